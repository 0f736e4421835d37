import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/oracle") else os.environ.get("GRAFT_REPO_ROOT", "."))
import nasa_niswan_amd as pkg
from nasa_niswan_amd import engine
from oracle import convlstm_oracle as O
pkg.load_library()
def run(C, hidden, ks, out, B, T, H, W, dtype="f32", wave=0, rows=0, seed=4):
    engine.FORCE_WAVE, engine.FORCE_TILE_ROWS = wave, rows
    L = len(hidden)
    rng = np.random.default_rng(seed)
    params = O.synth_params(C, hidden, ks, L, out_channels=out, seed=seed)
    X = torch.from_numpy(rng.standard_normal((B, T, C, H, W)).astype(np.float32))
    net = pkg.ConvLSTM(C, hidden, ks, L, out_channels=out, compute_dtype=dtype).cuda()
    net.load_state_dict(params)
    with torch.no_grad():
        pred = net(X.cuda()).cpu()
        po = O.convlstm_forward(X, params)
    e = float((pred - po).abs().max() / po.abs().max())
    print(f"C={C} hidden={hidden} k={ks} B={B} T={T} {H}x{W} wave={wave} rows={rows}: pred max-rel {e:.2e}", flush=True)
run(1, [32, 64, 16], [5, 3, 1], 1, 4, 2, 8, 49)
run(1, [32, 64, 16], [5, 3, 1], 1, 4, 2, 8, 49, rows=4)
run(1, [32, 64, 16], [5, 3, 1], 1, 4, 2, 8, 49, rows=8)
run(1, [32, 64, 16], [5, 3, 3], 1, 4, 2, 8, 49)
run(3, [32, 64, 16], [5, 3, 1], 1, 4, 2, 8, 49)
run(1, [32, 64], [5, 3], 1, 4, 2, 8, 49)
run(1, [32], [5], 1, 4, 2, 8, 49)
run(1, [32, 64, 16], [5, 3, 1], 1, 4, 1, 8, 49)
run(1, [32, 64, 16], [5, 3, 1], 1, 1, 2, 8, 49)
run(1, [32, 64, 16], [5, 3, 1], 1, 4, 2, 16, 48)
