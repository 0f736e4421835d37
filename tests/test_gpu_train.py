"""GPU tests of the fit-loop level: fused trainer vs the oracle's fit-loop step on the reference's
goldens, the reference-style autograd loop with FusedAdam, device preproc through the dataset,
and the train.py shell end to end (artefacts, checkpoint interchange)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import nasa_niswan_amd as p
    p.load_library()
    return p


def test_fused_trainer_matches_reference_fit_loop_cfg0(pkg):
    """BASELINE configs[0]: 1-layer ConvLSTM, 32x32, 4 channels, seq_len 4, batch 2 -- three optimiser
    steps against the goldens produced by the reference model.py + torch.optim.Adam."""
    from nasa_niswan_amd.trainer import FusedTrainer
    g = np.load(os.path.join(GOLD, "cfg0_train.npz"))
    params = {k[len("params0."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("params0.")}
    net = pkg.ConvLSTM(4, [8], [3], 1).cuda()
    net.load_state_dict(params)
    tr = FusedTrainer(net, lr=float(g["lr"]), betas=tuple(g["betas"]), halo=(0, 0))
    X, y = torch.from_numpy(g["X"]).cuda(), torch.from_numpy(g["y"]).cuda()
    for step in (1, 2, 3):
        loss = float(tr.step(X, y))
        print(f"  step {step}: loss {loss:.7f} golden {float(g[f'loss{step}']):.7f}")
        assert abs(loss - float(g[f"loss{step}"])) < 2e-6
        if step == 1:
            for i, (k, p) in enumerate(net.named_parameters()):
                e = float((tr.flat.grad_view(i).cpu() - torch.from_numpy(g["grad." + k])).abs().max())
                assert e <= 1e-3 * float(np.abs(g["grad." + k]).max()) + 1e-7, (k, e)
    # weights after 3 Adam steps: lr=1e-4, so a last-bit sign flip of a ~0 gradient moves a weight by <= 2e-4 per step
    for k, v in net.state_dict().items():
        d = (v.cpu() - torch.from_numpy(g["params3." + k])).abs()
        print(f"  {k}: max |dW| {float(d.max()):.2e}, mean {float(d.mean()):.2e}")
        assert float(d.max()) <= 6.5e-4 and float(d.mean()) <= 2e-6
    loss_e, r2_e = tr.epoch_stats()
    assert abs(loss_e - np.mean([float(g[f"loss{s}"]) for s in (1, 2, 3)])) < 1e-5 and r2_e < 0.5


def test_epoch_stats_are_the_reference_per_batch_means(pkg):
    """f-1: the logged numbers are the reference's statistics -- the mean over batches of loss.item() and of
    sklearn r2_score(y, pred) (train.py:113-117), and for validation at batch size 1 the mean of per-sample R2
    (utils.py:73-75) -- not a pooled R2.  Checked (1) on the cfg0 goldens along three optimiser steps and (2) on a
    synthetic 8-sample epoch cut into ragged batches 3+3+2 and into 8 batches of 1, tolerance 1e-6."""
    from nasa_niswan_amd.trainer import FusedTrainer
    from oracle import convlstm_oracle as O
    g = np.load(os.path.join(GOLD, "cfg0_train.npz"))
    params = {k[len("params0."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("params0.")}
    net = pkg.ConvLSTM(4, [8], [3], 1).cuda()
    net.load_state_dict(params)
    lr, betas = float(g["lr"]), tuple(float(b) for b in g["betas"])
    tr = FusedTrainer(net, lr=lr, betas=betas, halo=(0, 0))
    X, y = torch.from_numpy(g["X"]), torch.from_numpy(g["y"])
    # the golden prediction of step 1 pins the oracle's R2 of that batch
    r2_gold = O.r2_score_np(g["y"], g["pred_full"][:, 0])
    p, st, r2s, losses = params, None, [], []
    for step in (1, 2, 3):
        tr.step(X.cuda(), y.cuda())
        p, st, oloss, opred, _ = O.train_step(p, st, X, y, lr=lr, betas=betas, halo=(0, 0))
        r2s.append(O.r2_score_np(y.numpy(), opred.numpy()))
        losses.append(oloss)
    assert abs(r2s[0] - r2_gold) < 1e-7
    loss_e, r2_e, r2_pool = tr.epoch_stats(pooled=True)
    print(f"  cfg0: mean per-batch R2 {r2_e:.7f} (oracle {np.mean(r2s):.7f}), pooled {r2_pool:.7f}; loss {loss_e:.7f}")
    assert abs(r2_e - np.mean(r2s)) < 1e-6 and abs(loss_e - np.mean(losses)) < 2e-6

    params = O.synth_params(5, [8, 8], [3, 3], 2, seed=12)
    X, y = O.synth_batch(8, 3, 5, 20, 28, (10, 18), seed=12)
    y = y * torch.linspace(0.5, 3.0, 8).view(8, 1, 1) + torch.linspace(-1, 1, 8).view(8, 1, 1)   # per-sample spread: pooled != mean
    net = pkg.ConvLSTM(5, [8, 8], [3, 3], 2).cuda()
    net.load_state_dict(params)
    tr = FusedTrainer(net, lr=1e-3, halo=(5, 5))
    po = O.crop_pred(O.convlstm_forward(X, params), (5, 5), (10, 18))[:, 0]
    for cuts in ([(0, 3), (3, 6), (6, 8)], [(i, i + 1) for i in range(8)]):
        tr.reset_stats()
        for a, b in cuts:
            tr.evaluate(X[a:b].cuda(), y[a:b].cuda())
        loss_e, r2_e, r2_pool = tr.epoch_stats(pooled=True)
        want_r2 = np.mean([O.r2_score_np(y[a:b].numpy(), po[a:b].numpy()) for a, b in cuts])
        want_loss = np.mean([float(O.loss_mse_l1(y[a:b], po[a:b])) for a, b in cuts])
        want_pool = O.r2_score_np(y.numpy(), po.numpy())
        print(f"  {len(cuts)} batches: R2 {r2_e:.7f} (oracle {want_r2:.7f}), pooled {r2_pool:.7f} (oracle {want_pool:.7f})")
        assert abs(r2_e - want_r2) < 1e-6 and abs(loss_e - want_loss) < 2e-6 * abs(want_loss) + 1e-7
        assert abs(r2_pool - want_pool) < 1e-6 and abs(want_pool - want_r2) > 1e-3      # the two statistics do differ here


def test_reference_style_autograd_loop_equals_fused_trainer(pkg):
    from nasa_niswan_amd.optim import FlatParams, FusedAdam
    from nasa_niswan_amd.trainer import FusedTrainer
    from oracle import convlstm_oracle as O
    params = O.synth_params(5, [16, 8, 8], [5, 3, 3], 3, seed=9)
    X, y = O.synth_batch(2, 3, 5, 20, 28, (10, 18), seed=9)
    Xd, yd = X.cuda(), y.cuda()
    a = pkg.ConvLSTM(5, [16, 8, 8], [5, 3, 3], 3).cuda(); a.load_state_dict(params)
    b = pkg.ConvLSTM(5, [16, 8, 8], [5, 3, 3], 3).cuda(); b.load_state_dict(params)
    tr = FusedTrainer(a, lr=1e-3, betas=(0.5, 0.999), halo=(5, 5))
    opt = FusedAdam(FlatParams(b), lr=1e-3, betas=(0.5, 0.999))
    l1, l2 = torch.nn.MSELoss(), torch.nn.L1Loss()
    for _ in range(3):
        la = float(tr.step(Xd, yd))
        pred = b(Xd)[:, :, 5:15, 5:23].squeeze()                 # the reference loop body, train.py:96-110
        loss = l1(yd, pred) + l2(yd, pred)
        opt.zero_grad(); loss.backward(); opt.step()
        assert abs(la - float(loss)) < 1e-6
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert float((va - vb).abs().max()) < 1e-6, k


def test_dataset_device_batch_matches_oracle_preproc(pkg):
    from nasa_niswan_amd.dataset import SyntheticE33OMA_CRNN
    from oracle import preproc_oracle as PO
    for levels, mode in ((1, "reference"), (3, "reflect")):
        ds = SyntheticE33OMA_CRNN("train", padding=(100, 154), in_channels=3 * levels + 2, sequence_length=5,
                                  levels=levels, n_steps=30, pad_mode=mode, device="cuda")
        X, y = ds.device_batch([0, 7])
        torch.cuda.synchronize()
        assert X.shape == (2, 5, 3 * levels + 2, 100, 154)       # dataset_config.ipynb:712 (X: (T,5,100,154))
        for b, idx in enumerate((0, 7)):
            (u, v, w, pr, src), yr = ds.window(idx)
            ref = PO.preproc_sample(u if levels > 1 else u[:, 0], v if levels > 1 else v[:, 0], w if levels > 1 else w[:, 0],
                                    pr, src, ds.X_mean, ds.X_std, (100, 154), mode)
            np.testing.assert_allclose(X[b].cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
            yref = (yr - ds.y_mean) / ds.y_std
            np.testing.assert_allclose(y[b].cpu().numpy(), yref[0] if levels == 1 else yref, rtol=1e-6, atol=1e-6)
        Xi, yi = ds[7]
        assert torch.equal(Xi, X[1]) and yi.shape == ((90, 144) if levels == 1 else (levels, 90, 144))


def test_from_arrays_dataset_on_device_matches_oracle_preproc(pkg):
    """`E33OMA90D_CRNN.from_arrays` (dataset.py:551-637 on caller-held arrays; page-locked staging, one asynchronous upload):
    the device batch equals the numpy restatement of the whole chain -- stack, z-score with the training-part statistics,
    window, cyclic-lon / lat pad (dataset.py:584-634) -- and the slab path writes the same values."""
    from nasa_niswan_amd.dataset import E33OMA90D_CRNN
    from nasa_niswan_amd.trainer import FusedTrainer
    from oracle import preproc_oracle as PO
    rng = np.random.default_rng(8)
    n, H, W, T = 64, 90, 144, 6
    arrs = [(rng.standard_normal((n, H, W)) * s + m).astype(np.float32) for m, s in ((0.2, 6.5), (0.3, 5.3), (0, 6e-5), (2.2, 7.3), (0.2, 2.6), (5.0, 57.0))]
    ds = E33OMA90D_CRNN.from_arrays(*arrs, period="train", padding=(100, 154), sequence_length=T, device="cuda")
    assert ds._device_arrays()["u"].is_cuda and len(ds) == 45
    idx = [0, 17, 44]
    X, y = ds.device_batch(idx)
    torch.cuda.synchronize()
    assert X.shape == (3, T, 5, 100, 154) and y.shape == (3, 90, 144)
    for b, i in enumerate(idx):
        ref = PO.preproc_sample(*(a[i:i + T] for a in arrs[:5]), ds.X_mean, ds.X_std, (100, 154), "reference")
        np.testing.assert_allclose(X[b].cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(y[b].cpu().numpy(), (arrs[5][i + T - 1] - ds.y_mean) / ds.y_std, rtol=1e-6, atol=1e-6)
    # and it trains: the slab path and the tensor path give the same first-step loss in f32 (same values, same kernels)
    losses = []
    for feed in ("slab", "tensor"):
        torch.manual_seed(0)
        net = pkg.ConvLSTM(5, [8, 8], [3, 3], 2).cuda()
        tr = FusedTrainer(net, lr=1e-3, halo=(5, 5))
        Xb, yb = ds.slab_batch(idx) if feed == "slab" else ds.device_batch(idx)
        losses.append(float(tr.step(Xb, yb)))
    assert losses[0] == losses[1], losses


@pytest.mark.parametrize("fold", [True, False], ids=["xfold", "plain"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_slab_batch_is_bit_identical_to_preproc_then_pack(pkg, dtype, fold):
    """a-6 straight into the slab: `nint_preproc_fuse_pad_slab` (one launch per batch, no f32 intermediate) must
    write exactly the bytes that the oracle's preproc followed by the pack kernel's rounding produces: compared
    (1) byte for byte with device_batch -> nint_pack_btchw[_xfold] on the same windows, whose f32 values are themselves
    checked against preproc_oracle above, and (2) directly against preproc_oracle rounded to the slab type.
    With `fold` the thin 5-channel input is written HORIZONTALLY FOLDED (nint_layer.xfold: slab channel kx*C + c of
    pixel x = channel c of pixel x + kx - 1, zero outside the padded grid); the 62-channel input never folds."""
    from nasa_niswan_amd import engine
    from nasa_niswan_amd.dataset import SyntheticE33OMA_CRNN
    from nasa_niswan_amd.engine import LayerCfg, SeqEngine
    from oracle import preproc_oracle as PO
    engine.XFOLD = fold
    try:
        for levels, mode, idx in ((1, "reference", [3, 0, 9]), (20, "reflect", [1, 5])):
            Cin, k = 3 * levels + 2, 3
            ds = SyntheticE33OMA_CRNN("train", padding=(100, 154), in_channels=Cin, sequence_length=3, levels=levels,
                                      n_steps=24, pad_mode=mode, device="cuda")
            eng = SeqEngine([LayerCfg(Cin, 16, k)], dtype, "cuda")
            folded = eng.cfgs[0].xfold
            assert folded == (fold and levels == 1)
            B = len(idx)
            ws_a = eng.acquire(B, 3, 100, 154, False, False)
            ws_b = eng.acquire(B, 3, 100, 154, False, False)
            assert ws_a is not ws_b
            sb, y1 = ds.slab_batch(idx)
            eng.pack_input(ws_a, sb)
            X, y2 = ds.device_batch(idx)
            eng.pack_input(ws_b, X)
            torch.cuda.synchronize()
            assert torch.equal(ws_a.xs, ws_b.xs) and torch.equal(y1, y2), (levels, mode)
            # directly against the oracle: image t*B+b, interior [P, P+100) x [P, P+154), channels-last
            g, Cp = ws_a.g, ws_a.Cxp0
            et = torch.float32 if dtype == "f32" else torch.bfloat16
            full = ws_a.xs.view(et).view(3 * B, g.Hh, g.Wh, Cp)
            slab = full[:, g.P:g.P + 100, g.P:g.P + 154].float().cpu()
            nch = Cin * k if folded else Cin
            assert float(full[:, :, :, nch:].abs().max()) == 0.0                     # channel padding zero
            for b, i in enumerate(idx):
                fields, _ = ds.window(i)
                ref = torch.from_numpy(PO.preproc_sample(*fields, ds.X_mean, ds.X_std, (100, 154), mode))   # (T,C,Hp,Wp)
                want = ref.to(et).float().permute(0, 2, 3, 1)                        # (T,Hp,Wp,C)
                if folded:     # channel kx*C + c of pixel x = channel c of pixel x + kx - k//2, zero outside
                    pad = torch.nn.functional.pad(want, (0, 0, k // 2, k // 2))
                    want = torch.cat([pad[:, :, kx:kx + 154] for kx in range(k)], dim=3)
                got = slab[b::B][:, :, :, :nch]
                if dtype == "f32":
                    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
                else:   # the f32 value may differ from numpy's in the last bit before rounding: <= 1 bf16 ulp, and rarely
                    d = (got - want).abs()
                    assert float((d > 0).float().mean()) < 1e-3 and float((d / (want.abs() + 1e-6)).max()) <= 2 ** -7
            eng.release(ws_a); eng.release(ws_b)
    finally:
        engine.XFOLD = True


def test_train_py_end_to_end(pkg, tmp_path, monkeypatch):
    from nasa_niswan_amd import train as T
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    snap = tmp_path / "snap"
    argv = ["--model", "LSTM-test", "--in-channels", "5", "--hidden-channels", "8", "8", "--kernel-size", "3", "3",
            "--num-layers", "2", "--sequence-length", "4", "--num-epochs", "10", "--input-size", "100", "154",
            "--batch-size", "4", "--learning-rate", "1e-3", "--scheduler-config", "4", "0.5", "--snapshot-dir", str(snap),
            "--synthetic-steps", "36", "--dtype", "f32"]
    logger = T.main(T.get_arguments(argv))
    assert len(logger["MSELoss"]) == 10 and all(np.isfinite(logger["MSELoss"]))
    assert logger["MSELoss"][-1] < logger["MSELoss"][0]                     # it learns
    with open(snap / "logger.npy", "rb") as f:                               # three stacked np.save (train.py:138-142)
        a, b, c = np.load(f), np.load(f), np.load(f)
    assert a.shape == b.shape == c.shape == (10,)
    ck = torch.load(snap / "epoch-010" / "generator.pth.tar", weights_only=True)
    assert set(ck) == {"model_state_dict", "optimizer_state_dict", "learning_rate", "epoch"} and ck["epoch"] == 10
    assert abs(ck["learning_rate"][0] - 1e-3 * 0.25) < 1e-12                 # StepLR(4, 0.5) after 10 epochs
    assert list(ck["model_state_dict"]) == ["layers.0.conv.weight", "layers.0.conv.bias", "layers.1.conv.weight",
                                            "layers.1.conv.bias", "conv.weight", "conv.bias"]
    # resume: weights + Adam moments restored, LR forced to the CLI value (utils.py:34-50)
    logger2 = T.main(T.get_arguments(argv[:-4] + ["--synthetic-steps", "36", "--dtype", "f32", "--use-checkpoint",
                                                   "--restore-from", str(snap / "epoch-010"), "--num-epochs", "1"]))
    assert logger2["MSELoss"][0] < logger["MSELoss"][0]


def test_train_py_runs_baseline_config0(pkg, tmp_path, monkeypatch):
    """BASELINE.json configs[0] through the fit-loop entry point itself: 1-layer ConvLSTM, 32x32 synthetic grid,
    4 in-channels, seq_len 4, batch 2 (on the MI355X: this build has no CPU leg).  The crop is derived from
    --input-size vs --grid (halo 0 here; the reference hard-codes 5, train.py:102).  The first step's loss must
    equal the oracle's fit-loop step on the same first batch, and both data paths (slab-direct / f32 tensor) agree."""
    from nasa_niswan_amd import train as T
    from nasa_niswan_amd.dataset import SyntheticE33OMA_CRNN
    from nasa_niswan_amd.utils import shard_indices
    from oracle import convlstm_oracle as O
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    argv = ["--model", "LSTM-cfg0", "--in-channels", "4", "--hidden-channels", "8", "--kernel-size", "3", "--num-layers", "1",
            "--sequence-length", "4", "--input-size", "32", "32", "--grid", "32", "32", "--batch-size", "2", "--num-epochs", "2",
            "--learning-rate", "1e-4", "--synthetic-steps", "24", "--dtype", "f32", "--snapshot-dir", str(tmp_path / "s")]
    logger = T.main(T.get_arguments(argv))
    assert len(logger["MSELoss"]) == 2 and np.isfinite(logger["MSELoss"]).all() and np.isfinite(logger["r2_score_val"]).all()
    ds = SyntheticE33OMA_CRNN("train", padding=(32, 32), in_channels=4, sequence_length=4, n_steps=24, grid=(32, 32), device="cuda")
    assert ds.generic and len(ds) == 17
    idx = shard_indices(len(ds), 1, 0, 1, 2)
    assert len(idx) == 9 and len(idx[-1]) == 1                       # ragged tail batch kept (DataLoader drop_last=False)
    X, y = ds.device_batch(idx[0])
    params = O.init_params(4, [8], [3], 1, seed=0)                    # seed(0) then ConvLSTM(...): train.py:32,48
    _, _, oloss, _, _ = O.train_step(params, None, X.cpu(), y.cpu(), lr=1e-4, halo=(0, 0))
    print(f"  first-step loss {logger['first_step_loss']:.7f}, oracle {oloss:.7f}")
    assert abs(logger["first_step_loss"] - oloss) <= 2e-6 * abs(oloss)
    logger2 = T.main(T.get_arguments(argv + ["--f32-inputs"]))
    assert logger2["first_step_loss"] == logger["first_step_loss"] and logger2["MSELoss"] == logger["MSELoss"]
    with pytest.raises(SystemExit):                                   # a grid larger than the model input is refused
        T.main(T.get_arguments(argv[:-2] + ["--snapshot-dir", str(tmp_path / "t"), "--grid", "40", "40"]))


def test_bench_under_torchrun_with_rccl_group(pkg):
    """The N>1 launch contract on a 1-GPU box: torch.distributed.run, one rank, RCCL process group
    forced on, so init / broadcast / all-reduce / barrier / MAX-reduce of the timing all execute."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--force-dist", "--no-cpu-baseline", "--long-steps", "4"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    # ONE line on stdout: RCCL's version banner ("RCCL version : ...", five lines at process-group creation) goes to stderr
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), out.stdout[:1500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "dp1" and d["scaling"] == "weak"
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    # the collective by itself: timed on the step's stream around dist.all_reduce of the flat bucket
    assert d["allreduce_ms"] > 0 and d["allreduce_bytes"] == 4 * 945428 and d["bus_bw_GBs"] is None      # one rank: no bus traffic
    assert d["value_200steps"] > 0 and d["long_window"]["steps"] == 4
    # kernels priced inside the step by the probe stamps, the warm loop beside them
    r = d["roofline"]
    assert r["timing"].startswith("in-step") and r["ms_per_launch"] > 0 and r["ms_per_launch_loop"] > 0
    ph = d["phases"]["per_step_us"]
    # B = 2 runs the forward wavefront as merged grids (nint_seq.wave = 4): 10 grids of all three layers, one of two at either
    # end, the lone gate launches of the first / last wavefront step -- and the headline roofline is the merged grid's
    assert d["config"]["wave"] == 4 and ph["fwd_wavefront_grid_of_3"]["launches"] == 10 and ph["fwd_wavefront_grid_of_2"]["launches"] == 2
    assert ph["gate0"]["launches"] == 1 and ph["gate2"]["launches"] == 1 and r["kernel"].startswith("conv_lstm_multi8_kernel")
    assert ph["wgrad0"]["launches"] == 1 and 0 < d["phases"]["probe_pair_cost_us"] < 50
    # ... and BPTT as two grids per step (kinds 8 / 9), the same schedule as the timed steps
    assert ph["bptt_grid_dgrad0_with_launch_of_layer1"]["launches"] >= 10 and ph["bptt_grid_pointwise0_with_fused_step_of_layer2"]["launches"] == 10
    assert d["phases"]["schedule_differs_from_timed_steps"] is False


@pytest.mark.parametrize("hidden", [[16, 8, 8], [64, 32, 16]])
def test_timing_probes_bracket_every_launch_and_change_nothing(pkg, hidden):
    """nint_seq.probe (how bench.py prices kernels inside the step): stamp launches around the selected launches of
    nint_seq_fwd / nint_seq_bwd.  They must not change a bit of the step's results, every (kind, layer, t) must appear as a
    begin / end pair in launch order with non-decreasing timestamps, and a step without probes must leave the buffer alone.
    The forward pass of this small batch is a wavefront of merged grids (nint_seq.wave = 5): those are bracketed as kind 7
    (layer = gate launches in the grid, t = wavefront step), the lone launches at either end as gate launches.  With the
    reference's widths (64, 32, 16: classic BPTT steps in layers 0 and 1) the BPTT pairs of wave = 5 are on as well -- the schedule,
    and with it every bit of the result, must be the same with and without probes; those grids are bracketed as kinds 8 and 9."""
    import bench
    from nasa_niswan_amd.trainer import FusedTrainer
    from oracle import convlstm_oracle as O
    C_, ks, B, T, H, W = 5, [5, 3, 3], 2, 4, 20, 28
    params = O.synth_params(C_, hidden, ks, 3, seed=3)
    X, y = O.synth_batch(B, T, C_, H, W, (10, 18), seed=3)
    res = {}
    for probed in (False, True):
        net = pkg.ConvLSTM(C_, hidden, ks, 3, compute_dtype="bf16").cuda()
        net.load_state_dict(params)
        tr = FusedTrainer(net, lr=1e-3, halo=(5, 5))
        buf = torch.zeros(2 * 1024, dtype=torch.int64, device="cuda")
        if probed:
            tr.set_probe(buf, 0x3fe)                 # every kind
        loss = float(tr.step(X.cuda(), y.cuda()))
        torch.cuda.synchronize()
        res[probed] = (loss, tr.flat.grad.clone(), tr.flat.data.clone(), buf.cpu().numpy())
    assert res[False][0] == res[True][0] and torch.equal(res[False][1], res[True][1]) and torch.equal(res[False][2], res[True][2])
    assert not res[False][3].any()
    w = res[True][3]
    fwd, bwd = bench.probe_table(w[:1024]), bench.probe_table(w[1024:])
    assert [r[:4] for r in fwd[:2]] == [(0, 0, 0, 0), (0, 0, 0, 1)] and [r[:4] for r in bwd[:2]] == [(0, 0, 0, 0), (0, 0, 0, 1)]
    exp = []
    for w_ in range(T + 3 - 1):
        ls = [l for l in range(3) if 0 <= w_ - l < T]
        exp += [(7, len(ls), w_, e) for e in (0, 1)] if len(ls) > 1 else [(1, ls[0], w_ - ls[0], e) for e in (0, 1)]
    assert [r[:4] for r in fwd[2:]] == exp
    for tab in (fwd, bwd):
        ticks = [r[4] for r in tab]
        assert all(b >= a for a, b in zip(ticks, ticks[1:])) and ticks[-1] > ticks[0]
    kinds = {r[0] for r in bwd[2:]}
    assert {2, 3, 5, 6} <= kinds                      # pointwise, dgrad, weight gradients, fold (the fused step: kind 4, per schedule)
    if hidden[1] == 32:
        assert 9 in kinds                             # the fused step + the bottom pointwise backward as one grid (the dgrad pair, kind 8,
                                                      # has no merged kernel for this tiny grid's launch shapes: test_bench_under_torchrun... sees it)
    d = bench.probe_durations(w[:1024], w[1024:])
    assert len(d[(1, 0)]) == 1 and len(d[(7, 3)]) == T - 2 and len(d[(7, 2)]) == 2 and all(us > 0 for _, us in d[(7, 3)]) and len(d[(5, 0)]) == 1


def test_bench_starts_its_own_ranks(pkg):
    """`python bench.py --gpus 1 --force-dist` is what the driver runs for N = 1; for N > 1 the same form must start N
    ranks by itself.  On this one-GPU box: the parent decides from argv, the torchrun child runs the single rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    argv = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "1", "--force-dist", "--no-cpu-baseline",
            "--no-kernel-rooflines", "--long-steps", "0", "--master-port", "29579"]
    args = bench.parse_args(argv)
    code = ("import sys; sys.path.insert(0, %r); import bench; a = %r; "
            "raise SystemExit(bench.spawn_ranks(bench.parse_args(a), a))") % (root, argv)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["allreduce_ms"] > 0 and d["roofline"] is None


def test_inference_helpers_match_oracle(pkg):
    """test.ipynb cells 8 and 56 on the HIP forward path vs the oracle on the same (perturbed) inputs."""
    from nasa_niswan_amd.dataset import SyntheticE33OMA_CRNN
    from nasa_niswan_amd.inference import oat_sensitivity, predict
    from oracle import convlstm_oracle as O
    ds = SyntheticE33OMA_CRNN("test", padding=(100, 154), in_channels=5, sequence_length=3, n_steps=24, device="cuda")
    params = O.synth_params(5, [8, 8], [3, 3], 2, seed=4)
    net = pkg.ConvLSTM(5, [8, 8], [3, 3], 2).cuda()
    net.load_state_dict(params)
    idx = [0, 1, 2]
    gts, pds = predict(net, ds, batch_size=2, indices=idx)
    assert gts.shape == pds.shape == (3, 1, 90, 144)
    X, y = ds.device_batch(idx)
    ref = O.convlstm_forward(X.cpu(), params)[:, :, 5:95, 5:149].numpy() * ds.y_std + ds.y_mean
    np.testing.assert_allclose(pds, ref, rtol=1e-4, atol=1e-3 * float(np.abs(ref).max()))
    np.testing.assert_allclose(gts[:, 0], ds.yraw[ds.first[idx] + 2][:, 0], rtol=1e-4, atol=1e-3)   # de-normalised target = raw field
    sweep = oat_sensitivity(net, ds, num_ftrs=5, batch_size=3, indices=idx)
    assert sweep.shape == (5, 3, 1, 90, 144)
    Xp = X.cpu().clone(); Xp[:, :, 3] *= 1.05
    refp = O.convlstm_forward(Xp, params)[:, :, 5:95, 5:149].numpy() * ds.y_std + ds.y_mean
    np.testing.assert_allclose(sweep[3], refp, rtol=1e-4, atol=1e-3 * float(np.abs(refp).max()))
    assert not np.allclose(sweep[3], pds)


def test_bf16_tracks_f32_along_a_training_trajectory(pkg):
    """bf16 storage / f32 accumulate / f32 master weights must agree with the f32 path
    at every point of a real training trajectory.  Free-running the two optimisers diverges chaotically
    (Adam normalises every update to ~lr), so the bf16 model is re-synchronised to the f32 weights before
    each of 20 steps and compared there: loss within 1 %, gradient cosine >= 0.995."""
    from nasa_niswan_amd.trainer import FusedTrainer
    from oracle import convlstm_oracle as O
    params = O.synth_params(5, [16, 8, 8], [5, 3, 3], 3, seed=21)
    X, y = O.synth_batch(4, 6, 5, 30, 38, (20, 28), seed=21)
    Xd, yd = X.cuda(), y.cuda()
    ref = pkg.ConvLSTM(5, [16, 8, 8], [5, 3, 3], 3, compute_dtype="f32").cuda()
    ref.load_state_dict(params)
    tr = FusedTrainer(ref, lr=2e-3, betas=(0.5, 0.999), halo=(5, 5))
    low = pkg.ConvLSTM(5, [16, 8, 8], [5, 3, 3], 3, compute_dtype="bf16").cuda()
    worst_loss, worst_cos, first, last = 0.0, 1.0, None, None
    for step in range(20):
        low.load_state_dict(ref.state_dict())
        low.zero_grad()
        pred = low(Xd)[:, :, 5:25, 5:33].squeeze()
        loss_b = ((yd - pred) ** 2).mean() + (yd - pred).abs().mean()
        loss_b.backward()
        gb = torch.cat([p.grad.reshape(-1) for p in low.parameters()])
        loss_f = float(tr.step(Xd, yd))                     # flat.grad now holds the f32 gradient at the same weights
        gf = tr.flat.grad
        cos = float((gb * gf).sum() / (gb.norm() * gf.norm()))
        worst_loss = max(worst_loss, abs(float(loss_b) - loss_f) / abs(loss_f))
        worst_cos = min(worst_cos, cos)
        first = loss_f if first is None else first
        last = loss_f
    print(f"  20 steps: loss {first:.4f} -> {last:.4f}; worst |dloss|/loss {worst_loss:.2e}; worst grad cosine {worst_cos:.5f}")
    assert worst_loss <= 1e-2 and worst_cos >= 0.995 and last < 0.97 * first


@pytest.mark.parametrize("rows", [4, 8])
def test_both_tile_heights_keep_parity_on_every_shape(pkg, rows):
    """The gate / dgrad kernels pick 4- or 8-row pixel tiles per launch shape.  The choice is an explicit field of the
    layer descriptor (nint_layer.tile_rows; the library reads no environment and keeps no state): both heights are
    forced here for every layer of ragged, multi-layer and reference-size shapes and checked against the oracle."""
    from nasa_niswan_amd import engine
    from test_gpu_shapes import CASES, check, run_case
    engine.FORCE_TILE_ROWS = rows
    try:
        for name in ("ragged-grid-odd-channels", "k1-and-k3", "wide-hidden-48"):
            for dtype in ("f32", "bf16"):
                check(run_case(pkg, *CASES[name], dtype), dtype)
        check(run_case(pkg, 5, [64, 32, 16], [5, 3, 3], 1, 2, 2, 20, 36, "f32"), "f32")      # the reference stack (folded x)
        check(run_case(pkg, 62, [64, 32, 16], [5, 3, 3], 20, 1, 2, 100, 154, "bf16"), "bf16")  # bench geometry, T=2
    finally:
        engine.FORCE_TILE_ROWS = 0


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_thin_inputs_plain_layout_keeps_parity(pkg, dtype):
    """Thin first-layer inputs are fed horizontally folded by default (nint_layer.xfold, covered by every small-channel
    case of the suite and the reference-size goldens); the plain channel-padded layout stays available
    (engine.XFOLD = False) and must give the same results: forward, all weight gradients and the input gradient."""
    from nasa_niswan_amd import engine
    from test_gpu_shapes import check, run_case
    for fold in (False, True):
        engine.XFOLD = fold
        try:
            net = pkg.ConvLSTM(5, [16], [5], 1, compute_dtype=dtype).cuda()
            assert net._engine(torch.device("cuda", 0)).cfgs[0].xfold == fold
            check(run_case(pkg, 5, [64, 32, 16], [5, 3, 3], 1, 2, 3, 20, 36, dtype), dtype)   # reference stack, 5 inputs
            check(run_case(pkg, 4, [8], [3], 1, 2, 4, 32, 32, dtype), dtype)                  # BASELINE configs[0]
            check(run_case(pkg, 7, [24, 16], [3, 5], 1, 3, 2, 11, 19, dtype), dtype)          # ragged grid, odd channels
        finally:
            engine.XFOLD = True


def _ddp_rank(rank, world, port, out, overlap=False):
    """One rank of test_two_ranks_on_one_device_match_the_global_batch: gloo process group (it moves the device bucket through
    the host -- RCCL needs one device per rank), both ranks on cuda:0."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import nasa_niswan_amd as p
    from nasa_niswan_amd.trainer import FusedTrainer
    from nasa_niswan_amd.utils import shard_indices
    torch.manual_seed(100 + rank)                     # deliberately different init per rank: the trainer broadcasts rank 0's
    net = p.ConvLSTM(6, [16, 8], [3, 3], 2, out_channels=2, compute_dtype="f32").cuda()
    tr = FusedTrainer(net, lr=1e-2, halo=(2, 2), overlap_allreduce=overlap)
    g = torch.Generator().manual_seed(7)
    X = torch.randn(4, 3, 6, 20, 28, generator=g)
    y = torch.randn(4, 2, 16, 24, generator=g)
    losses = []
    for step in range(3):
        idx = shard_indices(4, step, rank, world, 2, shuffle=False)[0]
        losses.append(float(tr.step(X[idx].cuda(), y[idx].cuda())))
    torch.cuda.synchronize()
    torch.save({"w": tr.flat.data.detach().cpu(), "losses": losses}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_device_match_the_global_batch(pkg, tmp_path):
    """The N > 1 step on the HIP path (FusedTrainer: bucket broadcast, ONE all-reduce of the flat gradient bucket, 1/world in
    nint_adam_flat), two ranks sharing this box's one device: after three steps both ranks hold identical weights, and they
    are the weights of a single process that trained on the global batch (mean of equal shards' gradients = global gradient)."""
    import socket
    import torch.multiprocessing as mp
    from nasa_niswan_amd.trainer import FusedTrainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "ddp")
    mp.spawn(_ddp_rank, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["w"], r1["w"])
    # single process, global batch, rank 0's initial weights (seed 100)
    torch.manual_seed(100)
    net = pkg.ConvLSTM(6, [16, 8], [3, 3], 2, out_channels=2, compute_dtype="f32").cuda()
    tr = FusedTrainer(net, lr=1e-2, halo=(2, 2))
    g = torch.Generator().manual_seed(7)
    X = torch.randn(4, 3, 6, 20, 28, generator=g)
    y = torch.randn(4, 2, 16, 24, generator=g)
    losses = [float(tr.step(X.cuda(), y.cuda())) for _ in range(3)]
    w = tr.flat.data.detach().cpu()
    err = float((w - r0["w"]).abs().max() / w.abs().max())
    print(f"  two ranks vs global batch: weights rel max err {err:.2e}; losses {losses} vs mean of shards "
          f"{[(a + b) / 2 for a, b in zip(r0['losses'], r1['losses'])]}")
    assert err < 2e-5
    for a, b, c in zip(losses, r0["losses"], r1["losses"]):
        assert abs(a - (b + c) / 2) < 1e-5 * abs(a)


def test_two_ranks_with_the_exchange_in_two_pieces_end_on_the_same_bits(pkg, tmp_path):
    """FusedTrainer(overlap_allreduce=True): BPTT + the gradients of layers >= 1, then the all-reduce of everything but layer 0's
    slice of the bucket STARTED (async), layer 0's weight gradient, the all-reduce of its slice (nint_seq.bwd_parts; SURVEY.md
    8e).  Same launches per layer, two all-reduces of disjoint slices: after three steps the weights of both ranks equal the
    one-all-reduce run's bit for bit."""
    import socket
    import torch.multiprocessing as mp
    res = {}
    for overlap in (False, True):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        out = str(tmp_path / f"ddp{int(overlap)}")
        mp.spawn(_ddp_rank, args=(2, port, out, overlap), nprocs=2, join=True)
        res[overlap] = (torch.load(out + ".0"), torch.load(out + ".1"))
    assert torch.equal(res[True][0]["w"], res[True][1]["w"])
    assert torch.equal(res[True][0]["w"], res[False][0]["w"])
    assert res[True][0]["losses"] == res[False][0]["losses"] and res[True][1]["losses"] == res[False][1]["losses"]
