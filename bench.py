#!/usr/bin/env python3
"""Headline benchmark: training samples/sec of the Smart-NINT ConvLSTM on synthetic 90x144x20
grids at seq_len=12 (BASELINE.json metric / configs[1]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch resident in HBM: pack -> ConvLSTM forward
(12 steps x 3 layers) -> 1x1 head -> crop+MSE+L1 -> BPTT (pointwise, dgrad, wgrad) -> [RCCL
all-reduce of the flat gradient bucket] -> Adam.  Rank 0 prints ONE JSON line.

Workload (config.workload = "cfg1-20level"): ConvLSTM(62, (64,32,16), (5,3,3)), head out 20,
X (B,12,62,100,154) = 3 met fields x 20 levels + precipitation + emission on the 90x144 GISS grid
with the reference's 5-cell halo (launcher.sh:24), y (B,20,90,144); bf16 storage, f32 accumulate,
f32 master weights; B = 8 per GPU (launcher.sh:25), weak scaling across GPUs.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (C_in, hidden, kernels, out_channels, T, Hp, Wp, halo, grid)
    "cfg1-20level": (62, (64, 32, 16), (5, 3, 3), 20, 12, 100, 154, (5, 5), (90, 144)),
    "cfg1-refpinned": (5, (64, 32, 16), (5, 3, 3), 1, 12, 100, 154, (5, 5), (90, 144)),
    # BASELINE.json configs[3] / configs[4]: parity-test shapes, runnable here for sizing (not bench lines)
    "cfg3-1deg-hidden128": (62, (128, 128, 128), (3, 3, 3), 20, 24, 190, 298, (5, 5), (180, 288)),
    "cfg4-multitracer-40lev": (126, (64, 32, 16), (5, 3, 3), 200, 12, 100, 154, (5, 5), (90, 144)),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # dense peaks, MI355X_MICROARCH.md


def fwd_flops_per_sample(C, hidden, ks, out, T, Hp, Wp):
    """F_fwd = 2*Hp*Wp*T*sum_l k^2 (Cin+Ch) 4Ch (+ head)."""
    f, cin = 0, C
    for ch, k in zip(hidden, ks):
        f += k * k * (cin + ch) * 4 * ch
        cin = ch
    return 2 * Hp * Wp * (T * f + hidden[-1] * out)


def time_kernel(fn, iters, stream):
    """average device time of `fn` (one launch sequence on `stream`) with HIP events, in ms"""
    fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream)
    for _ in range(iters):
        fn()
    ev1.record(stream)
    ev1.synchronize()
    return ev0.elapsed_time(ev1) / iters


def host_cpu():
    """(model name, physical core count) of this box from /proc/cpuinfo"""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core)); phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return model, (len(cores) or os.cpu_count() or 1)


def cpu_baseline(name, steps=3):
    """The CPU oracle (plain PyTorch CPU ops = the ATen path the reference takes on CPU) timed on
    this box's host cores on a bounded sample of the same workload: B=2, f32, 1 warm-up + `steps`
    timed train steps.  Reported, never the target."""
    from oracle import convlstm_oracle as O      # checker / baseline only -- never on the product path
    C, hidden, ks, out, T, Hp, Wp, halo, grid = WORKLOADS[name]
    B = 2
    model, phys = host_cpu()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(phys, avail))           # one thread per physical core this process may use
    torch.set_num_threads(threads)
    params = O.synth_params(C, hidden, ks, len(hidden), out_channels=out, seed=0)
    import numpy as np
    rng = np.random.default_rng(0)
    X = torch.from_numpy(rng.standard_normal((B, T, C, Hp, Wp)).astype("float32"))
    y = torch.from_numpy(rng.standard_normal((B, out, grid[0], grid[1])).astype("float32"))
    state = None
    params, state, *_ = O.train_step(params, state, X, y, lr=1e-3, halo=halo)
    t0 = time.perf_counter()
    for _ in range(steps):
        params, state, *_ = O.train_step(params, state, X, y, lr=1e-3, halo=halo)
    dt = time.perf_counter() - t0
    return {"value": round(B * steps / dt, 4), "unit": "samples/s", "cores": threads, "kind": "port",
            "cpu_model": model, "physical_cores": phys, "logical_cpus_available": avail,
            "sample": f"{name} at B={B}, f32, {steps} timed train steps after 1 warm-up ({dt:.1f} s of CPU work)"}


def executed_flops_per_sample(C, hidden, ks, out, T, Hp, Wp):
    """FLOPs a training step really has to execute (algorithmic channel counts, no padding): the zero initial state
    removes the h half of K at t = 0 from the forward, its dgrad and its weight gradient (model.py:259-262), and the
    input gradient of layer 0 is never needed (train.py:109 only differentiates the parameters)."""
    px = 2 * Hp * Wp
    fwd = dgrad = wgrad = 0
    cin = C
    for l, (ch, k) in enumerate(zip(hidden, ks)):
        per = px * k * k * 4 * ch
        fwd += per * (T * cin + (T - 1) * ch)
        wgrad += per * (T * cin + (T - 1) * ch)
        dgrad += per * ((T * cin if l > 0 else 0) + (T - 1) * ch)
        cin = ch
    head = px * hidden[-1] * out
    return fwd + dgrad + wgrad + 3 * head


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--workload", default="cfg1-20level", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fuse-bwd", type=lambda v: int(v, 0), default=0,
                    help="nint_seq.fuse_bwd (BPTT schedule; 0 = the library's per-layer choice, 0x40000000|masks = explicit, see nint.h)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL even for one rank (exercises the N>1 code path on a 1-GPU box)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL on ROCm
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    import nasa_niswan_amd as pkg
    from nasa_niswan_amd.trainer import FusedTrainer
    from nasa_niswan_amd import engine as _engine
    _engine.FUSE_BWD = args.fuse_bwd
    pkg.load_library()

    C, hidden, ks, out, T, Hp, Wp, halo, grid = WORKLOADS[args.workload]
    B = args.batch
    torch.manual_seed(0)                                    # identical init on every rank (utils.py:77-88)
    model = pkg.ConvLSTM(C, list(hidden), list(ks), len(hidden), out_channels=out, compute_dtype=args.dtype).to(dev)
    trainer = FusedTrainer(model, lr=1e-3, betas=(0.5, 0.999), halo=halo, distributed=(world > 1 or args.force_dist))
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)   # each rank its own shard of synthetic data
    X = torch.randn(B, T, C, Hp, Wp, device=dev, generator=gen)
    y = torch.randn(B, out, grid[0], grid[1], device=dev, generator=gen)

    def sync():
        if world > 1 or args.force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(X, y)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(X, y)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1 or args.force_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss)

    # ---- rooflines of the four kernels that are ~3/4 of the step, each timed live with HIP events on the launch
    # stream (back-to-back launches on the bench's own slabs, i.e. real data): layer-0 gate kernel (forward), layer-0
    # weight gradient, layer-0 dgrad, layer-0 LSTM pointwise backward.  `roofline` = the dominant one.
    roof, roof_all = None, None
    if rank == 0:
        import ctypes as Ct
        eng = model._engine(dev)
        ws = eng.acquire(B, T, Hp, Wp, True, False)
        lib = pkg.load_library()
        ly, g = eng.layers[0], ws.g
        st = torch.cuda.current_stream()
        sp = Ct.c_void_p(st.cuda_stream)
        kc, es = eng.kc, eng.es
        halo_px, comp_px = g.Hh * g.Wh, Hp * Wp
        k0, ch0 = ks[0], hidden[0]
        hs, cs = B * halo_px * ly.Chp * es, B * comp_px * ly.Chp * 4
        gs, dgs = B * comp_px * 4 * ly.Ch16 * es, B * halo_px * 4 * ly.Ch16 * es
        xs1 = ws.xs.data_ptr() + 1 * B * halo_px * ly.Cxp * es
        vp = Ct.c_void_p
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        traffic = {}
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except (OSError, ValueError):
            pass

        def entry(key, kernel, bound, work, ms, note=None):
            """work = algorithmic FLOPs (mfma) or bytes (hbm) per launch"""
            if bound == "mfma":
                ach, pk, unit = work / (ms * 1e-3) / 1e12, peak, "TFLOP/s"
            else:
                ach, pk, unit = work / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
            # HBM bytes per launch from the committed PMC pass (PMC needs rocprofv3 around the process, so it is
            # not collected live); only quoted for the exact configuration it was measured on
            tr = traffic.get(f"{args.workload}/{args.dtype}/B{B}/{key}", {}).get("bytes_per_launch")
            e = {"kernel": kernel, "bound": bound, "achieved": round(ach, 2), "peak": pk, "unit": unit,
                 "frac": round(ach / pk, 4), "traffic": tr, "ms_per_launch": round(ms, 4),
                 ("flops_per_launch" if bound == "mfma" else "bytes_per_launch"): work}
            if note:
                e["note"] = note
            return e

        # t = 1 step of layer 0: x slab image B.., h_prev = slab 1, c_prev = slot 1 -> writes slot 2
        def k_fwd():
            assert lib.nint_cell_fwd(Ct.byref(ly), Ct.byref(g), eng.dt, B, vp(xs1), vp(ws.h[0].data_ptr() + hs),
                                     vp(ws.c[0].data_ptr() + cs), vp(ws.h[0].data_ptr() + 2 * hs), vp(ws.c[0].data_ptr() + 2 * cs),
                                     vp(ws.gates[0].data_ptr() + gs), sp) == 0

        def k_dgrad():      # as in the step: h columns only (no input gradient for layer 0)
            assert lib.nint_conv_dgrad(Ct.byref(ly), Ct.byref(g), eng.dt, B, vp(ws.dG[0].data_ptr() + dgs), None,
                                       vp(ws.dh[0].data_ptr()), sp) == 0

        def k_pw():
            assert lib.nint_cell_bwd_pointwise(Ct.byref(ly), Ct.byref(g), eng.dt, B, vp(ws.gates[0].data_ptr() + gs),
                                               vp(ws.c[0].data_ptr() + cs), vp(ws.c[0].data_ptr() + 2 * cs), vp(ws.dh[0].data_ptr()),
                                               vp(ws.dc[0].data_ptr()), vp(ws.dG[0].data_ptr() + dgs), sp) == 0

        dW0, db0 = trainer._dW[0], trainer._db[0]

        def k_wgrad():      # both sources (x and h) over all T time steps + the split-K fold + the bias-gradient fold
            assert lib.nint_conv_wgrad(Ct.byref(ly), Ct.byref(g), eng.dt, T * B, vp(ws.dG[0].data_ptr()), vp(ws.xs.data_ptr()),
                                       vp(ws.h[0].data_ptr()), vp(dW0.data_ptr()), vp(db0.data_ptr()), vp(eng.wg_partial.data_ptr()),
                                       eng.wg_partial.numel() * 4, eng.n_cu, sp) == 0

        f_gate = 2.0 * B * Hp * Wp * k0 * k0 * (C + ch0) * 4 * ch0           # algorithmic, per launch
        f_dgrad = 2.0 * B * Hp * Wp * k0 * k0 * 4 * ch0 * ch0
        f_wgrad = 2.0 * T * B * Hp * Wp * k0 * k0 * (C + ch0) * 4 * ch0
        b_pw = float(B * comp_px * ch0 * (9 * es + 16))     # gates + dG (4 ET each) + dh (ET); c_prev, c_new, dc in, dc out (f32)
        roof = entry("conv_igemm_fwd_layer0", "conv_igemm_kernel<LSTM epilogue> layer 0 (B images, one time step)", "mfma",
                     f_gate, time_kernel(k_fwd, 50, st))
        roof_all = [
            dict(roof, launches_per_step=T),
            dict(entry("wgrad_layer0", "wgrad_kernel layer 0: x part + h part over all T steps, incl. split-K and bias folds",
                       "mfma", f_wgrad, time_kernel(k_wgrad, 10, st)), launches_per_step=1),
            dict(entry("conv_igemm_dgrad_layer0", "conv_igemm_kernel<DGRAD epilogue> layer 0 (h columns; B images, one time step)",
                       "mfma", f_dgrad, time_kernel(k_dgrad, 50, st)), launches_per_step=T - 1),
            dict(entry("lstm_bwd_pointwise_layer0", "lstm_bwd_pointwise_kernel layer 0 (B images, one time step)", "hbm",
                       b_pw, time_kernel(k_pw, 50, st)), launches_per_step=T),
        ]
        eng.release(ws)

    samples = world * B * args.steps
    value = samples / elapsed
    if rank == 0:
        f_train = 3 * fwd_flops_per_sample(C, hidden, ks, out, T, Hp, Wp)
        f_exec = executed_flops_per_sample(C, hidden, ks, out, T, Hp, Wp)
        line = {
            "metric": "training samples/sec (90x144x20 grid, seq_len=12)",
            "value": round(value, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": args.workload, "in_channels": C, "hidden": list(hidden), "kernels": list(ks),
                       "out_channels": out, "seq_len": T, "padded_grid": [Hp, Wp], "grid": list(grid),
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}"},
            # whole step priced two ways: the survey's nominal 3 x F_fwd, and the FLOPs the step really executes
            # (zero-state h parts and the layer-0 input gradient are legitimately skipped)
            "whole_step_nominal_tflops": round(value * f_train / 1e12, 2),
            "whole_step_nominal_mfma_frac": round(value * f_train / 1e12 / world / MFMA_PEAK_TFLOPS[args.dtype], 4),
            "whole_step_executed_tflops": round(value * f_exec / 1e12, 2),
            "whole_step_executed_mfma_frac": round(value * f_exec / 1e12 / world / MFMA_PEAK_TFLOPS[args.dtype], 4),
            "final_loss": round(final_loss, 5),
            "roofline": roof,
            "roofline_kernels": roof_all,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.workload)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1 or args.force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
