#!/usr/bin/env python3
"""Headline benchmark: training samples/sec of the Smart-NINT ConvLSTM on synthetic 90x144x20
grids at seq_len=12 (BASELINE.json metric / configs[1]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 [--scaling strong]        # starts its own 8 ranks (torch.distributed.run as a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch resident in HBM: pack -> ConvLSTM forward
(12 steps x 3 layers) -> 1x1 head -> crop+MSE+L1 -> BPTT (pointwise, dgrad, wgrad) -> [RCCL
all-reduce of the flat gradient bucket] -> Adam.  Rank 0 prints ONE JSON line.

Workload (config.workload = "cfg1-20level"): ConvLSTM(62, (64,32,16), (5,3,3)), head out 20,
X (B,12,62,100,154) = 3 met fields x 20 levels + precipitation + emission on the 90x144 GISS grid
with the reference's 5-cell halo (launcher.sh:24), y (B,20,90,144); bf16 storage, f32 accumulate,
f32 master weights; B = 8 per GPU (launcher.sh:25), weak scaling across GPUs (--scaling strong: global batch 8).

Kernel rooflines are priced INSIDE the step: after the timed region a few more steps run with the library's timing
probes on (nint_seq.probe: one-thread stamp launches around every priced launch, calibrated by a back-to-back pair),
so `roofline` / `roofline_kernels` carry the duration a launch has between its real neighbours (`ms_per_launch`);
the old figure -- HIP events around a loop of back-to-back launches of the one kernel on warm slabs -- stays as
`ms_per_launch_loop`.  Whole phases (forward, BPTT chain + weight gradients) are bracketed with HIP events in the
same pass (`phases_ms`).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

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
XGMI_LINK_GBS = 153.0          # per-link, per direction (task statement: 7 links x ~153 GB/s per GPU)
PROBE_GATE, PROBE_POINTWISE, PROBE_DGRAD, PROBE_FUSED, PROBE_WGRAD, PROBE_FOLD, PROBE_WAVE, PROBE_BWD_PAIR, PROBE_BWD_PW = 1, 2, 3, 4, 5, 6, 7, 8, 9   # include/nint.h NINT_PROBE_*


def fwd_flops_per_sample(C, hidden, ks, out, T, Hp, Wp):
    """F_fwd = 2*Hp*Wp*T*sum_l k^2 (Cin+Ch) 4Ch (+ head)."""
    f, cin = 0, C
    for ch, k in zip(hidden, ks):
        f += k * k * (cin + ch) * 4 * ch
        cin = ch
    return 2 * Hp * Wp * (T * f + hidden[-1] * out)


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


def time_kernel(fn, iters, stream):
    """average device time of `fn` (one launch sequence on `stream`) with HIP events, in ms"""
    import torch
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


def cpu_baseline(name, steps=5):
    """The CPU oracle (plain PyTorch CPU ops = the ATen path the reference takes on CPU) timed on
    this box's host cores on a bounded sample of the same workload: B=2, f32, 1 warm-up + `steps`
    timed train steps, each timed by itself; the value is B / MEDIAN step time (a mean over 3 steps on 128 threads
    was 2.5x apart between two boxes of the same CPU model).  Reported, never the target."""
    import numpy as np
    import torch
    from oracle import convlstm_oracle as O      # checker / baseline only -- never on the product path
    C, hidden, ks, out, T, Hp, Wp, halo, grid = WORKLOADS[name]
    B = 2
    model, phys = host_cpu()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(phys, avail))           # one thread per physical core this process may use
    torch.set_num_threads(threads)
    params = O.synth_params(C, hidden, ks, len(hidden), out_channels=out, seed=0)
    rng = np.random.default_rng(0)
    X = torch.from_numpy(rng.standard_normal((B, T, C, Hp, Wp)).astype("float32"))
    y = torch.from_numpy(rng.standard_normal((B, out, grid[0], grid[1])).astype("float32"))
    state = None
    params, state, *_ = O.train_step(params, state, X, y, lr=1e-3, halo=halo)
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        params, state, *_ = O.train_step(params, state, X, y, lr=1e-3, halo=halo)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": round(B / med, 4), "unit": "samples/s", "cores": threads, "kind": "port",
            "cpu_model": model, "physical_cores": phys, "logical_cpus_available": avail,
            "step_s": [round(t, 3) for t in times],
            "sample": f"{name} at B={B}, f32, median of {steps} separately timed train steps after 1 warm-up "
                      f"({sum(times):.1f} s of CPU work)"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU (weak scaling) / global batch (strong scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch samples per GPU; strong: --batch samples in all, --batch / N per GPU (SURVEY 8d cfg 2)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--workload", default="cfg1-20level", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-rooflines", action="store_true", help="skip the probe pass and the per-kernel loops")
    ap.add_argument("--long-steps", type=int, default=200, help="steps of the extra long window (value_200steps); 0 = skip")
    ap.add_argument("--fuse-bwd", type=lambda v: int(v, 0), default=0,
                    help="nint_seq.fuse_bwd (BPTT schedule; 0 = the library's per-layer choice, 0x40000000|masks = explicit, see nint.h)")
    ap.add_argument("--wide", type=int, default=0, help="nint_layer.wide of every layer: weight-gradient kernel family (0 = the library's choice, 1 = 4-wave 64-column kernel, 2 = 8-wave 128-column kernel where instantiated)")
    ap.add_argument("--wave", type=int, default=-1, choices=[-1, 0, 1, 2, 3, 4, 5],
                    help="merged grids (nint_seq.wave): -1 = the engine's rule by batch size, 0 = off, 1 = forward wavefront + backward pair, 2 = forward wavefront only (8-row tiles), 4 = 2 + the BPTT pairs, 5 = the forward pass of 1 + the BPTT pairs of 4")
    ap.add_argument("--tile-rows", type=int, default=0, choices=[0, 1, 2, 4, 8], help="nint_layer.tile_rows of every layer (0 = per launch shape; tiny layers: 1 = stencil gate kernel, 2 = dense-K MFMA gate kernel)")
    ap.add_argument("--overlap-allreduce", action="store_true", help="reduce the gradient bucket in two pieces, all but layer 0's slice under layer 0's weight gradient (FusedTrainer(overlap_allreduce=True))")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL even for one rank (exercises the N>1 code path on a 1-GPU box)")
    ap.add_argument("--master-port", type=int, default=29533)
    ap.add_argument("--phase-events", type=int, default=0, help="after the timed region, bracket the phases of this many more steps with HIP events (phase_ms in the line: pack+forward, head/loss, BPTT+weight gradients, all-reduce+Adam)")
    ap.add_argument("--lib", default=None, help="load this build of the library (A/B copies made by nasa-niswan_amd/build.py --out=...) instead of the product one")
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (torch.distributed.run as a CHILD
    process -- decided from argv before anything touches the GPU; never an exec), relay their output and return the
    child's exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    for line in proc.stdout:
        # rank 0's JSON line goes to stdout untouched; launcher chatter goes to stderr
        (sys.stdout if line.startswith('{"metric"') else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


def probe_table(words):
    """[(kind, layer, t, end, ticks)] from a probe buffer (nint.h: tag = kind | layer << 8 | t << 16 | end << 31 | 1 << 63)"""
    out = []
    for tag, ticks in zip(words[0::2], words[1::2]):
        tag = int(tag) & 0xFFFFFFFFFFFFFFFF
        if not (tag >> 63) & 1:
            continue
        out.append((tag & 0xFF, (tag >> 8) & 0xFF, (tag >> 16) & 0x7FFF, (tag >> 31) & 1, int(ticks)))
    return out


def probe_durations(words_fwd, words_bwd):
    """{(kind, layer): [(t, microseconds)]} from the forward / backward halves of a probe buffer.  A launch's duration is
    (end stamp - begin stamp) - the calibration pair's difference (two stamps back to back: what the brackets cost)."""
    res = {}
    for words in (words_fwd, words_bwd):
        tab = probe_table(words)
        if len(tab) < 2:
            continue
        cal = (tab[1][4] - tab[0][4]) / 100.0
        open_ = {}
        for kind, layer, t, end, ticks in tab[2:]:
            if not end:
                open_[(kind, layer)] = (t, ticks)
            elif (kind, layer) in open_:
                t0, b = open_.pop((kind, layer))
                res.setdefault((kind, layer), []).append((t0, (ticks - b) / 100.0 - cal))
        res.setdefault("cal_us", []).append(cal)
    return res


def probe_steps(trainer, X, y, rank, pbuf, mask, psteps, slots, timer=None):
    """The in-step pricing pass: `psteps` + 1 more trainer.step() calls ON EVERY RANK -- under a process group every step
    ends in the gradient all-reduce, so a pass that only rank 0 ran would leave its collectives without a peer (an RCCL
    hang; the round-3 advisor's finding).  Only rank 0 attaches the probe buffer and parses it; the other ranks run the
    same steps without stamps.  `timer`: callable returning (record_start, record_stop_and_ms) -- HIP events on the GPU,
    wall clock in the CPU test.  Returns (durations, step_ms) on rank 0, (None, None) elsewhere."""
    import torch
    probing = rank == 0 and pbuf is not None
    if probing:
        trainer.set_probe(pbuf, mask)
    dur, step_ms = {}, []
    for _ in range(psteps):
        if probing:
            pbuf.zero_()
            start, stop_ms = timer() if timer else (lambda: None, lambda: 0.0)
            start()
        trainer.step(X, y)
        if probing:
            step_ms.append(stop_ms())
            w = pbuf.cpu().numpy()
            for key, v in probe_durations(w[:slots], w[slots:]).items():
                dur.setdefault(key, []).extend(v)
    if probing:
        trainer.set_probe(None)
    trainer.step(X, y)                      # one step without probes: workspaces leave the pass as the timed steps left them
    return (dur, step_ms) if probing else (None, None)


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args, argv))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    json_fd = None
    if use_dist:
        # RCCL writes its version banner (five lines: "RCCL version : ...", "HIP version", ...) to STDOUT when the process group is
        # created.  The contract is ONE JSON line there: file descriptor 1 goes to stderr for the rest of the run, and rank 0's
        # line is written to a duplicate of the original stdout.
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(args.master_port))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL on ROCm

    import nasa_niswan_amd as pkg
    from nasa_niswan_amd.trainer import FusedTrainer
    from nasa_niswan_amd import engine as _engine
    _engine.FUSE_BWD = args.fuse_bwd
    _engine.FORCE_WIDE = args.wide
    _engine.FORCE_WAVE = None if args.wave < 0 else args.wave
    _engine.FORCE_TILE_ROWS = args.tile_rows
    pkg.load_library(args.lib) if args.lib else pkg.load_library()

    C, hidden, ks, out, T, Hp, Wp, halo, grid = WORKLOADS[args.workload]
    if args.scaling == "strong":
        if args.batch % world:
            raise SystemExit(f"--scaling strong: global batch {args.batch} is not a multiple of {world} ranks")
        B = args.batch // world
    else:
        B = args.batch
    torch.manual_seed(0)                                    # identical init on every rank (utils.py:77-88)
    model = pkg.ConvLSTM(C, list(hidden), list(ks), len(hidden), out_channels=out, compute_dtype=args.dtype).to(dev)
    trainer = FusedTrainer(model, lr=1e-3, betas=(0.5, 0.999), halo=halo, distributed=use_dist, overlap_allreduce=args.overlap_allreduce)
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)   # each rank its own shard of synthetic data
    X = torch.randn(B, T, C, Hp, Wp, device=dev, generator=gen)
    y = torch.randn(B, out, grid[0], grid[1], device=dev, generator=gen)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(nsteps):
        sync()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            loss_ = trainer.step(X, y)
        sync()
        el = time.perf_counter() - t0
        timed.local = el
        if use_dist:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt)
        return el, loss_

    for _ in range(args.warmup):
        trainer.step(X, y)
    elapsed, loss = timed(args.steps)
    local_elapsed = timed.local
    final_loss = float(loss)
    long_elapsed = timed(args.long_steps)[0] if args.long_steps > 0 else None

    # ---- the gradient all-reduce by itself (N > 1 or --force-dist): HIP events around dist.all_reduce on the step's
    # stream, the flat bucket exactly as trainer.step() reduces it
    ar_ms = bus_bw = None
    if use_dist:
        st = torch.cuda.current_stream()
        buf = trainer.flat.grad
        for _ in range(3):
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_ar = 20
        e0.record(st)
        for _ in range(n_ar):
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        e1.record(st)
        e1.synchronize()
        ar = torch.tensor([e0.elapsed_time(e1) / n_ar], dtype=torch.float64, device=dev)
        dist.all_reduce(ar, op=dist.ReduceOp.MAX)
        ar_ms = float(ar)
        nbytes = buf.numel() * 4
        # ring all-reduce: every rank sends and receives 2 (N-1)/N of the bucket; "bus bandwidth" as nccl-tests define it
        bus_bw = (2.0 * (world - 1) / world) * nbytes / (ar_ms * 1e-3) / 1e9 if world > 1 else None

    # ---- phase split of the step by HIP events (diagnostic: which phase a slow process loses its time in)
    phase_ms = None
    if args.phase_events > 0:
        import numpy as _np
        rows = []
        for _ in range(args.phase_events):
            trainer._marks = []
            trainer.step(X, y)
            torch.cuda.synchronize()
            ev = trainer._marks
            rows.append([ev[i].elapsed_time(ev[i + 1]) for i in range(4)])
        trainer._marks = None
        med = _np.median(_np.array(rows), axis=0)
        phase_ms = {"pack_forward": round(float(med[0]), 4), "head_loss": round(float(med[1]), 4), "bptt_wgrad_fold": round(float(med[2]), 4),
                    "allreduce_adam": round(float(med[3]), 4), "steps": args.phase_events}

    # ---- per-rank step time (N > 1): the slowest rank sets `value`; the spread says whether one device lags
    rank_ms = None
    if use_dist:
        mine = torch.tensor([1e3 * local_elapsed / args.steps], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_ms = [round(float(t), 3) for t in allr]

    # ---- kernel rooflines, priced in the step (probe pass) and in a warm loop (HIP events)
    roof, roof_all, phases = None, None, None
    eng = model._engine(dev)
    wave_on = None
    for wsl in eng.pool.values():
        for ws_ in wsl:
            wave_on = int(ws_.seq.wave)             # the launch schedule of the timed steps (nint_seq.wave: 0 / 1 / 2 / 4)
    PSTEPS, SLOTS = 5, 2048
    dur = step_ms = None
    if not args.no_kernel_rooflines:
        # probe pass: PSTEPS more steps with stamps around the layer-0 gate / dgrad / pointwise launches, every layer's
        # weight-gradient block and the fold.  EVERY rank runs the steps (each ends in the all-reduce); rank 0 alone stamps.
        pbuf = torch.zeros(2 * SLOTS, dtype=torch.int64, device=dev) if rank == 0 else None
        mask = (1 << PROBE_GATE) | (1 << PROBE_POINTWISE) | (1 << PROBE_DGRAD) | (1 << PROBE_FUSED) | (1 << PROBE_WGRAD) | (1 << PROBE_FOLD) | (1 << PROBE_WAVE) | (1 << PROBE_BWD_PAIR) | (1 << PROBE_BWD_PW)
        st_ = torch.cuda.current_stream()

        def ev_timer():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def stop():
                e1.record(st_)
                e1.synchronize()
                return e0.elapsed_time(e1)
            return (lambda: e0.record(st_)), stop
        dur, step_ms = probe_steps(trainer, X, y, rank, pbuf, mask, PSTEPS, SLOTS, ev_timer)
    if rank == 0 and dur is not None:
        import ctypes as Ct
        import numpy as np
        lib = pkg.load_library()
        st = torch.cuda.current_stream()
        k0, ch0 = ks[0], hidden[0]

        def mean_us(kind, layer, pred=lambda t: True):
            v = [us for t, us in dur.get((kind, layer), []) if pred(t)]
            return (float(np.mean(v)), len(v) // PSTEPS) if v else (None, 0)

        ws = eng.acquire(B, T, Hp, Wp, True, False)
        ly, g = eng.layers[0], ws.g
        sp = Ct.c_void_p(st.cuda_stream)
        es = eng.es
        halo_px, comp_px = g.Hh * g.Wh, Hp * Wp
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

        def entry(key, kernel, bound, work, ms, ms_loop, note=None):
            """work = algorithmic FLOPs (mfma) or bytes (hbm) per launch; ms = in-step duration (probes), ms_loop = warm loop"""
            use = ms if ms is not None else ms_loop
            if bound == "mfma":
                ach, pk, unit = work / (use * 1e-3) / 1e12, peak, "TFLOP/s"
            else:
                ach, pk, unit = work / (use * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
            # HBM bytes per launch from the committed PMC pass (PMC needs rocprofv3 around the process, so it is
            # not collected live); only quoted for the exact configuration it was measured on
            tr = traffic.get(f"{args.workload}/{args.dtype}/B{B}/{key}", {}).get("bytes_per_launch")
            e = {"kernel": kernel, "bound": bound, "achieved": round(ach, 2), "peak": pk, "unit": unit,
                 "frac": round(ach / pk, 4), "traffic": tr, "ms_per_launch": round(use, 4),
                 "timing": "in-step (nint_seq.probe stamps)" if ms is not None else "warm loop (HIP events)",
                 "ms_per_launch_loop": round(ms_loop, 4), "frac_loop": round(ach / pk * use / ms_loop, 4),
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

        f_gate = 2.0 * B * Hp * Wp * k0 * k0 * (C + ch0) * 4 * ch0           # algorithmic, per full-K launch
        f_dgrad = 2.0 * B * Hp * Wp * k0 * k0 * 4 * ch0 * ch0
        f_wgrad = 2.0 * B * Hp * Wp * k0 * k0 * (T * C + (T - 1) * ch0) * 4 * ch0   # as executed in the step: the h part skips t = 0
        f_wgrad_loop = 2.0 * T * B * Hp * Wp * k0 * k0 * (C + ch0) * 4 * ch0        # the stand-alone entry reduces all T steps of both sources
        b_pw = float(B * comp_px * ch0 * (9 * es + 16))     # gates + dG (4 ET each) + dh (ET); c_prev, c_new, dc in, dc out (f32)
        us_gate, n_gate = mean_us(PROBE_GATE, 0, lambda t: t >= 1)           # full-K launches only (t = 0 has no h half)
        us_gate0, _ = mean_us(PROBE_GATE, 0, lambda t: t == 0)
        us_dgrad, n_dgrad = mean_us(PROBE_DGRAD, 0)
        us_pw, n_pw = mean_us(PROBE_POINTWISE, 0)
        us_wg, _ = mean_us(PROBE_WGRAD, 0)
        us_fold, _ = mean_us(PROBE_FOLD, 0)
        us_pair, _ = mean_us(PROBE_BWD_PAIR, 1)          # BPTT grid 1 (nint_seq.wave = 4 / 5): layer-0 dgrad of u+1 + layer-1 dgrad of u
        ms_ = lambda us: None if us is None else us * 1e-3
        gate_loop = time_kernel(k_fwd, 50, st)
        roof_gate = entry("conv_igemm_fwd_layer0", "layer-0 gate kernel, LSTM epilogue (B images, one time step, full K)", "mfma",
                          f_gate, ms_(us_gate), gate_loop,
                          note=None if us_gate0 is None else f"t = 0 launch (x half of K only): {us_gate0:.1f} us in the step")
        roof = roof_gate
        # With the forward wavefront merged (nint_seq.wave), the step's dominant kernel is the merged grid: one wavefront step --
        # gate(0, w), gate(1, w-1), gate(2, w-2), ... -- per launch.  Priced over ALL merged launches of a step (the two partial
        # ones at the ends of the wavefront included), so that ms_per_launch is comparable with the kernel's average duration in
        # a rocprofv3 --kernel-trace --stats of the same command.
        wave_rows = [(n_, w_, us) for key_, v_ in dur.items() if key_ != "cal_us" and key_[0] == PROBE_WAVE
                     for (w_, us) in v_ for n_ in [key_[1]]]
        roof_wave = None
        if wave_rows and wave_on:
            def gate_flops(l, t):
                cin = C if l == 0 else hidden[l - 1]
                return 2.0 * B * Hp * Wp * ks[l] ** 2 * (cin + (hidden[l] if t > 0 else 0)) * 4 * hidden[l]
            Lm = len(hidden)
            fl = [sum(gate_flops(l, w_ - l) for l in range(Lm) if 0 <= w_ - l < T) for (n_, w_, us) in wave_rows]
            full = [(f, us) for f, (n_, w_, us) in zip(fl, wave_rows) if n_ == Lm]
            ms_wave = float(np.mean([us for _, _, us in wave_rows])) * 1e-3
            roof_wave = entry("conv_lstm_multi8_fwd_wavefront",
                              "conv_lstm_multi8_kernel: one forward wavefront step as ONE grid (gate(0, w), gate(1, w-1), gate(2, w-2): every layer on 8-row tiles)",
                              "mfma", float(np.mean(fl)), ms_wave, ms_wave,
                              note=(f"average over the {len(wave_rows) // PSTEPS} merged launches of a step; the {len(full) // PSTEPS} launches that hold all "
                                    f"{Lm} layers: {np.mean([f for f, _ in full]) / 1e9:.1f} GFLOP in {np.mean([us for _, us in full]):.1f} us; no stand-alone "
                                    "loop exists for a merged grid (ms_per_launch_loop repeats the in-step figure); the layer-0 gate kernel "
                                    f"alone, one launch per time step: {gate_loop * 1e3:.1f} us in a warm loop") if full else None)
            roof = roof_wave
        wg_loop = time_kernel(k_wgrad, 10, st)
        roof_all = [
            dict(roof, launches_per_step=(len(wave_rows) // PSTEPS) if roof_wave is not None else (T - 1 if us_gate is not None else T)),
            dict(entry("wgrad_layer0", "layer-0 weight gradient (bf16: wgrad_wide_kernel<5, 2>, the 8-wave 128-column kernel; f32: wgrad_kernel): x part over T steps + h part over T-1 steps (in the step, without the fold launch)",
                       "mfma", f_wgrad, ms_(us_wg), wg_loop * f_wgrad / f_wgrad_loop,
                       note=f"loop figure = the stand-alone entry (all T steps of both sources + the fold, {wg_loop:.3f} ms) scaled by the executed / nominal FLOPs"),
                 launches_per_step=1),
            dict(entry("conv_igemm_dgrad_layer0", "layer-0 dgrad kernel (h columns; B images, one time step)",
                       "mfma", f_dgrad, ms_(us_dgrad), time_kernel(k_dgrad, 50, st),
                       note=None if us_dgrad is not None or us_pair is None else
                       f"in the timed steps this launch is one half of a merged BPTT grid ({us_pair:.1f} us together with layer 1's dgrad: phases.per_step_us); priced in a warm loop"),
                 launches_per_step=n_dgrad or T - 1),
            dict(entry("lstm_bwd_pointwise_layer0", "lstm_bwd_pointwise_kernel layer 0 (B images, one time step)", "hbm",
                       b_pw, ms_(us_pw), time_kernel(k_pw, 50, st)), launches_per_step=n_pw or T),
        ]
        eng.release(ws)
        # per (kind, layer) in-step totals: what the step spends where (microseconds per step)
        names = {PROBE_GATE: "gate", PROBE_POINTWISE: "pointwise", PROBE_DGRAD: "dgrad", PROBE_FUSED: "fused_bptt_step",
                 PROBE_WGRAD: "wgrad", PROBE_FOLD: "fold", PROBE_WAVE: "fwd_wavefront_grid_of_",
                 PROBE_BWD_PAIR: "bptt_grid_dgrad0_with_launch_of_layer", PROBE_BWD_PW: "bptt_grid_pointwise0_with_fused_step_of_layer"}
        phases = {"step_ms_with_probes": round(float(np.median(step_ms)), 3),
                  "probe_pair_cost_us": round(float(np.mean(dur.get("cal_us", [0.0]))), 2),
                  # probes do not change the schedule: merged grids are bracketed as such (kinds 7, 8, 9)
                  "wave": wave_on, "schedule_differs_from_timed_steps": False,
                  "per_step_us": {}}
        for key, v in sorted((k, v) for k, v in dur.items() if k != "cal_us"):
            phases["per_step_us"][f"{names[key[0]]}{key[1]}"] = {"launches": len(v) // PSTEPS,
                                                                  "us_per_step": round(sum(us for _, us in v) / PSTEPS, 1),
                                                                  "us_per_launch": round(float(np.mean([us for _, us in v])), 2)}

    samples = world * B * args.steps
    value = samples / elapsed
    if rank == 0:
        f_train = 3 * fwd_flops_per_sample(C, hidden, ks, out, T, Hp, Wp)
        f_exec = executed_flops_per_sample(C, hidden, ks, out, T, Hp, Wp)
        line = {
            "metric": "training samples/sec (90x144x20 grid, seq_len=12)",
            "value": round(value, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": args.workload, "in_channels": C, "hidden": list(hidden), "kernels": list(ks),
                       "out_channels": out, "seq_len": T, "padded_grid": [Hp, Wp], "grid": list(grid),
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "wave": wave_on, "overlap_allreduce": bool(args.overlap_allreduce)},
            # which build produced the line: the product library next to the package, or an A/B copy (--lib)
            "lib": os.path.realpath(args.lib) if args.lib else "product", "nint_version": int(pkg.load_library().nint_version()),
            # the same measurement over a window long enough that the timer does not matter (the headline window is 0.17 s)
            "value_200steps": None if long_elapsed is None else round(world * B * args.long_steps / long_elapsed, 3),
            "long_window": None if long_elapsed is None else {"steps": args.long_steps, "seconds": round(long_elapsed, 3)},
            # whole step priced two ways: the survey's nominal 3 x F_fwd, and the FLOPs the step really executes
            # (zero-state h parts and the layer-0 input gradient are legitimately skipped)
            "whole_step_nominal_tflops": round(value * f_train / 1e12, 2),
            "whole_step_nominal_mfma_frac": round(value * f_train / 1e12 / world / MFMA_PEAK_TFLOPS[args.dtype], 4),
            "whole_step_executed_tflops": round(value * f_exec / 1e12, 2),
            "whole_step_executed_mfma_frac": round(value * f_exec / 1e12 / world / MFMA_PEAK_TFLOPS[args.dtype], 4),
            # (keys of round 1, kept as aliases of the nominal figures so that trend readers keyed on them still line up)
            "whole_step_tflops": round(value * f_train / 1e12, 2),
            "whole_step_mfma_frac": round(value * f_train / 1e12 / world / MFMA_PEAK_TFLOPS[args.dtype], 4),
            "final_loss": round(final_loss, 5),
            # N > 1: every rank's own step time (the headline uses the slowest), and the exchange step's share of the step
            "rank_ms_per_step": None if rank_ms is None else {"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms},
            "allreduce_ms": None if ar_ms is None else round(ar_ms, 4),
            "allreduce_frac_of_step": None if ar_ms is None else round(ar_ms / (1e3 * elapsed / args.steps), 4),
            "allreduce_bytes": trainer.flat.grad.numel() * 4 if use_dist else None,
            "bus_bw_GBs": None if bus_bw is None else round(bus_bw, 2),
            "bus_bw_bound_GBs": XGMI_LINK_GBS if world > 1 else None,
            "phase_ms": phase_ms,
            "roofline": roof,
            "roofline_kernels": roof_all,
            "phases": phases,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.workload)
        else:
            line["cpu_baseline"] = None
        if json_fd is not None:
            os.write(json_fd, (json.dumps(line) + "\n").encode())
        else:
            print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
