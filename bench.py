#!/usr/bin/env python3
"""bench.py -- traces/s of the of1x1 hot path on MI355X (BASELINE.json metric).

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N > 1 it runs one rank per
GPU (RCCL) under torch.distributed.run -- either launched that way by the driver (WORLD_SIZE set)
or, when started as plain `python bench.py --gpus N`, by launching itself: the parent starts the
N ranks as a child process before it has touched the GPU, relays rank 0's JSON line and exits with
the child's code (the reference's fan-out: Pool.starmap over series, features.py:405-420).  One "step" = one
pass of the hot path over the resident batch of synthetic events (already in HBM when
the timed region starts).  Event batches are sharded across ranks by
``detprocess_amd.dist.run_sharded`` (weak scaling: every rank owns --traces traces); the
only collective is the all-gather of the feature matrix that ends each pass (SURVEY.md
section 8e: one gather per job; a step is one whole job here, so it is inside the timed
region, 32 bytes per event).  Rank 0 prints ONE JSON line.

--config 1 (default)  BASELINE configs[1]: 1M traces x 32768, of1x1_unconstrained, 1 template
--config 2            BASELINE configs[2]: of1x1_constrained (+-400 us) + integral + min/max
--config 3            BASELINE configs[3]: 4 channels x 32768, 3 template tags (pulse / glitch /
                      muon), the feature set of examples/processing/process_example.yaml:109-222
                      through the YAML driver; 262,144 events = 1M traces
--samples 25000       the same workloads at the trace length of the reference's own example YAML
                      (examples/processing/process_example.yaml:93; the k_fused25 kernel); the
                      default 32768 is BASELINE's length and the only one the driver runs
--stream-events E     BASELINE configs[4] rehearsal on this rank count: E events per GPU consumed
                      in --chunk pieces, generated on a producer stream beside the hot path
                      (run_sharded's streaming form); `value` then includes the generation.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SAMPLES = 32768
FS = 1.25e6
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md
CHANNELS3 = ["chA", "chB", "chC", "chD"]
TAGS3 = ["pulse", "glitch", "muon"]


def yaml_config3():
    """The per-channel feature block of the reference's example
    (examples/processing/process_example.yaml:109-222: of1x1_nodelay / _unconstrained /
    _constrained with lowchi2_fcutoff 50 kHz, baseline, baseline_end, maximum, minimum, integral,
    psd_amp), once per template tag for the OF algorithms, on four channels."""
    lines = [",".join(CHANNELS3) + ":"]
    for tag in TAGS3:
        lines += [f"    of1x1_nodelay_{tag}:", "        run: True",
                  "        base_algorithm: of1x1_nodelay", "        lowchi2_fcutoff: 50000",
                  f"        template_tag: {tag}", "        csd_tag: default",
                  f"    of1x1_unconstrained_{tag}:", "        run: True",
                  "        base_algorithm: of1x1_unconstrained",
                  f"        template_tag: {tag}", "        csd_tag: default",
                  f"    of1x1_constrained_{tag}:", "        run: True",
                  "        base_algorithm: of1x1_constrained",
                  "        window_min_from_trig_usec: -100",
                  "        window_max_from_trig_usec: 100", "        lowchi2_fcutoff: 50000",
                  f"        template_tag: {tag}", "        csd_tag: default"]
    lines += ["    baseline:", "        run: True", "        window_min_from_start_usec: 0",
              "        window_max_from_trig_usec: -2000",
              "    baseline_end:", "        run: True", "        base_algorithm: baseline",
              "        window_min_from_trig_usec: 2000", "        window_max_to_end_usec: 0",
              "    maximum:", "        run: True", "        window_min_from_trig_usec: -500",
              "        window_max_from_trig_usec: 500",
              "    minimum:", "        run: True", "        window_min_from_trig_usec: -500",
              "        window_max_from_trig_usec: 500",
              "    integral:", "        run: True", "        window_min_from_trig_usec: -10",
              "        window_max_from_trig_usec: 500",
              "    psd_amp:", "        run: True",
              "        f_lims: [[45.0, 75.0], [300.0, 500.0], [350.0, 450.0], [150, 250], [250, 350]]"]
    return "\n".join(lines) + "\n"


def filter_data3(pre):
    from detprocess_amd import FilterData, synth
    fd = FilterData()
    f = np.fft.fftfreq(N_SAMPLES, d=1 / FS)
    J = synth.make_psd(N_SAMPLES, FS)
    for ch in CHANNELS3:
        for tag in TAGS3:
            fd.set_template(ch, synth.make_template(N_SAMPLES, pre, FS, tag), sample_rate=FS,
                            pretrigger_length_samples=pre, tag=tag)
        fd.set_psd(ch, J, f, sample_rate=FS, tag="default")
    return fd


def measured_traffic(tag, traces_per_launch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*traffic.json,
    FETCH_SIZE / WRITE_SIZE collected separately and corrected as MI355X_MICROARCH.md
    prescribes); scaled to this run's traces per launch.  (bytes, source file) -- it is the stored
    figure of the profiled run of the same workload, not a measurement of THIS run: the counters
    need rocprofv3 around the process.  (None, None) if no matching profile."""
    import glob
    best, src = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            rec = json.load(open(f))
        except Exception:
            continue
        if rec.get("workload", "config1") == tag and "hbm_bytes_per_trace" in rec:
            best, src = rec, os.path.relpath(f, ROOT)
    if best is None:
        return None, None
    return best["hbm_bytes_per_trace"] * traces_per_launch, src


def secondary_ceiling(tag, traces_per_s, n_gpus):
    """The ceiling that actually binds the fused kernel: VALU issue.  Every CU issues at most one
    VALU wave-instruction per cycle (four SIMDs, four cycles per 64-lane instruction, packed fp32
    included), so the chip peaks at 256 x clock wave-instructions/s.  `insts_per_trace` is
    SQ_INSTS_VALU of the committed counter pass (profiles/*sq_counters*.json), `clock_mhz` the shader
    clock INSIDE the kernel from the stamped build (d s_memtime / d s_memrealtime,
    profiles/*phase_timeline*.json) -- both stored figures of profiled runs of this workload, named in
    `source`; `achieved` is this run's rate times insts_per_trace."""
    import glob
    sq = clk = None
    src = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*sq_counters*.json"))):
        try:
            rec = json.load(open(f))
        except Exception:
            continue
        wl = rec.get("workload")
        if wl is None:                             # round-1/2 files: the kernel name tells
            k = rec.get("kernel", "")
            wl = "config1" if k.startswith("k_fused<") else ("config1_n25000" if "k_fused25" in k else None)
        if wl == tag and "SQ_INSTS_VALU" in rec.get("per_trace", {}):
            sq, sqf = rec["per_trace"]["SQ_INSTS_VALU"], os.path.relpath(f, ROOT)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*phase_timeline*.json"))):
        try:
            rec = json.load(open(f))
        except Exception:
            continue
        if rec.get("workload", "config1") == tag and isinstance(rec.get("clock_mhz"), dict):
            clk, clkf = rec["clock_mhz"]["mean"], os.path.relpath(f, ROOT)
    if sq is None:
        return None, None
    ach = traces_per_s * sq                       # VALU wave-instructions per second, whole job
    out = {"bound": "valu_issue", "insts_per_trace": sq, "achieved": ach / 1e9,
           "unit": "G wave-instructions/s", "source": [sqf]}
    if clk is not None:
        peak = 256.0 * n_gpus * clk * 1e6
        out.update({"peak": peak / 1e9, "frac": ach / peak, "clock_mhz": clk})
        out["source"].append(clkf)
    else:
        out.update({"peak": None, "frac": None, "clock_mhz": None})
    return out, clk


def _cpu_setup():
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS",
              "NUMEXPR_NUM_THREADS"):
        os.environ[k] = "1"                        # mirrors features.py:35-38
    from detprocess_amd import synth
    from oracle import of1x1 as orc
    pre = N_SAMPLES // 2
    tmpl = synth.make_template(N_SAMPLES, pre, FS)
    psd = synth.make_psd(N_SAMPLES, FS)
    filt = orc.OFFilter(tmpl, psd, FS, pre)
    return orc, filt, tmpl, psd


def _cpu_kwargs(config):
    if config == 2:
        return "constrained", dict(window_min_from_trig_usec=-400, window_max_from_trig_usec=400)
    return "unconstrained", {}


def _cpu_worker(args):
    """One process of the all-cores leg: the per-event loop on its own events."""
    seconds, config, seed = args
    orc, filt, tmpl, psd = _cpu_setup()
    from detprocess_amd import synth
    traces, _, _ = synth.make_traces(64, tmpl, psd, FS, filt.ampres, seed=seed)
    mode, kw = _cpu_kwargs(config)
    orc.process_events(filt, traces[:4], mode, **kw)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        orc.process_events(filt, traces, mode, **kw)
        if config == 2:
            for tr in traces:
                orc.integral(tr, FS, 16372, 17009), orc.maximum(tr, 15759, 17009), \
                    orc.minimum(tr, 15759, 17009)
        done += len(traces)
    return done, time.perf_counter() - t0


def _usable_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota
    (a GPU box shows every host core in the mask but grants a share of them)."""
    n = len(os.sched_getaffinity(0))
    for quota_file, period_file in (("/sys/fs/cgroup/cpu.max", None),
                                    ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us",
                                     "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if period_file is None:
                quota, period = open(quota_file).read().split()[:2]
            else:
                quota, period = open(quota_file).read().strip(), open(period_file).read().strip()
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
            break
        except (OSError, ValueError):
            continue
    return min(n, 64)          # one process per core; more than this only measures fork time


def cpu_baseline(config, seconds=8.0):
    """The oracle (fp64 NumPy restatement of the detprocess+QETpy per-event path) timed on
    this host on a bounded sample of the same workload: `value` = 1
    core, per-event loop, thread caps as features.py:35-38; extra keys: one process per core
    on every core this process may use (the reference's Pool over series, features.py:405-417)
    and a batched scipy.fft variant with workers = all cores (a stronger CPU baseline that
    skips lowchi2)."""
    import multiprocessing as mp
    done, el = _cpu_worker((seconds, config, 123))
    rec = {"value": done / el, "unit": "traces/s", "cores": 1, "kind": "port",
           "sample": f"{done} traces x {N_SAMPLES} samples, per-event loop of oracle/of1x1.py "
                     f"(fp64 NumPy FFTs, 1 thread), {el:.1f} s; workload of --config {config}"
                     + (" (one channel, one template tag of it)" if config == 3 else "")}
    ncores = _usable_cores()
    try:
        with mp.get_context("spawn").Pool(ncores) as pool:
            res = pool.map(_cpu_worker, [(seconds, config, 200 + i) for i in range(ncores)])
        rec["all_cores"] = {"value": sum(d / e for d, e in res), "unit": "traces/s",
                            "cores": ncores, "sample": f"one process per core, {seconds:.0f} s each"}
    except Exception as exc:                       # pragma: no cover
        rec["all_cores"] = {"error": str(exc)}
    try:
        import scipy.fft as sf
        orc, filt, tmpl, psd = _cpu_setup()
        from detprocess_amd import synth
        X, _, _ = synth.make_traces(512, tmpl, psd, FS, filt.ampres, seed=7)
        K = N_SAMPLES // 2 + 1
        Wf = filt.Wf[:K]
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds / 2:
            V = sf.rfft(X, axis=1, workers=-1)
            A = sf.irfft(V * Wf, n=N_SAMPLES, axis=1, workers=-1) * N_SAMPLES
            (2.0 * (V.real ** 2 + V.imag ** 2) * filt.g[:K]).sum(axis=1)
            np.argmax(np.roll(A * A, filt.pre, axis=1), axis=1)
            reps += 1
        el = time.perf_counter() - t0
        rec["batched_scipy"] = {"value": reps * X.shape[0] / el, "unit": "traces/s",
                                "cores": ncores,
                                "sample": "scipy.fft rfft/irfft workers=-1 on 512-trace batches, "
                                          "chi2_0 + arg-max, no lowchi2"}
    except Exception as exc:                       # pragma: no cover
        rec["batched_scipy"] = {"error": str(exc)}
    return rec


def metric_name():
    """BASELINE.json's metric string (the file travels with the repository)."""
    fallback = ("traces/sec of1x1_unconstrained (32768-sample, 1ch) @1/2/4/8 GPU; "
                "% HBM roofline")
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as fh:
            return json.load(fh).get("metric", fallback)
    except (OSError, ValueError):
        return fallback


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks under
    torch.distributed.run as a CHILD process (never exec: this process must not have touched the
    GPU, and does not), pass its output through and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL over xGMI)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def main():
    global N_SAMPLES
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3])
    ap.add_argument("--traces", type=int, default=1 << 20,
                    help="traces resident per GPU (default 1M; config 3: 4 per event)")
    ap.add_argument("--engine", default="auto", choices=["auto", "fused", "rocfft", "lds"])
    ap.add_argument("--stream-events", type=int, default=0,
                    help="events per GPU of a streamed run (configs[4] rehearsal), config 1 only")
    ap.add_argument("--chunk", type=int, default=1 << 19, help="events per chunk when streaming")
    ap.add_argument("--samples", type=int, default=32768, help="trace length (32768 = BASELINE)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    args = ap.parse_args()
    N_SAMPLES = args.samples
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    from detprocess_amd import (FeatureProcessing, OFPlan, SynthSource, build_filter, search_range,
                                synth, utils)
    from detprocess_amd import dist as ofdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with "
                         f"--nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    pre = N_SAMPLES // 2
    tmpl = synth.make_template(N_SAMPLES, pre, FS)
    psd = synth.make_psd(N_SAMPLES, FS)
    ft = build_filter(tmpl, psd, FS, pre)
    C = 4 if args.config == 3 else 1
    streaming = args.stream_events > 0
    if streaming and args.config != 1:
        raise SystemExit("--stream-events runs the config 1 workload")
    # v = A roll(template, d) + coloured noise drawn from J; A log-uniform in
    # [3, 300] sigma_A for half of the events, d uniform in [-2000, 2000]; keyed by (seed, index)
    # (streaming: white noise of the PSD's median level -- the generator then costs one HBM write
    # per trace, like an ingest would; resident shards keep the coloured noise of SURVEY.md 8d)
    gen = SynthSource(N_SAMPLES, tmpl, psd, FS, 3 * ft.ampres, 300 * ft.ampres, 0.5, 2000,
                      seed=2026, device=local_rank, white=streaming)

    # events per GPU, capped by free HBM (input + output + slack)
    B = (args.chunk * 2 if streaming else args.traces // C)
    free, _ = torch.cuda.mem_get_info(dev)
    maxB = int((free - (8 << 30)) // (C * N_SAMPLES * 4 + 1024))
    B = min(B, maxB)
    if streaming:                  # two event buffers of one chunk each must fit
        args.chunk = max(1, min(args.chunk, B // 2))
    if world > 1:       # every rank holds the same number of events (weak scaling)
        tb = torch.tensor([B], dtype=torch.int64, device=dev)
        dist.all_reduce(tb, op=dist.ReduceOp.MIN)
        B = int(tb.item())

    # ------------------------------------------------------------------ the workload
    plans = []
    if args.config in (1, 2):
        plan = OFPlan(N_SAMPLES, pre, FS, max_batch=8192, device=local_rank, engine=args.engine)
        plan.set_filter(0, ft)
        if args.config == 1:
            plan.add_search(0, "delay")                       # of1x1_unconstrained
            out_floats = 4
            what = "of1x1_unconstrained (BASELINE configs[1]" + ("" if N_SAMPLES == 32768 else
                                                                 f"'s workload at {N_SAMPLES} samples") + ")"
        else:
            lo, hi = search_range(N_SAMPLES, pre, FS, -400, 400)
            plan.add_search(0, "delay", lo, hi)               # of1x1_constrained +-400 us
            wi = utils.get_window_indices(nb_samples=N_SAMPLES, nb_pretrigger_samples=pre, fs=FS,
                                          window_min_from_trig_usec=-10,
                                          window_max_from_trig_usec=500)
            wm = utils.get_window_indices(nb_samples=N_SAMPLES, nb_pretrigger_samples=pre, fs=FS,
                                          window_min_from_trig_usec=-500,
                                          window_max_from_trig_usec=500)
            plan.add_tdwindow(*wi)                            # integral
            plan.add_tdwindow(*wm)                            # maximum, minimum
            out_floats = 7 + 3
            what = ("of1x1_constrained (+-400 us) + integral + minimum + maximum fused "
                    "(BASELINE configs[2])")
        plans = [plan]
        row = plan.row_floats
        event_shape = (N_SAMPLES,)

        def process(ev, out):
            plan.process(ev, out=out)
    else:
        fp = FeatureProcessing(yaml_config3(), filter_data3(pre), CHANNELS3, FS, device=local_rank,
                               engine=args.engine, max_batch=8192)
        pl = fp.plans(N_SAMPLES)
        keys = list(pl)
        plans = [pl[k] for k in keys]
        cols = fp.device_columns()
        out_floats = sum(len(cols[k]) for k in keys) // len(keys)      # per channel
        rows = [p.row_floats for p in plans]
        row = sum(rows)
        event_shape = (C, N_SAMPLES)
        what = (f"4 channels x 3 template tags ({'/'.join(TAGS3)}) x the feature set of "
                f"process_example.yaml:109-222 ({out_floats} columns per channel) through the "
                f"YAML driver (BASELINE configs[3])")

        def process(ev, out):
            # one launch per channel plan; plan k writes its contiguous [n, row_k] block into
            # the flat storage of `out` ([n, sum row_k]): a bench layout -- the product API
            # (FeatureProcessing.process) hands every plan's matrix back under its column names
            n = ev.shape[0]
            flat = out.reshape(-1)
            o, outs = 0, {}
            for k, r in zip(keys, rows):
                outs[k] = flat[o:o + n * r].view(n, r)
                o += n * r
            fp.process_device(ev, outs=outs)

    engines = sorted({p.engine for p in plans})
    algo_bytes = N_SAMPLES * 4 + out_floats * 4          # per trace

    if streaming:
        shard = None
        total = world * args.stream_events
    else:
        shard = torch.empty((B,) + event_shape, dtype=torch.float32, device=dev)
        flat = shard.reshape(B * C, N_SAMPLES)
        for b0 in range(0, B * C, 1 << 16):
            b1 = min(b0 + (1 << 16), B * C)
            gen.fill(rank * B * C + b0, rank * B * C + b1, flat[b0:b1])
        torch.cuda.synchronize()
        total = world * B
    n_local = args.stream_events if streaming else B
    out = torch.empty((n_local, row), dtype=torch.float32, device=dev)
    bufs = None
    if streaming:
        bufs = [torch.empty((min(args.chunk, B),) + event_shape, dtype=torch.float32, device=dev)
                for _ in range(2)]

    def step():
        return ofdist.run_sharded(total, args.chunk if streaming else B,
                                  gen.fill if streaming else shard, process, row, event_shape,
                                  rank=rank, world=world, device=dev, out=out, buffers=bufs)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    for p in plans:
        p.enable_timing(True)
        p.kernel_time()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    kt = [p.kernel_time() for p in plans]

    if rank == 0:
        traces_per_step = world * n_local * C
        value = traces_per_step * args.steps / el
        k_n = sum(n for _, n in kt)
        k_ms = sum(ms * n for ms, n in kt) / max(1, k_n)          # average launch duration
        traces_per_launch = n_local * C * args.steps / max(1, k_n)
        ach = traces_per_launch * algo_bytes / (k_ms * 1e-3) / 1e9 if k_ms else 0.0
        wtag = f"config{args.config}" + ("" if N_SAMPLES == 32768 else f"_n{N_SAMPLES}")
        traffic, traffic_src = measured_traffic(wtag, traces_per_launch)
        secondary, clock_mhz = (None, None) if streaming else secondary_ceiling(wtag, value, world)
        rec = {
            "metric": metric_name(),
            "value": value, "unit": "traces/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n_local} events/GPU x {C} channel(s) x {N_SAMPLES} samples, "
                                   + what
                                   + (f"; streamed in chunks of {args.chunk} events generated "
                                      f"(white noise + pulse, k_synth) on a producer stream "
                                      f"(configs[4] rehearsal)"
                                      if streaming else ""),
                       "bench_config": args.config, "engine": "+".join(engines),
                       "events_per_gpu": n_local, "traces_per_gpu": n_local * C,
                       "events_per_s": value / C,
                       "n_samples": N_SAMPLES, "fs": FS, "streaming": streaming,
                       "parallelism": f"event-range shards x{world} (run_sharded), one all-gather "
                                      f"of the feature matrix per pass"},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "secondary": secondary, "clock_mhz": clock_mhz,
                         "kernel_ms": k_ms, "launches": k_n,
                         "traces_per_launch": traces_per_launch,
                         "algorithmic_bytes_per_trace": algo_bytes},
        }
        if not args.no_cpu_baseline and world == 1:
            rec["cpu_baseline"] = cpu_baseline(args.config, args.cpu_seconds)
        else:
            rec["cpu_baseline"] = None
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
