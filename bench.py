#!/usr/bin/env python3
"""bench.py -- traces/s of of1x1_unconstrained (32768-sample, 1 channel, 1 template).

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N > 1 it is
launched under torch.distributed.run, one rank per GPU (RCCL).  One "step" = one
pass of the hot path over the resident batch of synthetic traces (BASELINE.json
configs[1]: 1M traces x 32768 samples, fp32, already in HBM when the timed
region starts).  Event batches are sharded across ranks (weak scaling: every
rank owns --traces events); the only collective is the final all-gather of the
feature matrix.  Rank 0 prints ONE JSON line.
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
ALGO_BYTES_PER_TRACE = N_SAMPLES * 4 + 16
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md


def measured_traffic(engine, traces_per_launch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*traffic.json,
    FETCH_SIZE / WRITE_SIZE collected separately and corrected as MI355X_MICROARCH.md
    prescribes); scaled to this run's traces per launch.  None if no matching profile."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic.json"))):
        try:
            rec = json.load(open(f))
        except Exception:
            continue
        if rec.get("engine") == engine:
            best = rec
    if best is None:
        return None
    return best["hbm_bytes_per_trace"] * traces_per_launch


def cpu_baseline(seconds_target=12.0):
    """The oracle (fp64 NumPy restatement of the detprocess+QETpy per-event path)
    timed on this host, 1 core, on a bounded sample of the same workload."""
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS",
              "NUMEXPR_NUM_THREADS"):
        os.environ.setdefault(k, "1")             # mirrors features.py:35-38
    from detprocess_amd import build_filter, synth
    from oracle import of1x1 as orc
    pre = N_SAMPLES // 2
    tmpl = synth.make_template(N_SAMPLES, pre, FS)
    psd = synth.make_psd(N_SAMPLES, FS)
    filt = orc.OFFilter(tmpl, psd, FS, pre)
    ft = build_filter(tmpl, psd, FS, pre)
    nsamp = 256
    traces, _, _ = synth.make_traces(nsamp, tmpl, psd, FS, ft.ampres, seed=123)
    orc.process_events(filt, traces[:8], "unconstrained")          # warm-up
    done, t0 = 0, time.perf_counter()
    while True:
        orc.process_events(filt, traces, "unconstrained")
        done += nsamp
        el = time.perf_counter() - t0
        if el >= seconds_target:
            break
    return {"value": done / el, "unit": "traces/s", "cores": 1, "kind": "port",
            "sample": f"{done} traces x {N_SAMPLES} samples, per-event loop of "
                      f"oracle/of1x1.py (fp64 NumPy FFTs, 1 thread), {el:.1f} s"}


def metric_name():
    """BASELINE.json's metric string (the file travels with the repository)."""
    fallback = ("traces/sec of1x1_unconstrained (32768-sample, 1ch) @1/2/4/8 GPU; "
                "% HBM roofline")
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")) as fh:
            return json.load(fh).get("metric", fallback)
    except (OSError, ValueError):
        return fallback


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--traces", type=int, default=1 << 20,
                    help="events resident per GPU (default 1M = BASELINE configs[1])")
    ap.add_argument("--engine", default="auto", choices=["auto", "fused", "rocfft", "lds"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from detprocess_amd import OFPlan, build_filter, synth, synth_traces

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
    B = args.traces
    # cap the resident batch by free HBM (input + output + slack)
    free, _ = torch.cuda.mem_get_info(dev)
    maxB = int((free - (6 << 30)) // (N_SAMPLES * 4 + 64))
    if B > maxB:
        B = maxB
    if world > 1:       # every rank must hold the same number of events (weak scaling, all-gather)
        tb = torch.tensor([B], dtype=torch.int64, device=dev)
        dist.all_reduce(tb, op=dist.ReduceOp.MIN)
        B = int(tb.item())
    traces = torch.empty((B, N_SAMPLES), dtype=torch.float32, device=dev)
    chunk = 1 << 16
    for b0 in range(0, B, chunk):     # counter-based: (seed, global event index)
        nb = min(chunk, B - b0)
        # v = A roll(template, d) + coloured noise drawn from J; A log-uniform
        # in [3, 300] sigma_A for half of the events, d uniform in [-2000, 2000]
        synth_traces(nb, N_SAMPLES, tmpl, 0.0, 3 * ft.ampres, 300 * ft.ampres, 0.5,
                     2000, seed=2026, first_index=rank * B + b0, device=local_rank,
                     out=traces[b0:b0 + nb], return_truth=False, psd=psd, fs=FS)
    plan = OFPlan(N_SAMPLES, pre, FS, max_batch=8192, device=local_rank,
                  engine=args.engine)
    plan.set_filter(0, ft)
    plan.add_search(0, "delay")                       # of1x1_unconstrained
    out = torch.empty((B, plan.row_floats), dtype=torch.float32, device=dev)
    gathered = torch.empty((world * B, plan.row_floats), dtype=torch.float32,
                           device=dev) if world > 1 else None

    def step():
        plan.process(traces, out=out)
        if world > 1:
            dist.all_gather_into_tensor(gathered, out)   # the path's one collective

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    plan.enable_timing(True)
    plan.kernel_time()
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
    k_ms, k_n = plan.kernel_time()

    if rank == 0:
        value = world * B * args.steps / el
        launches_per_step = max(1, k_n // max(1, args.steps))
        traces_per_launch = B / launches_per_step
        ach = traces_per_launch * ALGO_BYTES_PER_TRACE / (k_ms * 1e-3) / 1e9 if k_ms else 0.0
        rec = {
            "metric": metric_name(),
            "value": value, "unit": "traces/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{B} traces/GPU x {N_SAMPLES} samples, 1 channel, "
                                   f"1 template, of1x1_unconstrained (BASELINE configs[1])",
                       "engine": plan.engine, "traces_per_gpu": B,
                       "n_samples": N_SAMPLES, "fs": FS,
                       "parallelism": f"event-range shards x{world}, all-gather of features"},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": measured_traffic(plan.engine, traces_per_launch),
                         "kernel_ms": k_ms, "launches": k_n,
                         "algorithmic_bytes_per_trace": ALGO_BYTES_PER_TRACE},
        }
        if not args.no_cpu_baseline and world == 1:
            rec["cpu_baseline"] = cpu_baseline()
        else:
            rec["cpu_baseline"] = None
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
