"""N > 1 path on CPU: world_size-2 gloo.  Event ranges are sharded with no
data-path collective; the only exchange is the final all-gather of the feature
matrix.  The per-shard compute is stood in by the oracle
here (tests may use it); the sharding / chunking / gather code is the product's:
``detprocess_amd.dist.run_sharded``, the same function ``bench.py --gpus N`` drives."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from detprocess_amd import dist as ofdist


def test_shard_ranges_cover_everything_once():
    for total in (0, 1, 7, 1000, 12_500_001):
        for world in (1, 2, 3, 8):
            spans = [ofdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a0 <= a1
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(total):
    from detprocess_amd import synth
    from oracle import of1x1 as orc
    n, fs, pre = 1024, 1.25e6, 512
    tmpl = synth.make_template(n, pre, fs)
    psd = synth.make_psd(n, fs)
    filt = orc.OFFilter(tmpl, psd, fs, pre)
    traces, _, _ = synth.make_traces(total, tmpl, psd, fs, filt.ampres, seed=1, max_delay=100)

    def source(lo, hi, buf):                       # events keyed by their global index
        buf[: hi - lo] = torch.as_tensor(traces[lo:hi])

    def process(events, out):
        r = orc.process_events(filt, events.numpy(), "unconstrained")
        out[:] = torch.tensor(np.stack([r["amp"], r["t0"], r["chi2"], r["lowchi2"]], axis=1))

    return n, traces, source, process


def _worker(rank, world, port, total, chunk, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, traces, source, process = _setup(total)
    lo, hi = ofdist.shard_range(total, rank, world)
    mk = lambda: torch.empty((hi - lo, 4), dtype=torch.float64)     # (fp64: the oracle's rows)
    full = ofdist.run_sharded(total, chunk, source, process, 4, (n,), rank=rank, world=world,
                              dtype=torch.float64, out=mk())
    # the resident form of the same call (this rank's shard already in memory)
    full2 = ofdist.run_sharded(total, chunk, torch.as_tensor(traces[lo:hi]), process, 4, (n,),
                               rank=rank, world=world, out=mk())
    assert torch.equal(full, full2)
    if rank == 0:
        q.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_chunk_spans():
    assert ofdist.chunk_spans(3, 3, 5) == []
    assert ofdist.chunk_spans(3, 14, 5) == [(3, 8), (8, 13), (13, 14)]
    assert ofdist.chunk_spans(0, 10, 10) == [(0, 10)]


@pytest.mark.parametrize("total,chunk", [(37, 7), (64, 32), (5, 100)])
def test_two_rank_gloo_equals_single_process(total, chunk):
    from detprocess_amd import synth
    from oracle import of1x1 as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, chunk, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, fs, pre = 1024, 1.25e6, 512
    tmpl = synth.make_template(n, pre, fs)
    psd = synth.make_psd(n, fs)
    filt = orc.OFFilter(tmpl, psd, fs, pre)
    traces, _, _ = synth.make_traces(total, tmpl, psd, fs, filt.ampres, seed=1, max_delay=100)
    r = orc.process_events(filt, traces, "unconstrained")
    want = np.stack([r["amp"], r["t0"], r["chi2"], r["lowchi2"]], axis=1)
    assert np.array_equal(got, want)          # pure sharding: bit-identical


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts its two ranks under
    torch.distributed.run as a child process and passes their exit code on: on a box without a
    GPU the ranks stop at "needs a GPU", never at "needs torch.distributed.run"."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("the CPU form of this check")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline"], env=env, capture_output=True,
                       text=True, timeout=300)
    text = r.stdout + r.stderr
    assert r.returncode != 0
    assert text.count("bench.py needs a GPU") >= 1, text[-2000:]   # (the launcher stops the other rank)
    assert "needs torch.distributed.run" not in text
