"""Multi-GPU sharding: one process per GPU, contiguous event ranges, no
data-path collective; the only exchange is the final all-gather of the feature
matrix over RCCL (backend "nccl" on ROCm) -- SURVEY.md section 8e.  The
reference's counterpart is Pool.starmap over series lists followed by
pd.concat (features.py:405-420).

``run_sharded`` is the one driver of a sharded run (BASELINE configs[4]: 100 M events,
12.5 M per GPU, consumed in chunks): the rank's event range is cut into chunks, chunk k+1 is
generated / ingested on a producer stream while the hot path runs chunk k on the compute
stream (two event buffers), and the features are gathered once at the end.  ``bench.py`` and
the gloo test (tests/test_dist.py) both go through it."""

import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Events [lo, hi) owned by `rank`: contiguous, sizes differ by at most 1."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def gather_features(local, total, rank, world, group=None):
    """All-gather the [B_local, F] feature matrices into [total, F] (every rank).

    Shards may differ by one row, so rows are padded to the largest shard for
    the collective and trimmed afterwards.
    """
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local           # (with a process group, world 1 still goes through the collective)
    sizes = [shard_range(total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    F = local.shape[1]
    if all(hi - lo == mx for lo, hi in sizes):           # equal shards: no padding, no trim
        out = torch.empty((world * mx, F), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = torch.zeros((mx, F), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * mx, F), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = [out[r * mx: r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, dim=0)


def chunk_spans(lo, hi, chunk):
    """[lo, hi) cut into consecutive spans of at most `chunk` events."""
    chunk = max(1, int(chunk))
    return [(a, min(a + chunk, hi)) for a in range(lo, hi, chunk)]


def run_sharded(total_events, chunk, source, process, row_floats, event_shape, rank=0, world=1,
                device=None, group=None, gather=True, buffers=None, out=None,
                dtype=torch.float32):
    """Process events [0, total_events) sharded by event range over `world` ranks.

    source(lo, hi, buf)   fills buf[: hi - lo] (a [chunk, *event_shape] tensor on `device`) with
                          the events [lo, hi) of the global run -- a generator keyed by the
                          global event index, or an ingest step.  Called on the producer
                          stream.  ``source`` may instead be a tensor holding this rank's whole
                          shard, resident on `device` ([hi - lo, *event_shape]): chunks are then
                          views of it and nothing is produced.
    process(events, out)  the hot path on `events` ([n, *event_shape]) writing out ([n,
                          row_floats]); called on the compute stream.
    Returns the [total_events, row_floats] feature matrix on every rank (gather=True: the
    path's one collective, at the end) or this rank's [B_local, row_floats] rows.

    On a CUDA device chunk k+1 is produced while chunk k is processed (two buffers, two
    streams, event-ordered); on the CPU (device None: the gloo tests) the same loop runs in
    order.
    """
    lo, hi = shard_range(total_events, rank, world)
    n_local = hi - lo
    dev = torch.device("cpu") if device is None else torch.device(device)
    cuda = dev.type == "cuda"
    if out is None:
        out = torch.empty((n_local, row_floats), dtype=torch.float32, device=dev)
    spans = chunk_spans(lo, hi, chunk)
    resident = isinstance(source, torch.Tensor)
    if resident:
        if source.shape[0] != n_local:
            raise ValueError(f"ERROR: resident shard holds {source.shape[0]} events, rank {rank} "
                             f"owns {n_local}")
        for a, b in spans:
            process(source[a - lo: b - lo], out[a - lo: b - lo])
    elif spans:
        cmax = max(b - a for a, b in spans)
        if buffers is None:
            nbuf = 2 if (cuda and len(spans) > 1) else 1
            buffers = [torch.empty((cmax,) + tuple(event_shape), dtype=dtype, device=dev)
                       for _ in range(nbuf)]
        nbuf = len(buffers)
        if any(bf.shape[0] < cmax for bf in buffers):
            raise ValueError(f"ERROR: event buffers hold {min(bf.shape[0] for bf in buffers)} "
                             f"events, chunks have up to {cmax}")
        if cuda:
            compute = torch.cuda.current_stream(dev)
            producer = torch.cuda.Stream(dev) if nbuf > 1 else compute
            filled = [torch.cuda.Event() for _ in spans]
            freed = [torch.cuda.Event() for _ in spans]
            producer.wait_stream(compute)

            def produce(k):
                a, b = spans[k]
                with torch.cuda.stream(producer):
                    if k >= nbuf:                      # the buffer's previous chunk is done
                        producer.wait_event(freed[k - nbuf])
                    source(a, b, buffers[k % nbuf])
                    filled[k].record(producer)

            for k in range(min(nbuf - 1, len(spans)) if nbuf > 1 else 0):
                produce(k)
            for k, (a, b) in enumerate(spans):
                if nbuf > 1:
                    if k + nbuf - 1 < len(spans):
                        produce(k + nbuf - 1)
                else:
                    produce(k)
                compute.wait_event(filled[k])
                process(buffers[k % nbuf][: b - a], out[a - lo: b - lo])
                freed[k].record(compute)
            compute.wait_stream(producer)
        else:
            for a, b in spans:
                source(a, b, buffers[0])
                process(buffers[0][: b - a], out[a - lo: b - lo])
    if not gather:
        return out
    return gather_features(out, total_events, rank, world, group=group)
