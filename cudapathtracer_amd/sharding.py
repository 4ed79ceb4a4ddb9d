"""Samples-per-pixel sharding across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).  The
samples of a render are split into contiguous ranges, one per rank; every rank
traces its range into its own int64 accumulation buffer and the buffers are
summed onto rank 0 with ONE reduce per render.  Because a path's random numbers
are keyed by (pixel, sample index) and the accumulators are integers, the
reduced image is bit-identical for every world size.

The reference has no multi-GPU path (one Scene* on the default device,
gpu_path_tracer/CudaRenderer.hpp:18); this is the new core's own.
"""


def sample_range(spp, rank, world_size):
    """Contiguous share of samples [begin, begin+count) of ``rank``; the first
    ``spp % world_size`` ranks take one extra sample."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(int(spp), world_size)
    begin = rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


def reduce_accum(accum, dist, dst=0):
    """ONE reduce(SUM) of the int64 accumulators onto ``dst``.  backend nccl (= RCCL): on the GPU
    tensor, over xGMI.  backend gloo (CPU tests; rehearsals of several ranks on one GPU): through a
    host copy, the result written back into ``accum`` on every rank (only ``dst`` holds the sum)."""
    if dist.get_backend() == "nccl" or not accum.is_cuda:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    else:
        host = accum.cpu()
        dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
        accum.copy_(host)


def render_sharded(trace, accum, spp, rank, world_size, dist=None, dst=0):
    """Trace this rank's share of ``spp`` samples and sum all shares onto ``dst``.

    trace(sample_begin, sample_count): adds the given samples of every pixel to
        ``accum`` (e.g. ``lambda b, n: ctx.render(mode, block, depth, b, n, accum=accum)``)
    accum: torch int64 tensor (on the rank's GPU for nccl, on the CPU for gloo), zeroed
    dist:  torch.distributed (None or world_size 1: no collective)
    Returns (sample_begin, sample_count) of this rank.  After the call ``accum``
    on rank ``dst`` holds the sum over all ranks.
    """
    begin, count = sample_range(spp, rank, world_size)
    if count:
        trace(begin, count)
    if dist is not None and world_size > 1:
        reduce_accum(accum, dist, dst)
    return begin, count
