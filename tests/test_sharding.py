"""Samples-per-pixel sharding: partition arithmetic, and the N > 1 data path
(world_size 2, gloo, CPU) -- each rank accumulates its own sample range into an
int64 buffer, ONE reduce sums them onto rank 0, and the result equals the
single-rank render bit for bit.  On the CPU the per-rank tracer is the oracle
(test infrastructure); on the GPU box the same function is driven by the HIP
renderer (test_gpu_sharding below)."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_sample_range_partitions():
    from cudapathtracer_amd.sharding import sample_range
    for spp in (0, 1, 7, 8, 4096, 4099):
        for world in (1, 2, 3, 4, 8):
            got = [sample_range(spp, r, world) for r in range(world)]
            assert got[0][0] == 0 and sum(c for _, c in got) == spp
            for (b0, c0), (b1, _) in zip(got, got[1:]):
                assert b0 + c0 == b1
            assert max(c for _, c in got) - min(c for _, c in got) <= 1
    with pytest.raises(ValueError):
        sample_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, depth, spp, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    from cudapathtracer_amd.sharding import render_sharded
    from orclib import RNG_PHILOX, TRIG_PORTABLE, Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    osc = Oracle().cornell(w, h)
    accum = torch.zeros(3 * w * h, dtype=torch.int64)

    def trace(begin, count):
        a, _ = osc.render(RNG_PHILOX, TRIG_PORTABLE, depth, count, sample_begin=begin, threads=2, fixed=True)
        accum.add_(torch.from_numpy(a.reshape(-1)))

    render_sharded(trace, accum, spp, rank, world, dist=dist, dst=0)
    if rank == 0:
        np.save(out_path, accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("spp", [5, 4])
def test_two_ranks_gloo_equal_one_rank(oracle, tmp_path, spp):
    import torch.multiprocessing as mp
    from orclib import RNG_PHILOX, TRIG_PORTABLE
    w, h, depth = 24, 16, 6
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(2, _free_port(), w, h, depth, spp, out), nprocs=2, join=True)
    got = np.load(out).reshape(-1, 3)
    want, _ = oracle.cornell(w, h).render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_gpu_sharding_single_device(hp, oracle, gpu_ctx):
    """The shard logic on one GPU: ranks traced one after the other into separate
    buffers and summed -- equals the unsharded render and the oracle."""
    import torch
    from cudapathtracer_amd.sharding import render_sharded
    from orclib import RNG_PHILOX, TRIG_PORTABLE
    w, h, depth, spp = 48, 32, 8, 7
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
    gpu_ctx.upload(sc)
    stream = torch.cuda.current_stream()
    total = torch.zeros(3 * w * h, dtype=torch.int64, device="cuda:0")
    for world in (1, 3):
        total.zero_()
        for rank in range(world):
            acc = torch.zeros_like(total)
            mode = 1 + (rank % 2)
            render_sharded(lambda b, n: gpu_ctx.render(mode, 256, depth, b, n, accum=acc, stream=stream), acc, spp, rank, world)
            total += acc
        torch.cuda.synchronize()
        want, _ = oracle.cornell(w, h).render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
        assert np.array_equal(total.cpu().numpy().reshape(-1, 3), want)
