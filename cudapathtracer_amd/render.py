"""Render a picture on N GPUs of one node, one process per GPU:

    python -m cudapathtracer_amd.render --gpus N <spp> <maxDepth> <threads> <mode> [--scene ...] [--out file.ppm]

The positional arguments are those of the reference's gpu_path_tracer CLI
(gpu_path_tracer/main.cpp:16-27: spp 32, maxDepth 10, threads 256, mode 1 = single kernel /
2 = streamed).  With --gpus N > 1 the module starts itself N times under
torch.distributed.run (as a child process); every rank uploads the scene to its own GPU, traces
its contiguous share of the samples into an int64 buffer (sharding.render_sharded) and ONE
reduce -- RCCL over xGMI with the default backend "nccl" -- sums the buffers onto rank 0, which
resolves them and writes the PPM with the reference's tone curve.  Because samples are keyed by
(pixel, sample index) and the sums are integers, the file is byte-identical for every N.

--backend gloo --device 0 puts all ranks on one GPU and reduces through host memory: a rehearsal
of the same flow for machines with a single GPU (and what the test-suite runs).
"""
import argparse
import os
import subprocess
import sys
import time


def build_scene(hp, name, width, height, depth):
    return hp.Scene.CreateNamedScene(name, depth, width, height)


def parse(argv):
    ap = argparse.ArgumentParser(prog="python -m cudapathtracer_amd.render")
    ap.add_argument("spp", nargs="?", type=int, default=32)
    ap.add_argument("maxDepth", nargs="?", type=int, default=10)
    ap.add_argument("threads", nargs="?", type=int, default=256)
    ap.add_argument("mode", nargs="?", type=int, default=1)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--scene", default="cornell", help="cornell | bunny | glass | heightfield:<n> | file:<scene file>")
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--out", default=None, help="PPM file (default out/gpu/<scene name>.ppm, like the reference)")
    ap.add_argument("--pfm", default=None, help="also write the linear radiance as a PFM file")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--device", type=int, default=None, help="GPU of every rank (default: LOCAL_RANK)")
    ap.add_argument("--master-port", type=int, default=29533)
    return ap.parse_args(argv)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        # start the ranks as child processes; this process never touches the GPU
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(args.master_port), "-m", "cudapathtracer_amd.render"] + argv
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        return subprocess.call(cmd, env=env)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist
    import cudapathtracer_amd as hp
    from .sharding import render_sharded

    if not torch.cuda.is_available():
        raise SystemExit("cudapathtracer_amd.render needs a GPU (there is no CPU fallback)")
    dev_index = local_rank if args.device is None else args.device
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    scene = build_scene(hp, args.scene, args.width, args.height, args.maxDepth)
    if rank == 0:
        print("Resolution: %dx%d\nSPP: %d\nTrace Depth: %d\nHIP Mode: %s\nHIP #threads: %d\nHIP #GPUs: %d" % (
            args.width, args.height, args.spp, args.maxDepth, "SingleKernel" if args.mode == 1 else "Streamed", args.threads, world),
            flush=True)
    ctx = hp.Context(dev_index).upload(scene)
    accum = torch.zeros(3 * args.width * args.height, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()
    ctx.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()

    def trace(begin, count):
        ctx.render(args.mode, args.threads, args.maxDepth, begin, count, accum=accum, stream=stream)
        ctx.synchronize()   # before the reduce reads the buffer (a collective may run on another stream)

    render_sharded(trace, accum, args.spp, rank, world, dist=dist if world > 1 else None, dst=0)
    st = ctx.stats()
    rays = torch.tensor([float(st["closest_queries"] + st["shadow_queries"])], dtype=torch.float64)
    if world > 1:   # the rays of all ranks (RCCL reduces device tensors, gloo host tensors; both in place)
        if args.backend == "nccl":
            on_dev = rays.to(dev)
            dist.all_reduce(on_dev)
            rays = on_dev.cpu()
        else:
            dist.all_reduce(rays)
    if rank == 0:
        img = ctx.resolve(max(args.spp, 1), accum=accum, stream=stream)
        dt = time.perf_counter() - t0
        print("Render complete: \nTime taken: %.3f seconds" % dt)
        out = args.out or os.path.join("out", "gpu", scene.name + ".ppm")
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        hp.SavePPM(out, img, args.width, args.height)
        if args.pfm:
            hp.SavePFM(args.pfm, img, args.width, args.height)
        print("%s: %d spp on %d GPU(s), %.1f Mrays/s (the rays of all ranks over rank 0's clock)" % (out, args.spp, world, float(rays.item()) / dt * 1e-6))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
