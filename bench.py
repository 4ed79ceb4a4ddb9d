#!/usr/bin/env python3
"""Headline benchmark: Cornell box 1024x1024, depth 20, single-kernel mode.

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch: --spp-per-step samples of
every pixel of the 1k x 1k image (default 1024 -> 1.07 G camera paths, about
12.5 G rays, one kernel launch; the 4096-spp configuration is four such steps).
With N > 1 (one process per GPU, launched by torch.distributed.run)
every rank traces its own, disjoint sample ranges (weak scaling: per-GPU work
per step is fixed) and the int64 accumulation buffers are summed onto rank 0
with ONE RCCL reduce at the end of the timed region, which also contains the
final fixed-point -> float resolve on rank 0.

--scaling strong: the job is --total-spp samples per pixel (default 4096: the
reference README's picture), split evenly over the steps and, inside a step,
over the ranks; same timed region, "scaling": "strong".  The default (weak) line
carries the same measurement as a side leg, `strong_config`: the 4096-spp Cornell
render sharded over the N ranks + one reduce, timed after the weak region.

Rank 0 prints one JSON line.  `value` = Mrays/s over the whole job, rays
counted by device counters (ray = one BVH query, closest-hit or shadow).
The scene is uploaded and resident in HBM before the timed region starts.
"""
import argparse
import json
import os
import sys
import time

# Before anything can initialise the GPU runtime (ROCr reads its settings once, when HSA starts: `import torch`
# may already do that): the host driver of this pool only supports dmabuf IPC, and RCCL's cross-process buffer
# sharing fails with `hipIpcGetMemHandle: invalid argument` when the legacy IPC mode is left on.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# dense FP32 vector peak of MI355X (157.3 TFLOP/s counts an FMA as 2 flops); HBM3E peak; aggregate
# L2 bandwidth of the eight XCDs (MI355X_MICROARCH.md: 8 TB/s and 34.5 TB/s)
PEAK_FP32_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0
L2_PEAK_GBS = 34500.0
L2_BYTES_PER_XCD = 4 << 20
# algorithmic work per unit: one box test = 24 flops (6 sub, 6 mul, ~12 compare/select) and one
# node record, priced at 64 B like a triangle record; one triangle test = 70 flops / 64 B; one
# shaded bounce = 350 flops.  Tests are counted as the reference traversal defines them, by the
# CPU restatement on a bounded sample of the benchmarked scene, not by what the GPU kernel
# happens to execute (that is reported beside it, from profiles/executed_tests.json).
FLOPS_BOX, FLOPS_TRI, FLOPS_BOUNCE = 24.0, 70.0, 350.0
BYTES_BOX, BYTES_TRI = 64.0, 64.0


def scene_parts(name):
    """(name, triangles, material) lists of the non-Cornell bench scenes (pure numpy)."""
    from cudapathtracer_amd import scenes
    if name == "bunny":
        return scenes.bunny_in_cornell_parts()
    if name == "glass":
        return scenes.glass_cornell_parts()
    if name.startswith("heightfield:"):
        return scenes.heightfield_in_cornell_parts(int(name.split(":")[1]))
    raise SystemExit("unknown --scene " + name)


def host_cpu():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count() or 1


def cpu_baseline(scene, width, height, depth, spp):
    """The CPU restatement of the reference renderer (its own RNG stream and libm), timed on the
    host cores on a bounded sample of the same workload: the benchmarked scene at the benchmark's
    depth, full resolution for the Cornell box, 512 x 512 for the large scenes (whose CPU rays are
    several times slower).  Also returns the per-ray box/triangle test counts of the REFERENCE
    traversal on that scene, which the roofline is priced with."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from orclib import RNG_MT, TRIG_LIBM, Material, Oracle
    model, ncpu = host_cpu()
    # the GPU box gives a job 16 of the machine's CPUs (OMP_NUM_THREADS is set to that share there)
    threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or ncpu, ncpu))
    if scene == "cornell":
        osc = Oracle().cornell(width, height)
    else:
        from cudapathtracer_amd import scenes
        width = height = 512
        osc = Oracle().scene(width, height, scenes.CORNELL_CAM, 40.0)
        for _, tris, mat in scene_parts(scene):
            osc.add_triangles(tris, Material.make(**mat))
        osc.build()
    t0 = time.perf_counter()
    _, st = osc.render(RNG_MT, TRIG_LIBM, depth, spp, threads=threads)
    dt = time.perf_counter() - t0
    rays = st["closest_queries"] + st["shadow_queries"]
    return {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "cpu_model": model, "host_cpus": ncpu,
        "threads_note": "OpenMP threads = OMP_NUM_THREADS (the job's CPU share on the GPU box) capped at the host's CPU count",
        "sample": "%s %dx%d depth %d, %d spp (%.1f M rays) in %.2f s; %.3f spp/s at this resolution" % (
            scene, width, height, depth, spp, rays / 1e6, dt, spp / dt),
        "spp_per_sec": spp / dt,
        "relation_to_reference": "the restatement renders the reference's image bit for bit and runs 1.09-1.15x as fast as the "
                                 "compiled reference on the same 8 cores (512x512 16 spp depth 5: 7.50 s vs 8.66 s; 1024x1024 16 spp depth 20: 38.9 s vs 42.5 s)",
    }, {"box_per_ray": st["box_tests"] / rays, "tri_per_ray": st["tri_tests"] / rays,
        "rays_per_path": rays / st["paths"]}


def load_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}


def kernel_name(mode, variant, extended_hint=""):
    """The kernel a render of this mode launches for a scene of this traversal variant (hpt_core.hip: the
    two-child-record walks through L1/L2, variants 5 and 7, run under the wave scheduler unless it is switched off)."""
    def off(name):
        return os.environ.get(name, "") not in ("", "1")
    scheduler = variant in (5, 7) and not off("HPT_SCHEDULER") and not (mode == 2 and off("HPT_WF_SCHEDULER"))
    mask = os.environ.get("HPT_MEGA_SCHEDULER_VARIANTS" if mode == 1 else "HPT_WF_SCHEDULER_VARIANTS")
    if mask is not None:
        scheduler = variant in (2, 3, 5, 6, 7) and bool((int(mask, 0) >> variant) & 1) and not off("HPT_SCHEDULER")
    if mode == 1:
        return "k_render_mega_sm" if scheduler else "k_render_mega"
    return ("k_wf_bounces_sm" if scheduler else "k_wf_bounces") + " (+ k_wf_generate)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-step", type=int, default=1024,
                    help="samples per pixel per step and GPU (weak scaling); 1024 = one launch of 2^30 paths, about 0.4 s on the Cornell box")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every GPU traces --spp-per-step samples per step; strong: --total-spp samples per pixel "
                         "are split over the steps and, inside a step, over the GPUs")
    ap.add_argument("--total-spp", type=int, default=4096, help="samples per pixel of the whole job with --scaling strong, and of the strong_config leg")
    ap.add_argument("--no-strong-leg", action="store_true", help="skip the strong_config side leg (the --total-spp render sharded over the ranks)")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=20)
    ap.add_argument("--mode", type=int, default=1, help="1 single kernel, 2 streamed (wavefront)")
    ap.add_argument("--block", type=int, default=256, help="workgroup size (the CLI's <threads>; 256 is the benchmark configuration's)")
    ap.add_argument("--cpu-spp", type=int, default=32, help="samples of the CPU baseline leg (0 = skip); 32 = 14 s of the GPU box's 16 host threads on the Cornell box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traversal", default="reference", choices=["reference", "culled"],
                    help="reference = the reference's box/triangle tests exactly (the parity path); culled = near-first "
                         "order with distance culling for scenes of more than 64 triangles (opt-in, not bit-guaranteed)")
    ap.add_argument("--bvh-build", default="default", choices=["default", "host", "gpu", "sah"],
                    help="default: what the host library builds by itself; host: the reference's tree; gpu: traversal records built on the "
                         "device at upload; sah: host build with surface-area splits (scenes other than cornell)")
    ap.add_argument("--scene", default="cornell", help="cornell (the headline workload) | bunny | heightfield:<n> "
                    "(side measurements of the other configurations; they never change the default line)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist
    import cudapathtracer_amd as hp
    from cudapathtracer_amd.sharding import reduce_accum, sample_range

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU fallback)")
    # HPT_BENCH_BACKEND=gloo + HPT_BENCH_DEVICE=0: rehearse the multi-rank flow on a one-GPU box
    backend = os.environ.get("HPT_BENCH_BACKEND", "nccl")
    dev_index = int(os.environ.get("HPT_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # HPT_BENCH_FORCE_DIST=1: create the process group and run the reduce even with one rank, so that a
    # one-GPU box exercises the RCCL path (communicator, int64 reduce on device memory) that N ranks take
    use_dist = world > 1 or os.environ.get("HPT_BENCH_FORCE_DIST", "0") not in ("", "0")
    json_fd = 1
    if use_dist:
        # RCCL writes a version banner to stdout when its communicator is created; the one line this
        # program owes its caller goes to the real stdout, everything else of the run to stderr
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # (a collective that cannot complete ends the run with an error after three minutes instead of holding the node)
        import datetime
        limit = datetime.timedelta(seconds=int(os.environ.get("HPT_BENCH_COLLECTIVE_TIMEOUT", "180")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)

    def reduce_sum(t):
        reduce_accum(t, dist, 0)

    def all_reduce(t, op):
        if backend == "nccl":
            dist.all_reduce(t, op=op)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)

    W, H, depth, spp_step = args.width, args.height, args.depth, args.spp_per_step
    strong = args.scaling == "strong"
    if strong and args.total_spp < args.steps * world:
        sys.exit("--scaling strong: --total-spp %d gives some rank of some step no sample (%d steps x %d GPUs)" % (args.total_spp, args.steps, world))
    if args.scene == "cornell":
        scene = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, W, H)
    else:
        from cudapathtracer_amd import scenes
        scene = scenes._assemble(scene_parts(args.scene), W, H, depth)
        if args.bvh_build == "gpu":
            scene.buildBVH(gpu=True)
        elif args.bvh_build == "sah":
            scene.buildBVH(sah=True)
        elif args.bvh_build == "host":
            scene.buildBVH(gpu=False)
    bvh_built = {hp.BVH_BUILD_HOST: "host", hp.BVH_BUILD_GPU: "gpu", hp.BVH_BUILD_SAH: "sah"}.get(scene.bvh_build, "host")
    ctx = hp.Context(dev_index)
    if args.traversal == "culled":
        ctx.set_traversal(1)
    ctx.upload(scene)
    accum = torch.zeros(3 * W * H, dtype=torch.int64, device=dev)
    rgb = torch.zeros(3 * W * H, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()

    def barrier():
        ctx.synchronize()            # the HIP runtime the kernel library launches through
        torch.cuda.synchronize()     # torch's (the same one when torch was loaded first)
        if use_dist:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    # what every rank traces in step k: (first sample, number of samples)
    #   weak:   step k of rank r = samples [(k*world + r)*spp_step, +spp_step)
    #   strong: the job's total_spp samples split evenly over the steps, a step's share evenly over the ranks
    def step_range(k, total=None):
        if total is None:
            return (k * world + rank) * spp_step, spp_step
        b0, n0 = sample_range(total, k, args.steps)
        b1, n1 = sample_range(n0, rank, world)
        return b0 + b1, n1

    total_spp = args.total_spp if strong else args.steps * world * spp_step
    warm_base = max(total_spp, args.total_spp)  # warm-up samples lie outside the timed ranges
    warm_spp = spp_step if not strong else max(1, args.total_spp // (args.steps * world))
    if use_dist:  # warm up the communicator as well
        reduce_sum(accum)
    for k in range(args.warmup):
        ctx.render(args.mode, args.block, depth, warm_base + (k * world + rank) * warm_spp, warm_spp, accum=accum, stream=stream)
    barrier()
    accum.zero_()
    ctx.stats_reset()

    def timed_region(total):
        """The K steps, the reduce and the resolve between two barriers; returns (seconds, reduce ms of this rank, wall-clock start, stop)."""
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        wall0 = time.time()
        t0 = time.perf_counter()
        for k in range(args.steps):
            b, n = step_range(k, total)
            if n:
                ctx.render(args.mode, args.block, depth, b, n, accum=accum, stream=stream)
        reduce_ms = None
        if use_dist:
            # the collective, seen from this rank's stream: from the end of its own renders to the reduced buffer
            ev0.record(stream)
            reduce_sum(accum)
            ev1.record(stream)
        if rank == 0:
            ctx.resolve_dev(total if total is not None else total_spp, rgb, accum=accum, stream=stream)
        barrier()
        dt = time.perf_counter() - t0
        wall1 = time.time()
        if use_dist:
            reduce_ms = float(ev0.elapsed_time(ev1))
        return dt, reduce_ms, wall0, wall1

    elapsed, reduce_ms, wall0, wall1 = timed_region(args.total_spp if strong else None)

    st = ctx.stats()
    rays = st["closest_queries"] + st["shadow_queries"]

    def whole_job(st_, dt):
        """(rays, paths, closest queries) summed and the time maximised over the ranks."""
        agg = torch.tensor([float(st_["closest_queries"] + st_["shadow_queries"]), float(st_["paths"]), float(st_["closest_queries"])],
                           dtype=torch.float64, device=dev)
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if use_dist:
            all_reduce(agg, dist.ReduceOp.SUM)
            all_reduce(tmax, dist.ReduceOp.MAX)
        return [float(v) for v in agg.tolist()] + [float(tmax.item())]

    total_rays, total_paths, total_closest, elapsed = whole_job(st, elapsed)

    # who took part: every rank's device, as torch.distributed saw the job
    me = {"rank": rank, "device_index": dev_index, "device_name": torch.cuda.get_device_name(dev_index),
          "pci_bus_id": getattr(torch.cuda.get_device_properties(dev_index), "pci_bus_id", None), "reduce_ms": reduce_ms}
    ranks = [me]
    if use_dist:
        ranks = [None] * dist.get_world_size()
        dist.all_gather_object(ranks, me)

    img_ok = True
    if rank == 0:
        img = rgb.view(-1, 3)
        mean = [float(v) for v in img.mean(dim=0).tolist()]
        img_ok = bool(torch.isfinite(img).all().item()) and mean[0] > 0.02 and int(total_paths) == W * H * total_spp
        if not img_ok:
            sys.stderr.write("bench.py: implausible render (mean %r, paths %d, expected %d)\n" % (mean, int(total_paths), W * H * total_spp))

    # Side leg (default weak line only): BASELINE's sharded configuration as it is stated -- --total-spp (4096) samples per
    # pixel of the benchmarked scene split over the ranks, ONE reduce, resolve on rank 0 -- timed like the region above.
    strong_leg = None
    if not strong and not args.no_strong_leg and args.total_spp >= args.steps * world:
        accum.zero_()
        ctx.stats_reset()
        s_dt, s_reduce_ms, s_w0, s_w1 = timed_region(args.total_spp)
        s_st = ctx.stats()
        s_rays, s_paths, _, s_dt = whole_job(s_st, s_dt)
        strong_leg = {"scaling": "strong", "total_spp": args.total_spp, "n_gpus": world, "seconds": s_dt,
                      "value": s_rays / s_dt / 1e6, "unit": "Mrays/s", "spp_per_sec": args.total_spp / s_dt,
                      "reduce_ms_rank0": s_reduce_ms, "paths": int(s_paths), "paths_expected": W * H * args.total_spp,
                      "wall_clock_start": s_w0, "wall_clock_stop": s_w1,
                      "note": "the same job for every N: %d spp of %s %dx%d depth %d sharded over the ranks, one reduce, resolve; "
                              "time = max over ranks between two barriers" % (args.total_spp, args.scene, W, H, depth)}
        if rank == 0 and int(s_paths) != W * H * args.total_spp:
            img_ok = False
            sys.stderr.write("bench.py: the strong leg traced %d paths, expected %d\n" % (int(s_paths), W * H * args.total_spp))

    if rank == 0 and img_ok:
        cpu, per_ray = None, None
        if world == 1 and not args.no_cpu_baseline and args.cpu_spp > 0:
            cpu, per_ray = cpu_baseline(args.scene, W, H, depth, args.cpu_spp)

        launches = max(1, st["launches"])
        if args.mode == 2:
            launches //= 2                                # a streamed batch = generate + bounce loop: priced as one unit
        launches = max(1, launches)
        kernel_ms = st["kernel_ms"] / launches            # rank 0's average launch, HIP events on its stream
        rays_per_launch = rays / launches
        closest_per_launch = st["closest_queries"] / launches
        # where the traversal's records live decides the bound: LDS (staged per workgroup) -> VALU issue;
        # read through L1/L2 -> L2 bandwidth while the records fit one XCD's L2, HBM beyond
        variant = st["variant"]
        desc = scene.describe()
        record_bytes = 64 * (desc.num_tris + desc.num_nodes // 2)    # triangle records + two-child records
        # (the 190 MB of the 1.0 M-triangle scene do not fit the L2s, but what crosses the fabric -- `traffic` --
        # stays under half of the HBM peak and half of the lookups still hit L2: the L2 request
        # path is the nearer bound there too; hbm_traffic_GBps / hbm_traffic_frac state the HBM side)
        bound = "valu" if variant in (1, 2, 3, 6) else "l2"
        # the PMC files are keyed by the samples ONE launch traces (a launch is at most 2^28 camera paths)
        tiles = ((W + 7) // 8) * ((H + 7) // 8)
        per_rank_step = spp_step if not strong else max(1, args.total_spp // (args.steps * world))
        spp_launch = max(1, min(per_rank_step, ((1 << 24) if args.mode == 1 else (1 << 22)) // tiles))   # single kernel: 2^30 paths; streamed: batches of 2^28
        # (the default tree -- surface-area splits for the large scenes -- has no suffix; --bvh-build host / gpu / sah name theirs)
        tree = "" if args.bvh_build == "default" or args.scene == "cornell" else "_" + ("reftree" if args.bvh_build == "host" else args.bvh_build)
        key = "%s_mode%d_%dx%d_d%d_spp%d" % (args.scene, args.mode, W, H, depth, spp_launch) + ("_culled" if args.traversal == "culled" else "") + tree
        pm = load_json("hbm_traffic.json").get(key, {})
        ex = load_json("executed_tests.json").get("%s_%dx%d_d%d_mode%d" % (args.scene, W, H, depth, args.mode)
                                                  + ("_culled" if args.traversal == "culled" else "") + tree, {})
        roof = {"bound": bound, "achieved": None, "peak": None, "unit": None, "frac": None,
                "traffic": pm.get("hbm_bytes_per_launch"),
                "kernel": kernel_name(args.mode, variant),
                "kernel_ms": kernel_ms, "launches": launches, "spp_per_launch": spp_launch, "traversal_variant": variant,
                "scene_record_bytes": record_bytes,
                "valu_active_lane_fraction": pm.get("valu_active_lane_fraction"),   # PMC, same source file as `traffic`
                "l2_hit_rate": pm.get("l2_hit_rate"), "counter_source": pm.get("source"), "counter_key": key,
                "records_fit_one_xcd_l2": record_bytes <= L2_BYTES_PER_XCD}
        if roof["traffic"]:
            roof["hbm_traffic_GBps"] = roof["traffic"] / (kernel_ms * 1e-3) / 1e9
            roof["hbm_traffic_frac"] = roof["hbm_traffic_GBps"] / HBM_PEAK_GBS
        if per_ray is not None:
            flops = rays_per_launch * (FLOPS_BOX * per_ray["box_per_ray"] + FLOPS_TRI * per_ray["tri_per_ray"]) \
                + closest_per_launch * FLOPS_BOUNCE
            byts = rays_per_launch * (BYTES_BOX * per_ray["box_per_ray"] + BYTES_TRI * per_ray["tri_per_ray"])
            tflops = flops / (kernel_ms * 1e-3) / 1e12
            gbps = byts / (kernel_ms * 1e-3) / 1e9
            if bound == "valu":
                roof.update(achieved=tflops, peak=PEAK_FP32_TFLOPS, unit="TFLOP/s", frac=tflops / PEAK_FP32_TFLOPS)
            else:
                roof.update(achieved=gbps, peak=L2_PEAK_GBS, unit="GB/s", frac=gbps / L2_PEAK_GBS)
            roof.update({
                "algorithmic_flops_per_launch": flops, "algorithmic_TFLOPs": tflops,
                "algorithmic_bytes_per_launch": byts, "algorithmic_record_GBps": gbps,
                "box_tests_per_ray": per_ray["box_per_ray"], "tri_tests_per_ray": per_ray["tri_per_ray"],
                "note": "algorithmic work = box/triangle tests as the REFERENCE traversal defines them on this scene (counted by the "
                        "CPU oracle on the cpu_baseline sample): 24 flops and one 64-byte record per box test, 70 flops and one record "
                        "per triangle test, 350 flops per shaded bounce. " +
                        ("The records are staged in LDS per workgroup, so the record byte rate is an LDS rate, not HBM traffic."
                         if bound == "valu" else
                         "The records are read through L1/L2" + ("; the scene fits one XCD's L2, HBM traffic is nil." if record_bytes <= L2_BYTES_PER_XCD else
                                                                 "; the scene is larger than the L2s: `traffic` is what crossed the fabric.")),
            })
            if ex:
                eflops = rays_per_launch * (FLOPS_BOX * ex["executed_box_tests_per_ray"] + FLOPS_TRI * ex["executed_tri_tests_per_ray"]) \
                    + closest_per_launch * FLOPS_BOUNCE
                ebytes = rays_per_launch * (0.5 * BYTES_BOX * ex["executed_box_tests_per_ray"] + BYTES_TRI * ex["executed_tri_tests_per_ray"])
                roof["executed"] = {
                    "box_tests_per_ray": ex["executed_box_tests_per_ray"], "tri_tests_per_ray": ex["executed_tri_tests_per_ray"],
                    "TFLOPs": eflops / (kernel_ms * 1e-3) / 1e12, "frac_of_fp32_peak": eflops / (kernel_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                    "record_GBps": ebytes / (kernel_ms * 1e-3) / 1e9,
                    "note": "tests the kernel executes (diagnostic counting build, profiles/executed_tests.json); a fetched record "
                            "holds two boxes (32 bytes per box test)"}
        scene_text = "Cornell" if args.scene == "cornell" else args.scene
        out = {
            "metric": "Mrays/sec + spp/sec (whole node), %s %s depth %d" % (
                scene_text, "1k^2" if (W, H) == (1024, 1024) else "%dx%d" % (W, H), depth),
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "spp_per_sec": total_spp / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32 (+f64 in the triangle test, int64 accumulators)",
            "data": "synthetic: %s, Philox-keyed samples" % ("built-in Cornell box (models/cornellbox)" if args.scene == "cornell" else args.scene),
            "config": {"workload": ("%s %dx%d, depth %d, mode %d (%s), block %d, " % (
                                        "Cornell box" if args.scene == "cornell" else args.scene, W, H, depth, args.mode,
                                        "single kernel" if args.mode == 1 else "streamed", args.block))
                                   + ("%d spp per step per GPU (%d launches of %d spp); %d spp in the timed region" % (
                                        spp_step, -(-spp_step // spp_launch), spp_launch, total_spp) if not strong else
                                      "%d spp in all, split over %d steps and %d GPUs" % (total_spp, args.steps, world)),
                       "rays_per_path": total_rays / total_paths,
                       "traversal": args.traversal,
                       "bvh_build": bvh_built if args.scene != "cornell" else "host",
                       "gpu_bvh_build_ms": ctx.build_info()["gpu_build_ms"] or None,
                       "parallelism": ("spp-sharded x%d, one %s reduce" % (world, "RCCL" if backend == "nccl" else backend)) if use_dist else "1 GPU",
                       "collective_backend": (dist.get_backend() if use_dist else None),
                       # what torch.distributed itself reports (not the flag, not the environment)
                       "world_size": (dist.get_world_size() if use_dist else 1),
                       "ranks": ranks,
                       "reduce_ms": reduce_ms,
                       "reduce_bytes": (accum.numel() * 8 if use_dist else 0),
                       "wall_clock_start": wall0, "wall_clock_stop": wall1},
            "strong_config": strong_leg,
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if not img_ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
