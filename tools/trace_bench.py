#!/usr/bin/env python3
"""Traversal-only microbenchmark: hpt_intersect_batch_dev on device-resident ray batches
(primary grid / random closest / shadow / bounce-like rays), Mrays/s per batch type."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cudapathtracer_amd as hp
from raybatches import primary_rays, random_rays, shadow_rays

sc = hp.Scene.CreateBuiltinScene(0, 20, 1024, 1024)
ctx = hp.Context(0).upload(sc)
f = sc.flat_arrays()
n = 1 << 22
def run(name, o, d, sh=None, tm=None, reps=5):
    o_d = torch.from_numpy(np.ascontiguousarray(o)).cuda(); d_d = torch.from_numpy(np.ascontiguousarray(d)).cuda()
    sh_d = None if sh is None else torch.from_numpy(sh).cuda(); tm_d = None if tm is None else torch.from_numpy(tm).cuda()
    prim = torch.zeros(len(o), dtype=torch.int32, device="cuda"); t = torch.zeros(len(o), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream()
    ctx.intersect_dev(len(o), o_d, d_d, prim, t, sh_d, tm_d, 256, st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): ctx.intersect_dev(len(o), o_d, d_d, prim, t, sh_d, tm_d, 256, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print("%-16s %8.1f Mrays/s  (%.3f ms, hit rate %.3f)" % (name, len(o) / dt / 1e6, dt * 1e3, float((prim >= 0).float().mean())))
o, d = primary_rays(2048, 2048, (278, 273, -800), f["cam_scale"]); run("primary 2048^2", o, d)
o, d = random_rays(n, 1); run("random closest", o, d)
o, d, sh, tm = shadow_rays(n, 2); run("shadow", o, d, sh, tm)
