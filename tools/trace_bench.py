#!/usr/bin/env python3
"""Traversal-only microbenchmark: hpt_intersect_batch_dev on device-resident ray batches
(primary grid / random closest / shadow / the rays camera paths trace at bounce 3, captured from the CPU
restatement), Mrays/s per batch type."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cudapathtracer_amd as hp
from raybatches import primary_rays, random_rays, shadow_rays
from orclib import Oracle, RNG_PHILOX, TRIG_PORTABLE

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
# (iv) incoherent rays as a render produces them: bounce 3 of the camera paths of one sample of a 1024^2 image
osc = Oracle().cornell(1024, 1024)
pix = np.arange(1024 * 1024)
o, d = osc.bounce_rays(RNG_PHILOX, TRIG_PORTABLE, 3, np.zeros_like(pix), pix)
reps = -(-n // len(o))
o, d = np.tile(o, (reps, 1))[:n], np.tile(d, (reps, 1))[:n]
run("bounce-3 rays", o, d)
