#!/usr/bin/env python3
"""Traversal-only microbenchmark on the larger scenes: hpt_intersect_batch_dev on random rays,
Mrays/s for the bunny and the 1.0 M-triangle height field (block sizes 64..256)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cudapathtracer_amd as hp
from cudapathtracer_amd import scenes
from raybatches import random_rays

n = 1 << 22
o, d = random_rays(n, 1)
o_d = torch.from_numpy(np.ascontiguousarray(o)).cuda(); d_d = torch.from_numpy(np.ascontiguousarray(d)).cuda()
prim = torch.zeros(n, dtype=torch.int32, device="cuda"); t = torch.zeros(n, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream()
for name, sc in (("bunny", scenes.bunny_in_cornell(256, 256, 20)), ("heightfield708", scenes.heightfield_in_cornell(708, 256, 256, 20))):
    ctx = hp.Context(0).upload(sc)
    for block in (64, 256):
        ctx.intersect_dev(n, o_d, d_d, prim, t, None, None, block, st); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): ctx.intersect_dev(n, o_d, d_d, prim, t, None, None, block, st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print("%-16s block %3d  random closest %8.1f Mrays/s (%.2f ms, hit rate %.3f)" % (name, block, n / dt / 1e6, dt * 1e3, float((prim >= 0).float().mean())), flush=True)
    ctx.close()
