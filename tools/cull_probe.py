#!/usr/bin/env python3
"""HPT_TRAVERSAL_CULLED against the reference traversal on the larger scenes: fraction of
rays whose (prim, t) differ, and traversal-only / rendering rates of both."""
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
st = torch.cuda.current_stream()
which = sys.argv[1:] or ["bunny", "heightfield708"]
for name in which:
    W = 512
    sc = scenes.bunny_in_cornell(W, W, 20) if name == "bunny" else scenes.heightfield_in_cornell(int(name[11:]), W, W, 20)
    res = {}
    for trav in (0, 1):
        ctx = hp.Context(0).set_traversal(trav)
        ctx.upload(sc)
        prim = torch.zeros(n, dtype=torch.int32, device="cuda"); t = torch.zeros(n, dtype=torch.float64, device="cuda")
        ctx.intersect_dev(n, o_d, d_d, prim, t, None, None, 256, st); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): ctx.intersect_dev(n, o_d, d_d, prim, t, None, None, 256, st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        line = "%-16s traversal %d variant %d  random closest %8.1f Mrays/s" % (name, trav, ctx.stats()["variant"], n / dt / 1e6)
        for mode in (1, 2):
            ctx.accum_clear(); ctx.stats_reset()
            ctx.render(mode, 256, 20, 0, 4); s = ctx.stats()
            ctx.stats_reset(); ctx.render(mode, 256, 20, 4, 16); s = ctx.stats()
            rays = s["closest_queries"] + s["shadow_queries"]
            line += "   mode %d %7.1f Mrays/s" % (mode, rays / s["kernel_ms"] / 1e3)
            res[(trav, mode)] = (ctx.resolve(20).copy(), rays)
        print(line, flush=True)
        res[trav] = (prim.cpu().numpy(), t.cpu().numpy())
        ctx.close()
    dp = res[0][0] != res[1][0]; dt_ = res[0][1] != res[1][1]
    print("%-16s differing ids %d, differing t %d of %d rays" % (name, dp.sum(), dt_.sum(), n))
    for mode in (1, 2):
        a, ra = res[(0, mode)]; b, rb = res[(1, mode)]
        print("%-16s mode %d: image max |diff| %.3g, pixels differing %d of %d, rays %d vs %d" % (name, mode, np.abs(a - b).max(), (a != b).any(-1).sum() if a.ndim > 1 else (a != b).sum(), a.shape[0], ra, rb))
