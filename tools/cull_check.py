#!/usr/bin/env python3
"""Larger comparison of HPT_TRAVERSAL_CULLED with the reference traversal: closest-hit (prim, t)
on random rays and occlusion on shadow rays towards the light, several seeds, both large scenes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cudapathtracer_amd as hp
from cudapathtracer_amd import scenes
from raybatches import random_rays

n = 1 << 22
st = torch.cuda.current_stream()
for name in ("bunny", "heightfield708"):
    sc = scenes.bunny_in_cornell(64, 64, 20) if name == "bunny" else scenes.heightfield_in_cornell(708, 64, 64, 20)
    ctxs = [hp.Context(0).set_traversal(t) for t in (0, 1)]
    for c in ctxs: c.upload(sc)
    tot = dict(rays=0, ids=0, t=0, srays=0, occ=0)
    for seed in range(8):
        o, d = random_rays(n, 100 + seed)
        rs = np.random.RandomState(200 + seed)
        p = (rs.rand(n, 3) * [556, 400, 559]).astype(np.float32)
        x = (rs.rand(n, 3) * [130, 0, 105] + [213, 548.7, 227]).astype(np.float32)
        px = x - p; ln = np.sqrt((px * px).sum(1, dtype=np.float32)).astype(np.float32)
        sd = (px / ln[:, None]).astype(np.float32); so = (p + sd * np.float32(1e-5)).astype(np.float32)
        res = []
        for c in ctxs:
            out = []
            for (oo, dd, sh, tm) in ((o, d, None, None), (so, sd, np.ones(n, np.uint8), ln.astype(np.float64))):
                t_o = torch.from_numpy(np.ascontiguousarray(oo)).cuda(); t_d = torch.from_numpy(np.ascontiguousarray(dd)).cuda()
                prim = torch.zeros(n, dtype=torch.int32, device="cuda"); t = torch.zeros(n, dtype=torch.float64, device="cuda")
                t_s = torch.from_numpy(sh).cuda() if sh is not None else None
                t_m = torch.from_numpy(tm).cuda() if tm is not None else None
                c.intersect_dev(n, t_o, t_d, prim, t, t_s, t_m, 256, st); torch.cuda.synchronize()
                out.append((prim.cpu().numpy(), t.cpu().numpy()))
            res.append(out)
        tot["rays"] += n; tot["ids"] += int((res[0][0][0] != res[1][0][0]).sum()); tot["t"] += int((res[0][0][1] != res[1][0][1]).sum())
        tot["srays"] += n; tot["occ"] += int(((res[0][1][0] >= 0) != (res[1][1][0] >= 0)).sum())
    print("%-16s closest: %d rays, %d ids differ, %d t differ; shadow: %d rays, %d occlusion results differ" % (name, tot["rays"], tot["ids"], tot["t"], tot["srays"], tot["occ"]), flush=True)
    for c in ctxs: c.close()
