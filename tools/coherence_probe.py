import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import cudapathtracer_amd as hp
from cudapathtracer_amd import scenes
from raybatches import random_rays, primary_rays
n = 1 << 22
st = torch.cuda.current_stream()
def run(ctx, name, o, d):
    o_d = torch.from_numpy(np.ascontiguousarray(o)).cuda(); d_d = torch.from_numpy(np.ascontiguousarray(d)).cuda()
    prim = torch.zeros(len(o), dtype=torch.int32, device="cuda"); t = torch.zeros(len(o), dtype=torch.float64, device="cuda")
    ctx.intersect_dev(len(o), o_d, d_d, prim, t, None, None, 256, st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): ctx.intersect_dev(len(o), o_d, d_d, prim, t, None, None, 256, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("%-40s %8.1f Mrays/s" % (name, len(o) / dt / 1e6), flush=True)
for sname, sc in (("bunny", scenes.bunny_in_cornell(256, 256, 20)), ("hf708", scenes.heightfield_in_cornell(708, 256, 256, 20))):
    ctx = hp.Context(0).upload(sc)
    f = sc.flat_arrays()
    o, d = primary_rays(2048, 2048, scenes.CORNELL_CAM, f["cam_scale"]); run(ctx, sname + " primary", o, d)
    o, d = random_rays(n, 1); run(ctx, sname + " random", o, d)
    # sorted random rays: by direction octant, then origin cell (8x8x8)
    oc = ((d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2))
    cell = np.clip((o / np.array([556, 548.8, 559.2], np.float32) * 8).astype(np.int64), 0, 7)
    key = (oc << 9) | (cell[:, 0] << 6) | (cell[:, 1] << 3) | cell[:, 2]
    idx = np.argsort(key, kind="stable"); run(ctx, sname + " random sorted octant+8^3 cell", o[idx], d[idx])
    cell = np.clip((o / np.array([556, 548.8, 559.2], np.float32) * 32).astype(np.int64), 0, 31)
    key = (oc << 15) | (cell[:, 0] << 10) | (cell[:, 1] << 5) | cell[:, 2]
    idx = np.argsort(key, kind="stable"); run(ctx, sname + " random sorted octant+32^3 cell", o[idx], d[idx])
    ctx.close()
