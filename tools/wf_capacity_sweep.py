import sys, time
sys.path.insert(0, "/root/repo")
import torch, cudapathtracer_amd as hp
W = H = 1024; depth = 20
sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, W, H)
for cap_log in (22, 23, 24, 25, 26, 27):
    ctx = hp.Context(0); ctx.set_wavefront_capacity(1 << cap_log); ctx.upload(sc)
    spp = 128
    ctx.render(2, 256, depth, 0, 16); torch.cuda.synchronize()
    ctx.stats_reset(); ctx.accum_clear()
    t0 = time.perf_counter(); ctx.render(2, 256, depth, 100, spp); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = ctx.stats(); rays = st["closest_queries"] + st["shadow_queries"]
    print("capacity 2^%d: %.1f Mrays/s (%.1f ms, kernel %.1f ms, %d launches)" % (cap_log, rays / dt / 1e6, dt * 1e3, st["kernel_ms"], st["launches"]), flush=True)
    ctx.close()
