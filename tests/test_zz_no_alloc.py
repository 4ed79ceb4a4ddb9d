"""hpt_render_static -- the render entry point that allocates nothing -- through the C ABI against
hpt_render on the same inputs: the int64 accumulators must be bit-identical (integer work: no
tolerance).  Two calls in a row, the second with a smaller input, so that stale contents of the
shared scratch array (path records, live counters of the first call) would show."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _accum(ctx, n, fn, *args):
    import torch
    acc = torch.zeros(n, dtype=torch.int64, device="cuda:0")
    fn(*args, accum=acc, stream=torch.cuda.current_stream())
    torch.cuda.synchronize()
    return acc.cpu().numpy()


@pytest.mark.parametrize("mode", [2, 1])
def test_static_render_equals_render(hp, gpu_ctx, mode, monkeypatch):
    import torch
    from cudapathtracer_amd import scenes
    depth = 8
    # first call: the Cornell box at the reference's own 512 x 512, one sample = exactly the limit
    big = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, 512, 512)
    gpu_ctx.upload(big)
    want = _accum(gpu_ctx, 3 * 512 * 512, gpu_ctx.render, mode, 256, depth, 3, 1)
    got = _accum(gpu_ctx, 3 * 512 * 512, gpu_ctx.render_static, mode, 256, depth, 3, 1)
    assert want.any() and np.array_equal(got, want)
    # the property the entry point exists for: it allocates nothing.  With HPT_ALLOC_LIMIT=0 the library's
    # allocator refuses every request (hpt_render would fail here: its wavefront queues are gone after the
    # upload below frees nothing but a streamed hpt_render of a larger batch would have to grow them), and
    # the card's free memory must not move over the calls.
    acc = torch.zeros(3 * 512 * 512, dtype=torch.int64, device="cuda:0")
    gpu_ctx.render_static(mode, 256, depth, 3, 1, accum=acc, stream=torch.cuda.current_stream())   # warm: code objects, kernel arguments
    torch.cuda.synchronize()
    acc.zero_()
    monkeypatch.setenv("HPT_ALLOC_LIMIT", "0")
    free_before = torch.cuda.mem_get_info(0)[0]
    for _ in range(3):
        gpu_ctx.render_static(mode, 256, depth, 3, 1, accum=acc, stream=torch.cuda.current_stream())
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(0)[0] == free_before, "hpt_render_static changed the card's free memory"
    assert np.array_equal(acc.cpu().numpy(), 3 * want)
    fresh = hp.Context(0)
    with pytest.raises(hp.HptError, match="HPT_ALLOC_LIMIT"):   # the limit does bite where something is allocated
        fresh.upload(big)
    fresh.close()
    monkeypatch.delenv("HPT_ALLOC_LIMIT")
    with pytest.raises(hp.HptError, match="at most 262144 camera paths"):
        gpu_ctx.render_static(mode, 256, depth, 0, 2)
    # second call, smaller: another scene (the bunny: path records through another traversal variant),
    # odd resolution with partial edge tiles, several samples
    w, h, spp = 100, 52, 7
    small = scenes.bunny_in_cornell(w, h, depth)
    gpu_ctx.upload(small)
    want = _accum(gpu_ctx, 3 * w * h, gpu_ctx.render, mode, 192, depth, 10, spp)
    got = _accum(gpu_ctx, 3 * w * h, gpu_ctx.render_static, mode, 192, depth, 10, spp)
    assert want.any() and np.array_equal(got, want)
    # and a third, smaller again, into the context's own accumulation buffer
    gpu_ctx.upload(hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, 40, 24))
    gpu_ctx.accum_clear()
    gpu_ctx.render(mode, 256, depth, 0, 5)
    want_img = gpu_ctx.resolve(5)
    gpu_ctx.accum_clear()
    gpu_ctx.render_static(mode, 256, depth, 0, 5)
    assert np.array_equal(gpu_ctx.resolve(5).view(np.int32), want_img.view(np.int32))


def test_static_render_rejects_what_render_rejects(hp, gpu_ctx):
    gpu_ctx.upload(hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, 5, 16, 16))
    with pytest.raises(hp.HptError):
        gpu_ctx.render_static(3, 256, 5, 0, 1)
    with pytest.raises(hp.HptError):
        gpu_ctx.render_static(1, 0, 5, 0, 1)
    fresh = hp.Context(0)
    with pytest.raises(hp.HptError):
        fresh.render_static(2, 256, 5, 0, 1)
    fresh.close()
    gpu_ctx.accum_clear()
    gpu_ctx.render_static(2, 256, 0, 0, 4)   # zero depth / zero samples: no-ops like hpt_render
    gpu_ctx.render_static(1, 256, 5, 0, 0)
    assert not gpu_ctx.resolve(1).any()


@pytest.mark.parametrize("mode", [2, 1])
def test_static_renders_of_two_contexts_on_two_streams_do_not_race(hp, mode):
    """Every hpt_render_static on a GPU uses that GPU's one scratch array.  Two contexts, each on a stream of its
    own, queue their calls back to back without waiting: the library orders them (each call waits for the event
    the previous call on the GPU recorded), so both get the sums hpt_render gives."""
    import torch
    depth, w, h, spp = 10, 256, 256, 3
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
    ctxs = [hp.Context(0).upload(sc) for _ in range(2)]
    try:
        want = []
        for k, c in enumerate(ctxs):
            acc = torch.zeros(3 * w * h, dtype=torch.int64, device="cuda:0")
            for r in range(4):
                c.render(mode, 256, depth, 10 * k + r * spp, spp, accum=acc, stream=torch.cuda.current_stream())
            torch.cuda.synchronize()
            want.append(acc.cpu().numpy())
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        accs = [torch.zeros(3 * w * h, dtype=torch.int64, device="cuda:0") for _ in ctxs]
        torch.cuda.synchronize()
        for r in range(4):
            for k, c in enumerate(ctxs):   # alternate the contexts; nothing waits on the host
                c.render_static(mode, 256, depth, 10 * k + r * spp, spp, accum=accs[k], stream=streams[k])
        for s_ in streams:
            s_.synchronize()
        torch.cuda.synchronize()
        for k in range(2):
            assert want[k].any() and np.array_equal(accs[k].cpu().numpy(), want[k]), "context %d" % k
    finally:
        for c in ctxs:
            c.close()
