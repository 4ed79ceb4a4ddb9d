"""Deterministic ray batches for parity tests and micro-benchmarks: inputs are regenerated from
seeds, only expected outputs are stored as fixtures.  Pure numpy; nothing here touches the oracle."""
import numpy as np


def primary_rays(width, height, cam_pos, scale, bias=0.5):
    """Camera rays with the reference's camera maths in float32 (common/Renderer.cpp:37-39); bias: the jitter of
    Renderer.cpp:32, one number (0.5 = pixel centres) or an array [height, width] of per-pixel draws."""
    f = np.float32
    j, i = np.meshgrid(np.arange(height, dtype=np.float32), np.arange(width, dtype=np.float32), indexing="ij")
    aspect = f(width) / f(height)
    bias = np.asarray(bias, np.float32)
    x = (f(2) * (i + bias) / f(width) - f(1)) * aspect * f(scale)
    y = (f(1) - f(2) * (j + bias) / f(height)) * f(scale)
    d = np.stack([-x, y, np.ones_like(x)], -1).reshape(-1, 3).astype(np.float32)
    inv = f(1) / np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32)
    d = (d * inv[:, None]).astype(np.float32)
    o = np.tile(np.asarray(cam_pos, np.float32), (len(d), 1))
    return o, d


def random_rays(n, seed, lo=(0, 0, 0), hi=(556, 548.8, 559.2)):
    """Origins uniform in the box [lo, hi], directions uniform on the sphere."""
    rs = np.random.RandomState(seed)
    o = (rs.rand(n, 3) * (np.asarray(hi) - np.asarray(lo)) + np.asarray(lo)).astype(np.float32)
    v = rs.randn(n, 3)
    d = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    return o, d


def shadow_rays(n, seed, lo=(0, 0, 0), hi=(556, 548.8, 559.2), light=((213, 548.7, 227), (343, 548.7, 332))):
    """Shadow rays from random points in the box towards random points on the
    light quad; tmax = distance (double of the float length, as Scene.hpp:203)."""
    rs = np.random.RandomState(seed)
    p = (rs.rand(n, 3) * (np.asarray(hi) - np.asarray(lo)) + np.asarray(lo)).astype(np.float32)
    l0, l1 = np.asarray(light[0]), np.asarray(light[1])
    x = (rs.rand(n, 3) * (l1 - l0) + l0).astype(np.float32)
    px = (x - p).astype(np.float32)
    ln = np.sqrt((px[:, 0] * px[:, 0] + px[:, 1] * px[:, 1]) + px[:, 2] * px[:, 2], dtype=np.float32)
    d = (px * (np.float32(1) / ln)[:, None]).astype(np.float32)
    o = (p + d * np.float32(1e-5)).astype(np.float32)
    return o, d, np.ones(n, np.uint8), ln.astype(np.float64)


def shadow_rays_down(n, seed):
    """Any-hit queries from high points towards the floor: a height field faces up, so only rays
    that come from above can be occluded by it."""
    rs = np.random.RandomState(seed)
    p = (rs.rand(n, 3) * [500, 300, 500] + [28, 200, 28]).astype(np.float32)
    x = (rs.rand(n, 3) * [500, 0, 500] + [28, 0, 28]).astype(np.float32)
    px = (x - p).astype(np.float32)
    ln = np.sqrt((px[:, 0] * px[:, 0] + px[:, 1] * px[:, 1]) + px[:, 2] * px[:, 2], dtype=np.float32)
    d = (px * (np.float32(1) / ln)[:, None]).astype(np.float32)
    return (p + d * np.float32(1e-5)).astype(np.float32), d, np.ones(n, np.uint8), ln.astype(np.float64)
