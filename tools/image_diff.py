#!/usr/bin/env python3
"""Compares two pictures written by the path tracers: PFM (linear radiance, HPT_PFM=1) or binary PPM (the tone-mapped
bytes of Utility::SavePPM).  Prints RMSE, RMSE of the values clamped to [0, 1] (what the north-star image check
gates: a GPU render against a cpu_path_tracer render of the same scene), per-channel means and the largest difference.

    tools/image_diff.py a.pfm b.pfm [--max-rmse-clamped X]     exit status 1 when the bound is exceeded
"""
import argparse
import sys

import numpy as np


def read_image(path):
    with open(path, "rb") as f:
        data = f.read()
    parts = data.split(b"\n", 3)
    if parts[0] == b"PF":
        w, h = (int(v) for v in parts[1].split())
        scale = float(parts[2])
        a = np.frombuffer(parts[3], "<f4" if scale < 0 else ">f4", w * h * 3).reshape(h, w, 3)
        return a[::-1].astype(np.float64), "pfm"     # PFM rows run bottom to top
    if parts[0] == b"P6":
        w, h = (int(v) for v in parts[1].split())
        a = np.frombuffer(parts[3], np.uint8, w * h * 3).reshape(h, w, 3)
        return a.astype(np.float64) / 255.0, "ppm"
    raise SystemExit("%s: neither PF nor P6" % path)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--max-rmse-clamped", type=float, default=None)
    ap.add_argument("--json", action="store_true", help="one JSON object instead of text")
    args = ap.parse_args()
    (a, ka), (b, kb) = read_image(args.a), read_image(args.b)
    if a.shape != b.shape:
        raise SystemExit("different sizes: %s vs %s" % (a.shape[:2], b.shape[:2]))
    if ka != kb:
        print("note: comparing a %s with a %s (linear radiance against tone-mapped bytes)" % (ka, kb))
    rmse = float(np.sqrt(np.mean((a - b) ** 2)))
    rmse_c = float(np.sqrt(np.mean((np.clip(a, 0, 1) - np.clip(b, 0, 1)) ** 2)))
    if args.json:
        import json
        print(json.dumps({"width": a.shape[1], "height": a.shape[0], "rmse": rmse, "rmse_clamped": rmse_c, "max_abs_diff": float(np.abs(a - b).max()),
                          "mean_a": a.mean((0, 1)).tolist(), "mean_b": b.mean((0, 1)).tolist(), "pixels_differing": int((a != b).any(2).sum())}))
        return 1 if args.max_rmse_clamped is not None and rmse_c > args.max_rmse_clamped else 0
    print("size %dx%d  rmse %.6g  rmse(clamped to [0,1]) %.6g  max |diff| %.6g" % (a.shape[1], a.shape[0], rmse, rmse_c, float(np.abs(a - b).max())))
    print("channel means a: %s  b: %s" % (np.round(a.mean((0, 1)), 6).tolist(), np.round(b.mean((0, 1)), 6).tolist()))
    print("identical" if np.array_equal(a, b) else "%d of %d pixels differ" % (int((a != b).any(2).sum()), a.shape[0] * a.shape[1]))
    if args.max_rmse_clamped is not None and rmse_c > args.max_rmse_clamped:
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
