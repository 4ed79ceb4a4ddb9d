"""The multi-GPU render entries on a one-GPU box: every piece of the N-GPU flow that can run
there does, and the N-way picture must equal the 1-way picture bit for bit (file: byte for byte).

* hpt_accum_reduce: several contexts in one process (the C ABI's reduce; RCCL between distinct
  devices, loaded and run with a one-device communicator here);
* HipRenderer::devices (C++): samples sharded over the listed GPUs, one host thread each;
* python -m cudapathtracer_amd.render --gpus 2: one process per rank, the ranks' buffers reduced
  with torch.distributed (gloo, both ranks on GPU 0 here; "nccl" = RCCL on a multi-GPU node);
* bench.py under torch.distributed.run with 2 ranks (gloo) and with the RCCL backend (1 rank).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from orclib import RNG_PHILOX, TRIG_PORTABLE

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(kw)
    return env


@pytest.mark.parametrize("force_rccl", ["0", "1"])
def test_accum_reduce_of_contexts_in_one_process(hp, oracle, monkeypatch, force_rccl):
    w, h, depth, spp = 48, 40, 8, 7
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
    ctxs = [hp.Context(0).upload(sc) for _ in range(3)]
    try:
        for rank, c in enumerate(ctxs):
            begin, count = hp.sample_range(spp, rank, len(ctxs))
            c.accum_clear()
            c.render(1 + rank % 2, 256, depth, begin, count)
        monkeypatch.setenv("HPT_REDUCE_FORCE_RCCL", force_rccl)
        hp.accum_reduce(ctxs, root=1)
        got = ctxs[1].resolve(spp)
        want_acc, _ = oracle.cornell(w, h).render(RNG_PHILOX, TRIG_PORTABLE, depth, spp, fixed=True)
        assert np.array_equal(got.view(np.int32), oracle.resolve(want_acc, spp).view(np.int32))
        with pytest.raises(hp.HptError):
            hp.accum_reduce(ctxs, root=3)
        with pytest.raises(hp.HptError):
            hp.accum_reduce([ctxs[0], ctxs[0]], root=0)
        other = hp.Context(0).upload(hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, 16, 16))
        with pytest.raises(hp.HptError):
            hp.accum_reduce([ctxs[0], other], root=0)
        other.close()
    finally:
        for c in ctxs:
            c.close()


def test_second_reduce_reuses_the_communicator(hp, monkeypatch):
    """The RCCL communicators of hpt_accum_reduce are made once per device list and kept: the first reduce (or
    hpt_accum_reduce_prepare) creates them, every later reduce over the same devices is one grouped ncclReduce."""
    w, h, depth = 32, 32, 4
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
    monkeypatch.setenv("HPT_REDUCE_FORCE_RCCL", "1")   # one GPU here: a communicator of one device all the same
    ctxs = [hp.Context(0).upload(sc) for _ in range(2)]
    try:
        before, _ = hp.accum_reduce_info()
        hp.accum_reduce_prepare(ctxs)
        made, _ = hp.accum_reduce_info()
        # (a set made by an earlier reduce of this process over the same device is still there while any context lives)
        assert made >= 1 and made in (before, before + 1)
        for rnd in range(3):
            for k, c in enumerate(ctxs):
                c.accum_clear()
                c.render(1, 256, depth, 2 * rnd + k, 1)
            hp.accum_reduce(ctxs, root=0)
            n, ms = hp.accum_reduce_info()
            assert n == made, "reduce %d created a communicator" % rnd
            assert ms > 0
            got = ctxs[0].accum_download()
            ctxs[1].accum_clear()
            ctxs[1].render(2, 256, depth, 2 * rnd, 2)
            assert np.array_equal(got, ctxs[1].accum_download())
        # another root over the same devices: still the same communicator set
        hp.accum_reduce(ctxs, root=1)
        assert hp.accum_reduce_info()[0] == made
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_render_then_pathtrace_counts_every_sample_once(hp, mode):
    """HipRenderer with several devices: Render() reduces the other contexts' buffers into the first; samples added
    afterwards with PathTrace() and reduced again must not bring Render()'s shares along a second time."""
    w, h, depth, spp = 48, 40, 6, 9
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
    one, _ = sc.render_cpp(spp, depth, 256, mode)
    for devices in ([0, 0], [0, 0, 0]):
        refined, st = sc.render_cpp(spp, depth, 256, mode, devices=devices, iterative=2)
        assert st["paths"] == w * h * spp
        assert np.array_equal(one.view(np.int32), refined.view(np.int32)), devices


@pytest.mark.parametrize("mode", [1, 2])
def test_cpp_renderer_sharded_over_devices(hp, mode):
    w, h, depth, spp = 64, 48, 6, 5
    sc = hp.Scene.CreateBuiltinScene(hp.Scene.CornellBox, depth, w, h)
    one, st1 = sc.render_cpp(spp, depth, 256, mode)
    many, st3 = sc.render_cpp(spp, depth, 256, mode, devices=[0, 0, 0])
    assert np.array_equal(one.view(np.int32), many.view(np.int32))
    assert (st1["paths"], st1["closest_queries"], st1["shadow_queries"]) == (st3["paths"], st3["closest_queries"], st3["shadow_queries"])
    it, _ = sc.render_cpp(spp, depth, 256, mode, devices=[0, 0], iterative=True)
    assert np.array_equal(one.view(np.int32), it.view(np.int32))


def test_cli_sharded_over_devices_writes_the_same_file(tmp_path):
    exe = os.path.join(ROOT, "build", "gpu_path_tracer")
    outs = []
    for tag, extra in (("one", {}), ("three", {"HPT_DEVICES": "0,0,0"})):
        d = tmp_path / tag
        (d / "out" / "gpu").mkdir(parents=True)
        r = subprocess.run([exe, "6", "8", "256", "2"], cwd=str(d), env=_env(HPT_MODELS_DIR=os.path.join(ROOT, "models"), HPT_WIDTH="96", HPT_HEIGHT="64", **extra),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=100)
        assert r.returncode == 0, r.stdout
        outs.append((d / "out" / "gpu" / "Cornell Box.ppm").read_bytes())
        if extra:
            assert "HIP #GPUs: 3" in r.stdout
    assert outs[0] == outs[1] and len(outs[0]) > 96 * 64 * 3


def test_render_module_two_ranks_write_the_file_one_rank_writes(tmp_path):
    files = []
    for gpus in (1, 2):
        out = str(tmp_path / ("r%d.ppm" % gpus))
        cmd = [sys.executable, "-m", "cudapathtracer_amd.render", "--gpus", str(gpus), "--backend", "gloo", "--device", "0",
               "--width", "96", "--height", "64", "--out", out, "--master-port", "29541", "7", "8", "256", "2"]
        r = subprocess.run(cmd, cwd=ROOT, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=110)
        assert r.returncode == 0, r.stdout[-3000:]
        assert "HIP #GPUs: %d" % gpus in r.stdout
        files.append(open(out, "rb").read())
    assert files[0] == files[1] and files[0].startswith(b"P6")


def test_bench_two_ranks_and_rccl_backend():
    """bench.py as the driver launches it for N > 1 (2 ranks on GPU 0, gloo), and its RCCL branch with
    one rank (communicator + int64 reduce on device memory through RCCL)."""
    common = ["--steps", "1", "--warmup", "1", "--spp-per-step", "2", "--width", "128", "--height", "128", "--no-cpu-baseline"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29551", os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common
    r = subprocess.run(cmd, cwd=ROOT, env=_env(HPT_BENCH_BACKEND="gloo", HPT_BENCH_DEVICE="0"), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["collective_backend"] == "gloo"
    assert line["config"]["world_size"] == 2 and [r["rank"] for r in line["config"]["ranks"]] == [0, 1]
    assert line["config"]["reduce_ms"] is not None and line["config"]["reduce_bytes"] == 3 * 128 * 128 * 8
    leg = line["strong_config"]   # the 4096-spp job split over the two ranks, beside the weak line
    assert leg["scaling"] == "strong" and leg["n_gpus"] == 2 and leg["paths"] == leg["paths_expected"] == 128 * 128 * 4096
    # the same program with --scaling strong: total work fixed, split over steps and ranks
    r = subprocess.run(cmd + ["--scaling", "strong", "--total-spp", "7", "--steps", "2"], cwd=ROOT, env=_env(HPT_BENCH_BACKEND="gloo", HPT_BENCH_DEVICE="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["n_gpus"] == 2 and line["steps"] == 2 and line["strong_config"] is None
    assert "7 spp in all" in line["config"]["workload"] and line["config"]["world_size"] == 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, cwd=ROOT, env=_env(HPT_BENCH_FORCE_DIST="1", MASTER_PORT="29561"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["collective_backend"] == "nccl"
