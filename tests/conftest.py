import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from orclib import ORACLE_SO, Oracle
    if not os.path.exists(ORACLE_SO):
        pytest.fail("oracle/liboracle.so is missing: run `make oracle` (or __graft_entry__.build())")
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The compiled reference; only exists in the build container."""
    from orclib import Ref, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference): `make ref`")
    return Ref()


@pytest.fixture(scope="session")
def hp():
    import cudapathtracer_amd
    cudapathtracer_amd.load_libraries()
    # HPT_TEST_BVH=sah | reference: every scene the tests create without naming a builder gets the surface-area tree
    # (also the small ones, which by default keep the reference's) / the reference's tree alone -- runs of the whole
    # suite on either tree; the default is "auto" (surface-area tree for scenes of more than 64 primitives)
    which = {"sah": cudapathtracer_amd.BVH_BUILD_SAH, "reference": cudapathtracer_amd.BVH_BUILD_HOST}.get(os.environ.get("HPT_TEST_BVH", ""))
    if which is not None:
        cudapathtracer_amd.api._ckh(cudapathtracer_amd.load_libraries()[1].hpt_host_set_default_bvh_build(which))
    return cudapathtracer_amd


@pytest.fixture(scope="session")
def gpu_ctx(hp):
    """A context on GPU 0; fails (never skips) when the HIP path is unusable."""
    if hp.device_count() < 1:
        pytest.fail("no GPU visible to the HIP runtime")
    ctx = hp.Context(0)
    yield ctx
    ctx.close()
