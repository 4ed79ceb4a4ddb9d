"""MI355X-native path-tracing core behind the host interface of
kevinsadi/CudaPathTracer (Scene / HipRenderer), see api.py."""
from .api import (Context, HipRenderer, HptError, Material, SavePPM, SavePFM, Scene, SingleKernel, Streamed,  # noqa: F401
                  TRAVERSAL_CULLED, TRAVERSAL_REFERENCE, BVH_BUILD_AUTO, BVH_BUILD_GPU, BVH_BUILD_HOST, BVH_BUILD_SAH,
                  accum_reduce, accum_reduce_info, accum_reduce_prepare, device_count, load_libraries, material_eval, material_sample, rng_floats)
from .sharding import reduce_accum, render_sharded, sample_range  # noqa: F401
