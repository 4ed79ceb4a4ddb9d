"""Python host side of the MI355X path-tracing core.

Two shared libraries are bound with ctypes:

* ``libhpt.so``       HIP kernels behind the C ABI of ``include/hpt.h``
* ``libhpt_host.so``  the C++ host classes (Scene / BVHAccel / HipRenderer)
                      behind ``include/hpt_host.h``

There is no CPU fallback: if a library is missing, or no GPU is present when a
device call is made, the call raises.  The classes mirror the reference's host
interface (gpu_path_tracer/CudaRenderer.hpp:9-33, common/Scene.hpp:20-93):
``Scene.CreateBuiltinScene`` / ``Add`` / ``buildBVH`` and ``HipRenderer`` with
``spp``, ``num_threads``, ``SetMode``, ``PrepareRender``, ``Render``,
``FinishRender`` and ``PathTrace``.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)
DEFAULT_MODELS_DIR = os.path.join(ROOT, "models")

SingleKernel = 1
Streamed = 2
FIXED_ONE = 16777216.0  # accumulators are 40.24 fixed point


class HptError(RuntimeError):
    pass


class hpt_material(C.Structure):
    _fields_ = [("type", C.c_int32), ("emission", C.c_float * 3), ("albedo", C.c_float * 3), ("metallic", C.c_float),
                ("roughness", C.c_float), ("ior", C.c_float), ("_pad", C.c_float * 2)]


class hpt_light(C.Structure):
    _fields_ = [("area", C.c_float), ("root_area", C.c_float), ("root_node", C.c_uint32), ("material", C.c_uint32)]


class hpt_scene_desc(C.Structure):
    _fields_ = [("num_nodes", C.c_uint32), ("nodes", C.POINTER(C.c_float)), ("node_area", C.POINTER(C.c_float)),
                ("num_tris", C.c_uint32), ("tris", C.POINTER(C.c_float)), ("tri_v12", C.POINTER(C.c_float)),
                ("num_materials", C.c_uint32), ("materials", C.POINTER(hpt_material)),
                ("num_lights", C.c_uint32), ("lights", C.POINTER(hpt_light)),
                ("sum_light_area", C.c_float), ("sum_light_power", C.c_float),
                ("cam_pos", C.c_float * 3), ("cam_scale", C.c_float), ("width", C.c_int32), ("height", C.c_int32),
                ("num_ref_nodes", C.c_uint32), ("ref_nodes", C.POINTER(C.c_float))]


class hpt_stats(C.Structure):
    _fields_ = [("closest_queries", C.c_uint64), ("shadow_queries", C.c_uint64), ("paths", C.c_uint64),
                ("kernel_ms", C.c_double), ("launches", C.c_uint32), ("variant", C.c_uint32),
                ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64)]

    def as_dict(self):
        return dict(closest_queries=int(self.closest_queries), shadow_queries=int(self.shadow_queries),
                    paths=int(self.paths), kernel_ms=float(self.kernel_ms), launches=int(self.launches),
                    variant=int(self.variant), box_tests=int(self.box_tests), tri_tests=int(self.tri_tests))


def Material(type=0, emission=(0.0, 0.0, 0.0), albedo=(0.9, 0.9, 0.9), metallic=0.0, roughness=1.0, ior=0.0):
    """Material with the reference's defaults (common/Material.hpp:137-142).
    type: 0 Lambertian, 1 MetallicWorkflow."""
    m = hpt_material()
    m.type = int(type)
    m.emission[:] = [float(np.float32(v)) for v in emission]
    m.albedo[:] = [float(np.float32(v)) for v in albedo]
    m.metallic, m.roughness, m.ior = float(metallic), float(roughness), float(ior)
    return m


_libs = None


def load_libraries():
    """Load (once) and return (libhpt, libhpt_host)."""
    global _libs
    if _libs is not None:
        return _libs
    # One HIP runtime per process: PyTorch's ROCm wheels bundle their own libamdhip64 / libhsa-runtime64,
    # and a process that loads libhpt.so first and torch afterwards ends up with two of each (streams and
    # synchronisation then belong to different runtimes).  With torch loaded first the kernel library
    # binds to the runtime that is already there.  Without torch installed the system's runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    # HPT_LIBHPT: load a differently built kernel library (A/B experiments)
    paths = [os.environ.get("HPT_LIBHPT") or os.path.join(_PKG, "libhpt.so"), os.path.join(_PKG, "libhpt_host.so")]
    for p in paths:
        if not os.path.exists(p):
            raise HptError("%s is missing: build it with `make hpt host` (or __graft_entry__.build()); "
                           "there is no CPU fallback" % p)
    try:
        hpt = C.CDLL(paths[0], mode=C.RTLD_GLOBAL)
        host = C.CDLL(paths[1])
    except OSError as e:
        raise HptError("cannot load the HIP extension: %s" % e)
    _declare(hpt, host)
    _libs = (hpt, host)
    return _libs


def _declare(L, H):
    vp, i32, u32, u64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64
    L.hpt_last_error.restype = C.c_char_p
    L.hpt_version.restype = C.c_char_p
    L.hpt_device_count.argtypes = [C.POINTER(C.c_int)]
    L.hpt_context_create.argtypes = [i32, C.POINTER(vp)]
    L.hpt_context_destroy.argtypes = [vp]
    L.hpt_scene_upload.argtypes = [vp, C.POINTER(hpt_scene_desc)]
    L.hpt_set_seed.argtypes = [vp, u32]
    L.hpt_set_integrator.argtypes = [vp, u32]
    L.hpt_set_wavefront_capacity.argtypes = [vp, u64]
    L.hpt_set_traversal.argtypes = [vp, C.c_int]
    L.hpt_octant_copies.argtypes = [C.POINTER(hpt_scene_desc), C.c_int, vp, vp]
    L.hpt_accum_clear.argtypes = [vp, vp, vp]
    L.hpt_render.argtypes = [vp, i32, i32, i32, u32, u32, vp, vp]
    L.hpt_render_static.argtypes = [vp, i32, i32, i32, u32, u32, vp, vp]
    L.hpt_resolve.argtypes = [vp, vp, u32, vp, vp]
    L.hpt_resolve_dev.argtypes = [vp, vp, u32, vp, vp]
    L.hpt_intersect_batch.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp, vp]
    L.hpt_intersect_batch_dev.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp, vp, i32, vp]
    L.hpt_cast_rays.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, i32, vp]
    L.hpt_cast_rays_dev.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, i32, vp, i32, vp]
    L.hpt_stats_get.argtypes = [vp, C.POINTER(hpt_stats)]
    L.hpt_stats_reset.argtypes = [vp]
    L.hpt_accum_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.hpt_accum_reduce.argtypes = [C.POINTER(vp), i32, i32]
    L.hpt_accum_reduce_prepare.argtypes = [C.POINTER(vp), i32]
    L.hpt_accum_reduce_info.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    L.hpt_synchronize.argtypes = [vp]
    L.hpt_accum_download.argtypes = [vp, vp, vp]
    H.hpt_host_last_error.restype = C.c_char_p
    H.hpt_host_scene_cornell.argtypes = [C.c_char_p, i32, i32, i32, C.POINTER(vp)]
    H.hpt_host_scene_builtin.argtypes = [i32, C.c_char_p, i32, i32, i32, C.POINTER(vp)]
    H.hpt_host_scene_named.argtypes = [C.c_char_p, C.c_char_p, i32, i32, i32, C.POINTER(vp)]
    H.hpt_host_scene_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, i32]
    H.hpt_host_scene_add_obj_transformed.argtypes = [vp, C.c_char_p, C.POINTER(hpt_material), vp, vp, vp]
    H.hpt_host_scene_add_sphere.argtypes = [vp, vp, C.c_float, C.POINTER(hpt_material)]
    H.hpt_host_scene_create.argtypes = [i32, i32, vp, C.c_double, i32, C.POINTER(vp)]
    H.hpt_host_scene_add_obj.argtypes = [vp, C.c_char_p, C.POINTER(hpt_material)]
    H.hpt_host_scene_add_triangles.argtypes = [vp, vp, i32, C.POINTER(hpt_material)]
    H.hpt_host_scene_build.argtypes = [vp]
    H.hpt_host_scene_set_resolution.argtypes = [vp, i32, i32]
    H.hpt_host_scene_destroy.argtypes = [vp]
    H.hpt_host_scene_describe.argtypes = [vp, C.POINTER(hpt_scene_desc)]
    H.hpt_host_scene_prim_ids.argtypes = [vp, vp]
    H.hpt_host_bvh_dump.argtypes = [vp, i32, C.POINTER(C.c_int), vp, vp, vp, vp]
    H.hpt_host_render.argtypes = [vp, i32, i32, i32, i32, i32, vp, C.POINTER(hpt_stats)]
    H.hpt_host_render_iterative.argtypes = [vp, i32, i32, i32, i32, i32, vp, C.POINTER(hpt_stats)]
    H.hpt_host_render_devices.argtypes = [vp, C.POINTER(C.c_int), i32, i32, i32, i32, i32, i32, vp, C.POINTER(hpt_stats)]
    H.hpt_host_save_ppm.argtypes = [C.c_char_p, vp, i32, i32]
    H.hpt_host_scene_intersect.argtypes = [vp, i32, C.c_size_t] + [vp] * 11
    H.hpt_host_scene_cast_rays.argtypes = [vp, i32, C.c_uint32, C.c_uint32, C.c_size_t] + [vp] * 5
    H.hpt_host_material_eval.argtypes = [C.POINTER(hpt_material), C.c_size_t] + [vp] * 5
    H.hpt_host_material_sample.argtypes = [C.POINTER(hpt_material), C.c_size_t] + [vp] * 4
    H.hpt_host_rng_floats.argtypes = [i32, i32, i32, vp]
    H.hpt_host_set_default_bvh_build.argtypes = [i32]
    H.hpt_host_default_bvh_build.argtypes = [C.POINTER(C.c_int)]
    H.hpt_host_scene_set_bvh_build.argtypes = [vp, i32]
    H.hpt_host_scene_bvh_build.argtypes = [vp, C.POINTER(C.c_int)]
    L.hpt_set_bvh_build.argtypes = [vp, C.c_int]
    L.hpt_scene_build_info.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    H.hpt_host_save_pfm.argtypes = [C.c_char_p, vp, i32, i32]


def _ck(rc):
    if rc != 0:
        raise HptError(load_libraries()[0].hpt_last_error().decode())


def _ckh(rc):
    if rc != 0:
        raise HptError(load_libraries()[1].hpt_host_last_error().decode())


def _ptr(x):
    """numpy array / torch tensor / int / None -> raw address."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    raise TypeError("cannot take the address of %r" % type(x))


def _stream(stream):
    if stream is None:
        return None
    return getattr(stream, "cuda_stream", stream)


TRAVERSAL_REFERENCE, TRAVERSAL_CULLED = 0, 1
BVH_BUILD_HOST, BVH_BUILD_GPU, BVH_BUILD_SAH, BVH_BUILD_AUTO = 0, 1, 2, 3
INTEGRATOR_RUSSIAN_ROULETTE, INTEGRATOR_REGULARIZE = 1, 2


class Scene:
    """Host scene (C++ ``Scene``): meshes, camera, light totals, two-level BVH
    and its flattened form."""

    CornellBox = 0
    BunnyInCornell = 1   # Cornell walls + light with the Stanford bunny in place of the boxes (not in the reference)
    GlassCornell = 2     # CornellBox with the short box switched to Material::Type::Glass (roughness 0.2, ior 1.5)

    def __init__(self, width, height, camPos=(0.0, 0.0, 0.0), fov=40.0, maxDepth=1, _handle=None):
        self._H = load_libraries()[1]
        self.width, self.height, self.maxDepth = int(width), int(height), int(maxDepth)
        self.name = "custom"
        if _handle is None:
            cam = np.asarray(camPos, np.float32)
            h = C.c_void_p()
            _ckh(self._H.hpt_host_scene_create(self.width, self.height, cam.ctypes.data, float(fov), self.maxDepth,
                                               C.byref(h)))
            _handle = h
        self._h = _handle
        self._built = False

    @staticmethod
    def CreateBuiltinScene(sceneId=0, maxDepth=10, width=0, height=0, models_dir=DEFAULT_MODELS_DIR):
        """Scene::CreateBuiltinScene (common/Scene.cpp:25-73); 512x512 unless a resolution is given."""
        if sceneId not in (Scene.CornellBox, Scene.BunnyInCornell, Scene.GlassCornell):
            raise HptError("Unsupported sceneId")
        H = load_libraries()[1]
        h = C.c_void_p()
        _ckh(H.hpt_host_scene_builtin(int(sceneId), models_dir.encode(), int(width), int(height), int(maxDepth), C.byref(h)))
        s = Scene(width or 512, height or 512, maxDepth=maxDepth, _handle=h)
        s.name = ("Cornell Box", "Bunny", "Glass Cornell")[sceneId]
        s._built = True
        return s

    @staticmethod
    def CreateNamedScene(name, maxDepth=10, width=0, height=0, models_dir=DEFAULT_MODELS_DIR, bvh_build=None):
        """Scene::CreateNamedScene: "cornell" | "bunny" | "glass" | "heightfield:<n>" | "file:<scene file>".
        bvh_build=BVH_BUILD_GPU: the scene is flattened for the device-side build (Scene::default_bvh_build);
        BVH_BUILD_SAH: host build with surface-area splits."""
        H = load_libraries()[1]
        h = C.c_void_p()
        if bvh_build is None:   # Scene::default_bvh_build as it stands (the reference's tree unless the process changed it)
            _ckh(H.hpt_host_scene_named(name.encode(), models_dir.encode(), int(width), int(height), int(maxDepth), C.byref(h)))
        else:
            before = C.c_int()
            _ckh(H.hpt_host_default_bvh_build(C.byref(before)))
            _ckh(H.hpt_host_set_default_bvh_build(int(bvh_build)))
            try:
                _ckh(H.hpt_host_scene_named(name.encode(), models_dir.encode(), int(width), int(height), int(maxDepth), C.byref(h)))
            finally:
                H.hpt_host_set_default_bvh_build(before.value)
        w, hh = C.c_int(), C.c_int()
        buf = C.create_string_buffer(256)
        _ckh(H.hpt_host_scene_info(h, C.byref(w), C.byref(hh), buf, 256))
        s = Scene(w.value, hh.value, maxDepth=maxDepth, _handle=h)
        s.name = buf.value.decode()
        s._built = True
        return s

    def Add(self, mesh, material, translation=None, rotation=None, scale=None):
        """mesh: path of an OBJ file (optionally with a Transform: translation, rotation in degrees about
        x, y, z, scale), or an array of triangle vertices (n, 3, 3)."""
        if isinstance(mesh, (str, bytes, os.PathLike)) and (translation is not None or rotation is not None or scale is not None):
            t = np.asarray(translation if translation is not None else (0, 0, 0), np.float32)
            r = np.asarray(rotation if rotation is not None else (0, 0, 0), np.float32)
            sc = np.broadcast_to(np.asarray(scale if scale is not None else 1, np.float32), (3,)).copy()
            _ckh(self._H.hpt_host_scene_add_obj_transformed(self._h, os.fspath(mesh).encode(), C.byref(material), t.ctypes.data,
                                                            r.ctypes.data, sc.ctypes.data))
        elif isinstance(mesh, (str, bytes, os.PathLike)):
            _ckh(self._H.hpt_host_scene_add_obj(self._h, os.fspath(mesh).encode(), C.byref(material)))
        else:
            v = np.ascontiguousarray(mesh, np.float32).reshape(-1, 9)
            _ckh(self._H.hpt_host_scene_add_triangles(self._h, v.ctypes.data, len(v), C.byref(material)))
        self._built = False

    def AddSphere(self, center, radius, material):
        """Scene::Add(Sphere*): center (3 floats), radius, a non-emitting material."""
        c = np.asarray(center, np.float32)
        _ckh(self._H.hpt_host_scene_add_sphere(self._h, c.ctypes.data, float(radius), C.byref(material)))
        self._built = False

    def buildBVH(self, gpu=None, sah=False):
        """Scene::buildBVH.  gpu=True: only the light meshes' trees are built on the host and the traversal records
        are built on the device at upload (BVH_BUILD_GPU); gpu=False: the reference's two-level BVH on the host;
        sah=True: the host build with surface-area splits for the top level and the non-emitting meshes
        (BVH_BUILD_SAH; same leaves, fewer box tests per ray); None keeps the scene's setting."""
        if sah:
            _ckh(self._H.hpt_host_scene_set_bvh_build(self._h, BVH_BUILD_SAH))
        elif gpu is not None:
            _ckh(self._H.hpt_host_scene_set_bvh_build(self._h, BVH_BUILD_GPU if gpu else BVH_BUILD_HOST))
        _ckh(self._H.hpt_host_scene_build(self._h))
        self._built = True
        return self

    @property
    def bvh_build(self):
        """How the scene was last built: BVH_BUILD_HOST, BVH_BUILD_GPU or BVH_BUILD_SAH."""
        b = C.c_int()
        _ckh(self._H.hpt_host_scene_bvh_build(self._h, C.byref(b)))
        return b.value

    @property
    def gpu_build(self):
        return self.bvh_build == BVH_BUILD_GPU

    def set_resolution(self, width, height):
        _ckh(self._H.hpt_host_scene_set_resolution(self._h, int(width), int(height)))
        self.width, self.height = int(width), int(height)

    def describe(self):
        d = hpt_scene_desc()
        _ckh(self._H.hpt_host_scene_describe(self._h, C.byref(d)))
        return d

    def flat_arrays(self):
        """Copies of the flattened arrays as numpy (for inspection and tests)."""
        d = self.describe()
        nn, nt = d.num_nodes, d.num_tris
        prim = np.zeros(nt, np.int32)
        _ckh(self._H.hpt_host_scene_prim_ids(self._h, prim.ctypes.data))
        return dict(
            nodes=np.ctypeslib.as_array(d.nodes, (nn, 8)).copy(),
            node_area=np.ctypeslib.as_array(d.node_area, (nn,)).copy(),
            tris=np.ctypeslib.as_array(d.tris, (nt, 16)).copy(),
            tri_v12=np.ctypeslib.as_array(d.tri_v12, (nt, 8)).copy(),
            ref_nodes=np.ctypeslib.as_array(d.ref_nodes, (d.num_ref_nodes, 8)).copy() if d.num_ref_nodes else None,
            prim_of_tri=prim,
            materials=np.array([(m.type,) + tuple(m.emission) + tuple(m.albedo) + (m.metallic, m.roughness, m.ior)
                                for m in (d.materials[i] for i in range(d.num_materials))], np.float32),
            lights=[(L.area, L.root_area, L.root_node, L.material) for L in (d.lights[i] for i in range(d.num_lights))],
            sum_light_area=d.sum_light_area, sum_light_power=d.sum_light_power, cam_scale=d.cam_scale,
            cam_pos=tuple(d.cam_pos), width=d.width, height=d.height)

    def octant_copies(self, leaf_bits=True):
        """The per-octant node copies hpt_scene_upload derives for a scene of up to 32 (64) triangles:
        (nodes float32[9, num_nodes, 8], table uint32[32]).  Host only."""
        L = load_libraries()[0]
        d = self.describe()
        nodes = np.zeros((9, d.num_nodes, 8), np.float32)
        table = np.zeros(32, np.uint32)
        _ck(L.hpt_octant_copies(C.byref(d), 1 if leaf_bits else 0, nodes.ctypes.data, table.ctypes.data))
        return nodes, table

    def bvh_dump(self, mesh):
        n = C.c_int()
        _ckh(self._H.hpt_host_bvh_dump(self._h, mesh, C.byref(n), None, None, None, None))
        b, p = np.zeros((n.value, 6), np.float32), np.zeros(n.value, np.int32)
        a, m = np.zeros(n.value, np.float32), np.zeros(n.value, np.int32)
        _ckh(self._H.hpt_host_bvh_dump(self._h, mesh, C.byref(n), b.ctypes.data, p.ctypes.data, a.ctypes.data,
                                       m.ctypes.data))
        return dict(bounds=b, payload=p, area=a, miss=m)

    def intersect(self, o, d, shadow=None, tmax=None, device=0):
        """Scene::intersect (common/Scene.hpp:95-97) of the C++ host class on a batch of rays, run on the GPU.
        Returns a dict of the Intersection fields (happened, prim, distance, coords, normal, triangleArea, material_type)."""
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        sh = None if shadow is None else np.ascontiguousarray(shadow, np.uint8)
        tm = None if tmax is None else np.ascontiguousarray(tmax, np.float64)
        out = dict(happened=np.zeros(n, np.uint8), prim=np.zeros(n, np.int32), distance=np.zeros(n, np.float64),
                   coords=np.zeros((n, 3), np.float32), normal=np.zeros((n, 3), np.float32),
                   triangleArea=np.zeros(n, np.float32), material_type=np.zeros(n, np.int32))
        _ckh(self._H.hpt_host_scene_intersect(self._h, int(device), n, o.ctypes.data, d.ctypes.data,
                                              None if sh is None else sh.ctypes.data, None if tm is None else tm.ctypes.data,
                                              *[out[k].ctypes.data for k in ("happened", "prim", "distance", "coords", "normal",
                                                                             "triangleArea", "material_type")]))
        return out

    def castRay(self, o, d, pixel_key, sample_key, device=0, seed=0, integrator=0):
        """Scene::castRay (common/Scene.hpp:274-285) of the C++ host class on a batch of eye rays, run on the GPU:
        radiance float32[n, 3] after up to the scene's maxDepth bounces; ray i draws from the random stream of the
        camera path (pixel_key[i], sample_key[i])."""
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        pk = np.ascontiguousarray(pixel_key, np.uint32).reshape(-1)
        sk = np.ascontiguousarray(sample_key, np.uint32).reshape(-1)
        n = len(o)
        if not (len(d) == len(pk) == len(sk) == n):
            raise ValueError("castRay: one direction and one (pixel, sample) key per origin")
        rgb = np.zeros((n, 3), np.float32)
        _ckh(self._H.hpt_host_scene_cast_rays(self._h, int(device), int(seed), int(integrator), n, o.ctypes.data, d.ctypes.data,
                                              pk.ctypes.data, sk.ctypes.data, rgb.ctypes.data))
        return rgb

    def render_cpp(self, spp, max_depth, num_threads=256, mode=SingleKernel, device=0, iterative=False, devices=None):
        """Render through the C++ HipRenderer class: PrepareRender/Render/FinishRender, or
        (iterative) PrepareRender, PathTrace(scene, iter) spp times, FinishRender.
        devices: list of GPU indices to shard the samples over (HipRenderer::devices); with devices, iterative=2 is
        Render() of the first spp // 2 samples followed by PathTrace for the rest."""
        rgb = np.zeros((self.width * self.height, 3), np.float32)
        st = hpt_stats()
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            _ckh(self._H.hpt_host_render_devices(self._h, arr, len(devices), spp, max_depth, num_threads, mode,
                                                 int(iterative), rgb.ctypes.data, C.byref(st)))
            return rgb, st.as_dict()
        fn = self._H.hpt_host_render_iterative if iterative else self._H.hpt_host_render
        _ckh(fn(self._h, device, spp, max_depth, num_threads, mode, rgb.ctypes.data, C.byref(st)))
        return rgb, st.as_dict()

    def close(self):
        if getattr(self, "_h", None):
            self._H.hpt_host_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One GPU: uploaded scene, accumulation buffer, ray counters (``hpt_ctx``)."""

    def __init__(self, device=0):
        self._L = load_libraries()[0]
        h = C.c_void_p()
        _ck(self._L.hpt_context_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.width = self.height = 0

    def upload(self, scene, bvh_build=None):
        """hpt_scene_upload; a Scene built with buildBVH(gpu=True) selects BVH_BUILD_GPU by itself."""
        d = scene.describe() if isinstance(scene, Scene) else scene
        if bvh_build is None:
            bvh_build = BVH_BUILD_GPU if (isinstance(scene, Scene) and scene.gpu_build) else BVH_BUILD_HOST
        if bvh_build == BVH_BUILD_SAH:   # a host build as far as the device library is concerned
            bvh_build = BVH_BUILD_HOST
        _ck(self._L.hpt_set_bvh_build(self._h, int(bvh_build)))
        _ck(self._L.hpt_scene_upload(self._h, C.byref(d)))
        self.width, self.height = d.width, d.height
        return self

    def build_info(self):
        """hpt_scene_build_info: device time of a GPU build (ms), number of two-child records, stack entries."""
        ms, n, d = C.c_double(), C.c_uint32(), C.c_uint32()
        _ck(self._L.hpt_scene_build_info(self._h, C.byref(ms), C.byref(n), C.byref(d)))
        return dict(gpu_build_ms=ms.value, records=n.value, stack_depth=d.value)

    def set_seed(self, seed):
        _ck(self._L.hpt_set_seed(self._h, int(seed)))

    def set_integrator(self, russian_roulette=False, regularize=False):
        """hpt_set_integrator: the two steps of Scene::TracePath the reference switches off at compile time."""
        _ck(self._L.hpt_set_integrator(self._h, (INTEGRATOR_RUSSIAN_ROULETTE if russian_roulette else 0)
                                       | (INTEGRATOR_REGULARIZE if regularize else 0)))
        return self

    def set_traversal(self, traversal):
        """TRAVERSAL_REFERENCE (default, bit-exact) or TRAVERSAL_CULLED; applies to the next upload."""
        _ck(self._L.hpt_set_traversal(self._h, int(traversal)))
        return self

    def set_wavefront_capacity(self, paths):
        _ck(self._L.hpt_set_wavefront_capacity(self._h, int(paths)))

    def accum_clear(self, accum=None, stream=None):
        _ck(self._L.hpt_accum_clear(self._h, _ptr(accum), _stream(stream)))

    def render(self, mode, block, max_depth, sample_begin, sample_count, accum=None, stream=None):
        _ck(self._L.hpt_render(self._h, int(mode), int(block), int(max_depth), int(sample_begin), int(sample_count),
                               _ptr(accum), _stream(stream)))

    def render_static(self, mode, block, max_depth, sample_begin, sample_count, accum=None, stream=None):
        """hpt_render_static: the render that allocates nothing (at most 2^18 camera paths per call)."""
        _ck(self._L.hpt_render_static(self._h, int(mode), int(block), int(max_depth), int(sample_begin),
                                      int(sample_count), _ptr(accum), _stream(stream)))

    def resolve(self, spp_total, accum=None, stream=None):
        rgb = np.zeros((self.width * self.height, 3), np.float32)
        _ck(self._L.hpt_resolve(self._h, _ptr(accum), int(spp_total), rgb.ctypes.data, _stream(stream)))
        return rgb

    def resolve_dev(self, spp_total, rgb_dev, accum=None, stream=None):
        _ck(self._L.hpt_resolve_dev(self._h, _ptr(accum), int(spp_total), _ptr(rgb_dev), _stream(stream)))

    def synchronize(self):
        """Wait for everything queued on this context's GPU (in the HIP runtime the library uses)."""
        _ck(self._L.hpt_synchronize(self._h))

    def accum_download(self, accum=None):
        """The exact int64 accumulators as numpy (pixels, 3)."""
        out = np.zeros((self.width * self.height, 3), np.int64)
        _ck(self._L.hpt_accum_download(self._h, _ptr(accum), out.ctypes.data))
        return out

    def accum_buffer(self):
        p, n = C.c_void_p(), C.c_size_t()
        _ck(self._L.hpt_accum_buffer(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def intersect(self, o, d, shadow=None, tmax=None):
        """Host arrays in, (prim int32[n], t float64[n]) out."""
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = len(o)
        sh = None if shadow is None else np.ascontiguousarray(shadow, np.uint8)
        tm = None if tmax is None else np.ascontiguousarray(tmax, np.float64)
        prim, t = np.zeros(n, np.int32), np.zeros(n, np.float64)
        _ck(self._L.hpt_intersect_batch(self._h, n, _ptr(o), _ptr(d), _ptr(sh), _ptr(tm), _ptr(prim), _ptr(t)))
        return prim, t

    def intersect_dev(self, n, o, d, prim, t, shadow=None, tmax=None, block=256, stream=None):
        """Device pointers / torch tensors; asynchronous on ``stream``."""
        _ck(self._L.hpt_intersect_batch_dev(self._h, int(n), _ptr(o), _ptr(d), _ptr(shadow), _ptr(tmax), _ptr(prim),
                                            _ptr(t), int(block), _stream(stream)))

    def cast_rays(self, o, d, pixel_key, sample_key, max_depth):
        """Scene::castRay (common/Scene.hpp:274-285) for a batch of eye rays: host arrays in, radiance float32[n, 3]
        out.  Ray r draws from the random stream of the camera path (pixel_key[r], sample_key[r])."""
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        pk = np.ascontiguousarray(pixel_key, np.uint32)
        sk = np.ascontiguousarray(sample_key, np.uint32)
        n = len(o)
        if not (o.shape == d.shape == (n, 3) and pk.shape == sk.shape == (n,)):
            raise ValueError("cast_rays: o and d are [n, 3], pixel_key and sample_key [n]")
        rgb = np.zeros((n, 3), np.float32)
        _ck(self._L.hpt_cast_rays(self._h, n, _ptr(o), _ptr(d), _ptr(pk), _ptr(sk), int(max_depth), _ptr(rgb)))
        return rgb

    def cast_rays_dev(self, n, o, d, pixel_key, sample_key, max_depth, rgb, block=256, stream=None):
        """Device pointers / torch tensors; asynchronous on ``stream``."""
        _ck(self._L.hpt_cast_rays_dev(self._h, int(n), _ptr(o), _ptr(d), _ptr(pixel_key), _ptr(sample_key), int(max_depth),
                                      _ptr(rgb), int(block), _stream(stream)))

    def stats(self):
        st = hpt_stats()
        _ck(self._L.hpt_stats_get(self._h, C.byref(st)))
        return st.as_dict()

    def stats_reset(self):
        _ck(self._L.hpt_stats_reset(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.hpt_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipRenderer:
    """Python mirror of the C++ ``HipRenderer`` (the reference's ``CudaRenderer``,
    gpu_path_tracer/CudaRenderer.hpp:15-33)."""

    def __init__(self, device=0):
        self.spp = 32
        self.num_threads = 256
        self._framebuffer = None
        self._stale = False
        self.device = device
        self._mode = SingleKernel
        self._ctx = None

    @property
    def framebuffer(self):
        """Renderer::framebuffer: accumulated / spp.  PathTrace only queues device work (like the
        reference's); the download happens here, when the picture is looked at, or in FinishRender."""
        self.RefreshFramebuffer()
        return self._framebuffer

    def RefreshFramebuffer(self):
        if self._stale and self._ctx is not None:
            self._framebuffer = self._ctx.resolve(max(self.spp, 1))
            self._stale = False

    def SetMode(self, mode):
        if mode not in (SingleKernel, Streamed):
            raise HptError("mode must be 1 (SingleKernel) or 2 (Streamed)")
        self._mode = mode

    def PrepareRender(self, scene):
        self.FinishRender(scene)
        self._ctx = Context(self.device).upload(scene)
        self._framebuffer = np.zeros((scene.width * scene.height, 3), np.float32)
        self._stale = False

    def Render(self, scene):
        if self._ctx is None:
            raise HptError("HipRenderer.Render: PrepareRender has not been called")
        self._ctx.accum_clear()
        if self.spp > 0:
            self._ctx.render(self._mode, self.num_threads, scene.maxDepth, 0, self.spp)
            self._framebuffer = self._ctx.resolve(self.spp)
            self._stale = False
        return self._framebuffer

    def PathTrace(self, scene, iter):
        if self._ctx is None:
            raise HptError("HipRenderer.PathTrace: PrepareRender has not been called")
        self._ctx.render(self._mode, self.num_threads, scene.maxDepth, iter, 1)
        self._stale = True

    def Stats(self):
        return self._ctx.stats() if self._ctx else {}

    def FinishRender(self, scene=None):
        if self._ctx is not None:
            self.RefreshFramebuffer()
            self._ctx.close()
            self._ctx = None


def material_eval(material, ray_source_dir, ray_out_dir, normal):
    """Material::bsdf and Material::pdf of the C++ host class (common/Material.hpp:257-384): (bsdf[n, 3], pdf[n])."""
    a, b, c = (np.ascontiguousarray(x, np.float32).reshape(-1, 3) for x in (ray_source_dir, ray_out_dir, normal))
    bsdf, pdf = np.zeros_like(a), np.zeros(len(a), np.float32)
    _ckh(load_libraries()[1].hpt_host_material_eval(C.byref(material), len(a), a.ctypes.data, b.ctypes.data, c.ctypes.data,
                                                    bsdf.ctypes.data, pdf.ctypes.data))
    return bsdf, pdf


def material_sample(material, u3, ray_out_dir, normal):
    """Material::sample of the C++ host class (common/Material.hpp:186-219) with the three uniform draws given per sample in the order the method draws them."""
    u, b, c = (np.ascontiguousarray(x, np.float32).reshape(-1, 3) for x in (u3, ray_out_dir, normal))
    out = np.zeros_like(b)
    _ckh(load_libraries()[1].hpt_host_material_sample(C.byref(material), len(b), u.ctypes.data, b.ctypes.data, c.ctypes.data,
                                                      out.ctypes.data))
    return out


def rng_floats(iter, index, count):
    """The first floats of the host RNG(iter, index) (common/MathUtils.hpp:22-36)."""
    out = np.zeros(count, np.float32)
    _ckh(load_libraries()[1].hpt_host_rng_floats(int(iter), int(index), int(count), out.ctypes.data))
    return out


def SavePPM(path, framebuffer, width, height):
    """Utility::SavePPM (common/Utility.cpp:4-22)."""
    rgb = np.ascontiguousarray(framebuffer, np.float32)
    _ckh(load_libraries()[1].hpt_host_save_ppm(os.fspath(path).encode(), rgb.ctypes.data, int(width), int(height)))


def SavePFM(path, framebuffer, width, height):
    """Utility::SavePFM: linear float RGB as a portable float map (not in the reference)."""
    rgb = np.ascontiguousarray(framebuffer, np.float32)
    _ckh(load_libraries()[1].hpt_host_save_pfm(os.fspath(path).encode(), rgb.ctypes.data, int(width), int(height)))


def accum_reduce(contexts, root=0):
    """hpt_accum_reduce: sum the contexts' own accumulation buffers into contexts[root]'s (one process
    driving several GPUs; RCCL between distinct devices)."""
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    _ck(load_libraries()[0].hpt_accum_reduce(arr, len(contexts), int(root)))


def accum_reduce_prepare(contexts):
    """hpt_accum_reduce_prepare: create the RCCL communicators a reduce over these contexts will use (kept
    and reused; nothing to do when they share one GPU)."""
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    _ck(load_libraries()[0].hpt_accum_reduce_prepare(arr, len(contexts)))


def accum_reduce_info():
    """hpt_accum_reduce_info: (communicator sets created by this process so far, host ms of the last reduce)."""
    n, ms = C.c_uint64(), C.c_double()
    _ck(load_libraries()[0].hpt_accum_reduce_info(C.byref(n), C.byref(ms)))
    return int(n.value), float(ms.value)


def device_count():
    n = C.c_int()
    _ck(load_libraries()[0].hpt_device_count(C.byref(n)))
    return n.value
