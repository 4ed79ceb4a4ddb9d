"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, where it
exists, the compiled reference (oracle/_ref/libref.so).

TEST INFRASTRUCTURE ONLY: the product package never imports this module.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = os.path.join(ROOT, "models")
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref.so")

RNG_MT, RNG_PHILOX = 0, 1
TRIG_LIBM, TRIG_PORTABLE = 0, 1
CAMERA_DEPTH = 0xFFFFFFFF


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("emission", C.c_float * 3), ("albedo", C.c_float * 3),
                ("metallic", C.c_float), ("roughness", C.c_float), ("ior", C.c_float)]

    @staticmethod
    def make(type=0, emission=(0, 0, 0), albedo=(.9, .9, .9), metallic=0.0, roughness=1.0, ior=0.0):
        m = Material()
        m.type = type
        m.emission[:] = [float(np.float32(v)) for v in emission]
        m.albedo[:] = [float(np.float32(v)) for v in albedo]
        m.metallic, m.roughness, m.ior = metallic, roughness, ior
        return m


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("closest_queries", "shadow_queries", "box_tests", "tri_tests", "paths",
                                          "stalled_bounces", "nan_resets")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Oracle:
    """The CPU restatement."""

    def __init__(self, path=ORACLE_SO):
        self.lib = L = C.CDLL(path)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_double]
        L.orc_scene_create_cornell.restype = C.c_void_p
        L.orc_scene_create_cornell.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_add_obj.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Material)]
        L.orc_scene_add_triangles.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Material)]
        L.orc_scene_build.argtypes = [C.c_void_p]
        L.orc_scene_add_sphere.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.POINTER(Material)]
        L.orc_scene_set_integrator.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_solve_quadratic.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_material_regularize.restype = C.c_float
        L.orc_material_regularize.argtypes = [C.c_float]
        L.orc_scene_set_resolution.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_scene_info.argtypes = [C.c_void_p] + [C.c_void_p] * 8
        L.orc_mesh_info.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
        L.orc_mesh_triangles.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_bvh_dump.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
        L.orc_intersect_batch.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 7 + [C.c_int]
        L.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.orc_cast_paths.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int, C.POINTER(Stats)]
        L.orc_cast_rays.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int64] + [C.c_void_p] * 5 + [
            C.c_int, C.POINTER(Stats)]
        L.orc_capture_bounce_rays.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int64] + [C.c_void_p] * 5 + [C.c_int]
        L.orc_rng_floats.argtypes = [C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_material_eval.argtypes = [C.POINTER(Material), C.c_int64] + [C.c_void_p] * 5
        L.orc_material_sample.argtypes = [C.POINTER(Material), C.c_int, C.c_int64] + [C.c_void_p] * 4
        L.orc_sample_hemisphere_cosine.argtypes = [C.c_int, C.c_int64] + [C.c_void_p] * 4
        L.orc_sample_light.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_power_heuristic.restype = C.c_float
        L.orc_power_heuristic.argtypes = [C.c_float, C.c_float]
        L.orc_tonemap.argtypes = [C.c_float]
        L.orc_philox4x32_10.argtypes = [C.c_void_p] * 3
        L.orc_philox_uniform.restype = C.c_float
        L.orc_philox_uniform.argtypes = [C.c_uint32] * 5
        L.orc_radiance_to_fixed.restype = C.c_int64
        L.orc_radiance_to_fixed.argtypes = [C.c_float]
        L.orc_fixed_to_radiance.restype = C.c_float
        L.orc_fixed_to_radiance.argtypes = [C.c_int64, C.c_int]
        L.orc_portable_trig.argtypes = [C.c_int64] + [C.c_void_p] * 4

    # ---- scenes -------------------------------------------------------
    def cornell(self, width=512, height=512, models_dir=MODELS):
        h = self.lib.orc_scene_create_cornell(models_dir.encode(), width, height)
        if not h:
            raise RuntimeError("cornell scene: OBJ files not found under " + models_dir)
        return OracleScene(self, h)

    def scene(self, width, height, cam_pos, fov):
        cam = _f32(cam_pos)
        return OracleScene(self, self.lib.orc_scene_create(width, height, _p(cam), fov))

    # ---- units --------------------------------------------------------
    def rng_floats(self, mode, iter, index, count, seed=0):
        out = np.zeros(count, np.float32)
        self.lib.orc_rng_floats(mode, seed, iter, index, count, _p(out))
        return out

    def material_eval(self, mat, wi, wo, n):
        wi, wo, n = _f32(wi), _f32(wo), _f32(n)
        bs = np.zeros_like(wi)
        pdf = np.zeros(len(wi), np.float32)
        self.lib.orc_material_eval(C.byref(mat), len(wi), _p(wi), _p(wo), _p(n), _p(bs), _p(pdf))
        return bs, pdf

    def material_sample(self, mat, u3, wo, n, trig=TRIG_LIBM):
        u3, wo, n = _f32(u3), _f32(wo), _f32(n)
        out = np.zeros_like(wo)
        self.lib.orc_material_sample(C.byref(mat), trig, len(wo), _p(u3), _p(wo), _p(n), _p(out))
        return out

    def sample_hemisphere_cosine(self, n, rx, ry, trig=TRIG_LIBM):
        n, rx, ry = _f32(n), _f32(rx), _f32(ry)
        out = np.zeros_like(n)
        self.lib.orc_sample_hemisphere_cosine(trig, len(n), _p(n), _p(rx), _p(ry), _p(out))
        return out

    def philox(self, ctr, key):
        ctr = np.ascontiguousarray(ctr, np.uint32)
        key = np.ascontiguousarray(key, np.uint32)
        out = np.zeros(4, np.uint32)
        self.lib.orc_philox4x32_10(_p(ctr), _p(key), _p(out))
        return out

    def philox_uniform(self, pixel, sample, depth, slot, seed=0):
        return float(self.lib.orc_philox_uniform(pixel, sample, depth, slot, seed))

    def portable_trig(self, x):
        x = _f32(x)
        s, c, a = np.zeros_like(x), np.zeros_like(x), np.zeros_like(x)
        self.lib.orc_portable_trig(x.size, _p(x), _p(s), _p(c), _p(a))
        return s, c, a

    def to_fixed(self, v):
        return int(self.lib.orc_radiance_to_fixed(float(v)))

    def resolve(self, accum, spp_total):
        """int64 accumulators -> float32 image, same expression as fixed_to_radiance."""
        return (accum.astype(np.float64) / (16777216.0 * float(spp_total))).astype(np.float32)


class OracleScene:
    def __init__(self, orc, handle):
        self.orc, self.lib, self.h = orc, orc.lib, handle

    def close(self):
        if self.h:
            self.lib.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_obj(self, path, mat):
        r = self.lib.orc_scene_add_obj(self.h, path.encode(), C.byref(mat))
        if r < 0:
            raise RuntimeError("cannot load " + path)
        return r

    def add_triangles(self, verts, mat):
        v = _f32(verts).reshape(-1, 9)
        return self.lib.orc_scene_add_triangles(self.h, _p(v), len(v), C.byref(mat))

    def add_sphere(self, center, radius, mat):
        c = _f32(center)
        return self.lib.orc_scene_add_sphere(self.h, _p(c), float(radius), C.byref(mat))

    def set_integrator(self, russian_roulette=False, path_regularization=False):
        self.lib.orc_scene_set_integrator(self.h, int(russian_roulette), int(path_regularization))
        return self

    def build(self):
        self.lib.orc_scene_build(self.h)
        return self

    def set_resolution(self, w, h):
        self.lib.orc_scene_set_resolution(self.h, w, h)

    def info(self):
        nm, nt, w, h, nl = (C.c_int() for _ in range(5))
        sa, sp, cs = (C.c_float() for _ in range(3))
        self.lib.orc_scene_info(self.h, C.byref(nm), C.byref(nt), C.byref(w), C.byref(h), C.byref(sa), C.byref(sp),
                                C.byref(nl), C.byref(cs))
        return dict(num_meshes=nm.value, num_tris=nt.value, width=w.value, height=h.value, sum_light_area=sa.value,
                    sum_light_power=sp.value, num_lights=nl.value, cam_scale=cs.value)

    def mesh_info(self, mesh):
        nt = C.c_int()
        area, root_area = C.c_float(), C.c_float()
        bbox = np.zeros(6, np.float32)
        self.lib.orc_mesh_info(self.h, mesh, C.byref(nt), C.byref(area), _p(bbox), C.byref(root_area))
        return dict(num_tris=nt.value, area=area.value, bbox=bbox, bvh_root_area=root_area.value)

    def mesh_triangles(self, mesh):
        out = np.zeros((self.mesh_info(mesh)["num_tris"], 13), np.float32)
        self.lib.orc_mesh_triangles(self.h, mesh, _p(out))
        return out

    def bvh_dump(self, mesh):
        n = self.lib.orc_bvh_dump(self.h, mesh, None, None, None, None)
        b, p = np.zeros((n, 6), np.float32), np.zeros(n, np.int32)
        a, m = np.zeros(n, np.float32), np.zeros(n, np.int32)
        r = self.lib.orc_bvh_dump(self.h, mesh, _p(b), _p(p), _p(a), _p(m))
        assert r == n
        return dict(bounds=b, payload=p, area=a, miss=m)

    def intersect(self, o, d, shadow=None, tmax=None, counts=False, threads=0):
        o, d = _f32(o), _f32(d)
        n = len(o)
        sh = None if shadow is None else np.ascontiguousarray(shadow, np.uint8)
        tm = None if tmax is None else np.ascontiguousarray(tmax, np.float64)
        prim, t = np.zeros(n, np.int32), np.zeros(n, np.float64)
        cnt = np.zeros((n, 2), np.int32) if counts else None
        self.lib.orc_intersect_batch(self.h, n, _p(o), _p(d), _p(sh), _p(tm), _p(prim), _p(t), _p(cnt), threads)
        return (prim, t, cnt) if counts else (prim, t)

    def render(self, rng, trig, max_depth, spp, sample_begin=0, spp_total=None, seed=0, threads=0, fixed=False):
        """Returns (float image | int64 accumulators, stats dict)."""
        i = self.info()
        P = i["width"] * i["height"]
        st = Stats()
        spp_total = spp if spp_total is None else spp_total
        if fixed:
            acc = np.zeros((P, 3), np.int64)
            self.lib.orc_render(self.h, rng, trig, seed, max_depth, sample_begin, spp, spp_total, threads, None, _p(acc),
                                C.byref(st))
            return acc, st.as_dict()
        rgb = np.zeros((P, 3), np.float32)
        self.lib.orc_render(self.h, rng, trig, seed, max_depth, sample_begin, spp, spp_total, threads, _p(rgb), None,
                            C.byref(st))
        return rgb, st.as_dict()

    def cast_paths(self, rng, trig, max_depth, samples, pixels, seed=0, threads=0):
        k = np.ascontiguousarray(samples, np.int32)
        m = np.ascontiguousarray(pixels, np.int32)
        out = np.zeros((len(k), 3), np.float32)
        st = Stats()
        self.lib.orc_cast_paths(self.h, rng, trig, seed, max_depth, len(k), _p(k), _p(m), _p(out), threads, C.byref(st))
        return out, st.as_dict()

    def cast_rays(self, rng, trig, max_depth, o, d, samples, pixels, seed=0, threads=0):
        o, d = _f32(o), _f32(d)
        k = np.ascontiguousarray(samples, np.int32)
        m = np.ascontiguousarray(pixels, np.int32)
        out = np.zeros((len(k), 3), np.float32)
        st = Stats()
        self.lib.orc_cast_rays(self.h, rng, trig, seed, max_depth, len(k), _p(o), _p(d), _p(k), _p(m), _p(out), threads, C.byref(st))
        return out, st.as_dict()

    def bounce_rays(self, rng, trig, bounce, samples, pixels, seed=0, threads=0):
        """(o, d) of the rays the camera paths (sample, pixel) trace at bounce `bounce`; paths that ended earlier are
        left out."""
        k = np.ascontiguousarray(samples, np.int32)
        m = np.ascontiguousarray(pixels, np.int32)
        o, d = np.zeros((len(k), 3), np.float32), np.zeros((len(k), 3), np.float32)
        alive = np.zeros(len(k), np.uint8)
        self.lib.orc_capture_bounce_rays(self.h, rng, trig, seed, bounce, len(k), _p(k), _p(m), _p(o), _p(d), _p(alive), threads)
        keep = alive != 0
        return o[keep], d[keep]

    def sample_light(self, u4):
        u4 = _f32(u4)
        out = np.zeros((len(u4), 11), np.float32)
        self.lib.orc_sample_light(self.h, len(u4), _p(u4), _p(out))
        return out


def have_ref():
    return os.path.exists(REF_SO)


class Ref:
    """The compiled reference (only in the build container).  One global scene."""

    def __init__(self, path=REF_SO):
        self.lib = L = C.CDLL(path)
        L.ref_scene_cornell.argtypes = [C.c_char_p, C.c_int]
        L.ref_scene_begin.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double]
        L.ref_scene_add_obj.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.ref_set_resolution.argtypes = [C.c_int, C.c_int]
        L.ref_set_max_depth.argtypes = [C.c_int]
        L.ref_scene_info.argtypes = [C.c_void_p] * 8
        L.ref_mesh_info.argtypes = [C.c_int] + [C.c_void_p] * 4
        L.ref_mesh_triangles.argtypes = [C.c_int, C.c_void_p]
        L.ref_bvh_dump.argtypes = [C.c_int] + [C.c_void_p] * 4
        L.ref_intersect_batch.argtypes = [C.c_long] + [C.c_void_p] * 8
        L.ref_render.argtypes = [C.c_int, C.c_void_p]
        L.ref_cast_paths.argtypes = [C.c_long, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_rng_floats.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ref_material_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_long] + \
            [C.c_void_p] * 5
        L.ref_material_sample.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_long,
                                          C.c_int, C.c_int] + [C.c_void_p] * 4
        L.ref_sample_hemisphere_cosine.argtypes = [C.c_long] + [C.c_void_p] * 4
        L.ref_sample_light.argtypes = [C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_power_heuristic.restype = C.c_float
        L.ref_power_heuristic.argtypes = [C.c_float, C.c_float]
        L.ref_tonemap.argtypes = [C.c_float]
        L.ref_solve_quadratic.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ref_material_regularize.restype = C.c_float
        L.ref_material_regularize.argtypes = [C.c_float]

    def cornell(self, max_depth=5, root=ROOT):
        rc = self.lib.ref_scene_cornell(root.encode(), max_depth)
        if rc:
            raise RuntimeError("ref_scene_cornell rc=%d" % rc)
        return self

    def begin(self, w, h, max_depth, cam_pos, fov):
        cam = _f32(cam_pos)
        self.lib.ref_scene_begin(w, h, max_depth, _p(cam), fov)
        return self

    def add_obj(self, path, mat):
        e = np.array(list(mat.emission), np.float32)
        a = np.array(list(mat.albedo), np.float32)
        return self.lib.ref_scene_add_obj(path.encode(), mat.type, _p(e), _p(a), mat.metallic, mat.roughness, mat.ior)

    def build(self):
        self.lib.ref_scene_build()
        return self

    def set_resolution(self, w, h):
        self.lib.ref_set_resolution(w, h)

    def set_max_depth(self, d):
        self.lib.ref_set_max_depth(d)

    def info(self):
        nm, nt, w, h, nl, nc = (C.c_int() for _ in range(6))
        sa, sp = C.c_float(), C.c_float()
        self.lib.ref_scene_info(C.byref(nm), C.byref(nt), C.byref(w), C.byref(h), C.byref(sa), C.byref(sp), C.byref(nl),
                                C.byref(nc))
        return dict(num_meshes=nm.value, num_tris=nt.value, width=w.value, height=h.value, sum_light_area=sa.value,
                    sum_light_power=sp.value, num_lights=nl.value, node_counter=nc.value)

    def mesh_info(self, mesh):
        nt = C.c_int()
        area, root_area = C.c_float(), C.c_float()
        bbox = np.zeros(6, np.float32)
        self.lib.ref_mesh_info(mesh, C.byref(nt), C.byref(area), _p(bbox), C.byref(root_area))
        return dict(num_tris=nt.value, area=area.value, bbox=bbox, bvh_root_area=root_area.value)

    def mesh_triangles(self, mesh):
        out = np.zeros((self.mesh_info(mesh)["num_tris"], 13), np.float32)
        self.lib.ref_mesh_triangles(mesh, _p(out))
        return out

    def bvh_dump(self, mesh):
        n = self.lib.ref_bvh_dump(mesh, None, None, None, None)
        b, p = np.zeros((n, 6), np.float32), np.zeros(n, np.int32)
        a, m = np.zeros(n, np.float32), np.zeros(n, np.int32)
        r = self.lib.ref_bvh_dump(mesh, _p(b), _p(p), _p(a), _p(m))
        assert r == n, "nextIfHit is not the pre-order successor"
        return dict(bounds=b, payload=p, area=a, miss=m)

    def intersect(self, o, d, shadow=None, tmax=None):
        o, d = _f32(o), _f32(d)
        n = len(o)
        sh = None if shadow is None else np.ascontiguousarray(shadow, np.uint8)
        tm = None if tmax is None else np.ascontiguousarray(tmax, np.float64)
        prim, t = np.zeros(n, np.int32), np.zeros(n, np.float64)
        agree, cnt = np.zeros(n, np.int32), np.zeros((n, 2), np.int32)
        self.lib.ref_intersect_batch(n, _p(o), _p(d), _p(sh), _p(tm), _p(prim), _p(t), _p(agree), _p(cnt))
        return prim, t, agree, cnt

    def render(self, spp):
        i = self.info()
        rgb = np.zeros((i["width"] * i["height"], 3), np.float32)
        self.lib.ref_render(spp, _p(rgb))
        return rgb

    def cast_paths(self, samples, pixels):
        k = np.ascontiguousarray(samples, np.int32)
        m = np.ascontiguousarray(pixels, np.int32)
        out = np.zeros((len(k), 3), np.float32)
        self.lib.ref_cast_paths(len(k), _p(k), _p(m), _p(out))
        return out

    def rng_floats(self, iter, index, count):
        out = np.zeros(count, np.float32)
        self.lib.ref_rng_floats(iter, index, count, _p(out))
        return out

    def material_eval(self, mat, wi, wo, n):
        wi, wo, n = _f32(wi), _f32(wo), _f32(n)
        e = np.array(list(mat.emission), np.float32)
        a = np.array(list(mat.albedo), np.float32)
        bs, pdf = np.zeros_like(wi), np.zeros(len(wi), np.float32)
        self.lib.ref_material_eval(mat.type, _p(e), _p(a), mat.metallic, mat.roughness, mat.ior, len(wi), _p(wi), _p(wo),
                                   _p(n), _p(bs), _p(pdf))
        return bs, pdf

    def material_sample(self, mat, iter0, index, wo, n):
        wo, n = _f32(wo), _f32(n)
        e = np.array(list(mat.emission), np.float32)
        a = np.array(list(mat.albedo), np.float32)
        wi, u = np.zeros_like(wo), np.zeros((len(wo), 3), np.float32)
        self.lib.ref_material_sample(mat.type, _p(e), _p(a), mat.metallic, mat.roughness, mat.ior, len(wo), iter0, index,
                                     _p(wo), _p(n), _p(wi), _p(u))
        return wi, u

    def sample_hemisphere_cosine(self, n, rx, ry):
        n, rx, ry = _f32(n), _f32(rx), _f32(ry)
        out = np.zeros_like(n)
        self.lib.ref_sample_hemisphere_cosine(len(n), _p(n), _p(rx), _p(ry), _p(out))
        return out

    def sample_light(self, count, iter0, index):
        out, u = np.zeros((count, 11), np.float32), np.zeros((count, 4), np.float32)
        self.lib.ref_sample_light(count, iter0, index, _p(out), _p(u))
        return out, u


# deterministic ray batches: tests/raybatches.py (no oracle involved); re-exported for the tests
from raybatches import primary_rays, random_rays, shadow_rays, shadow_rays_down  # noqa: E402,F401


def fnv1a64(prim, t):
    """FNV-1a-64 over int32_le(prim) || float64_le(t) per ray."""
    rec = np.zeros(len(prim), dtype=[("p", "<i4"), ("t", "<f8")])
    rec["p"], rec["t"] = prim, t
    h = 0xcbf29ce484222325
    for b in rec.tobytes():
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def fnv_bytes(data):
    """FNV-1a-64 folded over the little-endian 64-bit words of the zero-padded byte string and
    then over its padded length (a byte-wise loop takes minutes on a 2.0 M-node BVH dump)."""
    data = data + b"\0" * (-len(data) % 8)
    h = 0xcbf29ce484222325
    for w in np.frombuffer(data, "<u8").tolist() + [len(data)]:
        h = ((h ^ w) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def bvh_hash(d):
    """fnv_bytes over a BVH dump's node bounds, miss links and payloads (pre-order)."""
    return fnv_bytes(d["bounds"].astype("<f4").tobytes() + d["miss"].astype("<i4").tobytes()
                     + d["payload"].astype("<i4").tobytes())
