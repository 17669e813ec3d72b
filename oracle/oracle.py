"""ctypes wrapper around oracle/libcrt_oracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcrt_oracle.so")
MISS = 0xFFFFFFFF
MODE_LAMBERT = 100
MODE_PATH = 200

NODE_DTYPE = np.dtype([("lx0", "f4"), ("lx1", "f4"), ("ly0", "f4"), ("ly1", "f4"),
                       ("rx0", "f4"), ("rx1", "f4"), ("ry0", "f4"), ("ry1", "f4"),
                       ("lz0", "f4"), ("lz1", "f4"), ("rz0", "f4"), ("rz1", "f4"),
                       ("left", "i4"), ("right", "i4"), ("pad0", "i4"), ("pad1", "i4")])
NODE4_DTYPE = np.dtype([("minx", "f4", 4), ("maxx", "f4", 4), ("miny", "f4", 4), ("maxy", "f4", 4), ("minz", "f4", 4), ("maxz", "f4", 4),
                        ("ref", "i4", 4), ("pad", "i4", 4)])
NODE4Q_DTYPE = np.dtype([("lo", "f4", 3), ("s", "f4", 3), ("qlo_x", "u4"), ("qhi_x", "u4"), ("qlo_y", "u4"), ("qhi_y", "u4"), ("qlo_z", "u4"), ("qhi_z", "u4"), ("ref", "i4", 4)])
assert NODE4Q_DTYPE.itemsize == 64
EMPTY = -1
TRI_DTYPE = np.dtype([("v0", "f4", 3), ("inst", "u4"), ("e1", "f4", 3), ("prim", "u4"),
                      ("e2", "f4", 3), ("gid", "u4")])
SHADE_DTYPE = np.dtype([("n0", "f4", 3), ("n1", "f4", 3), ("n2", "f4", 3), ("material", "u4"), ("pad", "u4", 2)])
assert NODE4_DTYPE.itemsize == 128 and NODE_DTYPE.itemsize == 64 and TRI_DTYPE.itemsize == 48 and SHADE_DTYPE.itemsize == 48


class _Mesh(C.Structure):
    _fields_ = [("xyz", C.c_void_p), ("idx", C.c_void_p), ("normals", C.c_void_p), ("uvs", C.c_void_p),
                ("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32), ("material_index", C.c_int32)]


class _Light(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("intensity", C.c_float)]


class _Material(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("type", C.c_uint32), ("smooth", C.c_uint32), ("ior", C.c_float), ("texture", C.c_int32)]


class _Texture(C.Structure):
    _fields_ = [("type", C.c_uint32), ("color_a", C.c_float * 3), ("color_b", C.c_float * 3), ("scalar", C.c_float),
                ("pixels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint32)]


TEXTURE_TYPES = {"albedo": 0, "edges": 1, "checker": 2, "bitmap": 3}
UV_DTYPE = np.dtype([("uv0", "f4", 2), ("uv1", "f4", 2), ("uv2", "f4", 2)])


def make_texture(t, keep):
    """t: dict {type: albedo|edges|checker|bitmap, color_a, color_b, scalar, pixels (H,W,C uint8)}"""
    x = _Texture()
    x.type = TEXTURE_TYPES[t["type"]]
    x.color_a = (C.c_float * 3)(*[float(c) for c in t.get("color_a", (0, 0, 0))])
    x.color_b = (C.c_float * 3)(*[float(c) for c in t.get("color_b", (0, 0, 0))])
    x.scalar = float(t.get("scalar", 0.0))
    px = t.get("pixels")
    if px is not None:
        px = np.ascontiguousarray(px, dtype=np.uint8)
        keep.append(px)
        x.pixels = px.ctypes.data
        x.height, x.width, x.channels = px.shape
    return x


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("pixels", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build(force=False):
    """Compile the oracle (gcc).  Building the checker is not using it."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f))
                                               for f in ("crt_oracle.c", "crt_oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "libcrt_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_scene_create.restype = C.c_void_p
        L.oracle_scene_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.oracle_scene_create_ex.restype = C.c_void_p
        L.oracle_scene_create_ex.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int]
        L.oracle_scene_destroy.argtypes = [C.c_void_p]
        L.oracle_scene_set_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
        L.oracle_scene_set_width.argtypes = [C.c_void_p, C.c_int]
        for f in ("oracle_scene_node_count", "oracle_scene_tri_count", "oracle_scene_max_depth", "oracle_scene_node4_count", "oracle_scene_depth4"):
            getattr(L, f).restype = C.c_uint32
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("oracle_scene_nodes", "oracle_scene_tris", "oracle_scene_shade", "oracle_scene_nodes4", "oracle_scene_nodes4q"):
            getattr(L, f).restype = C.c_void_p
            getattr(L, f).argtypes = [C.c_void_p]
        L.oracle_render.restype = C.c_int
        L.oracle_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_int]
        L.oracle_ray_dir.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_sinf.restype = C.c_float
        L.oracle_sinf.argtypes = [C.c_float]
        L.oracle_shade_mode.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_unorm8.restype = C.c_uint8
        L.oracle_unorm8.argtypes = [C.c_float]
        L.oracle_intersect_tri.restype = C.c_int
        L.oracle_intersect_tri.argtypes = [C.c_void_p] * 5 + [C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_max_threads.restype = C.c_int
        L.oracle_scene_set_textures.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.oracle_scene_uvs.restype = C.c_void_p
        L.oracle_scene_uvs.argtypes = [C.c_void_p]
        L.oracle_texture_color.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.oracle_set_path_params.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.oracle_set_phong.argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_set_phong.restype = None
        L.oracle_set_cost_outputs.argtypes = [C.c_void_p] * 4
        L.oracle_set_stack_output.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _f32(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if n is not None:
        assert a.size == n
    return a


class OracleScene:
    """meshes: list of dicts {vertices (N,3) f32, triangles (M,3) u32, normals (N,3) f32|None, material_index}
    lights: list of (pos3, intensity); materials: list of dicts {albedo, type, smooth_shading, ior}."""

    def __init__(self, meshes, lights=(), materials=(), build_mode=0, textures=()):
        """build_mode 0 = binned SAH, 1 = LBVH (spec of the GPU builder); textures: list of dicts (make_texture),
        referenced by materials through {"texture": index}"""
        L = lib()
        self._keep = []
        marr = (_Mesh * max(1, len(meshes)))()
        for i, m in enumerate(meshes):
            v = _f32(m["vertices"]).reshape(-1, 3)
            t = np.ascontiguousarray(m["triangles"], dtype=np.uint32).reshape(-1, 3)
            nrm = m.get("normals")
            if nrm is not None:
                nrm = _f32(nrm).reshape(-1, 3)
                assert nrm.shape == v.shape
            uv = m.get("uvs")
            if uv is not None:
                uv = _f32(uv).reshape(-1, 3)
                assert uv.shape == v.shape
            self._keep += [v, t, nrm, uv]
            marr[i].xyz = v.ctypes.data
            marr[i].idx = t.ctypes.data
            marr[i].normals = nrm.ctypes.data if nrm is not None else None
            marr[i].uvs = uv.ctypes.data if uv is not None else None
            marr[i].n_vertices = v.shape[0]
            marr[i].n_triangles = t.shape[0]
            marr[i].material_index = int(m.get("material_index", 0))
        larr = (_Light * max(1, len(lights)))()
        for i, (p, inten) in enumerate(lights):
            larr[i].pos = (C.c_float * 3)(*[float(x) for x in p])
            larr[i].intensity = float(inten)
        matarr = (_Material * max(1, len(materials)))()
        for i, m in enumerate(materials):
            matarr[i].albedo = (C.c_float * 3)(*[float(x) for x in m.get("albedo", (1, 1, 1))])
            matarr[i].type = int(m.get("type", 1))
            matarr[i].smooth = int(bool(m.get("smooth_shading", False)))
            matarr[i].ior = float(m.get("ior", 1.0))
            matarr[i].texture = int(m.get("texture", -1))
        self.h = L.oracle_scene_create_ex(marr, len(meshes), larr, len(lights), matarr, len(materials), int(build_mode))
        if not self.h:
            raise RuntimeError("oracle_scene_create failed")
        if textures:
            tarr = (_Texture * len(textures))(*[make_texture(t, self._keep) for t in textures])
            assert L.oracle_scene_set_textures(self.h, tarr, len(textures)) == 0

    def close(self):
        if self.h:
            lib().oracle_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_nodes(self):
        return lib().oracle_scene_node_count(self.h)

    @property
    def n_tris(self):
        return lib().oracle_scene_tri_count(self.h)

    @property
    def max_depth(self):
        return lib().oracle_scene_max_depth(self.h)

    @property
    def n_nodes4(self):
        return lib().oracle_scene_node4_count(self.h)

    @property
    def depth4(self):
        return lib().oracle_scene_depth4(self.h)

    def nodes4(self):
        return self._view("oracle_scene_nodes4", NODE4_DTYPE, self.n_nodes4)

    def nodes4q(self):
        """the 64-byte quantised nodes the wide walk (and the kernels) test"""
        return self._view("oracle_scene_nodes4q", NODE4Q_DTYPE, self.n_nodes4)

    def set_width(self, width):
        """4 = wide tree (default, what the kernels walk), 2 = the binary tree it is collapsed from"""
        lib().oracle_scene_set_width(self.h, int(width))

    def _view(self, fn, dtype, n):
        p = getattr(lib(), fn)(self.h)
        if n == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (n * dtype.itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dtype).copy()

    def nodes(self):
        return self._view("oracle_scene_nodes", NODE_DTYPE, self.n_nodes)

    def tris(self):
        return self._view("oracle_scene_tris", TRI_DTYPE, self.n_tris)

    def shade(self):
        return self._view("oracle_scene_shade", SHADE_DTYPE, self.n_tris)

    def uvs(self):
        p = lib().oracle_scene_uvs(self.h)
        if not p or self.n_tris == 0:
            return None
        return np.frombuffer((C.c_char * (self.n_tris * 24)).from_address(p), dtype=UV_DTYPE).copy()

    def set_bvh(self, nodes, tris, shade=None):
        nodes = np.ascontiguousarray(nodes)
        tris = np.ascontiguousarray(tris)
        assert nodes.dtype.itemsize == 64 and tris.dtype.itemsize == 48
        sp = None
        if shade is not None:
            shade = np.ascontiguousarray(shade)
            assert shade.dtype.itemsize == 48 and len(shade) == len(tris)
            sp = shade.ctypes.data
        rc = lib().oracle_scene_set_bvh(self.h, nodes.ctypes.data, len(nodes), tris.ctypes.data, sp, len(tris))
        assert rc == 0

    def render(self, pos, rot, mode, w, h, miss_rgb=(0.0, 1.0, 1.0), rows=None, brute_force=False, n_threads=0,
               want=("rgba8", "hit_inst", "hit_prim", "hit_t", "rgb")):
        """rows = (y_begin, y_end, y_step) or None for the full frame. Returns dict of arrays + 'stats'."""
        pos = _f32(pos, 3)
        rot = _f32(rot, 9)
        miss = _f32(miss_rgb, 3)
        y0, y1, ys = rows if rows is not None else (0, h, 1)
        out = {}
        if "rgba8" in want:
            out["rgba8"] = np.zeros((h, w, 4), dtype=np.uint8)
        if "hit_inst" in want:
            out["hit_inst"] = np.full((h, w), MISS, dtype=np.uint32)
        if "hit_prim" in want:
            out["hit_prim"] = np.full((h, w), MISS, dtype=np.uint32)
        if "hit_t" in want:
            out["hit_t"] = np.zeros((h, w), dtype=np.float32)
        if "rgb" in want:
            out["rgb"] = np.zeros((h, w, 3), dtype=np.float32)
        st = Stats()

        def p(k):
            return out[k].ctypes.data if k in out else None

        rc = lib().oracle_render(self.h, pos.ctypes.data, rot.ctypes.data, int(mode), miss.ctypes.data, w, h,
                                 y0, y1, ys, p("rgba8"), p("hit_inst"), p("hit_prim"), p("hit_t"), p("rgb"),
                                 C.byref(st), int(bool(brute_force)), int(n_threads))
        if rc != 0:
            raise RuntimeError("oracle_render failed rc=%d" % rc)
        out["stats"] = st.as_dict()
        return out


def occluded(scene, o, d, tmin, tmax, brute_force=False):
    """any-hit query on the open interval (tmin, tmax) of the ray o + t d: 1 if some triangle lies in it"""
    L = lib()
    L.oracle_occluded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int]
    o, d = _f32(o, 3), _f32(d, 3)
    return int(L.oracle_occluded(scene.h, o.ctypes.data, d.ctypes.data, float(np.float32(tmin)), float(np.float32(tmax)), int(bool(brute_force))))


def ray_dir(rot, px, py, w, h):
    rot = _f32(rot, 9)
    o = np.zeros(3, dtype=np.float32)
    lib().oracle_ray_dir(rot.ctypes.data, px, py, w, h, o.ctypes.data)
    return o


def sinf(x):
    return float(lib().oracle_sinf(float(np.float32(x))))


def shade_mode(mode, inst, prim, t, u, v, o, d):
    o = _f32(o, 3)
    d = _f32(d, 3)
    out = np.zeros(3, dtype=np.float32)
    lib().oracle_shade_mode(mode, inst, prim, float(np.float32(t)), float(np.float32(u)), float(np.float32(v)),
                            o.ctypes.data, d.ctypes.data, out.ctypes.data)
    return out


def unorm8(c):
    return int(lib().oracle_unorm8(float(np.float32(c))))


def intersect_tri(o, d, v0, v1, v2, tmin=0.001, tmax=10000.0):
    arrs = [_f32(a, 3) for a in (o, d, v0, v1, v2)]
    t = C.c_float()
    u = C.c_float()
    v = C.c_float()
    hit = lib().oracle_intersect_tri(*[a.ctypes.data for a in arrs], tmin, tmax, C.byref(t), C.byref(u), C.byref(v))
    return bool(hit), t.value, u.value, v.value


def set_path_params(spp=4, max_bounces=3, seed=1234):
    """mode 200 parameters (process-wide in the oracle)"""
    lib().oracle_set_path_params(int(spp), int(max_bounces), int(seed))


def set_phong(ks_permille=0, exponent=32):
    """Phong specular term of mode 100 (process-wide, like the path parameters); 0 = off"""
    lib().oracle_set_phong(int(ks_permille), int(exponent))


def texture_color(t, u, v):
    keep = []
    x = make_texture(t, keep)
    out = np.zeros(3, dtype=np.float32)
    lib().oracle_texture_color(C.byref(x), float(np.float32(u)), float(np.float32(v)), out.ctypes.data)
    return out


def max_threads():
    return lib().oracle_max_threads()
