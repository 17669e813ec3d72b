"""directx-raytracer_amd -- Python binding (ctypes) of libcrt_hip.so, the MI355X-native render loop.

The product is the C-ABI shared library (include/crt_hip.h; sources in csrc/).  This module is the thin
test/bench harness on top of it: `Scene` wraps the crt_scene_* scene layer (CRTScene / CRTCamera surface),
`Renderer` wraps the crt_ctx renderer (DXRTRenderer surface: upload, set camera, changeShadingMode, renderFrame).
There is no CPU fallback: if the library or a HIP device is missing, construction raises.

The directory name contains a hyphen, so it is loaded by path (see __graft_entry__.load_package()).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRT_HIP_LIBRARY") or os.path.join(_HERE, "libcrt_hip.so")  # the override is for A/B builds (tools/variant_build.sh)
MISS = 0xFFFFFFFF
MODE_LAMBERT = 100
MODE_PATH = 200
TILE = 16

NODE_DTYPE = np.dtype([("lx0", "f4"), ("lx1", "f4"), ("ly0", "f4"), ("ly1", "f4"),
                       ("rx0", "f4"), ("rx1", "f4"), ("ry0", "f4"), ("ry1", "f4"),
                       ("lz0", "f4"), ("lz1", "f4"), ("rz0", "f4"), ("rz1", "f4"),
                       ("left", "i4"), ("right", "i4"), ("pad0", "i4"), ("pad1", "i4")])
NODE4_DTYPE = np.dtype([("minx", "f4", 4), ("maxx", "f4", 4), ("miny", "f4", 4), ("maxy", "f4", 4), ("minz", "f4", 4), ("maxz", "f4", 4),
                        ("ref", "i4", 4), ("pad", "i4", 4)])
NODE4Q_DTYPE = np.dtype([("lo", "f4", 3), ("s", "f4", 3), ("qlo_x", "u4"), ("qhi_x", "u4"), ("qlo_y", "u4"), ("qhi_y", "u4"), ("qlo_z", "u4"), ("qhi_z", "u4"), ("ref", "i4", 4)])
assert NODE4Q_DTYPE.itemsize == 64
BVH_EMPTY = -1
TRI_DTYPE = np.dtype([("v0", "f4", 3), ("inst", "u4"), ("e1", "f4", 3), ("prim", "u4"),
                      ("e2", "f4", 3), ("gid", "u4")])
SHADE_DTYPE = np.dtype([("n0", "f4", 3), ("n1", "f4", 3), ("n2", "f4", 3), ("material", "u4"), ("pad", "u4", 2)])

# every symbol include/crt_hip.h declares (tests/test_abi.py checks the library exports all of them)
ABI_SYMBOLS = [
    "crt_abi_version", "crt_create", "crt_destroy", "crt_last_error", "crt_upload_scene", "crt_set_textures", "crt_bvh_export_uv", "crt_set_camera",
    "crt_set_shading_mode", "crt_set_miss_color", "crt_set_counting", "crt_set_option", "crt_debug_read_timeline", "crt_debug_read_counters", "crt_render_frame", "crt_render_frame_device",
    "crt_tile_count", "crt_tile_slots", "crt_render_tiles_device", "crt_render_frames_batch_device", "crt_render_tiles_batch_device",
    "crt_untile_device", "crt_untile_batch_device", "crt_set_stream", "crt_reset_stream",
    "crt_synchronize", "crt_bvh_info", "crt_bvh_export", "crt_bvh_build_host", "crt_free", "crt_host_alloc", "crt_host_free", "crt_bvh_info4", "crt_bvh_export4", "crt_bvh_export4q", "crt_bvh_quantize4", "crt_comm_unique_id", "crt_comm_init", "crt_comm_init_host", "crt_comm_destroy", "crt_comm_info", "crt_render_frame_distributed", "crt_bvh_build_host4", "crt_build_stats",
    "crt_scene_load", "crt_scene_save", "crt_scene_new", "crt_scene_free", "crt_scene_add_mesh", "crt_scene_add_light",
    "crt_scene_add_material", "crt_scene_mesh_count", "crt_scene_mesh", "crt_scene_light_count", "crt_scene_light",
    "crt_scene_material_count", "crt_scene_material", "crt_scene_texture_count", "crt_scene_texture_color", "crt_scene_add_texture",
    "crt_scene_set_material_texture", "crt_scene_set_mesh_uvs", "crt_scene_settings",
    "crt_scene_camera_get", "crt_scene_camera_set", "crt_scene_camera_rotate", "crt_scene_camera_zoom",
    "crt_scene_camera_move_forward", "crt_scene_camera_move_right", "crt_scene_camera_pan", "crt_scene_camera_tilt",
    "crt_scene_camera_roll", "crt_scene_camera_pan_around_target", "crt_upload_scene_from", "crt_set_camera_from",
]


class CrtError(RuntimeError):
    pass


class MeshView(C.Structure):
    _fields_ = [("xyz", C.c_void_p), ("idx", C.c_void_p), ("normals", C.c_void_p), ("uvs", C.c_void_p),
                ("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32), ("material_index", C.c_int32)]


class Light(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("intensity", C.c_float)]


class Material(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("type", C.c_uint32), ("smooth", C.c_uint32), ("ior", C.c_float), ("texture", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("type", C.c_uint32), ("color_a", C.c_float * 3), ("color_b", C.c_float * 3), ("scalar", C.c_float),
                ("pixels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint32)]


TEXTURE_TYPES = {"albedo": 0, "edges": 1, "checker": 2, "bitmap": 3}
UV_DTYPE = np.dtype([("uv0", "f4", 2), ("uv1", "f4", 2), ("uv2", "f4", 2)])


class FrameStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("total_ms", C.c_double), ("rays_primary", C.c_uint64),
                ("rays_shadow", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force=False):
    """Compile libcrt_hip.so (g++ host code + hipcc --offload-arch=gfx950 kernels) in-tree."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", csrc, "-j8", "all"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    """Load the shared library; never builds implicitly and never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CrtError("libcrt_hip.so is missing (%s): run __graft_entry__.build() / make -C csrc; "
                       "there is no CPU fallback" % LIB_PATH)
    # One HIP runtime per process: libcrt_hip.so needs libamdhip64.so.7 and takes whichever copy the process has
    # already loaded.  PyTorch bundles its own (torch/lib, same SONAME); if ours (/opt/rocm) were loaded first torch
    # would end up with a mixed runtime and report "No HIP GPUs".  So when torch is around, let it load first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, i32, f32 = C.c_void_p, C.c_uint32, C.c_int32, C.c_float
    sig = {
        "crt_abi_version": (u32, []),
        "crt_create": (C.c_int, [C.POINTER(vp), C.c_int]),
        "crt_destroy": (None, [vp]),
        "crt_last_error": (C.c_char_p, [vp]),
        "crt_upload_scene": (C.c_int, [vp, vp, u32, vp, u32, vp, u32]),
        "crt_set_camera": (C.c_int, [vp, vp, vp]),
        "crt_set_textures": (C.c_int, [vp, vp, u32]),
        "crt_bvh_export_uv": (C.c_int, [vp, vp, C.POINTER(C.c_int)]),
        "crt_scene_texture_color": (C.c_int, [vp, u32, f32, f32, vp]),
        "crt_scene_add_texture": (C.c_int, [vp, C.c_char_p, C.c_char_p, vp, vp, f32, C.c_char_p]),
        "crt_scene_set_material_texture": (C.c_int, [vp, u32, C.c_char_p]),
        "crt_scene_set_mesh_uvs": (C.c_int, [vp, u32, vp]),
        "crt_set_shading_mode": (C.c_int, [vp, u32]),
        "crt_set_miss_color": (C.c_int, [vp, vp]),
        "crt_set_counting": (C.c_int, [vp, C.c_int]),
        "crt_set_option": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "crt_debug_read_timeline": (C.c_int, [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
        "crt_debug_read_counters": (C.c_int, [vp, vp]),
        "crt_render_frame": (C.c_int, [vp, u32, u32, vp, vp, vp, vp, vp, vp]),
        "crt_render_frame_device": (C.c_int, [vp, u32, u32, vp, vp, vp, vp, vp, vp]),
        "crt_tile_count": (u32, [u32, u32]),
        "crt_tile_slots": (u32, [u32, u32, u32]),
        "crt_render_tiles_device": (C.c_int, [vp, u32, u32, u32, u32, vp, vp]),
        "crt_untile_batch_device": (C.c_int, [vp, u32, u32, u32, u32, u32, vp, vp]),
        "crt_render_frames_batch_device": (C.c_int, [vp, u32, u32, u32, vp, vp, vp]),
        "crt_render_tiles_batch_device": (C.c_int, [vp, u32, u32, u32, u32, u32, vp, vp, vp]),
        "crt_untile_device": (C.c_int, [vp, u32, u32, u32, vp, vp]),
        "crt_set_stream": (C.c_int, [vp, vp]),
        "crt_reset_stream": (C.c_int, [vp]),
        "crt_synchronize": (C.c_int, [vp]),
        "crt_bvh_info": (C.c_int, [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]),
        "crt_bvh_export": (C.c_int, [vp, vp, vp, vp]),
        "crt_bvh_build_host": (C.c_int, [vp, u32, C.POINTER(vp), C.POINTER(u32), C.POINTER(vp), C.POINTER(vp),
                                         C.POINTER(u32), C.POINTER(u32)]),
        "crt_free": (None, [vp]),
        "crt_host_alloc": (vp, [C.c_size_t]),
        "crt_host_free": (None, [vp]),
        "crt_bvh_info4": (C.c_int, [vp, C.POINTER(u32), C.POINTER(u32)]),
        "crt_build_stats": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "crt_bvh_export4": (C.c_int, [vp, vp]),
        "crt_bvh_export4q": (C.c_int, [vp, vp]),
        "crt_comm_unique_id": (C.c_int, [vp]),
        "crt_comm_init": (C.c_int, [vp, u32, u32, vp]),
        "crt_comm_init_host": (C.c_int, [vp, u32, u32, C.c_char_p]),
        "crt_comm_destroy": (C.c_int, [vp]),
        "crt_comm_info": (C.c_int, [vp, C.POINTER(u32), C.POINTER(u32)]),
        "crt_render_frame_distributed": (C.c_int, [vp, u32, u32, vp, vp, C.POINTER(FrameStats)]),
        "crt_bvh_quantize4": (C.c_int, [vp, u32, vp]),
        "crt_bvh_build_host4": (C.c_int, [vp, u32, C.POINTER(vp), C.POINTER(u32), C.POINTER(u32)]),
        "crt_scene_load": (C.c_int, [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_size_t]),
        "crt_scene_new": (C.c_int, [C.POINTER(vp)]),
        "crt_scene_save": (C.c_int, [vp, C.c_char_p, C.c_char_p, C.c_size_t]),
        "crt_scene_free": (None, [vp]),
        "crt_scene_add_mesh": (C.c_int, [vp, vp, u32, vp, u32, i32]),
        "crt_scene_add_light": (C.c_int, [vp, vp, f32]),
        "crt_scene_add_material": (C.c_int, [vp, vp]),
        "crt_scene_mesh_count": (u32, [vp]),
        "crt_scene_mesh": (C.c_int, [vp, u32, vp]),
        "crt_scene_light_count": (u32, [vp]),
        "crt_scene_light": (C.c_int, [vp, u32, vp]),
        "crt_scene_material_count": (u32, [vp]),
        "crt_scene_material": (C.c_int, [vp, u32, vp]),
        "crt_scene_texture_count": (u32, [vp]),
        "crt_scene_settings": (C.c_int, [vp, C.POINTER(u32), C.POINTER(u32), vp]),
        "crt_scene_camera_get": (C.c_int, [vp, vp, vp]),
        "crt_scene_camera_set": (C.c_int, [vp, vp, vp]),
        "crt_scene_camera_rotate": (C.c_int, [vp, f32, f32]),
        "crt_scene_camera_zoom": (C.c_int, [vp, f32]),
        "crt_scene_camera_move_forward": (C.c_int, [vp, f32]),
        "crt_scene_camera_move_right": (C.c_int, [vp, f32]),
        "crt_scene_camera_pan": (C.c_int, [vp, f32]),
        "crt_scene_camera_tilt": (C.c_int, [vp, f32]),
        "crt_scene_camera_roll": (C.c_int, [vp, f32]),
        "crt_scene_camera_pan_around_target": (C.c_int, [vp, f32, vp]),
        "crt_upload_scene_from": (C.c_int, [vp, vp]),
        "crt_set_camera_from": (C.c_int, [vp, vp]),
    }
    assert set(sig) == set(ABI_SYMBOLS)
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _f32(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert n is None or a.size == n
    return a


def tile_count(w, h):
    return ((w + TILE - 1) // TILE) * ((h + TILE - 1) // TILE)


def tile_slots(w, h, n_ranks):
    return (tile_count(w, h) + n_ranks - 1) // n_ranks


def untile_host(gathered, w, h, n_ranks, n_frames=1, frame=0):
    """numpy statement of the tile-major -> row-major de-interleave (crt_untile_device's layout contract):
    gathered = uint32[n_ranks, n_frames, slots, 16, 16]; macro tile k (row-major) of frame f lives at rank k % n_ranks,
    frame f, slot k // n_ranks (n_frames = 1: one frame per all-gather)."""
    slots = tile_slots(w, h, n_ranks)
    g = np.asarray(gathered, dtype=np.uint32).reshape(n_ranks, n_frames, slots, TILE, TILE)[:, frame]
    tx, ty = (w + TILE - 1) // TILE, (h + TILE - 1) // TILE
    k = np.arange(tx * ty)
    tiles = g[k % n_ranks, k // n_ranks]                       # [k, 16, 16]
    full = tiles.reshape(ty, tx, TILE, TILE).transpose(0, 2, 1, 3).reshape(ty * TILE, tx * TILE)
    return np.ascontiguousarray(full[:h, :w])


def tile_host(frame_u32, w, h, rank, n_ranks):
    """inverse for one rank: the staging buffer crt_render_tiles_device fills (pixels outside the frame = 0)."""
    slots = tile_slots(w, h, n_ranks)
    tx, ty = (w + TILE - 1) // TILE, (h + TILE - 1) // TILE
    pad = np.zeros((ty * TILE, tx * TILE), dtype=np.uint32)
    pad[:h, :w] = np.asarray(frame_u32, dtype=np.uint32).reshape(h, w)
    tiles = pad.reshape(ty, TILE, tx, TILE).transpose(0, 2, 1, 3).reshape(tx * ty, TILE, TILE)
    out = np.zeros((slots, TILE, TILE), dtype=np.uint32)
    mine = np.arange(rank, tx * ty, n_ranks)
    out[:len(mine)] = tiles[mine]
    return out


class Scene:
    """crt_scene handle: CRTScene / CRTSceneParser / CRTCamera surface (host only, no GPU needed)."""

    def __init__(self, path=None):
        L = lib()
        h = C.c_void_p()
        if path is None:
            rc = L.crt_scene_new(C.byref(h))
            if rc:
                raise CrtError("crt_scene_new rc=%d" % rc)
        else:
            err = C.create_string_buffer(512)
            rc = L.crt_scene_load(os.fsencode(path), C.byref(h), err, len(err))
            if rc:
                raise CrtError("crt_scene_load(%s) rc=%d: %s" % (path, rc, err.value.decode()))
        self.h = h

    @classmethod
    def from_arrays(cls, sc):
        """sc: dict as produced by scenes.py (meshes / lights / materials / camera)."""
        s = cls()
        for m in sc["meshes"]:
            s.add_mesh(m["vertices"], m["triangles"], m.get("material_index", 0))
        for pos, inten in sc.get("lights", []):
            s.add_light(pos, inten)
        for m in sc.get("materials", []):
            s.add_material(m.get("albedo", (1, 1, 1)), m.get("type", 1), m.get("smooth_shading", False), m.get("ior", 1.0))
        for i, m in enumerate(sc["meshes"]):
            if m.get("uvs") is not None:
                s.set_mesh_uvs(i, m["uvs"])
        cam = sc.get("camera")
        if cam is not None:
            s.set_camera(cam["position"], cam["matrix"])
        return s

    def close(self):
        if getattr(self, "h", None):
            lib().crt_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ok(self, rc, what):
        if rc:
            raise CrtError("%s rc=%d" % (what, rc))

    def save(self, path):
        """binary cache (.crtbin)"""
        err = C.create_string_buffer(512)
        rc = lib().crt_scene_save(self.h, os.fsencode(path), err, len(err))
        if rc:
            raise CrtError("crt_scene_save(%s) rc=%d: %s" % (path, rc, err.value.decode()))

    def add_mesh(self, vertices, triangles, material_index=0):
        v = _f32(vertices).reshape(-1, 3)
        t = np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 3)
        self._ok(lib().crt_scene_add_mesh(self.h, v.ctypes.data, len(v), t.ctypes.data, len(t), int(material_index)), "crt_scene_add_mesh")

    def add_light(self, pos, intensity):
        p = _f32(pos, 3)
        self._ok(lib().crt_scene_add_light(self.h, p.ctypes.data, float(intensity)), "crt_scene_add_light")

    def add_material(self, albedo=(1, 1, 1), type=1, smooth_shading=False, ior=1.0):
        m = Material((C.c_float * 3)(*[float(x) for x in albedo]), int(type), int(bool(smooth_shading)), float(ior), -1)
        self._ok(lib().crt_scene_add_material(self.h, C.byref(m)), "crt_scene_add_material")

    def add_texture(self, name, type, color_a=(0, 0, 0), color_b=(0, 0, 0), scalar=0.0, file_path=None):
        a, b = _f32(color_a, 3), _f32(color_b, 3)
        self._ok(lib().crt_scene_add_texture(self.h, name.encode(), type.encode(), a.ctypes.data, b.ctypes.data, float(scalar),
                                             os.fsencode(file_path) if file_path else None), "crt_scene_add_texture")

    def set_material_texture(self, material, texture_name):
        self._ok(lib().crt_scene_set_material_texture(self.h, int(material), texture_name.encode()), "crt_scene_set_material_texture")

    def set_mesh_uvs(self, mesh, uvs):
        u = _f32(uvs).reshape(-1, 3)
        self._ok(lib().crt_scene_set_mesh_uvs(self.h, int(mesh), u.ctypes.data), "crt_scene_set_mesh_uvs")

    def texture_color(self, i, u, v):
        out = np.zeros(3, dtype=np.float32)
        self._ok(lib().crt_scene_texture_color(self.h, int(i), float(np.float32(u)), float(np.float32(v)), out.ctypes.data), "crt_scene_texture_color")
        return out

    # ---- getters (CRTScene::getObjects / getLights / getMaterials / getTextures / getSettings)
    @property
    def mesh_count(self):
        return lib().crt_scene_mesh_count(self.h)

    def mesh(self, i):
        mv = MeshView()
        self._ok(lib().crt_scene_mesh(self.h, i, C.byref(mv)), "crt_scene_mesh")

        def arr(ptr, n, dt):
            if not ptr or n == 0:
                return None
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dt).copy()
        v = arr(mv.xyz, mv.n_vertices * 3, np.float32)
        t = arr(mv.idx, mv.n_triangles * 3, np.uint32)
        n = arr(mv.normals, mv.n_vertices * 3, np.float32)
        uv = arr(mv.uvs, mv.n_vertices * 3, np.float32)
        return {"uvs": uv.reshape(-1, 3) if uv is not None else None, "vertices": v.reshape(-1, 3) if v is not None else np.zeros((0, 3), np.float32),
                "triangles": t.reshape(-1, 3) if t is not None else np.zeros((0, 3), np.uint32),
                "normals": n.reshape(-1, 3) if n is not None else None, "material_index": mv.material_index}

    def meshes(self):
        return [self.mesh(i) for i in range(self.mesh_count)]

    def lights(self):
        out = []
        for i in range(lib().crt_scene_light_count(self.h)):
            l = Light()
            self._ok(lib().crt_scene_light(self.h, i, C.byref(l)), "crt_scene_light")
            out.append((tuple(l.pos), l.intensity))
        return out

    def materials(self):
        out = []
        for i in range(lib().crt_scene_material_count(self.h)):
            m = Material()
            self._ok(lib().crt_scene_material(self.h, i, C.byref(m)), "crt_scene_material")
            out.append({"albedo": tuple(m.albedo), "type": m.type, "smooth_shading": bool(m.smooth), "ior": m.ior, "texture": m.texture})
        return out

    @property
    def texture_count(self):
        return lib().crt_scene_texture_count(self.h)

    def settings(self):
        w, h = C.c_uint32(), C.c_uint32()
        bg = np.zeros(3, dtype=np.float32)
        self._ok(lib().crt_scene_settings(self.h, C.byref(w), C.byref(h), bg.ctypes.data), "crt_scene_settings")
        return {"width": w.value, "height": h.value, "background_color": tuple(bg)}

    # ---- camera (CRTCamera)
    def camera(self):
        pos = np.zeros(3, dtype=np.float32)
        rot = np.zeros(9, dtype=np.float32)
        self._ok(lib().crt_scene_camera_get(self.h, pos.ctypes.data, rot.ctypes.data), "crt_scene_camera_get")
        return pos, rot

    def set_camera(self, pos=None, rot=None):
        p = _f32(pos, 3) if pos is not None else None
        r = _f32(rot, 9) if rot is not None else None
        self._ok(lib().crt_scene_camera_set(self.h, p.ctypes.data if p is not None else None,
                                            r.ctypes.data if r is not None else None), "crt_scene_camera_set")

    def rotate(self, dyaw, dpitch):
        self._ok(lib().crt_scene_camera_rotate(self.h, dyaw, dpitch), "rotate")

    def zoom(self, a):
        self._ok(lib().crt_scene_camera_zoom(self.h, a), "zoom")

    def move_forward(self, d):
        self._ok(lib().crt_scene_camera_move_forward(self.h, d), "moveForward")

    def move_right(self, d):
        self._ok(lib().crt_scene_camera_move_right(self.h, d), "moveRight")

    def pan(self, deg):
        self._ok(lib().crt_scene_camera_pan(self.h, deg), "pan")

    def tilt(self, deg):
        self._ok(lib().crt_scene_camera_tilt(self.h, deg), "tilt")

    def roll(self, deg):
        self._ok(lib().crt_scene_camera_roll(self.h, deg), "roll")

    def pan_around_target(self, deg, target):
        t = _f32(target, 3)
        self._ok(lib().crt_scene_camera_pan_around_target(self.h, deg, t.ctypes.data), "panAroundTarget")


def _mesh_views(meshes, keep):
    arr = (MeshView * max(1, len(meshes)))()
    for i, m in enumerate(meshes):
        v = _f32(m["vertices"]).reshape(-1, 3)
        t = np.ascontiguousarray(m["triangles"], dtype=np.uint32).reshape(-1, 3)
        n = m.get("normals")
        if n is not None:
            n = _f32(n).reshape(-1, 3)
        uv = m.get("uvs")
        if uv is not None:
            uv = _f32(uv).reshape(-1, 3)
        keep += [v, t, n, uv]
        arr[i].xyz = v.ctypes.data
        arr[i].idx = t.ctypes.data
        arr[i].normals = n.ctypes.data if n is not None else None
        arr[i].uvs = uv.ctypes.data if uv is not None else None
        arr[i].n_vertices = len(v)
        arr[i].n_triangles = len(t)
        arr[i].material_index = int(m.get("material_index", 0))
    return arr


def build_bvh_host(meshes):
    """crt_bvh_build_host: the product's BVH builder, host only (no GPU). Returns nodes, tris, shade, max_depth."""
    L = lib()
    keep = []
    mv = _mesh_views(meshes, keep)
    pn, pt, ps = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nn, nt, md = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = L.crt_bvh_build_host(mv, len(meshes), C.byref(pn), C.byref(nn), C.byref(pt), C.byref(ps), C.byref(nt), C.byref(md))
    if rc:
        raise CrtError("crt_bvh_build_host rc=%d: %s" % (rc, L.crt_last_error(None).decode()))

    def take(p, n, dt):
        if n == 0:
            out = np.zeros(0, dtype=dt)
        else:
            out = np.frombuffer((C.c_char * (n * dt.itemsize)).from_address(p.value), dtype=dt).copy()
        L.crt_free(p)
        return out
    return take(pn, nn.value, NODE_DTYPE), take(pt, nt.value, TRI_DTYPE), take(ps, nt.value, SHADE_DTYPE), md.value


def build_bvh4_host(meshes):
    """crt_bvh_build_host4: binary build + collapse to the 4-wide tree the kernels traverse. Returns nodes4, depth4."""
    L = lib()
    keep = []
    mv = _mesh_views(meshes, keep)
    pn, nn, d4 = C.c_void_p(), C.c_uint32(), C.c_uint32()
    rc = L.crt_bvh_build_host4(mv, len(meshes), C.byref(pn), C.byref(nn), C.byref(d4))
    if rc:
        raise CrtError("crt_bvh_build_host4 rc=%d: %s" % (rc, L.crt_last_error(None).decode()))
    out = np.zeros(0, dtype=NODE4_DTYPE) if nn.value == 0 else \
        np.frombuffer((C.c_char * (nn.value * 128)).from_address(pn.value), dtype=NODE4_DTYPE).copy()
    L.crt_free(pn)
    return out, d4.value


def quantize4(nodes4):
    """crt_bvh_quantize4: wide nodes -> the 64-byte quantised nodes the kernels traverse (host only)"""
    nodes4 = np.ascontiguousarray(nodes4, dtype=NODE4_DTYPE)
    out = np.zeros(len(nodes4), dtype=NODE4Q_DTYPE)
    rc = lib().crt_bvh_quantize4(nodes4.ctypes.data, len(nodes4), out.ctypes.data)
    if rc != 0:
        raise CrtError("crt_bvh_quantize4 failed rc=%d" % rc)
    return out


class Renderer:
    """crt_ctx handle: the DXRTRenderer surface over HIP. Raises CrtError when no MI355X / HIP device is usable."""

    def __init__(self, device=0):
        L = lib()
        h = C.c_void_p()
        rc = L.crt_create(C.byref(h), int(device))
        if rc:
            raise CrtError("crt_create rc=%d: %s" % (rc, L.crt_last_error(None).decode()))
        self.h = h
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self._free_pinned()
            lib().crt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ok(self, rc, what):
        if rc:
            raise CrtError("%s rc=%d: %s" % (what, rc, lib().crt_last_error(self.h).decode()))

    def set_textures(self, textures):
        """textures: list of dicts {type: albedo|edges|checker|bitmap, color_a, color_b, scalar, pixels (H,W,C uint8)}"""
        keep = []
        arr = (Texture * max(1, len(textures)))()
        for i, t in enumerate(textures):
            arr[i].type = TEXTURE_TYPES[t["type"]]
            arr[i].color_a = (C.c_float * 3)(*[float(c) for c in t.get("color_a", (0, 0, 0))])
            arr[i].color_b = (C.c_float * 3)(*[float(c) for c in t.get("color_b", (0, 0, 0))])
            arr[i].scalar = float(t.get("scalar", 0.0))
            px = t.get("pixels")
            if px is not None:
                px = np.ascontiguousarray(px, dtype=np.uint8)
                keep.append(px)
                arr[i].pixels = px.ctypes.data
                arr[i].height, arr[i].width, arr[i].channels = px.shape
        self._ok(lib().crt_set_textures(self.h, arr, len(textures)), "crt_set_textures")

    def bvh_export_uv(self):
        info = self.bvh_info()
        uv = np.zeros(info["n_tris"], dtype=UV_DTYPE)
        has = C.c_int()
        self._ok(lib().crt_bvh_export_uv(self.h, uv.ctypes.data, C.byref(has)), "crt_bvh_export_uv")
        return uv if has.value else None

    def upload(self, meshes, lights=(), materials=(), textures=None):
        keep = []
        mv = _mesh_views(meshes, keep)
        larr = (Light * max(1, len(lights)))()
        for i, (p, inten) in enumerate(lights):
            larr[i].pos = (C.c_float * 3)(*[float(x) for x in p])
            larr[i].intensity = float(inten)
        marr = (Material * max(1, len(materials)))()
        for i, m in enumerate(materials):
            marr[i].albedo = (C.c_float * 3)(*[float(x) for x in m.get("albedo", (1, 1, 1))])
            marr[i].type = int(m.get("type", 1))
            marr[i].smooth = int(bool(m.get("smooth_shading", False)))
            marr[i].ior = float(m.get("ior", 1.0))
            marr[i].texture = int(m.get("texture", -1))
        self._ok(lib().crt_upload_scene(self.h, mv, len(meshes), larr, len(lights), marr, len(materials)), "crt_upload_scene")
        self.set_textures(list(textures) if textures else [])

    def upload_scene(self, scene):
        self._ok(lib().crt_upload_scene_from(self.h, scene.h), "crt_upload_scene_from")

    def set_camera(self, pos, rot):
        p, r = _f32(pos, 3), _f32(rot, 9)
        self._ok(lib().crt_set_camera(self.h, p.ctypes.data, r.ctypes.data), "crt_set_camera")

    def set_camera_from(self, scene):
        self._ok(lib().crt_set_camera_from(self.h, scene.h), "crt_set_camera_from")

    def change_shading_mode(self, mode):
        self._ok(lib().crt_set_shading_mode(self.h, int(mode)), "crt_set_shading_mode")

    def set_miss_color(self, rgb):
        c = _f32(rgb, 3)
        self._ok(lib().crt_set_miss_color(self.h, c.ctypes.data), "crt_set_miss_color")

    def set_counting(self, on):
        self._ok(lib().crt_set_counting(self.h, int(bool(on))), "crt_set_counting")

    def set_path_params(self, spp=4, max_bounces=3, seed=1234):
        """mode 200 (path tracing) parameters"""
        self.set_option("spp", spp)
        self.set_option("max_bounces", max_bounces)
        self.set_option("seed", seed)

    def set_option(self, name, value):
        self._ok(lib().crt_set_option(self.h, name.encode(), int(value)), "crt_set_option")

    def read_counters(self):
        buf = np.zeros(32, dtype=np.uint64)
        self._ok(lib().crt_debug_read_counters(self.h, buf.ctypes.data), "crt_debug_read_counters")
        return buf

    def read_timeline(self, max_words=1 << 22):
        buf = np.zeros(max_words, dtype=np.uint64)
        n = C.c_size_t()
        self._ok(lib().crt_debug_read_timeline(self.h, buf.ctypes.data, max_words, C.byref(n)), "crt_debug_read_timeline")
        return buf[:n.value].reshape(-1, 3)

    def set_stream(self, stream_ptr):
        """run on an external hipStream_t handle (0 / None = HIP's default stream, which is torch's default)"""
        self._ok(lib().crt_set_stream(self.h, stream_ptr), "crt_set_stream")

    def reset_stream(self):
        self._ok(lib().crt_reset_stream(self.h), "crt_reset_stream")

    def synchronize(self):
        self._ok(lib().crt_synchronize(self.h), "crt_synchronize")

    def bvh_info(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._ok(lib().crt_bvh_info(self.h, C.byref(a), C.byref(b), C.byref(c)), "crt_bvh_info")
        return {"n_nodes": a.value, "n_tris": b.value, "max_depth": c.value}

    def bvh_export(self):
        info = self.bvh_info()
        nodes = np.zeros(info["n_nodes"], dtype=NODE_DTYPE)
        tris = np.zeros(info["n_tris"], dtype=TRI_DTYPE)
        shade = np.zeros(info["n_tris"], dtype=SHADE_DTYPE)
        self._ok(lib().crt_bvh_export(self.h, nodes.ctypes.data, tris.ctypes.data, shade.ctypes.data), "crt_bvh_export")
        return nodes, tris, shade

    def build_stats(self):
        a, b = C.c_double(), C.c_double()
        self._ok(lib().crt_build_stats(self.h, C.byref(a), C.byref(b)), "crt_build_stats")
        return {"upload_ms": a.value, "device_build_ms": b.value}

    def bvh_export4(self):
        a, b = C.c_uint32(), C.c_uint32()
        self._ok(lib().crt_bvh_info4(self.h, C.byref(a), C.byref(b)), "crt_bvh_info4")
        nodes4 = np.zeros(a.value, dtype=NODE4_DTYPE)
        self._ok(lib().crt_bvh_export4(self.h, nodes4.ctypes.data), "crt_bvh_export4")
        return nodes4, b.value

    # native RCCL frame assembly (crt_comm_*): no torch involved
    def comm_init(self, rank, n_ranks, unique_id=None):
        """rank 0 may pass unique_id=None to create one; returns the 128-byte id (to be handed to the other ranks)"""
        if unique_id is None:
            buf = C.create_string_buffer(128)
            rc = lib().crt_comm_unique_id(buf)
            if rc != 0:
                raise CrtError("crt_comm_unique_id failed rc=%d: %s" % (rc, lib().crt_last_error(None).decode()))
            unique_id = buf.raw
        self._ok(lib().crt_comm_init(self.h, rank, n_ranks, C.create_string_buffer(unique_id, 128)), "crt_comm_init")
        return unique_id

    def comm_init_host(self, rank, n_ranks, name):
        """the same frame assembly with a POSIX shared-memory object ("/name") as the transport: ranks that share one GPU (rehearsal)"""
        self._ok(lib().crt_comm_init_host(self.h, rank, n_ranks, name.encode()), "crt_comm_init_host")

    def comm_destroy(self):
        self._ok(lib().crt_comm_destroy(self.h), "crt_comm_destroy")

    def render_frame_distributed(self, w, h, d_rgba8=None, host=False, stats=False):
        out = np.zeros((h, w, 4), dtype=np.uint8) if host else None
        st = FrameStats()
        self._ok(lib().crt_render_frame_distributed(self.h, w, h, d_rgba8, out.ctypes.data if host else None, C.byref(st) if stats else None),
                 "crt_render_frame_distributed")
        res = {"stats": st.as_dict()} if stats else {}
        if host:
            res["rgba8"] = out
        return res

    def bvh_export4q(self):
        """the quantised 64-byte nodes as they sit in HBM"""
        a = C.c_uint32()
        self._ok(lib().crt_bvh_info4(self.h, C.byref(a), None), "crt_bvh_info4")
        q = np.zeros(a.value, dtype=NODE4Q_DTYPE)
        self._ok(lib().crt_bvh_export4q(self.h, q.ctypes.data), "crt_bvh_export4q")
        return q

    def pinned_frame(self, w, h):
        """RGBA8 frame buffer in page-locked host memory (crt_host_alloc), reused across calls of render_frame(pinned=True)."""
        key = (w, h)
        if getattr(self, "_pinned_key", None) != key:
            self._free_pinned()
            ptr = lib().crt_host_alloc(w * h * 4)
            if not ptr:
                raise CrtError("crt_host_alloc failed")
            self._pinned_ptr, self._pinned_key = ptr, key
            self._pinned = np.ctypeslib.as_array((C.c_uint8 * (w * h * 4)).from_address(ptr)).reshape(h, w, 4)
        return self._pinned

    def _free_pinned(self):
        if getattr(self, "_pinned_ptr", None):
            self._pinned = None
            lib().crt_host_free(self._pinned_ptr)
            self._pinned_ptr, self._pinned_key = None, None

    def render_frame(self, w, h, want=("rgba8", "hit_inst", "hit_prim", "hit_t", "rgb"), pinned=False):
        """renderFrame with host outputs. Returns dict of arrays + 'stats'. pinned=True: rgba8 lands in a reused page-locked
        buffer (valid until the next such call)."""
        out = {"rgba8": self.pinned_frame(w, h) if pinned else np.zeros((h, w, 4), dtype=np.uint8)}
        if "hit_inst" in want:
            out["hit_inst"] = np.zeros((h, w), dtype=np.uint32)
        if "hit_prim" in want:
            out["hit_prim"] = np.zeros((h, w), dtype=np.uint32)
        if "hit_t" in want:
            out["hit_t"] = np.zeros((h, w), dtype=np.float32)
        if "rgb" in want:
            out["rgb"] = np.zeros((h, w, 3), dtype=np.float32)
        st = FrameStats()

        def p(k):
            return out[k].ctypes.data if k in out else None
        self._ok(lib().crt_render_frame(self.h, w, h, p("rgba8"), p("hit_inst"), p("hit_prim"), p("hit_t"), p("rgb"),
                                        C.byref(st)), "crt_render_frame")
        out["stats"] = st.as_dict()
        return out

    def render_frame_device(self, w, h, d_rgba8, d_hit_inst=None, d_hit_prim=None, d_hit_t=None, d_rgb=None, stats=False):
        """device pointers are integers (e.g. torch.Tensor.data_ptr())."""
        st = FrameStats() if stats else None
        self._ok(lib().crt_render_frame_device(self.h, w, h, d_rgba8, d_hit_inst, d_hit_prim, d_hit_t, d_rgb,
                                               C.byref(st) if stats else None), "crt_render_frame_device")
        return st.as_dict() if stats else None

    def render_tiles_device(self, w, h, rank, n_ranks, d_staging, stats=False):
        st = FrameStats() if stats else None
        self._ok(lib().crt_render_tiles_device(self.h, w, h, rank, n_ranks, d_staging, C.byref(st) if stats else None),
                 "crt_render_tiles_device")
        return st.as_dict() if stats else None

    @staticmethod
    def _batch_args(cameras, d_out):
        n = len(d_out)
        outs = (C.c_void_p * n)(*[int(x) for x in d_out])
        cams = None
        if cameras is not None:
            cams = np.ascontiguousarray(np.concatenate([np.concatenate([_f32(p, 3), _f32(r, 9)]) for p, r in cameras]), dtype=np.float32)
            assert cams.size == 12 * n
        return n, cams, outs

    def render_frames_batch_device(self, w, h, d_rgba8_list, cameras=None, stats=False):
        """several frames in ONE launch; cameras = [(pos, rot3x3), ...] per frame or None (current camera for all)."""
        n, cams, outs = self._batch_args(cameras, d_rgba8_list)
        st = FrameStats() if stats else None
        self._ok(lib().crt_render_frames_batch_device(self.h, w, h, n, cams.ctypes.data if cams is not None else None, outs,
                                                      C.byref(st) if stats else None), "crt_render_frames_batch_device")
        return st.as_dict() if stats else None

    def render_tiles_batch_device(self, w, h, rank, n_ranks, d_staging_list, cameras=None, stats=False):
        n, cams, outs = self._batch_args(cameras, d_staging_list)
        st = FrameStats() if stats else None
        self._ok(lib().crt_render_tiles_batch_device(self.h, w, h, rank, n_ranks, n, cams.ctypes.data if cams is not None else None, outs,
                                                     C.byref(st) if stats else None), "crt_render_tiles_batch_device")
        return st.as_dict() if stats else None

    def untile_batch_device(self, w, h, n_ranks, n_frames, frame, d_gathered, d_frame):
        self._ok(lib().crt_untile_batch_device(self.h, w, h, n_ranks, n_frames, frame, d_gathered, d_frame), "crt_untile_batch_device")

    def untile_device(self, w, h, n_ranks, d_gathered, d_frame):
        self._ok(lib().crt_untile_device(self.h, w, h, n_ranks, d_gathered, d_frame), "crt_untile_device")
