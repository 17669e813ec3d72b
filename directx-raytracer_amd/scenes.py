"""Seeded synthetic scenes for the configurations BASELINE.json names.

None of the named assets (Cornell box, Stanford bunny, 1M/5M-triangle meshes) ship with the reference --
its only scene is Scenes/Dragon.crtscene -- so they are generated here, deterministically, as plain numpy
arrays in the layout the reference hands to its vertex/index buffers (float32 xyz stride 12, uint32 indices;
R/DXRTRenderer.cpp:391-392,314-315).  This module only makes INPUTS; it renders nothing.
"""
import json
import math

import numpy as np

IDENTITY = np.eye(3, dtype=np.float32).reshape(9)


def _mesh(vertices, triangles, material_index=0, normals=None):
    return {"vertices": np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3),
            "triangles": np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 3),
            "material_index": int(material_index), "normals": normals}


def _scene(meshes, lights=(), materials=(), cam_pos=(0, 0, 0), cam_rot=IDENTITY, width=1920, height=1080,
           background=(0.0, 0.5, 0.0)):
    return {"meshes": list(meshes), "lights": list(lights), "materials": list(materials),
            "camera": {"position": np.asarray(cam_pos, dtype=np.float32), "matrix": np.asarray(cam_rot, dtype=np.float32).reshape(9)},
            "settings": {"width": int(width), "height": int(height), "background_color": tuple(background)}}


def camera_matrix(yaw_deg, pitch_deg):
    """Rotation with columns right/up/forward exactly as CRTCamera::rotate builds it from yaw/pitch
    (R/CRTCamera.cpp:57-87); the camera looks along -forward.  float64 maths, rounded to float32."""
    yaw, pitch = math.radians(yaw_deg), math.radians(pitch_deg)
    f = np.array([math.cos(pitch) * math.sin(yaw), math.sin(pitch), math.cos(pitch) * math.cos(yaw)])
    f /= np.linalg.norm(f)
    r = np.cross([0.0, 1.0, 0.0], f)
    r /= np.linalg.norm(r)
    u = np.cross(f, r)
    return np.stack([r, u, f], axis=1).astype(np.float32).reshape(9)


def load_crtscene(path):
    """.crtscene JSON -> arrays (test-harness reader; the product's loader is the C++ one behind crt_scene_load)."""
    with open(path) as f:
        d = json.load(f)
    type_ids = {"diffuse": 1, "reflective": 2, "constant": 4}
    meshes = [_mesh(np.array(o["vertices"], dtype=np.float32), np.array(o["triangles"], dtype=np.uint32),
                    o.get("material_index", 0)) for o in d.get("objects", [])]
    lights = [(tuple(l["position"]), float(l["intensity"])) for l in d.get("lights", [])]
    mats = []
    for m in d.get("materials", []):
        alb = m.get("albedo", [1, 1, 1])
        mats.append({"albedo": tuple(alb) if isinstance(alb, list) else (1.0, 1.0, 1.0),
                     "type": type_ids.get(m.get("type"), 3), "smooth_shading": bool(m.get("smooth_shading", False)),
                     "ior": float(m.get("ior", 1.0))})
    st = d.get("settings", {})
    img = st.get("image_settings", {})
    cam = d.get("camera", {})
    return _scene(meshes, lights, mats, cam.get("position", (0, 0, 0)), cam.get("matrix", IDENTITY),
                  img.get("width", 1920), img.get("height", 1080), st.get("background_color", (0, 0, 0)))


def vertex_normals(vertices, triangles):
    """CRTMesh::calculateVertexNormals (R/CRTMesh.cpp:66-94): sum of UNIT face normals per vertex, in triangle
    order, then normalised; float32 throughout, same operation order as CRTVector's operators."""
    v = np.asarray(vertices, dtype=np.float32)
    t = np.asarray(triangles, dtype=np.int64)
    n = np.zeros_like(v)
    f32 = np.float32
    for i0, i1, i2 in t:
        e0, e1 = v[i1] - v[i0], v[i2] - v[i0]
        c = np.array([e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2], e0[0] * e1[1] - e0[1] * e1[0]], dtype=f32)
        ln = np.sqrt(f32(c[0] * c[0]) + f32(c[1] * c[1]) + f32(c[2] * c[2]), dtype=f32)
        c = c / ln
        n[i0] += c
        n[i1] += c
        n[i2] += c
    ln = np.sqrt((n[:, 0] * n[:, 0] + n[:, 1] * n[:, 1]) + n[:, 2] * n[:, 2], dtype=f32)
    with np.errstate(invalid="ignore", divide="ignore"):
        return (n / ln[:, None]).astype(f32)


# ----------------------------------------------------------------------------------------------------
# C1: Cornell box, 32 triangles
# ----------------------------------------------------------------------------------------------------
def _quad(a, b, c, d):
    return [a, b, c, d], [(0, 1, 2), (0, 2, 3)]


def _box(cx, cz, sx, sz, h, angle_deg, y0):
    """5-face box (no bottom) standing on y0, rotated about Y."""
    a = math.radians(angle_deg)
    ca, sa = math.cos(a), math.sin(a)
    base = [(-sx, -sz), (sx, -sz), (sx, sz), (-sx, sz)]
    pts = [(cx + x * ca - z * sa, cz + x * sa + z * ca) for x, z in base]
    lo = [(x, y0, z) for x, z in pts]
    hi = [(x, y0 + h, z) for x, z in pts]
    verts, tris = [], []

    def add(q):
        o = len(verts)
        verts.extend(q)
        tris.extend([(o, o + 1, o + 2), (o, o + 2, o + 3)])

    add([hi[0], hi[1], hi[2], hi[3]])
    for i in range(4):
        j = (i + 1) % 4
        add([lo[i], lo[j], hi[j], hi[i]])
    return verts, tris


def cornell_box(width=256, height=256):
    """Cornell box: 5 walls (10 tris) + ceiling light quad (2) + short box (10) + tall box (10) = 32 triangles,
    camera identity in front of the open side looking down -Z (BASELINE.json configs[0])."""
    s = 5.0
    walls_v, walls_t = [], []

    def add(q, dest_v, dest_t):
        o = len(dest_v)
        dest_v.extend(q)
        dest_t.extend([(o, o + 1, o + 2), (o, o + 2, o + 3)])

    add([(-s, -s, s), (s, -s, s), (s, -s, -s), (-s, -s, -s)], walls_v, walls_t)      # floor
    add([(-s, s, s), (-s, s, -s), (s, s, -s), (s, s, s)], walls_v, walls_t)          # ceiling
    add([(-s, -s, -s), (s, -s, -s), (s, s, -s), (-s, s, -s)], walls_v, walls_t)      # back
    left_v, left_t, right_v, right_t = [], [], [], []
    add([(-s, -s, s), (-s, -s, -s), (-s, s, -s), (-s, s, s)], left_v, left_t)        # left (red)
    add([(s, -s, -s), (s, -s, s), (s, s, s), (s, s, -s)], right_v, right_t)          # right (green)
    light_v, light_t = [], []
    add([(-1.2, s - 0.01, 1.2), (-1.2, s - 0.01, -1.2), (1.2, s - 0.01, -1.2), (1.2, s - 0.01, 1.2)], light_v, light_t)
    sb_v, sb_t = _box(1.7, 1.5, 1.4, 1.4, 2.8, -18.0, -s)
    tb_v, tb_t = _box(-1.6, -1.4, 1.4, 1.4, 5.8, 17.0, -s)
    meshes = [_mesh(walls_v, walls_t, 0), _mesh(left_v, left_t, 1), _mesh(right_v, right_t, 2),
              _mesh(light_v, light_t, 3), _mesh(sb_v, sb_t, 0), _mesh(tb_v, tb_t, 0)]
    assert sum(len(m["triangles"]) for m in meshes) == 32
    mats = [{"albedo": (0.73, 0.73, 0.73), "type": 1}, {"albedo": (0.65, 0.05, 0.05), "type": 1},
            {"albedo": (0.12, 0.45, 0.15), "type": 1}, {"albedo": (1.0, 1.0, 1.0), "type": 4}]
    lights = [((0.0, 4.5, 0.0), 600.0)]
    return _scene(meshes, lights, mats, cam_pos=(0.0, 0.0, 15.0), cam_rot=IDENTITY, width=width, height=height)


def textured_cornell(width=320, height=240):
    """The Cornell box with per-vertex uvs and one texture of each kind the reference parses
    (R/CRTSceneParser.cpp:221-306): checker walls, edges left wall, bitmap right wall, albedo-texture boxes."""
    sc = cornell_box(width, height)
    quad_uv = np.float32([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]])
    for m in sc["meshes"]:
        n = len(m["vertices"])
        if n % 4 == 0 and n <= 12:
            m["uvs"] = np.tile(quad_uv, (n // 4, 1))
        else:  # the boxes: planar projection with values outside [0,1] and negative ones (floor() in the checker, clamps in the bitmap)
            v = np.asarray(m["vertices"], dtype=np.float32)
            m["uvs"] = np.stack([v[:, 0] * np.float32(0.37), v[:, 1] * np.float32(0.21) + v[:, 2] * np.float32(0.11), np.zeros(len(v), np.float32)], axis=1)
    yy, xx = np.mgrid[0:13, 0:17]
    pixels = np.stack([(xx * 15) % 256, (yy * 19) % 256, ((xx + yy) * 9) % 256], axis=-1).astype(np.uint8)
    sc["textures"] = [
        {"type": "checker", "color_a": (0.9, 0.9, 0.9), "color_b": (0.2, 0.2, 0.6), "scalar": 0.125},
        {"type": "edges", "color_a": (1.0, 0.9, 0.1), "color_b": (0.65, 0.05, 0.05), "scalar": 0.04},
        {"type": "bitmap", "pixels": pixels},
        {"type": "albedo", "color_a": (0.3, 0.8, 0.4)},
    ]
    sc["materials"][0]["texture"] = 0
    sc["materials"][1]["texture"] = 1
    sc["materials"][2]["texture"] = 2
    sc["materials"].append({"albedo": (0.5, 0.5, 0.5), "type": 1, "texture": 3})
    sc["materials"].append({"albedo": (0.9, 0.9, 0.9), "type": 2, "texture": 0})  # a mirror ignores its texture
    sc["meshes"][4]["material_index"] = 4
    sc["meshes"][5]["material_index"] = 5
    return sc


# ----------------------------------------------------------------------------------------------------
# C2: ~70k-triangle closed mesh (stand-in for the Stanford bunny, which cannot be fetched)
# ----------------------------------------------------------------------------------------------------
def displaced_sphere(n_lat=188, n_lon=188, radius=6.0, seed=7, width=1280, height=720):
    """UV sphere with seeded low-frequency radial displacement: 2*n_lon*(n_lat-1) = 70 312 triangles at 188x188."""
    rng = np.random.default_rng(seed)
    k = rng.uniform(1.0, 5.0, size=(6, 3))
    ph = rng.uniform(0, 2 * np.pi, size=6)
    amp = rng.uniform(0.03, 0.12, size=6)
    lat = np.linspace(0.0, np.pi, n_lat + 1)[1:-1]
    lon = np.linspace(0.0, 2 * np.pi, n_lon, endpoint=False)
    la, lo = np.meshgrid(lat, lon, indexing="ij")
    d = np.stack([np.sin(la) * np.cos(lo), np.cos(la), np.sin(la) * np.sin(lo)], axis=-1).reshape(-1, 3)
    d = np.concatenate([[[0.0, 1.0, 0.0]], d, [[0.0, -1.0, 0.0]]], axis=0)
    disp = 1.0 + sum(a * np.sin(d @ kk * 2.0 + p) for a, kk, p in zip(amp, k, ph))
    v = (d * (radius * disp)[:, None]).astype(np.float32)
    tris = []
    ring = lambda r: 1 + r * n_lon  # noqa: E731
    j = np.arange(n_lon)
    jn = (j + 1) % n_lon
    tris.append(np.stack([np.zeros(n_lon, dtype=np.int64), ring(0) + jn, ring(0) + j], axis=1))
    for r in range(n_lat - 2):
        a, b = ring(r), ring(r + 1)
        tris.append(np.stack([a + j, a + jn, b + j], axis=1))
        tris.append(np.stack([a + jn, b + jn, b + j], axis=1))
    last = len(v) - 1
    a = ring(n_lat - 2)
    tris.append(np.stack([np.full(n_lon, last, dtype=np.int64), a + j, a + jn], axis=1))
    t = np.concatenate(tris, axis=0).astype(np.uint32)
    ground = _mesh([(-15, -8, 15), (15, -8, 15), (-15, -8, -15), (15, -8, -15)], [(0, 1, 2), (3, 2, 1)], 0)
    mats = [{"albedo": (0.8, 0.8, 0.8), "type": 1}, {"albedo": (0.9, 0.6, 0.9), "type": 1}]
    lights = [((9.0, 12.0, 6.0), 3000.0)]
    return _scene([ground, _mesh(v, t, 1)], lights, mats, cam_pos=(0.0, 2.0, 16.0), cam_rot=camera_matrix(0.0, 8.0),
                  width=width, height=height)


# ----------------------------------------------------------------------------------------------------
# C3: ~1M-triangle height field (+ the Dragon scene's ground quad)
# ----------------------------------------------------------------------------------------------------
def _value_noise(x, z, seed, cells=24):
    rng = np.random.default_rng(seed)
    g = rng.uniform(-1.0, 1.0, size=(cells + 2, cells + 2))
    fx, fz = x * cells, z * cells
    ix, iz = np.floor(fx).astype(np.int64), np.floor(fz).astype(np.int64)
    tx, tz = fx - ix, fz - iz
    tx, tz = tx * tx * (3 - 2 * tx), tz * tz * (3 - 2 * tz)
    a, b = g[ix, iz], g[ix + 1, iz]
    c, d = g[ix, iz + 1], g[ix + 1, iz + 1]
    return (a * (1 - tx) + b * tx) * (1 - tz) + (c * (1 - tx) + d * tx) * tz


def heightfield(n=708, seed=1234, extent=15.0, width=1920, height=1080, n_lights=1):
    """(n x n quads) x 2 = 1 002 528 triangles at n=708 (BASELINE.json configs[2]); vertices (x, h(x,z), z) with
    h = 4 seeded sinusoids + value noise, scaled to the Dragon scene's +-15 extent; plus the Dragon ground quad.
    Camera (0,7,6) pitched 60 degrees down so that the mesh covers ~80 % of the frame (SURVEY.md proposed the
    Dragon default position, from which 3/4 of the rays only see sky and cost nothing). Light = first Dragon light."""
    rng = np.random.default_rng(seed)
    u = np.linspace(0.0, 1.0, n + 1)
    ux, uz = np.meshgrid(u, u, indexing="ij")
    h = np.zeros_like(ux)
    for _ in range(4):
        fx, fz = rng.uniform(1.0, 7.0, size=2)
        ph = rng.uniform(0, 2 * np.pi)
        h += rng.uniform(0.25, 1.0) * np.sin(2 * np.pi * (fx * ux + fz * uz) + ph)
    h += 1.5 * _value_noise(ux, uz, seed + 1)
    h = h / np.abs(h).max() * 2.5 - 1.0
    v = np.stack([(ux * 2 - 1) * extent, h, (uz * 2 - 1) * extent], axis=-1).reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a = (i * (n + 1) + j).reshape(-1)
    b, c, d = a + 1, a + (n + 1), a + (n + 2)
    t = np.concatenate([np.stack([a, b, c], axis=1), np.stack([d, c, b], axis=1)], axis=1).reshape(-1, 3).astype(np.uint32)
    ground = _mesh([(-15, -5, 15), (15, -5, 15), (-15, -5, -15), (15, -5, -15)], [(0, 1, 2), (3, 2, 1)], 0)
    mats = [{"albedo": (0.8, 0.8, 0.8), "type": 2}, {"albedo": (0.9, 0.6, 0.9), "type": 1}]
    lights = [((9.0, 7.0, 0.0), 2000.0), ((-9.0, 16.0, 0.0), 2000.0), ((0.0, 9.0, 7.5), 500.0), ((0.0, 9.0, -7.5), 500.0)][:n_lights]
    return _scene([ground, _mesh(v, t, 1)], lights, mats, cam_pos=(0.0, 7.0, 6.0), cam_rot=camera_matrix(0.0, 60.0),
                  width=width, height=height)


# ----------------------------------------------------------------------------------------------------
# C3b: triangle soup of copied icospheres (less coherent than a height field)
# ----------------------------------------------------------------------------------------------------
def _icosphere(subdiv):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    v = [tuple(np.array(p) / np.linalg.norm(p)) for p in v]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = (np.array(v[a]) + np.array(v[b])) / 2.0
                v.append(tuple(m / np.linalg.norm(m)))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, dtype=np.float64), np.array(f, dtype=np.int64)


def icosphere_soup(n_spheres=3125, subdiv=2, seed=4321, extent=14.0, width=1920, height=1080):
    """n_spheres copied (not instanced: the reference only has identity transforms, R/DXRTRenderer.cpp:701-703)
    subdiv-2 icospheres of 320 triangles each = 1 000 000 triangles at the defaults, seeded positions/radii."""
    sv, sf = _icosphere(subdiv)
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, size=(n_spheres, 3)) * np.array([1.0, 0.45, 1.0]) + np.array([0.0, 2.0, 0.0])
    r = rng.uniform(0.25, 0.7, size=n_spheres)
    v = (c[:, None, :] + sv[None, :, :] * r[:, None, None]).reshape(-1, 3).astype(np.float32)
    t = (sf[None, :, :] + (np.arange(n_spheres) * len(sv))[:, None, None]).reshape(-1, 3).astype(np.uint32)
    ground = _mesh([(-15, -5, 15), (15, -5, 15), (-15, -5, -15), (15, -5, -15)], [(0, 1, 2), (3, 2, 1)], 0)
    mats = [{"albedo": (0.8, 0.8, 0.8), "type": 1}, {"albedo": (0.6, 0.8, 0.9), "type": 1}]
    lights = [((9.0, 16.0, 6.0), 3000.0)]
    return _scene([ground, _mesh(v, t, 1)], lights, mats, cam_pos=(0.0, 14.0, 26.0), cam_rot=camera_matrix(0.0, 25.0),
                  width=width, height=height)


def single_triangle(width=64, height=64):
    """One triangle facing an identity camera: analytically checkable (SURVEY.md section 8c iii)."""
    m = _mesh([(-1.0, -1.0, -3.0), (1.0, -1.0, -3.0), (0.0, 1.0, -3.0)], [(0, 1, 2)], 0)
    return _scene([m], [((0.0, 0.0, 0.0), 100.0)], [{"albedo": (1, 1, 1), "type": 1}], width=width, height=height)
