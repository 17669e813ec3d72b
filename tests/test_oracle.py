"""The CPU oracle against closed-form known answers.  The reference ships no tests, golden images or vectors
(SURVEY.md section 4), so the pins are: the HLSL source restated by hand here in numpy/float32 for chosen inputs
(rayGen, miss, the 7 closest-hit modes, UNORM store), analytically known scenes, and brute force == BVH."""
import math

import numpy as np
import pytest

f32 = np.float32
IDENT = np.eye(3, dtype=np.float32).reshape(9)


# ---------------------------------------------------------------------------------------------- rayGen
def _raygen_np(rot, px, py, w, h):
    """hlsl:21-55 in float64 (tolerance compare)."""
    x = (px + 0.5) / w
    y = (py + 0.5) / h
    x = 2 * x - 1
    y = 1 - 2 * y
    x *= w / h
    d = np.array([x, y, -1.0])
    d /= np.linalg.norm(d)
    dw = np.asarray(rot, dtype=np.float64).reshape(3, 3) @ d
    return dw / np.linalg.norm(dw)


def test_raygen_identity_camera(oracle):
    w, h = 1920, 1080
    for px, py in ((0, 0), (1919, 0), (0, 1079), (1919, 1079), (959, 539), (960, 540), (17, 803)):
        d = oracle.ray_dir(IDENT, px, py, w, h)
        np.testing.assert_allclose(d, _raygen_np(IDENT, px, py, w, h), rtol=0, atol=2e-7)
        assert abs(float(np.linalg.norm(d.astype(np.float64))) - 1.0) < 2e-7
    # centre pixels straddle the optical axis symmetrically and look down -Z (identity camera, hlsl:46)
    a, b = oracle.ray_dir(IDENT, 959, 539, w, h), oracle.ray_dir(IDENT, 960, 540, w, h)
    assert a[2] < -0.999 and a[0] == -b[0] and a[1] == -b[1]
    # fixed 90 degree vertical field of view: top row's centre is half a pixel inside tan = 1
    top = oracle.ray_dir(IDENT, 959, 0, w, h)
    assert abs(top[1] / -top[2] - (1 - 1 / h)) < 1e-6
    # horizontal extent scaled by the aspect ratio (hlsl:44)
    right = oracle.ray_dir(IDENT, 1919, 539, w, h)
    assert abs(right[0] / -right[2] - (w / h) * (1 - 1 / w)) < 1e-6


def test_raygen_uses_column_vector_convention(oracle, scenes):
    """dirWorld = R * dirCam with R's columns right/up/forward (R/DXRTRenderer.cpp:259-264, R/CRTCamera.cpp:81-86):
    a yawed camera's centre ray is -forward = -(column 2)."""
    rot = scenes.camera_matrix(30.0, 10.0)
    d = 0.5 * (oracle.ray_dir(rot, 959, 539, 1920, 1080) + oracle.ray_dir(rot, 960, 540, 1920, 1080))
    fwd = rot.reshape(3, 3)[:, 2]
    np.testing.assert_allclose(d / np.linalg.norm(d), -fwd, rtol=0, atol=1e-6)
    for px, py in ((3, 7), (1900, 1000)):
        np.testing.assert_allclose(oracle.ray_dir(rot, px, py, 1920, 1080), _raygen_np(rot, px, py, 1920, 1080), rtol=0, atol=3e-7)


# ------------------------------------------------------------------------------------------------ sin
def test_sin_contract_accuracy(oracle):
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-10, 10, 2000), rng.uniform(0, 1.4e7, 2000), rng.uniform(0, 5.6e10, 2000),
                         [0.0, 12.9898, 78.233, 4014 * 12.9898]]).astype(np.float32)
    for x in xs:
        got = oracle.sinf(x)
        ref = math.sin(float(x))  # the float value x exactly, in double
        ulp = np.spacing(np.float32(abs(ref))) if ref != 0 else 1e-45
        assert abs(got - ref) <= 1.0 * float(ulp) + 1e-12, (x, got, ref)
    assert oracle.sinf(0.0) == 0.0


# -------------------------------------------------------------------------------------------- shading
def _frac(x):
    return f32(x) - f32(np.floor(f32(x)))


def _hash_sin(oracle, x, k):
    return _frac(f32(f32(oracle.sinf(f32(x))) * f32(k)))


def test_mode0_random_triangle_colour(oracle):
    o, d = (0, 0, 0), (0, 0, -1)
    for prim in (0, 1, 2, 4013, 1002527):
        got = oracle.shade_mode(0, 0, prim, 1.0, 0.2, 0.3, o, d)
        fp = f32(prim)
        exp = [_hash_sin(oracle, fp * f32(k), 43758.5453) for k in (12.9898, 78.233, 45.164)]
        np.testing.assert_array_equal(got, f32(exp))
        assert np.all(got >= 0) and np.all(got < 1)
    np.testing.assert_array_equal(oracle.shade_mode(0, 5, 0, 1.0, 0, 0, o, d), f32([0, 0, 0]))  # sin(0) = 0


def test_mode1_object_cells(oracle):
    o, d, t = f32([1.0, 2.0, 3.0]), f32([0.6, 0.0, -0.8]), f32(7.5)
    got = oracle.shade_mode(1, 1, 99, t, 0, 0, o, d)
    wp = o + d * t
    cell = np.floor(wp / f32(2.0)).astype(np.int64)
    h = ((int(cell[0]) * 73856093) & 0xFFFFFFFF) ^ ((int(cell[1]) * 19349663) & 0xFFFFFFFF) ^ ((int(cell[2]) * 83492791) & 0xFFFFFFFF)
    var = _hash_sin(oracle, f32(np.uint32(h)) * f32(12.9898), 43758.5453)
    base = [_hash_sin(oracle, f32(1) * f32(12.9898), 43758.5453), _hash_sin(oracle, f32(1) * f32(78.233), 12345.6789),
            _hash_sin(oracle, f32(1) * f32(39.425), 34567.8901)]
    exp = [f32(b * f32(0.7)) + var * (f32(b * f32(1.3)) - f32(b * f32(0.7))) for b in base]
    np.testing.assert_allclose(got, f32(exp), rtol=0, atol=1e-7)
    # negative cells wrap like uint32 arithmetic (the HLSL multiplies signed ints that overflow, hlsl:107)
    got2 = oracle.shade_mode(1, 0, 0, f32(50.0), 0, 0, f32([-3, -1, 0]), f32([-0.6, -0.64, -0.48]))
    assert np.all(np.isfinite(got2))


def test_mode2_object_triangle(oracle):
    got = oracle.shade_mode(2, 1, 77, 3.0, 0.1, 0.1, (0, 0, 0), (0, 0, -1))
    base = [_hash_sin(oracle, f32(1) * f32(12.9898), 43758.5453), _hash_sin(oracle, f32(1) * f32(78.233), 12345.6789),
            _hash_sin(oracle, f32(1) * f32(39.425), 34567.8901)]
    shade = _hash_sin(oracle, f32(77) * f32(12.9898), 43758.5453)
    k = f32(0.6) + shade * (f32(1.0) - f32(0.6))
    np.testing.assert_allclose(got, f32([b * k for b in base]), rtol=0, atol=1e-7)


def test_mode3_barycentrics(oracle):
    got = oracle.shade_mode(3, 0, 0, 1.0, 0.25, 0.5, (0, 0, 0), (0, 0, -1))
    np.testing.assert_array_equal(got, f32([0.25, 0.25, 0.5]))


def test_mode4_height(oracle):
    for y, exp_h in ((-10.0, 0.0), (10.0, 1.0), (0.0, 0.5), (-30.0, 0.0), (50.0, 1.0)):
        got = oracle.shade_mode(4, 0, 0, 1.0, 0, 0, (0, y + 1.0, 0), (0, -1, 0))
        exp = [f32(a) + f32(exp_h) * (f32(0.9) - f32(a)) for a in (0.1, 0.2, 0.6)]
        np.testing.assert_allclose(got, f32(exp), rtol=0, atol=1e-7)


def test_mode5_distance(oracle):
    for t, c in ((0.0, 0.0), (10.0, 0.5), (20.0, 1.0), (400.0, 1.0)):
        np.testing.assert_allclose(oracle.shade_mode(5, 0, 0, t, 0, 0, (0, 0, 0), (0, 0, -1)), f32([c] * 3), atol=1e-7)


def test_mode6_checker_and_fallthrough(oracle):
    for (x, z), c in (((0.5, 0.5), 0.2), ((1.5, 0.5), 0.9), ((-0.5, 0.5), 0.9), ((-0.5, -0.5), 0.2), ((2.5, 1.5), 0.9)):
        got = oracle.shade_mode(6, 0, 0, 1.0, 0, 0, (x, 1.0, z), (0, -1, 0))
        np.testing.assert_array_equal(got, f32([c] * 3))
    np.testing.assert_array_equal(oracle.shade_mode(42, 0, 0, 1.0, 0, 0, (1.5, 1, 0.5), (0, -1, 0)), f32([0.9] * 3))


def test_unorm8_store(oracle):
    assert [oracle.unorm8(c) for c in (0.0, 1.0, -0.5, 2.0, float("nan"), 0.5, 0.2, 0.9)] == [0, 255, 0, 255, 0, 128, 51, 230]
    assert oracle.unorm8(1.3 * 0.9) == 255  # mode 1 can exceed 1 -> clamped by the UNORM store


# --------------------------------------------------------------------------------------- intersection
def test_moeller_trumbore_known_answers(oracle):
    v0, v1, v2 = (-1, -1, -3), (1, -1, -3), (0, 1, -3)
    hit, t, u, v = oracle.intersect_tri((0, -1 / 3, 0), (0, 0, -1), v0, v1, v2)
    assert hit and abs(t - 3) < 1e-6 and abs(u - 1 / 3) < 1e-6 and abs(v - 1 / 3) < 1e-6  # centroid
    hit, t, u, v = oracle.intersect_tri((0, -1 / 3, -6), (0, 0, 1), v0, v1, v2)
    assert hit and abs(t - 3) < 1e-6  # back face hits too: no culling (R/DXRTRenderer.cpp:590,697-699)
    assert not oracle.intersect_tri((2, 2, 0), (0, 0, -1), v0, v1, v2)[0]
    assert not oracle.intersect_tri((0, 0, 0), (1, 0, 0), v0, v1, v2)[0]            # parallel: det == 0
    assert not oracle.intersect_tri((0, 0, -2.9995), (0, 0, -1), v0, v1, v2)[0]     # t = 0.0005 < TMin (exclusive)
    assert not oracle.intersect_tri((0, 0, 0), (0, 0, -1), v0, v1, v2, tmax=3.0)[0]  # t == TMax excluded
    hit, t, u, v = oracle.intersect_tri((1, -1, 0), (0, 0, -1), v0, v1, v2)          # exactly on vertex v1
    assert hit and u == 1.0 and v == 0.0
    assert not oracle.intersect_tri((0, 0, 0), (0, 0, -1), (0, 0, -3), (0, 0, -3), (0, 0, -3))[0]  # degenerate


# ------------------------------------------------------------------------------ analytically known scenes
def test_single_triangle_scene(oracle, scenes):
    sc = scenes.single_triangle(64, 64)
    S = oracle.OracleScene(sc["meshes"])
    out = S.render((0, 0, 0), IDENT, 3, 64, 64)
    hit = out["hit_inst"] != oracle.MISS
    # rasterise analytically: pixel centre -> point on z=-3 plane -> inside test
    ys, xs = np.mgrid[0:64, 0:64]
    X = (2 * (xs + 0.5) / 64 - 1) * 3.0
    Y = (1 - 2 * (ys + 0.5) / 64) * 3.0
    inside = (Y >= -1) & (Y <= 1) & (np.abs(X) <= (1 - Y) / 2)
    edge = np.abs(np.abs(X) - (1 - Y) / 2) < 1e-4
    assert np.array_equal(hit[~edge], inside[~edge])
    assert hit.sum() > 100
    assert np.all(out["hit_prim"][hit] == 0) and np.all(out["hit_inst"][hit] == 0)
    np.testing.assert_array_equal(out["rgba8"][~hit], np.broadcast_to(np.uint8([0, 255, 255, 255]), (int((~hit).sum()), 4)))  # miss = cyan
    np.testing.assert_allclose(out["rgb"][hit].sum(axis=1), 1.0, atol=1e-6)  # barycentrics sum to 1
    assert np.all(out["rgba8"][..., 3] == 255)


def test_phong_term_closed_form(oracle, scenes):
    """Mode 100's optional Phong highlight (an extension: R/CRTMaterial.h:30-35 has no specular fields).  Light at the eye, the
    centre ray hits the triangle head on: mirror direction = view direction, so the highlight is exactly ks x the Lambert term;
    off by default; BVH == brute force with it on."""
    sc = scenes.single_triangle()
    cam = sc["camera"]
    O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    try:
        base = O.render(cam["position"], cam["matrix"], 100, 65, 65)["rgb"]
        oracle.set_phong(500, 7)
        ph = O.render(cam["position"], cam["matrix"], 100, 65, 65)
        brute = O.render(cam["position"], cam["matrix"], 100, 65, 65, brute_force=True)
        q = base[32, 32, 0]
        assert q > 0 and ph["rgb"][32, 32, 0] == np.float32(q + np.float32(0.5) * q)
        assert np.array_equal(ph["rgb"], brute["rgb"]) and np.array_equal(ph["rgba8"], brute["rgba8"])
        hit = ph["hit_inst"] != 0xFFFFFFFF
        assert np.all(ph["rgb"][hit] >= base[hit]) and np.array_equal(ph["rgb"][~hit], base[~hit])
        # a larger exponent narrows the highlight: off-centre pixels get less of it, the centre keeps all of it
        oracle.set_phong(500, 64)
        narrow = O.render(cam["position"], cam["matrix"], 100, 65, 65)["rgb"]
        assert narrow[32, 32, 0] == ph["rgb"][32, 32, 0] and np.all(narrow[hit] <= ph["rgb"][hit])
    finally:
        oracle.set_phong(0, 32)
    assert np.array_equal(O.render(cam["position"], cam["matrix"], 100, 65, 65)["rgb"], base)


def test_dragon_ground_plane_closed_form(oracle, dragon):
    """Pixels whose closest hit is the ground quad (instance 0, y = -5): t and the colours of modes 4,5,6 depend only
    on the plane, so they are checkable without any BVH (SURVEY.md section 8c iii)."""
    S = oracle.OracleScene(dragon["meshes"])
    cam = dragon["camera"]
    w, h = 480, 270
    outs = {m: S.render(cam["position"], cam["matrix"], m, w, h) for m in (4, 5, 6)}
    ground = outs[4]["hit_inst"] == 0
    assert ground.sum() > 5000
    ys, xs = np.nonzero(ground)
    d = np.array([_raygen_np(cam["matrix"], x, y, w, h) for x, y in zip(xs[::97], ys[::97])])
    t_exact = (-5.0 - float(cam["position"][1])) / d[:, 1]
    np.testing.assert_allclose(outs[4]["hit_t"][ys[::97], xs[::97]], t_exact, rtol=2e-6)
    wp = np.asarray(cam["position"], dtype=np.float64) + d * t_exact[:, None]
    hgt = np.clip((wp[:, 1] + 10) / 20, 0, 1)
    exp4 = np.stack([a + hgt * (0.9 - a) for a in (0.1, 0.2, 0.6)], axis=1)
    np.testing.assert_allclose(outs[4]["rgb"][ys[::97], xs[::97]], exp4, atol=1e-5)
    np.testing.assert_allclose(outs[5]["rgb"][ys[::97], xs[::97], 0], np.clip(t_exact * 0.05, 0, 1), atol=1e-5)
    fx, fz = wp[:, 0] - np.floor(wp[:, 0]), wp[:, 2] - np.floor(wp[:, 2])
    safe = (np.minimum(fx, 1 - fx) > 1e-3) & (np.minimum(fz, 1 - fz) > 1e-3)  # away from checker edges
    chk = (np.floor(wp[:, 0]).astype(np.int64) ^ np.floor(wp[:, 2]).astype(np.int64)) & 1
    np.testing.assert_allclose(outs[6]["rgb"][ys[::97], xs[::97], 0][safe], np.where(chk, 0.9, 0.2)[safe], atol=1e-6)


# ---------------------------------------------------------------------------------- BVH path pinned by brute force
@pytest.mark.parametrize("name", ["cornell", "dragon", "sphere_small"])
@pytest.mark.parametrize("mode", [3, 100])
def test_bvh_equals_brute_force(oracle, scenes, dragon, name, mode):
    if name == "cornell":
        sc, (w, h) = scenes.cornell_box(), (128, 128)
    elif name == "dragon":
        sc, (w, h) = dragon, (160, 90)
    else:
        sc, (w, h) = scenes.displaced_sphere(24, 24), (96, 54)
    if name == "dragon":
        sc = dict(sc)
        sc["meshes"] = [dict(m, normals=scenes.vertex_normals(m["vertices"], m["triangles"])) for m in sc["meshes"]]
    S = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    cam = sc["camera"]
    a = S.render(cam["position"], cam["matrix"], mode, w, h)
    b = S.render(cam["position"], cam["matrix"], mode, w, h, brute_force=True)
    for k in ("hit_inst", "hit_prim", "hit_t", "rgb", "rgba8"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert a["stats"]["rays_shadow"] == b["stats"]["rays_shadow"]
    assert a["stats"]["tris_tested"] < b["stats"]["tris_tested"]


def test_equal_t_tie_break_is_lowest_global_triangle(oracle):
    """Two coincident triangles in two meshes: DXR leaves the winner implementation-defined; the spec picks the lower
    global ordinal, independent of BVH order."""
    v = f32([(-1, -1, -3), (1, -1, -3), (0, 1, -3)])
    for order in (0, 1):
        meshes = [{"vertices": v, "triangles": [(0, 1, 2)]}, {"vertices": v, "triangles": [(0, 1, 2)]},
                  {"vertices": v + f32([5, 0, 0]), "triangles": [(0, 1, 2)]}]
        S = oracle.OracleScene(meshes)
        for bf in (False, True):
            out = S.render((0, 0, 0), IDENT, 3, 32, 32, brute_force=bf)
            hit = out["hit_inst"] != oracle.MISS
            assert hit.any() and np.all(out["hit_inst"][hit & (out["hit_inst"] < 2)] == 0)


def test_rows_subset_and_threads(oracle, scenes):
    sc = scenes.cornell_box()
    S = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    cam = sc["camera"]
    full = S.render(cam["position"], cam["matrix"], 100, 64, 64, n_threads=1)
    part = S.render(cam["position"], cam["matrix"], 100, 64, 64, rows=(3, 64, 8), n_threads=2)
    rows = np.arange(3, 64, 8)
    np.testing.assert_array_equal(part["rgba8"][rows], full["rgba8"][rows])
    assert part["stats"]["rays_primary"] == len(rows) * 64
    assert np.all(part["rgba8"][0] == 0)


def test_empty_scene_is_all_miss(oracle):
    S = oracle.OracleScene([])
    out = S.render((0, 0, 0), IDENT, 0, 8, 8)
    assert np.all(out["hit_inst"] == oracle.MISS) and np.all(out["rgba8"] == np.uint8([0, 255, 255, 255]))


# ------------------------------------------------------------------------------------------ path tracing (mode 200)
def _pcg_hash(v):
    state = (v * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return ((word >> 22) ^ word) & 0xFFFFFFFF


def test_path_tracer_is_deterministic_and_seeded(oracle, scenes):
    sc = scenes.cornell_box()
    cam = sc["camera"]
    S = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    oracle.set_path_params(4, 3, 1234)
    a = S.render(cam["position"], cam["matrix"], 200, 96, 96, miss_rgb=(0, 0, 0), n_threads=1)
    b = S.render(cam["position"], cam["matrix"], 200, 96, 96, miss_rgb=(0, 0, 0), n_threads=4)
    assert np.array_equal(a["rgb"], b["rgb"]) and a["stats"] == b["stats"]          # independent of threading
    oracle.set_path_params(4, 3, 99)
    c = S.render(cam["position"], cam["matrix"], 200, 96, 96, miss_rgb=(0, 0, 0))
    assert not np.array_equal(a["rgb"], c["rgb"])                                      # seeded
    oracle.set_path_params(4, 3, 1234)
    st = a["stats"]
    assert st["pixels"] == 96 * 96 and st["rays_primary"] >= 4 * 96 * 96 and st["rays_primary"] <= 4 * 4 * 96 * 96
    assert np.all(np.isfinite(a["rgb"])) and a["rgb"].min() >= 0.0
    # sample 0's camera ray is jittered inside its pixel: hits agree with the unjittered frame except at silhouettes
    ref = S.render(cam["position"], cam["matrix"], 3, 96, 96)
    assert (a["hit_inst"] == ref["hit_inst"]).mean() > 0.97


def test_path_tracer_zero_bounces_equals_direct_light_of_jittered_camera_rays(oracle, scenes):
    """max_bounces = 0: radiance = direct light at the first hit (diffuse), albedo (constant) or miss colour."""
    sc = scenes.cornell_box()
    cam = sc["camera"]
    S = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    oracle.set_path_params(1, 0, 7)
    a = S.render(cam["position"], cam["matrix"], 200, 64, 64, miss_rgb=(0.25, 0.5, 0.75))
    oracle.set_path_params(4, 3, 1234)
    miss = a["hit_inst"] == oracle.MISS
    np.testing.assert_array_equal(a["rgb"][miss], np.broadcast_to(np.float32([0.25, 0.5, 0.75]), (int(miss.sum()), 3)))
    light = a["hit_inst"] == 3  # the ceiling quad has a CONSTANT material: its albedo is returned
    assert light.any()
    np.testing.assert_array_equal(a["rgb"][light], np.broadcast_to(np.float32([1, 1, 1]), (int(light.sum()), 3)))
    assert a["stats"]["rays_primary"] == 64 * 64  # no bounce rays


def test_path_tracer_brute_force_equals_bvh(oracle, scenes):
    sc = scenes.cornell_box()
    sc["materials"][1] = {"albedo": (0.9, 0.9, 0.9), "type": 2}                    # left wall: mirror
    sc["materials"][2] = {"albedo": (1.0, 1.0, 1.0), "type": 3, "ior": 1.5}        # right wall: glass
    cam = sc["camera"]
    S = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    oracle.set_path_params(2, 3, 5)
    a = S.render(cam["position"], cam["matrix"], 200, 64, 64, miss_rgb=(0.1, 0.1, 0.1))
    b = S.render(cam["position"], cam["matrix"], 200, 64, 64, miss_rgb=(0.1, 0.1, 0.1), brute_force=True)
    oracle.set_path_params(4, 3, 1234)
    for k in ("hit_inst", "hit_prim", "hit_t", "rgb", "rgba8"):
        assert np.array_equal(a[k], b[k]), k
    assert a["stats"]["rays_primary"] == b["stats"]["rays_primary"] and a["stats"]["rays_shadow"] == b["stats"]["rays_shadow"]


def test_rng_known_answers(oracle):
    """pcg hash (RXS-M-XS 32) restated in Python: the first jitter of pixel 0 / sample 0 / seed 1234 is reproduced
    through the image: a 1x1 frame's camera ray direction identifies (jx, jy)."""
    st = _pcg_hash(0 ^ _pcg_hash(0 + _pcg_hash(1234)))
    st1 = _pcg_hash(st)
    st2 = _pcg_hash(st1)
    jx, jy = (st1 >> 8) * 2.0 ** -24, (st2 >> 8) * 2.0 ** -24
    assert 0 <= jx < 1 and 0 <= jy < 1
    # a big quad facing an identity camera at z=-2: mode-200 with 0 bounces and a CONSTANT material returns albedo,
    # and hit_t of sample 0 = 2 / -dir.z of the jittered ray
    quad = {"vertices": np.float32([(-50, -50, -2), (50, -50, -2), (50, 50, -2), (-50, 50, -2)]), "triangles": [(0, 1, 2), (0, 2, 3)], "material_index": 0}
    S = oracle.OracleScene([quad], [], [{"albedo": (0.5, 0.25, 0.125), "type": 4}])
    oracle.set_path_params(1, 0, 1234)
    out = S.render((0, 0, 0), IDENT, 200, 1, 1)
    oracle.set_path_params(4, 3, 1234)
    x = (2 * jx - 1) * 1.0
    y = 1 - 2 * jy
    t_exp = 2.0 * math.sqrt(x * x + y * y + 1)
    assert abs(float(out["hit_t"][0, 0]) - t_exp) < 1e-5
    np.testing.assert_array_equal(out["rgb"][0, 0], np.float32([0.5, 0.25, 0.125]))


def test_wide_and_binary_traversal_agree(oracle, scenes, dragon):
    """The 4-wide tree (what the kernels walk) is a collapse of the binary tree: both traversals must report the same
    hits, t and colours; the wide one takes about half the node fetches."""
    for sc, (w, h) in ((scenes.cornell_box(), (128, 128)), (dragon, (320, 180)), (scenes.displaced_sphere(40, 40), (160, 90))):
        S = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
        cam = sc["camera"]
        S.set_width(4)
        a = S.render(cam["position"], cam["matrix"], 100, w, h)
        S.set_width(2)
        b = S.render(cam["position"], cam["matrix"], 100, w, h)
        for k in ("hit_inst", "hit_prim", "hit_t", "rgb", "rgba8"):
            assert np.array_equal(a[k], b[k]), k
        assert a["stats"]["rays_shadow"] == b["stats"]["rays_shadow"]
        assert a["stats"]["nodes_visited"] < 0.75 * b["stats"]["nodes_visited"]


def test_ray_segments_cover_the_ray(oracle, scenes, dragon):
    """The idea behind split packets (csrc/split_packet.hip.h), on the oracle: a ray cut into K overlapping pieces is occluded
    exactly when one of its pieces is, BVH walk and brute force alike -- also for pieces that start in the middle of the geometry
    (a lower bound inside boxes and just in front of triangles)."""
    O = oracle.OracleScene(dragon["meshes"], dragon["lights"], dragon["materials"])
    rng = np.random.default_rng(3)
    f32 = np.float32
    n_occluded = 0
    for _ in range(400):
        o = f32(rng.uniform(-14, 14, 3)); o[1] = f32(rng.uniform(-5, 9))
        target = f32(rng.uniform(-6, 6, 3))
        d = target - o
        dist = f32(np.sqrt((d * d).sum(dtype=f32)))
        d = (d / dist).astype(f32)
        whole = oracle.occluded(O, o, d, 0.0, dist)
        assert whole == oracle.occluded(O, o, d, 0.0, dist, brute_force=True)
        n_occluded += whole
        for K in (4, 16):
            q = dist / f32(K)
            pieces = 0
            for k in range(K):
                a, b = q * f32(k), q * f32(k + 1)
                lo = f32(0) if k == 0 else a - abs(a) * f32(2.0 ** -10)
                hi = dist if k == K - 1 else b
                got = oracle.occluded(O, o, d, lo, hi)
                assert got == oracle.occluded(O, o, d, lo, hi, brute_force=True)
                pieces |= got
            assert pieces == whole
    assert 40 < n_occluded < 360
