"""Parity tests proper: the HIP path (through the C ABI of libcrt_hip.so) against the CPU oracle on the same inputs.
Bar (BASELINE.json north_star): hit ids identical, RGBA8 identical, float RGB within 1e-4 -- in practice the two
sides execute the same IEEE operation sequence and agree bit for bit, which is what is asserted for t and RGB too,
with the 1e-4 tolerance kept as the documented fallback for the float colour only."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-4  # north_star's tolerance for shaded floats
ALL_MODES = (0, 1, 2, 3, 4, 5, 6, 100)


@pytest.fixture(scope="module")
def renderer(pkg):
    r = pkg.Renderer(0)
    yield r
    r.close()


def _with_normals(scenes, sc):
    sc = dict(sc)
    sc["meshes"] = [dict(m, normals=scenes.vertex_normals(m["vertices"], m["triangles"])) for m in sc["meshes"]]
    return sc


def _compare(pkg, oracle, renderer, sc, w, h, modes, exact_float=True, count=True):
    cam = sc["camera"]
    renderer.upload(sc["meshes"], sc["lights"], sc["materials"])
    renderer.set_camera(cam["position"], cam["matrix"])
    O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    # the product's own BVH is what sits in HBM; the oracle built its own independently: they must be identical
    nodes, tris, shade = renderer.bvh_export()
    assert nodes.tobytes() == O.nodes().tobytes() and tris.tobytes() == O.tris().tobytes() and shade.tobytes() == O.shade().tobytes()
    nodes4, depth4 = renderer.bvh_export4()  # the wide tree, and its 64-byte quantised form: what is actually traversed
    assert nodes4.tobytes() == O.nodes4().tobytes() and depth4 == O.depth4
    assert renderer.bvh_export4q().tobytes() == O.nodes4q().tobytes()
    for mode in modes:
        renderer.change_shading_mode(mode)
        ref = O.render(cam["position"], cam["matrix"], mode, w, h)
        if count:  # the plain (non-instrumented) kernel variant is the product: it must give the same frame
            renderer.set_counting(False)
            plain = renderer.render_frame(w, h)
            for k in ("hit_inst", "hit_prim", "hit_t", "rgba8"):
                np.testing.assert_array_equal(plain[k], ref[k], err_msg="mode %d %s (plain kernel)" % (mode, k))
            assert np.array_equal(plain["rgb"], ref["rgb"], equal_nan=True), "mode %d rgb (plain kernel)" % mode
        renderer.set_counting(count)
        got = renderer.render_frame(w, h)
        np.testing.assert_array_equal(got["hit_inst"], ref["hit_inst"], err_msg="mode %d hit_inst" % mode)
        np.testing.assert_array_equal(got["hit_prim"], ref["hit_prim"], err_msg="mode %d hit_prim" % mode)
        np.testing.assert_array_equal(got["hit_t"], ref["hit_t"], err_msg="mode %d hit_t" % mode)
        np.testing.assert_array_equal(got["rgba8"], ref["rgba8"], err_msg="mode %d rgba8" % mode)
        err = np.abs(got["rgb"].astype(np.float64) - ref["rgb"].astype(np.float64))
        assert np.nanmax(err) <= RGB_TOL, "mode %d rgb err %g" % (mode, np.nanmax(err))
        if exact_float:
            assert np.array_equal(got["rgb"], ref["rgb"], equal_nan=True), "mode %d rgb not bit-exact" % mode
        if count:  # instrumented kernel variant counts exactly what the oracle's instrumented traversal counts
            st, rs = got["stats"], ref["stats"]
            assert st["rays_primary"] == rs["rays_primary"] == w * h
            assert (st["rays_shadow"], st["nodes_visited"], st["tris_tested"]) == (rs["rays_shadow"], rs["nodes_visited"], rs["tris_tested"]), mode
    renderer.set_counting(False)
    return O


def test_cornell_256_all_modes(pkg, oracle, scenes, renderer):
    """BASELINE.json configs[0]: Cornell box, 32 triangles, 256x256, primary rays (+ the other modes)."""
    _compare(pkg, oracle, renderer, scenes.cornell_box(), 256, 256, ALL_MODES)


def test_dragon_1080p_all_modes(pkg, oracle, scenes, dragon, renderer):
    """The only input the reference itself defines: Dragon.crtscene, 1920x1080 (R/DXRTRenderer.cpp:1348-1350),
    smooth-shaded materials with the reference's vertex normals for mode 100."""
    _compare(pkg, oracle, renderer, _with_normals(scenes, dragon), 1920, 1080, ALL_MODES)


def test_dragon_through_scene_loader(pkg, oracle, scenes, dragon, renderer, golden_dir):
    """Same scene through crt_scene_load + crt_upload_scene_from (the path a reference-side caller takes)."""
    s = pkg.Scene(os.path.join(golden_dir, "dragon.crtscene"))
    renderer.upload_scene(s)
    s.rotate(25.0, -10.0)
    s.move_forward(-3.0)
    renderer.set_camera_from(s)
    pos, rot = s.camera()
    sc = _with_normals(scenes, dragon)
    O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    for mode in (0, 100):
        renderer.change_shading_mode(mode)
        got = renderer.render_frame(640, 360)
        ref = O.render(pos, rot, mode, 640, 360)
        for k in ("hit_inst", "hit_prim", "rgba8", "hit_t"):
            np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
        assert np.array_equal(got["rgb"], ref["rgb"], equal_nan=True)


def test_bunny_standin_720p(pkg, oracle, scenes, renderer):
    """BASELINE.json configs[1] (70k triangles, 1280x720); the Stanford bunny cannot be fetched: seeded stand-in."""
    sc = scenes.displaced_sphere()
    assert 69000 < sum(len(m["triangles"]) for m in sc["meshes"]) < 72000
    _compare(pkg, oracle, renderer, sc, 1280, 720, (0, 3, 100))


def test_heightfield_1m_1080p(pkg, oracle, scenes, renderer):
    """BASELINE.json configs[2] at full size: 1 002 530 triangles, 1920x1080, primary + shadow rays."""
    sc = scenes.heightfield()
    assert sum(len(m["triangles"]) for m in sc["meshes"]) == 1002530
    O = _compare(pkg, oracle, renderer, sc, 1920, 1080, (3, 100))
    # size-independent properties at full size
    renderer.change_shading_mode(5)
    a = renderer.render_frame(1920, 1080)
    b = renderer.render_frame(1920, 1080)
    for k in ("rgba8", "hit_prim", "hit_t", "rgb"):
        assert np.array_equal(a[k], b[k]), "render is not idempotent: " + k
    hit = a["hit_inst"] != pkg.MISS
    assert 0.7 < hit.mean() < 0.9
    np.testing.assert_array_equal(a["rgb"][..., 0][hit], np.clip(a["hit_t"][hit] * np.float32(0.05), 0, 1))  # mode 5 = saturate(t/20)
    assert np.all(a["hit_t"][~hit] == np.float32(10000.0)) and np.all(a["hit_t"][hit] > 0.001)
    assert np.all(a["rgba8"][~hit] == np.uint8([0, 255, 255, 255]))


def test_icosphere_soup_1m(pkg, oracle, scenes, renderer):
    """Less coherent 1M-triangle variant (3 125 copied icospheres), 1920x1080, Lambert + shadow."""
    sc = scenes.icosphere_soup()
    assert sum(len(m["triangles"]) for m in sc["meshes"]) == 1000002
    _compare(pkg, oracle, renderer, sc, 1920, 1080, (100,))


@pytest.mark.parametrize("w,h", [(1, 1), (17, 33), (15, 16), (257, 130), (1000, 3)])
def test_ragged_frame_sizes(pkg, oracle, scenes, renderer, w, h):
    """frame sizes that are not multiples of the 16x16 macro tile / 8x8 wavefront tile"""
    _compare(pkg, oracle, renderer, scenes.cornell_box(), w, h, (2, 100), count=True)


def test_empty_and_tiny_scenes(pkg, oracle, scenes, renderer):
    f = np.float32
    empty = {"meshes": [], "lights": [], "materials": [], "camera": {"position": f([0, 0, 0]), "matrix": scenes.IDENTITY}}
    _compare(pkg, oracle, renderer, empty, 64, 48, (0, 100))
    _compare(pkg, oracle, renderer, scenes.single_triangle(), 64, 64, ALL_MODES)
    # coincident triangles in different meshes: equal-t tie goes to the lower global triangle ordinal
    v = f([(-1, -1, -3), (1, -1, -3), (0, 1, -3)])
    tie = dict(empty, meshes=[{"vertices": v, "triangles": [(0, 1, 2)]}, {"vertices": v, "triangles": [(0, 1, 2)]}])
    O = _compare(pkg, oracle, renderer, tie, 64, 64, (3,))
    out = renderer.render_frame(64, 64)
    assert set(np.unique(out["hit_inst"]).tolist()) == {0, pkg.MISS}


def test_degenerate_inputs(pkg, oracle, scenes, renderer):
    """zero-area triangles, axis-parallel rays through box faces, camera inside geometry bounds"""
    f = np.float32
    v = f([(0, 0, -5), (0, 0, -5), (0, 0, -5), (-2, -2, -4), (2, -2, -4), (0, 2, -4), (-50, -1, -50), (50, -1, -50), (0, -1, 50)])
    sc = {"meshes": [{"vertices": v, "triangles": [(0, 1, 2), (3, 4, 5), (6, 7, 8)]}], "lights": [((0, 5, 0), 300.0)],
          "materials": [{"albedo": (0.5, 0.6, 0.7)}], "camera": {"position": f([0, 0, 0]), "matrix": scenes.IDENTITY}}
    _compare(pkg, oracle, renderer, sc, 65, 65, ALL_MODES)  # odd size: the centre pixel's ray is exactly (0,0,-1)


def test_tile_partition_reassembles_full_frame(pkg, scenes, dragon, renderer):
    """N-GPU path on one GPU: every rank's tile launch + de-interleave kernel == the single-launch frame."""
    import torch
    sc = dragon
    renderer.upload(sc["meshes"], sc["lights"], sc["materials"])
    renderer.set_camera(sc["camera"]["position"], sc["camera"]["matrix"])
    renderer.change_shading_mode(100)
    for (w, h) in ((1920, 1080), (333, 77)):
        full = torch.zeros(h * w, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()  # the renderer's own stream does not order with torch's: the fill must have landed
        renderer.render_frame_device(w, h, full.data_ptr())
        renderer.synchronize()
        for n in (1, 2, 3, 8):
            slots = pkg.tile_slots(w, h, n)
            gathered = torch.zeros(n * slots * 256, dtype=torch.int32, device="cuda")
            frame = torch.zeros(h * w, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            for rank in range(n):
                st = renderer.render_tiles_device(w, h, rank, n, gathered.data_ptr() + rank * slots * 1024, stats=True)
                assert st["rays_primary"] > 0
            renderer.untile_device(w, h, n, gathered.data_ptr(), frame.data_ptr())
            renderer.synchronize()
            assert torch.equal(frame, full), (w, h, n)
            host = pkg.untile_host(gathered.cpu().numpy().view(np.uint32), w, h, n)
            assert np.array_equal(host.reshape(-1), full.cpu().numpy().view(np.uint32))


def test_c3_tiled_over_2_4_8_ranks_vs_oracle(pkg, oracle, scenes, renderer):
    """BASELINE.json configs[3]: the 1M-triangle mesh at 1920x1080, framebuffer tiled over 2 / 4 / 8 ranks (every rank's
    launch run on this one GPU), rank buffers laid out as the RCCL all-gather leaves them, de-interleaved by the untile
    kernel: the reassembled frame must be the ORACLE's frame, pixel for pixel.  Replaces the reference's single
    DispatchRays (R/DXRTRenderer.cpp:1346-1350, 1405)."""
    import torch
    sc = scenes.heightfield()
    assert sum(len(m["triangles"]) for m in sc["meshes"]) == 1002530
    cam = sc["camera"]
    w, h = 1920, 1080
    renderer.upload(sc["meshes"], sc["lights"], sc["materials"])
    renderer.set_camera(cam["position"], cam["matrix"])
    renderer.change_shading_mode(100)
    ref = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"]).render(cam["position"], cam["matrix"], 100, w, h, want=("rgba8",))
    ref = np.ascontiguousarray(ref["rgba8"]).view(np.uint32).reshape(-1)
    for n in (2, 4, 8):
        slots = pkg.tile_slots(w, h, n)
        assert slots * 1024 == {2: 4177920, 4: 2088960, 8: 1044480}[n]  # bytes each rank contributes to the all-gather
        gathered = torch.zeros(n * slots * 256, dtype=torch.int32, device="cuda")
        frame = torch.zeros(h * w, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        rays = 0
        for rank in range(n):
            rays += renderer.render_tiles_device(w, h, rank, n, gathered.data_ptr() + rank * slots * 1024, stats=True)["rays_primary"]
        assert rays == w * h  # the ranks' shares partition the frame
        renderer.untile_device(w, h, n, gathered.data_ptr(), frame.data_ptr())
        renderer.synchronize()
        np.testing.assert_array_equal(frame.cpu().numpy().view(np.uint32), ref, err_msg="n_ranks=%d" % n)
        # batched variant (several frames per launch, as bench.py does from 8 ranks up): same frames
        if n == 8:
            nb = 4  # bench.py's frames per launch at 8 ranks
            g2 = torch.zeros(n * nb * slots * 256, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            for rank in range(n):
                base = g2.data_ptr() + rank * nb * slots * 1024
                renderer.render_tiles_batch_device(w, h, rank, n, [base + f * slots * 1024 for f in range(nb)])
            for f in range(nb):
                renderer.untile_batch_device(w, h, n, nb, f, g2.data_ptr(), frame.data_ptr())
                renderer.synchronize()
                np.testing.assert_array_equal(frame.cpu().numpy().view(np.uint32), ref, err_msg="batched frame %d" % f)


def test_rccl_gather_path_of_bench_in_a_fresh_process():
    """The real N>1 code path of bench.py -- RCCL communicator init, tile staging, ONE all-gather per launch, de-interleave --
    in a fresh child process with world size 1 (all this box has) and NO launcher environment (no WORLD_SIZE / RANK / MASTER_*
    preset: bench.py finds its own rendezvous, as when the driver calls `python bench.py --gpus N`); the child compares the
    gathered frame with the oracle's and exits non-zero on any difference.  Both pipelining policies run (1 launch in flight,
    and 4 in flight)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "6", "--warmup", "2",
                          "--no-cpu-baseline", "--check-dist-frame"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["frame_matches_oracle"] is True
    assert line["config"]["launches_in_flight"] == 4 and line["config"]["frames_per_launch"] == 1  # the N > 1 default policy of `value`
    assert line["one_launch_in_flight"]["launches_in_flight"] == 1 and line["one_launch_in_flight"]["value"] > 0
    assert line["rank0_render_kernel_ms"] > 0 and line["rank0_gather_untile_us"] > 0
    assert "all-gather" in line["config"]["parallelism"] or line["n_gpus"] == 1


def test_bench_three_ranks_rehearsed_on_one_gpu():
    """`python bench.py --gpus 3` with no launcher environment: bench.py starts three rank processes itself; with --backend gloo
    --same-device they all render on GPU 0 and exchange their tiles through host memory (RCCL refuses several ranks on one GPU).
    Everything of the N > 1 path except the collective's transport runs as it will on three GPUs: the rank-dependent tile shares
    with split packets, four launches in flight on four streams, the spin-up whose length rank 0 broadcasts, the per-rank
    barriers, the one-launch-in-flight leg, and rank 0's comparison of the gathered frame with the oracle's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--backend", "gloo", "--same-device", "--steps", "8",
                          "--warmup", "2", "--spinup-ms", "30", "--no-cpu-baseline", "--check-dist-frame"],
                         capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 3 and line["frame_matches_oracle"] is True and line["frame_rows_checked"] == 1080
    assert line["config"]["launches_in_flight"] == 4 and line["one_launch_in_flight"]["value"] > 0
    assert line["message_bytes_per_rank"] == 2720 * 1024 and "rehearsal" in line


def test_bench_c5_config_runs_and_matches_the_oracle_on_sampled_rows():
    """`bench.py --config c5` (BASELINE.json configs[4]: 5M triangles, 3840x2160, 4 spp, 3 bounces) through the N>1 path with
    the one rank a box has: the tile-partitioned path-traced frame equals the oracle's on the rows the CPU sample covers."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "c5", "--gpus", "1", "--force-dist", "--steps", "3", "--warmup", "1",
                          "--no-extras", "--no-cpu-baseline", "--check-dist-frame", "--spinup-ms", "0"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["frame_matches_oracle"] is True and line["frame_rows_checked"] >= 8
    assert "4999124 triangles, 3840x2160" in line["config"]["workload"] and line["config"]["rays_per_frame"] > 4 * 3840 * 2160


def test_obj_scene_renders_like_the_oracle(pkg, oracle, scenes, renderer, tmp_path):
    """north_star: "so the same .obj scenes render": a Wavefront .obj written here (two objects, quads and triangles,
    negative and v/vt/vn index forms) goes through crt_scene_load -> crt_upload_scene_from -> the HIP kernels; the frames
    equal the oracle's on the arrays the loader produced."""
    sp = scenes.displaced_sphere(n_lat=24, n_lon=32)["meshes"][1]  # the sphere (mesh 0 is its ground quad)
    lines = ["# written by tests/test_gpu_parity.py", "o blob"]
    lines += ["v %.9g %.9g %.9g" % tuple(float(c) for c in v) for v in sp["vertices"]]
    nv = len(sp["vertices"])
    for i, t in enumerate(sp["triangles"]):
        a, b, c = (int(x) for x in t)
        if i % 3 == 0:
            lines.append("f %d %d %d" % (a + 1, b + 1, c + 1))
        elif i % 3 == 1:
            lines.append("f %d %d %d" % (a - nv, b - nv, c - nv))                 # negative (relative) indices
        else:
            lines.append("f %d/1/1 %d/1/1 %d/1/1" % (a + 1, b + 1, c + 1))       # v/vt/vn form
    lines += ["o floor", "v -20 -7 -20", "v 20 -7 -20", "v 20 -7 20", "v -20 -7 20", "f -4 -1 -2 -3"]  # a quad: split into two triangles
    path = tmp_path / "blob.obj"
    path.write_text("\n".join(lines) + "\n")
    s = pkg.Scene(str(path))
    assert s.mesh_count == 2 and len(s.mesh(1)["triangles"]) == 2
    s.add_light((6.0, 9.0, 4.0), 900.0)
    s.set_camera(pos=np.float32([0, 2, 16]), rot=scenes.camera_matrix(0.0, 8.0))
    renderer.upload_scene(s)
    renderer.set_camera_from(s)
    pos, rot = s.camera()
    meshes = [dict(vertices=m["vertices"], triangles=m["triangles"], normals=m["normals"], material_index=m["material_index"]) for m in s.meshes()]
    O = oracle.OracleScene(meshes, s.lights(), s.materials())
    for mode in (0, 3, 100):
        renderer.change_shading_mode(mode)
        got = renderer.render_frame(640, 360)
        ref = O.render(pos, rot, mode, 640, 360)
        for k in ("hit_inst", "hit_prim", "hit_t", "rgba8"):
            np.testing.assert_array_equal(got[k], ref[k], err_msg="mode %d %s" % (mode, k))
        assert np.array_equal(got["rgb"], ref["rgb"], equal_nan=True)
    assert (got["hit_inst"] == 0).any() and (got["hit_inst"] == 1).any()  # both objects are in view


def test_phong_highlight_matches_the_oracle(pkg, oracle, scenes, dragon, renderer):
    """north_star's "Lambert/Phong shading": mode 100 with the specular options on (Cornell flat normals, Dragon smooth
    normals, several coefficients / exponents) is bit-exact against the oracle; with ks = 0 it is the plain Lambert frame."""
    try:
        for sc, w, h in ((scenes.cornell_box(), 256, 256), (_with_normals(scenes, dragon), 640, 360)):
            cam = sc["camera"]
            renderer.upload(sc["meshes"], sc["lights"], sc["materials"])
            renderer.set_camera(cam["position"], cam["matrix"])
            renderer.change_shading_mode(100)
            O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
            plain = renderer.render_frame(w, h)["rgb"].copy()
            for ks, n in ((300, 32), (1000, 1), (50, 1000), (700, 5)):
                renderer.set_option("phong_ks", ks)
                renderer.set_option("phong_exponent", n)
                oracle.set_phong(ks, n)
                got = renderer.render_frame(w, h)
                ref = O.render(cam["position"], cam["matrix"], 100, w, h)
                for k in ("hit_inst", "hit_prim", "hit_t", "rgba8"):
                    np.testing.assert_array_equal(got[k], ref[k], err_msg="ks=%d n=%d %s" % (ks, n, k))
                assert np.array_equal(got["rgb"], ref["rgb"], equal_nan=True), (ks, n)
                assert (got["rgb"] > plain).any() and not (got["rgb"] < plain).any()
            renderer.set_option("phong_ks", 0)
            assert np.array_equal(renderer.render_frame(w, h)["rgb"], plain)
        with pytest.raises(pkg.CrtError):
            renderer.set_option("phong_exponent", 0)
    finally:
        oracle.set_phong(0, 32)
        renderer.set_option("phong_ks", 0)
        renderer.set_option("phong_exponent", 32)


def test_pinned_host_frame(pkg, oracle, scenes, renderer):
    """crt_host_alloc: a page-locked output buffer gives the same frame as a pageable one"""
    sc = scenes.cornell_box()
    cam = sc["camera"]
    renderer.upload(sc["meshes"], sc["lights"], sc["materials"])
    renderer.set_camera(cam["position"], cam["matrix"])
    renderer.change_shading_mode(100)
    a = renderer.render_frame(200, 120, want=())["rgba8"].copy()
    b = renderer.render_frame(200, 120, want=(), pinned=True)["rgba8"].copy()
    c = renderer.render_frame(64, 32, want=(), pinned=True)["rgba8"].copy()  # size change reallocates
    assert np.array_equal(a, b) and c.shape == (32, 64, 4)
    assert np.array_equal(a, oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"]).render(cam["position"], cam["matrix"], 100, 200, 120)["rgba8"])


def test_external_stream_and_error_paths(pkg, scenes, renderer):
    import torch
    r2 = pkg.Renderer(0)
    with pytest.raises(pkg.CrtError) as e:
        r2.render_frame(8, 8)
    assert "rc=5" in str(e.value)  # CRT_ESTATE: render before upload
    sc = scenes.cornell_box()
    r2.upload(sc["meshes"], sc["lights"], sc["materials"])
    with pytest.raises(pkg.CrtError):
        r2.render_frame(0, 8)
    with pytest.raises(pkg.CrtError):
        r2.render_tiles_device(64, 64, 3, 2, 1234)
    # run on torch's current stream: ordering with torch ops is then guaranteed
    r2.set_camera(sc["camera"]["position"], sc["camera"]["matrix"])
    r2.set_stream(torch.cuda.current_stream().cuda_stream)
    buf = torch.zeros(64 * 64, dtype=torch.int32, device="cuda")
    r2.render_frame_device(64, 64, buf.data_ptr())
    a = buf.clone()
    torch.cuda.synchronize()
    r2.reset_stream()
    ref = r2.render_frame(64, 64)["rgba8"].view(np.uint32).reshape(-1)
    assert np.array_equal(a.cpu().numpy().view(np.uint32), ref)
    r2.close()
    with pytest.raises(pkg.CrtError):
        pkg.Renderer(99)


def test_scheduling_knobs_never_change_results(pkg, oracle, scenes, dragon, renderer):
    """Wave scheduling (inner_min), XCD grouping, cost-feedback launch order, priority boost and LDS stack size are
    speed knobs only: every setting must give the oracle's frame bit for bit (and the same fetch counters)."""
    sc = _with_normals(scenes, dragon)
    cam = sc["camera"]
    renderer.upload(sc["meshes"], sc["lights"], sc["materials"])
    renderer.set_camera(cam["position"], cam["matrix"])
    renderer.change_shading_mode(100)
    w, h = 640, 360
    ref = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"]).render(cam["position"], cam["matrix"], 100, w, h)
    defaults = {"inner_min": -6, "inner_min_any": -6, "xcd_group": 16, "adaptive_order": 2, "boost_units": 512, "split_units": -1, "split_rays": 4, "split_segments": 16,
                "xcd_affine_order": 0, "stack_entries": 0}
    try:
        for name, values in (("inner_min", (1, 7, 33, 65, -1, -3, -8)), ("inner_min_any", (1, 32, -2, -8)), ("xcd_group", (1, 4)), ("adaptive_order", (0, 1, 2)),
                             ("boost_units", (0, 100000)), ("split_units", (0, 7, 3000, 65536)), ("split_rays", (16, 8)), ("split_segments", (4, 8)), ("xcd_affine_order", (1,)), ("stack_entries", (1, 2, 5, 16, 32))):  # 1..5: the spill arena carries most of the stack
            for v in values:
                renderer.set_option(name, v)
                if name in ("split_rays", "split_segments"):
                    renderer.set_option("split_units", 40)  # (these two only act on split packets)
                renderer.set_counting(True)
                for frame in range(10):  # from frame 4 on (ring of 4 slots) the launch order comes from an earlier frame's costs
                    got = renderer.render_frame(w, h)
                    for k in ("hit_inst", "hit_prim", "hit_t", "rgba8"):
                        np.testing.assert_array_equal(got[k], ref[k], err_msg="%s=%d frame %d %s" % (name, v, frame, k))
                    assert np.array_equal(got["rgb"], ref["rgb"], equal_nan=True)
                    assert got["stats"]["rays_primary"] == ref["stats"]["rays_primary"] and got["stats"]["rays_shadow"] == ref["stats"]["rays_shadow"]
                    if name in ("split_units", "split_rays", "split_segments") and (v > 0 if name == "split_units" else True):
                        # a split packet's rays are traced as four segments that each descend from the root: same hits, more fetches
                        assert got["stats"]["nodes_visited"] >= ref["stats"]["nodes_visited"]
                    else:
                        assert got["stats"]["nodes_visited"] == ref["stats"]["nodes_visited"]
                        assert got["stats"]["tris_tested"] == ref["stats"]["tris_tested"]
            renderer.set_option(name, defaults[name])
            renderer.set_option("split_units", defaults["split_units"])
        with pytest.raises(pkg.CrtError):
            renderer.set_option("no_such_option", 1)
        with pytest.raises(pkg.CrtError):
            renderer.set_option("stack_entries", 33)
    finally:
        renderer.set_counting(False)
        for name, v in defaults.items():
            renderer.set_option(name, v)


def test_split_packets_render_the_same_frame(pkg, oracle, scenes, renderer):
    """Split packets (split_packet.hip.h; options split_units / split_rays / split_segments; on by default for the tile share of a
    multi-rank frame): the most expensive 8x8 packets are rendered by 4 / 8 / 16 wavefronts whose lanes share the segments of
    the block's rays (closest hit = lowest segment with a hit; a shadow ray is occluded if any segment is).  C3 scene, whole frame
    and an 8-rank tile share, primary rays only and with shadow rays (plain and Phong): hit ids, t and colours are the oracle's
    bit for bit."""
    import torch
    sc = scenes.heightfield(n_lights=1)
    cam = sc["camera"]
    renderer.upload(sc["meshes"], sc["lights"], sc["materials"])
    renderer.set_camera(cam["position"], cam["matrix"])
    w, h = 1920, 1080
    O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    try:
        for mode, phong in ((3, 0), (100, 0), (100, 300)):
            renderer.change_shading_mode(mode)
            renderer.set_option("phong_ks", phong)
            oracle.set_phong(phong, 32)
            ref = O.render(cam["position"], cam["matrix"], mode, w, h)
            for split, rays, segs in ((300, 4, 16), (4096, 16, 4), (150, 8, 8)):
                renderer.set_option("split_units", split)
                renderer.set_option("split_rays", rays)
                renderer.set_option("split_segments", segs)
                for frame in range(10):  # the launch order (and with it the split) comes from an earlier frame's costs
                    got = renderer.render_frame(w, h)
                for k in ("hit_inst", "hit_prim", "hit_t", "rgba8"):
                    np.testing.assert_array_equal(got[k], ref[k], err_msg="mode %d phong %d split %d x %d x %d %s" % (mode, phong, split, rays, segs, k))
                assert np.array_equal(got["rgb"], ref["rgb"], equal_nan=True)
            # an 8-rank share with the default setting (what the split is for: its launch lasts as long as its slowest packet)
            renderer.set_option("split_units", -1)
            renderer.set_option("split_rays", 4)
            renderer.set_option("split_segments", 16)
            n = 8
            slots = pkg.tile_slots(w, h, n)
            gathered = torch.zeros(n * slots * 256, dtype=torch.int32, device="cuda")
            frame_t = torch.zeros(w * h, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            for rep in range(10):
                for rank in range(n):
                    renderer.render_tiles_device(w, h, rank, n, gathered.data_ptr() + rank * slots * 1024)
            renderer.untile_device(w, h, n, gathered.data_ptr(), frame_t.data_ptr())
            renderer.synchronize()
            np.testing.assert_array_equal(frame_t.cpu().numpy().view(np.uint32).reshape(h, w), ref["rgba8"].view(np.uint32).reshape(h, w))
    finally:
        renderer.set_option("split_units", -1)
        renderer.set_option("split_rays", 4)
        renderer.set_option("split_segments", 16)
        renderer.set_option("phong_ks", 0)
        oracle.set_phong(0, 32)
        renderer.change_shading_mode(0)


def test_split_packets_stress(pkg, oracle, scenes, dragon, renderer):
    """A quarter of all packets split, every combination of rays per wavefront and pieces per ray, on scenes with several lights,
    smooth normals, tiny and huge triangles and rays that start inside boxes: hit ids, t and colours stay the oracle's."""
    cases = [(_with_normals(scenes, dragon), 640, 360, (3, 100)), (scenes.cornell_box(), 256, 256, (100,)),
             (scenes.displaced_sphere(n_lat=60, n_lon=80), 480, 270, (0, 100))]
    try:
        for sc, w, h, modes in cases:
            cam = sc["camera"]
            renderer.upload(sc["meshes"], sc["lights"], sc["materials"])
            renderer.set_camera(cam["position"], cam["matrix"])
            O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
            for mode in modes:
                renderer.change_shading_mode(mode)
                ref = O.render(cam["position"], cam["matrix"], mode, w, h)
                for rays in (4, 8, 16):
                    for segs in (4, 8, 16):
                        renderer.set_option("split_units", 65536)
                        renderer.set_option("split_rays", rays)
                        renderer.set_option("split_segments", segs)
                        for frame in range(9):
                            got = renderer.render_frame(w, h)
                        for k in ("hit_inst", "hit_prim", "hit_t", "rgba8"):
                            np.testing.assert_array_equal(got[k], ref[k], err_msg="mode %d, %d rays x %d segments: %s" % (mode, rays, segs, k))
                        assert np.array_equal(got["rgb"], ref["rgb"], equal_nan=True)
    finally:
        renderer.set_option("split_units", -1)
        renderer.set_option("split_rays", 4)
        renderer.set_option("split_segments", 16)
        renderer.change_shading_mode(0)


def _compare_path(pkg, oracle, renderer, sc, w, h, spp, bounces, seed, miss=(0.0, 0.0, 0.0)):
    cam = sc["camera"]
    renderer.upload(sc["meshes"], sc["lights"], sc["materials"], sc.get("textures"))
    renderer.set_camera(cam["position"], cam["matrix"])
    renderer.set_miss_color(miss)
    renderer.change_shading_mode(pkg.MODE_PATH)
    renderer.set_path_params(spp, bounces, seed)
    renderer.set_counting(True)
    O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"], textures=sc.get("textures") or ())
    oracle.set_path_params(spp, bounces, seed)
    try:
        got = renderer.render_frame(w, h)
        ref = O.render(cam["position"], cam["matrix"], oracle.MODE_PATH, w, h, miss_rgb=miss)
        renderer.set_counting(False)  # the plain kernel variant is the product
        plain = renderer.render_frame(w, h)
        for k in ("hit_inst", "hit_prim", "hit_t", "rgba8"):
            np.testing.assert_array_equal(plain[k], ref[k], err_msg=k + " (plain kernel)")
        assert np.array_equal(plain["rgb"], ref["rgb"], equal_nan=True), "path-traced rgb (plain kernel)"
    finally:
        oracle.set_path_params(4, 3, 1234)
        renderer.set_path_params(4, 3, 1234)
        renderer.set_counting(False)
        renderer.set_miss_color((0.0, 1.0, 1.0))
    for k in ("hit_inst", "hit_prim", "hit_t", "rgba8"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    err = np.abs(got["rgb"].astype(np.float64) - ref["rgb"].astype(np.float64))
    assert np.nanmax(err) <= RGB_TOL
    assert np.array_equal(got["rgb"], ref["rgb"], equal_nan=True), "path-traced rgb not bit-exact"
    st, rs = got["stats"], ref["stats"]
    assert (st["rays_primary"], st["rays_shadow"], st["nodes_visited"], st["tris_tested"]) == \
           (rs["rays_primary"], rs["rays_shadow"], rs["nodes_visited"], rs["tris_tested"])
    return got


def test_path_tracing_cornell_all_material_types(pkg, oracle, scenes, renderer):
    """mode 200 (BASELINE.json configs[4]: 4 spp, 3 bounces): diffuse + constant (ceiling light) + mirror + glass."""
    sc = scenes.cornell_box()
    _compare_path(pkg, oracle, renderer, sc, 256, 256, 4, 3, 1234)
    sc["materials"][1] = {"albedo": (0.9, 0.9, 0.9), "type": 2}                 # left wall mirror
    sc["materials"][2] = {"albedo": (1.0, 1.0, 1.0), "type": 3, "ior": 1.5}     # right wall glass
    got = _compare_path(pkg, oracle, renderer, sc, 200, 120, 3, 5, 42, miss=(0.2, 0.3, 0.4))
    assert np.isfinite(got["rgb"]).all()
    _compare_path(pkg, oracle, renderer, sc, 64, 64, 1, 0, 7)                    # no bounces at all
    # the stages as separate launches over global queues (option path_pipeline = 1): same frames, same counters -- one pass, then
    # passes of 64 work items each (64 x 64 x 3 paths), then more samples than a pass carries (16 + 16 + 5, cross-pass sums)
    try:
        renderer.set_option("path_pipeline", 1)
        _compare_path(pkg, oracle, renderer, sc, 200, 120, 3, 5, 42, miss=(0.2, 0.3, 0.4))
        renderer.set_option("path_pass_paths", 65536)
        _compare_path(pkg, oracle, renderer, sc, 200, 120, 3, 5, 42, miss=(0.2, 0.3, 0.4))
        _compare_path(pkg, oracle, renderer, sc, 70, 50, 37, 2, 99)
        _compare_path(pkg, oracle, renderer, sc, 64, 64, 1, 0, 7)
        # nothing to trace at all, and a scene whose every queue is shorter than one wavefront
        nothing = {"meshes": [], "lights": [], "materials": [], "camera": {"position": np.float32([0, 0, 0]), "matrix": scenes.IDENTITY}}
        _compare_path(pkg, oracle, renderer, nothing, 64, 48, 2, 2, 5, miss=(0.1, 0.2, 0.3))
        _compare_path(pkg, oracle, renderer, scenes.single_triangle(), 33, 17, 3, 2, 11)
    finally:
        renderer.set_option("path_pass_paths", 1 << 24)
        renderer.set_option("path_pipeline", 0)
    # more samples than one pass of the pipeline carries (16 per 8x8 tile, 4 per 16x16 tile): the sample average must still run
    # in sample order across the passes (running sums kept in the workgroup's scratch)
    try:
        _compare_path(pkg, oracle, renderer, sc, 70, 50, 37, 2, 99)              # passes of 16 + 16 + 5
        renderer.set_option("path_tile", 16)
        renderer.set_option("path_ranges", 1)                                    # one shared work counter instead of one per XCD
        _compare_path(pkg, oracle, renderer, sc, 70, 50, 9, 2, 5)                # passes of 4 + 4 + 1, macro-tile workgroups
    finally:
        renderer.set_option("path_tile", 0)
        renderer.set_option("path_ranges", 8)


def test_path_tracing_dragon_smooth_normals_and_mirror_ground(pkg, oracle, scenes, dragon, renderer):
    """The shipped scene's own materials: reflective ground (type 2), diffuse smooth-shaded dragon, 4 lights."""
    _compare_path(pkg, oracle, renderer, _with_normals(scenes, dragon), 640, 360, 4, 3, 1234)


def test_path_tracing_large_mesh(pkg, oracle, scenes, renderer):
    """Path tracing on a 250k-triangle height field at 4K/16 (stand-in for configs[4]'s 5M-triangle 4K frame, which the
    CPU oracle cannot finish in test time): same kernel, same code paths, incoherent bounce rays."""
    sc = scenes.heightfield(n=354, n_lights=2)
    _compare_path(pkg, oracle, renderer, sc, 960, 540, 4, 3, 1234)


def test_c5_full_size_properties(pkg, scenes, renderer):
    """BASELINE.json configs[4] at full size: 4 999 124-triangle mesh, 3840x2160, 4 spp, 3 bounces.  The CPU oracle
    needs minutes for this frame, so the full-size run is checked through size-independent properties: determinism
    (idempotence under a different launch order), tile partition == single launch, structural sanity; the arithmetic
    itself is pinned bit-exactly against the oracle at smaller sizes above."""
    import torch
    sc = scenes.heightfield(n=1581, n_lights=2)
    assert sum(len(m["triangles"]) for m in sc["meshes"]) == 4999124
    renderer.upload(sc["meshes"], sc["lights"], sc["materials"])
    info = renderer.bvh_info()
    assert info["n_tris"] == 4999124 and info["max_depth"] <= 32
    renderer.set_camera(sc["camera"]["position"], sc["camera"]["matrix"])
    renderer.change_shading_mode(pkg.MODE_PATH)
    renderer.set_path_params(4, 3, 1234)
    w, h = 3840, 2160
    try:
        a = torch.zeros(h * w, dtype=torch.int32, device="cuda")
        b = torch.zeros(h * w, dtype=torch.int32, device="cuda")
        rgb = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()  # own stream vs torch's fill kernels
        renderer.set_counting(True)
        st = renderer.render_frame_device(w, h, a.data_ptr(), d_rgb=rgb.data_ptr(), stats=True)
        renderer.set_counting(False)
        assert st["rays_primary"] >= 4 * w * h and st["rays_shadow"] > 0
        renderer.render_frame_device(w, h, b.data_ptr())   # second frame runs in the cost-sorted order of the first
        renderer.synchronize()
        assert torch.equal(a, b)
        assert bool(torch.isfinite(rgb).all()) and float(rgb.min()) >= 0.0
        n = 8
        slots = pkg.tile_slots(w, h, n)
        gathered = torch.zeros(n * slots * 256, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for rank in range(n):
            renderer.render_tiles_device(w, h, rank, n, gathered.data_ptr() + rank * slots * 1024)
        renderer.untile_device(w, h, n, gathered.data_ptr(), b.data_ptr())
        renderer.synchronize()
        assert torch.equal(a, b)
        alpha = (a.cpu().numpy().view(np.uint32) >> 24)
        assert np.all(alpha == 255)
    finally:
        renderer.set_counting(False)
        renderer.change_shading_mode(0)


def test_headless_cpp_driver_matches_binding(pkg, oracle, scenes, dragon, tmp_path, golden_dir):
    """crt_render (csrc/crt_render_main.cpp): the C++ host path -- crt::Scene loader, crt::Renderer with DXRTRenderer's
    methods, scripted camera through the same Camera calls the reference's input handlers make -- produces the frames
    the oracle predicts for the same camera path."""
    import subprocess
    exe = os.path.join(os.path.dirname(pkg.LIB_PATH), "crt_render")
    assert os.path.exists(exe), "crt_render not built"
    prefix = str(tmp_path / "frame")
    out = subprocess.run([exe, os.path.join(golden_dir, "dragon.crtscene"), "--mode", "100", "--size", "480x270", "--frames", "3",
                          "--orbit", "15", "--forward", "1.5", "--out", prefix, "--count"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Mray/s" in out.stdout and "average kernel" in out.stdout
    sc = _with_normals(scenes, dragon)
    O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    s = pkg.Scene(os.path.join(golden_dir, "dragon.crtscene"))
    for f in range(3):
        if f > 0:
            s.rotate(15.0, 0.0)
            s.move_forward(-1.5)
        pos, rot = s.camera()
        ref = O.render(pos, rot, 100, 480, 270)["rgba8"][..., :3]
        raw = open("%s_%d.ppm" % (prefix, f), "rb").read()
        header_end = raw.index(b"255\n") + 4
        assert raw[:header_end] == b"P6\n480 270\n255\n"
        img = np.frombuffer(raw[header_end:], dtype=np.uint8).reshape(270, 480, 3)
        np.testing.assert_array_equal(img, ref, err_msg="frame %d" % f)
    # --png: the same frame as a PNG (8-bit RGB, stored deflate blocks -- several of them at this size), chunk CRCs and the
    # Adler-32 checked by zlib, and readable by the product's own PNG decoder
    import struct
    import zlib
    out = subprocess.run([exe, os.path.join(golden_dir, "dragon.crtscene"), "--mode", "100", "--size", "333x217", "--frames", "1",
                          "--out", prefix, "--png"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    raw = open(prefix + "_0.png", "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat, kinds = 8, b"", []
    while at < len(raw):
        n, kind = struct.unpack(">I4s", raw[at:at + 8])
        body = raw[at + 8:at + 8 + n]
        assert struct.unpack(">I", raw[at + 8 + n:at + 12 + n])[0] == zlib.crc32(kind + body), kind
        kinds.append(kind)
        if kind == b"IHDR":
            assert struct.unpack(">IIBBBBB", body) == (333, 217, 8, 2, 0, 0, 0)
        if kind == b"IDAT":
            idat += body
        at += 12 + n
    assert kinds[0] == b"IHDR" and kinds[-1] == b"IEND"
    lines = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(217, 1 + 3 * 333)
    assert not lines[:, 0].any()
    s0 = pkg.Scene(os.path.join(golden_dir, "dragon.crtscene"))
    pos, rot = s0.camera()
    ref = O.render(pos, rot, 100, 333, 217)["rgba8"][..., :3]
    np.testing.assert_array_equal(lines[:, 1:].reshape(217, 333, 3), ref)
    t = pkg.Scene()
    t.add_texture("frame", "bitmap", file_path=prefix + "_0.png")
    np.testing.assert_array_equal(t.texture_color(0, 0.0, 1.0), ref[0, 0].astype(np.float32) / np.float32(255.0))
    # failure is an error message and a non-zero exit code, never an assert/abort
    bad = subprocess.run([exe, str(tmp_path / "nope.crtscene")], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "cannot open" in bad.stderr


def _read_ppm(path, w, h):
    raw = open(path, "rb").read()
    header_end = raw.index(b"255\n") + 4
    assert raw[:header_end] == b"P6\n%d %d\n255\n" % (w, h)
    return np.frombuffer(raw[header_end:], dtype=np.uint8).reshape(h, w, 3)


def test_headless_driver_full_camera_controls_and_mode_switch(pkg, oracle, scenes, dragon, tmp_path, golden_dir):
    """The reference's whole control surface, scripted: A/D -> moveRight (R/DXRTApp.cpp:91-107), wheel -> zoom
    (R/DXRTViewportWidget.cpp:74-78), mouse -> rotate with pitch, the shading-mode combo box (R/DXRTMainWindow.cpp:114-121),
    per-frame flags plus a camera-path file, Phong and path-tracing parameters: every frame equals the oracle's for the camera
    the same Camera calls produce."""
    import subprocess
    exe = os.path.join(os.path.dirname(pkg.LIB_PATH), "crt_render")
    scene_file = os.path.join(golden_dir, "dragon.crtscene")
    path_file = tmp_path / "cam.txt"
    path_file.write_text("# frame 0: untouched\n\nrotate -4 2; right 0.75\nzoom 0.5; mode 5\npan 3; tilt -2; roll 1.5; forward 0.25\n")
    prefix = str(tmp_path / "f")
    w, h, frames = 320, 180, 5
    out = subprocess.run([exe, scene_file, "--mode", "100", "--size", "%dx%d" % (w, h), "--frames", str(frames), "--orbit", "3", "--pitch", "-1",
                          "--right", "0.5", "--zoom", "0.125", "--mode-at", "1:3", "--mode-at", "4:100", "--phong", "400:12",
                          "--path", str(path_file), "--out", prefix], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    sc = _with_normals(scenes, dragon)
    O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    s = pkg.Scene(scene_file)
    script = {2: lambda: (s.rotate(-4.0, 2.0), s.move_right(0.75)), 3: lambda: s.zoom(0.5),
              4: lambda: (s.pan(3.0), s.tilt(-2.0), s.roll(1.5), s.move_forward(-0.25))}
    modes = {0: 100, 1: 3, 2: 3, 3: 5, 4: 100}  # --mode-at 1:3, "mode 5" in the path file at frame 3, --mode-at 4:100
    try:
        oracle.set_phong(400, 12)
        for f in range(frames):
            if f > 0:
                s.rotate(3.0, -1.0)
                s.move_right(0.5)
                s.zoom(0.125)
            if f in script:
                script[f]()
            pos, rot = s.camera()
            ref = O.render(pos, rot, modes[f], w, h)["rgba8"][..., :3]
            np.testing.assert_array_equal(_read_ppm("%s_%d.ppm" % (prefix, f), w, h), ref, err_msg="frame %d" % f)
    finally:
        oracle.set_phong(0, 32)
    # path tracing parameters from the command line
    out = subprocess.run([exe, scene_file, "--mode", "200", "--size", "160x90", "--spp", "2", "--bounces", "1", "--seed", "77",
                          "--out", prefix + "p"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    oracle.set_path_params(2, 1, 77)
    try:
        s0 = pkg.Scene(scene_file)
        pos, rot = s0.camera()
        ref = O.render(pos, rot, 200, 160, 90)["rgba8"][..., :3]
    finally:
        oracle.set_path_params(4, 3, 1234)
    np.testing.assert_array_equal(_read_ppm(prefix + "p_0.ppm", 160, 90), ref)
    bad = subprocess.run([exe, scene_file, "--path", str(tmp_path / "none.txt")], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "camera path" in bad.stderr


def test_native_rccl_gather_behind_the_c_abi(pkg, oracle, scenes, dragon, renderer, tmp_path, golden_dir):
    """crt_comm_init + crt_render_frame_distributed (csrc/crt_api.cpp): tiles -> ncclAllGather on the context's stream ->
    untile, no torch involved.  With the one rank this box has: the distributed frame equals crt_render_frame_device's and the
    oracle's; crt_render --ranks 1 (the C++-only launcher: parent forks the rank processes before touching the GPU) writes
    the same images as the single-process run; and so do three rank processes sharing this GPU through the host-memory exchange."""
    import subprocess
    import torch
    sc = _with_normals(scenes, dragon)
    cam = sc["camera"]
    w, h = 1920, 1080
    r2 = pkg.Renderer(0)
    try:
        r2.upload(sc["meshes"], sc["lights"], sc["materials"])
        r2.set_camera(cam["position"], cam["matrix"])
        r2.change_shading_mode(100)
        with pytest.raises(pkg.CrtError):
            r2.render_frame_distributed(w, h, host=True)  # no communicator yet: CRT_ESTATE
        uid = r2.comm_init(0, 1)
        assert len(uid) == 128
        with pytest.raises(pkg.CrtError):
            r2.comm_init(0, 1, uid)  # already has one
        got = r2.render_frame_distributed(w, h, host=True, stats=True)
        assert got["stats"]["rays_primary"] == w * h and got["stats"]["kernel_ms"] > 0
        single = r2.render_frame(w, h, want=())["rgba8"]
        ref = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"]).render(cam["position"], cam["matrix"], 100, w, h, want=("rgba8",))["rgba8"]
        np.testing.assert_array_equal(got["rgba8"], single)
        np.testing.assert_array_equal(got["rgba8"], ref)
        # device-pointer form, several frames in flight on the context's stream, ragged size
        dev = torch.zeros(333 * 77, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for _ in range(6):
            r2.render_frame_distributed(333, 77, d_rgba8=dev.data_ptr())
        r2.synchronize()
        np.testing.assert_array_equal(dev.cpu().numpy().view(np.uint8).reshape(77, 333, 4), r2.render_frame(333, 77, want=())["rgba8"])
        r2.comm_destroy()
        r2.comm_destroy()  # idempotent
        # the host-memory transport behind the same calls (one rank: its own slice goes in and comes out)
        r2.comm_init_host(0, 1, "/crt_test_%d" % os.getpid())
        with pytest.raises(pkg.CrtError):
            r2.comm_init_host(0, 1, "/crt_test_again_%d" % os.getpid())  # already has one
        np.testing.assert_array_equal(r2.render_frame_distributed(w, h, host=True)["rgba8"], single)
        r2.comm_destroy()
        assert not [n for n in os.listdir("/dev/shm") if n.startswith("crt_test_")]
        with pytest.raises(pkg.CrtError):
            r2.comm_init_host(0, 1, "no_leading_slash")
    finally:
        r2.close()
    exe = os.path.join(os.path.dirname(pkg.LIB_PATH), "crt_render")
    scene_file = os.path.join(golden_dir, "dragon.crtscene")
    common = ["--mode", "100", "--size", "480x270", "--frames", "2", "--orbit", "10"]
    a = subprocess.run([exe, scene_file] + common + ["--out", str(tmp_path / "one")], capture_output=True, text=True, timeout=300)
    b = subprocess.run([exe, scene_file] + common + ["--out", str(tmp_path / "ranks"), "--ranks", "1"], capture_output=True, text=True, timeout=300)
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stdout + b.stderr
    assert "rank 0's tile share of 1 ranks" in b.stdout
    for f in range(2):
        np.testing.assert_array_equal(_read_ppm(str(tmp_path / ("one_%d.ppm" % f)), 480, 270), _read_ppm(str(tmp_path / ("ranks_%d.ppm" % f)), 480, 270))
    # THREE native rank processes on this one GPU: the launcher, the per-rank tile shares (split packets at their multi-rank
    # default), the frame assembly and rank 0's output run as they will on three GPUs, with shared host memory carrying the tiles
    # (crt_comm_init_host: RCCL refuses several ranks on one GPU) -- ragged size, a camera that moves, a mode switch
    c = subprocess.run([exe, scene_file, "--mode", "100", "--size", "333x217", "--frames", "3", "--orbit", "10", "--mode-at", "2:3",
                        "--out", str(tmp_path / "three"), "--ranks", "3", "--host-exchange", "--same-device"], capture_output=True, text=True, timeout=300)
    d = subprocess.run([exe, scene_file, "--mode", "100", "--size", "333x217", "--frames", "3", "--orbit", "10", "--mode-at", "2:3",
                        "--out", str(tmp_path / "solo")], capture_output=True, text=True, timeout=300)
    assert c.returncode == 0 and d.returncode == 0, c.stdout + c.stderr + d.stderr
    assert "rank 0's tile share of 3 ranks" in c.stdout
    for f in range(3):
        np.testing.assert_array_equal(_read_ppm(str(tmp_path / ("three_%d.ppm" % f)), 333, 217), _read_ppm(str(tmp_path / ("solo_%d.ppm" % f)), 333, 217))
    assert not [n for n in os.listdir("/dev/shm") if n.startswith("crt_render_")], "the exchange's shared memory was left behind"


def test_reference_scene_layer_bound_to_the_c_abi(pkg, oracle, scenes, dragon, tmp_path, golden_dir):
    """INTEGRATION.md variant B, compiled and run: integration/DXRTRenderer_hip.cpp -- a DXRTRenderer with the reference's
    public interface (R/DXRTRenderer.h:74-94) -- over the REFERENCE's own CRT* classes (compiled in place in the build
    container by oracle/Makefile into oracle/_ref/ref_shim_render, which travels to this box as a binary; the reference's
    sources do not), bound to libcrt_hip.so.  It loads the Dragon scene with the reference's parser, moves the camera with
    the reference's CRTCamera and renders modes 0 / 3 / 100; each frame must equal the oracle's for the camera it printed."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "_ref", "ref_shim_render")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_shim_render is built only where the reference tree is mounted")
    prefix = str(tmp_path / "shim")
    w, h = 640, 360
    out = subprocess.run([exe, os.path.join(golden_dir, "dragon.crtscene"), prefix, "%dx%d" % (w, h)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l.split() for l in out.stdout.splitlines() if l.startswith("frame ")]
    assert len(lines) == 3
    sc = _with_normals(scenes, dragon)
    O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    # the build's own scene layer reaches the same camera states with the same calls (bit for bit here; goldens allow 2e-7)
    s = pkg.Scene(os.path.join(golden_dir, "dragon.crtscene"))
    for f, l in enumerate(lines):
        mode = int(l[3])
        cam = np.float32([float(x) for x in l[5:17]])
        ref = O.render(cam[:3], cam[3:], mode, w, h)["rgba8"][..., :3]
        np.testing.assert_array_equal(_read_ppm("%s_%d.ppm" % (prefix, f), w, h), ref, err_msg="frame %d" % f)
        if f == 1:
            s.rotate(20.0, -5.0)
            s.move_forward(-2.0)
        if f == 2:
            s.move_right(1.5)
            s.zoom(0.25)
        pos, rot = s.camera()
        np.testing.assert_allclose(np.concatenate([pos, rot]), cam, rtol=0, atol=2e-6)


def test_context_lifecycle_and_reuse(pkg, oracle, scenes, dragon):
    """re-upload, frame-size changes, two contexts at once, NaN vertices, many small frames: no state leaks between
    frames/contexts; results stay the oracle's."""
    sc1, sc2 = scenes.cornell_box(), _with_normals(scenes, dragon)
    O1 = oracle.OracleScene(sc1["meshes"], sc1["lights"], sc1["materials"])
    O2 = oracle.OracleScene(sc2["meshes"], sc2["lights"], sc2["materials"])
    a, b = pkg.Renderer(0), pkg.Renderer(0)
    try:
        a.upload(sc1["meshes"], sc1["lights"], sc1["materials"])
        b.upload(sc2["meshes"], sc2["lights"], sc2["materials"])
        a.set_camera(sc1["camera"]["position"], sc1["camera"]["matrix"])
        b.set_camera(sc2["camera"]["position"], sc2["camera"]["matrix"])
        a.change_shading_mode(100)
        b.change_shading_mode(2)
        for (w, h) in ((64, 64), (300, 200), (64, 64), (1920, 1080), (33, 9)):
            ga, gb = a.render_frame(w, h), b.render_frame(w, h)
            ra = O1.render(sc1["camera"]["position"], sc1["camera"]["matrix"], 100, w, h)
            rb = O2.render(sc2["camera"]["position"], sc2["camera"]["matrix"], 2, w, h)
            assert np.array_equal(ga["rgba8"], ra["rgba8"]) and np.array_equal(gb["rgba8"], rb["rgba8"]), (w, h)
            assert np.array_equal(ga["hit_prim"], ra["hit_prim"]) and np.array_equal(gb["hit_prim"], rb["hit_prim"])
        # swap the scenes between the contexts
        a.upload(sc2["meshes"], sc2["lights"], sc2["materials"])
        a.set_camera(sc2["camera"]["position"], sc2["camera"]["matrix"])
        a.change_shading_mode(2)
        assert np.array_equal(a.render_frame(300, 200)["rgba8"], O2.render(sc2["camera"]["position"], sc2["camera"]["matrix"], 2, 300, 200)["rgba8"])
        # a mesh with a NaN vertex: the triangle can never be hit (comparisons with NaN fail), nothing crashes
        v = np.float32([(-1, -1, -3), (1, -1, -3), (0, 1, -3), (np.nan, 0, -2), (1, 0, -2), (0, 1, -2)])
        weird = {"meshes": [{"vertices": v, "triangles": [(0, 1, 2), (3, 4, 5)]}], "lights": [], "materials": []}
        a.upload(weird["meshes"])
        a.set_camera((0, 0, 0), scenes.IDENTITY)
        a.change_shading_mode(3)
        out = a.render_frame(64, 64)
        assert set(np.unique(out["hit_prim"]).tolist()) <= {0, pkg.MISS} and (out["hit_prim"] == 0).any()
    finally:
        a.close()
        b.close()


def test_gpu_bvh_build_matches_its_spec_and_renders_identically(pkg, oracle, scenes, dragon):
    """the acceleration structure built by HIP kernels (LBVH: Morton codes, radix sort, Karras
    hierarchy, bottom-up fit) is byte-identical to the oracle's CPU restatement of the same algorithm, and frames
    rendered over it are bit-identical to the oracle's frames over its own LBVH tree -- and equal in hit ids, t and
    RGBA8 to the frames over the SAH tree (a different tree finds the same closest hits)."""
    f = np.float32
    tri = f([(0, 0, -3), (1, 0, -3), (0, 1, -3)])
    small = [{"meshes": [{"vertices": np.concatenate([tri + f([1.5 * i, 0, 0]) for i in range(n)]),
                          "triangles": np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)}], "lights": [], "materials": [],
              "camera": {"position": f([2, 0.3, 0]), "matrix": scenes.IDENTITY}} for n in (1, 2, 4, 5, 9)]
    # the radix sort's corners: 5 000 coincident triangles (every Morton code equal: the order is the input order, i.e. the sort
    # must be stable) and 6 151 triangles in two clumps (three 2 048-key tiles, the last one ragged; two digits only)
    rng = np.random.default_rng(5)
    same = {"meshes": [{"vertices": np.tile(tri, (5000, 1)), "triangles": np.arange(15000, dtype=np.uint32).reshape(-1, 3)}], "lights": [],
            "materials": [], "camera": {"position": f([0.3, 0.3, 0]), "matrix": scenes.IDENTITY}}
    clump = np.concatenate([tri + (f([0, 0, 0]) if i % 3 else f([40, 0, 0])) + f(rng.uniform(0, 1e-3, 3)) for i in range(6151)])
    clumps = {"meshes": [{"vertices": clump.astype(np.float32), "triangles": np.arange(3 * 6151, dtype=np.uint32).reshape(-1, 3)}], "lights": [],
              "materials": [], "camera": {"position": f([0.3, 0.3, 0]), "matrix": scenes.IDENTITY}}
    cases = [(s, 64, 64, (3,)) for s in small] + [(same, 48, 48, (3,)), (clumps, 48, 48, (3,))] + [
        (scenes.cornell_box(), 256, 256, (3, 100, 200)),
        (_with_normals(scenes, dragon), 640, 360, (0, 100, 200)),
        (scenes.displaced_sphere(), 640, 360, (100,)),
        (scenes.heightfield(n_lights=1), 1920, 1080, (100,)),
    ]
    r = pkg.Renderer(0)
    try:
        r.set_option("gpu_build", 1)
        for sc, w, h, modes in cases:
            cam = sc["camera"]
            r.upload(sc["meshes"], sc["lights"], sc["materials"])
            st = r.build_stats()
            n_tris = sum(len(m["triangles"]) for m in sc["meshes"])
            assert st["upload_ms"] > 0 and (st["device_build_ms"] > 0 or n_tris <= 4)
            O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"], build_mode=1)
            nodes, tris, shade = r.bvh_export()
            assert nodes.tobytes() == O.nodes().tobytes(), "LBVH binary nodes differ (%d tris)" % n_tris
            assert tris.tobytes() == O.tris().tobytes() and shade.tobytes() == O.shade().tobytes()
            nodes4, depth4 = r.bvh_export4()
            assert nodes4.tobytes() == O.nodes4().tobytes() and depth4 == O.depth4
            assert r.bvh_info()["max_depth"] == O.max_depth
            S = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])  # SAH tree
            r.set_camera(cam["position"], cam["matrix"])
            for mode in modes:
                r.change_shading_mode(mode)
                r.set_counting(True)
                got = r.render_frame(w, h)
                r.set_counting(False)
                ref = O.render(cam["position"], cam["matrix"], mode, w, h)
                for k in ("hit_inst", "hit_prim", "hit_t", "rgba8"):
                    np.testing.assert_array_equal(got[k], ref[k], err_msg="mode %d %s" % (mode, k))
                assert np.array_equal(got["rgb"], ref["rgb"], equal_nan=True)
                assert (got["stats"]["nodes_visited"], got["stats"]["tris_tested"]) == (ref["stats"]["nodes_visited"], ref["stats"]["tris_tested"])
                if mode != 200:  # path tracing's random walk is tree independent too, but costs a second oracle render
                    sah = S.render(cam["position"], cam["matrix"], mode, w, h)
                    for k in ("hit_inst", "hit_prim", "hit_t", "rgba8"):
                        np.testing.assert_array_equal(got[k], sah[k], err_msg="vs SAH tree, mode %d %s" % (mode, k))
        # uvs and textures through the device-side gather (the records never exist on the host for this builder)
        sc = scenes.textured_cornell()
        cam = sc["camera"]
        r.upload(sc["meshes"], sc["lights"], sc["materials"], sc["textures"])
        O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"], build_mode=1, textures=sc["textures"])
        uv = r.bvh_export_uv()
        assert uv is not None and uv.tobytes() == O.uvs().tobytes()
        assert r.bvh_export4q().tobytes() == O.nodes4q().tobytes()
        r.set_camera(cam["position"], cam["matrix"])
        r.change_shading_mode(100)
        got = r.render_frame(320, 240)
        ref = O.render(cam["position"], cam["matrix"], 100, 320, 240)
        np.testing.assert_array_equal(got["rgba8"], ref["rgba8"])
        assert np.array_equal(got["rgb"], ref["rgb"], equal_nan=True)
        # switching back to the host SAH builder works on the same context
        r.set_option("gpu_build", 0)
        sc = scenes.cornell_box()
        r.upload(sc["meshes"], sc["lights"], sc["materials"])
        assert r.bvh_export()[0].tobytes() == oracle.OracleScene(sc["meshes"]).nodes().tobytes()
        assert r.build_stats()["device_build_ms"] == 0.0
    finally:
        r.close()


def test_textured_scene(pkg, oracle, scenes, renderer):
    """the reference's four texture kinds sampled on the device (diffuse hits of modes 100
    and 200), uvs carried through the BVH reorder; identical to the oracle, whose texture functions are pinned by
    tests/golden/texture_known_answers.json."""
    sc = scenes.textured_cornell()
    cam = sc["camera"]
    w, h = 320, 240
    renderer.upload(sc["meshes"], sc["lights"], sc["materials"], sc["textures"])
    renderer.set_camera(cam["position"], cam["matrix"])
    O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"], textures=sc["textures"])
    uv = renderer.bvh_export_uv()
    assert uv is not None and uv.tobytes() == O.uvs().tobytes()
    renderer.change_shading_mode(100)
    got = renderer.render_frame(w, h)
    ref = O.render(cam["position"], cam["matrix"], 100, w, h)
    np.testing.assert_array_equal(got["hit_prim"], ref["hit_prim"])
    np.testing.assert_array_equal(got["rgba8"], ref["rgba8"])
    assert np.array_equal(got["rgb"], ref["rgb"], equal_nan=True)
    # the textures really are in the picture: the same scene without them renders differently, and many distinct colours appear
    plain = renderer.render_frame(w, h)["rgba8"].copy()
    renderer.set_textures([])
    assert not np.array_equal(renderer.render_frame(w, h)["rgba8"], plain)
    renderer.set_textures(sc["textures"])
    assert len(np.unique(plain.reshape(-1, 4), axis=0)) > 500
    _compare_path(pkg, oracle, renderer, sc, w, h, spp=4, bounces=4, seed=11)
    # a texture index past the table, and a material that names a texture on a scene without uvs, fall back to the plain albedo
    sc2 = scenes.cornell_box()
    sc2["materials"][0]["texture"] = 0
    renderer.upload(sc2["meshes"], sc2["lights"], sc2["materials"], [sc["textures"][0]])
    assert renderer.bvh_export_uv() is None
    renderer.change_shading_mode(100)
    O2 = oracle.OracleScene(sc2["meshes"], sc2["lights"], sc2["materials"], textures=[sc["textures"][0]])
    got = renderer.render_frame(64, 64)
    ref = O2.render(cam["position"], cam["matrix"], 100, 64, 64)
    np.testing.assert_array_equal(got["rgba8"], ref["rgba8"])


def test_scene_file_with_png_and_tga_textures_on_the_device(pkg, oracle, scenes, renderer, tmp_path, golden_dir):
    """A .crtscene whose bitmap textures are a PNG (RGBA, fixed-Huffman deflate), a run-length coded TGA and a progressive 4:2:0
    JPEG -- files the reference's stbi_load accepts (R/CRTTextureBitmap.cpp:10) and this repo decodes itself -- through
    crt_scene_load -> crt_upload_scene_from -> the kernels: the frame equals the oracle's, whose bitmap is rebuilt texel by texel
    from the host class that tests/golden/bitmap_known_answers.json pins to the reference."""
    import json
    quad = lambda x0, x1, z: [[x0, -1, z], [x1, -1, z], [x1, 1, z], [x0, 1, z]]
    objs = []
    for i, (x0, x1) in enumerate(((-3.3, -1.2), (-1.05, 1.05), (1.2, 3.3))):
        objs.append({"material_index": i, "vertices": [c for v in quad(x0, x1, -3.0) for c in v], "triangles": [0, 1, 2, 0, 2, 3],
                     "uvs": [0, 0, 0, 1, 0, 0, 1, 1, 0, 0, 1, 0]})
    files = (("png", "tex_rgba8_fixed.png", 9, 6), ("tga", "tex_32_rle_topleft.tga", 9, 6), ("jpg", "tex_prog420.jpg", 21, 19))
    scene = {"settings": {"background_color": [0, 0, 0], "image_settings": {"width": 320, "height": 200}},
             "camera": {"matrix": [1, 0, 0, 0, 1, 0, 0, 0, 1], "position": [0, 0, 0]},
             "lights": [{"intensity": 400, "position": [0, 1, 2]}],
             "textures": [{"name": n, "type": "bitmap", "file_path": os.path.join(golden_dir, f)} for n, f, _, _ in files],
             "materials": [{"type": "diffuse", "albedo": n, "smooth_shading": False} for n, _, _, _ in files],
             "objects": objs}
    path = tmp_path / "textured.crtscene"
    path.write_text(json.dumps(scene))
    s = pkg.Scene(str(path))
    assert s.texture_count == 3 and [m["texture"] for m in s.materials()] == [0, 1, 2]
    renderer.upload_scene(s)
    renderer.set_camera_from(s)
    pos, rot = s.camera()
    # the oracle's copy of the bitmaps: every texel through the host class (v flipped)
    tex = []
    for i, (_, _, tw, th) in enumerate(files):
        px = np.zeros((th, tw, 3), np.uint8)
        for row in range(th):
            for col in range(tw):
                # (aimed a quarter texel inside: the class truncates u * (w - 1) and (1 - v) * (h - 1), and clamps u, v to [0, 1])
                c = s.texture_color(i, (np.float32(col) + np.float32(0.25)) / np.float32(tw - 1), np.float32(1) - (np.float32(row) + np.float32(0.25)) / np.float32(th - 1))
                px[row, col] = np.round(c * 255.0).astype(np.uint8)
        tex.append({"type": "bitmap", "pixels": px})
    meshes = [dict(vertices=m["vertices"], triangles=m["triangles"], normals=m["normals"], uvs=m["uvs"], material_index=m["material_index"]) for m in s.meshes()]
    O = oracle.OracleScene(meshes, s.lights(), s.materials(), textures=tex)
    renderer.change_shading_mode(100)
    got = renderer.render_frame(320, 200)
    ref = O.render(pos, rot, 100, 320, 200)
    np.testing.assert_array_equal(got["hit_prim"], ref["hit_prim"])
    np.testing.assert_array_equal(got["rgba8"], ref["rgba8"])
    assert np.array_equal(got["rgb"], ref["rgb"], equal_nan=True)
    assert len(np.unique(got["rgba8"].reshape(-1, 4), axis=0)) > 120  # all three bitmaps are in the picture


def test_frames_in_flight_and_launch_order_feedback(pkg, oracle, scenes, dragon, renderer):
    """Many frames issued back to back on alternating streams (the ring of per-frame scratch slots, the cost-bucketed
    launch queues taken from and refilled by every frame, the in-kernel recycling of a consumed queue): every frame,
    rendered into its own sentinel-filled buffer, must be the oracle's frame -- a work unit rendered twice or never
    would leave sentinel pixels.  Resolution changes in between invalidate the queues (stale, never-consumed ones are
    cleared on the stream)."""
    import torch
    sc = _with_normals(scenes, dragon)
    cam = sc["camera"]
    renderer.upload(sc["meshes"], sc["lights"], sc["materials"])
    renderer.set_camera(cam["position"], cam["matrix"])
    renderer.change_shading_mode(100)
    O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    refs = {}
    streams = [torch.cuda.Stream() for _ in range(5)]  # more streams than ring slots
    try:
      for policy, n_streams in ((1, 5), (2, 5), (2, 1), (1, 1), (0, 3)):  # feedback forced on / auto / off; 1 stream = auto turns it on
        renderer.set_option("adaptive_order", policy)
        n = 0
        for (w, h), frames in (((640, 360), 13), ((333, 217), 9), ((640, 360), 11), ((64, 64), 10)):
            if (w, h) not in refs:
                refs[(w, h)] = O.render(cam["position"], cam["matrix"], 100, w, h)["rgba8"].reshape(-1, 4).copy().view(np.uint32).ravel()
            bufs = [torch.full((w * h,), 0x7E57AB1E, dtype=torch.int32, device="cuda") for _ in range(frames)]
            torch.cuda.synchronize()
            for b in bufs:
                st = streams[n % n_streams]
                n += 1
                renderer.set_stream(st.cuda_stream)
                renderer.render_frame_device(w, h, b.data_ptr())
            torch.cuda.synchronize()
            for i, b in enumerate(bufs):
                got = b.cpu().numpy().view(np.uint32)
                assert np.array_equal(got, refs[(w, h)]), "policy %d, %d streams, %dx%d frame %d: %d pixels differ" % (
                    policy, n_streams, w, h, i, int((got != refs[(w, h)]).sum()))
    finally:
        torch.cuda.synchronize()
        renderer.set_option("adaptive_order", 2)
        renderer.reset_stream()


def test_path_frames_in_flight_share_scratch_safely(pkg, oracle, scenes, renderer):
    """Mode 200 frames issued back to back: on ONE stream they reuse one scratch arena and one work counter in stream order;
    on more streams than there are arenas an arena is taken over from the least recently used stream behind an event.
    Every frame, rendered into its own sentinel-filled buffer, must be the oracle's frame.  The arena is sized by the
    workgroups RESIDENT on the device, not by the frame: a 4K frame at 16 spp must not grow device memory by more than
    the documented bound (608 MB per stream at 16 samples -- the frame-sized scratch it replaced was 15 GB)."""
    import torch
    sc = scenes.cornell_box()
    sc["materials"][1] = {"albedo": (0.9, 0.9, 0.9), "type": 2}
    cam = sc["camera"]
    renderer.upload(sc["meshes"], sc["lights"], sc["materials"])
    renderer.set_camera(cam["position"], cam["matrix"])
    renderer.change_shading_mode(pkg.MODE_PATH)
    renderer.set_path_params(2, 2, 4321)
    oracle.set_path_params(2, 2, 4321)
    O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    streams = [torch.cuda.Stream() for _ in range(6)]
    try:
        refs = {}
        n = 0
        for n_streams in (1, 6, 2):
            for (w, h), frames in (((160, 120), 9), ((97, 61), 7), ((160, 120), 8)):
                if (w, h) not in refs:
                    refs[(w, h)] = O.render(cam["position"], cam["matrix"], oracle.MODE_PATH, w, h)["rgba8"].reshape(-1, 4).copy().view(np.uint32).ravel()
                bufs = [torch.full((w * h,), 0x7E57AB1E, dtype=torch.int32, device="cuda") for _ in range(frames)]
                torch.cuda.synchronize()
                for b in bufs:
                    renderer.set_stream(streams[n % n_streams].cuda_stream)
                    n += 1
                    renderer.render_frame_device(w, h, b.data_ptr())
                torch.cuda.synchronize()
                for i, b in enumerate(bufs):
                    got = b.cpu().numpy().view(np.uint32)
                    assert np.array_equal(got, refs[(w, h)]), "%d streams, %dx%d frame %d: %d pixels differ" % (
                        n_streams, w, h, i, int((got != refs[(w, h)]).sum()))
        # scratch bound: 4K, 16 samples per pixel, frames on one stream
        renderer.set_stream(streams[0].cuda_stream)
        renderer.set_path_params(16, 3, 1)
        frame = torch.zeros(3840 * 2160, dtype=torch.int32, device="cuda")
        renderer.render_frame_device(256, 256, frame.data_ptr())
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        for _ in range(3):
            renderer.render_frame_device(3840, 2160, frame.data_ptr())
        torch.cuda.synchronize()
        grown = free0 - torch.cuda.mem_get_info()[0]
        assert grown < 768 * 2**20, "a 4K / 16 spp path frame grew device memory by %.0f MB" % (grown / 2**20)
    finally:
        torch.cuda.synchronize()
        oracle.set_path_params(4, 3, 1234)
        renderer.set_path_params(4, 3, 1234)
        renderer.reset_stream()
        renderer.change_shading_mode(0)


def test_batch_launch_equals_single_frames(pkg, oracle, scenes, dragon, renderer):
    """crt_render_frames_batch_device / crt_render_tiles_batch_device: several frames (own camera each) in one launch are
    exactly the frames single launches with those cameras give -- full frames against the oracle, tile shares against the
    single-frame tile call; argument errors are reported."""
    import torch
    sc = _with_normals(scenes, dragon)
    cam = sc["camera"]
    renderer.upload(sc["meshes"], sc["lights"], sc["materials"])
    O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    w, h = 333, 217
    pos0 = np.float32(cam["position"])
    cams = [(pos0 + np.float32([0.7 * k, 0.1 * k, -0.4 * k]), scenes.camera_matrix(yaw_deg=4.0 * k, pitch_deg=-2.0 * k)) for k in range(4)]
    for mode, pipeline in ((100, 0), (pkg.MODE_PATH, 0), (pkg.MODE_PATH, 1)):  # path tracing: persistent kernel, then stage launches
        renderer.change_shading_mode(mode)
        renderer.set_option("path_pipeline", pipeline)
        refs = [O.render(p, r, mode, w, h)["rgba8"].reshape(-1, 4).copy().view(np.uint32).ravel() for p, r in cams]
        for n in (1, 2, 3, 4):
            bufs = [torch.full((w * h,), 0x7E57AB1E, dtype=torch.int32, device="cuda") for _ in range(n)]
            torch.cuda.synchronize()
            st = renderer.render_frames_batch_device(w, h, [b.data_ptr() for b in bufs], cams[:n], stats=True)
            assert st["kernel_ms"] > 0
            for k, b in enumerate(bufs):
                assert np.array_equal(b.cpu().numpy().view(np.uint32), refs[k]), "mode %d pipeline %d batch of %d, frame %d" % (mode, pipeline, n, k)
        if mode == pkg.MODE_PATH:  # a path-traced frame from three ranks' tile shares
            renderer.set_camera(*cams[0])
            slots3 = pkg.tile_slots(w, h, 3)
            gathered3 = torch.zeros(3 * slots3 * 256, dtype=torch.int32, device="cuda")
            out3 = torch.zeros(w * h, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            for rank in range(3):
                renderer.render_tiles_device(w, h, rank, 3, gathered3.data_ptr() + rank * slots3 * 1024, stats=True)
            renderer.untile_device(w, h, 3, gathered3.data_ptr(), out3.data_ptr())
            renderer.synchronize()
            assert np.array_equal(out3.cpu().numpy().view(np.uint32), refs[0]), "path tracing, pipeline %d, three tile shares" % pipeline
    renderer.set_option("path_pipeline", 0)
    # tile shares: batch == single-frame calls with the same cameras (n_ranks 3, every rank)
    renderer.change_shading_mode(100)
    slots = pkg.tile_slots(w, h, 3)
    for rank in range(3):
        single = []
        for p, r in cams[:3]:
            renderer.set_camera(p, r)
            s1 = torch.zeros(slots * 256, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            renderer.render_tiles_device(w, h, rank, 3, s1.data_ptr(), stats=True)
            single.append(s1)
        batch = [torch.zeros(slots * 256, dtype=torch.int32, device="cuda") for _ in range(3)]
        torch.cuda.synchronize()
        renderer.render_tiles_batch_device(w, h, rank, 3, [b.data_ptr() for b in batch], cams[:3], stats=True)
        for k in range(3):
            assert torch.equal(batch[k], single[k]), (rank, k)
        if rank == 0:
            all_single = {0: single}
        else:
            all_single[rank] = single
    # the batched all-gather layout [rank][frame][slot] + crt_untile_batch_device == the frames themselves
    gathered = torch.cat([torch.cat(all_single[rk]) for rk in range(3)])
    for k in range(3):
        out = torch.zeros(w * h, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        renderer.untile_batch_device(w, h, 3, 3, k, gathered.data_ptr(), out.data_ptr())
        renderer.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), refs_for(O, cams[k], w, h)), k
        assert np.array_equal(pkg.untile_host(gathered.cpu().numpy().view(np.uint32), w, h, 3, n_frames=3, frame=k).reshape(-1),
                              out.cpu().numpy().view(np.uint32))
    # cameras=None: every frame uses the current camera
    renderer.set_camera(*cams[1])
    two = [torch.zeros(w * h, dtype=torch.int32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    renderer.render_frames_batch_device(w, h, [b.data_ptr() for b in two], None, stats=True)
    assert torch.equal(two[0], two[1]) and np.array_equal(two[0].cpu().numpy().view(np.uint32), refs_for(O, cams[1], w, h))
    with pytest.raises(pkg.CrtError):
        renderer.render_frames_batch_device(w, h, [two[0].data_ptr()] * 5)
    with pytest.raises(pkg.CrtError):
        renderer.render_frames_batch_device(w, h, [two[0].data_ptr(), 0])
    renderer.set_camera(cam["position"], cam["matrix"])


def refs_for(O, cam, w, h):
    return O.render(cam[0], cam[1], 100, w, h)["rgba8"].reshape(-1, 4).copy().view(np.uint32).ravel()


def test_random_scenes_property(pkg, oracle, scenes, renderer):
    """Property test on the device: random small scenes on an integer lattice (coplanar and duplicated triangles -> equal-t
    ties, zero-area triangles, boxes of zero thickness), cameras on lattice points looking along axes or arbitrary angles
    (direction components exactly 0 -> the 1e-20 clamp, rays inside box faces), all of it against the oracle: hit ids,
    t, RGBA8, float colour and the fetch counters, for primary rays, shadow rays and path tracing."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    lattice = st.integers(-4, 4).map(float)
    coord = st.one_of(lattice, lattice, st.floats(-4, 4, width=32))
    tri = st.tuples(*[st.tuples(coord, coord, coord)] * 3)
    cam_pos = st.tuples(st.integers(-6, 6).map(float), st.integers(-6, 6).map(float), st.integers(3, 9).map(float))
    angles = st.one_of(st.sampled_from([(0.0, 0.0), (90.0, 0.0), (180.0, 0.0), (0.0, 45.0), (45.0, 0.0)]),
                       st.tuples(st.floats(-180, 180, width=32), st.floats(-80, 80, width=32)))
    w, h = 33, 17

    @settings(max_examples=int(os.environ.get("CRT_PROPERTY_EXAMPLES", "60")), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
    @given(st.lists(tri, min_size=1, max_size=24), st.integers(1, 3), cam_pos, angles, st.integers(0, 2 ** 31 - 1))
    def run(tris, n_meshes, pos, ang, seed):
        v = np.float32(tris).reshape(-1, 3)
        t = np.arange(len(v), dtype=np.uint32).reshape(-1, 3)
        cut = [len(t) * k // n_meshes for k in range(n_meshes + 1)]
        meshes = [{"vertices": v, "triangles": t[cut[k]:cut[k + 1]], "material_index": k % 4} for k in range(n_meshes) if cut[k + 1] > cut[k]]
        # seeded extras: smooth-shading normals, per-vertex uvs (outside [0,1] too) and one texture of a random kind
        g = np.random.default_rng(seed)
        nrm = g.normal(size=v.shape).astype(np.float32)
        nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-6).astype(np.float32)
        uv = np.concatenate([g.uniform(-1.5, 2.5, size=(len(v), 2)), np.zeros((len(v), 1))], axis=1).astype(np.float32)
        for m in meshes:
            m["normals"] = nrm
            m["uvs"] = uv
        kind = ("albedo", "edges", "checker", "bitmap")[seed % 4]
        tex = {"type": kind, "color_a": (0.9, 0.2, 0.1), "color_b": (0.1, 0.3, 0.9), "scalar": (0.07, 0.3, 1.0)[seed % 3],
               "pixels": g.integers(0, 256, size=(5, 7, 3), dtype=np.uint8)}
        mats = [{"albedo": (0.8, 0.7, 0.6), "type": 1, "texture": 0, "smooth_shading": bool(seed & 8)}, {"albedo": (0.9, 0.9, 0.9), "type": 2},
                {"albedo": (1.0, 1.0, 1.0), "type": 3, "ior": 1.5}, {"albedo": (0.5, 0.9, 0.4), "type": 1, "smooth_shading": True}]
        lights = [((2.0, 5.0, 3.0), 300.0), ((-3.0, -2.0, 6.0), 150.0)]
        rot = scenes.camera_matrix(yaw_deg=float(ang[0]), pitch_deg=float(ang[1]))
        renderer.upload(meshes, lights, mats, [tex])
        renderer.set_camera(pos, rot)
        O = oracle.OracleScene(meshes, lights, mats, textures=[tex])
        for mode in (3, 100, pkg.MODE_PATH):
            renderer.change_shading_mode(mode)
            if mode == pkg.MODE_PATH:
                renderer.set_path_params(2, 2, seed)
                oracle.set_path_params(2, 2, seed)
                renderer.set_option("path_pipeline", seed & 1)  # persistent kernel / stage launches over global queues, alternating
            for counting in (False, True):
                renderer.set_counting(counting)
                got = renderer.render_frame(w, h)
                ref = O.render(pos, rot, mode, w, h)
                for k in ("hit_inst", "hit_prim", "hit_t", "rgba8"):
                    np.testing.assert_array_equal(got[k], ref[k], err_msg="mode %d %s counting=%s" % (mode, k, counting))
                assert np.array_equal(got["rgb"], ref["rgb"], equal_nan=True), "mode %d rgb" % mode
                if counting:
                    st_, rs = got["stats"], ref["stats"]
                    assert (st_["rays_shadow"], st_["nodes_visited"], st_["tris_tested"]) == (rs["rays_shadow"], rs["nodes_visited"], rs["tris_tested"])

    try:
        run()
    finally:
        renderer.set_counting(False)
        renderer.set_option("path_pipeline", 0)
        renderer.set_path_params(4, 3, 1234)
        oracle.set_path_params(4, 3, 1234)


def test_gpu_bvh_build_property(pkg, oracle, scenes):
    """Property test of the GPU LBVH builder: random meshes with coincident centroids (equal Morton codes -> ordinal
    tie-break), degenerate and duplicated triangles, 1..300 triangles: the tree built by the HIP kernels equals the oracle's
    CPU restatement byte for byte (binary tree, wide tree, leaf-ordered records), and a frame over it equals the oracle's."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    lattice = st.integers(-3, 3).map(float)
    coord = st.one_of(lattice, st.floats(-3, 3, width=32))
    tri = st.tuples(*[st.tuples(coord, coord, coord)] * 3)
    r = pkg.Renderer(0)
    r.set_option("gpu_build", 1)

    @settings(max_examples=int(os.environ.get("CRT_PROPERTY_EXAMPLES", "40")), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
    @given(st.lists(tri, min_size=1, max_size=300), st.integers(1, 4))
    def run(tris, repeat):
        v = np.float32(tris * repeat).reshape(-1, 3)  # `repeat` copies: identical centroids and boxes
        t = np.arange(len(v), dtype=np.uint32).reshape(-1, 3)
        meshes = [{"vertices": v, "triangles": t, "material_index": 0}]
        r.upload(meshes, [((1.0, 4.0, 5.0), 200.0)], [{"albedo": (0.7, 0.7, 0.7), "type": 1}])
        O = oracle.OracleScene(meshes, [((1.0, 4.0, 5.0), 200.0)], [{"albedo": (0.7, 0.7, 0.7), "type": 1}], build_mode=1)
        nodes, tr, shade = r.bvh_export()
        assert nodes.tobytes() == O.nodes().tobytes(), "LBVH binary nodes differ (%d tris)" % len(t)
        assert tr.tobytes() == O.tris().tobytes() and shade.tobytes() == O.shade().tobytes()
        nodes4, depth4 = r.bvh_export4()
        assert nodes4.tobytes() == O.nodes4().tobytes() and depth4 == O.depth4
        pos, rot = np.float32([0.5, 0.5, 7.0]), scenes.IDENTITY
        r.set_camera(pos, rot)
        r.change_shading_mode(100)
        got = r.render_frame(40, 24)
        ref = O.render(pos, rot, 100, 40, 24)
        for k in ("hit_inst", "hit_prim", "hit_t", "rgba8"):
            np.testing.assert_array_equal(got[k], ref[k], err_msg=k)

    try:
        run()
    finally:
        r.close()


def test_runtime_state_machine_stress(pkg, oracle, scenes, dragon, renderer):
    """Randomised sequences over the runtime layer -- stream switches (own / torch streams), frame-size changes, camera and
    mode changes, adaptive_order policies, single frames, batches and tile shares issued back to back without host
    synchronisation -- every frame into its own sentinel-filled buffer, all checked at the end against references
    rendered one at a time.  Exercises the scratch ring, the cost/order buffers and their events, order reuse for an
    unchanged view, and the spill arenas under reordering."""
    import random
    import torch
    sc = _with_normals(scenes, dragon)
    base = sc["camera"]
    renderer.upload(sc["meshes"], sc["lights"], sc["materials"])
    rng = random.Random(20260104)
    sizes = [(160, 90), (97, 61), (256, 144)]
    cams = [(np.float32(base["position"]) + np.float32([0.9 * k, 0.2 * k, -0.5 * k]), scenes.camera_matrix(yaw_deg=5.0 * k, pitch_deg=-3.0 * k)) for k in range(3)]
    modes = [3, 100]
    # references: one synchronous frame per (size, camera, mode), checked against the oracle
    O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    refs = {}
    renderer.reset_stream()
    for si, (w, h) in enumerate(sizes):
        for ci, (p, r) in enumerate(cams):
            for m in modes:
                renderer.set_camera(p, r)
                renderer.change_shading_mode(m)
                got = renderer.render_frame(w, h, want=())["rgba8"].reshape(-1, 4).copy().view(np.uint32).ravel()
                if si == 0:
                    np.testing.assert_array_equal(got, O.render(p, r, m, w, h)["rgba8"].reshape(-1, 4).copy().view(np.uint32).ravel())
                refs[(si, ci, m)] = got
    streams = [torch.cuda.Stream() for _ in range(3)]
    pending = []  # (buffer, key, kind)
    try:
        for rounds in range(max(6, int(os.environ.get("CRT_PROPERTY_EXAMPLES", "60")) // 10)):
            si, ci, m = 0, 0, 100
            renderer.set_camera(*cams[ci]); renderer.change_shading_mode(m)
            # plan the round first and allocate every output up front, so that the issue loop below never synchronises
            plan = []
            for step in range(48):
                op = rng.random()
                pre = None
                if op < 0.15:
                    ci = rng.randrange(3); pre = ("cam", ci)
                elif op < 0.25:
                    m = rng.choice(modes); pre = ("mode", m)
                elif op < 0.35:
                    si = rng.randrange(3)
                elif op < 0.45:
                    pre = ("order", rng.choice([0, 1, 2]))
                elif op < 0.6:
                    pre = ("stream", rng.randrange(4))
                w, h = sizes[si]
                kind = rng.random()
                if kind < 0.6:
                    bufs = [torch.full((w * h,), 0x7E57AB1E, dtype=torch.int32, device="cuda")]
                    what = ("frame", None)
                elif kind < 0.8:
                    bufs = [torch.full((w * h,), 0x7E57AB1E, dtype=torch.int32, device="cuda") for _ in range(rng.randrange(1, 5))]
                    what = ("batch", None)
                else:
                    n = rng.choice([2, 3, 8]); rank = rng.randrange(n)
                    bufs = [torch.zeros(pkg.tile_slots(w, h, n) * 256, dtype=torch.int32, device="cuda")]
                    what = ("tiles", (rank, n))
                plan.append((pre, (w, h), what, bufs, (si, ci, m)))
            torch.cuda.synchronize()
            for pre, (w, h), (kind, extra), bufs, key in plan:
                if pre:
                    if pre[0] == "cam": renderer.set_camera(*cams[pre[1]])
                    elif pre[0] == "mode": renderer.change_shading_mode(pre[1])
                    elif pre[0] == "order": renderer.set_option("adaptive_order", pre[1])
                    elif pre[1] == 3: renderer.reset_stream()
                    else: renderer.set_stream(streams[pre[1]].cuda_stream)
                if kind == "frame":
                    renderer.render_frame_device(w, h, bufs[0].data_ptr())
                elif kind == "batch":
                    renderer.render_frames_batch_device(w, h, [b.data_ptr() for b in bufs])
                else:
                    renderer.render_tiles_device(w, h, extra[0], extra[1], bufs[0].data_ptr())
                pending += [(b, key, "tiles" if kind == "tiles" else "frame", extra) for b in bufs]
            torch.cuda.synchronize()
            renderer.synchronize()
            for b, key, kind, extra in pending:
                w, h = sizes[key[0]]
                if kind == "frame":
                    assert np.array_equal(b.cpu().numpy().view(np.uint32), refs[key]), (rounds, key)
                else:
                    rank, n = extra
                    exp = pkg.tile_host(refs[key].reshape(h, w), w, h, rank, n).reshape(-1)
                    assert np.array_equal(b.cpu().numpy().view(np.uint32), exp), (rounds, key, extra)
            pending.clear()
    finally:
        torch.cuda.synchronize()
        renderer.set_option("adaptive_order", 2)
        renderer.reset_stream()
        renderer.set_camera(base["position"], base["matrix"])
