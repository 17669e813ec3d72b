"""Host scene layer (crt::Scene / Camera / Mesh behind crt_scene_*) against known answers produced by the
reference's own CRT* sources (tests/golden/dragon_scene_layer.json, made by oracle/make_golden.py)."""
import json
import os

import numpy as np
import pytest

REF_DRAGON = "/root/reference/DirectX-RayTracer/DirectX-RayTracer/Scenes/Dragon.crtscene"


@pytest.fixture(scope="module")
def gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, "dragon_scene_layer.json")))


@pytest.fixture(scope="module")
def scene(pkg, golden_dir):
    return pkg.Scene(os.path.join(golden_dir, "dragon.crtscene"))


def test_record_sizes_pinned_by_reference(gold, pkg):
    # sizeof(CRTVector)=12, sizeof(CRTMatrix)=36, sizeof(CRTTriangle)=48: static_assert'ed in csrc/scene.h
    assert gold["sizeof"] == {"CRTVector": 12, "CRTMatrix": 36, "CRTTriangle": 48}
    text = open(os.path.join(os.path.dirname(pkg.LIB_PATH), "csrc", "scene.h")).read()
    for s in ("sizeof(Vector) == 12", "sizeof(Matrix) == 36", "sizeof(Triangle) == 48"):
        assert s in text


def test_parse_counts_and_settings(scene, gold):
    assert scene.mesh_count == len(gold["meshes"]) == 2
    for i, gm in enumerate(gold["meshes"]):
        m = scene.mesh(i)
        assert len(m["vertices"]) == gm["n_vertices"] and m["triangles"].size == gm["n_indices"]
        assert m["material_index"] == gm["material_index"]
        np.testing.assert_allclose(m["vertices"].astype(np.float64).sum(axis=0), gm["vertex_sum"], rtol=0, atol=1e-9)
        assert int(m["triangles"].astype(np.int64).sum()) == gm["index_sum"]
    st = scene.settings()
    assert (st["width"], st["height"]) == (gold["settings"]["width"], gold["settings"]["height"])
    np.testing.assert_array_equal(np.float32(st["background_color"]), np.float32(gold["settings"]["background_color"]))
    assert scene.texture_count == gold["n_textures"] == 0


def test_lights_and_materials(scene, gold):
    lights = scene.lights()
    assert len(lights) == len(gold["lights"]) == 4
    for (pos, inten), g in zip(lights, gold["lights"]):
        np.testing.assert_array_equal(np.float32(pos), np.float32(g["position"]))
        assert np.float32(inten) == np.float32(g["intensity"])
    mats = scene.materials()
    assert len(mats) == 2
    for m, g in zip(mats, gold["materials"]):
        assert m["type"] == g["type"] and m["smooth_shading"] == bool(g["smooth_shading"])
        np.testing.assert_array_equal(np.float32(m["albedo"]), np.float32(g["albedo"]))


def test_vertex_normals_match_reference(scene, gold):
    """CRTMesh::calculateVertexNormals (R/CRTMesh.cpp:66-94), 2 012 normals, bit-exact."""
    for i, gm in enumerate(gold["meshes"]):
        n = scene.mesh(i)["normals"]
        ref = np.array(gm["vertex_normals"], dtype=np.float32)
        assert n.shape == ref.shape
        np.testing.assert_array_equal(n, ref)


def test_camera_initial(scene, gold):
    pos, rot = scene.camera()
    np.testing.assert_array_equal(pos, np.float32(gold["camera"]["position"]))
    np.testing.assert_array_equal(rot, np.float32(gold["camera"]["matrix"]))


def test_camera_operation_sequence(pkg, golden_dir, gold):
    """CRTCamera::rotate/zoom/moveForward/moveRight/pan/tilt/roll/panAroundTarget (R/CRTCamera.cpp:9-130),
    including both pitch clamps, state after every step against the reference's values."""
    s = pkg.Scene(os.path.join(golden_dir, "dragon.crtscene"))
    ops = [lambda: None, lambda: s.rotate(10.0, 5.0), lambda: s.zoom(2.5), lambda: s.move_forward(-1.25),
           lambda: s.move_right(3.0), lambda: s.rotate(-35.0, -120.0), lambda: s.rotate(200.0, 300.0),
           lambda: s.move_forward(0.5), lambda: s.pan(30.0), lambda: s.tilt(-12.0), lambda: s.roll(7.0),
           lambda: s.pan_around_target(45.0, (0.0, 0.0, 0.0))]
    assert len(ops) == len(gold["camera_sequence"])
    for op, g in zip(ops, gold["camera_sequence"]):
        op()
        pos, rot = s.camera()
        np.testing.assert_allclose(pos, np.float32(g["position"]), rtol=0, atol=2e-6, err_msg=g["op"])
        np.testing.assert_allclose(rot, np.float32(g["matrix"]), rtol=0, atol=2e-7, err_msg=g["op"])


def test_python_vertex_normals_helper_matches(scenes, dragon, gold):
    n = scenes.vertex_normals(dragon["meshes"][1]["vertices"], dragon["meshes"][1]["triangles"])
    np.testing.assert_allclose(n, np.array(gold["meshes"][1]["vertex_normals"], dtype=np.float32), rtol=0, atol=1e-6)


@pytest.mark.skipif(not os.path.exists(REF_DRAGON), reason="reference tree not mounted (GPU box)")
def test_original_scene_file_parses_to_the_same_data(pkg, scene):
    """The reference's shipped Dragon.crtscene, read in place, gives the same arrays as the re-serialised fixture."""
    orig = pkg.Scene(REF_DRAGON)
    assert orig.mesh_count == scene.mesh_count
    for i in range(orig.mesh_count):
        a, b = orig.mesh(i), scene.mesh(i)
        np.testing.assert_array_equal(a["vertices"], b["vertices"])
        np.testing.assert_array_equal(a["triangles"], b["triangles"])
        np.testing.assert_array_equal(a["normals"], b["normals"])
    assert orig.lights() == scene.lights() and orig.materials() == scene.materials()
    assert orig.settings() == scene.settings()


def test_absent_keys_take_defaults(pkg, tmp_path):
    """Keys the reference dereferences when absent (uvs, textures, material_index, smooth_shading, ior) default here."""
    p = tmp_path / "min.crtscene"
    p.write_text('{"objects":[{"vertices":[0,0,0, 1,0,0, 0,1,0],"triangles":[0,1,2]}],'
                 '"materials":[{"type":"glass"},{"type":"diffuse","albedo":"checker_tex"}],'
                 '"textures":[{"name":"checker_tex","type":"checker","color_A":[1,0,0],"color_B":[0,0,1],"square_size":0.25}]}')
    s = pkg.Scene(str(p))
    assert s.mesh_count == 1 and s.mesh(0)["material_index"] == 0
    mats = s.materials()
    assert mats[0]["type"] == 3 and mats[0]["ior"] == 1.0 and mats[0]["albedo"] == (1.0, 1.0, 1.0) and not mats[0]["smooth_shading"]
    assert mats[1]["type"] == 1
    assert s.texture_count == 1
    pos, rot = s.camera()
    np.testing.assert_array_equal(rot, np.eye(3, dtype=np.float32).reshape(9))
    np.testing.assert_array_equal(s.mesh(0)["normals"], np.float32([[0, 0, 1]] * 3))


def test_parse_errors_are_reported_not_asserted(pkg, tmp_path):
    with pytest.raises(pkg.CrtError) as e:
        pkg.Scene(str(tmp_path / "missing.crtscene"))
    assert "rc=6" in str(e.value)  # CRT_EIO
    bad = tmp_path / "bad.crtscene"
    bad.write_text('{"objects":[{"vertices":[0,0,0],"triangles":[0,1,2]}]}')
    with pytest.raises(pkg.CrtError) as e:
        pkg.Scene(str(bad))
    assert "rc=7" in str(e.value) and "out of range" in str(e.value)
    bad.write_text('{"objects": [')
    with pytest.raises(pkg.CrtError):
        pkg.Scene(str(bad))


def test_obj_loader(pkg, tmp_path):
    p = tmp_path / "quad.obj"
    p.write_text("# quad + tri\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\no second\nv 0 0 1\nv 1 0 1\nv 0 1 1\nf -3/1/1 -2/2/2 -1/3/3\n")
    s = pkg.Scene(str(p))
    assert s.mesh_count == 2
    a, b = s.mesh(0), s.mesh(1)
    assert len(a["vertices"]) == 4 and a["triangles"].tolist() == [[0, 1, 2], [0, 2, 3]]
    assert len(b["vertices"]) == 3 and b["triangles"].tolist() == [[0, 1, 2]]
    np.testing.assert_array_equal(b["vertices"][:, 2], np.float32([1, 1, 1]))
    assert len(s.materials()) == 1


def test_programmatic_scene_roundtrip(pkg, scenes):
    sc = scenes.cornell_box()
    s = pkg.Scene.from_arrays(sc)
    assert s.mesh_count == 6 and sum(len(m["triangles"]) for m in s.meshes()) == 32
    pos, rot = s.camera()
    np.testing.assert_array_equal(pos, sc["camera"]["position"])


def test_binary_cache_roundtrip(pkg, scene, tmp_path, scenes):
    """.crtbin holds everything the JSON scene yields (normals included) and loads back identically"""
    p = str(tmp_path / "dragon.crtbin")
    scene.save(p)
    back = pkg.Scene(p)
    assert back.mesh_count == scene.mesh_count
    for i in range(scene.mesh_count):
        a, b = scene.mesh(i), back.mesh(i)
        for k in ("vertices", "triangles", "normals"):
            np.testing.assert_array_equal(a[k], b[k])
        assert a["material_index"] == b["material_index"]
    assert back.lights() == scene.lights() and back.materials() == scene.materials() and back.settings() == scene.settings()
    np.testing.assert_array_equal(back.camera()[0], scene.camera()[0])
    np.testing.assert_array_equal(back.camera()[1], scene.camera()[1])
    assert os.path.getsize(p) < 0.9 * os.path.getsize(os.path.join(os.path.dirname(__file__), "golden", "dragon.crtscene"))
    # a synthetic mesh built through the C API survives too, and loads much faster than its JSON would parse
    big = pkg.Scene.from_arrays(scenes.heightfield(n=120))
    pb = str(tmp_path / "hf.crtbin")
    big.save(pb)
    b2 = pkg.Scene(pb)
    np.testing.assert_array_equal(b2.mesh(1)["vertices"], big.mesh(1)["vertices"])
    np.testing.assert_array_equal(b2.mesh(1)["normals"], big.mesh(1)["normals"])


def test_binary_cache_rejects_damaged_files(pkg, scene, tmp_path):
    p = str(tmp_path / "d.crtbin")
    scene.save(p)
    raw = open(p, "rb").read()
    for name, data in (("trunc.crtbin", raw[:len(raw) // 2]), ("magic.crtbin", b"XXXX" + raw[4:]), ("tail.crtbin", raw + b"\\0"),
                       ("ver.crtbin", raw[:4] + b"\\x09\\0\\0\\0" + raw[8:])):
        q = tmp_path / name
        q.write_bytes(data)
        with pytest.raises(pkg.CrtError) as e:
            pkg.Scene(str(q))
        assert "rc=7" in str(e.value) and "crtbin" in str(e.value)
    # index out of range inside an otherwise well-formed file
    bad = bytearray(raw)
    idx_pos = raw.rfind(np.uint32([0, 1, 2]).tobytes())  # first triangle of the ground quad
    bad[idx_pos:idx_pos + 4] = np.uint32([0x7FFFFFFF]).tobytes()
    q = tmp_path / "idx.crtbin"
    q.write_bytes(bytes(bad))
    with pytest.raises(pkg.CrtError):
        pkg.Scene(str(q))
    with pytest.raises(pkg.CrtError):
        scene.save(str(tmp_path / "no_such_dir" / "x.crtbin"))


# ----------------------------------------------------------------------------------------- textures (row f3)
def _texture_goldens(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "texture_known_answers.json")))
    raw = open(os.path.join(golden_dir, "tex7x5.ppm"), "rb").read()
    img = np.frombuffer(raw[raw.index(b"255\n") + 4:], dtype=np.uint8).reshape(5, 7, 3)
    return g, img


def test_texture_classes_match_the_reference(pkg, oracle, golden_dir):
    """CRTTexture{Albedo,Edges,Checker,Bitmap}::getColor (R/CRTTexture*.cpp) on a 21x21 grid of (u,v): known answers
    produced by the reference's own classes (its bitmap one decoding tests/golden/tex7x5.ppm through its vendored
    stb_image); the host scene layer AND the oracle's restatement must reproduce every value exactly."""
    g, img = _texture_goldens(golden_dir)
    s = pkg.Scene()
    s.add_texture("a", "albedo", g["params"]["albedo"])
    e = g["params"]["edges"]
    s.add_texture("e", "edges", e["edge_color"], e["inner_color"], e["edge_width"])
    c = g["params"]["checker"]
    s.add_texture("c", "checker", c["color_A"], c["color_B"], c["square_size"])
    s.add_texture("b", "bitmap", file_path=os.path.join(golden_dir, "tex7x5.ppm"))
    assert s.texture_count == 4
    odesc = {"albedo": {"type": "albedo", "color_a": g["params"]["albedo"]},
             "edges": {"type": "edges", "color_a": e["edge_color"], "color_b": e["inner_color"], "scalar": e["edge_width"]},
             "checker": {"type": "checker", "color_a": c["color_A"], "color_b": c["color_B"], "scalar": c["square_size"]},
             "bitmap": {"type": "bitmap", "pixels": img}}
    for i, name in enumerate(("albedo", "edges", "checker", "bitmap")):
        assert len(g[name]) > 200
        for u, v, r, gg, b in g[name]:
            exp = np.float32([r, gg, b])
            np.testing.assert_array_equal(s.texture_color(i, u, v), exp, err_msg="%s host (%g,%g)" % (name, u, v))
            np.testing.assert_array_equal(oracle.texture_color(odesc[name], u, v), exp, err_msg="%s oracle (%g,%g)" % (name, u, v))


def test_bitmap_formats_match_the_reference(pkg, golden_dir):
    """Bitmap textures in the formats a .crtscene realistically names.  The reference decodes them with its vendored stb_image
    (R/CRTTextureBitmap.cpp:10); this repo with its own decoders (csrc/image_decode.cpp: PNG with its own inflate, BMP, TGA;
    csrc/jpeg_decode.cpp).  Known answers: the reference's CRTTextureBitmap::getColor over the same seeded files
    (oracle/make_golden.py), bit for bit -- PNG in every colour type (grey, grey + alpha, RGB, RGBA, palette with and without
    tRNS; colour-key tRNS on grey and RGB files, which adds an alpha channel) at every bit depth each allows (1 / 2 / 4 / 8 / 16,
    Adam7 also below 8 bits), all five scanline filters, stored / fixed / dynamic deflate blocks, several IDAT chunks, Adam7;
    BMP 24 / 32 bit / palette / top-down; TGA raw and run-length coded, colour and grey; JPEG baseline and progressive (also
    with optimised Huffman tables), 4:4:4 / 4:2:2 / 4:2:0 / 4:1:1, grey, Adobe CMYK, restart intervals, quality 10, and images one
    or two texels wide -- a JPEG's texels depend on the decoder's inverse DCT, chroma filter and colour matrix, so these pin the
    arithmetic, not only the parsing; GIF (first image, four channels): global and local colour tables, interlaced, transparent
    index, an image smaller than its canvas with a background index (red and blue exchanged there, as the reference's decoder
    leaves them); Radiance HDR flat and run-length coded through that decoder's tone curve; PSD 8 / 16 bit raw and PackBits RGBA
    un-matted from white; Softimage PIC raw, mixed and pure run-length packets with and without alpha, 16-bit run lengths.
    BMP also with 16 bits per pixel (5-5-5, 5-6-5), channel masks at odd places, 4-4-4-4 with alpha in a V4 header, 4- and 1-bit
    palettes; TGA also colour-mapped (8- and 16-bit indices, 24-bit and 5-5-5 entries, an index outside the palette), 5-5-5 true
    colour, grey + alpha, the ignored right-to-left bit; PNM with maxima of 100, 1000 and 65535.
    With these every format (and every variant of it) the reference's loader accepts is read here."""
    answers = json.load(open(os.path.join(golden_dir, "bitmap_known_answers.json")))
    assert len(answers) >= 72 and all(sum(n.endswith(e) for n in answers) >= k for e, k in ((".jpg", 14), (".gif", 6), (".hdr", 3), (".psd", 3), (".pic", 3), (".bmp", 10), (".tga", 8), (".png", 22)))
    f32 = np.float32
    for name, rows in sorted(answers.items()):
        s = pkg.Scene()
        s.add_texture("b", "bitmap", file_path=os.path.join(golden_dir, name))
        for iu, iv, r, g, b in rows:
            u = f32(f32(iu) * f32(0.05) + f32(iu % 3) * f32(0.003))
            v = f32(f32(iv) * f32(0.05) + f32(iv % 4) * f32(0.002))
            exp = np.array([f32(r) / f32(255.0), f32(g) / f32(255.0), f32(b) / f32(255.0)], dtype=np.float32)
            np.testing.assert_array_equal(s.texture_color(0, u, v), exp, err_msg="%s (%d,%d)" % (name, iu, iv))


def test_damaged_bitmap_files_are_errors_not_crashes(pkg, golden_dir, tmp_path):
    """every truncation of every fixture, and a few hundred single-byte corruptions, either decode or raise CrtError"""
    rng = np.random.default_rng(5)
    names = sorted(n for n in os.listdir(golden_dir) if n.startswith("tex_") and n != "tex_rgb8_big.png")
    tried = failed = 0
    for name in names:
        data = open(os.path.join(golden_dir, name), "rb").read()
        cuts = list(range(0, len(data), max(1, len(data) // 40)))
        variants = [data[:c] for c in cuts]
        for _ in range(40):
            b = bytearray(data)
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            variants.append(bytes(b))
        for i, blob in enumerate(variants):
            path = tmp_path / ("v%d_%s" % (i, name))
            path.write_bytes(blob)
            tried += 1
            try:
                s = pkg.Scene()
                s.add_texture("b", "bitmap", file_path=str(path))
                s.texture_color(0, 0.5, 0.5)
            except pkg.CrtError:
                failed += 1
            path.unlink()
    assert tried > 1000 and failed > tried // 4
    # a format nobody reads here (the reference's stb_image neither): a clear error naming the ones that are
    webp = tmp_path / "x.webp"
    webp.write_bytes(b"RIFF\x20\0\0\0WEBPVP8 " + b"\0" * 32)
    with pytest.raises(pkg.CrtError):
        pkg.Scene().add_texture("b", "bitmap", file_path=str(webp))
    scene = tmp_path / "webp.crtscene"
    scene.write_text('{"settings":{"background_color":[0,0,0],"image_settings":{"width":4,"height":4}},'
                     '"camera":{"matrix":[1,0,0,0,1,0,0,0,1],"position":[0,0,0]},"lights":[],"materials":[],'
                     '"textures":[{"name":"t","type":"bitmap","file_path":"x.webp"}],"objects":[]}')
    with pytest.raises(pkg.CrtError, match="not a PNG, JPEG, GIF, BMP, TGA, PSD, Radiance HDR, Softimage PIC"):
        pkg.Scene(str(scene))
    # a PIC without an extent
    pic = tmp_path / "x.pic"
    pic.write_bytes(b"\x53\x80\xf6\x34" + b"\0" * 84 + b"PICT" + b"\0" * 16)
    with pytest.raises(pkg.CrtError):
        pkg.Scene().add_texture("b", "bitmap", file_path=str(pic))
    # a PSD in a colour mode other than RGB, an HDR in another pixel format
    for name, blob in (("cmyk.psd", b"8BPS\0\1" + b"\0" * 6 + b"\0\4" + b"\0\0\0\2\0\0\0\2\0\x08\0\4" + b"\0" * 16),
                       ("xyze.hdr", b"#?RADIANCE\nFORMAT=32-bit_rle_xyze\n\n-Y 2 +X 2\n" + b"\0" * 16)):
        path = tmp_path / name
        path.write_bytes(blob)
        with pytest.raises(pkg.CrtError):
            pkg.Scene().add_texture("b", "bitmap", file_path=str(path))
    # a GIF whose canvas has no extent, and one without any image
    for blob in (b"GIF89a" + b"\0" * 64, b"GIF89a\x02\x00\x02\x00\x00\x00\x00\x3B"):
        gif = tmp_path / "x.gif"
        gif.write_bytes(blob)
        with pytest.raises(pkg.CrtError):
            pkg.Scene().add_texture("b", "bitmap", file_path=str(gif))
    # a JPEG kind this decoder does not read (arithmetic coding) says so
    arith = tmp_path / "arith.jpg"
    arith.write_bytes(b"\xff\xd8\xff\xc9\x00\x0b\x08\x00\x04\x00\x04\x01\x01\x11\x00\xff\xd9")
    with pytest.raises(pkg.CrtError):
        pkg.Scene().add_texture("b", "bitmap", file_path=str(arith))


def test_textured_scene_file(pkg, tmp_path, golden_dir):
    """a .crtscene with uvs, a texture-named albedo and all four texture kinds (the keys R/CRTSceneParser.cpp:83-306 reads)"""
    import shutil
    shutil.copy(os.path.join(golden_dir, "tex7x5.ppm"), tmp_path / "img.ppm")
    p = tmp_path / "tex.crtscene"
    p.write_text(json.dumps({
        "objects": [{"material_index": 0, "vertices": [0, 0, 0, 1, 0, 0, 0, 1, 0], "triangles": [0, 1, 2], "uvs": [0, 0, 0, 1, 0, 0, 0, 1, 0]}],
        "materials": [{"type": "diffuse", "albedo": "chk", "smooth_shading": False}, {"type": "diffuse", "albedo": "missing_name"}],
        "textures": [{"name": "alb", "type": "albedo", "albedo": [0.1, 0.2, 0.3]},
                     {"name": "edg", "type": "edges", "edge_color": [1, 1, 1], "inner_color": [0, 0, 0], "edge_width": 0.05},
                     {"name": "chk", "type": "checker", "color_A": [1, 0, 0], "color_B": [0, 1, 0], "square_size": 0.25},
                     {"name": "pic", "type": "bitmap", "file_path": "img.ppm"}]}))
    s = pkg.Scene(str(p))
    assert s.texture_count == 4
    mats = s.materials()
    assert mats[0]["texture"] == 2 and mats[1]["texture"] == -1  # by name; unknown names resolve to none
    np.testing.assert_array_equal(s.mesh(0)["uvs"], np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]]))
    np.testing.assert_array_equal(s.texture_color(2, 0.3, 0.1), np.float32([0, 1, 0]))
    np.testing.assert_array_equal(s.texture_color(0, 0.9, 0.9), np.float32([0.1, 0.2, 0.3]))
    assert np.all(s.texture_color(3, 0.5, 0.5) > 0)
    # survives the binary cache (bitmap re-read from its path next to the cache file)
    s.save(str(tmp_path / "tex.crtbin"))
    b = pkg.Scene(str(tmp_path / "tex.crtbin"))
    assert b.texture_count == 4 and b.materials()[0]["texture"] == 2
    np.testing.assert_array_equal(b.texture_color(3, 0.5, 0.5), s.texture_color(3, 0.5, 0.5))
    np.testing.assert_array_equal(b.mesh(0)["uvs"], s.mesh(0)["uvs"])
    # a scene written on the reference's platform may spell the path with backslashes: found with '/' in their place
    (tmp_path / "textures").mkdir()
    shutil.copy(os.path.join(golden_dir, "tex_444.jpg"), tmp_path / "textures" / "wood.jpg")
    q = tmp_path / "win.crtscene"
    q.write_text(json.dumps({"objects": [], "materials": [], "textures": [{"name": "pic", "type": "bitmap", "file_path": "textures\\wood.jpg"}]}))
    assert np.all(pkg.Scene(str(q)).texture_color(0, 0.5, 0.5) >= 0)
    # a damaged bitmap is a load error, not a crash (the reference never checks stbi_load's result)
    (tmp_path / "img.ppm").write_bytes(b"\\x89PNG....")
    with pytest.raises(pkg.CrtError) as e:
        pkg.Scene(str(p))
    assert "PPM" in str(e.value)


# ------------------------------------------------------------------------------------------- loader fuzzing
def test_loaders_survive_arbitrary_input(pkg, tmp_path, golden_dir):
    """Whatever bytes a scene file holds, crt_scene_load either parses it or reports CRT_EPARSE / CRT_EIO through the error
    channel: no crash, no hang, no out-of-range index left in a scene that loaded (the reference asserts or reads
    uninitialised members, R/CRTSceneParser.cpp:407-427).  Mutations of the real Dragon scene + generated JSON / OBJ text."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    base = open(os.path.join(golden_dir, "dragon.crtscene"), "rb").read()[:6000]  # header + the first vertices: enough structure

    def load(name, data):
        p = tmp_path / name
        p.write_bytes(data)
        try:
            s = pkg.Scene(str(p))
        except pkg.CrtError as e:
            assert str(e)  # a message, not an empty error
            return None
        # a scene that loaded must be self-consistent
        for i in range(s.mesh_count):
            m = s.mesh(i)
            nv = len(m["vertices"])
            if len(m["triangles"]):
                assert int(m["triangles"].max()) < nv
        return s

    json_leaf = st.one_of(st.none(), st.booleans(), st.integers(-5, 2 ** 33), st.floats(allow_nan=False, allow_infinity=False, width=32),
                          st.text(max_size=6))
    json_val = st.recursive(json_leaf, lambda c: st.one_of(st.lists(c, max_size=6),
                            st.dictionaries(st.sampled_from(["objects", "vertices", "triangles", "uvs", "material_index", "settings", "camera",
                                                             "matrix", "position", "lights", "intensity", "materials", "type", "albedo",
                                                             "textures", "name", "image_settings", "width", "height", "background_color"]),
                                            c, max_size=6)), max_leaves=25)

    @settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
    @given(st.data())
    def run(data):
        kind = data.draw(st.sampled_from(["mutate", "truncate", "json", "obj", "bin"]))
        if kind == "mutate":
            b = bytearray(base)
            for _ in range(data.draw(st.integers(1, 8))):
                b[data.draw(st.integers(0, len(b) - 1))] = data.draw(st.integers(0, 255))
            load("m.crtscene", bytes(b))
        elif kind == "truncate":
            load("t.crtscene", base[:data.draw(st.integers(0, len(base)))])
        elif kind == "json":
            load("j.crtscene", json.dumps(data.draw(json_val)).encode())
        elif kind == "obj":
            lines = data.draw(st.lists(st.one_of(
                st.builds(lambda a, b, c: "v %g %g %g" % (a, b, c), *[st.floats(-10, 10, width=32)] * 3),
                st.builds(lambda a, b, c: "f %d %d %d" % (a, b, c), *[st.integers(-3, 12)] * 3),
                st.builds(lambda a, b: "f %d/%d %d//%d x" % (a, b, a, b), st.integers(0, 9), st.integers(0, 9)),
                st.text(alphabet="vf 0123456789/.-#\t", max_size=20)), max_size=30))
            load("o.obj", "\n".join(lines).encode())
        else:
            load("b.crtbin", data.draw(st.binary(max_size=200)))

    run()
