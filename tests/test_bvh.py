"""The product's host BVH builder (crt_bvh_build_host, csrc/bvh_build.cpp) against the oracle's independent
restatement of the same spec (byte-identical trees) and against structural invariants."""
import numpy as np
import pytest


def _check_structure(nodes, tris, n_input, max_depth_reported):
    """every triangle in exactly one leaf, leaves <= 4 triangles, child boxes bound their triangles, depth <= 32,
    nodes in DFS pre-order."""
    if n_input == 0:
        assert len(nodes) == 0 and len(tris) == 0
        return
    seen = np.zeros(len(tris), dtype=np.int32)
    v0 = tris["v0"].astype(np.float64)
    v1 = v0 + tris["e1"]
    v2 = v0 + tris["e2"]
    tmin = np.minimum(np.minimum(v0, v1), v2)
    tmax = np.maximum(np.maximum(v0, v1), v2)
    order = []
    stack = [(0, 0)]
    deepest = 0
    while stack:
        i, depth = stack.pop()
        order.append(i)
        n = nodes[i]
        for side, ref in (("l", int(n["left"])), ("r", int(n["right"]))):
            lo = np.array([n[side + "x0"], n[side + "y0"], n[side + "z0"]], dtype=np.float64)
            hi = np.array([n[side + "x1"], n[side + "y1"], n[side + "z1"]], dtype=np.float64)
            if ref >= 0:
                assert ref > i, "children come after parents"
                c = nodes[ref]
                clo = np.minimum([c["lx0"], c["ly0"], c["lz0"]], [c["rx0"], c["ry0"], c["rz0"]])
                chi = np.maximum([c["lx1"], c["ly1"], c["lz1"]], [c["rx1"], c["ry1"], c["rz1"]])
                assert np.all(clo >= lo - 1e-6) and np.all(chi <= hi + 1e-6)
            else:
                code = ~ref & 0xFFFFFFFF
                first, cnt = code >> 3, code & 7
                assert cnt <= 4 and first + cnt <= len(tris)
                deepest = max(deepest, depth + 1)
                if cnt:
                    seen[first:first + cnt] += 1
                    assert np.all(tmin[first:first + cnt] >= lo - 1e-5) and np.all(tmax[first:first + cnt] <= hi + 1e-5)
                    # leaf boxes are tight unions (vertices are rebuilt from v0 + edge here, hence the ulp of slack)
                    np.testing.assert_allclose(tmin[first:first + cnt].min(axis=0), lo, rtol=1e-6, atol=1e-6)
                    np.testing.assert_allclose(tmax[first:first + cnt].max(axis=0), hi, rtol=1e-6, atol=1e-6)
        if int(n["right"]) >= 0:
            stack.append((int(n["right"]), depth + 1))
        if int(n["left"]) >= 0:
            stack.append((int(n["left"]), depth + 1))
    assert np.all(seen == 1), "every triangle in exactly one leaf"
    assert order == list(range(len(nodes))), "DFS pre-order"
    assert deepest <= 32 and max_depth_reported <= 32
    assert sorted(tris["gid"].tolist()) == list(range(n_input))


def _check_wide(nodes, nodes4, depth4, tris):
    """the wide tree is a collapse of the binary one: same leaves, each exactly once, every child box inside the
    union its binary ancestors gave it, DFS pre-order, at most 4 children, unused slots marked empty"""
    if len(nodes) == 0:
        assert len(nodes4) == 0
        return
    EMPTY = -1
    seen = np.zeros(len(tris), dtype=np.int32)
    order, stack, deepest = [], [(0, 1)], 0
    while stack:
        i, d = stack.pop()
        order.append(i)
        deepest = max(deepest, d)
        n = nodes4[i]
        refs = [int(r) for r in n["ref"]]
        used = [r for r in refs if r != EMPTY]
        # (a scene of one leaf is wrapped in a node whose second child is the leaf of no triangles = the unused-slot marker)
        assert (1 if len(nodes4) == 1 else 2) <= len(used) <= 4 and refs[:len(used)] == used, "empty slots trail"
        for k in reversed(range(len(used))):
            r = used[k]
            assert n["minx"][k] <= n["maxx"][k] and n["miny"][k] <= n["maxy"][k] and n["minz"][k] <= n["maxz"][k]
            if r >= 0:
                assert r > i
                stack.append((r, d + 1))
            else:
                code = ~r & 0xFFFFFFFF
                first, cnt = code >> 3, code & 7
                seen[first:first + cnt] += 1
    assert order == list(range(len(nodes4))), "DFS pre-order"
    assert np.all(seen == 1)
    assert deepest == depth4 <= 32
    assert len(nodes4) <= len(nodes)


def _check_quantised(nodes4, q):
    """the 64-byte nodes the kernels fetch: decoded child boxes (the kernel's own decode expression, fma(q, s, lo) in float32)
    contain the full-precision boxes and exceed them by less than two quanta; unused slots are marked; refs are copied"""
    assert len(q) == len(nodes4)
    if len(q) == 0:
        return
    EMPTY = -1
    np.testing.assert_array_equal(q["ref"], nodes4["ref"])
    used = nodes4["ref"] != EMPTY
    for a, (qlo, qhi, mn, mx) in enumerate((("qlo_x", "qhi_x", "minx", "maxx"), ("qlo_y", "qhi_y", "miny", "maxy"), ("qlo_z", "qhi_z", "minz", "maxz"))):
        lo, s = q["lo"][:, a].astype(np.float64), q["s"][:, a].astype(np.float64)
        assert np.all(s > 0)
        for k in range(4):
            l = ((q[qlo] >> (8 * k)) & 0xFF).astype(np.float64)
            h = ((q[qhi] >> (8 * k)) & 0xFF).astype(np.float64)
            dec_lo = (l * s + lo).astype(np.float32)  # one rounding of the exact value = fmaf
            dec_hi = (h * s + lo).astype(np.float32)
            u = used[:, k]
            fin = u & np.isfinite(nodes4[mn][:, k]) & np.isfinite(nodes4[mx][:, k])
            assert np.all(dec_lo[fin] <= nodes4[mn][:, k][fin]) and np.all(dec_hi[fin] >= nodes4[mx][:, k][fin]), "quantised box must contain the exact one"
            slack = 2.0 * s[fin] + 1e-6 * (np.abs(lo[fin]) + 255.0 * s[fin])
            assert np.all(nodes4[mn][:, k][fin] - dec_lo[fin] <= slack) and np.all(dec_hi[fin] - nodes4[mx][:, k][fin] <= slack), "and be tight to two quanta"
            assert np.all(l[~u] == 0) and np.all(h[~u] == 0)  # a point at the node's minimum corner


def _compare(pkg, oracle, meshes):
    nodes, tris, shade, md = pkg.build_bvh_host(meshes)
    O = oracle.OracleScene(meshes)
    nodes4, depth4 = pkg.build_bvh4_host(meshes)
    assert nodes4.tobytes() == O.nodes4().tobytes() and depth4 == O.depth4
    _check_wide(nodes, nodes4, depth4, tris)
    q = pkg.quantize4(nodes4)  # the product's quantiser against the oracle's restatement of the rule, byte for byte
    assert q.tobytes() == O.nodes4q().tobytes()
    _check_quantised(nodes4, q)
    assert nodes.tobytes() == O.nodes().tobytes()
    assert tris.tobytes() == O.tris().tobytes()
    assert shade.tobytes() == O.shade().tobytes()
    assert md == O.max_depth
    _check_structure(nodes, tris, sum(len(m["triangles"]) for m in meshes), md)
    return nodes, tris


@pytest.mark.parametrize("name", ["cornell", "dragon", "sphere", "single", "soup"])
def test_product_builder_equals_oracle_builder(pkg, oracle, scenes, dragon, name):
    sc = {"cornell": scenes.cornell_box, "dragon": lambda: dragon, "sphere": lambda: scenes.displaced_sphere(60, 60),
          "single": scenes.single_triangle, "soup": lambda: scenes.icosphere_soup(40, 2)}[name]()
    nodes, tris = _compare(pkg, oracle, sc["meshes"])
    if name == "dragon":
        assert len(tris) == 4014


def test_large_scene_parallel_build_is_deterministic(pkg, oracle, scenes):
    """> 32768 triangles takes the OpenMP task path; result must equal the serial oracle build and itself."""
    sc = scenes.heightfield(n=160)
    a = pkg.build_bvh_host(sc["meshes"])
    b = pkg.build_bvh_host(sc["meshes"])
    assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()
    _compare(pkg, oracle, sc["meshes"])


def test_chunk_parallel_sweeps_equal_the_serial_build(pkg, oracle, scenes):
    """Ranges of >= 131072 triangles sweep bounds / bins / the stable partition in parallel chunks (the top levels of the
    tree): same bytes as the oracle's plain recursive build, on a mesh whose triangle order is shuffled so that the
    partition really has to keep it stable."""
    sc = scenes.heightfield(n=330)  # 217 802 triangles
    m = sc["meshes"][0]
    rng = np.random.default_rng(5)
    perm = rng.permutation(len(m["triangles"]))
    meshes = [dict(m, triangles=np.ascontiguousarray(m["triangles"][perm]))] + sc["meshes"][1:]
    nodes, tris, shade, md = pkg.build_bvh_host(meshes)
    O = oracle.OracleScene(meshes)
    assert nodes.tobytes() == O.nodes().tobytes() and tris.tobytes() == O.tris().tobytes() and shade.tobytes() == O.shade().tobytes()
    nodes4, depth4 = pkg.build_bvh4_host(meshes)
    assert nodes4.tobytes() == O.nodes4().tobytes() and depth4 == O.depth4


def test_edge_cases(pkg, oracle):
    f = np.float32
    tri = f([(0, 0, 0), (1, 0, 0), (0, 1, 0)])
    # empty scene, empty mesh
    nodes, tris, shade, md = pkg.build_bvh_host([])
    assert len(nodes) == 0 and len(tris) == 0
    _compare(pkg, oracle, [{"vertices": np.zeros((0, 3), f), "triangles": np.zeros((0, 3), np.uint32)}])
    # scenes of 1..5 triangles (root-is-a-leaf wrapping up to 4)
    for n in range(1, 6):
        v = np.concatenate([tri + f([2 * i, 0, 0]) for i in range(n)])
        t = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
        nodes, _ = _compare(pkg, oracle, [{"vertices": v, "triangles": t}])
        assert len(nodes) >= 1
    # 37 identical triangles: all centroids equal -> no SAH plane exists -> median splits down to leaves of <= 4
    v = np.concatenate([tri] * 37)
    t = np.arange(3 * 37, dtype=np.uint32).reshape(-1, 3)
    _compare(pkg, oracle, [{"vertices": v, "triangles": t}])
    # degenerate (zero-area) and axis-aligned flat triangles
    v = f([(0, 0, 0), (0, 0, 0), (0, 0, 0), (1, 1, 1), (2, 1, 1), (1, 2, 1), (5, 5, 5), (5, 5, 5), (6, 5, 5)])
    _compare(pkg, oracle, [{"vertices": v, "triangles": np.uint32([(0, 1, 2), (3, 4, 5), (6, 7, 8)])}])


def test_bad_input_is_rejected(pkg):
    f = np.float32
    with pytest.raises(pkg.CrtError) as e:
        pkg.build_bvh_host([{"vertices": f([(0, 0, 0), (1, 0, 0), (0, 1, 0)]), "triangles": np.uint32([(0, 1, 3)])}])
    assert "out of range" in str(e.value)


def test_depth_bound_on_adversarial_input(pkg, oracle):
    """Geometric progression of sizes makes SAH peel one triangle per level; the depth guard must cap at 32."""
    f = np.float32
    n = 200
    s = (1.5 ** np.arange(n, dtype=np.float64) * 1e-6).astype(f)[:60]
    v = np.concatenate([f([(x, 0, 0), (x * 1.1, 0, 0), (x, x * 0.1, 0)]) for x in s])
    t = np.arange(3 * len(s), dtype=np.uint32).reshape(-1, 3)
    nodes, tris, shade, md = pkg.build_bvh_host([{"vertices": v, "triangles": t}])
    assert md <= 32
    _compare(pkg, oracle, [{"vertices": v, "triangles": t}])


def test_nan_vertices_do_not_break_the_builders(pkg, oracle):
    """a NaN vertex anywhere in a larger mesh: both builders still agree and nothing indexes out of range"""
    rng = np.random.default_rng(3)
    v = rng.uniform(-5, 5, size=(300, 3)).astype(np.float32)
    t = np.arange(300, dtype=np.uint32).reshape(-1, 3)
    for poison in (0, 150, 299):
        w = v.copy()
        w[poison, poison % 3] = np.nan
        nodes, tris, shade, md = pkg.build_bvh_host([{"vertices": w, "triangles": t}])
        O = oracle.OracleScene([{"vertices": w, "triangles": t}])
        assert nodes.tobytes() == O.nodes().tobytes() and tris.tobytes() == O.tris().tobytes()
        assert sorted(tris["gid"].tolist()) == list(range(100))


def test_random_meshes_both_builders_agree(pkg, oracle):
    """Property test: for arbitrary small meshes (coincident vertices, zero-area and repeated triangles, huge / tiny
    coordinates, several meshes) the product's builder and the oracle's independent implementation emit the same bytes
    (binary tree, wide tree, leaf-ordered records) and the structural invariants hold."""
    from hypothesis import given, settings, strategies as st

    coord = st.one_of(st.floats(-20, 20, width=32), st.sampled_from([0.0, 1.0, -1.0, 1e-20, 1e20, -1e20, 3.4e38]),
                      st.integers(-3, 3).map(float))
    mesh = st.integers(1, 40).flatmap(lambda nv: st.tuples(
        st.lists(st.tuples(coord, coord, coord), min_size=nv, max_size=nv),
        st.lists(st.tuples(*[st.integers(0, nv - 1)] * 3), min_size=0, max_size=60)))

    @settings(max_examples=120, deadline=None)
    @given(st.lists(mesh, min_size=1, max_size=3))
    def run(ms):
        meshes = [{"vertices": np.float32(v).reshape(-1, 3), "triangles": np.uint32(t).reshape(-1, 3), "material_index": i}
                  for i, (v, t) in enumerate(ms)]
        n_tris = sum(len(m["triangles"]) for m in meshes)
        nodes, tris, shade, md = pkg.build_bvh_host(meshes)
        O = oracle.OracleScene(meshes)
        assert nodes.tobytes() == O.nodes().tobytes() and tris.tobytes() == O.tris().tobytes() and shade.tobytes() == O.shade().tobytes()
        nodes4, depth4 = pkg.build_bvh4_host(meshes)
        assert nodes4.tobytes() == O.nodes4().tobytes() and depth4 == O.depth4 and md == O.max_depth
        assert len(tris) == n_tris and sorted(tris["gid"].tolist()) == list(range(n_tris)) and md <= 32
        # the 64-byte quantised nodes: both restatements of the rule agree on every input (huge extents are clamped alike);
        # containment / tightness hold wherever the coordinates stay below the documented 3e38 extent limit
        q = pkg.quantize4(nodes4)
        assert q.tobytes() == O.nodes4q().tobytes()
        if all(np.abs(m["vertices"]).max(initial=0.0) <= 1e30 for m in meshes):
            _check_quantised(nodes4, q)

    run()
