#!/usr/bin/env python3
"""Path tracing (mode 200) throughput: BASELINE.json configs[4] (C5: 4 999 124 triangles, 3840x2160, 4 spp, 3 bounces)
and the same settings on the 1M-triangle C3 scene.  tools/path_bench.py [--quick]"""


def main():
    import importlib, os, statistics, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
    import __graft_entry__ as e
    import torch
    pkg = e.load_package(); scenes = importlib.import_module(e.PKG_NAME + ".scenes")
    if os.environ.get("CRT_LIB"): pkg.LIB_PATH = os.path.abspath(os.environ["CRT_LIB"])  # a variant build
    r = pkg.Renderer(0)
    for o in [x for x in sys.argv[1:] if "=" in x]: r.set_option(o.split("=")[0], int(o.split("=")[1]))
    cases = [("C3 1M tris 1080p", lambda: scenes.heightfield(n_lights=1), 1920, 1080), ("C5 5M tris 4K", lambda: scenes.heightfield(n=1581, n_lights=1), 3840, 2160)]
    if "--quick" in sys.argv: cases = cases[:1]
    for name, mk, W, H in cases:
        sc = mk()
        r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"])
        r.change_shading_mode(pkg.MODE_PATH); r.set_path_params(4, 3, 1234)
        frame = torch.zeros(W * H, dtype=torch.int32, device="cuda")
        r.set_counting(True); c = r.render_frame_device(W, H, frame.data_ptr(), stats=True); r.set_counting(False)
        rays = c["rays_primary"] + c["rays_shadow"]
        for tile in ([0] if "--sweep" not in sys.argv else [8, 16]):
            for im in ([32] if "--sweep" not in sys.argv else [16, 32, 48]):
                if "--sweep" in sys.argv: r.set_option("path_tile", tile); r.set_option("inner_min", im)
                for _ in range(2): r.render_frame_device(W, H, frame.data_ptr(), stats=True)
                ms = statistics.median([r.render_frame_device(W, H, frame.data_ptr(), stats=True)["kernel_ms"] for _ in range(7)])
                print("%s mode 200 4 spp 3 bounces (path_tile %d, inner_min %d): %.3f ms/frame  rays %d (closest %d, shadow %d)  %.0f Mray/s  nodes/ray %.1f tris/ray %.1f" % (
                    name, tile, im, ms, rays, c["rays_primary"], c["rays_shadow"], rays / ms / 1e3, c["nodes_visited"] / rays, c["tris_tested"] / rays), flush=True)
        if "--sweep" in sys.argv: r.set_option("path_tile", 0)


if __name__ == "__main__":
    main()
