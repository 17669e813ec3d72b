#!/usr/bin/env python3
"""Mode 100 (fused kernel) against the path pipeline configured to do the same work (mode 200, 1 spp, 0 bounces: camera ray +
one shadow ray per lit hit; the jitter aside) on the C3 frame."""


def main():
    import importlib, os, statistics, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
    import __graft_entry__ as e
    import torch
    pkg = e.load_package(); scenes = importlib.import_module(e.PKG_NAME + ".scenes")
    sc = scenes.heightfield(n_lights=1)
    r = pkg.Renderer(0); r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"])
    W, H = 1920, 1080
    frame = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    for mode, spp, b, tile in ((100, 1, 0, 0), (200, 1, 0, 8), (200, 1, 0, 16), (200, 4, 0, 8), (100, 1, 0, 0)):
        r.change_shading_mode(mode); r.set_path_params(spp, b, 1234); r.set_option("path_tile", tile)
        for _ in range(12): r.render_frame_device(W, H, frame.data_ptr(), stats=True)
        ms = statistics.median([r.render_frame_device(W, H, frame.data_ptr(), stats=True)["kernel_ms"] for _ in range(30)])
        print("mode %d spp %d bounces %d path_tile %d: %.4f ms" % (mode, spp, b, tile, ms), flush=True)


if __name__ == "__main__":
    main()
