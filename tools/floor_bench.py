#!/usr/bin/env python3
"""Fixed costs of a 1080p frame: empty scene (launch + rayGen + store), tiny scene, and the C3 frame per mode."""


def main():
    import sys, os, importlib, statistics
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
    import numpy as np
    import __graft_entry__ as e
    import torch
    pkg = e.load_package(); scenes = importlib.import_module(e.PKG_NAME + ".scenes")
    W, H = 1920, 1080
    frame = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    def run(name, sc, mode):
        r = pkg.Renderer(0); r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"]); r.change_shading_mode(mode)
        for _ in range(5): r.render_frame_device(W, H, frame.data_ptr(), stats=True)
        ms = [r.render_frame_device(W, H, frame.data_ptr(), stats=True)["kernel_ms"] for _ in range(30)]
        print("%-28s mode %3d  kernel %.4f ms" % (name, mode, statistics.median(ms)), flush=True)
        r.close()
    empty = {"meshes": [], "lights": [], "materials": [], "camera": {"position": np.float32([0, 0, 0]), "matrix": scenes.IDENTITY}}
    run("empty scene", empty, 0)
    run("cornell (32 tris)", scenes.cornell_box(), 100)
    hf = scenes.heightfield(n_lights=1)
    for m in (3, 5, 0, 100):
        run("heightfield 1M", hf, m)
    run("icosphere soup 1M", scenes.icosphere_soup(), 100)


if __name__ == "__main__":
    main()
