#!/usr/bin/env python3
"""Render the C3 frame with two builds of the library and compare the RGBA8 frames: tools/frame_eq.py libA.so libB.so [--mode M]"""


def main():
    import argparse, importlib, os, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    import torch
    ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs=2); ap.add_argument("--mode", type=int, default=100)
    a = ap.parse_args()
    pkg = entry.load_package(); scenes = importlib.import_module(entry.PKG_NAME + ".scenes")
    W, H = 1920, 1080
    frames = []
    for path in a.libs:
        pkg._lib = None; pkg.LIB_PATH = os.path.abspath(path)
        for sc in (scenes.heightfield(n_lights=1), scenes.icosphere_soup()):
            r = pkg.Renderer(0); r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"]); r.change_shading_mode(a.mode)
            f = torch.zeros(W * H, dtype=torch.int32, device="cuda")
            r.set_counting(True); st = r.render_frame_device(W, H, f.data_ptr(), stats=True); r.set_counting(False)
            frames.append((f.cpu(), st["nodes_visited"], st["tris_tested"]))
    n = len(frames) // 2
    for i in range(n):
        A, B = frames[i], frames[n + i]
        print("scene %d: frames equal: %s   nodes %d vs %d (%+.2f %%)  tris %d vs %d (%+.2f %%)" % (i, bool((A[0] == B[0]).all()), A[1], B[1], 100.0 * (B[1] - A[1]) / A[1], A[2], B[2], 100.0 * (B[2] - A[2]) / A[2]))


if __name__ == "__main__":
    main()
