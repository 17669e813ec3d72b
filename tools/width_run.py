#!/usr/bin/env python3
"""Render N frames of the C3 scene with each tree layout of the packed-layout variant build (tools/variant_build.sh packed
"-DCRT_PACKED_LAYOUTS=1"); meant to run under rocprofv3 --pmc: the three layouts are three kernel instantiations, so one run
yields the counters of all of them.  tools/width_run.py [frames]"""


def main():
    import importlib, os, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
    import __graft_entry__ as e
    import torch
    pkg = e.load_package(); scenes = importlib.import_module(e.PKG_NAME + ".scenes")
    pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), "libcrt_hip_packed.so")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    sc = scenes.heightfield(n_lights=1)
    W, H = 1920, 1080
    frame = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    for width in (0, 4, 8):
        r = pkg.Renderer(0)
        r.set_option("bvh_width", width)
        r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"]); r.change_shading_mode(100)
        for _ in range(n):
            r.render_frame_device(W, H, frame.data_ptr(), stats=True)
        r.close()


if __name__ == "__main__":
    main()
