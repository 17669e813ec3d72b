#!/usr/bin/env python3
"""A grid of option values, median kernel time each: tools/path_sweep.py nameA=v1,v2,.. nameB=v1,v2,.. [--c3] [--mode M]
Default: mode 200 on C5 (5M triangles, 4K, 4 spp, 3 bounces); --c3: the 1M-triangle scene at 1080p; --mode 100 / 3: the shaded / primary-ray frame."""


def main():
    import importlib, itertools, os, statistics, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
    import __graft_entry__ as e
    import torch
    pkg = e.load_package(); scenes = importlib.import_module(e.PKG_NAME + ".scenes")
    if os.environ.get("CRT_LIB"): pkg.LIB_PATH = os.path.abspath(os.environ["CRT_LIB"])
    small = "--c3" in sys.argv
    sc = scenes.heightfield(n_lights=1) if small else scenes.heightfield(n=1581, n_lights=1)
    W, H = (1920, 1080) if small else (3840, 2160)
    r = pkg.Renderer(0)
    r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"])
    mode = int(sys.argv[sys.argv.index("--mode") + 1]) if "--mode" in sys.argv else pkg.MODE_PATH
    r.change_shading_mode(mode); r.set_path_params(4, 3, 1234)
    frame = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    grid = [(a.split("=")[0], [int(v) for v in a.split("=")[1].split(",")]) for a in sys.argv[1:] if "=" in a]
    names = [g[0] for g in grid]
    for combo in itertools.product(*[g[1] for g in grid]):
        for n, v in zip(names, combo): r.set_option(n, v)
        for _ in range(2 if mode >= 200 else 12): r.render_frame_device(W, H, frame.data_ptr(), stats=True)  # (below 200 the launch order settles first)
        ms = statistics.median([r.render_frame_device(W, H, frame.data_ptr(), stats=True)["kernel_ms"] for _ in range(5 if mode >= 200 else 25)])
        print("  ".join("%s=%d" % nv for nv in zip(names, combo)), "-> %.4f ms" % ms, flush=True)


if __name__ == "__main__":
    main()
