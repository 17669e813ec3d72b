#!/usr/bin/env python3
"""Acceleration-structure build: host binned-SAH vs GPU LBVH (option gpu_build) -- build time and the frame time over
each tree (1080p, mode 100)."""


def main():
    import sys, os, importlib, statistics, time
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
    import __graft_entry__ as e
    import torch
    pkg = e.load_package(); scenes = importlib.import_module(e.PKG_NAME + ".scenes")
    W, H = 1920, 1080
    frame = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    for name, sc in (("heightfield 1M", scenes.heightfield(n_lights=1)), ("heightfield 5M", scenes.heightfield(n=1581, n_lights=1)), ("icosphere soup 1M", scenes.icosphere_soup())):
        for gpu in (0, 1):
            r = pkg.Renderer(0)
            r.set_option("gpu_build", gpu)
            ups = []
            for _ in range(3):
                t0 = time.perf_counter(); r.upload(sc["meshes"], sc["lights"], sc["materials"]); ups.append((time.perf_counter() - t0) * 1e3)
            st = r.build_stats()
            r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"]); r.change_shading_mode(100)
            for _ in range(6): r.render_frame_device(W, H, frame.data_ptr(), stats=True)
            ms = statistics.median(r.render_frame_device(W, H, frame.data_ptr(), stats=True)["kernel_ms"] for _ in range(20))
            info = r.bvh_info()
            print("%-18s %-9s upload (build+collapse+H2D) %8.1f ms  [device build kernels %6.2f ms]  binary nodes %8d depth %2d  -> frame %.3f ms"
                  % (name, "GPU LBVH" if gpu else "host SAH", min(ups), st["device_build_ms"], info["n_nodes"], info["max_depth"], ms), flush=True)
            r.close()


if __name__ == "__main__":
    main()
