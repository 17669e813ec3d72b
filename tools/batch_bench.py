#!/usr/bin/env python3
"""Frames per launch on ONE stream, launches back to back: ms per frame for batches of 1..4 (C3, 1080p, mode 100)."""


def main():
    import importlib, os, sys, time
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
    import __graft_entry__ as e
    import torch
    pkg = e.load_package(); scenes = importlib.import_module(e.PKG_NAME + ".scenes")
    sc = scenes.heightfield(n_lights=1)
    r = pkg.Renderer(0); r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"]); r.change_shading_mode(100)
    W, H = 1920, 1080
    bufs = [torch.zeros(W * H, dtype=torch.int32, device="cuda") for _ in range(4)]
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    for B in (1, 2, 3, 4):
        ptrs = [b.data_ptr() for b in bufs[:B]]
        for _ in range(24): r.render_frames_batch_device(W, H, ptrs)
        torch.cuda.synchronize()
        K = 120
        t0 = time.perf_counter()
        for _ in range(K): r.render_frames_batch_device(W, H, ptrs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("batch %d: %.4f ms per frame (%.4f ms per launch)" % (B, dt / K / B * 1e3, dt / K * 1e3), flush=True)


if __name__ == "__main__":
    main()
