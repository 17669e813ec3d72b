#!/usr/bin/env python3
"""C3 with a camera that changes every frame (orbit step 0.01 degrees: practically the same view, so the difference to the
static camera is the cost of the per-frame feedback): the launch order is then measured and sorted again for
every frame (side stream), as an interactive viewer would see it.  ms per frame, launches back to back on one stream."""


def main():
    import importlib, os, sys, time
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
    import numpy as np
    import __graft_entry__ as e
    import torch
    pkg = e.load_package(); scenes = importlib.import_module(e.PKG_NAME + ".scenes")
    diag = os.path.join(os.path.dirname(pkg.LIB_PATH), "libcrt_hip_diag.so")  # tools/diag_build.sh: needed for the forced-measure case
    have_diag = os.path.exists(diag)
    if have_diag: pkg.LIB_PATH = diag
    sc = scenes.heightfield(n_lights=1)
    r = pkg.Renderer(0); r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.change_shading_mode(100)
    W, H = 1920, 1080
    frame = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    pos0 = np.float32(sc["camera"]["position"])
    def cam(i):
        a = np.radians(0.01 * i)
        c, s = np.cos(a), np.sin(a)
        R = np.float32([[c, 0, s], [0, 1, 0], [-s, 0, c]])
        return (R @ pos0).astype(np.float32), (R @ np.float32(sc["camera"]["matrix"]).reshape(3, 3)).astype(np.float32).reshape(9)
    wobble = "--wobble" in sys.argv  # the camera swings +-0.005 degrees about the static view instead of travelling along an orbit: same picture
    if wobble:
        orbit = cam
        cam = lambda i: orbit(130 + 0.5 * (1 if i % 2 else -1))
    for moving, every in ((False, 1), (False, -1), (True, 1), (True, 2), (True, 8), (True, 32), (False, 1)):
        if every < 0 and not have_diag: continue
        if have_diag: r.set_option("debug_force_measure", 1 if every < 0 else 0)  # -1: static view, but measured and sorted at every frame
        r.set_option("remeasure_every", abs(every))
        r.set_camera(*cam(130))  # the static reference view = the middle of the orbit segment the moving runs cover (30..230)
        for i in range(30):
            if moving: r.set_camera(*cam(i))
            r.render_frame_device(W, H, frame.data_ptr())
        torch.cuda.synchronize()
        K = 200
        cams = [cam(30 + i) for i in range(K)]  # precomputed: the timed loop issues only the two C calls per frame
        t0 = time.perf_counter()
        for i in range(K):
            if moving: r.set_camera(*cams[i])
            r.render_frame_device(W, H, frame.data_ptr())
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        print("   (host issue %.1f us per frame)" % (t_issue / K * 1e6))
        print("%s camera, remeasure_every %d: %.4f ms per frame" % ("moving" if moving else "static", every, (time.perf_counter() - t0) / K * 1e3), flush=True)


if __name__ == "__main__":
    main()
