#!/usr/bin/env python3
"""Wave-lifetime timeline of one frame (diagnostic build of the kernel): occupancy over time, lifetime distribution."""


def main():
    import sys, os, importlib
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import numpy as np
    import __graft_entry__ as e
    import torch
    pkg = e.load_package(); scenes = importlib.import_module(e.PKG_NAME + ".scenes")
    pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), "libcrt_hip_diag.so")  # tools/diag_build.sh
    sc = scenes.heightfield(n_lights=1)
    r = pkg.Renderer(0); r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"])
    r.change_shading_mode(int(sys.argv[1]) if len(sys.argv) > 1 else 100)
    W, H = 1920, 1080
    frame = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    if os.environ.get("CRT_AFFINE"): r.set_option("xcd_affine_order", int(os.environ["CRT_AFFINE"]))
    for _ in range(14):
        r.render_frame_device(W, H, frame.data_ptr(), stats=True)
    r.set_option("timeline", 1)
    st = r.render_frame_device(W, H, frame.data_ptr(), stats=True)
    tl = r.read_timeline()
    tl = tl[tl[:, 1] > 0]
    t0 = tl[:, 0].min()
    s = (tl[:, 0] - t0).astype(np.float64) / 100.0   # us
    en = (tl[:, 1] - t0).astype(np.float64) / 100.0
    life = en - s
    xcc = (tl[:, 2] >> 32).astype(np.int64)
    print("waves", len(tl), "kernel_ms(counting variant)", st["kernel_ms"], "span_us", en.max())
    print("lifetime us: mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % (life.mean(), np.median(life), np.percentile(life, 90), np.percentile(life, 99), life.max()))
    print("sum of lifetimes / span = avg resident waves: %.0f  (per CU %.1f)" % (life.sum() / en.max(), life.sum() / en.max() / 256))
    edges = np.linspace(0, en.max(), 21)
    for a, b in zip(edges[:-1], edges[1:]):
        mid = 0.5 * (a + b)
        res = ((s <= mid) & (en > mid)).sum()
        print("t=%7.1f us resident waves %5d  started in bin %5d" % (mid, res, ((s >= a) & (s < b)).sum()))
    print("per-XCC waves:", np.bincount(xcc, minlength=8), " per-XCC last end (us):", [round(float(en[xcc == x].max()), 1) if (xcc == x).any() else None for x in range(8)])
    idx = np.nonzero(r.read_timeline()[:, 1] > 0)[0]
    last = np.argsort(en)[-12:]
    print("latest-ending waves: (blockIdx, start_us, life_us, xcc, tile)")
    for k in last:
        print("   b=%6d start %7.1f life %7.1f xcc %d tile (%d,%d)" % (idx[k], s[k], life[k], xcc[k], int(tl[k, 2]) & 0xFFFF, (int(tl[k, 2]) >> 16) & 0xFFFF))
    firsts = np.argsort(idx)[:6400]
    print("first 6400 blockIdx: start time max %.1f us; lifetimes mean %.1f max %.1f" % (s[firsts].max(), life[firsts].mean(), life[firsts].max()))
    print("start time vs blockIdx correlation:", np.corrcoef(idx, s)[0, 1])


if __name__ == "__main__":
    main()
