#!/usr/bin/env python3
"""Wave-lifetime timeline of ONE rank's tile share of an N-rank frame (diagnostic build, tools/diag_build.sh): what bounds a
lone launch.  tools/timeline_share.py [mode] [N] [split_units]"""


def main():
    import sys, os, importlib
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import numpy as np
    import __graft_entry__ as e
    import torch
    pkg = e.load_package(); scenes = importlib.import_module(e.PKG_NAME + ".scenes"); host = importlib.import_module(e.PKG_NAME + ".multigpu")
    pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), "libcrt_hip_diag.so")
    mode = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    split = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    sc = scenes.heightfield(n_lights=1)
    r = pkg.Renderer(0); r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"])
    r.change_shading_mode(mode)
    r.set_option("split_units", split)
    if len(sys.argv) > 4: r.set_option("split_rays", int(sys.argv[4]))
    if len(sys.argv) > 5: r.set_option("split_segments", int(sys.argv[5]))
    W, H = 1920, 1080
    share = host.rank_share(W, H, 0, N)
    staging = torch.zeros(share["slots"] * 256, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(14):
        r.render_tiles_device(W, H, 0, N, staging.data_ptr(), stats=True)
    plain = r.render_tiles_device(W, H, 0, N, staging.data_ptr(), stats=True)["kernel_ms"]
    r.set_option("timeline", 1)
    st = r.render_tiles_device(W, H, 0, N, staging.data_ptr(), stats=True)
    tl = r.read_timeline()
    idx = np.nonzero(tl[:, 1] > 0)[0]
    tl = tl[tl[:, 1] > 0]
    t0 = tl[:, 0].min()
    s = (tl[:, 0] - t0).astype(np.float64) / 100.0   # us
    en = (tl[:, 1] - t0).astype(np.float64) / 100.0
    life = en - s
    print("mode %d N=%d split_units=%d: kernel %.1f us (with stamps %.1f us), waves %d, span %.1f us" % (mode, N, split, plain * 1e3, st["kernel_ms"] * 1e3, len(tl), en.max()))
    print("  lifetime us: mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f;  last start %.1f us;  sum of lifetimes / span = %.0f resident waves" % (
        life.mean(), np.median(life), np.percentile(life, 90), np.percentile(life, 99), life.max(), s.max(), life.sum() / en.max()))
    parts = (64 // (int(sys.argv[4]) if len(sys.argv) > 4 else 4)) if split else 1
    is_split = idx < split * parts
    if split:
        print("  split wavefronts: %d, lifetime mean %.1f max %.1f, last end %.1f us;  ordinary: lifetime mean %.1f max %.1f, last end %.1f us" % (
            int(is_split.sum()), life[is_split].mean(), life[is_split].max(), en[is_split].max(), life[~is_split].mean(), life[~is_split].max(), en[~is_split].max()))
    for k in np.argsort(en)[-10:]:
        print("     b=%6d start %6.1f life %6.1f end %6.1f tile (%d,%d)" % (idx[k], s[k], life[k], en[k], int(tl[k, 2]) & 0xFFFF, (int(tl[k, 2]) >> 16) & 0xFFFF))
    edges = np.linspace(0, en.max(), 11)
    print("  resident waves over time:", [int(((s <= 0.5 * (a + b)) & (en > 0.5 * (a + b))).sum()) for a, b in zip(edges[:-1], edges[1:])])


if __name__ == "__main__":
    main()
