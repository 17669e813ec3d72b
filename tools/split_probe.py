#!/usr/bin/env python3
"""Split packets (option split_units): frame identical to the unsplit one, and what a lone launch of an N-rank share costs.
tools/split_probe.py [--mode 100]"""


def main():
    import argparse, importlib, os, statistics, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
    import __graft_entry__ as e
    import numpy as np, torch
    ap = argparse.ArgumentParser(); ap.add_argument("--mode", type=int, default=100); ap.add_argument("--splits", default="0,64,128,256,512,1024")
    ap.add_argument("--rays", default="4", help="rays per wavefront of a split packet: 16, 8 or 4 (comma separated to sweep)")
    ap.add_argument("--ranks", default="1,8")
    ap.add_argument("--set", action="append", default=[], help="name=value set once on the renderer")
    ap.add_argument("--segments", default="16", help="pieces per split ray: 4, 8 or 16 (comma separated to sweep)")
    a = ap.parse_args()
    pkg = e.load_package(); scenes = importlib.import_module(e.PKG_NAME + ".scenes"); host = importlib.import_module(e.PKG_NAME + ".multigpu")
    if os.environ.get("CRT_LIB"): pkg.LIB_PATH = os.path.abspath(os.environ["CRT_LIB"])
    sc = scenes.heightfield(n_lights=1)
    r = pkg.Renderer(0); r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"]); r.change_shading_mode(a.mode)
    for o in a.set: r.set_option(o.split("=")[0], int(o.split("=")[1]))
    W, H = 1920, 1080
    ref = None
    for N in [int(v) for v in a.ranks.split(",")]:
        share = host.rank_share(W, H, 0, N)
        staging = torch.zeros(share["slots"] * 256, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for split, rays, segs in [(int(v), int(w), int(x)) for v in a.splits.split(",") for w in (a.rays.split(",") if int(v) else a.rays.split(",")[:1])
                                  for x in (a.segments.split(",") if int(v) else a.segments.split(",")[:1])]:
            r.set_option("split_units", split)
            r.set_option("split_rays", rays)
            r.set_option("split_segments", segs)
            for _ in range(14): r.render_tiles_device(W, H, 0, N, staging.data_ptr(), stats=True)
            ms = statistics.median([r.render_tiles_device(W, H, 0, N, staging.data_ptr(), stats=True)["kernel_ms"] for _ in range(25)])
            img = staging.cpu().numpy().copy()
            if split == 0: ref = img
            print("N=%d split_units=%4d x %2d rays x %2d segments: one launch alone %.1f us   staging %s" % (N, split, rays, segs, ms * 1e3, "identical" if np.array_equal(img, ref) else "DIFFERS (%d)" % int((img != ref).sum())), flush=True)
    r.set_option("split_units", -1)


if __name__ == "__main__":
    main()
