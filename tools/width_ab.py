#!/usr/bin/env python3
"""Tree layouts A/B in ONE process: the same scene uploaded once per layout (option "bvh_width": 0 = legacy 64-byte 4-wide
nodes, 4 / 8 = packed wide tree), frames compared (hit ids, t and RGBA8 never depend on the tree), interleaved timing rounds,
fetch counters:  tools/width_ab.py [--scene heightfield] [--mode 100] [--widths 0,4,8] [--lib other.so] [--gpu-build]"""


def main():
    import argparse, importlib, os, statistics, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="heightfield", help="heightfield (C3) | heightfield5m | soup | cornell | dragon")
    ap.add_argument("--mode", type=int, default=100)
    ap.add_argument("--widths", default="0,4,8")
    ap.add_argument("--lib", action="append", help="another build of the library, timed beside the product one")
    ap.add_argument("--lib-widths", default="-1", help="layouts to time with each --lib (-1: its default, option not set)")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--gpu-build", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="name=value set on every renderer")
    a = ap.parse_args()
    import numpy as np
    import torch
    pkg = entry.load_package()
    scenes = importlib.import_module(entry.PKG_NAME + ".scenes")
    sc = {"heightfield": lambda: scenes.heightfield(n_lights=1), "heightfield5m": lambda: scenes.heightfield(n=1581, n_lights=1),
          "soup": scenes.icosphere_soup, "cornell": scenes.cornell_box, "mesh70k": scenes.displaced_sphere}[a.scene]()
    W, H = [int(v) for v in a.size.split("x")]
    frame = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    product = pkg.lib()
    rs = []
    for wd in [int(v) for v in a.widths.split(",")]:
        r = pkg.Renderer(0)
        r.set_option("bvh_width", wd)
        if a.gpu_build:
            r.set_option("gpu_build", 1)
        for o in a.opt:
            r.set_option(o.split("=")[0], int(o.split("=")[1]))
        r.upload(sc["meshes"], sc["lights"], sc["materials"])
        r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"])
        r.change_shading_mode(a.mode)
        rs.append(("width %d" % wd, product, r))
    for path in a.lib or []:
        pkg._lib = None
        pkg.LIB_PATH = os.path.abspath(path)
        other = pkg.lib()
        for wd in [int(v) for v in a.lib_widths.split(",")]:
            r = pkg.Renderer(0)
            if wd >= 0:
                r.set_option("bvh_width", wd)
            for o in a.opt:
                r.set_option(o.split("=")[0], int(o.split("=")[1]))
            r.upload(sc["meshes"], sc["lights"], sc["materials"])
            r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"])
            r.change_shading_mode(a.mode)
            rs.append(("%s w%d" % (os.path.basename(path).replace("libcrt_hip_", "").replace(".so", ""), wd), other, r))
    # correctness first: every layout renders the same frame
    ref = None
    for name, L, r in rs:
        pkg._lib = L
        r.set_counting(True)
        got = r.render_frame(W, H)
        r.set_counting(False)
        st = got["stats"]
        print("%-24s nodes %d tris %d shadow %d  upload %.1f ms" % (name, st["nodes_visited"], st["tris_tested"], st["rays_shadow"], r.build_stats()["upload_ms"]), flush=True)
        if ref is None:
            ref = got
        else:
            for k in ("hit_inst", "hit_prim", "hit_t", "rgba8", "rgb"):
                same = np.array_equal(got[k], ref[k])
                if not same:
                    print("   %s DIFFERS from the first layout (%d elements)" % (k, int(np.count_nonzero(got[k] != ref[k]))), flush=True)
    res = {name: [] for name, _, _ in rs}
    for rnd in range(a.rounds + 1):
        for name, L, r in rs:
            pkg._lib = L
            if rnd == 0:
                for _ in range(12):
                    r.render_frame_device(W, H, frame.data_ptr(), stats=True)  # the launch order settles
            ms = [r.render_frame_device(W, H, frame.data_ptr(), stats=True)["kernel_ms"] for _ in range(a.frames)]
            if rnd:
                res[name].append(statistics.median(ms))
    for name in res:
        print("%-24s median %.4f ms  min %.4f ms" % (name, statistics.median(res[name]), min(res[name])), flush=True)


if __name__ == "__main__":
    main()
    import os
    os._exit(0)  # (unloading several builds of the library in one process can crash at exit)
