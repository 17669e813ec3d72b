#!/usr/bin/env python3
"""A/B two builds of libcrt_hip.so in ONE process (interleaved rounds): tools/ab_lib.py libA.so libB.so [--mode 100]"""


def main():
    import argparse, importlib, os, statistics, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs="+"); ap.add_argument("--mode", type=int, default=100)
    ap.add_argument("--opt", default=None, help="name=v1,v2,...: also sweep a crt_set_option knob for every library")
    ap.add_argument("--scene", default="heightfield", help="heightfield (C3) | heightfield5m | soup")
    ap.add_argument("--rounds", type=int, default=6); ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--set", action="append", default=[], help="name=value set once on every renderer")
    a = ap.parse_args()
    import torch
    pkg = entry.load_package(); scenes = importlib.import_module(entry.PKG_NAME + ".scenes")
    sc = {"heightfield": lambda: scenes.heightfield(n_lights=1), "heightfield5m": lambda: scenes.heightfield(n=1581, n_lights=1),
          "soup": scenes.icosphere_soup}[a.scene]()
    W, H = 1920, 1080
    frame = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    rs = []
    for path in a.libs:
        pkg._lib = None; pkg.LIB_PATH = os.path.abspath(path)
        L = pkg.lib()
        r = pkg.Renderer(0); r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"]); r.change_shading_mode(a.mode)
        for o in a.set: r.set_option(o.split("=")[0], int(o.split("=")[1]))
        rs.append((path, L, r))
    oname, ovals = (a.opt.split("=")[0], [int(v) for v in a.opt.split("=")[1].split(",")]) if a.opt else (None, [None])
    res = {(p, v): [] for p, _, _ in rs for v in ovals}
    for rnd in range(a.rounds + 1):
        for path, L, r in rs:
            pkg._lib = L
            for v in ovals:
                if oname:
                    r.set_option(oname, v)
                    for _ in range(12): r.render_frame_device(W, H, frame.data_ptr(), stats=True)  # the launch order settles again
                ms = [r.render_frame_device(W, H, frame.data_ptr(), stats=True)["kernel_ms"] for _ in range(a.frames)]
                if rnd: res[(path, v)].append(statistics.median(ms))
    for (path, v) in res:
        print("%-32s %-18s median %.4f ms  min %.4f ms" % (os.path.basename(path), "" if v is None else "%s=%d" % (oname, v), statistics.median(res[(path, v)]), min(res[(path, v)])), flush=True)


if __name__ == "__main__":
    main()
