#!/usr/bin/env python3
"""A/B harness for tuning knobs on the GPU box: interleaved rounds in ONE process (cdna_hip_programming.md rule 24).
usage: tools/sweep.py <option> v1 v2 ... [--scene heightfield|soup] [--mode 100] [--rounds 5]"""


def main():
    import argparse, importlib, os, sys, statistics
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry

    ap = argparse.ArgumentParser()
    ap.add_argument("option")
    ap.add_argument("values", nargs="+", type=int)
    ap.add_argument("--scene", default="heightfield")
    ap.add_argument("--mode", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--frames", type=int, default=20)
    a = ap.parse_args()
    import torch
    pkg = entry.load_package()
    scenes = importlib.import_module(entry.PKG_NAME + ".scenes")
    sc = {"heightfield": lambda: scenes.heightfield(n_lights=1), "heightfield5m": lambda: scenes.heightfield(n=1581, n_lights=1), "soup": scenes.icosphere_soup}[a.scene]()
    r = pkg.Renderer(0)
    r.upload(sc["meshes"], sc["lights"], sc["materials"])
    r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"])
    r.change_shading_mode(a.mode)
    W, H = 1920, 1080
    frame = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    res = {v: [] for v in a.values}
    for rnd in range(a.rounds + 1):
        for v in a.values:
            r.set_option(a.option, v); r.render_frame_device(W, H, frame.data_ptr(), stats=True)
            ms = [r.render_frame_device(W, H, frame.data_ptr(), stats=True)["kernel_ms"] for _ in range(a.frames)]
            if rnd:
                res[v].append(statistics.median(ms))
    for v in a.values:
        print("%s=%d: median %.4f ms  min %.4f ms" % (a.option, v, statistics.median(res[v]), min(res[v])), flush=True)


if __name__ == "__main__":
    main()
