#!/usr/bin/env python3
"""Run the CRT_PROF diagnostic build (libcrt_hip_prof.so): share of a wavefront's cycles in node steps / leaf steps / rest."""


def main():
    import sys, os, importlib
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
    import numpy as np
    import __graft_entry__ as e
    import torch
    pkg = e.load_package(); scenes = importlib.import_module(e.PKG_NAME + ".scenes")
    pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), "libcrt_hip_prof.so")
    sc = scenes.heightfield(n_lights=1)
    r = pkg.Renderer(0); r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"])
    W, H = 1920, 1080
    frame = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    if "--path" in sys.argv:  # mode 200: the three stages of the path kernel; name=value arguments are options (path_pipeline=1 ...)
        big = "--c5" in sys.argv
        if big:
            sc = scenes.heightfield(n=1581, n_lights=1); W, H = 3840, 2160
            r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"])
            frame = torch.zeros(W * H, dtype=torch.int32, device="cuda")
        for o in [x for x in sys.argv[1:] if "=" in x]: r.set_option(o.split("=")[0], int(o.split("=")[1]))
        r.change_shading_mode(pkg.MODE_PATH); r.set_path_params(4, 3, 1234)
        r.set_counting(True)  # (the instrumented variant measures the costs the launch order is sorted by, too)
        for _ in range(12): r.render_frame_device(W, H, frame.data_ptr(), stats=True)  # the launch order settles
        r.set_counting(True)
        st = r.render_frame_device(W, H, frame.data_ptr(), stats=True)
        c = r.read_counters().astype(np.float64)
        r.set_counting(False)
        print("mode 200 %dx%d: kernel %.3f ms (instrumented); rays closest %d shadow %d" % (W, H, st["kernel_ms"], c[3], c[2]))
        life = c[4]
        if c[27] > 0:  # how full the launch keeps its wavefront slots: sum of the wavefronts' lives / (wavefronts x launch duration)
            print("  %d persistent wavefronts alive %.1f%% of the launch on average (100 MHz clock: %.3f ms of %.3f ms)" % (
                c[27], 100.0 * c[26] / c[27] / (st["kernel_ms"] * 1e5), c[26] / c[27] / 1e5, st["kernel_ms"]))
        for i, name in enumerate(("A camera rays", "B shade + shadow rays" , "C bounce rays")):
            t, tn, tl, itn, itl, lan, lal = c[5 + 7 * i: 12 + 7 * i]
            if t == 0: continue
            print("  stage %-22s %5.1f%% of the wavefronts' life: node steps %4.1f%%  leaf steps %4.1f%%  rest %4.1f%%" % (name, 100 * t / life, 100 * tn / t, 100 * tl / t, 100 * (t - tn - tl) / t))
            if itn and itl:
                print("      node phases %d (%.0f cycles, %.1f lanes)   leaf phases %d (%.0f cycles, %.1f lanes)" % (itn, tn / itn, lan / itn, itl, tl / itl, lal / itl))
        return
    for mode in (3, 100):
        r.change_shading_mode(mode)
        for _ in range(4): r.render_frame_device(W, H, frame.data_ptr(), stats=True)
        r.set_counting(True)
        st = r.render_frame_device(W, H, frame.data_ptr(), stats=True)
        c = r.read_counters().astype(np.float64)
        r.set_counting(False)
        tot, tn, tl, itn, itl, lan, lal = c[4:11]
        print("mode %d: kernel %.3f ms (instrumented)  wave-cycles: node %.1f%%  leaf %.1f%%  other %.1f%%" % (mode, st["kernel_ms"], 100 * tn / tot, 100 * tl / tot, 100 * (tot - tn - tl) / tot))
        print("   node phases: %d  (%.0f cycles each, %.1f lanes active)   leaf phases: %d (%.0f cycles each, %.1f lanes active)" % (itn, tn / itn, lan / itn, itl, tl / itl, lal / itl))
        d = c[11:19]
        if d[0] > 0:
            print("   divergent node steps %d: %.1f lanes, %.1f runs of equal neighbours, %.1f distinct nodes each" % (d[0], d[1] / d[0], d[2] / d[0], d[3] / d[0]))
        if d[4] > 0:
            print("   divergent leaf steps %d: %.1f lanes, %.1f runs, %.1f distinct leaves each" % (d[4], d[5] / d[4], d[6] / d[4], d[7] / d[4]))
        print("   node fetches %d  tri fetches %d -> per node phase %.1f lane-steps, per leaf phase %.1f tri tests" % (c[0], c[1], c[0] / itn, c[1] / itl))


if __name__ == "__main__":
    main()
