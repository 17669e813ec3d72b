import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as e
import torch
pkg = e.load_package(); scenes = importlib.import_module(e.PKG_NAME + ".scenes")
sc = scenes.load_crtscene(os.path.join(ROOT, "tests/golden/dragon.crtscene"))
for path in sys.argv[1:]:
    pkg._lib = None; pkg.LIB_PATH = os.path.abspath(path); pkg.lib()
    r = pkg.Renderer(0); r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"]); r.change_shading_mode(100)
    for (w, h) in ((333, 77),):
        full = torch.zeros(h * w, dtype=torch.int32, device="cuda")
        r.render_frame_device(w, h, full.data_ptr()); r.synchronize()
        for rep in range(3):
          for n in (1, 2, 3, 8):
            slots = pkg.tile_slots(w, h, n)
            gathered = torch.zeros(n * slots * 256, dtype=torch.int32, device="cuda")
            for rank in range(n):
                r.render_tiles_device(w, h, rank, n, gathered.data_ptr() + rank * slots * 1024, stats=True)
            frame = torch.zeros(h * w, dtype=torch.int32, device="cuda")
            r.untile_device(w, h, n, gathered.data_ptr(), frame.data_ptr()); r.synchronize()
            d = (frame != full).cpu().numpy().reshape(h, w)
            if d.any():
                ys, xs = np.nonzero(d)
                print(os.path.basename(path), "rep", rep, "n", n, "differ", d.sum(), "x", xs.min(), xs.max(), "y", ys.min(), ys.max(), "tiles", sorted(set((int(y) // 16) * ((w + 15) // 16) + int(x) // 16 for y, x in zip(ys, xs)))[:10], flush=True)
            else:
                print(os.path.basename(path), "rep", rep, "n", n, "ok", flush=True)
    r.close()
