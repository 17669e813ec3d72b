#!/usr/bin/env python3
"""configs[4] at full size (5M triangles, 3840x2160, 4 spp, 3 bounces): the persistent path kernel and the stage-launch pipeline must give the same frame and the same counters."""
import importlib, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
import torch
pkg = e.load_package(); scenes = importlib.import_module(e.PKG_NAME + ".scenes")
sc = scenes.heightfield(n=1581, n_lights=1); W, H = 3840, 2160
r = pkg.Renderer(0); r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"])
r.change_shading_mode(pkg.MODE_PATH); r.set_path_params(4, 3, 1234)
out = []
for pipe in (0, 1):
    r.set_option("path_pipeline", pipe)
    f = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    r.set_counting(True); st = r.render_frame_device(W, H, f.data_ptr(), stats=True); r.set_counting(False)
    torch.cuda.synchronize()
    out.append((f.clone(), st))
print("frames equal:", bool(torch.equal(out[0][0], out[1][0])), {k: (out[0][1][k], out[1][1][k]) for k in ("rays_primary", "rays_shadow", "nodes_visited", "tris_tested")})
