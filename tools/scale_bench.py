#!/usr/bin/env python3
"""Resolution scaling of the C3 view: kernel ms and Mray/s at 540p .. 8K, modes 100 and 3."""


def main():
    import sys, os, importlib, statistics
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
    import __graft_entry__ as e
    import torch
    pkg=e.load_package(); scenes=importlib.import_module(e.PKG_NAME + ".scenes")
    sc=scenes.heightfield(n_lights=1); r=pkg.Renderer(0); r.upload(sc['meshes'],sc['lights'],sc['materials']); r.set_camera(sc['camera']['position'],sc['camera']['matrix'])
    for mode in (100,3):
        r.change_shading_mode(mode)
        for (W,H) in ((960,540),(1920,1080),(3840,2160),(7680,4320)):
            frame=torch.zeros(W*H,dtype=torch.int32,device='cuda')
            r.set_counting(True); c=r.render_frame_device(W,H,frame.data_ptr(),stats=True); r.set_counting(False)
            rays=c['rays_primary']+c['rays_shadow']
            ms=[r.render_frame_device(W,H,frame.data_ptr(),stats=True)['kernel_ms'] for _ in range(15)]
            m=statistics.median(ms)
            print('mode',mode,W,H,'ms %.4f'%m,'Mray/s %.0f'%(rays/m/1e3),'nodes/ray %.1f'%(c['nodes_visited']/rays), flush=True)


if __name__ == "__main__":
    main()
