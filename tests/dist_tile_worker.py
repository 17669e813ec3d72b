"""world_size-2 gloo worker for tests/test_tiling.py: the N>1 data path of bench.py on CPU.
Each rank produces ONLY its own macro tiles (the CPU oracle stands in for the HIP kernel, which cannot run here),
stages them tile-major exactly as crt_render_tiles_device does, one all_gather moves them, the de-interleave
rebuilds the row-major frame, which must equal the frame a single rank renders alone."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, n = dist.get_rank(), dist.get_world_size()
    pkg = entry.load_package()
    import importlib
    scenes = importlib.import_module(entry.PKG_NAME + ".scenes")
    host = importlib.import_module(entry.PKG_NAME + ".multigpu")
    oracle = entry.load_oracle()
    sc = scenes.cornell_box()
    cam = sc["camera"]
    w, h = 200, 120  # not multiples of 16: partial tiles on both edges
    S = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    full = S.render(cam["position"], cam["matrix"], 100, w, h, n_threads=2)["rgba8"].view(np.uint32).reshape(h, w)

    # this rank's tiles only: mask everything it does not own before staging
    tx = (w + 15) // 16
    ys, xs = np.mgrid[0:h, 0:w]
    owner = ((ys // 16) * tx + xs // 16) % n
    mine = np.where(owner == rank, full, 0).astype(np.uint32)
    staging = torch.from_numpy(pkg.tile_host(mine, w, h, rank, n).reshape(-1).view(np.int32).copy())
    frame = host.gather_frame(staging, w, h, untile=lambda g: torch.from_numpy(
        pkg.untile_host(g.numpy().view(np.uint32), w, h, n).view(np.int32).copy()))
    got = frame.numpy().view(np.uint32).reshape(h, w)
    assert np.array_equal(got, full), "gathered frame differs on rank %d" % rank
    # a batch of 3 frames (different images) in one all-gather: [rank][frame][slot] layout, de-interleaved per frame
    fulls = [full, np.ascontiguousarray(full[::-1]), (full ^ np.uint32(0x00FF00FF))]
    stag = np.concatenate([pkg.tile_host(np.where(owner == rank, f, 0).astype(np.uint32), w, h, rank, n).reshape(-1) for f in fulls])
    frames = host.gather_batch(torch.from_numpy(stag.view(np.int32).copy()), w, h, 3, untile=lambda g, f: torch.from_numpy(
        pkg.untile_host(g.numpy().view(np.uint32), w, h, n, n_frames=3, frame=f).view(np.int32).copy()))
    for f in range(3):
        assert np.array_equal(frames[f].numpy().view(np.uint32).reshape(h, w), fulls[f]), "batched gather: frame %d differs on rank %d" % (f, rank)
    share = host.rank_share(w, h, rank, n)
    assert share["tiles"] == len(range(rank, pkg.tile_count(w, h), n))
    dist.barrier()
    if rank == 0:
        print("TILE_GATHER_OK", n)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
