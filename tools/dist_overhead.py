#!/usr/bin/env python3
"""Per-frame cost of the N>1 step on ONE GPU: rank 0's share of an N-rank frame (1/N of the tiles) + a one-rank RCCL
all-gather of the same message size + the de-interleave, timed as (a) CPU issue time per step and (b) steps/s with
4 frames in flight.  Shows whether the N = 8 step is bounded by the GPU or by host-side launch overhead."""


def main():
    import importlib, os, sys, time
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
    import __graft_entry__ as e
    import torch, torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    pkg = e.load_package(); scenes = importlib.import_module(e.PKG_NAME + ".scenes"); host = importlib.import_module(e.PKG_NAME + ".multigpu")
    sc = scenes.heightfield(n_lights=1)
    r = pkg.Renderer(0); r.upload(sc["meshes"], sc["lights"], sc["materials"]); r.set_camera(sc["camera"]["position"], sc["camera"]["matrix"]); r.change_shading_mode(100)
    W, H = 1920, 1080
    n_fly = int(os.environ.get('NFLY', '4'))
    BATCH = int(os.environ.get('BATCH', '1'))
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(n_fly - 1)]
    import statistics
    for N, adaptive in ((1, 2), (2, 2), (4, 2), (8, 2)):
        r.set_option("adaptive_order", adaptive)
        share = host.rank_share(W, H, 0, N)
        PAD = 64  # room for experimental ownership layouts
        staging = [torch.zeros((share["slots"] + PAD) * 256, dtype=torch.int32, device="cuda") for _ in range(n_fly)]
        gathered = [torch.zeros(N * (share["slots"] + PAD) * 256, dtype=torch.int32, device="cuda") for _ in range(n_fly)]
        frames = [torch.zeros(W * H, dtype=torch.int32, device="cuda") for _ in range(n_fly)]
        r.set_stream(streams[0].cuda_stream)
        for _ in range(5): r.render_tiles_device(W, H, 0, N, staging[0].data_ptr(), stats=True)
        ms = statistics.median([r.render_tiles_device(W, H, 0, N, staging[0].data_ptr(), stats=True)["kernel_ms"] for _ in range(15)])
        print("N=%d adaptive_order=%d: one launch alone %.1f us" % (N, adaptive, ms * 1e3), flush=True)
        for what in ("render", "render+gather+untile"):
            def step(i):
                k = i % n_fly
                r.set_stream(streams[k].cuda_stream)
                with torch.cuda.stream(streams[k]):
                    if BATCH == 1:
                        r.render_tiles_device(W, H, 0, N, staging[k].data_ptr())
                    else:
                        r.render_tiles_batch_device(W, H, 0, N, [staging[k].data_ptr()] * BATCH)  # same buffer: timing only
                    for _ in range(BATCH):
                        if what != "render":
                            dist.all_gather_into_tensor(gathered[k][:staging[k].numel()], staging[k])
                        if what.endswith("untile"):
                            r.untile_device(W, H, N, gathered[k].data_ptr(), frames[k].data_ptr())
            for i in range(40): step(i)
            torch.cuda.synchronize()
            K = 400
            t0 = time.perf_counter()
            for i in range(K): step(i)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print("N=%d %-22s issue %.1f us/frame   throughput %.1f us/frame (batch %d)" % (N, what, (t1 - t0) / K / BATCH * 1e6, (t2 - t0) / K / BATCH * 1e6, BATCH), flush=True)
    r.set_stream(streams[0].cuda_stream)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
