#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): Mray/s and ms/frame at 1920x1080 on a 1M-triangle scene, 1 spp primary rays
+ 1 shadow ray per lit hit (configs[2]); algorithmic GB/s of the render kernel against MI355X's 8 TB/s HBM peak.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one frame.  N = 1: one launch of the fused rayGen->traverse->shade->store kernel over the whole frame, scene
and output resident in HBM.  N > 1: the SAME frame is tile-partitioned (16x16 macro tiles, tile k -> rank k % N),
every rank renders its tiles into a tile-major staging buffer, ONE RCCL all-gather moves them over xGMI and a
de-interleave kernel rebuilds the row-major frame ("scaling": "strong": total work is fixed).
value = rays traced by the whole job (primary + shadow, counted by the instrumented kernel variant) / wall time.

The CPU oracle (oracle/) is used here ONLY for the cpu_baseline leg: rank 0, N = 1, a bounded number of full frames.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
W, H = 1920, 1080
MODE = 100  # Lambert + one shadow ray per light


def usable_cores(omp_default):
    """threads the host really gives this process: affinity mask and cgroup CPU quota (the GPU box grants a share)"""
    n = min(omp_default, len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(oracle, sc, budget_s=10.0):
    """CPU restatement (the build's own oracle, NOT reference code: the reference has no CPU renderer) of the same
    workload on this host's cores: same scene, same BVH, same arithmetic; OpenMP over image rows."""
    cam = sc["camera"]
    O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    O.set_width(2)  # binary-tree walk: as fast on the CPU as the 4-wide walk with its SSE slab test (measured within 2 %); same results
    cores = usable_cores(oracle.max_threads())
    O.render(cam["position"], cam["matrix"], MODE, W, H, want=("rgba8",), n_threads=cores)  # warm caches / thread pool
    frames, rays, t0 = 0, 0, time.perf_counter()
    while True:
        st = O.render(cam["position"], cam["matrix"], MODE, W, H, want=("rgba8",), n_threads=cores)["stats"]
        frames += 1
        rays += st["rays_primary"] + st["rays_shadow"]
        dt = time.perf_counter() - t0
        if dt >= budget_s or frames >= 400:
            break
    return {"value": rays / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
            "sample": "%d full %dx%d frames of the same workload (%.1f s), CPU restatement (build's own oracle, not "
                      "reference code), binary-BVH traversal, gcc -O2 -mfma -ffp-contract=off + OpenMP schedule(dynamic,1 row)" % (frames, W, H, dt),
            "ms_per_frame": dt / frames * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scene", default="heightfield", choices=["heightfield", "soup", "heightfield5m"],
                    help="heightfield = BASELINE.json configs[2] (the headline); heightfield5m = same view over 4 999 124 triangles "
                         "(working set 650 MB > the 256 MB Infinity Cache: the HBM-regime data point)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=0, help="frames in flight: consecutive frames alternate over this many HIP "
                    "streams and frame buffers (the reference keeps 2 swap-chain buffers, R/DXRTRenderer.cpp:178-204). Default: 1 at "
                    "N=1 (launches back to back on ONE stream, so the HIP events over the timed region measure the kernel's average "
                    "launch duration and the rocprofv3 summary of the same command agrees with it), 4 at N>1 (the tail of one "
                    "rank's tile launch and the all-gather overlap the next frame)")
    ap.add_argument("--batch", type=int, default=0, help="N>1 only: frames per launch (1..4) of each rank's tile share; 0 = 2 from 8 ranks up, else 1")
    ap.add_argument("--force-dist", action="store_true", help="rehearse the N>1 code path (RCCL init, tile staging, all-gather, "
                    "de-interleave) with whatever world size the launcher gives, even 1")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm

    pkg = entry.load_package()
    scenes = importlib.import_module(entry.PKG_NAME + ".scenes")
    host = importlib.import_module(entry.PKG_NAME + ".multigpu")

    sc = {"heightfield": lambda: scenes.heightfield(n_lights=1), "soup": scenes.icosphere_soup,
          "heightfield5m": lambda: scenes.heightfield(n=1581, n_lights=1)}[args.scene]()
    n_tris = sum(len(m["triangles"]) for m in sc["meshes"])
    cam = sc["camera"]
    r = pkg.Renderer(local_rank)
    t0 = time.perf_counter()
    r.upload(sc["meshes"], sc["lights"], sc["materials"])
    upload_s = time.perf_counter() - t0
    r.set_camera(cam["position"], cam["matrix"])
    r.change_shading_mode(MODE)
    n_fly = args.inflight if args.inflight > 0 else (4 if multi else 1)
    # frames per launch at N > 1: a 1/8 share is bound by the launch's slowest packet, which a batch shares (66 -> 50 us per
    # frame measured for an 8-rank share); shares of 1/2 and 1/4 are not (no gain measured)
    batch = (max(1, min(4, args.batch)) if args.batch > 0 else (2 if world >= 8 else 1)) if multi else 1
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(n_fly - 1)]
    stream = streams[0]
    r.set_stream(stream.cuda_stream)  # the kernels run on torch streams: torch events and RCCL order with them

    frames = [torch.zeros(W * H, dtype=torch.int32, device="cuda") for _ in range(n_fly)]
    frame = frames[0]
    # instrumented variant, once, untimed: exact ray / node / triangle counts of the whole frame
    r.set_counting(True)
    cnt = r.render_frame_device(W, H, frame.data_ptr(), stats=True)
    r.set_counting(False)
    rays_per_frame = cnt["rays_primary"] + cnt["rays_shadow"]
    alg_bytes_frame = 128 * cnt["nodes_visited"] + 48 * cnt["tris_tested"] + 4 * W * H  # 128-B wide nodes, 48-B triangles

    if not multi:
        def step(i):
            k = i % n_fly
            r.set_stream(streams[k].cuda_stream)
            r.render_frame_device(W, H, frames[k].data_ptr())
    else:
        # N > 1: each launch carries `batch` consecutive frames of this rank's tile share (crt_render_tiles_batch_device): a
        # launch lasts as long as its slowest packet, and a 1/N share has the same slowest packet as the whole frame
        share = host.rank_share(W, H, rank, world)
        # one contiguous staging / gathered buffer per launch slot: the batch travels in ONE all-gather ([rank][frame][slot])
        per = share["slots"] * 256
        staging = [torch.zeros(batch * per, dtype=torch.int32, device="cuda") for _ in range(n_fly)]
        gathered = [torch.zeros(world * batch * per, dtype=torch.int32, device="cuda") for _ in range(n_fly)]
        out = [[torch.zeros(W * H, dtype=torch.int32, device="cuda") for _ in range(batch)] for _ in range(n_fly)]
        launches = [0]

        def step(i):
            # frames i, i+1, ... of one batch are issued when its first frame is due; the others are already covered
            if i % batch:
                return
            nb = min(batch, step.total - i)
            k = launches[0] % n_fly
            launches[0] += 1
            r.set_stream(streams[k].cuda_stream)
            with torch.cuda.stream(streams[k]):
                r.render_tiles_batch_device(W, H, rank, world, [staging[k].data_ptr() + 4 * per * f for f in range(nb)])
                host.gather_batch(staging[k][:nb * per], W, H, nb,
                                  lambda g, f, k=k, nb=nb: (r.untile_batch_device(W, H, world, nb, f, g.data_ptr(), out[k][f].data_ptr()), out[k][f])[1],
                                  gathered[k][:world * nb * per])
        step.total = 0

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    if not multi and n_fly == 1:
        # scene setup, like the BVH build: the launch order of an unchanged view settles after each of the library's 4 scratch
        # slots has measured the view twice (crt_api.cpp); done here so that any --warmup, even 0, times the settled state
        for _ in range(12):
            r.render_frame_device(W, H, frame.data_ptr())
        torch.cuda.synchronize()
    if multi:
        step.total = args.warmup
    for i in range(args.warmup):
        step(i)
    fence()
    if multi:
        step.total = args.steps
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(args.steps):
        step(i)
    for st in streams[1:]:
        stream.wait_stream(st)
    ev1.record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    stream_ms = ev0.elapsed_time(ev1)  # HIP events over the timed region (stream 0 joins the other streams first)
    r.set_stream(stream.cuda_stream)
    if multi:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # own-kernel share of this rank, measured per launch with the library's HIP events (render kernel only)
    kms = []
    for _ in range(min(20, args.steps)):
        if not multi:
            kms.append(r.render_frame_device(W, H, frame.data_ptr(), stats=True)["kernel_ms"])
        else:
            kms.append(r.render_tiles_device(W, H, rank, world, staging[0].data_ptr(), stats=True)["kernel_ms"])
    kernel_ms = float(np.median(kms))

    # PCIe-inclusive variant (host output buffer handed over the C ABI), for DESIGN.md; never `value`
    d2h_ms = d2h_pinned_ms = None
    if not multi:
        r.reset_stream()
        d2h = {}
        for pinned in (False, True):  # pageable numpy buffer vs page-locked crt_host_alloc buffer
            r.render_frame(W, H, want=(), pinned=pinned)
            t0 = time.perf_counter()
            for _ in range(5):
                r.render_frame(W, H, want=(), pinned=pinned)
            d2h[pinned] = (time.perf_counter() - t0) / 5 * 1e3
        d2h_ms, d2h_pinned_ms = d2h[False], d2h[True]
        r.set_stream(stream.cuda_stream)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_per_frame * args.steps / elapsed / 1e6
        line = {
            "metric": "Mray/s", "value": value, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %d triangles, %dx%d, 1 spp primary + 1 shadow ray per lit hit (mode 100, %d light)"
                                   % ({"heightfield": "seeded height field 708x708 quads + ground quad (BASELINE.json configs[2])",
                                       "heightfield5m": "seeded height field 1581x1581 quads + ground quad",
                                       "soup": "seeded soup of 3125 copied icospheres + ground quad"}[args.scene], n_tris, W, H, len(sc["lights"])),
                       "rays_per_frame": rays_per_frame, "primary_rays": cnt["rays_primary"], "shadow_rays": cnt["rays_shadow"],
                       "frames_in_flight": n_fly * batch, "launches_in_flight": n_fly, "frames_per_launch": batch,
                       "parallelism": "1 GPU, one launch per frame" if world == 1 else "framebuffer tiles 16x16 round-robin over %d GPUs + 1 RCCL all-gather/frame" % world,
                       "bvh": {"nodes": r.bvh_info()["n_nodes"], "max_depth": r.bvh_info()["max_depth"], "build_and_upload_s": upload_s}},
            "ms_per_frame": ms_per_step,
            "stream_ms_per_step": stream_ms / args.steps,
        }
        if not multi:
            # dominant (only) kernel: renderKernel<false, false>.  Its average launch duration = HIP events on its stream around
            # the K back-to-back launches of the timed region / K (with --inflight 1 the stream holds nothing else: the
            # 1-workgroup sortUnitsKernel that orders a later frame's launch runs on a side stream); the median of per-launch
            # event pairs is reported beside it.  With more frames in flight launches overlap and only the median is a
            # single launch's duration.
            launch_ms = stream_ms / args.steps if n_fly == 1 else kernel_ms
            achieved = alg_bytes_frame / (launch_ms * 1e-3) / 1e9
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(tfile):
                try:
                    traffic = json.load(open(tfile)).get(args.scene, {}).get("bytes_per_launch")
                except Exception:
                    traffic = None
            line["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                                "kernel": "renderKernel<false, false>", "algorithmic_bytes_per_launch": alg_bytes_frame,
                                "nodes_fetched": cnt["nodes_visited"], "tris_fetched": cnt["tris_tested"],
                                "kernel_ms_avg_timed_region": launch_ms, "kernel_ms_event_median": kernel_ms,
                                "note": "bytes = 128 B x wide-node records fetched + 48 B x triangle records fetched + 4 B x pixels; "
                                        "the 113 MB working set is served mostly by L2 / Infinity Cache, so achieved may exceed what HBM itself moves"}
            line["ms_per_frame_incl_d2h"] = d2h_ms
            line["ms_per_frame_incl_d2h_pinned"] = d2h_pinned_ms
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(entry.load_oracle(), sc)
                line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
        else:
            line["rank0_render_kernel_ms"] = kernel_ms
        print(json.dumps(line))
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
