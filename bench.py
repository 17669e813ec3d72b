#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): Mray/s and ms/frame at 1920x1080 on a 1M-triangle scene, 1 spp primary rays
+ 1 shadow ray per lit hit (configs[2]), with the render kernel placed against the chip's ceilings.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one frame.  N = 1: one launch of the fused rayGen->traverse->shade->store kernel over the whole frame, scene
and output resident in HBM.  N > 1: the SAME frame is tile-partitioned (16x16 macro tiles, tile k -> rank k % N),
every rank renders its tiles into a tile-major staging buffer, ONE RCCL all-gather moves them over xGMI and a
de-interleave kernel rebuilds the row-major frame ("scaling": "strong": total work is fixed).
value = rays traced by the whole job (primary + shadow, counted by the instrumented kernel variant) / wall time.

Pipelining policy: `value` is measured with the SAME policy at every N -- one launch in flight, one frame per launch,
frames issued back to back on one stream -- so the driver's 1 -> 8 curve compares like with like.  The throughput
policy (4 launches in flight on alternating streams, and from 8 ranks up 2 frames per launch) is timed right after
and reported as the labelled extra "pipelined"; it never replaces `value`.

The CPU oracle (oracle/) is used here ONLY as the checker: the cpu_baseline leg (rank 0, N = 1, a bounded number of
full frames) and the comparison of the last timed frame's RGBA8 with the frame that leg renders.
"""
import argparse
import importlib
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before torch / HIP initialise: the host driver only does dmabuf IPC

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0     # HBM3E 8 TB/s spec (6.3 TB/s achievable)
L2_PEAK_GBS = 34500.0     # aggregate L2 bandwidth
N_SIMDS = 256 * 4         # 256 CUs x 4 SIMD-32
CLOCK_HZ = 2.4e9          # max engine clock
VALU_ISSUE_CYCLES = 2.0   # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles ("v_fma_f32 (wave64): 2 cyc")
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
    workload on this host's cores: same scene, same BVH, same arithmetic; OpenMP over image rows.
    Returns the baseline record and the oracle's RGBA8 frame (the checker for the GPU's timed frame)."""
    cam = sc["camera"]
    O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    O.set_width(2)  # binary-tree walk: as fast on the CPU as the 4-wide walk with its SSE slab test (measured within 2 %); same results
    cores = usable_cores(oracle.max_threads())
    ref = O.render(cam["position"], cam["matrix"], MODE, W, H, want=("rgba8",), n_threads=cores)  # warm caches / thread pool
    frames, rays, t0 = 0, 0, time.perf_counter()
    while True:
        st = O.render(cam["position"], cam["matrix"], MODE, W, H, want=("rgba8",), n_threads=cores)["stats"]
        frames += 1
        rays += st["rays_primary"] + st["rays_shadow"]
        dt = time.perf_counter() - t0
        if dt >= budget_s or frames >= 400:
            break
    rec = {"value": rays / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
           "sample": "%d full %dx%d frames of the same workload (%.1f s), CPU restatement (build's own oracle, not "
                     "reference code), binary-BVH traversal, gcc -O2 -mfma -ffp-contract=off + OpenMP schedule(dynamic,1 row)" % (frames, W, H, dt),
           "ms_per_frame": dt / frames * 1e3}
    return rec, np.ascontiguousarray(ref["rgba8"]).view(np.uint32).reshape(-1)


def roofline_block(scene, launch_ms, kernel_ms_median, cnt, alg_bytes):
    """The render kernel against the chip's ceilings.  Instruction and byte counts per launch come from the committed
    rocprofv3 PMC summary of this workload (profiles/roofline_inputs.json, written by tools/parse_profile.py from
    separate --pmc passes as MI355X_MICROARCH.md prescribes; they are properties of the kernel + frame, identical from
    launch to launch); the launch duration is measured live in this run."""
    sec = launch_ms * 1e-3
    out = {"kernel": "renderKernel<false, false>", "kernel_ms_avg_timed_region": launch_ms, "kernel_ms_event_median": kernel_ms_median,
           "bound": "valu-issue", "achieved": None, "peak": N_SIMDS * CLOCK_HZ / VALU_ISSUE_CYCLES / 1e9, "unit": "G wave-instructions/s",
           "frac": None, "traffic": None,
           "peak_note": "1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md)",
           "logical_bytes_per_launch": alg_bytes, "logical_GBs": alg_bytes / sec / 1e9,
           "logical_note": "64 B x quantised wide-node records fetched + 48 B x triangle records fetched + 4 B x pixels (SURVEY.md 8d): per-ray "
                           "logical fetches, served mostly by the scalar cache, L1, L2 and the Infinity Cache -- NOT an HBM rate, no frac",
           "nodes_fetched": cnt["nodes_visited"], "tris_fetched": cnt["tris_tested"]}
    src = os.path.join(ROOT, "profiles", "roofline_inputs.json")
    try:
        inp = json.load(open(src)).get(scene)
    except (OSError, ValueError):
        inp = None
    if inp:
        out["counters_from"] = inp.get("from")
        if inp.get("valu_insts"):
            out["valu_insts_per_launch"] = inp["valu_insts"]
            out["achieved"] = inp["valu_insts"] / sec / 1e9
            out["frac"] = out["achieved"] / out["peak"]
            if inp.get("thread_cycles_valu"):
                out["lanes_active"] = inp["thread_cycles_valu"] / (inp["valu_insts"] * 64.0)
        if inp.get("hbm_read_bytes") is not None and inp.get("hbm_write_bytes") is not None:
            out["traffic"] = inp["hbm_read_bytes"] + inp["hbm_write_bytes"]
            out["hbm_measured_GBs"] = out["traffic"] / sec / 1e9
            out["hbm_peak_GBs"] = HBM_PEAK_GBS
            out["hbm_frac"] = out["hbm_measured_GBs"] / HBM_PEAK_GBS
            out["traffic_note"] = "FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, separate --pmc passes"
        if inp.get("l2_requests"):
            out["l2_GBs"] = inp["l2_requests"] * 128.0 / sec / 1e9
            out["l2_frac"] = out["l2_GBs"] / L2_PEAK_GBS
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scene", default="heightfield", choices=["heightfield", "soup", "heightfield5m"],
                    help="heightfield = BASELINE.json configs[2] (the headline); heightfield5m = same view over 4 999 124 triangles "
                         "(working set 650 MB > the 256 MB Infinity Cache: the HBM-regime data point)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the pipelined / moving-camera / D2H legs (profiling runs)")
    ap.add_argument("--inflight", type=int, default=1, help="launches in flight for `value` (default 1 at every N: frames back to "
                    "back on ONE stream, so the HIP events over the timed region measure the kernel's average launch duration and "
                    "the rocprofv3 summary of the same command agrees with it)")
    ap.add_argument("--batch", type=int, default=1, help="N>1 only: frames per launch (1..4) for `value`; default 1")
    ap.add_argument("--force-dist", action="store_true", help="run the N>1 code path (RCCL init, tile staging, all-gather, "
                    "de-interleave) with whatever world size the launcher gives, even 1")
    ap.add_argument("--check-dist-frame", action="store_true", help="N>1 path: compare the gathered frame with the oracle's "
                    "(tests/test_gpu_parity.py runs this in a child process)")
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
    main_stream = torch.cuda.current_stream()
    r.set_stream(main_stream.cuda_stream)  # the kernels run on torch streams: torch events and RCCL order with them

    frame0 = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    # instrumented variant, once, untimed: exact ray / node / triangle counts of the whole frame
    r.set_counting(True)
    cnt = r.render_frame_device(W, H, frame0.data_ptr(), stats=True)
    r.set_counting(False)
    rays_per_frame = cnt["rays_primary"] + cnt["rays_shadow"]
    alg_bytes_frame = 64 * cnt["nodes_visited"] + 48 * cnt["tris_tested"] + 4 * W * H  # 64-B quantised wide nodes, 48-B triangles

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    share = host.rank_share(W, H, rank, world)
    per = share["slots"] * 256

    def policy(n_fly, batch):
        """Build the per-frame step of one pipelining policy: n_fly launches in flight (alternating streams and buffers),
        `batch` frames per launch (N > 1 only).  Returns (step, set_total, streams, last_frame)."""
        streams = [main_stream] + [torch.cuda.Stream() for _ in range(n_fly - 1)]
        state = {"total": 0, "launches": 0, "last": None}
        if not multi:
            frames = [frame0] + [torch.zeros(W * H, dtype=torch.int32, device="cuda") for _ in range(n_fly - 1)]

            def step(i):
                k = i % n_fly
                r.set_stream(streams[k].cuda_stream)
                r.render_frame_device(W, H, frames[k].data_ptr())
                state["last"] = frames[k]
        else:
            # one contiguous staging / gathered buffer per launch slot: a batch travels in ONE all-gather ([rank][frame][slot])
            staging = [torch.zeros(batch * per, dtype=torch.int32, device="cuda") for _ in range(n_fly)]
            gathered = [torch.zeros(world * batch * per, dtype=torch.int32, device="cuda") for _ in range(n_fly)]
            out = [[torch.zeros(W * H, dtype=torch.int32, device="cuda") for _ in range(batch)] for _ in range(n_fly)]

            def step(i):
                # frames i, i+1, ... of one batch are issued when its first frame is due; the others are already covered
                if i % batch:
                    return
                nb = min(batch, state["total"] - i)
                k = state["launches"] % n_fly
                state["launches"] += 1
                r.set_stream(streams[k].cuda_stream)
                with torch.cuda.stream(streams[k]):
                    r.render_tiles_batch_device(W, H, rank, world, [staging[k].data_ptr() + 4 * per * f for f in range(nb)])
                    host.gather_batch(staging[k][:nb * per], W, H, nb,
                                      lambda g, f, k=k, nb=nb: (r.untile_batch_device(W, H, world, nb, f, g.data_ptr(), out[k][f].data_ptr()), out[k][f])[1],
                                      gathered[k][:world * nb * per])
                state["last"] = out[k][nb - 1]
        return step, state, streams

    def timed(step, state, streams, steps, warmup):
        state["total"] = warmup
        for i in range(warmup):
            step(i)
        fence()
        state["total"] = steps
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(main_stream)
        for i in range(steps):
            step(i)
        for st in streams[1:]:
            main_stream.wait_stream(st)
        ev1.record(main_stream)
        fence()
        elapsed = time.perf_counter() - t0
        stream_ms = ev0.elapsed_time(ev1)  # HIP events over the timed region (the main stream joins the other streams first)
        r.set_stream(main_stream.cuda_stream)
        if multi:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        return elapsed, stream_ms

    n_fly = max(1, args.inflight)
    batch = max(1, min(4, args.batch)) if multi else 1
    step, state, streams = policy(n_fly, batch)
    if n_fly == 1:
        # scene setup, like the BVH build: the launch order of an unchanged view settles after each of the library's 4 scratch
        # slots has measured the view twice (crt_api.cpp); done here so that any --warmup, even 0, times the settled state
        state["total"] = 12
        for i in range(12):
            step(i)
        fence()
    elapsed, stream_ms = timed(step, state, streams, args.steps, args.warmup)
    last_frame = state["last"].cpu().numpy().view(np.uint32) if rank == 0 else None

    # own-kernel share of this rank, measured per launch with the library's HIP events (render kernel only)
    kms = []
    scratch = torch.zeros(per, dtype=torch.int32, device="cuda")
    for _ in range(min(20, args.steps)):
        if not multi:
            kms.append(r.render_frame_device(W, H, frame0.data_ptr(), stats=True)["kernel_ms"])
        else:
            kms.append(r.render_tiles_device(W, H, rank, world, scratch.data_ptr(), stats=True)["kernel_ms"])
    kernel_ms = float(np.median(kms))

    extras = {}
    if not args.no_extras:
        # throughput policy, labelled extra: 4 launches in flight on alternating streams; from 8 ranks up 2 frames per launch
        p_fly, p_batch = 4, (2 if (multi and world >= 8) else 1)
        pstep, pstate, pstreams = policy(p_fly, p_batch)
        p_elapsed, _ = timed(pstep, pstate, pstreams, args.steps, max(args.warmup, 8))
        extras["pipelined"] = {"launches_in_flight": p_fly, "frames_per_launch": p_batch, "ms_per_frame": p_elapsed / args.steps * 1e3,
                               "value": rays_per_frame * args.steps / p_elapsed / 1e6, "unit": "Mray/s",
                               "note": "throughput policy; `value` above uses 1 launch in flight, 1 frame per launch at every N"}
    if not args.no_extras and not multi:
        # a camera that moves every frame (0.01 degrees of orbit: practically the same view, so the difference to the static
        # figure is the cost of measuring and sorting the launch order for every frame, as an interactive viewer pays it)
        pos0 = np.float32(cam["position"])
        rot0 = np.float32(cam["matrix"]).reshape(3, 3)

        def cam_at(i):
            a = np.radians(0.01 * i)
            c, s = np.cos(a), np.sin(a)
            R = np.float32([[c, 0, s], [0, 1, 0], [-s, 0, c]])
            return (R @ pos0).astype(np.float32), (R @ rot0).astype(np.float32).reshape(9)
        K = min(args.steps, 200)
        cams = [cam_at(i) for i in range(30 + K)]
        for i in range(30):
            r.set_camera(*cams[i])
            r.render_frame_device(W, H, frame0.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            r.set_camera(*cams[30 + i])
            r.render_frame_device(W, H, frame0.data_ptr())
        torch.cuda.synchronize()
        extras["moving_camera_ms_per_frame"] = (time.perf_counter() - t0) / K * 1e3
        r.set_camera(cam["position"], cam["matrix"])
        # PCIe-inclusive variant (host output buffer handed over the C ABI), for DESIGN.md; never `value`
        r.reset_stream()
        d2h = {}
        for pinned in (False, True):  # pageable numpy buffer vs page-locked crt_host_alloc buffer
            r.render_frame(W, H, want=(), pinned=pinned)
            t0 = time.perf_counter()
            for _ in range(5):
                r.render_frame(W, H, want=(), pinned=pinned)
            d2h[pinned] = (time.perf_counter() - t0) / 5 * 1e3
        extras["ms_per_frame_incl_d2h"], extras["ms_per_frame_incl_d2h_pinned"] = d2h[False], d2h[True]
        r.set_stream(main_stream.cuda_stream)

    ok = True
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
                       "camera": "static (launch order of the unchanged view settled before the timed region); moving camera: moving_camera_ms_per_frame",
                       "parallelism": "1 GPU, one launch per frame" if world == 1 else "framebuffer tiles 16x16 round-robin over %d GPUs + 1 RCCL all-gather/frame" % world,
                       "bvh": {"nodes": r.bvh_info()["n_nodes"], "max_depth": r.bvh_info()["max_depth"], "build_and_upload_s": upload_s}},
            "ms_per_frame": ms_per_step,
            "stream_ms_per_step": stream_ms / args.steps,
        }
        line.update(extras)
        if not multi:
            # dominant (only) kernel: renderKernel<false, false>.  Its average launch duration = HIP events on its stream around
            # the K back-to-back launches of the timed region / K (with 1 launch in flight the stream holds nothing else: the
            # 1-workgroup sortUnitsKernel that orders a later frame's launch runs on a side stream)
            launch_ms = stream_ms / args.steps if n_fly == 1 else kernel_ms
            line["roofline"] = roofline_block(args.scene, launch_ms, kernel_ms, cnt, alg_bytes_frame)
        else:
            line["rank0_render_kernel_ms"] = kernel_ms
        want_check = (not multi and not args.no_cpu_baseline) or args.check_dist_frame
        if want_check:
            base, ref = cpu_baseline(entry.load_oracle(), sc, budget_s=10.0 if not multi else 0.0)
            if not multi:
                line["cpu_baseline"] = base
                line["speedup_vs_cpu_baseline"] = value / base["value"]
            # the timed frame is the frame the oracle renders: RGBA8 identical, pixel for pixel
            ok = bool(np.array_equal(last_frame, ref))
            line["frame_matches_oracle"] = ok
            if not ok:
                line["frame_mismatching_pixels"] = int((last_frame != ref).sum())
        print(json.dumps(line))
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    r.close()
    if not ok:
        raise SystemExit("bench.py: the timed frame differs from the oracle's frame")


if __name__ == "__main__":
    main()
