#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): Mray/s and ms/frame at 1920x1080 on a 1M-triangle scene, 1 spp primary rays
+ 1 shadow ray per lit hit (configs[2]), with the render kernel placed against the chip's ceilings.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher's environment: starts its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  python bench.py --config c5 [--gpus N]                   BASELINE.json configs[4]: 5M triangles, 3840x2160, 4 spp path traced, 3 bounces

A step = one frame.  N = 1: one launch of the fused rayGen->traverse->shade->store kernel over the whole frame, scene
and output resident in HBM.  N > 1: the SAME frame is tile-partitioned (16x16 macro tiles, tile k -> rank k % N),
every rank renders its tiles into a tile-major staging buffer, ONE RCCL all-gather moves them over xGMI and a
de-interleave kernel rebuilds the row-major frame ("scaling": "strong": total work is fixed).
value = rays traced by the whole job (primary + shadow, counted by the instrumented kernel variant) / wall time.

Pipelining policy.  N = 1: `value` is one launch in flight, one frame per launch, frames back to back on one stream (the
roofline block needs the kernel's own launch duration over the timed region); the throughput policy (4 launches in flight
on alternating streams) is timed right after and reported as the labelled extra "pipelined".  N > 1: `value`
is the throughput policy -- 4 launches in flight, from 8 ranks up 4 frames per launch -- because a rank's share of a
1080p frame is tens of microseconds of work behind a tail of one grazing packet and a frame-end exchange, which only
other frames' work can hide (one-GPU emulation, tools/dist_overhead.py: 8 ranks 121 us per frame with one launch in
flight, 60 with four, 44 with four frames per launch: 6.5x one GPU's 285); the one-launch-in-flight figure is timed right
after and reported as "one_launch_in_flight".  Every frame of the timed region is complete, gathered and de-interleaved
on every rank before the closing barrier.  --inflight / --batch override both defaults.

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
# Issue cost of a wave64 vector instruction on one SIMD, measured on the MI355X (tools/micro/valu_cost.hip, 8 wavefronts per SIMD,
# profiles/r03_valu_cost.txt): v_fma / v_mul / v_add / v_sub / v_mov / v_and / v_or / v_xor / v_add_u32 issue every 2.5 cycles,
# everything else the kernels use (v_cvt_*, float and integer min / max / med3, compares, v_cndmask, bit-field and all VOP3-only
# integer forms, SDWA, v_fma_mix, v_perm) every 4.2 -- half rate -- and v_rcp / v_sqrt every 8.2.  The guide's "v_fma_f32
# (wave64): 2 cyc" is the first class only; round 2 priced every instruction at 2 cycles and under-stated the VALU share.
VALU_FAST_CYCLES = 2.5
VALU_SLOW_CYCLES = 4.2
VALU_TRANS_CYCLES = 8.2
KERNEL_SOURCES = ("traversal.hip.h", "shading.hip.h", "render_kernels.hip", "path_kernels.hip", "render_kernels.h")

CONFIGS = {
    # BASELINE.json configs[2]: the configuration the metric is quoted on
    "c3": {"scene": "heightfield", "w": 1920, "h": 1080, "mode": 100, "kernel": "renderKernel<false, false, LayLegacy>",
           "what": "1 spp primary + 1 shadow ray per lit hit (mode 100, %d light)"},
    # BASELINE.json configs[4]
    "c5": {"scene": "heightfield5m", "w": 3840, "h": 2160, "mode": 200, "spp": 4, "bounces": 3, "seed": 1234, "kernel": "pathKernel<false, LayLegacy>",
           "what": "4 spp path traced, 3 bounces, 1 shadow ray per light at every diffuse hit (mode 200, %d light)"},
}


def kernel_source_hash():
    """sha256 over the kernel sources the committed PMC counters were collected for (tools/parse_profile.py stores it)"""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "directx-raytracer_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def self_launch(n, argv):
    """`python bench.py --gpus N` with no launcher environment: start N ranks with torch.distributed.run as a CHILD process and
    relay its exit code.  Called before torch or anything else has touched a GPU (the parent never does)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, cwd=ROOT)


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


def cpu_baseline(oracle, sc, cfg, budget_s=10.0):
    """CPU restatement (the build's own oracle, NOT reference code: the reference has no CPU renderer) of the same
    workload on this host's cores: same scene, same BVH, same arithmetic; OpenMP over image rows.
    c3: whole frames.  c5: the frame takes the oracle minutes, so the sample is every `stride`-th row of the frame (rows are
    independent; the stride is chosen so that one pass lasts about the budget).
    Returns the baseline record, the oracle's RGBA8 frame and the rows of it that are valid (the checker for the GPU's frame)."""
    cam = sc["camera"]
    W, H, mode = cfg["w"], cfg["h"], cfg["mode"]
    O = oracle.OracleScene(sc["meshes"], sc["lights"], sc["materials"])
    O.set_width(2)  # binary-tree walk: as fast on the CPU as the 4-wide walk with its SSE slab test (measured within 2 %); same results
    cores = usable_cores(oracle.max_threads())
    if mode >= 200:
        oracle.set_path_params(cfg["spp"], cfg["bounces"], cfg["seed"])
        probe_rows = (0, H, max(1, H // 16))
        t0 = time.perf_counter()
        O.render(cam["position"], cam["matrix"], mode, W, H, rows=probe_rows, want=("rgba8",), n_threads=cores)
        per_row = (time.perf_counter() - t0) / len(range(*probe_rows))
        stride = max(1, int(np.ceil(per_row * H / max(budget_s, 1.0))))
        rows = (stride // 2, H, stride)
        t0 = time.perf_counter()
        ref = O.render(cam["position"], cam["matrix"], mode, W, H, rows=rows, want=("rgba8",), n_threads=cores)
        dt = time.perf_counter() - t0
        st = ref["stats"]
        rays = st["rays_primary"] + st["rays_shadow"]
        n_rows = len(range(*rows))
        rec = {"value": rays / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
               "sample": "%d of the %d rows of the same %dx%d frame (every %d-th row, %d rays, %.1f s), CPU restatement (build's own oracle, "
                         "not reference code), binary-BVH traversal, gcc -O2 -mfma -ffp-contract=off + OpenMP schedule(dynamic,1 row)"
                         % (n_rows, H, W, H, stride, rays, dt),
               "ms_per_frame": dt / n_rows * H * 1e3}
        return rec, np.ascontiguousarray(ref["rgba8"]).view(np.uint32).reshape(H, W), np.arange(*rows)
    ref = O.render(cam["position"], cam["matrix"], mode, W, H, want=("rgba8",), n_threads=cores)  # warm caches / thread pool
    frames, rays, t0 = 0, 0, time.perf_counter()
    while budget_s > 0.0:
        st = O.render(cam["position"], cam["matrix"], mode, W, H, want=("rgba8",), n_threads=cores)["stats"]
        frames += 1
        rays += st["rays_primary"] + st["rays_shadow"]
        dt = time.perf_counter() - t0
        if dt >= budget_s or frames >= 400:
            break
    rec = None
    if frames:
        rec = {"value": rays / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
               "sample": "%d full %dx%d frames of the same workload (%.1f s), CPU restatement (build's own oracle, not "
                         "reference code), binary-BVH traversal, gcc -O2 -mfma -ffp-contract=off + OpenMP schedule(dynamic,1 row)" % (frames, W, H, dt),
               "ms_per_frame": dt / frames * 1e3}
    return rec, np.ascontiguousarray(ref["rgba8"]).view(np.uint32).reshape(H, W), np.arange(H)


def roofline_block(key, kernel, launch_ms, kernel_ms_median, cnt, alg_bytes):
    """The dominant kernel against the chip's ceilings.  Instruction and byte counts per launch come from the committed
    rocprofv3 PMC summary of this workload (profiles/roofline_inputs.json, written by tools/parse_profile.py from
    separate --pmc passes as MI355X_MICROARCH.md prescribes; they are properties of the kernel + frame, identical from
    launch to launch); the launch duration is measured live in this run.  The summary records a hash of the kernel sources it
    was taken for: when the sources have changed since, the counter-derived fields are withheld ("counters_stale")."""
    sec = launch_ms * 1e-3
    out = {"kernel": kernel, "kernel_ms_avg_timed_region": launch_ms, "kernel_ms_event_median": kernel_ms_median,
           "bound": "valu-issue", "achieved": None, "peak": N_SIMDS * CLOCK_HZ / 1e9, "unit": "G SIMD issue cycles/s",
           "frac": None, "traffic": None,
           "peak_note": "1024 SIMDs x 2.4 GHz; a kernel's issue cycles = sum over its wave64 vector instructions of the measured issue cost "
                        "of their class (%.1f fma/mul/add/mov/and/or, %.1f conversions/min/max/compare/select/bit-field, %.1f rcp/sqrt: "
                        "tools/micro/valu_cost.hip)" % (VALU_FAST_CYCLES, VALU_SLOW_CYCLES, VALU_TRANS_CYCLES),
           "logical_bytes_per_launch": alg_bytes, "logical_GBs": alg_bytes / sec / 1e9,
           "logical_note": "64 B x quantised wide-node records fetched + 48 B x triangle records fetched + 4 B x pixels (SURVEY.md 8d): per-ray "
                           "logical fetches, served mostly by the scalar cache, L1, L2 and the Infinity Cache -- NOT an HBM rate, no frac",
           "nodes_fetched": cnt["nodes_visited"], "tris_fetched": cnt["tris_tested"]}
    src = os.path.join(ROOT, "profiles", "roofline_inputs.json")
    try:
        inp = json.load(open(src)).get(key)
    except (OSError, ValueError):
        inp = None
    if inp:
        out["counters_from"] = inp.get("from")
        if inp.get("kernel_source_hash") != kernel_source_hash():
            out["counters_stale"] = True  # the kernel that ran is not the kernel that was profiled: no counter-derived figure
            return out
        if inp.get("valu_insts"):
            out["valu_insts_per_launch"] = inp["valu_insts"]
            fp32 = inp.get("valu_fp32_fma_add_mul_insts")
            if fp32 is not None:
                cvt, trans = inp.get("valu_cvt_insts", 0.0), inp.get("valu_trans_insts", 0.0)
                rest = inp["valu_insts"] - fp32 - cvt - trans  # integer + float min / max / compare / select / move: classes the counters lump
                known = fp32 * VALU_FAST_CYCLES + cvt * VALU_SLOW_CYCLES + trans * VALU_TRANS_CYCLES
                lower, upper = known + rest * VALU_FAST_CYCLES, known + rest * VALU_SLOW_CYCLES
                out["valu_mix"] = {"fp32_fma_add_mul": fp32, "conversions": cvt, "transcendental": trans, "other": rest}
                out["valu_issue_cycles_per_launch"] = lower
                out["achieved"] = lower / sec / 1e9
                out["frac"] = out["achieved"] / out["peak"]
                out["frac_upper"] = upper / sec / 1e9 / out["peak"]
                out["frac_note"] = ("`frac` prices the instructions the counters do not classify (integer, float min / max, compares, selects, "
                                    "moves) at the full rate: a lower bound of the SIMDs' issue cycles; `frac_upper` prices them at half rate "
                                    "(most of them are: tools/micro/valu_cost.hip) and may exceed 1")
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


def default_policy(world, multi, path, inflight=0, batch=0):
    """(launches in flight, frames per launch) of the timed region: see the module docstring.  0 = default."""
    n_fly = inflight if inflight > 0 else (4 if multi else 1)
    if not multi:
        return n_fly, 1
    return n_fly, (max(1, min(4, batch)) if batch > 0 else (4 if (world >= 8 and not path) else 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS), help="c3 = BASELINE.json configs[2], the configuration the metric "
                    "is quoted on (default); c5 = configs[4]: 5M triangles, 3840x2160, 4 spp path traced, 3 bounces")
    ap.add_argument("--scene", default=None, choices=["heightfield", "soup", "heightfield5m"],
                    help="another scene under the chosen config's settings; heightfield5m = the same view over 4 999 124 triangles "
                         "(working set 650 MB > the 256 MB Infinity Cache: the HBM-regime data point)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the pipelined / moving-camera / D2H legs (profiling runs)")
    ap.add_argument("--inflight", type=int, default=0, help="launches in flight for `value`.  Default (0): 1 at N = 1 -- frames back to "
                    "back on ONE stream, so the HIP events over the timed region measure the kernel's average launch duration and the "
                    "rocprofv3 summary of the same command agrees with it; 4 at N > 1 -- a rank's share of a frame is a short launch "
                    "whose tail and whose frame-end exchange only overlap with other frames' work")
    ap.add_argument("--batch", type=int, default=0, help="N>1 only: frames per launch (1..4) for `value`.  Default (0): 1 below 8 ranks, "
                    "4 from 8 ranks up (a 1/8 share of a 1080p frame is 37 us of work: one launch and one all-gather per frame "
                    "are host-issue bound at 59 us, tools/dist_overhead.py)")
    ap.add_argument("--force-dist", action="store_true", help="run the N>1 code path (RCCL init, tile staging, all-gather, "
                    "de-interleave) with whatever world size the launcher gives, even 1")
    ap.add_argument("--check-dist-frame", action="store_true", help="N>1 path: compare the gathered frame with the oracle's even when "
                    "the CPU baseline is switched off")
    ap.add_argument("--spinup-ms", type=float, default=150.0, help="untimed frames before the timed region, by wall time: clocks and "
                    "caches at their steady state even for a short --steps")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: with --rendezvous-only (CPU tests), or with "
                    "--same-device: the N > 1 code path with every rank on GPU 0 and the tiles exchanged through host memory (RCCL "
                    "refuses two ranks on one GPU) -- a rehearsal of the rank logic on a one-GPU box, never a measurement")
    ap.add_argument("--same-device", action="store_true", help="every rank renders on GPU 0 (rehearsal with --backend gloo)")
    ap.add_argument("--rendezvous-only", action="store_true", help="the ranks meet, add up their ranks and leave: checks the launch path "
                    "without a GPU (tests/test_tiling.py)")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not launched:
        # no launcher set the rank environment: start the N ranks ourselves, as a child process, BEFORE importing torch or
        # touching a GPU (a process that has initialised the GPU must never be replaced or forked into ranks)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if args.rendezvous_only:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(args.backend)
        t = torch.tensor([rank + 1], dtype=torch.int64)
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            t = t.cuda()
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"rendezvous_ok": int(t.item()) == world * (world + 1) // 2, "n_gpus": world, "backend": args.backend}))
        dist.barrier()
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    if args.same_device:
        if args.backend != "gloo":
            raise SystemExit("bench.py: --same-device needs --backend gloo (RCCL refuses two ranks on one GPU)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_dist
    ctl = "cuda" if args.backend == "nccl" else "cpu"  # where the small control tensors of the collectives live
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    pkg = entry.load_package()
    scenes = importlib.import_module(entry.PKG_NAME + ".scenes")
    host = importlib.import_module(entry.PKG_NAME + ".multigpu")

    cfg = dict(CONFIGS[args.config])
    if args.scene:
        cfg["scene"] = args.scene
    W, H, MODE = cfg["w"], cfg["h"], cfg["mode"]
    path = MODE >= 200
    sc = {"heightfield": lambda: scenes.heightfield(n_lights=1), "soup": scenes.icosphere_soup,
          "heightfield5m": lambda: scenes.heightfield(n=1581, n_lights=1)}[cfg["scene"]]()
    n_tris = sum(len(m["triangles"]) for m in sc["meshes"])
    cam = sc["camera"]
    r = pkg.Renderer(local_rank)
    t0 = time.perf_counter()
    r.upload(sc["meshes"], sc["lights"], sc["materials"])
    upload_s = time.perf_counter() - t0
    r.set_camera(cam["position"], cam["matrix"])
    r.change_shading_mode(MODE)
    if path:
        r.set_path_params(cfg["spp"], cfg["bounces"], cfg["seed"])
    main_stream = torch.cuda.current_stream()
    r.set_stream(main_stream.cuda_stream)  # the kernels run on torch streams: torch events and RCCL order with them

    frame0 = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
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
        `batch` frames per launch (N > 1 only).  Returns (step, state, streams); state["gather_us"] collects the event-timed
        all-gather + de-interleave of the launches when state["time_gather"] is set."""
        streams = [main_stream] + [torch.cuda.Stream() for _ in range(n_fly - 1)]
        state = {"total": 0, "launches": 0, "last": None, "time_gather": False, "gather_ev": []}
        if not multi:
            frames = [frame0] + [torch.zeros(W * H, dtype=torch.int32, device="cuda") for _ in range(n_fly - 1)]

            def step(i):
                k = i % n_fly
                r.set_stream(streams[k].cuda_stream)
                r.render_frame_device(W, H, frames[k].data_ptr())
                state["last"] = frames[k]
        else:
            # one contiguous staging / gathered buffer per launch slot: a batch travels in ONE all-gather ([rank][frame][slot]).
            # With one launch in flight (the policy of `value`) the frame-end exchange -- all-gather + de-interleave -- runs on a
            # communication stream of its own with double buffers, so frame f's tiles travel while frame f + 1 renders: the render
            # launches still run strictly one after the other, the collective overlaps compute
            overlap = n_fly == 1
            slots = 2 if overlap else n_fly
            comm = torch.cuda.Stream() if overlap else None
            if overlap:
                streams.append(comm)
            staging = [torch.zeros(batch * per, dtype=torch.int32, device="cuda") for _ in range(slots)]
            gathered = [torch.zeros(world * batch * per, dtype=torch.int32, device="cuda") for _ in range(slots)]
            out = [[torch.zeros(W * H, dtype=torch.int32, device="cuda") for _ in range(batch)] for _ in range(slots)]
            exchanged = [None] * slots  # event: the exchange that last read staging[k] / wrote out[k] is done

            def step(i):
                # frames i, i+1, ... of one batch are issued when its first frame is due; the others are already covered
                if i % batch:
                    return
                nb = min(batch, state["total"] - i)
                k = state["launches"] % slots
                state["launches"] += 1
                rs = streams[0] if overlap else streams[k]   # render stream
                xs = comm if overlap else streams[k]         # exchange stream
                r.set_stream(rs.cuda_stream)
                if overlap and exchanged[k] is not None:
                    rs.wait_event(exchanged[k])              # the buffers of two frames ago are free again
                with torch.cuda.stream(rs):
                    r.render_tiles_batch_device(W, H, rank, world, [staging[k].data_ptr() + 4 * per * f for f in range(nb)])
                if overlap:
                    rendered = torch.cuda.Event()
                    rendered.record(rs)
                    xs.wait_event(rendered)
                    r.set_stream(xs.cuda_stream)             # (the de-interleave kernel goes where the collective goes)
                with torch.cuda.stream(xs):
                    if state["time_gather"]:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(xs)
                    host.gather_batch(staging[k][:nb * per], W, H, nb,
                                      lambda g, f, k=k, nb=nb: (r.untile_batch_device(W, H, world, nb, f, g.data_ptr(), out[k][f].data_ptr()), out[k][f])[1],
                                      gathered[k][:world * nb * per])
                    if state["time_gather"]:
                        e1.record(xs)
                        state["gather_ev"].append((e0, e1))
                    if overlap:
                        exchanged[k] = torch.cuda.Event()
                        exchanged[k].record(xs)
                if overlap:
                    r.set_stream(rs.cuda_stream)
                state["last"] = out[k][nb - 1]
        return step, state, streams

    def timed(step, state, streams, steps, warmup):
        state["total"] = 1 << 30
        i = 0
        t_spin = time.perf_counter()
        # untimed: the --warmup steps, then more until --spinup-ms of wall time have passed (same count on every rank: the
        # count is agreed on below), so that a short timed region starts at the steady state
        while i < warmup:
            step(i)
            i += 1
        torch.cuda.synchronize()
        extra = 0
        while extra < 4096:
            # every rank issues the same number of steps (they contain collectives): rank 0's clock decides, chunk by chunk
            go = (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms
            if multi:
                flag = torch.tensor([1 if go else 0], dtype=torch.int32, device=ctl)
                dist.broadcast(flag, src=0)
                go = bool(flag.item())
            if not go:
                break
            for _ in range(8):
                step(i)
                i += 1
            extra += 8
            torch.cuda.synchronize()
        fence()
        state["total"] = steps
        state["launches"] = 0
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
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        return elapsed, stream_ms, warmup + extra

    n_fly, batch = default_policy(world, multi, path, args.inflight, args.batch)
    step, state, streams = policy(n_fly, batch)
    if n_fly == 1 and not path:
        # scene setup, like the BVH build: the launch order of an unchanged view settles after each of the library's 4 scratch
        # slots has measured the view twice (crt_api.cpp); done here so that any --warmup, even 0, times the settled state
        state["total"] = 12
        for i in range(12):
            step(i)
        fence()
    elapsed, stream_ms, untimed_steps = timed(step, state, streams, args.steps, args.warmup)
    last_frame = state["last"].cpu().numpy().view(np.uint32).reshape(H, W) if rank == 0 else None

    # own-kernel share of this rank, measured per launch with the library's HIP events (render kernel only)
    kms = []
    scratch = torch.zeros(per, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(min(20, args.steps)):
        if not multi:
            kms.append(r.render_frame_device(W, H, frame0.data_ptr(), stats=True)["kernel_ms"])
        else:
            kms.append(r.render_tiles_device(W, H, rank, world, scratch.data_ptr(), stats=True)["kernel_ms"])
    kernel_ms = float(np.median(kms))
    gather_us = None
    if multi:
        # the frame-end exchange of this rank, event-timed: ONE all-gather of the staged tiles + the de-interleave kernel
        gstep, gstate, gstreams = policy(1, 1)
        gstate["total"] = 1 << 30
        gstate["time_gather"] = True
        for i in range(min(20, max(4, args.steps))):
            gstep(i)
        fence()
        gather_us = float(np.median([a.elapsed_time(b) for a, b in gstate["gather_ev"]])) * 1e3
        r.set_stream(main_stream.cuda_stream)

    extras = {}
    if not args.no_extras:
        # the OTHER policy, labelled extra, so that both are on record at every N: at N = 1 `value` is one launch in flight and this leg
        # is the throughput policy (4 launches in flight on alternating streams); at N > 1 `value` is the throughput policy and this
        # leg is one launch in flight, one frame per launch (frame f's exchange beside frame f + 1's render)
        p_fly, p_batch = (1, 1) if multi else (4, 1)
        pstep, pstate, pstreams = policy(p_fly, p_batch)
        p_elapsed, _, _ = timed(pstep, pstate, pstreams, args.steps, max(args.warmup, 8))
        extras["one_launch_in_flight" if multi else "pipelined"] = {
            "launches_in_flight": p_fly, "frames_per_launch": p_batch, "ms_per_frame": p_elapsed / args.steps * 1e3,
            "value": rays_per_frame * args.steps / p_elapsed / 1e6, "unit": "Mray/s",
            "note": "`value` above uses %d launch(es) in flight, %d frame(s) per launch" % (n_fly, batch)}
    if not args.no_extras and not multi and not path:
        # a camera that changes every frame but keeps showing the same picture (it swings +-0.005 degrees about the static view), so
        # that the difference to the static figure is the cost of measuring and sorting the launch order for every frame, as an
        # interactive viewer pays it -- along a real orbit the frame's own cost changes with the view as well (tools/moving_camera.py)
        pos0 = np.float32(cam["position"])
        rot0 = np.float32(cam["matrix"]).reshape(3, 3)

        def cam_at(i):
            a = np.radians(0.005 if i % 2 else -0.005)
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

    if not args.no_extras and not multi and not path:
        # What a rank of an N-GPU job would have to do, measured on this one GPU: rank 0's share of the frame (1/N of the tiles, split
        # packets at their multi-rank default) under the pipelining policy `value` uses at that N, the de-interleave of a gathered
        # buffer of the right size included, the collective itself not.  An estimate of what the partition leaves per rank, labelled as such.
        emu = {}
        for n in (2, 4, 8):
            e_fly, e_batch = default_policy(n, True, False)
            sh = host.rank_share(W, H, 0, n)
            e_per = sh["slots"] * 256
            e_streams = [torch.cuda.Stream() for _ in range(e_fly)]
            e_stage = [torch.zeros(e_batch * e_per, dtype=torch.int32, device="cuda") for _ in range(e_fly)]
            e_gath = [torch.zeros(n * e_batch * e_per, dtype=torch.int32, device="cuda") for _ in range(e_fly)]
            e_out = [torch.zeros(W * H, dtype=torch.int32, device="cuda") for _ in range(e_fly)]
            torch.cuda.synchronize()

            def launch(k):
                st = e_streams[k % e_fly]
                r.set_stream(st.cuda_stream)
                r.render_tiles_batch_device(W, H, 0, n, [e_stage[k % e_fly].data_ptr() + 4 * e_per * f for f in range(e_batch)])
                for f in range(e_batch):
                    r.untile_batch_device(W, H, n, e_batch, f, e_gath[k % e_fly].data_ptr(), e_out[k % e_fly].data_ptr())
            for k in range(24):
                launch(k)
            torch.cuda.synchronize()
            launches = max(8, 200 // e_batch)
            t0 = time.perf_counter()
            for k in range(launches):
                launch(k)
            torch.cuda.synchronize()
            emu[str(n)] = (time.perf_counter() - t0) / (launches * e_batch) * 1e3
        r.set_stream(main_stream.cuda_stream)
        extras["emulated_rank_share"] = {
            "ms_per_frame": emu, "note": "ONE GPU doing rank 0's share of an N-rank frame (tiles + de-interleave, no collective) under the N > 1 "
            "pipelining policy: what the tile partition leaves per rank, not a multi-GPU measurement"}

    ok = True
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_per_frame * args.steps / elapsed / 1e6
        scene_name = {"heightfield": "seeded height field 708x708 quads + ground quad (BASELINE.json configs[2])",
                      "heightfield5m": "seeded height field 1581x1581 quads + ground quad" + (" (BASELINE.json configs[4])" if path else ""),
                      "soup": "seeded soup of 3125 copied icospheres + ground quad"}[cfg["scene"]]
        line = {
            "metric": "Mray/s", "value": value, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %d triangles, %dx%d, %s" % (scene_name, n_tris, W, H, cfg["what"] % len(sc["lights"])),
                       "rays_per_frame": rays_per_frame, "closest_hit_rays": cnt["rays_primary"], "shadow_rays": cnt["rays_shadow"],
                       "frames_in_flight": n_fly * batch, "launches_in_flight": n_fly, "frames_per_launch": batch,
                       "untimed_steps_before_the_timed_region": untimed_steps,
                       "camera": "static" + ("" if path else " (launch order of the unchanged view settled before the timed region); moving camera: moving_camera_ms_per_frame"),
                       "parallelism": "1 GPU, one launch per frame" if world == 1 else "framebuffer tiles 16x16 round-robin over %d GPUs + 1 RCCL all-gather per launch; "
                                      "%d launches in flight on alternating streams, %d frame(s) per launch (one_launch_in_flight: the same with 1 and 1)" % (world, n_fly, batch),
                       "bvh": {"nodes": r.bvh_info()["n_nodes"], "max_depth": r.bvh_info()["max_depth"], "build_and_upload_s": upload_s}},
            "ms_per_frame": ms_per_step,
            "stream_ms_per_step": stream_ms / args.steps,
        }
        line.update(extras)
        if "emulated_rank_share" in extras:
            extras["emulated_rank_share"]["speedup_bound_vs_this_run"] = {n: ms_per_step / v for n, v in extras["emulated_rank_share"]["ms_per_frame"].items()}
        if args.same_device:
            line["rehearsal"] = "every rank on GPU 0, tiles exchanged through host memory over gloo: exercises the rank logic, measures nothing"
        if not multi:
            # dominant (only) kernel.  Its average launch duration = HIP events on its stream around the K back-to-back launches
            # of the timed region / K (with 1 launch in flight the stream holds nothing else: the 1-workgroup sortUnitsKernel
            # that orders a later frame's launch runs on a side stream)
            launch_ms = stream_ms / args.steps if n_fly == 1 else kernel_ms
            key = args.config if not args.scene else "%s:%s" % (args.config, args.scene)
            line["roofline"] = roofline_block(key, cfg["kernel"], launch_ms, kernel_ms, cnt, alg_bytes_frame)
        else:
            line["rank0_render_kernel_ms"] = kernel_ms
            line["rank0_gather_untile_us"] = gather_us
            line["message_bytes_per_rank"] = per * 4
        want_check = not args.no_cpu_baseline or args.check_dist_frame
        if want_check:
            budget = 0.0 if (multi or args.no_cpu_baseline) else (20.0 if path else 10.0)
            base, ref, rows = cpu_baseline(entry.load_oracle(), sc, cfg, budget_s=budget if not path else max(budget, 6.0))
            if not multi and not args.no_cpu_baseline and base:
                line["cpu_baseline"] = base
                line["speedup_vs_cpu_baseline"] = value / base["value"]
            # the timed frame is the frame the oracle renders: RGBA8 identical, pixel for pixel (c5: on the sampled rows)
            ok = bool(np.array_equal(last_frame[rows], ref[rows]))
            line["frame_matches_oracle"] = ok
            line["frame_rows_checked"] = int(len(rows))
            if not ok:
                line["frame_mismatching_pixels"] = int((last_frame[rows] != ref[rows]).sum())
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    r.close()
    if not ok:
        raise SystemExit("bench.py: the timed frame differs from the oracle's frame")


if __name__ == "__main__":
    main()
