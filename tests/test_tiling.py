"""Tile partition of the framebuffer over N ranks and the frame-end gather.  CPU tests:
the layout contract in numpy, and a world_size-2 gloo run in which each rank renders only its own tiles
(with the CPU oracle standing in for the GPU) and one all_gather + de-interleave reproduces the full frame."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("w,h,n", [(1920, 1080, 8), (1920, 1080, 3), (33, 17, 2), (16, 16, 4), (1, 1, 1), (100, 50, 7)])
def test_tile_untile_roundtrip(pkg, w, h, n):
    rng = np.random.default_rng(w * 31 + h)
    frame = rng.integers(0, 2 ** 32, size=(h, w), dtype=np.uint32)
    slots = pkg.tile_slots(w, h, n)
    assert slots == pkg.lib().crt_tile_slots(w, h, n) and pkg.tile_count(w, h) == pkg.lib().crt_tile_count(w, h)
    staged = np.stack([pkg.tile_host(frame, w, h, r, n) for r in range(n)])
    assert staged.shape == (n, slots, 16, 16)
    np.testing.assert_array_equal(pkg.untile_host(staged, w, h, n), frame)
    # ownership: macro tile k belongs to rank k % n (interleaved for load balance)
    tx = (w + 15) // 16
    k = 1 % pkg.tile_count(w, h)
    r, s = k % n, k // n
    ty0, tx0 = (k // tx) * 16, (k % tx) * 16
    blk = np.zeros((16, 16), np.uint32)
    sub = frame[ty0:ty0 + 16, tx0:tx0 + 16]
    blk[:sub.shape[0], :sub.shape[1]] = sub
    np.testing.assert_array_equal(staged[r, s], blk)


def test_1080p_message_sizes(pkg):
    """1080p has 120 x 68 macro tiles (last row half used); per-rank message at 8 GPUs."""
    assert pkg.tile_count(1920, 1080) == 8160
    assert pkg.tile_slots(1920, 1080, 8) * 1024 == 1044480  # bytes per rank, vs 1 036 800 unpadded


def test_world_size_2_gloo_gather():
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533",
                          os.path.join(ROOT, "tests", "dist_tile_worker.py")],
                         capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "TILE_GATHER_OK" in out.stdout


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher environment (how the driver calls it): the parent must start two ranks with
    torch.distributed.run as a child process before anything touches a GPU, and relay the rank-0 line and the exit code.
    Here (no GPU) the ranks only rendezvous -- over gloo -- add up their ranks and leave."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only", "--backend", "gloo"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(env, OMP_NUM_THREADS="2"))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"rendezvous_ok": True, "n_gpus": 2, "backend": "gloo"}
    # a rank that fails takes the whole launch down with a non-zero exit code
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(env, OMP_NUM_THREADS="2"))
    assert bad.returncode != 0  # (no GPU here: "bench.py needs an MI355X")


def test_bench_pipelining_policy_defaults():
    """bench.py: one launch in flight on one GPU (the roofline block times the kernel itself); four in flight on N > 1 ranks, four
    frames per launch from 8 ranks up (not for path tracing, whose share of a frame is milliseconds); flags override."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.default_policy(1, False, False) == (1, 1)
    assert b.default_policy(1, True, False) == (4, 1)       # --force-dist with the one rank a box has
    assert b.default_policy(2, True, False) == (4, 1) and b.default_policy(4, True, False) == (4, 1)
    assert b.default_policy(8, True, False) == (4, 4) and b.default_policy(8, True, True) == (4, 1)
    assert b.default_policy(8, True, False, inflight=1, batch=1) == (1, 1)
    assert b.default_policy(8, True, False, batch=9) == (4, 4) and b.default_policy(1, False, False, inflight=4, batch=3) == (4, 1)
