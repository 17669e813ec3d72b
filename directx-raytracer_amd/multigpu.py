"""Multi-GPU frame assembly: one process per GPU, framebuffer cut into 16x16 macro tiles dealt round-robin to the
ranks (tile k -> rank k % N), ONE collective per frame (the reference is single GPU only).

`torch.distributed` backend "nccl" is RCCL on ROCm; the all-gather moves crt_tile_slots()*1024 bytes per rank
(1.04 MB at 1080p / 8 GPUs) over xGMI.  The scene is replicated; there is no exchange during the frame.
The same function runs over gloo on CPU tensors (tests/test_tiling.py)."""
import torch
import torch.distributed as dist

TILE = 16


def rank_share(w, h, rank, n_ranks):
    tiles = ((w + TILE - 1) // TILE) * ((h + TILE - 1) // TILE)
    slots = (tiles + n_ranks - 1) // n_ranks
    mine = len(range(rank, tiles, n_ranks))
    return {"tiles": mine, "slots": slots, "staging_bytes": slots * TILE * TILE * 4}


def gather_frame(staging, w, h, untile, gathered=None):
    """staging: this rank's tile-major int32 tensor (slots*256 elements, device or CPU).
    Returns the row-major frame produced by `untile(gathered)`; every rank gets the frame (all-gather)."""
    n = dist.get_world_size() if dist.is_initialized() else 1
    if gathered is None:
        gathered = torch.empty(n * staging.numel(), dtype=staging.dtype, device=staging.device)
    if not dist.is_initialized():
        gathered.copy_(staging)
    else:
        dist.all_gather_into_tensor(gathered, staging)  # also with one rank: same code path as N > 1
    return untile(gathered)


def gather_batch(staging, w, h, n_frames, untile, gathered=None):
    """A batch of frames rendered by one launch (crt_render_tiles_batch_device) travels in ONE all-gather: `staging` is this
    rank's n_frames staging buffers back to back (n_frames*slots*256 elements), the gathered tensor holds
    [rank][frame][slot] tiles, and `untile(gathered, f)` rebuilds frame f (crt_untile_batch_device).  Returns the frames."""
    n = dist.get_world_size() if dist.is_initialized() else 1
    if gathered is None:
        gathered = torch.empty(n * staging.numel(), dtype=staging.dtype, device=staging.device)
    if not dist.is_initialized():
        gathered.copy_(staging)
    elif dist.get_backend() == "gloo" and staging.is_cuda:
        # rehearsal on one GPU (bench.py --backend gloo --same-device): the tiles travel through host memory
        host = torch.empty(n * staging.numel(), dtype=staging.dtype)
        dist.all_gather_into_tensor(host, staging.cpu())
        gathered.copy_(host)
    else:
        dist.all_gather_into_tensor(gathered, staging)
    return [untile(gathered, f) for f in range(n_frames)]
