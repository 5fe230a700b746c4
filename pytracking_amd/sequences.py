"""Multi-GPU partitioning of the tracking workload.

The hot path shards over independent video sequences: every sequence owns its filter, sample memory and RNG, so
rank r of G simply takes sequences {r, r+G, r+2G, ...} and no data-path collective exists.  The reference does the
same with a process pool (pytracking/evaluation/running.py:198-218).  The only exchange is the end-of-batch gather of
(frames_done, seconds) per rank -- 16 bytes over RCCL/xGMI on the GPU box, gloo in the CPU tests.
"""
import torch


def shard_sequences(num_sequences: int, world_size: int, rank: int):
    """Sequence ids owned by `rank` (round-robin, like the reference's pool hands sequences to workers)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, num_sequences, world_size))


def gather_throughput(frames_done: int, seconds: float, group=None, device="cpu"):
    """All ranks contribute (frames, seconds); every rank gets (total_frames, slowest_seconds, per_rank list).
    Whole-job throughput = total_frames / slowest_seconds (ranks run concurrently)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return int(frames_done), float(seconds), [(int(frames_done), float(seconds))]
    world = dist.get_world_size(group)
    mine = torch.tensor([float(frames_done), float(seconds)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    per_rank = [(int(t[0].item()), float(t[1].item())) for t in out]
    return sum(f for f, _ in per_rank), max(s for _, s in per_rank), per_rank
