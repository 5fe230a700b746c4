"""N>1 path on CPU: world_size-2 gloo run of the sequence sharding + end-of-batch throughput gather that bench.py
uses on the GPU box with RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pytracking_amd.sequences import gather_throughput, shard_sequences


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seqs = shard_sequences(5, world, rank)
    frames = 100 * len(seqs)
    secs = 1.0 + rank
    tot, slow, per = gather_throughput(frames, secs)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, seqs, tot, slow, per))


def test_shard_is_a_partition():
    for n in (0, 1, 7, 8, 50):
        for world in (1, 2, 8):
            got = sorted(s for r in range(world) for s in shard_sequences(n, world, r))
            assert got == list(range(n))
            sizes = [len(shard_sequences(n, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_gather_is_identity():
    assert gather_throughput(10, 2.0) == (10, 2.0, [(10, 2.0)])


def test_two_rank_gloo_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    for _, _, tot, slow, per in res:
        assert tot == 500 and slow == 2.0 and per == [(300, 1.0), (200, 2.0)]
