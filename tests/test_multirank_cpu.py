"""CPU suite, part 3: the N>1 path on gloo with world size 2 (the GPU runs use the same code
over RCCL).  The per-rank "kernel" here is the oracle -- a CPU stand-in, allowed in tests only --
so the test checks exactly what multi-GPU adds: disjoint covering shards, no dependence of a
stream's result on the rank that owned it, and the MAX/SUM reduction bench.py reports."""
import os
import sys
import zlib

import numpy as np
import pytest

from conftest import ROOT, load_blob
from rnnoise_amd import dist as rdist
from rnnoise_amd import synth

TOTAL, T = 5, 6


def _crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle.binding import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blob = load_blob("default")
    mine = rdist.shard_streams(TOTAL, world, rank)
    res = {}
    for s in mine:
        pcm = synth.stream_pcm(s, T).astype(np.float32).reshape(T, 480)
        res[s] = _crc(Oracle(blob).run(pcm)["out"])
    frames, elapsed = rdist.aggregate_throughput(len(mine) * T, 1.0 + rank, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, list(mine), frames, elapsed, gathered))


def test_shards_partition_the_streams():
    for total in (1, 7, 8, 4096, 524288):
        for world in (1, 2, 3, 8):
            parts = [rdist.shard_streams(total, world, r) for r in range(world)]
            flat = [s for p in parts for s in p]
            assert flat == list(range(total))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort()
    assert out[0][1] + out[1][1] == list(range(TOTAL))
    for rank, mine, frames, elapsed, gathered in out:
        assert frames == TOTAL * T and elapsed == 2.0  # SUM of frames, MAX of elapsed
    merged = {}
    for d in out[0][4]:
        merged.update(d)
    from oracle.binding import Oracle
    blob = load_blob("default")
    for s in range(TOTAL):  # same bits whichever rank owned the stream
        pcm = synth.stream_pcm(s, T).astype(np.float32).reshape(T, 480)
        assert merged[s] == _crc(Oracle(blob).run(pcm)["out"])
