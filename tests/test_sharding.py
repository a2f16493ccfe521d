"""Frame sharding across ranks (no data-path collective), covered on CPU
with a real 2-process gloo group."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shards_partition_the_frames():
    from x265_amd.sharding import shard_frames
    for nframes in (1, 2, 9, 17, 33, 100):
        for world in (1, 2, 3, 4, 8):
            for seg in (1, 4):
                for nrefs in (1, 4):
                    owned = []
                    for r in range(world):
                        s = shard_frames(nframes, world, r, nrefs=nrefs, segment=seg)
                        owned += list(range(s.first, s.first + s.count))
                        assert s.ref_first == max(s.first - nrefs, 0)
                        if s.count and r < world - 1:
                            assert s.count % seg == 0 or s.first + s.count == nframes
                    assert owned == list(range(1, nframes)), (nframes, world, seg)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import pyoracle
    from x265_amd import synth
    from x265_amd.sharding import shard_frames
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank derives the same sequence and encodes only its own frames (CPU oracle
        # stands in for the kernels here: this test is about WHO does WHAT, not the arithmetic)
        seq = synth.make_sequence(128, 64, 7, search_range=8, seed=3)
        sh = shard_frames(len(seq), world, rank)
        orc = pyoracle.Oracle()
        mine = np.zeros((len(seq), 2), pyoracle.MV_DTYPE)
        for f in range(sh.first, sh.first + sh.count):
            mine[f] = orc.me_fullsearch(seq[f], seq[f - 1], 64, 8)
        # gather the small results (motion vectors) -- the only cross-rank traffic of the design
        t = torch.from_numpy(mine.view(np.int32).reshape(len(seq), 4).copy())
        dist.all_reduce(t)  # disjoint ownership => sum == concatenation
        dist.barrier()
        if rank == 0:
            q.put(t.numpy().copy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_run_equals_single_process():
    import torch.multiprocessing as mp
    from oracle import pyoracle
    from x265_amd import synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seq = synth.make_sequence(128, 64, 7, search_range=8, seed=3)
    orc = pyoracle.Oracle()
    want = np.zeros((len(seq), 2), pyoracle.MV_DTYPE)
    for f in range(1, len(seq)):
        want[f] = orc.me_fullsearch(seq[f], seq[f - 1], 64, 8)
    np.testing.assert_array_equal(got, want.view(np.int32).reshape(len(seq), 4))
