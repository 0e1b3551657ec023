"""N>1 path on CPU: utterance sharding + the one end-of-batch gather, world_size 2, gloo."""
import os
import socket

import numpy as np
import pytest

from cmusphinx_amd import shard


def test_contiguous_shards_cover_exactly_once():
    for n in (0, 1, 5, 31, 100, 1024):
        for w in (1, 2, 3, 4, 8):
            parts = [shard.shard_contiguous(n, r, w) for r in range(w)]
            flat = [i for p in parts for i in p]
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_frame_balanced_shards():
    rng = np.random.default_rng(1)
    frames = rng.integers(50, 1500, 200)
    for w in (2, 4, 8):
        parts = [shard.shard_by_frames(frames, r, w) for r in range(w)]
        assert sorted(i for p in parts for i in p) == list(range(200))
        loads = [int(frames[p].sum()) for p in parts]
        assert max(loads) - min(loads) <= frames.max()


def test_record_roundtrip():
    r = shard.pack_record(7, 345, -1234567, [3, 1, 4, 1, 5])
    d = shard.unpack_record(r)
    assert d == dict(utt=7, n_frames=345, score=-1234567, words=[3, 1, 4, 1, 5])
    long = shard.unpack_record(shard.pack_record(0, 1, 0, range(200)))
    assert len(long["words"]) == shard.REC_WORDS


def _worker(rank, world, port, n_utt, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.shard_contiguous(n_utt, rank, world)
    recs = [shard.pack_record(u, 100 + u, -1000 * u, [u, u + 1, rank]) for u in mine]
    got = shard.gather_records(recs, n_utt, dist)
    if rank == 0:
        q.put(got)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_utt", [5, 8, 1])
def test_gather_two_ranks_gloo(n_utt):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, n_utt, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = q.get(timeout=120)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [d["utt"] for d in got] == list(range(n_utt))
    for d in got:
        assert d["n_frames"] == 100 + d["utt"] and d["score"] == -1000 * d["utt"]
        assert d["words"][:2] == [d["utt"], d["utt"] + 1]
    # contiguous split: rank of each utterance as recorded in the third word
    split = shard.shard_contiguous(n_utt, 0, 2)
    assert all((d["words"][2] == 0) == (d["utt"] in split) for d in got)
