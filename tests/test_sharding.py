"""N>1 path on CPU: utterance sharding + the ONE end-of-batch gather of hypothesis records (s3a_hyp_record_t,
packed and formatted by the C ABI), world_size 2, gloo; the -hyp / -hypseg files rank 0 writes from the gathered
records must be byte-identical to what one process writes."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

from cmusphinx_amd import lib, shard

WORDS = ["<s>", "</s>", "<sil>", "ONE", "TWO", "TWO(2)", "THREE", "++NOISE++"]
BASE = np.array([0, 1, 2, 3, 4, 4, 6, 7], np.int32)
FILL = np.array([1, 1, 1, 0, 0, 0, 0, 1], np.uint8)


def fmt(rec):
    L = lib.load()
    ws = (C.c_char_p * len(WORDS))(*[w.encode() for w in WORDS])
    m, s = C.create_string_buffer(1 << 14), C.create_string_buffer(1 << 14)
    lib.check(L.s3a_hyp_format(C.byref(rec), ws, lib._p(BASE), lib._p(FILL), 0, 1, np.float32(9.5), -3567, 0, m, len(m), s, len(s)), L)
    return m.value.decode(), s.value.decode()


def make_record(u):
    """a deterministic fake hypothesis for utterance u: <sil> w w ... </s>"""
    rng = np.random.default_rng(1000 + u)
    r = lib.HypRecord()
    r.uttid = f"utt{u:04d}".encode()
    r.utt_index, r.n_frames, r.status = u, 100 + 7 * u, 0
    n = 2 + int(rng.integers(1, 6))
    t = 0
    for k in range(n - 1):
        w = r.word[k]
        w.wid = 2 if k == 0 else int(rng.integers(3, 7))
        w.sf, w.ef = t, t + int(rng.integers(5, 20))
        t = w.ef + 1
        w.ascr, w.lscr, w.scale = -int(rng.integers(1000, 90000)), -int(rng.integers(100, 60000)), int(rng.integers(1, 999))
    r.word[n - 2].ef = r.n_frames - 1
    e = r.word[n - 1]
    e.wid, e.sf, e.ef, e.lscr = 1, r.n_frames, r.n_frames, -4242
    r.n_words, r.score, r.total_scale = n, -123456 - u, 31337 + u
    return r


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


def test_record_format_follows_match_write_and_matchseg_write():
    """srch_output.c:74-161: fillers / <s> / </s> and zero-width words dropped from -hyp, base word strings there and
    full word strings in -hypseg, lm_rawscore = (lscr - wip) / lw truncated, S = the frame normalisers' sum."""
    r = lib.HypRecord()
    r.uttid, r.n_words, r.n_frames, r.total_scale = b"u1", 4, 50, 777
    for k, (wid, sf, ef, a, l) in enumerate([(2, 0, 9, -1000, -3567), (5, 10, 30, -2000, -13067), (7, 31, 49, -300, -5000), (1, 50, 50, 0, -99)]):
        w = r.word[k]
        w.wid, w.sf, w.ef, w.ascr, w.lscr, w.scale = wid, sf, ef, a, l, 11
    m, s = fmt(r)
    assert m == "TWO (u1)\n"
    raw = [int(np.float32(x + 3567) / np.float32(9.5)) for x in (-3567, -13067, -5000)]
    assert s == f"u1 S 777 T {-3300 + sum(raw)} A -3300 L {sum(raw)} 0 -1000 {raw[0]} <sil> 10 -2000 {raw[1]} TWO(2) 31 -300 {raw[2]} ++NOISE++ 50\n"
    empty = lib.HypRecord()
    empty.uttid = b"e"
    assert fmt(empty)[0] == "(null) (e)\n"


def _worker(rank, world, port, n_utt, outdir, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.shard_contiguous(n_utt, rank, world)
    recs = shard.gather_records([make_record(u) for u in mine], n_utt, dist)
    if rank == 0:
        shard.write_outputs(recs, fmt, os.path.join(outdir, "w2.match"), os.path.join(outdir, "w2.matchseg"))
        q.put([r.utt_index for r in recs])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_utt", [5, 8, 1])
def test_gather_two_ranks_gloo_writes_the_single_process_files(n_utt, tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, n_utt, str(tmp_path), q)) for r in range(2)]
    for p in ps:
        p.start()
    got = q.get(timeout=120)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == list(range(n_utt))
    one = shard.gather_records([make_record(u) for u in range(n_utt)], n_utt)      # one process, no collective
    shard.write_outputs(one, fmt, str(tmp_path / "w1.match"), str(tmp_path / "w1.matchseg"))
    for e in ("match", "matchseg"):
        assert open(tmp_path / f"w1.{e}").read() == open(tmp_path / f"w2.{e}").read()
    assert open(tmp_path / "w1.match").read().count("\n") == n_utt


def test_gather_refuses_missing_and_overflowing_records():
    with pytest.raises(RuntimeError, match="missing or duplicated"):
        shard.gather_records([make_record(0), make_record(2)], 3)
    r = make_record(0)
    r.status = -3
    with pytest.raises(RuntimeError, match="exceed S3A_HYP_MAXW"):
        shard.gather_records([r], 1)


def test_failed_utterances_get_no_line_as_in_the_reference():
    """srch.c:495-498: when utt_end fails (no word exit, a decode error) the reference writes neither a -hyp nor a
    -hypseg line for the utterance; the gathered files must do the same"""
    recs = [make_record(u) for u in range(4)]
    recs[1].status, recs[1].n_words = -2, 0
    recs[3].status, recs[3].n_words = -1, 0
    got = shard.gather_records(recs, 4)
    log = []
    lines = shard.write_outputs(got, fmt, log=log)
    assert [l[0] for l in lines] == [fmt(make_record(0))[0], fmt(make_record(2))[0]]
    assert [(i, s) for i, _, s in log] == [(1, -2), (3, -1)]


def _hdr_words(u, n_words=None):
    r = make_record(u)
    h = lib.HypHeader.from_buffer_copy(bytes(r)[:C.sizeof(lib.HypHeader)])
    w = np.frombuffer(bytes(r)[C.sizeof(lib.HypHeader):], np.int32).reshape(-1, 6)[:r.n_words].copy()
    if n_words is not None:                 # a long hypothesis: repeat the middle words
        mid = np.tile(w[1:-1], (n_words // max(len(w) - 2, 1) + 1, 1))[:n_words - 2]
        w = np.concatenate([w[:1], mid, w[-1:]])
        h.n_words = len(w)
    return h, w


def fmt_var(h, w):
    L = lib.load()
    ws = (C.c_char_p * len(WORDS))(*[x.encode() for x in WORDS])
    m, s = C.create_string_buffer(1 << 20), C.create_string_buffer(1 << 20)
    w = np.ascontiguousarray(w, np.int32)
    lib.check(L.s3a_hyp_format_var(C.byref(h), lib._p(w), ws, lib._p(BASE), lib._p(FILL), 0, 1, np.float32(9.5), -3567, 0, m, len(m), s, len(s)), L)
    return m.value.decode(), s.value.decode()


def _worker_var(rank, world, port, lens, outdir, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.shard_contiguous(len(lens), rank, world)
    got = shard.gather_var([_hdr_words(u, lens[u]) for u in mine], len(lens), dist)
    if rank == 0:
        shard.write_outputs(got, fmt_var, os.path.join(outdir, "v2.match"), os.path.join(outdir, "v2.matchseg"))
        q.put([int(h.n_words) for h, _ in got])
    dist.barrier()
    dist.destroy_process_group()


def test_variable_length_gather_has_no_word_limit(tmp_path):
    """lengths gather + padded payload: hypotheses far beyond S3A_HYP_MAXW words cross the ranks intact, and what fits
    the fixed record formats to the same lines either way"""
    import torch.multiprocessing as mp
    lens = [None, 1200, None, 3, 4000]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_var, args=(r, 2, port, lens, str(tmp_path), q)) for r in range(2)]
    for p in ps:
        p.start()
    got = q.get(timeout=120)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[1] == 1200 and got[4] == 4000 and got[3] == 3
    one = shard.gather_var([_hdr_words(u, lens[u]) for u in range(len(lens))], len(lens))
    shard.write_outputs(one, fmt_var, str(tmp_path / "v1.match"), str(tmp_path / "v1.matchseg"))
    for e in ("match", "matchseg"):
        assert open(tmp_path / f"v1.{e}").read() == open(tmp_path / f"v2.{e}").read()
    assert fmt_var(*_hdr_words(0)) == fmt(make_record(0)) and fmt_var(*_hdr_words(2)) == fmt(make_record(2))
