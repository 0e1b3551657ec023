"""The C exchange with world > 1, on CPU: s3a_gather_init (file rendezvous) + s3a_gather_hyps (three all-gathers: counts,
headers padded to the largest rank, words padded to the largest rank; ordering by utterance index) run in 2 and 3 processes
against tests/mock_rccl.c, a stand-in librccl.so whose collectives go through files and take host pointers.  What the real
RCCL run on N GPUs does with device buffers is this arithmetic; on a one-GPU box only world = 1 can be exercised."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

WORKER = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from cmusphinx_amd import lib
rank, world, rdv, n_total, drop = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
run_id = int(sys.argv[7]) if len(sys.argv) > 7 else 0
if len(sys.argv) > 8 and rank > 0:
    import time; time.sleep(float(sys.argv[8]))           # a rank born late
g = lib.Gather(rank, world, rdv, run_id=run_id)
for rnd in range(2):                       # twice: the staging buffers are reused (and must grow in round 2)
    recs = []
    for u in range(n_total):
        if u % world != (rank + rnd) % world:
            continue
        if drop and rank == world - 1:
            continue                       # a rank that sends nothing although it was given a share
        n = (u * 7 + rnd * 13) % 11 * (1 + 5 * rnd)
        h = lib.HypHeader()
        h.utt_index, h.status, h.n_frames, h.score, h.n_words = u, (1 if u % 5 == 4 else 0), 100 + u, -1000 * u - rnd, n
        w = (np.arange(n * 6, dtype=np.int32).reshape(n, 6) + 1000 * u + rnd) if h.status == 0 else np.zeros((0, 6), np.int32)
        recs.append((h, w))
    try:
        out = g.gather(recs, n_total)
        res = [[h.utt_index, h.status, h.n_frames, h.score, h.n_words, w.ravel().tolist()] for h, w in out]
        err = None
    except lib.S3AError as e:
        res, err = None, str(e)
    if rank == 0:
        print(json.dumps({"round": rnd, "res": res, "err": err}))
'''


@pytest.fixture(scope="module")
def mock_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("mockrccl")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", str(d / "librccl.so"), os.path.join(ROOT, "tests", "mock_rccl.c")], check=True)
    return d


def run_world(mock_dir, tmp_path, world, n_total, drop=0, run_id=0, late=0.0):
    rdv = str(tmp_path / "rccl-id")
    env = dict(os.environ, LD_LIBRARY_PATH=f"{mock_dir}:" + os.environ.get("LD_LIBRARY_PATH", ""), MOCK_RCCL_DIR=str(tmp_path))
    extra = [str(run_id)] + ([str(late)] if late else [])
    ps = [subprocess.Popen([sys.executable, "-c", WORKER, ROOT, str(r), str(world), rdv, str(n_total), str(drop)] + extra, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=120) for p in ps]
    assert all(p.returncode == 0 for p in ps), "\n".join(o[1][-800:] for o in outs)
    return [json.loads(l) for l in outs[0][0].splitlines() if l.startswith("{")]


def expected(n_total, rnd):
    out = []
    for u in range(n_total):
        n = (u * 7 + rnd * 13) % 11 * (1 + 5 * rnd)
        st = 1 if u % 5 == 4 else 0
        w = (np.arange(n * 6, dtype=np.int32) + 1000 * u + rnd).tolist() if st == 0 else []
        out.append([u, st, 100 + u, -1000 * u - rnd, n, w])
    return out


@pytest.mark.parametrize("world,n_total", [(2, 23), (3, 31), (4, 5)])
def test_ranks_exchange_ragged_records(mock_dir, tmp_path, world, n_total):
    got = run_world(mock_dir, tmp_path, world, n_total)
    assert [g["round"] for g in got] == [0, 1]
    for g in got:
        assert g["err"] is None
        assert g["res"] == expected(n_total, g["round"])


def test_missing_utterances_are_reported_not_invented(mock_dir, tmp_path):
    """three ranks, one sends nothing although the control file gave it a share: the count check must fail loudly"""
    got = run_world(mock_dir, tmp_path, 3, 12, drop=1)
    for g in got:
        assert g["res"] is None and ("missing or duplicated" in g["err"] or "expected" in g["err"])


def test_stale_rendezvous_file_is_ignored(mock_dir, tmp_path):
    rdv = tmp_path / "rccl-id"
    rdv.write_bytes(b"x" * 128)
    os.utime(rdv, (1000000000, 1000000000))           # left behind by a run long ago
    got = run_world(mock_dir, tmp_path, 2, 9)
    assert got[0]["err"] is None and got[0]["res"] == expected(9, 0)


def test_run_id_rendezvous_compares_no_clocks(mock_dir, tmp_path):
    """s3a_gather_init_run: a rank born seconds after rank 0 wrote the file still takes it (the age test of the id-less form would
    have been at its limit with the 1 s window of round 4); a file of ANOTHER run id next to it, and an old file under the run's own
    name (rank 0 removes it before it writes), are not mistaken for it"""
    (tmp_path / "rccl-id.00000000000004d2").write_bytes(b"y" * 136)          # run 1234's, left behind
    (tmp_path / ("rccl-id.%016x" % 777)).write_bytes(b"z" * 136)             # this run's name, a crashed earlier attempt
    os.utime(tmp_path / ("rccl-id.%016x" % 777), (1000000000, 1000000000))
    got = run_world(mock_dir, tmp_path, 2, 11, run_id=777, late=2.5)
    assert got[0]["err"] is None and got[0]["res"] == expected(11, 0)
    assert got[1]["err"] is None and got[1]["res"] == expected(11, 1)


def test_the_rendezvous_file_does_not_outlive_the_init(mock_dir, tmp_path):
    """rank 0 removes the file once every rank holds the id (the init's own all-gather is the proof), so the SAME command run again
    under the SAME run id -- a launcher that reuses its port as the id -- cannot meet the first run's dead communicator id"""
    for attempt in range(2):
        got = run_world(mock_dir, tmp_path, 2, 7, run_id=29500)
        assert got[0]["err"] is None and got[0]["res"] == expected(7, 0), attempt
        assert got[1]["err"] is None and got[1]["res"] == expected(7, 1), attempt
        assert not (tmp_path / ("rccl-id.%016x" % 29500)).exists()
    got = run_world(mock_dir, tmp_path, 2, 5)              # (the id-less form too)
    assert got[0]["err"] is None and not (tmp_path / "rccl-id").exists()
