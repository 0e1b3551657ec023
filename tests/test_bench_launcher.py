"""bench.py --gpus N without a launcher around it (round 5): N ranks of the same command, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
their environment as under torch.distributed.run, rank 0's stdout passed through, a failing rank ends the others.  On CPU: the
launcher itself (bench.launch_ranks) with a stand-in command, and its ranks doing the C exchange over the mock librccl.so with
the run id the launcher handed them (MASTER_PORT) -- what `python bench.py --gpus 2` does on a GPU node with the real RCCL."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)

RANK = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
out = sys.argv[2]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
if len(sys.argv) > 3 and sys.argv[3] == "fail" and rank == 1:
    sys.exit(7)
if len(sys.argv) > 3 and sys.argv[3] == "gather":
    import numpy as np
    from cmusphinx_amd import lib
    g = lib.Gather(rank, world, os.path.join(out, "rccl-id"), run_id=int(os.environ["S3A_RUN_ID"]))
    recs = []
    for u in range(rank, 10, world):
        h = lib.HypHeader()
        h.utt_index, h.status, h.n_frames, h.score, h.n_words = u, 0, 100 + u, -u, u % 3
        recs.append((h, np.full((u % 3, 6), u, np.int32)))
    got = g.gather(recs, 10)
    rec["gathered"] = [[h.utt_index, h.n_words, int(w.sum())] for h, w in got]
if len(sys.argv) > 3 and sys.argv[3] == "fail" and rank != 1:
    import time; time.sleep(60)
json.dump(rec, open(os.path.join(out, f"rank{rank}.json"), "w"))
if rank == 0:
    print(json.dumps({"n_gpus": world}))
'''


def test_launcher_starts_n_ranks_with_the_launchers_environment(tmp_path, capfd):
    import bench
    rc = bench.launch_ranks(3, [sys.executable, "-c", RANK, ROOT, str(tmp_path)])
    assert rc == 0
    recs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(3)]
    assert [r["RANK"] for r in recs] == ["0", "1", "2"] and [r["LOCAL_RANK"] for r in recs] == ["0", "1", "2"]
    assert {r["WORLD_SIZE"] for r in recs} == {"3"} and {r["MASTER_ADDR"] for r in recs} == {"127.0.0.1"}
    assert len({r["MASTER_PORT"] for r in recs}) == 1 and int(recs[0]["MASTER_PORT"]) > 0
    out = capfd.readouterr().out
    assert out.count('"n_gpus": 3') == 1                    # rank 0's line, once


def test_a_failing_rank_ends_the_run(tmp_path):
    import bench
    import time
    t0 = time.time()
    rc = bench.launch_ranks(3, [sys.executable, "-c", RANK, ROOT, str(tmp_path), "fail"])
    assert rc == 7 and time.time() - t0 < 30               # the sleeping ranks were not waited for


def test_launched_ranks_exchange_over_the_mock_rccl(tmp_path):
    mock = tmp_path / "mock"
    mock.mkdir()
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", str(mock / "librccl.so"), os.path.join(ROOT, "tests", "mock_rccl.c")], check=True)
    import bench
    old = {k: os.environ.get(k) for k in ("LD_LIBRARY_PATH", "MOCK_RCCL_DIR")}
    os.environ["LD_LIBRARY_PATH"] = f"{mock}:" + os.environ.get("LD_LIBRARY_PATH", "")
    os.environ["MOCK_RCCL_DIR"] = str(tmp_path)
    try:
        rc = bench.launch_ranks(2, [sys.executable, "-c", RANK, ROOT, str(tmp_path), "gather"])
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert rc == 0
    want = [[u, u % 3, u * 6 * (u % 3)] for u in range(10)]
    for r in range(2):
        assert json.load(open(tmp_path / f"rank{r}.json"))["gathered"] == want


def test_world_size_must_agree_with_gpus(tmp_path):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(os.environ, WORLD_SIZE="4", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=4" in (p.stderr + p.stdout)
