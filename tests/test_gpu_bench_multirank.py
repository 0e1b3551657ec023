"""The multi-rank flow of bench.py rehearsed on ONE GPU (MI355X): two ranks under torch.distributed.run, both on GPU 0
(S3A_BENCH_ONE_GPU=1: the end-of-batch gather goes over gloo, because RCCL cannot put two ranks on one device) -- the
utterance sharding over ranks, the barriers around the timed region, the maximum over ranks, the ONE gather of
fixed-size hypothesis records, rank 0 checking that every utterance index arrived exactly once and printing ONE JSON
line.  On an 8-GPU node the same code runs with the nccl (= RCCL) backend and one GPU per rank."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(tmp_path, port, extra):
    env = dict(os.environ, S3A_BENCH_ONE_GPU="1", TMPDIR=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--lanes", "6", "--engines", "2", "--min-group", "2", "--frames", "150", "--utts", "10", "--no-scoring"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, errors="ignore", timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                # rank 0 only
    return json.loads(lines[0])


def test_two_ranks_split_the_fixed_batch_and_gather_their_hypotheses(tmp_path):
    """strong scaling (configs[3] as written): ONE 10-utterance control list, rank r decodes its contiguous shard
    (shard.shard_contiguous = -ctloffset/-ctlcount) inside the timed region, the hypotheses cross the ranks in two
    collectives (headers, then the padded words), and rank 0 compares EVERY utterance of the batch -- both ranks' --
    with the unmodified reference decoder run over the whole list (the CPU baseline's batch leg, 3 processes)."""
    d = run_bench(tmp_path, 29541, ["--cpu-procs", "3"])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong"
    assert d["identical_to_reference"] == {"hyp": True, "hypseg": True} and d["utterances_checked_against_reference"] == 10
    assert d["config"]["utterances_per_step"] == 10 and d["config"]["lanes_per_gpu"] == 6 and d["config"]["engines_per_gpu"] == 2
    assert sum(d["config"]["groups_rank0"]) == 5             # rank 0's share of the 10
    # value x time = the frames of the whole batch, once per step
    frames = d["value"] * d["ms_per_step"] * 1e-3 * d["steps"]
    assert abs(frames - 2 * d["config"]["frames_per_step"]) < 1e-3 * frames and 10 * 100 < d["config"]["frames_per_step"] < 10 * 400
    assert d["cpu_baseline"]["cores"] == 3 and d["cpu_baseline"]["utterances"] == 10        # (3 processes x 8 >= the batch)
    assert set(d["cpu_baseline"]["single_core_split_xCPU"]) == {"sen", "search", "tot"}
    assert d["weak_scaling"]["utterances"] == 20 and d["weak_scaling"]["value"] > 0
    assert d["roofline"]["frac"] <= 1.0 and "kernels" in d


def test_two_ranks_weak_scaling(tmp_path):
    d = run_bench(tmp_path, 29543, ["--scaling", "weak", "--no-cpu"])
    assert d["scaling"] == "weak" and d["config"]["utterances_per_step"] == 20
    assert d["identical_to_reference"]["hyp"] is True and d["utterances_checked_against_reference"] == 2


def test_bench_starts_its_ranks_itself(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (round 5): the command re-executes itself as two ranks (here both on
    GPU 0: S3A_BENCH_ONE_GPU=1) and rank 0 prints the ONE line with n_gpus = 2"""
    env = dict(os.environ, S3A_BENCH_ONE_GPU="1", TMPDIR=str(tmp_path))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--lanes", "6", "--engines", "2",
           "--min-group", "2", "--frames", "150", "--utts", "10", "--no-scoring", "--no-cpu", "--no-ps", "--no-wide-beam"]
    p = subprocess.run(cmd, capture_output=True, text=True, errors="ignore", timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rank_devices"] == [0, 0] and d["identical_to_reference"]["hyp"] is True
    assert d["config"]["utterances_per_step"] == 10
