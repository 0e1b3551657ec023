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


def test_two_ranks_share_the_gpu_and_gather_their_hypotheses(tmp_path):
    env = dict(os.environ, S3A_BENCH_ONE_GPU="1", TMPDIR=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--lanes", "6", "--engines", "2", "--frames", "150", "--utts", "10", "--no-cpu", "--no-scoring"]
    p = subprocess.run(cmd, capture_output=True, text=True, errors="ignore", timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["identical_to_reference"] is True
    assert d["config"]["lanes"] == 6 and d["config"]["engines"] == 2
    # 2 ranks x 2 steps x 6 lanes utterances (the task's sentences run a little past the nominal 150 frames), all of
    # them gathered on rank 0: value x time = the frames of BOTH ranks
    frames = d["value"] * d["ms_per_step"] * 1e-3 * d["steps"]
    assert 2 * 2 * 6 * 100 < frames < 2 * 2 * 6 * 400
    assert "roofline" in d and "kernels" in d
