"""s3a_gather_*: the end-of-batch exchange of hypothesis records over RCCL, in C (MI355X, one rank -- what a one-GPU box
allows; the multi-rank arithmetic -- padding to the largest rank, utterance order, failed utterances -- is the gloo tests'
of tests/test_sharding.py, whose Python implementation follows the same three steps)."""
import numpy as np
import pytest

from cmusphinx_amd import lib

pytestmark = pytest.mark.gpu


def test_one_rank_gather_runs_rccl_and_returns_the_records_in_utterance_order(gpu_lib):
    rng = np.random.default_rng(5)
    local = []
    for u in (3, 0, 2, 1, 4):                        # (a rank hands its records over in any order)
        h = gpu_lib.HypHeader()
        h.uttid = f"utt{u}".encode()
        h.utt_index, h.n_frames, h.score = u, 100 + u, -1000 * u
        n = [7, 0, 1500, 3, 1][u]                    # one beyond S3A_HYP_MAXW, one that could not be ended
        h.status, h.n_words = (-2, 0) if n == 0 else (0, n)
        local.append((h, rng.integers(-10**6, 10**6, (n, 6)).astype(np.int32)))
    g = gpu_lib.Gather(0, 1)
    out = g.gather(local, 5)
    assert [h.utt_index for h, _ in out] == [0, 1, 2, 3, 4]
    by = {h.utt_index: (h, w) for h, w in local}
    for h, w in out:
        assert h.uttid == by[h.utt_index][0].uttid and h.status == by[h.utt_index][0].status and h.n_frames == 100 + h.utt_index
        assert np.array_equal(w, by[h.utt_index][1])
    with pytest.raises(gpu_lib.S3AError, match="expected|missing"):
        g.gather(local[:3], 5)
