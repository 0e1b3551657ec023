"""The lane-refill schedule of s3a_uttdec_decode_queue (s3a_queue_schedule: host arithmetic, runs without a GPU): every
utterance gets a lane and a first engine frame; lanes never overlap, refills happen at window boundaries only, the queue's
order is kept, and no lane idles while the queue still holds an utterance it could have taken at the last boundary."""
import os

import numpy as np
import pytest

from cmusphinx_amd import lib, s3io
from conftest import GOLDEN


def model(n_lanes, boundary, n):
    """a plain restatement: at every boundary a free lane (index order) takes the next utterance"""
    lanes = min(n_lanes, len(n))
    until, busy, nxt, F = [0] * lanes, [False] * lanes, 0, 0
    lane, f0 = [0] * len(n), [0] * len(n)
    while True:
        for z in range(lanes):
            if busy[z] and until[z] <= F:
                busy[z] = False
            if not busy[z] and nxt < len(n):
                lane[nxt], f0[nxt], until[z], busy[z] = z, F, F + n[nxt], True
                nxt += 1
        if not any(busy):
            return lane, f0, F
        F += boundary


@pytest.mark.parametrize("seed", range(6))
def test_schedule_properties(seed):
    rng = np.random.default_rng(seed)
    n_utt, n_lanes, boundary = int(rng.integers(1, 200)), int(rng.integers(1, 40)), int(rng.choice([1, 8, 16, 64]))
    n = rng.integers(1, 700, n_utt).astype(np.int32)
    lane, f0, total = lib.queue_schedule(n_lanes, boundary, n)
    ml, mf, mt = model(n_lanes, boundary, list(map(int, n)))
    assert list(lane) == ml and list(f0) == mf and total == mt
    assert (f0 % boundary == 0).all() and total % boundary == 0
    assert lane.max() < min(n_lanes, n_utt)
    assert (np.diff(f0) >= 0).all()                            # queue order = start order
    for z in range(lane.max() + 1):                             # a lane's utterances follow each other without overlap
        idx = np.flatnonzero(lane == z)
        ends = f0[idx] + n[idx]
        assert (f0[idx][1:] >= ends[:-1]).all()
        assert (f0[idx][1:] - ends[:-1] < boundary).all()       # ... and the lane takes the next one at the first boundary
    assert total >= (f0 + n).max() and total - (f0 + n).max() < boundary


def test_bad_arguments_are_refused():
    with pytest.raises(lib.S3AError):
        lib.queue_schedule(4, 8, [10, 0, 5])
    with pytest.raises(lib.S3AError):
        lib.queue_schedule(0, 8, [10])


def test_ragged_tidigits_set_in_numbers():
    """what DESIGN.md quotes: 31 utterances of 82 .. 339 frames; 8 lanes, 8-frame windows"""
    D = os.path.join(GOLDEN, "tidigits_decode")
    utts = [l.split()[0] for l in open(f"{D}/tidigits.length.arb.regression") if l.strip()]
    n = [len(s3io.read_mfc(f"{D}/cepstra/{u}.mfc")) // 13 for u in utts]
    assert (len(n), sum(n), min(n), max(n)) == (31, 5395, 82, 339)
    lock_step = sum(max(n[i:i + 8]) for i in range(0, len(n), 8))
    assert lock_step == 1021
    assert lib.queue_schedule(8, 8, n)[2] == 784
    assert lib.queue_schedule(8, 64, n)[2] == 896
    assert lib.queue_schedule(16, 8, n)[2] == 456
