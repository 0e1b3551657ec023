"""The decoder bundle's record format (cmusphinx_amd/bundle.py reads what integration/sphinx3/s3amd_tst.c's
export_bundle writes: {tag, n, n x int32} records) -- CPU only: a hand-made bundle parses, a truncated one is refused."""
import numpy as np
import pytest

from cmusphinx_amd import bundle


def rec(tag, words):
    w = np.asarray(words, dtype="<i4")
    return np.concatenate([np.array([tag, len(w)], "<i4"), w])


def make(path, cut=0):
    parts = [rec(1, [2, 3, 5, 7, 11, 13, 17, 4, 6, 39]),                    # header: n_tree .. veclen
             rec(10, [3, 1, 1, 0]), rec(11, [4, 5, 6]), rec(14, [-1, -1, 9]),   # one tree: 3 nodes
             rec(10, [2, 1, 1, 1]), rec(11, [1, 2]),                            # a second tree
             rec(30, [100, 200, 300]), rec(31, list(range(100))),
             rec(40, [2, 0, 1, 1, 0, 1, 3]),
             rec(45, np.frombuffer(b"<s>\0</s>\0\0\0\0", "<i4"))]
    raw = np.concatenate(parts)
    if cut:
        raw = raw[:-cut]
    raw.tofile(path)


def test_hand_made_bundle_parses(tmp_path):
    p = str(tmp_path / "b.bundle")
    make(p)
    b = bundle.read(p)
    assert b["n_tree"] == 2 and b["veclen"] == 39 and b["n_ci_sen"] == 4
    assert (b["n_ug"], b["n_bg"], b["n_tg"]) == (100, 200, 300) and len(b["ug_prob"]) == 100
    assert b["words"] == ["<s>", "</s>"] and b["finishwid"] == 1
    trees = b["trees"]
    assert len(trees) == 2 and list(trees[0]["ssid"]) == [4, 5, 6] and list(trees[0]["wid"]) == [-1, -1, 9]
    assert trees[1]["n_node"] == 2 and trees[1]["type"] == 1 and list(trees[1]["ssid"]) == [1, 2]


@pytest.mark.parametrize("cut", [1, 2, 50])
def test_truncated_bundle_is_refused(tmp_path, cut):
    p = str(tmp_path / "t.bundle")
    make(p, cut=cut)
    with pytest.raises(ValueError, match="truncated bundle"):
        bundle.read(p)
