"""pocketsphinx's senone score dump (-senlogdir; SURVEY 8(f).3: the score interchange format between decoders):
s3a_senlog_* (host-only file format code in cmusphinx_amd/csrc/s3a_host.c) against a file written by the
unmodified pocketsphinx (tests/golden/ps_senlog_man.ah.111a.sen, made by tests/golden/make_golden.py senlog):
read every frame, write them back, the bytes must be the reference's; and the frames must mean what an
independent parse of the format says."""
import os
import re
import struct

import numpy as np
import pytest

from cmusphinx_amd import lib

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ps_senlog_man.ah.111a.sen")
DUMMY = 0x7fff


def parse(raw):
    """independent reading of acmod_write_scores' frame format"""
    end = raw.index(b"endhdr\n") + 7
    hdr = dict(l.split(None, 1) for l in raw[3:end - 7].decode().splitlines())
    assert struct.unpack("<I", raw[end:end + 4])[0] == 0x11223344
    n_sen, pos, frames = int(hdr["n_sen"]), end + 4, []
    while pos < len(raw):
        (na,) = struct.unpack_from("<h", raw, pos); pos += 2
        scr = np.full(n_sen, DUMMY, np.int16)
        if na == n_sen:
            scr[:] = np.frombuffer(raw, "<i2", na, pos); pos += 2 * na
            frames.append((scr, None))
        else:
            act = np.frombuffer(raw, np.uint8, na, pos); pos += na
            ids = np.cumsum(act.astype(np.int64))
            scr[ids] = np.frombuffer(raw, "<i2", na, pos); pos += 2 * na
            frames.append((scr, act.copy()))
    return hdr, frames


def test_reads_the_reference_dump_and_writes_it_back_byte_for_byte(tmp_path):
    lib.load()
    raw = open(G, "rb").read()
    hdr, frames = parse(raw)
    r = lib.SenLog.open(G)
    assert r.n_sen == int(hdr["n_sen"]) == 602 and abs(r.logbase - float(hdr["logbase"])) < 1e-9
    out = str(tmp_path / "copy.sen")
    w = lib.SenLog.create(out, hdr["mdef_file"], r.n_sen, r.logbase)
    n = 0
    while True:
        fr = r.read()
        if fr is None:
            break
        scr, act = fr
        exp_scr, exp_act = frames[n]
        assert np.array_equal(scr, exp_scr), n
        if exp_act is None:
            assert len(act) == r.n_sen
            w.write(scr)
        else:
            assert np.array_equal(act, exp_act), n
            w.write(scr, act)
        n += 1
    w.close(); r.close()
    assert n == len(frames) and n > 100
    assert any(a is None for _, a in frames) or any(a is not None for _, a in frames)
    assert open(out, "rb").read() == raw


def test_all_senones_frame_and_empty_frame_round_trip(tmp_path):
    lib.load()
    p = str(tmp_path / "x.sen")
    w = lib.SenLog.create(p, "some/mdef", 10, 1.0001)
    full = np.arange(10, dtype=np.int16) * 7
    w.write(full)                                   # every senone: no list in the file
    w.write(full, np.array([], np.uint8))           # nothing active
    w.write(full, np.array([0, 3, 6], np.uint8))    # senones 0, 3, 9
    w.close()
    raw = open(p, "rb").read()
    assert re.match(rb"s3\nversion 0.1\nmdef_file some/mdef\nn_sen 10\nlogbase 1.000100\nendhdr\n", raw)
    r = lib.SenLog.open(p)
    a, la = r.read(); b, lb = r.read(); c, lc = r.read()
    assert r.read() is None
    assert np.array_equal(a, full) and len(la) == 10
    assert (b == DUMMY).all() and len(lb) == 0
    assert list(lc) == [0, 3, 6] and c[0] == 0 and c[3] == 21 and c[9] == 63 and (np.delete(c, [0, 3, 9]) == DUMMY).all()


def test_bad_files_and_arguments(tmp_path):
    lib.load()
    p = tmp_path / "bad.sen"
    p.write_bytes(b"not a dump")
    with pytest.raises(lib.S3AError):
        lib.SenLog.open(str(p))
    with pytest.raises(lib.S3AError):
        lib.SenLog.create(str(tmp_path / "y.sen"), "m", 40000, 1.0001)      # n_active is an int16 in the file
    w = lib.SenLog.create(str(tmp_path / "z.sen"), "m", 4, 1.0001)
    with pytest.raises(lib.S3AError):
        w.write(np.zeros(4, np.int16), np.array([2, 2], np.uint8))          # runs past the last senone
