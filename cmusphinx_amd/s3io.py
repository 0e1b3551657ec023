"""Sphinx-3 acoustic-model file formats (numpy reader / writer).

The product's loader is host C (csrc/s3a_model.c); this module exists so that
the Python harness (tests, bench.py, synthetic-model generator) can write and
read the very same files the reference loads:

* envelope: ``s3\\n<key value>...endhdr\\n`` + uint32 byte-order magic
  0x11223344 + payload + optional rotating checksum
  (reference: sphinxbase/src/libsphinxbase/util/bio.c:187-262 bio_readhdr,
  :265-296 chksum_accum, :491-504 bio_verify_chksum)
* means / variances: n_mgau, n_feat, n_density, veclen[n_feat], n, float32[n]
  (sphinx3/src/libs3decoder/libam/cont_mgau.c:148-429 mgau_file_read)
* mixture_weights: n_mgau, n_feat, n_comp, n, float32[n]
  (cont_mgau.c:507-683 mgau_mixw_read)
* transition_matrices: n_tmat, n_src, n_dst, n, float32[n]
  (sphinx3/src/libs3decoder/libam/tmat.c:155-270 tmat_init)
"""
from __future__ import annotations

import os
import numpy as np

BYTE_ORDER_MAGIC = 0x11223344


def chksum_u32(words: np.ndarray, start: int = 0) -> int:
    """bio.c:265-296 for 4-byte elements: sum = rotl(sum, 20) + w.

    The recurrence is inherently sequential (rotation does not distribute over
    the carrying 32-bit add), so this is a plain loop: ~4 M words/s.  Large
    synthetic models are written with ``chksum=False`` (the reference and the
    product loader both skip verification when the header has no chksum0).
    """
    s = int(start) & 0xFFFFFFFF
    w = np.ascontiguousarray(words, dtype="<u4").ravel()
    for v in w.tolist():
        s = (((s << 20) | (s >> 12)) + v) & 0xFFFFFFFF
    return s


def _read_envelope(path: str):
    with open(path, "rb") as f:
        buf = f.read()
    if not buf.startswith(b"s3\n"):
        raise ValueError(f"{path}: not an s3 binary file")
    end = buf.find(b"endhdr\n")
    if end < 0:
        raise ValueError(f"{path}: header has no endhdr")
    hdr = {}
    for line in buf[3:end].decode("ascii").splitlines():
        parts = line.split()
        if not parts or parts[0].startswith("#"):
            continue
        hdr[parts[0]] = parts[1] if len(parts) > 1 else ""
    off = end + len(b"endhdr\n")
    magic = int(np.frombuffer(buf, dtype="<u4", count=1, offset=off)[0])
    if magic == BYTE_ORDER_MAGIC:
        bo = "<"
    elif int(np.frombuffer(buf, dtype=">u4", count=1, offset=off)[0]) == BYTE_ORDER_MAGIC:
        bo = ">"
    else:
        raise ValueError(f"{path}: bad byte-order magic {magic:08x}")
    return hdr, bo, buf, off + 4


def _finish(path, hdr, bo, buf, off, body_words, verify):
    """Check the trailing checksum (if the header announces one) and EOF."""
    if "chksum0" in hdr:
        file_sum = int(np.frombuffer(buf, dtype=bo + "u4", count=1, offset=off)[0])
        off += 4
        if verify:
            got = chksum_u32(body_words)
            if got != file_sum:
                raise ValueError(f"{path}: checksum error; file {file_sum:08x}, computed {got:08x}")
    if off != len(buf):
        raise ValueError(f"{path}: more data than expected")


def read_gau(path: str, verify: bool = True) -> np.ndarray:
    """means or variances file -> float32 [n_mgau][n_density][veclen] (1 stream)."""
    hdr, bo, buf, off = _read_envelope(path)
    head = np.frombuffer(buf, dtype=bo + "i4", count=3, offset=off)
    n_mgau, n_feat, n_density = (int(x) for x in head)
    veclen = np.frombuffer(buf, dtype=bo + "i4", count=n_feat, offset=off + 12)
    n = int(np.frombuffer(buf, dtype=bo + "i4", count=1, offset=off + 12 + 4 * n_feat)[0])
    blk = int(veclen.sum())
    if n_feat != 1:
        raise ValueError(f"{path}: #feature streams {n_feat} != 1 for a continuous model")
    if n != n_mgau * n_density * blk:
        raise ValueError(f"{path}: #float32s({n}) doesn't match dimensions")
    nwords = 3 + n_feat + 1 + n
    words = np.frombuffer(buf, dtype=bo + "u4", count=nwords, offset=off)
    data = np.frombuffer(buf, dtype=bo + "f4", count=n, offset=off + 4 * (4 + n_feat))
    _finish(path, hdr, bo, buf, off + 4 * nwords, words, verify)
    return data.astype("<f4").reshape(n_mgau, n_density, blk)


def read_mixw(path: str, verify: bool = True) -> np.ndarray:
    """mixture_weights -> float32 [n_mgau][n_feat][n_comp] (raw, unnormalised)."""
    hdr, bo, buf, off = _read_envelope(path)
    n_mgau, n_feat, n_comp, n = (int(x) for x in np.frombuffer(buf, dtype=bo + "i4", count=4, offset=off))
    if n != n_mgau * n_feat * n_comp:
        raise ValueError(f"{path}: #float32s({n}) doesn't match header dimensions")
    words = np.frombuffer(buf, dtype=bo + "u4", count=4 + n, offset=off)
    data = np.frombuffer(buf, dtype=bo + "f4", count=n, offset=off + 16)
    _finish(path, hdr, bo, buf, off + 4 * (4 + n), words, verify)
    return data.astype("<f4").reshape(n_mgau, n_feat, n_comp)


def read_tmat(path: str, verify: bool = True) -> np.ndarray:
    """transition_matrices -> float32 [n_tmat][n_src][n_src+1] (raw probabilities)."""
    hdr, bo, buf, off = _read_envelope(path)
    n_tmat, n_src, n_dst, n = (int(x) for x in np.frombuffer(buf, dtype=bo + "i4", count=4, offset=off))
    if n_dst != n_src + 1 or n != n_tmat * n_src * n_dst:
        raise ValueError(f"{path}: bad tmat dimensions {n_tmat}x{n_src}x{n_dst} ({n})")
    words = np.frombuffer(buf, dtype=bo + "u4", count=4 + n, offset=off)
    data = np.frombuffer(buf, dtype=bo + "f4", count=n, offset=off + 16)
    _finish(path, hdr, bo, buf, off + 4 * (4 + n), words, verify)
    return data.astype("<f4").reshape(n_tmat, n_src, n_dst)


def _write(path: str, head_ints, data: np.ndarray, chksum: bool = True):
    body = np.concatenate([
        np.asarray(head_ints, dtype="<i4").view("<u4"),
        np.ascontiguousarray(data, dtype="<f4").ravel().view("<u4"),
    ])
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(b"s3\nversion 1.0\n")
        if chksum:
            f.write(b"chksum0 yes\n")
        f.write(b"endhdr\n")
        f.write(np.array([BYTE_ORDER_MAGIC], dtype="<u4").tobytes())
        f.write(body.tobytes())
        if chksum:
            f.write(np.array([chksum_u32(body)], dtype="<u4").tobytes())


def write_gau(path: str, arr: np.ndarray, chksum: bool = True):
    n_mgau, n_density, veclen = arr.shape
    _write(path, [n_mgau, 1, n_density, veclen, arr.size], arr, chksum)


def write_gau_streams(path: str, flat: np.ndarray, n_mgau: int, n_density: int, featlen, chksum: bool = True):
    """Multi-stream means/variances file (ms_gauden.c:205-312 gauden_param_read): header
    n_mgau, n_feat, n_density, veclen[n_feat], total; data in [m][f][d][veclen f] order."""
    featlen = [int(v) for v in featlen]
    flat = np.ascontiguousarray(flat, "<f4").ravel()
    assert flat.size == n_mgau * n_density * sum(featlen)
    _write(path, [n_mgau, len(featlen), n_density] + featlen + [flat.size], flat, chksum)


def write_mixw(path: str, arr: np.ndarray, chksum: bool = True):
    if arr.ndim == 2:
        arr = arr[:, None, :]
    n_mgau, n_feat, n_comp = arr.shape
    _write(path, [n_mgau, n_feat, n_comp, arr.size], arr, chksum)


def write_lda(path: str, arr: np.ndarray, chksum: bool = True):
    """feature transform file (sphinxbase feat/lda.c:60-133 feat_read_lda: bio_fread_3d): [n_lda][rows = output dims][cols = stream length]"""
    if arr.ndim == 2:
        arr = arr[None]
    n, m, k = arr.shape
    _write(path, [n, m, k, arr.size], arr, chksum)


def write_tmat(path: str, arr: np.ndarray, chksum: bool = True):
    n_tmat, n_src, n_dst = arr.shape
    assert n_dst == n_src + 1
    _write(path, [n_tmat, n_src, n_dst, arr.size], arr, chksum)


def read_mfc(path: str) -> np.ndarray:
    """Sphinx cepstrum file: int32 count + float32[count], either endianness
    (reference: sphinxbase/src/libsphinxbase/feat/feat.c feat_s2mfc_read)."""
    raw = np.fromfile(path, dtype="<i4", count=1)
    size = os.path.getsize(path)
    n = int(raw[0])
    bo = "<"
    if n * 4 + 4 != size:
        n = int(raw.byteswap()[0])
        bo = ">"
        if n * 4 + 4 != size:
            raise ValueError(f"{path}: header count does not match file size")
    return np.fromfile(path, dtype=bo + "f4", offset=4, count=n).astype("<f4")


def read_mdef(path: str) -> dict:
    """Sphinx-3 text model-definition file (reference:
    sphinx3/src/libs3decoder/libam/mdef.c:672-842 mdef_init, :603-656
    sseq_compress).  Returns the arrays the hot path needs:

    n_ciphone, n_phone, n_emit_state, n_ci_sen, n_sen, n_tmat,
    ciphone names, per-phone (ci, lc, rc, wpos, tmat, filler),
    phone_ssid [n_phone], sseq [n_sseq][n_emit] int16 (unique state sequences
    in first-occurrence order, as sseq_compress numbers them),
    cd2cisen [n_sen] int16 (mdef.c:819-836).
    """
    with open(path, "r") as f:
        lines = [ln.rstrip("\n") for ln in f if not ln.startswith("#")]
    it = iter(lines)
    ver = next(it).strip()
    if ver != "0.3":
        raise ValueError(f"{path}: version {ver}, expecting 0.3")
    hdr = {}
    for _ in range(6):
        val, tag = next(it).split()[:2]
        hdr[tag] = int(val)
    n_ci, n_tri = hdr["n_base"], hdr["n_tri"]
    n_phone = n_ci + n_tri
    n_sen, n_ci_sen, n_tmat = hdr["n_tied_state"], hdr["n_tied_ci_state"], hdr["n_tied_tmat"]
    if hdr["n_state_map"] % n_phone:
        raise ValueError(f"{path}: n_state_map not a multiple of #phones")
    n_emit = hdr["n_state_map"] // n_phone - 1
    names, name2id = [], {}
    phones = np.zeros((n_phone, 6), np.int32)        # ci lc rc wpos tmat filler
    states = np.zeros((n_phone, n_emit), np.int16)
    wpos_code = {"-": -1, "b": 0, "e": 1, "s": 2, "i": 3}
    for p in range(n_phone):
        tok = next(it).split()
        base, lft, rt, wp, attrib, tmat = tok[:6]
        st = tok[6:6 + n_emit]
        if tok[6 + n_emit] != "N":
            raise ValueError(f"{path}: phone {p}: missing non-emitting state N")
        if p < n_ci:
            name2id[base] = p
            names.append(base)
            phones[p] = (p, -1, -1, -1, int(tmat), int(attrib == "filler"))
        else:
            phones[p] = (name2id[base], name2id[lft], name2id[rt], wpos_code[wp], int(tmat),
                         int(attrib == "filler"))
        states[p] = [int(x) for x in st]
    # sseq_compress: unique sequences numbered by first occurrence over phones
    seen, sseq, ssid = {}, [], np.zeros(n_phone, np.int32)
    for p in range(n_phone):
        key = states[p].tobytes()
        if key not in seen:
            seen[key] = len(sseq)
            sseq.append(states[p].copy())
        ssid[p] = seen[key]
    cd2cisen = np.zeros(n_sen, np.int16)
    cd2cisen[:n_ci_sen] = np.arange(n_ci_sen)
    for p in range(n_ci, n_phone):
        ci = phones[p, 0]
        for s in range(n_emit):
            cd2cisen[states[p, s]] = states[ci, s]
    return dict(n_ciphone=n_ci, n_phone=n_phone, n_emit_state=n_emit, n_ci_sen=n_ci_sen,
                n_sen=n_sen, n_tmat=n_tmat, ciphone=names, phones=phones, states=states,
                phone_ssid=ssid, sseq=np.array(sseq, np.int16), cd2cisen=cd2cisen)
