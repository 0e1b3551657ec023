"""Utterance-level sharding across GPUs and the one end-of-batch gather.

The reference has no distributed mode; its only batch-partitioning device is
-ctloffset / -ctlcount on the control file (sphinx3/src/programs/main_decode.c:164-169,
libcommon/corpus.c:538 ctl_process).  Utterances are independent (all
per-utterance state is reset in srch_utt_begin, srch.c:453-479), so the path
shards embarrassingly: rank r of W decodes its slice of the control list with a
replicated model and NO per-frame collective; ONE all_gather of fixed-size
hypothesis records (s3a_hyp_record_t, include/cmusphinx_amd.h: uttid, words with
sf/ef/ascr/lscr/scale, score, n_frames -- what -hyp and -hypseg are written from)
closes the batch, and rank 0 writes both files in control-file order so they diff
cleanly against a single-process run.  The collective is
torch.distributed's all_gather: backend "nccl" is RCCL over xGMI on the GPU box,
"gloo" in the CPU tests; the records themselves are packed, read and formatted by
the C ABI (s3a_uttdec_hyp / s3a_hyp_format).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib

REC_BYTES = C.sizeof(lib.HypRecord)


def shard_contiguous(n_utt: int, rank: int, world: int):
    """-ctloffset/-ctlcount style contiguous slices, sizes differing by at most 1."""
    base, extra = divmod(n_utt, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return list(range(lo, hi))


def shard_by_frames(n_frames, rank: int, world: int):
    """Longest-first greedy balancing by frame count (deterministic; ties by index)."""
    order = sorted(range(len(n_frames)), key=lambda i: (-int(n_frames[i]), i))
    load = [0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += int(n_frames[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


def gather_records(local_records, n_utt_total: int, dist=None, device="cpu"):
    """ONE all_gather of the ranks' hypothesis records; returns them in utterance order (every rank).

    local_records: list of lib.HypRecord with utt_index set.  Ranks may hold different counts: each pads to
    the maximum with records whose utt_index is -1 (one extra all_gather of a single int64 finds the maximum).
    Raises when an utterance is missing or duplicated, or a record reports a word-list overflow.
    """
    n_local = len(local_records)
    local = np.frombuffer(b"".join(bytes(r) for r in local_records), np.uint8).reshape(n_local, REC_BYTES) \
        if n_local else np.zeros((0, REC_BYTES), np.uint8)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        allrec = local
    else:
        import torch
        world = dist.get_world_size()
        cnt = torch.tensor([n_local], dtype=torch.int64, device=device)
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        m = max(int(c.item()) for c in cnts)
        pad = lib.HypRecord()
        pad.utt_index = -1
        buf = torch.from_numpy(np.frombuffer(bytes(pad) * max(m, 1), np.uint8).reshape(max(m, 1), REC_BYTES).copy()).to(device)
        if n_local:
            buf[:n_local] = torch.from_numpy(local.copy()).to(device)
        out = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(out, buf)                       # the batch's one exchange (RCCL over xGMI on the GPU box)
        allrec = torch.cat(out).cpu().numpy()
    recs = [lib.HypRecord.from_buffer_copy(allrec[i].tobytes()) for i in range(allrec.shape[0])]
    recs = sorted((r for r in recs if r.utt_index >= 0), key=lambda r: r.utt_index)
    if [r.utt_index for r in recs] != list(range(n_utt_total)):
        raise RuntimeError("gather: utterances missing or duplicated across ranks")
    over = [r.utt_index for r in recs if r.status == -3]
    if over:
        raise RuntimeError(f"gather: hypotheses of utterances {over} exceed S3A_HYP_MAXW words")
    return recs


def gather_var(local, n_utt_total: int, dist=None, device="cpu"):
    """The exchange without a word limit: local = list of (lib.HypHeader, words int32 [n_words, 6]).  Two collectives:
    the fixed-size headers (every rank learns every hypothesis' length), then the ranks' words, each rank's block padded
    to the largest rank's total.  Returns [(header, words)] in utterance order (every rank)."""
    HB = C.sizeof(lib.HypHeader)
    n_local = len(local)
    hdr = np.frombuffer(b"".join(bytes(h) for h, _ in local), np.uint8).reshape(n_local, HB) if n_local else np.zeros((0, HB), np.uint8)
    words = np.concatenate([np.asarray(w, np.int32).reshape(-1, 6) for _, w in local]) if n_local else np.zeros((0, 6), np.int32)
    assert [int(h.n_words) if h.status == 0 else 0 for h, _ in local] == [len(w) for _, w in local]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        blocks = [(hdr, words)]
    else:
        import torch
        world = dist.get_world_size()
        cnt = torch.tensor([n_local, len(words)], dtype=torch.int64, device=device)
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        mh, mw = (max(int(c[k].item()) for c in cnts) for k in (0, 1))
        pad = lib.HypHeader()
        pad.utt_index = -1
        hb = np.frombuffer(bytes(pad) * max(mh, 1), np.uint8).reshape(max(mh, 1), HB).copy()
        hb[:n_local] = hdr
        wb = np.zeros((max(mw, 1), 6), np.int32)
        wb[:len(words)] = words
        th, tw = torch.from_numpy(hb).to(device), torch.from_numpy(wb).to(device)
        oh, ow = [torch.empty_like(th) for _ in range(world)], [torch.empty_like(tw) for _ in range(world)]
        dist.all_gather(oh, th)                         # lengths (and everything else that is fixed-size)
        dist.all_gather(ow, tw)                         # the padded payload
        blocks = [(h.cpu().numpy(), w.cpu().numpy()) for h, w in zip(oh, ow)]
    out = []
    for hb, wb in blocks:
        pos = 0
        for i in range(hb.shape[0]):
            h = lib.HypHeader.from_buffer_copy(hb[i].tobytes())
            if h.utt_index < 0:
                continue
            n = int(h.n_words) if h.status == 0 else 0
            out.append((h, wb[pos:pos + n].copy()))
            pos += n
    out.sort(key=lambda t: t[0].utt_index)
    if [h.utt_index for h, _ in out] != list(range(n_utt_total)):
        raise RuntimeError("gather: utterances missing or duplicated across ranks")
    return out


def write_outputs(recs, fmt, hyp_path=None, hypseg_path=None, log=None):
    """-hyp / -hypseg files from gathered records (fmt: record -> (match line, matchseg line), e.g. bundle.Decoder.format;
    records may be (header, words) pairs, then fmt takes both).  An utterance that could not be ended (status != 0:
    decode error, no word exit) gets NO line in either file -- the reference logs `utt_end failed` and writes
    nothing for it (srch.c:495-498) -- so the files stay byte-identical to a single-process reference run."""
    lines = []
    for r in recs:
        h = r[0] if isinstance(r, tuple) else r
        if h.status != 0:
            if log is not None:
                log.append((h.utt_index, h.uttid.decode(errors="replace"), h.status))
            continue
        lines.append(fmt(*r) if isinstance(r, tuple) else fmt(r))
    if hyp_path:
        with open(hyp_path, "w") as f:
            f.writelines(l[0] for l in lines)
    if hypseg_path:
        with open(hypseg_path, "w") as f:
            f.writelines(l[1] for l in lines)
    return lines
