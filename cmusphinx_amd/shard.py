"""Utterance-level sharding across GPUs and the one end-of-batch gather.

The reference has no distributed mode; its only batch-partitioning device is
-ctloffset / -ctlcount on the control file (sphinx3/src/programs/main_decode.c:164-169,
libcommon/corpus.c:538 ctl_process).  Utterances are independent (all
per-utterance state is reset in srch_utt_begin, srch.c:453-479), so the path
shards embarrassingly: rank r of W decodes its slice of the control list with a
replicated model and NO per-frame collective; one all_gather of fixed-size
result records closes the batch, and rank 0 re-assembles them in control-file
order so the output diffs cleanly against a single-process run.
"""
from __future__ import annotations

import numpy as np

REC_WORDS = 64          # hypothesis words kept per record
REC_LEN = 4 + REC_WORDS  # [utt_index, n_frames, total_score, n_words, word ids...]


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


def pack_record(utt_index: int, n_frames: int, total_score: int, word_ids) -> np.ndarray:
    rec = np.full(REC_LEN, -1, np.int64)
    w = list(word_ids)[:REC_WORDS]
    rec[0], rec[1], rec[2], rec[3] = utt_index, n_frames, total_score, len(w)
    rec[4:4 + len(w)] = w
    return rec


def unpack_record(rec):
    n = int(rec[3])
    return dict(utt=int(rec[0]), n_frames=int(rec[1]), score=int(rec[2]),
                words=[int(v) for v in rec[4:4 + n]])


def gather_records(local_records, n_utt_total: int, dist=None, device="cpu"):
    """all_gather fixed-size records; returns the list in utterance order (every rank).

    `dist` is torch.distributed (backend nccl == RCCL on the GPU box, gloo in the
    CPU tests); None means single process.  Ranks may hold different counts:
    each pads to the maximum, padding rows carry utt = -1.
    """
    local = np.stack(local_records) if len(local_records) else np.zeros((0, REC_LEN), np.int64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        allrec = local
    else:
        import torch
        world = dist.get_world_size()
        cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        m = max(int(c.item()) for c in cnts)
        buf = torch.full((max(m, 1), REC_LEN), -1, dtype=torch.int64, device=device)
        if local.shape[0]:
            buf[:local.shape[0]] = torch.from_numpy(local).to(device)
        out = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(out, buf)
        allrec = torch.cat(out).cpu().numpy()
    allrec = allrec[allrec[:, 0] >= 0]
    got = sorted((unpack_record(r) for r in allrec), key=lambda d: d["utt"])
    if [d["utt"] for d in got] != list(range(n_utt_total)):
        raise RuntimeError("gather: utterances missing or duplicated across ranks")
    return got
