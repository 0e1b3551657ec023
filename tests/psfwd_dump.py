"""Reader of the backpointer-table dumps oracle/ref_ps_fwd.c writes with -bpdump (test infrastructure)."""
import numpy as np


def read_bpdump(path):
    """-> {uttid: dict(hdr fields, table [n][7] = frame wid bp score s_idx real_wid valid, bss, idx)}"""
    raw = open(path, "rb").read()
    out, pos = {}, 0
    while pos < len(raw):
        hdr = np.frombuffer(raw, np.int32, 16, pos); pos += 64
        assert hdr[0] == 0x50534250, "bad magic"
        ln, n_frame, bpidx, bss_head = (int(x) for x in hdr[1:5])
        uttid = raw[pos:pos + ln].decode(); pos += ln
        tab = np.frombuffer(raw, np.int32, bpidx * 7, pos).reshape(bpidx, 7); pos += bpidx * 28
        bss = np.frombuffer(raw, np.int32, bss_head, pos); pos += bss_head * 4
        nidx = int(hdr[14]) + 1
        idx = np.frombuffer(raw, np.int32, nidx, pos); pos += nidx * 4
        out[uttid] = dict(n_frame=n_frame, bpidx=bpidx, bss_head=bss_head, best_score=int(hdr[5]),
                          last_phone_best_score=int(hdr[6]), renormalized=int(hdr[7]), st=hdr[8:14].copy(),
                          output_frame=int(hdr[14]), table=tab, bss=bss, idx=idx)
    return out


def diff_bpdumps(a, b):
    """list of human-readable differences between two dumps (empty = identical)"""
    bad = []
    if list(a) != list(b):
        return [f"utterance lists differ: {list(a)[:3]}.. vs {list(b)[:3]}.."]
    for u in a:
        x, y = a[u], b[u]
        for k in ("n_frame", "bpidx", "bss_head", "best_score", "last_phone_best_score", "renormalized", "output_frame"):
            if x[k] != y[k]:
                bad.append(f"{u}: {k} {x[k]} != {y[k]}")
        if not np.array_equal(x["st"], y["st"]):
            bad.append(f"{u}: stats {x['st']} != {y['st']}")
        for k in ("table", "bss", "idx"):
            if x[k].shape != y[k].shape:
                bad.append(f"{u}: {k} shape {x[k].shape} != {y[k].shape}")
            elif not np.array_equal(x[k], y[k]):
                w = np.argwhere(x[k] != y[k])[0]
                bad.append(f"{u}: {k} first difference at {tuple(w)}: {x[k][tuple(w)]} != {y[k][tuple(w)]}")
    return bad


if __name__ == "__main__":
    import sys
    d = diff_bpdumps(read_bpdump(sys.argv[1]), read_bpdump(sys.argv[2]))
    print("\n".join(d) if d else "identical")
    sys.exit(1 if d else 0)
