#!/usr/bin/env python3
"""tests/golden/nbest_tidigits.npz: lattices as the UNMODIFIED reference decoder wrote them (-outlatdir, dag_write: dag->list order,
a source's links in succlist order -- the very orders s3a_uttdec_lattice hands out) and the N-best lists it wrote for them
(-nbestdir, astar.c nbest_search) -- the fixture of tests/test_nbest_host.py (the library's s3a_lattice_nbest, no GPU).
Runs only in the build container (needs oracle/_ref/sphinx3_decode = the reference built where it lies).

    python tests/golden/make_nbest_golden.py
"""
import gzip
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_oracle_dag as T  # noqa: E402

REFDEC = os.path.join(ROOT, "oracle", "_ref", "sphinx3_decode")
CASES = {  # name -> (arguments beside the task's, what the test must know of them)
    "plain": (["-nbest", "20"], dict(nbest=20, beam=1e-55, bestpathlw=0.0, maxppath=1000000)),
    "lw14_beam1e-30_n7": (["-nbest", "7", "-bestpathlw", "14", "-beam", "1e-30"], dict(nbest=7, beam=1e-30, bestpathlw=14.0, maxppath=1000000)),
    "maxppath60": (["-nbest", "50", "-maxppath", "60"], dict(nbest=50, beam=1e-55, bestpathlw=0.0, maxppath=60)),
}


def read_text(path):
    return gzip.open(path, "rb").read() if path.endswith(".gz") else open(path, "rb").read()


def parse_lattice(txt):
    """dag_write's file (dag.c:731-790) -> (n_frames, nodes [n, 6] = wid-less (word, sf, fef, lef), initial, final, links [m, 3])"""
    lines = [l for l in txt.decode().splitlines() if l and not l.startswith("#")]
    k = 0
    assert lines[k].startswith("Frames "); n_frames = int(lines[k].split()[1]); k += 1
    assert lines[k].startswith("Nodes "); n = int(lines[k].split()[1]); k += 1
    nodes = []
    for j in range(n):
        f = lines[k + j].split()
        assert int(f[0]) == j
        nodes.append((f[1], int(f[2]), int(f[3]), int(f[4])))
    k += n
    initial = int(lines[k].split()[1]); final = int(lines[k + 1].split()[1]); k += 2
    assert lines[k].startswith("BestSegAscr"); nb = int(lines[k].split()[1]); k += 1 + nb
    assert lines[k].startswith("Edges"); k += 1
    links = []
    while lines[k] != "End":
        links.append(tuple(int(x) for x in lines[k].split())); k += 1
    return n_frames, nodes, initial, final, links


def main():
    out = {}
    for name, (extra, meta) in CASES.items():
        with tempfile.TemporaryDirectory() as td:
            lat, nb = os.path.join(td, "lat"), os.path.join(td, "nb")
            os.makedirs(lat); os.makedirs(nb)
            args = T.tidigits_args() + extra + ["-outlatdir", lat, "-latext", "lat", "-nbestdir", nb, "-nbestext", "nbest", "-hyp", os.path.join(td, "h")]
            p = subprocess.run([REFDEC] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, errors="ignore")
            assert p.returncode == 0, p.stdout[-2000:]
            utts = sorted(f[:-4] for f in os.listdir(lat) if f.endswith(".lat"))
            kept = 0
            for u in utts:
                n_frames, nodes, initial, final, links = parse_lattice(read_text(os.path.join(lat, u + ".lat")))
                nbf = os.path.join(nb, u + ".nbest")
                txt = read_text(nbf) if os.path.exists(nbf) else b""
                key = f"{name}.{kept}"
                out[key + ".uttid"] = np.frombuffer(u.encode(), np.uint8)
                out[key + ".words"] = np.frombuffer("\n".join(w for w, _, _, _ in nodes).encode(), np.uint8)
                out[key + ".nodes"] = np.array([[sf, fef, lef] for _, sf, fef, lef in nodes], np.int32)
                out[key + ".links"] = np.array(links, np.int32).reshape(-1, 3)
                out[key + ".info"] = np.array([n_frames, initial, final], np.int32)
                out[key + ".text"] = np.frombuffer(txt, np.uint8)
                kept += 1
            out[name + ".n"] = np.array([kept], np.int32)
            out[name + ".meta"] = np.array([meta["nbest"], meta["maxppath"]], np.int32)
            out[name + ".fmeta"] = np.array([meta["beam"], meta["bestpathlw"]], np.float64)
            print(name, kept, "utterances,", sum(len(out[f"{name}.{k}.text"]) for k in range(kept)), "bytes of lists,",
                  sum(1 for k in range(kept) if len(out[f"{name}.{k}.text"]) == 0), "without a list")
    # the dictionary's word ids as dict_init assigns them (main dictionary in file order, then the filler dictionary; dict.c:300-420)
    words = []
    for fn in ("dictionary", "fillerdict"):
        for ln in open(os.path.join(T.D, fn), errors="ignore"):
            ln = ln.strip()
            if ln and not ln.startswith("#") and not ln.startswith(";;"):
                words.append(ln.split()[0])
    out["dict.words"] = np.frombuffer("\n".join(words).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "nbest_tidigits.npz"), **out)
    print(os.path.getsize(os.path.join(HERE, "nbest_tidigits.npz")) // 1024, "KB;", len(words), "dictionary words")


if __name__ == "__main__":
    main()
