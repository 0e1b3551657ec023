#!/usr/bin/env python3
"""tests/golden/dag_tables.npz: history tables as the reference's second pass read them and what the pinned restatement
(oracle/_ref/ref_s3odag_decode, S3O_DAGDUMP) produced from them -- the fixture of tests/test_gpu_dag.py.
Runs only in the build container (needs oracle/_ref and, for the RM1 cases, tests/_local_data).

    python tests/golden/make_dag_golden.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_oracle_dag as T  # noqa: E402

ODAG = os.path.join(ROOT, "oracle", "_ref", "ref_s3odag_decode")
CASES = {  # name -> (args, cfg overrides the device test applies)
    "tidigits": (T.tidigits_args() + ["-bestpath", "1"], {}),
    "tidigits_lw14_minendfr1": (T.tidigits_args() + ["-bestpath", "1", "-bestpathlw", "14", "-min_endfr", "1"], {"bestpathlw": 14.0, "min_endfr": 1}),
    "rm1": (T.rm_args(5) + ["-bestpath", "1"], {}),
    "rm1_minendfr0": (T.rm_args(3) + ["-bestpath", "1", "-min_endfr", "0"], {"min_endfr": 0}),
    "rm1_maxlpf5": (T.rm_args(8) + ["-bestpath", "1", "-maxlpf", "5"], {"maxlpf": 5}),
}
NAMES = {10: "wid", 11: "sf", 12: "ef", 13: "ascr", 14: "lscr", 15: "score", 16: "hyp_wid", 17: "hyp_sf", 20: "lmop",
         30: "o_wid", 31: "o_sf", 32: "o_ef", 33: "o_ascr", 34: "o_lscr"}


def parse(path):
    raw = np.fromfile(path, "<i4")
    pos, utts = 0, []
    while pos < len(raw):
        tag, n = int(raw[pos]), int(raw[pos + 1])
        d = raw[pos + 2: pos + 2 + n].copy()
        pos += 2 + n
        if tag == 1:
            utts.append(dict(hdr=d))
        else:
            utts[-1][NAMES[tag]] = d
    return utts


out = {}
for name, (args, _) in CASES.items():
    with tempfile.TemporaryDirectory() as td:
        dump = os.path.join(td, "dump")
        p = subprocess.run([ODAG] + args + ["-hyp", os.path.join(td, "h")], env=dict(os.environ, S3O_DAGDUMP=dump),
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert p.returncode == 0, name
        utts = parse(dump)
        out[f"{name}.n"] = np.array([len(utts)], np.int32)
        for k, u in enumerate(utts):
            for key, v in u.items():
                out[f"{name}.{k}.{key}"] = v
        print(name, len(utts), "utterances;", sum(len(u["wid"]) for u in utts), "entries;", [int(u["hdr"][4]) for u in utts])
np.savez_compressed(os.path.join(HERE, "dag_tables.npz"), **out)
print(os.path.getsize(os.path.join(HERE, "dag_tables.npz")) // 1024, "KB")
