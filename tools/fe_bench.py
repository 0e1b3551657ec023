#!/usr/bin/env python3
"""MFCC front end: device (s3a_fe_process_utt, host buffers: PCIe inclusive) vs the unmodified reference
(oracle/_ref/ref_dump fe, one core) on a 100 s utterance; also reports how many values are bit-identical."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cmusphinx_amd import lib  # noqa: E402
import oracle_lib as O  # noqa: E402

rng = np.random.default_rng(3)
n = 1_600_000
x = (rng.standard_normal(n) * 3000 * (np.sin(np.arange(n) / 9000.0) ** 2)).astype(np.int16)
fe = lib.FrontEnd()
got = fe.process_utt(x)
exp = O.OracleFe().process_utt(x)
same = (got.view(np.uint32) == exp.view(np.uint32)).mean()
print(f"frames {len(got)}  bit-identical values {same * 100:.4f} %  max |diff| {np.abs(got - exp).max():.3g}")
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    fe.process_utt(x)
dt = (time.perf_counter() - t0) / reps
print(f"device: {dt * 1e3:.3f} ms per 100 s utterance = {len(got) / dt:.0f} frames/s = {len(got) / dt / 100:.0f} x real time (H2D + kernel + D2H)")
ref = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
if os.path.exists(ref):
    d = tempfile.mkdtemp()
    raw = os.path.join(d, "x.raw")
    x.tofile(raw)
    out = subprocess.run([ref, "fe", raw, d], env=dict(os.environ, REF_FE_REPS="3"), capture_output=True, text=True).stdout
    f, s = out.split()[1], out.split()[3]
    print(f"reference, 1 core: {float(f) / float(s):.0f} frames/s = {float(f) / float(s) / 100:.0f} x real time")
