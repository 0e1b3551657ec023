#!/usr/bin/env python3
"""frame-synchronous scoring pass (1 frame per launch) on a model shape: tools/fs_bench.py hub4|wsj"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cmusphinx_amd import lib, synth
shape = synth.WSJ_STRESS if (len(sys.argv) > 1 and sys.argv[1] == "wsj") else synth.HUB4
m = synth.make_model(**shape)
g = lib.MgauModel.init_arrays(m["mean"], m["var"], m["mixw"], lib.LogMath(1.0003))
T = 200
f = synth.make_features(m, T, seed=99)
fd = lib.DevBuf(f.nbytes).upload(f)
sd = lib.DevBuf(T * g.S * 4)
g.bench(fd, T, sd, None, 1, 1)
us, kus, n = g.bench(fd, T, sd, None, 1, 5)
b = g.S * g.C * (2 * g.D + 2) * 4 + g.D * 4 + g.S * 4
print(f"{sys.argv[1:]}: {kus:.2f} us/launch, {b / kus / 1e3:.0f} GB/s = {b / kus / 1e3 / 8000 * 100:.1f} % of 8 TB/s")
