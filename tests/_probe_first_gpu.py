import sys, time, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from cmusphinx_amd import lib, synth
import oracle_lib as O
print('devices', lib.device_count(), 'missing syms', lib.MISSING)
lm = lib.LogMath(1.0003)
olm = O.OracleLogMath(1.0003)
print('table eq', np.array_equal(lm.table, olm.table))
# small degenerate model, C=5 (CP=8), D=39
m = synth.make_model(300, 30, 5, 39, 6, 3, 1234, degenerate=True)
fx = synth.make_features(m, 37, 99)
g = lib.MgauModel.init_arrays(m['mean'], m['var'], m['mixw'], lm)
og = O.OracleMgau(m['mean'], m['var'], m['mixw'], olm)
p = g.params()
nc = p['n_comp']
print('n_comp eq', np.array_equal(nc, og.n_comp))
sc, best = g.score_frames(fx)
osc = og.score_all(fx)
print('small score eq', np.array_equal(sc, osc), 'best eq', np.array_equal(best, osc.max(1)), (sc != osc).sum())
for nfr in (1, 5, 8, 15, 16):
    sc2 = g.score_frames(fx[:nfr], want_best=False)
    print('  nfr', nfr, np.array_equal(sc2, osc[:nfr]))
# mgau_eval single
ok = True
for s in (0, 7, 299):
    for t in (0, 3):
        ok &= g.eval(s, fx[t], t, 1) == og.eval(s, fx[t], t, 1)
print('mgau_eval eq', ok)
# hub4-shaped
t0 = time.time()
M = synth.make_model(**synth.HUB4)
X = synth.make_features(M, 1000, 7)
print('gen', time.time() - t0)
G = lib.MgauModel.init_arrays(M['mean'], M['var'], M['mixw'], lm)
OG = O.OracleMgau(M['mean'], M['var'], M['mixw'], olm)
t0 = time.time(); SC, B = G.score_frames(X); print('gpu host-api time', time.time() - t0)
t0 = time.time(); OSC = OG.score_all(X[:24]); dt = time.time() - t0; print('oracle 24 frames', dt, 'frames/s', 24 / dt)
print('hub4 score eq (24 frames)', np.array_equal(SC[:24], OSC), (SC[:24] != OSC).sum())
# bench
fd = lib.DevBuf(X.nbytes).upload(X)
sd = lib.DevBuf(1000 * 6144 * 4)
bd = lib.DevBuf(1000 * 4)
for fpl in (0, 1, 8, 64):
    for it in (3, 10):
        us, kus, nl = G.bench(fd, 1000, sd, None, fpl, it)
    print(f'frames_per_launch {fpl}: {us:.1f} us per 1000 frames -> {1000 / us * 1e6:.0f} frames/s, launches {nl}, per launch {kus:.2f} us')
out = sd.download(np.int32, (1000, 6144))
print('dev out eq host-api out', np.array_equal(out, SC))
G.set_precision(lib.GMM_FAST)
SCF, _ = G.score_frames(X)
d = (SCF.astype(np.int64) - SC)
print('fast mode max abs diff', np.abs(d).max(), 'mean', np.abs(d).mean())
for fpl in (0, 1):
    for it in (3, 10):
        us, kus, nl = G.bench(fd, 1000, sd, None, fpl, it)
    print(f'FAST frames_per_launch {fpl}: {us:.1f} us per 1000 frames -> {1000 / us * 1e6:.0f} frames/s')
