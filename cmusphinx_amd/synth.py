"""Synthetic acoustic models and feature streams in genuine S3 file formats.

The hub4 CD-GMM model named by BASELINE.json is not distributable (its mdef /
means / variances are absent from the reference checkout, SURVEY.md "facts"
item 1), but its dimensions are fully known: 6144 senones (144 CI first),
8 Gaussians, 39-dim 1s_c_d_dd features, 48 3-state transition matrices.  This
module writes models of that SHAPE with seeded synthetic values, in the
reference's own file formats, so that the unmodified reference, the CPU oracle
restatement and the HIP path all load the very same files.

Determinism: everything derives from numpy's PCG64 with an explicit seed and
float32 arithmetic in a fixed order; fixtures additionally record a checksum
of the generated arrays so a silent RNG change is caught by the tests.
"""
from __future__ import annotations

import os
import zlib

import numpy as np

from . import s3io

HUB4 = dict(n_sen=6144, n_ci_sen=144, n_comp=8, veclen=39, n_tmat=48, n_emit=3, seed=0x5EED0001)
WSJ_STRESS = dict(n_sen=8000, n_ci_sen=150, n_comp=32, veclen=39, n_tmat=50, n_emit=3, seed=0x5EED0002)

# per-dimension spread of real 1s_c_d_dd features (cepstra, deltas, delta-deltas):
# c0 dominates, higher cepstra and the dynamic streams are progressively smaller.
_DIM_SCALE = np.concatenate([
    np.array([4.0, 1.2, 0.9, 0.8, 0.7, 0.6, 0.55, 0.5, 0.45, 0.4, 0.38, 0.35, 0.33]),
    np.array([1.6, 0.6, 0.5, 0.45, 0.4, 0.36, 0.33, 0.3, 0.28, 0.26, 0.24, 0.22, 0.2]),
    np.array([0.9, 0.35, 0.3, 0.27, 0.25, 0.22, 0.2, 0.19, 0.18, 0.17, 0.16, 0.15, 0.14]),
]).astype(np.float32)


def array_crc(*arrays) -> int:
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return c


def make_model(n_sen, n_ci_sen, n_comp, veclen, n_tmat, n_emit, seed, degenerate=False):
    """Return dict(mean, var, mixw, tmat, cd2cisen) of raw (file-domain) arrays.

    degenerate=True plants the edge cases the reference's loader handles
    (cont_mgau.c:700-816, 624-661): variances below the floor, all-zero
    variance vectors and NaN means (component removed by mgau_uninit_compact),
    all-zero mixture-weight rows, single zero weights, tiny weights below the
    mixw floor.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    scale = np.resize(_DIM_SCALE, veclen).astype(np.float32)
    # senones of one CI phone cluster around a per-phone centre
    n_ciphone = max(n_ci_sen // n_emit, 1)
    centre = (rng.standard_normal((n_ciphone, veclen)).astype(np.float32) * scale)
    cd2cisen = np.empty(n_sen, np.int16)
    cd2cisen[:n_ci_sen] = np.arange(n_ci_sen)
    if n_sen > n_ci_sen:
        # CD senones contiguous per parent CI phone (mdef.h:206-208), state position cycling
        n_cd = n_sen - n_ci_sen
        parent_phone = np.sort(rng.integers(0, n_ciphone, n_cd))
        state = np.arange(n_cd) % n_emit
        cd2cisen[n_ci_sen:] = (parent_phone * n_emit + state).astype(np.int16)
    phone_of = (cd2cisen.astype(np.int32) // n_emit) % n_ciphone
    mean = centre[phone_of][:, None, :] + \
        rng.standard_normal((n_sen, n_comp, veclen)).astype(np.float32) * (0.6 * scale)
    mean = mean.astype(np.float32)
    # variances: log-uniform around (0.5*scale)^2
    logv = rng.uniform(np.log(0.25), np.log(4.0), (n_sen, n_comp, veclen)).astype(np.float32)
    var = (np.exp(logv) * (0.5 * scale) ** 2).astype(np.float32)
    mixw = rng.dirichlet(np.ones(n_comp), n_sen).astype(np.float32)
    # unnormalised counts, as trainers write them
    mixw = (mixw * rng.uniform(50.0, 5000.0, (n_sen, 1))).astype(np.float32)

    if degenerate:
        k = max(n_sen // 40, 2)
        idx = rng.choice(n_sen, size=6 * k, replace=False)
        var[idx[0:k], 1, 3] = 1e-6                       # below -varfloor 1e-4
        var[idx[k:2 * k], 2, :] = 0.0                    # zero vector -> component removed
        mean[idx[2 * k:3 * k], 0, 5] = np.nan            # NaN mean   -> component removed
        mixw[idx[3 * k:4 * k], :] = 0.0                  # all-zero row -> S3_LOGPROB_ZERO
        mixw[idx[4 * k:5 * k], n_comp - 1] = 0.0         # single zero weight
        mixw[idx[5 * k:6 * k], 0] = 1e-12                # below -mixwfloor after normalisation? (raw)
        var[idx[0], :, :] = 0.0                          # a senone that loses every component

    # upper-triangular no-skip transition matrices, raw counts
    tmat = np.zeros((n_tmat, n_emit, n_emit + 1), np.float32)
    for i in range(n_emit):
        stay = rng.uniform(0.4, 0.9, n_tmat).astype(np.float32)
        tmat[:, i, i] = stay
        tmat[:, i, i + 1] = 1.0 - stay
    tmat *= rng.uniform(1e3, 1e6, (n_tmat, 1, 1)).astype(np.float32)
    return dict(mean=mean, var=var, mixw=mixw, tmat=tmat.astype(np.float32), cd2cisen=cd2cisen,
                n_ci_sen=n_ci_sen, n_emit=n_emit)


def write_model(dirpath, model, chksum=False):
    os.makedirs(dirpath, exist_ok=True)
    s3io.write_gau(os.path.join(dirpath, "means"), model["mean"], chksum)
    s3io.write_gau(os.path.join(dirpath, "variances"), model["var"], chksum)
    s3io.write_mixw(os.path.join(dirpath, "mixture_weights"), model["mixw"], chksum)
    s3io.write_tmat(os.path.join(dirpath, "transition_matrices"), model["tmat"], chksum)
    model["cd2cisen"].astype("<i2").tofile(os.path.join(dirpath, "cd2cisen.i16"))
    return dirpath


def make_features(model, n_frames, seed, rho=0.9):
    """AR(1) trajectory hopping between model means, so the per-frame best
    senones and the CI beam behave like speech."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mean = np.nan_to_num(model["mean"])
    n_sen, n_comp, veclen = mean.shape
    scale = np.resize(_DIM_SCALE, veclen).astype(np.float32)
    x = np.empty((n_frames, veclen), np.float32)
    cur = mean[rng.integers(n_sen), rng.integers(n_comp)].copy()
    tgt = cur.copy()
    hold = 0
    for t in range(n_frames):
        if hold == 0:
            tgt = mean[rng.integers(n_sen), rng.integers(n_comp)]
            hold = int(rng.integers(3, 12))
        hold -= 1
        cur = (rho * cur + (1.0 - rho) * tgt).astype(np.float32)
        x[t] = cur + rng.standard_normal(veclen).astype(np.float32) * (0.25 * scale)
    return x
