"""Oracle pinning, part 1: integer log-domain arithmetic.

Checks oracle/s3o_logmath.c against (a) tables and values produced by the
unmodified reference (tests/golden/logmath.npz via oracle/_ref/ref_dump) and
(b) the known answers in the reference's own unit tests:
  sphinxbase/test/unit/test_logmath/test_log_int16.c  (log(1e-150) = -3454050,
      log(42) = 37378 at base 1.0001)
  sphinx3/src/tests/unit_tests/test_logs3/_testlogs3_1.{test,res}  (= 79150)
  sphinx3/src/tests/unit_tests/test_hmm/_testhmm_tidigits.res
      ("Log-Add table size = 99042 x 2 >> 0" at base 1.0001)
"""
import numpy as np
import pytest

import oracle_lib as O
from conftest import golden

CASES = [(1.0003, 0), (1.0001, 0), (1.0001, 1), (1.0001, 8), (1.002, 0)]


@pytest.mark.parametrize("base,shift", CASES)
def test_table_matches_reference(base, shift):
    g = golden("logmath.npz")
    key = f"b{base}_s{shift}"
    lm = O.OracleLogMath(base, shift, 1)
    assert np.array_equal(lm.table.astype(np.int64), g[key + "_table"].astype(np.int64))
    known = g[key + "_known"]
    assert (len(lm.table), lm.width, shift, lm.zero) == tuple(known[:4])


@pytest.mark.parametrize("base,shift", CASES)
def test_known_values_match_reference(base, shift):
    g = golden("logmath.npz")
    k = g[f"b{base}_s{shift}_known"]
    kf = g[f"b{base}_s{shift}_knownf"]
    lm = O.OracleLogMath(base, shift, 1)
    L = O.lib()
    n = len(lm.table)
    got = [lm.log(1e-150), lm.log(42.0), lm.log(1e-48),
           lm.add(lm.log(1e-48), lm.log(5e-48)), lm.add(lm.log(1e-48), lm.log(42.0)),
           L.s3o_logmath_log10_to_log(lm.p, -7.0), L.s3o_logmath_ln_to_log(lm.p, -123.456),
           lm.logs3(1e-80), lm.logs3(0.5), lm.add(O.LOGPROB_ZERO, -12345),
           lm.add(-12345, O.LOGPROB_ZERO), lm.add(-100, -100 - n)]
    assert got == list(k[4:16])
    assert L.s3o_logmath_log_to_ln(lm.p, O.LOGPROB_ZERO) == kf[0]
    assert L.s3o_logmath_exp(lm.p, -5000) == kf[1]
    assert L.s3o_logmath_log_to_ln(lm.p, -79150) == kf[2]


def test_sphinxbase_known_answers():
    lm = O.OracleLogMath(1.0001, 0, 1)
    assert lm.log(1e-150) == -3454050       # test_log_int16.c
    assert lm.log(42) == 37378
    assert len(lm.table) == 99042 and lm.width == 2   # _testhmm_tidigits.res line 2


def test_sphinx3_logs3_known_answer():
    lm = O.OracleLogMath(1.0003, 0, 1)
    v = int(O.lib().s3o_logmath_log10_to_log(lm.p, 0.8202) * 10.5 - lm.logs3(0.02))
    assert v == 79150                        # _testlogs3_1.res
    assert len(lm.table) == 29356 and lm.width == 2


def test_add_semantics():
    lm = O.OracleLogMath(1.0003, 0, 1)
    assert lm.zero == -536870912
    # x <= zero returns y, first (logmath.c:398-401) -- even when y is also "zero"
    assert lm.add(O.LOGPROB_ZERO, O.LOGPROB_ZERO - 5) == O.LOGPROB_ZERO - 5
    assert lm.add(-5, -5) == -5 + int(lm.table[0])
    assert lm.add(-5, -5 - 29355) == -5 + int(lm.table[29355])
    assert lm.add(-5, -5 - 29356) == -5
    # not associative: the oracle must add components in order
    a, b, c = -70000, -70100, -72000
    assert isinstance(lm.add(lm.add(a, b), c), int)
    assert O.lib().s3o_logs3(lm.p, 0.0) == O.LOGPROB_ZERO
    assert O.lib().s3o_logs3(lm.p, -1.0) == O.LOGPROB_ZERO
