"""BASELINE.json configs[0]: "pocketsphinx en-us PTM semi-continuous model, bundled test/data/goforward.raw, CPU
ps_decode_raw (plumbing, no GPU)".  The UNMODIFIED pocketsphinx (oracle/_ref/libpsref.so) through its utterance API --
ps_init / ps_decode_raw / ps_get_hyp (oracle/ref_ps_raw.c) -- on the reference's own audio with the US-English PTM
model of the checkout (pocketsphinx-extra/model/hmm/en_US/hub4_wsj_ptm256_3s_8k.cd_ptm_5000) and
goforward.fsg: the hypothesis pocketsphinx/test/unit/test_fsg.c:21-70 expects, and the path score the survey's probe
recorded.  Runs where /root/reference exists (model and audio are not redistributed); nothing of ours is in the path:
it pins what the ps_decoder_t utterance API returns for the secondary boundary (ps_mgaufuncs_t, tests/test_gpu_psms.py).
"""
import os
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference"
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_ps_raw")
PTM = f"{REF}/pocketsphinx-extra/model/hmm/en_US/hub4_wsj_ptm256_3s_8k.cd_ptm_5000"
SC = f"{REF}/pocketsphinx/model/hmm/en_US/hub4wsj_sc_8k"


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (model + audio are not redistributed)")
@pytest.mark.parametrize("hmm,score", [(PTM, -3003), (SC, None)])
def test_ps_decode_raw_goforward(hmm, score):
    assert os.path.exists(EXE), "make -C oracle ref"
    p = subprocess.run([EXE, hmm, f"{REF}/pocketsphinx/test/data/goforward.fsg", f"{REF}/pocketsphinx/model/lm/en/turtle.dic",
                        f"{REF}/pocketsphinx/test/data/goforward.raw"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-400:]
    assert "HYP: go forward ten meters (goforward " in p.stdout
    if score is not None:       # the PTM model: the path score of the survey's probe
        assert f"(goforward {score})" in p.stdout
