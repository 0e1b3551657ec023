"""The PMC traffic the bench line quotes (profiles/pmc_traffic.json) is stamped with the hash of the kernels' sources it was measured on
(tools/gpu_round6.sh); bench.py refuses a file whose stamp is another source's.  This test says so at commit time: a change under
cmusphinx_amd/csrc/ without new counter passes fails here, not silently in the line's `traffic`."""
import json
import os

import bench
from conftest import ROOT


def test_pmc_traffic_is_of_these_sources():
    p = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert p["csrc_hash"] == bench.csrc_hash(), "profiles/pmc_traffic.json was measured on other sources: tools/gpu_round6.sh"
    k = p["kernels"]["ku_frames"]
    assert k["dispatches_per_call"] >= 1 and 1.0e5 < k["hbm_bytes_per_lane_frame"] < 1.0e7


def test_pocketsphinx_traffic_is_of_the_lines_regime():
    p = json.load(open(os.path.join(ROOT, "profiles", "r6_pmc_ps.json")))
    assert p["lanes"] == 512 and p["utterances"] == 1024 and p["frames"] > 1000000
    assert any(n.startswith("k_psf_queue<3") for n in p["kernels"])
