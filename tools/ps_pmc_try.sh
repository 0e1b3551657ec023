#!/bin/bash
# does pocketsphinx's device search survive rocprofv3 --pmc with N lanes?  (experiment 10 / 21: it does not from some lane count on)
export S3A_ON_GPU_BOX=1 TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for L in ${1:-64 512}; do
python3 - $L <<'PY'
import sys, os
sys.path.insert(0, 'tests')
import psfwd_cases as P
from pathlib import Path
L = sys.argv[1]
tp = Path('/tmp/pspmc'); tp.mkdir(exist_ok=True)
args = P.cont_args(tp) + P.FIRST_PASS_ONLY + ["-fresh", "yes", "-batch", L, "-queue", "yes", "-hyp", "/tmp/pspmc/h", "-hypseg", "/tmp/pspmc/s"]
open('/tmp/pspmc/cmd', 'w').write(" ".join([os.path.join(P.REF, "ref_ps_amdfwd")] + args))
PY
( cd /tmp; rm -rf /tmp/prof_ps; timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_ps -o pmc -- $(cat /tmp/pspmc/cmd) > /tmp/pspmc/out_$L.txt 2>&1; echo "lanes $L rc=$?" )
grep -n "served by\|batch of\|FATAL\|ERROR\|Segmentation\|hipMalloc\|s3a_" /tmp/pspmc/out_$L.txt | head -8 | cut -c1-220
grep -n "^    @" /tmp/pspmc/out_$L.txt | head -12 | cut -c1-160
find /tmp/prof_ps -name "*counter_collection.csv" 2>/dev/null | head -2
done
