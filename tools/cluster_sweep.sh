#!/bin/bash
# ku_frames' cluster size against the lane count, one utterance per lane (what a rank of a strong-scaling run gets): bench.py --plain per
# (lanes, workgroups per lane); C = 0 is the library's choice.  usage (GPU box): bash tools/cluster_sweep.sh "96 128" "0 2 3 4"
mkdir -p gpurun_out/csweep; export S3A_ON_GPU_BOX=1
for L in ${1:-32 64 128 192}; do
  for C in ${2:-1 2 3 4 6 8}; do
    if [ $((L * C)) -gt 480 ]; then continue; fi
    S3A_UTT_CLUSTER=$C timeout 300 python bench.py --plain --no-cpu --utts $L --lanes $L --steps 2 --warmup 1 > gpurun_out/csweep/s_${L}_$C.json 2> gpurun_out/csweep/s_${L}_$C.err
    python3 -c "
import json
try:
    d=json.loads(open('gpurun_out/csweep/s_${L}_$C.json').read().strip().splitlines()[-1]); print($L, $C, d['value'], d['identical_to_reference']['hyp'], d['kernels']['ku_frames']['ms_per_step'])
except Exception as e: print($L, $C, 'failed', e)"
  done
done
