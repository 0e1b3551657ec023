#!/bin/bash
# a few short bench variants on the GPU box: each line of stdin = "ENV... -- bench args"
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; make -s -C oracle oracle
while IFS= read -r line; do
  [ -z "$line" ] && continue
  envs="${line%%--*}"; args="${line#*--}"
  out=$(env $envs python bench.py --steps 1 --warmup 1 --no-cpu --no-scoring $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print(d['value'], d['identical_to_reference']['hyp'], d['config']['groups_rank0'], d.get('strong_scaling_projection',{}).get('frames_per_sec_per_gpu'), {n:k[n]['avg_launch_us'] for n in ('ku_resolve','ku_emit_word','ku_hmm_eval','ku_enter3_mark')})")
  echo "[$line] $out"
done
