#!/bin/bash
# the pocketsphinx first pass with workgroup-size / register-budget variants of s3a_psfwd.hip (cmusphinx_amd/variants/lib_ps<NT>_<WPE>.so,
# built by hand: hipcc -DNT=.. -DPSF_WPE=.. + the other objects): 1024 utterances as a queue over L lanes
# usage: tools/psfwd_variants.sh "base:256 512" "ps512_4:512" ...
cd $(dirname $0)/..
O=gpurun_out/psvar; mkdir -p $O
cp cmusphinx_amd/libcmusphinx_amd.so /tmp/lib_base.so
for spec in "$@"; do
  v=${spec%%:*}; lanes=${spec#*:}
  if [ $v = base ]; then cp /tmp/lib_base.so cmusphinx_amd/libcmusphinx_amd.so; else cp cmusphinx_amd/variants/lib_$v.so cmusphinx_amd/libcmusphinx_amd.so; fi
  QUEUE="$lanes" bash tools/psfwd_bench.sh $O/$v 1024 1000 2>&1 | grep "queue lanes" | sed "s/^/$v /"
done
cp /tmp/lib_base.so cmusphinx_amd/libcmusphinx_amd.so
