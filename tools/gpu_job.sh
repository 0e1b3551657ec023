export S3A_ON_GPU_BOX=1
cp cmusphinx_amd/libcmusphinx_amd.so /tmp/lib_base.so
run() { python bench.py --plain > gpurun_out/plain_v.json 2> gpurun_out/plain_v.err; python -c "
import json; r=json.load(open('gpurun_out/plain_v.json')); print('$1', r['value'], r['identical_to_reference'])" 2>&1 | tail -1; }
for v in base e11 e13 e9; do
if [ $v = base ]; then cp /tmp/lib_base.so cmusphinx_amd/libcmusphinx_amd.so; else cp cmusphinx_amd/variants/lib_$v.so cmusphinx_amd/libcmusphinx_amd.so; fi
run $v
done
cp /tmp/lib_base.so cmusphinx_amd/libcmusphinx_amd.so
export S3A_UTT_GEVAL=61; run geval61
export S3A_UTT_GEVAL=53; run geval53
unset S3A_UTT_GEVAL; run base
