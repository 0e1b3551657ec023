cp cmusphinx_amd/libcmusphinx_amd.so /tmp/lib_base.so
for v in gl512 gl1024 gl2048 sweep gl512; do
if [ $v = sweep ]; then cp /tmp/lib_base.so cmusphinx_amd/libcmusphinx_amd.so; X="--variant resolve_sweep=1"; else cp cmusphinx_amd/variants/lib_$v.so cmusphinx_amd/libcmusphinx_amd.so; X=""; fi
python bench.py --plain $X > gpurun_out/plain_v.json 2> gpurun_out/plain_v.err; python -c "
import json; r=json.load(open('gpurun_out/plain_v.json')); print('$v', r['value'], r['identical_to_reference'])" 2>&1 | tail -1
done
cp /tmp/lib_base.so cmusphinx_amd/libcmusphinx_amd.so
