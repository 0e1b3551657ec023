export S3A_ON_GPU_BOX=1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4j_pytest_gpu.txt 2>&1; grep -n "passed\|failed" gpurun_out/r4j_pytest_gpu.txt | tail -2 | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
bash tools/gpu_round4.sh j 2>&1 | tail -22
cp cmusphinx_amd/libcmusphinx_amd.so /tmp/lib_base.so
run() { python bench.py --plain > gpurun_out/plain_v.json 2> gpurun_out/plain_v.err; python -c "
import json; r=json.load(open('gpurun_out/plain_v.json')); print('$1', r['value'], r['identical_to_reference'])" 2>&1 | tail -1; }
for v in hist11 hist13 base; do
if [ $v = base ]; then cp /tmp/lib_base.so cmusphinx_amd/libcmusphinx_amd.so; else cp cmusphinx_amd/variants/lib_$v.so cmusphinx_amd/libcmusphinx_amd.so; fi
run $v
done
cp /tmp/lib_base.so cmusphinx_amd/libcmusphinx_amd.so
