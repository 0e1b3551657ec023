export S3A_ON_GPU_BOX=1
python -m pytest tests/test_gpu_dropin.py tests/test_gpu_uttdec.py tests/test_gpu_pheur.py tests/test_gpu_queue.py -q -x > gpurun_out/shared2_tests.txt 2>&1; tail -2 gpurun_out/shared2_tests.txt | cut -c1-200
cp cmusphinx_amd/libcmusphinx_amd.so /tmp/lib_base.so
for v in base prev base prev; do
X=""
if [ $v = base ]; then cp /tmp/lib_base.so cmusphinx_amd/libcmusphinx_amd.so; else cp cmusphinx_amd/variants/lib_$v.so cmusphinx_amd/libcmusphinx_amd.so; fi
python bench.py --plain $X > gpurun_out/plain_v.json 2> gpurun_out/plain_v.err; python -c "
import json; r=json.load(open('gpurun_out/plain_v.json')); print('$v', r['value'], r['identical_to_reference'])" 2>&1 | tail -1
done
cp /tmp/lib_base.so cmusphinx_amd/libcmusphinx_amd.so
