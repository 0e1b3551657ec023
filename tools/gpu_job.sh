export S3A_ON_GPU_BOX=1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_queue.py tests/test_abi.py -q -x 2>&1 | grep -n "passed\|failed" | tail -1
python bench.py --steps 1 --warmup 1 --no-ps --no-wide-beam --no-scoring > gpurun_out/last_bench.json 2> gpurun_out/last_bench.err; python -c "
import json; r=json.load(open('gpurun_out/last_bench.json')); print(r['value'], r['identical_to_reference'], r['roofline']['kernel'], r['cpu_baseline']['value'])"
