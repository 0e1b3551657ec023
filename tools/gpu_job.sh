export S3A_ON_GPU_BOX=1
bash tools/pmc_only.sh r4pmc2 > gpurun_out/r4pmc2.log 2>&1; tail -3 gpurun_out/r4pmc2.log | cut -c1-200
[ -s gpurun_out/r4pmc2/pmc_traffic.json ] && cp gpurun_out/r4pmc2/pmc_traffic.json profiles/pmc_traffic.json
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4g_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r4g_pytest_gpu.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
bash tools/gpu_round4.sh g 2>&1 | tail -30
