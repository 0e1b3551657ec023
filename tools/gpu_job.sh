export S3A_ON_GPU_BOX=1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4i_pytest_gpu.txt 2>&1; grep -n "passed\|failed" gpurun_out/r4i_pytest_gpu.txt | tail -2 | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
bash tools/gpu_round4.sh i 2>&1 | tail -22
