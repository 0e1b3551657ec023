export S3A_ON_GPU_BOX=1
python -m pytest tests/test_gpu_queue.py tests/test_gpu_adcin.py tests/test_gpu_psfwd.py tests/test_gpu_psfwd_synth.py -q 2>&1 | tail -8
bash tools/psfwd_variants.sh "base:256 512" 2>&1 | tail -4
