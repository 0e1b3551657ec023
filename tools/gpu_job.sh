export S3A_ON_GPU_BOX=1
python -m pytest tests/test_gpu_psfwd.py tests/test_gpu_psfwd_synth.py -q -x 2>&1 | tail -3
bash tools/psfwd_variants.sh "base:256 512" 2>&1 | tail -4
grep -h "of which" gpurun_out/psvar/base/q512.log | tail -1
