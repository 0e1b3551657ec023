export S3A_ON_GPU_BOX=1
python -m pytest tests/test_gpu_psms.py tests/test_gpu_psfwd.py tests/test_gpu_psfwd_synth.py tests/test_gpu_dag.py -q 2>&1 | tail -8
bash tools/psfwd_variants.sh "base:512" 2>&1 | tail -2
grep -h "scor\|ms on" gpurun_out/psvar/base/q512.log | tail -5
