export S3A_ON_GPU_BOX=1
python -m pytest tests/test_gpu_psfwd.py tests/test_gpu_psfwd_synth.py -q -x 2>&1 | tail -2
bash tools/psfwd_variants.sh "base:512" 2>&1 | tail -2
bash tools/psfwd_pmc.sh gpurun_out/pspmc2 256 128 2>&1 | grep -A8 "k_psf_queue"
