export S3A_ON_GPU_BOX=1
cp cmusphinx_amd/libcmusphinx_amd.so /tmp/lib_base.so
cp cmusphinx_amd/variants/lib_wl512.so cmusphinx_amd/libcmusphinx_amd.so
python -m pytest tests -m gpu -q --deselect tests/test_gpu_psfwd.py --deselect tests/test_gpu_psfwd_synth.py --deselect tests/test_gpu_psms.py > gpurun_out/wl512_tests.txt 2>&1
tail -6 gpurun_out/wl512_tests.txt
cp /tmp/lib_base.so cmusphinx_amd/libcmusphinx_amd.so
