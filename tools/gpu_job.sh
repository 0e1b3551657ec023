export S3A_ON_GPU_BOX=1
cp cmusphinx_amd/libcmusphinx_amd.so /tmp/lib_base.so
cp cmusphinx_amd/variants/lib_nosurv.so cmusphinx_amd/libcmusphinx_amd.so
timeout 1500 python -m pytest tests/test_gpu_pheur.py -q > gpurun_out/pheur_nosurv.txt 2>&1; tail -12 gpurun_out/pheur_nosurv.txt | cut -c1-200
cp /tmp/lib_base.so cmusphinx_amd/libcmusphinx_amd.so
