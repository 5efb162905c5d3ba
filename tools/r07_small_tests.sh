cd /root/repo
mkdir -p gpurun_out/r07a; timeout 1500 python -m pytest tests/test_gpu_is_step_fused.py -x -q 2>&1 | tail -25 > gpurun_out/r07a/tests.log
