timeout 600 python -m pytest tests/test_gpu_compact.py tests/test_gpu_path.py tests/test_gpu_holes.py tests/test_gpu_tail.py -x -q 2>&1 | grep -v amdgpu.ids | tail -30
