timeout 300 python -m pytest tests/test_gpu_compact.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/ab_bench.sh r02_q "PP_CELL_LEAN=0" "PP_GROUP_BLOCKS_ASYNC=96" "PP_GROUP_BLOCKS_ASYNC=128" "PP_GROUP_BLOCKS_ASYNC=256" "PP_GROUP_BLOCKS_ASYNC=384" "PP_SPLIT_BUDGET=128" "PP_SPLIT_BUDGET=512" "PP_GEMM_KW=1" "PP_GEMM_KW=2" "PP_OBS_TPW=2" "PP_OBS_TPW=4" 2>&1 | grep -v amdgpu.ids
