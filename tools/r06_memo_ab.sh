#!/bin/bash
out=$PWD/gpurun_out/${1:-r06h}; mkdir -p $out
for x in 1 0 1 0 1 0; do
  PP_IS_MEMO_FAST=$x python tools/gumm_call_bench.py 200000 16 2>/dev/null | tail -1 | sed "s/^/PP_IS_MEMO_FAST=$x /" >> $out/gumm_memo_ab.txt
done
cat $out/gumm_memo_ab.txt
python -m pytest tests/test_gpu_rows.py tests/test_gpu_posterior_h512.py tests/test_gpu_is_fused.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -2
