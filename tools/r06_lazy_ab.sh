#!/bin/bash
# Round 6: the posterior call with forward() in every call (PP_IS_PLAN=0: what runs under pyprob as the host) - the observe embedding
# deferred into the first statement's launch (PP_IS_LAZY_INIT=1, default) against a separate pp_is_init launch (0)
out=$PWD/gpurun_out/${1:-r06k}; mkdir -p $out
for x in 1 0 1 0; do
  PP_IS_PLAN=0 PP_IS_LAZY_INIT=$x PROFILE_ROWS=1 python tools/is_call_profile.py 1000000 400 2>/dev/null | grep "wall per call" | tr '\n' ' ' | sed "s/^/PP_IS_LAZY_INIT=$x /" >> $out/noplan_lazy_ab.txt; echo >> $out/noplan_lazy_ab.txt
done
cat $out/noplan_lazy_ab.txt
python -m pytest tests/test_gpu_is_fused.py tests/test_gpu_logweight.py tests/test_gpu_posterior_h512.py tests/test_gpu_model.py tests/test_gpu_binding_session.py tests/test_gpu_ten_thousand_particles.py -m gpu -x -q > $out/tests.log 2>&1; grep -n "passed\|failed" $out/tests.log
