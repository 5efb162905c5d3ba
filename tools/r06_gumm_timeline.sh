#!/bin/bash
# Round 6: the Marsaglia posterior call (200 000 particles, lock step): branch decisions polled vs copied, kernel timeline + cProfile
out=$PWD/gpurun_out/${1:-r06g}; mkdir -p $out; REPO=$PWD
for x in 1 0 1 0; do
  PP_IS_PART_POLL=$x python tools/gumm_call_bench.py 200000 12 2>/dev/null | tail -1 >> $out/gumm_poll_ab.txt
done
cat $out/gumm_poll_ab.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $out/tl
rocprofv3 --kernel-trace -d $out/tl -o p -- python $REPO/tools/gumm_timeline.py 200000 > $out/timeline.log 2>&1
python $REPO/tools/rocprof_summary.py $out/tl/p_results.db $out/gumm_call_timeline.csv sequence 'is_fused_kernel<1' > /dev/null 2>&1
rm -rf $out/tl
cd $REPO
python tools/gumm_lockstep_profile.py > $out/gumm_cprofile.txt 2>&1
head -5 $out/gumm_call_timeline.csv; wc -l $out/gumm_call_timeline.csv
