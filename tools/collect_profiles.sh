# After a GPU call that ran tools/profile_round.sh / profile_is_step.sh with <tag>: copy what the judge should read from
# gpurun_out/ (scratch) into profiles/ UNDER THE SAME NAMES the JSON documents cite, and install the documents bench.py quotes.
#   bash tools/collect_profiles.sh <tag>
TAG=$1
for f in gpurun_out/${TAG}_*.csv gpurun_out/${TAG}_*bench*.json gpurun_out/${TAG}_*.jsonl gpurun_out/${TAG}_*tests*.log gpurun_out/${TAG}_*.txt; do
  [ -f "$f" ] && cp "$f" profiles/
done
for d in r06_kernel_avgs r06_pmc_traffic r06_is_pmc_traffic r06_is_fused_valu r06_gumm_traffic r06_mfma_busy; do  # (documents bench.py quotes)
  [ -f gpurun_out/${TAG}_$d.json ] && cp gpurun_out/${TAG}_$d.json profiles/$d.json
done
ls profiles/${TAG}_* 2>/dev/null | wc -l
