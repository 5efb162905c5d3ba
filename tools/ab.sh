cd /root/repo
timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/host_time.py | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_g -o e -- python /root/repo/bench.py --steps 100 --warmup 20 --no-cpu-baseline > /root/repo/gpurun_out/e.log 2>&1; python /root/repo/tools/prof_top.py /root/repo/gpurun_out/prof_g/e_results.db "%" 2>&1 | grep "head_tail\|obs_embed\|cell_fwd" | head -4
rm -rf /root/repo/gpurun_out/prof_g
