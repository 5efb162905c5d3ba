cd /root/repo
for i in 1 2; do python tools/host_time.py | tail -1; PP_NO_T0_BLOCKS=1 python tools/host_time.py | tail -1; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_g -o e -- python /root/repo/bench.py --steps 100 --warmup 20 --no-cpu-baseline > /root/repo/gpurun_out/e.log 2>&1; python /root/repo/tools/prof_top.py /root/repo/gpurun_out/prof_g/e_results.db "%gemm%" 2>&1 | head -8
rm -rf /root/repo/gpurun_out/prof_g
