cd /root/repo
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for t in 768 1024; do echo "target $t"; PP_GROUP_BLOCKS=$t python bench.py --steps 300 --warmup 30 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done
cd /tmp && export TMPDIR=/tmp
for t in 768; do
PP_GROUP_BLOCKS=$t rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_e$t -o e -- python /root/repo/bench.py --steps 200 --warmup 20 --no-cpu-baseline > /root/repo/gpurun_out/e.log 2>&1; echo "target $t"; python /root/repo/tools/prof_top.py /root/repo/gpurun_out/prof_e$t/e_results.db "%" 2>&1 | grep -v "at::" | head -24
done
