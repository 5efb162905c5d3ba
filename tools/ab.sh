cd /root/repo
timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for m in 0 1; do PP_ASYNC_KGATHER=$m python bench.py --workload train_gumm --steps 200 --warmup 30 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done
PP_GEMM_TRACE=1 python bench.py --workload train_gumm --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep pp_gemm | sort | uniq -c | head
