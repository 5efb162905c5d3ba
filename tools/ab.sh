cd /root/repo
timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for m in 0 1 0 1; do echo "aux $m"; PP_AUX_STREAM=$m python tools/host_time.py | tail -1; done
for m in 0 1; do PP_AUX_STREAM=$m python bench.py --workload train_gumm --steps 200 --warmup 30 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done
PP_AUX_STREAM=1 python bench.py --graph 1 --steps 200 --warmup 20 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph', d['ms_per_step'], d['value'])"
