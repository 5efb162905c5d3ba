cd /root/repo
timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 300 --warmup 30 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
python bench.py --workload train_gumm --steps 200 --warmup 30 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
