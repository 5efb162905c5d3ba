cd /root/repo
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --workload is --steps 50 --warmup 5 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
python bench.py --steps 300 --warmup 30 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
