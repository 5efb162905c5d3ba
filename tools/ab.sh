cd /root/repo
for b in 512 256; do for g in 768 512 384 256; do echo "budget $b group $g"; PP_SPLIT_BUDGET=$b PP_GROUP_BLOCKS=$g python bench.py --steps 300 --warmup 30 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done; done
