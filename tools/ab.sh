cd /root/repo
python bench.py --steps 300 --warmup 30 2>&1 | tail -1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
python bench.py --graph 1 --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
python __graft_entry__.py --smoke 2>&1 | tail -2
