cd /root/repo
timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python tools/host_time.py | tail -2
PP_DBG_STAMP=1 python tools/timeline5.py | tail -1
