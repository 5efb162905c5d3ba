"""posterior_results through the API, GUM only (for kernel traces of ONE posterior call): python tools/is_api_probe.py [calls]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyprob_amd import lib as L
lib = L.load()
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rec, dt, units = bench.api_posterior_bench(lib, torch.device('cuda:0'), 512, 1000000, calls, 3, 'gum')
print(rec['particles_per_sec'], rec['ms_per_call'], rec.get('particle_kernels', {}).get('us_per_call'))
