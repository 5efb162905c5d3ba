"""Where does a lock-step posterior call of a program with stochastic control flow spend its time? (GUMM, H = 512, N particles)
host: cProfile of one call; device + host operators: torch.profiler of one call.
usage: python tools/gumm_lockstep_profile.py [particles] [out_prefix]"""
import cProfile
import io
import os
import pstats
import sys
import time
import contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyprob_amd.state import InferenceEngine, InferenceNetwork

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
out = sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/gumm_lockstep'
GUM, GUMM = bench.api_models()
model = GUMM()
torch.manual_seed(123)
with contextlib.redirect_stdout(io.StringIO()):
    model.learn_inference_network(num_traces=96 * 1024, inference_network=InferenceNetwork.LSTM,
                                  observe_embeddings={'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, batch_size=1024, lstm_dim=512, seed=1)
IC = InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK
obs = {'obs0': 4, 'obs1': 5}
import warnings
warnings.simplefilter('ignore')
for i in range(3):
    model.posterior_results(n, IC, observe=obs, lock_step=True, seed=i)
torch.cuda.synchronize()
ts = []
for i in range(5):
    t0 = time.perf_counter()
    post = model.posterior_results(n, IC, observe=obs, lock_step=True, seed=10 + i)
    _ = post.effective_sample_size
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print('wall per call (ms):', [round(t * 1e3, 2) for t in ts], 'paths', post.num_paths)
pr = cProfile.Profile()
pr.enable()
post = model.posterior_results(n, IC, observe=obs, lock_step=True, seed=77)
_ = post.effective_sample_size
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45)
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(60)
open(out + '_cprofile.txt', 'w').write(s.getvalue())
pr.dump_stats(out + '_cprofile.pstats')
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    post = model.posterior_results(n, IC, observe=obs, lock_step=True, seed=78)
    _ = post.effective_sample_size
    torch.cuda.synchronize()
open(out + '_torchprof_cuda.txt', 'w').write(prof.key_averages().table(sort_by='cuda_time_total', row_limit=40, max_name_column_width=90))
open(out + '_torchprof_cpu.txt', 'w').write(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=40, max_name_column_width=90))
print('written', out)
