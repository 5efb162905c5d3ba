"""Where the time of ONE replayed GUM posterior call goes (host side): python tools/is_call_profile.py [particles] [calls]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
GUM, _ = bench.api_models()
model = GUM()
torch.manual_seed(123)
with contextlib.redirect_stdout(io.StringIO()):
    model.learn_inference_network(num_traces=16 * 1024, inference_network=InferenceNetwork.LSTM,
                                  observe_embeddings={'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, batch_size=1024, lstm_dim=512, seed=1)
IC = InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK
obs = [{'obs0': 8, 'obs1': 9}, {'obs0': 7.5, 'obs1': 8.25}, {'obs0': 8.6, 'obs1': 9.4}]
for i in range(6):
    p = model.posterior_results(n, IC, observe=obs[i % 3], lock_step=True, seed=i)
assert getattr(p, 'replayed_plan', False) or os.environ.get('PP_IS_PLAN') == '0'
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(calls):
    p = model.posterior_results(n, IC, observe=obs[i % 3], lock_step=True, seed=10 + i)
    _ = p.effective_sample_size
torch.cuda.synchronize()
print('wall per call %.1f us' % ((time.perf_counter() - t0) / calls * 1e6))
# the same loop with 1000 particles: the host floor (the device chain shrinks to its latency)
for m in (1000,):
    for i in range(6):
        model.posterior_results(m, IC, observe=obs[i % 3], lock_step=True, seed=i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(calls):
        p = model.posterior_results(m, IC, observe=obs[i % 3], lock_step=True, seed=10 + i)
        _ = p.effective_sample_size
    torch.cuda.synchronize()
    print('wall per call at %d particles %.1f us' % (m, (time.perf_counter() - t0) / calls * 1e6))
pr = cProfile.Profile()
pr.enable()
for i in range(calls):
    p = model.posterior_results(n, IC, observe=obs[i % 3], lock_step=True, seed=10 + i)
    _ = p.effective_sample_size
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(int(os.environ.get('PROFILE_ROWS', '14')))
print('\n'.join(l[:150] for l in s.getvalue().splitlines()[:18 + int(os.environ.get('PROFILE_ROWS', '14'))]))
