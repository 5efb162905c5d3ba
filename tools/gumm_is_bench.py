"""Importance sampling with the inference network for a program with stochastic control flow
(GaussianUnknownMeanMarsaglia), particles/s of the three executors:
  lock-step   posterior_results(lock_step=True): the program rewritten with tensor conditions, one forward() per path
  coroutines  posterior_results(lock_step=False): the reference's program AS WRITTEN (`while float(s) >= 1`), one greenlet
              per particle, parked at `sample`, served in address-grouped batches (pyprob_amd/coroutine.py)
  per trace   posterior_results(lock_step='per_trace'): the reference's loop, batch-1 network calls
usage: python tools/gumm_is_bench.py [lock-step particles] [coroutine particles]"""
import cProfile
import os
import pstats
import sys
import time
import warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from models import GaussianWithUnknownMeanMarsaglia, GaussianWithUnknownMeanMarsagliaLockStep
from pyprob_amd.state import InferenceEngine
from pyprob_amd.state import InferenceNetwork
LSTM = InferenceNetwork.LSTM
IC = InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK
EMB = {'obs0': {'dim': 32}, 'obs1': {'dim': 32}}
warnings.simplefilter('ignore')
obs = {'obs0': 4, 'obs1': 5}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 20000

torch.manual_seed(1)
model = GaussianWithUnknownMeanMarsagliaLockStep()
model.learn_inference_network(inference_network=LSTM, num_traces=20000, observe_embeddings=EMB, batch_size=256, lstm_dim=512, seed=1)
model.posterior_results(10000, IC, observe=obs, lock_step=True, seed=1)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    post = model.posterior_results(n, IC, observe=obs, lock_step=True, seed=2 + rep)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print('lock-step: %d particles, %d control-flow paths, %.1f ms -> %.2f M particles/s; mean %.3f std %.3f ESS %.0f' % (
        n, post.num_paths, (t1 - t0) * 1e3, n / (t1 - t0) / 1e6, post.mean, post.stddev, post.effective_sample_size))

torch.manual_seed(1)
ref_style = GaussianWithUnknownMeanMarsaglia()      # `while float(s) >= 1:` - the reference's source
ref_style.learn_inference_network(inference_network=LSTM, num_traces=20000, observe_embeddings=EMB, batch_size=256, lstm_dim=512, seed=1)
ref_style.posterior_results(500, IC, observe=obs, lock_step=False)
for rep in range(2):
    t0 = time.perf_counter()
    post = ref_style.posterior_results(nc, IC, observe=obs, lock_step=False, seed=rep)
    t1 = time.perf_counter()
    st = post.coroutine_stats
    print('coroutines: %d particles in %.2f s -> %.0f particles/s (%d rounds, %d group calls, %d statements); mean %.3f ESS %.0f' % (
        nc, t1 - t0, nc / (t1 - t0), st['rounds'], st['group_calls'], st['statements'], post.mean, post.effective_sample_size))
for workers in (16, 32, 64):
    big = int(sys.argv[3]) if len(sys.argv) > 3 else 200000
    for rep in range(3):      # the first call forks the persistent workers of this (program, worker count)
        t0 = time.perf_counter()
        post = ref_style.posterior_results(big, IC, observe=obs, lock_step=False, seed=3 + rep, num_workers=workers)
        t1 = time.perf_counter()
        st = post.coroutine_stats
        print('coroutines in %d worker processes (call %d%s): %d particles in %.2f s -> %.0f particles/s (%d rounds, %d group calls); '
              'mean %.3f ESS %.0f' % (st['workers'], rep + 1, ', forks the workers' if rep == 0 else '', big, t1 - t0,
                                      big / (t1 - t0), st['rounds'], st['group_calls'], post.mean, post.effective_sample_size))
        print('      inside run(): %.2f s; parent waited %.2f s for the workers, served %.2f s, replied %.2f s'
              % (st['seconds'], st['parent_seconds']['wait'], st['parent_seconds']['serve'], st['parent_seconds']['send']))
from pyprob_amd.coroutine import close_worker_pools
close_worker_pools()
if os.environ.get('PP_PROFILE'):
    pr = cProfile.Profile()
    pr.enable()
    ref_style.posterior_results(5000, IC, observe=obs, lock_step=False, seed=9)
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
t0 = time.perf_counter()
one = ref_style.posterior_results(300, IC, lock_step='per_trace', observe=obs)
t1 = time.perf_counter()
print('one particle per forward() (reference loop): %.0f particles/s (mean %.3f)' % (300 / (t1 - t0), one.mean))
