"""InferenceNetworkFeedForward (pyprob/nn/inference_network_feedforward.py) on the GUM benchmark shape: training step
and importance sampling rates with inputs resident in HBM, and end-to-end online training through
learn_inference_network (the reference's default network)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from helpers import synthetic_gum_arrays
from models import GaussianWithUnknownMean
from pyprob_amd.engine import ICEngine
from pyprob_amd.is_engine import ISRunner, gum_posterior
from pyprob_amd.packed import PackedBatch
from pyprob_amd.spec import NetSpec

warnings.simplefilter('ignore')
EMB = {'obs0': {'dim': 32}, 'obs1': {'dim': 32}}
spec = NetSpec(EMB, network='feedforward')
spec.add_address('mu', 'Normal')
eng = ICEngine(spec, seed=0)
a = synthetic_gum_arrays(1024, seed=1)
pb = PackedBatch.from_ragged(a['trace_len'], a['addr_idx'], a['values'], a['prior'], a['obs'], 1).to(eng.device)
for _ in range(20):
    eng.train_step(pb, 1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 500
for _ in range(K):
    eng.train_step(pb, 1e-3)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print('FF training step (batch 1024, %d parameters): %.1f us = %.2f M traces/s' % (spec.num_parameters(), dt * 1e6, 1024 / dt / 1e6))
run = ISRunner(eng)
for n in (1 << 20, 1 << 24):
    gum_posterior(eng, n, runner=run)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(5):
        st = gum_posterior(eng, n, seed=r, runner=run)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print('FF importance sampling: %d particles in %.2f ms = %.2f G particles/s' % (n, dt * 1e3, n / dt / 1e9))
model = GaussianWithUnknownMean()
torch.manual_seed(1)
t0 = time.perf_counter()
model.learn_inference_network(num_traces=2000000, observe_embeddings=EMB, batch_size=1024, seed=1, prior_chunk_traces=131072)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
net = model._inference_network
print('FF online training end to end: %d traces in %.2f s = %.2f M traces/s (loss %.3f -> %.3f)' % (
    net._total_train_traces, dt, net._total_train_traces / dt / 1e6, net._loss_init, net._loss_previous))
post = model.posterior_results(1000000, __import__('pyprob_amd').InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK,
                               observe={'obs0': 8, 'obs1': 9}, lock_step=True)
print('posterior mean %.3f std %.3f ESS %.0f (analytic 7.25 / 0.913)' % (post.mean, post.stddev, post.effective_sample_size))
