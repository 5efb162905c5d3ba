"""Lock-step importance sampling of a program with stochastic control flow (GaussianUnknownMeanMarsaglia, SURVEY.md
8f.2): particles/s of posterior_results(N, lock_step=True) against the one-particle-per-forward() engine."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from models import GaussianWithUnknownMeanMarsagliaLockStep
from pyprob_amd.state import InferenceEngine
from pyprob_amd.state import InferenceNetwork
LSTM = InferenceNetwork.LSTM
IC = InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK
EMB = {'obs0': {'dim': 32}, 'obs1': {'dim': 32}}
warnings.simplefilter('ignore')
torch.manual_seed(1)
model = GaussianWithUnknownMeanMarsagliaLockStep()
model.learn_inference_network(inference_network=LSTM, num_traces=20000, observe_embeddings=EMB, batch_size=256, lstm_dim=512, seed=1)
obs = {'obs0': 4, 'obs1': 5}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
model.posterior_results(10000, IC, observe=obs, lock_step=True, seed=1)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    post = model.posterior_results(n, IC, observe=obs, lock_step=True, seed=2 + rep)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print('lock-step: %d particles, %d control-flow paths, %.1f ms -> %.2f M particles/s; mean %.3f std %.3f ESS %.0f' % (
        n, post.num_paths, (t1 - t0) * 1e3, n / (t1 - t0) / 1e6, post.mean, post.stddev, post.effective_sample_size))
t0 = time.perf_counter()
ref = model.posterior_results(300, IC, lock_step=False, observe=obs)
t1 = time.perf_counter()
print('one particle per forward(): %.0f particles/s (mean %.3f)' % (300 / (t1 - t0), ref.mean))
