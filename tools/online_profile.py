"""Where the host time of vectorised online training goes (cProfile of learn_inference_network, warm network)."""
import cProfile, os, pstats, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from models import GaussianWithUnknownMean
from pyprob_amd.state import InferenceNetwork
warnings.simplefilter('ignore')
EMB = {'obs0': {'dim': 32}, 'obs1': {'dim': 32}}
net = InferenceNetwork.FEEDFORWARD if (len(sys.argv) > 1 and sys.argv[1] == 'feedforward') else InferenceNetwork.LSTM
model = GaussianWithUnknownMean()
kw = dict(inference_network=net, observe_embeddings=EMB, batch_size=1024, lstm_dim=512, seed=1, prior_chunk_traces=131072)
model.learn_inference_network(num_traces=200000, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
model.learn_inference_network(num_traces=4000000, **kw)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('steady state: 4 M traces in %.2f s = %.2f M traces/s end to end' % (dt, 4.0 / dt))
pr = cProfile.Profile()
pr.enable()
model.learn_inference_network(num_traces=1000000, **kw)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
