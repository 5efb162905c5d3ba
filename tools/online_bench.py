"""End-to-end online training rate of learn_inference_network (prior generation + packing + upload + training step):
vectorised prior traces against one forward() per trace."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from models import GaussianWithUnknownMean, GaussianWithUnknownMeanMarsagliaLockStep
EMB = {'obs0': {'dim': 32}, 'obs1': {'dim': 32}}
from pyprob_amd.state import InferenceNetwork
LSTM = InferenceNetwork.LSTM
warnings.simplefilter('ignore')
for cls, n in ((GaussianWithUnknownMean, 2000000), (GaussianWithUnknownMeanMarsagliaLockStep, 600000)):
    for vec, m in ((True, n), (False, 20000)):
        torch.manual_seed(1)
        model = cls()
        t0 = time.perf_counter()
        model.learn_inference_network(inference_network=LSTM, num_traces=m, observe_embeddings=EMB, batch_size=1024, lstm_dim=512, seed=1,
                                      vectorised_prior=vec, prior_chunk_traces=131072)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        net = model._inference_network
        print('%s vectorised=%s: %d traces in %.2f s = %.0f traces/s end to end (loss %.3f -> %.3f)' % (
            cls.__name__, vec, net._total_train_traces, dt, net._total_train_traces / dt, net._loss_init, net._loss_previous))
