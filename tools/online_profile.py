"""cProfile of learn_inference_network (vectorised online training) - where the host time goes."""
import cProfile, pstats, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from models import GaussianWithUnknownMean
warnings.simplefilter('ignore')
model = GaussianWithUnknownMean()
from pyprob_amd.state import InferenceNetwork
LSTM = InferenceNetwork.LSTM
EMB = {'obs0': {'dim': 32}, 'obs1': {'dim': 32}}
model.learn_inference_network(inference_network=LSTM, num_traces=50000, observe_embeddings=EMB, batch_size=1024, lstm_dim=512, seed=1, prior_chunk_traces=131072)
pr = cProfile.Profile(); pr.enable()
model.learn_inference_network(inference_network=LSTM, num_traces=1000000, observe_embeddings=EMB, batch_size=1024, lstm_dim=512, seed=1, prior_chunk_traces=131072)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumtime').print_stats(28)
