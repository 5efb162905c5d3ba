"""End-to-end online training rate through Model.learn_inference_network (prior generation + training), H = 512, B = 1024,
device-generated chunks (PP_PRIOR_DEVICE=1, the default) against host-generated ones, alternating in ONE process:
    python tools/online_rate.py [traces]"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyprob_amd.state import InferenceNetwork
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 1024 * 1024
GUM, GUMM = bench.api_models()
EMB = {'obs0': {'dim': 32}, 'obs1': {'dim': 32}}
for prog, cls, total in (('gum', GUM, n), ('gumm', GUMM, n // 4)):
    for mode in ('1', '0', '1', '0'):
        os.environ['PP_PRIOR_DEVICE'] = mode
        model = cls()
        torch.manual_seed(1)
        with contextlib.redirect_stdout(io.StringIO()):
            model.learn_inference_network(num_traces=128 * 1024, inference_network=InferenceNetwork.LSTM, observe_embeddings=EMB,
                                          batch_size=1024, lstm_dim=512, seed=1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.learn_inference_network(num_traces=total, inference_network=InferenceNetwork.LSTM, observe_embeddings=EMB,
                                          batch_size=1024, lstm_dim=512, seed=1)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print('%s online, chunks drawn on the %s: %d traces in %.3f s = %.2f M traces/s' % (
            prog, 'device' if mode == '1' else 'host', total, dt, total / dt / 1e6), flush=True)
