"""Probe behind tests/test_gpu_model.py::test_feedforward_network_with_control_flow_and_categorical: the FeedForward network on the
rejection-loop program, trained T times with the test's seeds (float atomics make the gradient sums run-dependent), each network
asked for the same posterior by the row-list executor (nested paths on / off) and the boolean-mask executor: are the executors
identical on one network, and how does the effective sample size vary BETWEEN trainings?   python tools/ff_executor_probe.py [T]"""
import contextlib
import io
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np                                             # noqa: E402
import torch                                                   # noqa: E402
from models import GaussianWithUnknownMeanMarsagliaLockStep    # noqa: E402
from pyprob_amd.state import InferenceEngine                   # noqa: E402

IC = InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK
EMB = {'obs0': {'dim': 32}, 'obs1': {'dim': 32}}
OBS = {'obs0': 8, 'obs1': 9}
warnings.simplefilter('ignore')
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for trial in range(T):
    torch.manual_seed(31)
    model = GaussianWithUnknownMeanMarsagliaLockStep()
    with contextlib.redirect_stdout(io.StringIO()):
        model.learn_inference_network(num_traces=120000, observe_embeddings=EMB, batch_size=256, seed=10)
    res = []
    for rows, nest in (('0', '0'), ('1', '0'), ('1', '1')):
        os.environ['PP_IS_ROWS'] = rows
        os.environ['PP_IS_NEST'] = nest
        post = model.posterior_results(40000, IC, observe=OBS, lock_step=True, seed=5)
        res.append((float(post.mean), float(post.effective_sample_size), post.num_paths, post._all_log_weights.cpu().numpy(),
                    post._all_values.cpu().numpy()))
    base = res[0]
    same = all(np.array_equal(r[3], base[3], equal_nan=True) and np.array_equal(r[4], base[4]) for r in res[1:])
    lw = base[3][np.isfinite(base[3])]
    eng = model._inference_network._engine
    print('trial %2d  mean %.4f  ess %9.2f  paths %d  addresses %d  loss %.4f  executors identical %s  top log-weights %s'
          % (trial, base[0], base[1], base[2], len(eng.spec.addresses), float(model._inference_network._loss_previous), same,
             np.round(np.sort(lw)[-3:], 3)), flush=True)
