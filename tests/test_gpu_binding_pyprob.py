"""Real pyprob + the HIP kernels in ONE process: runs only on a machine that has both a ROCm device and
an importable `pyprob` (PYTHONPATH or site-packages) - pyprob's own `Model.learn_inference_network`, `optimize()` loop and
`posterior_results` with `pyprob_amd.binding.install()`, the network's parameters living in HBM. Everywhere else it is
skipped: tests/test_binding_reference.py runs the same host code over the oracle-backed CPU kernels, tests/test_gpu_binding.py
the operators on the device."""
import math
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
pyprob = pytest.importorskip('pyprob')


def test_pyprob_trains_and_infers_through_the_hip_binding():
    from pyprob import InferenceEngine, InferenceNetwork, Model
    from pyprob.distributions import Normal
    import pyprob_amd.binding as hip

    class GaussianWithUnknownMean(Model):                      # pyprob/tests: the benchmark program, unmodified
        def __init__(self):
            super().__init__('Gaussian with unknown mean')

        def forward(self):
            mu = pyprob.sample(Normal(1, math.sqrt(5)))
            likelihood = Normal(mu, math.sqrt(2))
            pyprob.observe(likelihood, name='obs0')
            pyprob.observe(likelihood, name='obs1')
            return mu

    hip.install()
    try:
        pyprob.seed(1)
        model = GaussianWithUnknownMean()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            model.learn_inference_network(num_traces=20000, batch_size=128, observe_embeddings={'obs0': {'dim': 32}, 'obs1': {'dim': 32}},
                                          inference_network=InferenceNetwork.LSTM, lstm_dim=64, learning_rate_init=1e-3)
        net = model._inference_network
        assert type(net).__name__ == 'InferenceNetworkLSTMHip' and net._hip_engine.params.is_cuda
        hist = np.asarray(net._history_train_loss)
        assert np.isfinite(hist).all() and hist[-20:].mean() < hist[:5].mean() + 0.05      # (156 iterations: it has started to learn)
        post = model.posterior_results(2000, InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK,
                                       observe={'obs0': 8, 'obs1': 9})
        assert abs(float(post.mean) - 7.25) < 1.0 and float(post.effective_sample_size) > 20      # analytic posterior N(7.25, 0.913^2)
    finally:
        hip.uninstall()
