"""Prior trace generation on the device (state.PriorLockStep + pp_prior_draw) against the per-trace
generator of the reference's OnlineDataset loop (pyprob/nn/dataset.py:50-62): address tables, path probabilities and
moments; a device-resident chunk trains to the same loss level as the host route."""
import math

import numpy as np
import pytest

from models import GaussianWithUnknownMean, GaussianWithUnknownMeanMarsagliaLockStep
from pyprob_amd.state import InferenceNetwork, TraceMode

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
EMB = {'obs0': {'dim': 32}, 'obs1': {'dim': 32}}


def test_device_draws_have_the_moments_of_their_distributions():
    from pyprob_amd.ops import ops
    dev = torch.device('cuda:0')
    n = 1 << 20
    m, s = torch.tensor([1.5], device=dev), torch.tensor([0.7], device=dev)
    x = ops.prior_draw(0, m, s, n, 123, 0, 5)
    assert abs(float(x.mean()) - 1.5) < 4 * 0.7 / math.sqrt(n) and abs(float(x.std()) - 0.7) < 5e-3
    assert abs(float(((x - 1.5) / 0.7).pow(4).mean()) - 3.0) < 0.05                  # Gaussian kurtosis
    lo, hi = torch.full((n,), -1.0, device=dev), torch.linspace(1.0, 3.0, n, device=dev)
    u = ops.prior_draw(1, lo, hi, n, 123, 0, 6)
    assert bool((u >= lo).all()) and bool((u < hi).all())
    z = ((u - lo) / (hi - lo))
    assert abs(float(z.mean()) - 0.5) < 2e-3 and abs(float(z.var()) - 1 / 12) < 1e-3
    # same key, counter and statement -> same values; another statement -> independent values
    assert torch.equal(x, ops.prior_draw(0, m, s, n, 123, 0, 5))
    y = ops.prior_draw(0, m, s, n, 123, 0, 7)
    assert abs(float(((x - 1.5) * (y - 1.5)).mean())) < 5e-3


def test_device_chunk_equals_the_per_trace_generator_gum():
    torch.manual_seed(1)
    model = GaussianWithUnknownMean()
    n = 200000
    lens, table, ids, vals, prior, obs = model.prior_traces_packed(n, ['obs0', 'obs1'], device='cuda:0')
    res = model._last_prior_resident
    assert res is not None and res['values'].is_cuda and res['values'].numel() == n and res['obs'].shape == (n, 2)
    np.testing.assert_array_equal(res['values'].cpu().numpy(), vals)              # the resident columns ARE the chunk
    np.testing.assert_array_equal(res['obs'].cpu().numpy(), obs)
    tr = next(model._trace_generator(trace_mode=TraceMode.PRIOR_FOR_INFERENCE_NETWORK))
    assert [t[0] for t in table] == [v.address for v in tr.variables_controlled]   # the reference's address strings
    assert np.all(lens == 1) and np.all(prior == np.array([1.0, math.sqrt(5.0)], np.float32))
    assert abs(vals.mean() - 1.0) < 0.02 and abs(vals.std() - math.sqrt(5.0)) < 0.02
    r = obs - vals[:, None]                                                          # y_j - mu ~ N(0, sqrt 2), independent
    assert abs(r.mean()) < 0.01 and abs(r.std() - math.sqrt(2.0)) < 0.01
    assert abs(np.corrcoef(r[:, 0], r[:, 1])[0, 1]) < 0.01 and abs(np.corrcoef(r[:, 0], vals)[0, 1]) < 0.01


def test_device_chunk_equals_the_per_trace_generator_marsaglia():
    torch.manual_seed(2)
    model = GaussianWithUnknownMeanMarsagliaLockStep()
    n = 100000
    lens, table, ids, vals, prior, obs = model.prior_traces_packed(n, ['obs0', 'obs1'], device='cuda:0')
    assert model._last_prior_resident is None                                        # several paths: the host packer's route
    p = math.pi / 4
    for k in range(1, 5):                                                            # P(T = 2 k) = p (1 - p)^(k - 1)
        frac = float(np.mean(lens == 2 * k))
        assert abs(frac - p * (1 - p) ** (k - 1)) < 4 * math.sqrt(p * (1 - p) ** (k - 1) / n) + 1e-3, (k, frac)
    gen = model._trace_generator(trace_mode=TraceMode.PRIOR_FOR_INFERENCE_NETWORK)
    seen = set()
    for _ in range(60):
        seen.update(v.address for v in next(gen).variables_controlled)
    assert seen <= {t[0] for t in table}                                             # same address strings
    assert np.all(vals >= -1) and np.all(vals < 1) and abs(vals.mean()) < 0.01
    off = np.concatenate([[0], np.cumsum(lens)])
    last = np.stack([vals[off[1:] - 2], vals[off[1:] - 1]], 1)                        # the accepted pair of every trace
    assert np.all((last ** 2).sum(1) < 1)
    first_rejected = lens > 2
    rej = np.stack([vals[off[:-1]][first_rejected], vals[off[:-1] + 1][first_rejected]], 1)
    assert np.all((rej ** 2).sum(1) >= 1)


def test_online_training_from_device_resident_chunks(monkeypatch):
    """learn_inference_network online: chunks drawn on the device and trained from HBM (pp_train_resident) reach the
    loss level of the host-generated route; the counters follow the reference's bookkeeping."""
    losses = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('PP_PRIOR_DEVICE', mode)
        torch.manual_seed(3)
        model = GaussianWithUnknownMean()
        model.learn_inference_network(inference_network=InferenceNetwork.LSTM, num_traces=60 * 256, observe_embeddings=EMB,
                                      batch_size=256, lstm_dim=64, seed=1)
        net = model._inference_network
        assert net._total_train_traces == 60 * 256 and net._total_train_iterations == 60
        assert net._engine.spec.addresses[0].total_train_iterations == 60
        losses[mode] = (net._loss_init, float(np.mean(net._history_train_loss[-10:])))
    assert losses['1'][1] < losses['1'][0] - 0.2 and abs(losses['1'][1] - losses['0'][1]) < 0.25
