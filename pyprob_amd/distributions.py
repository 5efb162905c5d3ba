"""Priors / likelihoods of the host-side trace runtime and the weighted-sample container.

Mirrors the slice of pyprob.distributions that the hot path touches (pyprob/distributions/{distribution,normal,
uniform,categorical,empirical}.py): thin objects carrying `name`, parameters, `sample()` and
`log_prob(value, sum=False)`. They are evaluated on the host like in the reference; the
proposal distributions themselves (Mixture / TruncatedNormal) live in the HIP kernels.
"""
import math

import numpy as np
import torch


def _t(x):
    return x if torch.is_tensor(x) else torch.as_tensor(x, dtype=torch.float32)


class Distribution:
    def __init__(self, name, address_suffix, torch_dist=None):
        self.name = name
        self._address_suffix = address_suffix
        self._torch_dist = torch_dist

    def sample(self):
        return self._torch_dist.sample()

    def log_prob(self, value, sum=False):
        lp = self._torch_dist.log_prob(_t(value).to(self._device()))
        return torch.sum(lp) if sum else lp

    def _device(self):
        return torch.device('cpu')

    @property
    def mean(self):
        return self._torch_dist.mean

    @property
    def variance(self):
        return self._torch_dist.variance

    @property
    def stddev(self):
        return self.variance.sqrt()


class Normal(Distribution):
    """pyprob/distributions/normal.py:7-31"""

    def __init__(self, loc, scale):
        loc, scale = _t(loc).float(), _t(scale).float()
        if scale.device != loc.device:
            scale = scale.to(loc.device)
        # per-particle parameters of a lock-step run may hold stale (even NaN) entries for particles that are not on the
        # current control-flow path: no argument validation for vectors
        super().__init__('Normal', 'Normal', torch.distributions.Normal(loc, scale, validate_args=None if loc.numel() == 1 else False))

    def _device(self):
        return self._torch_dist.loc.device

    def __repr__(self):
        return 'Normal({}, {})'.format(self.mean.tolist(), self.stddev.tolist())


class Uniform(Distribution):
    """pyprob/distributions/uniform.py:7-25"""

    def __init__(self, low, high):
        low, high = _t(low).float(), _t(high).float()
        super().__init__('Uniform', 'Uniform', torch.distributions.Uniform(low, high, validate_args=False))

    def _device(self):
        return self._torch_dist.low.device

    @property
    def low(self):
        return self._torch_dist.low

    @property
    def high(self):
        return self._torch_dist.high


class Poisson(Distribution):
    """pyprob/distributions/poisson.py:7-21. The inference network proposes a continuous TruncatedNormal mixture on
    [0, 40] for a Poisson variable (proposal_poisson_truncated_normal_mixture.py), so log_prob must accept non-integer
    values like the reference's torch did: no argument validation."""

    def __init__(self, rate):
        rate = _t(rate).float()
        super().__init__('Poisson', 'Poisson', torch.distributions.Poisson(rate, validate_args=False))

    @property
    def rate(self):
        return self._torch_dist.mean


class Categorical(Distribution):
    """pyprob/distributions/categorical.py:7-39"""

    def __init__(self, probs):
        probs = _t(probs).float()
        if probs.dim() == 0:
            raise ValueError('probs cannot be a scalar.')
        td = torch.distributions.Categorical(probs=probs)
        self._probs = td.probs
        self._num_categories = self._probs.size(-1)
        super().__init__('Categorical', 'Categorical(len_probs:{})'.format(self._num_categories), td)

    @property
    def num_categories(self):
        return self._num_categories

    @property
    def probs(self):
        return self._probs


class Empirical:
    """Weighted samples (pyprob/distributions/empirical.py: add :315-340, finalize :298-309, expectation :451-466,
    effective_sample_size :758-766). Memory-backed only; values are floats/tensors, weights are log-weights."""

    def __init__(self, values=None, log_weights=None, name='Empirical'):
        self.name = name
        self._values = [] if values is None else values
        self._log_weights = [] if log_weights is None else log_weights
        self._finalized = False
        self._metadata = {}

    def add(self, value, log_weight=None):
        self._values.append(value)
        self._log_weights.append(0.0 if log_weight is None else float(log_weight))

    def finalize(self):
        lw = np.asarray(self._log_weights.cpu() if torch.is_tensor(self._log_weights) else self._log_weights, np.float64)
        self._lw = lw
        self.length = len(lw)
        m = np.max(lw) if self.length else 0.0
        w = np.exp(lw - m)
        self._w = w / w.sum() if self.length else w
        self._finalized = True
        return self

    def __len__(self):
        return self.length

    def rename(self, name):
        self.name = name
        return self

    def add_metadata(self, **kwargs):
        self._metadata.update(kwargs)

    @property
    def log_weights(self):
        return self._lw

    def values_numpy(self):
        v = self._values
        if torch.is_tensor(v):
            return v.detach().cpu().double().numpy()
        return np.asarray([float(x) for x in v], np.float64)

    def expectation(self, func):
        v = self.values_numpy()
        return float(np.sum(self._w * np.asarray([func(x) for x in v]) if not isinstance(func(v[0]), np.ndarray)
                            else self._w * func(v)))

    @property
    def mean(self):
        return float(np.sum(self._w * self.values_numpy()))

    @property
    def variance(self):
        v = self.values_numpy()
        return float(np.sum(self._w * (v - self.mean) ** 2))

    @property
    def stddev(self):
        return math.sqrt(self.variance)

    @property
    def effective_sample_size(self):
        return float(1.0 / np.sum(self._w ** 2))
