"""Priors / likelihoods of the host-side trace runtime and the weighted-sample container.

Mirrors the slice of pyprob.distributions that the hot path touches (pyprob/distributions/{distribution,normal,
uniform,categorical,empirical}.py): thin objects carrying `name`, parameters, `sample()` and
`log_prob(value, sum=False)`. They are evaluated on the host like in the reference; the
proposal distributions themselves (Mixture / TruncatedNormal) live in the HIP kernels.
"""
import math
import warnings

import numpy as np
import torch


def _t(x):
    return x if torch.is_tensor(x) else torch.as_tensor(x, dtype=torch.float32)


class Distribution:
    def __init__(self, name, address_suffix, torch_dist=None):
        self.name = name
        self._address_suffix = address_suffix
        self._td = torch_dist

    # The torch.distributions object behind a prior is built on first use: constructing one costs ~45 us of host time
    # (parameter broadcasting), and a particle of an importance-sampling run with the inference network never asks its
    # priors for samples or log-probs - the device does (measured: a fifth of a coroutine worker's time).
    @property
    def _torch_dist(self):
        if self._td is None:
            self._td = self._make_torch_dist()
        return self._td

    def _make_torch_dist(self):
        raise NotImplementedError

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
        # A state.ParticleTensor (per-particle value of a lock-step run) pays a Python __torch_function__ round trip for EVERY
        # torch call and attribute read (~4 us each; a posterior call builds two of these objects per statement): the
        # constructor only looks at metadata, so it runs with the subclass dispatch off, and a float32 tensor is taken as it is.
        with torch._C.DisableTorchFunctionSubclass():
            if not (torch.is_tensor(loc) and loc.dtype == torch.float32):
                loc = _t(loc).float()
            if not (torch.is_tensor(scale) and scale.dtype == torch.float32):
                scale = _t(scale).float()
            if scale.device != loc.device and not (scale.numel() == 1 and scale.device.type == 'cpu'):
                scale = scale.to(loc.device)      # (a host scalar next to device locations stays where it is: the device kernels
                                                  #  take it as a cached constant, and a posterior call pays no copy for it)
            if scale.numel() == 1 and scale.device.type == 'cpu' and not float(scale) > 0.0:   # (what torch's validation rejects)
                raise ValueError('Normal: the scale must be positive, got {}'.format(float(scale)))
            # per-particle parameters are handed out as they are (see _raw_params)
            self._raw = loc.shape == scale.shape or type(loc).__name__ == 'ParticleTensor' or type(scale).__name__ == 'ParticleTensor'
        self._loc, self._scale = loc, scale
        super().__init__('Normal', 'Normal')

    def _make_torch_dist(self):
        # per-particle parameters of a lock-step run may hold stale (even NaN) entries for particles that are not on the
        # current control-flow path: no argument validation for vectors
        scale = self._scale if self._scale.device == self._loc.device else self._scale.to(self._loc.device)
        return torch.distributions.Normal(self._loc, scale, validate_args=None if self._loc.numel() == 1 else False)

    def _device(self):
        return self._loc.device

    def _raw_params(self):
        # per-particle parameters of a lock-step run (state.ParticleTensor) are handed out as they are: building the
        # torch.distributions object would broadcast - i.e. READ - a value whose draw may still be deferred
        return self._td is None and getattr(self, '_raw', False)

    @property
    def mean(self):
        return self._loc if self._raw_params() else self._torch_dist.mean

    @property
    def stddev(self):
        return self._scale if self._raw_params() else self._torch_dist.stddev

    @property
    def variance(self):
        return self.stddev ** 2

    def __repr__(self):
        return 'Normal({}, {})'.format(self.mean.tolist(), self.stddev.tolist())


class Uniform(Distribution):
    """pyprob/distributions/uniform.py:7-25"""

    def __init__(self, low, high):
        self._low, self._high = _t(low).float(), _t(high).float()
        super().__init__('Uniform', 'Uniform')

    def _make_torch_dist(self):
        return torch.distributions.Uniform(self._low, self._high, validate_args=False)

    def _device(self):
        return self._low.device

    @property
    def low(self):
        return self._low if self._td is None and self._low.shape == self._high.shape else self._torch_dist.low

    @property
    def high(self):
        return self._high if self._td is None and self._low.shape == self._high.shape else self._torch_dist.high


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


class Bernoulli(Distribution):
    """pyprob/distributions/bernoulli.py"""

    def __init__(self, probs):
        probs = _t(probs).float()
        super().__init__('Bernoulli', 'Bernoulli', torch.distributions.Bernoulli(probs=probs, validate_args=False))

    @property
    def probs(self):
        return self._torch_dist.probs


class Empirical:
    """Weighted samples: the result type of Model.prior / posterior (pyprob/distributions/empirical.py). Memory-backed;
    values are floats / tensors / arbitrary objects (or ONE device tensor for lock-step runs), weights are log-weights.
    The reductions (finalize :298-309, expectation :451-466, moments :668-690, effective_sample_size :758-766) are
    vectorised over the weight array instead of looping over the values; the transformation methods (map, condition,
    resample, thin, unweighted, slicing, concatenation :568-664, :768-773) return new Empiricals like the reference."""

    def __init__(self, values=None, log_weights=None, weights=None, concat_empiricals=None, name='Empirical'):
        self.name = name
        self._metadata = {}
        self._finalized = False
        self.length = 0
        if concat_empiricals is not None:        # ParallelModel's merge (model.py:395-404)
            parts = list(concat_empiricals)
            self._values = [v for e in parts for v in e.get_values()]
            self._log_weights = np.concatenate([e.log_weights_numpy() for e in parts]) if parts else []
            self.finalize()
            return
        if weights is not None:
            if log_weights is not None:
                raise ValueError('Expecting log_weights or weights, not both.')
            with np.errstate(divide='ignore'):
                log_weights = np.log(np.asarray(weights, np.float64))
        self._values = [] if values is None else values
        if log_weights is None:
            log_weights = [] if values is None else np.zeros(len(values))
        self._log_weights = log_weights
        if values is not None and len(values) > 0:
            self.finalize()

    @classmethod
    def from_device(cls, values, log_weights, device_stats, name='Empirical'):
        """The result of a lock-step run: ONE device tensor of values, one of log-weights and the importance statistics
        the device reduced in float64 (pp_is_stats / pp_is_fused). mean / variance / effective_sample_size answer from
        those; the host copies behind everything else (weights, sampling, slicing ...) are made on first use - a
        posterior call of 10^6 particles does not pay a 4 MB read-back and a host-side exp per call."""
        out = cls(name=name)
        out._values, out._log_weights = values, log_weights
        out.length = int(log_weights.numel())
        out.device_stats = device_stats
        out._finalized = True
        for k in ('_lw', '_w', '_uniform', '_cum'):      # (set by finalize(); absent = not materialised yet)
            out.__dict__.pop(k, None)
        return out

    def __getattr__(self, name):
        if name in ('_lw', '_w', '_uniform', '_cum') and '_log_weights' in self.__dict__ and self.__dict__.get('_finalized'):
            self.finalize()          # host-side weights of a device-backed Empirical, on first use
            if name in self.__dict__:
                return self.__dict__[name]
        raise AttributeError(name)

    def _host_weights_ready(self):
        return '_w' in self.__dict__

    # ---- construction ------------------------------------------------------------------------------------
    def add(self, value, log_weight=None, weight=None):
        if weight is not None:
            log_weight = math.log(weight) if weight > 0 else float('-inf')
        if not isinstance(self._values, list) or not isinstance(self._log_weights, list):
            lw = self._lw if self._finalized else self._log_weights
            lw = lw.cpu().numpy() if torch.is_tensor(lw) else lw
            self._values = list(self.get_values()) if self._finalized else list(self._values)
            self._log_weights = [float(x) for x in np.asarray(lw, np.float64)]
        self._values.append(value)
        self._log_weights.append(0.0 if log_weight is None else float(log_weight))
        self._finalized = False

    def add_sequence(self, values, log_weights=None, weights=None):
        for i, v in enumerate(values):
            self.add(v, None if log_weights is None else log_weights[i], None if weights is None else weights[i])

    def finalize(self):
        lw = np.asarray(self._log_weights.cpu() if torch.is_tensor(self._log_weights) else self._log_weights, np.float64)
        self._lw = lw
        self.length = len(lw)
        m = np.max(lw) if self.length else 0.0
        w = np.exp(lw - m)
        self._w = w / w.sum() if self.length else w
        self._uniform = bool(self.length) and bool(np.all(lw == lw[0]))
        self._cum = None
        self._finalized = True
        return self

    def _check_finalized(self):
        if not self._finalized:
            raise RuntimeError('Empirical not finalized. Call finalize first.')

    def __len__(self):
        return self.length

    def rename(self, name):
        self.name = name
        return self

    def add_metadata(self, **kwargs):
        self._metadata.update(kwargs)

    @property
    def metadata(self):
        return self._metadata

    def _like(self, values, log_weights, **meta):
        out = Empirical(name=self.name)
        out._values, out._log_weights = values, log_weights
        out.finalize()
        out._metadata = dict(self._metadata)
        out.add_metadata(**meta)
        return out

    # ---- access ---------------------------------------------------------------------------------------------
    @property
    def log_weights(self):
        return self._lw

    @property
    def weights(self):
        return torch.from_numpy(self._w)

    @property
    def weighted(self):
        return not self._uniform

    def weights_numpy(self):
        return self._w

    def log_weights_numpy(self):
        return self._lw

    def get_values(self):
        self._check_finalized()
        v = self._values
        return list(v.detach().cpu().unbind(0)) if torch.is_tensor(v) else v

    def values_numpy(self):
        v = self._values
        if torch.is_tensor(v):
            return v.detach().cpu().double().numpy()
        return np.asarray([float(x) for x in v], np.float64)

    def _value(self, i):
        v = self._values
        return v[i].detach().cpu() if torch.is_tensor(v) else v[i]

    def __iter__(self):
        self._check_finalized()
        for i in range(self.length):
            yield self._value(i)

    def __getitem__(self, index):
        self._check_finalized()
        if isinstance(index, slice):
            return self._like(self.get_values()[index], self._lw[index], op='slice', index=str(index))
        if isinstance(index, (int, np.integer)):
            return self._value(int(index))
        raise RuntimeError('Cannot use the given value ({}) as index'.format(index))

    def _draw_indices(self, n, lo=None, hi=None):
        """n indices distributed like the (renormalised) weights of [lo, hi); randomness from torch's global generator
        (so pyprob.seed / torch.manual_seed make it reproducible)."""
        lo = 0 if lo is None else lo
        hi = self.length if hi is None else hi
        if hi <= lo:
            raise ValueError('empty index range')
        u = torch.rand(n, dtype=torch.float64).numpy()
        if self._uniform:
            return lo + np.minimum((u * (hi - lo)).astype(np.int64), hi - lo - 1)
        if self._cum is None:
            self._cum = np.cumsum(self._w)
        base = self._cum[lo - 1] if lo > 0 else 0.0
        total = self._cum[hi - 1] - base
        return np.minimum(np.searchsorted(self._cum, base + u * total, side='right'), hi - 1)

    def sample(self, min_index=None, max_index=None):
        """One value drawn according to the weights (empirical.py:392-408; min_index inclusive, max_index exclusive)."""
        self._check_finalized()
        return self._value(int(self._draw_indices(1, min_index, max_index)[0]))

    # ---- transformations (each returns a new Empirical) ------------------------------------------------------
    def map(self, func, min_index=None, max_index=None):
        self._check_finalized()
        if self.length == 0:
            return self
        sl = slice(min_index or 0, self.length if max_index is None else max_index)
        return self._like([func(v) for v in self.get_values()[sl]], self._lw[sl], op='map', length=self.length)

    def condition(self, criterion, min_index=None, max_index=None):
        self._check_finalized()
        if self.length == 0:
            return self
        lo, hi = min_index or 0, self.length if max_index is None else max_index
        vals = self.get_values()
        keep = [i for i in range(lo, hi) if criterion(vals[i])]
        return self._like([vals[i] for i in keep], self._lw[keep], op='condition', length=self.length, length_after=len(keep))

    def filter(self, *args, **kwargs):
        warnings.warn('Empirical.filter will be deprecated in future releases. Use Empirical.condition instead.')
        return self.condition(*args, **kwargs)

    def resample(self, num_samples, map_func=None, min_index=None, max_index=None):
        """num_samples unweighted draws (empirical.py:617-637)."""
        self._check_finalized()
        ess = self.effective_sample_size
        idx = self._draw_indices(int(num_samples), min_index, max_index)
        vals = self.get_values()
        out = [vals[i] if map_func is None else map_func(vals[i]) for i in idx]
        return self._like(out, np.zeros(len(out)), op='resample', length=self.length, num_samples=int(num_samples),
                          ess_before=ess)

    def thin(self, num_samples, map_func=None, min_index=None, max_index=None):
        self._check_finalized()
        lo, hi = min_index or 0, self.length if max_index is None else max_index
        step = max(1, math.floor((hi - lo) / num_samples))
        idx = list(range(lo, hi, step))
        vals = self.get_values()
        out = [vals[i] if map_func is None else map_func(vals[i]) for i in idx]
        return self._like(out, self._lw[idx], op='thin', length=self.length, num_samples=int(num_samples), step=int(step))

    def unweighted(self):
        return self._like(list(self.get_values()), np.zeros(self.length), op='discard_weights')

    # ---- reductions ------------------------------------------------------------------------------------------
    def expectation(self, func):
        """sum_i w_i func(x_i) (empirical.py:451-466). func is tried on the whole value array (numpy, then a torch tensor
        for functions like torch.sin) before falling back to one call per value."""
        self._check_finalized()
        v = self.values_numpy()
        for conv in ((lambda a: a), torch.from_numpy):
            try:
                fv = func(conv(v))
                fv = np.asarray(fv.detach().cpu() if torch.is_tensor(fv) else fv, np.float64)
                if fv.shape == v.shape:
                    return float(np.sum(self._w * fv))
            except Exception:   # noqa: BLE001 - func works on one value at a time, or on the other array type
                pass
        return float(np.sum(self._w * np.asarray([float(func(x)) for x in self.get_values()])))

    def _device_moments(self):
        st = self.__dict__.get('device_stats')
        # ONE source of truth per object: while the values live on the device the float64 device statistics answer mean /
        # variance / ESS - also after something (weights_numpy, sampling, slicing) materialised the host-side weights, so the
        # same property returns the same number whatever was called before
        return st if (st is not None and torch.is_tensor(self._values)) else None

    @property
    def mean(self):
        st = self._device_moments()
        if st is not None:
            return float(st['mean'])
        return float(np.sum(self._w * self.values_numpy()))

    @property
    def variance(self):
        st = self._device_moments()
        if st is not None:
            return float(max(st['var'], 0.0))
        v = self.values_numpy()
        return float(np.sum(self._w * (v - self.mean) ** 2))

    @property
    def stddev(self):
        return math.sqrt(self.variance)

    @property
    def skewness(self):
        v = self.values_numpy()
        return float(np.sum(self._w * ((v - self.mean) / self.stddev) ** 3))

    @property
    def kurtosis(self):
        v = self.values_numpy()
        return float(np.sum(self._w * ((v - self.mean) / self.stddev) ** 4))

    @property
    def effective_sample_size(self):
        st = self._device_moments()
        if st is not None:
            return float(st['ess'])
        return float(1.0 / np.sum(self._w ** 2))

    @property
    def mode(self):
        """Highest-weight value; with uniform weights the most frequent (hashable) value (empirical.py:693-712)."""
        self._check_finalized()
        if self._uniform:
            counts = {}
            for v in self.get_values():
                k = float(v) if torch.is_tensor(v) and v.numel() == 1 else v
                counts[k] = counts.get(k, 0) + 1
            return max(counts.items(), key=lambda kv: kv[1])[0]
        return self._value(int(np.argmax(self._lw)))

    @property
    def median(self):
        self._check_finalized()
        if self._uniform:
            v = self._values
            if torch.is_tensor(v) or (len(v) and torch.is_tensor(v[0])):
                # tensor values: torch.median, the LOWER of the two middle values of an even count (empirical.py:719-721)
                return float(np.sort(self.values_numpy())[(self.length - 1) // 2])
            return float(np.median(self.values_numpy()))       # :723-724
        return self.resample(1000).median           # empirical.py:727

    @property
    def min(self):
        return float(np.min(self.values_numpy()))

    @property
    def max(self):
        return float(np.max(self.values_numpy()))

    def arg_max(self, map_func):
        vals = self.get_values()
        return vals[int(np.argmax([float(map_func(v)) for v in vals]))]

    def arg_min(self, map_func):
        vals = self.get_values()
        return vals[int(np.argmin([float(map_func(v)) for v in vals]))]

    def __repr__(self):
        try:
            return 'Empirical(items:{}, weighted:{}, mean:{}, stddev:{})'.format(self.length, self.weighted, self.mean, self.stddev)
        except Exception:   # noqa: BLE001 - values that are not scalars
            return 'Empirical(items:{})'.format(self.length)
