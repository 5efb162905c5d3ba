"""Model._lockstep_plan_key (the launch-plan cache of a static lock-step program): a replay does not run forward(), so the
key must hold the VALUE of everything forward() can read - private instance attributes, class attributes, module globals,
closure cells, the code of the methods it calls - or refuse to plan (ADVICE r04 model.py:158).
Host logic only (the engine is a stand-in): runs without a GPU."""
import types

import numpy as np
import torch

import pyprob_amd as pyprob
from pyprob_amd.distributions import Normal
from pyprob_amd.model import Model

GLOBAL_SCALE = 2.0
GLOBAL_OBJECT = object()


class _Spec:
    addresses = ['a']


class _Eng:
    device = torch.device('cuda:0')
    spec = _Spec()
    token = 7


class _Net:
    _engine = _Eng()


def _key(model, observe={'obs0': 1.0}):
    model._inference_network = _Net()
    return model._lockstep_plan_key(100, observe, 1.0, (), {})


class Private(Model):
    def __init__(self):
        super().__init__()
        self.prior_mean = 1.0
        self._sigma = 2.0
        self._unrelated = object()       # never named by forward(): does not rule the plan out

    def forward(self):
        mu = pyprob.sample(Normal(self.prior_mean, 1.0))
        pyprob.observe(Normal(mu, self._sigma), name='obs0')
        return mu


def test_private_attribute_read_by_forward_is_in_the_key():
    m = Private()
    k0 = _key(m)
    assert k0 is not None
    m._sigma = 3.0
    k1 = _key(m)
    assert k1 is not None and k1 != k0
    m._sigma = 2.0
    assert _key(m) == k0
    m._sigma = object()                  # cannot be fingerprinted by value: no plan
    assert _key(m) is None


class ViaMethod(Model):
    SCALE = 1.5                          # a class constant

    def __init__(self):
        super().__init__()
        self._tau = 0.5

    def helper(self, mu):
        return Normal(mu, self._tau * self.SCALE)

    def forward(self):
        mu = pyprob.sample(Normal(0.0, 1.0))
        pyprob.observe(self.helper(mu), name='obs0')
        return mu


def test_callee_methods_and_class_constants_are_in_the_key():
    m = ViaMethod()
    k0 = _key(m)
    assert k0 is not None
    m._tau = 0.75                        # read by helper(), not by forward() itself
    k1 = _key(m)
    assert k1 is not None and k1 != k0
    m._tau = 0.5
    old = ViaMethod.SCALE
    try:
        ViaMethod.SCALE = 9.0
        assert _key(m) != k0
    finally:
        ViaMethod.SCALE = old
    assert _key(m) == k0
    # a new body for the callee is a new program
    orig = ViaMethod.helper
    try:
        ViaMethod.helper = lambda self, mu: Normal(mu, 4.0)
        assert _key(m) != k0
    finally:
        ViaMethod.helper = orig


class ReadsGlobal(Model):
    def forward(self):
        mu = pyprob.sample(Normal(0.0, 1.0))
        pyprob.observe(Normal(mu, GLOBAL_SCALE), name='obs0')
        return mu


class ReadsGlobalObject(Model):
    def forward(self):
        mu = pyprob.sample(Normal(0.0, 1.0))
        pyprob.observe(Normal(mu, 1.0 if GLOBAL_OBJECT else 2.0), name='obs0')
        return mu


def test_module_globals_are_in_the_key_or_rule_the_plan_out():
    global GLOBAL_SCALE
    m = ReadsGlobal()
    k0 = _key(m)
    assert k0 is not None
    GLOBAL_SCALE = 5.0
    try:
        assert _key(m) not in (None, k0)
    finally:
        GLOBAL_SCALE = 2.0
    assert _key(m) == k0
    assert _key(ReadsGlobalObject()) is None


def test_closure_cells_are_in_the_key():
    def make(scale):
        class Closed(Model):
            def forward(self):
                mu = pyprob.sample(Normal(0.0, 1.0))
                pyprob.observe(Normal(mu, scale), name='obs0')
                return mu
        return Closed()
    a, b = make(1.0), make(2.0)
    ka, kb = _key(a), _key(b)
    assert ka is not None and kb is not None and ka != kb
    assert _key(make({'x': 1})) is None                 # a cell that holds an object


def test_small_tensors_and_containers_fingerprint_by_value():
    class T(Model):
        def __init__(self):
            super().__init__()
            self._loc = torch.tensor([1.0, 2.0])
            self._pair = (0.5, np.float64(2.0).item())

        def forward(self):
            mu = pyprob.sample(Normal(self._loc[0], self._pair[0]))
            pyprob.observe(Normal(mu, self._pair[1]), name='obs0')
            return mu
    m = T()
    k0 = _key(m)
    assert k0 is not None
    m._loc[1] = 7.0                                     # in-place change of a tensor constant
    assert _key(m) not in (None, k0)
    m._loc = torch.zeros(1000)                          # too large to fingerprint per call
    assert _key(m) is None


def test_engine_identity_is_a_token_not_an_address():
    m = Private()
    k0 = _key(m)
    net = _Net()
    net._engine = types.SimpleNamespace(device=torch.device('cuda:0'), spec=_Spec(), token=8)
    m._inference_network = net
    assert m._lockstep_plan_key(100, {'obs0': 1.0}, 1.0, (), {}) != k0
    assert m._lockstep_plan_key(100, {'obs0': 1.0}, 1.0, (1,), {}) is None        # call arguments: no plan


# ---- values read THROUGH a module, a class or an instance-attribute function (model.py:381) -------------------
settings = types.ModuleType('user_settings')
settings.PRIOR_MEAN = 1.0
settings.helper = lambda x: x


class Config:
    MU = 1.5
    SIGMA = 2.0


def _scaled(x, k=2.0):
    return k * x


class ThroughModule(Model):
    def forward(self):
        mu = pyprob.sample(Normal(settings.PRIOR_MEAN, 1.0))
        pyprob.observe(Normal(mu, Config.SIGMA), name='obs0')
        return mu


def test_constants_read_through_a_module_or_a_class_are_in_the_key():
    m = ThroughModule()
    k0 = _key(m)
    assert k0 is not None
    settings.PRIOR_MEAN = 4.0
    try:
        k1 = _key(m)
        assert k1 is not None and k1 != k0          # (the identity fast path notices too: the float is a new object)
    finally:
        settings.PRIOR_MEAN = 1.0
    assert _key(m) == k0
    Config.SIGMA = 3.0
    try:
        k2 = _key(m)
        assert k2 is not None and k2 != k0
        Config.SIGMA = object()                      # no value fingerprint: no plan
        assert _key(m) is None
    finally:
        Config.SIGMA = 2.0
    assert _key(m) == k0


class ThroughModuleFunction(Model):
    def forward(self):
        mu = pyprob.sample(Normal(settings.helper(1.0), 1.0))
        pyprob.observe(Normal(mu, 1.0), name='obs0')
        return mu


def test_code_reached_through_a_module_rules_the_plan_out():
    assert _key(ThroughModuleFunction()) is None     # settings.helper's reads are not analysed: forward() runs every call


class InstanceFunction(Model):
    def __init__(self, fn):
        super().__init__()
        self.helper = fn

    def forward(self):
        mu = pyprob.sample(Normal(self.helper(1.0), 1.0))
        pyprob.observe(Normal(mu, 1.0), name='obs0')
        return mu


def test_function_kept_on_the_instance_is_keyed_by_code_defaults_and_closure():
    m = InstanceFunction(_scaled)
    k0 = _key(m)
    assert k0 is not None
    old = _scaled.__defaults__
    _scaled.__defaults__ = (3.0,)
    try:
        k1 = _key(m)
        assert k1 is not None and k1 != k0
    finally:
        _scaled.__defaults__ = old
    assert _key(m) == k0
    m.helper = lambda x: 2.0 * x                      # another function: another key
    assert _key(m) not in (None, k0)
    scale = [2.0]
    state = object()
    m.helper = lambda x: x if state else scale[0]    # a closure over an object without a value: no plan
    assert _key(m) is None
