"""The reference's test programs written against the pyprob_amd host API (same source as
reference tests/test_inference.py:97-109 and :252-275 with `pyprob` -> `pyprob_amd`)."""
import math

import torch

import pyprob_amd as pyprob
from pyprob_amd.distributions import Normal, Uniform, Categorical, Poisson, Bernoulli
from pyprob_amd.model import Model


class GaussianWithUnknownMean(Model):
    def __init__(self, prior_mean=1, prior_stddev=math.sqrt(5), likelihood_stddev=math.sqrt(2)):
        self.prior_mean = prior_mean
        self.prior_stddev = prior_stddev
        self.likelihood_stddev = likelihood_stddev
        super().__init__('Gaussian with unknown mean')

    def forward(self):
        mu = pyprob.sample(Normal(self.prior_mean, self.prior_stddev))
        likelihood = Normal(mu, self.likelihood_stddev)
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


class GaussianWithUnknownMeanMarsaglia(Model):
    def __init__(self, prior_mean=1, prior_stddev=math.sqrt(5), likelihood_stddev=math.sqrt(2)):
        self.prior_mean = prior_mean
        self.prior_stddev = prior_stddev
        self.likelihood_stddev = likelihood_stddev
        super().__init__('Gaussian with unknown mean (Marsaglia)')

    def marsaglia(self, mean, stddev):
        uniform = Uniform(-1, 1)
        s = 1
        while float(s) >= 1:
            x = pyprob.sample(uniform)
            y = pyprob.sample(uniform)
            s = x * x + y * y
        return mean + stddev * (x * torch.sqrt(-2 * torch.log(s) / s))

    def forward(self):
        mu = self.marsaglia(self.prior_mean, self.prior_stddev)
        likelihood = Normal(mu, self.likelihood_stddev)
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


class CategoricalThenNormal(Model):
    def forward(self):
        c = pyprob.sample(Categorical([0.2, 0.3, 0.5]))
        mu = pyprob.sample(Normal(c.float() * 2.0 - 1.0, 1.5))
        likelihood = Normal(mu, 0.8)
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


class GaussianWithUnknownMeanMarsagliaLockStep(GaussianWithUnknownMeanMarsaglia):
    """The same program with the rejection loop written as a tensor condition (`while s >= 1:` instead of
    `while float(s) >= 1:`): runs unchanged one particle at a time AND with all particles in lock step, where the
    condition is a per-particle branch (pyprob_amd.state.ParticleTensor)."""

    def marsaglia(self, mean, stddev):
        uniform = Uniform(-1, 1)
        s = 1
        while s >= 1:
            x = pyprob.sample(uniform)
            y = pyprob.sample(uniform)
            s = x * x + y * y
        return mean + stddev * (x * torch.sqrt(-2 * torch.log(s) / s))


class PoissonThenNormal(Model):
    """n ~ Poisson(4); mu ~ Normal(n / 2, 1); two Normal observations (the program of the `poi` golden case):
    exercises the Poisson proposal head (TruncatedNormal mixture on [0, 40])."""

    def forward(self):
        n = pyprob.sample(Poisson(4.0))
        mu = pyprob.sample(Normal(n * 0.5, 1.0))
        likelihood = Normal(mu, 0.8)
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


class BernoulliThenNormal(Model):
    """b ~ Bernoulli(0.3); mu ~ Normal(2 b - 1, 1); two Normal observations (the program of the `ber` golden case)."""

    def forward(self):
        b = pyprob.sample(Bernoulli(0.3))
        mu = pyprob.sample(Normal(b * 2.0 - 1.0, 1.0))
        likelihood = Normal(mu, 0.8)
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu
