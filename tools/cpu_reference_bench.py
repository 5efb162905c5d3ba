#!/usr/bin/env python3
"""Time the UNMODIFIED reference (pyprob v1.5.0, /root/reference) on this host's cores: BASELINE.md section 3 / SURVEY.md 8(d).

Runs only where /root/reference exists (the build container; the GPU box has no reference).  The reference is imported
read-only with the import stubs of oracle/refstubs (termcolor, sqlitedict, zmq, flatbuffers, pydotplus are absent here and
none is on the timed path).  Figures (all torch CPU, `torch.set_num_threads(cores)`; core count in the output):

  (i)   end-to-end  Model.learn_inference_network(GUM, LSTM hidden 512, batch 1024, online dataset):
        _total_train_traces / _total_train_seconds  (pyprob/nn/inference_network.py:529-531) - the 10x target's denominator
  (ii)  NN-only     zero_grad -> _loss(batch) -> backward -> optimizer.step on a PREBUILT pyprob.nn.Batch
        (pyprob/nn/inference_network.py:486-496), >= 3 warm-up + >= 20 timed steps
  (iii) posterior_results(N, IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK, observe={'obs0': 8, 'obs1': 9}) particles/s
        (pyprob/model.py:180-181, 47-88)
  (iv)  (ii) for GaussianUnknownMeanMarsaglia (ragged traces, one head per address), H = 512, batch 1024

    python tools/cpu_reference_bench.py [--out profiles/r03_cpu_reference.json] [--quick]
"""
import argparse
import io
import json
import math
import os
import sys
import time
import contextlib
import warnings

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get('PYPROB_REFERENCE', '/root/reference')


def host_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def import_reference():
    if not os.path.isdir(os.path.join(REFERENCE, 'pyprob')):
        raise SystemExit('no reference checkout at %s' % REFERENCE)
    sys.path.insert(0, os.path.join(REPO, 'oracle', 'refstubs'))
    sys.path.insert(1, REFERENCE)
    import pyprob          # noqa: F401
    return pyprob


def make_models(pyprob):
    import torch
    from pyprob import Model
    from pyprob.distributions import Normal, Uniform

    class GaussianWithUnknownMean(Model):                   # reference tests/test_inference.py:97-109
        def __init__(self):
            self.prior_mean, self.prior_stddev, self.likelihood_stddev = 1, math.sqrt(5), math.sqrt(2)
            super().__init__('Gaussian with unknown mean')

        def forward(self):
            mu = pyprob.sample(Normal(self.prior_mean, self.prior_stddev))
            likelihood = Normal(mu, self.likelihood_stddev)
            pyprob.observe(likelihood, name='obs0')
            pyprob.observe(likelihood, name='obs1')
            return mu

    class GaussianWithUnknownMeanMarsaglia(Model):          # reference tests/test_inference.py:252-275
        def __init__(self):
            self.prior_mean, self.prior_stddev, self.likelihood_stddev = 1, math.sqrt(5), math.sqrt(2)
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
    return GaussianWithUnknownMean, GaussianWithUnknownMeanMarsaglia


OBS = {'obs0': {'dim': 32}, 'obs1': {'dim': 32}}


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def end_to_end(pyprob, model_cls, num_traces, lstm_dim, batch):
    from pyprob import InferenceNetwork
    model = model_cls()
    with quiet():
        model.learn_inference_network(num_traces=num_traces, inference_network=InferenceNetwork.LSTM, observe_embeddings=OBS,
                                      batch_size=batch, lstm_dim=lstm_dim, lstm_depth=1, proposal_mixture_components=10)
    net = model._inference_network
    return model, dict(traces=int(net._total_train_traces), seconds=round(float(net._total_train_seconds), 3),
                       traces_per_sec=round(net._total_train_traces / net._total_train_seconds, 1),
                       params=int(net._history_num_params[-1]),
                       definition='_total_train_traces / _total_train_seconds, pyprob/nn/inference_network.py:529-531')


def nn_only(pyprob, model, batch, warm, timed):
    """The loop body of optimize() (pyprob/nn/inference_network.py:486-496) on one prebuilt Batch."""
    import torch
    from pyprob.nn import Batch
    net = model._inference_network
    with quiet():
        traces = [next(model._trace_generator(trace_mode=pyprob.TraceMode.PRIOR_FOR_INFERENCE_NETWORK)) for _ in range(batch)]
    b = Batch(traces)
    net._polymorph(b)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    t_batch0 = time.perf_counter()
    Batch(traces)
    t_batch = time.perf_counter() - t_batch0
    times = []
    for i in range(warm + timed):
        t0 = time.perf_counter()
        opt.zero_grad()
        ok, loss = net._loss(b)
        assert ok
        loss.backward()
        opt.step()
        float(loss)
        times.append(time.perf_counter() - t0)
    t = times[warm:]
    mean = sum(t) / len(t)
    return dict(steps=len(t), warmup=warm, batch=batch, sub_batches=len(b.sub_batches),
                ms_per_step=round(mean * 1e3, 2), traces_per_sec=round(batch / mean, 1),
                batch_init_ms=round(t_batch * 1e3, 2), params=sum(p.numel() for p in net.parameters()),
                definition='zero_grad -> _loss(Batch) -> backward -> Adam.step on a prebuilt pyprob.nn.Batch, '
                           'pyprob/nn/inference_network.py:486-496')


def posterior(pyprob, model, n):
    from pyprob import InferenceEngine
    t0 = time.perf_counter()
    with quiet():
        post = model.posterior_results(n, inference_engine=InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK,
                                       observe={'obs0': 8, 'obs1': 9})
    dt = time.perf_counter() - t0
    return dict(particles=n, seconds=round(dt, 3), particles_per_sec=round(n / dt, 1),
                posterior_mean=round(float(post.mean), 4), posterior_stddev=round(float(post.stddev), 4),
                ess=round(float(post.effective_sample_size), 1),
                definition='Model.posterior_results(N, IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK), pyprob/model.py:180-181')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=None)
    ap.add_argument('--quick', action='store_true', help='small trace counts (smoke test of this script)')
    ap.add_argument('--lstm-dim', type=int, default=512)
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--threads', type=int, default=0)
    args = ap.parse_args()
    warnings.filterwarnings('ignore')
    pyprob = import_reference()
    import torch
    cores = host_cores()
    threads = args.threads or cores
    torch.set_num_threads(threads)
    pyprob.seed(123)
    GUM, GUMM = make_models(pyprob)
    H, B = args.lstm_dim, args.batch
    n_e2e = (3 if args.quick else 24) * B
    warm, timed = (1, 3) if args.quick else (3, 20)
    t_all = time.time()
    out = dict(reference='pyprob v%s at %s (unmodified; import stubs oracle/refstubs)' % (pyprob.__version__, REFERENCE),
               torch=torch.__version__, host_cores=cores, torch_threads=threads, lstm_dim=H, batch=B,
               cpu_model=next((l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')), '?'))
    gum, out['gum_end_to_end'] = end_to_end(pyprob, GUM, n_e2e, H, B)
    out['gum_nn_only'] = nn_only(pyprob, gum, B, warm, timed)
    out['gum_posterior'] = posterior(pyprob, gum, 200 if args.quick else 3000)
    gumm, out['gumm_end_to_end'] = end_to_end(pyprob, GUMM, (2 if args.quick else 8) * B, H, B)
    out['gumm_nn_only'] = nn_only(pyprob, gumm, B, warm, max(3, timed // 2))
    out['gumm_posterior'] = posterior(pyprob, gumm, 100 if args.quick else 1000)
    out['wall_s'] = round(time.time() - t_all, 1)
    text = json.dumps(out, indent=1)
    print(text)
    if args.out:
        with open(args.out, 'w') as f:
            f.write(text + '\n')


if __name__ == '__main__':
    main()
