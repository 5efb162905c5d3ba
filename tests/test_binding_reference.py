"""The reference-side binding EXECUTED: pyprob's own `Model.learn_inference_network`, `optimize()`
loop, `_save` / `_load` and `posterior` run through pyprob_amd/binding.py.

Runs only where the reference is available (this container: /root/reference + the import stubs of oracle/refstubs);
skipped on the GPU box. The device is absent here, so the `pyprob_hip::*` operators execute their oracle-backed CPU
kernels (tests/oracle_ops.py) and the engine's buffers are CPU tensors - everything ABOVE the operators (parameter
re-binding, `_polymorph` growth, the autograd.Function, `grad is None` semantics, HipAdam inside the reference's
optimizer / scheduler protocol, pickling, the coroutine `_traces`) is the code that ships.
The GPU twin of the gradient check is tests/test_gpu_binding.py."""
import functools
import math
import os
import sys
import warnings

import numpy as np
import pytest
import torch

REFERENCE = '/root/reference'
if not os.path.isdir(os.path.join(REFERENCE, 'pyprob')):
    pytest.skip('the reference tree is not available on this machine', allow_module_level=True)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle', 'refstubs'))
sys.path.insert(1, REFERENCE)

import pyprob  # noqa: E402
from pyprob import InferenceEngine, InferenceNetwork, Model  # noqa: E402
from pyprob.distributions import Normal, Uniform  # noqa: E402

import oracle_ops  # noqa: E402  (CPU kernels of the operators)
import pyprob_amd.binding as hip  # noqa: E402
from oracle import ic_oracle as O  # noqa: E402



def _cpu_engine(spec, device):
    eng = oracle_ops.CpuBufferEngine(spec)
    eng._use_ops = True                    # (ICEngine's compute methods go through the operators: oracle-backed here)
    return eng


hip._HipNetworkMixin._hip_device = 'cpu'
hip._HipNetworkMixin._hip_engine_factory = staticmethod(_cpu_engine)
os.environ['PP_PYTHON_LOOP'] = '1'         # the runs of minibatches inside one C call (pp_train_steps) need the device
# The tests that compare a bound run with a stock run SEED BY SEED need both to consume torch's generator identically: they pin
# pyprob's own per-trace loop (PYPROB_HIP_FAST_TRAIN=0 / PYPROB_HIP_LOCKSTEP=0). The batched paths that install() selects by
# default (pyprob_amd/pyprob_host.py) have their own tests at the end of this file.
os.environ['PYPROB_HIP_FAST_TRAIN'] = '0'
os.environ['PYPROB_HIP_LOCKSTEP'] = '0'

IC = InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK
EMB = {'obs0': {'dim': 16}, 'obs1': {'dim': 16}}


class GaussianWithUnknownMean(Model):           # reference tests/test_inference.py:97-109
    def __init__(self):
        super().__init__('Gaussian with unknown mean')

    def forward(self):
        mu = pyprob.sample(Normal(1, math.sqrt(5)))
        likelihood = Normal(mu, math.sqrt(2))
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


class GaussianWithUnknownMeanMarsaglia(Model):  # reference tests/test_inference.py:252-275, as written
    def __init__(self):
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
        mu = self.marsaglia(1, math.sqrt(5))
        likelihood = Normal(mu, math.sqrt(2))
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


@pytest.fixture
def installed():
    hip.install()
    yield
    hip.uninstall()


def _train(model_cls, use_hip, num_traces, seed=3, **kw):
    (hip.install if use_hip else hip.uninstall)()
    try:
        pyprob.seed(seed)
        model = model_cls()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            model.learn_inference_network(num_traces=num_traces, batch_size=32, observe_embeddings=EMB,
                                          inference_network=InferenceNetwork.LSTM, lstm_dim=24, learning_rate_init=1e-3,
                                          **kw)
        return model
    finally:
        hip.uninstall()


@pytest.mark.parametrize('program', [GaussianWithUnknownMean, GaussianWithUnknownMeanMarsaglia], ids=['gum', 'gumm'])
def test_learn_inference_network_matches_the_stock_reference(program):
    """Same seed, same program: the reference's optimize() loop driving the HIP-side network produces the stock
    reference's loss trajectory, parameter set and final weights (fp32 reference vs fp64 oracle kernels: 1e-4)."""
    stock = _train(program, False, 320)
    bound = _train(program, True, 320)
    ns, nb = stock._inference_network, bound._inference_network
    assert type(nb).__name__ == 'InferenceNetworkLSTMHip' and isinstance(nb, type(ns))
    assert isinstance(nb, torch.nn.Module)
    assert [n for n, _ in nb.named_parameters()] == [n for n, _ in ns.named_parameters()]
    assert nb._total_train_traces == ns._total_train_traces and nb._total_train_iterations == ns._total_train_iterations
    np.testing.assert_allclose(nb._history_train_loss, ns._history_train_loss, rtol=2e-4, atol=2e-4)
    assert nb._history_num_params == ns._history_num_params
    sd_s, sd_b = ns.state_dict(), nb.state_dict()
    for k in sd_s:
        np.testing.assert_allclose(sd_b[k].numpy(), sd_s[k].numpy(), rtol=5e-3, atol=5e-4, err_msg=k)
    # every parameter IS a view of the engine's flat buffer
    eng = nb._hip_engine
    for name, p in nb.named_parameters():
        assert isinstance(p, torch.nn.Parameter) and p.data_ptr() == eng.tensor(name).data_ptr()
    for a, layer in nb._layers_proposal.items():
        assert layer._total_train_iterations == ns._layers_proposal[a]._total_train_iterations


def test_gradients_and_grad_none_set_match_the_reference(installed):
    """One minibatch through both `_loss` implementations on identical weights: same loss, same gradients, and
    `grad is None` for exactly the parameters the reference's autograd leaves untouched (the presence map of
    _distributed_sync_grad, inference_network.py:300)."""
    from pyprob.nn import Batch
    stock = _train(GaussianWithUnknownMeanMarsaglia, False, 320)
    hip.install()
    ns = stock._inference_network
    pyprob.seed(11)
    gen = stock._trace_generator(trace_mode=pyprob.TraceMode.PRIOR_FOR_INFERENCE_NETWORK)
    traces = []
    while len(traces) < 24:            # short traces only: several addresses of the network stay untouched
        t = next(gen)
        if t.length_controlled <= 4:
            traces.append(t)
    batch = Batch(traces)
    nb = hip.InferenceNetworkLSTMHip(model=stock, observe_embeddings=EMB, lstm_dim=24)
    nb._init_layers_observe_embedding(EMB, example_trace=traces[0])
    nb._init_layers()
    nb._layers_initialized = True
    full = Batch([next(gen) for _ in range(200)])
    ns._polymorph(full)
    nb._polymorph(full)
    nb.load_state_dict(ns.state_dict())               # in place: writes through the views into the flat buffer
    ns.zero_grad()
    ok_s, loss_s = ns._loss(batch)
    loss_s.backward()
    ok_b, loss_b = nb._loss(batch)
    assert ok_s and ok_b and loss_b.dim() == 0 and loss_b.requires_grad
    loss_b.backward()
    assert abs(float(loss_b) - float(loss_s)) < 1e-4 * abs(float(loss_s))
    gs = dict((n, p.grad) for n, p in ns.named_parameters())
    none_b = {n for n, p in nb.named_parameters() if p.grad is None}
    assert none_b == {n for n, g in gs.items() if g is None} and len(none_b) > 0
    for n, p in nb.named_parameters():
        if p.grad is not None:
            ref = gs[n].numpy()
            err = np.abs(p.grad.numpy() - ref).max() / max(np.abs(ref).max(), 1e-6)
            assert err < 2e-3, (n, err)
    # a scaled loss scales the gradients (autograd contract of the Function)
    g1 = {n: p.grad.clone() for n, p in nb.named_parameters() if p.grad is not None}
    for p in nb.parameters():
        p.grad = None
    ok, loss = nb._loss(batch)
    (3.0 * loss).backward()
    for n, p in nb.named_parameters():
        if n in g1:
            np.testing.assert_allclose(p.grad.numpy(), 3.0 * g1[n].numpy(), rtol=1e-5, atol=1e-7)
    # unknown address -> (False, 0) like inference_network_lstm.py:150-152
    nb2 = hip.InferenceNetworkLSTMHip(model=stock, observe_embeddings=EMB, lstm_dim=24)
    nb2._init_layers_observe_embedding(EMB, example_trace=traces[0])
    nb2._init_layers()
    nb2._layers_initialized = True
    nb2._polymorph(Batch([t for t in traces if t.length_controlled == 2][:4]))
    longer = [t for t in full.traces if t.length_controlled >= 4][:4]
    assert nb2._loss(Batch(longer)) == (False, 0)


def test_save_load_round_trip_and_continue_training(installed, tmp_path, monkeypatch):
    """`_save` pickles the module (inference_network.py:162-196): the HIP-side state is rebuilt on load, Adam moments and
    per-tensor step counts come back through HipAdam.load_state_dict, and training continues on the loaded network."""
    monkeypatch.setattr(torch, 'load', functools.partial(torch.load, weights_only=False))
    bound = _train(GaussianWithUnknownMean, True, 320)
    hip.install()
    net = bound._inference_network
    fn = str(tmp_path / 'net.network')
    bound.save_inference_network(fn)
    other = GaussianWithUnknownMean()
    other.load_inference_network(fn)
    ln = other._inference_network
    assert type(ln).__name__ == 'InferenceNetworkLSTMHip' and ln._hip_engine is not None
    for (n1, p1), (n2, p2) in zip(net.named_parameters(), ln.named_parameters()):
        assert n1 == n2 and torch.equal(p1.detach(), p2.detach())
        assert p2.data_ptr() == ln._hip_engine.tensor(n2).data_ptr()
    assert torch.equal(ln._hip_engine.tensor_step, net._hip_engine.tensor_step)
    assert torch.equal(ln._hip_engine.exp_avg, net._hip_engine.exp_avg)
    assert ln._total_train_traces == net._total_train_traces and ln._history_train_loss == net._history_train_loss
    before = ln._total_train_traces
    pyprob.seed(5)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        other.learn_inference_network(num_traces=64, batch_size=32, observe_embeddings=EMB,
                                      inference_network=InferenceNetwork.LSTM, lstm_dim=24)
    assert ln._total_train_traces == before + 64 and int(ln._hip_engine.tensor_step.max()) == 12
    assert np.isfinite(ln._history_train_loss[-1])


def test_learning_rate_scheduler_and_weight_decay_drive_hip_adam(installed):
    """LambdaLR (POLY2, inference_network.py:357-379) steps the HipAdam instance like any torch optimizer."""
    pyprob.seed(2)
    model = GaussianWithUnknownMean()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model.learn_inference_network(num_traces=160, batch_size=32, observe_embeddings=EMB, lstm_dim=24,
                                      inference_network=InferenceNetwork.LSTM, learning_rate_init=1e-3, learning_rate_end=1e-5,
                                      learning_rate_scheduler_type=pyprob.LearningRateScheduler.POLY2, num_traces_end=1000,
                                      weight_decay=1e-4)
    net = model._inference_network
    assert isinstance(net._optimizer, hip.HipAdam)
    lr = net._optimizer.param_groups[0]['lr']
    want = (1e-3 - 1e-5) * (1 - 160 / 1000) ** 2 + 1e-5
    assert abs(lr - want) < 1e-9
    assert net._optimizer.param_groups[0]['weight_decay'] == 1e-4


@pytest.mark.parametrize('program,case', [(GaussianWithUnknownMean, 'gum'), (GaussianWithUnknownMeanMarsaglia, 'gumm')],
                         ids=['gum', 'gumm'])
def test_posterior_through_the_binding_is_rescored_by_the_oracle(installed, program, case):
    """`Model.posterior(IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK)` of the reference, particles as coroutines parked in
    `_infer_step`: the Empirical holds the reference's own Trace objects; their log-weights (summed by the reference's
    Trace.end from the reference's log p and the served log q) equal the oracle's re-scoring of their values."""
    from is_helpers import rescore
    bound = _train(program, True, 640)
    hip.install()
    net = bound._inference_network
    observe = {'obs0': 8, 'obs1': 9}
    pyprob.seed(4)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        post = bound.posterior(60, inference_engine=IC, observe=observe)
    assert post.length == 60 and hasattr(post, '_hip_coroutine_stats')
    traces = [post._get_value(i) for i in range(post.length)]
    params = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    meta = dict(obs_names=['obs0', 'obs1'], mixture_components=10)
    ref = rescore(case, meta, params, traces, observe, math.sqrt(2))
    lw = np.array([float(t.log_importance_weight) for t in traces])
    np.testing.assert_allclose(lw, ref, rtol=1e-4, atol=1e-4)
    st = post._hip_coroutine_stats
    assert st['statements'] == sum(t.length_controlled for t in traces)
    if case == 'gumm':
        assert st['group_calls'] < st['statements']        # batches, not one network call per statement
    # the per-particle path of the reference (no coroutines) agrees statistically and is scored the same way
    os.environ['PYPROB_HIP_COROUTINES'] = '0'
    try:
        pyprob.seed(4)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            one = bound.posterior(12, inference_engine=IC, observe=observe)
    finally:
        del os.environ['PYPROB_HIP_COROUTINES']
    traces1 = [one._get_value(i) for i in range(one.length)]
    ref1 = rescore(case, meta, params, traces1, observe, math.sqrt(2))
    np.testing.assert_allclose([float(t.log_importance_weight) for t in traces1], ref1, rtol=1e-4, atol=1e-4)


def test_operators_fail_loudly_without_a_kernel_for_the_device():
    """The product registers device kernels only; a dispatch key without a kernel raises (no silent fallback)."""
    x = torch.zeros(4, device='meta')
    with pytest.raises(NotImplementedError):
        torch.ops.pyprob_hip.log_prob(0, x, 0, x, 0, x, 4)


@pytest.mark.parametrize('n,batch_size,world,num_buckets', [(1024, 16, 4, 5), (960, 16, 2, None), (2048, 32, 8, 3)])
def test_array_sampler_equals_the_reference_sampler(monkeypatch, n, batch_size, world, num_buckets):
    """pyprob_amd.parallel.DistributedTraceBatchSampler (one index table + row ranges) against the reference's list-of-lists
    sampler (pyprob/nn/dataset.py:328-400) on datasets that need no random drop: same minibatches, same buckets, same
    epoch-seeded bucket order, same per-rank selection."""
    import torch.distributed as dist
    from pyprob.nn.dataset import DistributedTraceBatchSampler as RefSampler, OfflineDataset
    from pyprob_amd.parallel import DistributedTraceBatchSampler
    ds = object.__new__(OfflineDataset)
    ds._sorted_indices = list(np.random.RandomState(1).permutation(n))
    ds._length = n
    monkeypatch.setattr(OfflineDataset, '__len__', lambda self: self._length)
    monkeypatch.setattr(dist, 'get_world_size', lambda *a, **k: world)
    for rank in range(world):
        monkeypatch.setattr(dist, 'get_rank', lambda *a, **k: rank)
        ref = RefSampler(ds, batch_size, shuffle_batches=False, num_buckets=num_buckets)
        mine = DistributedTraceBatchSampler(ds._sorted_indices, batch_size, rank, world, num_buckets, shuffle_batches=False)
        assert len(mine) == len(ref) and mine._bucket_size == ref._bucket_size
        assert [[list(b) for b in bk] for bk in mine._buckets] == [[list(b) for b in bk] for bk in ref._buckets]
        for epoch in range(3):
            a = [list(b) for b in ref]
            b = [list(x) for x in mine]
            assert a == b and mine._current_bucket_id == ref._current_bucket_id


@pytest.mark.parametrize('opt', ['SGD', 'ADAM_LARC', 'SGD_LARC'])
def test_the_other_optimizers_match_the_stock_reference(opt, tmp_path, monkeypatch):
    """Optimizer.SGD (nesterov momentum) and the LARC-wrapped optimizers of _create_optimizer (inference_network.py:343-355)
    through HipSGD / HipAdam(larc=True): same loss trajectory and weights as the stock reference, with weight decay, on the
    program whose minibatches leave some proposal layers without gradient; the optimizer state survives _save / _load."""
    kw = dict(optimizer_type=getattr(pyprob.Optimizer, opt), weight_decay=1e-3, momentum=0.8)
    stock = _train(GaussianWithUnknownMeanMarsaglia, False, 320, **kw)
    bound = _train(GaussianWithUnknownMeanMarsaglia, True, 320, **kw)
    ns, nb = stock._inference_network, bound._inference_network
    want = 'HipSGD' if opt.startswith('SGD') else 'HipAdam'
    assert type(nb._optimizer).__name__ == want and nb._optimizer._larc == opt.endswith('LARC')
    np.testing.assert_allclose(nb._history_train_loss, ns._history_train_loss, rtol=2e-4, atol=2e-4)
    sd_s, sd_b = ns.state_dict(), nb.state_dict()
    for k in sd_s:
        np.testing.assert_allclose(sd_b[k].numpy(), sd_s[k].numpy(), rtol=5e-3, atol=5e-4, err_msg=k)
    # optimizer state in torch's format, and back
    inner = ns._optimizer.optim if opt.endswith('LARC') else ns._optimizer
    st_s, st_b = inner.state_dict()['state'], nb._optimizer.state_dict()['state']
    assert set(st_b.keys()) == set(st_s.keys())
    key = 'momentum_buffer' if opt.startswith('SGD') else 'exp_avg'
    for i in st_s:
        np.testing.assert_allclose(st_b[i][key].numpy(), st_s[i][key].numpy(), rtol=5e-3, atol=1e-5)
    monkeypatch.setattr(torch, 'load', functools.partial(torch.load, weights_only=False))
    hip.install()
    try:
        fn = str(tmp_path / 'net.network')
        bound.save_inference_network(fn)
        other = GaussianWithUnknownMeanMarsaglia()
        other.load_inference_network(fn)
        ln = other._inference_network
        assert type(ln._optimizer).__name__ == want and ln._optimizer._larc == opt.endswith('LARC')
        assert torch.equal(ln._hip_engine.exp_avg, nb._hip_engine.exp_avg)
        assert ln._optimizer.param_groups[0]['lr'] == nb._optimizer.param_groups[0]['lr']
    finally:
        hip.uninstall()


def _dp_worker(rank, world, port, use_hip, dataset_dir, out):
    """One rank of pyprob's own data-parallel training (optimize(distributed_backend=...), inference_network.py:381-599)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import contextlib
    import io
    (hip.install if use_hip else hip.uninstall)()
    pyprob.seed(7)
    model = GaussianWithUnknownMeanMarsaglia()
    with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
        warnings.simplefilter('ignore')
        model.learn_inference_network(num_traces=256, batch_size=16, dataset_dir=dataset_dir, observe_embeddings=EMB,
                                      inference_network=InferenceNetwork.LSTM, lstm_dim=24, learning_rate_init=1e-3,
                                      distributed_backend='gloo', distributed_num_buckets=2, pre_generate_layers=True,
                                      distributed_params_sync_every_iter=5)
    net = model._inference_network
    torch.save(dict(hist=list(net._history_train_loss), dist_hist=list(net._distributed_history_train_loss),
                    sd={k: v.detach().clone() for k, v in net.state_dict().items()}, cls=type(net).__name__,
                    iters=net._total_train_iterations, traces=net._total_train_traces,
                    lr=net._optimizer.param_groups[0]['lr']), '{}.{}.{}'.format(out, int(use_hip), rank))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_training_through_the_binding_matches_the_stock_reference(tmp_path, monkeypatch):
    """pyprob's OWN distributed loop on two gloo ranks - DistributedTraceBatchSampler over an OfflineDataset, _polymorph
    skipped (pre-generated layers), _distributed_sync_parameters, loss.backward(), _distributed_sync_grad, optimizer.step(),
    _distributed_update_train_loss (inference_network.py:290-333, 461-531) - with the HIP-side network bound in: its ONE
    all-reduce of [flat gradients | presence map] and the 1 / world_size inside the optimizer kernel give the stock
    reference's trajectory; the ranks touch different proposal layers and still hold identical parameters."""
    import torch.multiprocessing as mp
    monkeypatch.setattr(torch, 'load', functools.partial(torch.load, weights_only=False))
    hip.uninstall()
    pyprob.seed(21)
    d = str(tmp_path / 'ds')
    os.makedirs(d)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        GaussianWithUnknownMeanMarsaglia().save_dataset(d, 256, 64)
    out = str(tmp_path / 'dp')
    world = 2
    for use_hip in (False, True):
        port = 39500 + (os.getpid() + 7 * use_hip) % 2000
        mp.spawn(_dp_worker, args=(world, port, use_hip, d, out), nprocs=world, join=True)
    res = {(h, r): torch.load('{}.{}.{}'.format(out, h, r), weights_only=False) for h in (0, 1) for r in range(world)}
    assert res[(1, 0)]['cls'] == 'InferenceNetworkLSTMHip' and res[(0, 0)]['cls'] == 'InferenceNetworkLSTM'
    for h in (0, 1):                                  # the ranks of one run never diverge
        a, b = res[(h, 0)], res[(h, 1)]
        assert a['iters'] == b['iters'] and a['traces'] == b['traces'] and a['lr'] == b['lr']
        for k in a['sd']:
            assert torch.equal(a['sd'][k], b['sd'][k]), (h, k)
        np.testing.assert_allclose(a['dist_hist'], b['dist_hist'], rtol=1e-6)
    for r in range(world):                            # bound run = stock run, rank by rank
        s, b = res[(0, r)], res[(1, r)]
        assert s['iters'] == b['iters'] > 0 and s['traces'] == b['traces']
        assert abs(s['lr'] - b['lr']) < 1e-12 and abs(s['lr'] - 1e-3 * math.sqrt(world)) < 1e-9       # :448
        np.testing.assert_allclose(b['hist'], s['hist'], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(b['dist_hist'], s['dist_hist'], rtol=2e-4, atol=2e-4)
        for k in s['sd']:
            np.testing.assert_allclose(b['sd'][k].numpy(), s['sd'][k].numpy(), rtol=5e-3, atol=5e-4, err_msg=k)


class _CategoricalThenNormal(Model):            # the program of the `cat` golden case, with the reference's classes
    def __init__(self):
        super().__init__('categorical then normal')

    def forward(self):
        from pyprob.distributions import Categorical
        c = pyprob.sample(Categorical([0.2, 0.3, 0.5]))
        mu = pyprob.sample(Normal(c.float() * 2.0 - 1.0, 1.5))
        likelihood = Normal(mu, 0.8)
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


class _PoissonThenNormal(Model):                # `poi`
    def __init__(self):
        super().__init__('poisson then normal')

    def forward(self):
        from pyprob.distributions import Poisson
        n = pyprob.sample(Poisson(4.0))
        mu = pyprob.sample(Normal(n * 0.5, 1.0))
        likelihood = Normal(mu, 0.8)
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


class _BernoulliThenNormal(Model):              # `ber`
    def __init__(self):
        super().__init__('bernoulli then normal')

    def forward(self):
        from pyprob.distributions import Bernoulli
        b = pyprob.sample(Bernoulli(0.3))
        mu = pyprob.sample(Normal(b * 2.0 - 1.0, 1.0))
        likelihood = Normal(mu, 0.8)
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


@pytest.mark.parametrize('program,network', [(_CategoricalThenNormal, InferenceNetwork.LSTM),
                                             (_PoissonThenNormal, InferenceNetwork.LSTM),
                                             (_BernoulliThenNormal, InferenceNetwork.LSTM),
                                             (GaussianWithUnknownMeanMarsaglia, InferenceNetwork.FEEDFORWARD),
                                             (_CategoricalThenNormal, InferenceNetwork.FEEDFORWARD)],
                         ids=['cat-lstm', 'poi-lstm', 'ber-lstm', 'gumm-ff', 'cat-ff'])
def test_every_proposal_layer_and_both_networks_through_the_binding(program, network):
    """The reference's training loop on programs with Categorical / Poisson / Bernoulli priors (ProposalCategoricalCategorical,
    ProposalPoissonTruncatedNormalMixture, ProposalBernoulliBernoulli, inference_network_lstm.py:52-66) and with
    InferenceNetworkFeedForward (inference_network_feedforward.py), HIP-side network bound in: the stock reference's loss
    trajectory, parameter names and weights; then a posterior through the bound network (finite weights, the reference's
    Empirical)."""
    def train(use_hip):
        (hip.install if use_hip else hip.uninstall)()
        try:
            pyprob.seed(5)
            model = program()
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                model.learn_inference_network(num_traces=256, batch_size=32, observe_embeddings=EMB, inference_network=network,
                                              lstm_dim=24, learning_rate_init=1e-3)
            return model
        finally:
            hip.uninstall()
    stock, bound = train(False), train(True)
    ns, nb = stock._inference_network, bound._inference_network
    assert type(nb).__name__ == type(ns).__name__ + 'Hip' and isinstance(nb, type(ns))
    assert [n for n, _ in nb.named_parameters()] == [n for n, _ in ns.named_parameters()]
    np.testing.assert_allclose(nb._history_train_loss, ns._history_train_loss, rtol=3e-4, atol=3e-4)
    sd_s, sd_b = ns.state_dict(), nb.state_dict()
    for k in sd_s:
        np.testing.assert_allclose(sd_b[k].numpy(), sd_s[k].numpy(), rtol=5e-3, atol=5e-4, err_msg=k)
    hip.install()
    # (the Poisson proposal is a mixture of truncated normals, proposal_poisson_truncated_normal_mixture.py: its draws are not
    # integers, and torch's Poisson.log_prob rejects them since argument validation became the default - in the stock
    # reference too; the golden records were made with validation off as well, tests/golden/make_golden.py)
    validate = torch.distributions.Distribution._validate_args
    torch.distributions.Distribution.set_default_validate_args(False)
    try:
        pyprob.seed(6)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            post = bound.posterior(40, inference_engine=IC, observe={'obs0': 0.5, 'obs1': 0.8})
        assert post.length == 40
        lw = np.array([float(post._get_value(i).log_importance_weight) for i in range(post.length)])
        assert np.isfinite(lw).all() and np.isfinite(float(post.map(lambda t: t.result).mean))
    finally:
        torch.distributions.Distribution.set_default_validate_args(validate)
        hip.uninstall()


@pytest.mark.parametrize('kw', [dict(lstm_depth=2), dict(observe_embeddings={'obs0': {'dim': 16, 'depth': 3}, 'obs1': {'dim': 8, 'depth': 1}})],
                         ids=['lstm-depth-2', 'embedding-depths-3-1'])
def test_stacked_lstm_and_embedding_depths_through_the_binding(kw):
    """nn.LSTM(input, hidden, lstm_depth) (inference_network_lstm.py:31) and EmbeddingFeedForward(depth)
    (inference_network.py:110-118) created by the reference's code, re-bound to the flat buffer: the stock reference's
    trajectory and weights, parameter names included (`weight_ih_l1`, `_layers.2.weight`, ...)."""
    args = dict(observe_embeddings=EMB, lstm_depth=1)
    args.update(kw)

    def train(use_hip):
        (hip.install if use_hip else hip.uninstall)()
        try:
            pyprob.seed(8)
            model = GaussianWithUnknownMeanMarsaglia()
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                model.learn_inference_network(num_traces=256, batch_size=32, inference_network=InferenceNetwork.LSTM, lstm_dim=16,
                                              learning_rate_init=1e-3, **args)
            return model._inference_network
        finally:
            hip.uninstall()
    ns, nb = train(False), train(True)
    names = [n for n, _ in ns.named_parameters()]
    assert [n for n, _ in nb.named_parameters()] == names
    if 'lstm_depth' in kw:
        assert '_layers_lstm.weight_hh_l1' in names and nb._hip_engine.spec.lstm_depth == 2
    else:
        assert '_layers_observe_embedding.obs0._layers.2.weight' in names
        assert '_layers_observe_embedding.obs1._layers.1.weight' not in names
    np.testing.assert_allclose(nb._history_train_loss, ns._history_train_loss, rtol=3e-4, atol=3e-4)
    sd_s, sd_b = ns.state_dict(), nb.state_dict()
    for k in sd_s:
        np.testing.assert_allclose(sd_b[k].numpy(), sd_s[k].numpy(), rtol=5e-3, atol=5e-4, err_msg=k)


def test_array_sampler_equals_the_reference_sampler_on_arbitrary_sizes(monkeypatch):
    """The same comparison on arbitrary dataset sizes (hypothesis), including those where the reference drops a seed-0 random
    subset of traces so that the number of minibatches divides by the world size (dataset.py:337-343, util.drop_items): the
    SAME traces are dropped, minibatches, buckets and per-rank order are identical; configurations the reference rejects are
    rejected."""
    import contextlib
    import io
    import torch.distributed as dist
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st
    from pyprob.nn.dataset import DistributedTraceBatchSampler as RefSampler, OfflineDataset
    from pyprob_amd.parallel import DistributedTraceBatchSampler
    monkeypatch.setattr(OfflineDataset, '__len__', lambda self: self._length)
    seen = dict(dropped=0, ok=0)

    @settings(derandomize=True, max_examples=150, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
    @given(st.integers(8, 700), st.sampled_from([1, 3, 8, 16]), st.integers(1, 5), st.sampled_from([None, 1, 2, 3, 7]),
           st.integers(0, 10 ** 6))
    def check(n, batch_size, world, num_buckets, seed):
        ds = object.__new__(OfflineDataset)
        ds._sorted_indices = list(np.random.RandomState(seed).permutation(n))
        ds._length = n
        monkeypatch.setattr(dist, 'get_world_size', lambda *a, **k: world)
        for rank in range(world):
            monkeypatch.setattr(dist, 'get_rank', lambda *a, **k: rank)
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    ref = RefSampler(ds, batch_size, shuffle_batches=False, num_buckets=num_buckets)
            except (RuntimeError, IndexError, ZeroDivisionError, ValueError):
                with pytest.raises((RuntimeError, ZeroDivisionError, ValueError)):
                    DistributedTraceBatchSampler(ds._sorted_indices, batch_size, rank, world, num_buckets, shuffle_batches=False)
                continue
            mine = DistributedTraceBatchSampler(ds._sorted_indices, batch_size, rank, world, num_buckets, shuffle_batches=False)
            assert [[list(b) for b in bk] for bk in mine._buckets] == [[list(b) for b in bk] for bk in ref._buckets]
            for epoch in range(2):
                assert [list(b) for b in ref] == [list(x) for x in mine]
            seen['ok'] += 1
            seen['dropped'] += int((n // batch_size) % world != 0)
    check()
    assert seen['ok'] > 50 and seen['dropped'] > 10


def test_empirical_statistics_equal_the_reference_class():
    """pyprob_amd.distributions.Empirical (vectorised reductions) against pyprob.distributions.Empirical
    (pyprob/distributions/empirical.py: finalize :298-309, expectation :451-466, moments :668-690, ESS :758-766, mode /
    median / min / max) on arbitrary weighted samples, log-weights down to -150 (the range importance sampling produces)."""
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st
    from pyprob.distributions import Empirical as RefEmpirical
    from pyprob_amd.distributions import Empirical

    @settings(derandomize=True, max_examples=80, deadline=None, suppress_health_check=[HealthCheck.too_slow])
    @given(st.integers(1, 200), st.integers(0, 10 ** 6), st.sampled_from([None, 1.0, 30.0, 150.0]))
    def check(n, seed, spread):
        rng = np.random.RandomState(seed)
        values = rng.normal(size=n) * 3 + 1
        lw = None if spread is None else -rng.uniform(0, spread, n)
        as_tensors = bool(seed & 1)            # forward() results are 0-d tensors or plain numbers
        vals = [torch.tensor(float(v)) for v in values] if as_tensors else [float(v) for v in values]
        ref = RefEmpirical(values=list(vals), log_weights=None if lw is None else [float(w) for w in lw])
        mine = Empirical(values=list(vals), log_weights=None if lw is None else [float(w) for w in lw])
        assert mine.length == ref.length == n
        for name in ('mean', 'variance', 'stddev', 'effective_sample_size'):
            a, b = float(getattr(mine, name)), float(getattr(ref, name))
            assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (name, a, b)
        if n >= 3 and float(ref.stddev) > 1e-3:
            for name in ('skewness', 'kurtosis'):
                a, b = float(getattr(mine, name)), float(getattr(ref, name))
                assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (name, a, b)
        # (the median of a weighted Empirical is the median of a random resample in both classes, the mode of an unweighted one
        # a count over hashable values: compared where they are deterministic)
        for name in ('min', 'max') + (('median',) if lw is None else ('mode',)):
            a, b = float(getattr(mine, name)), float(getattr(ref, name))
            assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (name, a, b)
        a, b = float(mine.expectation(lambda x: x * x + 1)), float(ref.expectation(lambda x: x * x + 1))
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b))
        np.testing.assert_allclose(np.asarray(mine.weights_numpy(), np.float64), ref.weights_numpy(), rtol=1e-4, atol=1e-9)
    check()


def test_mirror_runtime_produces_the_reference_addresses():
    """Addresses are `<bytecode offset>__<call stack>__<distribution>__<instance>` (pyprob/state.py:34-35, 168-186): the same
    method source executed under pyprob and under the pyprob_amd trace runtime gives the same address strings, instance
    counters included (the golden networks recorded from the reference are keyed by them)."""
    import models as M
    pyprob.seed(1)
    ref_model = GaussianWithUnknownMeanMarsaglia()
    ref_traces = [next(ref_model._trace_generator(trace_mode=pyprob.TraceMode.PRIOR)) for _ in range(60)]
    torch.manual_seed(1)
    mir_model = M.GaussianWithUnknownMeanMarsaglia()
    from pyprob_amd.state import TraceMode
    gen = mir_model._trace_generator(trace_mode=TraceMode.PRIOR)
    mir_traces = [next(gen) for _ in range(60)]

    def by_length(traces):
        out = {}
        for t in traces:
            out.setdefault(len(t.variables_controlled), [v.address for v in t.variables_controlled])
        return out
    ref, mir = by_length(ref_traces), by_length(mir_traces)
    common = sorted(set(ref) & set(mir))
    assert len(common) >= 2 and 2 in common
    for n in common:                       # x_1, y_1, x_2, y_2, ...: same offsets, same instance numbering
        assert ref[n] == mir[n], (n, ref[n], mir[n])
    meta = __import__('json').load(open(os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'gumm_meta.json')))
    assert set(ref[2]) <= set(meta['addresses'])          # ... and they are the addresses of the recorded golden network
    # the named observables: same distribution suffix and instance (the offset differs: the two forward() sources do)
    for name in ('obs0', 'obs1'):
        a, b = ref_traces[0].named_variables[name].address, mir_traces[0].named_variables[name].address
        assert a.split('__', 1)[1] == b.split('__', 1)[1], (a, b)


def test_host_distributions_equal_the_reference_classes():
    """pyprob_amd.distributions.{Normal, Uniform, Poisson, Categorical, Bernoulli}.log_prob (the host-side prior / likelihood
    scoring of per-trace runs, pyprob/distributions/*.py) against the reference classes on arbitrary parameters and values,
    support edges included (Uniform: closed lower, open upper edge)."""
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st
    import pyprob.distributions as R
    import pyprob_amd.distributions as D
    validate = torch.distributions.Distribution._validate_args
    torch.distributions.Distribution.set_default_validate_args(False)

    @settings(derandomize=True, max_examples=200, deadline=None, suppress_health_check=[HealthCheck.too_slow])
    @given(st.integers(0, 10 ** 6))
    def check(seed):
        rng = np.random.RandomState(seed)
        mean, sd = float(rng.normal() * 3), float(rng.uniform(0.05, 4))
        low = float(rng.normal() * 2)
        high = low + float(rng.uniform(0.1, 5))
        rate = float(rng.uniform(0.1, 20))
        probs = rng.dirichlet(np.ones(int(rng.randint(2, 7)))).astype(np.float32)
        p = float(rng.uniform(0.01, 0.99))
        cases = [(D.Normal(mean, sd), R.Normal(mean, sd), [mean, mean + 3 * sd, float(rng.normal() * 10)]),
                 (D.Uniform(low, high), R.Uniform(low, high), [low, (low + high) / 2, high - 1e-4 * (high - low), high + 1.0, low - 1.0]),
                 (D.Poisson(rate), R.Poisson(rate), [0.0, float(rng.randint(0, 40)), 3.0]),
                 (D.Categorical(probs.tolist()), R.Categorical(probs.tolist()), [0, len(probs) - 1, int(rng.randint(0, len(probs)))]),
                 (D.Bernoulli(p), R.Bernoulli(p), [0.0, 1.0])]
        for mine, ref, values in cases:
            for v in values:
                a = float(mine.log_prob(torch.tensor(float(v)), sum=True))
                b = float(ref.log_prob(torch.tensor(float(v)), sum=True))
                if np.isinf(b) or np.isnan(b):
                    assert (np.isinf(a) and a < 0) or np.isnan(a) == np.isnan(b), (type(ref).__name__, v, a, b)
                else:
                    assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (type(ref).__name__, v, a, b)
    try:
        check()
    finally:
        torch.distributions.Distribution.set_default_validate_args(validate)


@pytest.mark.parametrize('cfg', [dict(lstm_dim=8, K=3, dims=(4, 8), depth=1, seed=101),
                                 dict(lstm_dim=24, K=10, dims=(16, 16), depth=1, seed=102),
                                 dict(lstm_dim=12, K=5, dims=(8, 4), depth=2, seed=103),
                                 dict(lstm_dim=40, K=7, dims=(32, 8), depth=1, seed=104)],
                         ids=lambda c: 'H{lstm_dim}-K{K}-depth{depth}'.format(**c))
def test_oracle_equals_the_live_reference_on_other_network_shapes(cfg):
    """The oracle is pinned on nine recorded cases of fixed shape (tests/golden); here the reference itself runs next to it on
    freshly initialised networks of other shapes - LSTM width, mixture components K (proposal_mixture_components), observe
    embedding dims, stacked LSTM - on a ragged minibatch: `_loss` (inference_network_lstm.py:136-220) and every gradient of
    loss.backward() against O.loss_and_grads on the same state_dict and the same traces."""
    import importlib.util
    from pyprob.nn import Batch
    spec_ = importlib.util.spec_from_file_location('make_golden', os.path.join(HERE, 'golden', 'make_golden.py'))
    G = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(G)
    hip.uninstall()
    pyprob.seed(cfg['seed'])
    model = GaussianWithUnknownMeanMarsaglia()
    emb = {'obs0': {'dim': cfg['dims'][0]}, 'obs1': {'dim': cfg['dims'][1]}}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model.learn_inference_network(num_traces=96, batch_size=48, observe_embeddings=emb, inference_network=InferenceNetwork.LSTM,
                                      lstm_dim=cfg['lstm_dim'], lstm_depth=cfg['depth'], proposal_mixture_components=cfg['K'],
                                      learning_rate_init=1e-3)
    net = model._inference_network
    gen = model._trace_generator(trace_mode=pyprob.TraceMode.PRIOR_FOR_INFERENCE_NETWORK)
    known = set(net._layers_proposal.keys())
    traces = []
    while len(traces) < 40:
        t = next(gen)
        if all(v.address in known for v in t.variables_controlled):
            traces.append(t)
    batch = Batch(traces)
    net.zero_grad()
    ok, loss = net._loss(batch)
    assert ok
    loss.backward()
    arrays, meta = G.dump_batch(traces, ['obs0', 'obs1'])
    params = {k: v.detach().numpy().astype(np.float64) for k, v in net.state_dict().items()}
    onet = O.Net(params, ['obs0', 'obs1'], K=cfg['K'])
    assert onet.depth == cfg['depth']
    out = O.loss_and_grads(onet, arrays, meta['addresses'], meta['dist_names'])
    assert abs(out['loss'] - float(loss)) <= 5e-6 * abs(float(loss)), (out['loss'], float(loss))
    for n, p in net.named_parameters():
        if p.grad is None:
            assert not np.any(out['grads'][n]), n
            continue
        ref = p.grad.numpy()
        # (freshly initialised networks: gradients of ~1e-3 and below whose bias entries are cancelled sums over the rows -
        # the fp32 reference's own round-off is ~1e-3 of a tensor's largest entry there, like for the gumm2 golden)
        err = np.abs(out['grads'][n] - ref).max() / max(np.abs(ref).max(), 1e-6)
        assert err < 3e-3, (n, err)


class _AllFamilies(Model):
    """One controlled variable of every prior family the inference networks have proposal layers for."""

    def __init__(self, categories):
        super().__init__('all families')
        self.categories = categories

    def forward(self):
        from pyprob.distributions import Bernoulli, Categorical, Poisson
        a = pyprob.sample(Normal(0.0, 1.0))
        b = pyprob.sample(Uniform(-1.0, 2.0))
        c = pyprob.sample(Categorical([1.0 / self.categories] * self.categories))
        d = pyprob.sample(Poisson(3.0))
        e = pyprob.sample(Bernoulli(0.4))
        mu = a + b + c.float() + 0.1 * d + e
        pyprob.observe(Normal(mu, 1.0), name='obs0')
        pyprob.observe(Normal(mu, 2.0), name='obs1')
        return mu


@pytest.mark.parametrize('network', ['lstm', 'feedforward'])
def test_parameter_layout_equals_the_reference_modules(network):
    """NetSpec (names, shapes, order of creation, parameter count - the layout of the flat HBM buffer and of checkpoints)
    against the modules the reference builds (inference_network_lstm.py:23-80, inference_network_feedforward.py,
    embedding_feedforward.py, the five proposal layers) for arbitrary hyper-parameters (hypothesis): LSTM width and depth,
    embedding dims and depths, sample / address / distribution-type embedding dims, mixture components, categories."""
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st
    from pyprob.nn import Batch, InferenceNetworkFeedForward, InferenceNetworkLSTM
    from pyprob_amd.spec import NetSpec
    hip.uninstall()

    @settings(derandomize=True, max_examples=12, deadline=None, suppress_health_check=[HealthCheck.too_slow])
    @given(st.sampled_from([4, 10, 33]), st.integers(1, 3), st.sampled_from([3, 10]), st.sampled_from([2, 5]),
           st.sampled_from([(8, 2), (5, 1), (16, 3)]), st.sampled_from([(3, 6, 2), (4, 64, 8)]))
    def check(lstm_dim, lstm_depth, K, categories, emb, dims):
        model = _AllFamilies(categories)
        pyprob.seed(1)
        gen = model._trace_generator(trace_mode=pyprob.TraceMode.PRIOR_FOR_INFERENCE_NETWORK)
        traces = [next(gen) for _ in range(4)]
        obs_emb = {'obs0': {'dim': emb[0], 'depth': emb[1]}, 'obs1': {'dim': 2 * emb[0]}}
        if network == 'lstm':
            net = InferenceNetworkLSTM(model=model, observe_embeddings=obs_emb, lstm_dim=lstm_dim, lstm_depth=lstm_depth,
                                       sample_embedding_dim=dims[0], address_embedding_dim=dims[1],
                                       distribution_type_embedding_dim=dims[2], proposal_mixture_components=K)
        else:
            net = InferenceNetworkFeedForward(model=model, observe_embeddings=obs_emb, proposal_mixture_components=K)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            net._init_layers_observe_embedding(obs_emb, example_trace=traces[0])
            net._init_layers()
            net._layers_initialized = True
            net._polymorph(Batch(traces))
        ref = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        spec = NetSpec({'obs0': {'dim': emb[0], 'depth': emb[1], 'input_dim': 1}, 'obs1': {'dim': 2 * emb[0], 'input_dim': 1}},
                       lstm_dim=lstm_dim, sample_embedding_dim=dims[0], address_embedding_dim=dims[1],
                       distribution_type_embedding_dim=dims[2], proposal_mixture_components=K, network=network,
                       lstm_depth=lstm_depth)
        for v in traces[0].variables_controlled:
            d = v.distribution
            spec.add_address(v.address, d.name, d.num_categories if d.name == 'Categorical' else None)
        mine = {k: tuple(shape) for k, (off, shape) in spec.tensors.items()}
        assert mine == ref, (set(mine) ^ set(ref), [(k, mine[k], ref[k]) for k in mine if k in ref and mine[k] != ref[k]])
        assert spec.num_parameters() == sum(p.numel() for p in net.parameters())
    check()


# ---- the batched paths install() selects by default (pyprob_amd/pyprob_host.py) ------------------------------------------------
@pytest.fixture
def batched(monkeypatch):
    monkeypatch.setenv('PYPROB_HIP_FAST_TRAIN', '1')
    monkeypatch.setenv('PYPROB_HIP_LOCKSTEP', '1')
    hip.install()
    yield
    hip.uninstall()


class GaussianWithUnknownMeanMarsagliaTensorLoop(Model):
    """The rejection-sampling program with a TENSOR condition (`while s >= 1:` - legal pyprob, same semantics one trace at a
    time): the form the lock-step executors can run for all particles together."""

    def __init__(self):
        super().__init__('Gaussian with unknown mean (Marsaglia, tensor loop)')

    def marsaglia(self, mean, stddev):
        uniform = Uniform(-1, 1)
        s = torch.ones(())
        while s >= 1:
            x = pyprob.sample(uniform)
            y = pyprob.sample(uniform)
            s = x * x + y * y
        return mean + stddev * (x * torch.sqrt(-2 * torch.log(s) / s))

    def forward(self):
        mu = self.marsaglia(1, math.sqrt(5))
        likelihood = Normal(mu, math.sqrt(2))
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


def _quiet_learn(model, **kw):
    import contextlib
    import io
    with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
        warnings.simplefilter('ignore')
        kw.setdefault('inference_network', InferenceNetwork.LSTM)
        model.learn_inference_network(observe_embeddings=EMB, lstm_dim=24, learning_rate_init=1e-3, **kw)
    return model._inference_network


def test_online_training_of_a_pyprob_model_takes_the_batched_data_path(batched, tmp_path, monkeypatch):
    """pyprob's OWN Model.learn_inference_network with install(): the OnlineDataset + per-minibatch loop of optimize()
    (nn/dataset.py:50-62, inference_network.py:461-499) are replaced by lock-step prior generation and runs of minibatches;
    the module tree, address strings, layer creation, bookkeeping, optimizer object and checkpoints stay pyprob's."""
    monkeypatch.setattr(torch, 'load', functools.partial(torch.load, weights_only=False))
    pyprob.seed(4)
    model = GaussianWithUnknownMean()
    net = _quiet_learn(model, num_traces=32 * 40, batch_size=32)
    assert net._hip_last_optimize.startswith('batched')
    assert type(net).__name__ == 'InferenceNetworkLSTMHip' and isinstance(net, torch.nn.Module)
    assert net._total_train_iterations == 40 and net._total_train_traces == 32 * 40 and len(net._history_train_loss) == 40
    assert net._history_train_loss_trace == [32 * (i + 1) for i in range(40)]
    hist = np.asarray(net._history_train_loss)
    assert np.isfinite(hist).all() and net._loss_init == hist[0] and net._loss_min == hist.min() and net._loss_previous == hist[-1]
    # the address the batched run created layers for is the one pyprob's own runtime extracts from the program
    pyprob.seed(1)
    stock_trace = next(model._trace_generator(trace_mode=pyprob.TraceMode.PRIOR_FOR_INFERENCE_NETWORK))
    address = stock_trace.variables_controlled[0].address
    assert list(net._layers_proposal.keys()) == [address] and net._layers_proposal[address]._total_train_iterations == 40
    # same parameter set as a stock network of this program (names and shapes): checkpoints interchange
    hip.uninstall()
    stock = _train(GaussianWithUnknownMean, False, 64)._inference_network
    hip.install()
    assert [(n, tuple(p.shape)) for n, p in net.named_parameters()] == [(n, tuple(p.shape)) for n, p in stock.named_parameters()]
    assert net._history_num_params == stock._history_num_params
    assert isinstance(net._optimizer, hip.HipAdam) and int(net._hip_engine.tensor_step.max()) == 40
    # it learns: the loss of the stock network after the same number of traces is the yardstick (different random streams)
    assert abs(hist[-10:].mean() - np.mean(stock._history_train_loss)) < 0.6 and hist[-10:].mean() < hist[:5].mean() + 0.1
    # _save / _load (pyprob's tarball + pickle) and training continues on either loop from the saved optimizer state
    f = str(tmp_path / 'batched.network')
    model.save_inference_network(f)
    clone = GaussianWithUnknownMean()
    clone.load_inference_network(f)
    cn = clone._inference_network
    assert cn._total_train_traces == 32 * 40 and cn._history_train_loss == net._history_train_loss
    for (n0, p0), (n1, p1) in zip(net.named_parameters(), cn.named_parameters()):
        assert n0 == n1 and torch.equal(p0.detach().cpu(), p1.detach().cpu())
    _quiet_learn(clone, num_traces=64, batch_size=32)                          # batched again
    assert cn._total_train_iterations == 42 and cn._hip_last_optimize.startswith('batched')
    assert int(cn._hip_engine.tensor_step.max()) == 42                         # Adam's step count carried through the checkpoint
    monkeypatch.setenv('PYPROB_HIP_FAST_TRAIN', '0')
    _quiet_learn(clone, num_traces=64, batch_size=32)                          # pyprob's own loop on the same network
    assert cn._total_train_iterations == 44 and cn._hip_last_optimize == "pyprob's loop"


def test_a_program_that_reads_sampled_values_keeps_pyprobs_loop(batched):
    """`while float(s) >= 1` (the reference's own Marsaglia program) cannot run with N-wide values: the probe says so, training
    takes pyprob's per-trace loop and importance sampling the particle coroutines - same results as before."""
    pyprob.seed(2)
    model = GaussianWithUnknownMeanMarsaglia()
    net = _quiet_learn(model, num_traces=96, batch_size=32)
    assert net._hip_last_optimize == "pyprob's loop" and net._total_train_iterations == 3
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        post = model.posterior_results(24, IC, observe={'obs0': 4, 'obs1': 5})
    assert type(post).__name__ == 'Empirical' and post.length == 24
    assert pyprob.sample is pyprob.state.sample and pyprob.util.to_tensor.__module__ == 'pyprob.util'      # everything restored


@pytest.mark.parametrize('program', [GaussianWithUnknownMean, GaussianWithUnknownMeanMarsagliaTensorLoop], ids=['gum', 'gumm_tensor_loop'])
def test_posterior_results_of_a_pyprob_model_in_lock_step(batched, program):
    """pyprob's OWN Model.posterior_results with install(): forward() runs once per control-flow path with all particles
    (pyprob.sample / observe forwarded for the call, priors and likelihoods read off pyprob's Distribution objects), and what
    comes back is a pyprob Empirical. Every particle's log-weight is re-scored by the oracle from the values the run drew
    (state.py:203-219, trace.py:123-125; 1e-4)."""
    pyprob.seed(6)
    model = program()
    net = _quiet_learn(model, num_traces=32 * 12, batch_size=32)
    assert net._hip_last_optimize.startswith('batched')
    observe = {'obs0': 8.0, 'obs1': 9.0} if program is GaussianWithUnknownMean else {'obs0': 4.0, 'obs1': 5.0}
    n = 300
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        post = model.posterior_results(n, IC, observe=observe)
    from pyprob.distributions import Empirical
    assert isinstance(post, Empirical) and type(post).__name__ == 'HipEmpirical' and post._hip_executor['executor'] == 'lock step'
    assert post.length == n and len(post) == n and post.name.startswith('Posterior, IC, traces: {:,}'.format(n))
    assert pyprob.sample is pyprob.state.sample and pyprob.observe is pyprob.state.observe and pyprob.util.to_tensor.__module__ == 'pyprob.util'
    values, lw = (t.detach().cpu().double().numpy() for t in post.values_device())
    # the base class's view of the same particles (lists made on first use) and its statistics
    assert len(post.values) == n and abs(float(post.values[3]) - values[3]) < 1e-6
    w = np.exp(lw - lw.max())
    w /= w.sum()
    assert abs(float(post.mean) - float((w * values).sum())) < 1e-5 * max(1.0, abs(float(post.mean)))
    assert abs(float(post.effective_sample_size) - 1.0 / float((w ** 2).sum())) < 1e-3 * float(post.effective_sample_size)
    assert abs(float(post.expectation(lambda x: x)) - float(post.mean)) < 1e-4
    # oracle re-scoring of every particle from what it drew at each statement
    params = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    onet = O.Net(params, ['obs0', 'obs1'], K=10)
    obs = np.array([observe['obs0'], observe['obs1']], np.float64)
    log = post._hip.statement_log
    addresses = list(net._layers_proposal.keys())
    dist_name = 'Normal' if program is GaussianWithUnknownMean else 'Uniform'
    prior_pair = [1.0, math.sqrt(5.0)] if program is GaussianWithUnknownMean else [-1.0, 1.0]
    trace_len, addr_idx, vals, prior = [], [], [], []
    if program is GaussianWithUnknownMean:
        assert post._hip_executor['control_flow_paths'] == 1
        for b in range(n):
            trace_len.append(1), addr_idx.append(0), vals.append(values[b]), prior.append(prior_pair)
        mu = values
    else:
        assert post._hip_executor['control_flow_paths'] > 1
        stmts = [{a: v[0].detach().cpu().double().numpy() for a, v in entry.items()} for entry in log]
        # (a particle that loops deeper than any trace of the short training run meets addresses without proposal layers:
        # the executor draws those from the prior like inference_network_lstm.py:132-134 - left out of the re-scoring)
        keep = []
        for b in range(n):
            k, known, mine = 0, True, []
            while True:
                (ax, xs), (ay, ys) = list(stmts[2 * k].items())[0], list(stmts[2 * k + 1].items())[0]
                known = known and ax in addresses and ay in addresses
                if known:
                    mine.extend([(addresses.index(ax), xs[b]), (addresses.index(ay), ys[b])])
                s_ = float(np.float32(np.float32(xs[b]) * np.float32(xs[b]) + np.float32(ys[b]) * np.float32(ys[b])))
                k += 1
                if s_ < 1.0:
                    break
            if known:
                keep.append(b)
                trace_len.append(2 * k)
                addr_idx.extend(a for a, _ in mine)
                vals.extend(v for _, v in mine)
                prior.extend([prior_pair] * len(mine))
        assert len(keep) > 0.8 * n
        keep = np.asarray(keep)
        mu, lw = values[keep], lw[keep]
    _, _, _, want = O.is_rescore(onet, obs, np.asarray(trace_len), np.asarray(addr_idx), np.asarray(vals, np.float64),
                                 np.asarray(prior, np.float64), addresses, [dist_name] * len(addresses))
    for b in range(len(want)):
        want[b] += sum(float(O.normal_log_prob(y, mu[b], math.sqrt(2.0))) for y in obs)
    np.testing.assert_allclose(lw, want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('network', [InferenceNetwork.LSTM, InferenceNetwork.FEEDFORWARD], ids=['lstm', 'feedforward'])
@pytest.mark.parametrize('program,first', [(_CategoricalThenNormal, 'Categorical'), (_PoissonThenNormal, 'Poisson')], ids=['cat', 'poi'])
def test_other_proposal_layers_in_lock_step_through_pyprob_model(batched, program, first, network):
    """A Categorical / Poisson first statement followed by a Normal whose mean depends on it, through pyprob's OWN
    learn_inference_network (batched online path: the one-hot / Poisson layers are created by pyprob's `_polymorph`) and
    posterior_results (lock step): addresses in pyprob's format, every particle re-scored by the oracle."""
    pyprob.seed(8)
    model = program()
    net = _quiet_learn(model, num_traces=32 * 10, batch_size=32, inference_network=network)
    assert net._hip_last_optimize.startswith('batched')
    assert type(net).__name__ == ('InferenceNetworkLSTMHip' if network == InferenceNetwork.LSTM else 'InferenceNetworkFeedForwardHip')
    addresses = list(net._layers_proposal.keys())
    assert len(addresses) == 2 and first in addresses[0] and 'Normal' in addresses[1]
    observe = {'obs0': 1.0, 'obs1': 1.5}
    n = 200
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        post = model.posterior_results(n, IC, observe=observe)
    assert type(post).__name__ == 'HipEmpirical' and post._hip_executor == dict(executor='lock step', control_flow_paths=1)
    values, lw = (t.detach().cpu().double().numpy() for t in post.values_device())
    log = post._hip.statement_log
    v0 = list(log[0].values())[0][0].detach().cpu().double().numpy()
    v1 = list(log[1].values())[0][0].detach().cpu().double().numpy()
    np.testing.assert_allclose(v1, values)                                   # forward() returns the second draw
    prior = np.zeros((2 * n, 3))
    if first == 'Categorical':
        prior[0::2] = [0.2, 0.3, 0.5]
        prior[1::2, 0], prior[1::2, 1] = v0 * 2.0 - 1.0, 1.5
        sigma = 0.8
    else:
        prior[0::2, 0] = 4.0
        prior[1::2, 0], prior[1::2, 1] = v0 * 0.5, 1.0
        sigma = 0.8
    vals = np.empty(2 * n)
    vals[0::2], vals[1::2] = v0, v1
    params = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    onet = O.Net(params, ['obs0', 'obs1'], K=10)
    obs = np.array([observe['obs0'], observe['obs1']], np.float64)
    rescore_fn = O.is_rescore if network == InferenceNetwork.LSTM else O.is_rescore_feedforward
    _, _, _, want = rescore_fn(onet, obs, np.full(n, 2), np.tile([0, 1], n), vals, prior, addresses, [first, 'Normal'])
    for b in range(n):
        want[b] += sum(float(O.normal_log_prob(y, v1[b], sigma)) for y in obs)
    np.testing.assert_allclose(lw, want, rtol=1e-4, atol=2e-4)


from pyprob import observe as _captured_observe, sample as _captured_sample  # noqa: E402  (what a program may have done)


class _CapturedNames(Model):
    """A program that holds pyprob's functions by name: the forwarding of `pyprob.sample` / `pyprob.observe` cannot reach it."""

    def __init__(self):
        super().__init__('captured names')

    def forward(self):
        mu = _captured_sample(Normal(1, math.sqrt(5)))
        likelihood = Normal(mu, math.sqrt(2))
        _captured_observe(likelihood, name='obs0')
        _captured_observe(likelihood, name='obs1')
        return mu


def test_a_program_that_captured_pyprobs_functions_is_not_batched(batched):
    """pyprob's original `sample` without a current trace returns one bare prior draw (state.py:162-163): a batched run of such a
    program would be silently wrong. The forwarding context makes the original functions fail, the probes reject the program,
    and training / inference take pyprob's own loops (and are right)."""
    pyprob.seed(3)
    model = _CapturedNames()
    net = _quiet_learn(model, num_traces=32 * 6, batch_size=32)
    assert net._hip_last_optimize == "pyprob's loop" and net._total_train_iterations == 6
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        post = model.posterior_results(40, IC, observe={'obs0': 8.0, 'obs1': 9.0})
    assert type(post).__name__ == 'Empirical' and post.length == 40           # coroutines / pyprob's loop, not the lock-step executor
    values = np.array([float(v) for v in post.values])
    assert len(set(np.round(values, 5))) > 30                                  # distinct particles (not one bare prior draw)
    assert pyprob.state._current_trace is None or type(pyprob.state._current_trace).__name__ != '_NoDirectCalls'


def test_offline_training_of_a_pyprob_model_from_packed_shards(batched, tmp_path, monkeypatch):
    """pyprob's OWN Model.save_dataset / learn_inference_network(dataset_dir=, dataset_valid_dir=, pre_generate_layers=True,
    POLY2 schedule, log file, checkpoints) with install(): the dataset is packed shards (no shelve / pickle / zlib decode per
    trace, pyprob/nn/dataset.py:121-205), layers are pre-generated from its address table, minibatches come from the reference's
    sampler over the sorted index and runs of them train inside one C call; module tree, bookkeeping, scheduler object and
    checkpoints stay pyprob's. A shelve dataset written by stock pyprob converts once (hip.convert_dataset) and trains the same way."""
    import glob
    monkeypatch.setattr(torch, 'load', functools.partial(torch.load, weights_only=False))
    pyprob.seed(12)
    model = GaussianWithUnknownMeanMarsaglia()             # (`while float(s) >= 1`: traces generated one forward() at a time)
    d, dv = str(tmp_path / 'train'), str(tmp_path / 'valid')
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        model.save_dataset(d, 256, 128)
        model.save_dataset(dv, 64, 64)
    assert sorted(os.listdir(d)) == ['pyprob_traces_packed_000000_128', 'pyprob_traces_packed_000001_128']
    from pyprob_amd.dataset import PackedTraceDataset
    ds = PackedTraceDataset(d)
    stock_trace = next(model._trace_generator(trace_mode=pyprob.TraceMode.PRIOR_FOR_INFERENCE_NETWORK))
    assert len(ds) == 256 and ds.addresses[0][0] == stock_trace.variables_controlled[0].address      # pyprob's address strings
    log, prefix = str(tmp_path / 'log.csv'), str(tmp_path / 'ckpt')
    net = _quiet_learn(model, num_traces=32 * 8, batch_size=32, dataset_dir=d, dataset_valid_dir=dv, valid_every=64,
                       pre_generate_layers=True, learning_rate_end=1e-5, num_traces_end=1000,
                       learning_rate_scheduler_type=pyprob.LearningRateScheduler.POLY2, log_file_name=log,
                       save_file_name_prefix=prefix, save_every_sec=0)
    assert net._hip_last_optimize == 'batched (pyprob_host.optimize_packed)' and net._layers_pre_generated
    assert set(net._layers_proposal.keys()) == {a for a, _, _ in ds.addresses}            # every address of the dataset, up front
    assert net._total_train_iterations == 8 and net._total_train_traces == 256 and len(net._history_train_loss) == 8
    assert np.isfinite(net._history_train_loss).all() and len(net._history_valid_loss) >= 1
    want_lr = (1e-3 - 1e-5) * (1 - 256 / 1000) ** 2 + 1e-5                                # inference_network.py:357-379 at 256 traces
    assert abs(net._optimizer.param_groups[0]['lr'] - want_lr) < 1e-9 and net._learning_rate_scheduler.last_epoch == 256
    assert len(open(log).read().strip().splitlines()) == 1 + 8
    assert glob.glob(prefix + '_*_pre_generated.network') and len(glob.glob(prefix + '_*_traces_*.network')) >= 2
    # a dataset written by STOCK pyprob (shelve files) is opened by pyprob's own OfflineDataset, and converts once
    hip.uninstall()
    ds_stock = str(tmp_path / 'shelve')
    os.makedirs(ds_stock)
    with contextlib.redirect_stdout(io.StringIO()):
        GaussianWithUnknownMeanMarsaglia().save_dataset(ds_stock, 96, 48)
    hip.install()
    from pyprob.nn import OfflineDataset
    with contextlib.redirect_stdout(io.StringIO()):
        assert isinstance(hip._open_offline_dataset(ds_stock), OfflineDataset)
        packed = str(tmp_path / 'converted')
        assert hip.convert_dataset(ds_stock, packed, num_traces_per_file=64) == 96
    conv = PackedTraceDataset(packed)
    assert len(conv) == 96 and conv.obs_names == ['obs0', 'obs1']
    src = OfflineDataset(ds_stock) if False else None
    model2 = GaussianWithUnknownMeanMarsaglia()
    net2 = _quiet_learn(model2, num_traces=64, batch_size=32, dataset_dir=packed)
    assert net2._hip_last_optimize == 'batched (pyprob_host.optimize_packed)' and net2._total_train_iterations == 2
    assert np.isfinite(net2._history_train_loss).all()
