"""The host side of the training / inference loop WITHOUT a GPU: `Model.learn_inference_network`, `InferenceNetworkLSTM.optimize`
(the mirror of pyprob/nn/inference_network.py:381-599), checkpoints and `posterior_results` driven end to end with the
engine's buffers on the host and the `pyprob_hip::*` operators backed by the oracle (tests/oracle_ops.py) - the code above
the operators is the code that ships; tests/test_gpu_model.py runs the same calls on the device."""

import numpy as np
import pytest
import torch

import oracle_ops
from models import GaussianWithUnknownMean, GaussianWithUnknownMeanMarsaglia, GaussianWithUnknownMeanMarsagliaLockStep
from pyprob_amd.state import InferenceEngine, InferenceNetwork, Optimizer

LSTM = InferenceNetwork.LSTM
IC = InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK
EMB = {'obs0': {'dim': 8}, 'obs1': {'dim': 8}}
KW = dict(inference_network=LSTM, observe_embeddings=EMB, lstm_dim=16, batch_size=16, device='cpu')


def _factory(spec, device='cpu', seed=None):
    eng = oracle_ops.CpuBufferEngine(spec, seed=seed)
    eng._use_ops = True
    return eng


@pytest.fixture(autouse=True)
def host_engine(monkeypatch):
    from pyprob_amd import nn as N
    monkeypatch.setattr(N.InferenceNetworkLSTM, '_engine_factory', staticmethod(_factory))
    monkeypatch.setenv('PP_PYTHON_LOOP', '1')          # (the runs inside one C call need the device)


def test_online_training_creates_layers_and_learns(capsys):
    """Online training of the single-statement program: prior traces generated in lock step, layers created at the first
    minibatch (inference_network_lstm.py:34-80), loss bookkeeping of inference_network.py:497-531."""
    torch.manual_seed(1)
    model = GaussianWithUnknownMean()
    model.learn_inference_network(num_traces=16 * 60, learning_rate_init=1e-2, seed=2, **KW)
    net = model._inference_network
    hist = np.asarray(net._history_train_loss)
    assert net._total_train_iterations == 60 and net._total_train_traces == 16 * 60 and len(hist) == 60
    assert np.isfinite(hist).all() and hist[-10:].mean() < hist[:5].mean() - 0.2
    assert net._loss_init == hist[0] and net._loss_min == hist.min() and net._loss_previous == hist[-1]
    assert [a.dist_name for a in net._engine.spec.addresses] == ['Normal']
    assert net._history_num_params[-1] == net._engine.spec.num_parameters()
    out = capsys.readouterr().out
    assert 'New layers, address' in out and 'Stop condition reached' in out
    assert int(net._engine.tensor_step.max()) == 60


def test_ragged_program_grows_its_network_and_resets_the_optimizer():
    """The rejection-sampling program: new addresses keep appearing; each growth rebuilds the optimizer like
    inference_network.py:481-483 (step counts restart), existing parameters keep their values."""
    torch.manual_seed(3)
    model = GaussianWithUnknownMeanMarsagliaLockStep()
    model.learn_inference_network(num_traces=16 * 30, learning_rate_init=1e-3, seed=4, prior_chunk_traces=64, **KW)
    net = model._inference_network
    n_addr = len(net._engine.spec.addresses)
    assert n_addr >= 4 and len(set(net._history_num_params)) >= 2            # the parameter count grew during training
    assert int(net._engine.tensor_step.max()) < 30                          # ... and Adam was rebuilt when it did
    assert np.isfinite(net._history_train_loss).all()
    for a in net._engine.spec.addresses:
        assert a.total_train_iterations > 0                                  # :198


def test_offline_dataset_schedule_log_validation_and_checkpoints(tmp_path):
    """learn_inference_network(dataset_dir=..., dataset_valid_dir=...): packed shards, the reference's sampler over the
    sorted index, POLY2 learning rate driven by the trace count (:357-379, :568), one log line per iteration (:575-578),
    validation losses (:535-548), periodic and final checkpoints (:550-556, :596-599) that load back."""
    import glob
    torch.manual_seed(5)
    model = GaussianWithUnknownMeanMarsaglia()
    d, dv = str(tmp_path / 'train'), str(tmp_path / 'valid')
    model.save_dataset(d, 320, 160)
    model.save_dataset(dv, 64, 64)
    log, prefix = str(tmp_path / 'log.csv'), str(tmp_path / 'ckpt')
    model.learn_inference_network(num_traces=16 * 20, dataset_dir=d, dataset_valid_dir=dv, valid_every=64, learning_rate_init=1e-3,
                                  learning_rate_end=1e-5, learning_rate_scheduler_type='POLY2', num_traces_end=1000,
                                  log_file_name=log, save_file_name_prefix=prefix, save_every_sec=0, pre_generate_layers=True,
                                  seed=6, **KW)
    net = model._inference_network
    assert net._total_train_iterations == 20 and net._layers_pre_generated
    want = (1e-3 - 1e-5) * (1 - 320 / 1000) ** 2 + 1e-5
    assert abs(net._learning_rate() - want) < 1e-12
    lines = open(log).read().strip().splitlines()
    assert lines[0].startswith('time, iteration, trace, loss') and len(lines) == 21
    lrs = [float(ln.split(',')[4]) for ln in lines[1:]]
    assert lrs[0] > lrs[-1] > want - 1e-12 and all(a >= b for a, b in zip(lrs, lrs[1:]))
    assert len(net._history_valid_loss) >= 3 and np.isfinite(net._history_valid_loss).all()
    files = sorted(glob.glob(prefix + '_*.network'))
    assert any(f.endswith('_00000000_pre_generated.network') for f in files)
    final = [f for f in files if f.endswith('_traces_%d.network' % net._total_train_traces)]
    assert final
    other = GaussianWithUnknownMeanMarsaglia()
    other.load_inference_network(final[-1], device='cpu')
    ln = other._inference_network
    assert torch.equal(ln._engine.params, net._engine.params) and torch.equal(ln._engine.exp_avg, net._engine.exp_avg)
    assert ln._history_train_loss == net._history_train_loss and ln._learning_rate_scheduler_type == 'POLY2'


def test_a_non_finite_minibatch_is_skipped_or_stops_training(tmp_path, capsys):
    """'Cannot compute loss, skipping batch' (inference_network.py:488-492): a NaN observation leaves the parameters and
    the trace counters untouched; with stop_with_bad_loss training returns at that iteration."""
    from pyprob_amd.dataset import PackedTraceDataset
    from helpers import synthetic_gum_arrays
    arr = synthetic_gum_arrays(64, seed=9)
    arr['obs'][:16, 0] = np.nan                       # (sorted by length: a single-statement dataset keeps its order)
    table = [('mu', 'Normal', None)]
    ds = PackedTraceDataset.from_columns(['obs0', 'obs1'], [1, 1], arr['trace_len'], table, arr['addr_idx'], arr['values'],
                                         arr['prior'], arr['obs'])
    for stop in (False, True):
        torch.manual_seed(7)
        model = GaussianWithUnknownMean()
        model.learn_inference_network(num_traces=64, dataset=ds, stop_with_bad_loss=stop, learning_rate_init=1e-3, seed=8, **KW)
        net = model._inference_network
        out = capsys.readouterr().out
        assert 'Cannot compute loss, skipping batch' in out
        assert np.isfinite(net._engine.params.numpy()).all()
        bad = int(np.isnan(ds.gather(np.arange(64))[4]).any(1).sum()) // 16
        assert bad >= 1
        if stop:
            assert net._total_train_iterations < 4
        else:
            assert net._total_train_traces >= 64 and np.isfinite(net._history_train_loss).all()


@pytest.mark.parametrize('opt', ['SGD', 'ADAM_LARC', 'SGD_LARC'])
def test_optimizer_choice_survives_a_checkpoint(opt, tmp_path):
    """optimizer_type / momentum (inference_network.py:439-442) and the optimizer state through save -> load -> continue."""
    torch.manual_seed(11)
    model = GaussianWithUnknownMean()
    kw = dict(learning_rate_init=1e-2, optimizer_type=getattr(Optimizer, opt), momentum=0.8, weight_decay=1e-4, seed=12, **KW)
    model.learn_inference_network(num_traces=16 * 6, **kw)
    net = model._inference_network
    assert net._engine.optimizer == dict(kind='sgd' if opt.startswith('SGD') else 'adam', larc=opt.endswith('LARC'), momentum=0.8)
    fn = str(tmp_path / 'n.network')
    model.save_inference_network(fn)
    other = GaussianWithUnknownMean()
    other.load_inference_network(fn, device='cpu')
    ln = other._inference_network
    assert ln._optimizer_type == opt and ln._momentum == 0.8 and ln._engine.optimizer == net._engine.optimizer
    assert torch.equal(ln._engine.exp_avg, net._engine.exp_avg) and float(ln._engine.exp_avg.abs().max()) > 0
    before = ln._engine.exp_avg.clone()
    other.learn_inference_network(num_traces=16 * 2, **kw)
    assert ln._total_train_iterations == 8 and not torch.equal(before, ln._engine.exp_avg)
    with pytest.raises(ValueError):
        GaussianWithUnknownMean().learn_inference_network(num_traces=16, optimizer_type='LBFGS', **KW)


def test_posterior_after_training_in_lock_step_and_in_coroutines():
    """posterior_results with the freshly trained network: all particles in lock step (tensor program) and as coroutines
    (program as written); both give finite weights and a mean near the analytic posterior of the model."""
    torch.manual_seed(13)
    model = GaussianWithUnknownMean()
    model.learn_inference_network(num_traces=16 * 150, learning_rate_init=1e-2, seed=14, **KW)
    obs = {'obs0': 4.0, 'obs1': 5.0}
    exact = (1 / 5 + 9 / 2) / (1 / 5 + 2 / 2)
    lock = model.posterior_results(400, IC, observe=obs, lock_step=True, seed=15)
    assert np.isfinite(lock.log_weights).all() and lock.effective_sample_size > 4 and abs(lock.mean - exact) < 1.0
    co = model.posterior_results(200, IC, observe=obs, lock_step=False, seed=16)
    assert np.isfinite(co.log_weights).all() and co.effective_sample_size > 3 and abs(co.mean - exact) < 1.2


@pytest.mark.parametrize('grow', [False, True])
def test_run_planner_equals_the_per_step_loop(tmp_path, monkeypatch, grow):
    """nn.optimize's run planner (runs of up to 64 steps handed to ONE call of pp_train_steps; here the call restated on the
    host, oracle_ops.CpuBufferEngine.train_run) against the per-step loop on the same offline dataset: same minibatches,
    same POLY2 learning rates, same bookkeeping, bit-identical parameters - with layers pre-generated (one run per epoch
    tail) and with the network growing while it trains (a run ends before the minibatch that brings a new address, and
    Adam restarts there like inference_network.py:481-483)."""
    d = str(tmp_path / 'train')
    torch.manual_seed(5)
    GaussianWithUnknownMeanMarsaglia().save_dataset(d, 320, 160)
    out = {}
    for loop in ('python', 'planner'):
        monkeypatch.setenv('PP_PYTHON_LOOP', '1' if loop == 'python' else '0')
        torch.manual_seed(7)
        model = GaussianWithUnknownMeanMarsaglia()
        model.learn_inference_network(num_traces=16 * 45, dataset_dir=d, learning_rate_init=1e-3, learning_rate_end=1e-5,
                                      learning_rate_scheduler_type='POLY2', num_traces_end=2000, pre_generate_layers=not grow,
                                      seed=6, **KW)
        net = model._inference_network
        out[loop] = (net._engine.params.clone(), list(net._history_train_loss), net._total_train_iterations,
                     net._total_train_traces, [a.total_train_iterations for a in net._engine.spec.addresses],
                     net._engine.tensor_step.clone(), list(getattr(net._engine, 'run_lengths', [])))
    p, q = out['python'], out['planner']
    assert q[6] and max(q[6]) > 1 and sum(q[6]) == 45 and not p[6]          # the planner really handed out runs
    assert p[2] == q[2] == 45 and p[3] == q[3] == 16 * 45 and p[4] == q[4]
    assert torch.equal(p[5], q[5])
    np.testing.assert_allclose(p[1], q[1], rtol=1e-6)
    assert torch.equal(p[0], q[0])
