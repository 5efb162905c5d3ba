"""The reference-side binding: pyprob's OWN `Model.learn_inference_network()` / `posterior_results()` running on the HIP
engine (seams B1 and B2). Importing this module needs `pyprob` itself (it subclasses its classes).

    import pyprob, pyprob_amd.binding as hip
    hip.install()                        # Model.learn_inference_network now builds the HIP-backed networks
    model.learn_inference_network(..., inference_network=pyprob.InferenceNetwork.LSTM)     # unchanged user code
    model.posterior_results(1000, pyprob.InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK, observe=...)

B1 - `InferenceNetworkLSTMHip(pyprob.nn.InferenceNetworkLSTM)` / `InferenceNetworkFeedForwardHip`: the reference's
module tree, `state_dict` names, `_polymorph`, `optimize()` loop, `_save` / `_load` stay as they are; what changes is
where the arithmetic runs:
  * every `nn.Parameter` is re-bound (`.data`) to a view of ONE flat HBM buffer owned by an `ICEngine`; new layers
    created by `_polymorph` (pyprob/nn/inference_network_lstm.py:34-80) are appended to it with the reference's own
    initial values;
  * `_loss(batch)` (inference_network_lstm.py:136-220) packs the minibatch (pyprob_amd/packed.py) and calls the
    `pyprob_hip::ic_loss` operator (forward + backward in one C call) through `HipLoss` (pyprob_amd/autograd.py), an `autograd.Function` whose
    inputs are exactly the parameters that take part in the minibatch: `loss.backward()` hands each of them its view of
    the flat gradient buffer and leaves `grad = None` on every other parameter, like the reference's autograd does
    (the presence map of `_distributed_sync_grad`, inference_network.py:300-315, keeps working);
  * `_create_optimizer` (inference_network.py:343-355) returns `HipAdam`, a `torch.optim.Optimizer` whose `step()` is
    the `pyprob_hip::adam_step` operator over the flat buffers (per-tensor step counts and `grad is None` skipping as
    in torch.optim.Adam); LambdaLR schedulers and `state_dict()` work on it;
  * `_distributed_sync_grad` is ONE all-reduce of `[gradients | presence map]`;
  * `_infer_init` / `_infer_step` (inference_network.py:141-148, inference_network_lstm.py:82-134) call
    `pyprob_hip::is_init` / `is_step`; `_infer_step` returns an object with the `.sample()` / `.log_prob(value, sum=True)`
    protocol `state.sample` uses (state.py:207-212).
B2 - `install()` wraps `Model._traces` (pyprob/model.py:47-88): importance sampling with a HIP-backed network runs the
particles as coroutines over the reference's own trace runtime - they park inside `_infer_step` and are served in
address-grouped batches (pyprob_amd/coroutine.py) - and returns the same `Empirical` the reference builds.

Everything else of pyprob (state, Trace, distributions, Empirical, datasets, the optimize loop) is the reference's code.
"""
import os
import warnings

import torch

import pyprob  # noqa: F401
from pyprob import util as _util
from pyprob.nn import InferenceNetworkFeedForward as _RefFeedForward
from pyprob.nn import InferenceNetworkLSTM as _RefLSTM

from .autograd import HipAdam, HipLoss, HipSGD  # noqa: F401  (the binding's public names)
from .coroutine import ParticleScheduler
from .hip_network import _PROPOSAL_DIST, _HipNetworkMixin  # noqa: F401
from .nn import ProposalSample


class _FastOptimize:
    """`optimize` (inference_network.py:381-599) of the HIP-backed classes: online training with Adam goes through the batched
    data path of pyprob_amd/pyprob_host.py (prior traces generated in lock step, runs of minibatches inside one C call) when
    the program allows it; everything else - offline datasets, validation, schedulers, SGD / LARC, a log file, a process group,
    `PYPROB_HIP_FAST_TRAIN=0` - is the reference's own loop over `_loss` / `HipAdam`, unchanged."""

    def optimize(self, num_traces, dataset, dataset_valid=None, num_traces_end=1e9, batch_size=64, valid_every=None,
                 optimizer_type=None, learning_rate_init=0.0001, learning_rate_end=1e-6, learning_rate_scheduler_type=None,
                 momentum=0.9, weight_decay=1e-5, save_file_name_prefix=None, save_every_sec=600, distributed_backend=None,
                 distributed_params_sync_every_iter=10000, distributed_num_buckets=10, dataloader_offline_num_workers=0,
                 stop_with_bad_loss=False, log_file_name=None):
        from pyprob import LearningRateScheduler, Optimizer
        from pyprob.nn import OnlineDataset
        optimizer_type = Optimizer.ADAM if optimizer_type is None else optimizer_type
        learning_rate_scheduler_type = LearningRateScheduler.NONE if learning_rate_scheduler_type is None else learning_rate_scheduler_type
        fast = (type(dataset) is OnlineDataset and dataset_valid is None and distributed_backend is None and log_file_name is None
                and (self._optimizer_type or optimizer_type) == Optimizer.ADAM
                and (self._learning_rate_scheduler_type or learning_rate_scheduler_type) == LearningRateScheduler.NONE
                and not self._layers_pre_generated and os.environ.get('PYPROB_HIP_FAST_TRAIN', '1') != '0')
        if fast:
            from . import pyprob_host
            if pyprob_host.optimize_online(self, dataset, num_traces, batch_size, learning_rate_init, learning_rate_end, weight_decay,
                                           num_traces_end, save_file_name_prefix, save_every_sec, stop_with_bad_loss,
                                           optimizer_type, momentum):
                self._hip_last_optimize = 'batched (pyprob_host.optimize_online)'
                return
        from .dataset import PackedTraceDataset
        if isinstance(dataset, PackedTraceDataset):
            # an offline dataset of packed shards (install() makes learn_inference_network(dataset_dir=...) open them): the
            # reference's DataLoader + Batch route cannot read it - single process, Adam; anything else is an error, not a fallback
            if distributed_backend is not None or (self._optimizer_type or optimizer_type) != Optimizer.ADAM or \
                    not (dataset_valid is None or isinstance(dataset_valid, PackedTraceDataset)):
                raise NotImplementedError('packed offline datasets train in one process with Optimizer.ADAM (and a packed validation '
                                          'set); use pyprob.nn.OfflineDataset files (PYPROB_HIP_PACKED_DATASET=0) for the other paths')
            from . import pyprob_host
            pyprob_host.optimize_packed(self, dataset, dataset_valid, num_traces, batch_size, valid_every, learning_rate_init,
                                        learning_rate_end, learning_rate_scheduler_type, weight_decay, num_traces_end,
                                        save_file_name_prefix, save_every_sec, stop_with_bad_loss, optimizer_type, momentum,
                                        log_file_name, distributed_num_buckets)
            self._hip_last_optimize = 'batched (pyprob_host.optimize_packed)'
            return
        self._hip_last_optimize = "pyprob's loop"
        return super().optimize(num_traces=num_traces, dataset=dataset, dataset_valid=dataset_valid, num_traces_end=num_traces_end,
                                batch_size=batch_size, valid_every=valid_every, optimizer_type=optimizer_type,
                                learning_rate_init=learning_rate_init, learning_rate_end=learning_rate_end,
                                learning_rate_scheduler_type=learning_rate_scheduler_type, momentum=momentum,
                                weight_decay=weight_decay, save_file_name_prefix=save_file_name_prefix,
                                save_every_sec=save_every_sec, distributed_backend=distributed_backend,
                                distributed_params_sync_every_iter=distributed_params_sync_every_iter,
                                distributed_num_buckets=distributed_num_buckets,
                                dataloader_offline_num_workers=dataloader_offline_num_workers,
                                stop_with_bad_loss=stop_with_bad_loss, log_file_name=log_file_name)


    def _pre_generate_layers(self, dataset, batch_size=64, save_file_name_prefix=None):
        from .dataset import PackedTraceDataset
        if isinstance(dataset, PackedTraceDataset):       # inference_network.py:269-288 from the dataset's address table
            from . import pyprob_host
            return pyprob_host.pre_generate_layers_packed(self, dataset, save_file_name_prefix)
        return super()._pre_generate_layers(dataset, batch_size=batch_size, save_file_name_prefix=save_file_name_prefix)


class InferenceNetworkLSTMHip(_FastOptimize, _HipNetworkMixin, _RefLSTM):
    _hip_kind = 'lstm'

    def _init_layers(self):
        super()._init_layers()          # nn.LSTM with the reference's initialisation; bound at the first _polymorph


class InferenceNetworkFeedForwardHip(_FastOptimize, _HipNetworkMixin, _RefFeedForward):
    _hip_kind = 'feedforward'


# ---- B2: Model._traces with particle coroutines ------------------------------------------------------------------------
class _InferStepScheduler(ParticleScheduler):
    """Particles are the reference's own `forward()` runs over the reference's own trace runtime; they park inside
    `InferenceNetwork._infer_step` and receive (value, log q); log p, the weight and Trace.end stay pyprob's code."""

    def __init__(self, network, num_traces, seed):
        super().__init__(network._hip_is, network._hip_engine.spec, num_traces, seed, 0)
        self.network = network

    def infer_step(self, a, prev_a, distribution, prev_variable):
        from pyprob import state
        if prev_variable is not None:
            # the previous variable's value as the trace holds it (it may have been drawn from the prior on the host when
            # its address had no proposal layers): written into the device column before the group is stepped
            self.current.prev_host_value = float(prev_variable.value)
        ctx = (state._current_trace, state._current_trace_previous_variable, state._current_trace_execution_start)
        value, logq = self.park(a, prev_a, distribution)
        state._current_trace, state._current_trace_previous_variable, state._current_trace_execution_start = ctx
        return ProposalSample(value, logq)


_original_traces = None


def _traces_with_coroutines(self, num_traces=10, trace_mode=None, prior_inflation=None, inference_engine=None,
                            inference_network=None, map_func=None, silent=False, observe=None, file_name=None,
                            likelihood_importance=1., *args, **kwargs):
    import pyprob.model as M
    from pyprob import InferenceEngine, PriorInflation, TraceMode, state
    from pyprob.distributions import Empirical
    trace_mode = TraceMode.PRIOR if trace_mode is None else trace_mode
    prior_inflation = PriorInflation.DISABLED if prior_inflation is None else prior_inflation
    inference_engine = InferenceEngine.IMPORTANCE_SAMPLING if inference_engine is None else inference_engine
    fast = (inference_engine == InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK
            and isinstance(inference_network, _HipNetworkMixin) and num_traces > 1 and _have_greenlet()
            and os.environ.get('PYPROB_HIP_COROUTINES', '1') != '0')
    batched = (inference_engine == InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK
               and isinstance(inference_network, _HipNetworkMixin) and num_traces > 1 and file_name is None
               and trace_mode == TraceMode.POSTERIOR and prior_inflation == PriorInflation.DISABLED
               and map_func is getattr(M, 'trace_result', None))
    if batched:
        # posterior_results: only forward()'s return value of every particle is asked for - all particles in lock step when the
        # program allows it (pyprob_host.traces_lockstep; None = it does not)
        from . import pyprob_host
        emp = pyprob_host.traces_lockstep(self, inference_network, num_traces, observe, likelihood_importance, args, kwargs)
        if emp is not None:
            return emp
    if not fast:
        return _original_traces(self, num_traces=num_traces, trace_mode=trace_mode, prior_inflation=prior_inflation,
                                inference_engine=inference_engine, inference_network=inference_network, map_func=map_func,
                                silent=silent, observe=observe, file_name=file_name,
                                likelihood_importance=likelihood_importance, *args, **kwargs)
    # model.py:39-45 (_trace_generator) for N interleaved particles instead of one after the other
    state._init_traces(func=self.forward, trace_mode=trace_mode, prior_inflation=prior_inflation,
                       inference_engine=inference_engine, inference_network=inference_network, observe=observe,
                       address_dictionary=self._address_dictionary, likelihood_importance=likelihood_importance)
    sched = _InferStepScheduler(inference_network, num_traces, seed=int(torch.randint(0, 2 ** 31, (1,)).item()))
    inference_network._hip_scheduler = sched

    def particle_main(p):
        state._begin_trace()
        result = self.forward(*args, **kwargs)
        p.trace = state._end_trace(result)
        p.done = True
    try:
        particles = sched.run_particles(particle_main)
    finally:
        inference_network._hip_scheduler = None
        state._current_trace = None
    # model.py:48-88: the Empirical of map_func(trace) with the traces' log-weights; non-finite traces are discarded
    traces = Empirical(file_name=file_name)
    map_func = (lambda t: t) if map_func is None else map_func
    for p in particles:
        log_weight = p.trace.log_importance_weight
        if _util.has_nan_or_inf(log_weight):
            warnings.warn('Encountered trace with nan, inf, or -inf log_weight. Discarding trace.')
        else:
            traces.add(map_func(p.trace), log_weight)
    traces.finalize()
    traces._hip_coroutine_stats = dict(rounds=sched.rounds, group_calls=sched.group_calls, statements=sched.statements,
                                       seconds=sched.seconds)
    return traces


_original_offline = None
_original_save_dataset = None


def _open_offline_dataset(dataset_dir):
    from . import pyprob_host
    return pyprob_host.open_offline_dataset(dataset_dir)


def _save_dataset(self, dataset_dir, num_traces, num_traces_per_file, prior_inflation=None, *args, **kwargs):
    from pyprob import PriorInflation
    prior_inflation = PriorInflation.DISABLED if prior_inflation is None else prior_inflation
    if os.environ.get('PYPROB_HIP_PACKED_DATASET', '1') == '0':
        return _original_save_dataset(self, dataset_dir, num_traces, num_traces_per_file, prior_inflation, *args, **kwargs)
    from . import pyprob_host
    return pyprob_host.save_dataset_packed(self, dataset_dir, num_traces, num_traces_per_file, prior_inflation, args, kwargs)


def convert_dataset(shelve_dir, packed_dir, num_traces_per_file=100000, obs_names=None):
    """An existing pyprob dataset directory (shelve files) -> packed shards `learn_inference_network(dataset_dir=packed_dir)` trains
    from at full speed (pyprob_host.convert_dataset)."""
    from . import pyprob_host
    return pyprob_host.convert_dataset(shelve_dir, packed_dir, num_traces_per_file, obs_names)


def _have_greenlet():
    try:
        import greenlet  # noqa: F401
        return True
    except ImportError:
        return False


def install():
    """Make pyprob build HIP-backed inference networks and serve importance sampling in batches: replaces the two
    classes `Model.learn_inference_network` instantiates (pyprob/model.py:199-204) and wraps `Model._traces`. Networks
    saved afterwards unpickle as the HIP classes (they must be importable: `import pyprob_amd.binding`)."""
    global _original_traces, _original_save_dataset, _original_offline
    import pyprob.model as M
    M.InferenceNetworkLSTM = InferenceNetworkLSTMHip
    M.InferenceNetworkFeedForward = InferenceNetworkFeedForwardHip
    if _original_traces is None:
        _original_traces = M.Model._traces
        M.Model._traces = _traces_with_coroutines
    if _original_offline is None:
        # offline datasets (pyprob/model.py:186-195, 227-232): `save_dataset` writes packed shards and `dataset_dir` opens them
        # (pyprob's shelve files are still opened by pyprob's own OfflineDataset; PYPROB_HIP_PACKED_DATASET=0 writes them too)
        _original_offline, _original_save_dataset = M.OfflineDataset, M.Model.save_dataset
        M.OfflineDataset = _open_offline_dataset
        M.Model.save_dataset = _save_dataset


def uninstall():
    global _original_traces, _original_save_dataset, _original_offline
    import pyprob.model as M
    M.InferenceNetworkLSTM = _RefLSTM
    M.InferenceNetworkFeedForward = _RefFeedForward
    if _original_traces is not None:
        M.Model._traces = _original_traces
        _original_traces = None
    if _original_offline is not None:
        M.OfflineDataset, M.Model.save_dataset = _original_offline, _original_save_dataset
        _original_offline = _original_save_dataset = None
