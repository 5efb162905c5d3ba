"""pyprob as the HOST of the batched executors (seam B2 and the data path of seam B1's `optimize`).

`binding.install()` routes two things of a REAL `pyprob.Model` through the executors this package already runs for its own
mirror host - so that what is measured on the MI355X behind `pyprob_amd.Model` (lock-step importance sampling, device-side
prior generation + `pp_train_resident` runs) is the code a pyprob user gets:

  * `Model._traces` (pyprob/model.py:47-88) with IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK and `map_func=trace_result`
    (`posterior_results`): the user's `forward()` runs ONCE per control-flow path with N-wide device tensors
    (`pyprob_amd.model.Model._traces_lockstep`), `pyprob.sample` / `pyprob.observe` forwarded for the duration of the call
    (pyprob/state.py:118-293) and the prior / likelihood parameters read off pyprob's own `Distribution` objects. A program the
    probe rejects (`float(sampled value)`, `tag`, a family without a device kernel ...) takes the coroutine executor of
    binding.py or pyprob's loop, as before;
  * `InferenceNetwork.optimize` (pyprob/nn/inference_network.py:381-599) on an `OnlineDataset` (pyprob/nn/dataset.py:50-62):
    prior traces are generated in lock step (`VectorisedOnlineDataset`, on the device for single-path programs) and runs of
    minibatches train inside one C call (`pyprob_amd.nn.InferenceNetworkLSTM.optimize`, the loop the mirror host uses) -
    instead of one `forward()` per trace, `Batch.__init__` and one Python iteration per minibatch. The module tree, the
    `_polymorph` growth (new layers are CREATED by pyprob's own code, with its initial values), the bookkeeping attributes and
    the optimizer object stay pyprob's.

Nothing here computes: the arithmetic is the engine's. Importing this module needs pyprob (it is imported by binding.py only).
"""
import contextlib
import os
import warnings
import weakref

import torch

import pyprob
from pyprob import state as _pp_state
from pyprob import util as _pp_util
from pyprob.distributions import Empirical as _PPEmpirical

from . import distributions as D
from . import state as S
from .model import Model as _MirrorModel
from .nn import InferenceNetworkFeedForward as _MirrorFF
from .nn import InferenceNetworkLSTM as _MirrorLSTM


# ---- pyprob Distribution -> the executors' distribution records --------------------------------------------------------------
def _shared(t):
    """torch.distributions broadcasts the parameters (`Normal(mu[N], sqrt(2))` holds scale as a stride-0 view of N elements):
    a shared parameter goes back to its single element - the device kernels take it with stride 0."""
    if torch.is_tensor(t) and t.dim() >= 1 and t.numel() > 1 and all(s == 0 for s in t.stride()):
        return t.reshape(-1)[:1].reshape(())
    return t


def convert(distribution):
    if isinstance(distribution, D.Distribution) or distribution is None:
        return distribution
    name = getattr(distribution, 'name', None)
    with torch._C.DisableTorchFunctionSubclass():      # (metadata only: no per-attribute dispatch of a ParticleTensor)
        if name == 'Normal':
            return D.Normal(_shared(distribution.loc), _shared(distribution.scale))
        if name == 'Uniform':
            return D.Uniform(_shared(distribution.low), _shared(distribution.high))
        if name == 'Poisson':
            return D.Poisson(_shared(distribution.rate))
        if name == 'Categorical':
            return D.Categorical(distribution._probs if hasattr(distribution, '_probs') else distribution.probs)
        if name == 'Bernoulli':
            return D.Bernoulli(_shared(distribution._probs if hasattr(distribution, '_probs') else distribution.probs))
    raise NotImplementedError('no batched executor for distribution {}'.format(name))


# ---- pyprob.sample / pyprob.observe for the duration of a batched call --------------------------------------------------------
def _hip_sample(distribution, control=True, name=None, address=None):
    return S.sample(convert(distribution), name=name, address=address, control=control)


def _hip_observe(distribution, value=None, name=None, address=None):
    return S.observe(convert(distribution), value=value, name=name, address=address)


def _refuse(*args, **kwargs):
    raise NotImplementedError('pyprob.tag / pyprob.factor have no batched executor')


def _to_tensor_in_place(value, dtype=torch.float32):
    """`pyprob.util.to_tensor` (util.py:133-143) for the duration of a batched call: a tensor STAYS WHERE IT IS (the reference moves
    everything to `util._device`: an N-wide per-particle value would be pulled to the host by `Normal(mu, s)` of the user's
    program), a Python number becomes a HOST scalar (no host-to-device copy per distribution object; the kernels take shared
    parameters as cached constants)."""
    if value is None:
        return None
    if torch.is_tensor(value):
        return value if value.dtype == dtype else value.to(dtype=dtype)
    import numpy as np
    if isinstance(value, (np.integer, np.floating)):
        value = float(value)
    return torch.tensor(value, dtype=dtype)


class _NoDirectCalls:
    """`pyprob.state._current_trace` for the duration of a batched call: pyprob's ORIGINAL `sample` / `observe` - reached by a program
    that captured them (`from pyprob import sample`) - would find no current trace and quietly hand out ONE prior draw with no
    weight (state.py:162-163); with this object in place they fail, the probe rejects the program, and it takes the executors
    that run pyprob's own trace runtime."""

    def __getattr__(self, name):
        raise NotImplementedError('pyprob.state.sample / observe were called directly during a batched run (the program captured the '
                                  'functions instead of calling pyprob.sample / pyprob.observe)')


@contextlib.contextmanager
def forwarded(device):
    """`pyprob.sample` / `pyprob.observe` (and the `state` module's names) forward to this package's trace runtime; pyprob's
    `util.to_tensor` leaves tensors on their device (`_to_tensor_in_place`); torch.distributions' argument validation - a device
    synchronisation per object, and a READ of values whose draw may still be deferred - is off. Everything is restored on exit. A program that imported the names
    (`from pyprob import sample`) keeps pyprob's own functions: it then fails the probe and takes the other executors."""
    saved = dict(sample=pyprob.sample, observe=pyprob.observe, tag=getattr(pyprob, 'tag', None), factor=getattr(pyprob, 'factor', None),
                 s_sample=_pp_state.sample, s_observe=_pp_state.observe, to_tensor=_pp_util.to_tensor,
                 skip=S._address_frame_skip, validate=torch.distributions.Distribution._validate_args)
    pyprob.sample, pyprob.observe = _hip_sample, _hip_observe
    _pp_state.sample, _pp_state.observe = _hip_sample, _hip_observe
    if saved['tag'] is not None:
        pyprob.tag = _refuse
    if saved['factor'] is not None:
        pyprob.factor = _refuse
    _pp_util.to_tensor = _to_tensor_in_place
    saved['trace'] = _pp_state._current_trace
    _pp_state._current_trace = _NoDirectCalls()
    S._address_frame_skip = 1
    torch.distributions.Distribution.set_default_validate_args(False)
    try:
        yield
    finally:
        pyprob.sample, pyprob.observe = saved['sample'], saved['observe']
        _pp_state.sample, _pp_state.observe = saved['s_sample'], saved['s_observe']
        if saved['tag'] is not None:
            pyprob.tag = saved['tag']
        if saved['factor'] is not None:
            pyprob.factor = saved['factor']
        _pp_util.to_tensor = saved['to_tensor']
        _pp_state._current_trace = saved['trace']
        S._address_frame_skip = saved['skip']
        torch.distributions.Distribution.set_default_validate_args(saved['validate'])


class ProgramAdapter(_MirrorModel):
    """A pyprob.Model seen by the batched executors: `forward` is the user's, everything else the executors' own."""

    def __init__(self, pp_model):
        super().__init__(getattr(pp_model, 'name', 'pyprob model'))
        try:
            self._pp_ref = weakref.ref(pp_model)          # (the adapter is cached per model: no cycle that keeps the model alive)
        except TypeError:
            self._pp_ref = lambda: pp_model

    def forward(self, *args, **kwargs):
        return self._pp_ref().forward(*args, **kwargs)

    def _lockstep_plan_key(self, *args, **kwargs):
        # a launch-plan replay does not run forward(): its key would have to fingerprint everything the USER's forward() reads
        # through `self._pp` - not analysed for a foreign host, so forward() runs in every call
        return None


def network_view(net):
    """The binding's network (an nn.Module over pyprob's classes) as the executors see a network: the SAME engine and
    importance-sampling runner, no parameters of its own."""
    cls = _MirrorFF if net._hip_kind == 'feedforward' else _MirrorLSTM
    kw = dict(proposal_mixture_components=net._proposal_mixture_components, device=net._hip_device)
    if net._hip_kind == 'lstm':
        kw.update(lstm_dim=net._lstm_dim, lstm_depth=net._lstm_depth, sample_embedding_dim=net._sample_embedding_dim,
                  address_embedding_dim=net._address_embedding_dim, distribution_type_embedding_dim=net._distribution_type_embedding_dim)
    view = cls(model=None, observe_embeddings=dict(net._observe_embeddings), **kw)
    view._engine, view._is = net._hip_engine, net._hip_is
    view._obs_names = list(net._hip_obs_names)
    view._layers_initialized = True
    view._layers_pre_generated = bool(net._layers_pre_generated)
    return view


_ADAPTERS = weakref.WeakKeyDictionary()      # pyprob model -> ProgramAdapter (not an attribute of the model: it must stay picklable)


def adapter_for(pp_model, net):
    """One adapter per pyprob model (the lock-step probe's verdict is cached on it); the network view follows the engine."""
    try:
        ad = _ADAPTERS.get(pp_model)
        if ad is None:
            ad = _ADAPTERS[pp_model] = ProgramAdapter(pp_model)
    except TypeError:                        # (a model class with __slots__ and no weak references: no caching)
        ad = ProgramAdapter(pp_model)
    view = ad.__dict__.get('_inference_network')
    if view is None or view._engine is not net._hip_engine or view._is is not net._hip_is:
        ad._inference_network = network_view(net)
    for address, layer in net._layers_proposal.items():      # per-address counters live on pyprob's layers (lstm.py:198)
        a = net._hip_engine.spec.address_id.get(address)
        if a is not None:
            net._hip_engine.spec.addresses[a].total_train_iterations = layer._total_train_iterations
    return ad


# ---- the Empirical pyprob's posterior() hands back ----------------------------------------------------------------------------
class HipEmpirical(_PPEmpirical):
    """pyprob's `Empirical` (pyprob/distributions/empirical.py) over the device tensors of a batched run. `mean`, `variance`,
    `stddev`, `effective_sample_size` and `length` answer from the float64 statistics the device reduced (the reference's own
    caches `_mean` ... are pre-filled: util.py:398-399, empirical.py:668-678, 758-766); the Python lists behind every other method
    of the base class (`values`, `log_weights`, the Categorical over the weights) are made on first use - a posterior of 10^6
    particles does not pay 10^6 `add()` calls unless somebody iterates it."""

    def __init__(self, device_empirical, name='Empirical'):
        self._hip = device_empirical
        self._hip_values = self._hip_log_weights = self._hip_categorical = None
        super().__init__(name=name)
        self._length = int(device_empirical.length)
        self._finalized = True
        self._mean = _pp_util.to_tensor(float(device_empirical.mean))
        self._variance = _pp_util.to_tensor(float(device_empirical.variance))
        self._effective_sample_size = _pp_util.to_tensor(float(device_empirical.effective_sample_size))
        self.add_metadata(op='finalize', length=self._length)

    # base-class attributes, lazily backed
    @property
    def values(self):
        if self._hip_values is None:
            v = self._hip._values
            self._hip_values = list(v.detach().cpu().unbind(0)) if torch.is_tensor(v) else list(v)
        return self._hip_values

    @values.setter
    def values(self, v):
        if v:                                   # (the base constructor assigns [])
            self._hip_values = v

    @property
    def log_weights(self):
        if self._hip_log_weights is None and self.__dict__.get('_hip') is not None:
            lw = self._hip._log_weights
            lw = lw.detach().cpu() if torch.is_tensor(lw) else torch.as_tensor(lw)
            self._hip_log_weights = list(lw.to(torch.float32).unbind(0))
        return self._hip_log_weights if self._hip_log_weights is not None else []

    @log_weights.setter
    def log_weights(self, v):
        if v:
            self._hip_log_weights = v

    @property
    def _categorical(self):
        if self._hip_categorical is None and self.__dict__.get('_hip') is not None and self.__dict__.get('_finalized'):
            lw = self._hip._log_weights
            lw = lw.detach().cpu() if torch.is_tensor(lw) else torch.as_tensor(lw)
            self._hip_categorical = torch.distributions.Categorical(logits=lw.to(torch.float64))
            self._uniform_weights = bool(torch.eq(lw, lw[0]).all())
        return self._hip_categorical

    @_categorical.setter
    def _categorical(self, v):
        if v is not None:
            self._hip_categorical = v

    @property
    def _uniform_weights(self):
        if '_hip_uniform' not in self.__dict__ and self.__dict__.get('_hip') is not None and self.__dict__.get('_finalized'):
            _ = self._categorical
        return self.__dict__.get('_hip_uniform', False)

    @_uniform_weights.setter
    def _uniform_weights(self, v):
        self.__dict__['_hip_uniform'] = bool(v)

    def values_device(self):
        """The particles' values / log-weights where they were computed (no host copy)."""
        return self._hip._values, self._hip._log_weights


# ---- seam B2: Model._traces in lock step -------------------------------------------------------------------------------------
def traces_lockstep(pp_model, net, num_traces, observe, likelihood_importance, args, kwargs):
    """pyprob/model.py:47-88 for IC + trace_result, all particles together. Returns a finalized pyprob Empirical, or None when the
    program cannot run in lock step (decided once per model by a probe of four particles)."""
    if os.environ.get('PYPROB_HIP_LOCKSTEP', '1') == '0':
        return None
    net._hip_ensure()
    ad = adapter_for(pp_model, net)
    with forwarded(net._hip_engine.device):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            safe = ad._lock_step_safe(observe, *args, **kwargs)
        if not safe:
            return None
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())            # follows torch's global seed (pyprob.seed)
        emp = ad._traces_lockstep(num_traces, observe, seed=seed, likelihood_importance=likelihood_importance, *args, **kwargs)
    out = HipEmpirical(emp)
    out._hip_executor = dict(executor='lock step', control_flow_paths=int(getattr(emp, 'num_paths', 1)))
    return out


# ---- seam B1: optimize() on an OnlineDataset ---------------------------------------------------------------------------------
_BOOKKEEPING = ('_total_train_seconds', '_total_train_traces', '_total_train_traces_end', '_total_train_iterations', '_loss_init',
                '_loss_min', '_loss_max', '_loss_previous', '_history_train_loss', '_history_train_loss_trace', '_history_valid_loss',
                '_history_valid_loss_trace', '_learning_rate_init', '_learning_rate_end', '_weight_decay', '_momentum')


class _OptimizeView(_MirrorLSTM):
    """`pyprob_amd.nn.InferenceNetworkLSTM.optimize` driving the binding's network: the engine is shared, new layers are
    created by pyprob's own `_polymorph` on the nn.Module (its initial values, its `_history_num_params`), the optimizer object
    is re-created where the reference re-creates it (inference_network.py:481-483)."""
    _hip_owner = None

    def _polymorph(self, batch):
        owner = self._hip_owner
        new = [a for a in getattr(batch, 'new_addresses', []) if a[0] not in owner._layers_proposal]
        if not new:
            return False
        changed = owner._polymorph(_example_batch(new))
        if changed:
            owner._create_optimizer()                        # a NEW optimizer object (and empty engine state), as in the reference
            owner._create_lr_scheduler()
        return changed

    def _save(self, file_name):
        self._sync_back()
        self._hip_owner._save(file_name)

    def _sync_back(self):
        owner = self._hip_owner
        for k in _BOOKKEEPING:
            setattr(owner, k, getattr(self, k))
        spec = owner._hip_engine.spec
        for address, layer in owner._layers_proposal.items():
            layer._total_train_iterations = spec.addresses[spec.address_id[address]].total_train_iterations


class _OptimizeViewFF(_OptimizeView, _MirrorFF):
    pass


def _example_batch(new_addresses):
    """What `_polymorph` reads of a minibatch (inference_network_lstm.py:36-41): one example trace per sub-batch with the
    controlled variables' address, distribution (an instance of pyprob's class: it dispatches with isinstance) and value
    shape. new_addresses: [(address, distribution name, number of categories or None)]."""
    from pyprob import distributions as PD
    from pyprob.trace import Variable

    class _T:
        pass
    variables = []
    for address, dname, ncat in new_addresses:
        if dname == 'Normal':
            dist = PD.Normal(0., 1.)
        elif dname == 'Uniform':
            dist = PD.Uniform(0., 1.)
        elif dname == 'Poisson':
            dist = PD.Poisson(1.)
        elif dname == 'Bernoulli':
            dist = PD.Bernoulli(0.5)
        elif dname == 'Categorical':
            dist = PD.Categorical([1.0 / ncat] * int(ncat))
        else:
            raise NotImplementedError(dname)
        variables.append(Variable(distribution=dist, value=_pp_util.to_tensor(0.), address=address, control=True))
    t = _T()
    t.variables_controlled = variables
    b = _T()
    b.sub_batches = [[t]]
    return b


def optimize_online(net, dataset, num_traces, batch_size, learning_rate_init, learning_rate_end, weight_decay, num_traces_end,
                    save_file_name_prefix, save_every_sec, stop_with_bad_loss, optimizer_type, momentum):
    """`InferenceNetwork.optimize` for an OnlineDataset with Adam and no learning-rate schedule, validation set, log file or
    process group (those take pyprob's own loop). Returns False when the program's prior cannot be generated in lock step."""
    from .dataset import VectorisedOnlineDataset
    from pyprob import PriorInflation
    if not net._layers_initialized:                                                          # inference_network.py:382-385
        net._init_layers_observe_embedding(net._observe_embeddings, example_trace=dataset.__getitem__(0))
        net._init_layers()
        net._layers_initialized = True
    obs_names = list(net._layers_observe_embedding.keys())
    ad = adapter_for(dataset._model, net) if net._hip_engine is not None else ProgramAdapter(dataset._model)
    inflation = S.PriorInflation.ENABLED if dataset._prior_inflation == PriorInflation.ENABLED else S.PriorInflation.DISABLED
    with forwarded(net._hip_device):
        try:
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                ad.prior_traces_packed(8, obs_names, prior_inflation=inflation)             # the probe
        except Exception:   # noqa: BLE001 - any failure of the probe means "not lock-step safe"
            return False
        gen_dev = net._hip_device if str(net._hip_device).startswith('cuda') and os.environ.get('PP_PRIOR_DEVICE', '1') != '0' else 'cpu'
        vds = VectorisedOnlineDataset(ad, obs_names, chunk_traces=max(64 * batch_size, 16384), prior_inflation=inflation,
                                      device=gen_dev)
        _run_view_optimize(net, vds, None, num_traces, batch_size, learning_rate_init, learning_rate_end, weight_decay, num_traces_end,
                           save_file_name_prefix, save_every_sec, stop_with_bad_loss, optimizer_type, momentum,
                           pyprob.LearningRateScheduler.NONE, None, None, None)
    return True


def _run_view_optimize(net, dataset, dataset_valid, num_traces, batch_size, learning_rate_init, learning_rate_end, weight_decay,
                       num_traces_end, save_file_name_prefix, save_every_sec, stop_with_bad_loss, optimizer_type, momentum,
                       learning_rate_scheduler_type, valid_every, log_file_name, distributed_num_buckets):
    """The settings the reference fixes at the first call (inference_network.py:438-452), the optimizer / scheduler objects, then
    `pyprob_amd.nn.InferenceNetworkLSTM.optimize` over the binding's engine (`_OptimizeView`), and the bookkeeping back."""
    if net._optimizer_type is None:
        net._optimizer_type = optimizer_type
    if net._momentum is None:
        net._momentum = momentum
    if net._weight_decay is None:
        net._weight_decay = weight_decay
    if net._learning_rate_scheduler_type is None:
        net._learning_rate_scheduler_type = learning_rate_scheduler_type
    if net._learning_rate_init is None:
        net._learning_rate_init = learning_rate_init
    if net._learning_rate_end is None:
        net._learning_rate_end = learning_rate_end
    if net._total_train_traces_end is None:
        net._total_train_traces_end = num_traces_end
    net.train()
    if net._hip_engine is None:
        net._hip_bind()                          # the layers that exist so far (observe embedding, LSTM) into the flat buffer
    if net._optimizer is None:
        net._create_optimizer()
        net._create_lr_scheduler()
    cls = _OptimizeViewFF if net._hip_kind == 'feedforward' else _OptimizeView
    base = network_view(net)
    view = cls.__new__(cls)
    view.__dict__.update(base.__dict__)
    view._hip_owner = net
    for k in _BOOKKEEPING:
        setattr(view, k, getattr(net, k))
    view._optimizer_type = 'ADAM'
    sched = str(net._learning_rate_scheduler_type).split('.')[-1].upper()
    view._learning_rate_scheduler_type = None if sched == 'NONE' else sched
    try:
        view.optimize(num_traces, dataset, batch_size=batch_size, learning_rate_init=net._learning_rate_init,
                      learning_rate_end=net._learning_rate_end, weight_decay=net._weight_decay,
                      num_traces_end=net._total_train_traces_end, stop_with_bad_loss=stop_with_bad_loss,
                      save_file_name_prefix=save_file_name_prefix, save_every_sec=save_every_sec, verbose=False,
                      optimizer_type='ADAM', learning_rate_scheduler_type=view._learning_rate_scheduler_type,
                      dataset_valid=dataset_valid, valid_every=valid_every, log_file_name=log_file_name,
                      distributed_num_buckets=distributed_num_buckets)
    finally:
        view._sync_back()
        if net._learning_rate_scheduler is not None:
            # the reference steps its LambdaLR with the TRACE COUNT every iteration (inference_network.py:567-568): one step to
            # where training stands leaves the scheduler object (and the optimizer's lr) where pyprob's own loop would have
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                net._learning_rate_scheduler.step(net._total_train_traces)
    net._hip_grads_clean = False


# ---- offline datasets (pyprob/nn/dataset.py:121-263, pyprob/model.py:186-232) -------------------------------------------------
def open_offline_dataset(dataset_dir):
    """What `Model.learn_inference_network(dataset_dir=...)` opens: packed shards (pyprob_amd/dataset.py: columnar `.npy` files +
    `meta.json`, written by `save_dataset_packed` / `convert_dataset`) as a `PackedTraceDataset`, anything else - pyprob's shelve
    files - as pyprob's own `OfflineDataset`."""
    from pyprob.nn import OfflineDataset
    from .dataset import PackedTraceDataset
    packed = os.path.isdir(dataset_dir) and any(os.path.exists(os.path.join(dataset_dir, d, 'meta.json')) for d in os.listdir(dataset_dir))
    if packed and os.environ.get('PYPROB_HIP_PACKED_DATASET', '1') != '0':
        return PackedTraceDataset(dataset_dir)
    return OfflineDataset(dataset_dir=dataset_dir)


def save_dataset_packed(pp_model, dataset_dir, num_traces, num_traces_per_file, prior_inflation, args, kwargs):
    """`Model.save_dataset` (pyprob/model.py:227-232) writing packed shards: prior traces generated in lock step when the program
    allows it, else one `forward()` per trace through this package's trace runtime (same address strings either way)."""
    from pyprob import PriorInflation
    from .dataset import save_dataset
    inflation = S.PriorInflation.ENABLED if prior_inflation == PriorInflation.ENABLED else S.PriorInflation.DISABLED
    with forwarded('cpu'):
        return save_dataset(ProgramAdapter(pp_model), dataset_dir, int(num_traces), int(num_traces_per_file), None, *args,
                            prior_inflation=inflation, **kwargs)


def convert_dataset(shelve_dir, packed_dir, num_traces_per_file=100000, obs_names=None):
    """An existing pyprob dataset (shelve files, read with pyprob's own `OfflineDataset`: the slow per-trace decode, once) into
    packed shards. obs_names: the observables to keep, in the order of the network's `observe_embeddings` (default: every named
    variable of the pruned traces, `nn/dataset.py:64-119`, in their order). Returns the number of traces."""
    from pyprob.nn import OfflineDataset
    from .dataset import PackedTraceWriter
    src = OfflineDataset(dataset_dir=shelve_dir)
    os.makedirs(packed_dir, exist_ok=True)
    n, shard, writer, names = len(src), 0, None, (None if obs_names is None else list(obs_names))
    for i in range(n):
        trace = src[i]
        if names is None:
            names = list(trace.named_variables.keys())
        if writer is None:
            m = min(num_traces_per_file, n - i)
            writer = PackedTraceWriter(os.path.join(packed_dir, 'pyprob_traces_packed_{:06d}_{}'.format(shard, m)), names)
            left = m
        writer.add_trace(trace)
        left -= 1
        if left == 0:
            writer.close()
            writer, shard = None, shard + 1
    if writer is not None:
        writer.close()
    return n


def pre_generate_layers_packed(net, dataset, save_file_name_prefix=None):
    """`InferenceNetwork._pre_generate_layers` (inference_network.py:269-288) for a packed dataset: its address table names every
    layer there is to create - no pass over the traces."""
    if not net._layers_initialized:
        net._init_layers_observe_embedding(net._observe_embeddings, example_trace=dataset[0])
        net._init_layers()
        net._layers_initialized = True
    net._layers_pre_generated = True
    new = [a for a in dataset.addresses if a[0] not in net._layers_proposal]
    if new and net._polymorph(_example_batch(new)) and save_file_name_prefix is not None:
        net._save('{}_00000000_pre_generated.network'.format(save_file_name_prefix))


def optimize_packed(net, dataset, dataset_valid, num_traces, batch_size, valid_every, learning_rate_init, learning_rate_end,
                    learning_rate_scheduler_type, weight_decay, num_traces_end, save_file_name_prefix, save_every_sec,
                    stop_with_bad_loss, optimizer_type, momentum, log_file_name, distributed_num_buckets):
    """`InferenceNetwork.optimize` on a packed offline dataset (single process, Adam): the reference's sampler over the sorted
    index, minibatches packed from memory-mapped columns, runs of steps inside one C call; validation loss, POLY1 / POLY2
    schedule, log file and checkpoints as in inference_network.py:381-599."""
    if not net._layers_initialized:
        net._init_layers_observe_embedding(net._observe_embeddings, example_trace=dataset[0])
        net._init_layers()
        net._layers_initialized = True
    _run_view_optimize(net, dataset, dataset_valid, num_traces, batch_size, learning_rate_init, learning_rate_end, weight_decay,
                       num_traces_end, save_file_name_prefix, save_every_sec, stop_with_bad_loss, optimizer_type, momentum,
                       learning_rate_scheduler_type, valid_every, log_file_name, distributed_num_buckets)
    return True
