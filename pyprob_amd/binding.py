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

import pyprob
from pyprob import util as _util
from pyprob.nn import InferenceNetworkFeedForward as _RefFeedForward
from pyprob.nn import InferenceNetworkLSTM as _RefLSTM

from . import lib as L
from .autograd import HipAdam, HipLoss, HipSGD, presence
from .coroutine import ParticleScheduler
from .is_engine import ISRunner
from .nn import ProposalSample
from .ops import ops  # noqa: F401  (importing registers the pyprob_hip operators)
from .packed import pack_traces
from .spec import NetSpec

_PROPOSAL_DIST = {'ProposalNormalNormalMixture': 'Normal', 'ProposalUniformTruncatedNormalMixture': 'Uniform',
                  'ProposalCategoricalCategorical': 'Categorical', 'ProposalPoissonTruncatedNormalMixture': 'Poisson',
                  'ProposalBernoulliBernoulli': 'Bernoulli'}


def _default_engine_factory(spec, device):
    from .engine import ICEngine
    return ICEngine(spec, device=device)


class _HipNetworkMixin:
    """The HIP side of an `InferenceNetwork` subclass. State (not pickled): `_hip_engine` (flat buffers + network
    description), `_hip_is` (importance-sampling runner)."""
    _hip_kind = 'lstm'
    _hip_device = os.environ.get('PYPROB_HIP_DEVICE', 'cuda:0')
    _hip_engine_factory = staticmethod(_default_engine_factory)
    _hip_engine = None
    _hip_is = None
    _hip_scheduler = None
    _hip_grads_clean = False
    _hip_grad_scale = 1.0
    _hip_status = None
    _hip_sync_status = True      # `_loss` reads the non-finite flag back (the reference's has_nan_or_inf sync, :202-217)

    # ---- binding the module tree to the flat buffer ------------------------------------------------------------------
    def _hip_named_parameters(self):
        return list(self.named_parameters())

    def _hip_presence(self, bring_home=False):
        return presence(self, bring_home)

    def _hip_obs_spec(self):
        obs = {}
        for name, layer in self._layers_observe_embedding.items():
            layers = getattr(layer, '_layers', None)
            if type(layer).__name__ != 'EmbeddingFeedForward' or layers is None or not 1 <= len(layers) <= L.PP_MAX_OBS_DEPTH:
                raise NotImplementedError('the HIP engine embeds observations with ObserveEmbedding.FEEDFORWARD of depth '
                                          '1..{} (observable {}: {})'.format(L.PP_MAX_OBS_DEPTH, name, type(layer).__name__))
            obs[name] = dict(input_dim=int(layer._input_dim), dim=int(layer._output_dim), depth=len(layers))
        return obs

    def _hip_address_items(self):
        items = []
        for address, layer in self._layers_proposal.items():
            dist = _PROPOSAL_DIST.get(type(layer).__name__)
            if dist is None:
                raise NotImplementedError('no HIP proposal head for {}'.format(type(layer).__name__))
            ncat = int(layer._ff._layers[-1].out_features) if dist == 'Categorical' else None
            items.append((address, dist, ncat))
        return items

    def _hip_bind(self):
        """(Re)build the engine for the module tree as it is now and re-bind every parameter to its flat view. Called
        when the layers were created by the reference's code: after `_init_layers`, after `_polymorph`, after `_load`."""
        if not self._layers_initialized and self._layers_observe_embedding_final is None:
            raise RuntimeError('inference network layers are not initialised yet')
        named = self._hip_named_parameters()
        if self._hip_engine is None:
            kw = dict(proposal_mixture_components=self._proposal_mixture_components, network=self._hip_kind)
            if self._hip_kind == 'lstm':
                kw.update(lstm_dim=self._lstm_dim, lstm_depth=self._lstm_depth, sample_embedding_dim=self._sample_embedding_dim,
                          address_embedding_dim=self._address_embedding_dim,
                          distribution_type_embedding_dim=self._distribution_type_embedding_dim)
            spec = NetSpec(self._hip_obs_spec(), **kw)
            for address, dist, ncat in self._hip_address_items():
                spec.add_address(address, dist, ncat)
            self._hip_engine = type(self)._hip_engine_factory(spec, self._hip_device)
            fresh = [n for n, _ in named]
        else:
            known = set(self._hip_engine.spec.tensors.keys())
            self._hip_engine.add_addresses([it for it in self._hip_address_items()
                                            if it[0] not in self._hip_engine.spec.address_id])
            fresh = [n for n, _ in named if n not in known]
        eng = self._hip_engine
        if set(eng.spec.tensors.keys()) != set(n for n, _ in named):
            raise RuntimeError('HIP binding: parameter sets differ: {}'.format(
                sorted(set(eng.spec.tensors.keys()) ^ set(n for n, _ in named))[:4]))
        with torch.no_grad():
            for name, p in named:
                flat = eng.tensor(name)
                if tuple(flat.shape) != tuple(p.shape):
                    raise RuntimeError('HIP binding: shape of {} is {}, the engine expects {}'.format(
                        name, tuple(p.shape), tuple(flat.shape)))
                if name in fresh:
                    flat.copy_(p.data.to(flat.device))      # the reference's own initial values (or a loaded checkpoint)
                p.data = flat
                p.grad = None
        self._hip_is = ISRunner(eng)
        self._hip_grads_clean = False
        self._hip_obs_names = list(self._layers_observe_embedding.keys())
        for address, layer in self._layers_proposal.items():      # per-address counters live on the reference's layers
            eng.spec.addresses[eng.spec.address_id[address]].total_train_iterations = layer._total_train_iterations

    def _hip_ensure(self):
        if self._hip_engine is None:       # first use, or the module was unpickled (the engine is not part of the pickle)
            self._hip_bind()

    # ---- the reference's hooks -------------------------------------------------------------------------------------
    def to(self, device=None, *args, **kwargs):
        """The parameters live in the engine's HBM buffer whatever `util._device` says (model.py:214 calls .to())."""
        self._device = torch.device(self._hip_device)
        self._on_cuda = 'cuda' in str(self._hip_device)
        return self

    def _polymorph(self, batch):
        changed = super()._polymorph(batch)
        if changed or self._hip_engine is None:
            self._hip_bind()
        return changed

    def _create_optimizer(self, state_dict=None):
        if self._optimizer_type is None:           # happens when loading a pre-generated network (:344-345)
            return
        self._hip_ensure()
        self._hip_engine.reset_optimizer()         # a NEW optimizer: state is lost like in the reference (:481-483)
        larc = self._optimizer_type in (pyprob.Optimizer.ADAM_LARC, pyprob.Optimizer.SGD_LARC)    # :351-352
        if self._optimizer_type in (pyprob.Optimizer.ADAM, pyprob.Optimizer.ADAM_LARC):           # :347-348
            self._optimizer = HipAdam(self, lr=self._learning_rate_init, weight_decay=self._weight_decay, larc=larc)
        else:                                                                                      # :349-350
            self._optimizer = HipSGD(self, lr=self._learning_rate_init, momentum=self._momentum, weight_decay=self._weight_decay,
                                     nesterov=True, larc=larc)
        if state_dict is not None:
            self._optimizer.load_state_dict(state_dict)

    def _loss(self, batch):
        self._hip_ensure()
        eng = self._hip_engine
        spec = eng.spec
        for sub_batch in batch.sub_batches:
            for variable in sub_batch[0].variables_controlled:
                if variable.address not in spec.address_id:
                    print('Address unknown by inference network: {}'.format(variable.address))
                    return False, 0                                               # :150-152, :164-166
        packed = pack_traces(batch.traces, spec, self._hip_obs_names)
        for sub_batch in batch.sub_batches:                                       # :198, once per (sub-batch, time step)
            for variable in sub_batch[0].variables_controlled:
                self._layers_proposal[variable.address]._total_train_iterations += 1
        act = spec.active_mask(packed.cur_counts, packed.prev_counts)
        named = self._hip_named_parameters()
        index = {n: i for i, n in enumerate(spec.tensors.keys())}
        taking_part = [(n, p) for n, p in named if act[index[n]] > 0]
        loss = HipLoss.apply(self, packed, [n for n, _ in taking_part], *[p for _, p in taking_part])
        if self._hip_sync_status and int(self._hip_status.item()) != 0:
            print('Nan or Inf present in proposal log_prob.')
            return False, 0                                                       # :214-217
        return True, loss

    def _distributed_sync_grad(self, world_size):
        """inference_network.py:296-325 as ONE all-reduce of [flat gradients | presence map]; the division by the world
        size happens inside the optimizer kernel (grad_scale)."""
        import torch.distributed as dist
        eng = self._hip_engine
        present = self._hip_presence(bring_home=True)
        eng.active.copy_(torch.tensor(present, dtype=torch.float32).to(eng.device))
        eng._active_key = None
        eng.loss_buf.zero_()
        dist.all_reduce(eng.grads_full)
        merged = eng.active.cpu()
        index = {n: i for i, n in enumerate(eng.spec.tensors.keys())}
        for name, p in self._hip_named_parameters():
            if merged[index[name]] > 0 and p.grad is None:   # someone else had a gradient: a (zero) local one joins the update
                p.grad = eng.tensor(name, eng.grads)
        self._hip_grad_scale = 1.0 / float(world_size)

    def _distributed_update_train_loss(self, loss, world_size):
        """inference_network.py:327-333 with the scalar on the engine's device (an nccl group cannot reduce CPU tensors)."""
        import torch.distributed as dist
        t = torch.tensor([float(loss)], dtype=torch.float32).to(self._hip_engine.device)
        dist.all_reduce(t)
        self._distributed_train_loss = _util.to_tensor(float(t.item()) / float(world_size))
        self._distributed_history_train_loss.append(float(self._distributed_train_loss))
        self._distributed_history_train_loss_trace.append(self._total_train_traces)
        return self._distributed_train_loss

    def _distributed_sync_parameters(self):
        import torch.distributed as dist
        self._hip_ensure()
        dist.broadcast(self._hip_engine.params, 0)              # :290-294 as one broadcast of the flat buffer

    # ---- importance sampling -----------------------------------------------------------------------------------------
    def _infer_init(self, observe=None):
        self._hip_ensure()
        self._infer_observe = observe
        vals = []
        for name in self._hip_obs_names:
            vals.extend(torch.as_tensor(observe[name], dtype=torch.float32).reshape(-1).tolist())
        self._hip_is.init(vals)
        self._infer_observe_embedding = self._hip_is.e_obs[:self._hip_engine.spec.e_obs].reshape(1, -1)
        self._hip_prev_address = None

    def _infer_step(self, variable, prev_variable=None, proposal_min_train_iterations=None):
        spec = self._hip_engine.spec
        address, distribution = variable.address, variable.distribution
        if spec.feedforward:
            prev_variable = None
        if address not in spec.address_id or (prev_variable is not None and prev_variable.address not in spec.address_id):
            warnings.warn('Using prior. No proposal for address: {}'.format(address))
            return distribution
        a = spec.address_id[address]
        layer = self._layers_proposal[address]
        if proposal_min_train_iterations is not None and layer._total_train_iterations < proposal_min_train_iterations:
            warnings.warn('Using prior. Proposal not sufficiently trained ({}/{}) for address: {}'.format(
                layer._total_train_iterations, proposal_min_train_iterations, address))
            return distribution
        prev = None if prev_variable is None else spec.address_id[prev_variable.address]
        sched = self._hip_scheduler
        if sched is not None:                                  # a particle coroutine: park, served in a batch
            return sched.infer_step(a, prev, distribution, prev_variable)
        run = self._hip_is
        if prev_variable is None:
            run.begin(1)
        else:
            run.prev_value = torch.as_tensor(prev_variable.value, dtype=torch.float32).reshape(1).to(run.dev)
        from .packed import distribution_params
        prior = torch.tensor([distribution_params(distribution)], dtype=torch.float32).to(run.dev)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())     # follows torch's global seed (pyprob.seed)
        value, logq = run.step(a, prev, prior, seed=seed)
        return ProposalSample(value.cpu(), logq.cpu())

    # ---- pickling (torch.save of the module, inference_network.py:162-196) -------------------------------------------
    def __getstate__(self):
        state = dict(self.__dict__)
        for k in [k for k in state if k.startswith('_hip_')]:
            del state[k]
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)


class InferenceNetworkLSTMHip(_HipNetworkMixin, _RefLSTM):
    _hip_kind = 'lstm'

    def _init_layers(self):
        super()._init_layers()          # nn.LSTM with the reference's initialisation; bound at the first _polymorph


class InferenceNetworkFeedForwardHip(_HipNetworkMixin, _RefFeedForward):
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
    from pyprob import InferenceEngine, PriorInflation, TraceMode, state
    from pyprob.distributions import Empirical
    trace_mode = TraceMode.PRIOR if trace_mode is None else trace_mode
    prior_inflation = PriorInflation.DISABLED if prior_inflation is None else prior_inflation
    inference_engine = InferenceEngine.IMPORTANCE_SAMPLING if inference_engine is None else inference_engine
    fast = (inference_engine == InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK
            and isinstance(inference_network, _HipNetworkMixin) and num_traces > 1 and _have_greenlet()
            and os.environ.get('PYPROB_HIP_COROUTINES', '1') != '0')
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
    global _original_traces
    import pyprob.model as M
    M.InferenceNetworkLSTM = InferenceNetworkLSTMHip
    M.InferenceNetworkFeedForward = InferenceNetworkFeedForwardHip
    if _original_traces is None:
        _original_traces = M.Model._traces
        M.Model._traces = _traces_with_coroutines


def uninstall():
    global _original_traces
    import pyprob.model as M
    M.InferenceNetworkLSTM = _RefLSTM
    M.InferenceNetworkFeedForward = _RefFeedForward
    if _original_traces is not None:
        M.Model._traces = _original_traces
        _original_traces = None
