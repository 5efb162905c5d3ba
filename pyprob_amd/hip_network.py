"""The HIP side of pyprob's `InferenceNetwork` subclass protocol (seam B1) - WITHOUT an import of pyprob.

`_HipNetworkMixin` is everything `pyprob_amd/binding.py` adds to `pyprob.nn.InferenceNetworkLSTM` / `...FeedForward`: the
module tree stays the base class's (layer classes, `state_dict` names, `_polymorph`, pickling), the arithmetic moves to the
engine. It only uses the attributes the reference's classes define (pyprob/nn/inference_network.py:25-80,
inference_network_lstm.py:11-80): `_layers_observe_embedding`, `_layers_proposal`, `_lstm_dim`, `_optimizer_type`, ... - so
the SAME class body runs
  * under the real pyprob (binding.py composes it with the reference's classes; tests/test_binding_reference.py), and
  * on the GPU box, where the reference cannot exist, composed with a data-holder module tree rebuilt from sessions RECORDED
    from the stock reference (tests/binding_standin.py, tests/test_gpu_binding_session.py): the recorded minibatches go through
    `_polymorph` / `_loss` / `backward` / `HipAdam.step` / `_infer_step` of this class on the device and must reproduce the
    stock network's recorded loss trajectory, weights and proposal log-probabilities.
"""
import os
import warnings

import torch

from . import lib as L
from .autograd import HipAdam, HipLoss, HipSGD, presence
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
        kind = str(self._optimizer_type).split('.')[-1].upper()        # pyprob.Optimizer member (or its name)
        if kind not in ('ADAM', 'SGD', 'ADAM_LARC', 'SGD_LARC'):
            raise ValueError('Unknown optimizer_type: {}'.format(self._optimizer_type))
        larc = kind.endswith('_LARC')                                                              # :351-352
        if kind.startswith('ADAM'):                                                                # :347-348
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
        self._distributed_train_loss = torch.tensor(float(t.item()) / float(world_size), dtype=torch.float32)
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
        # (a VIEW of the runner's embedding row: the launch that fills it is deferred into the first statement, ISRunner.init)
        self._infer_observe_embedding = self._hip_is._e_obs[:self._hip_engine.spec.e_obs].reshape(1, -1)
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
