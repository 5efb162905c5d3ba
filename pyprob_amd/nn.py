"""Host-side mirror of pyprob.nn for the hot path: Batch, OnlineDataset, InferenceNetworkLSTM.

Same names, argument meaning and error behaviour as the reference (pyprob/nn/dataset.py:21-62,
pyprob/nn/inference_network.py:25-599, pyprob/nn/inference_network_lstm.py:11-220); the arithmetic is the HIP
engine's (`ICEngine`, one C-ABI call per loss / optimizer step / IS statement).
"""
import math
import os
import time
import warnings

import numpy as np
import torch

from .engine import ICEngine
from .is_engine import ISRunner
from .packed import PackedBatch, pack_traces
from .spec import NetSpec


class Batch:
    """pyprob/nn/dataset.py:21-47: a minibatch of traces, sub-batched by address sequence."""

    def __init__(self, traces):
        self.traces = traces
        self.size = len(traces)
        sub_batches = {}
        total = 0
        for trace in traces:
            tl = trace.length_controlled
            if tl == 0:
                raise ValueError('Trace of length zero.')
            total += tl
            h = ''.join([v.address for v in trace.variables_controlled])
            sub_batches.setdefault(h, []).append(trace)
        self.sub_batches = list(sub_batches.values())
        self.mean_length_controlled = total / self.size

    def __len__(self):
        return len(self.traces)

    def __getitem__(self, key):
        return self.traces[key]


class _PackedIds:
    """A minibatch of a PackedTraceDataset before packing: trace ids + the addresses the network has not seen yet."""

    def __init__(self, dataset, ids, new_addresses):
        self.dataset, self.ids, self.new_addresses = dataset, ids, new_addresses
        self.size = len(ids)
        self.sub_batches = []
        self.mean_length_controlled = 0.0


class OnlineDataset:
    """pyprob/nn/dataset.py:50-62: every item runs the model once in PRIOR_FOR_INFERENCE_NETWORK mode."""

    def __init__(self, model, length=None, prior_inflation=None):
        from .state import PriorInflation
        self._model = model
        self._length = int(1e6) if length is None else length
        self._prior_inflation = PriorInflation.DISABLED if prior_inflation is None else prior_inflation

    def __len__(self):
        return self._length

    def __getitem__(self, idx):
        from .state import TraceMode
        return next(self._model._trace_generator(trace_mode=TraceMode.PRIOR_FOR_INFERENCE_NETWORK,
                                                 prior_inflation=self._prior_inflation))


class ProposalSample:
    """What `_infer_step` returns: the proposal of one controlled variable, already sampled and scored on the device
    (reference: a Mixture/Categorical object with .sample() and .log_prob(value, sum=True))."""

    def __init__(self, value, log_q):
        self._value, self._log_q = value, log_q

    def sample(self):
        return self._value

    def log_prob(self, value, sum=False):
        # state.sample scores exactly the value it just drew (state.py:208-212)
        return self._log_q.reshape(-1)[0] if sum else self._log_q


class InferenceNetworkLSTM:
    # observe_embeddings example: {'obs1': {'dim': 32}}   (FEEDFORWARD, depth 2)
    _network = 'lstm'           # NetSpec(network=...): 'lstm' | 'feedforward' (class InferenceNetworkFeedForward below)
    _engine_factory = ICEngine  # (spec, device=, seed=) -> engine; the CPU tests of the training loop put buffers on the host

    def __init__(self, model=None, observe_embeddings={}, lstm_dim=512, lstm_depth=1, sample_embedding_dim=4,
                 address_embedding_dim=64, distribution_type_embedding_dim=8, proposal_mixture_components=10,
                 device='cuda:0', seed=None):
        self._lstm_depth = int(lstm_depth)
        self._model = model
        self._observe_embeddings = observe_embeddings
        self._lstm_dim = lstm_dim
        self._sample_embedding_dim = sample_embedding_dim
        self._address_embedding_dim = address_embedding_dim
        self._distribution_type_embedding_dim = distribution_type_embedding_dim
        self._proposal_mixture_components = proposal_mixture_components
        self._device = device
        self._seed = seed
        self._engine = None
        self._is = None
        self._layers_initialized = False
        self._layers_pre_generated = False
        self._optimizer_ready = False
        self._learning_rate_init = None
        self._learning_rate_end = None
        self._learning_rate_scheduler_type = None
        self._weight_decay = None
        self._optimizer_type = None          # 'ADAM' | 'SGD' | 'ADAM_LARC' | 'SGD_LARC' (inference_network.py:39,439-440)
        self._momentum = None
        self._total_train_seconds = 0
        self._total_train_traces = 0
        self._total_train_traces_end = None
        self._total_train_iterations = 0
        self._loss_init = None
        self._loss_min = float('inf')
        self._loss_max = None
        self._loss_previous = float('inf')
        self._history_train_loss = []
        self._history_valid_loss = []
        self._history_valid_loss_trace = []
        self._history_train_loss_trace = []
        self._history_num_params = []
        self._history_num_params_trace = []
        self._distributed_world_size = 1
        self._infer_prev_addr_id = None

    # ---- layer creation ------------------------------------------------------------------------------------
    def _init_layers_observe_embedding(self, observe_embeddings, example_trace):
        """inference_network.py:80-130 for FEEDFORWARD embeddings: input width from the example trace."""
        if len(observe_embeddings) == 0:
            raise ValueError('At least one observe embedding is needed to initialize inference network.')
        if isinstance(observe_embeddings, set):
            observe_embeddings = {o: {} for o in observe_embeddings}
        obs = {}
        for name, value in observe_embeddings.items():
            v = dict(value)
            variable = example_trace.named_variables[name]
            v['input_dim'] = int(np.prod(v['reshape'])) if 'reshape' in v else int(torch.as_tensor(variable.value).numel())
            if 'dim' not in v:
                print('Observable {}: embedding dim not specified, using the default 256.'.format(name))
            obs[name] = v
        self._obs_spec = obs
        self._obs_names = list(obs.keys())

    def _init_layers(self):
        spec = NetSpec(self._obs_spec, lstm_dim=self._lstm_dim, sample_embedding_dim=self._sample_embedding_dim,
                       address_embedding_dim=self._address_embedding_dim,
                       distribution_type_embedding_dim=self._distribution_type_embedding_dim,
                       proposal_mixture_components=self._proposal_mixture_components, network=self._network,
                       lstm_depth=self._lstm_depth)
        self._engine = type(self)._engine_factory(spec, device=self._device, seed=self._seed)
        self._is = ISRunner(self._engine)

    def _polymorph(self, batch):
        """inference_network_lstm.py:34-80: new address => new embeddings + proposal head + sample embedding."""
        items = []
        spec = self._engine.spec
        seen = set()
        if isinstance(batch, _PackedIds):
            for a, dname, ncat in batch.new_addresses:
                items.append((a, dname, ncat))
                print('New layers, address: {}, distribution: {}'.format(a[:60], dname))
        for sub_batch in ([] if isinstance(batch, _PackedIds) else batch.sub_batches):
            for variable in sub_batch[0].variables_controlled:
                a = variable.address
                if a not in spec.address_id and a not in seen:
                    d = variable.distribution
                    ncat = d.num_categories if d.name == 'Categorical' else None
                    items.append((a, d.name, ncat))
                    seen.add(a)
                    print('New layers, address: {}, distribution: {}'.format(a[:60], d.name))
        layers_changed = self._engine.add_addresses(items) if items else False
        if layers_changed:
            n = spec.num_parameters()
            print('Total addresses: {:,}, distribution types: {:,}, parameters: {:,}'.format(
                len(spec.addresses), len(spec.dtypes), n))
            self._history_num_params.append(n)
            self._history_num_params_trace.append(self._total_train_traces)
        return layers_changed

    def _pre_generate_layers(self, dataset, batch_size=64, save_file_name_prefix=None):
        """inference_network.py:270-288: create the layers of every address the dataset contains before training; later
        minibatches are not polymorphed (optimize skips _polymorph, :476-479). Needed for data-parallel runs, where all
        ranks must hold the same parameter set."""
        if not self._layers_initialized:
            self._init_layers_observe_embedding(self._observe_embeddings, example_trace=dataset[0])
            self._init_layers()
            self._layers_initialized = True
        self._layers_pre_generated = True
        self._check_dataset_observables(dataset)
        changed = False
        if hasattr(dataset, 'addresses') and hasattr(dataset, 'trace_types'):       # packed dataset: the address table
            new = [a for a in dataset.addresses if a[0] not in self._engine.spec.address_id]
            changed = bool(new) and self._polymorph(_PackedIds(dataset, [], new))
        else:
            n = len(dataset) if hasattr(dataset, '__len__') else 0
            for i in range(0, min(n, int(1e5)), batch_size):
                changed |= bool(self._polymorph(Batch([dataset[k] for k in range(i, min(i + batch_size, n))])))
        if changed and save_file_name_prefix is not None:
            self._save('{}_00000000_pre_generated.network'.format(save_file_name_prefix))

    def _check_dataset_observables(self, dataset):
        """A packed dataset stores the observed values as columns in ITS obs_names order; the kernels read them with the
        network's layout. Both must name the same observables with the same widths, in the same order."""
        names = getattr(dataset, 'obs_names', None)
        if names is None or not hasattr(dataset, 'obs_width'):
            return
        widths = [int(w) for w in getattr(dataset, 'obs_widths', None) or []]
        want = [o[1] for o in self._engine.spec.obs]
        if list(names) != list(self._obs_names) or (widths and widths != want) or int(dataset.obs_width) != sum(want):
            raise ValueError('dataset observables {} (widths {}) do not match the observe embeddings of the inference '
                             'network {} (widths {}); write the dataset with save_dataset(obs_names={})'.format(
                                 list(names), widths, list(self._obs_names), want, list(self._obs_names)))

    # ---- training loss --------------------------------------------------------------------------------------
    def _pack(self, batch):
        if isinstance(batch, PackedBatch):
            return batch
        spec = self._engine.spec
        for sub_batch in batch.sub_batches:
            for v in sub_batch[0].variables_controlled:
                if v.address not in spec.address_id:
                    print('Address unknown by inference network: {}'.format(v.address))
                    return None
        return pack_traces(batch.traces, spec, self._obs_names)

    def _loss(self, batch, backward=False):
        """`_loss(batch)` -> (success, loss) like inference_network_lstm.py:136-220. `loss` is a 1-element device tensor
        (already divided by batch.size); with backward=True the gradients are left in the engine's flat buffer."""
        pb = self._pack(batch)
        if pb is None:
            return False, 0
        for info_id, n in enumerate(pb.cur_counts):
            if n > 0:
                self._engine.spec.addresses[info_id].total_train_iterations += 1      # :198
        loss = self._engine.loss(pb, backward=backward)
        return True, loss

    # ---- importance sampling ---------------------------------------------------------------------------------
    def _infer_init(self, observe=None):
        """inference_network.py:141-148"""
        self._infer_observe = observe
        vals = []
        for name in self._obs_names:
            vals.extend(torch.as_tensor(observe[name], dtype=torch.float32).reshape(-1).tolist())
        self._is.init(vals)
        self._infer_prev_addr_id = None

    def _prior_tensor(self, distribution, n=1):
        """The prior's parameter pair as the proposal heads read it, [rows, 2] on the device (scalar pairs are cached: no
        host-to-device copy per statement)."""
        key = None
        if distribution.name in ('Normal', 'Uniform'):
            a, b = (distribution._loc, distribution._scale) if distribution.name == 'Normal' else (distribution._low, distribution._high)
            if a.numel() == 1 and b.numel() == 1 and a.device.type == 'cpu' and b.device.type == 'cpu':
                key = (distribution.name, float(a), float(b))
                cache = self.__dict__.setdefault('_prior_cache', {})
                hit = cache.get(key)
                if hit is not None:
                    return hit
                if len(cache) > 1024:
                    cache.clear()
        p = self._prior_tensor_uncached(distribution)
        if key is not None:
            self._prior_cache[key] = p
        return p

    def _prior_tensor_uncached(self, distribution):
        if distribution.name == 'Normal':
            m, sd = distribution.mean.reshape(-1), distribution.stddev.reshape(-1)
            if sd.device != m.device:         # (a host scalar next to per-particle device means, distributions.Normal)
                sd = sd.to(m.device)
            p = torch.stack(torch.broadcast_tensors(m, sd), 1)
        elif distribution.name == 'Uniform':
            p = torch.stack([distribution.low.reshape(-1), distribution.high.reshape(-1)], 1)
        elif distribution.name == 'Poisson':      # the head's fixed interval, not the rate
            from .packed import POISSON_LOW_HIGH
            p = torch.tensor([POISSON_LOW_HIGH], dtype=torch.float32)
        else:
            return None
        return p.to(self._engine.device, torch.float32).contiguous()

    def _infer_step(self, variable, prev_variable=None, proposal_min_train_iterations=None):
        """inference_network_lstm.py:82-134 for one particle: returns the prior when the address is unknown or the
        proposal is not trained enough, else the (sampled, scored) proposal."""
        spec = self._engine.spec
        address = variable.address
        distribution = variable.distribution
        if spec.feedforward:            # inference_network_feedforward.py:52-66: no state, no previous variable
            prev_variable = None
        if address not in spec.address_id or (prev_variable is not None and prev_variable.address not in spec.address_id):
            warnings.warn('Using prior. No proposal for address: {}'.format(address))
            return distribution
        a = spec.address_id[address]
        if proposal_min_train_iterations is not None and \
                spec.addresses[a].total_train_iterations < proposal_min_train_iterations:
            warnings.warn('Using prior. Proposal not sufficiently trained for address: {}'.format(address))
            return distribution
        if prev_variable is None:
            self._is.begin(1)
            prev = None
        else:
            prev = spec.address_id[prev_variable.address]
            self._is.prev_value = torch.as_tensor(prev_variable.value, dtype=torch.float32).reshape(1).to(self._engine.device)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())       # follows torch's global seed (pyprob.seed)
        value, logq = self._is.step(a, prev, self._prior_tensor(distribution), seed=seed)
        return ProposalSample(value.cpu(), logq.cpu())

    def _infer_step_lockstep(self, address, distribution, ls):
        """One controlled sample statement of a lock-step run for the ACTIVE particles of the current execution path:
        LSTM step + proposal + draw + log q in one C-ABI call (all particles, or the gathered rows of a diverged path),
        then lw += log p(v) - log q(v) (state.py:211-217). A statement inside the replayed prefix of a path returns the
        values recorded when a superset of these particles executed it. Returns a ParticleTensor [n]."""
        from .state import ParticleTensor
        spec = self._engine.spec
        runner = ls.runner
        j = ls.statement
        ls.statement += 1
        if j < ls.replay_statements:
            values, a = ls.log[j][address]
            if a is not None:
                ls.prev_addr_id = a
                runner.prev_value = runner.last_value = values
            wrappers = ls.wrappers      # one ParticleTensor per recorded tensor: every replay hands out the same object
            w = wrappers.get(id(values))
            if w is None or w[0] is not values:
                w = wrappers[id(values)] = (values, ParticleTensor.wrap(values))
            return w[1]
        prev_unknown = getattr(ls, 'prev_unknown', False) and not spec.feedforward    # (FF: no previous-variable input)
        ls.prev_unknown = address not in spec.address_id
        if ls.prev_unknown or prev_unknown:
            # no proposal layers for this address or for the previous one (never seen in training): the prior is the
            # proposal, log p - log q = 0, and the LSTM state is not advanced (inference_network_lstm.py:100-104, 132-134)
            warnings.warn('Using prior. No proposal for address: {}'.format(address))
            entry = ls.log[j].get(address) if j < len(ls.log) else None
            values = entry[0] if entry is not None else torch.zeros(ls.n, dtype=torch.float32, device=runner.dev)
            draw = None
            if runner.dev.type == 'cuda' and distribution.name in ('Normal', 'Uniform'):
                # one prior draw per particle on the device (pp_prior_draw, Philox: counter = particle index, keyed by the
                # statement and the path like the proposal draws) - a host draw of n values + an upload per statement was
                # 0.7 ms + 0.05 ms at n = 200 000 (profiles/r04c_gumm_cprofile.txt)
                from .ops import ops
                a0, a1 = (distribution.mean, distribution.stddev) if distribution.name == 'Normal' else (distribution.low, distribution.high)
                p0 = torch.as_tensor(a0, dtype=torch.float32).reshape(-1).to(runner.dev)
                p1 = torch.as_tensor(a1, dtype=torch.float32).reshape(-1).to(runner.dev)
                if p0.numel() in (1, ls.n) and p1.numel() in (1, ls.n):
                    draw = ops.prior_draw(0 if distribution.name == 'Normal' else 1, p0, p1, ls.n,
                                          ls.seed + 7919 * j + 104729 * ls.path_id, ls.offset, 0x50)
            if draw is None:
                draw = distribution.sample()
                draw = torch.as_tensor(draw, dtype=torch.float32).reshape(-1)
                if draw.numel() == 1:     # shared prior: one draw per particle
                    draw = distribution._torch_dist.sample((ls.n,)).reshape(-1).float()
                draw = draw.to(runner.dev)
            if ls.rows is None:
                values = draw
            elif getattr(ls, 'by_rows', False):
                runner.copy_rows(draw, values, ls.rows)      # (in place: the other paths' recorded values stay where they are)
            else:
                values = torch.where(ls.active, draw, values)
            while len(ls.log) <= j:
                ls.log.append({})
            ls.log[j][address] = (values, spec.address_id.get(address))
            if not ls.prev_unknown:      # a known address after an unknown one: it is the next statement's "previous"
                ls.prev_addr_id = spec.address_id[address]
                runner.prev_value = runner.last_value = values
            return ParticleTensor.wrap(values)
        a = spec.address_id[address]
        prior = self._prior_tensor(distribution)
        seed = ls.seed + 7919 * j + 104729 * ls.path_id
        ls.flush()        # an earlier deferred draw is this statement's previous value: it must exist now
        info = spec.addresses[a]
        prior_term = runner.dist_term(distribution) if (ls.fused and ls.rows is None and ls.prev_addr_id is None) else None
        if (prior_term is not None and prior is not None and prior.numel() == 2 and
                info.dist_name in ('Normal', 'Uniform', 'Poisson')):
            # First statement of a trace, full width: every particle has the same proposal. Only the network runs now; the
            # draw, - log q, + log p and the observe terms that follow become ONE pass over the particles at the next flush.
            runner.step_net(a, None)
            values = torch.empty(ls.n, dtype=torch.float32, device=runner.dev)
            ls.draw = dict(addr=a, prior=prior, values=values, seed=seed, prior_term=prior_term)
            while len(ls.log) <= j:
                ls.log.append({})
            ls.log[j][address] = (values, a)
            runner.prev_value = runner.last_value = values
            ls.prev_addr_id = a
            return ParticleTensor.wrap(values)
        m_active = ls.n if ls.rows is None else ls.n_active
        if distribution.name == info.dist_name and runner.whole_statement_ok(a, ls.prev_addr_id, m_active, info.dist_name, prior):
            # the whole statement in ONE launch: previous values read at the particles' rows, the draw written to values[rows],
            # lw[rows] += log p(v) - log q(v) (state.py:211-217) - no gather, scatter, prior or axpy launches around it
            entry = ls.log[j].get(address) if j < len(ls.log) else None
            # (a full-width statement writes every row of a new record; a diverged path's new record is zero-filled: the rows of
            # particles that never execute this address are visible through Empirical.statement_log and enter the program's
            # full-width arithmetic - zeros there, as before round 4, not uninitialised memory)
            if entry is not None and ls.rows is not None:
                values = entry[0]
            elif ls.rows is None:
                values = torch.empty(ls.n, dtype=torch.float32, device=runner.dev)
            else:
                values = torch.zeros(ls.n, dtype=torch.float32, device=runner.dev)
            runner.statement_rows(ls.rows if ls.rows is not None else None, a, ls.prev_addr_id, prior, values, ls.lw,
                                  info.dist_name, seed=seed)
            while len(ls.log) <= j:
                ls.log.append({})
            ls.log[j][address] = (values, a)
            ls.prev_addr_id = a
            return ParticleTensor.wrap(values)
        if ls.rows is None:
            value, logq = runner.step(a, ls.prev_addr_id, prior, seed=seed)
            values = value
        else:
            entry = ls.log[j].get(address) if j < len(ls.log) else None
            values = entry[0] if entry is not None else torch.zeros(ls.n, dtype=torch.float32, device=runner.dev)
            value, logq = runner.step_rows(ls.rows, a, ls.prev_addr_id, prior, seed=seed)
            values.index_copy_(0, ls.rows, value)
        while len(ls.log) <= j:
            ls.log.append({})
        ls.log[j][address] = (values, a)
        runner.prev_value = runner.last_value = values     # previous-sample embedding input of the next statement (full size)
        self._accumulate_prior(ls, distribution, values)             # + log p(value)   state.py:211
        if ls.rows is None:
            runner.axpy(ls.lw, -1.0, logq)                           # - log q(value)   state.py:212,217
        else:
            ls.lw.index_add_(0, ls.rows, logq, alpha=-1.0)
        ls.prev_addr_id = a
        return ParticleTensor.wrap(values)

    def _accumulate_prior(self, ls, distribution, value):
        """+ log p(value) of the program's own prior (state.py:211), on the device for the families with a kernel."""
        term = ls.runner.dist_term(distribution)
        if term is None:   # a family without a device kernel: scored on the host
            lp = distribution.log_prob(value.cpu()).to(self._engine.device, torch.float32).contiguous()
            if ls.rows is not None:
                lp = torch.where(ls.active, lp, torch.zeros_like(lp))
            ls.runner.axpy(ls.lw, 1.0, lp)
            return
        if getattr(ls, 'by_rows', False) and ls.rows is not None:
            ls.runner.accumulate_rows(ls.lw, term, value, ls.rows, 1.0)
        else:
            ls.runner.accumulate_masked(ls.lw, None, None, None, value, ls.active, term=term)

    def _validation_loss(self, dataset_valid, batch_size):
        """Mean `_loss` over the minibatches of a packed validation dataset, forward only (inference_network.py:538-543);
        a minibatch with an address the network does not know is skipped like `_loss` returning (False, 0)."""
        self._check_dataset_observables(dataset_valid)
        sampler = dataset_valid.sampler(min(batch_size, len(dataset_valid)), 0, 1, None, False, False)
        total, n = None, 0
        for ids in sampler:
            try:
                pb = dataset_valid.device_batch(ids, self._engine.spec, self._engine.device)
            except KeyError:
                print('Address unknown by inference network (validation minibatch skipped)')
                continue
            loss = self._engine.loss(pb)
            total = loss.clone() if total is None else total + loss
            n += 1
        return float(total.item()) / n if n else float('nan')

    def _learning_rate(self, traces=None):
        """POLY1 / POLY2 decay driven by the trace count (inference_network.py:357-379, :568)."""
        traces = self._total_train_traces if traces is None else traces
        t = self._learning_rate_scheduler_type
        t = None if t is None else str(t).split('.')[-1].upper()      # 'POLY1' or LearningRateScheduler.POLY1
        if t in (None, 'NONE'):
            return self._learning_rate_init
        if t not in ('POLY1', 'POLY2'):
            raise ValueError('Unknown learning_rate_scheduler_type: {}'.format(self._learning_rate_scheduler_type))
        power = 1.0 if t == 'POLY1' else 2.0
        frac = max(0.0, 1.0 - traces / self._total_train_traces_end)
        return (self._learning_rate_init - self._learning_rate_end) * (frac ** power) + self._learning_rate_end

    def optimize(self, num_traces, dataset, batch_size=64, learning_rate_init=0.0001, learning_rate_end=1e-6,
                 learning_rate_scheduler_type=None, weight_decay=1e-5, num_traces_end=1e9, distributed_backend=None,
                 distributed_params_sync_every_iter=10000, stop_with_bad_loss=False, log_file_name=None, verbose=True,
                 distributed_num_buckets=None, save_file_name_prefix=None, save_every_sec=600, dataset_valid=None,
                 valid_every=None, optimizer_type=None, momentum=0.9):
        """The training loop of inference_network.py:381-599: per minibatch _polymorph -> zero_grad -> _loss -> backward ->
        [all-reduce, divide by world] -> optimizer step (Optimizer.ADAM | SGD | ADAM_LARC | SGD_LARC, :343-355; the runs
        inside one C call are Adam's, the other optimizers step per minibatch), traces/s bookkeeping."""
        if not self._layers_initialized:
            self._init_layers_observe_embedding(self._observe_embeddings, example_trace=dataset[0])
            self._init_layers()
            self._layers_initialized = True
        self._check_dataset_observables(dataset)
        world, rank = 1, 0
        if distributed_backend is not None:
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group(backend=distributed_backend)
            world, rank = dist.get_world_size(), dist.get_rank()
        self._distributed_world_size = world
        self._engine.world_size = world
        if world > 1 and distributed_backend == 'nccl' and os.environ.get('PP_DP_NATIVE', '1') == '1':
            # this library's own RCCL communicator: the gradient exchange is issued from the C side, and the native loop
            # pp_train_steps gets its data-parallel branch (every rank agrees on the outcome; otherwise torch.distributed)
            from .parallel import init_native_comm
            self._engine.native_dp = bool(init_native_comm(self._engine.device))
        if world > 1:
            # dL/dW_hh is zero on a rank whose traces all have one controlled variable: when EVERY rank finds that in its own
            # dataset (packed datasets know their trace lengths; an online generator does not) the range leaves the gradient
            # all-reduce (ICEngine.agree_skip_recurrent: one MIN-all-reduce of the per-rank finding, bit-identical parameters)
            lens = getattr(dataset, 'trace_len', None)
            try:
                single = lens is not None and len(lens) > 0 and int(np.max(lens)) == 1
            except (TypeError, ValueError):
                single = False
            self._engine.agree_skip_recurrent(bool(single))
        if self._learning_rate_init is None:
            self._learning_rate_init = learning_rate_init * math.sqrt(world)            # :448
        if self._learning_rate_end is None:
            self._learning_rate_end = learning_rate_end
        if self._learning_rate_scheduler_type is None:
            self._learning_rate_scheduler_type = learning_rate_scheduler_type
        if self._weight_decay is None:
            self._weight_decay = weight_decay
        if self._optimizer_type is None:                                                  # :439-442
            self._optimizer_type = 'ADAM' if optimizer_type is None else str(optimizer_type).split('.')[-1].upper()
        if self._momentum is None:
            self._momentum = momentum
        if self._optimizer_type not in ('ADAM', 'SGD', 'ADAM_LARC', 'SGD_LARC'):
            raise ValueError('Unknown optimizer_type: {}'.format(self._optimizer_type))
        want = dict(kind='sgd' if self._optimizer_type.startswith('SGD') else 'adam', larc=self._optimizer_type.endswith('LARC'),
                    momentum=float(self._momentum))
        if self._engine.optimizer != want:
            self._engine.set_optimizer(**want)          # (a new optimizer: empty state)
        plain_adam = want['kind'] == 'adam' and not want['larc']
        if self._total_train_traces_end is None:
            self._total_train_traces_end = num_traces_end
        prev_seconds = self._total_train_seconds
        time_start = time.time()
        trace = 0
        log_file = open(log_file_name, 'w', buffering=1) if (rank == 0 and log_file_name) else None
        if log_file:
            log_file.write('time, iteration, trace, loss, learning_rate, mean_trace_length_controlled, sub_mini_batches, '
                           'traces_per_second\n')
        stop = False
        last = time_start
        i_item = 0
        # Packed offline dataset (pyprob_amd/dataset.py): minibatches come from the reference's bucketed sampler over the
        # (length, type)-sorted index space (dataset.py:328-400) and are packed from memory-mapped columns - no Trace
        # objects. Anything else is indexed trace by trace like the reference's DataLoader does.
        # (stop_with_bad_loss does not force per-iteration syncs: a flagged minibatch never touches the parameters - Adam
        # skips it on the device - and training stops when its status is read back, at most two runs later)
        # data parallel: the C loop's all-reduce branch needs this library's communicator on every rank; the ranks plan the
        # same runs (same sampler length, same cuts: chunk size, parameter broadcasts, epoch ends, num_traces - and the
        # loss / non-finite flag they book are the all-reduced ones), tests/test_dp_gloo.py. PP_DP_NATIVE_LOOP=0: per-step loop
        dp_native = world > 1 and bool(self._engine.native_dp) and os.environ.get('PP_DP_NATIVE_LOOP', '1') == '1'
        sync_every = 1 if (log_file_name or (world > 1 and not dp_native)) else 16
        save_state = [time_start - (save_every_sec or 0)]       # (the reference saves at the first iteration, :461-462)
        if valid_every is None:
            valid_every = max(100, num_traces / 1000)                                     # :435-436
        valid_state = [-valid_every + 1]                                                  # last_validation_trace, :437
        ring_n = 64
        loss_ring = torch.zeros(ring_n, dtype=torch.float32, device=self._engine.device)
        status_ring = torch.zeros(ring_n, dtype=torch.int32, device=self._engine.device)
        pending = []
        packed = hasattr(dataset, 'gather') and hasattr(dataset, 'sorted_indices')
        if packed:
            sampler = dataset.sampler(batch_size, rank, world, distributed_num_buckets)
            sampler_iter = iter(sampler)

        def book(pending, losses, bad):
            """Per-iteration bookkeeping of inference_network.py:497-531 for the iterations whose loss / non-finite flag
            were just read back. Returns True when training must stop (stop_with_bad_loss)."""
            nonlocal trace, stop, last
            now = time.time()
            dt_each = (now - last) / len(pending)
            for k, (bsize, mean_len, n_sub) in enumerate(pending):
                loss = float(losses[k]) / (world if world > 1 else 1)                # tail is the all-reduced SUM
                if bad[k] != 0:
                    print('Cannot compute loss, skipping batch. Loss: {}'.format(loss))
                    trace -= bsize * world
                    stop = trace >= num_traces
                    if stop_with_bad_loss:
                        return True
                    continue
                if self._loss_init is None:
                    self._loss_init = self._loss_max = loss
                self._loss_min = min(self._loss_min, loss)
                self._loss_max = max(self._loss_max, loss)
                self._loss_previous = loss
                self._total_train_iterations += 1
                self._total_train_traces += bsize * world
                self._total_train_seconds = prev_seconds + (last + dt_each * (k + 1) - time_start)
                self._history_train_loss.append(loss)
                self._history_train_loss_trace.append(self._total_train_traces)
                if log_file:
                    log_file.write('{}, {}, {}, {}, {}, {}, {}, {}\n'.format(
                        self._total_train_seconds, self._total_train_iterations, self._total_train_traces, loss,
                        self._learning_rate(), mean_len, n_sub, bsize * world / max(dt_each, 1e-9)))
            if dataset_valid is not None and trace - valid_state[0] > valid_every:        # :535-548
                valid_state[0] = trace - 1
                self._history_valid_loss.append(self._validation_loss(dataset_valid, batch_size))
                self._history_valid_loss_trace.append(self._total_train_traces)
            last = now
            if rank == 0 and save_file_name_prefix is not None and save_every_sec is not None \
                    and now - save_state[0] > save_every_sec:                      # inference_network.py:550-556
                save_state[0] = now
                self._save('{}_{}_traces_{}.network'.format(save_file_name_prefix, time.strftime('%Y%m%d_%H%M%S'),
                                                            self._total_train_traces))
            return False

        # Single rank + packed dataset: runs of up to `chunk_steps` iterations execute inside ONE C call (pp_train_steps:
        # pack -> upload -> loss + backward -> Adam per step, no Python in between); Python plans the minibatches of a
        # run, polymorphs at exactly the iteration where a new address first appears (a run ends before it), computes
        # the learning rates, and reads the run's losses back once. PP_PYTHON_LOOP=1 keeps the per-step Python loop.
        # (a Bernoulli head's rows carry statistics of their sub-batch step, filled in by the Python packer only
        # - dataset._bernoulli_step_stats -: such programs keep the per-step loop)
        has_bernoulli = any(a.dist_name == 'Bernoulli' for a in self._engine.spec.addresses) or \
            (packed and any(a[1] == 'Bernoulli' for a in getattr(dataset, 'addresses', [])))
        # (data parallel: the C loop has an all-reduce branch - tests/test_gpu_dp_native.py on a one-rank group, the run
        # planning below on two gloo ranks in tests/test_dp_gloo.py)
        native = packed and plain_adam and (world == 1 or dp_native) and \
            not has_bernoulli and os.environ.get('PP_PYTHON_LOOP', '0') != '1'
        traces0 = self._total_train_traces
        planned_iters = 0
        chunk_steps = 1 if sync_every == 1 else 64
        carry = None
        type_key, type_known = None, None
        # While a run trains inside the C call, a worker thread generates the next chunk of prior traces
        # (VectorisedOnlineDataset.start_prefetch). torch's intra-op pool must not fan that work out over all host cores:
        # measured (round-1 probe, 256-core host) a 128-thread torch.normal next to pp_train_steps slows a
        # step from 180 us to 650-1100 us, with one intra-op thread there is no interference.
        prefetching = native and hasattr(dataset, 'start_prefetch')
        cpu_threads = torch.get_num_threads()
        if prefetching:
            torch.set_num_threads(1)

        def end_prefetch():
            if hasattr(dataset, 'wait_prefetch'):
                dataset.wait_prefetch()    # the prior generator shares the module-global trace state with model code
            if prefetching:
                torch.set_num_threads(cpu_threads)
        # The bookkeeping of a run lags one run behind: pp_train_steps returns while its last steps are still queued,
        # the losses travel to the host asynchronously, and Python plans and launches the NEXT run before it settles the
        # previous one - the GPU never waits for the interpreter. (Runs of one step - log file, stop_with_bad_loss - are
        # settled immediately, like the reference's float(loss) per iteration.)
        inflight = None
        try:
            while native:
                if stop:          # everything is planned: settle the run in flight (a skipped minibatch reopens the loop)
                    if inflight is not None:
                        (metas_p, wait), inflight = inflight, None
                        if book(metas_p, *wait()):
                            return
                    if stop:
                        break
                    continue
                steps, metas, planned, epoch_end = [], [], trace, False
                res = getattr(dataset, 'resident', None) if world == 1 else None
                if res is not None:
                    # The chunk of prior traces was drawn ON THE DEVICE (state.PriorLockStep + pp_prior_draw) and every trace
                    # has the same single statement: its columns stay in HBM as minibatch blocks (packed.ColumnarDataset) and a
                    # run of steps is one pp_train_resident call - no host copy of the chunk, no packing, no upload.
                    a_new = [res['address']] if res['address'][0] not in self._engine.spec.address_id else []
                    if a_new and not self._layers_pre_generated and self._polymorph(_PackedIds(dataset, [], a_new)):
                        self._engine.reset_optimizer()                                    # :481-483
                    a_id = self._engine.spec.address_id[res['address'][0]]
                    n_addr = len(self._engine.spec.addresses)
                    if res.get('pos') is None:
                        # ONE persistent block buffer per dataset: a new chunk is copied into it (stream-ordered behind the run
                        # in flight) and the PackedBatch objects of its blocks are built once, not per chunk (64 x ~15 us of
                        # Python per chunk were 15 % of the step time)
                        from .packed import ColumnarDataset
                        slot = dataset.__dict__.get('_resident_slot')
                        key = (batch_size, a_id, n_addr)
                        if slot is None or slot['key'] != key or not slot['cd'].refill(res['obs'], res['values'], res['prior']):
                            cd = ColumnarDataset(res['obs'], res['values'], res['prior'], batch_size)
                            cache = {}
                            slot = dict(key=key, cd=cd, batches=[cd.batch(i, a_id, n_addr, cache) for i in range(cd.n_batches)])
                            dataset.__dict__['_resident_slot'] = slot
                        res['slot'], res['pos'] = slot, 0
                    cd = res['slot']['cd']
                    batches = []
                    while len(batches) < chunk_steps and planned < num_traces and res['pos'] < cd.n_batches:
                        batches.append(res['slot']['batches'][res['pos']])
                        res['pos'] += 1
                        metas.append((batch_size, 1.0, 1))
                        planned += batch_size
                    if batches:
                        seen = traces0 + trace + np.concatenate([[0], np.cumsum([m[0] for m in metas])[:-1]])
                        lrs = [self._learning_rate(t) for t in seen]
                        losses_t, status_t = self._engine.train_resident(batches, lrs, weight_decay=self._weight_decay)
                        self._engine.spec.addresses[a_id].total_train_iterations += len(batches)
                        launched = (metas, self._engine.read_back(losses_t, status_t))
                        trace = planned
                        stop = trace >= num_traces
                        if chunk_steps == 1:
                            inflight, launched = launched, None
                        if inflight is not None:
                            (metas_p, wait), inflight = inflight, None
                            if book(metas_p, *wait()):
                                return
                        inflight = launched
                    if res['pos'] >= cd.n_batches and not stop:      # chunk used up: the next one (prefetched meanwhile)
                        dataset.refresh()
                        if hasattr(dataset, 'start_prefetch'):
                            dataset.start_prefetch()
                        sampler = dataset.sampler(batch_size, rank, world, distributed_num_buckets)
                        sampler_iter = iter(sampler)
                    continue
                limit = chunk_steps
                if world > 1:      # _distributed_sync_parameters every N iterations (:473-474): a run ends where one is due
                    if planned_iters % distributed_params_sync_every_iter == 0:
                        self._engine.broadcast_params()
                    limit = min(limit, distributed_params_sync_every_iter - planned_iters % distributed_params_sync_every_iter)
                while len(steps) < limit and planned < num_traces:
                    if carry is not None:
                        ids, carry = carry, None
                    else:
                        try:
                            ids = next(sampler_iter)
                        except StopIteration:
                            epoch_end = True
                            break
                    # Planning cost per minibatch is one fancy index: a trace TYPE whose addresses all have layers is
                    # "known" (table rebuilt when the network grows or the dataset chunk changes); only a minibatch with
                    # an unknown type takes the address-by-address route of _polymorph.
                    tkey = (getattr(dataset, 'generated', 0), len(self._engine.spec.addresses))
                    if tkey != type_key:
                        aid = self._engine.spec.address_id
                        type_known = np.asarray([all(dataset.addresses[a][0] in aid for a in seq)
                                                 for _, seq in dataset.trace_types], bool)
                        type_key = tkey
                    ttypes = dataset.trace_type[ids]
                    if not self._layers_pre_generated and not type_known[ttypes].all():
                        types = dataset.types_of(ids)
                        new = [a for a in dataset.addresses_of(ids, types) if a[0] not in self._engine.spec.address_id]
                        if new and steps:
                            carry = ids             # train the planned iterations with the current layers first
                            break
                        if new and self._polymorph(_PackedIds(dataset, ids, new)):
                            self._engine.reset_optimizer()                                # :481-483
                    steps.append(ids)
                    if log_file:
                        metas.append((len(ids), float(dataset.trace_len[ids].mean()), len(dataset.types_of(ids))))
                    else:
                        metas.append((len(ids), 0.0, 0))
                    planned += len(ids) * world
                if steps:
                    planned_iters += len(steps)
                    seen = traces0 + trace + world * np.concatenate([[0], np.cumsum([m[0] for m in metas])[:-1]])
                    lrs = [self._learning_rate(t) for t in seen]
                    losses_t, status_t = self._engine.train_run(dataset, steps, lrs, weight_decay=self._weight_decay)
                    launched = (metas, self._engine.read_back(losses_t, status_t))
                    trace = planned
                    stop = trace >= num_traces
                    if chunk_steps == 1:
                        inflight, launched = launched, None
                    if inflight is not None:
                        (metas_p, wait), inflight = inflight, None
                        if book(metas_p, *wait()):
                            return
                    inflight = launched
                if epoch_end:
                    if hasattr(dataset, 'refresh'):                                       # online: fresh prior traces
                        dataset.refresh()
                        sampler = dataset.sampler(batch_size, rank, world, distributed_num_buckets)
                        if hasattr(dataset, 'start_prefetch') and not stop:
                            dataset.start_prefetch()     # the next chunk is generated while this one trains (C call, no GIL)
                    sampler_iter = iter(sampler)                                          # next epoch (:461-464)
        finally:
            end_prefetch()
        while not stop:
            if packed:
                try:
                    ids = next(sampler_iter)
                except StopIteration:
                    if hasattr(dataset, 'refresh'):                                   # online: fresh prior traces
                        dataset.refresh()
                        sampler = dataset.sampler(batch_size, rank, world, distributed_num_buckets)
                    sampler_iter = iter(sampler)                                      # next epoch (:461-464)
                    ids = next(sampler_iter)
                types = dataset.types_of(ids)
                new = [a for a in dataset.addresses_of(ids, types) if a[0] not in self._engine.spec.address_id]
                batch = _PackedIds(dataset, ids, new)
            else:
                traces = [dataset[i_item + k] for k in range(batch_size)]
                i_item += batch_size
                batch = Batch(traces)
            if world > 1 and self._total_train_iterations % distributed_params_sync_every_iter == 0:
                self._engine.broadcast_params()                                       # :473-474
            layers_changed = False if self._layers_pre_generated else self._polymorph(batch)
            if layers_changed:
                self._engine.reset_optimizer()                                        # :481-483
            # The loss and the non-finite status of a step stay on the device: a slot of a small ring per iteration, read
            # back every `sync_every` iterations (1 = the reference's behaviour: float(loss) every iteration,
            # inference_network.py:497; used whenever a log file wants per-iteration timestamps or ranks must agree).
            # Adam checks the status flag itself (pp_adam_step `skip`), so a bad batch is skipped without a host sync.
            slot = len(pending) % ring_n
            l_out, s_out = loss_ring[slot:slot + 1], status_ring[slot:slot + 1]
            if packed:
                dev_batch = dataset.device_batch(ids, self._engine.spec, self._engine.device)
                batch.mean_length_controlled = dev_batch.mean_length_controlled
                batch.sub_batches = [None] * len(types)
                pb = dev_batch
            else:
                pb = self._pack(batch)
                if pb is None:
                    print('Cannot compute loss, skipping batch. Loss: {}'.format(0))
                    if stop_with_bad_loss:
                        return
                    continue
            for info_id, n in enumerate(pb.cur_counts):
                if n > 0:
                    self._engine.spec.addresses[info_id].total_train_iterations += 1      # :198
            if world > 1:
                self._engine.loss(pb, backward=True)                                  # loss in the all-reduced tail
                self._engine.allreduce_grads()                                        # :494-495
                l_out.copy_(self._engine.loss_buf[:1])
                # the non-finite flag was reduced with the gradients: every rank skips (and books) the same iterations
                s_out.copy_(self._engine.status_tail[:1])
                self._engine.optimizer_step(self._learning_rate(traces0 + trace), weight_decay=self._weight_decay, zero_grads=True,
                                            skip=self._engine.reduced_status())
            else:
                self._engine.loss(pb, backward=True, loss_out=l_out, status_out=s_out)
                self._engine.optimizer_step(self._learning_rate(traces0 + trace), weight_decay=self._weight_decay, zero_grads=True, skip=s_out)
            pending.append((batch.size, batch.mean_length_controlled, len(batch.sub_batches)))
            trace += batch.size * world
            stop = trace >= num_traces
            if len(pending) < sync_every and not stop:
                continue
            # ---- read back and book-keep the pending iterations ------------------------------------------------------
            if book(pending, loss_ring[:len(pending)].cpu().numpy(), status_ring[:len(pending)].cpu().numpy()):
                return
            pending = []
        if rank == 0 and save_file_name_prefix is not None:                               # inference_network.py:596-599
            self._save('{}_{}_traces_{}.network'.format(save_file_name_prefix, time.strftime('%Y%m%d_%H%M%S'),
                                                        self._total_train_traces))
        if verbose and rank == 0:
            print('Stop condition reached. num_traces: {}  loss {:+.3e}  traces/s {:,.0f}'.format(
                num_traces, self._loss_previous, self._total_train_traces / max(self._total_train_seconds, 1e-9)))
        if log_file:
            log_file.close()

    # ---- checkpoint (state_dict interchange with the reference's tensor names) ---------------------------------
    def state_dict(self):
        return self._engine.state_dict()

    # training state that survives a save / load like the reference's pickled module (inference_network.py:162-196)
    _PERSISTED = ('_learning_rate_init', '_learning_rate_end', '_learning_rate_scheduler_type', '_weight_decay',
                  '_optimizer_type', '_momentum',
                  '_total_train_seconds', '_total_train_traces', '_total_train_traces_end', '_total_train_iterations',
                  '_loss_init', '_loss_min', '_loss_max', '_loss_previous', '_history_train_loss', '_history_valid_loss',
                  '_history_valid_loss_trace', '_history_train_loss_trace', '_history_num_params',
                  '_history_num_params_trace', '_layers_pre_generated', '_distributed_world_size')

    def _save(self, file_name):
        spec = self._engine.spec
        sched = self._learning_rate_scheduler_type
        state = {k: getattr(self, k) for k in self._PERSISTED}
        state['_learning_rate_scheduler_type'] = None if sched is None else str(sched).split('.')[-1].upper()
        torch.save(dict(state_dict=self.state_dict(), obs_spec=self._obs_spec, lstm_dim=self._lstm_dim, network=self._network,
                        lstm_depth=self._lstm_depth,
                        K=self._proposal_mixture_components,
                        dims=dict(sample_embedding_dim=self._sample_embedding_dim,
                                  address_embedding_dim=self._address_embedding_dim,
                                  distribution_type_embedding_dim=self._distribution_type_embedding_dim),
                        addresses=[(a.address, a.dist_name, a.num_categories, a.total_train_iterations) for a in spec.addresses],
                        total_train_traces=self._total_train_traces, total_train_iterations=self._total_train_iterations,
                        train_state=state,
                        exp_avg=self._engine.exp_avg.cpu(), exp_avg_sq=self._engine.exp_avg_sq.cpu(),
                        tensor_step=self._engine.tensor_step.cpu()), file_name)

    @staticmethod
    def _load(file_name, device='cuda:0'):
        d = torch.load(file_name, weights_only=False)
        cls = InferenceNetworkFeedForward if d.get('network', 'lstm') == 'feedforward' else InferenceNetworkLSTM
        net = cls(observe_embeddings=d['obs_spec'], lstm_dim=d['lstm_dim'], proposal_mixture_components=d['K'], device=device,
                  lstm_depth=d.get('lstm_depth', 1), **d.get('dims', {}))
        net._obs_spec = d['obs_spec']
        net._obs_names = list(d['obs_spec'].keys())
        net._init_layers()
        net._layers_initialized = True
        net._engine.add_addresses([(a, dn, nc) for a, dn, nc, _ in d['addresses']])
        for info, (_, _, _, it) in zip(net._engine.spec.addresses, d['addresses']):
            info.total_train_iterations = it
        net._engine.load_state_dict(d['state_dict'])
        ot = d.get('train_state', {}).get('_optimizer_type')
        if ot is not None:         # before the optimizer state is restored: choosing an optimizer empties it
            net._engine.set_optimizer('sgd' if ot.startswith('SGD') else 'adam', ot.endswith('LARC'),
                                      d['train_state'].get('_momentum') or 0.9)
        net._engine.exp_avg.copy_(d['exp_avg'])
        net._engine.exp_avg_sq.copy_(d['exp_avg_sq'])
        net._engine.tensor_step.copy_(d['tensor_step'])
        net._engine.moments_written()
        net._total_train_traces = d['total_train_traces']
        net._total_train_iterations = d['total_train_iterations']
        for k, v in d.get('train_state', {}).items():      # (files written before the training state was persisted lack it)
            setattr(net, k, v)
        return net


class InferenceNetworkFeedForward(InferenceNetworkLSTM):
    """pyprob/nn/inference_network_feedforward.py: the proposal layer of every address reads the observe embedding (no
    LSTM, no address / sample embeddings; `lstm_dim` and the embedding dimensions are ignored). Same engine, same C
    calls: the network description carries lstm_dim = 0 (NetSpec(network='feedforward'))."""
    _network = 'feedforward'
