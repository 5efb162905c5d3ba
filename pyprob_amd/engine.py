"""ICEngine: device-resident state of one LSTM inference network and the calls into libpyprob_amd.so.

Owns (as PyTorch-ROCm tensors -- plumbing for HBM allocation and streams only): the flat parameter buffer, the
identically laid out gradient and Adam-moment buffers, the per-tensor step/presence arrays and the kernel workspace.
Every compute step is a C-ABI call (include/pyprob_amd.h); nothing here falls back to torch ops or to the oracle.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L
from .spec import NetSpec


import os
_ADAM_CLEARS = os.environ.get('PP_ADAM_ZERO', '1') != '0'   # A/B knob: let pp_ic_loss memset the gradients instead
import itertools
_engine_tokens = itertools.count(1)      # a per-process monotonic identity (id() values are reused after collection)


class ICEngine:
    def __init__(self, spec: NetSpec, device='cuda:0', seed=None):
        self.lib = L.load()
        self.token = next(_engine_tokens)
        if not torch.cuda.is_available():
            raise L.HipLibraryError('pyprob_amd needs a ROCm device (torch.cuda.is_available() is False); there is '
                                    'no CPU fallback.')
        self.spec = spec
        self.device = torch.device(device)
        self.rng = np.random.default_rng(seed)
        self.params = torch.zeros(0, dtype=torch.float32, device=self.device)
        self.workspace = None
        self.ws_bytes = 0
        self.ws_shape = (0, 0)
        self.status_buf = torch.zeros(4, dtype=torch.int32, device=self.device)
        self.world_size = 1
        self.force_allreduce = False   # run the collective even with one rank (exercises the RCCL path)
        self.optimizer = dict(kind='adam', larc=False, momentum=0.9)     # set_optimizer()
        self._resize(initialise=list(spec.tensors.keys()))

    # ---- buffers -----------------------------------------------------------------------------------------
    def _resize(self, initialise):
        """(Re)allocate the flat buffers for the current spec, keep existing values, initialise new tensors, and
        reset the optimizer state (the reference rebuilds Adam when layers change, inference_network.py:481-483)."""
        spec = self.spec
        n = spec.n_params
        new = torch.zeros(n + 1024, dtype=torch.float32, device=self.device)[:n]
        old_n = self.params.numel()
        if old_n:
            new[:old_n].copy_(self.params)   # tensors are only ever appended
        host = {}
        for name in initialise:
            host[name] = spec.init_tensor(name, self.rng)
        self.params = new
        for name, arr in host.items():
            self.set_tensor(name, arr)
        # grads carry a tail: [n_tensors presence flags | loss | non-finite flag] so that DP needs ONE all-reduce
        #  and every rank sees the SAME skip decision: a batch whose loss is not finite on ANY rank is
        # skipped by ALL ranks (the sum of the flags is > 0 everywhere; the NaN gradients it summed in are discarded)
        self.grads_full = torch.zeros(n + spec.n_tensors + 2, dtype=torch.float32, device=self.device)
        self.grads = self.grads_full[:n]
        self.active = self.grads_full[n:n + spec.n_tensors]
        self.loss_buf = self.grads_full[n + spec.n_tensors:n + spec.n_tensors + 1]   # the loss kernel writes into the tail
        self.status_tail = self.grads_full[n + spec.n_tensors + 1:]                  # float copy of the non-finite flag
        self.dp_skip = []      # (offset, count) float ranges left out of the gradient all-reduce, see skip_recurrent_weights
        self.dp_overlap = getattr(self, 'dp_overlap', [])     # (offset, count) ranges reduced EARLY, see enable_dp_overlap
        # set by the caller after pyprob_amd.parallel.init_native_comm() returned True on ALL ranks (kept when the network grows)
        self.native_dp = getattr(self, 'native_dp', False)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.tensor_step = torch.zeros(spec.n_tensors, dtype=torch.int32, device=self.device)
        self.arrived = torch.zeros(L.PP_ADAM_SCRATCH * spec.n_tensors, dtype=torch.int32, device=self.device)   # pp_adam_step scratch
        self._grads_clean = False      # the last Adam step cleared the gradient buffer (zero_grad of the next step)
        self.chunk_tensor = torch.from_numpy(spec.chunk_tensor_map()).to(self.device)
        self.addr_table = torch.from_numpy(spec.address_table()).to(self.device)
        self.net = spec.c_struct(self.addr_table.data_ptr())
        from . import ops
        ops.unregister_net(getattr(self, 'net_handle', None))
        self.net_handle = ops.register_net(self.net, spec)          # the operators' handle of this layer set (ops.py)
        self._active_cache = {}
        self._active_key = None
        self.ws_shape = (0, 0)

    def add_addresses(self, items):
        """items: iterable of (address, dist_name, num_categories). Returns True if layers were created."""
        created = []
        for address, dist_name, ncat in items:
            created += self.spec.add_address(address, dist_name, ncat)
        if created:
            self._resize(initialise=created)
        return bool(created)

    def _ensure_workspace(self, n_traces, n_rows):
        if n_traces <= self.ws_shape[0] and n_rows <= self.ws_shape[1]:
            return
        bt, br = max(n_traces, self.ws_shape[0]), max(n_rows, self.ws_shape[1])
        need = self.lib.pp_ic_workspace_bytes(C.byref(self.net), bt, br)
        if need == 0:
            raise RuntimeError('pp_ic_workspace_bytes failed: %s' % self.lib.pp_last_error().decode())
        self.workspace = torch.zeros(need, dtype=torch.uint8, device=self.device)     # (zero-filled: the header's contract)
        self.ws_bytes = need
        self.ws_shape = (bt, br)

    # ---- tensor access (reference state_dict names) ------------------------------------------------------------
    def tensor(self, name, buf=None):
        off, shape = self.spec.tensors[name]
        buf = self.params if buf is None else buf
        return buf[off:off + int(np.prod(shape))].view(shape)

    def set_tensor(self, name, array):
        t = torch.as_tensor(np.asarray(array, np.float32)).to(self.device)
        self.tensor(name).copy_(t.view(self.spec.tensors[name][1]))

    def state_dict(self):
        return {name: self.tensor(name).detach().cpu().clone() for name in self.spec.tensors}

    def load_state_dict(self, sd):
        for name in self.spec.tensors:
            self.set_tensor(name, sd[name].detach().cpu().numpy() if hasattr(sd[name], 'detach') else sd[name])

    def grad_dict(self):
        return {name: self.tensor(name, self.grads).detach().cpu().numpy().copy() for name in self.spec.tensors}

    # ---- the hot path ----------------------------------------------------------------------------------------
    def loss(self, batch, backward=False, keep_lp=False, zero_grads=True, loss_out=None, status_out=None):
        """InferenceNetworkLSTM._loss(batch) (+ backward). Returns the device loss scalar (1-element view); the
        non-finite status stays on the device in self.status_buf[0] (no host sync here)."""
        if batch.c is None:
            batch.to(self.device)
        self._ensure_workspace(batch.n_traces, batch.n_rows)
        flags = 0
        if backward:
            flags |= L.PP_LOSS_BACKWARD
            if zero_grads and not self._grads_clean:
                flags |= L.PP_LOSS_ZERO_GRADS
            self._grads_clean = False
        lp = None
        if keep_lp:
            flags |= L.PP_LOSS_KEEP_LP
            lp = torch.empty(batch.n_rows, dtype=torch.float32, device=self.device)
        if getattr(self, '_use_ops', False):     # through the operators (pyprob_amd/ops.py) instead of the direct C call
            from .ops import ops
            bdev, bhost = batch.op_tensors(self.device)
            l_, s_, lp_ = ops.ic_loss(self.params, self.grads, self.workspace, bdev, bhost, self.net_handle, flags)
            (self.loss_buf if loss_out is None else loss_out)[:1].copy_(l_)
            (self.status_buf if status_out is None else status_out)[:1].copy_(s_)
            if keep_lp:
                lp.copy_(lp_)
        else:
            rc = self.lib.pp_ic_loss(C.byref(self.net), C.byref(batch.c), self.params.data_ptr(),
                                     self.grads.data_ptr() if backward else None, self.workspace.data_ptr(), self.ws_bytes,
                                     (self.loss_buf if loss_out is None else loss_out).data_ptr(),
                                     (self.status_buf if status_out is None else status_out).data_ptr(), L.ptr(lp), flags,
                                     L.stream_ptr())
            L.check(rc, 'pp_ic_loss')
        if backward:
            self._set_active(batch)
        return (self.loss_buf[:1], lp) if keep_lp else self.loss_buf[:1]

    def _set_active(self, batch):
        key = batch.presence_key
        if key == self._active_key and self.world_size == 1 and not self.force_allreduce:
            return          # presence map already in place (it is only overwritten by the DP all-reduce)
        act = self._active_cache.get(key)
        if act is None:
            act = torch.from_numpy(self.spec.active_mask(batch.cur_counts, batch.prev_counts)).to(self.device)
            self._active_cache[key] = act
        if self.world_size == 1 and not self.force_allreduce:
            # the optimizer reads the cached map where it lies: no copy launch between the backward pass and the optimizer (a
            # ragged minibatch changes the map almost every step: 4 us of copy + 12 us of idle gaps around it)
            self._presence = act
        else:
            self.active.copy_(act)      # data parallel: the map travels in the tail of the reduced buffer
            self._presence = self.active
        self._active_key = key

    def presence(self):
        """The presence map the optimizer reads ([n_tensors] floats, > 0: the tensor had a gradient this step): the cached map of
        the last `loss(backward=True)`, or the tail of the gradient buffer when somebody else wrote it (`_active_key` None: the
        binding's optimizers, data parallel)."""
        return self._presence if self._active_key is not None else self.active

    def adam_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, zero_grads=False, skip=None, grad_scale=None):
        """optimizer.step() for optim.Adam (inference_network.py:348,496); grads are divided by world_size first
        when data-parallel (inference_network.py:324-325). zero_grads=True also performs the NEXT step's
        optimizer.zero_grad() (:486) in the same pass: the consumed gradient chunks are cleared, gradients of tensors
        without a gradient this step are zero already."""
        gs = 1.0 / self.world_size if grad_scale is None else grad_scale
        if getattr(self, '_use_ops', False):
            from .ops import ops
            ops.adam_step(self.params, self.grads, self.exp_avg, self.exp_avg_sq, self.chunk_tensor, self.presence(),
                          self.tensor_step, self.arrived, lr, beta1, beta2, eps, weight_decay, gs,
                          L.PP_ADAM_ZERO_GRADS if zero_grads else 0, skip)
        else:
            rc = self.lib.pp_adam_step(self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
                                       self.exp_avg_sq.data_ptr(), self.spec.n_params, self.chunk_tensor.data_ptr(),
                                       self.presence().data_ptr(), self.tensor_step.data_ptr(), self.arrived.data_ptr(),
                                       self.spec.n_tensors, lr, beta1, beta2, eps, weight_decay, gs,
                                       L.PP_ADAM_ZERO_GRADS if zero_grads else 0, L.ptr(skip), L.stream_ptr())
            L.check(rc, 'pp_adam_step')
        self._grads_clean = bool(zero_grads)

    def set_optimizer(self, kind='adam', larc=False, momentum=0.9):
        """Which optimizer `optimizer_step` / `train_step` run: Optimizer.ADAM / SGD / ADAM_LARC / SGD_LARC of
        InferenceNetwork._create_optimizer (inference_network.py:343-355; SGD is built with nesterov=True there). A new
        optimizer starts from empty state, like in the reference."""
        kind = str(kind).lower()
        if kind not in ('adam', 'sgd'):
            raise ValueError('unknown optimizer {!r} (adam | sgd)'.format(kind))
        self.optimizer = dict(kind=kind, larc=bool(larc), momentum=float(momentum))
        self.reset_optimizer()

    def sgd_step(self, lr, momentum=0.9, nesterov=True, weight_decay=0.0, zero_grads=False, skip=None, grad_scale=None):
        """optimizer.step() for optim.SGD(momentum, nesterov=True) (inference_network.py:350). The momentum buffer lives in
        `exp_avg` (same layout as the parameters)."""
        gs = 1.0 / self.world_size if grad_scale is None else grad_scale
        flags = L.PP_ADAM_ZERO_GRADS if zero_grads else 0
        if getattr(self, '_use_ops', False):
            from .ops import ops
            ops.sgd_step(self.params, self.grads, self.exp_avg, self.chunk_tensor, self.presence(), float(lr), float(momentum),
                         bool(nesterov), float(weight_decay), float(gs), flags, skip)
        else:
            rc = self.lib.pp_sgd_step(self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(), self.spec.n_params,
                                      self.chunk_tensor.data_ptr(), self.presence().data_ptr(), self.spec.n_tensors, lr, momentum,
                                      int(bool(nesterov)), weight_decay, gs, flags, L.ptr(skip), L.stream_ptr())
            L.check(rc, 'pp_sgd_step')
        self._grads_clean = bool(zero_grads)

    def larc_scale(self, lr, weight_decay=0.0, skip=None, trust_coefficient=0.002, clip=True, eps=1e-8, epsilon=1.0 / 16000.0):
        """The LARC wrapper's gradient rewrite (optimizer_larc.py:72-103, defaults as constructed at inference_network.py:352):
        afterwards the gradients carry the weight decay, the 1 / world_size averaging and the per-tensor adaptive factor,
        and the wrapped optimizer steps with weight_decay = 0, grad_scale = 1."""
        need = L.larc_scratch_floats(self.spec.n_params, self.spec.n_tensors)
        if getattr(self, '_larc_scratch', None) is None or self._larc_scratch.numel() < need:
            self._larc_scratch = torch.empty(need, dtype=torch.float32, device=self.device)
        if getattr(self, '_use_ops', False):
            from .ops import ops
            ops.larc_scale(self.params, self.grads, self.chunk_tensor, self.presence(), float(lr), float(weight_decay),
                           1.0 / self.world_size, float(trust_coefficient), float(eps), float(epsilon), bool(clip),
                           self._larc_scratch, skip)
        else:
            rc = self.lib.pp_larc_scale(self.params.data_ptr(), self.grads.data_ptr(), self.spec.n_params,
                                        self.chunk_tensor.data_ptr(), self.presence().data_ptr(), self.spec.n_tensors, lr, weight_decay,
                                        1.0 / self.world_size, trust_coefficient, eps, epsilon, int(bool(clip)),
                                        self._larc_scratch.data_ptr(), L.ptr(skip), L.stream_ptr())
            L.check(rc, 'pp_larc_scale')

    def optimizer_step(self, lr, weight_decay=0.0, zero_grads=False, skip=None):
        """optimizer.step() of the optimizer chosen with set_optimizer (inference_network.py:496)."""
        opt = self.optimizer
        scale = None
        if opt['larc']:
            self.larc_scale(lr, weight_decay, skip=skip)
            weight_decay, scale = 0.0, 1.0
        if opt['kind'] == 'sgd':
            self.sgd_step(lr, opt['momentum'], True, weight_decay, zero_grads, skip, grad_scale=scale)
        elif scale is None:
            self.adam_step(lr, weight_decay=weight_decay, zero_grads=zero_grads, skip=skip)
        else:
            self.adam_step(lr, weight_decay=0.0, zero_grads=zero_grads, skip=skip, grad_scale=1.0)

    def _adam_only(self, what):
        if self.optimizer['kind'] != 'adam' or self.optimizer['larc']:
            raise RuntimeError('{} runs Optimizer.ADAM; other optimizers step through train_step'.format(what))

    def train_step(self, batch, lr, weight_decay=0.0):
        """zero_grad -> loss -> backward -> [all-reduce] -> optimizer step (inference_network.py:486-496). No host sync.
        Data parallel: the non-finite flag travels in the reduced tail, so every rank skips the same batches."""
        loss = self.loss(batch, backward=True)
        if self.world_size > 1 or self.force_allreduce:
            self.allreduce_grads()
            self.optimizer_step(lr, weight_decay=weight_decay, zero_grads=_ADAM_CLEARS, skip=self.reduced_status())
        else:
            self.optimizer_step(lr, weight_decay=weight_decay, zero_grads=_ADAM_CLEARS, skip=self.status_buf)
        return loss

    def train_run(self, dataset, id_lists, lrs, weight_decay=0.0, beta1=0.9, beta2=0.999, eps=1e-8):
        """A run of training steps in ONE C call (pp_train_steps): step i trains on the traces id_lists[i] of a packed
        dataset (pyprob_amd/dataset.py) with learning rate lrs[i] - packing, upload, loss + backward and Adam per step
        without returning to Python (single rank). Returns (losses, statuses): device tensors [n_steps], not synchronised.
        The caller polymorphs first; per-address iteration counters (inference_network_lstm.py:198) are updated here."""
        self._adam_only('train_run (pp_train_steps)')
        dp = self.world_size != 1 or self.force_allreduce
        if dp and not (self.native_dp and self.lib.pp_dp_world() == self.world_size):
            raise RuntimeError('train_run under data parallelism needs the native RCCL communicator '
                               '(pyprob_amd.parallel.init_native_comm); without it the steps go through train_step')
        n_steps = len(id_lists)
        spec = self.spec
        shards, n_shards, first = dataset.native_columns(spec)
        sizes = np.fromiter((len(x) for x in id_lists), np.int64, n_steps)
        step_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        ids = np.ascontiguousarray(np.concatenate([np.asarray(x, np.int64).reshape(-1) for x in id_lists]))
        if ids.size == 0 or np.any(sizes <= 0):
            raise ValueError('empty batch')
        if np.any(ids < 0) or np.any(ids >= len(dataset.trace_len)):
            raise IndexError('trace index out of range')
        lens = dataset.trace_len[ids].astype(np.int64)
        b_max, t_max = int(sizes.max()), int(lens.max())
        r_max = int(np.add.reduceat(lens, step_off[:-1]).max())
        self._ensure_workspace(b_max, r_max)
        key = (id(spec), spec.n_tensors)
        if getattr(self, '_roles_key', None) != key:
            off, addr, role = spec.tensor_roles()
            self._roles_arrays = (off, addr, role)
            self._roles = L.pp_tensor_roles(off.ctypes.data, addr.ctypes.data, role.ctypes.data)
            self._roles_key = key
        n_slots = 16       # two halves of 8: a group of up to 8 minibatches is uploaded with one copy
        slot_words = int(self.lib.pp_train_slot_words(b_max, r_max, t_max, dataset.obs_width, len(spec.addresses),
                                                      spec.n_tensors))
        if getattr(self, '_slot_words', 0) < slot_words:
            L.check(self.lib.pp_train_sync(), 'pp_train_sync')      # pending uploads still read the old staging memory
            torch.cuda.synchronize(self.device)
            self._staging = torch.empty(n_slots * slot_words, dtype=torch.float32).pin_memory()
            self._device_batch = torch.empty(n_slots * slot_words, dtype=torch.float32, device=self.device)
            self._slot_words = slot_words
        if getattr(self, '_ring_n', 0) < n_steps:
            self._ring_n = max(n_steps, 64)
            self._loss_ring = torch.zeros(self._ring_n, dtype=torch.float32, device=self.device)
            self._status_ring = torch.zeros(self._ring_n, dtype=torch.int32, device=self.device)
        tb = L.pp_train_buffers(self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
                                self.exp_avg_sq.data_ptr(), self.chunk_tensor.data_ptr(), self.tensor_step.data_ptr(),
                                self.arrived.data_ptr(), self.workspace.data_ptr(), self.ws_bytes,
                                self._staging.data_ptr(), self._device_batch.data_ptr(), self._slot_words,
                                self._loss_ring.data_ptr(), self._status_ring.data_ptr(), spec.n_tensors, n_slots)
        if dp:      # (self.grads is the head of grads_full: the reduced tail follows the gradients)
            tb.dp_world, tb.dp_n_skip = self.world_size, len(self.dp_skip)
            for k, (off, cnt) in enumerate(self.dp_skip):
                tb.dp_skip_off[k], tb.dp_skip_cnt[k] = off, cnt
        lr = np.ascontiguousarray(lrs, np.float32).reshape(-1)
        if len(lr) != n_steps:
            raise ValueError('one learning rate per step')
        iters = np.zeros(max(len(spec.addresses), 1), np.int64)
        rc = self.lib.pp_train_steps(C.byref(self.net), C.byref(tb), C.byref(self._roles), shards, n_shards,
                                     first.ctypes.data, dataset.obs_width, ids.ctypes.data, step_off.ctypes.data, n_steps,
                                     lr.ctypes.data, beta1, beta2, eps, weight_decay, 1 if self._grads_clean else 0,
                                     iters.ctypes.data, L.stream_ptr())
        if rc != 0:
            msg = self.lib.pp_last_error().decode()
            self._grads_clean = False
            if 'address' in msg:
                raise KeyError('Address unknown by inference network ({})'.format(msg))
            raise RuntimeError('pp_train_steps failed (rc=%d): %s' % (rc, msg))
        self._grads_clean = True
        self._active_key = None        # (self.active was not used: the per-step maps live in the batch slots)
        for a, k in enumerate(iters[:len(spec.addresses)]):
            spec.addresses[a].total_train_iterations += int(k)
        return self._loss_ring[:n_steps], self._status_ring[:n_steps]

    def train_resident(self, batches, lrs, weight_decay=0.0, beta1=0.9, beta2=0.999, eps=1e-8):
        """A run of training steps over minibatches that are already in HBM (PackedBatch objects on this device) in ONE C
        call (pp_train_resident): step i = zero_grad -> loss + backward on batches[i] -> [all-reduce] -> Adam with lrs[i].
        Returns (losses, statuses): device tensors [n_steps], not synchronised. Data parallel: needs the native communicator
        (self.native_dp), every rank must pass the same number of steps."""
        n_steps = len(batches)
        self._adam_only('train_resident (pp_train_resident)')
        dp = self.world_size != 1 or self.force_allreduce
        if dp and not (self.native_dp and self.lib.pp_dp_world() == self.world_size):
            raise RuntimeError('train_resident under data parallelism needs the native RCCL communicator')
        b_max = max(b.n_traces for b in batches)
        r_max = max(b.n_rows for b in batches)
        self._ensure_workspace(b_max, r_max)
        if getattr(self, '_ring_n', 0) < n_steps:
            self._ring_n = max(n_steps, 64)
            self._loss_ring = torch.zeros(self._ring_n, dtype=torch.float32, device=self.device)
            self._status_ring = torch.zeros(self._ring_n, dtype=torch.int32, device=self.device)
        acts = []
        for b in batches:
            if b.c is None:
                b.to(self.device)
            key = b.presence_key
            act = self._active_cache.get(key)
            if act is None:
                act = torch.from_numpy(self.spec.active_mask(b.cur_counts, b.prev_counts)).to(self.device)
                self._active_cache[key] = act
            acts.append(act)
        bptr = (C.c_void_p * n_steps)(*[C.addressof(b.c) for b in batches])
        aptr = (C.c_void_p * n_steps)(*[a.data_ptr() for a in acts])
        lr = np.ascontiguousarray(lrs, np.float32).reshape(-1)
        if len(lr) != n_steps:
            raise ValueError('one learning rate per step')
        tb = L.pp_train_buffers(self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
                                self.exp_avg_sq.data_ptr(), self.chunk_tensor.data_ptr(), self.tensor_step.data_ptr(),
                                self.arrived.data_ptr(), self.workspace.data_ptr(), self.ws_bytes, None, None, 0,
                                self._loss_ring.data_ptr(), self._status_ring.data_ptr(), self.spec.n_tensors, 0)
        if dp:
            tb.dp_world, tb.dp_n_skip = self.world_size, len(self.dp_skip)
            for k, (off, cnt) in enumerate(self.dp_skip):
                tb.dp_skip_off[k], tb.dp_skip_cnt[k] = off, cnt
        rc = self.lib.pp_train_resident(C.byref(self.net), C.byref(tb), bptr, aptr, n_steps, lr.ctypes.data, beta1, beta2, eps,
                                        weight_decay, 1 if self._grads_clean else 0, L.stream_ptr())
        L.check(rc, 'pp_train_resident')
        self._grads_clean = True
        self._active_key = None
        return self._loss_ring[:n_steps], self._status_ring[:n_steps]

    def read_back(self, losses, statuses):
        """Start an asynchronous device-to-host copy of a run's (losses, statuses) and return a zero-argument function
        that waits for THAT copy only (not for work enqueued later) and returns the two numpy arrays."""
        n = losses.numel()
        pool = getattr(self, '_readback_pool', None)
        if pool is None:
            pool = self._readback_pool = []
        k = getattr(self, '_readback_next', 0)
        self._readback_next = k + 1
        if len(pool) < 4:
            pool.append(None)
        slot = k % len(pool)
        if pool[slot] is None or pool[slot][0].numel() < n:
            pool[slot] = (torch.empty(max(n, 64), dtype=torch.float32).pin_memory(),
                          torch.empty(max(n, 64), dtype=torch.int32).pin_memory(), torch.cuda.Event())
        h_loss, h_status, event = pool[slot]
        h_loss[:n].copy_(losses, non_blocking=True)
        h_status[:n].copy_(statuses, non_blocking=True)
        event.record()

        def wait():
            event.synchronize()
            return h_loss[:n].numpy().copy(), h_status[:n].numpy().copy()
        return wait

    # ---- HIP graph replay of the step (static shapes) ------------------------------------------------------------
    def capture_train_step(self, batch, lr, weight_decay=0.0):
        """Capture zero_grad -> loss -> backward -> Adam for a batch of FIXED shape and FIXED buffer addresses into a
        HIP graph (torch.cuda.CUDAGraph is the stream-capture plumbing; every node is one of this library's kernels or
        a memset). Replaying it removes the per-launch host cost of the ~40 launches of a step. New data is fed by
        copying the next minibatch into the captured batch's buffers (`PackedBatch.copy_columns_`). With data
        parallelism the all-reduce stays outside: two graphs (loss+backward | Adam)."""
        self._ensure_workspace(batch.n_traces, batch.n_rows)
        self.loss(batch, backward=True)              # warm-up outside capture: workspace, presence map, lazy init
        self.adam_step(lr, weight_decay=weight_decay)
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        if self.world_size == 1:
            with torch.cuda.graph(g1):
                self.loss(batch, backward=True)
                self.adam_step(lr, weight_decay=weight_decay)
            self._graphs = (g1, None)
        else:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                self.loss(batch, backward=True)
            with torch.cuda.graph(g2):
                self.adam_step(lr, weight_decay=weight_decay)
            self._graphs = (g1, g2)
        return self._graphs

    def replay_train_step(self):
        g1, g2 = self._graphs
        g1.replay()
        if g2 is not None:
            self.allreduce_grads()
            g2.replay()
        return self.loss_buf[:1]

    # ---- data parallel ------------------------------------------------------------------------------------
    def allreduce_grads(self):
        """ONE RCCL all-reduce (SUM) over [flat grads | presence map | loss | non-finite flag]
        (replaces the per-tensor loop of _distributed_sync_grad, inference_network.py:296-333). With `dp_skip` ranges
        (gradients that are zero on every rank by construction) the remaining pieces are reduced instead."""
        if self.native_dp and self.lib.pp_dp_world() == self.world_size and not self.grads_full.is_cpu:
            # this library's own communicator (pyprob_amd.parallel.init_native_comm): flag into the tail + ONE grouped
            # ncclAllReduce from C, no torch operator on the way (the tail keeps the SUMS, like the torch path below)
            n = self.grads.numel()
            k = len(self.dp_skip)
            off = (C.c_int64 * max(k, 1))(*[o for o, _ in self.dp_skip])
            cnt = (C.c_int64 * max(k, 1))(*[c for _, c in self.dp_skip])
            L.check(self.lib.pp_dp_reduce_grads(self.grads_full.data_ptr(), n, self.spec.n_tensors, None,
                                                self.status_buf.data_ptr(), off, cnt, k, None, None, L.stream_ptr()),
                    'pp_dp_reduce_grads')
            return
        from .parallel import allreduce_flat_
        self.status_tail.copy_(self.status_buf[:1])          # int32 flag -> float, into the reduced tail
        if self.dp_overlap:
            # the bucketed order of enable_dp_overlap on torch.distributed: the early ranges go out first as asynchronous
            # collectives, the rest follows, the early ones are waited for last (same pieces, same order on every rank)
            import torch.distributed as dist
            from .parallel import allreduce_pieces_
            early = [dist.all_reduce(self.grads_full[o:o + c], async_op=True) for o, c in self.dp_overlap]
            pieces, pos = [], 0
            for off, cnt in sorted(list(self.dp_skip) + list(self.dp_overlap)):
                if off > pos:
                    pieces.append((pos, off))
                pos = off + cnt
            pieces.append((pos, self.grads_full.numel()))
            allreduce_pieces_([self.grads_full[a:b] for a, b in pieces if b > a])
            for work in early:
                work.wait()
            return
        if not self.dp_skip:
            allreduce_flat_(self.grads_full)
            return
        # the pieces around the skipped ranges go out as ONE collective launch without staging copies: RCCL aggregates
        # the all-reduces issued between ncclGroupStart / ncclGroupEnd (torch's coalescing manager). (Staging the pieces
        # into one contiguous buffer, as before, cost eight ~1 MB device copies per step: ~30 us of a 125 us step.)
        pieces, pos = [], 0
        for off, cnt in self.dp_skip:
            if off > pos:
                pieces.append((pos, off))
            pos = off + cnt
        pieces.append((pos, self.grads_full.numel()))
        from .parallel import allreduce_pieces_
        allreduce_pieces_([self.grads_full[a:b] for a, b in pieces])

    def reduced_status(self):
        """The all-reduced non-finite flag as the int32 word pp_adam_step's `skip` reads: a sum of 0.0 / 1.0 floats is
        non-zero (any bit set) exactly when some rank flagged its batch."""
        return self.status_tail.view(torch.int32)

    def skip_recurrent_weights(self, enable=True):
        """Data-parallel runs over a dataset in which EVERY trace has one controlled variable (GaussianUnknownMean): no
        time step has a predecessor, so dL/dW_hh is exactly zero on every rank, every step (h_0 = 0,
        inference_network_lstm.py:186-187). W_hh is 2/3 (H = 512) to 3/4 (H = 1024) of the flat gradient: leaving its
        range out of the all-reduce changes no bit of the result (Adam still steps the tensor with its zero gradient,
        like the reference). The caller asserts the property for the whole dataset - it must hold on all ranks."""
        self.dp_skip = []
        if enable and not self.spec.feedforward:
            off, shape = self.spec.tensors['_layers_lstm.weight_hh_l0']
            n = int(np.prod(shape))
            self.dp_skip = [(off, ((n + 1023) // 1024) * 1024)]

    def enable_dp_overlap(self, enable=True):
        """Data parallel: reduce the first LSTM layer's gradients (W_ih, W_hh unless it is skipped, both bias vectors - 0.4 to
        0.7 of what a step exchanges) EARLY - the backward pass issues its weight-gradient launch in two parts and the ranges'
        all-reduce runs on a side stream under the second part (pp_dp_overlap, csrc/dp.hip; the reference's bucketed
        `_distributed_sync_grad`, inference_network.py:300-325). Every rank must make the same call (it follows
        `agree_skip_recurrent`, which all ranks settle together). Returns the ranges."""
        # NOT the default: on a one-rank RCCL group the two-part launch + the two cross-stream event waits cost the 67 us step
        # ~26 us (profiles/r06e_dp_overlap_one_rank.txt) - more than the <= 8-10 us of an 8-rank exchange the second part can
        # cover. PP_DP_OVERLAP=1 selects it (larger networks / slower links, where the early ranges' collective is long)
        self.dp_overlap = []
        if enable and not self.spec.feedforward and os.environ.get('PP_DP_OVERLAP', '0') == '1':
            names = ['_layers_lstm.weight_ih_l0', '_layers_lstm.weight_hh_l0', '_layers_lstm.bias_ih_l0', '_layers_lstm.bias_hh_l0']
            pieces = []
            for n in names:
                off, shape = self.spec.tensors[n]
                cnt = ((int(np.prod(shape)) + 1023) // 1024) * 1024
                if any(o <= off and off + cnt <= o + c for o, c in self.dp_skip):
                    continue                                   # (zero on every rank: it leaves the exchange altogether)
                if pieces and pieces[-1][0] + pieces[-1][1] == off:
                    pieces[-1] = (pieces[-1][0], pieces[-1][1] + cnt)
                else:
                    pieces.append((off, cnt))
            self.dp_overlap = pieces[:2] if len(pieces) <= 2 else []
        if self.native_dp and not self.grads_full.is_cpu and self.lib.pp_dp_world() >= 1:
            k = len(self.dp_overlap)
            off = (C.c_int64 * max(k, 1))(*[o for o, _ in self.dp_overlap])
            cnt = (C.c_int64 * max(k, 1))(*[c for _, c in self.dp_overlap])
            L.check(self.lib.pp_dp_overlap(off, cnt, k), 'pp_dp_overlap')
        return list(self.dp_overlap)

    def agree_skip_recurrent(self, single_statement):
        """COLLECTIVE (every rank calls it): leave W_hh out of the gradient all-reduce iff the data of EVERY rank has one
        controlled variable per trace. `single_statement` is this rank's own finding, read from its data (the trace lengths
        of its dataset shard / its resident minibatches) - not a caller's assertion about the other ranks; one MIN-all-reduce
        of a flag settles it, so all ranks issue all-reduces of the same layout. A per-step decision on the device is not
        possible without a host round trip per step: the piece list of an RCCL call is a host-side argument."""
        flag = 1 if single_statement else 0
        if self.world_size > 1:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                dev = self.device if dist.get_backend() == 'nccl' else 'cpu'
                t = torch.tensor([flag], dtype=torch.int32, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                flag = int(t.item())
        self.skip_recurrent_weights(bool(flag))
        if self.world_size > 1 or self.force_allreduce:
            self.enable_dp_overlap(True)       # (after the skip is settled: a skipped W_hh is not part of the early ranges)
        return bool(flag)

    def broadcast_params(self):
        """_distributed_sync_parameters (inference_network.py:290-294) as one broadcast of the flat buffer."""
        import torch.distributed as dist
        dist.broadcast(self.params, 0)

    def reset_optimizer(self):
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.tensor_step.zero_()
        self.arrived.view(-1, L.PP_ADAM_SCRATCH)[:, L.PP_ADAM_SEEN] = 0     # moments are zero again (pp_adam_step)

    def moments_written(self):
        """The caller filled exp_avg / exp_avg_sq itself (checkpoint load): pp_adam_step must not assume zero moments for
        tensors that have not seen a gradient in THIS process (PP_ADAM_SEEN, include/pyprob_amd.h)."""
        self.arrived.view(-1, L.PP_ADAM_SCRATCH)[:, L.PP_ADAM_SEEN] = 1
