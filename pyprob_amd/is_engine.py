"""Lock-step importance sampling with the inference network: N particles advance through the model's controlled
`sample` statements together.

The reference runs one particle at a time (pyprob/model.py:59) and, inside it, one batch-1 `_infer_step` per
`pyprob.sample` (pyprob/state.py:203-219, pyprob/nn/inference_network_lstm.py:82-134). Here every statement is one
C-ABI call for all particles of this rank: LSTM step + proposal head + device-side sampling + log q, followed by
fused prior/likelihood log-prob kernels that accumulate the per-particle log-weight
(log w = sum_t [log p(v_t) - log q(v_t)] + sum_j log p(y_j | .), pyprob/trace.py:123-125).
"""
import ctypes as C
import os
import time

import numpy as np
import torch

from . import lib as L
from .ops import ops


class ISRunner:
    def __init__(self, engine):
        self.eng = engine
        self.lib = engine.lib
        self.dev = engine.device
        self._e_obs = torch.zeros(self.e_obs_floats(), dtype=torch.float32, device=self.dev)
        self.ws = None
        self.ws_bytes = 0
        self.n = 0
        self.offset = 0
        self._consts = {}
        self._stats = torch.zeros(8, dtype=torch.float64, device=self.dev)
        self._st = None        # stream of the current posterior call (begin); None = the default stream
        self._stats_scratch = torch.zeros(L.PP_IS_STATS_SCRATCH, dtype=torch.float64, device=self.dev)

    def e_obs_floats(self):
        """Size of the embedding row buffer: pp_is_first_statement writes the (at most 8) observation copies at
        `e_out[round4(e_obs) + lane]` (include/pyprob_amd.h) - round4(e_obs) + 8 floats, not e_obs + 8."""
        return ((self.eng.spec.e_obs + 3) & ~3) + 8

    def _pins(self, k):
        """Pinned host staging of a posterior call: the observation vector (read in place by pp_is_first_statement, or copied
        to its device twin) and the statistics record pp_is_fused writes (polled, not copied back)."""
        if getattr(self, '_obs_np', None) is None or self._obs_pin.numel() < k:
            self._obs_pin = torch.zeros(max(k, 16), dtype=torch.float32).pin_memory()
            self._obs_np = self._obs_pin.numpy()
            self._obs_dev = torch.zeros(max(k, 16), dtype=torch.float32, device=self.dev)
        if getattr(self, '_stats_np', None) is None:
            self._stats_pin = torch.zeros(8, dtype=torch.float64).pin_memory()
            self._stats_np = self._stats_pin.numpy()

    def _const(self, v):
        """1-element device tensor holding v (cached: no allocation / H2D copy in the steady state)."""
        t = self._consts.get(v)
        if t is None:
            if len(self._consts) > 4096:      # (observed values of many posterior calls: keep the cache bounded)
                self._consts.clear()
            t = torch.tensor([v], dtype=torch.float32, device=self.dev)
            self._consts[v] = t
        return t

    def _ensure_ws(self, n):
        need = self.lib.pp_is_workspace_bytes(C.byref(self.eng.net), n)
        if need > self.ws_bytes:
            self.ws = torch.zeros(need, dtype=torch.uint8, device=self.dev)      # (zero-filled: the header's contract)
            self.ws_bytes = need

    def init(self, observe):
        """InferenceNetwork._infer_init(observe) (inference_network.py:141-148): embed the observation once."""
        vals = np.asarray(observe, np.float32).reshape(-1)
        if vals.size != self.eng.spec.obs_width:
            raise ValueError('observe has %d values, the network expects %d' % (vals.size, self.eng.spec.obs_width))
        self._ensure_ws(1)
        if self.dev.type == 'cuda':
            # pinned staging + the C ABI directly into the runner's embedding row (what run_plan does): no pageable upload, no
            # zero-fill launch and no operator dispatch per posterior call
            if torch.cuda.current_device() != (self.dev.index or 0):
                torch.cuda.set_device(self.dev)
            k = int(vals.size)
            self._pins(k)
            self._obs_np[:k] = vals
            if self._e_obs.numel() != self.e_obs_floats() or self._e_obs.device != self.dev:
                self._e_obs = torch.zeros(self.e_obs_floats(), dtype=torch.float32, device=self.dev)
            self._st = L.stream_ptr()
            # The embedding launch is DEFERRED to the first statement (round 6): a trace's first statement on a network
            # pp_is_first_statement takes runs embedding + LSTM row + proposal layer as ONE launch that reads the observation
            # from the pinned staging in place (what a launch-plan replay does) - also when forward() runs in every call
            # (PP_IS_PLAN=0, a program without a plan, pyprob as the host). Anything else that needs the embedding first
            # (`_ensure_init`) launches pp_is_init as before.
            self._init_pending = k
            if os.environ.get('PP_IS_LAZY_INIT', '1') == '0':
                self._ensure_init()
            return
        obs = torch.as_tensor(vals).to(self.dev)
        self._e_obs = ops.is_init(self.eng.params, self.ws, self.eng.net_handle, obs)

    _init_pending = 0

    @property
    def e_obs(self):
        """The observe embedding row (+ the device copies of the observation behind it) - computed by now if it was deferred."""
        self._ensure_init()
        return self._e_obs

    @e_obs.setter
    def e_obs(self, value):
        self._e_obs = value

    def _ensure_init(self):
        """The observe embedding of the call's observation exists on the device after this (see `init`)."""
        k, self._init_pending = self._init_pending, 0
        if k:
            self._obs_dev.copy_(self._obs_pin, non_blocking=True)
            L.check(self.lib.pp_is_init(C.byref(self.eng.net), self.eng.params.data_ptr(), self._obs_dev.data_ptr(),
                                        self._e_obs.data_ptr(), self.ws.data_ptr(), self.ws_bytes, self._st), 'pp_is_init')

    def begin(self, n, offset=0):
        """Start n traces in lock step (state._begin_trace, state.py:339-345): LSTM state is reset by the first step."""
        H = max(self.eng.spec.lstm_dim, 1)       # (FeedForward network: no LSTM state, 1-wide placeholders)
        D = self.eng.spec.lstm_depth             # nn.LSTM(I, H, depth): state of every layer, [depth, n, H]
        if n * H >= 2 ** 32:
            # the statement kernels address a particle's state row with 32-bit element offsets (is_step_fused.hip)
            raise ValueError('at most 2^32 / lstm_dim - 1 = %d particles per call and device (got %d): shard the call' % (2 ** 32 // H - 1, n))
        if n != self.n:
            self.h = torch.empty(D, n, H, dtype=torch.float32, device=self.dev)
            self.c = torch.empty(D, n, H, dtype=torch.float32, device=self.dev)
            self.n = n
        self.prev_value = None
        self.last_value = None
        self.state_rows = 1
        self.offset = int(offset)
        self._ensure_ws(n)
        # the stream of this posterior call, looked up once (torch.cuda.current_stream() costs ~6 us of device-index resolution
        # per lookup: 59 lookups per call of the Marsaglia program, profiles/r05d_gumm_cprofile.txt) - the direct C-ABI calls
        # below (partition, rows terms, whole statements) use it
        if self.dev.type == 'cuda':
            if torch.cuda.current_device() != (self.dev.index or 0):
                torch.cuda.set_device(self.dev)
            self._st = L.stream_ptr()
        else:
            self._st = None

    def step(self, addr_id, prev_addr_id, prior, value_in=None, seed=0):
        """One controlled sample statement for all particles. prior: device tensor [1,2] (shared) or [n,2].
        Returns (values [n], log q [n]) as device tensors."""
        n = self.n
        self._ensure_init()
        value, logq = ops.is_step(self.eng.params, self.ws, self.eng.net_handle, int(addr_id),
                                  -1 if prev_addr_id is None else int(prev_addr_id), n, self._e_obs, self.prev_value, prior,
                                  self.h, self.c, self.state_rows, value_in, int(seed), self.offset)
        self.state_rows = 1 if prev_addr_id is None else n   # see include/pyprob_amd.h
        self.prev_value = value
        self.last_value = value
        return value, logq

    def step_net(self, addr_id, prev_addr_id):
        """The network part of `step` only (LSTM step + proposal layer): the head outputs stay in the workspace and the
        draw happens in `fused`, together with the log-weight terms of the statements that follow."""
        k = self._init_pending
        if k and prev_addr_id is None and k <= 8 and self.dev.type == 'cuda' and \
                self.lib.pp_is_first_statement_supported(C.byref(self.eng.net), int(addr_id)):
            # embedding + one-row LSTM step + proposal layer in ONE launch, the observation read from pinned memory in place
            self._init_pending = 0
            L.check(self.lib.pp_is_first_statement(C.byref(self.eng.net), self.eng.params.data_ptr(), self._obs_pin.data_ptr(),
                                                   int(addr_id), self._e_obs.data_ptr(), self.h.data_ptr(), self.c.data_ptr(),
                                                   self.ws.data_ptr(), self.ws_bytes, self._st), 'pp_is_first_statement')
            self.state_rows = 1
            return
        self._ensure_init()
        ops.is_step_net(self.eng.params, self.ws, self.eng.net_handle, int(addr_id),
                        -1 if prev_addr_id is None else int(prev_addr_id), self.n, self._e_obs, self.prev_value, self.h, self.c,
                        self.state_rows)
        self.state_rows = 1 if prev_addr_id is None else self.n

    def fused(self, addr_id, prior, terms, value, lw, overwrite, seed=0, stats=False):
        """ONE pass over the particles: [draw from the shared proposal of `addr_id` (None: values are given), - log q,]
        lw (+)= sum of `terms`, [importance statistics]. terms = [((kind, p0, s0, p1, s1), x, scale, flags)], flags bit
        0 / 1 / 2: p0 / p1 / x is the particle's value. Returns the statistics dict when asked for."""
        if stats and self.dev.type == 'cuda' and value.is_contiguous() and lw.is_contiguous() and value.dtype == torch.float32 \
                and lw.dtype == torch.float32 and lw.numel() == value.numel():
            return self._fused_stats_direct(addr_id, prior, terms, value, lw, overwrite, seed)
        kinds, p0s, s0s, p1s, s1s, xs, scales, flags = [], [], [], [], [], [], [], []
        for (kind, p0, s0, p1, s1), x, scale, fl in terms:
            kinds.append(int(kind)); p0s.append(p0); s0s.append(int(s0)); p1s.append(p1); s1s.append(int(s1))
            xs.append(x); scales.append(float(scale)); flags.append(int(fl))
        out = ops.is_fused(self.ws, self.eng.net_handle, -1 if addr_id is None else int(addr_id),
                           None if prior is None else prior.reshape(-1), kinds, p0s, s0s, p1s, s1s, xs, scales, flags, value, lw,
                           bool(overwrite), int(seed), self.offset, self._stats_scratch if stats else None)
        return self._stats_dict(out) if stats else None

    def _fused_stats_direct(self, addr_id, prior, terms, value, lw, overwrite, seed):
        """`fused(..., stats=True)` straight through the C ABI with the statistics record in PINNED host memory (the count is
        stored last behind a system-scope fence and polled here): no operator dispatch, no device-to-host copy with its
        synchronisation - the last flush of every lock-step posterior call ends in this."""
        n = value.numel()
        count = len(terms)
        arr = (L.pp_lw_term * max(count, 1))()
        fl = (C.c_int32 * max(count, 1))()
        for q, ((kind, p0, s0, p1, s1), x, scale, flags) in enumerate(terms):
            for t, what in ((p0, 'p0'), (p1, 'p1'), (x, 'x')):
                # (a term tensor is one shared value or one value per particle; Categorical: one probability row or n rows)
                if t is not None and int(kind) != 5 and t.numel() not in (1, n):
                    raise RuntimeError('is_fused: term %d: %s has %d elements (1 or n = %d)' % (q, what, t.numel(), n))
                if t is not None and (t.device != value.device or t.dtype != torch.float32):
                    raise RuntimeError('is_fused: term %d: %s must be a float32 tensor on %s' % (q, what, value.device))
            arr[q].kind = int(kind)
            arr[q].p0, arr[q].p1, arr[q].x = L.ptr(p0), L.ptr(p1), L.ptr(x)
            arr[q].p0_stride, arr[q].p1_stride = int(s0), int(s1)
            arr[q].x_stride = 0 if (x is None or x.numel() == 1) else 1
            arr[q].scale = float(scale)
            fl[q] = int(flags)
        self._pins(1)
        snp = self._stats_np
        snp[5] = -1.0
        pr = None if prior is None else prior.reshape(-1)
        st = self._st if self._st is not None else L.stream_ptr()
        L.check(self.lib.pp_is_fused(C.byref(self.eng.net), -1 if addr_id is None else int(addr_id), n, L.ptr(pr), arr, fl, count,
                                     value.data_ptr(), lw.data_ptr(), 1 if overwrite else 0, int(seed), int(self.offset),
                                     self._stats_pin.data_ptr(), self._stats_scratch.data_ptr(), self.ws.data_ptr(), self.ws_bytes,
                                     st), 'pp_is_fused')
        spins = 0
        while snp[5] < 0.0:
            spins += 1
            if spins > 2000000:
                torch.cuda.synchronize(self.dev)
                if snp[5] < 0.0:
                    raise L.HipLibraryError('pp_is_fused: the statistics never arrived in host memory')
        return self._stats_dict(snp)

    def run_plan(self, plan, obs_values, n, offset, seed):
        """Replay of a recorded single-statement posterior call (Model._replay_lockstep_plan) straight through the C ABI, with no
        copy in either direction: the observation is written into PINNED host memory that the first kernel reads in place
        (pp_is_first_statement: observe embedding, the one-row LSTM step and the proposal layer in one launch), then
        pp_is_fused, whose statistics land in pinned host memory too - the count last, behind a system-scope fence - and are
        polled there instead of being copied back. Networks pp_is_first_statement does not take (and PP_IS_FIRST=0) go through
        a staged copy + pp_is_init + pp_is_step_net as before. plan['c'] caches the term array.
        Returns (values, log-weights, statistics dict)."""
        lib, net = self.lib, C.byref(self.eng.net)
        if torch.cuda.current_device() != (self.dev.index or 0):
            torch.cuda.set_device(self.dev)
        k = len(obs_values)
        self._pins(k)
        self._obs_np[:k] = obs_values
        self._init_pending = 0          # (this call issues its own first statement)
        self.begin(n, offset)
        st = self._st
        params, ws = self.eng.params.data_ptr(), self.ws.data_ptr()
        # (asked per call, not cached in the plan: the C side re-reads PP_IS_FIRST on every call, a plan recorded under another
        # setting must take the staged path instead of failing with PP_EINVAL - ADVICE r05)
        first = plan['first'] = bool(k <= 8 and lib.pp_is_first_statement_supported(net, plan['addr']))
        if self._e_obs.numel() != self.e_obs_floats():
            self._e_obs = torch.zeros(self.e_obs_floats(), dtype=torch.float32, device=self.dev)
        if first:
            L.check(lib.pp_is_first_statement(net, params, self._obs_pin.data_ptr(), plan['addr'], self._e_obs.data_ptr(),
                                              self.h.data_ptr(), self.c.data_ptr(), ws, self.ws_bytes, st), 'pp_is_first_statement')
            obs_base = self._e_obs.data_ptr() + 4 * ((self.eng.spec.e_obs + 3) & ~3)      # the device copies behind the embedding
        else:
            self._obs_dev.copy_(self._obs_pin, non_blocking=True)
            L.check(lib.pp_is_init(net, params, self._obs_dev.data_ptr(), self._e_obs.data_ptr(), ws, self.ws_bytes, st), 'pp_is_init')
            L.check(lib.pp_is_step_net(net, params, plan['addr'], -1, n, self._e_obs.data_ptr(), None, self.h.data_ptr(),
                                       self.c.data_ptr(), 1, ws, self.ws_bytes, st), 'pp_is_step_net')
            obs_base = self._obs_dev.data_ptr()
        self.state_rows = 1
        buf = torch.empty(2 * n, dtype=torch.float32, device=self.dev)      # values | log-weights: one allocation per call
        values, lw = buf[:n], buf[n:]
        c = plan.get('c')
        if c is None or c['obs_base'] != obs_base:
            terms = [(plan['prior_term'], None, 1.0, 4)] + [((kind, (a[1] if a[0] == 'const' else None), s0,
                                                               (b[1] if b[0] == 'const' else None), s1), xsrc, scale,
                                                              (1 if a[0] == 'value' else 0) | (2 if b[0] == 'value' else 0))
                                                             for kind, a, s0, b, s1, xsrc, scale in plan['terms']]
            arr = (L.pp_lw_term * len(terms))()
            fl = (C.c_int32 * len(terms))()
            for q, ((kind, p0, s0, p1, s1), xsrc, scale, flags) in enumerate(terms):
                arr[q].kind, arr[q].p0_stride, arr[q].p1_stride, arr[q].x_stride = int(kind), int(s0), int(s1), 0
                arr[q].p0, arr[q].p1 = L.ptr(p0), L.ptr(p1)
                arr[q].x = None if xsrc is None else obs_base + 4 * plan['obs_index'][xsrc[1]]
                arr[q].scale = float(scale)
                fl[q] = int(flags)
            c = plan['c'] = dict(arr=arr, fl=fl, count=len(terms), obs_base=obs_base, prior=plan['prior'].reshape(-1))
        snp = self._stats_np
        snp[5] = -1.0                                                        # (the kernel stores the count last)
        L.check(lib.pp_is_fused(net, plan['addr'], n, c['prior'].data_ptr(), c['arr'], c['fl'], c['count'], values.data_ptr(),
                                lw.data_ptr(), 1, int(seed), int(self.offset), self._stats_pin.data_ptr(),
                                self._stats_scratch.data_ptr(), ws, self.ws_bytes, st), 'pp_is_fused')
        self.prev_value = self.last_value = values
        # a short busy poll (the record arrives ~40 us after the launch), then yield the GIL between polls, bounded by TIME: after
        # one second let the runtime report what happened (also the path of non-coherent host allocations, HIP_HOST_COHERENT=0)
        spins, deadline = 0, None
        while snp[5] < 0.0:
            spins += 1
            if spins > 4000:
                now = time.perf_counter()
                if deadline is None:
                    deadline = now + 1.0
                elif now > deadline:
                    torch.cuda.synchronize(self.dev)
                    if snp[5] < 0.0:
                        raise L.HipLibraryError('pp_is_fused: the statistics never arrived in host memory')
                time.sleep(0)
        return values, lw, self._stats_dict(snp)

    PRIOR_KIND = {'Normal': 0, 'Uniform': 1}

    def whole_statement_ok(self, addr_id, prev_addr_id, m, dist_name, prior):
        """Can `statement_rows` take this statement (fused statement kernel, shared Normal / Uniform prior pair)?"""
        return (prev_addr_id is not None and self.dev.type == 'cuda' and
                dist_name in self.PRIOR_KIND and prior is not None and prior.numel() == 2 and self.prev_value is not None and
                self.prev_value.numel() == self.n and
                bool(self.lib.pp_is_step_fused_supported(C.byref(self.eng.net), int(addr_id), int(m))))

    def statement_rows(self, rows, addr_id, prev_addr_id, prior, values_full, lw_full, dist_name, seed=0):
        """The whole statement for the particles `rows` (None: all) in one launch: previous values read at the rows, value
        written to values_full[rows], lw_full[rows] += log p(v) - log q(v) (pp_is_statement_rows)."""
        m = self.n if rows is None else int(rows.numel())
        self._ensure_init()
        state_rows = self.state_rows
        if rows is not None and state_rows == 1 and self.n > 1:
            self.h[:, 1:] = self.h[:, :1]      # the shared first-statement state (row 0) becomes per-particle
            self.c[:, 1:] = self.c[:, :1]
            state_rows = self.n
        self._ensure_ws(m)
        # (straight through the C ABI: whole_statement_ok has checked the shapes, the tensors are this executor's own; state_rows =
        # the rows of one layer of the state buffer: a path of ONE particle is not the shared first state, include/pyprob_amd.h)
        L.check(self.lib.pp_is_statement_rows(C.byref(self.eng.net), self.eng.params.data_ptr(), int(addr_id), int(prev_addr_id), m,
                                              self._e_obs.data_ptr(), self.prev_value.data_ptr(), prior.data_ptr(), 0,
                                              self.h.data_ptr(), self.c.data_ptr(), 1 if state_rows == 1 else self.n,
                                              L.ptr(rows if self.n > 1 else None),
                                              values_full.data_ptr(), lw_full.data_ptr(), self.PRIOR_KIND[dist_name], int(seed),
                                              int(self.offset), self.ws.data_ptr(), self.ws_bytes, self._st), 'pp_is_statement_rows')
        torch.autograd.graph.increment_version(values_full)      # written by the kernel: memoised results of it are stale
        self.state_rows = self.n
        self.prev_value = self.last_value = values_full

    def step_rows(self, rows, addr_id, prev_addr_id, prior, seed=0, prior_compact=False):
        """The same statement for a SUBSET of the particles (a diverged control-flow path): the rows' LSTM state and
        previous values are gathered into a compact batch, stepped, and scattered back. rows: int64 device tensor; prior:
        [1, 2], or one row per particle ([n, 2]; prior_compact: one row per entry of `rows`)."""
        m = int(rows.numel())
        self._ensure_init()
        if (prev_addr_id is not None and self.dev.type == 'cuda' and self.n > 1 and
                self.lib.pp_is_step_fused_supported(C.byref(self.eng.net), int(addr_id), m)):
            # the fused statement kernel reads and writes the rows' state in place through the index list
            if self.state_rows == 1 and self.n > 1:
                self.h[:, 1:] = self.h[:, :1]
                self.c[:, 1:] = self.c[:, :1]
                self.state_rows = self.n
            prev = self.prev_value.index_select(0, rows)
            if prior is not None and prior.shape[0] != 1 and not prior_compact:
                prior = prior.index_select(0, rows).contiguous()
            self._ensure_ws(m)
            return ops.is_step_rows(self.eng.params, self.ws, self.eng.net_handle, int(addr_id), int(prev_addr_id), m, self._e_obs,
                                    prev, prior, self.h, self.c, self.n, rows, None, int(seed), self.offset)
        if self.state_rows == 1 and self.n > 1 and prev_addr_id is not None:
            self.h[:, 1:] = self.h[:, :1]      # the shared first-statement state (row 0) becomes per-particle
            self.c[:, 1:] = self.c[:, :1]
            self.state_rows = self.n
        h = self.h.index_select(1, rows)
        c = self.c.index_select(1, rows)
        prev = None if (prev_addr_id is None or self.prev_value is None) else self.prev_value.index_select(0, rows).contiguous()
        if prior is not None and prior.shape[0] != 1 and not prior_compact:
            prior = prior.index_select(0, rows).contiguous()
        self._ensure_ws(m)
        value, logq = ops.is_step(self.eng.params, self.ws, self.eng.net_handle, int(addr_id),
                                  -1 if prev_addr_id is None else int(prev_addr_id), m, self._e_obs, prev, prior, h, c,
                                  1 if prev_addr_id is None else m, None, int(seed), self.offset)
        if prev_addr_id is None:     # the call left the (shared) new state in row 0 of every layer
            h = h[:, :1].expand(-1, m, -1)
            c = c[:, :1].expand(-1, m, -1)
        self.h.index_copy_(1, rows, h)
        self.c.index_copy_(1, rows, c)
        return value, logq

    # ---- log-weight terms ---------------------------------------------------------------------------------
    def dist_term(self, distribution, n=None):
        """(kind, p0, p0_stride, p1, p1_stride) of pp_logweight_accumulate for a prior / likelihood distribution object
        (duck-typed: .name and the parameter attributes of pyprob/distributions/*.py); None if the family has no device
        kernel. Parameters may be shared (one element) or per particle (n elements)."""
        with torch._C.DisableTorchFunctionSubclass():     # (metadata only: no Python dispatch per attribute of a ParticleTensor)
            return self._dist_term(distribution)

    def _dist_term(self, distribution):
        dev = self.dev

        def t(v):
            if isinstance(v, (int, float)) or (torch.is_tensor(v) and v.device.type == 'cpu' and v.numel() == 1):
                return self._const(float(v))          # cached device scalar: no host-to-device copy per statement
            if torch.is_tensor(v) and v.dtype == torch.float32 and v.device == dev and v.dim() == 1 and v.is_contiguous():
                return v.as_subclass(torch.Tensor)    # (per-particle parameters of a lock-step run: one call, not five)
            return torch.as_tensor(v, dtype=torch.float32).as_subclass(torch.Tensor).reshape(-1).to(dev).contiguous()

        def s(v):
            return 0 if v.numel() == 1 else 1
        name = distribution.name
        if name == 'Normal':
            p0, p1 = t(distribution.mean), t(distribution.stddev)
            return 0, p0, s(p0), p1, s(p1)
        if name == 'Uniform':
            p0, p1 = t(distribution.low), t(distribution.high)
            return 1, p0, s(p0), p1, s(p1)
        if name == 'Poisson':
            p0 = t(distribution.rate)
            return 3, p0, s(p0), None, 0
        if name == 'Bernoulli':
            p0 = t(distribution.probs)
            return 4, p0, s(p0), None, 0
        if name == 'Categorical':
            C_ = int(distribution.num_categories)
            p0 = t(distribution.probs)
            return 5, p0, (0 if p0.numel() == C_ else C_), None, C_
        return None

    def log_prob(self, term, x, n=None):
        """log_prob(dist; x) per particle as a device tensor [n] (no accumulation)."""
        kind, p0, s0, p1, s1 = term
        n = int(x.numel()) if n is None else n
        return ops.log_prob(int(kind), p0, int(s0), p1, int(s1), x, n)

    def _term_accumulate(self, term, x, scale, lw):
        kind, p0, s0, p1, s1 = term
        ops.logweight_terms(lw, [int(kind)], [p0], [int(s0)], [p1], [int(s1)], [x], [float(scale)], False)

    def accumulate_masked(self, lw, kind, p0, p1, x, active, scale=1.0, term=None):
        """accumulate() for the active particles of a diverged path: the term is evaluated for every particle (stale
        entries of inactive particles may be anything) and added where `active` (None = everywhere)."""
        if term is None:
            term = (kind, p0, 0 if p0.numel() == 1 else 1, p1, 0 if p1.numel() == 1 else 1)
        if active is None:
            return self._term_accumulate(term, x, scale, lw)
        lp = self.log_prob(term, x, lw.numel())
        lw.add_(torch.where(active, lp, torch.zeros_like(lp)), alpha=float(scale))

    # ---- the particles of a control-flow path as a row list (LockStepState.by_rows): direct C-ABI calls ------------------------
    def partition_launch(self, cond, rows, m):
        """A branch: split the path's rows (None: particles 0..m-1) by the bool [n] condition - the launches only; the counts
        are read by partition_read (the one synchronisation of a branch)."""
        buf = torch.empty(2 * max(m, 1), dtype=torch.int64, device=self.dev)
        need = (m + 1023) // 1024 + 1
        scratch = getattr(self, '_part_scratch', None)
        if scratch is None or scratch.numel() < need:
            scratch = self._part_scratch = torch.empty(max(need, 1024), dtype=torch.int32, device=self.dev)
        if m > 0 and self.dev.type == 'cuda' and os.environ.get('PP_IS_PART_POLL', '1') != '0':
            # the decision comes back through PINNED host memory the kernel writes in place (counts, then a sequence word behind a
            # system-scope fence) and the host polls: no 8-byte device-to-host copy, whose blocking call cost every branch ~45 us
            ring = getattr(self, '_part_ring', None)
            if ring is None:
                self._part_pin = torch.zeros(64 * 4, dtype=torch.int32).pin_memory()
                ring = self._part_ring = self._part_pin.numpy().reshape(64, 4)
                self._part_seq = 0
            self._part_seq = self._part_seq % 0x3fffffff + 1
            slot = self._part_seq & 63          # (nested paths keep a few partitions in flight: one slot each)
            ring[slot, 2] = 0
            L.check(self.lib.pp_partition_rows_polled(cond.data_ptr(), L.ptr(rows), int(m), buf.data_ptr(), buf.data_ptr() + 8 * m,
                                                      self._part_pin.data_ptr() + 16 * slot, self._part_seq, scratch.data_ptr(),
                                                      self._st), 'pp_partition_rows_polled')
            return buf, (ring[slot], self._part_seq), m, cond, rows
        counts = torch.empty(2, dtype=torch.int32, device=self.dev)
        L.check(self.lib.pp_partition_rows(cond.data_ptr(), L.ptr(rows), int(m), buf.data_ptr(), buf.data_ptr() + 8 * m,
                                           counts.data_ptr(), scratch.data_ptr(), self._st), 'pp_partition_rows')
        return buf, counts, m, cond, rows      # (cond / rows stay alive until the kernels have run)

    def partition_read(self, handle):
        """(rows where the condition holds, the other rows, their counts) - ascending int64 device vectors."""
        buf, counts, m = handle[:3]
        if isinstance(counts, tuple):
            slot, seq = counts
            spins, deadline = 0, None
            while slot[2] != seq:
                spins += 1
                if spins > 4000:                 # a long statement kernel is still running: yield between polls, bounded by time
                    now = time.perf_counter()
                    if deadline is None:
                        deadline = now + 30.0
                    elif now > deadline:
                        torch.cuda.synchronize(self.dev)
                        if slot[2] != seq:
                            raise L.HipLibraryError('pp_partition_rows_polled: the counts never arrived in host memory')
                    time.sleep(0)
            n_true, n_false = int(slot[0]), int(slot[1])
        else:
            n_true, n_false = counts.tolist()
        return buf[:n_true], buf[m:m + n_false], n_true, n_false

    def partition(self, cond, rows, m):
        return self.partition_read(self.partition_launch(cond, rows, m))

    def accumulate_rows(self, lw, term, x, rows, scale):
        """lw[rows] += scale * log_prob(term; x[rows]) - one launch on the path's rows (pp_logweight_accumulate_rows)."""
        kind, p0, s0, p1, s1 = term
        x = x.as_subclass(torch.Tensor) if type(x) is not torch.Tensor else x
        if x.dtype != torch.float32 or x.device != self.dev or not x.is_contiguous():
            x = x.to(self.dev, torch.float32).contiguous()
        for t in (p0, p1, x):
            if t is not None and t.numel() not in (1, lw.numel()):
                raise RuntimeError('lock-step log-weight term: tensors of 1 or n elements')
        L.check(self.lib.pp_logweight_accumulate_rows(int(kind), L.ptr(p0), int(s0), L.ptr(p1), int(s1), x.data_ptr(),
                                                      0 if x.numel() == 1 else 1, float(scale), lw.data_ptr(), rows.data_ptr(),
                                                      int(rows.numel()), self._st), 'pp_logweight_accumulate_rows')

    def copy_rows(self, src, dst, rows):
        """dst[rows] = src[rows] (src: n values or one shared value), in place (pp_copy_rows)."""
        src = src.as_subclass(torch.Tensor) if type(src) is not torch.Tensor else src
        if src.dtype != torch.float32 or src.device != self.dev or not src.is_contiguous():
            src = src.to(self.dev, torch.float32).contiguous()
        if src.numel() not in (1, dst.numel()):
            raise RuntimeError('copy_rows: a source of 1 or n elements')
        L.check(self.lib.pp_copy_rows(src.data_ptr(), 0 if src.numel() == 1 else 1, dst.data_ptr(), rows.data_ptr(), int(rows.numel()),
                                      self._st), 'pp_copy_rows')
        torch.autograd.graph.increment_version(dst)       # written by the kernel: memoised results of it are stale

    def accumulate(self, lw, kind, p0, p1, x, scale=1.0, term=None):
        """lw += scale * log_prob(dist(p0, p1); x); p0/p1/x are device tensors of 1 (broadcast) or n elements."""
        if term is None:
            term = (kind, p0, 0 if p0.numel() == 1 else 1, p1, 0 if p1.numel() == 1 else 1)
        self._term_accumulate(term, x, scale, lw)

    def accumulate_terms(self, lw, terms, overwrite=False):
        """One pass for up to four terms; terms = [(kind, p0, p1, x, scale)] (kind 2 = the tensor x itself) or
        [(dist_term(...), x, scale)]."""
        def s(t):
            return 0 if (t is None or t.numel() == 1) else 1
        kinds, p0s, s0s, p1s, s1s, xs, scales = [], [], [], [], [], [], []
        for item in terms:
            if len(item) == 3:          # (dist_term tuple, x, scale)
                (kind, p0, s0, p1, s1), x, scale = item
            else:
                kind, p0, p1, x, scale = item
                s0, s1 = s(p0), s(p1)
            kinds.append(int(kind)); p0s.append(p0); s0s.append(int(s0)); p1s.append(p1); s1s.append(int(s1))
            xs.append(x); scales.append(float(scale))
        ops.logweight_terms(lw, kinds, p0s, s0s, p1s, s1s, xs, scales, bool(overwrite))

    def axpy(self, lw, scale, term):
        """lw += scale * term (e.g. -log q)."""
        ops.logweight_terms(lw, [2], [None], [0], [None], [0], [term], [float(scale)], False)

    def stats(self, lw, x=None):
        """Importance statistics (Empirical.finalize / expectation / effective_sample_size,
        pyprob/distributions/empirical.py:298-309, 451-466, 758-766) reduced on the device in float64."""
        self._stats = ops.is_stats(lw, x, self._stats_scratch)
        return self._stats_dict(self._stats)

    @staticmethod
    def _stats_dict(stats):
        m, sw, sw2, swx, swx2, cnt = (float(v) for v in stats[:6]) if isinstance(stats, np.ndarray) else stats.tolist()[:6]
        mean = swx / sw if sw > 0 else float('nan')
        var = swx2 / sw - mean * mean if sw > 0 else float('nan')
        return dict(max_lw=m, sum_w=sw, sum_w2=sw2, sum_wx=swx, sum_wx2=swx2, count=cnt,
                    ess=(sw * sw / sw2) if sw2 > 0 else 0.0, mean=mean, var=var,
                    log_evidence=m + np.log(sw / max(cnt, 1)) if sw > 0 else float('-inf'))


def gum_posterior(engine, num_particles, obs=(8.0, 9.0), prior_mean=1.0, prior_stddev=5.0 ** 0.5,
                  likelihood_stddev=2.0 ** 0.5, seed=0, offset=0, runner=None, return_particles=False):
    """posterior_results(N, IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK, observe={'obs0':.., 'obs1':..}) for the
    GaussianUnknownMean program (reference tests/test_inference.py:97-109), executed lock-step:
        mu = sample(Normal(m0, s0)); observe(Normal(mu, s), obs0); observe(Normal(mu, s), obs1); return mu
    log w = log N(mu; m0, s0) - log q(mu | y) + log N(y0; mu, s) + log N(y1; mu, s)   (SURVEY.md §8a)."""
    run = runner or ISRunner(engine)
    dev = engine.device
    run.init(obs)
    run.begin(num_particles, offset=offset)
    key = ('prior', float(prior_mean), float(prior_stddev))
    prior = run._consts.get(key)
    if prior is None:
        prior = run._consts[key] = torch.tensor([[prior_mean, prior_stddev]], dtype=torch.float32, device=dev)
    mu, logq = run.step(0, None, prior, seed=seed)
    lw = torch.empty(num_particles, dtype=torch.float32, device=dev)
    s = run._const(likelihood_stddev)
    # + log p(mu) (state.py:211)  - log q(mu) (state.py:212,217)  + log p(y_j | mu) (state.py:147-149): one fused pass
    terms = [(0, prior[0, 0:1], prior[0, 1:2], mu, 1.0), (2, None, None, logq, -1.0)]
    terms += [(0, mu, s, run._const(float(y)), 1.0) for y in obs]
    run.accumulate_terms(lw, terms, overwrite=True)
    st = run.stats(lw, mu)
    st['std'] = float(np.sqrt(max(st['var'], 0.0)))
    if return_particles:
        st['values'], st['log_weights'] = mu, lw
    return st
