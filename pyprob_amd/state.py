"""The probabilistic-program interpreter for the engines on the hot path (mirror of pyprob/state.py).

Module-global trace state, `sample` / `observe`, address extraction from the call site, two execution modes:

  * per-trace mode  -- one particle / one prior trace per run of `forward()`, as pyprob does (state.py:158-293,
                       118-155); the IC branch (state.py:203-219) calls `InferenceNetworkLSTM._infer_step`, which
                       is one `pp_is_step` C call (LSTM step + head + sample + log q) with n = 1;
  * lock-step mode  -- N particles run `forward()` with N-wide tensors as values (ParticleTensor): every `sample`
                       is one `pp_is_step` for all particles of the current control-flow path, every `observe` one
                       fused log-prob/accumulate kernel. A per-particle condition (`while s >= 1:`) splits the run
                       into one execution per distinct path (LockStepState). The same machinery generates PRIOR
                       traces for training N at a time (PriorLockStep: vectorised OnlineDataset).
                       Programs that turn sampled values into Python scalars (`float(s)`) must use per-trace mode.

MCMC engines, `tag`/`factor` and the address dictionary are out of scope.
"""
import enum
import opcode
import os
import sys
import time

import torch

from .distributions import Normal
from .trace import Trace, Variable


class TraceMode(enum.Enum):
    PRIOR = 1
    POSTERIOR = 2
    PRIOR_FOR_INFERENCE_NETWORK = 3


class InferenceEngine(enum.Enum):
    IMPORTANCE_SAMPLING = 0
    IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK = 1


class InferenceNetwork(enum.Enum):     # pyprob/__init__.py
    FEEDFORWARD = 0
    LSTM = 1


class LearningRateScheduler(enum.Enum):    # pyprob/__init__.py
    NONE = 0
    POLY1 = 1
    POLY2 = 2


class Optimizer(enum.Enum):                # pyprob/__init__.py
    ADAM = 0
    SGD = 1
    ADAM_LARC = 2
    SGD_LARC = 3


class PriorInflation(enum.Enum):       # pyprob/__init__.py; state.py:87-93
    DISABLED = 0
    ENABLED = 1


_trace_mode = TraceMode.PRIOR
_prior_inflation = PriorInflation.DISABLED
_inference_engine = InferenceEngine.IMPORTANCE_SAMPLING
_likelihood_importance = 1.
_current_trace = None
_current_trace_root_function_name = None
_current_trace_inference_network = None
_current_trace_previous_variable = None
_current_trace_observed_variables = {}
_current_trace_execution_start = None
_lock_step = None      # LockStepState when N particles advance together
_coroutine = None      # coroutine.CoroutineIS while particle coroutines run an unmodified program (one greenlet per particle)


_address_frame_skip = 0


def _make_address(distribution, address):
    if address is None:
        base = _extract_address_from_caller() + '__' + distribution._address_suffix
    else:
        base = address + '__' + distribution._address_suffix
    instance = _current_trace.last_instance(base) + 1
    return base, base + '__' + str(instance), instance


_STORE_FAST, _STORE_NAME, _STORE_GLOBAL = (opcode.opmap[n] for n in ('STORE_FAST', 'STORE_NAME', 'STORE_GLOBAL'))
_LOAD_FAST, _LOAD_NAME, _LOAD_GLOBAL, _LOAD_CONST = (opcode.opmap[n] for n in ('LOAD_FAST', 'LOAD_NAME', 'LOAD_GLOBAL', 'LOAD_CONST'))
_STORE_SUBSCR, _RETURN_VALUE = opcode.opmap['STORE_SUBSCR'], opcode.opmap['RETURN_VALUE']
_target_cache = {}      # (code object, f_lasti) -> name of the assignment target, or a resolver for subscripted targets


def _assignment_target(frame):
    """What the value of the `sample` / `observe` call at frame.f_lasti is assigned to: the reference puts this name into
    the address (pyprob/state.py:29-83), so it is part of the checkpoint format. `x = sample(..)` -> 'x'; `a[3] = ..` /
    `a[i] = ..` (int i) -> 'a[3]'; `return sample(..)` -> 'return'; anything else -> '?'. CPython >= 3.6 word code:
    the instruction after the call sits at f_lasti + 2. Decoded once per call site."""
    code, at = frame.f_code, frame.f_lasti
    key = (code, at)
    hit = _target_cache.get(key)
    if hit is None:
        co = code.co_code
        op = co[at + 2] if at + 2 < len(co) else -1
        arg = co[at + 3] if at + 3 < len(co) else 0
        hit = '?'
        if op == _STORE_FAST:
            hit = code.co_varnames[arg]
        elif op in (_STORE_NAME, _STORE_GLOBAL):
            hit = code.co_names[arg]
        elif op == _RETURN_VALUE:
            hit = 'return'
        elif op in (_LOAD_FAST, _LOAD_NAME, _LOAD_GLOBAL) and at + 6 < len(co) and co[at + 6] == _STORE_SUBSCR \
                and co[at + 4] in (_LOAD_CONST, _LOAD_FAST):
            base = (code.co_varnames if op == _LOAD_FAST else code.co_names)[arg]
            if co[at + 4] == _LOAD_CONST:
                index = code.co_consts[co[at + 5]]
                hit = base + '[' + str(index) + ']' if type(index) is int else '?'
            else:
                hit = (base, code.co_varnames[co[at + 5]])      # the index is a local: resolved per call
        _target_cache[key] = hit
    if type(hit) is tuple:
        index = frame.f_locals[hit[1]]
        return hit[0] + '[' + str(index) + ']' if type(index) is int else '?'
    return hit


def _extract_address_from_caller():
    """'<instruction offset>__<root function>__..__<calling function>__<assignment target>' exactly as the reference
    builds it (pyprob/state.py:29-47), so that networks and datasets interchange."""
    # one extra frame because of _make_address; `_address_frame_skip` more when sample / observe were entered through a
    # wrapper (pyprob_amd/pyprob_host.py: pyprob's own entry points forwarding to this runtime)
    frame = sys._getframe(3 + _address_frame_skip)
    ip = frame.f_lasti
    names = [_assignment_target(frame)]
    while frame is not None:
        n = frame.f_code.co_name
        if n.startswith('<') and n != '<listcomp>':
            break
        names.append(n)
        if n == _current_trace_root_function_name:
            break
        frame = frame.f_back
    return '{}__{}'.format(ip, '__'.join(reversed(names)))


class ParticleTensor(torch.Tensor):
    """A per-particle value of a lock-step run: an ordinary tensor (every torch op works and keeps the type) whose truth
    value is a BRANCH of the program. `while s >= 1:` on N particles asks the executor (LockStepState.branch): if the
    active particles agree that is the answer, otherwise the run splits - this execution continues with the particles
    that said True, the others are re-run later down the other branch with their recorded values."""

    @staticmethod
    def wrap(t):
        return t if isinstance(t, ParticleTensor) else torch.as_tensor(t).as_subclass(ParticleTensor)

    # While a draw of the lock-step executor is still DEFERRED (LockStepState.draw: the values' storage exists but is
    # filled by the fused kernel at the next flush), only operations that do not read the data may run on a ParticleTensor
    # without forcing that flush: attribute getters, views, passing it as a distribution parameter.
    _NO_READ = frozenset(('__get__', 'as_subclass', 'reshape', 'view', 'expand', 'numel', 'size', 'dim', 'is_contiguous',
                          'data_ptr', 'stride', 'storage_offset', 'detach', 'is_floating_point', 'is_complex', 'as_tensor',
                          'unsqueeze', 'squeeze', 'flatten', 'element_size', 'nelement', '__len__', 'ndimension'))

    # Pure elementwise operators whose results the lock-step executor may REUSE within one posterior call: every control-flow
    # path re-runs forward() from the top, so a path at depth d recomputes the arithmetic of its d - 1 replayed iterations on
    # full-width tensors (`x * x + y * y`, `s >= 1` ...: O(d^2) launches per call, each ~5 us of interpreter + dispatch). The
    # result of such an operator is a function of its inputs' contents only; it is kept under (operator, storage address + version of
    # every tensor argument, scalar arguments) and handed out again while the result itself has not been modified
    # (`Tensor._version` counts in-place writes of inputs and results alike).
    _PURE = frozenset(('add', 'sub', 'mul', 'div', 'true_divide', 'neg', 'pow', 'sqrt', 'rsqrt', 'log', 'log1p', 'log2', 'exp', 'expm1',
                       'abs', 'sin', 'cos', 'tan', 'tanh', 'sigmoid', 'ge', 'gt', 'le', 'lt', 'eq', 'ne', 'maximum', 'minimum', 'square',
                       'reciprocal', '__add__', '__radd__', '__sub__', '__rsub__', '__mul__', '__rmul__', '__truediv__', '__rtruediv__',
                       '__pow__', '__rpow__', '__neg__', '__ge__', '__gt__', '__le__', '__lt__', '__eq__', '__ne__'))

    # In-place forms. A memoised result that has been handed out MORE THAN ONCE is shared between two names of the program
    # (`a = x * x; b = x * x` are the same tensor object): `a += 1` on it is computed out of place (Python rebinds `a` to the
    # returned tensor, `b` keeps the product); a method-call form (`a.add_(1)`, `a[0] = ..`) cannot be redirected and raises
    # instead of silently changing `b` (PP_IS_MEMO=0 runs such a program without result reuse).
    def _shared_result(self):
        ls = _lock_step
        shared = getattr(ls, 'memo_shared', None) if ls is not None else None
        if not shared:
            return False
        with torch._C.DisableTorchFunctionSubclass():
            return self.untyped_storage()._cdata in shared

    def __iadd__(self, other):
        return self + other if self._shared_result() else super().__iadd__(other)

    def __isub__(self, other):
        return self - other if self._shared_result() else super().__isub__(other)

    def __imul__(self, other):
        return self * other if self._shared_result() else super().__imul__(other)

    def __itruediv__(self, other):
        return self / other if self._shared_result() else super().__itruediv__(other)

    def __ipow__(self, other):
        return self ** other if self._shared_result() else super().__ipow__(other)

    MEMO_BYTES = int(os.environ.get('PP_IS_MEMO_BYTES', str(4 << 30)))     # results kept per call before the memo starts over

    @staticmethod
    def _memo_key(name, args):
        with torch._C.DisableTorchFunctionSubclass():      # (metadata reads: no Python dispatch per attribute of a ParticleTensor)
            return ParticleTensor._memo_key_plain(name, args)

    @staticmethod
    def _memo_key_plain(name, args):
        key = [name]
        for a in args:
            if isinstance(a, torch.Tensor):
                # storage identity, not object identity: a replayed statement hands out a NEW wrapper of the recorded values;
                # shape, strides and offset: views of one storage (unsqueeze / expand / slices) share address, size and version
                key.append((a.data_ptr(), a.storage_offset(), a.shape, a.stride(), a.dtype, a._version))
            elif isinstance(a, (bool, int, float)):
                key.append(('s', type(a).__name__, a))
            else:
                return None
        return tuple(key)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        ls = _lock_step
        name = getattr(func, '__name__', '') if ls is not None else ''
        if ls is not None and getattr(ls, 'draw', None) is not None:
            lazy_ok = name in cls._NO_READ or (name in ('float', 'contiguous') and len(args) == 1 and torch.is_tensor(args[0]) and
                                               args[0].dtype == torch.float32 and args[0].is_contiguous())
            if not lazy_ok:
                ls.flush()
        memo = getattr(ls, 'memo', None) if ls is not None else None
        if memo is not None and not kwargs and name in cls._PURE and len(types) == 1:
            # One stretch with the subclass dispatch off: the key (metadata reads), the lookup, and on a miss the operator itself on
            # the plain tensors + ONE as_subclass of its result - torch's default __torch_function__ (re-dispatch + _convert of the
            # result) costs ~6 us per operator, a third of what a deep level of a control-flow program spends per statement
            with torch._C.DisableTorchFunctionSubclass():
                key = cls._memo_key_plain(name, args)
                if key is not None:
                    hit = memo.get(key)
                    if hit is not None and hit[0]._version == hit[1]:
                        # handed out a second time: two names of the program share it now. Keyed by STORAGE identity (views
                        # of the result are covered) and the tensor is held for the rest of the call, so that neither
                        # memo.clear() nor the program dropping both names can hand its address to an unrelated tensor
                        with torch._C.DisableTorchFunctionSubclass():
                            ls.memo_shared[hit[0].untyped_storage()._cdata] = hit[0]
                        return hit[0]
                    out = func(*args)
                    if isinstance(out, torch.Tensor):
                        if type(out) is not cls:
                            out = out.as_subclass(cls)
                        nbytes = out.numel() * out.element_size()
                        ls.memo_bytes += nbytes
                        if ls.memo_bytes > cls.MEMO_BYTES:      # bound what one call pins (many-path programs at 1e6 particles)
                            memo.clear()
                            ls.memo_bytes = nbytes
                        memo[key] = (out, out._version, args)      # (the arguments stay alive: their storage is not reused)
                    return out
        elif memo is not None and ls.memo_shared and args and isinstance(args[0], torch.Tensor) and \
                (name == '__setitem__' or (name.endswith('_') and not name.startswith('__'))):
            with torch._C.DisableTorchFunctionSubclass():
                shared = args[0].untyped_storage()._cdata in ls.memo_shared
            if shared:
                raise RuntimeError('lock-step executor: in-place `%s` on a tensor that two expressions of the program share '
                                   '(a reused elementwise result); write it out of place or set PP_IS_MEMO=0' % name)
        return super().__torch_function__(func, types, args, kwargs or {})

    def __bool__(self):
        ls = _lock_step
        if ls is not None and getattr(ls, 'draw', None) is not None:
            ls.flush()        # `if sample(...):` reads the values: a deferred draw has to exist before the branch looks at them
        plain = self.as_subclass(torch.Tensor)
        if ls is None or plain.numel() != ls.width:
            return bool(plain)
        return ls.branch(plain)


class PathExecutor:
    """Control-flow bookkeeping shared by the lock-step executors: one EXECUTION of the program follows one
    control-flow path for the particles in `active`; statements already executed for these particles (a prefix shared
    with an earlier execution) are REPLAYED from the statement log instead of being sampled again. `pending` holds the
    paths still to run: (active mask, recorded branch decisions, statements done, observes done)."""

    compact = False       # True: a re-run path works on tensors of ITS particles only (width = their number)

    def __init__(self, n, device):
        self.n = n
        self.dev = device
        self.log = []                 # per statement index: {address: recorded values ...}
        self.pending = []
        self.path_id = 0
        self.start_path(None, [], 0, 0)

    def start_path(self, active, decisions, statements_done, observes_done):
        """`active`: which particles this execution is for - None = all; a bool mask [n] (full-width executors); or, for
        compact executors, the int64 indices of its particles (the execution's tensors then have that many elements)."""
        if self.compact:
            self.base = active                    # global particle index of every element of this execution's tensors
            self.width = self.n if active is None else int(active.numel())
            self.active = None                    # narrowed by branch(): bool [width]
            self.rows = None
            self.n_active = self.width
        else:
            self.base = None
            self.width = self.n
            self.active = active          # bool [n], or None = every particle
            self.rows = None if active is None else torch.nonzero(active).reshape(-1)
            self.n_active = self.n if active is None else int(self.rows.numel())
        self.decisions = list(decisions)
        self.decisions_seen = 0
        self.replay_statements = statements_done
        self.replay_observes = observes_done
        self.statement = 0
        self.observes = 0
        self.prev_addr_id = None
        self.prev_unknown = False

    def global_rows(self, mask=None):
        """Global particle indices of the (masked) elements of this execution's tensors; None = every particle."""
        if mask is None:
            return self.base
        idx = torch.nonzero(mask).reshape(-1)
        return idx if self.base is None else self.base[idx]

    def next_path(self):
        """Switch to the next queued path; False when none is left."""
        if not self.pending:
            return False
        entry = self.pending.pop()
        active, decisions, st_done, ob_done = entry[:4]
        self._next_prefix = entry[4] if len(entry) > 4 else None      # (LockStepState: the statements of the replay prefix, in order)
        self.path_count = getattr(self, 'path_count', 0) + 1      # (ids in starting order; a nested run returns to its caller's id)
        self.path_id = self.path_count
        if self.path_id > 4096:
            raise RuntimeError('lock-step execution: more than 4096 control-flow paths')
        self.start_path(active, decisions, st_done, ob_done)
        return True

    def branch(self, cond):
        """Truth value of a per-particle condition for the active particles (ParticleTensor.__bool__)."""
        k = self.decisions_seen
        self.decisions_seen += 1
        if k < len(self.decisions):
            return self.decisions[k]             # replayed prefix: this path already knows its way
        if getattr(self, 'draw', None) is not None:
            self.flush()      # (any path that reaches a branch with a deferred draw pending: the condition may alias its storage)
        c = cond.reshape(-1).to(self.dev) != 0
        t = c if self.active is None else (c & self.active)
        n_true = int(t.sum().item())
        if n_true == self.n_active:
            decision = True
        elif n_true == 0:
            decision = False
        else:   # diverge: the False side is queued with everything this execution has done so far as its replay prefix
            other = ~c if self.active is None else (self.active & ~c)
            self.pending.append((self.global_rows(other) if self.compact else other, self.decisions + [False],
                                 self.statement, self.observes))
            self.active = t
            self.rows = torch.nonzero(t).reshape(-1)
            self.n_active = n_true
            decision = True
        self.decisions.append(decision)
        return decision


class LockStepState(PathExecutor):
    """Lock-step importance sampling over n particles (particles with stochastic control flow served
    in address-grouped batches). All per-particle quantities are full-size [n] device tensors."""
    mode = 'is'

    def __init__(self, runner, n, seed, offset):
        import os
        self.runner = runner
        self.seed = seed
        self.offset = offset
        self._lw = torch.empty(n, dtype=torch.float32, device=runner.dev)
        self._lw_valid = False        # nothing has been accumulated yet: the first fused pass overwrites instead of adding
        # Deferred work of the current execution (full-width statements only): `draw` = the first statement's draw from the
        # shared proposal (its network part has run, pp_is_step_net), `terms` = log-weight terms queued since. flush() runs
        # them as ONE pass over the particles (pp_is_fused); PP_IS_FUSED=0 keeps one kernel per term (A/B, tests).
        self.fused = os.environ.get('PP_IS_FUSED', '1') != '0'
        self.draw = None
        self.terms = []
        self.final_stats = None
        # what Model._traces_lockstep needs to recognise a program whose whole call is ONE draw + ONE fused pass (launch plan,
        # model.py): the number of flushes, where every deferred term's value came from, anything that read a draw early
        self.memo = {} if os.environ.get('PP_IS_MEMO', '1') != '0' else None     # ParticleTensor._PURE results of this call
        self.memo_shared = {}         # storage identity -> memoised result handed out more than once (kept alive for the call)
        self.memo_bytes = 0
        self.flushes = 0
        self.plan_terms = []          # (term, source, scale): source = ('obs', name) | ('value',) | ('const', tensor)
        self.plan_ok = True
        # A path's particles are its ROW LIST (ascending int64 indices, None = every particle): branches split it with one
        # partition call (pp_partition_rows: two launches + one 8-byte read-back; the boolean-mask bookkeeping of PathExecutor
        # cost a masked sum + .item() and two nonzero() - ten launches and three synchronisations - per branch), statements,
        # observes and the result copy work on the rows. The boolean mask exists only if somebody asks for it (`active`).
        self.by_rows = runner.dev.type == 'cuda' and os.environ.get('PP_IS_ROWS', '1') != '0'
        self._mask = None
        # A branch has to WAIT for the device (the condition is computed from values the queued statement kernels are still
        # drawing). Instead of idling there, the host first runs a queued path (Model._traces_lockstep sets `nest`): its
        # replay and its handful of small launches - queued behind the statement kernels - overlap the wait. One level deep.
        self.nest = None
        self._nest_depth = 0
        self.wrappers = {}            # id(recorded values) -> (values, its ParticleTensor): replays hand out one object per tensor
        self.nest_ok = os.environ.get('PP_IS_NEST', '1') != '0'
        super().__init__(n, runner.dev)

    _PATH_FIELDS = ('rows', '_mask', 'n_active', 'decisions', 'decisions_seen', 'replay_statements', 'replay_observes', 'statement',
                    'observes', 'prev_addr_id', 'prev_unknown', 'path_id', 'history', 'prefix')

    def _run_nested(self):
        """Run the most recently queued path to its end inside the current execution (which is waiting at a branch)."""
        global _current_trace, _current_trace_previous_variable, _current_trace_execution_start
        saved = {k: getattr(self, k) for k in self._PATH_FIELDS}
        saved_runner = (self.runner.prev_value, self.runner.last_value)
        saved_trace = (_current_trace, _current_trace_previous_variable, _current_trace_execution_start)
        self._nest_depth += 1
        try:
            self.nest()
        finally:
            self._nest_depth -= 1
            for k, v in saved.items():
                setattr(self, k, v)
            self.runner.prev_value, self.runner.last_value = saved_runner
            _current_trace, _current_trace_previous_variable, _current_trace_execution_start = saved_trace

    def start_path(self, active, decisions, statements_done, observes_done):
        # `history`: the controlled statements of this execution in order, (Variable, recorded values, address id, ParticleTensor).
        # A queued path carries its parent's history as `prefix`: state.sample() replays statement j of the prefix from
        # prefix[j] without building the address from the call stack again (a path at depth d replays 2 d statements: the
        # frame walk, the Variable and the log look-up were ~8 us each, ~0.4 ms per call of the Marsaglia program)
        self.history = []
        self.prefix = self.__dict__.pop('_next_prefix', None)
        if not self.by_rows:
            return super().start_path(active, decisions, statements_done, observes_done)
        super().start_path(None, decisions, statements_done, observes_done)
        self.rows = active            # (queued by branch(): the row list of the False side)
        self._mask = None
        self.n_active = self.n if active is None else int(active.numel())

    @property
    def active(self):
        """bool [n] of this execution's particles, None = all of them."""
        if not self.by_rows:
            return self._mask
        if self.rows is None:
            return None
        if self._mask is None:
            self._mask = torch.zeros(self.n, dtype=torch.bool, device=self.dev).index_fill_(0, self.rows, True)
        return self._mask

    @active.setter
    def active(self, mask):
        self._mask = mask

    def branch(self, cond):
        if not self.by_rows:
            return super().branch(cond)
        k = self.decisions_seen
        self.decisions_seen += 1
        if k < len(self.decisions):
            return self.decisions[k]             # replayed prefix: this path already knows its way
        if self.draw is not None:
            self.flush()      # (the condition may alias a deferred draw's storage)
        c = cond.reshape(-1)
        if c.dtype != torch.bool or c.device != self.dev or not c.is_contiguous():
            c = (c.to(self.dev) != 0).contiguous()
        handle = self.runner.partition_launch(c, self.rows, self.n_active)
        if self.pending and self.nest is not None and self.nest_ok and self._nest_depth == 0:
            self._run_nested()
        rows_true, rows_false, n_true, n_false = self.runner.partition_read(handle)
        if n_false == 0:
            decision = True
        elif n_true == 0:
            decision = False
        else:   # diverge: the False side is queued with everything this execution has done so far as its replay prefix
            self.pending.append((rows_false, self.decisions + [False], self.statement, self.observes, list(self.history)))
            self.rows = rows_true
            self._mask = None
            self.n_active = n_true
            decision = True
        self.decisions.append(decision)
        return decision

    @property
    def lw(self):
        """The per-particle log-weight accumulator; eager accumulation paths see a zero-initialised vector."""
        if not self._lw_valid:
            self._lw.zero_()
            self._lw_valid = True
        return self._lw

    def defer_term(self, term, x, scale, source=None):
        """Queue lw += scale * log_prob(term; x) for the next fused pass."""
        self.terms.append((term, x, float(scale)))
        self.plan_terms.append((term, source, float(scale)))
        if len(self.terms) >= 7:
            self.flush()

    def flush(self, final=False):
        """Run the deferred draw and the queued terms as one pass over the particles; with final=True the importance
        statistics of (lw, values) come out of the same pass (returned, and kept in final_stats)."""
        draw, terms = self.draw, self.terms
        if draw is None and not terms and not final:
            return None
        self.flushes += 1
        if final:
            self.plan_final = dict(draw=draw, n_terms=len(terms))
        self.draw, self.terms = None, []          # (cleared first: the calls below may touch ParticleTensors)
        value = draw['values'] if draw is not None else None
        vptr = None if value is None else value.data_ptr()
        fterms = []
        if draw is not None:      # + log p(v) of the program's own prior (state.py:211); - log q(v) is part of the draw
            fterms.append((draw['prior_term'], value, 1.0, 4))

        def is_value(t):
            return vptr is not None and t is not None and t.numel() == self.n and t.data_ptr() == vptr
        for (kind, p0, s0, p1, s1), x, scale in terms:
            fl = (1 if is_value(p0) else 0) | (2 if is_value(p1) else 0) | (4 if is_value(x) else 0)
            fterms.append(((kind, p0, s0, p1, s1), x, scale, fl))
        stats_x = value if value is not None else getattr(self, 'stats_values', None)
        want_stats = final and stats_x is not None
        out = self.runner.fused(None if draw is None else draw['addr'], None if draw is None else draw['prior'], fterms,
                                stats_x if stats_x is not None else self._lw, self._lw, overwrite=not self._lw_valid,
                                seed=0 if draw is None else draw['seed'], stats=want_stats) \
            if (draw is not None or fterms or want_stats) else None
        if draw is not None or fterms:
            self._lw_valid = True
        if want_stats:
            self.final_stats = out
            self.final_stats_of = (stats_x.data_ptr(), stats_x.numel())     # what the statistics were reduced over
        return out


def _inflate(distribution):
    """state._inflate, pyprob/state.py:87-93: with prior inflation the VALUE of a Normal is drawn with 3x the standard
    deviation and that of a Categorical uniformly; the variable keeps the original distribution (the prior parameters
    the proposal heads see, and log_prob, are those of the program's own prior)."""
    if _prior_inflation == PriorInflation.ENABLED:
        from .distributions import Categorical, Normal
        if distribution.name == 'Categorical':
            return Categorical(torch.full((distribution.num_categories,), 1. / distribution.num_categories))
        if distribution.name == 'Normal':
            return Normal(distribution.mean, distribution.stddev * 3)
    return None


def _vector_draw(distribution, n):
    """n independent draws: one per particle for shared parameters, elementwise for per-particle parameters."""
    td = distribution._torch_dist
    batch = td.batch_shape.numel() if len(td.batch_shape) else 1
    draw = td.sample((n,)).reshape(-1) if batch == 1 else td.sample().reshape(-1)
    return draw.float()


def _vector_params(distribution, n, device):
    """Prior parameters as the proposal heads read them (packed.distribution_params), one row per particle."""
    name = distribution.name
    if name == 'Normal':
        p = (distribution.mean, distribution.stddev)
    elif name == 'Uniform':
        p = (distribution.low, distribution.high)
    elif name == 'Categorical':
        p = (torch.zeros(1), torch.zeros(1))
    elif name == 'Poisson':      # the Poisson head's fixed interval (packed.distribution_params)
        from .packed import POISSON_LOW_HIGH
        p = (torch.tensor([POISSON_LOW_HIGH[0]]), torch.tensor([POISSON_LOW_HIGH[1]]))
    else:
        raise RuntimeError('Distribution currently unsupported: {}'.format(name))
    return [torch.as_tensor(q, dtype=torch.float32).as_subclass(torch.Tensor).reshape(-1).to(device).expand(n) for q in p]


class PriorLockStep(PathExecutor):
    """n PRIOR traces generated together (trace mode PRIOR_FOR_INFERENCE_NETWORK, pyprob/nn/dataset.py:50-62 run n
    times): every `sample` draws from the prior for the particles of the current control-flow path, every `observe`
    draws the synthetic observation; per path the statement list is recorded, so the traces come out directly as the
    ragged columns a minibatch is packed from (no Trace objects)."""
    mode = 'prior'
    compact = True      # a re-run path draws and computes for its own particles only (GUMM: 1.3 n instead of 8 n work)

    def __init__(self, n, device='cpu'):
        self.obs_log = []             # per observe index: {name: values [n]}
        self.paths = []               # finished executions: (particle indices or None, [(j, address)], [(i, name)])
        # device chunks: Normal / Uniform columns are drawn by pp_prior_draw (Philox, csrc/is_kernels.hip) where the training
        # step reads them; the key comes from torch's generator, so pyprob.seed / torch.manual_seed reproduce a chunk
        self.seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self._consts = {}
        super().__init__(n, torch.device(device))

    def _param(self, v):
        """A distribution parameter as a float32 device vector (scalars cached: no host-to-device copy per statement)."""
        if not torch.is_tensor(v) or (v.device.type == 'cpu' and v.numel() == 1):
            key = float(v)
            t = self._consts.get(key)
            if t is None:
                t = self._consts[key] = torch.tensor([key], dtype=torch.float32, device=self.dev)
            return t
        return v.as_subclass(torch.Tensor).reshape(-1).to(self.dev, torch.float32).contiguous()

    def _draw(self, distribution, stream_id):
        """One value per particle of this execution from `distribution` (prior inflation applied by the caller)."""
        if self.dev.type != 'cpu' and distribution.name in ('Normal', 'Uniform'):
            from .ops import ops
            if distribution.name == 'Normal':
                kind, p0, p1 = 0, self._param(distribution.mean), self._param(distribution.stddev)
            else:
                kind, p0, p1 = 1, self._param(distribution.low), self._param(distribution.high)
            if p0.numel() in (1, self.width) and p1.numel() in (1, self.width):
                # counters: particle index inside this execution; the path id separates re-run paths of one chunk
                return ops.prior_draw(kind, p0, p1, self.width, self.seed, self.path_id << 32, stream_id)
        return _vector_draw(distribution, self.width).to(self.dev)

    def start_path(self, active, decisions, statements_done, observes_done):
        super().start_path(active, decisions, statements_done, observes_done)
        self.seq, self.obs_seq = [], []
        self._active_rows = None

    def _store(self, full, new):
        """Write this execution's (active) elements of `new` [width] into the full-size record `full` [n] (None: create;
        elements of particles that never execute the statement stay uninitialised and are never read)."""
        new = new.to(self.dev)
        if self.base is None and self.active is None:
            return new
        if new.dim() and new.stride(0) == 0:      # one value for every particle (e.g. the bounds of Uniform(-1, 1))
            if full is None:
                return new[:1].expand(self.n)
            if full.stride(0) == 0 and bool(full[0] == new[0]):
                return full
        if full is None:
            full = torch.empty(self.n, dtype=new.dtype, device=self.dev)
        elif full.dim() and full.stride(0) == 0:
            full = full.contiguous()              # (a broadcast parameter recorded by an earlier path)
        if self.active is None:
            full.index_copy_(0, self.base, new)
        else:
            full.index_copy_(0, self.active_rows(), new[self.active])
        return full

    def active_rows(self):
        """Global indices of the active particles of this execution (cached until branch() narrows the set)."""
        if self._active_rows is None or self._active_rows[0] is not self.active:
            self._active_rows = (self.active, self.global_rows(self.active))
        return self._active_rows[1]

    def _view(self, full):
        """This execution's view [width] of a full-size record."""
        return full if self.base is None else full.index_select(0, self.base)

    def sample_statement(self, address, distribution, control):
        j = self.statement
        self.statement += 1
        if j < self.replay_statements:
            if control:
                self.seq.append((j, address))
            return ParticleTensor.wrap(self._view(self.log[j][address][0]))
        while len(self.log) <= j:
            self.log.append({})
        old = self.log[j].get(address)
        draw = self._draw(_inflate(distribution) or distribution, j)
        if self.dev.type != 'cpu' and distribution.name in ('Normal', 'Uniform'):
            # cached device constants: a host-to-device copy here would wait for every training step in flight on the stream
            pp = (distribution.mean, distribution.stddev) if distribution.name == 'Normal' else (distribution.low, distribution.high)
            p0, p1 = (self._param(q).expand(self.width) if self._param(q).numel() == 1 else self._param(q) for q in pp)
        else:
            p0, p1 = _vector_params(distribution, self.width, self.dev)
        o = (None, None, None) if old is None else old
        ncat = distribution.num_categories if distribution.name == 'Categorical' else None
        self.log[j][address] = (self._store(o[0], draw), self._store(o[1], p0), self._store(o[2], p1), distribution.name, ncat)
        if control:
            self.seq.append((j, address))
        return ParticleTensor.wrap(draw)

    def observe_statement(self, name, distribution, value):
        i = self.observes
        self.observes += 1
        if i >= self.replay_observes:
            while len(self.obs_log) <= i:
                self.obs_log.append({})
            if value is None:
                draw = self._draw(distribution, 0x4000 + i)
            else:
                draw = torch.as_tensor(value, dtype=torch.float32).reshape(-1).to(self.dev).expand(self.width)
            self.obs_log[i][name] = self._store(self.obs_log[i].get(name), draw)
            out = draw
        else:
            out = self._view(self.obs_log[i][name])
        if name is not None:
            self.obs_seq.append((i, name))
        return ParticleTensor.wrap(out)

    def finish_path(self):
        rows = self.active_rows() if self.active is not None else self.base
        self.paths.append((rows, list(self.seq), list(self.obs_seq)))

    def columns_device(self, obs_names):
        """The chunk as device columns when every trace took the same path with ONE controlled statement (e.g.
        GaussianUnknownMean): dict(address=(address, distribution, n_categories), values [n], prior [n, 2], obs [n, W]) -
        what packed.ColumnarDataset blocks into minibatches; None otherwise."""
        if self.dev.type == 'cpu' or len(self.paths) != 1:
            return None
        rows, seq, obs_seq = self.paths[0]
        if rows is not None or len(seq) != 1:
            return None
        j, address = seq[0]
        e = self.log[j][address]
        if e[3] not in ('Normal', 'Uniform'):
            return None
        by_name = dict((name, i) for i, name in obs_seq)
        try:
            cols = [self.obs_log[by_name[name]][name] for name in obs_names]
        except KeyError:
            return None
        if any(c.dim() != 1 or c.numel() not in (1, self.n) for c in cols):
            return None

        def full(t):
            t = t.as_subclass(torch.Tensor).reshape(-1).to(self.dev, torch.float32)
            return t.expand(self.n) if t.numel() == 1 else t
        return dict(address=(address, e[3], e[4]), values=full(e[0]).contiguous(),
                    prior=torch.stack([full(e[1]), full(e[2])], 1).contiguous(),
                    obs=torch.stack([full(c) for c in cols], 1).contiguous())

    def columns(self, obs_names, return_types=False):
        """(trace_len [n], address table [(address, distribution, n_categories)], address ids [R], values [R],
        prior [R, 2], obs [n, W]) as numpy arrays; traces are grouped by control-flow path. return_types appends
        (type index per trace, [address-id sequence per type]): every path is one trace type."""
        import numpy as np
        table, ids_of = [], {}
        lens, ids, vals, pri, obs = [], [], [], [], []
        type_of, seqs = [], []
        for rows, seq, obs_seq in self.paths:
            m = self.n if rows is None else int(rows.numel())
            rows = slice(None) if rows is None else rows
            if m == 0:
                continue
            if not seq:
                raise ValueError('Trace of length zero.')
            row_ids = []
            for j, address in seq:
                if address not in ids_of:
                    ids_of[address] = len(table)
                    e = self.log[j][address]
                    table.append((address, e[3], e[4]))
                row_ids.append(ids_of[address])
            lens.append(np.full(m, len(seq), np.int64))
            type_of.append(np.full(m, len(seqs), np.int64))
            seqs.append(list(row_ids))
            ids.append(np.tile(np.asarray(row_ids, np.int64), m))
            def column(rec):          # the path's elements of a full-size record, as a numpy vector (or a scalar)
                if rec.dim() and rec.stride(0) == 0:
                    return float(rec[0])
                return rec[rows].cpu().numpy()
            v = np.empty((m, len(seq)), np.float32)
            p = np.empty((m, len(seq), 2), np.float32)
            for k, (j, a) in enumerate(seq):
                e = self.log[j][a]
                v[:, k] = column(e[0])
                p[:, k, 0] = column(e[1])
                p[:, k, 1] = column(e[2])
            vals.append(v.reshape(-1))
            pri.append(p.reshape(-1, 2))
            by_name = dict((name, i) for i, name in obs_seq)
            o = np.empty((m, len(obs_names)), np.float32)
            for k, name in enumerate(obs_names):
                o[:, k] = column(self.obs_log[by_name[name]][name])
            obs.append(o)
        out = (np.concatenate(lens), table, np.concatenate(ids), np.concatenate(vals).astype(np.float32),
               np.concatenate(pri).astype(np.float32), np.concatenate(obs).astype(np.float32))
        return out + ((np.concatenate(type_of), seqs),) if return_types else out


def _lock_step_likelihood(distribution, value, obs_name=None):
    """lw += likelihood_importance * log p(value | .) for the active particles of a lock-step IS run (state.py:147-149;
    also the 'Variable is observed' branch of state.sample, :175-180). A replayed prefix has already been scored."""
    ls = _lock_step
    if not torch.is_tensor(value) or (value.device.type == 'cpu' and value.numel() == 1):
        v = ls.runner._const(float(value))       # cached device scalar (no host-to-device copy per statement)
    else:
        v = torch.as_tensor(value, dtype=torch.float32).as_subclass(torch.Tensor).reshape(-1).to(ls.runner.dev)
    if v.numel() not in (1, ls.width):
        # a vector-valued observation (k != n values): the device terms read one value per particle - refuse instead of
        # scoring element i against particle i (the coroutine executor, lock_step=False, sums vector observes on the host)
        raise RuntimeError('lock-step importance sampling scores one observed value per particle (or one shared value); '
                           'got {} values for {} particles - run this program with lock_step=False'.format(v.numel(), ls.width))
    ls.observes += 1
    if ls.observes <= ls.replay_observes:
        return
    term = ls.runner.dist_term(distribution)
    if term is None:
        raise RuntimeError('lock-step importance sampling has no device likelihood for {}'.format(distribution.name))
    if ls.fused and ls.rows is None:       # full width: joins the next fused pass (with the draw, if one is pending)
        ls.defer_term(term, v, _likelihood_importance, source=('obs', obs_name) if (obs_name is not None and v.numel() == 1) else None)
        return
    ls.plan_ok = False
    ls.flush()
    if getattr(ls, 'by_rows', False) and ls.rows is not None:
        ls.runner.accumulate_rows(ls.lw, term, v, ls.rows, _likelihood_importance)
    else:
        ls.runner.accumulate_masked(ls.lw, None, None, None, v, ls.active, scale=_likelihood_importance, term=term)


def observe(distribution, value=None, name=None, address=None):
    """state.observe, pyprob/state.py:118-155."""
    if _current_trace is None:
        return None
    base, addr, instance = _make_address(distribution, address)
    if name in _current_trace_observed_variables:
        value = _current_trace_observed_variables[name]
    elif value is not None:
        value = torch.as_tensor(value, dtype=torch.float32)
    elif _trace_mode == TraceMode.PRIOR_FOR_INFERENCE_NETWORK and distribution is not None:
        value = distribution.sample()
    else:
        value = None
    if _lock_step is not None and _lock_step.mode == 'prior':
        given = _current_trace_observed_variables.get(name) if name in _current_trace_observed_variables else (
            None if value is None else value)
        return _lock_step.observe_statement(name, distribution, given)
    if _coroutine is not None and value is not None and _trace_mode == TraceMode.POSTERIOR:
        _coroutine.observe(distribution, value)       # the weight term joins the next round's likelihood kernels
        _current_trace.add(Variable(distribution=distribution, value=value, address_base=base, address=addr, instance=instance,
                                    log_prob=None, log_importance_weight=None, observed=True, name=name))
        return value
    if _lock_step is not None and value is not None:
        _lock_step_likelihood(distribution, value, obs_name=name if name in _current_trace_observed_variables else None)
        variable = Variable(distribution=distribution, value=value, address_base=base, address=addr, instance=instance,
                            log_prob=None, log_importance_weight=None, observed=True, name=name)
        _current_trace.add(variable)
        return value
    if value is None:
        observed, log_prob, lw = False, None, None
    else:
        observed = True
        log_prob = _likelihood_importance * distribution.log_prob(value, sum=True)
        lw = float(log_prob)                                   # state.py:147-149
    variable = Variable(distribution=distribution, value=value, address_base=base, address=addr, instance=instance,
                        log_prob=log_prob, log_importance_weight=lw, observed=observed, name=name)
    _current_trace.add(variable)
    return variable.value


def sample(distribution, name=None, address=None, control=True):
    """state.sample, pyprob/state.py:158-293 (PRIOR / IS / IS-with-inference-network branches)."""
    global _current_trace_previous_variable
    if _current_trace is None:
        return distribution.sample()
    ls = _lock_step
    if ls is not None and ls.mode == 'is' and ls.prefix is not None and ls.statement < ls.replay_statements and \
            ls.statement < len(ls.prefix) and (name is None or name not in _current_trace_observed_variables):
        # a statement of the replay prefix, by position: what the parent execution recorded (same decisions, same statements)
        entry = ls.prefix[ls.statement]
        ls.statement += 1
        variable, values, a, wrapper = entry
        _current_trace.add(variable)          # (instance counting of the statements after the prefix)
        if a is not None:
            ls.prev_addr_id = a
            ls.runner.prev_value = ls.runner.last_value = values
        ls.history.append(entry)
        return wrapper
    base, addr, instance = _make_address(distribution, address)
    if name in _current_trace_observed_variables and _coroutine is not None and _trace_mode == TraceMode.POSTERIOR:
        value = torch.as_tensor(_current_trace_observed_variables[name], dtype=torch.float32)
        _coroutine.observe(distribution, value)
        _current_trace.add(Variable(distribution=distribution, value=value, address_base=base, address=addr, instance=instance,
                                    log_prob=None, log_importance_weight=None, observed=True, name=name))
        return value
    if name in _current_trace_observed_variables and _lock_step is not None and _lock_step.mode == 'is':
        # an observed `sample` in a lock-step run: its likelihood joins every particle's log-weight like observe()
        value = torch.as_tensor(_current_trace_observed_variables[name], dtype=torch.float32)
        _lock_step_likelihood(distribution, value)
        _current_trace.add(Variable(distribution=distribution, value=value, address_base=base, address=addr, instance=instance,
                                    log_prob=None, log_importance_weight=None, observed=True, name=name))
        return value
    if name in _current_trace_observed_variables:
        value = _current_trace_observed_variables[name]
        log_prob = _likelihood_importance * distribution.log_prob(value, sum=True)
        variable = Variable(distribution=distribution, value=value, address_base=base, address=addr, instance=instance,
                            log_prob=log_prob, log_importance_weight=float(log_prob), observed=True, name=name)
        _current_trace.add(variable)
        return variable.value

    ic = (_trace_mode == TraceMode.POSTERIOR and
          _inference_engine == InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK and control)
    if _lock_step is not None and _lock_step.mode == 'prior':
        value = _lock_step.sample_statement(addr, distribution, control)
        _current_trace.add(Variable(distribution=distribution, value=value, address_base=base, address=addr, instance=instance,
                                    control=control, name=name))     # (instance counting of the next statements)
        return value
    if _lock_step is not None:
        if not ic:
            raise RuntimeError('lock-step mode runs controlled samples with the inference network only')
        ls = _lock_step
        net = _current_trace_inference_network
        value = net._infer_step_lockstep(addr, distribution, ls)   # samples, scores and weights (or replays) the statement
        variable = Variable(distribution=distribution, value=value, address_base=base, address=addr, instance=instance,
                            log_prob=None, control=True, name=name)
        _current_trace.add(variable)
        hist = getattr(ls, 'history', None)
        if hist is not None:
            rec = ls.log[ls.statement - 1].get(addr) if ls.statement - 1 < len(ls.log) else None
            hist.append((variable, rec[0] if rec is not None else value.as_subclass(torch.Tensor), rec[1] if rec is not None else None, value))
        return value

    log_importance_weight = None
    if ic and _coroutine is not None:
        return _coroutine.sample(distribution, base, addr, instance, name)     # parks until the statement is served
    if ic:
        variable = Variable(distribution=distribution, value=None, address_base=base, address=addr, instance=instance,
                            log_prob=0., control=control, name=name)
        proposal = _current_trace_inference_network._infer_step(variable, prev_variable=_current_trace_previous_variable)
        value = proposal.sample()
        if value.dim() > 0:
            value = value[0]
        log_prob = distribution.log_prob(value, sum=True)
        proposal_log_prob = proposal.log_prob(value, sum=True)
        log_importance_weight = float(log_prob) - float(proposal_log_prob)          # state.py:217
        variable = Variable(distribution=distribution, value=value, address_base=base, address=addr, instance=instance,
                            log_prob=log_prob, log_importance_weight=log_importance_weight, control=control, name=name)
        _current_trace_previous_variable = variable
    else:
        inflated = _inflate(distribution)            # state.py:194-202, 280-288
        value = distribution.sample() if inflated is None else inflated.sample()
        log_prob = distribution.log_prob(value, sum=True)
        if inflated is not None:                     # to account for prior inflation
            log_importance_weight = float(log_prob) - float(inflated.log_prob(value, sum=True))
        variable = Variable(distribution=distribution, value=value, address_base=base, address=addr, instance=instance,
                            log_prob=log_prob, log_importance_weight=log_importance_weight, control=control, name=name)
    _current_trace.add(variable)
    return variable.value


def _init_traces(func, trace_mode=TraceMode.PRIOR, inference_engine=InferenceEngine.IMPORTANCE_SAMPLING,
                 inference_network=None, observe=None, likelihood_importance=1., lock_step=None,
                 prior_inflation=PriorInflation.DISABLED):
    """state._init_traces, pyprob/state.py:296-336."""
    global _trace_mode, _inference_engine, _likelihood_importance, _current_trace_root_function_name
    global _current_trace_inference_network, _current_trace_observed_variables, _lock_step, _prior_inflation
    _prior_inflation = prior_inflation
    _trace_mode = trace_mode
    _inference_engine = inference_engine
    _likelihood_importance = likelihood_importance
    _current_trace_root_function_name = func.__code__.co_name
    _current_trace_observed_variables = {} if observe is None else observe
    _current_trace_inference_network = inference_network
    _lock_step = lock_step
    if inference_engine == InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK:
        if inference_network is None:
            raise ValueError('Expecting an inference network')
        inference_network._infer_init(_current_trace_observed_variables)


def _begin_trace():
    global _current_trace, _current_trace_previous_variable, _current_trace_execution_start
    _current_trace_execution_start = time.time()
    _current_trace = Trace()
    _current_trace_previous_variable = None


def _end_trace(result):
    global _current_trace
    trace = _current_trace
    _current_trace = None
    trace.end(result, time.time() - _current_trace_execution_start)
    return trace
