"""Model facade for the hot path (mirror of pyprob/model.py:24-215): prior, posterior_results with importance
sampling (with or without the inference network), learn_inference_network, save/load_inference_network."""

import numpy as np
import torch

from . import state
from .distributions import Empirical
from .nn import InferenceNetworkFeedForward, InferenceNetworkLSTM, OnlineDataset
from .state import InferenceEngine, InferenceNetwork, Optimizer, PriorInflation, TraceMode
from operator import is_ as _is
from os import environ as _environ
import types


def trace_result(trace):
    return trace.result


def _drop_non_finite(values, log_weights):
    """Model._traces discards traces whose log-weight is nan / inf / -inf with a warning (pyprob/model.py:64-66); the
    lock-step executors do the same for their particle tensors, so one bad particle cannot poison the normalisation."""
    ok = torch.isfinite(log_weights)
    if bool(ok.all()):
        return values, log_weights
    import warnings
    warnings.warn('Encountered {} traces with nan, inf, or -inf log_weight. Discarding them.'.format(int((~ok).sum())))
    return values[ok].contiguous(), log_weights[ok].contiguous()


class Model:
    def __init__(self, name='Unnamed PyProb model'):
        self.name = name
        self._inference_network = None

    def forward(self):
        raise RuntimeError('Model instances must provide a forward method.')

    def _trace_generator(self, trace_mode=TraceMode.PRIOR, inference_engine=InferenceEngine.IMPORTANCE_SAMPLING,
                         inference_network=None, observe=None, likelihood_importance=1.,
                         prior_inflation=PriorInflation.DISABLED, *args, **kwargs):
        """pyprob/model.py:39-45"""
        state._init_traces(func=self.forward, trace_mode=trace_mode, inference_engine=inference_engine,
                           inference_network=inference_network, observe=observe, likelihood_importance=likelihood_importance,
                           prior_inflation=prior_inflation)
        while True:
            state._begin_trace()
            result = self.forward(*args, **kwargs)
            yield state._end_trace(result)

    def _traces(self, num_traces=10, trace_mode=TraceMode.PRIOR, inference_engine=InferenceEngine.IMPORTANCE_SAMPLING,
                inference_network=None, map_func=None, observe=None, likelihood_importance=1.,
                prior_inflation=PriorInflation.DISABLED, *args, **kwargs):
        """pyprob/model.py:47-88: one particle per forward() run; non-finite weights are discarded."""
        gen = self._trace_generator(trace_mode=trace_mode, inference_engine=inference_engine,
                                    inference_network=inference_network, observe=observe,
                                    likelihood_importance=likelihood_importance, prior_inflation=prior_inflation,
                                    *args, **kwargs)
        traces = Empirical()
        map_func = map_func or (lambda t: t)
        for _ in range(num_traces):
            trace = next(gen)
            lw = 1. if trace_mode == TraceMode.PRIOR else trace.log_importance_weight
            if lw != lw or lw in (float('inf'), float('-inf')):
                continue
            traces.add(map_func(trace), lw)
        return traces.finalize()

    def _traces_lockstep(self, num_traces, observe, seed=0, offset=0, likelihood_importance=1., *args, **kwargs):
        """All particles of this rank advance through forward() together. A program whose control flow depends on
        sampled values (written with tensor conditions: `while s >= 1:`) is executed once per distinct control-flow
        path: an execution serves the particles that agree at every branch, the others are queued and re-run with their
        recorded values as replay prefix (state.LockStepState). Every new sample statement is one C-ABI call for the
        path's particles - the general batched IS executor of SURVEY.md 8f.2."""
        runner = self._inference_network._is
        if runner.dev.type == 'cuda' and torch.cuda.current_device() != (runner.dev.index or 0):
            # the C-ABI calls launch on the network's device: scoped, so that a process that drives several devices gets its
            # current device back (ISRunner.begin / init / run_plan only switch when it differs)
            with torch.cuda.device(runner.dev):
                return self._traces_lockstep_on_device(num_traces, observe, seed, offset, likelihood_importance, *args, **kwargs)
        return self._traces_lockstep_on_device(num_traces, observe, seed, offset, likelihood_importance, *args, **kwargs)

    def _traces_lockstep_on_device(self, num_traces, observe, seed=0, offset=0, likelihood_importance=1., *args, **kwargs):
        net = self._inference_network
        runner = net._is
        # Launch plan: a program whose whole call was ONE deferred draw from the first statement's shared
        # proposal + ONE fused pass (all observes queued, nothing read the draw early, one control-flow path) issues the same
        # three C calls every time - observe embedding, first-statement network, fused pass. After such a call they are
        # replayed with the new observation values and seed without running forward() (no Trace / Variable / ParticleTensor
        # bookkeeping: ~170 us of interpreter per call). See _lockstep_plan_key / _record_lockstep_plan for when it is valid.
        plan_key = self._lockstep_plan_key(num_traces, observe, likelihood_importance, args, kwargs)
        if plan_key is not None:
            plan = self.__dict__.setdefault('_lockstep_plans', {}).get(plan_key)
            if plan is not None and (plan['verified'] or plan['observe'] == self._observe_values(observe)):
                emp = self._replay_lockstep_plan(plan, num_traces, observe, seed, offset)
                if emp is not None:
                    return emp
        ls = state.LockStepState(runner, num_traces, seed, offset)
        state._init_traces(func=self.forward, trace_mode=TraceMode.POSTERIOR,
                           inference_engine=InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK,
                           inference_network=net, observe=observe, likelihood_importance=likelihood_importance, lock_step=ls)
        runner.begin(num_traces, offset=offset)
        values = None
        n_paths = 0

        def run_path():
            """The program for the executor's current path, to its end."""
            nonlocal values, n_paths
            state._begin_trace()
            result = self.forward(*args, **kwargs)
            n_paths += 1
            if not torch.is_tensor(result):
                raise RuntimeError('lock-step importance sampling: forward() must return a per-particle tensor')
            # no queued control-flow path (and nobody waiting for this nested run): the statistics ride in this path's last pass
            last = not ls.pending and ls._nest_depth == 0
            if last and ls.rows is None and torch.is_tensor(result) and result.numel() == num_traces:
                ls.stats_values = result.as_subclass(torch.Tensor).reshape(-1)
            ls.flush(final=last)            # the deferred draw / log-weight terms of this execution (one pass)
            result = result.as_subclass(torch.Tensor).reshape(-1).to(runner.dev, torch.float32)
            if ls.rows is None:
                values = result if result.numel() == num_traces else result.expand(num_traces).clone()
            else:
                if values is None:
                    values = torch.zeros(num_traces, dtype=torch.float32, device=runner.dev)
                if ls.by_rows and result.numel() in (1, num_traces):
                    runner.copy_rows(result.contiguous(), values, ls.rows)
                else:
                    values = torch.where(ls.active, result.expand(num_traces) if result.numel() == 1 else result, values)

        def nested():      # (LockStepState.branch, while the host would wait for the device)
            if ls.next_path():
                run_path()
        ls.nest = nested
        try:
            while True:
                run_path()
                if not ls.next_path():
                    break
        finally:
            state._lock_step = None
            state._current_trace = None
            ls.memo = None          # (the call's intermediate results are released)
            ls.nest = None
        all_values, all_lw = values, ls.lw
        # the fused pass's statistics count only when they were reduced over the tensor that is returned (a draw still
        # pending at the last flush reduces over ITS values: forward() may return something else)
        same = (ls.final_stats is not None and values is not None and
                getattr(ls, 'final_stats_of', None) == (values.data_ptr(), values.numel()))
        stats = ls.final_stats if same else runner.stats(all_lw, values)
        lw = all_lw
        if int(stats['count']) != num_traces:      # non-finite log-weights are discarded like Model._traces does (model.py:64-66)
            values, lw = _drop_non_finite(values, all_lw)
            stats = runner.stats(lw, values)
        emp = Empirical.from_device(values, lw, stats)      # (host arrays are made on demand, not per call)
        emp._all_values, emp._all_log_weights = all_values, all_lw     # (full shard: the distributed gather needs fixed sizes)
        emp.num_paths = n_paths
        emp.statement_log = ls.log       # per statement index: {address: (values [n], address id)} - what each path drew
        if plan_key is not None:
            self._record_lockstep_plan(plan_key, ls, n_paths, all_values, same, observe)
        return emp

    # ---- launch plan of a static lock-step program ---------------------------------------------------------------------
    @staticmethod
    def _observe_values(observe):
        try:
            return tuple(sorted((k, float(v)) for k, v in (observe or {}).items()))
        except (TypeError, ValueError):
            return None

    # attributes the Model base class itself keeps on the instance (never read by a program as constants)
    _BASE_ATTRS = frozenset(('_inference_network', '_lockstep_plans', '_lock_step_ok', '_last_prior_resident', '_plan_code_cache',
                             '_plan_key_cache', '_plan_indirect_cache', '_plan_indirect_pairs'))

    @staticmethod
    def _fingerprint(v):
        """A cheap by-VALUE fingerprint of something a program may read as a constant, or None when there is none (an object
        with state of its own, a large or device tensor, a container of such)."""
        if v is None or isinstance(v, (bool, int, float, str, bytes)):
            return ('v', type(v).__name__, v)
        if isinstance(v, (torch.Tensor, np.ndarray)):
            if int(np.prod(tuple(v.shape))) > 16 or (isinstance(v, torch.Tensor) and v.device.type != 'cpu'):
                return None
            return ('t', tuple(v.shape), tuple(float(x) for x in np.asarray(v.detach() if isinstance(v, torch.Tensor) else v,
                                                                             dtype=np.float64).reshape(-1)))
        if isinstance(v, (tuple, list)) and len(v) <= 16:
            parts = tuple(Model._fingerprint(x) for x in v)
            return None if any(x is None for x in parts) else ('s', type(v).__name__, parts)
        return None

    def _program_reads(self):
        """Everything forward() can read besides its arguments and the observation, found statically: the code objects of
        forward and of every method / plain function it names (transitively, bounded), and for those the names their
        bytecode mentions (instance / class attributes), their closure cells and the module-global slots they name.
        Returns (codes, names, cells, globs) - cells as (name, cell), globs as (name, globals dict): the VALUES are read by
        the caller at every call - or None when the program cannot be analysed. The analysis is cached per model and
        re-validated by re-resolving every function it found (a re-bound method or a replaced __code__ starts over)."""
        import types
        fwd = getattr(self.forward, '__func__', self.forward)
        if not isinstance(fwd, types.FunctionType):
            return None
        cache = self.__dict__.get('_plan_code_cache')
        if cache is not None:
            ok = cache[0] is fwd
            if ok:
                for holder, name, fn, code in cache[1]:
                    cur = holder.get(name) if isinstance(holder, dict) else getattr(holder, name, None)
                    cur = getattr(cur, '__func__', cur)
                    if cur is not fn or fn.__code__ is not code:
                        ok = False
                        break
            if ok:
                return cache[2]
        seen, order, names = set(), [], set()
        cells, globs, found = [], [], [(type(self), 'forward', fwd, fwd.__code__)] if getattr(type(self), 'forward', None) is fwd \
            else [({'forward': fwd}, 'forward', fwd, fwd.__code__)]
        work = [fwd]
        pkg = __name__.split('.')[0]
        while work:
            fn = work.pop()
            code = fn.__code__
            if code in seen:
                continue
            if len(seen) >= 64:
                return None
            seen.add(code)
            order.append(code)
            stack = [code]
            conames = set()
            while stack:                      # nested code objects (lambdas, comprehensions, inner functions)
                c = stack.pop()
                conames.update(c.co_names)
                for k in c.co_consts:
                    if isinstance(k, types.CodeType):
                        stack.append(k)
                        order.append(k)
            names |= conames
            for name, cell in zip(code.co_freevars, fn.__closure__ or ()):
                cells.append((name, cell))
            g = fn.__globals__
            for n in conames:
                if n in g:
                    globs.append((n, g))
                    v = getattr(g[n], '__func__', g[n])
                    if isinstance(v, types.FunctionType) and not (v.__module__ or '').startswith((pkg, 'torch', 'numpy', 'math')):
                        found.append((g, n, v, v.__code__))
                        work.append(v)
                # methods reachable by name from the model's class
                m = getattr(type(self), n, None)
                m = getattr(m, '__func__', m)
                if isinstance(m, types.FunctionType) and not (m.__module__ or '').startswith(pkg + '.'):
                    found.append((type(self), n, m, m.__code__))
                    work.append(m)
        reads = (tuple(order), frozenset(names), tuple(cells), tuple(globs))
        self.__dict__['_plan_code_cache'] = (fwd, tuple(found), reads)
        return reads

    def _lockstep_plan_key(self, num_traces, observe, likelihood_importance, args, kwargs):
        """What a recorded plan is valid for: this forward() (its code objects and those of the methods / functions it names),
        the VALUES of everything those code objects can read - instance attributes (private ones included), class attributes,
        module globals, closure cells - this network (the engine's own token and address table size), this particle count and
        set of observed names. A replay does not run forward(), so whatever cannot be fingerprinted by value rules the plan
        out: None = no plan (also: call arguments, non-scalar observations, PP_IS_PLAN=0)."""
        if args or kwargs or _environ.get('PP_IS_PLAN', '1') == '0' or self._observe_values(observe) is None:
            return None
        net = self._inference_network
        eng = getattr(net, '_engine', None)
        if eng is None or eng.device.type != 'cuda':
            return None
        reads = self._program_reads()
        if reads is None:
            return None
        codes, names, cells, globs = reads
        # Fast path of a repeated call: every object the key was built from is STILL THE SAME OBJECT (the cache holds the old
        # objects, so an address cannot be reused) and each of them is immutable or a tensor whose in-place version is unchanged.
        head = (eng.token, len(eng.spec.addresses), int(num_traces), tuple(sorted(observe or {})), float(likelihood_importance), reads)
        watch = self._plan_watch(reads)
        kc = self.__dict__.get('_plan_key_cache')
        if kc is not None and kc[0] == head and watch is not None and len(kc[1]) == len(watch) and \
                all(map(_is, kc[1], watch)) and all(watch[i]._version == ver for i, ver in kc[2]):
            return kc[3]
        key = self._lockstep_plan_key_slow(eng, num_traces, observe, likelihood_importance, reads)
        stable = (type(None), bool, int, float, str, bytes, torch.Tensor, types.ModuleType, types.FunctionType,
                  types.BuiltinFunctionType, types.MethodType, type)
        if key is not None and watch is not None and all(isinstance(v, stable) or v is self for v in watch):
            self.__dict__['_plan_key_cache'] = (head, watch, [(i, v._version) for i, v in enumerate(watch) if isinstance(v, torch.Tensor)], key)
        else:
            self.__dict__.pop('_plan_key_cache', None)
        return key

    def _plan_watch(self, reads):
        """Every object `_lockstep_plan_key` looks at, in a fixed order: attribute names and values of the instance (the base
        class's own bookkeeping aside), the class attributes, module globals and closure cells the program's bytecode names."""
        codes, names, cells, globs = reads
        out = []
        base = self._BASE_ATTRS
        d = vars(self)
        for k, v in d.items():
            if k not in base:
                out.append(k)
                out.append(v)
        for k in self._plan_class_names(names):
            if k in d:
                continue
            for klass in type(self).__mro__:
                if klass is Model or klass is object:
                    break
                kv = vars(klass)
                if k in kv:
                    out.append(kv[k])
                    break
        try:
            for k, g in globs:
                out.append(g[k])
            for k, cell in cells:
                out.append(cell.cell_contents)
        except (KeyError, ValueError):
            return None
        # values read THROUGH the objects above (modules, classes, functions kept on the instance): whether there are any
        # depends only on the identity of those holders - the common "none" case is remembered and costs an identity compare
        ic = self.__dict__.get('_plan_indirect_pairs')
        if ic is not None and ic[0] is reads and len(ic[1]) == len(out) and all(map(_is, ic[1], out)):
            return out
        ind = self._plan_indirect(reads)
        if ind is None:
            self.__dict__.pop('_plan_indirect_pairs', None)
            return None
        stable = (type(None), bool, int, float, str, bytes, torch.Tensor, types.ModuleType, types.FunctionType,
                  types.BuiltinFunctionType, types.MethodType, type)
        if not ind and all(isinstance(v, stable) or v is self for v in out):
            self.__dict__['_plan_indirect_pairs'] = (reads, list(out))
        else:
            self.__dict__.pop('_plan_indirect_pairs', None)
        for _, _, v in ind:
            out.append(v)
        return out

    _PLAN_CODE_MODULES = ('math', 'torch', 'numpy', 'builtins', 'operator', 'functools', 'itertools')

    def _plan_indirect(self, reads):
        """What the program can read THROUGH an object it names: for every module or class reached from a module
        global, a closure cell or an instance attribute - `settings.PRIOR_MEAN`, `Config.MU` - the attributes of it that the
        program's bytecode names (co_names holds the name after the LOAD_GLOBAL / LOAD_ATTR); for every plain function kept
        as an instance attribute and named by the program (`self.helper = other_fn`) its code object, defaults and closure
        contents. Returns [(label, name, value)] in a fixed order, or None when something cannot be enumerated. Modules that
        are code only (math, torch, numpy, this package) are not walked."""
        import types
        codes, names, cells, globs = reads
        pkg = __name__.split('.')[0]
        cache = self.__dict__.get('_plan_indirect_cache')
        if cache is None or cache[0] is not reads:
            cache = self.__dict__['_plan_indirect_cache'] = (reads, tuple(sorted(names)))
        sorted_names = cache[1]
        out = []
        code_modules = self._PLAN_CODE_MODULES

        def walk(label, h, instance_attr):
            # (called per posterior call from the key's fast path: the common holder - a code-only module, a class of this
            # package, a plain value - costs two type checks)
            if isinstance(h, types.ModuleType):
                root = (getattr(h, '__name__', '') or '').split('.')[0]
                if root in code_modules or root == pkg:
                    return True
                d = vars(h)
                for n in sorted_names:
                    if n in d:
                        out.append((label, n, d[n]))
            elif isinstance(h, type):
                if h is type(self) or (getattr(h, '__module__', '') or '').split('.')[0] in code_modules + (pkg,):
                    return True
                for n in sorted_names:
                    for klass in h.__mro__:
                        if klass is object:
                            break
                        if n in vars(klass):
                            out.append((label, n, vars(klass)[n]))
                            break
            elif instance_attr and isinstance(h, types.FunctionType):
                out.append((label, '__code__', h.__code__))
                out.append((label, '__defaults__', h.__defaults__))
                out.append((label, '__kwdefaults__', h.__kwdefaults__))
                try:
                    for cn, cell in zip(h.__code__.co_freevars, h.__closure__ or ()):
                        out.append((label, 'cell:' + cn, cell.cell_contents))
                except ValueError:
                    return False
            return True
        try:
            for k, g in globs:
                if not walk('g:' + k, g[k], False):
                    return None
            for k, cell in cells:
                if not walk('f:' + k, cell.cell_contents, False):
                    return None
        except (KeyError, ValueError):
            return None
        base = self._BASE_ATTRS
        for k, v in vars(self).items():      # (insertion order: stable for one instance)
            if k in names and k not in base and not walk('i:' + k, v, True):
                return None
        return out

    def _plan_class_names(self, names):
        """The names of the program's bytecode that can be class attributes of the model's own classes (not members of the
        Model base): cached per (class, name set) - classes gain attributes rarely, and a new one that shadows nothing the
        program read before cannot change a recorded call."""
        cache = type(self).__dict__.get('_plan_class_names_cache')
        if cache is None or cache[0] is not names:
            own = set()
            for klass in type(self).__mro__:
                if klass is Model or klass is object:
                    break
                own.update(vars(klass))
            cache = (names, tuple(sorted(k for k in names if k in own and not hasattr(Model, k))))
            try:
                type(self)._plan_class_names_cache = cache
            except (AttributeError, TypeError):
                pass
        return cache[1]

    def _lockstep_plan_key_slow(self, eng, num_traces, observe, likelihood_importance, reads):
        import types
        codes, names, cells, globs = reads
        skip_types = (types.ModuleType, types.FunctionType, types.BuiltinFunctionType, types.MethodType, type)
        plain = []
        # instance attributes: every public one (as before) and every private one the program's bytecode names
        for k, v in sorted(vars(self).items()):
            if k in self._BASE_ATTRS or k == 'name':
                continue
            if k.startswith('_') and k not in names:
                continue
            if callable(v) and not isinstance(v, (torch.Tensor, np.ndarray)):
                if k in names and not isinstance(v, skip_types):
                    return None              # a callable object with state of its own, named by the program
                continue
            fp = self._fingerprint(v)
            if fp is None:
                return None
            plain.append(('i', k, fp))
        # class attributes the program names (constants kept on the class; not the Model base's own members)
        for k in sorted(names):
            if k in vars(self) or hasattr(Model, k):
                continue
            for klass in type(self).__mro__:
                if klass in (Model, object):
                    break
                if k in vars(klass):
                    v = vars(klass)[k]
                    if isinstance(v, (types.FunctionType, staticmethod, classmethod, property)):
                        break
                    fp = self._fingerprint(v)
                    if fp is None:
                        return None
                    plain.append(('c', k, fp))
                    break
        # module globals and closure cells: modules, functions, classes are code (keyed through `codes` where they are Python
        # functions of the user's); anything else must have a value fingerprint
        for tag, items in (('g', globs), ('f', cells)):
            for k, holder in items:
                try:
                    v = holder[k] if tag == 'g' else holder.cell_contents
                except (KeyError, ValueError):
                    return None
                if isinstance(v, skip_types) or v is self:
                    continue
                fp = self._fingerprint(v)
                if fp is None:
                    return None
                plain.append((tag, k, fp))
        # values read through a module / class / instance-attribute function the program names: by value, or no plan
        ind = self._plan_indirect(reads)
        if ind is None:
            return None
        extra_codes = []
        for label, n, v in ind:
            if isinstance(v, types.CodeType):
                extra_codes.append(v)
                continue
            if isinstance(v, (types.ModuleType, types.BuiltinFunctionType)):
                continue
            if isinstance(v, (types.FunctionType, types.MethodType, type, staticmethod, classmethod, property)) or callable(v) \
                    and not isinstance(v, (torch.Tensor, np.ndarray)):
                return None          # code (or an object with state) reached indirectly: its reads are not analysed - no plan
            fp = self._fingerprint(v)
            if fp is None:
                return None
            plain.append(('x', label, n, fp))
        codes = codes + tuple(extra_codes)
        return (eng.token, len(eng.spec.addresses), int(num_traces), tuple(sorted(observe or {})), float(likelihood_importance),
                tuple(plain), codes)

    def _record_lockstep_plan(self, key, ls, n_paths, values, stats_fused, observe):
        """After a normal lock-step call: keep its launch list when the call WAS one draw + one fused pass. A plan becomes
        `verified` (replayed for any observation values) once a second recording under DIFFERENT observation values produced
        the same constants - a program that computes distribution parameters from the observed values in Python is thereby
        replayed only for the observations it was recorded with."""
        plans = self.__dict__.setdefault('_lockstep_plans', {})
        final = getattr(ls, 'plan_final', None)
        ok = (n_paths == 1 and ls.plan_ok and ls.flushes == 1 and final is not None and final['draw'] is not None and stats_fused and
              len(ls.log) == 1 and len(ls.plan_terms) == final['n_terms'])
        if ok:
            draw = final['draw']
            vptr = draw['values'].data_ptr()
            ok = values is not None and values.data_ptr() == vptr and values.numel() == ls.n
        terms = []
        if ok:
            def src(t):
                if t is None:
                    return ('none',)
                if t.data_ptr() == vptr and t.numel() == ls.n:
                    return ('value',)
                if t.numel() == 1:
                    return ('const', t)
                return None
            for (kind, p0, s0, p1, s1), xsrc, scale in ls.plan_terms:
                a, b = src(p0), src(p1)
                if a is None or b is None or xsrc is None or xsrc[0] != 'obs' or xsrc[1] not in (observe or {}):
                    ok = False
                    break
                terms.append((int(kind), a, int(s0), b, int(s1), xsrc, float(scale)))
        if not ok:
            plans.pop(key, None)
            if len(plans) > 64:
                plans.clear()
            return
        sig = tuple((k, a[0], float(a[1]) if a[0] == 'const' else None, s0, b[0], float(b[1]) if b[0] == 'const' else None, s1,
                     x, sc) for k, a, s0, b, s1, x, sc in terms)
        (address, (_, addr_id)), = ls.log[0].items()
        net = self._inference_network
        obs_index = None
        spec = net._engine.spec
        if all(o[1] == 1 for o in spec.obs) and all(x[1] in net._obs_names for _, _, _, _, _, x, _ in terms) and \
                draw['prior'].numel() == 2 and all((a[0] != 'const' or a[1].is_cuda) and (b[0] != 'const' or b[1].is_cuda)
                                                   for _, a, _, b, _, _, _ in terms):
            obs_index = {name: i for i, name in enumerate(net._obs_names)}       # position in the observation vector
        new = dict(addr=int(draw['addr']), address=address, prior=draw['prior'], prior_term=draw['prior_term'], terms=terms, sig=sig,
                   observe=self._observe_values(observe), verified=False, obs_index=obs_index)
        old = plans.get(key)
        if old is not None and old['sig'] == sig and old['addr'] == new['addr'] and old['observe'] != new['observe']:
            new['verified'] = True       # same constants under different observations: they do not depend on the observed values
        elif old is not None and old['verified'] and old['sig'] == sig:
            new['verified'] = True
        plans[key] = new

    def _replay_lockstep_plan(self, plan, num_traces, observe, seed, offset):
        net = self._inference_network
        runner = net._is
        if plan.get('obs_index') is not None:
            # every observable is one number: the observation vector of _infer_init holds the x of every observe term
            try:
                vec = [float(observe[name]) for name in net._obs_names]
            except (KeyError, TypeError, ValueError):
                return None
            net._infer_observe = observe
            net._infer_prev_addr_id = None
            all_values, all_lw, stats = runner.run_plan(plan, vec, num_traces, offset, seed)
            values, lw = all_values, all_lw
            if int(stats['count']) != num_traces:
                values, lw = _drop_non_finite(values, all_lw)
                stats = runner.stats(lw, values)
            emp = Empirical.from_device(values, lw, stats)
            emp._all_values, emp._all_log_weights = all_values, all_lw
            emp.num_paths = 1
            emp.statement_log = [{plan['address']: (all_values, plan['addr'])}]
            emp.replayed_plan = True
            return emp
        net._infer_init(observe)                        # InferenceNetwork._infer_init (inference_network.py:141-148)
        runner.begin(num_traces, offset=offset)
        runner.step_net(plan['addr'], None)             # first statement: one shared row through the LSTM and the head
        values = torch.empty(num_traces, dtype=torch.float32, device=runner.dev)
        lw = torch.empty(num_traces, dtype=torch.float32, device=runner.dev)
        fterms = [(plan['prior_term'], values, 1.0, 4)]
        for kind, a, s0, b, s1, xsrc, scale in plan['terms']:
            p0 = values if a[0] == 'value' else (a[1] if a[0] == 'const' else None)
            p1 = values if b[0] == 'value' else (b[1] if b[0] == 'const' else None)
            x = runner._const(float(observe[xsrc[1]]))
            fterms.append(((kind, p0, s0, p1, s1), x, scale, (1 if a[0] == 'value' else 0) | (2 if b[0] == 'value' else 0)))
        stats = runner.fused(plan['addr'], plan['prior'], fterms, values, lw, overwrite=True, seed=seed, stats=True)
        runner.prev_value = runner.last_value = values
        all_values, all_lw = values, lw
        if int(stats['count']) != num_traces:
            values, lw = _drop_non_finite(values, all_lw)
            stats = runner.stats(lw, values)
        emp = Empirical.from_device(values, lw, stats)
        emp._all_values, emp._all_log_weights = all_values, all_lw
        emp.num_paths = 1
        emp.statement_log = [{plan['address']: (all_values, plan['addr'])}]
        emp.replayed_plan = True
        return emp

    def _traces_coroutines(self, num_traces, observe, map_func=None, seed=0, offset=0, likelihood_importance=1.,
                           *args, num_workers=None, **kwargs):
        """Importance sampling with the inference network for a program AS WRITTEN (`while float(s) >= 1:` ...): one
        greenlet per particle, parked at `sample` and served in address-grouped batches (pyprob_amd/coroutine.py).
        Replaces the per-particle loop of pyprob/model.py:59-71. map_func(trace) values like Model._traces; with the
        default (trace_result) numeric results are stacked into one tensor."""
        from .coroutine import CoroutineIS
        net = self._inference_network
        state._init_traces(func=self.forward, trace_mode=TraceMode.POSTERIOR,
                           inference_engine=InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK,
                           inference_network=net, observe=observe, likelihood_importance=likelihood_importance)
        # One interpreter runs ~7-11 k particles/s of a torch-scalar program; large runs shard the particles over forked
        # worker processes (the parent serves the device): num_workers / PP_IS_WORKERS, default one per 5 000 particles up
        # to 32 (forking a process that holds a device context costs ~40 ms: 64 workers were slower than 16 at 200 k
        # particles, profiles/r02_m_is_executors.log).
        import os
        if num_workers is None:
            num_workers = int(os.environ.get('PP_IS_WORKERS', 0)) or min(max(num_traces // 5000, 1),
                                                                         max((os.cpu_count() or 2) // 2, 1), 32)
        plain = map_func is None or map_func is trace_result
        if num_workers > 1:
            from .coroutine import ShardedCoroutineIS
            sh = ShardedCoroutineIS(state, self.forward, net, num_traces, num_workers, seed=seed, offset=offset,
                                    likelihood_importance=likelihood_importance, map_func=None if plain else map_func)
            try:
                results, lw, cstats = sh.run(*args, **kwargs)
            finally:
                state._current_trace = None
            if plain and isinstance(results, np.ndarray):
                values = torch.from_numpy(results).to(lw.device)
            else:
                values = list(results)
        else:
            co = CoroutineIS(state, self.forward, net, num_traces, seed=seed, offset=offset,
                             likelihood_importance=likelihood_importance)
            state._coroutine = co
            try:
                results, lw, traces = co.run(*args, **kwargs)
            finally:
                state._coroutine = None
                state._current_trace = None
            cstats = dict(rounds=co.rounds, group_calls=co.group_calls, statements=co.statements, seconds=co.seconds, workers=1)
            if plain:
                values = results
                try:
                    values = torch.stack([torch.as_tensor(r, dtype=torch.float32).reshape(()) for r in results]).to(lw.device)
                except (RuntimeError, TypeError, ValueError):
                    pass
            else:
                values = [map_func(t) for t in traces]
        if torch.is_tensor(values):
            values, lw = _drop_non_finite(values, lw)
        else:
            ok = torch.isfinite(lw).cpu().numpy()
            if not ok.all():
                values = [v for v, k in zip(values, ok) if k]
                lw = lw[torch.from_numpy(ok).to(lw.device)].contiguous()
        emp = Empirical(values=values, log_weights=lw)
        emp.finalize()
        if torch.is_tensor(values):
            emp.device_stats = net._is.stats(lw, values)
        emp.coroutine_stats = cstats
        return emp

    def prior_traces_packed(self, num_traces, obs_names, device='cpu', *args, return_types=False,
                            prior_inflation=PriorInflation.DISABLED, resident_only=False, **kwargs):
        """num_traces traces of the program in PRIOR_FOR_INFERENCE_NETWORK mode, generated TOGETHER (one execution of
        forward() per distinct control-flow path, state.PriorLockStep) and returned as ragged columns
        (trace_len, address table, address ids, values, prior parameters, observations) - what a training minibatch is
        packed from. The vectorised replacement of OnlineDataset's one-forward()-per-trace loop
        (pyprob/nn/dataset.py:50-62). The program must be lock-step safe (tensor conditions)."""
        ls = state.PriorLockStep(num_traces, device)
        state._init_traces(func=self.forward, trace_mode=TraceMode.PRIOR_FOR_INFERENCE_NETWORK, lock_step=ls,
                           prior_inflation=prior_inflation)
        try:
            while True:
                state._begin_trace()
                self.forward(*args, **kwargs)
                ls.finish_path()
                if not ls.next_path():
                    break
        finally:
            state._lock_step = None
            state._current_trace = None
        self._last_prior_resident = ls.columns_device(obs_names)      # (device chunks of single-statement programs)
        if resident_only and self._last_prior_resident is not None:
            return None          # the caller trains from the device columns: no host copy of the chunk
        return ls.columns(obs_names, return_types)

    def _lock_step_safe(self, observe, *args, **kwargs):
        """Can this program run with all particles in lock step? Decided once per model by a probe of 4 particles: a
        program that turns a sampled value into a Python scalar (`float(s)`, `.item()`, `if` on a plain tensor of several
        elements) raises there and is served one particle per forward() instead."""
        ok = getattr(self, '_lock_step_ok', None)
        if ok is None:
            import warnings
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    self._traces_lockstep(4, observe, *args, **kwargs)
                ok = True
            except Exception:   # noqa: BLE001 - any failure of the probe means "not lock-step safe"
                ok = False
            self._lock_step_ok = ok
        return ok

    def prior(self, num_traces=10, prior_inflation=PriorInflation.DISABLED, map_func=None, likelihood_importance=1.,
              *args, **kwargs):
        """pyprob/model.py:97-101: an Empirical of prior traces (or of map_func(trace))."""
        prior = self._traces(num_traces=num_traces, trace_mode=TraceMode.PRIOR, map_func=map_func,
                             prior_inflation=prior_inflation, likelihood_importance=likelihood_importance, *args, **kwargs)
        prior.rename('Prior, traces: {:,}'.format(prior.length))
        prior.add_metadata(op='prior', num_traces=num_traces, prior_inflation=str(prior_inflation),
                           likelihood_importance=likelihood_importance)
        return prior

    def prior_results(self, num_traces=10, prior_inflation=PriorInflation.DISABLED, map_func=trace_result, *args, **kwargs):
        """pyprob/model.py:103-104"""
        return self.prior(num_traces=num_traces, prior_inflation=prior_inflation, map_func=map_func, *args, **kwargs)

    def posterior(self, num_traces=10, inference_engine=InferenceEngine.IMPORTANCE_SAMPLING, map_func=None, observe=None,
                  likelihood_importance=1., *args, **kwargs):
        """pyprob/model.py:106-117 for the IS engines: an Empirical of posterior TRACES (or of map_func(trace)), one
        particle per forward() like the reference. `posterior_results` is the fast path when only the results are
        needed (all particles in lock step on the device)."""
        if inference_engine == InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK and self._inference_network is None:
            raise RuntimeError('Cannot run inference engine IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK because no '
                               'inference network for this model is available. Use learn_inference_network or '
                               'load_inference_network first.')
        net = self._inference_network if inference_engine == InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK else None
        post = self._traces(num_traces, TraceMode.POSTERIOR, inference_engine, net, map_func, observe, likelihood_importance,
                            *args, **kwargs)
        kind = 'IC' if net is not None else 'IS'
        post.rename('Posterior, {}, traces: {:,}, ESS: {:,.2f}'.format(kind, post.length, post.effective_sample_size))
        post.add_metadata(op='posterior', num_traces=num_traces, inference_engine=str(inference_engine),
                          effective_sample_size=post.effective_sample_size, likelihood_importance=likelihood_importance)
        return post

    def posterior_results(self, num_traces=10, inference_engine=InferenceEngine.IMPORTANCE_SAMPLING, observe=None,
                          lock_step=None, seed=0, offset=0, likelihood_importance=1., *args, **kwargs):
        """pyprob/model.py:106-117,180-181 for the IS engines. With the inference network, lock_step=True runs all
        particles together on the device (one forward() per control-flow path; the program's conditions must be tensor
        expressions); lock_step=False runs the program as written, one greenlet per particle, parked at `sample` and
        served in address-grouped batches (pyprob_amd/coroutine.py); 'per_trace' is the reference's loop (one particle
        per forward(), batch-1 network calls). None (default) = lock step when a probe run of a few particles shows the
        program allows it (same auto-detection as the vectorised prior of learn_inference_network), else coroutines."""
        if inference_engine == InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK:
            if self._inference_network is None:
                raise RuntimeError('Cannot run inference engine IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK because no '
                                   'inference network for this model is available. Use learn_inference_network or '
                                   'load_inference_network first.')
            if lock_step is None:
                lock_step = self._lock_step_safe(observe, *args, **kwargs)
            if lock_step == 'per_trace':        # the reference's loop: one particle per forward(), batch-1 network calls
                post = self._traces(num_traces, TraceMode.POSTERIOR, inference_engine, self._inference_network,
                                    trace_result, observe, likelihood_importance, *args, **kwargs)
            elif lock_step:
                post = self._traces_lockstep(num_traces, observe, seed=seed, offset=offset,
                                             likelihood_importance=likelihood_importance, *args, **kwargs)
            else:                               # program as written: particle coroutines, address-grouped batches
                post = self._traces_coroutines(num_traces, observe, trace_result, seed, offset, likelihood_importance,
                                               *args, **kwargs)
            post.rename('Posterior, IC, traces: {:,}, ESS: {:,.2f}'.format(post.length, post.effective_sample_size))
        else:
            post = self._traces(num_traces, TraceMode.POSTERIOR, inference_engine, None, trace_result, observe,
                                likelihood_importance, *args, **kwargs)
            post.rename('Posterior, IS, traces: {:,}, ESS: {:,.2f}'.format(post.length, post.effective_sample_size))
        return post

    def posterior_results_distributed(self, num_traces, observe=None, seed=0, likelihood_importance=1., *args, **kwargs):
        """posterior_results(IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK) with the particles sharded over the ranks of the
        initialised torch.distributed group (one process per GPU): every rank runs its contiguous shard in lock step with
        its own Philox counter range, one all-gather returns all particles to every rank. The reference's ParallelModel
        (pyprob/model.py:339-406) shards the same way over processes and merges the per-process files."""
        import torch.distributed as dist
        from .parallel import gather_particles, shard_range
        if self._inference_network is None:
            raise RuntimeError('Cannot run inference engine IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK because no '
                               'inference network for this model is available. Use learn_inference_network or '
                               'load_inference_network first.')
        offset, count = shard_range(num_traces, dist.get_rank(), dist.get_world_size())
        local = self._traces_lockstep(count, observe, seed=seed, offset=offset,
                                      likelihood_importance=likelihood_importance, *args, **kwargs)
        values, lw = gather_particles(local._all_values, local._all_log_weights, num_traces)
        values, lw = _drop_non_finite(values, lw)
        post = Empirical(values=values, log_weights=lw)
        post.device_stats = self._inference_network._is.stats(lw.contiguous(), values.contiguous())
        post.rename('Posterior, IC, traces: {:,}, ESS: {:,.2f}'.format(post.length, post.effective_sample_size))
        return post

    def learn_inference_network(self, num_traces, inference_network=InferenceNetwork.FEEDFORWARD, observe_embeddings={},
                                batch_size=64, lstm_dim=512, lstm_depth=1,
                                proposal_mixture_components=10, learning_rate_init=0.001, learning_rate_end=1e-6,
                                learning_rate_scheduler_type=None, weight_decay=0., distributed_backend=None,
                                device='cuda:0', seed=None, dataset=None, log_file_name=None, dataset_dir=None,
                                distributed_num_buckets=None, vectorised_prior=None, prior_chunk_traces=None,
                                prior_inflation=PriorInflation.DISABLED, num_traces_end=1e9, optimizer_type=Optimizer.ADAM,
                                momentum=0.9, save_file_name_prefix=None, save_every_sec=600, pre_generate_layers=False,
                                distributed_params_sync_every_iter=10000, dataloader_offline_num_workers=0,
                                stop_with_bad_loss=True, dataset_valid_dir=None, valid_every=None):
        """pyprob/model.py:186-215 (inference_network: FEEDFORWARD, the reference's default, or LSTM). `dataset_dir` = a directory written by
        `save_dataset` (packed shards, pyprob_amd/dataset.py): offline training like the reference's OfflineDataset."""
        if dataset is None and dataset_dir is not None:
            from .dataset import PackedTraceDataset
            dataset = PackedTraceDataset(dataset_dir)
        dataset_valid = None
        if dataset_valid_dir is not None:          # model.py:192-195 (a directory written by save_dataset)
            from .dataset import PackedTraceDataset
            dataset_valid = PackedTraceDataset(dataset_valid_dir)
        if dataset is None and vectorised_prior is not False and observe_embeddings:
            # online training: generate prior traces in lock step (one forward() per control-flow path and chunk) when
            # the program allows it - it must not turn sampled values into Python scalars; else one trace per forward()
            from .dataset import VectorisedOnlineDataset
            try:
                self.prior_traces_packed(8, list(observe_embeddings.keys()))
                import os
                # the chunk is drawn on the training device when there is one (pp_prior_draw; PP_PRIOR_DEVICE=0: host draws)
                gen_dev = device if (str(device).startswith('cuda') and torch.cuda.is_available() and
                                     os.environ.get('PP_PRIOR_DEVICE', '1') != '0') else 'cpu'
                dataset = VectorisedOnlineDataset(self, list(observe_embeddings.keys()),
                                                  chunk_traces=prior_chunk_traces or max(64 * batch_size, 16384),
                                                  prior_inflation=prior_inflation, device=gen_dev)
            except Exception as exc:   # noqa: BLE001 - any failure of the probe means "not lock-step safe"
                if vectorised_prior:
                    raise
                print('Prior traces are generated one forward() at a time (program is not lock-step safe: {})'.format(exc))
        if dataset is None:
            dataset = OnlineDataset(model=self, prior_inflation=prior_inflation)
        if self._inference_network is None:
            print('Creating new inference network...')
            if inference_network == InferenceNetwork.LSTM:
                cls = InferenceNetworkLSTM
            elif inference_network == InferenceNetwork.FEEDFORWARD:
                cls = InferenceNetworkFeedForward
            else:
                raise ValueError('Unknown inference_network: {}'.format(inference_network))   # model.py:203-204
            self.__dict__.pop('_lockstep_plans', None)
            self._inference_network = cls(model=self, observe_embeddings=observe_embeddings, lstm_dim=lstm_dim,
                                          lstm_depth=lstm_depth, proposal_mixture_components=proposal_mixture_components,
                                          device=device, seed=seed)
            if pre_generate_layers:        # model.py:205-209: all layers before the first step (and no _polymorph later)
                self._inference_network._pre_generate_layers(dataset, batch_size=batch_size,
                                                             save_file_name_prefix=save_file_name_prefix)
        else:
            print('Continuing to train existing inference network...')
        self._inference_network.optimize(num_traces=num_traces, dataset=dataset, batch_size=batch_size,
                                         num_traces_end=num_traces_end, save_file_name_prefix=save_file_name_prefix,
                                         save_every_sec=save_every_sec, stop_with_bad_loss=stop_with_bad_loss,
                                         distributed_params_sync_every_iter=distributed_params_sync_every_iter,
                                         dataset_valid=dataset_valid, valid_every=valid_every,
                                         learning_rate_init=learning_rate_init, learning_rate_end=learning_rate_end,
                                         learning_rate_scheduler_type=learning_rate_scheduler_type,
                                         weight_decay=weight_decay, distributed_backend=distributed_backend,
                                         log_file_name=log_file_name, distributed_num_buckets=distributed_num_buckets,
                                         optimizer_type=optimizer_type, momentum=momentum)

    def save_dataset(self, dataset_dir, num_traces, num_traces_per_file, prior_inflation=PriorInflation.DISABLED,
                     *args, **kwargs):
        """pyprob/model.py:227-232: prior traces for offline training, as packed shards (pyprob_amd/dataset.py)."""
        from .dataset import save_dataset
        return save_dataset(self, dataset_dir, num_traces, num_traces_per_file, *args, prior_inflation=prior_inflation,
                            **kwargs)

    def save_inference_network(self, file_name):
        if self._inference_network is None:
            raise RuntimeError('The model has no trained inference network.')
        self._inference_network._save(file_name)

    def load_inference_network(self, file_name, device='cuda:0'):
        self._inference_network = InferenceNetworkLSTM._load(file_name, device=device)
        self._inference_network._model = self
        self.__dict__.pop('_lockstep_plans', None)      # (plans are keyed on the engine's token; the old ones are dead weight)
