"""Importance sampling with the inference network for UNMODIFIED programs: particle coroutines.

The lock-step executor (state.LockStepState) needs programs whose conditions are tensor expressions. The reference's
own programs are plain Python - `while float(s) >= 1:` (tests/test_inference.py:262), `.item()`, `if` on a scalar - and
pyprob serves them one particle at a time: one `forward()` per particle, one batch-1 network evaluation per
`pyprob.sample` (pyprob/model.py:59-71, pyprob/state.py:203-219). Here every particle runs `forward()` in its own
greenlet, as written, and PARKS inside `sample`: when every live particle is parked (or finished) the
scheduler serves the pending statements in address-grouped batches -

    per group (address, previous address):  one pp_is_step  (LSTM step with the particles' gathered states + proposal
                                            head + draw + log q), one prior log_prob kernel, lw[rows] += log p - log q
    per round:                              the likelihood terms queued by `observe` since the last round, one kernel per
                                            family, lw[rows] += likelihood_importance * log p(y | .)

- and resumes the particles with their values. The per-particle log-weight lives on the device for the whole run and
is the same fp32 accumulator the lock-step path uses (pinned on the reference's records in tests/test_gpu_logweight.py).
What stays per particle is the program's own Python.

The executor talks to the network through `ISRunner` (pp_is_step / pp_logweight_accumulate) only; `backend` may be any
object with that interface (the CPU tests drive it with a stand-in built on the oracle).
"""
import time
import warnings

import numpy as np
import torch

from .packed import distribution_params


class _Particle:
    __slots__ = ('pid', 'glet', 'request', 'reply', 'trace', 'result', 'done', 'prev_host_value')

    def __init__(self, pid):
        self.pid = pid
        self.glet = None
        self.request = None            # (address id, previous address id, prior distribution) while parked
        self.reply = None              # (value, prior log_prob) from the scheduler
        self.trace = None
        self.result = None
        self.done = False
        self.prev_host_value = None    # value of a previous variable that was drawn on the host (unknown address)


class _Params:
    """Per-particle parameter columns of one statement group, in the duck type ISRunner.dist_term reads."""
    FIELDS = {'Normal': ('mean', 'stddev'), 'Uniform': ('low', 'high'), 'Poisson': ('rate',), 'Bernoulli': ('probs',),
              'Categorical': ('probs',)}

    @classmethod
    def from_columns(cls, name, columns):
        """columns: dict field -> array (what `columns()` returns; several of them concatenate row-wise)."""
        self = cls.__new__(cls)
        self.name = name
        for k, v in columns.items():
            setattr(self, k, v)
        if name == 'Categorical':
            self.num_categories = int(self.probs.shape[1])
        return self

    def columns(self):
        return {k: getattr(self, k) for k in self.FIELDS[self.name]}

    def __init__(self, name, dists):
        self.name = name
        if name == 'Normal':
            self.mean = np.array([float(d.mean) for d in dists], np.float32)
            self.stddev = np.array([float(d.stddev) for d in dists], np.float32)
        elif name == 'Uniform':
            self.low = np.array([float(d.low) for d in dists], np.float32)
            self.high = np.array([float(d.high) for d in dists], np.float32)
        elif name == 'Poisson':
            self.rate = np.array([float(d.rate) for d in dists], np.float32)
        elif name == 'Bernoulli':
            self.probs = np.array([float(d.probs) for d in dists], np.float32)
        elif name == 'Categorical':
            self.num_categories = int(dists[0].num_categories)
            self.probs = np.stack([np.asarray(d.probs.detach().reshape(-1).tolist(), np.float32) for d in dists])
        else:
            raise RuntimeError('Distribution currently unsupported: {}'.format(name))


def _is_scalar(v):
    return not hasattr(v, 'numel') or v.numel() == 1


class ParticleScheduler:
    """N particle greenlets over one ISRunner: particles park with a request (address id, previous address id, prior
    distribution); when all live particles are parked the scheduler serves the requests in groups of equal (address,
    previous address) - one pp_is_step per group on the gathered LSTM rows - and resumes them. Subclasses decide what a
    served particle receives (`_reply`) and what else happens per group (`_after_group`)."""

    def __init__(self, runner, spec, num_traces, seed=0, offset=0):
        import greenlet
        self._greenlet = greenlet
        self.runner = runner
        self.spec = spec
        self.n = int(num_traces)
        self.seed, self.offset = int(seed), int(offset)
        self.dev = runner.dev
        self.last_value = torch.zeros(self.n, dtype=torch.float32, device=self.dev)
        self.current = None
        self.hub = None
        self.rounds = 0
        self.group_calls = 0
        self.statements = 0
        self.seconds = 0.0

    def resolve(self, address, prev_address, min_train_iterations=None):
        """(address id, previous address id) of a statement, or None when the network has no layers for the address or
        for the previous one (the prior is then the proposal, inference_network_lstm.py:100-104, 132-134) - or when the
        address's proposal layer has trained for fewer than `min_train_iterations` iterations (:108-112)."""
        spec = self.spec
        a = spec.address_id.get(address)
        if a is None:
            return None
        if min_train_iterations is not None and spec.addresses[a].total_train_iterations < min_train_iterations:
            return None
        if spec.feedforward or prev_address is None:      # inference_network_feedforward.py:52-66: no previous variable
            return a, None
        prev_a = spec.address_id.get(prev_address)
        return None if prev_a is None else (a, prev_a)

    def park(self, a, prev_a, distribution):
        """Called inside a particle: wait for this statement to be served; returns the reply."""
        p = self.current
        p.request = (a, prev_a, distribution)
        self.hub.switch()
        reply, p.reply = p.reply, None
        return reply

    def run_particles(self, particle_main):
        """particle_main(particle) runs one trace inside the particle's greenlet. Returns the particles."""
        g = self._greenlet
        self.hub = g.getcurrent()
        self._begin()
        particles = [_Particle(i) for i in range(self.n)]
        t0 = time.time()
        for p in particles:                        # every particle runs to its first controlled sample (or to the end)
            p.glet = g.greenlet(particle_main, parent=self.hub)
            self.current = p
            p.glet.switch(p)
        parked = [p for p in particles if not p.done]
        while True:
            self._between_rounds()
            if not parked:
                break
            self._serve(parked)
            nxt = []
            for p in parked:
                self.current = p
                p.glet.switch()
                if not p.done:
                    nxt.append(p)
            parked = nxt
            self.rounds += 1
        self.current = None
        self.seconds = time.time() - t0
        return particles

    def _begin(self):
        self.runner.begin(self.n, offset=self.offset)
        self.runner.state_rows = self.n            # (per-particle rows from the start: groups gather / scatter them)
        # h0 = c0 = 0 per trace (inference_network_lstm.py:89-91), also for a particle whose FIRST controlled sample fell
        # back to the prior (unknown address) and is first served at a later statement with a known previous address
        self.runner.h.zero_()
        self.runner.c.zero_()

    def _between_rounds(self):
        pass

    def _serve(self, parked):
        """All pending sample statements, grouped by (address, previous address): one pp_is_step per group."""
        runner = self.runner
        groups = {}
        for p in parked:
            a, prev_a, _ = p.request
            groups.setdefault((a, prev_a), []).append(p)
        order = sorted(groups.items(), key=lambda kv: (kv[0][0], -1 if kv[0][1] is None else kv[0][1]))
        for gi, ((a, prev_a), members) in enumerate(order):
            m = len(members)
            pids = np.fromiter((p.pid for p in members), np.int64, m)
            rows = torch.from_numpy(pids).to(self.dev)
            host_prev = [(k, p.prev_host_value) for k, p in enumerate(members) if p.prev_host_value is not None]
            if host_prev:          # previous variable was drawn from the prior on the host (unknown address)
                idx = torch.tensor([pids[k] for k, _ in host_prev], device=self.dev)
                self.last_value.index_copy_(0, idx, torch.tensor([v for _, v in host_prev], dtype=torch.float32, device=self.dev))
                for k, _ in host_prev:
                    members[k].prev_host_value = None
            dists = [p.request[2] for p in members]
            head = np.asarray([distribution_params(d) for d in dists], np.float32).reshape(m, 2)
            prior = torch.from_numpy(head).to(self.dev)
            runner.prev_value = self.last_value
            seed = self.seed + 7919 * self.rounds + 104729 * gi
            value, logq = runner.step_rows(rows, a, prev_a, prior, seed=seed, prior_compact=True)
            self.last_value.index_copy_(0, rows, value)
            replies = self._after_group(a, rows, dists, value, logq)
            for p, r in zip(members, replies):
                p.reply = r
                p.request = None
            self.group_calls += 1
            self.statements += m

    def _after_group(self, a, rows, dists, value, logq):
        """Replies of a served group: (value, log q) per particle, as 0-d host tensors (ONE device-to-host copy)."""
        host = torch.stack([value, logq]).cpu()
        return list(zip(host[0].unbind(0), host[1].unbind(0)))


class CoroutineIS(ParticleScheduler):
    """One posterior run of `num_traces` particle coroutines over the trace runtime `state` (pyprob_amd.state), with the
    whole log-weight on the device. `run()` returns (results, log_weights [n] device tensor, traces): results[i] =
    forward()'s return value of particle i."""

    def __init__(self, state, forward, network, num_traces, seed=0, offset=0, likelihood_importance=1.0):
        super().__init__(network._is, network._engine.spec, num_traces, seed, offset)
        self.state = state
        self.forward = forward
        self.net = network
        self.scale = float(likelihood_importance)
        self.lw = torch.zeros(self.n, dtype=torch.float32, device=self.dev)
        self.likelihoods = []          # (pid, distribution, value) queued by observe since the last round

    # ---- called from state.sample / state.observe inside a particle ------------------------------------------------
    def sample(self, distribution, base, addr, instance, name):
        """The IC branch of state.sample (state.py:203-219) for the particle that is running: park until the scheduler
        has served this statement, then record the variable like the reference does."""
        from .trace import Variable
        state = self.state
        prev = state._current_trace_previous_variable
        ids = self.resolve(addr, None if prev is None else prev.address,
                           getattr(state, '_current_trace_inference_network_proposal_min_train_iterations', None))
        if ids is None:
            # no proposal layers for this address (or the previous one): the prior is the proposal and
            # log p - log q = 0 (inference_network_lstm.py:100-104, 132-134); the LSTM state is not advanced
            warnings.warn('Using prior. No proposal for address: {}'.format(addr))
            value = distribution.sample()
            if value.dim() > 0:
                value = value[0]
            variable = Variable(distribution=distribution, value=value, address_base=base, address=addr, instance=instance,
                                log_prob=distribution.log_prob(value, sum=True), log_importance_weight=0.0, control=True,
                                name=name)
            self.current.prev_host_value = float(value)
            state._current_trace.add(variable)
            state._current_trace_previous_variable = variable
            return variable.value
        variable = Variable(distribution=distribution, value=None, address_base=base, address=addr, instance=instance,
                            log_prob=0., control=True, name=name)
        ctx = (state._current_trace, state._current_trace_execution_start)
        reply = self.park(ids[0], ids[1], distribution)     # ---- parked; the scheduler serves the statement ----
        state._current_trace, state._current_trace_execution_start = ctx
        variable.value, variable.log_prob = reply
        state._current_trace.add(variable)
        state._current_trace_previous_variable = variable      # (no other particle runs between here and the next park)
        return variable.value

    def observe(self, distribution, value):
        """state.observe's weight term (state.py:147-149), deferred to the next round's likelihood kernels."""
        self.likelihoods.append((self.current.pid, distribution, value))

    # ---- scheduler -------------------------------------------------------------------------------------------------
    def run(self, *args, **kwargs):
        state = self.state

        def particle_main(p):
            state._begin_trace()
            result = self.forward(*args, **kwargs)
            p.trace = state._end_trace(result)
            p.result = result
            p.done = True
        particles = self.run_particles(particle_main)
        state._current_trace = None
        lw_host = self.lw.cpu().numpy().astype(np.float64)
        for p, w in zip(particles, lw_host):
            # every weight term of a served statement lives in the device accumulator (trace.py:123-125 on the device);
            # Trace.end summed the host-side ones (prior-as-proposal fallbacks: 0)
            p.trace.log_importance_weight = float(w)
        return [p.result for p in particles], self.lw, [p.trace for p in particles]

    def _after_group(self, a, rows, dists, value, logq):
        runner = self.runner
        term = runner.dist_term(_Params(self.spec.addresses[a].dist_name, dists))
        prior_lp = runner.log_prob(term, value)
        self.lw.index_add_(0, rows, prior_lp - logq)                       # state.py:211-217
        host = torch.stack([value, prior_lp]).cpu()                        # ONE device-to-host copy per group
        return list(zip(host[0].unbind(0), host[1].unbind(0)))

    def _between_rounds(self):
        self._flush_likelihoods()

    def _flush_likelihoods(self):
        """lw[rows] += likelihood_importance * log p(y | .) for the observes since the last round, one kernel per family."""
        if not self.likelihoods:
            return
        runner = self.runner
        by_family = {}
        for pid, d, v in self.likelihoods:
            by_family.setdefault(d.name if _is_scalar(v) else '__host__', []).append((pid, d, v))
        self.likelihoods = []
        for name, items in by_family.items():
            dists = [d for _, d, _ in items]
            try:      # vector-valued observations (state.py:147-149 sums log_prob over the event) are scored on the host
                term = None if name == '__host__' else runner.dist_term(_Params(name, dists))
            except RuntimeError:
                term = None
            rows = torch.tensor([pid for pid, _, _ in items], dtype=torch.int64, device=self.dev)
            if term is None:       # a family without a device kernel: scored on the host like the reference
                lp = torch.tensor([float(d.log_prob(v, sum=True)) for _, d, v in items], dtype=torch.float32, device=self.dev)
            else:
                x = torch.tensor([float(v) for _, _, v in items], dtype=torch.float32, device=self.dev)
                lp = runner.log_prob(term, x)
            self.lw.index_add_(0, rows, lp, alpha=self.scale)


# ---- particle shards in worker processes ---------------------------------------------------------------------------
# One Python thread runs ~7 k particles/s of a torch-scalar program (the program's own interpreter time: ~140 us per
# GUMM particle). The particles are independent, so the host side shards like the reference's ParallelModel
# (pyprob/model.py:339-406): W forked workers run the greenlets of their contiguous particle range and talk to the parent
# over pipes; the parent owns the device - per round it merges the workers' parked statements by (address, previous
# address), serves every group with ONE pp_is_step over the gathered LSTM rows, adds the weight terms, and sends the
# values back. Workers never touch the device.
class _WorkerScheduler(CoroutineIS):
    """CoroutineIS inside a forked worker: same trace-runtime hooks, but `_serve` ships the round to the parent."""

    def __init__(self, state, forward, spec, lo, hi, conn, feed_forward_seed):
        import greenlet
        self._greenlet = greenlet
        self.state, self.forward, self.spec = state, forward, spec
        self.n, self.lo = hi - lo, lo
        self.conn = conn
        self.current = self.hub = None
        self.rounds = self.group_calls = self.statements = 0
        self.seconds = 0.0
        self.likelihoods = []
        self.runner = None

    def _begin(self):
        pass

    def _likelihood_payload(self):
        by_family = {}
        for pid, d, v in self.likelihoods:
            by_family.setdefault(d.name if _is_scalar(v) else '__host__', []).append((pid, d, v))
        self.likelihoods = []
        out = []
        for name, items in by_family.items():
            pids = np.array([self.lo + pid for pid, _, _ in items], np.int64)
            if name == '__host__':      # vector-valued observations: log p summed over the event here, added by the parent
                lp = np.array([float(d.log_prob(v, sum=True)) for _, d, v in items], np.float32)
                out.append((name, {}, lp, pids))
                continue
            cols = _Params(name, [d for _, d, _ in items]).columns()
            out.append((name, cols, np.array([float(v) for _, _, v in items], np.float32), pids))
        return out

    def _between_rounds(self):
        pass          # the queued likelihood terms travel with the next message

    def _serve(self, parked):
        groups = {}
        for p in parked:
            a, prev_a, _ = p.request
            groups.setdefault((a, prev_a), []).append(p)
        payload, order = [], []
        for (a, prev_a), members in groups.items():
            dists = [p.request[2] for p in members]
            head = np.asarray([distribution_params(d) for d in dists], np.float32).reshape(len(members), 2)
            prev_host = np.array([np.nan if p.prev_host_value is None else p.prev_host_value for p in members], np.float32)
            for p in members:
                p.prev_host_value = None
            payload.append((a, prev_a, np.array([self.lo + p.pid for p in members], np.int64), head,
                            _Params(self.spec.addresses[a].dist_name, dists).columns(), prev_host))
            order.append(members)
        self.conn.send(dict(done=False, groups=payload, likelihoods=self._likelihood_payload()))
        replies = self.conn.recv()
        for members, (vals, lps) in zip(order, replies):
            v, l = torch.from_numpy(vals).unbind(0), torch.from_numpy(lps).unbind(0)
            for k, p in enumerate(members):
                p.reply = (v[k], l[k])
                p.request = None
            self.statements += len(members)
        self.group_calls += len(order)


def _run_shard(conn, worker, lo, hi, state, forward, spec, map_func, seed, args, kwargs):
    """One particle shard inside a worker process: the greenlets of particles [lo, hi), every round shipped to the parent."""
    try:
        torch.manual_seed(seed + 7919 * (worker + 1))       # (prior-as-proposal fallbacks draw on the host)
        sched = _WorkerScheduler(state, forward, spec, lo, hi, conn, seed)
        state._coroutine = sched

        def particle_main(p):
            state._begin_trace()
            result = forward(*args, **kwargs)
            p.trace = state._end_trace(result)
            p.result = result
            p.done = True
        particles = sched.run_particles(particle_main)
        if map_func is None:
            try:
                results = np.array([float(p.result) for p in particles], np.float32)
            except (TypeError, ValueError):
                results = [p.result for p in particles]
        else:
            results = [map_func(p.trace) for p in particles]
        conn.send(dict(done=True, groups=[], likelihoods=sched._likelihood_payload(), results=results,
                       stats=(sched.rounds, sched.group_calls, sched.statements)))
        conn.recv()      # the parent's acknowledgement: the pipe is drained before the next job / the end of the process
        return True
    except BaseException as exc:      # noqa: BLE001 - reported to the parent, which raises it
        import traceback
        try:
            conn.send(dict(done=True, error='%s\n%s' % (exc, traceback.format_exc())))
        except Exception:  # noqa: BLE001
            pass
        return False
    finally:
        state._coroutine = None


def _worker_main(conn, worker, lo, hi, state, forward, spec, map_func, seed, args, kwargs):
    """A worker forked for ONE posterior call."""
    import os
    _child_hardening()
    try:
        torch.set_num_threads(1)
        _run_shard(conn, worker, lo, hi, state, forward, spec, map_func, seed, args, kwargs)
    finally:
        os._exit(0)       # no destructors of the parent's device state in the child


# ---- what a posterior call sets in the trace runtime --------------------------------------------------------------------
# state._init_traces (pyprob/state.py:296-336) writes module globals: the observed values, the trace mode, the inference
# engine, likelihood_importance ... A worker forked for an EARLIER call holds the values of THAT call, so every job carries
# a snapshot of them (and of the model's plain hyper-parameter attributes) and the worker installs it before it runs the
# shard: a second posterior_results(observe=other) must score the new y in observe() and in observed samples.
_RUNTIME_GLOBALS = ('_trace_mode', '_inference_engine', '_prior_inflation', '_likelihood_importance',
                    '_current_trace_root_function_name', '_current_trace_observed_variables', '_address_dictionary',
                    '_current_trace_inference_network_proposal_min_train_iterations', '_lock_step')


def _plain_value(v):
    if v is None or isinstance(v, (bool, int, float, str, np.generic)):
        return True
    if isinstance(v, torch.Tensor):
        return v.numel() <= 64 and not v.is_cuda
    if isinstance(v, np.ndarray):
        return v.size <= 64
    return False


def _runtime_snapshot(state, forward):
    snap = {k: getattr(state, k) for k in _RUNTIME_GLOBALS if hasattr(state, k)}
    obs = snap.get('_current_trace_observed_variables')
    if isinstance(obs, dict):      # device tensors do not cross a pipe; the workers score on the host
        snap['_current_trace_observed_variables'] = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v)
                                                     for k, v in obs.items()}
    model = getattr(forward, '__self__', None)
    attrs = {}
    if model is not None and hasattr(model, '__dict__'):
        attrs = {k: v for k, v in vars(model).items() if not k.startswith('_') and _plain_value(v)}
    return snap, attrs


def _runtime_install(state, forward, snapshot):
    snap, attrs = snapshot
    for k, v in snap.items():
        setattr(state, k, v)
    model = getattr(forward, '__self__', None)
    if model is not None:
        for k, v in attrs.items():
            try:
                setattr(model, k, v)
            except AttributeError:
                pass


# ---- persistent workers ------------------------------------------------------------------------------------------------
# Forking a worker copies the page tables of a process that holds a HIP context: 16 workers cost ~1 s, 64 workers 2.5 s
# per posterior call (tools/gumm_is_bench.py) - more than the particles themselves. The pool forks the workers of a
# (program, worker count) pair ONCE; a posterior call sends each worker its job (particle range, seed, network
# description, arguments) and the worker goes back to waiting afterwards. The workers hold the PROGRAM as it was when the
# pool was forked (the network lives in the parent): `close_worker_pools()` after changing the model object, PP_IS_POOL=0
# to fork per call as before.
def _child_hardening(persistent=False):
    """First thing in a forked particle worker. The parent's heap holds device tensors (also in garbage cycles waiting for
    a collection); a collection in the child that examined them would run their destructors - HIP calls in a forked
    process, which the ROCm runtime does not survive (a worker that dies without a message: EOFError in the parent).
    A fork-per-call worker simply never collects (its own garbage is bounded by its one job). A POOL worker serves many
    posterior calls: it freezes the inherited heap - frozen objects are never examined by a collection, so the parent's
    device tensors stay untouched - and then collects its own cycles (traces, frames, greenlets) between jobs."""
    import gc
    gc.disable()
    if persistent:
        gc.freeze()


def _pool_worker_main(conn, worker, state, forward, inherited):
    import gc
    import os
    _child_hardening(persistent=True)
    try:
        for c in inherited:      # pipe ends of the workers forked before this one
            try:
                c.close()
            except OSError:
                pass
        torch.set_num_threads(1)
        import pickle
        while True:
            try:
                job = conn.recv()
            except EOFError:
                break
            if job is None:
                break
            lo, hi, spec, seed, blob = job
            map_func, args, kwargs, runtime = pickle.loads(blob)
            _runtime_install(state, forward, runtime)       # this call's observe / trace mode / engine, not the fork's
            _run_shard(conn, worker, lo, hi, state, forward, spec, map_func, seed, args, kwargs)
            del job, blob, map_func, args, kwargs, runtime
            gc.collect()         # this job's cycles only: everything inherited at the fork is frozen
    finally:
        os._exit(0)


class _WorkerPool:
    def __init__(self, state, forward, workers):
        import multiprocessing as mp
        ctx = mp.get_context('fork')          # the model is an arbitrary user object: inherited, not pickled
        self.state, self.forward, self.workers = state, forward, workers
        import weakref
        model = getattr(forward, '__self__', None)
        try:      # id() of a freed model can be re-used by a new one: the pool is valid for THIS object only
            self.model_ref = weakref.ref(model) if model is not None else None
        except TypeError:
            self.model_ref = None
        self.conns, self.procs = [], []
        import gc
        gc.collect()          # device tensors in garbage cycles are freed HERE, by the process that owns the device ...
        gc.freeze()           # ... and what is alive now is never examined by a collector in a child
        try:
            for w in range(workers):
                parent, child = ctx.Pipe()
                pr = ctx.Process(target=_pool_worker_main, args=(child, w, state, forward, list(self.conns)), daemon=True)
                pr.start()
                child.close()
                self.conns.append(parent)
                self.procs.append(pr)
        finally:
            gc.unfreeze()

    def healthy(self):
        return all(pr.is_alive() for pr in self.procs)

    def owns(self, forward):
        model = getattr(forward, '__self__', None)
        if self.model_ref is None:
            return model is None or not hasattr(model, '__weakref__')
        return self.model_ref() is model

    def close(self, kill=False):
        for c in self.conns:
            try:
                if not kill:
                    c.send(None)
                c.close()
            except (OSError, BrokenPipeError):
                pass
        for pr in self.procs:
            pr.join(timeout=0.2 if kill else 5)
            if pr.is_alive():
                pr.terminate()
        self.conns, self.procs = [], []


_POOLS = {}


def _pool_key(state, forward, workers):
    return (id(state), id(getattr(forward, '__self__', None)), id(getattr(forward, '__func__', forward)), int(workers))


def _get_pool(state, forward, workers):
    import os
    if os.environ.get('PP_IS_POOL', '1') == '0':
        return None
    key = _pool_key(state, forward, workers)
    pool = _POOLS.get(key)
    if pool is not None and not (pool.healthy() and pool.owns(forward)):
        pool.close(kill=True)      # dead workers, or workers forked from another (freed) model at the same address
        pool = None
    if pool is None:
        if not _POOLS:
            import atexit
            atexit.register(close_worker_pools)
        pool = _POOLS[key] = _WorkerPool(state, forward, workers)
    return pool


def close_worker_pools():
    """Stop the persistent particle workers (they hold the program objects as they were when first used)."""
    for key in list(_POOLS):
        _POOLS.pop(key).close()


class ShardedCoroutineIS:
    """posterior run of `num_traces` particle coroutines spread over `num_workers` forked processes; the parent serves the
    device. `run()` returns (results, log_weights [n] device tensor, stats)."""

    def __init__(self, state, forward, network, num_traces, num_workers, seed=0, offset=0, likelihood_importance=1.0,
                 map_func=None):
        self.state, self.forward, self.net = state, forward, network
        self.runner, self.spec = network._is, network._engine.spec
        self.n, self.workers = int(num_traces), max(1, min(int(num_workers), int(num_traces)))
        self.seed, self.offset, self.scale = int(seed), int(offset), float(likelihood_importance)
        self.map_func = map_func
        self.dev = self.runner.dev

    def run(self, *args, **kwargs):
        import multiprocessing as mp
        from .parallel import shard_range
        ctx = mp.get_context('fork')          # the model is an arbitrary user object: inherited, not pickled
        runner = self.runner
        runner.begin(self.n, offset=self.offset)
        runner.state_rows = self.n
        runner.h.zero_()       # (see ParticleScheduler._begin)
        runner.c.zero_()
        lw = torch.zeros(self.n, dtype=torch.float32, device=self.dev)
        last_value = torch.zeros(self.n, dtype=torch.float32, device=self.dev)
        conns, procs, bounds = [], [], []
        t0 = time.time()
        pool, blob = None, None
        try:      # persistent workers need the per-call arguments as bytes
            import pickle
            runtime = _runtime_snapshot(self.state, self.forward)
            try:
                blob = pickle.dumps((self.map_func, args, kwargs, runtime))
            except Exception:  # noqa: BLE001 - e.g. a lambda as map_func
                import cloudpickle
                blob = cloudpickle.dumps((self.map_func, args, kwargs, runtime))
            pool = _get_pool(self.state, self.forward, self.workers)
        except Exception:  # noqa: BLE001 - not picklable at all: fork per call (arguments inherited)
            pool = None
        if pool is not None:
            conns = pool.conns
            for w in range(self.workers):
                lo, cnt = shard_range(self.n, w, self.workers)
                conns[w].send((lo, lo + cnt, self.spec, self.seed, blob))
                bounds.append((lo, lo + cnt))
        else:
            import gc
            gc.collect()
            gc.freeze()           # (see _WorkerPool.__init__)
            try:
                for w in range(self.workers):
                    lo, cnt = shard_range(self.n, w, self.workers)
                    parent, child = ctx.Pipe()
                    pr = ctx.Process(target=_worker_main, args=(child, w, lo, lo + cnt, self.state, self.forward, self.spec,
                                                                self.map_func, self.seed, args, kwargs), daemon=True)
                    pr.start()
                    child.close()
                    conns.append(parent); procs.append(pr); bounds.append((lo, lo + cnt))
            finally:
                gc.unfreeze()
        live = set(range(self.workers))
        results = [None] * self.workers
        stats = [0, 0, 0]
        rounds = 0
        try:
            tm = dict(wait=0.0, serve=0.0, send=0.0)      # parent's time: waiting for the workers / device + merge / replies
            while live:
                tq = time.time()
                msgs = {w: conns[w].recv() for w in sorted(live)}
                tm['wait'] += time.time() - tq
                tq = time.time()
                for w, m in msgs.items():
                    if m.get('error'):
                        raise RuntimeError('particle worker %d failed: %s' % (w, m['error']))
                # likelihood terms queued by the workers since their previous message: one kernel per family
                fam = {}
                for m in msgs.values():
                    for name, cols, x, pids in m['likelihoods']:
                        fam.setdefault(name, []).append((cols, x, pids))
                for name, parts in fam.items():
                    cols = {k: np.concatenate([p[0][k] for p in parts]) for k in parts[0][0]}
                    x = torch.from_numpy(np.concatenate([p[1] for p in parts])).to(self.dev)
                    rows = torch.from_numpy(np.concatenate([p[2] for p in parts])).to(self.dev)
                    lp = x if name == '__host__' else runner.log_prob(runner.dist_term(_Params.from_columns(name, cols)), x)
                    lw.index_add_(0, rows, lp, alpha=self.scale)
                # parked statements of all workers, merged by (address, previous address)
                merged = {}
                for w, m in msgs.items():
                    for gi, (a, prev_a, pids, head, cols, prev_host) in enumerate(m['groups']):
                        merged.setdefault((a, prev_a), []).append((w, gi, pids, head, cols, prev_host))
                replies = {w: [None] * len(m['groups']) for w, m in msgs.items()}
                order = sorted(merged.items(), key=lambda kv: (kv[0][0], -1 if kv[0][1] is None else kv[0][1]))
                for k, ((a, prev_a), parts) in enumerate(order):
                    pids = np.concatenate([p[2] for p in parts])
                    rows = torch.from_numpy(pids).to(self.dev)
                    prev_host = np.concatenate([p[5] for p in parts])
                    known = ~np.isnan(prev_host)
                    if known.any():      # previous values that were drawn on the host (unknown address)
                        last_value.index_copy_(0, rows[torch.from_numpy(known).to(self.dev)],
                                               torch.from_numpy(prev_host[known]).to(self.dev))
                    prior = torch.from_numpy(np.concatenate([p[3] for p in parts])).to(self.dev)
                    runner.prev_value = last_value
                    value, logq = runner.step_rows(rows, a, prev_a, prior, seed=self.seed + 7919 * rounds + 104729 * k,
                                                   prior_compact=True)
                    last_value.index_copy_(0, rows, value)
                    cols = {c: np.concatenate([p[4][c] for p in parts]) for c in parts[0][4]}
                    term = runner.dist_term(_Params.from_columns(self.spec.addresses[a].dist_name, cols))
                    prior_lp = runner.log_prob(term, value)
                    lw.index_add_(0, rows, prior_lp - logq)
                    host = torch.stack([value, prior_lp]).cpu().numpy()
                    pos = 0
                    for w, gi, p_ids, _, _, _ in parts:
                        m_ = len(p_ids)
                        replies[w][gi] = (host[0, pos:pos + m_].copy(), host[1, pos:pos + m_].copy())
                        pos += m_
                    stats[1] += 1
                tm['serve'] += time.time() - tq
                tq = time.time()
                for w, m in msgs.items():
                    if m['done']:
                        results[w] = m['results']
                        stats[2] += m['stats'][2]
                        conns[w].send(None)
                        live.discard(w)
                    else:
                        conns[w].send(replies[w])
                tm['send'] += time.time() - tq
                rounds += 1
        except BaseException:
            if pool is not None:      # workers may be mid-protocol: this pool cannot be reused
                _POOLS.pop(_pool_key(self.state, self.forward, self.workers), None)
                pool.close(kill=True)
            raise
        finally:
            if pool is None:
                for pr in procs:
                    pr.join(timeout=5)
                    if pr.is_alive():
                        pr.terminate()
                for c in conns:
                    c.close()
        stats[0] = rounds
        merged_results = []
        if all(isinstance(r, np.ndarray) for r in results):
            merged_results = np.concatenate(results)
        else:
            for r in results:
                merged_results.extend(list(r))
        return merged_results, lw, dict(rounds=stats[0], group_calls=stats[1], statements=stats[2], seconds=time.time() - t0,
                                        workers=self.workers, parent_seconds=tm)
