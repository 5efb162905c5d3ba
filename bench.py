#!/usr/bin/env python3
"""Benchmark of the north-star path on MI355X.

  python bench.py --gpus N --steps K --warmup W            # IC training traces/s (BASELINE.json configs[1])
  python bench.py --workload is ...                        # IS posterior particles/s (configs[3])

A "step" of the default workload is one pass of the hot path over one minibatch of 1024 synthetic
GaussianUnknownMean traces already resident in HBM: zero_grad -> InferenceNetworkLSTM._loss -> backward ->
[flat-gradient all-reduce over RCCL when N > 1] -> Adam (pyprob/nn/inference_network.py:486-496) with LSTM
hidden 512, 1 643 583 parameters, fp32. For N > 1 the driver launches one process per GPU
(torch.distributed.run); per-GPU work is fixed (weak scaling), value = whole-job traces/s.

The JSON line also carries
  roofline     : the dominant kernel of the step - the row-panel launch (csrc/panel.hip: the whole forward + backward data
                 path of the minibatch, fp32 MFMA 4x4x1 with weights streamed from L2) - timed live with HIP events on the
                 stream it runs on inside the timed region (every stride-th launch), priced against the fp32-matrix peak
                 (157.3 TFLOP/s); `second_kernel` = the grouped weight-gradient launch, `hbm_kernels` = the memory-side
                 launches (observe embedding + LSTM input rows; the Adam pass), `whole_step` = SURVEY.md 8(d)'s algorithmic
                 FLOPs of the whole step / wall-clock step time; `traffic` = HBM bytes per launch from the committed PMC
                 passes (`traffic_source`), `rocprof_avg_us` = the committed rocprofv3 --kernel-trace --stats average
  cpu_baseline : the numpy oracle port of the same step (forward + backward + Adam) timed on this host's cores on a
                 bounded sample (rank 0, N = 1 only); `cpu_baseline_torch` = the torch-CPU restatement of the reference's
                 step (nn.LSTM / nn.Linear / autograd / optim.Adam, oracle/torch_ref.py) timed the same way
"""
import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP32_MATRIX_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBS = 8000.0


def make_engine(lstm_dim, device, seed):
    from pyprob_amd.engine import ICEngine
    from pyprob_amd.spec import NetSpec
    spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=lstm_dim, proposal_mixture_components=10)
    spec.add_address('16__forward__mu__Normal__1', 'Normal')
    return ICEngine(spec, device=device, seed=seed)


def synth_gum_dataset(n, device, seed):
    """Packed offline dataset of GaussianUnknownMean prior traces in HBM (columnar, pre-shuffled):
    mu ~ N(1, sqrt 5); y0, y1 ~ N(mu, sqrt 2)   (reference tests/test_inference.py:97-109)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    mu = 1.0 + (5.0 ** 0.5) * torch.randn(n, generator=g, device=device)
    obs = mu[:, None] + (2.0 ** 0.5) * torch.randn(n, 2, generator=g, device=device)
    prior = torch.tensor([1.0, 5.0 ** 0.5], device=device).repeat(n, 1).contiguous()
    return obs.contiguous(), mu.contiguous(), prior


def host_cores():
    """Cores this process tree may really use: the cgroup CPU quota when there is one (the GPU boxes of this project
    show 256 logical CPUs and a quota of 16, tools/cpu_scaling_probe.py), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline_train(lstm_dim, batch, budget_s=12.0):
    """Oracle port of one training step (numpy fp32: _loss forward, manual backward, Adam), timed on host cores."""
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from helpers import synthetic_gum_arrays
    from oracle import ic_oracle as O
    from pyprob_amd.spec import NetSpec
    spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=lstm_dim)
    spec.add_address('mu', 'Normal')
    rng = np.random.default_rng(0)
    P = {n: spec.init_tensor(n, rng) for n in spec.tensors}
    M = {n: np.zeros_like(v) for n, v in P.items()}
    V = {n: np.zeros_like(v) for n, v in P.items()}
    steps, t0 = 0, time.time()
    while True:
        arrays = synthetic_gum_arrays(batch, seed=steps)
        net = O.Net(P, ['obs0', 'obs1'], K=10, dtype=np.float32)
        out = O.loss_and_grads(net, arrays, ['mu'], ['Normal'])
        for n in P:
            O.adam_step(P[n], out['grads'][n].astype(np.float32), M[n], V[n], steps + 1, 1e-3)
        steps += 1
        if steps >= 3 and time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    return dict(value=steps * batch / dt, unit='traces/s', cores=host_cores(), kind='port',
                sample='%d steps of %d GUM traces, H=%d: oracle/ic_oracle.py loss_and_grads + adam_step (numpy fp32, '
                       'BLAS threads = all logical CPUs, %d usable under the cgroup quota)' % (steps, batch, lstm_dim, host_cores()))


def cpu_baseline_torch(lstm_dim, batch, budget_s=10.0):
    """The reference's own host kernels for this step (torch CPU: nn.LSTM, nn.Linear, autograd, optim.Adam), vectorised
    over the minibatch (oracle/torch_ref.py): an upper bound of what pyprob itself reaches on these cores."""
    from oracle.torch_ref import time_training_steps
    cores = host_cores()
    rate, steps, threads = time_training_steps(lstm_dim, batch, budget_s=budget_s, threads=min(cores, 64))
    return dict(value=rate, unit='traces/s', cores=threads, kind='port',
                sample='%d steps of %d GUM traces, H=%d: oracle/torch_ref.py (torch %s CPU nn.LSTM + nn.Linear + autograd + '
                       'optim.Adam, %d intra-op threads of %d host cores; no per-trace Python, i.e. an upper bound of the '
                       'reference)' % (steps, batch, lstm_dim, torch.__version__, threads, cores))


def cpu_baseline_reference(lstm_dim, batch, workload='train'):
    """The UNMODIFIED reference timed live on this host (tools/cpu_reference_bench.py) - only where a pyprob checkout is
    importable (the build container; the GPU box has none). Returns None otherwise."""
    ref = os.environ.get('PYPROB_REFERENCE')      # (only when asked for: bench.py never looks for /root/reference by itself)
    if not ref or not os.path.isdir(os.path.join(ref, 'pyprob')):
        return None
    try:
        sys.path.insert(0, os.path.join(REPO, 'tools'))
        import cpu_reference_bench as R
        pyprob = R.import_reference()
        cores = host_cores()
        torch.set_num_threads(cores)
        pyprob.seed(123)
        GUM, GUMM = R.make_models(pyprob)
        if workload == 'is':
            model, _ = R.end_to_end(pyprob, GUM, 2 * batch, lstm_dim, batch)
            post = R.posterior(pyprob, model, 2000)
            return dict(value=post['particles_per_sec'], unit='particles/s', cores=cores, kind='reference',
                        sample='%d particles, %s' % (post['particles'], post['definition']), detail=post)
        cls = GUMM if workload == 'train_gumm' else GUM
        model, e2e = R.end_to_end(pyprob, cls, 8 * batch, lstm_dim, batch)
        nn = R.nn_only(pyprob, model, batch, 3, 8)
        return dict(value=e2e['traces_per_sec'], unit='traces/s', cores=cores, kind='reference',
                    sample='%d traces end to end through pyprob %s learn_inference_network (online dataset, H=%d, batch %d): %s'
                           % (e2e['traces'], pyprob.__version__, lstm_dim, batch, e2e['definition']),
                    nn_only=nn, end_to_end=e2e)
    except Exception as exc:      # a broken checkout must not take the GPU measurement down with it
        print('cpu_baseline_reference failed: %r' % (exc,), file=sys.stderr)
        return None


def recorded_reference():
    """The reference's figures recorded in the build container (profiles/r06_cpu_reference.json, re-measured every round by
    tools/cpu_reference_bench.py): carried in the line when the reference itself cannot run on this box."""
    src = os.path.join('profiles', 'r06_cpu_reference.json')
    try:
        with open(os.path.join(REPO, src)) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    keep = {k: {a: b for a, b in d[k].items() if a != 'definition'} for k in
            ('gum_end_to_end', 'gum_nn_only', 'gum_posterior', 'gumm_end_to_end', 'gumm_nn_only', 'gumm_posterior') if k in d}
    keep.update(source=src, host_cores=d.get('host_cores'), cpu_model=d.get('cpu_model'), torch_threads=d.get('torch_threads'),
                note='unmodified pyprob v1.5.0 on the build container (no GPU box has the reference); the 10x target of '
                     'BASELINE.json is against gum_end_to_end')
    return keep


def cpu_baseline_gumm(lstm_dim, batch, budget_s=10.0):
    """Oracle port of one ragged (GaussianUnknownMeanMarsaglia) training step."""
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from helpers import synthetic_gumm_arrays
    from oracle import ic_oracle as O
    from pyprob_amd.spec import NetSpec
    spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=lstm_dim)
    arr, addresses = synthetic_gumm_arrays(batch, seed=1, max_iter=6)
    for a in addresses:
        spec.add_address(a, 'Uniform')
    rng = np.random.default_rng(0)
    P = {n: spec.init_tensor(n, rng) for n in spec.tensors}
    steps, t0 = 0, time.time()
    while True:
        net = O.Net(P, ['obs0', 'obs1'], K=10, dtype=np.float32)
        O.loss_and_grads(net, arr, addresses, ['Uniform'] * len(addresses))
        steps += 1
        if time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    return dict(value=steps * batch / dt, unit='traces/s', cores=host_cores(), kind='port',
                sample='%d steps of %d GUMM traces (ragged, %d heads), H=%d: oracle/ic_oracle.py loss_and_grads (numpy fp32, '
                       'per-sub-batch loops like the reference, no optimizer)' % (steps, batch, len(addresses), lstm_dim))


def cpu_baseline_is(n=20000):
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from oracle import ic_oracle as O
    from pyprob_amd.spec import NetSpec
    spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=512)
    spec.add_address('mu', 'Normal')
    rng = np.random.default_rng(0)
    P = {k: spec.init_tensor(k, rng) for k in spec.tensors}
    net = O.Net(P, ['obs0', 'obs1'], K=10, dtype=np.float32)
    vals = rng.normal(7, 1, n).astype(np.float32)
    prior = np.tile(np.array([[1.0, 5 ** 0.5]], np.float32), (n, 1))
    t0 = time.time()
    m = 0
    while time.time() - t0 < 10.0:
        k = min(500, n - m)
        if k <= 0:
            break
        O.is_rescore(net, [8.0, 9.0], np.ones(k, np.int32), np.zeros(k, np.int32), vals[m:m + k], prior[m:m + k], ['mu'],
                     ['Normal'])
        m += k
    dt = time.time() - t0
    return dict(value=m / dt, unit='particles/s', cores=1, kind='port',
                sample='%d particles re-scored one at a time (oracle is_rescore, per-particle batch-1 network like the '
                       'reference)' % m)


def api_models():
    """The reference's two test programs (tests/test_inference.py:97-109, 252-275) written against the drop-in API
    (pyprob_amd.sample / observe / Model); the Marsaglia rejection loop uses a tensor condition so that it runs in lock step."""
    import math
    import pyprob_amd as pyprob
    from pyprob_amd import Model
    from pyprob_amd.distributions import Normal, Uniform

    class GaussianWithUnknownMean(Model):
        def __init__(self):
            self.prior_mean, self.prior_stddev, self.likelihood_stddev = 1, math.sqrt(5), math.sqrt(2)
            super().__init__('Gaussian with unknown mean')

        def forward(self):
            mu = pyprob.sample(Normal(self.prior_mean, self.prior_stddev))
            likelihood = Normal(mu, self.likelihood_stddev)
            pyprob.observe(likelihood, name='obs0')
            pyprob.observe(likelihood, name='obs1')
            return mu

    class GaussianWithUnknownMeanMarsaglia(Model):
        def __init__(self):
            self.prior_mean, self.prior_stddev, self.likelihood_stddev = 1, math.sqrt(5), math.sqrt(2)
            super().__init__('Gaussian with unknown mean (Marsaglia)')

        def marsaglia(self, mean, stddev):
            uniform = Uniform(-1, 1)
            s = 1
            while s >= 1:
                x = pyprob.sample(uniform)
                y = pyprob.sample(uniform)
                s = x * x + y * y
            return mean + stddev * (x * torch.sqrt(-2 * torch.log(s) / s))

        def forward(self):
            mu = self.marsaglia(self.prior_mean, self.prior_stddev)
            likelihood = Normal(mu, self.likelihood_stddev)
            pyprob.observe(likelihood, name='obs0')
            pyprob.observe(likelihood, name='obs1')
            return mu
    return GaussianWithUnknownMean, GaussianWithUnknownMeanMarsaglia


def csrc_sha():
    """sha256 over the kernel sources (csrc/*, include/*): profiles/r05_*.json carry the hash of the sources they were
    measured on; a line quotes them only when it matches the tree that is running."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, 'pyprob_amd', 'csrc', '*')) + glob.glob(os.path.join(REPO, 'include', '*.h'))):
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_profile(name):
    """(document, note): profiles/<name> when its `csrc_sha` equals the running tree's, else (None, why)."""
    path = os.path.join('profiles', name)
    try:
        with open(os.path.join(REPO, path)) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, '%s not found' % path
    have = csrc_sha()
    if d.get('csrc_sha') != have:
        return None, '%s was measured on kernel sources %s, this tree is %s: not quoted (re-run tools/profile_round.sh)' % (
            path, d.get('csrc_sha'), have)
    return d, path


def api_posterior_bench(lib, device, lstm_dim, particles, calls, warm, program='gum', seed0=7, offset=0, train_traces=None,
                        prof_class=4):
    """BASELINE.json configs[3] through the drop-in API: Model.learn_inference_network (a short run: the network only has to
    exist and be sane) then `calls` x Model.posterior_results(particles, IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK,
    observe=...) in lock step, the observation cycling through three values. What a timed call runs is reported, not
    assumed: for a static single-statement program (GUM) the default is the LAUNCH PLAN (Model._traces_lockstep: after two
    recordings the call replays three C-ABI calls with the new observation and forward() is NOT run; `plan_replays` counts
    them) - the same loop with PP_IS_PLAN=0, where state.sample / state.observe / Trace.end run on the user's forward() every
    call (pyprob/model.py:47-88, 180-181; state.py:118-155, 203-219), is timed next to it (`noplan`). The Empirical statistics
    are read back every call either way. Returns (record, seconds, units)."""
    import contextlib
    import io
    from pyprob_amd.state import InferenceEngine, InferenceNetwork
    GUM, GUMM = api_models()
    model = (GUM if program == 'gum' else GUMM)()
    torch.manual_seed(123)
    with contextlib.redirect_stdout(io.StringIO()):
        model.learn_inference_network(num_traces=train_traces or (64 * 1024 if program == 'gum' else 96 * 1024),
                                      inference_network=InferenceNetwork.LSTM,
                                      observe_embeddings={'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, batch_size=1024, lstm_dim=lstm_dim,
                                      seed=1)
    IC = InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK
    base = {'obs0': 8, 'obs1': 9} if program == 'gum' else {'obs0': 4, 'obs1': 5}
    # three observations (the first is the configuration's own): a plan that baked an observation in would show here
    observes = [base, {k: v - 0.5 + 0.25 * j for j, (k, v) in enumerate(base.items())}, {k: v + 0.6 + 0.2 * j for j, (k, v) in enumerate(base.items())}]
    for i in range(max(warm, 3)):
        post = model.posterior_results(particles, IC, observe=observes[i % 3], lock_step=True, seed=i, offset=offset)
    cap = calls * (4 if prof_class == 4 else 64)

    def timed_loop():
        replays, last = 0, {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(calls):
            post = model.posterior_results(particles, IC, observe=observes[i % 3], lock_step=True, seed=seed0 + i, offset=offset)
            _ = post.effective_sample_size           # (the caller looks at the result: mean / ESS are read back every call)
            replays += bool(getattr(post, 'replayed_plan', False))
            last[i % 3] = post
        torch.cuda.synchronize()
        return time.perf_counter() - t0, replays, last

    # the event pairs of the in-stream kernel timing (two hipEventRecord per launch) cost a ~70 us GUM call ~20 us and a Marsaglia
    # call (~25 statement launches) ~0.5 ms: the wall time is taken from an UN-instrumented loop, the kernel times from a second,
    # instrumented one
    dt, replays, last = timed_loop()
    lib.pp_prof_arm(prof_class, cap)
    timed_loop()
    post = last[0]                                   # the configuration's own observation: its posterior is what is reported
    ms = np.zeros(cap, np.float32)
    fl = np.zeros(cap, np.float64)
    cnt = C.c_int32(0)
    lib.pp_prof_collect(ms.ctypes.data, cap, C.byref(cnt), fl.ctypes.data)
    lib.pp_prof_arm(prof_class, 0)
    plans = getattr(model, '_lockstep_plans', None) or {}
    first_kernel = any(p.get('first') for p in plans.values() if isinstance(p, dict))
    executes = (('launch-plan replay: two C-ABI calls per call (pp_is_first_statement, pp_is_fused), forward() NOT run'
                 if first_kernel else
                 'launch-plan replay: three C-ABI calls per call (pp_is_init, pp_is_step_net, pp_is_fused), forward() NOT run')
                if replays == calls else
                "the user's forward() in lock step (state.sample / state.observe / Trace.end per control-flow path)" if replays == 0
                else 'mixed: %d of %d timed calls were launch-plan replays' % (replays, calls))
    rec = dict(particles_per_sec=round(particles * calls / dt, 1), ms_per_call=round(dt / calls * 1e3, 4), particles_per_call=particles,
               calls=calls, program='GaussianUnknownMean' if program == 'gum' else 'GaussianUnknownMeanMarsaglia (tensor-condition loop)',
               api='Model.posterior_results(N, IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK, observe=..., lock_step=True)',
               timed_call_executes=executes, plan_replays=int(replays),
               timing='wall time of `calls` un-instrumented calls (statistics read back every call); kernel figures from a second, instrumented loop', observations_cycled=[dict(o) for o in observes],
               posterior_mean=round(float(post.mean), 4), posterior_stddev=round(float(post.stddev), 4),
               ess=round(float(post.effective_sample_size), 1), control_flow_paths=int(getattr(post, 'num_paths', 1)),
               network_params=model._inference_network._engine.spec.num_parameters())
    if replays:
        # the same loop with the plan off: forward() runs in every timed call
        old_env = os.environ.get('PP_IS_PLAN')
        os.environ['PP_IS_PLAN'] = '0'
        try:
            for i in range(3):
                model.posterior_results(particles, IC, observe=observes[i % 3], lock_step=True, seed=i, offset=offset)
            dt0, r0, _ = timed_loop()
        finally:
            if old_env is None:
                os.environ.pop('PP_IS_PLAN', None)
            else:
                os.environ['PP_IS_PLAN'] = old_env
        rec['noplan'] = dict(particles_per_sec=round(particles * calls / dt0, 1), ms_per_call=round(dt0 / calls * 1e3, 4), plan_replays=int(r0),
                             timed_call_executes="PP_IS_PLAN=0: the user's forward() in lock step in every timed call")
    if cnt.value > 0 and prof_class == 5:
        # the N-row statement kernel (csrc/is_step_fused.hip), every launch of the timed calls: MFMA bound, priced on the
        # reference algorithm's FLOPs per particle-statement (2 (I + H) 4H + 2 (H hid + hid 3K); a statement on
        # the shared first state has no per-particle recurrent product: 2 I 4H + head)
        us = float(ms[:cnt.value].sum()) * 1e3 / calls
        flops = float(fl[:cnt.value].sum()) / calls
        # what the kernel EXECUTES per particle-statement (it multiplies H + 8 of the H + 212 input columns: the rest is one shared
        # row, folded into a bias; head layers on 16- / 8-padded widths): a launch on the shared first state is recognised by
        # its work = particles per call x the shared-state figure
        H_, I_ = lstm_dim, 64 + 4 + 2 * (64 + 8)
        hid_, nout_ = int((H_ + 30) / 2), 30
        head_alg = 2.0 * (H_ * hid_ + hid_ * nout_)
        alg_ns, alg_sh = 2.0 * (I_ + H_) * 4 * H_ + head_alg, 2.0 * I_ * 4 * H_ + head_alg
        head_exe = 2.0 * H_ * 16 * ((hid_ + 15) // 16) + 2.0 * 8 * ((hid_ + 7) // 8) * 32
        exe_ns, exe_sh = 2.0 * (H_ + 8) * 4 * H_ + head_exe, 2.0 * 8 * 4 * H_ + head_exe
        executed = 0.0
        for w in fl[:cnt.value]:
            shared_launch = abs(w - particles * alg_sh) < 0.5 * alg_sh
            executed += (w / alg_sh) * exe_sh if shared_launch else (w / alg_ns) * exe_ns
        executed /= calls
        prof, note = committed_profile('r06_is_pmc_traffic.json')
        tr = (prof or {}).get('kernels', {}).get('is_step_fused', {})
        rec['statement_kernel'] = dict(
            bound='mfma', achieved=round(flops / (us * 1e-6) / 1e12, 2), peak=FP32_MATRIX_PEAK_TFLOPS, unit='TFLOP/s',
            frac=round(flops / (us * 1e-6) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4), us_per_call=round(us, 1),
            launches_per_call=round(cnt.value / calls, 2), flops_per_call=flops, executed_flops_per_call=executed,
            frac_executed=round(executed / (us * 1e-6) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4),
            frac_note='ONE meaning of `frac` in every record of this file: achieved / peak with achieved = the REFERENCE '
                      "algorithm's FLOPs (SURVEY.md 8d) over the measured time; `frac_executed` beside it = the FLOPs that run on "
                      'the MFMA pipe. Here 8d prices all 212 input columns per particle-statement while the kernel multiplies 8 of '
                      'them per particle and folds the rest into a bias row: `frac` can exceed 1 for this kernel, `frac_executed` '
                      'is the pipe utilisation',
            traffic=tr.get('traffic_bytes_per_particle_statement'), algorithmic_bytes=tr.get('algorithmic_bytes_per_particle_statement'),
            traffic_source=note,
            kernel='is_step_fused_kernel (one launch per statement after the first: [s_prev | h] [W_s | W_hh]^T + bias on '
                   'v_mfma_f32_32x32x2_f32, LSTM cell on the accumulators, both head layers from the fresh h tile, Philox draw + '
                   'log q; (h, c) read and written once, in place, through the row index list of a diverged path)',
            timing='HIP event pair around EVERY launch of the class inside the timed calls',
            wall_over_statement_kernels=round(dt / calls * 1e6 / us, 2),
            wall_note='host wall per call / time inside the N-row statement kernels only: the shared first statement, the '
                      'per-term log-weight kernels and torch index operations of the path bookkeeping are outside the event pairs')
    elif cnt.value > 0:
        per_call_us = float(ms[:cnt.value].sum()) * 1e3 / calls
        nbytes = float(fl[:cnt.value].sum()) / calls
        # The pass is VALU-bound (a Philox block, the draw, K exps and a logsumexp per particle; 8 B per particle reach memory): it
        # is priced on the VALU pipe with SQ counters from a committed PMC run of the same sources (tools/pmc_is_fused.sh), not on
        # HBM - the memory figure rides along for the record
        vprof, vnote = committed_profile('r06_is_fused_valu.json')
        rec['particle_kernels'] = dict(
            bound='valu', unit='share of the VALU pipes\' time busy (SQ_ACTIVE_INST_VALU x 4 / SIMD cycles of the launch at 2.4 GHz)', peak=1.0,
            achieved=(vprof or {}).get('valu_busy_fraction'), frac=(vprof or {}).get('valu_busy_fraction'),
            valu_issue_fraction=(vprof or {}).get('valu_issue_fraction'),
            valu_active_over_wave_cycles=(vprof or {}).get('valu_active_over_wave_cycles'),
            wait_any_over_wave_cycles=(vprof or {}).get('wait_any_over_wave_cycles'),
            counters=(vprof or {}).get('counters'), counters_source=vnote,
            frac_note=(vprof or {}).get('valu_busy_fraction_note'),
            hbm_gbs=round(nbytes / (per_call_us * 1e-6) / 1e9, 2), hbm_frac=round(nbytes / (per_call_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
            us_per_call=round(per_call_us, 3), launches_per_call=round(cnt.value / calls, 2), bytes_per_call=nbytes,
            kernel='is_fused_kernel (draw from the shared proposal + log q + prior / likelihood terms + float64 statistics '
                   'partials in one pass: 8 B per particle written) [+ per-row draw kernels of later statements]; HIP event '
                   'pairs around the launches of a second, instrumented loop',
            wall_over_kernel=round(dt / calls * 1e6 / per_call_us, 2))
        # every launch of the call, not only the pass over the particles: kernel class 6 brackets pp_is_init ... pp_is_fused (the
        # observe embedding, the two one-row network kernels, the pass with its statistics) - a few extra calls, outside `dt`
        k6 = 8
        lib.pp_prof_arm(6, k6 * 2)
        for i in range(k6):
            post = model.posterior_results(particles, IC, observe=observes[i % 3], lock_step=True, seed=seed0 + calls + i, offset=offset)
            _ = post.effective_sample_size
        torch.cuda.synchronize()
        ms6 = np.zeros(k6 * 2, np.float32)
        c6 = C.c_int32(0)
        lib.pp_prof_collect(ms6.ctypes.data, k6 * 2, C.byref(c6), None)
        lib.pp_prof_arm(6, 0)
        if c6.value > 0:
            chain_us = float(np.median(ms6[:c6.value])) * 1e3
            rec['device_chain'] = dict(us_per_call=round(chain_us, 2), wall_over_device_chain=round(dt / calls * 1e6 / chain_us, 2),
                                       launches='first_row_net_kernel (observe embedding + one-row LSTM step) -> first_row_head_kernel -> is_fused_kernel -> is_stats_combine_kernel: ONE HIP event pair around the whole chain')
    return rec, dt, particles * calls


def online_training_bench(device, lstm_dim, batch, traces, program='gum'):
    """End to end through the host API, trace generation INCLUDED: Model.learn_inference_network(num_traces) on the program itself
    (no dataset handed in) - prior traces generated in lock step (on the device for a single-path program, pp_prior_draw), layers
    created at the first minibatch, runs of minibatches inside one C call. This is the loop pyprob_amd/pyprob_host.py routes a
    real pyprob.Model's learn_inference_network through (InferenceNetwork.optimize on an OnlineDataset, pyprob/nn/dataset.py:50-62,
    inference_network.py:381-599) and the counterpart of the reference's end-to-end figure (`_total_train_traces /
    _total_train_seconds`, inference_network.py:529-534). Host: pyprob_amd.Model (pyprob itself is not on the GPU box)."""
    import contextlib
    import io
    from pyprob_amd.state import InferenceNetwork
    GUM, GUMM = api_models()
    model = (GUM if program == 'gum' else GUMM)()
    kw = dict(inference_network=InferenceNetwork.LSTM, observe_embeddings={'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, batch_size=batch,
              lstm_dim=lstm_dim, seed=1)
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        model.learn_inference_network(num_traces=128 * batch, **kw)          # layers, code objects, first chunk
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.learn_inference_network(num_traces=traces, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    net = model._inference_network
    return dict(traces_per_sec=round(traces / dt, 1), traces=traces, seconds=round(dt, 4), program=program,
                bookkept_traces_per_sec=round(net._total_train_traces / max(net._total_train_seconds, 1e-9), 1),
                final_loss=round(float(net._history_train_loss[-1]), 4), host='pyprob_amd.Model',
                api='Model.learn_inference_network(num_traces, LSTM, observe_embeddings, batch_size=%d, lstm_dim=%d): online, prior '
                    'generation + layer creation + training' % (batch, lstm_dim))


def offline_training_bench(device, lstm_dim, batch, traces):
    """BASELINE.json configs[1] as the reference states it - 1 M OFFLINE traces: Model.save_dataset(dir, N, N / 4) writes packed shards
    (pyprob_amd/dataset.py), Model.learn_inference_network(dataset_dir=dir) opens them, pre-sorted index, the reference's sampler,
    minibatches packed from memory-mapped columns, runs of steps inside pp_train_steps. The route pyprob_host.optimize_packed gives a
    real pyprob.Model (the reference: shelve + pickle + zlib decode per trace, ~1.1-1.4 k traces/s). One epoch."""
    import contextlib
    import io
    import shutil
    import tempfile
    from pyprob_amd.state import InferenceNetwork
    GUM, _ = api_models()
    model = GUM()
    root = tempfile.mkdtemp(prefix='pp_bench_ds_')
    kw = dict(inference_network=InferenceNetwork.LSTM, observe_embeddings={'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, batch_size=batch,
              lstm_dim=lstm_dim, seed=1)
    try:
        torch.manual_seed(2)
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter()
            model.save_dataset(root, traces, max(traces // 4, batch))
            t_write = time.perf_counter() - t0
            model.learn_inference_network(num_traces=64 * batch, dataset_dir=root, **kw)      # layers, code objects, page cache
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.learn_inference_network(num_traces=traces, dataset_dir=root, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(root) for f in fs)
    finally:
        shutil.rmtree(root, ignore_errors=True)
    net = model._inference_network
    return dict(traces_per_sec=round(traces / dt, 1), traces=traces, seconds=round(dt, 4), save_dataset_traces_per_sec=round(traces / t_write, 1),
                bytes_per_trace_on_disk=round(size / traces, 1), final_loss=round(float(net._history_train_loss[-1]), 4), host='pyprob_amd.Model',
                api='Model.save_dataset(dir, %d, %d) + Model.learn_inference_network(num_traces=%d, dataset_dir=dir, batch_size=%d, '
                    'lstm_dim=%d)' % (traces, max(traces // 4, batch), traces, batch, lstm_dim))


def _claim_stdout():
    """stdout must carry exactly ONE line, rank 0's JSON. Everything else this process writes to file descriptor 1 - RCCL
    prints a five-line version banner at its first communicator, libraries warn now and then - is sent to stderr: returns a
    private duplicate of the original stdout for `_emit_line`."""
    try:
        sys.stdout.flush()
        fd = os.dup(1)
        os.dup2(2, 1)
        return fd
    except OSError:
        return None


def _emit_line(fd, text):
    sys.stdout.flush()
    if fd is None:
        print(text, flush=True)
        return
    os.write(fd, (text + '\n').encode())


def _self_launch(n):
    """Re-execute this command under torch.distributed.run with one rank per GPU (127.0.0.1 rendezvous on a free port)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='train', choices=['train', 'train_gumm', 'is', 'dropin'])
    ap.add_argument('--lstm-dim', type=int, default=512)
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--dataset', type=int, default=1000000, help='offline traces resident in HBM (per job)')
    ap.add_argument('--particles', type=int, default=1000000, help='IS particles per job')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-is', action='store_true', help='skip the posterior_results sub-record of the default line')
    ap.add_argument('--graph', type=int, default=0, help='replay the step as a captured HIP graph (1) or launch eagerly (0)')
    ap.add_argument('--prewarm-s', type=float, default=0.35, help='untimed steady-state pre-warm before the --warmup steps')
    args = ap.parse_args()
    if (args.gpus > 1 or os.environ.get('PP_BENCH_SELF_LAUNCH') == '1') and 'RANK' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one process per GPU ourselves, the way the reference spawns
        # its own workers (pyprob/model.py:339-406); rank 0 of the children prints the line
        raise SystemExit(_self_launch(args.gpus))
    json_fd = _claim_stdout()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with python -m torch.distributed.run '
                         '--nproc-per-node %d, or without a launcher)' % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    # one process per GPU over RCCL; also taken with a single rank when launched through torch.distributed.run, so that
    # the collective path can be exercised on a 1-GPU box
    use_dist = world > 1 or ('RANK' in os.environ and os.environ.get('PP_BENCH_DIST', '1') != '0')
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group(backend='nccl', device_id=device)

    from pyprob_amd import lib as L
    from pyprob_amd.packed import ColumnarDataset, PackedBatch
    lib = L.load()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    eng = make_engine(args.lstm_dim, device, seed=123)
    eng.world_size = world
    eng.force_allreduce = use_dist
    dp_exchange = None
    if use_dist:
        dp_exchange = 'torch.distributed all_reduce (RCCL), pieces coalesced into one launch'
        if os.environ.get('PP_DP_NATIVE', '1') == '1':      # this library's own communicator, exchange issued from C
            from pyprob_amd.parallel import init_native_comm        # (all ranks agree; any failure -> the torch path above)
            if init_native_comm(device, lib):
                eng.native_dp = True
                dp_exchange = 'pp_dp_reduce_grads (ncclAllReduce from the C side, grouped pieces)'

        # the collective really spans N ranks: what RCCL itself reports (this library's communicator when it is up, else
        # torch.distributed's), checked against --gpus - a run that silently fell back to fewer ranks must not print a line
        rccl_ranks = int(lib.pp_dp_world()) if eng.native_dp else int(dist.get_world_size())
        if rccl_ranks != world:
            raise SystemExit('bench.py: RCCL reports %d ranks, --gpus %d' % (rccl_ranks, world))
        out_rccl = rccl_ranks
        eng.broadcast_params()
    out = {}
    if use_dist:
        out['rccl_ranks'] = out_rccl
    K, W = args.steps, args.warmup

    if args.workload == 'train':
        B = args.batch
        per_rank = max(args.dataset // world, B * 8)
        obs, mu, prior = synth_gum_dataset(per_rank, device, seed=1000 + rank)
        ds = ColumnarDataset(obs, mu, prior, B)
        cache = {}
        lr = 1e-3 * (world ** 0.5)          # inference_network.py:448
        if args.graph:
            stage = ds.staging_batch(0, 1, cache)
            ds.load_into_staging(0)
            eng.capture_train_step(stage, lr)

            def step(i):
                ds.load_into_staging(i % ds.n_batches)      # one 20 KB device copy: the next minibatch into the captured buffers
                eng.replay_train_step()
            walk = 1
        else:
            # every minibatch of the resident dataset; the steps walk them with a stride coprime to their number, so that the
            # timed region samples the WHOLE 1 M-trace dataset (each step a region of HBM it has not read recently), not its
            # first K + W minibatches
            batches = [ds.batch(i, 0, 1, cache) for i in range(ds.n_batches)]
            if use_dist:
                # dL/dW_hh is zero on a rank whose traces all have ONE controlled variable; when that holds on EVERY rank (each
                # reads it off its own resident minibatches, a MIN-all-reduce of the flag decides) the range - 2/3 of the flat
                # gradient at H = 512 - stays out of the all-reduce: bit-identical result (ICEngine.agree_skip_recurrent)
                single = all(b.t_max == 1 for b in batches) and os.environ.get('PP_DP_SKIP_WHH', '1') != '0'
                out['dp_skip_recurrent'] = eng.agree_skip_recurrent(single)
            walk = next(q for q in range(max(1, ds.n_batches // max(K, 1)), ds.n_batches + 2) if np.gcd(q, ds.n_batches) == 1)

            def pick(i):
                return batches[(i * walk) % len(batches)]

            def step(i):
                eng.train_step(pick(i), lr)
        # Two host loops over the same kernels: one Python iteration per step (two C calls) or the native loop
        # pp_train_resident (up to 256 steps per C call: ~4x less host time per step, the robust choice on a slow or busy
        # host). The untimed pre-warm measures both in the steady state and the timed region uses the faster one
        # (PP_BENCH_LOOP=python|native pins it). Data parallel: the native loop needs the C-side exchange.
        native_ok = (not args.graph) and (not use_dist or eng.native_dp)
        forced = os.environ.get('PP_BENCH_LOOP', 'auto')

        def run_steps(i0, n, native):
            if not native:
                for i in range(i0, i0 + n):
                    step(i)
                return
            for c0 in range(0, n, 256):
                m = min(256, n - c0)
                eng.train_resident([pick(i0 + c0 + j) for j in range(m)], [lr] * m)

        def wall(native, n):
            barrier()
            t = time.perf_counter()
            run_steps(0, n, native)
            barrier()
            return (time.perf_counter() - t) / n

        # ---- untimed pre-warm: code objects, clocks, host caches; then a probe of both loops (the
        # driver's 20-step region after 5 warm-up steps used to start a few launches after process start)
        # (the timed region is ~1.3 ms at the driver's K = 20: a generational collection of the interpreter inside it would be a
        # visible fraction - collected HERE, before the pre-warm, and held off until the timed region is over. Not right before the
        # timed region: a collection idles the device for tens of ms and the first steps after an idle period run at ramping
        # clocks - 0.077 instead of 0.067 ms/step over 20 steps, profiles/r05_experiments_not_kept.txt)
        gc.collect()
        gc.disable()
        t_pre = time.perf_counter()
        run_steps(0, 30, False)
        prewarm_steps = 30
        probe = {}
        n_probe = 200
        for name, nat in (('python', False), ('native', True)):
            if (nat and not native_ok) or (forced in ('python', 'native') and forced != name):
                continue
            wall(nat, 20)
            probe[name] = min(wall(nat, n_probe), wall(nat, n_probe))
            prewarm_steps += 20 + 2 * n_probe
        if use_dist:      # every rank must take the same loop: decide on the slowest rank's numbers
            tt = torch.tensor([probe.get('python', 1e9), probe.get('native', 1e9)], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            probe = {k: float(v) for k, v in zip(('python', 'native'), tt.tolist()) if v < 1e8}
        native_loop = min(probe, key=probe.get) == 'native'
        while time.perf_counter() - t_pre < args.prewarm_s:
            run_steps(0, 100, native_loop)
            prewarm_steps += 100
            if use_dist:
                break         # (ranks would disagree on the count; the probes above already ran > 0.1 s)
        run_steps(0, W, native_loop)
        # the launch timed live inside the timed region: class 0 = the row-panel kernel (csrc/panel.hip: forward + backward data
        # path of the step, the longest launch) where it runs, else class 1 = the grouped weight-gradient launch. An event
        # pair costs the stream ~3 us: every `stride`-th launch carries one, so the timed region stays within ~1 % of an
        # untimed one
        panel_expected = os.environ.get('PP_PANEL', '1') != '0' and args.lstm_dim == 512 and \
            os.environ.get('PP_DETERMINISTIC', '0') != '1'
        live_class = 0 if panel_expected else 1
        stride = max(4, K // 50)
        if not args.graph:
            lib.pp_prof_stride(stride)
            lib.pp_prof_arm(live_class, K // stride + 1)
        barrier()
        t0 = time.perf_counter()
        run_steps(W, K, native_loop)
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()

        def collect(which, n):
            ms = np.zeros(n, np.float32)
            fl = np.zeros(n, np.float64)
            cnt = C.c_int32(0)
            lib.pp_prof_collect(ms.ctypes.data, n, C.byref(cnt), fl.ctypes.data)
            lib.pp_prof_arm(which, 0)
            return (float(ms[:cnt.value].mean()), float(fl[0]), int(cnt.value)) if cnt.value > 0 else None

        def eager_pass(which, n):
            # HIP events cannot be recorded inside a captured graph, and only one kernel class is timed at a time: a
            # short eager pass of the same step right after the timed region (same kernels, same shapes).
            bl = [ds.batch(i, 0, 1, cache) for i in range(min(ds.n_batches, n))]
            lib.pp_prof_arm(which, n)
            for i in range(n):
                eng.train_step(bl[i % len(bl)], lr)
            torch.cuda.synchronize()
            return collect(which, n)

        live = eager_pass(live_class, min(K, 50)) if args.graph else collect(live_class, K)
        lib.pp_prof_stride(1)
        # per-step times (event pairs between consecutive steps of one more untimed pass): the median next to the mean
        n_med = min(max(K, 20), 200)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_med + 1)]
        evs[0].record()
        for i in range(n_med):
            step(i)
            evs[i + 1].record()
        torch.cuda.synchronize()
        per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n_med))
        out['ms_per_step_median'] = round(per_step[n_med // 2], 4)
        other = eager_pass(1 - live_class, min(K, 50))
        first, dominant = (live, other) if live_class == 0 else (other, live)      # class 0 / class 1 samples
        gather = eager_pass(2, min(K, 50))
        adam = eager_pass(3, min(K, 50))
        final_loss = float(eng.loss_buf[0].item())
        units = B * K
        metric, unit = 'ic_train_traces_per_sec', 'traces/s'
        # HBM bytes per launch (PMC passes) and rocprofv3 --kernel-trace --stats averages of this command, committed by
        # tools/profile_round.sh together with the hash of the kernel sources they were measured on: quoted only on a match
        std_shape = B == 1024 and args.lstm_dim == 512
        pmc_doc, pmc_file = committed_profile('r06_pmc_traffic.json') if std_shape else (None, 'not the profiled shape')
        pmc = (pmc_doc or {}).get('kernels', {})
        avg_doc, avg_file = committed_profile('r06_kernel_avgs.json') if std_shape else (None, 'not the profiled shape')
        rocprof_avgs = avg_doc or {}
        if std_shape and (pmc_doc is None or avg_doc is None):
            out['profile_note'] = pmc_file if pmc_doc is None else avg_file

        def roof(sample, key, label, bound='mfma', algorithmic=None):
            """achieved = ALGORITHMIC work per launch (what the reference's algorithm does in this launch) /
            the measured launch duration. `algorithmic` overrides the work the C side reported for the launch, which is what
            the kernel EXECUTES: with compact LSTM-input rows the embedding-table columns of W_ih never enter a
            GEMM, so the executed FLOPs are fewer than the reference's; both are in the line."""
            avg_ms, work, n = sample
            executed = work
            if algorithmic is not None:
                work = algorithmic
            if bound == 'mfma':
                ach, peak, u = work / (avg_ms * 1e-3) / 1e12, FP32_MATRIX_PEAK_TFLOPS, 'TFLOP/s'
            else:
                ach, peak, u = work / (avg_ms * 1e-3) / 1e9, HBM_PEAK_GBS, 'GB/s'
            d = dict(bound=bound, achieved=round(ach, 3), peak=peak, unit=u, frac=round(ach / peak, 4),
                     traffic=pmc.get(key, {}).get('traffic_bytes_per_launch'),
                     algorithmic_bytes=pmc.get(key, {}).get('algorithmic_bytes_per_launch') if bound == 'mfma' else work,
                     kernel=label, avg_launch_us=round(avg_ms * 1e3, 3), launches_timed=n)
            d['flops_per_launch' if bound == 'mfma' else 'bytes_per_launch'] = work
            if algorithmic is not None and bound == 'mfma':
                d['executed_flops_per_launch'] = executed
                d['frac_executed'] = round(executed / (avg_ms * 1e-3) / 1e12 / peak, 4)
            d['timing'] = 'HIP event pair around the launch in an untimed pass of the same step right after the timed region'
            ravg = rocprof_avgs.get(key)
            if ravg:
                d['rocprof_avg_us'] = ravg
                d['frac_rocprof'] = round(work / (ravg * 1e-6) / (1e12 if bound == 'mfma' else 1e9) / peak, 4)
            d['traffic_source'] = (pmc_file + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, csrc_sha %s)'
                                   % pmc_doc.get('csrc_sha')) if d['traffic'] is not None else pmc_file
            return d
        H, I = args.lstm_dim, eng.spec.lstm_in
        if dominant and first:
            hid_ = int((H + 30) / 2)
            e_obs = eng.spec.e_obs
            timing_live = ('HIP event pair around every %d-th launch INSIDE the timed region, on the stream the kernel runs on (an '
                           'event pair adds ~2-3 us to the interval: frac is a lower bound; rocprof_avg_us / frac_rocprof = the '
                           'committed rocprofv3 --kernel-trace --stats average of this command)' % stride)
            timing_post = 'HIP event pair around the launch in an untimed pass of the same step right after the timed region'
            # reference algorithm, weight gradients of one step: dW_ih [4H, I], dW1 [hid, H], dW2 [30, hid], the observe
            # embedding's four weight matrices (64x64 twice, 32x16 twice); K = batch rows
            wgrad_alg = 2.0 * B * (4 * H * I + hid_ * H + 30 * hid_ + 2 * 64 * 64 + 2 * 32 * 16)
            t1 = os.environ.get('PP_WGRAD_T1', '1') != '0' and os.environ.get('PP_DETERMINISTIC', '0') != '1'
            wgrad = roof(dominant, 'wgrad_group',
                         ('wgrad_t1_kernel (csrc/wgrad_t1.hip: ' if t1 else 'gemm_f32_async_grouped_aux_kernel (') +
                         'last launch of the backward pass: the weight gradients dW_ih[:, :e_obs] %dx%dx%d, dW1, dW2 and the '
                         'observe-embedding leaves as 64x64 MFMA tiles over row ranges of the minibatch%s, combined with float '
                         'atomics, and behind them the reduction jobs - column sums, table-column gradients from the per-address '
                         "sums of dG, LSTM bias gradients, loss; algorithmic FLOPs = the reference's dW_ih %dx%dx%d + leaves, "
                         'SURVEY.md 8(d))' % (4 * H, e_obs, B, ' (operands streamed as MFMA fragments straight from memory, no LDS '
                                              'staging)' if t1 else '', 4 * H, I, B),
                         algorithmic=wgrad_alg)
            panel_exec = 2.0 * B * (2.0 * 3.0 * H * e_obs + 2.0 * H * hid_ + 2.0 * hid_ * 30)
            if abs(first[1] - panel_exec) < 1.0:
                # the row-panel launch carries the whole data path of the step: forward X W_ih^T + cell + both head layers +
                # mixture log-prob, backward dy, dz1, dh, cell, dX. Algorithmic FLOPs = the reference's products of those
                # (full LSTM input width I, all four gates); executed = 64 observe-embedding columns, three gates (c_prev = 0)
                panel_alg = 2.0 * B * (2.0 * 4 * H * I + 2.0 * H * hid_ + 2.0 * hid_ * 30)
                p16 = os.environ.get('PP_PANEL', '2') not in ('0', '1') and H in (512, 1024)
                what = ('forward input product + LSTM cell + proposal head + mixture log-prob + loss and the backward data path '
                        'dy, dz1, dh, cell, dX of all %d traces' % B)
                if p16:
                    label = ('panel16_kernel (csrc/panel16.hip: one launch = %s; 16-row panels, four workgroups per panel (a '
                             'quarter of the hidden units each), v_mfma_f32_16x16x4_f32 with the weights streamed as fragment '
                             "images through a register ring; algorithmic FLOPs = the reference's X W_ih^T %dx%dx%d and dG W_ih, "
                             'head layers forward and backward)' % (what, B, 4 * H, I))
                else:
                    label = ('panel_t1_kernel (csrc/panel.hip: one launch = %s, 8-row panels, two workgroups per panel, '
                             "v_mfma_f32_4x4x1 with weights streamed k-major from L2; algorithmic FLOPs = the reference's "
                             'X W_ih^T %dx%dx%d and dG W_ih, head layers forward and backward)' % (what, B, 4 * H, I))
                out['roofline'] = roof(first, 'panel', label, algorithmic=panel_alg)
                out['roofline']['timing'] = timing_live if live_class == 0 else timing_post
                wgrad['timing'] = timing_post if live_class == 0 else timing_live
                out['roofline']['second_kernel'] = wgrad
            else:
                out['roofline'] = wgrad
                out['roofline']['timing'] = timing_live if live_class == 1 else timing_post
                out['roofline']['second_kernel'] = roof(first, 'input_gemm', 'gemm_f32_async_lstm_kernel (forward LSTM input '
                                                        'product [E | s_prev] W_ih[:, :c2]^T + per-address bias, LSTM cell in '
                                                        "the epilogue; algorithmic FLOPs = the reference's X W_ih^T %dx%dx%d)"
                                                        % (B, 4 * H, I), algorithmic=2.0 * B * I * 4 * H)
                out['roofline']['second_kernel']['timing'] = timing_post if live_class == 1 else timing_live
            hbm = []
            if gather:
                d = roof(gather, 'obs_embed_fwd', 'obs_embed_fwd_kernel (observe embedding + fused address-dispatch '
                         'gather of the LSTM input rows)', bound='hbm')
                d['bound'] = 'latency'      # %d traces x ~3 KB cannot load the memory system: priced against HBM for the record only
                hbm.append(d)
            if adam:
                # bytes the pass really moves: a tensor that never received a non-zero gradient (W_hh of a single-statement
                # program: 63 % of the parameters) has zero moments and its chunks are no-ops that only read the gradient
                # (4 B per parameter); a stepped chunk reads params, grads, both moments and writes params, both moments and
                # the cleared gradient (32 B); tensors without a gradient this step are not touched. (The C side reports
                # 32 B x all padded parameters.) 46 MB of flat buffers sit inside the 256 MB Infinity Cache: this is a
                # cache-resident rate, not an HBM one.
                seen = eng.arrived.view(-1, L.PP_ADAM_SCRATCH)[:, L.PP_ADAM_SEEN].cpu().numpy() != 0
                act = eng.presence().cpu().numpy() > 0
                ct = eng.spec.chunk_tensor_map()
                per_tensor = np.bincount(ct, minlength=eng.spec.n_tensors).astype(np.float64) * 1024
                adam_bytes = float((per_tensor * (act & seen)).sum() * 32 + (per_tensor * (act & ~seen)).sum() * 4)
                d = roof(adam, 'adam', 'adam_kernel (one pass over the stepped chunks of params, grads, both moments; clears the '
                         'gradients; chunks of never-touched tensors only read their zero gradient)', bound='hbm',
                         algorithmic=adam_bytes)
                d['note'] = 'buffers are Infinity-Cache resident (46 MB); bytes = 32 B x stepped parameters + 4 B x idle ones'
                hbm.append(d)
            out['roofline']['hbm_kernels'] = hbm
            # training FLOPs per GUM trace = 3 x (18 496 + 2 I 4H + 2 (H hid + hid 3K) + 8)
            hid = int((H + 30) / 2)
            flops_trace = 3.0 * (18496 + 2 * I * 4 * H + 2 * (H * hid + hid * 30) + 8)
            step_s = dt / K
            ws = flops_trace * B / step_s / 1e12
            out['roofline']['whole_step'] = dict(bound='mfma', achieved=round(ws, 3), peak=FP32_MATRIX_PEAK_TFLOPS, unit='TFLOP/s',
                                                 frac=round(ws / FP32_MATRIX_PEAK_TFLOPS, 4), flops_per_step=flops_trace * B,
                                                 note='algorithmic FLOPs of the whole step (SURVEY.md 8d, x3 for training) / '
                                                      'wall-clock step time of the timed region')
        config = dict(workload='GaussianUnknownMean IC training, offline traces resident in HBM, LSTM hidden=%d, '
                               'batch=%d per GPU' % (args.lstm_dim, B),
                      traces_in_hbm=per_rank * world, params=eng.spec.num_parameters(), global_batch=B * world,
                      parallelism='dp%d' % world, optimizer='Adam lr=1e-3*sqrt(world)', final_loss=round(final_loss, 4),
                      launch='hip_graph_replay' if args.graph else
                      ('native loop: up to 256 steps per C call (pp_train_resident), eager launches' if native_loop else
                       'one Python iteration per step (ICEngine.train_step), eager launches'),
                      prewarm_steps=prewarm_steps, prewarm_s=args.prewarm_s, minibatches_resident=ds.n_batches,
                      minibatch_walk='step i trains minibatch (i * %d) mod %d of the resident dataset' % (walk, ds.n_batches),
                      csrc_sha=csrc_sha(),
                      loop_probe_us_per_step={k: round(v * 1e6, 2) for k, v in probe.items()},
                      allreduce_bytes_per_step=(4 * (eng.grads_full.numel() - sum(c for _, c in eng.dp_skip)) if use_dist else 0),
                      dp_exchange=dp_exchange)
        if world == 1 and not args.no_is:
            # the other half of BASELINE.json's metric in the same line: particles/s of posterior_results through the API
            # (configs[3] on one GPU; `--workload is` is the full record incl. the control-flow program)
            out['is'] = api_posterior_bench(lib, device, args.lstm_dim, 1000000, 100, 6, 'gum')[0]      # (100 calls: ~7 ms timed)
            # ... and a program with stochastic control flow in lock step (BASELINE.json configs[2]'s model): statements after
            # the first one run per particle - the N-row statement kernel, with its own MFMA roofline
            out['gumm_lockstep'] = api_posterior_bench(lib, device, args.lstm_dim, 200000, 5, 2, 'gumm', prof_class=5)[0]
            # the same program with 10^6 particles per call: a 200 000-particle call is bound by the interpreter re-running
            # forward() once per control-flow path (~10 ms for ~9 paths, whatever the particle count); with five times the
            # particles the device is what a call waits for
            g1m = api_posterior_bench(lib, device, args.lstm_dim, 1000000, 3, 1, 'gumm', prof_class=5)[0]
            out['gumm_lockstep_1m'] = {k: g1m[k] for k in ('particles_per_sec', 'ms_per_call', 'particles_per_call', 'calls',
                                                           'control_flow_paths', 'ess', 'posterior_mean') if k in g1m}
            if 'statement_kernel' in g1m:
                out['gumm_lockstep_1m']['statement_kernel'] = {k: g1m['statement_kernel'][k] for k in
                                                               ('achieved', 'frac', 'frac_executed',
                                                                'us_per_call', 'launches_per_call', 'wall_over_statement_kernels')}
        if world == 1 and not args.no_is:
            out['online_e2e'] = online_training_bench(device, args.lstm_dim, B, 2 * 1024 * 1024)
            config['online_e2e_traces_per_sec'] = out['online_e2e']['traces_per_sec']
            try:      # (needs a writable temporary directory for the 40 MB of packed shards)
                out['offline_e2e'] = offline_training_bench(device, args.lstm_dim, B, 1000000)
                config['offline_e2e_traces_per_sec'] = out['offline_e2e']['traces_per_sec']
            except OSError as exc:
                config['offline_e2e_traces_per_sec'] = None
                out['offline_e2e'] = dict(error=repr(exc))
        # the driver's record keeps the flat scalars of `config` / `roofline` and only the NAMES of nested objects: every
        # number README.md quotes is repeated here as a flat key
        config['ms_per_step_median'] = out.get('ms_per_step_median')
        for key, flat in (('is', 'is_particles_per_sec'), ('gumm_lockstep', 'gumm_lockstep_particles_per_sec'),
                          ('gumm_lockstep_1m', 'gumm_lockstep_1m_particles_per_sec')):
            if key in out:
                config[flat] = out[key].get('particles_per_sec')
        if 'is' in out:
            config['is_ms_per_call'] = out['is'].get('ms_per_call')
            config['is_noplan_particles_per_sec'] = out['is'].get('noplan', {}).get('particles_per_sec')
            config['is_host'] = 'pyprob_amd.Model (the stand-alone mirror host; pyprob itself is not on the GPU box)'
        if 'gumm_lockstep' in out:
            sk = out['gumm_lockstep'].get('statement_kernel', {})
            config['gumm_lockstep_ms_per_call'] = out['gumm_lockstep'].get('ms_per_call')
            config['gumm_statement_frac'] = sk.get('frac')
            config['gumm_statement_frac_executed'] = sk.get('frac_executed')
            config['gumm_wall_over_statement_kernels'] = sk.get('wall_over_statement_kernels')
        rl = out.get('roofline', {})
        # MFMA utilisation by COUNTER (SQ_VALU_MFMA_BUSY_CYCLES over the launch's SIMD cycles, tools/pmc_kernel.sh; quoted on a
        # source-hash match): what `frac_executed` estimates from the FLOP count
        mdoc, mnote = committed_profile('r06_mfma_busy.json') if (args.batch == 1024 and args.lstm_dim == 512) else (None, 'not the profiled shape')
        if mdoc is not None and 'kernel' in rl:
            key = 'panel' if 'panel16' in rl['kernel'] else 'wgrad_group'
            rl['mfma_busy_frac'] = mdoc['kernels'].get(key, {}).get('mfma_busy_frac')
            rl['wgrad_mfma_busy_frac'] = mdoc['kernels'].get('wgrad_group', {}).get('mfma_busy_frac')
            rl['mfma_busy_source'] = mnote
        if 'whole_step' in rl:
            rl['whole_step_frac'] = rl['whole_step']['frac']
        if 'second_kernel' in rl:
            rl['wgrad_frac'] = rl['second_kernel'].get('frac')
            rl['wgrad_frac_executed'] = rl['second_kernel'].get('frac_executed')
            rl['wgrad_us'] = rl['second_kernel'].get('avg_launch_us')
            t2, a2 = rl['second_kernel'].get('traffic'), rl['second_kernel'].get('algorithmic_bytes')
            rl['wgrad_traffic_ratio'] = round(t2 / a2, 3) if t2 and a2 else None
        if rl.get('traffic') and rl.get('algorithmic_bytes'):
            rl['traffic_ratio'] = round(rl['traffic'] / rl['algorithmic_bytes'], 3)
        for d in rl.get('hbm_kernels', []):
            rl[d['kernel'].split(' ')[0].replace('_kernel', '') + '_us'] = d.get('avg_launch_us')
    elif args.workload == 'train_gumm':
        # BASELINE.json configs[2]: GaussianUnknownMeanMarsaglia (stochastic control flow -> variable-length traces, one
        # proposal head per address), batch 1024, hidden 512. Ragged minibatches are packed on the host and uploaded
        # BEFORE the timed region (inputs resident in HBM); a parity case, reported for completeness.
        sys.path.insert(0, os.path.join(REPO, 'tests'))
        from helpers import synthetic_gumm_arrays
        B = args.batch
        nb = 16
        arrays0, addresses = synthetic_gumm_arrays(8, seed=0, max_iter=6)
        eng.add_addresses([(a, 'Uniform', None) for a in addresses])
        batches = []
        for i in range(nb):
            arr, _ = synthetic_gumm_arrays(B, seed=100 + rank * nb + i, max_iter=6)
            ids = np.array([eng.spec.address_id[addresses[j]] for j in arr['addr_idx']])
            batches.append(PackedBatch.from_ragged(arr['trace_len'], ids, arr['values'], arr['prior'], arr['obs'],
                                                   len(eng.spec.addresses)).to(device))
        lr = 1e-3 * (world ** 0.5)
        for i in range(W):
            eng.train_step(batches[i % nb], lr)
        # the weight-gradient launch timed live: an event pair costs the stream ~5 us (two boundaries around the launch, visible
        # as gaps in profiles/s5v_ragged_step_sequence.csv) - every 4-th launch carries one, as in the default workload
        lib.pp_prof_stride(4)
        lib.pp_prof_arm(1, K // 4 + 1)
        barrier()
        t0 = time.perf_counter()
        for i in range(K):
            eng.train_step(batches[(W + i) % nb], lr)
        barrier()
        dt = time.perf_counter() - t0
        ms = np.zeros(K, np.float32)
        fl = np.zeros(K, np.float64)
        cnt = C.c_int32(0)
        lib.pp_prof_collect(ms.ctypes.data, K, C.byref(cnt), fl.ctypes.data)
        lib.pp_prof_arm(1, 0)
        lib.pp_prof_stride(1)
        units = B * K
        metric, unit = 'ic_train_traces_per_sec', 'traces/s'
        mean_len = float(np.mean([b.mean_length_controlled for b in batches]))
        config = dict(workload='GaussianUnknownMeanMarsaglia IC training (ragged traces, mean controlled length %.2f, %d '
                               'addresses/heads), LSTM hidden=%d, batch=%d per GPU' % (mean_len, len(addresses),
                                                                                       args.lstm_dim, B),
                      params=eng.spec.num_parameters(), global_batch=B * world, parallelism='dp%d' % world,
                      final_loss=round(float(eng.loss_buf[0].item()), 4), launch='eager')
        flops_step = 0.0
        for b in batches[:1]:
            n_later = b.n_rows - b.n_traces
            flops_step = 3.0 * (B * 18496 + b.n_rows * (868352 + 293764 + 8) + n_later * 2097152)
        out['roofline'] = dict(bound='mfma', achieved=round(flops_step * K / dt / 1e12, 3), peak=FP32_MATRIX_PEAK_TFLOPS,
                               unit='TFLOP/s', frac=round(flops_step * K / dt / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4), traffic=None,
                               kernel='whole step (all GEMM + elementwise kernels), algorithmic FLOPs of SURVEY.md 8(d) x3 for '
                                      'training / wall-clock step time')
        # HBM bytes of the WHOLE step (all launches) from the committed PMC passes of this command (tools/profile_round6.sh), quoted
        # on a source-hash match; algorithmic bytes per step: every parameter read in the forward and the backward
        # pass, its gradient written, Adam's read of (w, g, m, v) and write of (w, m, v) = 44 B per parameter, + the materialised
        # LSTM input rows (written, read by the weight gradients) and the per-row gate / hidden / cell rows written forward and
        # read backward
        P_ = eng.spec.num_parameters()
        rows_ = float(np.mean([b.n_rows for b in batches]))
        H_ = args.lstm_dim
        alg_bytes = 44.0 * P_ + rows_ * 4.0 * (2 * 68 + 2 * (4 * H_ + 2 * H_) + 2 * (int((H_ + 30) / 2) + 30))
        gdoc, gnote = committed_profile('r06_gumm_traffic.json') if (B == 1024 and H_ == 512) else (None, 'not the profiled shape')
        out['roofline']['traffic'] = (gdoc or {}).get('traffic_bytes_per_step')
        out['roofline']['algorithmic_bytes'] = round(alg_bytes)
        out['roofline']['traffic_source'] = gnote if gdoc is None else '%s (csrc_sha %s)' % (gnote, gdoc.get('csrc_sha'))
        if out['roofline']['traffic']:
            out['roofline']['traffic_ratio'] = round(out['roofline']['traffic'] / alg_bytes, 3)
        out['roofline']['launches_per_step'] = (gdoc or {}).get('launches_per_step')
        if cnt.value > 0:
            avg_ms, flops = float(ms[:cnt.value].mean()), float(fl[:cnt.value].mean())
            ach = flops / (avg_ms * 1e-3) / 1e12
            out['roofline']['dominant_kernel'] = dict(
                bound='mfma', achieved=round(ach, 3), peak=FP32_MATRIX_PEAK_TFLOPS, unit='TFLOP/s',
                frac=round(ach / FP32_MATRIX_PEAK_TFLOPS, 4), avg_launch_us=round(avg_ms * 1e3, 3), launches_timed=int(cnt.value),
                flops_per_launch=flops, kernel='grouped weight-gradient launch holding dW_ih / dW_hh of the backward pass '
                                               '(the head products of the %d addresses ride in the same grouped launches)' % len(addresses))
    elif args.workload == 'dropin':
        # What a pyprob user gets after binding.install(): the executors pyprob_amd/pyprob_host.py routes pyprob.Model through,
        # end to end (trace generation, layer creation, bookkeeping included) - measured here through pyprob_amd.Model, the host
        # that exists on the GPU box; under pyprob itself every sample / observe statement additionally constructs one pyprob
        # Distribution object and forward() runs in every posterior call (no launch plan: `is_noplan`).
        del eng
        B = args.batch
        rec = online_training_bench(device, args.lstm_dim, B, max(K, 1) * 64 * B, 'gum')
        dt, units = rec['seconds'], rec['traces']
        K = units // B
        metric, unit = 'ic_train_traces_per_sec', 'traces/s'
        old_env = os.environ.get('PP_IS_PLAN')
        os.environ['PP_IS_PLAN'] = '0'
        try:
            isr = api_posterior_bench(lib, device, args.lstm_dim, args.particles, 20, 3, 'gum')[0]
            gm = api_posterior_bench(lib, device, args.lstm_dim, 200000, 5, 2, 'gumm', prof_class=5)[0]
        finally:
            if old_env is None:
                os.environ.pop('PP_IS_PLAN', None)
            else:
                os.environ['PP_IS_PLAN'] = old_env
        gmm = online_training_bench(device, args.lstm_dim, B, 256 * B, 'gumm')
        off = offline_training_bench(device, args.lstm_dim, B, args.dataset)
        out['offline_e2e'] = off
        config = dict(workload='drop-in paths end to end (pyprob_host executors through pyprob_amd.Model): GaussianUnknownMean online '
                               'IC training incl. prior generation, LSTM hidden=%d, batch=%d' % (args.lstm_dim, B),
                      parallelism='dp1', online_e2e_traces_per_sec=rec['traces_per_sec'],
                      online_e2e_bookkept_traces_per_sec=rec['bookkept_traces_per_sec'], online_final_loss=rec['final_loss'],
                      gumm_online_e2e_traces_per_sec=gmm['traces_per_sec'],
                      offline_e2e_traces_per_sec=off['traces_per_sec'], offline_traces_on_disk=off['traces'],
                      save_dataset_traces_per_sec=off['save_dataset_traces_per_sec'],
                      is_noplan_particles_per_sec=isr['particles_per_sec'], is_noplan_ms_per_call=isr['ms_per_call'],
                      gumm_lockstep_particles_per_sec=gm['particles_per_sec'], api=rec['api'])
        out['roofline'] = dict(bound='host', achieved=None, peak=None, unit='traces/s', frac=None, traffic=None,
                               note='an end-to-end rate through the host API: see the default workload for the kernels\' rooflines')
    else:
        # BASELINE.json configs[3]: posterior_results through the drop-in API, particles sharded over the ranks (no collective on
        # the data path: distinct Philox counter ranges per rank)
        n = args.particles // world
        del eng
        rec, dt, units = api_posterior_bench(lib, device, args.lstm_dim, n, K, max(W, 3), 'gum', offset=rank * n)
        metric, unit = 'is_posterior_particles_per_sec', 'particles/s'
        pk = rec.pop('particle_kernels', None)
        out['roofline'] = pk if pk else dict(bound='hbm', achieved=None, peak=HBM_PEAK_GBS, unit='GB/s', frac=None, traffic=None)
        out['roofline'].setdefault('traffic', None)
        config = dict(workload='GaussianUnknownMean posterior_results IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK through '
                               'Model.posterior_results on the user program, LSTM hidden=%d, %d particles per call per GPU'
                               % (args.lstm_dim, n), parallelism='particles sharded x%d' % world, **rec)
        if world == 1:      # the N-row network step: a program with stochastic control flow in lock step
            g, _, _ = api_posterior_bench(lib, device, args.lstm_dim, 200000, max(3, K // 10), 2, 'gumm', prof_class=5)
            out['gumm_lockstep'] = g

    if use_dist and args.workload == 'train' and eng.native_dp and not args.graph:
        # the exchange of a step, by HIP event pairs (a short pass of its own: a record costs the stream ~1.5 us): the early
        # ranges' all-reduce on the side stream, the rest's on the step's stream, and what the step then still waited for the side
        # stream. exposed = rest + wait (what sits between the backward pass and Adam); PP_DP_OVERLAP=0: one collective, all of it
        # exposed
        us = (C.c_float * 3)()
        samples = []
        if eng.dp_overlap:
            lib.pp_dp_overlap_stats(1, None)
            for i in range(12):
                step(K + W + i)
                if lib.pp_dp_overlap_stats(1, us) == 1:
                    samples.append([float(v) for v in us])
            lib.pp_dp_overlap_stats(0, None)
        if samples:
            med = np.median(np.asarray(samples[2:] or samples), axis=0)
            config['allreduce_us'] = round(float(med[0] + med[1]), 2)
            config['exposed_allreduce_us'] = round(float(med[1] + med[2]), 2)
            config['early_bucket_allreduce_us'] = round(float(med[0]), 2)
            config['early_bucket_bytes'] = int(4 * sum(c for _, c in eng.dp_overlap))
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            vals = []
            for i in range(12):
                eng.loss(pick(K + W + i), backward=True)
                e0.record()
                eng.allreduce_grads()
                e1.record()
                eng.optimizer_step(lr, zero_grads=True, skip=eng.reduced_status())
                torch.cuda.synchronize()
                vals.append(e0.elapsed_time(e1) * 1e3)
            config['allreduce_us'] = config['exposed_allreduce_us'] = round(float(np.median(vals[2:])), 2)
        config['dp_overlap_ranges'] = [[int(o), int(c)] for o, c in eng.dp_overlap]
    # max over ranks
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = units * world / dt
    if rank == 0:
        line = dict(metric=metric, value=round(value, 1), unit=unit, n_gpus=world, steps=K, warmup=W,
                    ms_per_step=round(dt / K * 1e3, 4), higher_is_better=True, scaling='weak', vs_baseline=None,
                    dtype='f32', data='synthetic', config=config)
        line.update(out)
        if world == 1 and not args.no_cpu_baseline:
            # the live reference where it can be imported (kind "reference"); on the GPU box the torch port of the step
            # (the stronger of the two ports) with the reference's recorded figures next to it
            live = cpu_baseline_reference(args.lstm_dim, args.batch, args.workload)
            if live is not None:
                line['cpu_baseline'] = live
            elif args.workload in ('train', 'dropin'):
                line['cpu_baseline'] = cpu_baseline_torch(args.lstm_dim, args.batch)
                if args.workload == 'train':
                    line['cpu_baseline_numpy'] = cpu_baseline_train(args.lstm_dim, args.batch, budget_s=6.0)
            elif args.workload == 'train_gumm':
                line['cpu_baseline'] = cpu_baseline_gumm(args.lstm_dim, args.batch)
            else:
                line['cpu_baseline'] = cpu_baseline_is()
            if live is None:
                line['cpu_baseline_reference_recorded'] = recorded_reference()
        _emit_line(json_fd, json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
