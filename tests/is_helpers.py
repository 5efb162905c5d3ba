"""Shared helpers of the importance-sampling executor tests (CPU: oracle-backed operators; GPU: the HIP engine): build a
host-mirror network from a golden case, re-score recorded traces with the oracle."""
import math

import numpy as np

from conftest import load_golden
from helpers import spec_from_golden
from oracle import ic_oracle as O
from pyprob_amd.is_engine import ISRunner
from pyprob_amd.nn import InferenceNetworkFeedForward, InferenceNetworkLSTM


def _engine(spec, device):
    if str(device) == 'cpu':
        import oracle_ops
        return oracle_ops.CpuBufferEngine(spec)
    from pyprob_amd.engine import ICEngine
    return ICEngine(spec, device=device, seed=0)


def network_from_golden(case, device='cpu'):
    meta, params, batch, loss, isr = load_golden(case)
    spec = spec_from_golden(meta, params)
    cls = InferenceNetworkFeedForward if spec.feedforward else InferenceNetworkLSTM
    depths = meta.get('observe_embedding_depths', {})
    net = cls(observe_embeddings={n: {'dim': meta['observe_embedding_dims'][n], 'depth': depths.get(n, 2)} for n in meta['obs_names']},
              lstm_dim=meta['lstm_dim'] or 512, lstm_depth=meta.get('lstm_depth', 1), device=device)
    net._obs_names = list(meta['obs_names'])
    net._engine = _engine(spec, device)
    net._engine.load_state_dict(params)
    net._is = ISRunner(net._engine)
    net._layers_initialized = True
    return net, meta, params, isr


def rescore(case, meta, params, traces, observe, sigma):
    """log-weights of the given traces by the oracle: per-trace batch-1 re-scoring + Normal likelihoods."""
    spec_addresses, dist_names, trace_len, addr_idx, values, prior = [], [], [], [], [], []
    for tr in traces:
        trace_len.append(len(tr.variables_controlled))
        for v in tr.variables_controlled:
            if v.address not in spec_addresses:
                spec_addresses.append(v.address)
                dist_names.append(v.distribution.name)
            addr_idx.append(spec_addresses.index(v.address))
            values.append(float(v.value))
            d = v.distribution
            if d.name == 'Normal':
                prior.append([float(d.mean), float(d.stddev), 0.0])
            elif d.name == 'Uniform':
                prior.append([float(d.low), float(d.high), 0.0])
            elif d.name == 'Poisson':
                prior.append([float(d.rate), 0.0, 0.0])
            elif d.name == 'Bernoulli':
                prior.append([float(d.probs), 0.0, 0.0])
            else:
                prior.append([float(p) for p in d.probs.reshape(-1)])
    net = O.Net(params, meta['obs_names'], K=meta['mixture_components'])
    fn = O.is_rescore_feedforward if meta.get('network', 'lstm') == 'feedforward' else O.is_rescore
    obs = np.array([float(observe[n]) for n in meta['obs_names']])
    _, _, _, lw = fn(net, obs, np.asarray(trace_len), np.asarray(addr_idx), np.asarray(values, np.float64),
                     np.asarray(prior, np.float64), spec_addresses, dist_names)
    for b, tr in enumerate(traces):
        mu = float(tr.result)
        lw[b] += sum(float(O.normal_log_prob(y, mu, sigma)) for y in obs)
    return lw


def lockstep_network(device='cpu'):
    """The golden GUMM network with its addresses renamed to the call sites of the lock-step variant of the program
    (`while s >= 1:` compiles to different instruction offsets than `while float(s) >= 1:`)."""
    from models import GaussianWithUnknownMeanMarsagliaLockStep
    from pyprob_amd.state import TraceMode
    meta, params, batch, loss, isr = load_golden('gumm')
    model = GaussianWithUnknownMeanMarsagliaLockStep()
    tr = next(model._trace_generator(trace_mode=TraceMode.PRIOR))
    new_x, new_y = (v.address.split('__')[0] for v in tr.variables[:2])
    old_x, old_y = (a.split('__')[0] for a in meta['addresses'][:2])

    def rename(s):
        return s.replace(old_x + '__forward__marsaglia__x', new_x + '__forward__marsaglia__x').replace(
            old_y + '__forward__marsaglia__y', new_y + '__forward__marsaglia__y')
    params = {rename(k): v for k, v in params.items()}
    meta = dict(meta, addresses=[rename(a) for a in meta['addresses']])
    spec = spec_from_golden(meta, params)
    net = InferenceNetworkLSTM(observe_embeddings={n: {'dim': 32} for n in meta['obs_names']}, lstm_dim=meta['lstm_dim'], device=device)
    net._obs_names = list(meta['obs_names'])
    net._engine = _engine(spec, device)
    net._engine.load_state_dict(params)
    net._is = ISRunner(net._engine)
    net._layers_initialized = True
    model._inference_network = net
    return model, net, meta, params


def rescore_lockstep_run(post, net, meta, params, observe, sigma):
    """Rebuild every particle's trace from the statement log of a lock-step GUMM run (pairs x_k, y_k until
    x^2 + y^2 < 1) and re-score it with the oracle's batch-1 restatement. Returns (log-weights, results)."""
    log = post.statement_log
    n = post._all_log_weights.numel()
    trace_len, addr_idx, values, results = [], [], [], []
    addresses = [a.address for a in net._engine.spec.addresses]
    for i in range(n):
        k = 0
        while True:
            (ax, (vx, _)), = log[2 * k].items()
            (ay, (vy, _)), = log[2 * k + 1].items()
            x, y = float(vx[i]), float(vy[i])
            if ax in addresses:      # (iterations the network never saw in training are proposed from the prior:
                addr_idx += [addresses.index(ax), addresses.index(ay)]      # log p - log q = 0, no LSTM step)
                values += [x, y]
                known = k + 1
            k += 1
            s = np.float32(x) * np.float32(x) + np.float32(y) * np.float32(y)
            if s < 1:
                break
        trace_len.append(2 * known)
        results.append(1.0 + math.sqrt(5.0) * x * math.sqrt(-2 * math.log(s) / s))
    onet = O.Net(params, meta['obs_names'], K=meta['mixture_components'])
    prior = np.tile(np.array([[-1.0, 1.0, 0.0]]), (len(values), 1))
    obs = np.array([float(observe[k]) for k in meta['obs_names']])
    _, _, _, lw = O.is_rescore(onet, obs, np.asarray(trace_len), np.asarray(addr_idx), np.asarray(values, np.float64), prior,
                               addresses, ['Uniform'] * len(addresses))
    lw = lw + np.array([sum(float(O.normal_log_prob(y, mu, sigma)) for y in obs) for mu in results])
    return lw, np.asarray(results)
