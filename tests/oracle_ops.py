"""TEST INFRASTRUCTURE: oracle-backed "CPU" kernels for the `pyprob_hip::*` operators (pyprob_amd/ops.py).

The product registers the operators for the device only (HIP); there is no CPU path in pyprob_amd. To execute the HOST
logic that sits on top of the operators without a GPU - the reference-side binding (pyprob_amd/binding.py), the particle
coroutine scheduler (pyprob_amd/coroutine.py), the lock-step executor - the CPU test-suite registers these stand-ins,
each a restatement of the operator's contract with the numpy oracle (oracle/ic_oracle.py). Importing this module is what
registers them; nothing under pyprob_amd/ imports it.
"""
import math

import numpy as np
import torch

from oracle import ic_oracle as O
from pyprob_amd import lib as L
from pyprob_amd import ops as P
from pyprob_amd.engine import ICEngine
from pyprob_amd.spec import CHUNK

DT = np.float64


class CpuBufferEngine(ICEngine):
    """ICEngine's buffers (flat parameters, gradients, Adam moments, network description) on CPU tensors. The compute
    methods of ICEngine call the C ABI directly and stay unusable here; the operators are what runs."""

    def __init__(self, spec, device='cpu', seed=None):
        self.lib = L.load()
        self.spec = spec
        self.device = torch.device('cpu')
        self.rng = np.random.default_rng(seed)
        self.params = torch.zeros(0, dtype=torch.float32)
        self.workspace = None
        self.ws_bytes = 0
        self.ws_shape = (0, 0)
        self.status_buf = torch.zeros(4, dtype=torch.int32)
        self.world_size = 1
        self.force_allreduce = False
        self.optimizer = dict(kind='adam', larc=False, momentum=0.9)
        self._resize(initialise=list(spec.tensors.keys()))
        self.run_lengths = []      # steps of every train_run call (the cuts of the run planner, nn.optimize)

    def train_run(self, dataset, id_lists, lrs, weight_decay=0.0, beta1=0.9, beta2=0.999, eps=1e-8):
        """pp_train_steps (csrc/train_loop.hip) restated on the host, step by step through the operators: pack the step's
        traces -> zero_grad, loss, backward -> [one all-reduce of gradients | presence | loss | flag] -> Adam, which skips a
        flagged step. Returns the per-step (loss, status) the C loop leaves in its rings: under data parallelism the
        all-reduced SUM of the losses and the reduced flag."""
        self._adam_only('train_run')
        if len(lrs) != len(id_lists):
            raise ValueError('one learning rate per step')
        dp = self.world_size != 1 or self.force_allreduce
        losses, statuses = [], []
        for ids, lr in zip(id_lists, lrs):
            pb = dataset.device_batch(np.asarray(ids), self.spec, 'cpu')
            for a, n in enumerate(pb.cur_counts):
                if n > 0:
                    self.spec.addresses[a].total_train_iterations += 1      # inference_network_lstm.py:198
            self.train_step(pb, float(lr), weight_decay=weight_decay)
            losses.append(self.loss_buf[:1].clone())
            statuses.append((self.reduced_status() if dp else self.status_buf)[:1].clone())
        self.run_lengths.append(len(id_lists))
        return torch.cat(losses), torch.cat(statuses)

    def read_back(self, losses, statuses):
        lo, st = losses.numpy().copy(), statuses.numpy().copy()
        return lambda: (lo, st)


def _net_from_flat(params, spec, dtype=DT):
    flat = params.detach().numpy()
    d = {}
    for name, (off, shape) in spec.tensors.items():
        d[name] = flat[off:off + int(np.prod(shape))].reshape(shape)
    return O.Net(d, [o[0] for o in spec.obs], K=spec.K, dtype=dtype)


def unpack_batch(batch_dev, batch_host):
    """Step-major packed batch -> the trace-major ragged arrays the oracle reads (packed trace order)."""
    bh = batch_host.numpy()
    B, R, T, W, A = (int(v) for v in bh[:5])
    off = {k: int(bh[5 + i]) for i, k in enumerate(P._BH_COLS)}
    f = batch_dev.detach().numpy()
    iv = f.view(np.int32)
    n_active = bh[P.BH_FIXED:P.BH_FIXED + T]
    row_off = bh[P.BH_FIXED + T:P.BH_FIXED + 2 * T + 1]
    obs = f[off['obs']:off['obs'] + B * W].reshape(B, W)
    value = f[off['value']:off['value'] + R]
    prior = f[off['prior']:off['prior'] + 2 * R].reshape(R, 2)
    addr = iv[off['addr']:off['addr'] + R]
    trace_len = np.array([(n_active > b).sum() for b in range(B)], np.int32)
    rows = np.concatenate([[row_off[t] + b for t in range(trace_len[b])] for b in range(B)]).astype(np.int64)
    return dict(trace_len=trace_len, addr_idx=addr[rows].astype(np.int32), values=value[rows], prior=prior[rows], obs=obs), rows


def _ic_loss_cpu(params, grads, workspace, batch_dev, batch_host, net, flags):
    spec = P.net_spec(net)
    batch, rows = unpack_batch(batch_dev, batch_host)
    onet = _net_from_flat(params, spec)
    addresses = [a.address for a in spec.addresses]
    dist_names = [a.dist_name for a in spec.addresses]
    bwd = bool(flags & L.PP_LOSS_BACKWARD)
    fn = O.loss_and_grads_feedforward if spec.feedforward else O.loss_and_grads
    out = fn(onet, batch, addresses, dist_names, want_grads=bwd)
    loss = torch.tensor([out['loss']], dtype=torch.float32)
    status = torch.tensor([0 if math.isfinite(out['loss']) else 1], dtype=torch.int32)
    if bwd:
        g = grads.numpy()
        if flags & L.PP_LOSS_ZERO_GRADS:
            g[:spec.n_params] = 0.0
        for name, (off, shape) in spec.tensors.items():
            n = int(np.prod(shape))
            g[off:off + n] += np.asarray(out['grads'][name], np.float32).reshape(-1)
    lp = torch.empty(0, dtype=torch.float32)
    if flags & L.PP_LOSS_KEEP_LP:
        # out['lp'] is ordered (sub-batch, time step, trace of the sub-batch): map back to packed rows
        subs = out['sub_batches']
        off = np.concatenate([[0], np.cumsum(batch['trace_len'])])
        full = np.zeros(len(rows), np.float32)
        k = 0
        for sb in subs:
            for t in range(int(batch['trace_len'][sb[0]])):
                full[rows[off[np.asarray(sb)] + t]] = out['lp'][k]
                k += 1
        lp = torch.from_numpy(full)
    return loss, status, lp


def _adam_step_cpu(params, grads, exp_avg, exp_avg_sq, chunk_tensor, active, tensor_step, scratch, lr, beta1, beta2, eps,
                   weight_decay, grad_scale, flags, skip):
    skipped = skip is not None and int(skip.view(torch.int32)[0]) != 0
    ct = chunk_tensor.numpy()
    p, g, m, v = params.numpy(), grads.numpy(), exp_avg.numpy(), exp_avg_sq.numpy()
    for t in range(tensor_step.numel()):
        if not float(active[t]) > 0:
            continue
        chunks = np.nonzero(ct == t)[0]
        sl = slice(int(chunks[0]) * CHUNK, (int(chunks[-1]) + 1) * CHUNK)
        if not skipped:
            step = int(tensor_step[t]) + 1
            P64 = p[sl].astype(np.float64)
            M64, V64 = m[sl].astype(np.float64), v[sl].astype(np.float64)
            O.adam_step(P64, g[sl].astype(np.float64) * grad_scale, M64, V64, step, lr, beta1, beta2, eps, weight_decay)
            p[sl], m[sl], v[sl] = P64, M64, V64
            tensor_step[t] = step
        if flags & L.PP_ADAM_ZERO_GRADS:
            g[sl] = 0.0


def _tensor_slices(chunk_tensor, active):
    ct = chunk_tensor.numpy()
    for t in range(active.numel()):
        if not float(active[t]) > 0:
            continue
        chunks = np.nonzero(ct == t)[0]
        yield t, slice(int(chunks[0]) * CHUNK, (int(chunks[-1]) + 1) * CHUNK)


def _sgd_step_cpu(params, grads, momentum_buf, chunk_tensor, active, lr, momentum, nesterov, weight_decay, grad_scale, flags, skip):
    skipped = skip is not None and int(skip.view(torch.int32)[0]) != 0
    p, g, m = params.numpy(), grads.numpy(), momentum_buf.numpy()
    for t, sl in _tensor_slices(chunk_tensor, active):
        if not skipped:
            P64, M64 = p[sl].astype(np.float64), m[sl].astype(np.float64)
            O.sgd_step(P64, g[sl].astype(np.float64) * grad_scale, M64, lr, momentum, nesterov, weight_decay)
            p[sl], m[sl] = P64, M64
        if flags & L.PP_ADAM_ZERO_GRADS:
            g[sl] = 0.0


def _larc_scale_cpu(params, grads, chunk_tensor, active, lr, weight_decay, grad_scale, trust_coefficient, eps, epsilon, clip,
                    scratch, skip):
    if skip is not None and int(skip.view(torch.int32)[0]) != 0:
        return
    p, g = params.numpy(), grads.numpy()
    for t, sl in _tensor_slices(chunk_tensor, active):
        g[sl] = O.larc_scale(p[sl].astype(np.float64), g[sl].astype(np.float64) * grad_scale, lr, weight_decay,
                             trust_coefficient, clip, eps, epsilon)


def _is_init_cpu(params, workspace, net, obs):
    spec = P.net_spec(net)
    onet = _net_from_flat(params, spec)
    E, _ = O.embed_observe(onet, obs.numpy().astype(DT).reshape(1, -1))
    e = torch.zeros(spec.e_obs + 8, dtype=torch.float32)
    e[:spec.e_obs] = torch.from_numpy(E[0].astype(np.float32))
    return e


def _draw(dist_name, params, prior, rng, n):
    """One draw per row from the proposal `params` of head_forward (Mixture.sample mixture.py:47-63: component index,
    then that component; TruncatedNormal by inverse CDF truncated_normal.py:94-112)."""
    from scipy.special import erfinv
    if dist_name in ('Categorical',):
        p = params[0]
        return np.array([rng.choice(p.shape[1], p=p[i] / p[i].sum()) for i in range(n)], DT)
    if dist_name == 'Bernoulli':
        return (rng.uniform(size=n) < params[0].reshape(-1)).astype(DT)
    mu, sd, p = params[:3]
    k = np.array([rng.choice(p.shape[1], p=p[i] / p[i].sum()) for i in range(n)])
    m, s = mu[np.arange(n), k], sd[np.arange(n), k]
    if dist_name == 'Normal':
        return m + s * rng.standard_normal(n)
    if dist_name == 'Poisson':
        low, high = np.full(n, O.POISSON_LOW), np.full(n, O.POISSON_HIGH)
    else:
        low, high = prior[:, 0], prior[:, 1]
    a, b = O.std_normal_cdf((low - m) / s), O.std_normal_cdf((high - m) / s)
    u = rng.uniform(size=n)
    x = m + s * math.sqrt(2.0) * erfinv(2.0 * (a + u * (b - a)) - 1.0)
    return np.clip(x, low, np.nextafter(high, low))


def _is_step_cpu(params, workspace, net, addr_id, prev_addr_id, n, e_obs, prev_value, prior, h, c, state_rows, value_in, seed,
                 offset):
    spec = P.net_spec(net)
    onet = _net_from_flat(params, spec)
    info = spec.addresses[addr_id]
    a_cur, d_cur = info.address, info.dist_name
    E = e_obs.numpy()[:spec.e_obs].astype(DT)
    pr = None if prior is None else prior.numpy().astype(DT).reshape(-1, 2)
    if pr is None:
        pr = np.zeros((n, 2), DT)
    elif pr.shape[0] == 1:
        pr = np.tile(pr, (n, 1))
    else:
        pr = pr[:n]
    first = prev_addr_id < 0
    rng = np.random.default_rng([int(seed) & 0xFFFFFFFF, int(offset) & 0xFFFFFFFF, addr_id, n])
    if spec.feedforward:
        hs = np.tile(E[None], (n, 1))
    else:
        Pm = onet.P
        I = spec.lstm_in
        rows = 1 if first else n
        x = np.zeros((1, rows, I), DT)
        x[0, :, :spec.e_obs] = E
        c1 = spec.e_obs
        if not first:
            pinfo = spec.addresses[prev_addr_id]
            s, _ = O.sample_embedding(onet, pinfo.address, pinfo.dist_name, prev_value.numpy()[:n].astype(DT))
            x[0, :, c1:c1 + spec.smp_dim] = s
            x[0, :, c1 + spec.smp_dim:c1 + spec.smp_dim + spec.dtype_dim] = Pm['_layers_distribution_type_embedding.' + pinfo.dist_name]
            x[0, :, c1 + spec.smp_dim + spec.dtype_dim:c1 + spec.smp_dim + spec.dtype_dim + spec.addr_dim] = \
                Pm['_layers_address_embedding.' + pinfo.address]
        c2 = c1 + spec.smp_dim + spec.dtype_dim + spec.addr_dim
        x[0, :, c2:c2 + spec.dtype_dim] = Pm['_layers_distribution_type_embedding.' + d_cur]
        x[0, :, c2 + spec.dtype_dim:c2 + spec.dtype_dim + spec.addr_dim] = Pm['_layers_address_embedding.' + a_cur]
        H = spec.lstm_dim
        D = spec.lstm_depth
        hv, cv = h.reshape(D, -1, H), c.reshape(D, -1, H)     # [depth, n, H] (views of the caller's state)
        xin = x
        for k in range(D):
            if first:
                h0 = c0 = None
            elif state_rows == 1:      # the shared first-statement state of row 0 (include/pyprob_amd.h, pp_is_step)
                h0 = np.tile(hv[k, :1].numpy().astype(DT), (n, 1))
                c0 = np.tile(cv[k, :1].numpy().astype(DT), (n, 1))
            else:
                h0, c0 = hv[k, :n].numpy().astype(DT), cv[k, :n].numpy().astype(DT)
            out, _, (hn, cn) = O.lstm_forward(xin, *onet.lstm_layer(k), h0, c0)
            hv[k, :rows] = torch.from_numpy(hn.astype(np.float32))
            cv[k, :rows] = torch.from_numpy(cn.astype(np.float32))
            xin = out
        hs = np.tile(out[0], (n, 1)) if first else out[0]
    dummy = np.zeros(n, DT)
    if value_in is None:
        _, _, params_q = O.head_forward(onet, a_cur, d_cur, hs, pr, dummy)
        v = _draw(d_cur, params_q, pr, rng, n)
    else:
        v = value_in.numpy()[:n].astype(DT)
    v = v.astype(np.float32).astype(DT)        # the device hands out fp32 values; score exactly those
    lp, _, params_q = O.head_forward(onet, a_cur, d_cur, hs, pr, v)
    if d_cur == 'Bernoulli':     # (head_bernoulli restates the TRAINING loss's [n, n] broadcast; one particle scores its own value)
        lp = O.bernoulli_log_prob(v, params_q[0].reshape(-1))
    return torch.from_numpy(v.astype(np.float32)), torch.from_numpy(np.asarray(lp, np.float32))


def _term(kind, p0, s0, p1, s1, x, n):
    xv = x.numpy().astype(DT).reshape(-1)
    xv = np.full(n, xv[0]) if xv.size == 1 else xv[:n]

    def col(t, s):
        a = t.numpy().astype(DT).reshape(-1)
        return np.full(n, a[0]) if s == 0 else a[:n]
    if kind == 2:
        return xv
    if kind == 5:
        C = s1
        probs = p0.numpy().astype(DT).reshape(-1, C)
        probs = np.tile(probs[:1], (n, 1)) if s0 == 0 else probs[:n]
        return O.categorical_log_prob(xv, probs)
    a = col(p0, s0)
    if kind == 3:
        return O.poisson_log_prob(xv, a)
    if kind == 4:
        return O.bernoulli_log_prob(xv, a)
    b = col(p1, s1)
    if kind == 0:
        return O.normal_log_prob(xv, a, b)
    if kind == 1:
        return O.uniform_log_prob(xv, a, b)
    raise RuntimeError('kind %d' % kind)


def _log_prob_cpu(kind, p0, p0_stride, p1, p1_stride, x, n):
    return torch.from_numpy(np.asarray(_term(kind, p0, p0_stride, p1, p1_stride, x, n), np.float32))


def _logweight_terms_cpu(lw, kinds, p0, p0_strides, p1, p1_strides, x, scales, overwrite):
    n = lw.numel()
    acc = np.zeros(n, np.float32) if overwrite else lw.numpy().copy()
    for q in range(len(kinds)):
        t = np.asarray(_term(kinds[q], p0[q], p0_strides[q], p1[q], p1_strides[q], x[q], n), np.float32)
        acc = (acc + np.float32(scales[q]) * t).astype(np.float32)     # fp32 accumulator like the kernel
    lw.copy_(torch.from_numpy(acc))


def _is_stats_cpu(lw, x, scratch):
    l = lw.numpy().astype(DT)
    ok = np.isfinite(l)
    out = torch.zeros(8, dtype=torch.float64)
    if ok.any():
        m = l[ok].max()
        w = np.exp(l[ok] - m)
        xv = np.zeros(ok.sum()) if x is None else x.numpy().astype(DT)[ok]
        out[:6] = torch.tensor([m, w.sum(), (w * w).sum(), (w * xv).sum(), (w * xv * xv).sum(), float(ok.sum())])
    else:
        out[0] = -math.inf
    return out


def _prior_draw_cpu(kind, p0, p1, n, seed, offset, stream_id):
    rng = np.random.default_rng([int(seed) & 0xFFFFFFFF, int(offset) & 0xFFFFFFFF, int(stream_id) & 0xFFFFFFFF, n])
    a = np.broadcast_to(p0.numpy().astype(DT).reshape(-1), (n,))
    b = np.broadcast_to(p1.numpy().astype(DT).reshape(-1), (n,))
    v = a + b * rng.standard_normal(n) if kind == 0 else a + (b - a) * rng.random(n)
    return torch.from_numpy(np.asarray(v, np.float32))


_NET_STASH = {}      # workspace address -> arguments of the last is_step_net (the device keeps the head outputs there)


def _is_step_net_cpu(params, workspace, net, addr_id, prev_addr_id, n, e_obs, prev_value, h, c, state_rows):
    _NET_STASH[workspace.data_ptr()] = (params, net, addr_id, prev_addr_id, n, e_obs, prev_value, h, c, state_rows)


def _is_fused_cpu(workspace, net, addr_id, prior, kinds, p0, p0_strides, p1, p1_strides, x, scales, flags, value, lw, overwrite,
                  seed, offset, stats_scratch):
    n = value.numel()
    acc = np.zeros(n, np.float32) if overwrite else lw.numpy().copy()
    if addr_id >= 0:
        params, net_, a, prev_a, n_, e_obs, prev_value, h, c, state_rows = _NET_STASH.pop(workspace.data_ptr())
        assert a == addr_id and n_ == n and net_ == net
        v, logq = _is_step_cpu(params, workspace, net, addr_id, prev_a, n, e_obs, prev_value, prior.reshape(1, 2), h, c,
                               state_rows, None, seed, offset)
        value.copy_(v)
        acc = (acc - logq.numpy()).astype(np.float32)
    for q in range(len(kinds)):
        f = int(flags[q])
        a0 = value if f & 1 else p0[q]
        a1 = value if f & 2 else p1[q]
        xx = value if f & 4 else x[q]
        s0 = 1 if f & 1 else p0_strides[q]
        s1 = 1 if f & 2 else p1_strides[q]
        t = np.asarray(_term(kinds[q], a0, s0, a1, s1, xx, n), np.float32)
        acc = (acc + np.float32(scales[q]) * t).astype(np.float32)
    lw.copy_(torch.from_numpy(acc))
    if stats_scratch is None:
        return torch.zeros(8, dtype=torch.float64)
    return _is_stats_cpu(lw, value, stats_scratch)


_registered = False


def register():
    global _registered
    if _registered:
        return
    for name, fn in (('ic_loss', _ic_loss_cpu), ('adam_step', _adam_step_cpu), ('sgd_step', _sgd_step_cpu),
                     ('larc_scale', _larc_scale_cpu), ('is_init', _is_init_cpu),
                     ('is_step', _is_step_cpu), ('is_step_net', _is_step_net_cpu), ('prior_draw', _prior_draw_cpu), ('is_fused', _is_fused_cpu),
                     ('log_prob', _log_prob_cpu), ('logweight_terms', _logweight_terms_cpu),
                     ('is_stats', _is_stats_cpu)):
        P._lib.impl(name, fn, 'CPU')
    _registered = True


register()
