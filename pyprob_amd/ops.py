"""PyTorch-ROCm custom operators over the C ABI (SURVEY.md §8b seam B3): `torch.ops.pyprob_hip.*`.

Each operator is a thin wrapper over ONE `extern "C"` entry point of libpyprob_amd.so (include/pyprob_amd.h): it checks
shapes / dtypes / contiguity / device, passes the tensors' device pointers and torch's current HIP stream, and turns a
non-zero return code into a RuntimeError. All buffers are PyTorch-owned tensors (caching allocator); the kernels never
allocate. Only the device ("CUDA" dispatch key = HIP on ROCm) implementation exists in the product: calling an operator
with CPU tensors fails in the dispatcher - there is no CPU fallback. (The CPU test-suite registers oracle-backed "CPU"
kernels for these operators from tests/, to execute the host logic above them without a GPU.)

    ic_loss      InferenceNetworkLSTM._loss (+ backward)            pyprob/nn/inference_network_lstm.py:136-220
    adam_step    optim.Adam.step over the flat buffer                pyprob/nn/inference_network.py:348,496
    sgd_step     optim.SGD(momentum, nesterov).step                  pyprob/nn/inference_network.py:350
    larc_scale   the LARC wrapper's gradient rescaling               pyprob/nn/optimizer_larc.py:72-107
    is_init      InferenceNetwork._infer_init                        pyprob/nn/inference_network.py:141-148
    is_step      _infer_step + proposal.sample() + log_prob          pyprob/nn/inference_network_lstm.py:82-134,
    is_step_rows   (the same for the rows of a diverged path, state in place)
                                                                     pyprob/state.py:207-212
    log_prob     prior / likelihood log_prob terms of the log-weight pyprob/state.py:211, 147-149

Non-tensor state travels as follows: the network description (`pp_net`, host struct with offsets into the flat
parameter buffer) is registered once per layer set with `register_net` and referenced by an integer handle; the
host-side arrays of a packed minibatch (`pp_batch.n_active / row_off / grp_off / nxt_off`) travel as one small CPU int32
tensor next to the device buffer (`PackedBatch.op_tensors`).
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L

NAMESPACE = 'pyprob_hip'

_lib = torch.library.Library(NAMESPACE, 'DEF')
_lib.define('ic_loss(Tensor params, Tensor(a!) grads, Tensor(b!) workspace, Tensor batch_dev, Tensor batch_host, int net, '
            'int flags) -> (Tensor, Tensor, Tensor)')
_lib.define('adam_step(Tensor(a!) params, Tensor(b!) grads, Tensor(c!) exp_avg, Tensor(d!) exp_avg_sq, Tensor chunk_tensor, '
            'Tensor active, Tensor(e!) tensor_step, Tensor(f!) scratch, float lr, float beta1, float beta2, float eps, '
            'float weight_decay, float grad_scale, int flags, Tensor? skip) -> ()')
_lib.define('sgd_step(Tensor(a!) params, Tensor(b!) grads, Tensor(c!) momentum_buf, Tensor chunk_tensor, Tensor active, float lr, '
            'float momentum, bool nesterov, float weight_decay, float grad_scale, int flags, Tensor? skip) -> ()')
_lib.define('larc_scale(Tensor params, Tensor(a!) grads, Tensor chunk_tensor, Tensor active, float lr, float weight_decay, '
            'float grad_scale, float trust_coefficient, float eps, float epsilon, bool clip, Tensor(b!) scratch, Tensor? skip) -> ()')
_lib.define('is_init(Tensor params, Tensor(a!) workspace, int net, Tensor obs) -> Tensor')
_lib.define('is_step(Tensor params, Tensor(a!) workspace, int net, int addr_id, int prev_addr_id, int n, Tensor e_obs, '
            'Tensor? prev_value, Tensor? prior, Tensor(b!) h, Tensor(c!) c, int state_rows, Tensor? value_in, int seed, '
            'int offset) -> (Tensor, Tensor)')
_lib.define('is_step_rows(Tensor params, Tensor(a!) workspace, int net, int addr_id, int prev_addr_id, int n, Tensor e_obs, '
            'Tensor prev_value, Tensor? prior, Tensor(b!) h, Tensor(c!) c, int state_rows, Tensor rows, Tensor? value_in, int seed, '
            'int offset) -> (Tensor, Tensor)')
_lib.define('is_statement_rows(Tensor params, Tensor(a!) workspace, int net, int addr_id, int prev_addr_id, int n, Tensor e_obs, '
            'Tensor prev_value_full, Tensor prior, Tensor(b!) h, Tensor(c!) c, int state_rows, Tensor? rows, Tensor(d!) value_full, '
            'Tensor(e!) lw_full, int prior_kind, int seed, int offset) -> ()')
_lib.define('is_step_net(Tensor params, Tensor(a!) workspace, int net, int addr_id, int prev_addr_id, int n, Tensor e_obs, '
            'Tensor? prev_value, Tensor(b!) h, Tensor(c!) c, int state_rows) -> ()')
_lib.define('is_fused(Tensor(a!) workspace, int net, int addr_id, Tensor? prior, int[] kinds, Tensor?[] p0, int[] p0_strides, '
            'Tensor?[] p1, int[] p1_strides, Tensor?[] x, float[] scales, int[] flags, Tensor(b!) value, Tensor(c!) lw, '
            'bool overwrite, int seed, int offset, Tensor(d!)? stats_scratch) -> Tensor')
_lib.define('prior_draw(int kind, Tensor p0, Tensor p1, int n, int seed, int offset, int stream_id) -> Tensor')
_lib.define('log_prob(int kind, Tensor p0, int p0_stride, Tensor? p1, int p1_stride, Tensor x, int n) -> Tensor')
_lib.define('logweight_terms(Tensor(a!) lw, int[] kinds, Tensor?[] p0, int[] p0_strides, Tensor?[] p1, int[] p1_strides, '
            'Tensor[] x, float[] scales, bool overwrite) -> ()')
_lib.define('is_stats(Tensor lw, Tensor? x, Tensor(a!) scratch) -> Tensor')

# ---- network registry -------------------------------------------------------------------------------------------
_NETS = {}
_next_handle = [1]


def register_net(net_struct, spec):
    """Make a pp_net (ctypes struct built by NetSpec.c_struct) addressable from operator calls. Returns the handle."""
    h = _next_handle[0]
    _next_handle[0] += 1
    _NETS[h] = (net_struct, spec)
    return h


def unregister_net(handle):
    _NETS.pop(handle, None)


def net_struct(handle):
    return _NETS[handle][0]


def net_spec(handle):
    return _NETS[handle][1]


# ---- packed batch <-> (device buffer, host int32 descriptor) --------------------------------------------------------
BH_FIXED = 16
_BH_COLS = ('obs', 'value', 'prior', 'addr', 'prev_row', 'grp_rows', 'trace', 'row_off_dev', 'nxt_rows')


def batch_descriptor(n_traces, n_rows, t_max, obs_width, n_addr, offsets, n_active, row_off, grp_off, nxt_off):
    """CPU int32 tensor: [B, R, T, obs_width, n_addr, 9 word offsets into the device buffer (pp_batch device columns in
    _BH_COLS order), pad to 16 | n_active [T] | row_off [T+1] | grp_off [n_addr+1] | nxt_off [n_addr+1]]."""
    head = np.zeros(BH_FIXED, np.int32)
    head[:5] = (n_traces, n_rows, t_max, obs_width, n_addr)
    head[5:5 + len(_BH_COLS)] = [offsets[k] for k in _BH_COLS]
    arr = np.concatenate([head, np.asarray(n_active, np.int32), np.asarray(row_off, np.int32),
                          np.asarray(grp_off, np.int32), np.asarray(nxt_off, np.int32)])
    return torch.from_numpy(arr)


def _batch_struct(batch_dev, batch_host):
    bh = batch_host.numpy()
    B, R, T, W, A = (int(v) for v in bh[:5])
    if bh.size != BH_FIXED + T + (T + 1) + 2 * (A + 1):
        raise RuntimeError('pyprob_hip: malformed batch descriptor')
    c = L.pp_batch()
    c.n_traces, c.n_rows, c.t_max, c.obs_width = B, R, T, W
    base = batch_dev.data_ptr()
    for i, k in enumerate(_BH_COLS):
        setattr(c, k, base + 4 * int(bh[5 + i]))
    ip = C.POINTER(C.c_int32)
    o = BH_FIXED
    host = np.ascontiguousarray(bh[o:], np.int32)          # kept alive by the caller until the C call returns
    p = host.ctypes.data
    c.n_active = C.cast(p, ip)
    c.row_off = C.cast(p + 4 * T, ip)
    c.grp_off = C.cast(p + 4 * (2 * T + 1), ip)
    c.nxt_off = C.cast(p + 4 * (2 * T + 1 + A + 1), ip)
    return c, host, (B, R, T, W, A)


def _f32(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError('pyprob_hip: %s must be a contiguous float32 tensor' % name)
    return t


def _same_device(ref, *tensors):
    for t in tensors:
        if t is not None and t.device != ref.device:
            raise RuntimeError('pyprob_hip: tensors on different devices (%s, %s)' % (ref.device, t.device))


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


# ---- device implementations ("CUDA" dispatch key = HIP) ---------------------------------------------------------------
def _ic_loss_hip(params, grads, workspace, batch_dev, batch_host, net, flags):
    lib = L.load()
    netc = net_struct(net)
    _f32(params, 'params')
    _same_device(params, grads, workspace, batch_dev)
    if params.numel() < netc.n_params:
        raise RuntimeError('pyprob_hip::ic_loss: parameter buffer smaller than the network (%d < %d)'
                           % (params.numel(), netc.n_params))
    bwd = bool(flags & L.PP_LOSS_BACKWARD)
    if bwd and (_f32(grads, 'grads').numel() < netc.n_params):
        raise RuntimeError('pyprob_hip::ic_loss: gradient buffer smaller than the network')
    c, keep, (B, R, T, W, A) = _batch_struct(batch_dev, batch_host)
    need = lib.pp_ic_workspace_bytes(C.byref(netc), B, R)
    ws_bytes = workspace.numel() * workspace.element_size()
    if need == 0 or ws_bytes < need:
        raise RuntimeError('pyprob_hip::ic_loss: workspace too small (%d < %d bytes)' % (ws_bytes, need))
    loss = torch.empty(1, dtype=torch.float32, device=params.device)
    status = torch.zeros(1, dtype=torch.int32, device=params.device)
    lp = torch.empty(R if (flags & L.PP_LOSS_KEEP_LP) else 0, dtype=torch.float32, device=params.device)
    with torch.cuda.device(params.device):
        rc = lib.pp_ic_loss(C.byref(netc), C.byref(c), params.data_ptr(), grads.data_ptr() if bwd else None,
                            workspace.data_ptr(), ws_bytes, loss.data_ptr(), status.data_ptr(),
                            lp.data_ptr() if lp.numel() else None, int(flags), _stream(params))
    L.check(rc, 'pp_ic_loss')
    del keep
    return loss, status, lp


def _adam_step_hip(params, grads, exp_avg, exp_avg_sq, chunk_tensor, active, tensor_step, scratch, lr, beta1, beta2, eps,
                   weight_decay, grad_scale, flags, skip):
    lib = L.load()
    n = params.numel()
    for t, name in ((params, 'params'), (grads, 'grads'), (exp_avg, 'exp_avg'), (exp_avg_sq, 'exp_avg_sq'), (active, 'active')):
        _f32(t, name)
    _same_device(params, grads, exp_avg, exp_avg_sq, chunk_tensor, active, tensor_step, scratch, skip)
    if min(grads.numel(), exp_avg.numel(), exp_avg_sq.numel()) < n or chunk_tensor.numel() * 1024 != n:
        raise RuntimeError('pyprob_hip::adam_step: buffer sizes do not match the parameter buffer')
    n_tensors = tensor_step.numel()
    if active.numel() < n_tensors or scratch.numel() < L.PP_ADAM_SCRATCH * n_tensors:
        raise RuntimeError('pyprob_hip::adam_step: per-tensor arrays too small')
    with torch.cuda.device(params.device):
        rc = lib.pp_adam_step(params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), n,
                              chunk_tensor.data_ptr(), active.data_ptr(), tensor_step.data_ptr(), scratch.data_ptr(),
                              n_tensors, lr, beta1, beta2, eps, weight_decay, grad_scale, int(flags), L.ptr(skip),
                              _stream(params))
    L.check(rc, 'pp_adam_step')


def _sgd_step_hip(params, grads, momentum_buf, chunk_tensor, active, lr, momentum, nesterov, weight_decay, grad_scale, flags, skip):
    lib = L.load()
    n = params.numel()
    for t, name in ((params, 'params'), (grads, 'grads'), (momentum_buf, 'momentum_buf'), (active, 'active')):
        _f32(t, name)
    _same_device(params, grads, momentum_buf, chunk_tensor, active, skip)
    if min(grads.numel(), momentum_buf.numel()) < n or chunk_tensor.numel() * 1024 != n:
        raise RuntimeError('pyprob_hip::sgd_step: buffer sizes do not match the parameter buffer')
    with torch.cuda.device(params.device):
        rc = lib.pp_sgd_step(params.data_ptr(), grads.data_ptr(), momentum_buf.data_ptr(), n, chunk_tensor.data_ptr(),
                             active.data_ptr(), active.numel(), lr, momentum, int(bool(nesterov)), weight_decay, grad_scale,
                             int(flags), L.ptr(skip), _stream(params))
    L.check(rc, 'pp_sgd_step')


def _larc_scale_hip(params, grads, chunk_tensor, active, lr, weight_decay, grad_scale, trust_coefficient, eps, epsilon, clip,
                    scratch, skip):
    lib = L.load()
    n = params.numel()
    for t, name in ((params, 'params'), (grads, 'grads'), (active, 'active'), (scratch, 'scratch')):
        _f32(t, name)
    _same_device(params, grads, chunk_tensor, active, scratch, skip)
    if grads.numel() < n or chunk_tensor.numel() * 1024 != n:
        raise RuntimeError('pyprob_hip::larc_scale: buffer sizes do not match the parameter buffer')
    if scratch.numel() < L.larc_scratch_floats(n, active.numel()):
        raise RuntimeError('pyprob_hip::larc_scale: scratch too small')
    with torch.cuda.device(params.device):
        rc = lib.pp_larc_scale(params.data_ptr(), grads.data_ptr(), n, chunk_tensor.data_ptr(), active.data_ptr(),
                               active.numel(), lr, weight_decay, grad_scale, trust_coefficient, eps, epsilon, int(bool(clip)),
                               scratch.data_ptr(), L.ptr(skip), _stream(params))
    L.check(rc, 'pp_larc_scale')


def _is_ws(lib, netc, workspace, n):
    need = lib.pp_is_workspace_bytes(C.byref(netc), n)
    have = workspace.numel() * workspace.element_size()
    if have < need:
        raise RuntimeError('pyprob_hip: importance-sampling workspace too small (%d < %d bytes)' % (have, need))
    return have


def _is_init_hip(params, workspace, net, obs):
    lib = L.load()
    netc = net_struct(net)
    _same_device(params, workspace, obs)
    ws_bytes = _is_ws(lib, netc, workspace, 1)
    if _f32(obs, 'obs').numel() != sum(netc.obs_in[o] for o in range(netc.n_obs)):
        raise RuntimeError('pyprob_hip::is_init: observe has %d values, the network expects %d'
                           % (obs.numel(), sum(netc.obs_in[o] for o in range(netc.n_obs))))
    e = torch.zeros(netc.e_obs + 8, dtype=torch.float32, device=params.device)
    with torch.cuda.device(params.device):
        rc = lib.pp_is_init(C.byref(netc), params.data_ptr(), obs.data_ptr(), e.data_ptr(), workspace.data_ptr(), ws_bytes,
                            _stream(params))
    L.check(rc, 'pp_is_init')
    return e


def _is_step_hip(params, workspace, net, addr_id, prev_addr_id, n, e_obs, prev_value, prior, h, c, state_rows, value_in, seed,
                 offset, rows=None):
    lib = L.load()
    netc = net_struct(net)
    _same_device(params, workspace, e_obs, prev_value, prior, h, c, value_in)
    ws_bytes = _is_ws(lib, netc, workspace, n)
    if not (0 <= addr_id < netc.n_addr) or prev_addr_id >= netc.n_addr:
        raise RuntimeError('pyprob_hip::is_step: address id out of range')
    H = netc.lstm_dim
    depth = max(1, int(netc.lstm_depth))
    if rows is not None:
        if rows.dtype != torch.int64 or not rows.is_contiguous() or rows.numel() < n or rows.device != params.device:
            raise RuntimeError('pyprob_hip::is_step_rows: rows must be a contiguous int64 device tensor with n entries')
        if not lib.pp_is_step_fused_supported(C.byref(netc), int(addr_id), 1 << 30) or prev_addr_id < 0:
            raise RuntimeError('pyprob_hip::is_step_rows: no fused statement kernel for this network / statement')
    elif H > 0 and (h.numel() < depth * n * H or c.numel() < depth * n * H):
        raise RuntimeError('pyprob_hip::is_step: LSTM state smaller than [depth, n, H]')
    stride = 0
    if prior is not None:
        _f32(prior, 'prior')
        stride = 0 if prior.numel() == 2 else 1
        if stride and prior.numel() < 2 * n:
            raise RuntimeError('pyprob_hip::is_step: prior must be [1, 2] or [n, 2]')
    for t, name in ((prev_value, 'prev_value'), (value_in, 'value_in')):
        if t is not None and _f32(t, name).numel() < n:
            raise RuntimeError('pyprob_hip::is_step: %s shorter than n' % name)
    value = torch.empty(n, dtype=torch.float32, device=params.device)
    logq = torch.empty(n, dtype=torch.float32, device=params.device)
    with torch.cuda.device(params.device):
        if rows is None:
            rc = lib.pp_is_step(C.byref(netc), params.data_ptr(), int(addr_id), int(prev_addr_id), int(n), e_obs.data_ptr(),
                                L.ptr(prev_value), L.ptr(prior), stride, L.ptr(h), L.ptr(c), int(state_rows), L.ptr(value_in),
                                value.data_ptr(), logq.data_ptr(), int(seed), int(offset), workspace.data_ptr(), ws_bytes,
                                _stream(params))
        else:
            rc = lib.pp_is_step_rows(C.byref(netc), params.data_ptr(), int(addr_id), int(prev_addr_id), int(n), e_obs.data_ptr(),
                                     L.ptr(prev_value), L.ptr(prior), stride, L.ptr(h), L.ptr(c), int(state_rows),
                                     rows.data_ptr(), L.ptr(value_in), value.data_ptr(), logq.data_ptr(), int(seed), int(offset),
                                     workspace.data_ptr(), ws_bytes, _stream(params))
    L.check(rc, 'pp_is_step')
    return value, logq


def _is_step_rows_hip(params, workspace, net, addr_id, prev_addr_id, n, e_obs, prev_value, prior, h, c, state_rows, rows, value_in,
                      seed, offset):
    return _is_step_hip(params, workspace, net, addr_id, prev_addr_id, n, e_obs, prev_value, prior, h, c, state_rows, value_in,
                        seed, offset, rows=rows)


def _is_statement_rows_hip(params, workspace, net, addr_id, prev_addr_id, n, e_obs, prev_value_full, prior, h, c, state_rows, rows,
                           value_full, lw_full, prior_kind, seed, offset):
    """pp_is_statement_rows: the whole statement (previous values by row, draw, value scatter, log-weight update) in one launch."""
    lib = L.load()
    netc = net_struct(net)
    _same_device(params, workspace, e_obs, prev_value_full, prior, h, c, rows, value_full, lw_full)
    ws_bytes = _is_ws(lib, netc, workspace, n)
    if not (0 <= addr_id < netc.n_addr) or not (0 <= prev_addr_id < netc.n_addr):
        raise RuntimeError('pyprob_hip::is_statement_rows: address id out of range')
    if rows is not None and (rows.dtype != torch.int64 or not rows.is_contiguous() or rows.numel() < n):
        raise RuntimeError('pyprob_hip::is_statement_rows: rows must be a contiguous int64 tensor with n entries')
    for t, name in ((prev_value_full, 'prev_value_full'), (value_full, 'value_full'), (lw_full, 'lw_full')):
        if _f32(t, name).numel() < n:
            raise RuntimeError('pyprob_hip::is_statement_rows: %s shorter than n' % name)
    stride = 0 if _f32(prior, 'prior').numel() == 2 else 1
    if stride and prior.numel() < 2 * n:
        raise RuntimeError('pyprob_hip::is_statement_rows: prior must be [1, 2] or [n, 2]')
    with torch.cuda.device(params.device):
        rc = lib.pp_is_statement_rows(C.byref(netc), params.data_ptr(), int(addr_id), int(prev_addr_id), int(n), e_obs.data_ptr(),
                                      prev_value_full.data_ptr(), prior.data_ptr(), stride, h.data_ptr(), c.data_ptr(), int(state_rows),
                                      L.ptr(rows), value_full.data_ptr(), lw_full.data_ptr(), int(prior_kind), int(seed), int(offset),
                                      workspace.data_ptr(), ws_bytes, _stream(params))
    L.check(rc, 'pp_is_statement_rows')


def _is_step_net_hip(params, workspace, net, addr_id, prev_addr_id, n, e_obs, prev_value, h, c, state_rows):
    lib = L.load()
    netc = net_struct(net)
    _same_device(params, workspace, e_obs, prev_value, h, c)
    ws_bytes = _is_ws(lib, netc, workspace, n)
    if not (0 <= addr_id < netc.n_addr) or prev_addr_id >= netc.n_addr:
        raise RuntimeError('pyprob_hip::is_step_net: address id out of range')
    with torch.cuda.device(params.device):
        rc = lib.pp_is_step_net(C.byref(netc), params.data_ptr(), int(addr_id), int(prev_addr_id), int(n), e_obs.data_ptr(),
                                L.ptr(prev_value), L.ptr(h), L.ptr(c), int(state_rows), workspace.data_ptr(), ws_bytes,
                                _stream(params))
    L.check(rc, 'pp_is_step_net')


def _is_fused_hip(workspace, net, addr_id, prior, kinds, p0, p0_strides, p1, p1_strides, x, scales, flags, value, lw, overwrite,
                  seed, offset, stats_scratch):
    lib = L.load()
    netc = net_struct(net)
    count = len(kinds)
    n = value.numel()
    _same_device(value, lw, workspace, prior, stats_scratch)
    if _f32(lw, 'lw').numel() != n or not value.is_contiguous() or not lw.is_contiguous():
        raise RuntimeError('pyprob_hip::is_fused: value and lw must be contiguous float32 vectors of one length')
    arr = (L.pp_lw_term * max(count, 1))()
    fl = (C.c_int32 * max(count, 1))()
    for q in range(count):
        _same_device(value, p0[q], p1[q], x[q])
        for t, what in ((p0[q], 'p0'), (p1[q], 'p1'), (x[q], 'x')):
            # a term tensor is one shared value or one value per particle (Categorical: one probability row or n rows); a
            # k-element tensor with k != n (a vector-valued observation) would be read at the particle index
            if t is not None and int(kinds[q]) != 5 and t.numel() not in (1, n):
                raise RuntimeError('pyprob_hip::is_fused: term %d: %s has %d elements (1 or n = %d)' % (q, what, t.numel(), n))
        arr[q].kind = int(kinds[q])
        arr[q].p0, arr[q].p1, arr[q].x = L.ptr(p0[q]), L.ptr(p1[q]), L.ptr(x[q])
        arr[q].p0_stride, arr[q].p1_stride = int(p0_strides[q]), int(p1_strides[q])
        arr[q].x_stride = 0 if (x[q] is None or x[q].numel() == 1) else 1
        arr[q].scale = float(scales[q])
        fl[q] = int(flags[q])
    out = torch.empty(8, dtype=torch.float64, device=value.device)      # (the combine kernel writes the six statistics)
    if stats_scratch is not None and (stats_scratch.dtype != torch.float64 or stats_scratch.numel() < L.PP_IS_STATS_SCRATCH):
        raise RuntimeError('pyprob_hip::is_fused: scratch must hold PP_IS_STATS_SCRATCH doubles')
    with torch.cuda.device(value.device):
        rc = lib.pp_is_fused(C.byref(netc), int(addr_id), n, L.ptr(prior), arr, fl, count, value.data_ptr(), lw.data_ptr(),
                             1 if overwrite else 0, int(seed), int(offset), out.data_ptr() if stats_scratch is not None else None,
                             L.ptr(stats_scratch), workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                             _stream(value))
    L.check(rc, 'pp_is_fused')
    return out


def _prior_draw_hip(kind, p0, p1, n, seed, offset, stream_id):
    lib = L.load()
    _same_device(p0, p1)
    out = torch.empty(n, dtype=torch.float32, device=p0.device)
    for t, name in ((p0, 'p0'), (p1, 'p1')):
        if _f32(t, name).numel() not in (1, n):
            raise RuntimeError('pyprob_hip::prior_draw: %s must have 1 or n elements' % name)
    with torch.cuda.device(p0.device):
        rc = lib.pp_prior_draw(int(kind), p0.data_ptr(), 0 if p0.numel() == 1 else 1, p1.data_ptr(), 0 if p1.numel() == 1 else 1,
                               int(n), int(seed), int(offset), int(stream_id) & 0xFFFFFFFF, out.data_ptr(), _stream(p0))
    L.check(rc, 'pp_prior_draw')
    return out


def _log_prob_hip(kind, p0, p0_stride, p1, p1_stride, x, n):
    lib = L.load()
    _same_device(x, p0, p1)
    lp = torch.empty(n, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.pp_logweight_accumulate(int(kind), _f32(p0, 'p0').data_ptr(), int(p0_stride), L.ptr(p1), int(p1_stride),
                                         _f32(x, 'x').data_ptr(), 0 if x.numel() == 1 else 1, 1.0, None, lp.data_ptr(), int(n),
                                         _stream(x))
    L.check(rc, 'pp_logweight_accumulate')
    return lp


def _logweight_terms_hip(lw, kinds, p0, p0_strides, p1, p1_strides, x, scales, overwrite):
    lib = L.load()
    count = len(kinds)
    arr = (L.pp_lw_term * count)()
    for q in range(count):
        _same_device(lw, p0[q], p1[q], x[q])
        arr[q].kind = int(kinds[q])
        arr[q].p0, arr[q].p1, arr[q].x = L.ptr(p0[q]), L.ptr(p1[q]), x[q].data_ptr()
        arr[q].p0_stride, arr[q].p1_stride = int(p0_strides[q]), int(p1_strides[q])
        arr[q].x_stride = 0 if x[q].numel() == 1 else 1
        arr[q].scale = float(scales[q])
    with torch.cuda.device(lw.device):
        rc = lib.pp_logweight_terms(arr, count, _f32(lw, 'lw').data_ptr(), lw.numel(), 1 if overwrite else 0, _stream(lw))
    L.check(rc, 'pp_logweight_terms')


def _is_stats_hip(lw, x, scratch):
    lib = L.load()
    _same_device(lw, x, scratch)
    if scratch.dtype != torch.float64 or scratch.numel() < L.PP_IS_STATS_SCRATCH:
        raise RuntimeError('pyprob_hip::is_stats: scratch must hold PP_IS_STATS_SCRATCH doubles')
    out = torch.zeros(8, dtype=torch.float64, device=lw.device)
    with torch.cuda.device(lw.device):
        rc = lib.pp_is_stats(_f32(lw, 'lw').data_ptr(), L.ptr(x), lw.numel(), out.data_ptr(), scratch.data_ptr(), _stream(lw))
    L.check(rc, 'pp_is_stats')
    return out


_lib.impl('ic_loss', _ic_loss_hip, 'CUDA')
_lib.impl('adam_step', _adam_step_hip, 'CUDA')
_lib.impl('sgd_step', _sgd_step_hip, 'CUDA')
_lib.impl('larc_scale', _larc_scale_hip, 'CUDA')
_lib.impl('is_init', _is_init_hip, 'CUDA')
_lib.impl('is_step', _is_step_hip, 'CUDA')
_lib.impl('is_step_rows', _is_step_rows_hip, 'CUDA')
_lib.impl('is_statement_rows', _is_statement_rows_hip, 'CUDA')
_lib.impl('is_step_net', _is_step_net_hip, 'CUDA')
_lib.impl('is_fused', _is_fused_hip, 'CUDA')
_lib.impl('prior_draw', _prior_draw_hip, 'CUDA')
_lib.impl('log_prob', _log_prob_hip, 'CUDA')
_lib.impl('logweight_terms', _logweight_terms_hip, 'CUDA')
_lib.impl('is_stats', _is_stats_hip, 'CUDA')

ops = getattr(torch.ops, NAMESPACE)
