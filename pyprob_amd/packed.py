"""Step-major ragged trace batch: the boundary data structure between pyprob's per-trace Python objects and the
HIP kernels (layout documented in include/pyprob_amd.h).

The reference groups a minibatch into sub-batches of identical address sequence and loops over them in Python
(pyprob/nn/dataset.py:21-37, pyprob/nn/inference_network_lstm.py:138). Here the whole minibatch becomes ONE
packed tensor set: traces sorted longest-first, row (t, b) = row_off[t] + b, rows additionally indexed by address
(the head dispatch) and by previous address (embedding-table gradients).
"""
import ctypes as C

import numpy as np

from . import lib as L


class PackedBatch:
    """Host (numpy) form; `.to(device)` uploads two buffers and fills the pp_batch struct."""

    def __init__(self, n_traces, n_rows, n_addr, obs, value, prior, addr, prev_row, trace, n_active, row_off,
                 grp_rows, grp_off, nxt_rows, nxt_off, order, src_row, mean_length):
        self.size = self.n_traces = int(n_traces)
        self.n_rows = int(n_rows)
        self.n_addr = int(n_addr)
        self.t_max = len(n_active)
        self.obs, self.value, self.prior = obs, value, prior
        self.addr, self.prev_row, self.trace = addr, prev_row, trace
        self.n_active, self.row_off = n_active, row_off
        self.grp_rows, self.grp_off, self.nxt_rows, self.nxt_off = grp_rows, grp_off, nxt_rows, nxt_off
        self.order = order          # packed trace position -> original trace index
        self.src_row = src_row      # packed row -> row in the trace-major input arrays
        self.mean_length_controlled = mean_length
        self.cur_counts = np.diff(grp_off)
        self.prev_counts = np.diff(nxt_off)
        self.dev = None
        self.c = None

    @property
    def presence_key(self):
        """Which addresses appear as the current / the previous variable of a row (hashable; decides the presence map of the
        optimizer step, NetSpec.active_mask). Computed once: a packed batch does not change, and a resident minibatch is
        trained on again and again (4 us of numpy per step otherwise)."""
        k = self.__dict__.get('_presence_key')
        if k is None:
            k = self._presence_key = (tuple((np.asarray(self.cur_counts) > 0).tolist()),
                                      tuple((np.asarray(self.prev_counts) > 0).tolist()))
        return k

    # ---------------------------------------------------------------------------------------------------
    @staticmethod
    def from_ragged(trace_len, addr_ids, values, prior, obs, n_addr):
        """trace-major ragged arrays -> step-major packed batch, by the native packer of the C ABI (pp_pack_ragged: one
        pass over the rows into ONE buffer that `.to()` uploads with one copy). Same result as `from_ragged_numpy`.

        trace_len [B] controlled length of every trace; addr_ids [R] engine address id per variable;
        values [R]; prior [R, >=2]; obs [B, obs_width]."""
        lib = L.load()
        trace_len = np.ascontiguousarray(trace_len, np.int32).reshape(-1)
        B = len(trace_len)
        if B == 0:
            raise ValueError('empty batch')
        if np.any(trace_len <= 0):
            raise ValueError('Trace of length zero.')        # pyprob/nn/dataset.py:28-29
        addr_ids = np.ascontiguousarray(addr_ids, np.int32).reshape(-1)
        values = np.ascontiguousarray(values, np.float32).reshape(-1)
        prior = np.ascontiguousarray(prior, np.float32)
        pw = prior.shape[1] if prior.ndim == 2 else 0
        obs = np.ascontiguousarray(obs, np.float32).reshape(B, -1)
        R, T = int(trace_len.sum()), int(trace_len.max())
        if len(addr_ids) != R or len(values) != R or (pw and len(prior) != R):
            raise ValueError('ragged columns do not match the trace lengths')
        words = lib.pp_pack_words(B, R, T, obs.shape[1], int(n_addr))
        buf = np.empty(words, np.float32)
        info = L.pp_pack_info()
        L.check(lib.pp_pack_ragged(trace_len.ctypes.data, addr_ids.ctypes.data, values.ctypes.data,
                                   prior.ctypes.data if pw else None, pw, obs.ctypes.data, B, obs.shape[1], int(n_addr),
                                   buf.ctypes.data, words, C.byref(info)), 'pp_pack_ragged')
        return PackedBatch._wrap_native(buf, info, obs.shape[1], n_addr)

    @staticmethod
    def _wrap_native(buf, info, W, n_addr):
        """Views into the one buffer the native packer filled (pp_pack_info word offsets)."""
        ib = buf.view(np.int32)
        B, R, T = int(info.n_traces), int(info.n_rows), int(info.t_max)

        def f(o, n):
            return buf[o:o + n]

        def i(o, n):
            return ib[o:o + n]
        nx = max(R - B, 0)
        pb = PackedBatch(B, R, n_addr, f(info.obs, B * W).reshape(B, W), f(info.value, R), f(info.prior, 2 * R).reshape(R, 2),
                         i(info.addr, R), i(info.prev_row, R), i(info.trace, R), i(info.n_active, T), i(info.row_off, T + 1),
                         i(info.grp_rows, R), i(info.grp_off, n_addr + 1), i(info.nxt_rows, nx), i(info.nxt_off, n_addr + 1),
                         i(info.order, B), i(info.src_row, R), float(R) / B)
        pb._buf, pb._dev_words, pb._info = buf, int(info.device_words), info
        return pb

    @staticmethod
    def from_ragged_numpy(trace_len, addr_ids, values, prior, obs, n_addr):
        """The numpy statement of the packing algorithm (the checker of the native packer in the tests).

        trace_len [B] controlled length of every trace; addr_ids [R] engine address id per variable;
        values [R]; prior [R, >=2]; obs [B, obs_width]."""
        trace_len = np.asarray(trace_len, np.int64)
        if trace_len.size == 0:
            raise ValueError('empty batch')
        if np.any(trace_len <= 0):
            raise ValueError('Trace of length zero.')        # pyprob/nn/dataset.py:28-29
        addr_ids = np.asarray(addr_ids, np.int64)
        values = np.asarray(values, np.float32)
        prior = np.asarray(prior, np.float32)
        obs = np.asarray(obs, np.float32)
        B = len(trace_len)
        off = np.concatenate([[0], np.cumsum(trace_len)])
        R = int(off[-1])
        # longest first; ties keep traces with the same first address adjacent (locality for the head gather)
        order = np.lexsort((np.arange(B), addr_ids[off[:-1]], -trace_len))
        lens = trace_len[order]
        T = int(lens[0])
        n_active = np.array([(lens > t).sum() for t in range(T)], np.int32)
        row_off = np.concatenate([[0], np.cumsum(n_active)]).astype(np.int32)
        src_row = np.empty(R, np.int64)
        prev_row = np.full(R, -1, np.int32)
        trace = np.empty(R, np.int32)
        for t in range(T):
            n = int(n_active[t])
            r0 = int(row_off[t])
            src_row[r0:r0 + n] = off[order[:n]] + t
            trace[r0:r0 + n] = np.arange(n)
            if t > 0:
                prev_row[r0:r0 + n] = row_off[t - 1] + np.arange(n)
        addr = addr_ids[src_row].astype(np.int32)
        grp_rows = np.argsort(addr, kind='stable').astype(np.int32)
        grp_off = np.concatenate([[0], np.cumsum(np.bincount(addr, minlength=n_addr))]).astype(np.int32)
        later = np.nonzero(prev_row >= 0)[0]
        prev_addr = addr[prev_row[later]]
        nxt_rows = later[np.argsort(prev_addr, kind='stable')].astype(np.int32)
        nxt_off = np.concatenate([[0], np.cumsum(np.bincount(prev_addr, minlength=n_addr))]).astype(np.int32)
        pr = np.zeros((R, 2), np.float32)
        w = min(2, prior.shape[1]) if prior.ndim == 2 else 0
        if w:
            pr[:, :w] = prior[src_row, :w]
        return PackedBatch(B, R, n_addr, np.ascontiguousarray(obs[order]), values[src_row], pr, addr, prev_row, trace,
                           n_active, row_off, grp_rows, grp_off, nxt_rows, nxt_off, order, src_row,
                           float(trace_len.sum()) / B)

    # ---------------------------------------------------------------------------------------------------
    def to(self, device):
        """Upload (one float buffer + one int32 buffer) and build the C struct. PyTorch is the allocator."""
        import torch
        B, R = self.n_traces, self.n_rows
        nx = len(self.nxt_rows)
        nf = B * self.obs.shape[1] + 3 * R
        if getattr(self, '_pinned', None) is not None:   # packed natively into page-locked memory: one asynchronous DMA
            buf = torch.empty(self._dev_words, dtype=torch.float32, device=device)
            buf.copy_(self._pinned[:self._dev_words], non_blocking=True)
            host = None
        elif getattr(self, '_buf', None) is not None:    # packed natively: the device part is already one buffer
            host = self._buf[:self._dev_words]
        else:
            # ONE host buffer and ONE copy: float columns first, then the int32 index columns (bit patterns in float32)
            host = np.empty(nf + 4 * R + (self.t_max + 1) + max(nx, 1), np.float32)
            host[:nf] = np.concatenate([self.obs.reshape(-1), self.value, self.prior.reshape(-1)])
            host[nf:].view(np.int32)[:] = np.concatenate([self.addr, self.prev_row, self.grp_rows, self.trace, self.row_off,
                                                          self.nxt_rows if nx else np.zeros(1, np.int32)])
        if host is not None:
            buf = torch.from_numpy(host).to(device, non_blocking=False)
        f = buf[:nf]
        i = buf[nf:].view(torch.int32)
        ow = self.obs.shape[1]
        d = {}
        d['obs'] = f[:B * ow].view(B, ow)
        d['value'] = f[B * ow:B * ow + R]
        d['prior'] = f[B * ow + R:].view(R, 2)
        o = 0
        for name, n in (('addr', R), ('prev_row', R), ('grp_rows', R), ('trace', R), ('row_off_dev', self.t_max + 1),
                        ('nxt_rows', max(nx, 1))):
            d[name] = i[o:o + n]
            o += n
        d['_f'], d['_i'], d['_buf'] = f, i, buf
        self.dev = d
        self._fill_struct(ow)
        return self

    def op_tensors(self, device):
        """(device buffer, CPU int32 descriptor) for the `pyprob_hip::ic_loss` operator (pyprob_amd/ops.py)."""
        from . import ops
        if self.dev is None:
            self.to(device)
        buf = self.dev.get('_buf')
        if buf is None:
            raise RuntimeError('op_tensors needs a batch that was uploaded as one buffer (PackedBatch.to)')
        base = buf.data_ptr()
        offsets = {k: (self.dev[k].data_ptr() - base) // 4 for k in ops._BH_COLS}
        desc = ops.batch_descriptor(self.n_traces, self.n_rows, self.t_max, self.dev['obs'].shape[1], self.n_addr, offsets,
                                    self.n_active, self.row_off, self.grp_off, self.nxt_off)
        return buf, desc

    def _fill_struct(self, obs_width):
        d = self.dev
        c = L.pp_batch()
        c.n_traces, c.n_rows, c.t_max, c.obs_width = self.n_traces, self.n_rows, self.t_max, obs_width
        self._h_n_active = np.ascontiguousarray(self.n_active, np.int32)
        self._h_row_off = np.ascontiguousarray(self.row_off, np.int32)
        self._h_grp_off = np.ascontiguousarray(self.grp_off, np.int32)
        self._h_nxt_off = np.ascontiguousarray(self.nxt_off, np.int32)
        ip = C.POINTER(C.c_int32)
        c.n_active = self._h_n_active.ctypes.data_as(ip)
        c.row_off = self._h_row_off.ctypes.data_as(ip)
        c.grp_off = self._h_grp_off.ctypes.data_as(ip)
        c.nxt_off = self._h_nxt_off.ctypes.data_as(ip)
        for k in ('obs', 'value', 'prior', 'addr', 'prev_row', 'grp_rows', 'trace', 'row_off_dev', 'nxt_rows'):
            setattr(c, k, d[k].data_ptr())
        self.c = c

    # ---------------------------------------------------------------------------------------------------
    @staticmethod
    def from_device_columns(obs, value, prior, addr_id, n_addr, index_cache):
        """Homogeneous length-1 traces already resident in HBM (a slice of a packed offline dataset): no host work,
        no copies. obs [B, w], value [B], prior [B, 2] are device tensors; the index arrays (identity) come from
        `index_cache`, a dict shared by all batches of the same size."""
        import torch
        B = value.shape[0]
        key = (B, int(addr_id), int(n_addr), str(value.device))
        if key not in index_cache:
            dev = value.device
            ar = torch.arange(B, dtype=torch.int32, device=dev)
            index_cache[key] = dict(addr=torch.full((B,), int(addr_id), dtype=torch.int32, device=dev),
                                    prev_row=torch.full((B,), -1, dtype=torch.int32, device=dev),
                                    ident=ar, row_off_dev=torch.tensor([0, B], dtype=torch.int32, device=dev),
                                    nxt=torch.zeros(1, dtype=torch.int32, device=dev))
        ix = index_cache[key]
        grp_off = np.zeros(n_addr + 1, np.int32)
        grp_off[addr_id + 1:] = B
        pb = PackedBatch(B, B, n_addr, None, None, None, None, None, None, np.array([B], np.int32),
                         np.array([0, B], np.int32), None, grp_off, np.zeros(0, np.int32), np.zeros(n_addr + 1, np.int32),
                         None, None, 1.0)
        pb.dev = dict(obs=obs, value=value, prior=prior, addr=ix['addr'], prev_row=ix['prev_row'], grp_rows=ix['ident'],
                      trace=ix['ident'], row_off_dev=ix['row_off_dev'], nxt_rows=ix['nxt'])
        pb._fill_struct(obs.shape[1])
        return pb


class ColumnarDataset:
    """Packed offline dataset of homogeneous length-1 traces resident in HBM, stored minibatch-blocked:
    block i = [obs (B x w) | value (B) | prior (B x 2)] contiguous, so a minibatch is ONE contiguous slice
    (zero-copy view for eager steps, one device copy into the captured buffers for HIP-graph replay)."""

    def __init__(self, obs, value, prior, batch_size):
        self.B, self.w = batch_size, obs.shape[1]
        self.n_batches = value.shape[0] // batch_size
        self.blocks = self.block_layout(obs, value, prior, batch_size)      # [n_batches, B*(w+3)]

    @staticmethod
    def block_layout(obs, value, prior, batch_size):
        import torch
        B, w = batch_size, obs.shape[1]
        nb = value.shape[0] // B
        n = nb * B
        return torch.cat([obs[:n].reshape(nb, B * w), value[:n].reshape(nb, B), prior[:n].reshape(nb, B * 2)], dim=1).contiguous()

    def refill(self, obs, value, prior):
        """New traces into the SAME buffers (stream-ordered copy): the PackedBatch objects of every block stay valid."""
        new = self.block_layout(obs, value, prior, self.B)
        if new.shape != self.blocks.shape:
            return False
        self.blocks.copy_(new)
        return True

    def columns(self, i, block=None):
        B, w = self.B, self.w
        row = self.blocks[i % self.n_batches] if block is None else block
        return row[:B * w].view(B, w), row[B * w:B * w + B], row[B * w + B:].view(B, 2)

    def batch(self, i, addr_id, n_addr, index_cache):
        return PackedBatch.from_device_columns(*self.columns(i), addr_id, n_addr, index_cache)

    def staging_batch(self, addr_id, n_addr, index_cache):
        """A batch whose buffers stay at fixed addresses (for graph capture); fill with `load_into`."""
        import torch
        self._stage = torch.empty_like(self.blocks[0])
        return PackedBatch.from_device_columns(*self.columns(0, self._stage), addr_id, n_addr, index_cache)

    def load_into_staging(self, i):
        self._stage.copy_(self.blocks[i % self.n_batches])


def pack_traces(traces, spec, obs_names):
    """list of pyprob-style Trace objects (duck-typed: .variables_controlled[*].{address, value, distribution},
    .named_variables[name].value) -> PackedBatch (host). Mirrors what Batch.__init__ + the torch.stack calls of
    `_loss` extract (pyprob/nn/dataset.py:21-37, inference_network_lstm.py:168,195-196,
    proposal_normal_normal_mixture.py:26-27, inference_network.py:135)."""
    trace_len, addr_ids, values, prior, obs = [], [], [], [], []
    for tr in traces:
        vc = tr.variables_controlled
        if len(vc) == 0:
            raise ValueError('Trace of length zero.')
        trace_len.append(len(vc))
        for v in vc:
            addr_ids.append(spec.address_id[v.address])
            values.append(float(v.value))
            prior.append(distribution_params(v.distribution))
        row = []
        for n in obs_names:
            val = tr.named_variables[n].value
            row.extend(np.asarray(val, np.float32).reshape(-1).tolist() if not hasattr(val, 'detach')
                       else val.detach().float().reshape(-1).tolist())
        obs.append(row)
    prior = np.asarray(prior, np.float32)
    bernoulli = [a for a, info in enumerate(spec.addresses) if info.dist_name == 'Bernoulli']
    if bernoulli:
        prior = bernoulli_group_stats(trace_len, addr_ids, values, prior, bernoulli)
    return PackedBatch.from_ragged(trace_len, addr_ids, values, prior, np.asarray(obs, np.float32), len(spec.addresses))


def bernoulli_group_stats(trace_len, addr_ids, values, prior, bernoulli_ids):
    """The `prior` pair of a row whose address has a Bernoulli proposal: (n, sum of values) over the rows of the same
    SUB-BATCH STEP (traces with the same address sequence, Batch.__init__ pyprob/nn/dataset.py:21-37, same time step) -
    the reference scores every proposal of such a step against every value of it (PP_HEAD_BERNOULLI, pyprob_amd.h)."""
    trace_len = np.asarray(trace_len, np.int64)
    addr_ids = np.asarray(addr_ids, np.int64)
    values = np.asarray(values, np.float64)
    prior = np.array(prior, np.float32, copy=True).reshape(len(addr_ids), -1)[:, :2]
    off = np.concatenate([[0], np.cumsum(trace_len)])
    is_b = np.isin(addr_ids, np.asarray(bernoulli_ids, np.int64))
    groups = {}
    for b in range(len(trace_len)):
        key = tuple(addr_ids[off[b]:off[b + 1]].tolist())
        for t in range(int(trace_len[b])):
            r = off[b] + t
            if is_b[r]:
                groups.setdefault((key, t), []).append(r)
    for rows in groups.values():
        prior[rows, 0] = len(rows)
        prior[rows, 1] = values[rows].sum()
    return prior


POISSON_LOW_HIGH = (0.0, 40.0)


def distribution_params(dist):
    """(p0, p1) of a prior as the heads need it: Normal -> (mean, stddev) (proposal_normal_normal_mixture.py:26-27),
    Uniform -> (low, high) (proposal_uniform_truncated_normal_mixture.py:28-29), Categorical -> unused."""
    name = dist.name
    if name == 'Normal':
        return (float(dist.mean), float(dist.stddev))
    if name == 'Uniform':
        return (float(dist.low), float(dist.high))
    if name == 'Categorical':
        return (0.0, 0.0)
    if name == 'Poisson':      # the Poisson head proposes on a FIXED interval (proposal_poisson_truncated_normal_mixture.py:10)
        return POISSON_LOW_HIGH
    if name == 'Bernoulli':    # placeholder: pack_traces fills in the sub-batch step statistics (bernoulli_group_stats)
        return (1.0, 0.0)
    raise RuntimeError('Distribution currently unsupported: {}'.format(name))
