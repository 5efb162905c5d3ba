"""Packed on-disk trace dataset and loader.

The reference stores offline traces as pickled, zlib-compressed `Trace` objects in shelve/sqlite files
(pyprob/nn/dataset.py:121-172 `OfflineDatasetFile`, pyprob/util.py:347-355), prunes every trace on write
(`_prune_trace`, dataset.py:64-119), sorts them on open by (controlled length, Python `hash()` of the address string)
into yet another shelf (dataset.py:217-259) and decodes one trace at a time when training (~1.1-1.4 k traces/s). A
minibatch only needs, per trace, the observed values and, per controlled sample statement, (address, value, prior
parameters) - so this module stores exactly those columns:

    <dir>/meta.json        version, counts, observable names/widths, address table, trace-type table
    <dir>/trace_len.npy    int32  [N]     controlled length of every trace
    <dir>/trace_type.npy   int32  [N]     index into the trace-type table (one type = one address sequence)
    <dir>/row_off.npy      int64  [N+1]   first row of every trace
    <dir>/obs.npy          float32[N, W]  observed values, observables concatenated in `obs_names` order
    <dir>/value.npy        float32[R]     sampled values (R = sum of lengths)
    <dir>/prior.npy        float32[R, 2]  prior parameters as the proposal heads need them (Normal: mean, stddev;
                                          Uniform: low, high; Categorical: unused)
    <dir>/addr.npy         int32  [R]     index into the address table

Traces are written SORTED by (length, type hash) - the order the reference's sampler wants (dataset.py:328-343) - with
a process-independent FNV-1a hash instead of Python's salted `hash()` (dataset.py:237). Every column is a plain .npy
file, opened with `np.load(mmap_mode='r')`: a minibatch is a handful of fancy-index reads and goes straight into
`PackedBatch.from_ragged`, no per-trace Python objects. `PackedTraceDataset` concatenates any number of such
directories (shards), exposes the reference's `OfflineDataset` surface (`len`, `dataset[i]` as a pruned Trace) and
yields device batches through a prefetching loader; `DistributedTraceBatchSampler` (parallel.py) partitions the
sorted index space across ranks exactly like the reference.
"""
import json
import os
import threading

import numpy as np

from .packed import PackedBatch, distribution_params

FORMAT_VERSION = 1
_PINNED_RING = dict(slots=[], next=0)   # staging buffers of PackedTraceDataset.device_batch
_COLUMNS = ('trace_len', 'trace_type', 'row_off', 'obs', 'value', 'prior', 'addr')


def fnv1a64(data):
    """64-bit FNV-1a of a bytes object (stable across processes and machines)."""
    h = 0xcbf29ce484222325
    for b in data:
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def trace_type_hash(addresses):
    """Hash of a controlled address sequence: what dataset.py:237 computes with the salted built-in hash()."""
    return fnv1a64('\0'.join(addresses).encode('utf-8'))


class PackedTraceWriter:
    """Accumulates traces (objects or columns) and writes one sorted shard directory on close()."""

    def __init__(self, path, obs_names, obs_widths=None):
        self.path = path
        self.obs_names = list(obs_names)
        self.obs_widths = None if obs_widths is None else [int(w) for w in obs_widths]
        self._addr_table = []        # [(address, distribution name, n_categories)]
        self._addr_id = {}
        self._types = []             # [(hash, tuple(address ids))]
        self._type_id = {}
        self._len, self._type, self._obs, self._value, self._prior, self._addr = [], [], [], [], [], []

    # ---- address / type tables --------------------------------------------------------------------------
    def _address(self, address, dist_name, n_categories=None):
        i = self._addr_id.get(address)
        if i is None:
            i = self._addr_id[address] = len(self._addr_table)
            self._addr_table.append((address, dist_name, None if n_categories is None else int(n_categories)))
        return i

    def _trace_type(self, ids):
        key = tuple(int(i) for i in ids)
        t = self._type_id.get(key)
        if t is None:
            t = self._type_id[key] = len(self._types)
            self._types.append((trace_type_hash([self._addr_table[i][0] for i in key]), key))
        return t

    # ---- input ------------------------------------------------------------------------------------------
    def add_trace(self, trace):
        """One pyprob-style Trace: keeps what `_prune_trace` keeps for training (dataset.py:64-119)."""
        vc = trace.variables_controlled
        if len(vc) == 0:
            raise ValueError('Trace of length zero.')
        ids = []
        for v in vc:
            d = v.distribution
            ids.append(self._address(v.address, d.name, getattr(d, 'num_categories', None) if d.name == 'Categorical' else None))
            self._value.append(np.float32(float(v.value)))
            self._prior.append(distribution_params(d))
        self._addr.append(np.asarray(ids, np.int32))
        row = []
        for n in self.obs_names:
            val = trace.named_variables[n].value
            val = val.detach().cpu().numpy() if hasattr(val, 'detach') else np.asarray(val)
            row.append(val.astype(np.float32).reshape(-1))
        if self.obs_widths is None:
            self.obs_widths = [len(r) for r in row]
        self._obs.append(np.concatenate(row) if row else np.zeros(0, np.float32))
        self._len.append(len(vc))
        self._type.append(self._trace_type(ids))

    def add_columns(self, trace_len, addresses, address_ids, values, prior, obs, trace_types=None):
        """Vectorised input (trace generators, converters): `addresses` = [(address, distribution name, n_categories)]
        is the table `address_ids` [R] indexes; trace_len [B], values [R], prior [R, 2], obs [B, W]. `trace_types` =
        (type index per trace [B], [address-id sequence of every type]) when the producer already knows which traces
        share an address sequence (the lock-step generator: one type per control-flow path); otherwise the types are
        found with a row-wise unique per trace length (an argsort over all traces)."""
        trace_len = np.asarray(trace_len, np.int64)
        if np.any(trace_len <= 0):
            raise ValueError('Trace of length zero.')
        remap = np.asarray([self._address(*a) for a in addresses], np.int32)
        ids = remap[np.asarray(address_ids, np.int64)]
        off = np.concatenate([[0], np.cumsum(trace_len)])
        obs = np.asarray(obs, np.float32).reshape(len(trace_len), -1)
        if self.obs_widths is None:
            self.obs_widths = [obs.shape[1]] if len(self.obs_names) == 1 else [obs.shape[1] // max(len(self.obs_names), 1)] * len(self.obs_names)
        self._value.append(np.asarray(values, np.float32).reshape(-1))
        self._prior.append(np.asarray(prior, np.float32).reshape(-1, 2))
        self._addr.append(ids)
        self._obs.append(obs)
        self._len.append(trace_len.astype(np.int32))
        if trace_types is not None:
            type_of, seqs = trace_types
            tids = np.asarray([self._trace_type(remap[np.asarray(q, np.int64)]) for q in seqs], np.int32)
            self._type.append(tids[np.asarray(type_of, np.int64)])
            return
        # trace types: unique address sequences, found per distinct length with a row-wise unique
        types = np.empty(len(trace_len), np.int32)
        for L in np.unique(trace_len):
            sel = np.nonzero(trace_len == L)[0]
            seqs = ids[(off[sel][:, None] + np.arange(L)[None, :])]
            uniq, inv = np.unique(seqs, axis=0, return_inverse=True)
            tids = np.asarray([self._trace_type(u) for u in uniq], np.int32)
            types[sel] = tids[inv.reshape(-1)]
        self._type.append(types)

    # ---- output -----------------------------------------------------------------------------------------
    def close(self):
        def cat(parts, dtype, shape_tail=()):
            arrs = [np.asarray(p, dtype).reshape((-1,) + shape_tail) for p in parts]
            return np.concatenate(arrs) if arrs else np.zeros((0,) + shape_tail, dtype)
        trace_len = cat(self._len, np.int32)
        N = len(trace_len)
        if N == 0:
            raise ValueError('empty dataset')
        W = int(sum(self.obs_widths or [0]))
        trace_type = cat(self._type, np.int32)
        obs = cat(self._obs, np.float32, (W,))
        value = cat(self._value, np.float32)
        prior = cat(self._prior, np.float32, (2,))
        addr = cat(self._addr, np.int32)
        off = np.concatenate([[0], np.cumsum(trace_len.astype(np.int64))])
        # sort by (length, type hash): the order OfflineDataset builds on open (dataset.py:217-259)
        hashes = np.asarray([h for h, _ in self._types], np.uint64)
        order = np.lexsort((np.arange(N), hashes[trace_type], trace_len))
        lens = trace_len[order].astype(np.int64)
        new_off = np.concatenate([[0], np.cumsum(lens)])
        rows = np.repeat(off[order] - new_off[:-1], lens) + np.arange(int(new_off[-1]))
        os.makedirs(self.path, exist_ok=True)
        cols = dict(trace_len=trace_len[order], trace_type=trace_type[order], row_off=new_off.astype(np.int64),
                    obs=obs[order], value=value[rows], prior=prior[rows], addr=addr[rows])
        for name in _COLUMNS:
            np.save(os.path.join(self.path, name + '.npy'), np.ascontiguousarray(cols[name]))
        meta = dict(format='pyprob_amd packed traces', version=FORMAT_VERSION, n_traces=int(N), n_rows=int(new_off[-1]),
                    obs_names=self.obs_names, obs_widths=[int(w) for w in (self.obs_widths or [])],
                    addresses=[dict(address=a, distribution=d, n_categories=c) for a, d, c in self._addr_table],
                    trace_types=[dict(hash='%016x' % h, address_ids=list(k)) for h, k in self._types],
                    sorted_by=['trace_len', 'trace_type_hash'])
        with open(os.path.join(self.path, 'meta.json'), 'w') as f:
            json.dump(meta, f)
        return meta

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()


class _Shard:
    def __init__(self, path):
        with open(os.path.join(path, 'meta.json')) as f:
            self.meta = json.load(f)
        if self.meta.get('version') != FORMAT_VERSION:
            raise RuntimeError('unsupported packed trace format in {}: {}'.format(path, self.meta.get('version')))
        for name in _COLUMNS:
            setattr(self, name, np.load(os.path.join(path, name + '.npy'), mmap_mode='r'))
        self.n = int(self.meta['n_traces'])


class _MemoryShard:
    """The columns of one shard held in memory (vectorised online generation): same attributes as _Shard."""

    def __init__(self, obs_names, obs_widths, trace_len, address_table, address_ids, values, prior, obs, trace_types=None):
        w = PackedTraceWriter(None, obs_names, obs_widths)
        w.add_columns(trace_len, address_table, address_ids, values, prior, obs, trace_types)
        self.meta = dict(version=FORMAT_VERSION, n_traces=int(len(trace_len)), obs_names=list(obs_names),
                         obs_widths=[int(x) for x in w.obs_widths],
                         addresses=[dict(address=a, distribution=d, n_categories=c) for a, d, c in w._addr_table],
                         trace_types=[dict(hash='%016x' % h, address_ids=list(k)) for h, k in w._types])
        self.trace_len = np.asarray(trace_len, np.int32)
        self.trace_type = np.concatenate(w._type).astype(np.int32)
        self.row_off = np.concatenate([[0], np.cumsum(self.trace_len.astype(np.int64))])
        self.obs = np.asarray(obs, np.float32).reshape(len(trace_len), -1)
        self.value = np.asarray(values, np.float32).reshape(-1)
        self.prior = np.asarray(prior, np.float32).reshape(-1, 2)
        self.addr = np.concatenate(w._addr).astype(np.int32)
        self.n = int(len(trace_len))


class PackedTraceDataset:
    """All shards under `dataset_dir` (or an explicit list of shard directories) as one indexable dataset.

    Mirrors OfflineDataset (dataset.py:175-263): `len(ds)`, `ds[i]` (a pruned Trace, slow path), sorted indices for the
    sampler; plus the vectorised `gather` / `batch` / `loader` the hot path uses."""

    @classmethod
    def from_columns(cls, obs_names, obs_widths, trace_len, address_table, address_ids, values, prior, obs, trace_types=None):
        """An in-memory dataset from ragged columns (e.g. Model.prior_traces_packed): no files, same interface."""
        return cls([_MemoryShard(obs_names, obs_widths, trace_len, address_table, address_ids, values, prior, obs, trace_types)])

    def __init__(self, dataset_dir):
        if isinstance(dataset_dir, (list, tuple)) and dataset_dir and not isinstance(dataset_dir[0], str):
            paths, shards = [], list(dataset_dir)      # ready-made shard objects
        elif isinstance(dataset_dir, (list, tuple)):
            paths, shards = list(dataset_dir), None
        elif os.path.exists(os.path.join(dataset_dir, 'meta.json')):
            paths, shards = [dataset_dir], None
        else:
            paths, shards = sorted(os.path.join(dataset_dir, d) for d in os.listdir(dataset_dir)
                                   if os.path.exists(os.path.join(dataset_dir, d, 'meta.json'))), None
        if shards is None:
            if not paths:
                raise RuntimeError('no packed trace shards in {}'.format(dataset_dir))
            shards = [_Shard(p) for p in paths]
        self._shards = shards
        first = self._shards[0].meta
        self.obs_names, self.obs_widths = first['obs_names'], first['obs_widths']
        self.obs_width = int(sum(self.obs_widths))
        # global address / trace-type tables and per-shard remaps
        self.addresses, self._addr_id = [], {}
        self.trace_types, self._type_id = [], {}
        self._addr_remap, self._type_remap = [], []
        for s in self._shards:
            if s.meta['obs_names'] != self.obs_names or s.meta['obs_widths'] != self.obs_widths:
                raise RuntimeError('shards disagree on the observables')
            ar = []
            for a in s.meta['addresses']:
                key = a['address']
                if key not in self._addr_id:
                    self._addr_id[key] = len(self.addresses)
                    self.addresses.append((a['address'], a['distribution'], a['n_categories']))
                ar.append(self._addr_id[key])
            ar = np.asarray(ar, np.int32)
            tr = []
            for t in s.meta['trace_types']:
                key = tuple(int(ar[i]) for i in t['address_ids'])
                if key not in self._type_id:
                    self._type_id[key] = len(self.trace_types)
                    self.trace_types.append((int(t['hash'], 16), key))
                tr.append(self._type_id[key])
            self._addr_remap.append(ar)
            self._type_remap.append(np.asarray(tr, np.int32))
        self._first = np.concatenate([[0], np.cumsum([s.n for s in self._shards])]).astype(np.int64)
        self._length = int(self._first[-1])
        self.trace_len = np.concatenate([np.asarray(s.trace_len) for s in self._shards])
        self.trace_type = np.concatenate([r[np.asarray(s.trace_type)] for s, r in zip(self._shards, self._type_remap)])
        self._sorted = None

    def __len__(self):
        return self._length

    # ---- sampler support --------------------------------------------------------------------------------
    def sorted_indices(self):
        """Trace indices ordered by (length, type hash) (dataset.py:217-259); one shard is already in this order."""
        if self._sorted is None and len(self._shards) == 1:
            self._sorted = np.arange(self._length)
        if self._sorted is None:
            hashes = np.asarray([h for h, _ in self.trace_types], np.uint64)[self.trace_type]
            self._sorted = np.lexsort((np.arange(self._length), hashes, self.trace_len))
        return self._sorted

    def sampler(self, batch_size, rank=0, world_size=1, num_buckets=None, shuffle_batches=True, shuffle_buckets=True):
        from .parallel import DistributedTraceBatchSampler
        return DistributedTraceBatchSampler(self.sorted_indices(), batch_size, rank, world_size, num_buckets,
                                            shuffle_batches, shuffle_buckets)

    # ---- vectorised access ------------------------------------------------------------------------------
    def gather(self, indices):
        """Ragged columns of the given traces, in the given order: (trace_len [B], addr [R] global address ids,
        value [R], prior [R, 2], obs [B, W])."""
        idx = np.asarray(indices, np.int64)
        if idx.size == 0:
            raise ValueError('empty batch')
        if np.any(idx < 0) or np.any(idx >= self._length):
            raise IndexError('trace index out of range')
        shard = np.searchsorted(self._first, idx, side='right') - 1
        lens = self.trace_len[idx].astype(np.int64)
        off = np.concatenate([[0], np.cumsum(lens)])
        R = int(off[-1])
        addr = np.empty(R, np.int32)
        value = np.empty(R, np.float32)
        prior = np.empty((R, 2), np.float32)
        obs = np.empty((len(idx), self.obs_width), np.float32)
        for s_id in np.unique(shard):
            s = self._shards[s_id]
            sel = np.nonzero(shard == s_id)[0]
            local = idx[sel] - self._first[s_id]
            l = lens[sel]
            src0 = s.row_off[local]
            tot = int(l.sum())
            pos = np.concatenate([[0], np.cumsum(l)])[:-1]
            src = np.repeat(np.asarray(src0, np.int64) - pos, l) + np.arange(tot)
            dst = np.repeat(off[sel] - pos, l) + np.arange(tot)
            addr[dst] = self._addr_remap[s_id][s.addr[src]]
            value[dst] = s.value[src]
            prior[dst] = s.prior[src]
            obs[sel] = s.obs[local]
        return lens, addr, value, prior, obs

    def batch(self, indices, spec, out=None):
        """Host PackedBatch of the given traces for a network with address table `spec`: ONE call of the native packer
        (pp_pack_indexed) reads the shard columns (memory-mapped) in place and writes the step-major layout - no
        intermediate ragged arrays. An address the network does not know raises KeyError (the caller polymorphs
        first). `gather` + `PackedBatch.from_ragged` is the two-step equivalent (and the test oracle of this path)."""
        import ctypes as C
        from . import lib as L
        lib = L.load()
        ids = np.ascontiguousarray(indices, np.int64).reshape(-1)
        if ids.size == 0:
            raise ValueError('empty batch')
        if np.any(ids < 0) or np.any(ids >= self._length):
            raise IndexError('trace index out of range')
        n_addr = len(spec.addresses)
        self.native_columns(spec)
        lens = self.trace_len[ids]
        B, R, T = len(ids), int(lens.sum()), int(lens.max())
        words = lib.pp_pack_words(B, R, T, self.obs_width, n_addr)
        buf = np.empty(words, np.float32) if out is None else out(words)     # out: callable giving a float32 buffer
        info = L.pp_pack_info()
        rc = lib.pp_pack_indexed(self._native_shards, len(self._shards), self._first.ctypes.data, ids.ctypes.data, B,
                                 self.obs_width, n_addr, buf.ctypes.data, words, C.byref(info))
        if rc != 0:
            msg = lib.pp_last_error().decode()
            if 'address' in msg:
                raise KeyError('Address unknown by inference network ({})'.format(msg))
            raise RuntimeError('pp_pack_indexed failed: ' + msg)
        pb = PackedBatch._wrap_native(buf, info, self.obs_width, n_addr)
        bern = [a for a, info_ in enumerate(spec.addresses) if getattr(info_, 'dist_name', None) == 'Bernoulli']
        if bern:
            self._bernoulli_step_stats(pb, ids, bern)
        return pb

    def _bernoulli_step_stats(self, pb, ids, bernoulli_ids):
        """The `prior` pair of a row with a Bernoulli proposal = (n, sum of values) over the rows of its SUB-BATCH STEP
        (PP_HEAD_BERNOULLI, include/pyprob_amd.h; packed.bernoulli_group_stats for Trace objects): the reference scores
        every proposal of a sub-batch step against every value of it. Traces of one sub-batch share the address sequence,
        i.e. the dataset's trace type: grouped by (type, time step), written in place before the batch is uploaded."""
        types = self.trace_type[ids][np.asarray(pb.order)]            # type of every packed trace
        t_of_row = np.repeat(np.arange(pb.t_max), pb.n_active)        # time step of every packed row (step-major)
        rows = np.nonzero(np.isin(pb.addr, np.asarray(bernoulli_ids, np.int32)))[0]
        if rows.size == 0:
            return
        key = types[pb.trace[rows]].astype(np.int64) * (pb.t_max + 1) + t_of_row[rows]
        uniq, inv = np.unique(key, return_inverse=True)
        n = np.bincount(inv, minlength=len(uniq)).astype(np.float32)
        n1 = np.bincount(inv, weights=pb.value[rows].astype(np.float64), minlength=len(uniq)).astype(np.float32)
        pb.prior[rows, 0] = n[inv]
        pb.prior[rows, 1] = n1[inv]

    def native_columns(self, spec):
        """(pp_shard_columns array, n_shards, `first` array) of this dataset for a network with address table `spec`: the
        shard columns in place (memory-mapped files or arrays) + per-shard maps from shard-local address ids to the
        network's. Cached until the network grows."""
        from . import lib as L
        key = (id(spec), len(spec.addresses))
        if getattr(self, '_native_key', None) != key:      # (re)build the per-shard address maps for this network
            to_engine = np.asarray([spec.address_id.get(a[0], -1) for a in self.addresses], np.int32)
            self._native_remap = [np.ascontiguousarray(to_engine[r]) for r in self._addr_remap]
            arr = (L.pp_shard_columns * len(self._shards))()
            keep = []
            for k, s in enumerate(self._shards):
                cols = [np.asarray(getattr(s, n)) if not isinstance(getattr(s, n), np.memmap) else getattr(s, n)
                        for n in ('trace_len', 'row_off', 'obs', 'value', 'prior', 'addr')]
                keep.append(cols)
                (arr[k].trace_len, arr[k].row_off, arr[k].obs, arr[k].value, arr[k].prior, arr[k].addr) = [c.ctypes.data for c in cols]
                arr[k].addr_remap = self._native_remap[k].ctypes.data
            self._native_shards, self._native_keep, self._native_key = arr, keep, key
        return self._native_shards, len(self._shards), self._first

    def device_batch(self, indices, spec, device):
        """`batch(...).to(device)` through a ring of PINNED host buffers: the packer writes straight into page-locked
        memory and the upload is one asynchronous DMA (a pageable source is staged and synchronised by the runtime:
        55-570 us per copy measured; round-1 probe). A slot is reused only after its copy has completed."""
        import torch
        if torch.device(device).type != 'cuda':
            return self.batch(indices, spec).to(device)
        ring = _PINNED_RING        # process-wide: page-locking memory costs ~1 ms per buffer, datasets come and go
        depth = 8
        if len(ring['slots']) < depth:
            ring['slots'].append(dict(tensor=None, event=None))
        slot = ring['slots'][ring['next'] % len(ring['slots'])]
        ring['next'] += 1
        if slot['event'] is not None:
            slot['event'].synchronize()

        def out(words):
            if slot['tensor'] is None or slot['tensor'].numel() < words:
                slot['tensor'] = torch.empty(max(words, 1 << 16), dtype=torch.float32).pin_memory()
            return slot['tensor'].numpy()[:words]
        pb = self.batch(indices, spec, out=out)
        # the host-side arrays of the batch must outlive the slot: detach them from the ring
        for name in ('n_active', 'row_off', 'grp_off', 'nxt_off', 'order', 'src_row', 'cur_counts', 'prev_counts'):
            setattr(pb, name, np.array(getattr(pb, name)))
        pb._pinned = slot['tensor']
        pb.to(device)
        if slot['event'] is None:
            slot['event'] = torch.cuda.Event()
        slot['event'].record()
        pb._buf = pb._pinned = None       # the device copy is the batch now; host views into the ring are gone
        pb.obs = pb.value = pb.prior = pb.addr = pb.prev_row = pb.trace = pb.grp_rows = pb.nxt_rows = None
        return pb

    def types_of(self, indices):
        """Distinct trace types (ids into self.trace_types) of the given traces: one per reference sub-batch."""
        t = self.trace_type[np.asarray(indices, np.int64)]
        if len(self.trace_types) <= 64:                       # a bit set beats the sort inside np.unique
            return np.flatnonzero(np.bincount(t, minlength=len(self.trace_types)))
        return np.unique(t)

    def addresses_of(self, indices, types=None):
        """[(address, distribution name, n_categories)] used by the given traces, in first-statement order: what
        `_polymorph` needs (inference_network_lstm.py:34-80) without materialising Trace objects."""
        types = self.types_of(indices) if types is None else types
        seen, out = set(), []
        for t in types:
            for a in self.trace_types[t][1]:
                if a not in seen:
                    seen.add(a)
                    out.append(self.addresses[a])
        return out

    def loader(self, spec, batch_size, device, rank=0, world_size=1, num_buckets=None, prefetch=6, epochs=None,
               shuffle_batches=True, shuffle_buckets=True, workers=0):
        """Iterator of device PackedBatches, in sampler order. workers=0 (default): a minibatch is packed (`gather` + the
        native packer, ~110 us) and uploaded (one 40-60 KB H2D copy, ~55 us) in the consumer's thread, right before it is
        used - its 165 us hide behind the previous step's GPU time (measured 242 us per step = 4.2 M traces/s for GUM,
        round-1 probe). workers >= 1 moves that work to background threads with a bounded look-ahead; for these
        small minibatches that is SLOWER (437 us per step with one worker: the threads convoy on the GIL with the thread
        that enqueues the kernels), it only pays when packing is much heavier than a step. `epochs=None` repeats forever
        like the reference's training loop."""
        import torch
        sampler = self.sampler(batch_size, rank, world_size, num_buckets, shuffle_batches, shuffle_buckets)
        if int(workers) <= 0:
            e = 0
            while epochs is None or e < epochs:
                for ids in sampler:
                    yield self.device_batch(ids, spec, device)
                e += 1
            return
        lock = threading.Condition()
        state = dict(next_seq=0, done=False, error=None, want=0)
        ready = {}                       # seq -> uploaded batch
        stop = threading.Event()

        def tickets():
            e = 0
            while epochs is None or e < epochs:
                for ids in sampler:
                    yield ids
                e += 1
        source = tickets()

        def work():
            try:
                while not stop.is_set():
                    with lock:
                        while not stop.is_set() and state['next_seq'] - state['want'] >= max(int(prefetch), 1):
                            lock.wait(0.05)      # bounded look-ahead
                        if stop.is_set():
                            return
                        try:
                            ids = next(source)
                        except StopIteration:
                            state['done'] = True
                            lock.notify_all()
                            return
                        seq = state['next_seq']
                        state['next_seq'] += 1
                    # pack + upload on the device's default stream: the 40-60 KB copy is ordered before the consumer's
                    # kernels by the stream itself (a side stream + event per batch cost 460 us per minibatch in torch's
                    # per-stream allocator and event plumbing; round-1 probe)
                    host = self.batch(ids, spec).to(device)
                    with lock:
                        ready[seq] = host
                        lock.notify_all()
            except BaseException as exc:   # surfaced in the consumer
                with lock:
                    state['error'] = exc
                    lock.notify_all()

        threads = [threading.Thread(target=work, daemon=True) for _ in range(max(int(workers), 1))]
        for th in threads:
            th.start()
        try:
            while True:
                with lock:
                    while state['want'] not in ready and state['error'] is None and \
                            not (state['done'] and state['want'] >= state['next_seq']):
                        lock.wait(0.05)
                    if state['error'] is not None:
                        raise state['error']
                    if state['want'] not in ready:
                        return
                    batch = ready.pop(state['want'])
                    state['want'] += 1
                    lock.notify_all()
                yield batch
        finally:
            stop.set()
            with lock:
                lock.notify_all()

    # ---- OfflineDataset compatibility (slow path) ---------------------------------------------------------
    def __getitem__(self, i):
        """A pruned Trace like OfflineDataset.__getitem__ (dataset.py:197-205): controlled variables with address,
        value and prior distribution, named observed variables with their values."""
        from . import distributions as D
        from .trace import Trace, Variable
        if i < 0:
            i += self._length
        lens, addr, value, prior, obs = self.gather([i])
        tr = Trace()
        for r in range(int(lens[0])):
            address, dname, ncat = self.addresses[addr[r]]
            if dname == 'Normal':
                dist = D.Normal(float(prior[r, 0]), float(prior[r, 1]))
            elif dname == 'Uniform':
                dist = D.Uniform(float(prior[r, 0]), float(prior[r, 1]))
            elif dname == 'Categorical':
                dist = D.Categorical([1.0 / ncat] * ncat)
            elif dname == 'Poisson':    # the rate is not stored: training only reads the head's fixed interval
                dist = D.Poisson(1.0)
            elif dname == 'Bernoulli':  # (the probability is not stored either: the head reads the values only)
                dist = D.Bernoulli(0.5)
            else:
                raise RuntimeError('Distribution currently unsupported: {}'.format(dname))
            v = Variable(distribution=dist, value=float(value[r]), address=address, control=True)
            tr.add(v)
        c = 0
        for name, w in zip(self.obs_names, self.obs_widths):
            v = Variable(value=obs[0, c:c + w].copy(), address='__observed__' + name, name=name, observed=True)
            tr.add(v)
            c += w
        # a pruned trace carries no log-probabilities (dataset.py:64-119): fill the collections end() would fill
        tr.variables_controlled = [v for v in tr.variables if v.control]
        tr.variables_observed = [v for v in tr.variables if v.observed]
        tr.named_variables = {v.name: v for v in tr.variables if v.name is not None}
        tr.length = len(tr.variables)
        tr.length_controlled = len(tr.variables_controlled)
        return tr


class VectorisedOnlineDataset:
    """OnlineDataset (pyprob/nn/dataset.py:50-62) without the one-forward()-per-trace loop: chunks of `chunk_traces`
    prior traces are generated in lock step (Model.prior_traces_packed) and served as an in-memory PackedTraceDataset;
    `refresh()` draws the next chunk (online training never sees a trace twice)."""

    def __init__(self, model, obs_names, chunk_traces=65536, device='cpu', prior_inflation=None):
        from .state import PriorInflation
        self._model, self.obs_names, self._chunk, self._device = model, list(obs_names), int(chunk_traces), device
        self._prior_inflation = PriorInflation.DISABLED if prior_inflation is None else prior_inflation
        self.generated = 0
        self._worker = self._prepared = None
        self.resident = None          # device columns of the current chunk (single-statement programs on a device), else None
        self.refresh()

    def _generate(self):
        """(host dataset of the chunk, device columns or None). On a device: a program whose first chunk came out as ONE
        single-statement path keeps its later chunks on the device only (no host copy: the host dataset of the first chunk
        stays as the carrier of the address / trace-type tables); a program with several paths generates on a side stream, so
        that the branch decisions and the copies to the host do not queue behind the training steps in flight."""
        on_device = str(self._device) != 'cpu'
        have_ds, single = self.__dict__.get('_ds') is not None, self.__dict__.get('_single_path', False)   # (__getattr__ forwards to _ds)
        keep = on_device and single and have_ds
        side = None
        if on_device and not single and have_ds:
            import torch
            side = self.__dict__.setdefault('_gen_stream', torch.cuda.Stream(device=self._device))
        import contextlib
        import torch
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            cols = self._model.prior_traces_packed(self._chunk, self.obs_names, device=self._device, return_types=True,
                                                   prior_inflation=self._prior_inflation, resident_only=keep)
            if side is not None:
                side.synchronize()
        resident = getattr(self._model, '_last_prior_resident', None) if on_device else None
        self._model._last_prior_resident = None
        if cols is None:
            return self._ds, resident
        self._single_path = resident is not None
        return PackedTraceDataset.from_columns(self.obs_names, None, *cols), resident

    def refresh(self):
        """Serve the next chunk of fresh prior traces (the one `start_prefetch` prepared, if any)."""
        worker, self._worker = self._worker, None
        if worker is not None or self._prepared is not None:
            if worker is not None:
                worker.join()
            made, err = self._prepared
            self._prepared = None
            if err is not None:
                raise err
        else:
            made = self._generate()
        self._ds, self.resident = made
        self.generated += self._chunk

    def wait_prefetch(self):
        """Block until a running prefetch has finished (its chunk is kept for the next refresh): model code and the
        generator share the module-global trace state, so the caller waits before running the model itself."""
        if self._worker is not None:
            self._worker.join()

    def start_prefetch(self):
        """Generate the NEXT chunk in a worker thread while the caller trains on the current one. Meant for the native
        training loop: the trainer spends its time inside one C call per run of steps (pp_train_steps, GIL released), so
        the generator's Python runs concurrently. The caller must not execute model code until the next refresh()."""
        import threading
        if self._worker is not None or self._prepared is not None:
            return

        def work():
            try:
                self._prepared = (self._generate(), None)
            except BaseException as exc:      # noqa: BLE001 - re-raised by refresh() in the training thread
                self._prepared = (None, exc)
        self._worker = threading.Thread(target=work, name='pyprob_amd-prior-generator', daemon=True)
        self._worker.start()

    def __len__(self):
        return int(1e9)

    def __getitem__(self, i):
        return self._ds[i % len(self._ds)]

    def __getattr__(self, name):        # gather / batch / sampler / addresses_of / sorted_indices / trace_type ...
        return getattr(self._ds, name)


def save_dataset(model, dataset_dir, num_traces, num_traces_per_file, obs_names=None, *args, prior_inflation=None,
                 **kwargs):
    """Model.save_dataset (pyprob/model.py:227-232, pyprob/nn/dataset.py:50-62 + 121-144): run the model in
    PRIOR_FOR_INFERENCE_NETWORK mode and write `ceil(num_traces / num_traces_per_file)` packed shards."""
    from .state import PriorInflation, TraceMode
    prior_inflation = PriorInflation.DISABLED if prior_inflation is None else prior_inflation
    os.makedirs(dataset_dir, exist_ok=True)
    gen = model._trace_generator(trace_mode=TraceMode.PRIOR_FOR_INFERENCE_NETWORK, prior_inflation=prior_inflation,
                                 *args, **kwargs)
    existing = [d for d in os.listdir(dataset_dir) if d.startswith('pyprob_traces_packed_')]
    shard, written = len(existing), 0
    names = obs_names
    if names is None:    # one per-trace run tells which named variables are observed
        probe = next(gen)
        names = [k for k, v in probe.named_variables.items() if v.observed]     # recorded by observe(), not named samples
    vectorised = True
    try:                 # lock-step-safe programs are generated n traces at a time (Model.prior_traces_packed)
        model.prior_traces_packed(8, names, *args, **kwargs)
    except Exception:    # noqa: BLE001 - e.g. float(tensor) in the program: one forward() per trace
        vectorised = False
    while written < num_traces:
        n = min(num_traces_per_file, num_traces - written)
        path = os.path.join(dataset_dir, 'pyprob_traces_packed_{:06d}_{}'.format(shard, n))
        with PackedTraceWriter(path, names) as w:
            if vectorised:
                w.add_columns(*model.prior_traces_packed(n, names, *args, return_types=True,
                                                         prior_inflation=prior_inflation, **kwargs))
            else:
                for _ in range(n):
                    w.add_trace(next(gen))
        shard += 1
        written += n
    return shard
