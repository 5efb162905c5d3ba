"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol (no compute calls without a
GPU), the packed-batch builder, the parameter layout, the build recipe."""
import os
import re

import numpy as np
import pytest

from conftest import REPO, load_golden
from helpers import spec_from_golden, packed_from_golden, synthetic_gumm_arrays


@pytest.fixture(scope='module')
def lib():
    from pyprob_amd import build as B
    B.build()
    from pyprob_amd import lib as L
    return L.load()


def test_library_exports_every_header_symbol(lib):
    hdr = open(os.path.join(REPO, 'include', 'pyprob_amd.h')).read()
    declared = set(re.findall(r'\b(pp_[a-z0-9_]+)\s*\(', hdr))
    from pyprob_amd import lib as L
    assert declared == set(L.PROTOTYPES.keys()), declared ^ set(L.PROTOTYPES.keys())
    for name in declared:
        assert hasattr(lib, name)
    hdr_abi = int(re.search(r'#define PP_ABI_VERSION (\d+)', hdr).group(1))
    assert lib.pp_abi_version() == L.PP_ABI_VERSION == hdr_abi
    # the document a maintainer binds from quotes the same number (INTEGRATION.md said 8 under a header at 11)
    doc = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    assert [int(x) for x in re.findall(r'pp_abi_version\(\) == (\d+)', doc)] == [hdr_abi]


def test_struct_sizes_match_header_layout():
    import ctypes as C
    from pyprob_amd import lib as L
    assert C.sizeof(L.pp_addr) == 6 * 4 + 8 * 8
    assert C.sizeof(L.pp_gemm_args) == 9 * 8 + 5 * 4 + 4 + 4 * 8 + 2 * 4 + 8 + 8  # incl. padding after b_kmajor
    assert C.sizeof(L.pp_batch) % 8 == 0


def test_no_gpu_reports_zero_devices_or_more(lib):
    assert lib.pp_device_count() >= 0


def test_product_refuses_to_run_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from pyprob_amd.engine import ICEngine
    from pyprob_amd.lib import HipLibraryError
    meta, params, batch, loss, isr = load_golden('gum')
    with pytest.raises(HipLibraryError):
        ICEngine(spec_from_golden(meta, params))


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(REPO, 'pyprob_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r'^\s*(from|import)\s+[^\n]*oracle', src, re.M), os.path.join(root, f)
                assert 'ic_oracle' not in src and 'sys.path' not in src, os.path.join(root, f)


def test_param_layout_matches_reference_state_dict(golden):
    case, meta, params, batch, loss, isr = golden
    spec = spec_from_golden(meta, params)
    assert set(spec.tensors.keys()) == set(params.keys())
    for n, (off, shape) in spec.tensors.items():
        assert shape == params[n].shape, n
        assert off % 1024 == 0
    assert spec.num_parameters() == meta['num_params']
    m = spec.chunk_tensor_map()
    assert len(m) == spec.n_params // 1024 and m[0] == 0 and m[-1] == spec.n_tensors - 1


def test_survey_parameter_counts():
    from pyprob_amd.spec import NetSpec
    obs = {'obs0': {'dim': 32}, 'obs1': {'dim': 32}}
    for H, count in ((512, 1643583), (1024, 5636415), (64, 85215)):   # SURVEY.md Appendix B / notebook :403
        s = NetSpec(obs, lstm_dim=H)
        s.add_address('a', 'Normal')
        assert s.num_parameters() == count
    s = NetSpec(obs, lstm_dim=512)
    for i in range(10):
        s.add_address('u%d' % i, 'Uniform')
    assert s.num_parameters() == 2968878   # GUMM notebook, 10 addresses


def test_packed_batch_structure(golden):
    case, meta, params, batch, loss, isr = golden
    spec = spec_from_golden(meta, params)
    pb = packed_from_golden(meta, batch, spec)
    B, R = len(batch['trace_len']), int(batch['trace_len'].sum())
    assert pb.n_traces == B and pb.n_rows == R
    lens = batch['trace_len'][pb.order]
    assert np.all(np.diff(lens) <= 0)
    assert pb.row_off[-1] == R and np.all(np.diff(pb.n_active) <= 0)
    # every packed row maps to a distinct source row, values/addresses carried over
    assert sorted(pb.src_row.tolist()) == list(range(R))
    np.testing.assert_array_equal(pb.value, batch['values'][pb.src_row])
    # prev_row walks back one time step of the same trace
    for r in range(R):
        p = pb.prev_row[r]
        if p >= 0:
            assert pb.trace[p] == pb.trace[r] and pb.src_row[p] == pb.src_row[r] - 1
    # head groups partition the rows by address
    for a in range(len(spec.addresses)):
        rows = pb.grp_rows[pb.grp_off[a]:pb.grp_off[a + 1]]
        assert np.all(pb.addr[rows] == a)
        nxt = pb.nxt_rows[pb.nxt_off[a]:pb.nxt_off[a + 1]]
        assert np.all(pb.addr[pb.prev_row[nxt]] == a)
    assert pb.grp_off[-1] == R and pb.nxt_off[-1] == R - B
    assert abs(pb.mean_length_controlled - R / B) < 1e-12


def test_packed_rejects_empty_traces():
    from pyprob_amd.packed import PackedBatch
    with pytest.raises(ValueError):
        PackedBatch.from_ragged([1, 0], [0], [0.0], np.zeros((1, 2)), np.zeros((2, 2)), 1)
    with pytest.raises(ValueError):
        PackedBatch.from_ragged([], [], [], np.zeros((0, 2)), np.zeros((0, 2)), 1)


def test_active_mask_matches_reference_has_grad(golden):
    case, meta, params, batch, loss, isr = golden
    spec = spec_from_golden(meta, params)
    pb = packed_from_golden(meta, batch, spec)
    act = spec.active_mask(pb.cur_counts, pb.prev_counts)
    names = list(spec.tensors.keys())
    ref = dict(zip(meta['param_names'], meta['has_grad']))
    for i, n in enumerate(names):
        assert bool(act[i]) == bool(ref[n]), n
    # the key the engine caches presence maps under: which addresses appear as current / previous variable; computed once per batch
    key = pb.presence_key
    assert key == (tuple(bool(v) for v in pb.cur_counts > 0), tuple(bool(v) for v in pb.prev_counts > 0))
    assert pb.presence_key is key and hash(key) == hash((tuple(pb.cur_counts > 0), tuple(pb.prev_counts > 0)))


def test_ragged_synthetic_generator_lengths():
    arr, addresses = synthetic_gumm_arrays(2000, seed=1)
    # GUMM trace length: mean 2/(pi/4) = 2.546 (reference tests/test_model.py:80 pins 2.563 +- tolerance)
    assert abs(arr['trace_len'].mean() - 2.546) < 0.1
    assert arr['trace_len'].min() == 2


def test_native_packer_matches_the_numpy_statement():
    """pp_pack_ragged (C ABI, host code) against packed.PackedBatch.from_ragged_numpy on ragged / uniform / single-trace
    minibatches, wide prior columns, and its error behaviour (dataset.py:28-29)."""
    from pyprob_amd.packed import PackedBatch
    rng = np.random.default_rng(5)
    fields = ('obs', 'value', 'prior', 'addr', 'prev_row', 'trace', 'n_active', 'row_off', 'grp_rows', 'grp_off', 'nxt_rows',
              'nxt_off', 'order', 'src_row', 'cur_counts', 'prev_counts')
    for B, max_len, n_addr, pw, W in ((1, 1, 1, 2, 2), (7, 1, 3, 2, 1), (64, 5, 4, 3, 2), (1024, 12, 12, 2, 2), (33, 3, 2, 1, 0)):
        lens = rng.integers(1, max_len + 1, B)
        R = int(lens.sum())
        ids = rng.integers(0, n_addr, R)
        vals = rng.normal(size=R).astype(np.float32)
        prior = rng.normal(size=(R, pw)).astype(np.float32)
        obs = rng.normal(size=(B, W)).astype(np.float32)
        a = PackedBatch.from_ragged(lens, ids, vals, prior, obs, n_addr)
        b = PackedBatch.from_ragged_numpy(lens, ids, vals, prior, obs, n_addr)
        for n in fields:
            assert np.array_equal(getattr(a, n), getattr(b, n)), (B, n)
        assert a.t_max == b.t_max and a.n_rows == b.n_rows and abs(a.mean_length_controlled - b.mean_length_controlled) < 1e-12
    with pytest.raises(ValueError, match='Trace of length zero'):
        PackedBatch.from_ragged([2, 0], [0, 0], [0.0, 0.0], np.zeros((2, 2)), np.zeros((2, 1)), 1)
    with pytest.raises(ValueError):
        PackedBatch.from_ragged([], [], [], np.zeros((0, 2)), np.zeros((0, 1)), 1)
    with pytest.raises(RuntimeError):      # address id outside the table: rejected by the C side
        PackedBatch.from_ragged([1], [5], [0.0], np.zeros((1, 2)), np.zeros((1, 1)), 2)


def test_tensor_roles_reproduce_the_presence_map(golden):
    """spec.tensor_roles() (the tables of pp_train_steps) evaluates to spec.active_mask() for any occurrence pattern."""
    case, meta, params, batch, loss, isr = golden
    spec = spec_from_golden(meta, params)
    off, addr, role = spec.tensor_roles()
    assert len(off) == spec.n_tensors + 1 and len(role) == spec.n_tensors
    rng = np.random.default_rng(0)
    n = len(spec.addresses)
    for _ in range(20):
        cur = rng.integers(0, 2, n) * rng.integers(1, 5, n)
        prev = rng.integers(0, 2, n) * rng.integers(1, 5, n)
        want = spec.active_mask(cur, prev)
        got = np.zeros(spec.n_tensors, np.float32)
        for t in range(spec.n_tensors):
            on = bool(role[t] & 4)
            for a in addr[off[t]:off[t + 1]]:
                on = on or bool(role[t] & 1 and cur[a] > 0) or bool(role[t] & 2 and prev[a] > 0)
            got[t] = on
        assert np.array_equal(got, want)


def test_train_steps_rejects_incomplete_arguments(lib):
    """pp_train_steps validates its arguments before touching the device (callable without a GPU)."""
    import ctypes as C
    from pyprob_amd import lib as L
    assert lib.pp_train_steps(None, None, None, None, 0, None, 0, None, None, 0, None, 0.9, 0.999, 1e-8, 0.0, 0, None, None) == -1
    assert b'null pointer' in lib.pp_last_error()
    spec = spec_from_golden(*load_golden('gum')[:2])
    net = spec.c_struct(0)
    tb, roles = L.pp_train_buffers(), L.pp_tensor_roles()
    one = np.zeros(4, np.int64)
    lr = np.zeros(1, np.float32)
    shards = (L.pp_shard_columns * 1)()
    rc = lib.pp_train_steps(C.byref(net), C.byref(tb), C.byref(roles), shards, 1, one.ctypes.data, 2, one.ctypes.data,
                            one.ctypes.data, 0, lr.ctypes.data, 0.9, 0.999, 1e-8, 0.0, 0, None, None)
    assert rc == -1 and b'pp_train_buffers' in lib.pp_last_error()
    assert lib.pp_train_slot_words(1024, 1024, 1, 2, 1, 20) >= lib.pp_pack_words(1024, 1024, 1, 2, 1) + 20


def test_bench_stdout_carries_only_the_json_line():
    """bench.py's contract: rank 0 prints ONE JSON line. RCCL writes a version banner to the C-level stdout at its first
    communicator (seen under torch.distributed.run): bench.py claims file descriptor 1 at start and sends everything but the
    line to stderr."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import ctypes, sys, bench\n"
            "fd = bench._claim_stdout()\n"
            "libc = ctypes.CDLL(None)\n"
            "libc.puts(b'banner written by a C library')\n"
            "libc.fflush(None)\n"
            "print('python chatter')\n"
            "bench._emit_line(fd, '{\"ok\": 1}')\n")
    r = subprocess.run([sys.executable, '-c', code], cwd=repo, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"ok": 1}\n'
    assert 'banner written by a C library' in r.stderr and 'python chatter' in r.stderr


# ---- host logic of the streaming weight-gradient launch (csrc/wgrad_t1.hip), no device work -----------------------------
def _wgrad_plan(products, zero_blocks=None):
    """products: dicts M, N, K, lda, ldb, ldc [, gather] -> list of problems {M, N, K, S, ks, gather, first, c_off, a_off, b_off}."""
    import ctypes as C
    from pyprob_amd import lib as L
    lib = L.load()
    n = len(products)
    arr = (L.pp_gemm_args * n)()
    for i, p in enumerate(products):
        base = 0x10000000 * (i + 1)                      # fake device addresses: never dereferenced by the planner
        a = arr[i]
        a.A, a.lda, a.B, a.ldb, a.C, a.ldc = base, p['lda'], base + 0x4000000, p['ldb'], base + 0x8000000, p['ldc']
        a.b_idx = base + 0xC000000 if p.get('gather') else None
        a.M, a.N, a.K, a.a_kmajor, a.b_kmajor, a.accumulate, a.split_k = p['M'], p['N'], p['K'], 1, 1, 1, 1
        for k, v in p.get('extra', {}).items():
            setattr(a, k, v)
    zb = None
    if zero_blocks is not None:
        zb = np.ascontiguousarray(zero_blocks, np.int32).reshape(n, 2, 6)
    out = np.zeros((64, 10), np.int64)
    nb = C.c_int32(0)
    k = lib.pp_debug_wgrad_plan(arr, zb.ctypes.data if zb is not None else None, n, out.ctypes.data, 64, C.byref(nb))
    keys = ('M', 'N', 'K', 'S', 'ks', 'gather', 'first', 'c_off', 'a_off', 'b_off')
    return [dict(zip(keys, map(int, out[i]))) for i in range(k)], nb.value


def test_wgrad_plan_single_statement_step(monkeypatch):
    """The products of a GUM step (B = 1024, H = 512): dW_ih 2048 x 68 with the forget-gate rows and the previous-variable
    columns zero for ALL rows (left out), the head and observe-embedding leaves; ~240 workgroups -> three row ranges each."""
    monkeypatch.delenv('PP_DETERMINISTIC', raising=False)
    H, B, I = 512, 1024, 212
    prods = [dict(M=4 * H, N=68, K=B, lda=4 * H, ldb=68, ldc=I), dict(M=30, N=271, K=B, lda=32, ldb=272, ldc=271),
             dict(M=271, N=512, K=B, lda=272, ldb=512, ldc=512), dict(M=64, N=64, K=B, lda=64, ldb=64, ldc=64),
             dict(M=32, N=16, K=B, lda=64, ldb=16, ldc=16)]
    zb = np.zeros((5, 2, 6), np.int32)
    zb[0, 0] = (0, 4 * H, 64, 68, 0, B)                  # columns [64, 68): no previous variable at t = 0
    zb[0, 1] = (H, 2 * H, 0, 68, 0, B)                   # forget gate: c_{-1} = 0
    plan, blocks = _wgrad_plan(prods, zb)
    ih = [p for p in plan if p['N'] == 64 and p['M'] in (512, 1024)]
    assert sorted((p['M'], p['c_off'], p['a_off']) for p in ih) == [(512, 0, 0), (1024, 2 * H * I, 2 * H)]
    assert all(p['K'] == B and p['S'] == 3 and p['ks'] % 4 == 0 and p['ks'] * 3 >= B and not p['gather'] for p in plan)
    assert len(plan) == 6
    tiles = sum(-(-p['M'] // 64) * -(-p['N'] // 64) for p in plan)
    assert tiles == 8 + 16 + 5 + 40 + 1 + 1 and blocks == 3 * tiles
    firsts = sorted(p['first'] for p in plan)
    assert firsts[0] == 0 and len(set(firsts)) == len(plan)          # the problems tile the launch without overlap


def test_wgrad_plan_ragged_step_prefix_blocks_and_gather(monkeypatch):
    """A ragged step: the zero blocks of dW_ih cover only the first B rows (first time steps) - those cells START at row B;
    dW_hh gathers h_{t-1} by prev_row (the index pointer moves with the first row, the operand pointer does not)."""
    monkeypatch.delenv('PP_DETERMINISTIC', raising=False)
    H, B, R, I = 512, 1024, 2600, 212
    prods = [dict(M=4 * H, N=68, K=R, lda=4 * H, ldb=68, ldc=I), dict(M=4 * H, N=H, K=R - B, lda=4 * H, ldb=H, ldc=H, gather=True)]
    zb = np.zeros((2, 2, 6), np.int32)
    zb[0, 0] = (0, 4 * H, 64, 200, 0, B)
    zb[0, 1] = (H, 2 * H, 0, 68, 0, B)
    plan, blocks = _wgrad_plan(prods, zb)
    cells = {(p['c_off'], p['M'], p['N']): p for p in plan if not p['gather']}
    assert set(cells) == {(0, 512, 64), (H * I, 512, 64), (2 * H * I, 1024, 64), (64, 2048, 4)}
    assert cells[(0, 512, 64)]['K'] == R and cells[(0, 512, 64)]['a_off'] == 0
    assert cells[(H * I, 512, 64)]['K'] == R - B and cells[(H * I, 512, 64)]['a_off'] == B * 4 * H + H      # rows from B on
    assert cells[(H * I, 512, 64)]['b_off'] == B * 68
    assert cells[(64, 2048, 4)]['K'] == R - B and cells[(64, 2048, 4)]['b_off'] == B * 68 + 64
    hh = [p for p in plan if p['gather']]
    assert len(hh) == 1 and hh[0]['M'] == 2048 and hh[0]['N'] == 512 and hh[0]['K'] == R - B and hh[0]['b_off'] == 0
    assert plan[0]['K'] == max(p['K'] for p in plan)                  # longest row ranges first
    assert all(p['S'] >= 1 and p['ks'] * p['S'] >= p['K'] and p['K'] // max(p['S'], 1) >= 64 for p in plan)
    assert blocks == sum(-(-p['M'] // 64) * -(-p['N'] // 64) * p['S'] for p in plan)


def test_wgrad_plan_refuses_what_only_the_tile_kernels_do(monkeypatch):
    monkeypatch.delenv('PP_DETERMINISTIC', raising=False)
    ok = dict(M=64, N=64, K=256, lda=64, ldb=64, ldc=64)
    assert _wgrad_plan([ok])[0]
    for extra in (dict(a_kmajor=0), dict(b_kmajor=0), dict(accumulate=0), dict(relu=1), dict(a_idx=0x1000), dict(c_idx=0x1000),
                  dict(colsum=0x1000), dict(bias=0x1000)):
        assert _wgrad_plan([dict(ok, extra=extra)])[0] == [], extra
    assert _wgrad_plan([dict(ok)] * 41)[0] == []                       # more problems than one launch carries
