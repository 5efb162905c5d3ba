"""The library's own RCCL communicator (csrc/dp.hip, pyprob_amd.parallel.init_native_comm) on a group of ONE rank - what a
1-GPU box can execute: the C-side exchange [flat grads | presence | loss | flag] gives the same training trajectory as the
torch.distributed path, the native loop pp_train_steps runs its data-parallel branch (all-reduce between backward and
Adam, reduced presence map / flag / loss), a non-finite batch is skipped through the reduced flag, and pieces around a
skipped range are reduced in one grouped launch. Multi-rank behaviour is RCCL's; the buffer layout and the skip logic
are covered by tests/test_dp_gloo.py."""
import os

import numpy as np
import pytest

from helpers import synthetic_gum_arrays, synthetic_gumm_arrays

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def one_rank_group():
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29700 + os.getpid() % 2000))
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    from pyprob_amd import lib as L
    from pyprob_amd.parallel import init_native_comm
    lib = L.load()
    assert init_native_comm(torch.device('cuda', 0), lib), lib.pp_last_error()
    assert lib.pp_dp_world() == 1
    yield lib
    lib.pp_dp_destroy()
    assert lib.pp_dp_world() == 0
    if created:
        dist.destroy_process_group()


def _engine(H, addresses, dist_name, seed):
    from pyprob_amd.engine import ICEngine
    from pyprob_amd.spec import NetSpec
    spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=H)
    for a in addresses:
        spec.add_address(a, dist_name)
    return ICEngine(spec, device='cuda:0', seed=seed)


def test_native_exchange_equals_the_torch_exchange(one_rank_group):
    lib = one_rank_group
    from pyprob_amd.packed import PackedBatch
    arr = synthetic_gum_arrays(512, seed=2)
    runs = {}
    for mode in ('torch', 'native'):
        eng = _engine(128, ['mu'], 'Normal', seed=9)
        eng.force_allreduce = True
        eng.skip_recurrent_weights(True)
        eng.native_dp = mode == 'native'        # (False: the torch.distributed path)
        ids = np.zeros(512, np.int64)
        pb = PackedBatch.from_ragged(arr['trace_len'], ids, arr['values'], arr['prior'], arr['obs'], 1).to(eng.device)
        losses = [float(eng.train_step(pb, 1e-3).item()) for _ in range(6)]
        torch.cuda.synchronize()
        runs[mode] = (losses, eng.params.cpu().numpy().copy(), float(eng.status_tail.item()))
    np.testing.assert_allclose(runs['native'][0], runs['torch'][0], rtol=2e-5)
    d = np.abs(runs['native'][1] - runs['torch'][1])
    assert np.linalg.norm(d) < 1e-3 * np.linalg.norm(runs['torch'][1])
    assert runs['native'][2] == 0.0


def test_native_loop_runs_its_data_parallel_branch(one_rank_group):
    from pyprob_amd.dataset import PackedTraceDataset
    arrays, addresses = synthetic_gumm_arrays(3000, seed=12, max_iter=4)
    table = [(a, 'Uniform', None) for a in addresses]
    ds = PackedTraceDataset.from_columns(['obs0', 'obs1'], [1, 1], arrays['trace_len'], table, arrays['addr_idx'],
                                         arrays['values'], arrays['prior'], arrays['obs'])
    rng = np.random.default_rng(5)
    steps = [rng.choice(3000, size=n, replace=False) for n in (256, 200, 31, 256, 128, 256)]
    lrs = [1e-3] * len(steps)
    a, b = _engine(64, addresses, 'Uniform', seed=3), _engine(64, addresses, 'Uniform', seed=3)
    la, sa = a.train_run(ds, steps, lrs)                  # single-rank loop
    b.force_allreduce = True                              # data-parallel branch over the one-rank communicator
    b.native_dp = True
    lb, sb = b.train_run(ds, steps, lrs)
    torch.cuda.synchronize()
    np.testing.assert_allclose(lb.cpu().numpy(), la.cpu().numpy(), rtol=2e-5)
    assert not sa.cpu().numpy().any() and not sb.cpu().numpy().any()
    pa, pb_ = a.params.cpu().numpy(), b.params.cpu().numpy()
    assert np.linalg.norm(pa - pb_) < 1e-3 * np.linalg.norm(pa)
    # a non-finite minibatch: flagged through the REDUCED tail, nothing moves
    before = b.params.clone()
    arrays2 = dict(arrays)
    arrays2['obs'] = arrays['obs'].copy()
    arrays2['obs'][7, 0] = np.nan
    ds2 = PackedTraceDataset.from_columns(['obs0', 'obs1'], [1, 1], arrays2['trace_len'], table, arrays2['addr_idx'],
                                          arrays2['values'], arrays2['prior'], arrays2['obs'])
    losses, status = b.train_run(ds2, [np.array([7, 1, 2, 3])], [1e-3])
    torch.cuda.synchronize()
    assert int(status.cpu().numpy()[0]) == 1
    assert torch.equal(before, b.params)


def test_resident_loop_with_the_exchange(one_rank_group):
    """pp_train_resident with its data-parallel branch (one-rank communicator) = the per-step loop with the native exchange."""
    from pyprob_amd.packed import PackedBatch
    a, b = _engine(64, ['mu'], 'Normal', seed=6), _engine(64, ['mu'], 'Normal', seed=6)
    for eng in (a, b):
        eng.force_allreduce = True
        eng.native_dp = True
        eng.skip_recurrent_weights(True)
    ba, bb = [], []
    for k, n in enumerate((256, 64, 300)):
        arr = synthetic_gum_arrays(n, seed=40 + k)
        for eng, lst in ((a, ba), (b, bb)):
            lst.append(PackedBatch.from_ragged(arr['trace_len'], np.zeros(n, np.int64), arr['values'], arr['prior'], arr['obs'],
                                               1).to(eng.device))
    for pb in ba:
        a.train_step(pb, 1e-3)
    losses, status = b.train_resident(bb, [1e-3] * 3)
    torch.cuda.synchronize()
    assert not status.cpu().numpy().any() and np.isfinite(losses.cpu().numpy()).all()
    pa, pb_ = a.params.cpu().numpy(), b.params.cpu().numpy()
    assert np.linalg.norm(pa - pb_) < 1e-3 * np.linalg.norm(pa)


def test_grouped_pieces(one_rank_group):
    lib = one_rank_group
    import ctypes as C
    from pyprob_amd import lib as L
    x = torch.arange(4096, dtype=torch.float32, device='cuda:0')
    ref = x.clone()
    off = (C.c_int64 * 3)(0, 1024, 3000)
    cnt = (C.c_int64 * 3)(512, 1000, 1096)
    L.check(lib.pp_dp_allreduce(x.data_ptr(), off, cnt, 3, L.stream_ptr()), 'pp_dp_allreduce')
    torch.cuda.synchronize()
    assert torch.equal(x, ref)          # one rank: the sum over the ranks is the buffer itself


@pytest.mark.parametrize('shape', ['single_statement_panel', 'ragged'])
def test_early_bucket_on_the_side_stream_changes_no_bit(one_rank_group, shape, monkeypatch):
    """pp_dp_overlap (ABI 13): the backward pass issues its weight-gradient launch in two parts, the LSTM layer's gradient
    ranges are reduced on a SIDE stream under the second part, pp_dp_reduce_grads reduces the rest and joins the streams
    before Adam - per-step calls, the resident loop and the packing loop, on the one-rank communicator. The same
    training trajectory as the single exchange (the split changes WHEN products run, not what they compute), and the
    event-pair statistics arrive."""
    lib = one_rank_group
    import ctypes as C
    from pyprob_amd.packed import PackedBatch
    monkeypatch.setenv('PP_DP_OVERLAP', '1')          # (off by default: profiles/r06e_dp_overlap_one_rank.txt)
    if shape == 'single_statement_panel':
        H, addresses, dist_name = 512, ['mu'], 'Normal'
        arrs = [synthetic_gum_arrays(1024, seed=70 + k) for k in range(3)]
        ids = [np.zeros(1024, np.int64)] * 3
    else:
        H, dist_name = 64, 'Uniform'
        arrs = []
        for k in range(3):
            arr, addresses = synthetic_gumm_arrays(300, seed=80 + k, max_iter=4)
            arrs.append(arr)
        ids = None
    runs = {}
    for mode in ('single', 'early'):
        eng = _engine(H, addresses, dist_name, seed=11)
        eng.force_allreduce = True
        eng.native_dp = True
        single = shape == 'single_statement_panel'
        eng.skip_recurrent_weights(single)
        ranges = eng.enable_dp_overlap(mode == 'early')
        assert (len(ranges) == (2 if single else 1)) if mode == 'early' else ranges == []
        batches = []
        for k, arr in enumerate(arrs):
            a_ids = ids[k] if ids is not None else np.array([eng.spec.address_id[addresses[j]] for j in arr['addr_idx']])
            batches.append(PackedBatch.from_ragged(arr['trace_len'], a_ids, arr['values'], arr['prior'], arr['obs'],
                                                   len(eng.spec.addresses)).to(eng.device))
        losses = [float(eng.train_step(pb, 1e-3).item()) for pb in batches]                  # per-step C calls
        l2, st = eng.train_resident(batches, [1e-3] * 3)                                      # the resident loop
        if mode == 'early':
            assert lib.pp_dp_overlap_stats(1, None) == 0
            eng.train_step(batches[0], 1e-3)
            us = (C.c_float * 3)()
            assert lib.pp_dp_overlap_stats(0, us) == 1 and all(0.0 <= v < 1e5 for v in us) and us[0] > 0.0
        else:
            eng.train_step(batches[0], 1e-3)
        torch.cuda.synchronize()
        runs[mode] = (losses, l2.cpu().numpy().copy(), eng.params.cpu().numpy().copy(), eng.tensor_step.cpu().numpy().copy())
        eng.enable_dp_overlap(False)
    # (float atomics order the row splits of a weight-gradient tile differently from launch to launch - with or without the
    # two-part launch -: 2e-6 on the losses, not bitwise)
    np.testing.assert_allclose(runs['early'][0], runs['single'][0], rtol=2e-6)
    np.testing.assert_allclose(runs['early'][1], runs['single'][1], rtol=2e-5)
    assert np.linalg.norm(runs['early'][2] - runs['single'][2]) < 1e-3 * np.linalg.norm(runs['single'][2])
    assert (runs['early'][3] == runs['single'][3]).all()
