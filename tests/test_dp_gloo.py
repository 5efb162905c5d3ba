"""world_size-2 data-parallel protocol on CPU (gloo): one all-reduce over [flat grads | presence | loss] reproduces
the reference's per-tensor gradient sync (pyprob/nn/inference_network.py:296-333); bucketed sampler; particle shards.
The gradients come from the oracle (test infrastructure) -- no GPU needed."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, load_golden
from helpers import spec_from_golden


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:      # (a fixed port can collide with an ephemeral one)
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import ic_oracle as O
    from pyprob_amd.parallel import allreduce_flat_, finish_reduce
    meta, params, batch, loss, isr = load_golden('gumm')
    spec = spec_from_golden(meta, params)
    # rank-local minibatch: traces [rank::world] of the golden batch
    B = len(batch['trace_len'])
    off = np.concatenate([[0], np.cumsum(batch['trace_len'])])
    idx = np.arange(rank, B - (B % world), world)
    rows = np.concatenate([np.arange(off[b], off[b + 1]) for b in idx])
    local = dict(trace_len=batch['trace_len'][idx], addr_idx=batch['addr_idx'][rows], values=batch['values'][rows],
                 prior=batch['prior'][rows], obs=batch['obs'][idx])
    net = O.Net(params, meta['obs_names'], K=10)
    o = O.loss_and_grads(net, local, meta['addresses'], meta['dist_names'])
    # flat buffer [grads | presence | loss] in the product's layout
    buf = torch.zeros(spec.n_params + spec.n_tensors + 1, dtype=torch.float64)
    names = list(spec.tensors.keys())
    for n in names:
        o_, shape = spec.tensors[n]
        buf[o_:o_ + int(np.prod(shape))] = torch.from_numpy(o['grads'][n].reshape(-1))
    ids = np.array([spec.address_id[meta['addresses'][i]] for i in local['addr_idx']])
    cur = np.bincount(ids, minlength=len(spec.addresses))
    lens = local['trace_len']
    loff = np.concatenate([[0], np.cumsum(lens)])
    not_last = np.ones(len(ids), bool)
    not_last[loff[1:] - 1] = False
    prev = np.bincount(ids[not_last], minlength=len(spec.addresses))
    buf[spec.n_params:spec.n_params + spec.n_tensors] = torch.from_numpy(spec.active_mask(cur, prev).astype(np.float64))
    buf[-1] = o['loss']
    allreduce_flat_(buf)
    grads, active, l = finish_reduce(buf, spec.n_params, spec.n_tensors, world)
    if rank == 0:
        torch.save(dict(grads=grads.clone(), active=active.clone(), loss=l.clone(), idx=idx), out)
    dist.barrier()
    dist.destroy_process_group()


def test_flat_allreduce_equals_global_batch_gradient(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / 'dp.pt')
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r = torch.load(out, weights_only=False)
    from oracle import ic_oracle as O
    meta, params, batch, loss, isr = load_golden('gumm')
    spec = spec_from_golden(meta, params)
    B = len(batch['trace_len'])
    keep = np.arange(B - (B % world))
    off = np.concatenate([[0], np.cumsum(batch['trace_len'])])
    rows = np.concatenate([np.arange(off[b], off[b + 1]) for b in keep])
    glob = dict(trace_len=batch['trace_len'][keep], addr_idx=batch['addr_idx'][rows], values=batch['values'][rows],
                prior=batch['prior'][rows], obs=batch['obs'][keep])
    o = O.loss_and_grads(O.Net(params, meta['obs_names'], K=10), glob, meta['addresses'], meta['dist_names'])
    # mean of the per-rank mean losses == global mean loss (equal per-rank batch sizes); same for gradients
    assert abs(float(r['loss']) - o['loss']) < 1e-9
    for n, (o_, shape) in spec.tensors.items():
        g = r['grads'][o_:o_ + int(np.prod(shape))].numpy().reshape(shape)
        np.testing.assert_allclose(g, o['grads'][n], rtol=1e-9, atol=1e-12)
    # merged presence: a tensor is active if any rank touched it
    ids = np.array([spec.address_id[meta['addresses'][i]] for i in glob['addr_idx']])
    assert np.all((r['active'].numpy() > 0) >= (spec.active_mask(np.bincount(ids, minlength=len(spec.addresses)),
                                                                 np.zeros(len(spec.addresses))) > 0))


def test_distributed_sampler_partitions_batches():
    from pyprob_amd.parallel import DistributedTraceBatchSampler
    idx = list(range(1000))
    world = 4
    per_rank = []
    for rank in range(world):
        s = DistributedTraceBatchSampler(idx, batch_size=16, rank=rank, world_size=world, num_buckets=5,
                                         shuffle_batches=False)
        per_rank.append([tuple(b) for b in s])
    n = {len(p) for p in per_rank}
    assert len(n) == 1                                        # same number of iterations on every rank
    flat = [b for p in per_rank for b in p]
    assert len(set(flat)) == len(flat)                        # disjoint minibatches
    assert all(len(b) == 16 for b in flat)
    # all ranks walk buckets in the same order: the k-th batch of every rank comes from the same bucket
    s0 = DistributedTraceBatchSampler(idx, 16, 0, world, num_buckets=5, shuffle_batches=False)
    bucket_of = {tuple(b): i for i, bk in enumerate(s0._buckets) for b in bk}
    for k in range(len(per_rank[0])):
        assert len({bucket_of[p[k]] for p in per_rank}) == 1
    with pytest.raises(RuntimeError):
        DistributedTraceBatchSampler(list(range(64)), 16, 0, 8, num_buckets=4)


def test_particle_shards_cover_the_range():
    from pyprob_amd.parallel import shard_range
    for n in (1000000, 1000003, 7):
        for world in (1, 2, 8):
            shards = [shard_range(n, r, world) for r in range(world)]
            assert shards[0][0] == 0 and sum(c for _, c in shards) == n
            for (o0, c0), (o1, _) in zip(shards, shards[1:]):
                assert o0 + c0 == o1


def _gather_worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pyprob_amd.parallel import gather_particles, shard_range
    n = 1003
    off, cnt = shard_range(n, rank, world)
    v, lw = gather_particles(torch.arange(off, off + cnt, dtype=torch.float32), -torch.arange(off, off + cnt, dtype=torch.float32), n)
    ok = torch.equal(v, torch.arange(n, dtype=torch.float32)) and torch.equal(lw, -torch.arange(n, dtype=torch.float32))
    bad = False
    try:
        gather_particles(torch.zeros(cnt + 1), torch.zeros(cnt + 1), n)
    except ValueError:
        bad = True
    torch.save(dict(ok=ok, bad=bad), out + '.%d' % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_particle_gather_concatenates_the_rank_shards(tmp_path):
    """IS shards: every rank ends up with all particles in shard order (ParallelModel's merge)."""
    world, port = 2, _free_port()
    out = str(tmp_path / 'g.pt')
    mp.spawn(_gather_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        d = torch.load(out + '.%d' % r)
        assert d['ok'] and d['bad']


# ---- the ENGINE's data-parallel step (buffer layout, reduced tail, skip logic) on two gloo ranks ---------------------
def _local_batches(meta, batch, rank):
    """Rank 0 trains on the 2-statement traces of the golden GUMM minibatch, rank 1 on the longer ones: the ranks touch
    DIFFERENT proposal heads / embeddings (the presence map has to be merged)."""
    lens = batch['trace_len']
    off = np.concatenate([[0], np.cumsum(lens)])
    pick = np.nonzero(lens == 2)[0][:16] if rank == 0 else np.nonzero(lens > 2)[0][:16]
    rows = np.concatenate([np.arange(off[b], off[b + 1]) for b in pick])
    return dict(trace_len=lens[pick], addr_idx=batch['addr_idx'][rows], values=batch['values'][rows].copy(),
                prior=batch['prior'][rows], obs=batch['obs'][pick].copy())


def _engine_worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oracle_ops
    from pyprob_amd.packed import PackedBatch
    meta, params, batch, loss, isr = load_golden('gumm')
    spec = spec_from_golden(meta, params)
    eng = oracle_ops.CpuBufferEngine(spec)
    eng._use_ops = True
    eng.load_state_dict(params)
    eng.world_size = world
    local = _local_batches(meta, batch, rank)
    ids = np.array([spec.address_id[meta['addresses'][i]] for i in local['addr_idx']])

    def packed(arr):
        return PackedBatch.from_ragged(arr['trace_len'], ids, arr['values'], arr['prior'], arr['obs'], len(spec.addresses)).to('cpu')
    rec = {}
    eng.train_step(packed(local), lr=1e-3)                                   # step 1: both ranks fine
    rec['p1'] = eng.params.clone()
    rec['steps1'] = eng.tensor_step.clone()
    rec['active1'] = eng.active.clone()
    rec['loss1'] = float(eng.loss_buf[0]) / world
    bad = {k: v.copy() for k, v in local.items()}
    if rank == 1:
        bad['obs'][0, 0] = np.nan                                           # step 2: rank 1's minibatch is broken
    eng.train_step(packed(bad), lr=1e-3)
    rec['p2'] = eng.params.clone()
    rec['steps2'] = eng.tensor_step.clone()
    rec['status2'] = float(eng.status_tail[0])
    rec['grads_clean2'] = bool((eng.grads == 0).all())
    eng.train_step(packed(local), lr=1e-3)                                   # step 3: training goes on, no NaN left behind
    rec['p3'] = eng.params.clone()
    rec['steps3'] = eng.tensor_step.clone()
    torch.save(rec, out + '.%d' % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_engine_data_parallel_step_merges_presence_and_skips_together(tmp_path):
    """ICEngine.train_step with world_size 2 (buffers on CPU, operators backed by the oracle): ONE all-reduce of
    [grads | presence | loss | non-finite flag]; ranks that touch different heads end with IDENTICAL parameters equal to
    Adam on the averaged gradient with the merged presence map (inference_network.py:296-333); a non-finite loss on one
    rank makes BOTH ranks skip that iteration (no parameter, no step count changes) and leaves no NaN behind."""
    world, port = 2, _free_port()
    out = str(tmp_path / 'eng.pt')
    mp.spawn(_engine_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = torch.load(out + '.0', weights_only=False), torch.load(out + '.1', weights_only=False)
    for k in ('p1', 'p2', 'p3', 'steps1', 'steps2', 'steps3', 'active1'):
        assert torch.equal(r0[k], r1[k]), k                                  # the ranks never diverge
    from oracle import ic_oracle as O
    meta, params, batch, loss, isr = load_golden('gumm')
    spec = spec_from_golden(meta, params)
    net = O.Net(params, meta['obs_names'], K=10)
    outs = [O.loss_and_grads(net, _local_batches(meta, batch, r), meta['addresses'], meta['dist_names']) for r in range(world)]
    assert abs(r0['loss1'] - 0.5 * (outs[0]['loss'] + outs[1]['loss'])) < 1e-5
    masks = []
    for r in range(world):
        lb = _local_batches(meta, batch, r)
        ids = np.array([spec.address_id[meta['addresses'][i]] for i in lb['addr_idx']])
        loff = np.concatenate([[0], np.cumsum(lb['trace_len'])])
        not_last = np.ones(len(ids), bool)
        not_last[loff[1:] - 1] = False
        masks.append(spec.active_mask(np.bincount(ids, minlength=len(spec.addresses)),
                                      np.bincount(ids[not_last], minlength=len(spec.addresses))))
    merged = np.maximum(masks[0], masks[1])
    assert (masks[0] != masks[1]).any()                                      # the ranks really touched different tensors
    assert np.array_equal(r0['active1'].numpy() > 0, merged > 0)
    names = list(spec.tensors.keys())
    p1 = r0['p1'].numpy()
    for i, n in enumerate(names):
        o_, shape = spec.tensors[n]
        cnt = int(np.prod(shape))
        got = p1[o_:o_ + cnt].reshape(shape)
        if merged[i] > 0:
            want = params[n].astype(np.float64).copy()
            g = 0.5 * (outs[0]['grads'][n] + outs[1]['grads'][n])
            O.adam_step(want, g, np.zeros_like(want), np.zeros_like(want), 1, 1e-3)
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6, err_msg=n)
            assert int(r0['steps1'][i]) == 1
        else:
            np.testing.assert_array_equal(got, params[n])
            assert int(r0['steps1'][i]) == 0
    # step 2: skipped everywhere
    assert r0['status2'] > 0 and r1['status2'] > 0
    assert torch.equal(r0['p2'], r0['p1']) and torch.equal(r0['steps2'], r0['steps1'])
    assert r0['grads_clean2'] and r1['grads_clean2']
    # step 3: finite, one more Adam step on the touched tensors
    assert torch.isfinite(r0['p3']).all() and not torch.equal(r0['p3'], r0['p2'])
    assert torch.equal(r0['steps3'], r0['steps1'] * 2)


def _skip_worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oracle_ops
    from helpers import packed_from_golden
    meta, params, batch, loss, isr = load_golden('gum')
    spec = spec_from_golden(meta, params)
    res = []
    for skip in (False, True):
        eng = oracle_ops.CpuBufferEngine(spec)
        eng._use_ops = True
        eng.load_state_dict(params)
        eng.world_size = world
        eng.skip_recurrent_weights(skip)
        b = {k: v[rank::world] if k != 'addr_idx' else v[rank::world] for k, v in batch.items()}
        pb = packed_from_golden(meta, b, spec).to('cpu')
        for _ in range(2):
            eng.train_step(pb, lr=1e-3)
        res.append((eng.params.clone(), eng.tensor_step.clone()))
    torch.save(dict(same_params=torch.equal(res[0][0], res[1][0]), same_steps=torch.equal(res[0][1], res[1][1]),
                    segments=len(eng.dp_skip)), out + '.%d' % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_recurrent_weight_range_can_be_left_out_of_the_allreduce(tmp_path):
    """Single-statement datasets (GUM): dL/dW_hh = 0 on every rank; reducing the flat buffer WITHOUT that range gives
    bit-identical parameters and step counts."""
    world, port = 2, _free_port()
    out = str(tmp_path / 'skip.pt')
    mp.spawn(_skip_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        d = torch.load(out + '.%d' % r)
        assert d['same_params'] and d['same_steps'] and d['segments'] == 1


def _agree_worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oracle_ops
    from helpers import packed_from_golden
    res = {}
    # (a) both ranks hold single-statement traces (GUM): every rank finds t_max == 1 in ITS data -> the range is skipped;
    # (b) rank 1 holds GUMM traces (several statements): its finding vetoes the skip on BOTH ranks
    for tag, cases in (('both_single', ('gum', 'gum')), ('one_ragged', ('gum', 'gumm'))):
        meta, params, batch, loss, isr = load_golden(cases[rank])
        spec = spec_from_golden(meta, params)
        eng = oracle_ops.CpuBufferEngine(spec)
        eng.world_size = world
        pb = packed_from_golden(meta, batch, spec).to('cpu')
        res[tag] = (eng.agree_skip_recurrent(pb.t_max == 1), len(eng.dp_skip), int(pb.t_max))
    torch.save(res, out + '.%d' % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_agree_on_the_recurrent_skip_from_their_own_data(tmp_path):
    """ICEngine.agree_skip_recurrent: the W_hh range leaves the all-reduce only when EVERY rank found single-statement traces
    in its own data (a MIN-all-reduce of the per-rank finding) - no caller assertion about the other ranks' data."""
    world, port = 2, _free_port()
    out = str(tmp_path / 'agree.pt')
    mp.spawn(_agree_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = (torch.load(out + '.%d' % r) for r in range(world))
    assert r0['both_single'][:2] == r1['both_single'][:2] == (True, 1)
    assert r0['one_ragged'][2] == 1 and r1['one_ragged'][2] > 1
    assert r0['one_ragged'][:2] == r1['one_ragged'][:2] == (False, 0)


def _larc_worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oracle_ops
    from pyprob_amd.packed import PackedBatch
    meta, params, batch, loss, isr = load_golden('gumm')
    spec = spec_from_golden(meta, params)
    local = _local_batches(meta, batch, rank)
    ids = np.array([spec.address_id[meta['addresses'][i]] for i in local['addr_idx']])
    pb = PackedBatch.from_ragged(local['trace_len'], ids, local['values'], local['prior'], local['obs'], len(spec.addresses)).to('cpu')
    rec = {}
    for kind, larc in (('sgd', False), ('sgd', True), ('adam', True)):
        eng = oracle_ops.CpuBufferEngine(spec)
        eng._use_ops = True
        eng.load_state_dict(params)
        eng.world_size = world
        eng.set_optimizer(kind, larc=larc, momentum=0.9)
        for _ in range(2):
            eng.train_step(pb, lr=0.05, weight_decay=1e-3)
        rec[(kind, larc)] = eng.params.clone()
    torch.save(rec, out + '.%d' % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_sgd_and_larc_under_data_parallelism(tmp_path):
    """Optimizer.SGD / SGD_LARC / ADAM_LARC with world_size 2: LARC's norms are those of the AVERAGED gradient (the
    reference divides by the world size before optimizer.step(), inference_network.py:324-325, 496), the ranks stay
    identical, and two steps equal the oracle's optimizers on the averaged gradients with the merged presence map."""
    world, port = 2, _free_port()
    out = str(tmp_path / 'larc.pt')
    mp.spawn(_larc_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = torch.load(out + '.0', weights_only=False), torch.load(out + '.1', weights_only=False)
    from oracle import ic_oracle as O
    meta, params, batch, loss, isr = load_golden('gumm')
    spec = spec_from_golden(meta, params)
    masks = []
    for r in range(world):
        lb = _local_batches(meta, batch, r)
        ids = np.array([spec.address_id[meta['addresses'][i]] for i in lb['addr_idx']])
        loff = np.concatenate([[0], np.cumsum(lb['trace_len'])])
        not_last = np.ones(len(ids), bool)
        not_last[loff[1:] - 1] = False
        masks.append(spec.active_mask(np.bincount(ids, minlength=len(spec.addresses)),
                                      np.bincount(ids[not_last], minlength=len(spec.addresses))))
    merged = np.maximum(masks[0], masks[1])
    names = list(spec.tensors.keys())
    for (kind, larc), got in r0.items():
        assert torch.equal(got, r1[(kind, larc)]), (kind, larc)
        P = {n: params[n].astype(np.float64).copy() for n in names}
        B = {n: np.zeros_like(P[n]) for n in names}
        M = {n: np.zeros_like(P[n]) for n in names}
        V = {n: np.zeros_like(P[n]) for n in names}
        for step in (1, 2):
            net = O.Net({n: P[n] for n in names}, meta['obs_names'], K=10)
            outs = [O.loss_and_grads(net, _local_batches(meta, batch, r), meta['addresses'], meta['dist_names']) for r in range(world)]
            for i, n in enumerate(names):
                if not merged[i] > 0:
                    continue
                g, decay = 0.5 * (outs[0]['grads'][n] + outs[1]['grads'][n]), 1e-3
                if larc:
                    g, decay = O.larc_scale(P[n], g, 0.05, decay), 0.0
                if kind == 'sgd':
                    O.sgd_step(P[n], g, B[n], 0.05, 0.9, True, decay)
                else:
                    O.adam_step(P[n], g, M[n], V[n], step, 0.05, weight_decay=decay)
        flat = got.numpy()
        for i, n in enumerate(names):
            o_, shape = spec.tensors[n]
            have = flat[o_:o_ + int(np.prod(shape))].reshape(shape)
            if merged[i] > 0:
                # (Adam's first steps are g / (|g| + eps): elements whose gradient and decay terms cancel are sensitive to the
                # fp32 storage of the buffers between the oracle-backed operators - compared on all but those)
                tol = dict(rtol=1e-4, atol=1e-5) if kind == 'sgd' else dict(rtol=5e-3, atol=2e-3)
                np.testing.assert_allclose(have, P[n], err_msg='%s %s %s' % (kind, larc, n), **tol)
            else:
                np.testing.assert_array_equal(have, params[n])


def _cpu_engine_factory(spec, device='cpu', seed=None):
    import oracle_ops
    eng = oracle_ops.CpuBufferEngine(spec, seed=seed)
    eng._use_ops = True
    return eng


def _gumm_dataset():
    from helpers import synthetic_gumm_arrays
    from pyprob_amd.dataset import PackedTraceDataset
    arrays, addresses = synthetic_gumm_arrays(768, seed=17, max_iter=3)
    table = [(a, 'Uniform', None) for a in addresses]
    return PackedTraceDataset.from_columns(['obs0', 'obs1'], [1, 1], arrays['trace_len'], table, arrays['addr_idx'],
                                           arrays['values'], arrays['prior'], arrays['obs'])


def _loop_worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import contextlib
    import io
    from models import GaussianWithUnknownMeanMarsaglia
    from pyprob_amd import nn as N
    from pyprob_amd.state import InferenceNetwork
    N.InferenceNetworkLSTM._engine_factory = staticmethod(_cpu_engine_factory)
    ds = _gumm_dataset()
    used = []
    device_batch = ds.device_batch

    def logged(ids, spec, device):
        used.append(np.asarray(ids).copy())
        return device_batch(ids, spec, device)
    ds.device_batch = logged
    model = GaussianWithUnknownMeanMarsaglia()
    with contextlib.redirect_stdout(io.StringIO()):
        model.learn_inference_network(num_traces=16 * 2 * 12, dataset=ds, inference_network=InferenceNetwork.LSTM,
                                      observe_embeddings={'obs0': {'dim': 8}, 'obs1': {'dim': 8}}, lstm_dim=16, batch_size=16,
                                      learning_rate_init=1e-3, weight_decay=0.0, distributed_backend='gloo',
                                      distributed_num_buckets=3, pre_generate_layers=True, device='cpu', seed=5,
                                      distributed_params_sync_every_iter=4)
    net = model._inference_network
    torch.save(dict(params=net._engine.params.clone(), used=used, iters=net._total_train_iterations,
                    traces=net._total_train_traces, hist=list(net._history_train_loss), lr=net._learning_rate_init,
                    steps=net._engine.tensor_step.clone()), out + '.%d' % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_training_loop_on_two_ranks_equals_steps_on_the_averaged_gradients(tmp_path):
    """InferenceNetworkLSTM.optimize (the mirror of inference_network.py:381-599) with world_size 2 on gloo, buffers on the
    host and operators backed by the oracle: bucketed sampler per rank, parameter broadcast every 4 iterations, one
    all-reduce per iteration, lr * sqrt(world) (:448), loss read-back per iteration. The ranks see disjoint minibatches, end
    with identical parameters and counters, and the parameters equal a single-process replay of the SAME minibatch pairs:
    gradients of both minibatches summed, presence maps merged, Adam on the average."""
    world, port = 2, _free_port()
    out = str(tmp_path / 'loop.pt')
    mp.spawn(_loop_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = torch.load(out + '.0', weights_only=False), torch.load(out + '.1', weights_only=False)
    assert torch.equal(r0['params'], r1['params']) and torch.equal(r0['steps'], r1['steps'])
    assert r0['iters'] == r1['iters'] == 12 and r0['traces'] == r1['traces'] == 16 * 2 * 12
    assert abs(r0['lr'] - 1e-3 * np.sqrt(2.0)) < 1e-12
    np.testing.assert_allclose(r0['hist'], r1['hist'], rtol=1e-6)           # the all-reduced mean loss
    assert len(r0['used']) == len(r1['used']) == 12
    for a, b in zip(r0['used'], r1['used']):
        assert len(a) == len(b) == 16 and not set(a.tolist()) & set(b.tolist())
    # single-process replay
    import oracle_ops  # noqa: F401
    from pyprob_amd.spec import NetSpec
    ds = _gumm_dataset()
    spec = NetSpec({'obs0': {'dim': 8, 'input_dim': 1}, 'obs1': {'dim': 8, 'input_dim': 1}}, lstm_dim=16)
    eng = _cpu_engine_factory(spec, seed=5)
    eng.add_addresses([a for a in ds.addresses])
    eng.force_allreduce = True      # (the presence map is rewritten by every loss call, as in a data-parallel run: this replay
    hist = []                       # overwrites it with the merged one)
    for a, b in zip(r0['used'], r1['used']):
        eng.loss(ds.device_batch(a, eng.spec, 'cpu'), backward=True)
        g = eng.grads_full.clone()
        eng.loss(ds.device_batch(b, eng.spec, 'cpu'), backward=True)
        g += eng.grads_full
        eng.grads_full.copy_(g)
        hist.append(float(eng.loss_buf[0]) / 2)
        eng.adam_step(r0['lr'], weight_decay=0.0, zero_grads=True, grad_scale=0.5)
    np.testing.assert_allclose(r0['hist'], hist, rtol=1e-5)
    d = (eng.params - r0['params']).abs().max().item()
    assert d < 1e-6, d


def _native_engine_factory(spec, device='cpu', seed=None):
    eng = _cpu_engine_factory(spec, device, seed)
    eng.native_dp = True        # (what nn.optimize sets after init_native_comm() succeeded on every rank)
    return eng


def _native_loop_worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.pop('PP_DP_NATIVE_LOOP', None)
    os.environ.pop('PP_PYTHON_LOOP', None)
    import contextlib
    import io
    from models import GaussianWithUnknownMeanMarsaglia
    from pyprob_amd import nn as N
    from pyprob_amd.state import InferenceNetwork
    N.InferenceNetworkLSTM._engine_factory = staticmethod(_native_engine_factory)
    ds = _gumm_dataset()
    used = []
    device_batch = ds.device_batch

    def logged(ids, spec, device):
        used.append(np.asarray(ids).copy())
        return device_batch(ids, spec, device)
    ds.device_batch = logged
    model = GaussianWithUnknownMeanMarsaglia()
    with contextlib.redirect_stdout(io.StringIO()):
        model.learn_inference_network(num_traces=16 * 2 * 12, dataset=ds, inference_network=InferenceNetwork.LSTM,
                                      observe_embeddings={'obs0': {'dim': 8}, 'obs1': {'dim': 8}}, lstm_dim=16, batch_size=16,
                                      learning_rate_init=1e-3, learning_rate_end=1e-5, learning_rate_scheduler_type='POLY2',
                                      num_traces_end=16 * 2 * 12, weight_decay=0.0, distributed_backend='gloo',
                                      distributed_num_buckets=3, pre_generate_layers=True, device='cpu', seed=5,
                                      distributed_params_sync_every_iter=5)
    net = model._inference_network
    torch.save(dict(params=net._engine.params.clone(), used=used, iters=net._total_train_iterations,
                    traces=net._total_train_traces, hist=list(net._history_train_loss), runs=list(net._engine.run_lengths),
                    steps=net._engine.tensor_step.clone(),
                    per_address=[a.total_train_iterations for a in net._engine.spec.addresses]), out + '.%d' % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_run_planner_under_data_parallelism_cuts_the_same_runs_on_every_rank(tmp_path):
    """The loop that hands RUNS of steps to one C call (pp_train_steps; nn.optimize's `while native`) on two gloo ranks, the C
    call restated on the host (oracle_ops.CpuBufferEngine.train_run: pack -> loss + backward -> one all-reduce -> Adam per
    step): both ranks cut the same runs (16-step chunks, a cut where a parameter broadcast is due - every 5 iterations -,
    the end), book the same all-reduced losses and counters, end with identical parameters - and everything equals a
    single-process replay of the same minibatch pairs with the POLY2 learning rate of the reference's trace count
    (inference_network.py:357-379, 448, 473-474, 486-531)."""
    world, port = 2, _free_port()
    out = str(tmp_path / 'native.pt')
    mp.spawn(_native_loop_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = torch.load(out + '.0', weights_only=False), torch.load(out + '.1', weights_only=False)
    assert r0['runs'] == r1['runs'] == [5, 5, 2]
    assert torch.equal(r0['params'], r1['params']) and torch.equal(r0['steps'], r1['steps'])
    assert r0['iters'] == r1['iters'] == 12 and r0['traces'] == r1['traces'] == 16 * 2 * 12
    np.testing.assert_allclose(r0['hist'], r1['hist'], rtol=1e-6)
    for a, b in zip(r0['used'], r1['used']):
        assert len(a) == len(b) == 16 and not set(a.tolist()) & set(b.tolist())
    import oracle_ops  # noqa: F401
    from pyprob_amd.spec import NetSpec
    ds = _gumm_dataset()
    spec = NetSpec({'obs0': {'dim': 8, 'input_dim': 1}, 'obs1': {'dim': 8, 'input_dim': 1}}, lstm_dim=16)
    eng = _cpu_engine_factory(spec, seed=5)
    eng.add_addresses([a for a in ds.addresses])
    eng.force_allreduce = True
    hist, seen, end = [], 0, 16 * 2 * 12
    lr0, lr1 = 1e-3 * np.sqrt(2.0), 1e-5
    for a, b in zip(r0['used'], r1['used']):
        eng.loss(ds.device_batch(a, eng.spec, 'cpu'), backward=True)
        g = eng.grads_full.clone()
        eng.loss(ds.device_batch(b, eng.spec, 'cpu'), backward=True)
        g += eng.grads_full
        eng.grads_full.copy_(g)
        hist.append(float(eng.loss_buf[0]) / 2)
        lr = (lr0 - lr1) * max(0.0, 1.0 - seen / end) ** 2 + lr1
        eng.adam_step(lr, weight_decay=0.0, zero_grads=True, grad_scale=0.5)
        seen += 32
    np.testing.assert_allclose(r0['hist'], hist, rtol=1e-5)
    d = (eng.params - r0['params']).abs().max().item()
    assert d < 1e-6, d
    # (per-address iteration counters count a rank's OWN minibatches, like the reference's proposal layers)
    assert max(r0['per_address']) == max(r1['per_address']) == 12


def _gum_model_on_host():
    import oracle_ops  # noqa: F401  (CPU kernels of the operators)
    from is_helpers import network_from_golden
    from models import GaussianWithUnknownMean
    net, meta, params, isr = network_from_golden('gum')
    model = GaussianWithUnknownMean()
    model._inference_network = net
    return model


def _posterior_worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    model = _gum_model_on_host()
    post = model.posterior_results_distributed(101, observe={'obs0': 8.0, 'obs1': 9.0}, seed=3)
    torch.save(dict(values=post._all_values.clone() if hasattr(post, '_all_values') else torch.as_tensor(post.values_numpy()),
                    lw=torch.as_tensor(np.asarray(post.log_weights)), mean=float(post.mean), ess=float(post.effective_sample_size),
                    stats=dict(post.device_stats)), out + '.%d' % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_distributed_posterior_on_two_ranks_is_the_concatenation_of_the_shards(tmp_path):
    """Model.posterior_results_distributed (ParallelModel's sharding, pyprob/model.py:339-406) on two gloo ranks with the golden
    GUM network on the host: shards of 51 + 50 particles with their own counter offsets, ONE all-gather; every rank holds
    all 101 particles in shard order, equal to the two shards computed in one process, with the same statistics."""
    world, port = 2, _free_port()
    out = str(tmp_path / 'post.pt')
    mp.spawn(_posterior_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = torch.load(out + '.0', weights_only=False), torch.load(out + '.1', weights_only=False)
    assert torch.equal(r0['values'], r1['values']) and torch.equal(r0['lw'], r1['lw'])
    assert r0['mean'] == r1['mean'] and r0['ess'] == r1['ess'] and r0['values'].numel() == 101
    from pyprob_amd.parallel import shard_range
    model = _gum_model_on_host()
    vals, lws = [], []
    for r in range(world):
        off, cnt = shard_range(101, r, world)
        local = model._traces_lockstep(cnt, {'obs0': 8.0, 'obs1': 9.0}, seed=3, offset=off)
        vals.append(local._all_values)
        lws.append(local._all_log_weights)
    assert [v.numel() for v in vals] == [51, 50]
    assert torch.equal(torch.cat(vals).cpu(), r0['values'].cpu())
    np.testing.assert_allclose(torch.cat(lws).cpu().numpy(), r0['lw'].numpy(), rtol=0, atol=0)
    assert abs(r0['stats']['ess'] - r0['ess']) < 1e-6 * r0['ess'] and r0['stats']['count'] == 101
    assert abs(r0['mean'] - 7.25) < 1.5        # (the golden network is only briefly trained)


# ---- the early-bucket order of the gradient exchange (ICEngine.enable_dp_overlap; csrc/dp.hip on the device) --------------------
def _overlap_worker(rank, world, port, out):
    os.environ['PP_DP_OVERLAP'] = '1'          # (off by default, pyprob_amd/engine.py enable_dp_overlap)
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oracle_ops
    from pyprob_amd.packed import PackedBatch
    meta, params, batch, loss, isr = load_golden('gumm')
    spec = spec_from_golden(meta, params)
    local = _local_batches(meta, batch, rank)
    ids = np.array([spec.address_id[meta['addresses'][i]] for i in local['addr_idx']])
    pb = PackedBatch.from_ragged(local['trace_len'], ids, local['values'], local['prior'], local['obs'], len(spec.addresses)).to('cpu')
    rec = {}
    for mode in ('plain', 'overlap', 'skip', 'overlap_skip'):
        eng = oracle_ops.CpuBufferEngine(spec)
        eng._use_ops = True
        eng.load_state_dict(params)
        eng.world_size = world
        if mode.endswith('skip'):
            eng.skip_recurrent_weights(True)      # (only the RANGES are under test: W_hh's gradient is not zero on this data -
            #                                        the yardstick is the same skip without the early ranges)
        rec[mode + '_ranges'] = eng.enable_dp_overlap(mode.startswith('overlap'))
        for k in range(3):
            eng.train_step(pb, lr=1e-3)
        rec[mode] = eng.params.clone()
        rec[mode + '_steps'] = eng.tensor_step.clone()
        rec[mode + '_loss'] = float(eng.loss_buf[0]) / world
    rec['whh'] = spec.tensors['_layers_lstm.weight_hh_l0']
    torch.save(rec, out + '.%d' % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_early_bucket_order_changes_no_bit_of_the_exchange(tmp_path):
    """The bucketed order (the first LSTM layer's gradient ranges as asynchronous collectives first, the rest after, the early
    ones waited for last - what csrc/dp.hip does with a side stream) on two gloo ranks: the same parameters, bit for bit, as
    the single all-reduce; with W_hh skipped the early ranges are the two pieces around it."""
    world, port = 2, _free_port()
    out = str(tmp_path / 'ov')
    mp.spawn(_overlap_worker, args=(world, port, out), nprocs=world, join=True)
    r = [torch.load(out + '.%d' % k, weights_only=False) for k in range(world)]
    for k in range(world):
        assert r[k]['plain_ranges'] == []
        (o0, c0), = r[k]['overlap_ranges']                        # W_ih | W_hh | b_ih | b_hh are adjacent: one range
        off, shape = r[k]['whh']
        assert o0 < off and off + int(np.prod(shape)) <= o0 + c0
        assert len(r[k]['overlap_skip_ranges']) == 2 and r[k]['overlap_skip_ranges'][0][0] == o0
        assert r[k]['overlap_skip_ranges'][0][0] + r[k]['overlap_skip_ranges'][0][1] == off      # ... ends where W_hh begins
        assert torch.equal(r[k]['overlap'], r[k]['plain']) and torch.equal(r[k]['overlap_steps'], r[k]['plain_steps'])
        assert r[k]['overlap_loss'] == r[k]['plain_loss']
        assert torch.equal(r[k]['overlap'], r[0]['overlap'])       # the ranks hold identical parameters
        assert r[k]['skip_ranges'] == [] and torch.equal(r[k]['overlap_skip'], r[k]['skip'])
