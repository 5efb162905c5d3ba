"""GPU parity of the whole hot path (InferenceNetworkLSTM._loss + backward, Adam, importance sampling) through the
C ABI, against (a) the golden vectors recorded from the reference and (b) the numpy oracle at benchmark sizes, plus
size-independent properties. Tolerance: north_star asks 1e-4 relative on log-weights; most checks are tighter."""

import numpy as np
import pytest

from conftest import load_golden
from helpers import (engine_from_golden, grad_check, packed_from_golden, rel_err, synthetic_gum_arrays,
                     synthetic_gumm_arrays)
from oracle import ic_oracle as O

pytestmark = pytest.mark.gpu
# gradient bars against the float64 oracle: <= 10x the largest error measured on MI355X (profiles/r04_grad_errors.jsonl:
# 8.6e-7 relative on the single-statement shapes; 5.0e-9 ABSOLUTE on the ragged ones, whose gradients are ~1e-6)
GRAD_BAR = 1e-5
GRAD_ABS_RAGGED = 5e-8
torch = pytest.importorskip('torch')


def _unpack_lp(pb, lp_rows, batch):
    """per-row log_prob (packed row order) -> trace-major order of the golden arrays"""
    out = np.empty(pb.n_rows, np.float64)
    out[pb.src_row] = lp_rows
    return out


def test_golden_loss_logprob_and_gradients(golden):
    case, meta, params, batch, loss, isr = golden
    eng = engine_from_golden(meta, params)
    pb = packed_from_golden(meta, batch, eng.spec).to(eng.device)
    l, lp = eng.loss(pb, backward=True, keep_lp=True)
    torch.cuda.synchronize()
    assert int(eng.status_buf[0].item()) == 0
    ref_loss = float(loss['loss'])
    assert abs(float(l.item()) - ref_loss) <= 1e-5 * abs(ref_loss), (float(l.item()), ref_loss)
    # per-row log_prob against the reference's Mixture/Categorical.log_prob outputs
    lp_tm = _unpack_lp(pb, lp.cpu().numpy(), batch)
    off = np.concatenate([[0], np.cumsum(batch['trace_len'])])
    for k, (si, t) in enumerate(meta['lp_index']):
        rows = off[np.array(meta['sub_batches'][si])] + t
        ref = loss['lp_%d_%d' % (si, t)]
        if ref.size == len(rows) ** 2 and len(rows) > 1:   # Bernoulli proposal: the reference's [n, n] broadcast matrix
            ref = ref.reshape(len(rows), len(rows)).sum(1)
        np.testing.assert_allclose(lp_tm[rows], ref, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(ref).max()))
    # every parameter gradient against loss.backward() of the reference
    # (the reference's gradients are fp32 themselves: the bar is 1e-5 of a tensor's largest element + the 5e-8 absolute fp32
    # summation floor of helpers.grad_check - measured against these goldens on MI355X: profiles/r06_grad_errors.jsonl)
    g = eng.grad_dict()
    for i, n in enumerate(meta['param_names']):
        ref = loss['g%d' % i]
        if not meta['has_grad'][i]:
            assert np.all(g[n] == 0), n
            continue
        grad_check('golden_%s/%s' % (case, n), g[n], ref, 1e-5, 5e-8)
    # presence map = which tensors had grad != None in the reference
    act = eng.presence().cpu().numpy()
    names = list(eng.spec.tensors.keys())
    ref_has = dict(zip(meta['param_names'], meta['has_grad']))
    assert [bool(a) for a in act] == [bool(ref_has[n]) for n in names]


def _oracle_run(spec, params, arrays, addresses, dist_names, want_grads=True):
    net = O.Net(params, [o[0] for o in spec.obs], K=spec.K)
    return O.loss_and_grads(net, arrays, addresses, dist_names, want_grads=want_grads)


def _fresh_engine(lstm_dim, addresses, dist, seed=0):
    from pyprob_amd.engine import ICEngine
    from pyprob_amd.spec import NetSpec
    spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=lstm_dim)
    for a in addresses:
        spec.add_address(a, dist)
    return ICEngine(spec, seed=seed)


def _packed(arrays, spec):
    from pyprob_amd.packed import PackedBatch
    return PackedBatch.from_ragged(arrays['trace_len'], arrays['addr_idx'], arrays['values'], arrays['prior'],
                                   arrays['obs'], len(spec.addresses))


@pytest.mark.parametrize('H', [512])
def test_benchmark_size_gum_against_oracle(H):
    """config 2 shape: GUM, batch 1024, LSTM hidden 512 (1 643 583 parameters)."""
    eng = _fresh_engine(H, ['mu'], 'Normal')
    assert eng.spec.num_parameters() == 1643583
    arrays = synthetic_gum_arrays(1024, seed=3)
    pb = _packed(arrays, eng.spec).to(eng.device)
    l, lp = eng.loss(pb, backward=True, keep_lp=True)
    torch.cuda.synchronize()
    params = {k: v.numpy() for k, v in eng.state_dict().items()}
    out = _oracle_run(eng.spec, params, arrays, ['mu'], ['Normal'])
    assert abs(float(l.item()) - out['loss']) <= 2e-5 * abs(out['loss'])
    lp_tm = _unpack_lp(pb, lp.cpu().numpy(), arrays)
    np.testing.assert_allclose(lp_tm, out['lp'][0], rtol=1e-4, atol=1e-4)
    g = eng.grad_dict()
    for n in eng.spec.tensors:
        if np.abs(out['grads'][n]).max() >= 1e-7:
            grad_check('gum_h%d_b1024/%s' % (H, n), g[n], out['grads'][n], GRAD_BAR)


@pytest.mark.parametrize('B', [300, 1024])
def test_hidden_1024_against_oracle(B):
    """config 5 network (LSTM hidden 1024, 5 636 415 parameters; head hidden 527 -> 3 float4 groups per lane), at an odd
    batch size and at the per-rank shape of config 5 (batch 1024)."""
    eng = _fresh_engine(1024, ['mu'], 'Normal')
    assert eng.spec.num_parameters() == 5636415
    arrays = synthetic_gum_arrays(B, seed=6)
    pb = _packed(arrays, eng.spec).to(eng.device)
    l = eng.loss(pb, backward=True)
    torch.cuda.synchronize()
    params = {k: v.numpy() for k, v in eng.state_dict().items()}
    out = _oracle_run(eng.spec, params, arrays, ['mu'], ['Normal'])
    assert abs(float(l.item()) - out['loss']) <= 2e-5 * abs(out['loss'])
    g = eng.grad_dict()
    for n in eng.spec.tensors:
        if np.abs(out['grads'][n]).max() >= 1e-7:
            grad_check('gum_h1024_b%d/%s' % (B, n), g[n], out['grads'][n], GRAD_BAR)


def test_benchmark_size_gumm_ragged_against_oracle():
    """config 3 shape: GUMM (variable-length traces, one head per address), batch 1024, hidden 512."""
    arrays, addresses = synthetic_gumm_arrays(1024, seed=4, max_iter=6)
    eng = _fresh_engine(512, addresses, 'Uniform')
    pb = _packed(arrays, eng.spec).to(eng.device)
    l, lp = eng.loss(pb, backward=True, keep_lp=True)
    torch.cuda.synchronize()
    params = {k: v.numpy() for k, v in eng.state_dict().items()}
    out = _oracle_run(eng.spec, params, arrays, addresses, ['Uniform'] * len(addresses))
    assert abs(float(l.item()) - out['loss']) <= 2e-5 * abs(out['loss'])
    g = eng.grad_dict()
    for n in eng.spec.tensors:
        if np.abs(out['grads'][n]).max() > 1e-7:
            grad_check('gumm_ragged_h512_b1024/%s' % n, g[n], out['grads'][n], GRAD_BAR, GRAD_ABS_RAGGED)


def test_feedforward_network_benchmark_size_against_oracle():
    """InferenceNetworkFeedForward (inference_network_feedforward.py:68-98) at batch 1024 on ragged GUMM traces: the
    gradient of the observe embedding sums over each trace's time steps, one head per address."""
    from pyprob_amd.engine import ICEngine
    from pyprob_amd.spec import NetSpec
    arrays, addresses = synthetic_gumm_arrays(1024, seed=8, max_iter=6)
    spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, network='feedforward')
    for a in addresses:
        spec.add_address(a, 'Uniform')
    assert spec.lstm_dim == 0 and not any(n.startswith('_layers_lstm') for n in spec.tensors)
    eng = ICEngine(spec, seed=2)
    pb = _packed(arrays, eng.spec).to(eng.device)
    l, lp = eng.loss(pb, backward=True, keep_lp=True)
    fwd = eng.loss(pb).clone()                       # forward-only entry point: same loss
    torch.cuda.synchronize()
    assert int(eng.status_buf[0].item()) == 0
    params = {k: v.numpy() for k, v in eng.state_dict().items()}
    net = O.Net(params, [o[0] for o in spec.obs], K=spec.K)
    out = O.loss_and_grads_feedforward(net, arrays, addresses, ['Uniform'] * len(addresses))
    assert abs(float(l.item()) - out['loss']) <= 2e-5 * abs(out['loss'])
    assert abs(float(fwd.item()) - out['loss']) <= 2e-5 * abs(out['loss'])
    g = eng.grad_dict()
    for n in eng.spec.tensors:
        if np.abs(out['grads'][n]).max() > 1e-7:
            grad_check('ff_ragged_b1024/%s' % n, g[n], out['grads'][n], GRAD_BAR, GRAD_ABS_RAGGED)
    # a few Adam steps reduce the loss
    first = float(l.item())
    for _ in range(30):
        last = eng.train_step(pb, 1e-3)
    assert float(last.item()) < first


def test_resident_loop_matches_the_per_step_calls():
    """pp_train_resident (a run of steps over minibatches already in HBM, one C call) against ICEngine.train_step per
    minibatch: same losses, same parameters, same Adam step counts."""
    from pyprob_amd.packed import PackedBatch
    a, b = _fresh_engine(64, ['mu'], 'Normal', seed=4), _fresh_engine(64, ['mu'], 'Normal', seed=4)
    batches_a, batches_b = [], []
    for k, n in enumerate((256, 300, 128, 200, 256, 512, 128)):
        arr = synthetic_gum_arrays(n, seed=30 + k)
        for eng, lst in ((a, batches_a), (b, batches_b)):
            lst.append(PackedBatch.from_ragged(arr['trace_len'], np.zeros(n, np.int64), arr['values'], arr['prior'], arr['obs'],
                                               1).to(eng.device))
    lrs = [1e-3 * (1 + 0.1 * k) for k in range(len(batches_a))]
    ref = [float(a.train_step(pb, lr, weight_decay=1e-5).item()) for pb, lr in zip(batches_a, lrs)]
    losses, status = b.train_resident(batches_b[:3], lrs[:3], weight_decay=1e-5)
    got = losses.cpu().numpy().tolist()
    losses, status2 = b.train_resident(batches_b[3:], lrs[3:], weight_decay=1e-5)
    got += losses.cpu().numpy().tolist()
    assert not status.cpu().numpy().any() and not status2.cpu().numpy().any()
    np.testing.assert_allclose(got, ref, rtol=2e-5)
    # (Adam normalises the gradient: an element whose gradient is at round-off level may move by +-lr in either run)
    # per element: Adam normalises the gradient, so an element whose gradient is at round-off level may move by +-lr in either
    # run and step (a sign flip of m / sqrt(v)): no element may differ by more than the sum of the learning rates both ways, and
    # all but a handful of such elements must agree to fp32 round-off of the seven updates
    pa, pb_ = a.params.cpu().numpy().astype(np.float64), b.params.cpu().numpy().astype(np.float64)
    diff = np.abs(pa - pb_)
    assert diff.max() <= 2.0 * sum(lrs) * 1.001, diff.max()
    loose = diff > 2e-6 + 2e-5 * np.abs(pa)
    assert loose.mean() < 2e-3, (loose.mean(), diff.max())
    assert torch.equal(b.tensor_step.cpu(), a.tensor_step.cpu())


def test_native_training_loop_matches_the_per_step_calls():
    """pp_train_steps (pack -> upload -> loss + backward -> Adam for a run of minibatches in one C call) against the same
    minibatches stepped one C call at a time: same losses, same parameters, same per-address iteration counters; a
    non-finite minibatch is flagged and leaves the parameters alone."""
    from pyprob_amd.dataset import PackedTraceDataset
    arrays, addresses = synthetic_gumm_arrays(6000, seed=12, max_iter=5)
    table = [(a, 'Uniform', None) for a in addresses]
    ds = PackedTraceDataset.from_columns(['obs0', 'obs1'], [1, 1], arrays['trace_len'], table, arrays['addr_idx'],
                                         arrays['values'], arrays['prior'], arrays['obs'])
    rng = np.random.default_rng(5)
    steps = [rng.choice(6000, size=n, replace=False) for n in (256, 256, 300, 17, 256, 1, 256, 500, 256, 256, 256, 64)]
    lrs = [1e-3 * (1 + 0.1 * k) for k in range(len(steps))]
    a, b = _fresh_engine(64, addresses, 'Uniform', seed=3), _fresh_engine(64, addresses, 'Uniform', seed=3)
    ref_losses = []
    for ids, lr in zip(steps, lrs):
        pb = ds.device_batch(ids, a.spec, a.device)
        for info_id, n in enumerate(pb.cur_counts):
            a.spec.addresses[info_id].total_train_iterations += int(n > 0)
        ref_losses.append(float(a.loss(pb, backward=True).item()))
        a.adam_step(lr, weight_decay=1e-5, zero_grads=True)
    losses, status = b.train_run(ds, steps[:5], lrs[:5], weight_decay=1e-5)
    got = losses.cpu().numpy().tolist()
    losses, status2 = b.train_run(ds, steps[5:], lrs[5:], weight_decay=1e-5)       # a second run continues the first
    got += losses.cpu().numpy().tolist()
    assert not status.cpu().numpy().any() and not status2.cpu().numpy().any()
    np.testing.assert_allclose(got, ref_losses, rtol=2e-5)
    # (Adam normalises the gradient: an element whose gradient is at round-off level may move by +-lr in either run)
    for n in a.spec.tensors:
        d = np.abs(b.tensor(n).cpu().numpy() - a.tensor(n).cpu().numpy())
        assert d.mean() < 2e-5 and d.max() < 4e-3, (n, d.mean(), d.max())
    assert [i.total_train_iterations for i in b.spec.addresses] == [i.total_train_iterations for i in a.spec.addresses]
    assert torch.equal(b.tensor_step.cpu(), a.tensor_step.cpu())
    # a minibatch with a non-finite observation (torch.relu keeps the NaN, so does the embedding here): flagged,
    # parameters untouched, the run goes on
    bad = dict(arrays)
    bad['obs'] = arrays['obs'].copy()
    bad['obs'][10] = np.nan
    ds2 = PackedTraceDataset.from_columns(['obs0', 'obs1'], [1, 1], bad['trace_len'], table, bad['addr_idx'], bad['values'],
                                          bad['prior'], bad['obs'])
    victim = int(np.nonzero(np.isnan(ds2.gather(np.arange(6000))[4]).any(1))[0][0])
    before = b.params.clone()
    losses, status = b.train_run(ds2, [np.array([victim, 1, 2, 3])], [1e-3])
    assert int(status.cpu()[0]) != 0 and torch.equal(b.params, before)
    losses, status = b.train_run(ds2, [np.arange(1, 200) if victim == 0 else np.delete(np.arange(200), victim)], [1e-3])
    assert int(status.cpu()[0]) == 0 and not torch.equal(b.params, before)


def test_loss_is_permutation_invariant_and_additive():
    """Size-independent properties: the loss does not depend on trace order, and the loss of a union of two
    batches is the size-weighted mean of their losses."""
    arrays, addresses = synthetic_gumm_arrays(600, seed=11, max_iter=5)
    eng = _fresh_engine(64, addresses, 'Uniform', seed=5)
    off = np.concatenate([[0], np.cumsum(arrays['trace_len'])])

    def subset(idx):
        rows = np.concatenate([np.arange(off[b], off[b + 1]) for b in idx])
        return dict(trace_len=arrays['trace_len'][idx], addr_idx=arrays['addr_idx'][rows], values=arrays['values'][rows],
                    prior=arrays['prior'][rows], obs=arrays['obs'][idx])

    full = float(eng.loss(_packed(arrays, eng.spec).to(eng.device)).item())
    perm = np.random.default_rng(0).permutation(600)
    shuffled = float(eng.loss(_packed(subset(perm), eng.spec).to(eng.device)).item())
    assert abs(full - shuffled) <= 2e-6 * abs(full)
    a, b = perm[:250], perm[250:]
    la = float(eng.loss(_packed(subset(a), eng.spec).to(eng.device)).item())
    lb = float(eng.loss(_packed(subset(b), eng.spec).to(eng.device)).item())
    assert abs(full - (250 * la + 350 * lb) / 600) <= 3e-6 * abs(full)


@pytest.mark.parametrize('case,B,H,parts', [('gum', 8192, 512, 8), ('gumm', 4096, 256, 4)])
def test_large_minibatch_equals_the_mean_of_its_parts(case, B, H, parts):
    """Beyond the benchmark's batch size (288 GB of HBM invite larger minibatches per GPU): the loss and EVERY gradient of a
    minibatch of B traces equal the size-weighted mean over its `parts` slices of B / parts traces - slices of the size the
    oracle comparisons above pin. Size-independent property; 1e-5 on the loss, 2e-4 of each tensor's largest gradient."""
    if case == 'gum':
        arrays, addresses, dist = synthetic_gum_arrays(B, seed=31), ['mu'], 'Normal'
        arrays['addr_idx'] = np.zeros(B, np.int32)
    else:
        arrays, addresses = synthetic_gumm_arrays(B, seed=32, max_iter=5)
        dist = 'Uniform'
    eng = _fresh_engine(H, addresses, dist, seed=2)
    off = np.concatenate([[0], np.cumsum(arrays['trace_len'])])

    def piece(b0, b1):
        r0, r1 = off[b0], off[b1]
        return dict(trace_len=arrays['trace_len'][b0:b1], addr_idx=arrays['addr_idx'][r0:r1], values=arrays['values'][r0:r1],
                    prior=arrays['prior'][r0:r1], obs=arrays['obs'][b0:b1])

    full = eng.loss(_packed(arrays, eng.spec).to(eng.device), backward=True)
    torch.cuda.synchronize()
    assert int(eng.status_buf[0].item()) == 0
    l_full, g_full = float(full.item()), eng.grads.double().clone()
    step = B // parts
    l_sum, g_sum = 0.0, torch.zeros_like(g_full)
    for k in range(parts):
        lk = eng.loss(_packed(piece(k * step, (k + 1) * step), eng.spec).to(eng.device), backward=True)
        torch.cuda.synchronize()
        l_sum += float(lk.item()) / parts
        g_sum += eng.grads.double() / parts
    assert abs(l_full - l_sum) <= 1e-5 * abs(l_sum), (l_full, l_sum)
    for n, (o, shape) in eng.spec.tensors.items():
        cnt = int(np.prod(shape))
        a, b = g_full[o:o + cnt], g_sum[o:o + cnt]
        scale = float(b.abs().max().item())
        assert float((a - b).abs().max().item()) <= 2e-4 * scale + 1e-9, (n, float((a - b).abs().max().item()), scale)


def test_adam_matches_torch_semantics():
    """Optimizer steps against the oracle's Adam fed with the SAME gradients, including tensors without gradient
    (skipped, step count not advanced) -- torch.optim.Adam as the reference configures it (inference_network.py:348)."""
    meta, params, batch, loss, isr = load_golden('gumm')
    eng = engine_from_golden(meta, params)
    pb = packed_from_golden(meta, batch, eng.spec).to(eng.device)
    P = {k: v.astype(np.float64).copy() for k, v in params.items()}
    M = {k: np.zeros_like(v) for k, v in P.items()}
    V = {k: np.zeros_like(v) for k, v in P.items()}
    names = list(eng.spec.tensors.keys())
    act = eng.spec.active_mask(pb.cur_counts, pb.prev_counts)
    for step in range(1, 5):
        eng.loss(pb, backward=True)
        g = eng.grad_dict()
        eng.adam_step(1e-3, weight_decay=1e-5 if step == 4 else 0.0)
        torch.cuda.synchronize()
        for i, n in enumerate(names):
            if act[i]:
                O.adam_step(P[n], g[n].astype(np.float64), M[n], V[n], step, 1e-3, weight_decay=1e-5 if step == 4 else 0.0)
        sd = eng.state_dict()
        worst = max(rel_err(sd[n].numpy(), P[n]) for n in names)
        assert worst < 2e-6, (step, worst)
    steps = eng.tensor_step.cpu().numpy()
    assert np.all(steps[act > 0] == 4) and np.all(steps[act == 0] == 0)
    for i, n in enumerate(names):       # untouched tensors are bit-identical
        if not act[i]:
            np.testing.assert_array_equal(sd[n].numpy(), params[n])


def test_adam_clears_consumed_gradients_and_train_step_skips_the_memset():
    """PP_ADAM_ZERO_GRADS: the optimizer pass performs the next step's zero_grad (inference_network.py:486). Two engines
    from the same parameters - one zeroing in pp_ic_loss every step, one relying on Adam's clearing - stay identical."""
    meta, params, batch, loss, isr = load_golden('gumm')
    a, b = engine_from_golden(meta, params), engine_from_golden(meta, params)
    pa, pb_ = (packed_from_golden(meta, batch, e.spec).to(e.device) for e in (a, b))
    for step in range(3):
        a.loss(pa, backward=True)                  # explicit zero_grad inside pp_ic_loss
        a.adam_step(1e-3)
        b.train_step(pb_, lr=1e-3)                 # Adam clears, the next loss runs without PP_LOSS_ZERO_GRADS
        assert b._grads_clean
        assert float(b.grads.abs().max().item()) == 0.0
        assert int(b.arrived.view(-1, 1056)[:, :1025].abs().max().item()) == 0     # arrival counters reset themselves
    sa, sb = a.state_dict(), b.state_dict()
    # float atomics make the last bits of a gradient run-dependent, and the first Adam steps move every weight by
    # ~lr * sign(g): a near-zero gradient whose sign flips costs 2 lr. Compare with the tolerance of one such flip
    # (bitwise equality of repeated runs is what PP_DETERMINISTIC=1 gives: tests/test_gpu_holes.py)
    # (three steps of lr 1e-3: up to three flips of one element = 6e-3 absolute; 1e-3 relative was within reach of two)
    assert max(rel_err(sa[n].numpy(), sb[n].numpy()) for n in sa) < 4e-3
    assert torch.equal(a.tensor_step, b.tensor_step)
    b.params.copy_(a.params)                       # (what follows is about stale gradients, not about the flips above)
    g = b.loss(pb_, backward=True)                 # flag consumed: this call must NOT see stale gradients
    ga = a.loss(pa, backward=True)
    assert abs(float(g.item()) - float(ga.item())) < 1e-5 * abs(float(ga.item()))
    gb, gaa = b.grad_dict(), a.grad_dict()
    assert max(rel_err(gb[n], gaa[n]) for n in gb) < 1e-3


def test_train_steps_track_the_oracle():
    """Three full train steps (loss -> backward -> Adam) against the oracle doing the same in float64: the loss
    trajectory agrees; parameters agree up to Adam's sign-sensitivity for near-zero gradients."""
    meta, params, batch, loss, isr = load_golden('gum')
    eng = engine_from_golden(meta, params)
    pb = packed_from_golden(meta, batch, eng.spec).to(eng.device)
    P = {k: v.astype(np.float64).copy() for k, v in params.items()}
    M = {k: np.zeros_like(v) for k, v in P.items()}
    V = {k: np.zeros_like(v) for k, v in P.items()}
    act = eng.spec.active_mask(pb.cur_counts, pb.prev_counts)
    names = list(eng.spec.tensors.keys())
    for step in range(1, 4):
        l = float(eng.train_step(pb, lr=1e-3).item())
        out = O.loss_and_grads(O.Net(P, meta['obs_names'], K=10), batch, meta['addresses'], meta['dist_names'])
        assert abs(l - out['loss']) < 1e-4 * abs(out['loss']), (step, l, out['loss'])
        for i, n in enumerate(names):
            if act[i]:
                O.adam_step(P[n], out['grads'][n], M[n], V[n], step, 1e-3)
    sd = eng.state_dict()
    assert max(rel_err(sd[n].numpy(), P[n]) for n in names) < 5e-3


def test_training_reduces_loss_on_gum():
    eng = _fresh_engine(64, ['mu'], 'Normal', seed=1)
    arrays = synthetic_gum_arrays(2048, seed=8)
    pb = _packed(arrays, eng.spec).to(eng.device)
    first = float(eng.loss(pb).item())
    for _ in range(150):
        eng.train_step(pb, lr=1e-3)
    last = float(eng.loss(pb).item())
    assert np.isfinite(last) and last < first - 0.3, (first, last)


def test_polymorph_growth_keeps_existing_parameters():
    eng = _fresh_engine(64, ['a0'], 'Uniform', seed=2)
    before = eng.state_dict()
    assert eng.add_addresses([('a1', 'Uniform', None), ('c', 'Categorical', 5)])
    assert not eng.add_addresses([('a1', 'Uniform', None)])
    after = eng.state_dict()
    for n in before:
        np.testing.assert_array_equal(before[n].numpy(), after[n].numpy())
    assert eng.spec.num_parameters() > sum(v.numel() for v in before.values())
    assert int(eng.tensor_step.sum().item()) == 0


# ---- importance sampling ------------------------------------------------------------------------------------
def _is_runner(eng):
    from pyprob_amd.is_engine import ISRunner
    return ISRunner(eng)


def test_is_rescoring_matches_reference_records(golden):
    """Re-score the reference-sampled particles: proposal log_prob, prior log_prob and the per-trace
    log-importance-weight of state.py:211-217 / trace.py:123-125, within 1e-4 relative."""
    case, meta, params, batch, loss, isr = golden
    eng = engine_from_golden(meta, params)
    run = _is_runner(eng)
    run.init(isr['observe'])
    addresses = meta['is_addresses']
    off = np.concatenate([[0], np.cumsum(isr['trace_len'])])
    lw_all = np.zeros(len(isr['trace_len']))
    q_all = np.zeros(len(isr['value']))
    for b in range(len(isr['trace_len'])):
        run.begin(1)
        prev = None
        for t in range(int(isr['trace_len'][b])):
            r = off[b] + t
            a = eng.spec.address_id[addresses[isr['addr'][r]]]
            info = eng.spec.addresses[a]
            v = torch.tensor([isr['value'][r]], dtype=torch.float32, device=eng.device)
            pr = isr['prior'][r, :2] if info.dist_name != 'Poisson' else np.array([0.0, 40.0], np.float32)   # fixed interval
            pr = torch.tensor(pr.reshape(1, 2), dtype=torch.float32, device=eng.device)
            _, logq = run.step(a, prev, pr, value_in=v)
            q_all[r] = float(logq.item())
            prev = a
        lw_all[b] = 0.0
    np.testing.assert_allclose(q_all, isr['prop_lp'], rtol=1e-4, atol=1e-4)
    # trace log weight: sum_t (log p - log q) + observed likelihood terms (taken from the record)
    lw = np.array([np.sum(isr['prior_lp'][off[b]:off[b + 1]] - q_all[off[b]:off[b + 1]]) for b in range(len(off) - 1)])
    np.testing.assert_allclose(lw + isr['obs_lw'], isr['lw'], rtol=1e-4, atol=1e-4)


def test_is_first_statement_is_shared_and_batched_matches_single():
    """The first sample statement has identical LSTM input for every particle: the batched call (network evaluated
    once) must equal N independent batch-1 calls."""
    meta, params, batch, loss, isr = load_golden('gumm')
    eng = engine_from_golden(meta, params)
    run = _is_runner(eng)
    run.init(isr['observe'])
    a0 = eng.spec.address_id[meta['is_addresses'][0]]
    a1 = eng.spec.address_id[meta['is_addresses'][1]]
    n = 1000
    vals = torch.linspace(-0.99, 0.99, n, device=eng.device)
    prior = torch.tensor([[-1.0, 1.0]], device=eng.device)
    run.begin(n)
    _, q0 = run.step(a0, None, prior, value_in=vals)
    _, q1 = run.step(a1, a0, prior, value_in=vals.flip(0))
    q0, q1 = q0.cpu().numpy().copy(), q1.cpu().numpy().copy()
    for i in (0, 17, 999):
        run.begin(1)
        _, s0 = run.step(a0, None, prior, value_in=vals[i:i + 1])
        _, s1 = run.step(a1, a0, prior, value_in=vals.flip(0)[i:i + 1])
        assert abs(float(s0.item()) - q0[i]) < 1e-5 and abs(float(s1.item()) - q1[i]) < 2e-5


def test_is_sampling_distribution_and_weights():
    """Device-side sampling: draws follow the proposal (moments of a Normal mixture), truncated draws stay inside
    the support, self-normalised weights with q = proposal and target = proposal are all equal (ESS = N)."""
    meta, params, batch, loss, isr = load_golden('gum')
    eng = engine_from_golden(meta, params)
    run = _is_runner(eng)
    run.init(isr['observe'])
    n = 200000
    prior = torch.tensor([[1.0, 5.0 ** 0.5]], device=eng.device)
    run.begin(n)
    v, logq = run.step(0, None, prior, seed=1234)
    mu, sd, p = [np.asarray(x, np.float64) for x in np.split(isr['prop_params'][0], 3)]
    mean_ref = float((p * mu).sum())
    var_ref = float((p * (sd ** 2 + mu ** 2)).sum() - mean_ref ** 2)
    vs = v.cpu().numpy().astype(np.float64)
    assert abs(vs.mean() - mean_ref) < 5 * np.sqrt(var_ref / n) + 1e-3
    assert abs(vs.var() - var_ref) / var_ref < 0.03
    # log q of the sampled values equals an independent evaluation of the mixture density
    comp = O.normal_log_prob(vs[:2000, None], mu[None], sd[None])
    np.testing.assert_allclose(logq.cpu().numpy()[:2000], O.mixture_log_prob(comp, np.tile(p, (2000, 1))), rtol=1e-4, atol=1e-4)
    stats = run.stats(logq - logq, v)
    assert abs(stats['ess'] - n) < 1e-6 * n
    # determinism: same seed and offset -> same particles
    run.begin(n)
    v_again, _ = run.step(0, None, prior, seed=1234)
    assert torch.equal(v, v_again)


def test_is_truncated_sampling_stays_in_support():
    meta, params, batch, loss, isr = load_golden('gumm')
    eng = engine_from_golden(meta, params)
    run = _is_runner(eng)
    run.init(isr['observe'])
    a0 = eng.spec.address_id[meta['is_addresses'][0]]
    n = 100000
    prior = torch.tensor([[-1.0, 1.0]], device=eng.device)
    run.begin(n)
    v, logq = run.step(a0, None, prior, seed=99)
    vs = v.cpu().numpy()
    assert np.all(np.isfinite(vs)) and vs.min() >= -1.0 and vs.max() < 1.0
    assert np.all(np.isfinite(logq.cpu().numpy()))
    # importance-weighted normalising constant of q over its support is 1: E_q[ 1/(2 q(v)) * 1 ] with uniform target
    w = np.exp(-np.log(2.0) - logq.cpu().numpy().astype(np.float64))
    assert abs(w.mean() - 1.0) < 0.05


def test_gum_posterior_statistics_after_training():
    """End to end on the analytically solvable GUM model (reference tests/test_inference.py:173-202 thresholds):
    train on prior traces, run IS with the network for obs (8, 9): posterior mean 7.25, std sqrt(1/1.2)."""
    from pyprob_amd.is_engine import gum_posterior
    eng = _fresh_engine(64, ['mu'], 'Normal', seed=3)
    rng = np.random.default_rng(0)
    for it in range(600):
        arrays = synthetic_gum_arrays(256, seed=int(rng.integers(1 << 30)))
        eng.train_step(_packed(arrays, eng.spec).to(eng.device), lr=1e-3)
    res = gum_posterior(eng, 50000, obs=(8.0, 9.0), seed=5)
    assert abs(res['mean'] - 7.25) < 0.75
    assert abs(res['std'] - np.sqrt(1 / 1.2)) < 0.75
    assert res['ess'] > 0.15 * 50000


def test_adam_skips_untouched_zero_gradient_tensors_exactly():
    """A tensor that is 'present' with an all-zero gradient and has never had a non-zero one (W_hh of a single-statement
    program): Adam's update is exactly zero - the kernel reads its gradient chunks only - but its step count advances like
    torch's. Once a non-zero gradient arrives, or moments are loaded from a checkpoint, it takes the full path."""
    from pyprob_amd import lib as L
    meta, params, batch, loss, isr = load_golden('gum')
    eng = engine_from_golden(meta, params)
    pb = packed_from_golden(meta, batch, eng.spec).to(eng.device)
    names = list(eng.spec.tensors.keys())
    k = names.index('_layers_lstm.weight_hh_l0')
    before = eng.tensor('_layers_lstm.weight_hh_l0').clone()
    for _ in range(2):
        eng.train_step(pb, lr=1e-2)
    seen = eng.arrived.view(-1, L.PP_ADAM_SCRATCH)[:, L.PP_ADAM_SEEN].cpu().numpy()
    assert seen[k] == 0 and seen[names.index('_layers_lstm.weight_ih_l0')] == 1
    assert torch.equal(eng.tensor('_layers_lstm.weight_hh_l0'), before)
    assert int(eng.tensor_step[k]) == 2
    assert float(eng.tensor('_layers_lstm.weight_hh_l0', eng.exp_avg).abs().max()) == 0.0
    # loaded moments: the flag is raised and the zero-gradient step applies them (m decays, the weight moves)
    eng.tensor('_layers_lstm.weight_hh_l0', eng.exp_avg).fill_(0.5)
    eng.tensor('_layers_lstm.weight_hh_l0', eng.exp_avg_sq).fill_(0.25)
    eng.moments_written()
    eng.train_step(pb, lr=1e-2)
    m = eng.tensor('_layers_lstm.weight_hh_l0', eng.exp_avg)
    assert abs(float(m.max()) - 0.45) < 1e-6 and not torch.equal(eng.tensor('_layers_lstm.weight_hh_l0'), before)
    step = 3
    want = before.cpu().numpy().astype(np.float64) - (1e-2 / (1 - 0.9 ** step)) * 0.45 / (np.sqrt(0.25 * 0.999) / np.sqrt(1 - 0.999 ** step) + 1e-8)
    np.testing.assert_allclose(eng.tensor('_layers_lstm.weight_hh_l0').cpu().numpy(), want, rtol=1e-5, atol=1e-6)
