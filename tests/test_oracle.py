"""Pins the CPU oracle (oracle/ic_oracle.py) to the reference: known-answer values from the reference's own
tests, and golden vectors recorded by running the reference (tests/golden/make_golden.py). CPU only."""
import numpy as np
import pytest

from oracle import ic_oracle as O


# ---- known answers held by the reference's tests/test_distributions.py -------------------------------------
def test_kat_normal():  # :1190, :1223
    assert np.allclose(O.normal_log_prob(0.0, 0.0, 1.0), -0.918939, atol=1e-5)
    assert np.allclose(O.normal_log_prob(np.array([0., 2.]), np.array([0., 2.]), np.array([1., 3.])),
                       [-0.918939, -2.01755], atol=1e-5)


def test_kat_truncated_normal():  # :1326, :1363, :1439
    assert np.allclose(O.truncated_normal_log_prob(2.0, 2.0, 3.0, -4.0, 4.0), -1.69563, atol=1e-4)
    lp = O.truncated_normal_log_prob(np.array([0., 2.]), np.array([0., 2.]), np.array([1., 3.]),
                                     np.array([-1., -4.]), np.array([1., 4.]))
    assert np.allclose(lp, [-0.537223, -1.69563], atol=1e-4)
    # clamp_mean_between_low_high=True case: means [0,2] clamp to [0.5,1]
    lp = O.truncated_normal_log_prob(np.array([0.75, -3.]), np.array([0.5, 1.]), np.array([1., 3.]),
                                     np.array([0.5, -4.]), np.array([1., 1.]))
    assert np.allclose(lp, [0.702875, -2.11283], atol=1e-4)
    assert np.isneginf(O.truncated_normal_log_prob(5.0, 2.0, 3.0, -4.0, 4.0))


def test_kat_categorical():  # :1474, :1503
    assert np.allclose(O.categorical_log_prob(0, np.array([0.1, 0.2, 0.7])), -2.30259, atol=1e-5)
    lp = O.categorical_log_prob([0, 1], np.array([[0.1, 0.2, 0.7], [0.2, 0.5, 0.3]]))
    assert np.allclose(lp, [-2.30259, -0.693147], atol=1e-5)


def test_kat_mixture():  # :2101, :2137
    v = np.array([0.7, 8.1])
    mu = np.array([[0., 2., 3.], [1., 5., 10.]])
    sd = np.array([[.1, .1, .1], [1., 1., 1.]])
    pr = np.array([[0.7, 0.2, 0.1], [0.1, 0.2, 0.7]])
    lp = O.mixture_log_prob(O.normal_log_prob(v[:, None], mu, sd), pr)
    assert np.allclose(lp, [-23.473, -3.06649], atol=1e-3)


def test_kat_uniform():  # :1565
    assert O.uniform_log_prob(0.5, 0.0, 1.0) == 0.0


# ---- architecture pins ------------------------------------------------------------
def test_param_counts(golden):
    case, meta, params, batch, loss, isr = golden
    assert sum(v.size for v in params.values()) == meta['num_params']
    if case == 'gum':
        assert meta['num_params'] == 85215   # H=64 GUM (1 address), measured in the survey


# ---- golden vectors recorded from the reference -------------------------------------------------
def _run(golden, dtype=np.float64, want_grads=True):
    case, meta, params, batch, loss, isr = golden
    net = O.Net(params, meta['obs_names'], K=meta['mixture_components'], dtype=dtype)
    fn = O.loss_and_grads_feedforward if meta.get('network') == 'feedforward' else O.loss_and_grads
    return fn(net, batch, meta['addresses'], meta['dist_names'], want_grads=want_grads)


def test_sub_batching_matches_reference(golden):
    case, meta, params, batch, loss, isr = golden
    subs, _ = O.split_sub_batches(batch['trace_len'], batch['addr_idx'])
    assert [list(map(int, s)) for s in subs] == meta['sub_batches']


def test_loss_forward_matches_reference(golden):
    case, meta, params, batch, loss, isr = golden
    out = _run(golden, want_grads=False)
    assert abs(out['loss'] - float(loss['loss'])) <= 2e-6 * abs(float(loss['loss']))
    for i in range(len(out.get('lstm_in', []))):
        np.testing.assert_allclose(out['lstm_in'][i], loss['lstm_in_%d' % i], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(out['lstm_out'][i], loss['lstm_out_%d' % i], rtol=1e-5, atol=2e-6)
    for k, (si, t) in enumerate(meta['lp_index']):
        ref = loss['lp_%d_%d' % (si, t)]
        n = len(out['lp'][k])
        if ref.size == n * n and n > 1:      # Bernoulli proposal: the reference's log_prob is the [n, n] broadcast matrix
            ref = ref.reshape(n, n).sum(1)
        np.testing.assert_allclose(out['lp'][k], ref, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(ref).max()))


def test_loss_forward_fp32_mode(golden):
    case, meta, params, batch, loss, isr = golden
    out = _run(golden, dtype=np.float32, want_grads=False)
    assert abs(out['loss'] - float(loss['loss'])) <= 2e-5 * abs(float(loss['loss']))


def test_gradients_match_reference(golden):
    case, meta, params, batch, loss, isr = golden
    out = _run(golden)
    worst = 0.0
    # (gumm2: a 32-wide, barely trained network whose gradients are ~1e-6 in magnitude - the fp32 reference's own
    # round-off is a few 1e-4 of that; the float64 oracle is compared at 1e-3 there)
    tol = 1e-3 if case == 'gumm2' else 5e-4
    for i, n in enumerate(meta['param_names']):
        ref = loss['g%d' % i]
        got = out['grads'][n]
        scale = max(np.abs(ref).max(), 1e-6)
        err = np.abs(got - ref).max() / scale
        worst = max(worst, err)
        assert err < tol, (n, err)
        if not meta['has_grad'][i]:
            assert np.all(got == 0), n
    assert worst < tol


def test_is_log_weights_match_reference(golden):
    case, meta, params, batch, loss, isr = golden
    net = O.Net(params, meta['obs_names'], K=meta['mixture_components'])
    # IS addresses index into meta['is_addresses']; dist type from the training-batch table
    addr_to_dist = dict(zip(meta['addresses'], meta['dist_names']))
    addresses = meta['is_addresses']
    dist_names = [addr_to_dist[a] for a in addresses]
    rescore = O.is_rescore_feedforward if meta.get('network') == 'feedforward' else O.is_rescore
    p_lp, q_lp, qparams, lw = rescore(net, isr['observe'], isr['trace_len'], isr['addr'], isr['value'],
                                           isr['prior'], addresses, dist_names)
    np.testing.assert_allclose(p_lp, isr['prior_lp'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(q_lp, isr['prop_lp'], rtol=1e-4, atol=1e-4)
    # trace log weight = sum_t (log p - log q) + sum_obs log lik  (trace.py:123-125)
    np.testing.assert_allclose(lw + isr['obs_lw'], isr['lw'], rtol=1e-4, atol=1e-4)
    # proposal parameters of the first particle
    r0 = qparams[0]
    if len(r0) == 3:
        got = np.concatenate([r0[0][0], r0[1][0], r0[2][0]])
        np.testing.assert_allclose(got, isr['prop_params'][0], rtol=2e-4, atol=1e-5)


def test_ess_formula():
    lw = np.array([0.0, 0.0, 0.0, 0.0])
    assert abs(O.effective_sample_size(lw) - 4.0) < 1e-12
    assert abs(O.effective_sample_size(np.array([0.0, -1e9])) - 1.0) < 1e-12


def test_neg_inf_rescue():
    """-inf log_prob rows are replaced by log(1e-8) (inference_network_lstm.py:207-213) and carry no gradient."""
    y = np.zeros((2, 30))
    prior = np.array([[-1., 1.], [-1., 1.]])
    lp, dy, _ = O.head_truncated_normal_mixture(y, prior, np.array([0.3, 1.5]), 10)
    assert np.isfinite(lp[0]) and np.isneginf(lp[1])
    assert np.all(np.isfinite(dy[0])) and np.all(dy[1, :20] == 0)


def test_torch_restatement_matches_the_reference_records():
    """oracle/torch_ref.py (the torch-CPU restatement bench.py times as `cpu_baseline_torch`): loss and every gradient of
    the `gum` golden minibatch, as the reference recorded them (same torch kernels: agreement to the last bits)."""
    import torch
    from conftest import load_golden
    from oracle.torch_ref import GumNetwork
    meta, params, batch, loss, isr = load_golden('gum')
    net = GumNetwork(meta['lstm_dim'], K=meta['mixture_components'])
    address = meta['addresses'][0]
    net.load_reference_state(params, meta['obs_names'], address)
    out = net.loss(torch.tensor(batch['obs']), torch.tensor(batch['values']), torch.tensor(batch['prior'][:, 0]),
                   torch.tensor(batch['prior'][:, 1]))
    assert abs(float(out.detach()) - float(loss['loss'])) < 1e-6
    out.backward()
    gold = {n: loss['g%d' % i] for i, n in enumerate(meta['param_names'])}
    pairs = {'_layers_lstm.weight_ih_l0': net.lstm.weight_ih_l0, '_layers_lstm.weight_hh_l0': net.lstm.weight_hh_l0,
             '_layers_lstm.bias_ih_l0': net.lstm.bias_ih_l0,
             '_layers_proposal.%s._ff._layers.0.weight' % address: net.proposal[0].weight,
             '_layers_proposal.%s._ff._layers.1.bias' % address: net.proposal[1].bias,
             '_layers_observe_embedding_final._layers.0.weight': net.final[0].weight,
             '_layers_observe_embedding.obs0._layers.0.weight': net.obs[0][0].weight,
             '_layers_address_embedding.' + address: net.address_embedding}
    for name, p in pairs.items():
        ref = gold[name]
        got = np.zeros_like(ref) if p.grad is None else p.grad.numpy()
        assert np.abs(got - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-6), name


@pytest.mark.parametrize('name', ['ADAM', 'SGD', 'ADAM_LARC', 'SGD_LARC'])
def test_optimizer_restatements_follow_the_recorded_trajectories(name):
    """O.adam_step / O.sgd_step / O.larc_scale against trajectories recorded from torch.optim.Adam / SGD(nesterov) and the
    reference's LARC wrapper (tests/golden/make_optim_golden.py; inference_network.py:343-355, optimizer_larc.py:72-107),
    float64, including a tensor without gradient at one step and an all-zero tensor (LARC's epsilon branch)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'optim_steps.npz'))
    lr, momentum = float(g['lr']), float(g['momentum'])
    for w, wd in enumerate(g['weight_decays']):
        key = '{}_{}'.format(name, w)
        P = [g['{}_p0_{}'.format(key, k)].copy() for k in range(3)]
        B = [np.zeros_like(p) for p in P]
        M = [np.zeros_like(p) for p in P]
        V = [np.zeros_like(p) for p in P]
        steps = [0, 0, 0]
        for it in range(6):
            present = g['{}_present{}'.format(key, it)]
            for k in range(3):
                if not present[k]:
                    continue
                grad, decay = g['{}_g{}_{}'.format(key, it, k)], float(wd)
                if name.endswith('LARC'):
                    grad, decay = O.larc_scale(P[k], grad, lr, decay), 0.0
                if name.startswith('SGD'):
                    O.sgd_step(P[k], grad, B[k], lr, momentum, True, decay)
                else:
                    steps[k] += 1
                    O.adam_step(P[k], grad, M[k], V[k], steps[k], lr, weight_decay=decay)
            for k in range(3):
                np.testing.assert_allclose(P[k], g['{}_p{}_{}'.format(key, it + 1, k)], rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize('case', ['gum', 'gumm'])
def test_is_log_weights_of_ten_thousand_reference_particles(case):
    """10^4 particles sampled and scored by the reference with the golden networks
    (tests/golden/make_is_10k.py): log-weights from -2.7 down to -152. The oracle re-scores every 4th of them (the CPU
    suite's time budget) - prior log_prob, proposal log_prob and the trace log-weight - to the north star's 1e-4."""
    import os
    from conftest import GOLDEN, load_golden
    meta, params, batch, loss, isr = load_golden(case)
    big = np.load(os.path.join(GOLDEN, case + '_is10k.npz'))
    assert len(big['lw']) == 10000 and float(big['lw'].max() - big['lw'].min()) > 100
    net = O.Net(params, meta['obs_names'], K=meta['mixture_components'])
    addresses = [str(a) for a in big['addresses']]
    dist_names = ['Normal' if '__Normal__' in a else 'Uniform' for a in addresses]
    off = np.concatenate([[0], np.cumsum(big['trace_len'])])
    pick = np.arange(0, 10000, 4)
    rows = np.concatenate([np.arange(off[b], off[b + 1]) for b in pick])
    p_lp, q_lp, _, lw = O.is_rescore(net, big['observe'], big['trace_len'][pick], big['addr'][rows], big['value'][rows],
                                     big['prior'][rows], addresses, dist_names)
    np.testing.assert_allclose(p_lp, big['prior_lp'][rows], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(q_lp, big['prop_lp'][rows], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(lw + big['obs_lw'][pick], big['lw'][pick], rtol=1e-4, atol=1e-4)
    # the records are self-consistent: lw = sum (log p - log q) + observed terms, for all 10^4 (trace.py:123-125)
    d = big['prior_lp'] - big['prop_lp']
    sums = np.add.reduceat(d, off[:-1])
    np.testing.assert_allclose(sums + big['obs_lw'], big['lw'], rtol=1e-5, atol=1e-4)


def _lockstep_steps(trace_len, addr, value, prior, addresses, dist_names):
    """Ragged trace-major records -> the statement list of O.is_rescore_lockstep: statement (t, address) for the traces whose
    t-th controlled variable has that address. Returns (steps, row index of every step's entries in the ragged arrays)."""
    off = np.concatenate([[0], np.cumsum(trace_len)]).astype(np.int64)
    steps, where = [], []
    for t in range(int(np.max(trace_len))):
        live = np.nonzero(trace_len > t)[0]
        r = off[live] + t
        for a in np.unique(addr[r]):
            sel = addr[r] == a
            steps.append(dict(address=addresses[a], dist_name=dist_names[a], values=value[r[sel]], prior=prior[r[sel]],
                              rows=live[sel]))
            where.append(r[sel])
    return steps, where


@pytest.mark.parametrize('case', ['gum', 'gumm', 'cat', 'poi', 'ber'])
def test_vectorised_rescoring_equals_the_reference_records(case):
    """O.is_rescore_lockstep (particles in lock step, what the GPU parity tests of large posteriors use) pinned on the
    reference's own per-particle records and on the per-trace restatement."""
    from conftest import load_golden
    meta, params, batch, loss, isr = load_golden(case)
    net = O.Net(params, meta['obs_names'], K=meta['mixture_components'])
    addr_to_dist = dict(zip(meta['addresses'], meta['dist_names']))
    addresses = meta['is_addresses']
    dist_names = [addr_to_dist[a] for a in addresses]
    steps, where = _lockstep_steps(isr['trace_len'], isr['addr'], isr['value'], isr['prior'], addresses, dist_names)
    per_step, lw = O.is_rescore_lockstep(net, isr['observe'], steps, len(isr['trace_len']), chunk=7)
    p_lp, q_lp = np.zeros(len(isr['value'])), np.zeros(len(isr['value']))
    for (p, q), r in zip(per_step, where):
        p_lp[r], q_lp[r] = p, q
    np.testing.assert_allclose(p_lp, isr['prior_lp'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(q_lp, isr['prop_lp'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(lw + isr['obs_lw'], isr['lw'], rtol=1e-4, atol=1e-4)
    p1, q1, _, lw1 = O.is_rescore(net, isr['observe'], isr['trace_len'], isr['addr'], isr['value'], isr['prior'], addresses,
                                  dist_names)
    np.testing.assert_allclose(q_lp, q1, rtol=0, atol=1e-6)
    np.testing.assert_allclose(lw, lw1, rtol=0, atol=1e-5)


@pytest.mark.parametrize('case', ['gum', 'gumm'])
def test_vectorised_rescoring_of_all_ten_thousand_reference_particles(case):
    import os
    from conftest import GOLDEN, load_golden
    meta, params, batch, loss, isr = load_golden(case)
    big = np.load(os.path.join(GOLDEN, case + '_is10k.npz'))
    net = O.Net(params, meta['obs_names'], K=meta['mixture_components'])
    addresses = [str(a) for a in big['addresses']]
    dist_names = ['Normal' if '__Normal__' in a else 'Uniform' for a in addresses]
    steps, where = _lockstep_steps(big['trace_len'], big['addr'], big['value'], big['prior'], addresses, dist_names)
    per_step, lw = O.is_rescore_lockstep(net, big['observe'], steps, len(big['trace_len']))
    q_lp = np.zeros(len(big['value']))
    for (p, q), r in zip(per_step, where):
        q_lp[r] = q
    np.testing.assert_allclose(q_lp, big['prop_lp'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(lw + big['obs_lw'], big['lw'], rtol=1e-4, atol=1e-4)
