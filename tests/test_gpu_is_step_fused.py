"""The fused importance-sampling statement (csrc/is_step_fused.hip: gates + LSTM cell + both head layers + draw + log q of N
particles with per-particle state in ONE kernel; `_infer_step` pyprob/nn/inference_network_lstm.py:82-134 + state.sample's IC
branch pyprob/state.py:203-219) against the float64 oracle, through the C ABI (pp_is_step / pp_is_step_rows via the
`pyprob_hip` operators): new (h, c), log q of the drawn values, the in-place row-index path, the shared-state second
statement, ragged panel tails, the categorical / Bernoulli heads that are sampled by their own kernel."""
import ctypes as C

import numpy as np
import pytest

from oracle import ic_oracle as O
from pyprob_amd import lib as L
from pyprob_amd.spec import NetSpec

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

EMB = {'obs0': {'dim': 32}, 'obs1': {'dim': 32}}
ADDRS = [('a_normal', 'Normal', None), ('a_uniform', 'Uniform', None), ('a_cat', 'Categorical', 7), ('a_poisson', 'Poisson', None),
         ('a_bern', 'Bernoulli', None)]


@pytest.fixture(autouse=True, params=['2', '3'], ids=['one_kernel', 'split'])
def _force_fused(monkeypatch, request):
    """Every case runs the statement both ways: PP_IS_STEP_FUSED=2 = the one-kernel statement at any n, 3 = the two-launch split
    statement (is_small_lstm_kernel + the head-only instantiation) at any n. By default pp_is_step picks by the number of
    particles (split up to 4 096); the parity cases below use small panel counts on purpose."""
    monkeypatch.setenv('PP_IS_STEP_FUSED', request.param)
    return request.param


def _engine(H, seed=0, depth=1, emb=None):
    from pyprob_amd.engine import ICEngine
    from pyprob_amd.is_engine import ISRunner
    spec = NetSpec(emb or EMB, lstm_dim=H, lstm_depth=depth)
    eng = ICEngine(spec, device='cuda:0', seed=seed)
    eng.add_addresses(ADDRS)
    rng = np.random.default_rng(seed + 1)
    # trained-looking weights: larger than the default initialisation so that gates and mixtures are not near-uniform
    sd = {k: (v.numpy() * (3.0 if ('lstm' in k or 'proposal' in k) else 1.0)).astype(np.float32) for k, v in eng.state_dict().items()}
    for k in sd:
        if k.endswith('bias') or 'bias_' in k:
            sd[k] = (sd[k] + 0.1 * rng.standard_normal(sd[k].shape)).astype(np.float32)
    eng.load_state_dict(sd)
    run = ISRunner(eng)
    run.init([8.0, 9.0])
    return eng, run, sd


def _oracle_statement(sd, H, observe, prev, cur, prev_val, h0, c0, values, prior, depth=1, emb=None):
    """One `_infer_step` for n particles in float64: returns (h, c, log q, head outputs y); depth > 1: (h0, c0) and the returned
    states are [depth, n, H] (nn.LSTM(I, H, depth): layer k reads the new hidden rows of layer k - 1)."""
    net = O.Net(sd, list(emb or EMB), K=10)
    a_prev, d_prev = prev
    a_cur, d_cur = cur
    n = len(prev_val)
    E, _ = O.embed_observe(net, np.asarray(observe, np.float64).reshape(1, -1))
    W_ih, W_hh, b_ih, b_hh = net.lstm_layer(0)
    x = np.zeros((n, W_ih.shape[1]))
    col = E.shape[1]
    x[:, :col] = E[0]
    cat = 7 if d_prev == 'Categorical' else None
    s, _ = O.sample_embedding(net, a_prev, d_prev, prev_val, cat)
    S, Ed, Ea = s.shape[1], 8, 64
    x[:, col:col + S] = s
    x[:, col + S:col + S + Ed] = net.P['_layers_distribution_type_embedding.' + d_prev]
    x[:, col + S + Ed:col + S + Ed + Ea] = net.P['_layers_address_embedding.' + a_prev]
    c2 = col + S + Ed + Ea
    x[:, c2:c2 + Ed] = net.P['_layers_distribution_type_embedding.' + d_cur]
    x[:, c2 + Ed:] = net.P['_layers_address_embedding.' + a_cur]
    if depth == 1:
        _, _, (h, c) = O.lstm_forward(x[None], W_ih, W_hh, b_ih, b_hh, h0.astype(np.float64), c0.astype(np.float64))
        top = h
    else:
        hs, cs, inp = [], [], x
        for l in range(depth):
            _, _, (hl, cl) = O.lstm_forward(inp[None], *net.lstm_layer(l), h0[l].astype(np.float64), c0[l].astype(np.float64))
            hs.append(hl); cs.append(cl); inp = hl
        h, c, top = np.stack(hs), np.stack(cs), inp
    Ws, bs = net.ff('_layers_proposal.%s._ff' % a_cur)
    y, _ = O.ff_forward(top, Ws, bs, False)
    h_top = top
    lq = None
    if values is not None:
        if d_cur == 'Bernoulli':
            lq = np.concatenate([np.diag(O.head_forward(net, a_cur, d_cur, h_top[i:i + 128], prior[i:i + 128], values[i:i + 128])[2][1])
                                 for i in range(0, n, 128)])
        else:
            lq, _, _ = O.head_forward(net, a_cur, d_cur, h_top, prior, values)
    return h, c, lq, y


def _prior_for(dist, n, rng):
    if dist == 'Normal':
        return np.stack([rng.normal(0, 1, n), rng.uniform(0.5, 2.0, n)], 1).astype(np.float32)
    if dist == 'Uniform':
        lo = rng.uniform(-2, 0, n)
        return np.stack([lo, lo + rng.uniform(0.5, 3, n)], 1).astype(np.float32)
    return np.zeros((n, 2), np.float32)


def _prev_values(dist, n, rng):
    if dist == 'Categorical':
        return rng.integers(0, 7, n).astype(np.float32)
    return rng.normal(0, 1.5, n).astype(np.float32)


def _ids(eng, name):
    return eng.spec.address_id[name]


@pytest.mark.parametrize('H,n,prev,cur,depth', [
    (512, 1000, ('a_normal', 'Normal'), ('a_uniform', 'Uniform'), 1),        # TruncatedNormal mixture, ragged last panel
    (512, 32, ('a_uniform', 'Uniform'), ('a_normal', 'Normal'), 1),          # exactly one panel
    (512, 4133, ('a_cat', 'Categorical'), ('a_normal', 'Normal'), 1),        # one-hot sample embedding of the previous value
    (512, 257, ('a_normal', 'Normal'), ('a_poisson', 'Poisson'), 1),         # Poisson head (TN mixture on [0, 40])
    (512, 300, ('a_normal', 'Normal'), ('a_cat', 'Categorical'), 1),         # head outputs -> the categorical kernel
    (512, 300, ('a_uniform', 'Uniform'), ('a_bern', 'Bernoulli'), 1),
    (256, 777, ('a_uniform', 'Uniform'), ('a_uniform', 'Uniform'), 1),
    # H = 1024: the LSTM step is the wide launch (two workgroups per 32 particles, half of the hidden units each), head layers and
    # draw the head-only launch (UB = 4: the activations of layer 1 go over the hidden tile in LDS)
    (1024, 1000, ('a_normal', 'Normal'), ('a_uniform', 'Uniform'), 1),
    (1024, 33, ('a_uniform', 'Uniform'), ('a_normal', 'Normal'), 1),
    (1024, 300, ('a_cat', 'Categorical'), ('a_cat', 'Categorical'), 1),
    # small networks (is_step_small.hip: 64 particles per workgroup, wave = 32 rows x 32 units x four gates; BASELINE.json
    # configs[0]'s plumbing network is H = 64, the reference's own tests train lstm_dim 32 / 64), one to three layers
    (64, 500, ('a_normal', 'Normal'), ('a_uniform', 'Uniform'), 1),
    (128, 200, ('a_uniform', 'Uniform'), ('a_normal', 'Normal'), 1),
    (32, 1000, ('a_normal', 'Normal'), ('a_uniform', 'Uniform'), 1),
    (32, 64, ('a_uniform', 'Uniform'), ('a_normal', 'Normal'), 2),           # exactly one workgroup; the gumm2 golden's shape
    (32, 4133, ('a_cat', 'Categorical'), ('a_normal', 'Normal'), 2),
    (64, 257, ('a_normal', 'Normal'), ('a_poisson', 'Poisson'), 2),
    (64, 300, ('a_normal', 'Normal'), ('a_cat', 'Categorical'), 2),
    (128, 777, ('a_uniform', 'Uniform'), ('a_uniform', 'Uniform'), 2),
    (128, 300, ('a_uniform', 'Uniform'), ('a_bern', 'Bernoulli'), 2),
    (64, 300, ('a_uniform', 'Uniform'), ('a_bern', 'Bernoulli'), 3),
    (128, 300, ('a_uniform', 'Uniform'), ('a_uniform', 'Uniform'), 4),
    (64, 1, ('a_uniform', 'Uniform'), ('a_normal', 'Normal'), 4),
    # the other multiples of 32 up to 256 (and H = 256 from two layers on): the same kernel with run-time indices
    (96, 200, ('a_uniform', 'Uniform'), ('a_normal', 'Normal'), 1),
    (160, 333, ('a_cat', 'Categorical'), ('a_uniform', 'Uniform'), 2),
    (224, 100, ('a_normal', 'Normal'), ('a_poisson', 'Poisson'), 1),
    (256, 300, ('a_normal', 'Normal'), ('a_uniform', 'Uniform'), 2),
    # shapes without a fused kernel: the chain of GEMM launches, same oracle
    (48, 200, ('a_uniform', 'Uniform'), ('a_normal', 'Normal'), 1),
    (512, 150, ('a_normal', 'Normal'), ('a_uniform', 'Uniform'), 2),
])
def test_fused_statement_against_the_oracle(H, n, prev, cur, depth):
    from pyprob_amd.ops import ops
    eng, run, sd = _engine(H, depth=depth)
    # (the fixture forces the fused statement at any n; by default H = 1024 takes it from 2 049 particles on)
    fused = (H in (512, 1024) and depth == 1) or (H % 32 == 0 and H <= 256)
    assert eng.lib.pp_is_step_fused_supported(C.byref(eng.net), _ids(eng, cur[0]), n) == (1 if fused else 0)
    rng = np.random.default_rng(5)
    h0 = (0.5 * rng.standard_normal((depth, n, H))).astype(np.float32).clip(-0.99, 0.99)
    c0 = rng.standard_normal((depth, n, H)).astype(np.float32)
    pv = _prev_values(prev[1], n, rng)
    prior = _prior_for(cur[1], n, rng)
    dev = eng.device
    h = torch.from_numpy(h0.copy()).to(dev).contiguous()
    c = torch.from_numpy(c0.copy()).to(dev).contiguous()
    run._ensure_ws(n)
    pt = torch.from_numpy(prior).to(dev) if cur[1] in ('Normal', 'Uniform') else (
        torch.zeros(1, 2, device=dev) if cur[1] == 'Poisson' else None)
    if cur[1] == 'Poisson':
        pt = torch.tensor([[0.0, 40.0]], device=dev)
        prior = np.tile(np.array([[0.0, 40.0]], np.float32), (n, 1))
    value, logq = ops.is_step(eng.params, run.ws, eng.net_handle, _ids(eng, cur[0]), _ids(eng, prev[0]), n, run.e_obs,
                              torch.from_numpy(pv).to(dev), pt, h, c, n, None, 1234, 0)
    torch.cuda.synchronize()
    v = value.cpu().numpy()
    assert np.all(np.isfinite(v))
    if cur[1] == 'Uniform':
        assert np.all((v >= prior[:, 0]) & (v < prior[:, 1]))
    o_h0, o_c0 = (h0[0], c0[0]) if depth == 1 else (h0, c0)
    href, cref, lq_ref, _ = _oracle_statement(sd, H, [8.0, 9.0], prev, cur, pv, o_h0, o_c0, v.astype(np.float64), prior.astype(np.float64),
                                              depth=depth)
    # the LSTM cell on v_exp_f32 / v_rcp_f32 sigmoid / tanh: absolute error of (h, c) asserted here
    eh = np.abs(h.cpu().numpy().reshape(np.shape(href)) - href).max()
    ec = np.abs(c.cpu().numpy().reshape(np.shape(cref)) - cref).max()
    tol = (1.0 if H <= 512 else 2.0) * (1.0 if depth == 1 else 1.5)      # (K = 1028 terms per gate at H = 1024)
    assert eh < 4e-6 * tol and ec < 2e-5 * tol, (eh, ec)
    lq = logq.cpu().numpy()
    ok = np.isfinite(lq_ref)
    assert ok.mean() > 0.99
    err = np.abs(lq[ok] - lq_ref[ok]) / np.maximum(1.0, np.abs(lq_ref[ok]))
    assert err.max() < 1e-4, err.max()
    # re-scoring the same values (value_in) reproduces log q bit for bit and leaves the values alone
    h2 = torch.from_numpy(h0.copy()).to(dev).contiguous()
    c2 = torch.from_numpy(c0.copy()).to(dev).contiguous()
    v2, lq2 = ops.is_step(eng.params, run.ws, eng.net_handle, _ids(eng, cur[0]), _ids(eng, prev[0]), n, run.e_obs,
                          torch.from_numpy(pv).to(dev), pt, h2, c2, n, value, 99, 0)
    assert torch.equal(v2, value) and torch.equal(lq2, logq) and torch.equal(h2, h) and torch.equal(c2, c)


@pytest.mark.parametrize('H,depth', [(512, 1), (1024, 1), (32, 2), (64, 1), (128, 2), (192, 2)])
def test_fused_statement_equals_the_unfused_chain(monkeypatch, _force_fused, H, depth):
    """A/B inside one process: PP_IS_STEP_FUSED=0 takes the gather -> GEMM -> GEMM -> cell -> head chain. Same Philox
    counters, so the draws agree to the rounding of the proposal parameters; states agree to fp32 summation order."""
    from pyprob_amd.ops import ops
    n = 2000
    eng, run, sd = _engine(H, seed=3, depth=depth)
    rng = np.random.default_rng(9)
    h0 = (0.5 * rng.standard_normal((depth, n, H))).astype(np.float32)
    c0 = rng.standard_normal((depth, n, H)).astype(np.float32)
    pv = rng.normal(0, 1, n).astype(np.float32)
    prior = _prior_for('Uniform', n, rng)
    dev = eng.device
    outs = []
    for flag in (_force_fused, '0'):
        monkeypatch.setenv('PP_IS_STEP_FUSED', flag)
        h = torch.from_numpy(h0.copy()).to(dev).contiguous()
        c = torch.from_numpy(c0.copy()).to(dev).contiguous()
        run._ensure_ws(n)
        v, lq = ops.is_step(eng.params, run.ws, eng.net_handle, _ids(eng, 'a_uniform'), _ids(eng, 'a_normal'), n, run.e_obs,
                            torch.from_numpy(pv).to(dev), torch.from_numpy(prior).to(dev), h, c, n, None, 7, 11)
        outs.append((v.cpu().numpy(), lq.cpu().numpy(), h.cpu().numpy(), c.cpu().numpy()))
    (v1, l1, h1, c1), (v0, l0, h0_, c0_) = outs
    tol = (1.0 if H <= 512 else 2.0) * depth
    assert np.abs(h1 - h0_).max() < 5e-6 * tol and np.abs(c1 - c0_).max() < 2e-5 * tol
    rel = np.abs(v1 - v0) / np.maximum(1e-3, np.abs(v0))
    assert np.median(rel) < 1e-5 and np.quantile(rel, 0.99) < 1e-3
    close = rel < 1e-5
    np.testing.assert_allclose(l1[close], l0[close], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('H,depth,m', [(512, 1, 1777), (1024, 1, 1777), (64, 1, 1777), (32, 2, 1777), (128, 2, 900), (512, 1, 1), (64, 2, 1), (96, 1, 700), (256, 2, 700)])
def test_row_index_list_updates_the_state_in_place(H, depth, m):
    """pp_is_step_rows: the particles of a diverged path own scattered rows of (h, c) - [depth, total, H], state_rows = total -;
    the rows are read and written in place, every other row is untouched, and the result equals the compact call on the
    gathered rows. m = 1: a path of ONE particle is not the shared first state (up to ABI 13 the callers passed n as state_rows
    and such a path read row 0)."""
    from pyprob_amd.ops import ops
    total = 5000
    eng, run, sd = _engine(H, seed=4, depth=depth)
    rng = np.random.default_rng(2)
    h0 = (0.5 * rng.standard_normal((depth, total, H))).astype(np.float32)
    c0 = rng.standard_normal((depth, total, H)).astype(np.float32)
    rows = np.sort(rng.choice(np.arange(1, total), m, replace=False)).astype(np.int64)
    pv = rng.normal(0, 1, m).astype(np.float32)
    prior = _prior_for('Normal', m, rng)
    dev = eng.device
    h = torch.from_numpy(h0.copy()).to(dev).contiguous()
    c = torch.from_numpy(c0.copy()).to(dev).contiguous()
    run._ensure_ws(max(m, 2))
    a, p = _ids(eng, 'a_normal'), _ids(eng, 'a_uniform')
    v, lq = ops.is_step_rows(eng.params, run.ws, eng.net_handle, a, p, m, run.e_obs, torch.from_numpy(pv).to(dev),
                             torch.from_numpy(prior).to(dev), h, c, total, torch.from_numpy(rows).to(dev), None, 5, 0)
    hg = torch.from_numpy(h0[:, rows].copy()).to(dev).contiguous()
    cg = torch.from_numpy(c0[:, rows].copy()).to(dev).contiguous()
    if m == 1:      # (a compact call on ONE row would be the shared-state statement: two copies of the row instead)
        hg, cg = hg.repeat(1, 2, 1).contiguous(), cg.repeat(1, 2, 1).contiguous()
        v2, lq2 = ops.is_step(eng.params, run.ws, eng.net_handle, a, p, 2, run.e_obs, torch.from_numpy(np.repeat(pv, 2)).to(dev),
                              torch.from_numpy(np.repeat(prior, 2, 0)).to(dev), hg, cg, 2, None, 5, 0)
        v2, lq2, hg, cg = v2[:1], lq2[:1], hg[:, :1], cg[:, :1]
    else:
        v2, lq2 = ops.is_step(eng.params, run.ws, eng.net_handle, a, p, m, run.e_obs, torch.from_numpy(pv).to(dev),
                              torch.from_numpy(prior).to(dev), hg, cg, m, None, 5, 0)
    assert torch.equal(v, v2) and torch.equal(lq, lq2)
    hn, cn = h.cpu().numpy(), c.cpu().numpy()
    assert np.array_equal(hn[:, rows], hg.cpu().numpy()) and np.array_equal(cn[:, rows], cg.cpu().numpy())
    rest = np.setdiff1d(np.arange(total), rows)
    assert np.array_equal(hn[:, rest], h0[:, rest]) and np.array_equal(cn[:, rest], c0[:, rest])
    o_h0, o_c0 = (h0[0, rows], c0[0, rows]) if depth == 1 else (h0[:, rows], c0[:, rows])
    href, cref, lq_ref, _ = _oracle_statement(sd, H, [8.0, 9.0], ('a_uniform', 'Uniform'), ('a_normal', 'Normal'), pv, o_h0, o_c0,
                                              v.cpu().numpy().astype(np.float64), prior.astype(np.float64), depth=depth)
    assert np.abs(hn[:, rows].reshape(np.shape(href)) - href).max() < (4e-6 if H <= 512 else 8e-6) * depth
    np.testing.assert_allclose(lq.cpu().numpy(), lq_ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('H,depth', [(512, 1), (1024, 1), (32, 2), (64, 3), (128, 2), (160, 2)])
def test_second_statement_with_the_shared_first_state(H, depth):
    """state_rows = 1: row 0 (of every layer) holds the state every particle left the first statement with; its recurrent
    product joins the bias row, the cell reads the one shared previous cell state, all n rows are written."""
    from pyprob_amd.ops import ops
    n = 3001
    eng, run, sd = _engine(H, seed=6, depth=depth)
    rng = np.random.default_rng(3)
    h_row = (0.5 * rng.standard_normal((depth, 1, H))).astype(np.float32)
    c_row = rng.standard_normal((depth, 1, H)).astype(np.float32)
    pv = rng.uniform(-1, 1, n).astype(np.float32)
    prior = np.tile(np.array([[-1.0, 1.0]], np.float32), (n, 1))
    dev = eng.device
    h = torch.zeros(depth, n, H, device=dev)
    c = torch.zeros(depth, n, H, device=dev)
    h[:, 0] = torch.from_numpy(h_row[:, 0]).to(dev)
    c[:, 0] = torch.from_numpy(c_row[:, 0]).to(dev)
    run._ensure_ws(n)
    v, lq = ops.is_step(eng.params, run.ws, eng.net_handle, _ids(eng, 'a_uniform'), _ids(eng, 'a_uniform'), n, run.e_obs,
                        torch.from_numpy(pv).to(dev), torch.tensor([[-1.0, 1.0]], device=dev), h, c, 1, None, 21, 0)
    o_h0, o_c0 = np.repeat(h_row, n, 1), np.repeat(c_row, n, 1)
    if depth == 1:
        o_h0, o_c0 = o_h0[0], o_c0[0]
    href, cref, lq_ref, _ = _oracle_statement(sd, H, [8.0, 9.0], ('a_uniform', 'Uniform'), ('a_uniform', 'Uniform'), pv,
                                              o_h0, o_c0, v.cpu().numpy().astype(np.float64), prior.astype(np.float64), depth=depth)
    tol = (1.0 if H <= 512 else 2.0) * depth
    assert np.abs(h.cpu().numpy().reshape(np.shape(href)) - href).max() < 4e-6 * tol
    assert np.abs(c.cpu().numpy().reshape(np.shape(cref)) - cref).max() < 2e-5 * tol
    np.testing.assert_allclose(lq.cpu().numpy(), lq_ref, rtol=1e-4, atol=1e-4)


def test_fast_activations_against_libm():
    """The cell's sigmoid / tanh run on v_exp_f32 / v_rcp_f32: with W_hh = 0 and a zero previous state the new cell state is
    sigmoid(b_i) * tanh(b_g) of the bias row - swept over [-12, 12] and compared with float64."""
    from pyprob_amd.ops import ops
    H, n = 512, 64
    eng, run, sd = _engine(H, seed=8)
    sd = dict(sd)
    sd['_layers_lstm.weight_hh_l0'] = np.zeros_like(sd['_layers_lstm.weight_hh_l0'])
    sd['_layers_lstm.weight_ih_l0'] = np.zeros_like(sd['_layers_lstm.weight_ih_l0'])
    sweep = np.linspace(-12, 12, H).astype(np.float32)
    b = np.concatenate([sweep, sweep[::-1], 0.37 * sweep, -sweep]).astype(np.float32)
    sd['_layers_lstm.bias_ih_l0'] = b
    sd['_layers_lstm.bias_hh_l0'] = np.zeros_like(b)
    eng.load_state_dict(sd)
    run.init([8.0, 9.0])
    dev = eng.device
    c0 = np.tile(np.linspace(-2, 2, H).astype(np.float32), (n, 1))
    h = torch.zeros(1, n, H, device=dev)
    c = torch.from_numpy(c0.copy()).to(dev).reshape(1, n, H).contiguous()
    run._ensure_ws(n)
    ops.is_step(eng.params, run.ws, eng.net_handle, _ids(eng, 'a_normal'), _ids(eng, 'a_normal'), n, run.e_obs,
                torch.zeros(n, device=dev), torch.tensor([[0.0, 1.0]], device=dev), h, c, n, None, 1, 0)
    b64 = b.astype(np.float64)
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    cn = sig(b64[H:2 * H]) * c0[0] + sig(b64[:H]) * np.tanh(b64[2 * H:3 * H])
    hn = sig(b64[3 * H:]) * np.tanh(cn)
    assert np.abs(c.cpu().numpy()[0] - cn).max() < 1e-6
    assert np.abs(h.cpu().numpy()[0] - hn).max() < 5e-7


@pytest.mark.parametrize('dist,use_rows,H,depth', [('Uniform', True, 512, 1), ('Normal', True, 512, 1), ('Uniform', False, 512, 1),
                                                   ('Uniform', True, 1024, 1), ('Normal', False, 1024, 1),
                                                   ('Uniform', True, 64, 2), ('Normal', False, 32, 2), ('Normal', True, 128, 1)])
def test_whole_statement_in_one_launch(dist, use_rows, H, depth):
    """pp_is_statement_rows: previous values read at the particles' rows, the drawn value scattered to values[rows] and
    lw[rows] += log p(v) - log q(v) inside the statement kernel = pp_is_step_rows + the gather / scatter / log-weight launches
    around it (same Philox counters: identical values; the two fp32 additions in the same order: identical log-weights)."""
    from pyprob_amd.ops import ops
    total = 6000
    eng, run, sd = _engine(H, seed=12, depth=depth)
    rng = np.random.default_rng(4)
    m = 2500 if use_rows else total
    rows = np.sort(rng.choice(total, m, replace=False)).astype(np.int64) if use_rows else np.arange(total)
    h0 = (0.5 * rng.standard_normal((depth, total, H))).astype(np.float32)
    c0 = rng.standard_normal((depth, total, H)).astype(np.float32)
    prev_full = rng.normal(0, 1, total).astype(np.float32)
    lw0 = rng.normal(-3, 1, total).astype(np.float32)
    prior = np.array([[-1.5, 2.0]], np.float32) if dist == 'Uniform' else np.array([[0.3, 1.7]], np.float32)
    cur = 'a_uniform' if dist == 'Uniform' else 'a_normal'
    dev = eng.device
    a, p = _ids(eng, cur), _ids(eng, 'a_normal')
    pt = torch.from_numpy(prior).to(dev)
    rt = torch.from_numpy(rows).to(dev)
    # reference: compact call + the operations the executor used to issue around it
    h = torch.from_numpy(h0.copy()).to(dev).contiguous()
    c = torch.from_numpy(c0.copy()).to(dev).contiguous()
    run._ensure_ws(m)
    pf = torch.from_numpy(prev_full).to(dev)
    v, lq = ops.is_step_rows(eng.params, run.ws, eng.net_handle, a, p, m, run.e_obs, pf.index_select(0, rt).contiguous(), pt, h, c, total, rt,
                             None, 31, 0)
    vals_ref = torch.zeros(total, device=dev)
    vals_ref.index_copy_(0, rt, v)
    lw_ref = torch.from_numpy(lw0.copy()).to(dev)
    kind = 1 if dist == 'Uniform' else 0
    plp = ops.log_prob(kind, pt[0, :1].contiguous(), 0, pt[0, 1:].contiguous(), 0, v, m)
    lw_rows = lw_ref.index_select(0, rt)
    lw_rows = lw_rows + plp
    lw_rows = lw_rows + (-1.0) * lq
    lw_ref.index_copy_(0, rt, lw_rows)
    # the whole statement in one launch
    h2 = torch.from_numpy(h0.copy()).to(dev).contiguous()
    c2 = torch.from_numpy(c0.copy()).to(dev).contiguous()
    vals = torch.zeros(total, device=dev)
    lw = torch.from_numpy(lw0.copy()).to(dev)
    ops.is_statement_rows(eng.params, run.ws, eng.net_handle, a, p, m, run.e_obs, pf, pt, h2, c2, total, rt if use_rows else None, vals, lw,
                          kind, 31, 0)
    assert torch.equal(vals, vals_ref) and torch.equal(h2, h) and torch.equal(c2, c)
    np.testing.assert_allclose(lw.cpu().numpy(), lw_ref.cpu().numpy(), rtol=0, atol=2e-6)
    rest = np.setdiff1d(np.arange(total), rows)
    assert np.array_equal(lw.cpu().numpy()[rest], lw0[rest])


def _mixture_cdf(x, params, dist, lo=None, hi=None):
    """CDF of the proposal mixture of ONE row (float64): Normal components, or TruncatedNormal components on [lo, hi]
    (pyprob/distributions/mixture.py:47-63 draws a component by its probability, then from the component)."""
    mu, sd, p = (np.asarray(a, np.float64).reshape(-1) for a in params)
    z = O.std_normal_cdf((x[:, None] - mu) / sd)
    if dist == 'Uniform':
        a, b = O.std_normal_cdf((lo - mu) / sd), O.std_normal_cdf((hi - mu) / sd)
        z = np.clip((z - a) / (b - a), 0.0, 1.0)
    return (z * p).sum(1)


@pytest.mark.parametrize('cur,H,depth', [(('a_normal', 'Normal'), 512, 1), (('a_uniform', 'Uniform'), 512, 1), (('a_uniform', 'Uniform'), 1024, 1),
                                         (('a_normal', 'Normal'), 64, 2), (('a_uniform', 'Uniform'), 32, 1)])
def test_the_drawn_values_follow_the_proposal(cur, H, depth):
    """The N-row draw itself (sixteen lanes per particle: component pick by the inclusive prefix of the clamped weights, then
    the component's Normal / inverse-CDF TruncatedNormal draw - is_step_fused.hip's tail, is_draw.hpp): re-scoring proves
    log q AT the drawn value, not that the value is drawn FROM q. Here every particle gets the same
    state, previous value and prior - one proposal q for all - and H = 512, both head kinds, one kernel and split (the
    autouse fixture): (1) Kolmogorov-Smirnov distance of the 40 000 draws to the oracle's mixture CDF, (2) the importance
    identity E_q[p(v) / q(v)] = 1 for a known target p within four standard errors, (3) log q of the device at those values."""
    from pyprob_amd.ops import ops
    n = 40000
    prev = ('a_uniform', 'Uniform') if cur[1] == 'Normal' else ('a_normal', 'Normal')
    eng, run, sd = _engine(H, seed=3, depth=depth)
    rng = np.random.default_rng(11)
    h0 = np.tile((0.5 * rng.standard_normal((depth, 1, H))).astype(np.float32).clip(-0.99, 0.99), (1, n, 1))
    c0 = np.tile(rng.standard_normal((depth, 1, H)).astype(np.float32), (1, n, 1))
    pv = np.full(n, 0.37, np.float32)
    prior = np.tile(np.array([[0.4, 1.3]] if cur[1] == 'Normal' else [[-1.0, 1.5]], np.float32), (n, 1))
    dev = eng.device
    h = torch.from_numpy(h0.copy()).to(dev).contiguous()
    c = torch.from_numpy(c0.copy()).to(dev).contiguous()
    run._ensure_ws(n)
    value, logq = ops.is_step(eng.params, run.ws, eng.net_handle, _ids(eng, cur[0]), _ids(eng, prev[0]), n, run.e_obs,
                              torch.from_numpy(pv).to(dev), torch.from_numpy(prior).to(dev), h, c, n, None, 4242, 0)
    torch.cuda.synchronize()
    v = value.cpu().numpy().astype(np.float64)
    assert np.all(np.isfinite(v)) and len(np.unique(v)) > 0.99 * n          # distinct Philox counters per particle
    href, _, _, y = _oracle_statement(sd, H, [8.0, 9.0], prev, cur, pv[:1], h0[0, :1] if depth == 1 else h0[:, :1],
                                       c0[0, :1] if depth == 1 else c0[:, :1], None, None, depth=depth)
    net = O.Net(sd, list(EMB), K=10)
    y_all = np.broadcast_to(y[:1], (n, y.shape[1]))
    lq_ref, _, params = O.head_forward(net, cur[0], cur[1], None, prior.astype(np.float64), v, y=y_all)
    one = tuple(np.asarray(a)[0] for a in params)
    lo, hi = float(prior[0, 0]), float(prior[0, 1])
    # (1) Kolmogorov-Smirnov: sup |F_n - F| of n i.i.d. draws exceeds 1.95 / sqrt(n) with probability 1e-3
    xs = np.sort(v)
    F = _mixture_cdf(xs, one, cur[1], lo, hi)
    i = np.arange(1, n + 1)
    D = max(np.abs(i / n - F).max(), np.abs((i - 1) / n - F).max())
    assert D < 1.95 / np.sqrt(n), (D, 1.95 / np.sqrt(n))
    # (2) importance identity with the oracle's q: target = a Normal narrower than the mixture / the Uniform prior on the support
    mu, sdv, p = (np.asarray(a, np.float64) for a in one)
    if cur[1] == 'Normal':
        m = float((p * mu).sum())
        s = float(np.sqrt((p * (sdv ** 2 + mu ** 2)).sum() - m * m))
        lp = O.normal_log_prob(v, m, 0.7 * s)
    else:
        assert np.all((v >= lo) & (v <= hi))
        lp = np.full(n, -np.log(hi - lo))
    w = np.exp(lp - lq_ref)
    se = w.std() / np.sqrt(n)
    assert abs(w.mean() - 1.0) < 4.0 * se + 1e-3, (w.mean(), se)
    # (3) the device's log q at its own draws
    assert np.abs(logq.cpu().numpy() - lq_ref).max() < 1e-4 * max(1.0, np.abs(lq_ref).max())


def test_the_golden_networks_of_the_reference_take_the_fused_statement():
    """`pp_is_step_fused_supported` is true for the `gumm2` golden's network (lstm_dim 32, nn.LSTM depth 2 -
    recorded from the reference, tests/golden/make_golden.py) and for the one-layer H = 64 goldens, for every address they hold."""
    from is_helpers import network_from_golden
    for case in ('gumm2', 'gumm', 'gum', 'gumd'):
        net, meta, params, _ = network_from_golden(case, 'cuda:0')
        eng = net._engine
        assert len(eng.spec.addresses) >= 1
        for a in range(len(eng.spec.addresses)):
            assert eng.lib.pp_is_step_fused_supported(C.byref(eng.net), a, 1000) == 1, (case, a)


WIDE_EMB = {'obs0': {'dim': 700}, 'obs1': {'dim': 500}}      # lstm_in = 1 200 + 4 + 2 (8 + 64) = 1 348


@pytest.mark.parametrize('H,depth', [(512, 1), (64, 2)])
def test_fused_statement_with_a_wide_observe_embedding(H, depth):
    """LSTM inputs wider than 1 024 columns (pyprob's default observe embedding is 256 wide per observable): the per-call bias row
    W_ih x_shared is accumulated through LDS in chunks of 1 024 columns - the fused statement kernels take any lstm_in."""
    from pyprob_amd.ops import ops
    n = 700
    eng, run, sd = _engine(H, seed=2, depth=depth, emb=WIDE_EMB)
    assert eng.spec.lstm_in > 1024
    prev, cur = ('a_normal', 'Normal'), ('a_uniform', 'Uniform')
    assert eng.lib.pp_is_step_fused_supported(C.byref(eng.net), _ids(eng, cur[0]), n) == 1
    rng = np.random.default_rng(5)
    h0 = (0.5 * rng.standard_normal((depth, n, H))).astype(np.float32).clip(-0.99, 0.99)
    c0 = rng.standard_normal((depth, n, H)).astype(np.float32)
    pv = _prev_values(prev[1], n, rng)
    prior = _prior_for(cur[1], n, rng)
    dev = eng.device
    h = torch.from_numpy(h0.copy()).to(dev).contiguous()
    c = torch.from_numpy(c0.copy()).to(dev).contiguous()
    run._ensure_ws(n)
    value, logq = ops.is_step(eng.params, run.ws, eng.net_handle, _ids(eng, cur[0]), _ids(eng, prev[0]), n, run.e_obs,
                              torch.from_numpy(pv).to(dev), torch.from_numpy(prior).to(dev), h, c, n, None, 1234, 0)
    torch.cuda.synchronize()
    v = value.cpu().numpy()
    o_h0, o_c0 = (h0[0], c0[0]) if depth == 1 else (h0, c0)
    href, cref, lq_ref, _ = _oracle_statement(sd, H, [8.0, 9.0], prev, cur, pv, o_h0, o_c0, v.astype(np.float64), prior.astype(np.float64),
                                              depth=depth, emb=WIDE_EMB)
    # (1 348 input terms per gate pre-activation: twice the bars of the 212-wide input)
    assert np.abs(h.cpu().numpy().reshape(np.shape(href)) - href).max() < 8e-6 * depth
    assert np.abs(c.cpu().numpy().reshape(np.shape(cref)) - cref).max() < 4e-5 * depth
    ok = np.isfinite(lq_ref)
    err = np.abs(logq.cpu().numpy()[ok] - lq_ref[ok]) / np.maximum(1.0, np.abs(lq_ref[ok]))
    assert ok.mean() > 0.99 and err.max() < 1e-4, err.max()
