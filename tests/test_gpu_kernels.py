"""GPU parity tests of the individual kernels behind the C ABI, against numpy (float64) restatements.
Run on the MI355X box: python -m pytest tests -m gpu."""
import ctypes as C

import numpy as np
import pytest

from oracle import ic_oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def env():
    from pyprob_amd import lib as L
    lib = L.load()
    assert torch.cuda.is_available(), 'the gpu tests need a ROCm device'
    assert lib.pp_device_count() >= 1, 'no gfx950 device visible to libpyprob_amd'
    return L, lib, torch.device('cuda:0')


def dev(a, device, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return t if dtype is None else t.to(dtype)


def run_gemm(env, A, B, M, N, K, akm, bkm, a_idx=None, b_idx=None, c_idx=None, bias=None, bias2=None, mask=None,
             relu=False, acc_init=None, c_rows=None, lda=None, ldb=None, split_k=0):
    L, lib, device = env
    dA, dB = dev(A, device), dev(B, device)
    crow = M if c_rows is None else c_rows
    ldc = N + 3
    Cm = np.zeros((crow, ldc), np.float32) if acc_init is None else acc_init.copy()
    dC = dev(Cm, device)
    g = L.pp_gemm_args()
    g.A, g.lda, g.B, g.ldb = dA.data_ptr(), (A.shape[1] if lda is None else lda), dB.data_ptr(), (B.shape[1] if ldb is None else ldb)
    g.C, g.ldc = dC.data_ptr(), ldc
    keep = []
    for name, arr in (('a_idx', a_idx), ('b_idx', b_idx), ('c_idx', c_idx)):
        if arr is not None:
            t = dev(np.asarray(arr, np.int32), device)
            keep.append(t)
            setattr(g, name, t.data_ptr())
    for name, arr in (('bias', bias), ('bias2', bias2)):
        if arr is not None:
            t = dev(np.asarray(arr, np.float32), device)
            keep.append(t)
            setattr(g, name, t.data_ptr())
    if mask is not None:
        t = dev(mask, device)
        keep.append(t)
        g.mask, g.ldmask = t.data_ptr(), mask.shape[1]
    g.M, g.N, g.K, g.a_kmajor, g.b_kmajor = M, N, K, int(akm), int(bkm)
    g.relu, g.accumulate = int(relu), int(acc_init is not None)
    g.split_k = int(split_k)
    L.check(lib.pp_gemm_f32(C.byref(g), L.stream_ptr()), 'pp_gemm_f32')
    torch.cuda.synchronize()
    return dC.cpu().numpy()


def ref_gemm(A, B, M, N, K, akm, bkm, a_idx, b_idx):
    A64, B64 = A.astype(np.float64), B.astype(np.float64)
    if akm:
        Ak = A64[(np.arange(K) if a_idx is None else a_idx[:K]), :M].T   # [M,K]
    else:
        Ak = A64[(np.arange(M) if a_idx is None else a_idx[:M]), :K]
    if bkm:
        Bk = B64[(np.arange(K) if b_idx is None else b_idx[:K]), :N].T   # [N,K]
    else:
        Bk = B64[(np.arange(N) if b_idx is None else b_idx[:N]), :K]
    return Ak @ Bk.T


@pytest.mark.parametrize('akm', [False, True])
@pytest.mark.parametrize('bkm', [False, True])
@pytest.mark.parametrize('shape', [(64, 64, 32), (100, 70, 45), (1024, 2048, 212), (33, 30, 271), (257, 129, 8),
                                   (5, 3, 1), (300, 512, 64)])
def test_gemm_layouts(env, akm, bkm, shape):
    M, N, K = shape
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    # leading dims: vector path needs multiples of 4; odd ones exercise the scalar path
    for pad in (0, 1):
        a_rows, a_cols = (K, M) if akm else (M, K)
        b_rows, b_cols = (K, N) if bkm else (N, K)
        lda = ((a_cols + 3) // 4) * 4 + pad
        ldb = ((b_cols + 3) // 4) * 4 + pad
        A = rng.uniform(-1, 1, (a_rows, lda)).astype(np.float32)
        B = rng.uniform(-1, 1, (b_rows, ldb)).astype(np.float32)
        got = run_gemm(env, A, B, M, N, K, akm, bkm)[:, :N]
        ref = ref_gemm(A, B, M, N, K, akm, bkm, None, None)
        err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-9)
        assert err < 2e-6, (shape, akm, bkm, pad, err)


@pytest.mark.parametrize('akm', [False, True])
@pytest.mark.parametrize('bkm', [False, True])
@pytest.mark.parametrize('shape', [(512, 200, 1500), (300, 130, 999), (2100, 1100, 600), (64, 70, 4100), (1, 2048, 212),
                                   (3, 271, 512), (4, 30, 271)])
def test_gemm_tile_families(env, akm, bkm, shape):
    """Shapes that route to every tile family of pp_gemm_f32: LDS-DMA ring tiles with four and eight waves (long K,
    K tails that are not multiples of 32 or of 4), the register-staged fallback (odd leading dimension), the
    wave-direct 32x32 tiles and the few-row GEMV; with the bias/ReLU/accumulate epilogue and a row gather."""
    M, N, K = shape
    rng = np.random.default_rng(M * 11 + N * 5 + K)
    for pad in (0, 1):
        a_rows, a_cols = (K, M) if akm else (M, K)
        b_rows, b_cols = (K, N) if bkm else (N, K)
        lda = ((a_cols + 3) // 4) * 4 + pad
        ldb = ((b_cols + 3) // 4) * 4 + pad
        A = rng.uniform(-1, 1, (a_rows + 7, lda)).astype(np.float32)
        B = rng.uniform(-1, 1, (b_rows, ldb)).astype(np.float32)
        a_idx = rng.permutation(a_rows + 7)[:a_rows] if not akm else None   # row gather of a k-contiguous operand
        bias = rng.uniform(-1, 1, N).astype(np.float32)
        init = rng.uniform(-1, 1, (M, N + 3)).astype(np.float32)
        got = run_gemm(env, A, B, M, N, K, akm, bkm, a_idx=a_idx, bias=bias, relu=True)[:, :N]
        ref = ref_gemm(A, B, M, N, K, akm, bkm, a_idx, None)
        scale = max(np.abs(ref).max(), 1e-9)
        assert np.abs(got - np.maximum(ref + bias, 0)).max() / scale < 3e-6, (shape, akm, bkm, pad)
        got = run_gemm(env, A, B, M, N, K, akm, bkm, a_idx=a_idx, acc_init=init)[:, :N]
        assert np.abs(got - (ref + init[:, :N])).max() / scale < 3e-6, (shape, akm, bkm, pad)


def test_gemm_is_transpose_detecting(env):
    """A = I with an asymmetric B: a swapped C write or operand would show."""
    M = N = K = 64
    A = np.eye(64, dtype=np.float32)
    B = (np.arange(64)[:, None] * 100 + np.arange(64)[None, :]).astype(np.float32)   # B(n,k) = 100 n + k
    got = run_gemm(env, A, B, M, N, K, False, False)[:, :N]
    np.testing.assert_array_equal(got, B.T)       # C[m,n] = sum_k I[m,k] B[n,k] = B[n,m]


def test_gemm_gather_scatter_epilogue(env):
    rng = np.random.default_rng(5)
    M, N, K = 150, 90, 72
    A = rng.uniform(-1, 1, (400, 72)).astype(np.float32)
    B = rng.uniform(-1, 1, (N, 72)).astype(np.float32)
    a_idx = rng.permutation(400)[:M]
    c_idx = rng.permutation(200)[:M]
    bias = rng.uniform(-1, 1, N).astype(np.float32)
    bias2 = rng.uniform(-1, 1, N).astype(np.float32)
    got = run_gemm(env, A, B, M, N, K, False, False, a_idx=a_idx, c_idx=c_idx, bias=bias, bias2=bias2, relu=True, c_rows=200)
    ref = np.maximum(ref_gemm(A, B, M, N, K, False, False, a_idx, None) + bias + bias2, 0)
    np.testing.assert_allclose(got[c_idx, :N], ref, rtol=2e-6, atol=2e-6)
    untouched = np.setdiff1d(np.arange(200), c_idx)
    assert np.all(got[untouched] == 0)
    # k-gather on a k-major operand + accumulate + mask
    Akm = rng.uniform(-1, 1, (500, 64)).astype(np.float32)     # [K rows, M]
    Bkm = rng.uniform(-1, 1, (500, 48)).astype(np.float32)
    kidx = rng.permutation(500)[:300]
    init = rng.uniform(-1, 1, (64, 48 + 3)).astype(np.float32)
    got = run_gemm(env, Akm, Bkm, 64, 48, 300, True, True, b_idx=kidx, a_idx=kidx, acc_init=init)
    ref = init[:, :48] + ref_gemm(Akm, Bkm, 64, 48, 300, True, True, kidx, kidx)
    np.testing.assert_allclose(got[:, :48], ref, rtol=1e-5, atol=3e-5)
    mask = rng.uniform(-1, 1, (M, N)).astype(np.float32)
    got = run_gemm(env, A, B, M, N, K, False, False, mask=mask)
    ref = np.where(mask > 0, ref_gemm(A, B, M, N, K, False, False, None, None), 0)
    np.testing.assert_allclose(got[:, :N], ref, rtol=2e-6, atol=2e-6)


def test_gemm_split_k_and_fused_colsum(env):
    """Deep-K / few-tile products take the split-K path (float atomics); masked dgrad products can emit the column
    sums (bias gradient) from the epilogue."""
    L, lib, device = env
    rng = np.random.default_rng(21)
    for (M, N, K, akm, bkm) in [(64, 64, 1024, True, True), (2048, 212, 1024, True, True), (1024, 212, 2048, False, True),
                                (271, 512, 1000, True, True)]:
        a_rows, a_cols = (K, M) if akm else (M, K)
        b_rows, b_cols = (K, N) if bkm else (N, K)
        A = rng.uniform(-1, 1, (a_rows, ((a_cols + 3) // 4) * 4)).astype(np.float32)
        B = rng.uniform(-1, 1, (b_rows, ((b_cols + 3) // 4) * 4)).astype(np.float32)
        ref = ref_gemm(A, B, M, N, K, akm, bkm, None, None)
        got = run_gemm(env, A, B, M, N, K, akm, bkm, split_k=1)[:, :N]           # fresh destination (memset inside)
        assert np.abs(got - ref).max() / np.abs(ref).max() < 3e-6, (M, N, K)
        init = rng.uniform(-1, 1, (M, N + 3)).astype(np.float32)
        got = run_gemm(env, A, B, M, N, K, akm, bkm, acc_init=init, split_k=1)[:, :N]   # accumulate onto existing values
        again = run_gemm(env, A, B, M, N, K, akm, bkm)[:, :N]                    # split_k=0 is bit-reproducible
        assert np.array_equal(again, run_gemm(env, A, B, M, N, K, akm, bkm)[:, :N])
        assert np.abs(got - (ref + init[:, :N])).max() / np.abs(ref).max() < 3e-6, (M, N, K)
    # fused colsum with mask
    M, N, K = 1000, 271, 30
    A = rng.uniform(-1, 1, (M, 32)).astype(np.float32)
    B = rng.uniform(-1, 1, (K, 272)).astype(np.float32)
    mask = rng.uniform(-1, 1, (M, N)).astype(np.float32)
    dA, dB, dM = dev(A, device), dev(B, device), dev(mask, device)
    dC = torch.zeros(M, N + 1, device=device)
    cs = torch.ones(N, device=device)
    g = L.pp_gemm_args()
    g.A, g.lda, g.B, g.ldb, g.b_kmajor = dA.data_ptr(), 32, dB.data_ptr(), 272, 1
    g.C, g.ldc, g.M, g.N, g.K = dC.data_ptr(), N + 1, M, N, K
    g.mask, g.ldmask, g.colsum = dM.data_ptr(), N, cs.data_ptr()
    L.check(lib.pp_gemm_f32(C.byref(g), L.stream_ptr()))
    ref = np.where(mask > 0, A[:, :K].astype(np.float64) @ B[:, :N].astype(np.float64), 0)
    np.testing.assert_allclose(dC.cpu().numpy()[:, :N], ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(cs.cpu().numpy(), 1 + ref.sum(0), rtol=1e-4, atol=1e-3)


def test_gemm_grouped_launch(env):
    """Independent products of different shapes in one launch equal the separate launches."""
    L, lib, device = env
    rng = np.random.default_rng(33)
    shapes = [(30, 271, 1024), (271, 512, 1024), (2048, 212, 1024), (64, 64, 1024), (16, 4, 100)]
    args = (L.pp_gemm_args * len(shapes))()
    keep, refs = [], []
    for q, (M, N, K) in enumerate(shapes):
        A = rng.uniform(-1, 1, (K, ((M + 3) // 4) * 4)).astype(np.float32)
        B = rng.uniform(-1, 1, (K, ((N + 3) // 4) * 4)).astype(np.float32)
        init = rng.uniform(-1, 1, (M, N)).astype(np.float32)
        dA, dB, dC = dev(A, device), dev(B, device), dev(init, device)
        keep += [dA, dB, dC]
        g = args[q]
        g.A, g.lda, g.a_kmajor, g.B, g.ldb, g.b_kmajor = dA.data_ptr(), A.shape[1], 1, dB.data_ptr(), B.shape[1], 1
        g.C, g.ldc, g.M, g.N, g.K, g.accumulate, g.split_k = dC.data_ptr(), N, M, N, K, 1, 1
        refs.append((dC, init + A[:, :M].astype(np.float64).T @ B[:, :N].astype(np.float64)))
    L.check(lib.pp_gemm_f32_grouped(args, len(shapes), L.stream_ptr()), 'pp_gemm_f32_grouped')
    torch.cuda.synchronize()
    for dC, ref in refs:
        assert np.abs(dC.cpu().numpy() - ref).max() / np.abs(ref).max() < 3e-6


def test_gemm_large_tile_path(env):
    rng = np.random.default_rng(9)
    M, N, K = 8192, 2048, 212          # >= 4096 tiles -> 128x128 configuration
    A = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    B = rng.uniform(-1, 1, (N, K)).astype(np.float32)
    got = run_gemm(env, A, B, M, N, K, False, False)[:, :N]
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-6


def test_colsum(env):
    L, lib, device = env
    rng = np.random.default_rng(2)
    X = rng.uniform(-1, 1, (1000, 75)).astype(np.float32)
    idx = rng.permutation(1000)[:700].astype(np.int32)
    dX, dI = dev(X, device), dev(idx, device)
    out = torch.ones(40, device=device)
    out2 = torch.zeros(40, device=device)
    L.check(lib.pp_colsum_f32(dX.data_ptr() + 4 * 5, 75, dI.data_ptr(), 700, 40, out.data_ptr(), out2.data_ptr(),
                              L.stream_ptr()))
    ref = X[idx, 5:45].astype(np.float64).sum(0)
    np.testing.assert_allclose(out.cpu().numpy(), ref + 1, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(out2.cpu().numpy(), ref, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('n,H', [(37, 64), (1024, 512)])
def test_lstm_cell_forward_backward(env, n, H):
    L, lib, device = env
    rng = np.random.default_rng(n)
    G = rng.normal(0, 1.5, (n, 4 * H)).astype(np.float32)
    c_prev = rng.normal(0, 1, (n, H)).astype(np.float32)
    dG, dcp = dev(G, device), dev(c_prev, device)
    c = torch.empty(n, H, device=device)
    h = torch.empty(n, H, device=device)
    L.check(lib.pp_lstm_cell_fwd(dG.data_ptr(), dcp.data_ptr(), c.data_ptr(), h.data_ptr(), n, H, L.stream_ptr()))
    g64 = G.astype(np.float64)
    i, f = O.sigmoid(g64[:, :H]), O.sigmoid(g64[:, H:2 * H])
    gg, o = np.tanh(g64[:, 2 * H:3 * H]), O.sigmoid(g64[:, 3 * H:])
    c_ref = f * c_prev + i * gg
    h_ref = o * np.tanh(c_ref)
    np.testing.assert_allclose(c.cpu().numpy(), c_ref, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(h.cpu().numpy(), h_ref, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(dG.cpu().numpy(), np.concatenate([i, f, gg, o], 1), rtol=1e-5, atol=2e-6)
    # backward: n_next < n rows receive a carried dc
    n_next = n // 2
    dh = rng.normal(0, 1, (n, H)).astype(np.float32)
    dc_in = rng.normal(0, 1, (n, H)).astype(np.float32)
    ddh, ddc = dev(dh, device), dev(dc_in, device)
    L.check(lib.pp_lstm_cell_bwd(dG.data_ptr(), dcp.data_ptr(), c.data_ptr(), ddh.data_ptr(), ddc.data_ptr(), n, n_next, H,
                                 L.stream_ptr()))
    tc = np.tanh(c_ref)
    carry = np.where(np.arange(n)[:, None] < n_next, dc_in, 0.0)
    dc = carry + dh * o * (1 - tc * tc)
    ref = np.concatenate([dc * gg * i * (1 - i), dc * c_prev * f * (1 - f), dc * i * (1 - gg * gg), dh * tc * o * (1 - o)], 1)
    np.testing.assert_allclose(dG.cpu().numpy(), ref, rtol=2e-5, atol=5e-6)
    np.testing.assert_allclose(ddc.cpu().numpy(), dc * f, rtol=2e-5, atol=5e-6)
    # first step of a trace: c_prev = NULL means zeros
    dG2 = dev(G, device)
    L.check(lib.pp_lstm_cell_fwd(dG2.data_ptr(), None, c.data_ptr(), h.data_ptr(), n, H, L.stream_ptr()))
    np.testing.assert_allclose(c.cpu().numpy(), i * gg, rtol=1e-5, atol=2e-6)
    # ... and the forget gate is not read at all (pp_ic_loss does not even compute its pre-activation for those rows):
    # garbage there changes nothing, the recorded gate is 0 so that the backward pass gives it a zero gradient
    G3 = G.copy()
    G3[:, H:2 * H] = np.nan
    dG3 = dev(G3, device)
    L.check(lib.pp_lstm_cell_fwd(dG3.data_ptr(), None, c.data_ptr(), h.data_ptr(), n, H, L.stream_ptr()))
    np.testing.assert_allclose(c.cpu().numpy(), i * gg, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(h.cpu().numpy(), o * np.tanh(i * gg), rtol=1e-5, atol=2e-6)
    assert float(dG3[:, H:2 * H].abs().max()) == 0.0
    L.check(lib.pp_lstm_cell_bwd(dG3.data_ptr(), None, c.data_ptr(), ddh.data_ptr(), ddc.data_ptr(), n, 0, H, L.stream_ptr()))
    assert torch.isfinite(dG3).all() and float(dG3[:, H:2 * H].abs().max()) == 0.0


def _head_case(kind, n, K, rng):
    if kind == 2:
        Cn = 7
        y = rng.normal(0, 2, (n, Cn))
        v = rng.integers(0, Cn, n).astype(np.float64)
        prior = np.zeros((n, 2))
        lp, dy, _ = O.head_categorical(y, v)
        return y, v, prior, lp, dy, Cn
    y = rng.normal(0, 1, (n, 3 * K))
    if kind == 0:
        prior = np.stack([rng.normal(0, 2, n), rng.uniform(0.5, 3, n)], 1)
        v = prior[:, 0] + prior[:, 1] * rng.normal(0, 1, n)
        lp, dy, _ = O.head_normal_mixture(y, prior, v, K)
    else:
        low = rng.uniform(-3, 0, n)
        prior = np.stack([low, low + rng.uniform(0.5, 4, n)], 1)
        v = rng.uniform(prior[:, 0], prior[:, 1])
        v[::17] = prior[::17, 1] + 1.0        # outside the support: -inf -> log(1e-8), zero gradient
        lp, dy, _ = O.head_truncated_normal_mixture(y, prior, v, K)
    return y, v, prior, lp, dy, 3 * K


@pytest.mark.parametrize('kind', [0, 1, 2])
@pytest.mark.parametrize('K', [10, 3, 16])
def test_head_logprob_and_gradient(env, kind, K):
    L, lib, device = env
    rng = np.random.default_rng(kind * 10 + K)
    n = 777
    y, v, prior, lp_ref, dy_ref, n_out = _head_case(kind, n, K, rng)
    ldy = ((n_out + 3) // 4) * 4
    Y = np.zeros((n, ldy), np.float32)
    Y[:, :n_out] = y
    rows = rng.permutation(n).astype(np.int32)      # row i of Y belongs to trace row rows[i]
    value = np.zeros(n, np.float32); value[rows] = v
    pr = np.zeros((n, 2), np.float32); pr[rows] = prior
    dY, dR, dV, dP = dev(Y, device), dev(rows, device), dev(value, device), dev(pr, device)
    lp = torch.zeros(n, device=device)
    dyo = torch.zeros(n, ldy, device=device)
    acc = torch.zeros(1, device=device)
    flag = torch.zeros(1, dtype=torch.int32, device=device)
    gs = -1.0 / 64
    L.check(lib.pp_head_logprob(kind, dY.data_ptr(), ldy, dR.data_ptr(), dV.data_ptr(), dP.data_ptr(), n, n_out, gs,
                                lp.data_ptr(), dyo.data_ptr(), acc.data_ptr(), flag.data_ptr(), L.stream_ptr()))
    got_lp = lp.cpu().numpy()[rows]
    # the oracle works on the fp32-rounded inputs too
    y32, v32, p32 = Y[:, :n_out].astype(np.float64), value[rows].astype(np.float64), pr[rows].astype(np.float64)
    if kind == 0:
        lp_ref, dy_ref, _ = O.head_normal_mixture(y32, p32, v32, K)
    elif kind == 1:
        lp_ref, dy_ref, _ = O.head_truncated_normal_mixture(y32, p32, v32, K)
    else:
        lp_ref, dy_ref, _ = O.head_categorical(y32, v32)
    fin = np.isfinite(lp_ref)
    assert np.array_equal(np.isneginf(got_lp), np.isneginf(lp_ref))
    np.testing.assert_allclose(got_lp[fin], lp_ref[fin], rtol=1e-4, atol=2e-5)
    dy_ref = np.where(fin[:, None], dy_ref, 0.0) * gs
    scale = np.abs(dy_ref).max()
    assert np.abs(dyo.cpu().numpy()[:, :n_out] - dy_ref).max() / scale < 2e-4
    loss_ref = -(np.where(fin, lp_ref, O.LOG_EPSILON)).sum()
    assert abs(float(acc.item()) - loss_ref) / abs(loss_ref) < 1e-5
    assert int(flag.item()) == 0


def test_head_rejects_too_many_components(env):
    L, lib, device = env
    y = torch.zeros(4, 64, device=device)
    rc = lib.pp_head_logprob(0, y.data_ptr(), 64, None, y.data_ptr(), y.data_ptr(), 4, 51, 1.0, None, None, None, None,
                             L.stream_ptr())
    assert rc == -1 and b'components' in lib.pp_last_error()
