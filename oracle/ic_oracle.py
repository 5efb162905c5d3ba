"""CPU ORACLE (test infrastructure, NOT product code) for the pyprob inference-compilation hot path.

A plain-numpy restatement of the reference algorithm (pyprob v1.5.0, /root/reference), written to mirror the
REFERENCE's structure (per-sub-batch loops, string-keyed layers), not the MI355X product's packed layout.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(pyprob_amd/) never does and fails loudly without its HIP library.

Parity status: PINNED. tests/test_oracle.py checks this file against
  * the known-answer log_probs of the reference's own tests (tests/test_distributions.py:1190,1223,1326,1363,
    1439,1474,1503,2101,2137), and
  * golden vectors produced by running the reference itself (tests/golden/make_golden.py): loss, LSTM
    input/output, per-row proposal log_prob, every parameter gradient, and importance-sampling log-weights
    (nine recorded programs / network shapes; 10 000 reference particles per program, tests/golden/make_is_10k.py;
    optimizer trajectories of torch.optim.Adam / SGD and the reference's LARC class, tests/golden/make_optim_golden.py);
and, where /root/reference exists (the build container), tests/test_binding_reference.py runs the reference next to it on
freshly initialised networks of other shapes.

Each function cites the reference lines it restates. Arithmetic is float64 by default (the reference is fp32;
agreement is to fp32 round-off), or float32 with dtype=np.float32.

Part of the arithmetic lives in a third-party dependency that is not under /root/reference: PyTorch
(setup.py:33 `torch>=1.5.1`; this container: 2.10.0). nn.Linear, nn.LSTM (gate order i,f,g,o),
torch.softmax/logsumexp, torch.distributions.Normal/Categorical are restated from their documented formulas.
"""
import math

import numpy as np
from scipy.special import erf as _erf

EPSILON = 1e-8                      # pyprob/util.py:34
LOG_EPSILON = math.log(EPSILON)     # pyprob/util.py:35
FP32_EPS = float(np.finfo(np.float32).eps)  # util.clamp_probs, pyprob/util.py:393-395
HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


# ------------------------------------------------------------------------------------------------
# distributions (pyprob/distributions/*.py -> torch.distributions)
# ------------------------------------------------------------------------------------------------
def normal_log_prob(v, mu, sd):
    """torch.distributions.Normal.log_prob via pyprob/distributions/normal.py:11, distribution.py:38-43."""
    return -((v - mu) ** 2) / (2.0 * sd * sd) - np.log(sd) - HALF_LOG_2PI


def std_normal_cdf(x):
    """Normal(0,1).cdf, pyprob/distributions/normal.py:23-24 (torch: 0.5*(1+erf(x/sqrt(2))))."""
    return 0.5 * (1.0 + _erf(x / math.sqrt(2.0)))


def std_normal_pdf(x):
    return np.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


def truncated_normal_log_prob(v, mu, sd, low, high):
    """pyprob/distributions/truncated_normal.py:25-30 (constructor) and :40-54 (log_prob)."""
    alpha = (low - mu) / sd
    beta = (high - mu) / sd
    Z = std_normal_cdf(beta) - std_normal_cdf(alpha)
    with np.errstate(divide='ignore'):
        inside = np.log(np.asarray((v >= low) & (v <= high), np.float64))
        return inside + normal_log_prob((v - mu) / sd, 0.0, 1.0) - np.log(sd * Z)


def uniform_log_prob(v, low, high):
    """torch.distributions.Uniform.log_prob via pyprob/distributions/uniform.py:11 (support [low, high))."""
    with np.errstate(divide='ignore'):
        inside = np.log(np.asarray((v >= low) & (v < high), np.float64))
    return inside - np.log(high - low)


def mixture_log_probs(probs):
    """Mixture.__init__ normalisation and clamp: pyprob/distributions/mixture.py:14-16."""
    p = probs / probs.sum(-1, keepdims=True)
    return p, np.log(np.clip(p, FP32_EPS, 1.0 - FP32_EPS))


def logsumexp(a, axis=-1):
    m = np.max(a, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    with np.errstate(divide='ignore'):
        return (m + np.log(np.sum(np.exp(a - m), axis=axis, keepdims=True))).squeeze(axis)


def mixture_log_prob(comp_lp, probs):
    """Mixture.log_prob, batched branch: pyprob/distributions/mixture.py:42-44. comp_lp, probs: [B,K]."""
    _, logp = mixture_log_probs(probs)
    return logsumexp(logp + comp_lp, axis=-1)


def categorical_log_prob(index, probs):
    """torch.distributions.Categorical(probs=...).log_prob via pyprob/distributions/categorical.py:17:
    probs are normalised, logits = log(clamp(p, eps, 1-eps)), log_prob gathers the logit."""
    p = probs / probs.sum(-1, keepdims=True)
    logits = np.log(np.clip(p, FP32_EPS, 1.0 - FP32_EPS))
    index = np.asarray(index).astype(np.int64)
    if logits.ndim == 1:
        return logits[index]
    return logits[np.arange(logits.shape[0]), index]


def softmax(z):
    z = z - z.max(-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(-1, keepdims=True)


def sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


# ------------------------------------------------------------------------------------------------
# network description
# ------------------------------------------------------------------------------------------------
class Net:
    """String-keyed view of an InferenceNetworkLSTM state_dict (names as in SURVEY.md Appendix B).

    params: dict name -> ndarray (reference state_dict names)
    addresses / dist_names: per address index, as in the golden batch meta
    obs_names: order of `_layers_observe_embedding` (insertion order, inference_network.py:134)
    """

    def __init__(self, params, obs_names, K=10, dtype=np.float64):
        self.dtype = dtype
        self.P = {k: np.asarray(v, dtype) for k, v in params.items()}
        self.obs_names = list(obs_names)
        self.K = K
        # (InferenceNetworkFeedForward has no LSTM: its heads read the observe embedding)
        self.H = self.P['_layers_lstm.weight_hh_l0'].shape[1] if '_layers_lstm.weight_hh_l0' in self.P else 0
        self.depth = sum(1 for k in self.P if k.startswith('_layers_lstm.weight_hh_l'))     # nn.LSTM(I, H, depth)

    def lstm_layer(self, k):
        """(W_ih, W_hh, b_ih, b_hh) of layer k (torch.nn.LSTM parameter names weight_ih_l<k> ...)."""
        return tuple(self.P['_layers_lstm.%s_l%d' % (n, k)] for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'))

    def ff(self, prefix):
        """(W list, b list) of an EmbeddingFeedForward stored under `prefix`._layers.N.{weight,bias}."""
        Ws, bs, i = [], [], 0
        while '%s._layers.%d.weight' % (prefix, i) in self.P:
            Ws.append(self.P['%s._layers.%d.weight' % (prefix, i)])
            bs.append(self.P['%s._layers.%d.bias' % (prefix, i)])
            i += 1
        return Ws, bs


def ff_forward(x, Ws, bs, relu_last):
    """EmbeddingFeedForward.forward, pyprob/nn/embedding_feedforward.py:35-48. Returns output and the
    per-layer inputs/outputs needed by ff_backward."""
    acts = [x]
    for i, (W, b) in enumerate(zip(Ws, bs)):
        x = x @ W.T + b
        if i < len(Ws) - 1 or relu_last:
            x = np.maximum(x, 0.0)
        acts.append(x)
    return x, acts


def ff_backward(dy, acts, Ws, relu_last):
    """Gradients of ff_forward: returns dx, [dW], [db]."""
    n = len(Ws)
    dWs, dbs = [None] * n, [None] * n
    for i in reversed(range(n)):
        if i < n - 1 or relu_last:
            dy = dy * (acts[i + 1] > 0)
        dWs[i] = dy.T @ acts[i]
        dbs[i] = dy.sum(0)
        dy = dy @ Ws[i]
    return dy, dWs, dbs


def embed_observe(net, obs):
    """InferenceNetwork._embed_observe, pyprob/nn/inference_network.py:132-139. obs: [B, n_obs] scalars."""
    parts, caches = [], []
    for j, name in enumerate(net.obs_names):
        Ws, bs = net.ff('_layers_observe_embedding.' + name)
        y, acts = ff_forward(obs[:, j:j + 1].astype(net.dtype), Ws, bs, True)
        parts.append(y)
        caches.append(acts)
    cat = np.concatenate(parts, axis=1)
    Ws, bs = net.ff('_layers_observe_embedding_final')
    E, acts = ff_forward(cat, Ws, bs, True)
    return E, (caches, acts)


def sample_embedding(net, address, dist_name, values, num_categories=None):
    """`_layers_sample_embedding[address](value)`: one Linear + ReLU (inference_network_lstm.py:54,169);
    one-hot input for Categorical (:63, embedding_feedforward.py:36-37)."""
    Ws, bs = net.ff('_layers_sample_embedding.' + address)
    if dist_name == 'Categorical':
        C = Ws[0].shape[1]
        x = np.zeros((len(values), C), net.dtype)
        x[np.arange(len(values)), np.asarray(values).astype(np.int64)] = 1.0
    else:
        x = np.asarray(values, net.dtype).reshape(-1, 1)
    y, acts = ff_forward(x, Ws, bs, True)
    return y, acts


def lstm_forward(x, W_ih, W_hh, b_ih, b_hh, h0=None, c0=None):
    """torch.nn.LSTM, 1 layer, as called at inference_network_lstm.py:185-188 (h0=c0=0) and :123.
    x: [T,B,I]. Gate order i,f,g,o. Returns h[T,B,H] and caches."""
    T, B, _ = x.shape
    H = W_hh.shape[1]
    h = np.zeros((B, H), x.dtype) if h0 is None else h0
    c = np.zeros((B, H), x.dtype) if c0 is None else c0
    hs, cache = [], []
    for t in range(T):
        g = x[t] @ W_ih.T + b_ih + h @ W_hh.T + b_hh
        i = sigmoid(g[:, :H])
        f = sigmoid(g[:, H:2 * H])
        gg = np.tanh(g[:, 2 * H:3 * H])
        o = sigmoid(g[:, 3 * H:])
        c_new = f * c + i * gg
        tc = np.tanh(c_new)
        h_new = o * tc
        cache.append((h, c, i, f, gg, o, tc))
        h, c = h_new, c_new
        hs.append(h)
    return np.stack(hs), cache, (h, c)


def lstm_backward(dh_seq, x, cache, W_ih, W_hh):
    """Gradients of lstm_forward w.r.t. x and the four parameter tensors, given dL/dh_t for every t."""
    T, B, _ = x.shape
    H = W_hh.shape[1]
    dW_ih = np.zeros_like(W_ih)
    dW_hh = np.zeros_like(W_hh)
    db = np.zeros(4 * H, x.dtype)
    dx = np.zeros_like(x)
    dh_next = np.zeros((B, H), x.dtype)
    dc_next = np.zeros((B, H), x.dtype)
    for t in reversed(range(T)):
        h_prev, c_prev, i, f, gg, o, tc = cache[t]
        dh = dh_seq[t] + dh_next
        do = dh * tc
        dc = dc_next + dh * o * (1.0 - tc * tc)
        di = dc * gg
        dgg = dc * i
        df = dc * c_prev
        dc_next = dc * f
        dG = np.concatenate([di * i * (1 - i), df * f * (1 - f), dgg * (1 - gg * gg), do * o * (1 - o)], axis=1)
        dW_ih += dG.T @ x[t]
        dW_hh += dG.T @ h_prev
        db += dG.sum(0)
        dx[t] = dG @ W_ih
        dh_next = dG @ W_hh
    return dx, dW_ih, dW_hh, db


# ------------------------------------------------------------------------------------------------
# proposal heads: forward to log_prob, and d log_prob / d(head output y)
# ------------------------------------------------------------------------------------------------
def head_normal_mixture(y, prior, v, K):
    """ProposalNormalNormalMixture.forward (pyprob/nn/proposal_normal_normal_mixture.py:18-35) followed by
    Mixture.log_prob. y: [B,3K] FF output, prior: [B,2] (mean, stddev), v: [B]. Returns lp[B], dlp/dy[B,3K],
    and the proposal parameters (mu, sd, probs)."""
    mu_p, sd_p = prior[:, 0:1], prior[:, 1:2]
    mu = mu_p + y[:, :K] * sd_p
    sd = np.exp(y[:, K:2 * K]) * sd_p
    pi = softmax(y[:, 2 * K:])
    comp = normal_log_prob(v[:, None], mu, sd)
    p, logp = mixture_log_probs(pi)
    a = logp + comp
    lp = logsumexp(a, axis=1)
    with np.errstate(invalid='ignore'):
        r = np.exp(a - lp[:, None])                        # responsibilities = d lp / d a_k
    t = (v[:, None] - mu) / sd
    d_mu = r * t / sd
    d_sd = r * (t * t - 1.0) / sd
    dy = np.zeros_like(y)
    dy[:, :K] = d_mu * sd_p
    dy[:, K:2 * K] = d_sd * sd
    dy[:, 2 * K:] = _mixture_logit_grad(r, pi, p)
    return lp, dy, (mu, sd, p)


def _mixture_logit_grad(r, pi, p):
    """d lp / d logits through log(clamp(p)) with p = pi/sum(pi), pi = softmax(logits)."""
    inside = (p >= FP32_EPS) & (p <= 1.0 - FP32_EPS)
    dp = np.where(inside, r / p, 0.0)
    S = pi.sum(-1, keepdims=True)
    dpi = dp / S - (dp * p).sum(-1, keepdims=True) / S
    return pi * (dpi - (dpi * pi).sum(-1, keepdims=True))


def head_truncated_normal_mixture(y, prior, v, K):
    """ProposalUniformTruncatedNormalMixture.forward (pyprob/nn/proposal_uniform_truncated_normal_mixture.py:18-36)
    followed by Mixture.log_prob over TruncatedNormal components. prior: [B,2] (low, high)."""
    low, high = prior[:, 0:1], prior[:, 1:2]
    rng = high - low
    sm = sigmoid(y[:, :K])
    ss = sigmoid(y[:, K:2 * K])
    mu = low + sm * rng
    sd = rng / 1000 + ss * rng * 10
    pi = softmax(y[:, 2 * K:])
    comp = truncated_normal_log_prob(v[:, None], mu, sd, low, high)
    p, logp = mixture_log_probs(pi)
    a = logp + comp
    lp = logsumexp(a, axis=1)
    with np.errstate(invalid='ignore'):
        r = np.exp(a - lp[:, None])
    r = np.where(np.isfinite(lp)[:, None], r, 0.0)
    alpha = (low - mu) / sd
    beta = (high - mu) / sd
    Z = std_normal_cdf(beta) - std_normal_cdf(alpha)
    pa, pb = std_normal_pdf(alpha), std_normal_pdf(beta)
    t = (v[:, None] - mu) / sd
    d_mu = r * (t / sd - (pa - pb) / (sd * Z))
    d_sd = r * ((t * t - 1.0) / sd - (alpha * pa - beta * pb) / (sd * Z))
    dy = np.zeros_like(y)
    dy[:, :K] = d_mu * rng * sm * (1 - sm)
    dy[:, K:2 * K] = d_sd * rng * 10 * ss * (1 - ss)
    dy[:, 2 * K:] = _mixture_logit_grad(r, pi, p)
    return lp, dy, (mu, sd, p)


POISSON_LOW, POISSON_HIGH = 0.0, 40.0   # ProposalPoissonTruncatedNormalMixture defaults (low=0, high=40)


def head_poisson_truncated_normal_mixture(y, v, K, low=POISSON_LOW, high=POISSON_HIGH):
    """ProposalPoissonTruncatedNormalMixture.forward (pyprob/nn/proposal_poisson_truncated_normal_mixture.py:19-37)
    followed by Mixture.log_prob: means = low + sigmoid(y) (high - low), stddevs = exp(y) (not scaled), TruncatedNormal
    components on the FIXED interval [low, high] - the prior's rate does not enter the proposal."""
    rng = high - low
    sm = sigmoid(y[:, :K])
    mu = low + sm * rng
    sd = np.exp(y[:, K:2 * K])
    pi = softmax(y[:, 2 * K:])
    lo = np.full((y.shape[0], 1), low, y.dtype)
    hi = np.full((y.shape[0], 1), high, y.dtype)
    comp = truncated_normal_log_prob(v[:, None], mu, sd, lo, hi)
    p, logp = mixture_log_probs(pi)
    a = logp + comp
    lp = logsumexp(a, axis=1)
    with np.errstate(invalid='ignore'):
        r = np.exp(a - lp[:, None])
    r = np.where(np.isfinite(lp)[:, None], r, 0.0)
    alpha = (lo - mu) / sd
    beta = (hi - mu) / sd
    Z = std_normal_cdf(beta) - std_normal_cdf(alpha)
    pa, pb = std_normal_pdf(alpha), std_normal_pdf(beta)
    t = (v[:, None] - mu) / sd
    d_mu = r * (t / sd - (pa - pb) / (sd * Z))
    d_sd = r * ((t * t - 1.0) / sd - (alpha * pa - beta * pb) / (sd * Z))
    dy = np.zeros_like(y)
    dy[:, :K] = d_mu * rng * sm * (1 - sm)
    dy[:, K:2 * K] = d_sd * sd
    dy[:, 2 * K:] = _mixture_logit_grad(r, pi, p)
    return lp, dy, (mu, sd, p)


def poisson_log_prob(v, rate):
    """torch.distributions.Poisson.log_prob: v log(rate) - rate - lgamma(v + 1) (defined for non-integer v too)."""
    from math import lgamma
    v = np.asarray(v, np.float64)
    lg = np.vectorize(lgamma)(v + 1.0)
    return v * np.log(rate) - rate - lg


def head_categorical(y, v):
    """ProposalCategoricalCategorical.forward (pyprob/nn/proposal_categorical_categorical.py:16-20) + log_prob."""
    pi = softmax(y)
    q = pi + EPSILON
    S = q.sum(-1, keepdims=True)
    p = q / S
    idx = np.asarray(v).astype(np.int64)
    rows = np.arange(y.shape[0])
    lp = np.log(np.clip(p, FP32_EPS, 1 - FP32_EPS))[rows, idx]
    dp = np.zeros_like(p)
    pv = p[rows, idx]
    dp[rows, idx] = np.where((pv >= FP32_EPS) & (pv <= 1 - FP32_EPS), 1.0 / pv, 0.0)
    dq = dp / S - (dp * p).sum(-1, keepdims=True) / S
    dy = pi * (dq - (dq * pi).sum(-1, keepdims=True))
    return lp, dy, (p,)


def head_bernoulli(y, v):
    """ProposalBernoulliBernoulli.forward (pyprob/nn/proposal_bernoulli_bernoulli.py:16-20): probs = sigmoid(y) + 1e-8
    with shape [n, 1]. `_loss` then calls Bernoulli.log_prob on the stacked values of shape [n]
    (inference_network_lstm.py:195-202): torch broadcasts [n, 1] against [n] to an [n, n] MATRIX - every trace's proposal
    scored against every value of the sub-batch step - and `-torch.sum(log_prob)` adds all n^2 entries. Restated as the
    reference computes it: returns the row sums lp[i] = sum_j log Bernoulli(v_j; p_i), d lp[i] / d y[i], and the matrix.
    torch's Bernoulli clamps probs to [eps, 1 - eps] when turning them into logits (torch/distributions/utils.py)."""
    y = np.asarray(y)
    v = np.asarray(v, y.dtype).reshape(-1)
    sg = sigmoid(y[:, 0])
    p = sg + EPSILON
    inside = (p >= FP32_EPS) & (p <= 1 - FP32_EPS)
    pc = np.clip(p, FP32_EPS, 1 - FP32_EPS)
    mat = v[None, :] * np.log(pc)[:, None] + (1 - v)[None, :] * np.log1p(-pc)[:, None]
    n, n1 = float(len(v)), float(v.sum())
    dp = np.where(inside, n1 / pc - (n - n1) / (1 - pc), 0.0)
    dy = (dp * sg * (1 - sg))[:, None]
    return mat.sum(1), dy, (p[:, None], mat)


def bernoulli_log_prob(v, probs):
    """torch.distributions.Bernoulli(probs).log_prob(v) (probs clamped to [eps, 1 - eps])."""
    pc = np.clip(np.asarray(probs, np.float64), FP32_EPS, 1 - FP32_EPS)
    v = np.asarray(v, np.float64)
    return v * np.log(pc) + (1 - v) * np.log1p(-pc)


def head_forward(net, address, dist_name, h, prior, v, y=None):
    """`_layers_proposal[address].forward(h, variables)` + `.log_prob(values)` (inference_network_lstm.py:197-202).
    Returns lp, caches for backward, proposal params. (y: the layer's outputs when the caller already has them - rows that
    share one hidden state.)"""
    Ws, bs = net.ff('_layers_proposal.%s._ff' % address)
    if y is None:
        y, acts = ff_forward(h, Ws, bs, False)
    else:
        acts = None
    if dist_name == 'Normal':
        lp, dy, params = head_normal_mixture(y, prior, v, net.K)
    elif dist_name == 'Uniform':
        lp, dy, params = head_truncated_normal_mixture(y, prior, v, net.K)
    elif dist_name == 'Categorical':
        lp, dy, params = head_categorical(y, v)
    elif dist_name == 'Poisson':
        lp, dy, params = head_poisson_truncated_normal_mixture(y, v, net.K)
    elif dist_name == 'Bernoulli':
        lp, dy, params = head_bernoulli(y, v)
    else:
        raise RuntimeError('Distribution currently unsupported: ' + dist_name)
    return lp, (dy, acts, Ws), params


# ------------------------------------------------------------------------------------------------
# the training loss: InferenceNetworkLSTM._loss, pyprob/nn/inference_network_lstm.py:136-220
# ------------------------------------------------------------------------------------------------
def split_sub_batches(trace_len, addr_idx):
    """Batch.__init__ grouping by address sequence, pyprob/nn/dataset.py:21-37. Returns (list of lists of trace
    indices in first-seen order, row offset of every trace in the trace-major ragged arrays)."""
    off = np.concatenate([[0], np.cumsum(trace_len)]).astype(np.int64)
    groups = {}
    for b in range(len(trace_len)):
        if trace_len[b] == 0:
            raise ValueError('Trace of length zero.')
        key = tuple(int(a) for a in addr_idx[off[b]:off[b + 1]])
        groups.setdefault(key, []).append(b)
    return list(groups.values()), off


def loss_and_grads(net, batch, addresses, dist_names, want_grads=True):
    """Forward (and optionally backward) of `_loss(batch)`.

    batch: dict with trace-major ragged arrays trace_len[B], addr_idx[R], values[R], prior[R,>=2], obs[B,n_obs]
    Returns dict(loss, lstm_in, lstm_out, lp (list per (sub-batch, t)), grads (name -> ndarray), sub_batches).
    """
    dt = net.dtype
    P = net.P
    Ea, Ed = 0, 0
    for k in P:
        if k.startswith('_layers_address_embedding.'):
            Ea = P[k].shape[0]
        if k.startswith('_layers_distribution_type_embedding.'):
            Ed = P[k].shape[0]
    trace_len = np.asarray(batch['trace_len'])
    addr_idx = np.asarray(batch['addr_idx'])
    values = np.asarray(batch['values'], dt)
    prior = np.asarray(batch['prior'], dt)
    obs = np.asarray(batch['obs'], dt)
    B = len(trace_len)
    subs, off = split_sub_batches(trace_len, addr_idx)
    W_ih, W_hh = P['_layers_lstm.weight_ih_l0'], P['_layers_lstm.weight_hh_l0']
    b_ih, b_hh = P['_layers_lstm.bias_ih_l0'], P['_layers_lstm.bias_hh_l0']
    grads = {k: np.zeros_like(v) for k, v in P.items()} if want_grads else None
    out = dict(lstm_in=[], lstm_out=[], lp=[], sub_batches=subs)
    total = 0.0
    for sb in subs:
        sb = np.asarray(sb)
        n = len(sb)
        T = int(trace_len[sb[0]])
        rows = off[sb][None, :] + np.arange(T)[:, None]             # [T, n] row index of (t, trace)
        seq = [int(a) for a in addr_idx[off[sb[0]]:off[sb[0] + 1]]]
        E, obs_cache = embed_observe(net, obs[sb])
        S_emb = next(iter(net.ff('_layers_sample_embedding.' + addresses[seq[0]])[0])).shape[0]
        x = np.zeros((T, n, W_ih.shape[1]), dt)
        smp_caches = [None] * T
        for t in range(T):
            a_cur = addresses[seq[t]]
            d_cur = dist_names[seq[t]]
            col = E.shape[1]
            x[t, :, :col] = E
            if t > 0:
                a_prev, d_prev = addresses[seq[t - 1]], dist_names[seq[t - 1]]
                s, acts = sample_embedding(net, a_prev, d_prev, values[rows[t - 1]])
                smp_caches[t] = acts
                x[t, :, col:col + S_emb] = s
                x[t, :, col + S_emb:col + S_emb + Ed] = P['_layers_distribution_type_embedding.' + d_prev]
                x[t, :, col + S_emb + Ed:col + S_emb + Ed + Ea] = P['_layers_address_embedding.' + a_prev]
            c2 = col + S_emb + Ed + Ea
            x[t, :, c2:c2 + Ed] = P['_layers_distribution_type_embedding.' + d_cur]
            x[t, :, c2 + Ed:c2 + Ed + Ea] = P['_layers_address_embedding.' + a_cur]
        # nn.LSTM(I, H, depth) (inference_network_lstm.py:31): layer k reads the hidden states of layer k - 1
        layer_in, layer_cache = [x], []
        for k in range(net.depth):
            hs, cache, _ = lstm_forward(layer_in[k], *net.lstm_layer(k))
            layer_cache.append(cache)
            layer_in.append(hs)
        out['lstm_in'].append(x)
        out['lstm_out'].append(hs)
        dh_seq = np.zeros_like(hs)
        for t in range(T):
            a_cur, d_cur = addresses[seq[t]], dist_names[seq[t]]
            lp, (dy, acts, Ws), _ = head_forward(net, a_cur, d_cur, hs[t], prior[rows[t]], values[rows[t]])
            out['lp'].append(lp.copy())
            # -inf -> log(1e-8) (inference_network_lstm.py:207-213, util.py:278-284); such rows carry no gradient
            neg_inf = np.isneginf(lp)
            lp = np.where(neg_inf, LOG_EPSILON, lp)
            total += -lp.sum()
            if want_grads:
                g = np.where(neg_inf, 0.0, -1.0 / B)[:, None]
                dyg = np.where(neg_inf[:, None], 0.0, dy) * g
                dh, dWs, dbs = ff_backward(dyg, acts, Ws, False)
                dh_seq[t] = dh
                for i in range(len(Ws)):
                    grads['_layers_proposal.%s._ff._layers.%d.weight' % (a_cur, i)] += dWs[i]
                    grads['_layers_proposal.%s._ff._layers.%d.bias' % (a_cur, i)] += dbs[i]
        if not want_grads:
            continue
        dx = dh_seq
        for k in reversed(range(net.depth)):
            Wk_ih, Wk_hh, _, _ = net.lstm_layer(k)
            dx, dW_ih, dW_hh, db = lstm_backward(dx, layer_in[k], layer_cache[k], Wk_ih, Wk_hh)
            grads['_layers_lstm.weight_ih_l%d' % k] += dW_ih
            grads['_layers_lstm.weight_hh_l%d' % k] += dW_hh
            grads['_layers_lstm.bias_ih_l%d' % k] += db
            grads['_layers_lstm.bias_hh_l%d' % k] += db
        col = E.shape[1]
        dE = dx[:, :, :col].sum(0)
        for t in range(T):
            a_cur, d_cur = addresses[seq[t]], dist_names[seq[t]]
            c2 = col + S_emb + Ed + Ea
            grads['_layers_distribution_type_embedding.' + d_cur] += dx[t, :, c2:c2 + Ed].sum(0)
            grads['_layers_address_embedding.' + a_cur] += dx[t, :, c2 + Ed:c2 + Ed + Ea].sum(0)
            if t > 0:
                a_prev, d_prev = addresses[seq[t - 1]], dist_names[seq[t - 1]]
                grads['_layers_distribution_type_embedding.' + d_prev] += dx[t, :, col + S_emb:col + S_emb + Ed].sum(0)
                grads['_layers_address_embedding.' + a_prev] += dx[t, :, col + S_emb + Ed:c2].sum(0)
                Ws, _ = net.ff('_layers_sample_embedding.' + a_prev)
                _, dWs, dbs = ff_backward(dx[t, :, col:col + S_emb], smp_caches[t], Ws, True)
                grads['_layers_sample_embedding.%s._layers.0.weight' % a_prev] += dWs[0]
                grads['_layers_sample_embedding.%s._layers.0.bias' % a_prev] += dbs[0]
        # observe embedding backward
        caches, acts_final = obs_cache
        Ws, _ = net.ff('_layers_observe_embedding_final')
        dcat, dWs, dbs = ff_backward(dE, acts_final, Ws, True)
        for i in range(len(Ws)):
            grads['_layers_observe_embedding_final._layers.%d.weight' % i] += dWs[i]
            grads['_layers_observe_embedding_final._layers.%d.bias' % i] += dbs[i]
        c = 0
        for j, name in enumerate(net.obs_names):
            Ws, _ = net.ff('_layers_observe_embedding.' + name)
            w = Ws[-1].shape[0]
            _, dWs, dbs = ff_backward(dcat[:, c:c + w], caches[j], Ws, True)
            c += w
            for i in range(len(Ws)):
                grads['_layers_observe_embedding.%s._layers.%d.weight' % (name, i)] += dWs[i]
                grads['_layers_observe_embedding.%s._layers.%d.bias' % (name, i)] += dbs[i]
    out['loss'] = total / B
    out['grads'] = grads
    return out


# ------------------------------------------------------------------------------------------------
# InferenceNetworkFeedForward: pyprob/nn/inference_network_feedforward.py
# ------------------------------------------------------------------------------------------------
def loss_and_grads_feedforward(net, batch, addresses, dist_names, want_grads=True):
    """InferenceNetworkFeedForward._loss, pyprob/nn/inference_network_feedforward.py:68-98: per sub-batch the observe
    embedding is computed once (:72) and EVERY time step's proposal layer reads it (:85); no LSTM, no address / sample
    embeddings. Same rescue of -inf rows (:87-96) and the same division by batch.size (:98). Returns
    dict(loss, lp (list per (sub-batch, t)), grads, sub_batches)."""
    dt = net.dtype
    P = net.P
    trace_len = np.asarray(batch['trace_len'])
    addr_idx = np.asarray(batch['addr_idx'])
    values = np.asarray(batch['values'], dt)
    prior = np.asarray(batch['prior'], dt)
    obs = np.asarray(batch['obs'], dt)
    B = len(trace_len)
    subs, off = split_sub_batches(trace_len, addr_idx)
    grads = {k: np.zeros_like(v) for k, v in P.items()} if want_grads else None
    out = dict(lp=[], sub_batches=subs)
    total = 0.0
    for sb in subs:
        sb = np.asarray(sb)
        T = int(trace_len[sb[0]])
        rows = off[sb][None, :] + np.arange(T)[:, None]
        seq = [int(a) for a in addr_idx[off[sb[0]]:off[sb[0] + 1]]]
        E, obs_cache = embed_observe(net, obs[sb])
        dE = np.zeros_like(E)
        for t in range(T):
            a_cur, d_cur = addresses[seq[t]], dist_names[seq[t]]
            lp, (dy, acts, Ws), _ = head_forward(net, a_cur, d_cur, E, prior[rows[t]], values[rows[t]])
            out['lp'].append(lp.copy())
            neg_inf = np.isneginf(lp)
            lp = np.where(neg_inf, LOG_EPSILON, lp)
            total += -lp.sum()
            if want_grads:
                g = np.where(neg_inf, 0.0, -1.0 / B)[:, None]
                dyg = np.where(neg_inf[:, None], 0.0, dy) * g
                dh, dWs, dbs = ff_backward(dyg, acts, Ws, False)
                dE += dh
                for i in range(len(Ws)):
                    grads['_layers_proposal.%s._ff._layers.%d.weight' % (a_cur, i)] += dWs[i]
                    grads['_layers_proposal.%s._ff._layers.%d.bias' % (a_cur, i)] += dbs[i]
        if not want_grads:
            continue
        caches, acts_final = obs_cache
        Ws, _ = net.ff('_layers_observe_embedding_final')
        dcat, dWs, dbs = ff_backward(dE, acts_final, Ws, True)
        for i in range(len(Ws)):
            grads['_layers_observe_embedding_final._layers.%d.weight' % i] += dWs[i]
            grads['_layers_observe_embedding_final._layers.%d.bias' % i] += dbs[i]
        c = 0
        for j, name in enumerate(net.obs_names):
            Ws, _ = net.ff('_layers_observe_embedding.' + name)
            w = Ws[-1].shape[0]
            _, dWs, dbs = ff_backward(dcat[:, c:c + w], caches[j], Ws, True)
            c += w
            for i in range(len(Ws)):
                grads['_layers_observe_embedding.%s._layers.%d.weight' % (name, i)] += dWs[i]
                grads['_layers_observe_embedding.%s._layers.%d.bias' % (name, i)] += dbs[i]
    out['loss'] = total / B
    out['grads'] = grads
    return out


def is_rescore_feedforward(net, observe, trace_len, addr_idx, values, prior, addresses, dist_names):
    """is_rescore for InferenceNetworkFeedForward._infer_step (inference_network_feedforward.py:52-66): the proposal of
    every controlled variable is its address's layer applied to the observe embedding of `_infer_init`."""
    dt = net.dtype
    E, _ = embed_observe(net, np.asarray(observe, dt).reshape(1, -1))
    off = np.concatenate([[0], np.cumsum(trace_len)]).astype(np.int64)
    R = int(off[-1])
    prior_lp, prop_lp, prop_params = np.zeros(R), np.zeros(R), []
    lw = np.zeros(len(trace_len))
    for b in range(len(trace_len)):
        for r in range(off[b], off[b + 1]):
            a_cur, d_cur = addresses[addr_idx[r]], dist_names[addr_idx[r]]
            v = np.asarray(values[r:r + 1], dt)
            q_lp, _, params = head_forward(net, a_cur, d_cur, E, np.asarray(prior[r:r + 1], dt), v)
            if d_cur == 'Categorical':
                C = params[0].shape[1]
                p_lp = categorical_log_prob(v, np.asarray(prior[r, :C], dt))
            else:
                p_lp = prior_log_prob(d_cur, np.asarray(prior[r], dt), v[0])
            prior_lp[r] = float(np.float32(np.asarray(p_lp).reshape(-1)[0]))
            prop_lp[r] = float(np.float32(q_lp[0]))
            prop_params.append(params)
            lw[b] += prior_lp[r] - prop_lp[r]
    return prior_lp, prop_lp, prop_params, lw


# ------------------------------------------------------------------------------------------------
# importance sampling: state.sample IC branch (pyprob/state.py:203-219), state.observe (:118-155),
# Trace.end (pyprob/trace.py:123-125), driven per particle by Model._traces (pyprob/model.py:59-71)
# ------------------------------------------------------------------------------------------------
def prior_log_prob(dist_name, prior, v):
    if dist_name == 'Normal':
        return normal_log_prob(v, prior[..., 0], prior[..., 1])
    if dist_name == 'Uniform':
        return uniform_log_prob(v, prior[..., 0], prior[..., 1])
    if dist_name == 'Categorical':
        return categorical_log_prob(v, prior)
    if dist_name == 'Poisson':
        return poisson_log_prob(v, prior[..., 0])
    if dist_name == 'Bernoulli':
        return bernoulli_log_prob(v, prior[..., 0])
    raise RuntimeError(dist_name)


def is_rescore(net, observe, trace_len, addr_idx, values, prior, addresses, dist_names):
    """Re-score reference-sampled particles: for every controlled variable run `_infer_step`
    (inference_network_lstm.py:82-134; LSTM state carried across the variables of ONE trace, reset when
    prev_variable is None) and return per-row (prior log_prob, proposal log_prob, proposal params) and the
    per-trace sum of (log p - log q) exactly as state.py:211-217 / trace.py:123-125 accumulate it
    (each term rounded to fp32 first, then summed in double)."""
    dt = net.dtype
    P = net.P
    Ea = P['_layers_address_embedding.' + addresses[0]].shape[0]
    Ed = P['_layers_distribution_type_embedding.' + dist_names[0]].shape[0]
    E, _ = embed_observe(net, np.asarray(observe, dt).reshape(1, -1))     # _infer_init, inference_network.py:141-148
    W_ih, W_hh = P['_layers_lstm.weight_ih_l0'], P['_layers_lstm.weight_hh_l0']
    b_ih, b_hh = P['_layers_lstm.bias_ih_l0'], P['_layers_lstm.bias_hh_l0']
    off = np.concatenate([[0], np.cumsum(trace_len)]).astype(np.int64)
    R = int(off[-1])
    prior_lp = np.zeros(R)
    prop_lp = np.zeros(R)
    prop_params = []
    lw = np.zeros(len(trace_len))
    for b in range(len(trace_len)):
        state = [(None, None)] * net.depth          # (h, c) of every layer, reset when prev_variable is None (:84-91)
        for t in range(int(trace_len[b])):
            r = off[b] + t
            a_cur, d_cur = addresses[addr_idx[r]], dist_names[addr_idx[r]]
            S_emb = net.ff('_layers_sample_embedding.' + a_cur)[0][0].shape[0]
            x = np.zeros((1, 1, W_ih.shape[1]), dt)
            col = E.shape[1]
            x[0, 0, :col] = E[0]
            if t > 0:
                a_prev, d_prev = addresses[addr_idx[r - 1]], dist_names[addr_idx[r - 1]]
                s, _ = sample_embedding(net, a_prev, d_prev, values[r - 1:r])
                x[0, 0, col:col + S_emb] = s[0]
                x[0, 0, col + S_emb:col + S_emb + Ed] = P['_layers_distribution_type_embedding.' + d_prev]
                x[0, 0, col + S_emb + Ed:col + S_emb + Ed + Ea] = P['_layers_address_embedding.' + a_prev]
            c2 = col + S_emb + Ed + Ea
            x[0, 0, c2:c2 + Ed] = P['_layers_distribution_type_embedding.' + d_cur]
            x[0, 0, c2 + Ed:c2 + Ed + Ea] = P['_layers_address_embedding.' + a_cur]
            hs = x
            for k in range(net.depth):
                hs, _, state[k] = lstm_forward(hs, *net.lstm_layer(k), *state[k])
            v = np.asarray(values[r:r + 1], dt)
            q_lp, _, params = head_forward(net, a_cur, d_cur, hs[0], np.asarray(prior[r:r + 1], dt), v)
            if d_cur == 'Categorical':
                C = params[0].shape[1]
                p_lp = categorical_log_prob(v, np.asarray(prior[r, :C], dt))
            else:
                p_lp = prior_log_prob(d_cur, np.asarray(prior[r], dt), v[0])
            prior_lp[r] = float(np.float32(np.asarray(p_lp).reshape(-1)[0]))
            prop_lp[r] = float(np.float32(q_lp[0]))
            prop_params.append(params)
            lw[b] += prior_lp[r] - prop_lp[r]
    return prior_lp, prop_lp, prop_params, lw


def is_rescore_lockstep(net, observe, steps, n, chunk=16384, return_state=False):
    """The same re-scoring as `is_rescore`, vectorised over PARTICLES: n traces advance statement by statement (the way the
    lock-step executor runs them), `_infer_step` (inference_network_lstm.py:82-134) is evaluated for all particles of a
    statement at once in `net.dtype`. One-layer LSTM. steps: list of dicts
        address, dist_name   the statement's address / distribution name
        values  [m]          the value of every particle that executes the statement
        prior   [m, P] or [1, P]
        rows    None (all n particles, m = n) or the m particle indices that execute it (a diverged control-flow path);
                every particle's statements must appear in its program order.
    Per particle: LSTM state reset at its first statement (prev_variable is None, :84-91), the previous statement's value /
    address / distribution feed the sample, address and type embeddings (:106-121). Returns per-step (prior_lp, prop_lp) and
    lw [n] = sum of fp32-rounded (log p - log q) per particle (state.py:211-217, trace.py:123-125) [, (h, c)]."""
    dt = net.dtype
    P = net.P
    assert net.depth == 1
    E, _ = embed_observe(net, np.asarray(observe, dt).reshape(1, -1))
    W_ih, W_hh, b_ih, b_hh = net.lstm_layer(0)
    H = W_hh.shape[1]
    h = np.zeros((n, H), dt)
    c = np.zeros((n, H), dt)
    prev_step = np.full(n, -1, np.int64)          # index of the particle's previous statement
    prev_val = np.zeros(n, dt)
    lw = np.zeros(n)
    out = []
    col = E.shape[1]
    for j, st in enumerate(steps):
        a_cur, d_cur = st['address'], st['dist_name']
        rows = np.arange(n) if st.get('rows') is None else np.asarray(st['rows'], np.int64)
        m = len(rows)
        values = np.asarray(st['values'], dt).reshape(-1)
        prior = np.asarray(st['prior'], dt).reshape(-1, np.asarray(st['prior']).shape[-1])
        if prior.shape[0] == 1:
            prior = np.broadcast_to(prior, (m, prior.shape[1]))
        assert values.shape[0] == m and prior.shape[0] == m
        ps = prev_step[rows]
        assert (ps == ps[0]).all(), 'the particles of a statement share their previous statement'
        Ed = P['_layers_distribution_type_embedding.' + d_cur].shape[0]
        Ea = P['_layers_address_embedding.' + a_cur].shape[0]
        S_emb = net.ff('_layers_sample_embedding.' + a_cur)[0][0].shape[0]
        x_shared = np.zeros(W_ih.shape[1], dt)
        x_shared[:col] = E[0]
        first = ps[0] < 0
        if not first:
            a_prev, d_prev = steps[ps[0]]['address'], steps[ps[0]]['dist_name']
            x_shared[col + S_emb:col + S_emb + Ed] = P['_layers_distribution_type_embedding.' + d_prev]
            x_shared[col + S_emb + Ed:col + S_emb + Ed + Ea] = P['_layers_address_embedding.' + a_prev]
        c2 = col + S_emb + Ed + Ea
        x_shared[c2:c2 + Ed] = P['_layers_distribution_type_embedding.' + d_cur]
        x_shared[c2 + Ed:c2 + Ed + Ea] = P['_layers_address_embedding.' + a_cur]
        p_lp_all, q_lp_all = np.zeros(m), np.zeros(m)
        for lo in range(0, m, chunk):
            r = rows[lo:lo + chunk]
            x = np.broadcast_to(x_shared, (len(r), x_shared.shape[0])).copy()
            if not first:
                s_emb, _ = sample_embedding(net, a_prev, d_prev, prev_val[r])
                x[:, col:col + S_emb] = s_emb
            y_rows = None
            if first:
                # every particle has the same input row and a zero state (:84-91): one row through the LSTM and the head
                _, _, (h1, c1) = lstm_forward(x[None, :1], W_ih, W_hh, b_ih, b_hh, np.zeros((1, H), dt), np.zeros((1, H), dt))
                hn, cn = np.broadcast_to(h1, (len(r), H)), np.broadcast_to(c1, (len(r), H))
                Wp, bp = net.ff('_layers_proposal.%s._ff' % a_cur)
                y_rows = np.broadcast_to(ff_forward(h1, Wp, bp, False)[0], (len(r), bp[-1].shape[0])).copy()
            else:
                _, _, (hn, cn) = lstm_forward(x[None], W_ih, W_hh, b_ih, b_hh, h[r], c[r])
            h[r], c[r] = hn, cn
            v = values[lo:lo + chunk]
            pr = prior[lo:lo + chunk]
            if d_cur == 'Bernoulli':
                # batch-1 semantics of `_infer_step`: particle i's value against ITS proposal - the diagonal of the [n, n]
                # matrix the training-time restatement (head_bernoulli) builds for a sub-batch
                q_lp = np.concatenate([np.diag(head_forward(net, a_cur, d_cur, hn[i:i + 256], pr[i:i + 256], v[i:i + 256])[2][1])
                                       for i in range(0, len(r), 256)])
            else:
                q_lp, _, _ = head_forward(net, a_cur, d_cur, hn, pr, v, y=y_rows)
            if d_cur == 'Categorical':
                C = net.ff('_layers_proposal.%s._ff' % a_cur)[0][-1].shape[0]
                p_lp = np.array([categorical_log_prob(v[i:i + 1], pr[i, :C])[0] for i in range(len(r))])
            else:
                p_lp = prior_log_prob(d_cur, pr, v)
            p_lp_all[lo:lo + chunk] = np.asarray(p_lp, np.float32).astype(np.float64)
            q_lp_all[lo:lo + chunk] = np.asarray(q_lp, np.float32).astype(np.float64)
        lw[rows] += p_lp_all - q_lp_all
        prev_step[rows] = j
        prev_val[rows] = values
        out.append((p_lp_all, q_lp_all))
    return (out, lw, (h, c)) if return_state else (out, lw)


def effective_sample_size(log_weights):
    """util.effective_sample_size, pyprob/util.py:398-399 (float64 softmax of the log-weights)."""
    lw = np.asarray(log_weights, np.float64)
    w = np.exp(lw - logsumexp(lw, axis=0))
    return 1.0 / np.sum(w * w)


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam (non-amsgrad) single-tensor update, as configured at inference_network.py:348."""
    if weight_decay != 0.0:
        g = g + weight_decay * p
    m[:] = beta1 * m + (1 - beta1) * g
    v[:] = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = np.sqrt(v) / math.sqrt(bc2) + eps
    p[:] = p - (lr / bc1) * m / denom
    return p


def sgd_step(p, g, buf, lr, momentum=0.9, nesterov=True, weight_decay=0.0):
    """torch.optim.SGD single-tensor update (dampening 0) as configured at inference_network.py:350. buf: the momentum
    buffer, ZERO before the parameter's first step (torch then sets buf = clone(g): the same value)."""
    if weight_decay != 0.0:
        g = g + weight_decay * p
    if momentum != 0.0:
        buf[:] = momentum * buf + g
        g = g + momentum * buf if nesterov else buf
    p[:] = p - lr * g
    return p


def larc_scale(p, g, lr, weight_decay=0.0, trust_coefficient=0.002, clip=True, eps=1e-8, epsilon=1.0 / 16000.0):
    """The LARC wrapper's rewrite of one tensor's gradient (pyprob/nn/optimizer_larc.py:82-102); the wrapped optimizer then
    steps with weight_decay 0. Returns the new gradient."""
    pn = math.sqrt(float(np.sum(np.square(p, dtype=np.float64))))
    gn = math.sqrt(float(np.sum(np.square(g, dtype=np.float64))))
    if pn != 0.0 and gn != 0.0:
        local = trust_coefficient * pn / (gn + pn * weight_decay + eps)
    else:
        local = epsilon
    adaptive = min(local / lr, 1.0) if clip else local
    return (g + weight_decay * p) * adaptive
