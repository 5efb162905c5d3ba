// Importance sampling with the inference network, lock-step over N particles (pyprob/state.py:203-219,
// pyprob/nn/inference_network_lstm.py:82-134, pyprob/trace.py:123-125, pyprob/model.py:59-71): device-side proposal
// sampling (Philox4x32-10), proposal / prior / likelihood log-probs, per-particle log-weight accumulation and the
// wavefront-reduced importance statistics (ESS, weighted mean / variance).
#include "common.hpp"
#include "gather.hpp"
#include "is_draw.hpp"
#include "is_step_fused.hpp"
#include "obs_embed.hpp"

#include <math.h>
#include <string.h>

#include <algorithm>

namespace pp {

int lstm_cell_fwd(float* G, const float* c_prev, float* c, float* h, int n, int H, hipStream_t st, int c_prev_shared = 0);
bool obs_fused_supported(const pp_net* net);

static inline int64_t round4(int64_t x) { return (x + 3) & ~int64_t(3); }

// First statement of a lock-step run: every particle has the same LSTM state and prior, hence the SAME proposal.
// The per-component quantities are computed once per workgroup into LDS (thread k owns component k); a particle then
// costs one Philox block, the draw, and K fused multiply-adds + exps for log q instead of re-deriving softmax, means
// and scales (10 exp + 30 divisions) itself. Same formulas as is_mixture_kernel below.
template <int KIND>
__global__ __launch_bounds__(256) void is_mixture_shared_kernel(const float* __restrict__ y, const float* __restrict__ prior,
                                                                int n, int K, const float* __restrict__ value_in,
                                                                float* __restrict__ value_out,
                                                                float* __restrict__ logq_out, uint64_t seed,
                                                                uint64_t offset) {
    __shared__ float s_mu[MAXK], s_sd[MAXK], s_inv[MAXK], s_c[MAXK], s_cum[MAXK], s_ca[MAXK], s_cb[MAXK];
    const float pa = prior[0], pb = prior[1];
    if (threadIdx.x < MAXK) {
        const int k = threadIdx.x;
        float zmax = -INFINITY;
        for (int j = 0; j < K; ++j) zmax = fmaxf(zmax, y[2 * K + j]);
        float zs = 0.0f;
        for (int j = 0; j < K; ++j) zs += expf(y[2 * K + j] - zmax);
        float ps = 0.0f, cum = 0.0f;
        for (int j = 0; j < K; ++j) ps += expf(y[2 * K + j] - zmax) / zs;
        for (int j = 0; j <= k && j < K; ++j) cum += (expf(y[2 * K + j] - zmax) / zs) / ps;
        if (k < K) {
            const float pk = (expf(y[2 * K + k] - zmax) / zs) / ps;
            float mu, sd;
            if (KIND == 0) {
                mu = pa + y[k] * pb;
                sd = expf(y[K + k]) * pb;
            } else {
                const float rng = pb - pa;
                mu = pa + sigmoidf_(y[k]) * rng;
                sd = KIND == 2 ? expf(y[K + k]) : rng / 1000.0f + sigmoidf_(y[K + k]) * rng * 10.0f;
            }
            const float lpk = logf(fminf(fmaxf(pk, kFp32Eps), 1.0f - kFp32Eps));
            s_mu[k] = mu;
            s_sd[k] = sd;
            s_inv[k] = 1.0f / sd;
            s_cum[k] = cum;
            if (KIND == 0) {
                s_c[k] = lpk - logf(sd) - kHalfLog2Pi;
                s_ca[k] = s_cb[k] = 0.0f;
            } else {
                const float ca = std_cdf((pa - mu) / sd), cb = std_cdf((pb - mu) / sd);
                s_ca[k] = ca;
                s_cb[k] = cb;
                s_c[k] = lpk - kHalfLog2Pi - logf(sd * (cb - ca));
            }
        }
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v;
    if (value_in) {
        v = value_in[i];
    } else {
        Philox rng(seed, offset + (uint64_t)i, 0x1C);
        v = NAN;
        for (int attempt = 0; attempt < 64; ++attempt) {
            uint32_t r[4];
            rng.next(r);
            const float u0 = u01(r[0]), u1 = u01(r[1]), u2 = u01(r[2]);
            int kk = K - 1;   // component index ~ Categorical(p)   (Mixture.sample, distributions/mixture.py:47-63)
            for (int k = K - 2; k >= 0; --k)
                if (u0 < s_cum[k]) kk = k;
            const float mk = s_mu[kk], sk = s_sd[kk];
            if (KIND == 0) {
                v = mk + sk * box_muller_fast(u1, u2);
                break;
            } else {
                const float uu = s_ca[kk] + u1 * (s_cb[kk] - s_ca[kk]);
                v = mk + sk * kSqrt2 * erfinvf(2.0f * uu - 1.0f);
                if (isfinite(v) && v >= pa && v < pb) break;
                v = NAN;
            }
        }
    }
    // log q(v) = logsumexp_k ( log p_k + log N(v; mu_k, sd_k) [- log Z_k] )   (Mixture.log_prob, mixture.py:42-44)
    const bool inside = (KIND == 0) || (v >= pa && v <= pb);
    float a[MAXK], amax = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) {
            const float t = (v - s_mu[k]) * s_inv[k];
            a[k] = inside ? s_c[k] - 0.5f * t * t : -INFINITY;
            amax = fmaxf(amax, a[k]);
        }
    float lp = amax;
    if (amax > -INFINITY) {
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) sum += expf(a[k] - amax);
        lp = amax + logf(sum);
    }
    value_out[i] = v;
    logq_out[i] = lp;
}

// KIND 0: Normal mixture around a Normal prior; KIND 1: TruncatedNormal mixture inside a Uniform prior; KIND 2: the
// Poisson head (TruncatedNormal mixture on [0, 40], stddev = exp(y)).
template <int KIND>
__global__ __launch_bounds__(256) void is_mixture_kernel(const float* __restrict__ Y, int64_t ldy, int y_shared,
                                                         const float* __restrict__ prior, int prior_stride, int n,
                                                         int K, const float* __restrict__ value_in,
                                                         float* __restrict__ value_out, float* __restrict__ logq_out,
                                                         uint64_t seed, uint64_t offset) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* y = Y + (y_shared ? 0 : (int64_t)i * ldy);
    const float pa = prior[(int64_t)i * 2 * prior_stride], pb = prior[(int64_t)i * 2 * prior_stride + 1];
    float v, lp;
    mixture_particle<KIND>(y, pa, pb, K, value_in != nullptr, value_in ? value_in[i] : 0.0f, seed, offset + (uint64_t)i, v, lp);
    value_out[i] = v;
    logq_out[i] = lp;
}

__global__ __launch_bounds__(256) void is_categorical_kernel(const float* __restrict__ Y, int64_t ldy, int y_shared,
                                                             int n, int C, const float* __restrict__ value_in,
                                                             float* __restrict__ value_out,
                                                             float* __restrict__ logq_out, uint64_t seed,
                                                             uint64_t offset) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* y = Y + (y_shared ? 0 : (int64_t)i * ldy);
    float zmax = -INFINITY;
    for (int k = 0; k < C; ++k) zmax = fmaxf(zmax, y[k]);
    float zs = 0.0f;
    for (int k = 0; k < C; ++k) zs += expf(y[k] - zmax);
    float S = 0.0f;
    for (int k = 0; k < C; ++k) S += expf(y[k] - zmax) / zs + 1e-8f;
    int vi;
    if (value_in) {
        vi = (int)value_in[i];
        vi = vi < 0 ? 0 : (vi >= C ? C - 1 : vi);
    } else {
        Philox rng(seed, offset + (uint64_t)i, 0x1C);
        uint32_t r[4];
        rng.next(r);
        const float u0 = u01(r[0]);
        float cum = 0.0f;
        vi = C - 1;
        for (int k = 0; k < C; ++k) {
            cum += (expf(y[k] - zmax) / zs + 1e-8f) / S;
            if (u0 < cum) {
                vi = k;
                break;
            }
        }
    }
    const float pv = (expf(y[vi] - zmax) / zs + 1e-8f) / S;
    value_out[i] = (float)vi;
    logq_out[i] = logf(fminf(fmaxf(pv, kFp32Eps), 1.0f - kFp32Eps));
}

// Bernoulli proposal (proposal_bernoulli_bernoulli.py:16-20): probs = sigmoid(y) + 1e-8; value ~ Bernoulli(probs),
// log q with torch's clamp of probs to [eps, 1 - eps].
__global__ __launch_bounds__(256) void is_bernoulli_kernel(const float* __restrict__ Y, int64_t ldy, int y_shared, int n,
                                                           const float* __restrict__ value_in,
                                                           float* __restrict__ value_out, float* __restrict__ logq_out,
                                                           uint64_t seed, uint64_t offset) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float y = Y[y_shared ? 0 : (int64_t)i * ldy];
    const float p = sigmoidf_(y) + 1e-8f;
    float v;
    if (value_in) {
        v = value_in[i];
    } else {
        Philox rng(seed, offset + (uint64_t)i, 0x1C);
        uint32_t r[4];
        rng.next(r);
        v = u01(r[0]) < p ? 1.0f : 0.0f;
    }
    const float pc = fminf(fmaxf(p, kFp32Eps), 1.0f - kFp32Eps);
    value_out[i] = v;
    logq_out[i] = p == p ? v * logf(pc) + (1.0f - v) * log1pf(-pc) : p;
}

// ---- the shared first statement of a trace as two launches ------------------------------------------------------------------
// Every particle has the same LSTM input row and a zero state (inference_network_lstm.py:84-91, 106-121): the network runs for
// ONE row. The generic chain spent five launches on it (input gather, W_ih product, cell, two head layers: ~30 us of
// launch-bound work per posterior call); here
//   first_row_lstm_kernel   builds the row x = [E | 0 | 0 | 0 | d_cur | a_cur] in LDS, takes the gate pre-activations of 4 hidden
//                           units per wave (i, g, o: the forget gate multiplies c0 = 0) as wave-reduced dot products and runs the
//                           cell: h, c of row 0;
//   first_row_head_kernel   a1 = relu(W1 h + b1), one wave per hidden column; the LAST workgroup to arrive (device-scope ticket)
//                           finishes y = W2 a1 + b2 from the coherent copies of a1.
// One-layer LSTM; the FeedForward network and deeper LSTMs keep the chain.
__global__ __launch_bounds__(256) void first_row_lstm_kernel(GatherDims d, const float* __restrict__ P,
                                                             const int64_t* __restrict__ at, const float* __restrict__ E,
                                                             int addr_id, int64_t w_ih, int64_t b_ih, int64_t b_hh, int H,
                                                             float* __restrict__ h, float* __restrict__ c) {
    __shared__ float sx[1024];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    // this wave's 4 hidden units x gates i, g, o: all weight loads are issued before the input row is complete (they do not
    // depend on it): one memory round trip for the 12 rows, one for the row's table lookups, overlapped
    const int u0 = (blockIdx.x * 4 + wave) * 4;
    constexpr int KI = 4;      // lstm_in <= 256 per 64 lanes x 4
    float wv[4][3][KI];
    float bsum[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int gsel = 0; gsel < 3; ++gsel) {
            const int n = (gsel == 0 ? 0 : gsel + 1) * H + min(u0 + q, H - 1);      // gates i, g, o
            const float* wr = P + w_ih + (int64_t)n * d.I;
#pragma unroll
            for (int kk = 0; kk < KI; ++kk) wv[q][gsel][kk] = (lane + 64 * kk) < d.I ? wr[lane + 64 * kk] : 0.0f;
            bsum[q][gsel] = P[b_ih + n] + P[b_hh + n];
        }
    for (int k = tid; k < d.I; k += 256) sx[k] = k < d.e_obs ? E[k] : gather_embedding_elem(d, P, at, k, -1, 0.0f, addr_id);
    __syncthreads();
    float xv[KI];
#pragma unroll
    for (int kk = 0; kk < KI; ++kk) xv[kk] = (lane + 64 * kk) < d.I ? sx[lane + 64 * kk] : 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float g3[3];
#pragma unroll
        for (int gsel = 0; gsel < 3; ++gsel) {
            float acc = 0.0f;
#pragma unroll
            for (int kk = 0; kk < KI; ++kk) acc += wv[q][gsel][kk] * xv[kk];
            g3[gsel] = wave_sum(acc) + bsum[q][gsel];
        }
        if (lane == 0 && u0 + q < H) {
            const float cn = sigmoidf_(g3[0]) * tanhf(g3[1]);
            c[u0 + q] = cn;
            h[u0 + q] = sigmoidf_(g3[2]) * tanhf(cn);
        }
    }
}

// ... and with the observe embedding of the one row in the SAME launch (pp_is_first_statement: _infer_init + the first
// _infer_step of a trace, inference_network.py:141-148 + inference_network_lstm.py:82-134): every workgroup stages the embedding's
// weights (38 KB, one round trip, behind its own W_ih rows' loads), wave 0 walks the row (obs_embed.hpp: the walk and summation
// order of obs_embed_fwd_kernel - the same embedding bit for bit), the other lanes look the table columns of x up meanwhile.
// `obs` may be host-mapped (pinned) memory: a posterior call hands its observation over without a copy launch; workgroup 0 leaves
// the embedding in e_out[0 .. e_obs) and the raw observation (at most 8 numbers: the x of the call's observe terms) behind it.
template <int NOBS>
__global__ __launch_bounds__(256) void first_row_net_kernel(const ObsFusedArgs ain, GatherDims d, const float* __restrict__ P,
                                                            const int64_t* __restrict__ at, const float* __restrict__ obs,
                                                            int addr_id, int64_t w_ih, int64_t b_ih, int64_t b_hh, int H,
                                                            float* __restrict__ e_out, int e4, float* __restrict__ h,
                                                            float* __restrict__ c) {
    __shared__ float lds[10240 + 1024];
    float* const sx = lds + 10240;
    const ObsFusedArgs a = ain;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int u0 = (blockIdx.x * 4 + wave) * 4;
    constexpr int KI = 4;      // lstm_in <= 256 per 64 lanes x 4
    float wv[4][3][KI];
    float bsum[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int gsel = 0; gsel < 3; ++gsel) {
            const int n = (gsel == 0 ? 0 : gsel + 1) * H + min(u0 + q, H - 1);      // gates i, g, o
            const float* wr = P + w_ih + (int64_t)n * d.I;
#pragma unroll
            for (int kk = 0; kk < KI; ++kk) wv[q][gsel][kk] = (lane + 64 * kk) < d.I ? wr[lane + 64 * kk] : 0.0f;
            bsum[q][gsel] = P[b_ih + n] + P[b_hh + n];
        }
    float ov = 0.0f;
    if (wave == 0 && lane < a.width) ov = obs[lane];      // (a.width <= 64: inputs of at most 8 observables x 8)
    obs_stage_all<NOBS>(a, P, lds, tid);
    for (int k = d.e_obs + tid; k < d.I; k += 256) sx[k] = gather_embedding_elem(d, P, at, k, -1, 0.0f, addr_id);
    if (wave == 0 && lane < a.width) sx[d.I + lane] = ov;      // the observation row (I + width <= 1024, checked on the host)
    __syncthreads();
    if (wave == 0) {
        const float e = obs_forward_row<NOBS>(a, lds, sx + d.I, lane);
        if (lane < a.e_obs) {
            sx[lane] = e;
            if (blockIdx.x == 0) e_out[lane] = e;
        }
        if (blockIdx.x == 0 && lane < min(a.width, 8)) e_out[e4 + lane] = ov;
    }
    __syncthreads();
    float xv[KI];
#pragma unroll
    for (int kk = 0; kk < KI; ++kk) xv[kk] = (lane + 64 * kk) < d.I ? sx[lane + 64 * kk] : 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float g3[3];
#pragma unroll
        for (int gsel = 0; gsel < 3; ++gsel) {
            float acc = 0.0f;
#pragma unroll
            for (int kk = 0; kk < KI; ++kk) acc += wv[q][gsel][kk] * xv[kk];
            g3[gsel] = wave_sum(acc) + bsum[q][gsel];
        }
        if (lane == 0 && u0 + q < H) {
            const float cn = sigmoidf_(g3[0]) * tanhf(g3[1]);
            c[u0 + q] = cn;
            h[u0 + q] = sigmoidf_(g3[2]) * tanhf(cn);
        }
    }
}

__global__ __launch_bounds__(256) void first_row_head_kernel(const float* __restrict__ P, const float* __restrict__ top, int Hin,
                                                             int64_t w1, int64_t b1, int hid, int64_t w2, int64_t b2, int n_out,
                                                             float* __restrict__ A1, float* __restrict__ Y, unsigned int* ticket) {
    __shared__ float sa[1024];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = blockIdx.x * 4 + wave;
    // W2 for the layer-2 pass of the LAST workgroup, requested by EVERY workgroup before it knows whether it is the last one
    // (n_out x hid floats, 32 KB: nothing against the 0.5 MB of W1) - the last one finds them in registers instead of starting
    // five dependent round trips after the ticket. Heads of at most 32 outputs and 320 hidden units; wider ones load below.
    constexpr int OW = 8, KW = 5;      // outputs per wave, k per lane
    const bool w2_pre = n_out <= 4 * OW && hid <= 64 * KW;
    float w2r[OW][KW];
#pragma unroll
    for (int q = 0; q < OW; ++q)
#pragma unroll
        for (int i = 0; i < KW; ++i) {
            const int o = min(wave * OW + q, n_out - 1), k = min(lane + 64 * i, hid - 1);
            w2r[q][i] = w2_pre ? P[w2 + (int64_t)o * hid + k] : 0.0f;
        }
    if (j < hid) {
        const float* wr = P + w1 + (int64_t)j * Hin;
        float acc = 0.0f;
        for (int k = lane * 4; k + 3 < Hin; k += 256) {      // (Hin is a multiple of 4: LSTM widths, observe-embedding widths)
            const f32x4 wq = *reinterpret_cast<const f32x4*>(wr + k);
            const f32x4 tq = *reinterpret_cast<const f32x4*>(top + k);
            acc += wq[0] * tq[0] + wq[1] * tq[1] + wq[2] * tq[2] + wq[3] * tq[3];
        }
        const float v = relu_keep_nan(wave_sum(acc) + P[b1 + j]);
        if (lane == 0) __hip_atomic_store(A1 + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __shared__ int s_last;
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const unsigned int t = atomicAdd(ticket, 1u);
        s_last = t == gridDim.x - 1u;
        if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    // the last workgroup: a1 (coherent copies) -> LDS in one round trip, then W2 a1 + b2 with the rows' loads in flight together
    for (int k = tid; k < hid; k += 256) sa[k] = __hip_atomic_load(A1 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int o0 = wave * OW; o0 < n_out; o0 += 4 * OW) {
        float acc[OW];
#pragma unroll
        for (int q = 0; q < OW; ++q) {
            acc[q] = 0.0f;
            if (w2_pre) {      // (one pass: o0 = wave * OW)
#pragma unroll
                for (int i = 0; i < KW; ++i)
                    if (lane + 64 * i < hid) acc[q] += w2r[q][i] * sa[lane + 64 * i];
            } else {
                const float* wr = P + w2 + (int64_t)min(o0 + q, n_out - 1) * hid;
                for (int k = lane; k < hid; k += 64) acc[q] += wr[k] * sa[k];
            }
        }
#pragma unroll
        for (int q = 0; q < OW; ++q) {
            const float v = wave_sum(acc[q]);
            if (lane == 0 && o0 + q < n_out) Y[o0 + q] = v + P[b2 + o0 + q];
        }
    }
}

// ... and head layer 1 in the SAME launch as the one-row LSTM step (pp_is_first_statement's default): blocks [0, nl) are the
// LSTM workgroups of first_row_net_kernel - their h values leave as agent-scope atomic stores and each workgroup counts itself
// in `ready` once its waves' stores are acknowledged -, blocks [nl, nl + cdiv(hid, 4)) are first_row_head_kernel's: they fetch
// their rows of W1 (and W2 for the last arriver) FIRST - the loads do not depend on h -, then wait for ready == nl and read h
// with agent-scope atomic loads. The head's cold weight loads (~2 us) and one kernel boundary disappear from the call's chain.
// Progress: the LSTM blocks have the lower block ids (dispatched first), 100 workgroups fit any device this library runs on.
template <int NOBS>
__global__ __launch_bounds__(256) void first_row_all_kernel(const ObsFusedArgs ain, GatherDims d, const float* __restrict__ P,
                                                            const int64_t* __restrict__ at, const float* __restrict__ obs,
                                                            int addr_id, int64_t w_ih, int64_t b_ih, int64_t b_hh, int H,
                                                            float* __restrict__ e_out, int e4, float* __restrict__ h,
                                                            float* __restrict__ c, int nl, int64_t w1, int64_t b1, int hid,
                                                            int64_t w2, int64_t b2, int n_out, float* __restrict__ A1,
                                                            float* __restrict__ Y, unsigned int* ticket, unsigned int* ready, const int acqrel) {
    __shared__ float lds[10240 + 1024];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    if ((int)blockIdx.x < nl) {
        // ---- the LSTM workgroups (first_row_net_kernel) ----
        float* const sx = lds + 10240;
        const ObsFusedArgs a = ain;
        const int u0 = (blockIdx.x * 4 + wave) * 4;
        constexpr int KI = 4;
        float wv[4][3][KI];
        float bsum[4][3];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int gsel = 0; gsel < 3; ++gsel) {
                const int n = (gsel == 0 ? 0 : gsel + 1) * H + min(u0 + q, H - 1);
                const float* wr = P + w_ih + (int64_t)n * d.I;
#pragma unroll
                for (int kk = 0; kk < KI; ++kk) wv[q][gsel][kk] = (lane + 64 * kk) < d.I ? wr[lane + 64 * kk] : 0.0f;
                bsum[q][gsel] = P[b_ih + n] + P[b_hh + n];
            }
        float ov = 0.0f;
        if (wave == 0 && lane < a.width) ov = obs[lane];
        obs_stage_all<NOBS>(a, P, lds, tid);
        for (int k = d.e_obs + tid; k < d.I; k += 256) sx[k] = gather_embedding_elem(d, P, at, k, -1, 0.0f, addr_id);
        if (wave == 0 && lane < a.width) sx[d.I + lane] = ov;
        __syncthreads();
        if (wave == 0) {
            const float e = obs_forward_row<NOBS>(a, lds, sx + d.I, lane);
            if (lane < a.e_obs) {
                sx[lane] = e;
                if (blockIdx.x == 0) e_out[lane] = e;
            }
            if (blockIdx.x == 0 && lane < min(a.width, 8)) e_out[e4 + lane] = ov;
        }
        __syncthreads();
        float xv[KI];
#pragma unroll
        for (int kk = 0; kk < KI; ++kk) xv[kk] = (lane + 64 * kk) < d.I ? sx[lane + 64 * kk] : 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float g3[3];
#pragma unroll
            for (int gsel = 0; gsel < 3; ++gsel) {
                float acc = 0.0f;
#pragma unroll
                for (int kk = 0; kk < KI; ++kk) acc += wv[q][gsel][kk] * xv[kk];
                g3[gsel] = wave_sum(acc) + bsum[q][gsel];
            }
            if (lane == 0 && u0 + q < H) {
                const float cn = sigmoidf_(g3[0]) * tanhf(g3[1]);
                c[u0 + q] = cn;
                __hip_atomic_store(h + u0 + q, sigmoidf_(g3[2]) * tanhf(cn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's h stores are acknowledged
        __syncthreads();
        // release / acquire on `ready`: the count publishes the workgroup's h and c stores at agent scope
        // (PP_IS_FIRST_SYNC=0: relaxed count behind the waves' own s_waitcnt - the stores above are agent-scope atomics that
        // bypass the non-coherent caches; A/B of the fence's cost)
        if (tid == 0) {
            if (acqrel) __hip_atomic_fetch_add(ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    // ---- head layer 1 (first_row_head_kernel), the last arriver finishes layer 2 ----
    float* const sa = lds;
    const int j = ((int)blockIdx.x - nl) * 4 + wave;
    const int nh = (int)gridDim.x - nl;
    constexpr int OW = 8, KW = 5;
    const bool w2_pre = n_out <= 4 * OW && hid <= 64 * KW;
    float w2r[OW][KW];
#pragma unroll
    for (int q = 0; q < OW; ++q)
#pragma unroll
        for (int i = 0; i < KW; ++i) {
            const int o = min(wave * OW + q, n_out - 1), k = min(lane + 64 * i, hid - 1);
            w2r[q][i] = w2_pre ? P[w2 + (int64_t)o * hid + k] : 0.0f;
        }
    // this wave's row of W1, all of it, before h exists (H <= 1024: four float4 per lane)
    constexpr int WQ = 4;
    f32x4 wq[WQ];
    const float* wr = P + w1 + (int64_t)min(j, hid - 1) * H;
#pragma unroll
    for (int i = 0; i < WQ; ++i) wq[i] = (lane * 4 + 256 * i + 3 < H) ? *reinterpret_cast<const f32x4*>(wr + lane * 4 + 256 * i) : f32x4{0, 0, 0, 0};
    const float b1v = P[b1 + min(j, hid - 1)];
    {
        int spins = 0;
        while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nl) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1 << 22)) __builtin_trap();
        }
        if (acqrel) (void)__hip_atomic_load(ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (j < hid) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < WQ; ++i) {
            const int k = lane * 4 + 256 * i;
            if (k + 3 < H) {      // (first_row_head_kernel's expression: the same rounding, bit-identical outputs)
                const float t0 = __hip_atomic_load(h + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float t1 = __hip_atomic_load(h + k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float t2 = __hip_atomic_load(h + k + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float t3 = __hip_atomic_load(h + k + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc += wq[i][0] * t0 + wq[i][1] * t1 + wq[i][2] * t2 + wq[i][3] * t3;
            }
        }
        const float v = relu_keep_nan(wave_sum(acc) + b1v);
        if (lane == 0) __hip_atomic_store(A1 + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == (unsigned)nh - 1u;
        if (s_last) {      // every head workgroup has passed its wait: both words are ready for the next launch
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ready, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!s_last) return;
    for (int k = tid; k < hid; k += 256) sa[k] = __hip_atomic_load(A1 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int o0 = wave * OW; o0 < n_out; o0 += 4 * OW) {
        float acc[OW];
#pragma unroll
        for (int q = 0; q < OW; ++q) {
            acc[q] = 0.0f;
            if (w2_pre) {
#pragma unroll
                for (int i = 0; i < KW; ++i)
                    if (lane + 64 * i < hid) acc[q] += w2r[q][i] * sa[lane + 64 * i];
            } else {
                const float* wr2 = P + w2 + (int64_t)min(o0 + q, n_out - 1) * hid;
                for (int k = lane; k < hid; k += 64) acc[q] += wr2[k] * sa[k];
            }
        }
#pragma unroll
        for (int q = 0; q < OW; ++q) {
            const float v = wave_sum(acc[q]);
            if (lane == 0 && o0 + q < n_out) Y[o0 + q] = v + P[b2 + o0 + q];
        }
    }
}

struct IsWorkspace {
    float *X, *G, *A1, *Y, *rec, *c0;
    float *obs_h, *cat, *f1;
    int64_t i4, hid4, out4, e4, maxohid4;
    IsFusedBuffers fz;   // operand images of the fused statement kernel (is_step_fused.hip)
    float* ticket;       // arrival counter of first_row_head_kernel (a 32-bit word: zero when the workspace is allocated, left at
                         // zero by every launch)
    size_t bytes;
};

static void is_carve(const pp_net* net, int n, void* p, IsWorkspace& w) {
    char* base = static_cast<char*>(p);
    size_t off = 0;
    auto take = [&](int64_t count) {
        off = (off + 255) & ~size_t(255);
        float* q = base ? reinterpret_cast<float*>(base + off) : nullptr;
        off += (size_t)std::max<int64_t>(count, 1) * sizeof(float);
        return q;
    };
    w.ticket = take(64);   // FIRST: a fixed place whatever n is (the scratch of an n-row call must never run over it)
    const int H = net->lstm_dim;
    w.i4 = round4(net->lstm_in);
    w.e4 = round4(net->e_obs);
    int64_t hid = 1, out = 1;
    for (int a = 0; a < net->n_addr; ++a) {
        hid = std::max<int64_t>(hid, net->addrs[a].hid);
        out = std::max<int64_t>(out, net->addrs[a].n_out);
    }
    w.hid4 = round4(hid);
    w.out4 = round4(out);
    w.maxohid4 = 4;
    for (int o = 0; o < net->n_obs; ++o) w.maxohid4 = std::max<int64_t>(w.maxohid4, round4(net->obs_hid[o]));
    w.X = take((int64_t)n * w.i4);
    w.G = take((int64_t)n * 4 * H);
    w.A1 = take((int64_t)n * w.hid4);
    w.Y = take((int64_t)n * w.out4);
    w.rec = take(4 * H);   // shared recurrent row h0 W_hh^T + b_hh of the second statement
    w.c0 = take(H);        // copy of the shared cell state row (c is rewritten in place)
    w.obs_h = take(PP_MAX_OBS * w.maxohid4);
    w.cat = take(w.e4);
    w.f1 = take(w.e4);
    is_fused_carve_sizes(net, w.fz);
    w.fz.whh = take(w.fz.n_whh);
    w.fz.w1 = take(w.fz.n_w1);
    w.fz.w2 = take(w.fz.n_w2);
    w.fz.bias = take(w.fz.n_bias);
    w.bytes = off + 256;
}

// PP_IS_STEP_FUSED (read per call; A/B and the parity cases of tests/test_gpu_is_step_fused.py): 0 = the chain of GEMM launches,
// 1 (default) = one kernel from SPLIT_MAX_ROWS + 1 particles on, the two-launch split statement below that, 2 = always the
// one-kernel statement, 3 = always the split statement
static int is_step_fused_mode() {
    const char* e = getenv("PP_IS_STEP_FUSED");
    return e ? atoi(e) : 1;
}
// One workgroup of the one-kernel statement streams all 4.7 MB of weights for its 32 particles: up to 256 workgroups (8 192
// particles) that is one generation of ~0.23 ms whatever n. The split statement's LSTM launch is MFMA-bound from ~2 000
// particles on (0.11 ms at 8 192) and its head launch costs ~0.04 ms: below 4 096 particles it wins clearly.
constexpr int SPLIT_MAX_ROWS = 4096;
// H = 1024: a workgroup of the wide LSTM launch streams 8.4 MB of weights for its 32 particles whatever n - 0.33 ms per statement up
// to 4 096 particles; the GEMM chain takes 0.14 / 0.20 / 0.33 / 0.69 ms at 512 / 1 024 / 2 048 / 4 096
// (profiles/s5w_h1024_statement_sweep.jsonl): from 2 049 particles on the fused statement
constexpr int WIDE_MIN_ROWS = 2048;
static bool is_step_fused_pays(const pp_net* net, int n) {      // mode 1: is the fused statement the faster path for n particles?
    return !(net->lstm_dim == 1024 && is_step_fused_mode() == 1 && n <= WIDE_MIN_ROWS);
}
static bool is_step_split(int n) {
    const int mode = is_step_fused_mode();
    return mode == 3 || (mode == 1 && n <= SPLIT_MAX_ROWS);
}

static int lin(const float* x, int64_t ldx, const float* W, const float* b, const float* b2, float* y, int64_t ldy, int n,
               int in, int out, bool relu, bool accumulate, hipStream_t st) {
    pp_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = x; g.lda = ldx;
    g.B = W; g.ldb = in;
    g.C = y; g.ldc = ldy;
    g.M = n; g.N = out; g.K = in;
    g.bias = b; g.bias2 = b2; g.relu = relu; g.accumulate = accumulate;
    return gemm_f32(&g, st);
}

int is_init(const pp_net* net, const float* P, const float* obs, float* e_out, void* ws, size_t ws_bytes,
            hipStream_t st) {
    PP_CHECK_ARG(net && P && obs && e_out && ws, "pp_is_init: null pointer");
    IsWorkspace w;
    is_carve(net, 1, ws, w);
    if (w.bytes > ws_bytes) {
        set_error("pp_is_init: workspace too small (%zu < %zu bytes)", ws_bytes, w.bytes);
        return PP_ENOSPACE;
    }
    if (obs_fused_supported(net)) {   // one fused launch (obs_embed.hip); e_out has the same row stride round4(e_obs)
        float* oh[PP_MAX_OBS];
        for (int o = 0; o < PP_MAX_OBS; ++o) oh[o] = w.obs_h + (int64_t)o * w.maxohid4;
        return obs_embed_fwd_fused(net, P, obs, 1, oh, w.cat, w.f1, e_out, st);
    }
    int ci = 0, co = 0, width = 0;
    for (int o = 0; o < net->n_obs; ++o) width += net->obs_in[o];
    for (int o = 0; o < net->n_obs; ++o) {
        // EmbeddingFeedForward(num_layers = depth): the hidden rows ping-pong between two slots of obs_h
        const int depth = net->obs_depth[o] ? net->obs_depth[o] : 2;
        const float* x = obs + ci;
        int64_t ldx = width;
        int in = net->obs_in[o];
        for (int l = 0; l < depth; ++l) {
            const bool last = l == depth - 1;
            const int out = last ? net->obs_out[o] : net->obs_hid[o];
            float* y = last ? w.cat + co : w.obs_h + (int64_t)(l & 1) * w.maxohid4;
            const int64_t wl = net->obs_depth[o] ? net->obs_w[o][l] : (l == 0 ? net->obs_w0[o] : net->obs_w1[o]);
            const int64_t bl = net->obs_depth[o] ? net->obs_b[o][l] : (l == 0 ? net->obs_b0[o] : net->obs_b1[o]);
            PP_TRY(lin(x, ldx, P + wl, P + bl, nullptr, y, last ? w.e4 : w.maxohid4, 1, in, out, true, false, st));
            x = y; ldx = last ? w.e4 : w.maxohid4; in = out;
        }
        ci += net->obs_in[o];
        co += net->obs_out[o];
    }
    PP_TRY(lin(w.cat, w.e4, P + net->fin_w0, P + net->fin_b0, nullptr, w.f1, w.e4, 1, net->e_obs, net->e_obs, true, false, st));
    PP_TRY(lin(w.f1, w.e4, P + net->fin_w1, P + net->fin_b1, nullptr, e_out, w.e4, 1, net->e_obs, net->e_obs, true, false, st));
    return 0;
}

// _infer_init + the network part of a trace's first _infer_step in two launches (first_row_net_kernel, first_row_head_kernel)
bool is_first_statement_supported(const pp_net* net, int addr_id) {
    const char* ev = getenv("PP_IS_FIRST");      // (read per call: the A/B tests flip it inside one process)
    if ((ev && atoi(ev) == 0) || !net || net->lstm_dim == 0 || std::max(1, (int)net->lstm_depth) != 1 || !net->addr_table) return false;
    if (addr_id < 0 || addr_id >= net->n_addr || net->lstm_in > 256 || (net->lstm_dim % 4) != 0 || net->addrs[addr_id].hid > 1024) return false;
    if (!obs_fused_supported(net)) return false;
    int width = 0;
    for (int o = 0; o < net->n_obs; ++o) width += net->obs_in[o];
    return width <= 64 && net->lstm_in + width <= 1024;
}

int is_first_statement(const pp_net* net, const float* P, const float* obs, int addr_id, float* e_out, float* h, float* c, void* ws,
                       size_t ws_bytes, hipStream_t st) {
    PP_CHECK_ARG(net && P && obs && e_out && h && c && ws, "pp_is_first_statement: null pointer");
    PP_CHECK_ARG(is_first_statement_supported(net, addr_id), "pp_is_first_statement: unsupported network (pp_is_first_statement_supported)");
    IsWorkspace w;
    is_carve(net, 1, ws, w);
    if (w.bytes > ws_bytes) {
        set_error("pp_is_first_statement: workspace too small (%zu < %zu bytes)", ws_bytes, w.bytes);
        return PP_ENOSPACE;
    }
    float* oh[PP_MAX_OBS];
    for (int o = 0; o < PP_MAX_OBS; ++o) oh[o] = w.obs_h + (int64_t)o * w.maxohid4;
    ObsFusedArgs a;
    PP_CHECK_ARG(obs_fused_args(net, oh, a), "pp_is_first_statement: the embedding image does not fit");
    const pp_addr& ad = net->addrs[addr_id];
    const int H = net->lstm_dim;
    GatherDims gd{net->e_obs, net->smp_dim, net->dtype_dim, net->addr_dim, net->lstm_in};
    const char* ev = getenv("PP_IS_FIRST");      // 1 (default): one launch; 2: the LSTM and the head launches separately (A/B)
    // The head workgroups of the one-launch variant WAIT for the LSTM workgroups of the same launch: every workgroup of the
    // grid must be resident at once (gated on an occupancy query like panel16_supported, not on a block count
    // alone) - otherwise the two-launch path
    const int grid_one = cdiv(H, 16) + cdiv(ad.hid, 4);
    bool one = !(ev && atoi(ev) == 2) && H <= 1024 && grid_one <= 200;
    if (one) {
        static int resident[4] = {-1, -1, -1, -1};
        const int vi = a.n_obs <= 1 ? 0 : a.n_obs <= 2 ? 1 : a.n_obs <= 4 ? 2 : 3;
        if (resident[vi] < 0) {
            int per_cu = 0, dev = 0, cus = 0;
            const void* fn = vi == 0 ? (const void*)first_row_all_kernel<1> : vi == 1 ? (const void*)first_row_all_kernel<2>
                           : vi == 2 ? (const void*)first_row_all_kernel<4> : (const void*)first_row_all_kernel<8>;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess)
                per_cu = cus = 0;
            resident[vi] = per_cu * cus;
        }
        one = resident[vi] >= grid_one;
    }
    unsigned int* tk = reinterpret_cast<unsigned int*>(w.ticket);
    // `ready` with release / acquire semantics (PP_IS_FIRST_SYNC=1) or relaxed behind the producers' own s_waitcnt vmcnt(0) +
    // barrier (default). Measured (profiles/r06c_ab.txt): the agent-scope release costs every LSTM workgroup an L2 write-back and
    // the call 3.5 us of ~70 (0.0735 vs 0.0700 ms at 10^6 particles). The relaxed protocol is the hand-off protocol of
    // handoff.hpp: h and the count are agent-scope atomics (they bypass the per-XCD caches), a wave's stores are acknowledged
    // before its workgroup's count goes up, a consumer wave issues its h loads after its count load has returned.
    const char* es = getenv("PP_IS_FIRST_SYNC");
    const int acqrel = es ? atoi(es) : 0;
    if (one) {
#define PP_FIRST_ALL(N)                                                                                                          \
    hipLaunchKernelGGL(first_row_all_kernel<N>, dim3(cdiv(H, 16) + cdiv(ad.hid, 4)), dim3(256), 0, st, a, gd, P, net->addr_table, \
                       obs, addr_id, net->w_ih, net->b_ih, net->b_hh, H, e_out, (int)w.e4, h, c, cdiv(H, 16), ad.w1, ad.b1,       \
                       ad.hid, ad.w2, ad.b2, ad.n_out, w.A1, w.Y, tk, tk + 1, acqrel)
        if (a.n_obs <= 1) PP_FIRST_ALL(1);
        else if (a.n_obs <= 2) PP_FIRST_ALL(2);
        else if (a.n_obs <= 4) PP_FIRST_ALL(4);
        else PP_FIRST_ALL(8);
#undef PP_FIRST_ALL
    } else {
#define PP_FIRST_NET(N)                                                                                                       \
    hipLaunchKernelGGL(first_row_net_kernel<N>, dim3(cdiv(H, 16)), dim3(256), 0, st, a, gd, P, net->addr_table, obs, addr_id, \
                       net->w_ih, net->b_ih, net->b_hh, H, e_out, (int)w.e4, h, c)
        if (a.n_obs <= 1) PP_FIRST_NET(1);
        else if (a.n_obs <= 2) PP_FIRST_NET(2);
        else if (a.n_obs <= 4) PP_FIRST_NET(4);
        else PP_FIRST_NET(8);
#undef PP_FIRST_NET
        hipLaunchKernelGGL(first_row_head_kernel, dim3(cdiv(ad.hid, 4)), dim3(256), 0, st, P, (const float*)h, H, ad.w1, ad.b1, ad.hid,
                           ad.w2, ad.b2, ad.n_out, w.A1, w.Y, tk);
    }
    PP_LAUNCH_CHECK("pp_is_first_statement");
    return 0;
}

int is_step(const pp_net* net, const float* P, int addr_id, int prev_addr_id, int n, const float* e_obs_vec,
            const float* prev_value, const float* prior, int prior_stride, float* h, float* c, int state_rows,
            const float* value_in, float* value_out, float* logq_out, uint64_t seed, uint64_t offset, void* ws,
            size_t ws_bytes, hipStream_t st, bool net_only = false, const int64_t* rows = nullptr,
            const IsStatementOut* whole = nullptr) {
    const bool ff = net && net->lstm_dim == 0;   // FeedForward network: the proposal layer reads the observe embedding
    PP_CHECK_ARG(net && P && e_obs_vec && (ff || (h && c)) && (net_only || whole || (value_out && logq_out)) && ws, "pp_is_step: null pointer");
    PP_CHECK_ARG(addr_id >= 0 && addr_id < net->n_addr && prev_addr_id < net->n_addr, "pp_is_step: address id out of range");
    PP_CHECK_ARG(ff || prev_addr_id < 0 || prev_value, "pp_is_step: prev_value is required after the first statement");
    // (with a row index list state_rows is 1 or the row count of one layer of the state buffer: the rows point anywhere into it)
    PP_CHECK_ARG(ff || prev_addr_id < 0 || state_rows == 1 || (rows ? state_rows >= 1 : state_rows == n),
                 "pp_is_step: state_rows must be 1 or n (with a row list: 1 or the rows of a layer of the state buffer)");
    PP_CHECK_ARG(ff || !rows || state_rows != 1 || std::max(1, (int)net->lstm_depth) == 1,
                 "pp_is_step_rows: a shared first state with a row list on an LSTM of depth > 1: expand the state rows first");
    if (n <= 0) return 0;
    const pp_addr& ad = net->addrs[addr_id];
    PP_CHECK_ARG(net_only || ad.kind == PP_HEAD_CATEGORICAL || ad.kind == PP_HEAD_BERNOULLI || prior,
                 "pp_is_step: prior parameters required");
    // First statement of a trace: identical LSTM input and zero state for every particle -> ONE row is evaluated and
    // only row 0 of (h, c) is written (the caller's state_rows becomes 1). Second statement: the inputs differ (previous
    // value) but the recurrent term h0 W_hh^T is still one shared row -> it enters the batched GEMM as a bias.
    // FeedForward network (inference_network_feedforward.py:52-66): no state at all - every statement's proposal is its
    // layer applied to the ONE observe embedding row, identical for all particles.
    const bool shared = ff || prev_addr_id < 0;
    const int m = shared ? 1 : n;
    const int H = ff ? net->e_obs : net->lstm_dim, I = net->lstm_in;
    IsWorkspace w;
    is_carve(net, m, ws, w);
    if (w.bytes > ws_bytes) {
        set_error("pp_is_step: workspace too small (%zu < %zu bytes)", ws_bytes, w.bytes);
        return PP_ENOSPACE;
    }
    // A statement after the first one on a one-layer LSTM of a supported width: ONE kernel (is_step_fused.hip) - gates,
    // cell, both head layers and the draw; (h, c) are read and written once, in place, optionally through a row index list.
    // Small launches take the same statement as two launches (the LSTM step split over the gate columns, then head + draw):
    // is_step_split above. PP_IS_STEP_FUSED=0 keeps the chain of GEMM launches below (A/B, tests).
    bool head_done = false;
    // (H = 1024 below WIDE_MIN_ROWS particles: the chain, unless the caller needs the row list / the whole-statement tail)
    if (!shared && is_step_fused_supported(net, addr_id) && is_step_fused_mode() != 0 && (rows || whole || is_step_fused_pays(net, n))) {
        bool sampled = false;
        // (32-bit element offsets into the state rows: a row list's indices are the caller's to bound - ISRunner.begin does)
        PP_CHECK_ARG(rows || (int64_t)n * H < (int64_t(1) << 32), "pp_is_step: more than 2^32 state elements per call: shard the particles");
        // (the split statement's new hidden rows go through the chain's gate buffer w.G: [n][4 H] >= [n][H])
        PP_TRY(is_step_fused(net, P, addr_id, prev_addr_id, n, e_obs_vec, prev_value, prior, prior_stride, h, c, state_rows, rows,
                             value_in, value_out, logq_out, seed, offset, w.fz, w.c0, w.Y, w.out4, net_only, &sampled, st, whole,
                             (H == 1024 || (state_rows != 1 && is_step_split(n))) ? w.G : nullptr,
                             rows && state_rows != 1 ? state_rows : n));
        if (sampled) return 0;
        PP_CHECK_ARG(!whole, "pp_is_statement_rows: mixture heads only");
        head_done = true;     // the head outputs are in w.Y: the sampling kernels below (or pp_is_fused) take over
    } else {
        PP_CHECK_ARG(!rows && !whole, "pp_is_step_rows / pp_is_statement_rows need the fused statement kernel (pp_is_step_fused_supported)");
    }
    // the shared first statement on a one-layer LSTM: two launches (first_row_lstm_kernel, first_row_head_kernel) instead of the
    // chain's five
    const bool first_row = shared && !ff && std::max(1, (int)net->lstm_depth) == 1 && net->lstm_in <= 256 && (H % 4) == 0 &&
                           ad.hid <= 1024 && net->addr_table;
    if (first_row) {
        GatherDims gd{net->e_obs, net->smp_dim, net->dtype_dim, net->addr_dim, net->lstm_in};
        hipLaunchKernelGGL(first_row_lstm_kernel, dim3(cdiv(H, 16)), dim3(256), 0, st, gd, P, net->addr_table, e_obs_vec, addr_id,
                           net->w_ih, net->b_ih, net->b_hh, H, h, c);
        hipLaunchKernelGGL(first_row_head_kernel, dim3(cdiv(ad.hid, 4)), dim3(256), 0, st, P, (const float*)h, H, ad.w1, ad.b1,
                           ad.hid, ad.w2, ad.b2, ad.n_out, w.A1, w.Y, reinterpret_cast<unsigned int*>(w.ticket));
        PP_LAUNCH_CHECK("pp_is_step(first statement)");
        head_done = true;
    }
    if (head_done) {
        if (net_only) return 0;
    } else {
    // H = 1024, one layer: the LSTM step as ONE launch (is_step_fused.hip is_lstm_wide: gates on the accumulators, cell in place;
    // no gathered input rows, no gate matrix in memory), then the head GEMMs below. Launches of up to 2 048 particles keep the
    // GEMM chain: a workgroup of the wide launch streams 8.4 MB of weights for its 32 particles whatever n - 0.33 ms per statement
    // up to 4 096 particles, the chain 0.14 / 0.20 / 0.33 / 0.69 ms at 512 / 1 024 / 2 048 / 4 096
    // (profiles/s5w_h1024_statement_sweep.jsonl). (Heads the fused statement's head-only launch does not take: wider than 32 outputs
    // or 576 hidden units; else pp_is_step took the fused statement above.)
    const int fmode = is_step_fused_mode();
    const bool wide = !ff && !shared && is_lstm_wide_supported(net) && (fmode >= 2 || (fmode == 1 && n > WIDE_MIN_ROWS)) &&
                      (int64_t)n * H < (int64_t(1) << 32);
    if (!ff && !wide)
        PP_TRY(lstm_input_gather(net, P, e_obs_vec, 0, nullptr, prev_value, nullptr, nullptr, addr_id, prev_addr_id, m, w.X,
                                 w.i4, st));
    const float* top = h;    // hidden rows the proposal layer reads
    if (ff) {
        top = e_obs_vec;
    } else if (wide) {
        // the new hidden rows pass through the chain's gate buffer (other workgroups still read the old rows); with the shared
        // state of the second statement nobody does
        float* hn = state_rows == 1 ? h : w.G;
        PP_TRY(is_lstm_wide(net, P, addr_id, prev_addr_id, n, e_obs_vec, prev_value, h, c, state_rows, w.fz, w.c0, hn, st));
        if (hn != h) (void)hipMemcpyAsync(h, hn, (size_t)n * H * sizeof(float), hipMemcpyDeviceToDevice, st);
    } else {
        // nn.LSTM(I, H, depth): layer k reads the new hidden rows of layer k - 1; (h, c) hold [depth, n, H]
        const int L = std::max(1, (int)net->lstm_depth);
        for (int l = 0; l < L; ++l) {
            float* hl = h + (int64_t)l * n * H;
            float* cl = c + (int64_t)l * n * H;
            const float* in = l == 0 ? w.X : h + (int64_t)(l - 1) * n * H;
            const int64_t in_ld = l == 0 ? w.i4 : H;
            const int in_w = l == 0 ? I : H;
            const float* Wih = P + (l == 0 ? net->w_ih : net->lstm_w_ih[l]);
            const float* Whh = P + (l == 0 ? net->w_hh : net->lstm_w_hh[l]);
            const float* bih = P + (l == 0 ? net->b_ih : net->lstm_b_ih[l]);
            const float* bhh = P + (l == 0 ? net->b_hh : net->lstm_b_hh[l]);
            if (shared) {
                PP_TRY(lin(in, in_ld, Wih, bih, bhh, w.G, 4 * H, 1, in_w, 4 * H, false, false, st));
                PP_TRY(lstm_cell_fwd(w.G, nullptr, cl, hl, 1, H, st));
            } else if (state_rows == 1) {
                PP_TRY(lin(hl, H, Whh, bhh, nullptr, w.rec, 4 * H, 1, H, 4 * H, false, false, st));
                PP_TRY(lin(in, in_ld, Wih, bih, w.rec, w.G, 4 * H, n, in_w, 4 * H, false, false, st));
                (void)hipMemcpyAsync(w.c0, cl, (size_t)H * sizeof(float), hipMemcpyDeviceToDevice, st);   // c is rewritten in place
                PP_TRY(lstm_cell_fwd(w.G, w.c0, cl, hl, n, H, st, /*c_prev_shared=*/1));
            } else {
                PP_TRY(lin(in, in_ld, Wih, bih, bhh, w.G, 4 * H, n, in_w, 4 * H, false, false, st));
                PP_TRY(lin(hl, H, Whh, nullptr, nullptr, w.G, 4 * H, n, H, 4 * H, false, true, st));
                PP_TRY(lstm_cell_fwd(w.G, cl, cl, hl, n, H, st));
            }
        }
        top = h + (int64_t)(L - 1) * n * H;
    }
    PP_TRY(lin(top, H, P + ad.w1, P + ad.b1, nullptr, w.A1, w.hid4, m, H, ad.hid, true, false, st));
    PP_TRY(lin(w.A1, w.hid4, P + ad.w2, P + ad.b2, nullptr, w.Y, w.out4, m, ad.hid, ad.n_out, false, false, st));
    if (net_only) return 0;      // the head outputs stay in w.Y for pp_is_fused
    }
    dim3 grid(cdiv(n, 256)), block(256);
    // kernel class 4 of the in-stream timing: draw + log q per particle (writes value and log q: 8 algorithmic bytes each)
    prof_begin(4, st);
    if (ad.kind == PP_HEAD_CATEGORICAL) {
        hipLaunchKernelGGL(is_categorical_kernel, grid, block, 0, st, w.Y, w.out4, shared ? 1 : 0, n, ad.n_out, value_in,
                           value_out, logq_out, seed, offset);
    } else if (ad.kind == PP_HEAD_BERNOULLI) {
        hipLaunchKernelGGL(is_bernoulli_kernel, grid, block, 0, st, w.Y, w.out4, shared ? 1 : 0, n, value_in, value_out,
                           logq_out, seed, offset);
    } else {
        PP_CHECK_ARG(ad.n_out % 3 == 0 && ad.n_out / 3 <= MAXK, "pp_is_step: at most %d mixture components", MAXK);
        const bool same_proposal = shared && prior_stride == 0;
        if (same_proposal && ad.kind == PP_HEAD_NORMAL_MIXTURE)
            hipLaunchKernelGGL(is_mixture_shared_kernel<0>, grid, block, 0, st, w.Y, prior, n, ad.n_out / 3, value_in, value_out,
                               logq_out, seed, offset);
        else if (same_proposal && ad.kind == PP_HEAD_TRUNCNORMAL_MIXTURE)
            hipLaunchKernelGGL(is_mixture_shared_kernel<1>, grid, block, 0, st, w.Y, prior, n, ad.n_out / 3, value_in, value_out,
                               logq_out, seed, offset);
        else if (same_proposal)
            hipLaunchKernelGGL(is_mixture_shared_kernel<2>, grid, block, 0, st, w.Y, prior, n, ad.n_out / 3, value_in, value_out,
                               logq_out, seed, offset);
        else if (ad.kind == PP_HEAD_NORMAL_MIXTURE)
            hipLaunchKernelGGL(is_mixture_kernel<0>, grid, block, 0, st, w.Y, w.out4, shared ? 1 : 0, prior, prior_stride,
                               n, ad.n_out / 3, value_in, value_out, logq_out, seed, offset);
        else if (ad.kind == PP_HEAD_TRUNCNORMAL_MIXTURE)
            hipLaunchKernelGGL(is_mixture_kernel<1>, grid, block, 0, st, w.Y, w.out4, shared ? 1 : 0, prior, prior_stride,
                               n, ad.n_out / 3, value_in, value_out, logq_out, seed, offset);
        else
            hipLaunchKernelGGL(is_mixture_kernel<2>, grid, block, 0, st, w.Y, w.out4, shared ? 1 : 0, prior, prior_stride,
                               n, ad.n_out / 3, value_in, value_out, logq_out, seed, offset);
    }
    prof_end(4, 8.0 * n + (value_in ? 4.0 * n : 0.0), st);
    PP_LAUNCH_CHECK("pp_is_step(sample)");
    return 0;
}

// log_prob of the prior / likelihood families state.sample and state.observe score (state.py:211, 147-149), as torch
// evaluates them in fp32:
//   0 Normal(mean a, stddev b)       torch.distributions.Normal.log_prob
//   1 Uniform(low a, high b)         support [low, high)
//   3 Poisson(rate a)                xlogy(v, rate) - rate - lgamma(v + 1)
//   4 Bernoulli(probs a)             probs clamped to [eps, 1 - eps] (probs_to_logits), v log p + (1 - v) log(1 - p)
//   5 Categorical(probs row p0[i * s0 .. + C), C = s1)   log(clamp(p[v] / sum p, eps, 1 - eps))
__device__ __forceinline__ float term_log_prob(int kind, const float* __restrict__ p0, int s0, const float* __restrict__ p1,
                                               int s1, float v, int64_t i) {
    if (kind == 5) {
        const float* p = p0 + i * s0;
        const int C = s1;
        float sum = 0.0f;
        for (int c = 0; c < C; ++c) sum += p[c];
        const int k = (int)v;
        if (k < 0 || k >= C) return -INFINITY;
        const float q = fminf(fmaxf(p[k] / sum, kFp32Eps), 1.0f - kFp32Eps);
        return logf(q);
    }
    const float a = p0[i * s0];
    if (kind == 3) return (v == 0.0f ? 0.0f : v * logf(a)) - a - lgammaf(v + 1.0f);
    if (kind == 4) {
        const float q = fminf(fmaxf(a, kFp32Eps), 1.0f - kFp32Eps);
        return v * logf(q) + (1.0f - v) * log1pf(-q);
    }
    const float b = p1[i * s1];
    if (kind == 0) {
        const float t = v - a;
        return -(t * t) / (2.0f * b * b) - logf(b) - kHalfLog2Pi;
    }
    return (v >= a && v < b) ? -logf(b - a) : -INFINITY;
}

// lw[i] += scale * log_prob(dist(p0_i, p1_i); x_i)
__global__ __launch_bounds__(256) void logweight_kernel(int kind, const float* __restrict__ p0, int s0,
                                                        const float* __restrict__ p1, int s1,
                                                        const float* __restrict__ x, int sx, float scale,
                                                        float* __restrict__ lw, float* __restrict__ lp_out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float lp = term_log_prob(kind, p0, s0, p1, s1, x[(int64_t)i * sx], i);
    if (lp_out) lp_out[i] = lp;
    if (lw) lw[i] += scale * lp;
}

// The same for the particles of a diverged control-flow path: lw[r] += scale * log_prob(dist(p0_r, p1_r); x_r), r = rows[j]
// (one launch on m rows instead of log_prob over all n particles + zeros + where + add).
__global__ __launch_bounds__(256) void logweight_rows_kernel(int kind, const float* __restrict__ p0, int s0,
                                                             const float* __restrict__ p1, int s1,
                                                             const float* __restrict__ x, int sx, float scale,
                                                             float* __restrict__ lw, const int64_t* __restrict__ rows, int m) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int i = (int)rows[j];
    lw[i] += scale * term_log_prob(kind, p0, s0, p1, s1, x[(int64_t)i * sx], i);
}

// dst[r] = src[r * stride], r = rows[j]: what a path returns (or a shared scalar, stride 0) into the call's result vector
__global__ __launch_bounds__(256) void rows_copy_kernel(const float* __restrict__ src, int stride, float* __restrict__ dst,
                                                        const int64_t* __restrict__ rows, int m) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int64_t i = rows[j];
    dst[i] = src[i * stride];
}

// ---- a branch of the lock-step executor: the rows of a path split by a per-particle condition ------------------------------
// (PathExecutor.branch: `while s >= 1:` on the particles of a path.) Stable partition of the path's rows - all particles when
// `rows` is null - into those whose condition byte is non-zero and the others, both in ascending order, and the two counts:
// two launches and ONE 8-byte read-back where torch spent a masked sum + .item(), and two nonzero() (four launches and a
// synchronisation each). Tile = 1024 rows per workgroup; a thread owns 4 consecutive rows.
constexpr int PART_TILE = 1024;
__device__ __forceinline__ int wave_sum_int(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void partition_count_kernel(const uint8_t* __restrict__ cond, const int64_t* __restrict__ rows, int m,
                                                              int32_t* __restrict__ block_true) {
    __shared__ int sh[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j0 = blockIdx.x * PART_TILE + tid * 4;
    int c = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + q;
        if (j < m) c += cond[rows ? rows[j] : (int64_t)j] != 0;
    }
    c = wave_sum_int(c);
    if (lane == 0) sh[wave] = c;
    __syncthreads();
    if (tid == 0) block_true[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void partition_scatter_kernel(const uint8_t* __restrict__ cond, const int64_t* __restrict__ rows, int m,
                                                                const int32_t* __restrict__ block_true, int64_t* __restrict__ rows_true,
                                                                int64_t* __restrict__ rows_false, int32_t* __restrict__ counts,
                                                                const int seq) {
    __shared__ int sh[4], shw[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x;
    // true rows in the tiles before this one (the last workgroup also sums its own: the totals)
    int before = 0;
    for (int q = tid; q < b; q += 256) before += block_true[q];
    before = wave_sum_int(before);
    if (lane == 0) sh[wave] = before;
    __syncthreads();
    before = sh[0] + sh[1] + sh[2] + sh[3];
    const int j0 = b * PART_TILE + tid * 4;
    int64_t r[4];
    int f[4], mine = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + q;
        r[q] = j < m ? (rows ? rows[j] : (int64_t)j) : 0;
        f[q] = j < m ? (cond[r[q]] != 0) : 0;
        mine += f[q];
    }
    // exclusive prefix of `mine` over the workgroup's threads: inside the wave by shifts, across the waves through LDS
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) shw[wave] = incl;
    __syncthreads();
    int wave_before = 0;
    for (int w = 0; w < wave; ++w) wave_before += shw[w];
    int t_pos = before + wave_before + incl - mine;             // true rows before this thread's first row
    int f_pos = (j0 - t_pos);                                   // false rows before it (every earlier row is one or the other)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (j0 + q >= m) break;
        if (f[q]) rows_true[t_pos++] = r[q];
        else rows_false[f_pos++] = r[q];
    }
    if (b == gridDim.x - 1 && tid == 255) {      // the last thread of the last tile has seen every row
        const int total_true = before + wave_before + incl;
        counts[0] = total_true;
        counts[1] = m - total_true;
        if (seq != 0) {      // polled by the host in mapped (pinned) memory: the sequence word goes last, behind a system-scope fence
            __threadfence_system();
            __hip_atomic_store(counts + 2, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(float scale, const float* __restrict__ t, float* __restrict__ lw, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) lw[i] += scale * t[i];
}

// Importance statistics, ONE pass over the particles: workgroup b keeps its slice relative to its OWN maximum
//   m_b = max finite lw,  S_b = (sum e, sum e^2, sum e x, sum e x^2, count),  e = exp(lw - m_b)
// (fp64 exp of the difference <= 0, fp64 sums) and a one-workgroup combine rescales the partials to the global
// maximum: sum w = sum_b S_b[0] exp(m_b - M), sum w^2 = sum_b S_b[1] exp(2 (m_b - M)), ...
// (two passes over the data with 64 workgroups took 8 + 23 us for 1M particles; a single workgroup > 1 ms).
constexpr int STAT_BLOCKS = 256;      // scratch: STAT_BLOCKS x 6 doubles (PP_IS_STATS_SCRATCH)
constexpr int STAT_PER_THREAD = 16;   // particles per thread per tile, held in registers between the two sweeps

__global__ __launch_bounds__(256) void is_stats_partial_kernel(const float* __restrict__ lw, const float* __restrict__ x,
                                                               int n, double* __restrict__ scratch) {
    __shared__ float shmax[4];
    __shared__ double sh[4][5];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tile = 256 * STAT_PER_THREAD;
    double M = -INFINITY;   // running maximum of this workgroup
    double S[5] = {0, 0, 0, 0, 0};
    for (int base = blockIdx.x * tile; base < n; base += gridDim.x * tile) {
        float l[STAT_PER_THREAD], xv[STAT_PER_THREAD];
        float m = -INFINITY;
#pragma unroll
        for (int q = 0; q < STAT_PER_THREAD; ++q) {
            const int i = base + q * 256 + tid;
            l[q] = i < n ? lw[i] : -INFINITY;
            xv[q] = (x && i < n) ? x[i] : 0.0f;
            if (isfinite(l[q])) m = fmaxf(m, l[q]);   // Model._traces drops non-finite weights (model.py:65-68)
        }
        m = wave_max(m);
        if (lane == 0) shmax[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(shmax[0], shmax[1]), fmaxf(shmax[2], shmax[3]));
        __syncthreads();
        if (m == -INFINITY) continue;   // workgroup-uniform: no finite weight in this tile
        if ((double)m > M) {            // rescale what was accumulated so far (workgroup-uniform)
            const double r = M == -INFINITY ? 0.0 : exp(M - (double)m);
            S[0] *= r; S[1] *= r * r; S[2] *= r; S[3] *= r;
            M = (double)m;
        }
#pragma unroll
        for (int q = 0; q < STAT_PER_THREAD; ++q) {
            if (!isfinite(l[q])) continue;
            // fp64 exponent: Empirical / util.effective_sample_size normalise in float64 (empirical.py:300, util.py:398-399)
            const double e = exp((double)l[q] - M), xd = (double)xv[q];
            S[0] += e; S[1] += e * e; S[2] += e * xd; S[3] += e * xd * xd; S[4] += 1.0;
        }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const double r = wave_sum(S[q]);
        if (lane == 0) sh[wave][q] = r;
    }
    __syncthreads();
    if (tid < 5) scratch[blockIdx.x * 6 + 1 + tid] = sh[0][tid] + sh[1][tid] + sh[2][tid] + sh[3][tid];
    if (tid == 0) scratch[blockIdx.x * 6] = M;
}

// The combine of the per-workgroup partials (one workgroup of 256 threads).
__device__ __forceinline__ void stats_combine_body(const double* __restrict__ scratch, int nblocks, double* __restrict__ out,
                                                   double (&shm)[4], double (&sh)[4][5]) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    auto ld = [&](int i) -> double { return scratch[i]; };
    double m = -INFINITY;       // maximum over this thread's partials b = tid, tid + 256, ...
    for (int b = tid; b < nblocks; b += 256) m = fmax(m, ld(b * 6));
    double gm = m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gm = fmax(gm, __shfl_xor(gm, o, 64));
    if (lane == 0) shm[wave] = gm;
    __syncthreads();
    gm = fmax(fmax(shm[0], shm[1]), fmax(shm[2], shm[3]));
    double S[5] = {0, 0, 0, 0, 0};
    for (int b = tid; b < nblocks; b += 256) {
        const double mb = ld(b * 6);
        if (!(mb > -INFINITY)) continue;
        const double r = exp(mb - gm);
        S[0] += ld(b * 6 + 1) * r;
        S[1] += ld(b * 6 + 2) * r * r;
        S[2] += ld(b * 6 + 3) * r;
        S[3] += ld(b * 6 + 4) * r;
        S[4] += ld(b * 6 + 5);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const double r = wave_sum(S[q]);
        if (lane == 0) sh[wave][q] = r;
    }
    __syncthreads();
    // `out` may be host-mapped (pinned) memory that the caller polls instead of reading the statistics back with a copy: the
    // count - negative while the caller waits - is stored LAST, behind a system-scope fence
    if (tid < 4) out[1 + tid] = sh[0][tid] + sh[1][tid] + sh[2][tid] + sh[3][tid];
    if (tid == 0) out[0] = gm;
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(out + 5, sh[0][4] + sh[1][4] + sh[2][4] + sh[3][4], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void is_stats_combine_kernel(const double* __restrict__ scratch, int nblocks,
                                                               double* __restrict__ out) {
    __shared__ double shm[4];
    __shared__ double sh[4][5];
    stats_combine_body(scratch, nblocks, out, shm, sh);
}

// Several log-weight terms in one pass over the particles (state.py:211-217, 147-149):
//   lw[i] += sum_t scale_t * term_t(i),  term kinds: term_log_prob's, and 2 = identity (x itself)
struct LwTerm {
    int kind, s0, s1, sx;
    const float *p0, *p1, *x;
    float scale;
};
struct LwTerms {
    LwTerm t[4];
    int count;
};

__global__ __launch_bounds__(256) void logweight_multi_kernel(const LwTerms terms, float* __restrict__ lw, int n,
                                                              int overwrite) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = overwrite ? 0.0f : lw[i];
    for (int q = 0; q < terms.count; ++q) {
        const LwTerm& t = terms.t[q];
        const float v = t.x[(int64_t)i * t.sx];
        float lp;
        if (t.kind == 2)
            lp = v;
        else
            lp = term_log_prob(t.kind, t.p0, t.s0, t.p1, t.s1, v, i);
        acc += t.scale * lp;
    }
    lw[i] = acc;
}

// ---- one pass per posterior statement -------------------------------------------------------------------------------
// Draw from the (shared) proposal, log q, the prior term and every queued likelihood term, the new log-weight and the
// importance statistics in ONE grid-stride kernel: a particle costs a Philox block, the draw, K fused multiply-adds + exps
// and a handful of log-probs; memory sees 8 bytes per particle (value and log-weight, written once). The per-component
// constants of the proposal are computed once per workgroup. The statistics use the scheme of is_stats_partial_kernel
// (per-workgroup maximum, fp64 sums relative to it, rescaled by is_stats_combine_kernel).
//   term parameters flagged "value" read the particle's freshly drawn value (e.g. the mean of the likelihood Normal(mu, s)
//   of observe statements that follow `mu = sample(...)`), so the program's statements up to the next sample are one pass.
constexpr int FUSED_PT = 4;          // particles per thread at 1M particles (grid sizing)
constexpr int FUSED_BLOCKS = 1024;   // workgroups at most (4 per CU: the per-particle chain is long and latency-bound)
constexpr int FUSED_MAX_TERMS = 8;
struct FusedTerm {
    int kind, s0, s1, sx, flags;     // flags: 1 p0 = value, 2 p1 = value, 4 x = value
    const float *p0, *p1, *x;
    float scale;
};
struct FusedTerms {
    FusedTerm t[FUSED_MAX_TERMS];
    int count;
};

// LEAN: every term is a Normal with ONE scale for all particles (or the identity) - the terms of a program whose observes are
// Normal(f(latent), sigma): the term loop is then a handful of instructions instead of eight unrolled copies of the general
// log-prob switch (9 600 lines of ISA, 128 VGPRs and two spilled quads for the general kernel).
// KC: the number of mixture components as a compile-time constant (10, pyprob's default - nn/proposal_*_mixture.py
// mixture_components = 10), 0 = read K: with KC the component loops are straight-line code without the sixteen k < K guards.
template <int KIND, bool LEAN, int KC = 0>      // KIND -1: no draw (values are read), 0 / 1 / 2: mixture head kinds as is_mixture_shared_kernel
__global__ __launch_bounds__(256, 4) void is_fused_kernel(const float* __restrict__ y, const float* __restrict__ prior, int n, int K_rt,
                                                       const FusedTerms terms, float* __restrict__ value,
                                                       float* __restrict__ lw, int overwrite, uint64_t seed, uint64_t offset,
                                                       double* __restrict__ scratch) {
    __shared__ float s_mu[MAXK], s_sd[MAXK], s_inv[MAXK], s_c[MAXK], s_cum[MAXK], s_ca[MAXK], s_cb[MAXK];
    __shared__ float shmax[4];
    __shared__ double sh[4][5];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int K = KC ? KC : K_rt;
    float pa = 0.0f, pb = 1.0f;
    if (KIND >= 0) {
        pa = prior[0]; pb = prior[1];
        if (wave == 0) {       // the proposal's components: the arithmetic (and summation order) of is_mixture_shared_kernel, the
            const int k = lane;   // softmax terms computed ONCE - lane j holds term j, the sums walk the lanes in order
            const float z = k < K ? y[2 * K + k] : -INFINITY;
            float zmax = -INFINITY;
            for (int j = 0; j < K; ++j) zmax = fmaxf(zmax, __shfl(z, j, 64));
            const float e = k < K ? expf(z - zmax) : 0.0f;
            float zs = 0.0f;
            for (int j = 0; j < K; ++j) zs += __shfl(e, j, 64);
            const float q = e / zs;
            float ps = 0.0f;
            for (int j = 0; j < K; ++j) ps += __shfl(q, j, 64);
            const float pk = q / ps;
            float cum = 0.0f;
            for (int j = 0; j < K; ++j) {
                const float rj = __shfl(pk, j, 64);
                if (j <= k) cum += rj;
            }
            if (k < K) {
                float mu, sd;
                if (KIND == 0) {
                    mu = pa + y[k] * pb;
                    sd = expf(y[K + k]) * pb;
                } else {
                    const float rng = pb - pa;
                    mu = pa + sigmoidf_(y[k]) * rng;
                    sd = KIND == 2 ? expf(y[K + k]) : rng / 1000.0f + sigmoidf_(y[K + k]) * rng * 10.0f;
                }
                const float lpk = logf(fminf(fmaxf(pk, kFp32Eps), 1.0f - kFp32Eps));
                s_mu[k] = mu; s_sd[k] = sd; s_inv[k] = 1.0f / sd; s_cum[k] = cum;
                if (KIND == 0) {
                    s_c[k] = lpk - logf(sd) - kHalfLog2Pi;
                    s_ca[k] = s_cb[k] = 0.0f;
                } else {
                    const float ca = std_cdf((pa - mu) / sd), cb = std_cdf((pb - mu) / sd);
                    s_ca[k] = ca; s_cb[k] = cb;
                    s_c[k] = lpk - kHalfLog2Pi - logf(sd * (cb - ca));
                }
            }
        }
    }
    // Normal terms whose scale is one number for all particles: their constants once per workgroup (LDS; by the second wave,
    // next to the first wave's softmax - one barrier for both).
    __shared__ float s_tc0[FUSED_MAX_TERMS], s_tc1[FUSED_MAX_TERMS];
    __shared__ int s_tcok[FUSED_MAX_TERMS];
    if (tid >= 64 && tid < 64 + FUSED_MAX_TERMS) {
        const int t = tid - 64;
        s_tcok[t] = 0; s_tc0[t] = 0.0f; s_tc1[t] = 0.0f;
        if (t < terms.count && terms.t[t].kind == 0 && !(terms.t[t].flags & 2) && terms.t[t].s1 == 0) {
            const float b = terms.t[t].p1[0];
            s_tcok[t] = 1;
            s_tc0[t] = -logf(b) - kHalfLog2Pi;
            s_tc1[t] = 1.0f / (2.0f * b * b);
        }
    }
    __syncthreads();
    // statistics: every thread keeps (m, sums relative to m) of ITS particles - a new maximum rescales the thread's sums
    // (rare after its first particles) - and the workgroup's threads meet once at the end; no per-tile barrier, no per-
    // particle arrays: the loop body is one particle, 8 waves per SIMD hide its latency chain
    float m_t = -INFINITY;
    double S[5] = {0, 0, 0, 0, 0};
    for (int i = blockIdx.x * 256 + tid; i < n; i += gridDim.x * 256) {
            float v, acc = overwrite ? 0.0f : lw[i];
        if (KIND < 0) {
            v = value[i];
        } else {
            Philox rng(seed, offset + (uint64_t)i, 0x1C);
            v = NAN;
            for (int attempt = 0; attempt < 64; ++attempt) {
                uint32_t r[4];
                rng.next(r);
                const float u0 = u01(r[0]), u1 = u01(r[1]), u2 = u01(r[2]);
                int kk = K - 1;
                if (KC) {
#pragma unroll
                    for (int k = KC - 2; k >= 0; --k)
                        if (u0 < s_cum[k]) kk = k;
                } else {
                    for (int k = K - 2; k >= 0; --k)
                        if (u0 < s_cum[k]) kk = k;
                }
                const float mk = s_mu[kk], sk = s_sd[kk];
                if (KIND == 0) {
                    v = mk + sk * box_muller_fast(u1, u2);
                    break;
                } else {
                    const float uu = s_ca[kk] + u1 * (s_cb[kk] - s_ca[kk]);
                    v = mk + sk * kSqrt2 * erfinvf(2.0f * uu - 1.0f);
                    if (isfinite(v) && v >= pa && v < pb) break;
                    v = NAN;
                }
            }
            const bool inside = (KIND == 0) || (v >= pa && v <= pb);
            float a[MAXK], amax = -INFINITY;
#pragma unroll
            for (int k = 0; k < MAXK; ++k)
                if (k < K) {
                    const float t = (v - s_mu[k]) * s_inv[k];
                    a[k] = inside ? s_c[k] - 0.5f * t * t : -INFINITY;
                    amax = fmaxf(amax, a[k]);
                }
            float lq = amax;
            if (amax > -INFINITY) {      // (hardware exp2 / log2: ~1e-6 absolute on log q, the 1e-4 bar is on log-weights)
                float sum = 0.0f;
#pragma unroll
                for (int k = 0; k < MAXK; ++k)
                    if (k < K) sum += __expf(a[k] - amax);
                lq = amax + __logf(sum);
            }
            acc -= lq;                 // - log q(v)   (state.py:212, 217)
            value[i] = v;
        }
        if (LEAN) {
#pragma unroll 1
            for (int t = 0; t < terms.count; ++t) {
                const FusedTerm& T = terms.t[t];
                float lp = (T.flags & 4) ? v : T.x[(int64_t)i * T.sx];
                if (T.kind == 0) {
                    const float d = lp - ((T.flags & 1) ? v : T.p0[(int64_t)i * T.s0]);
                    lp = s_tc0[t] - d * d * s_tc1[t];
                }
                acc += T.scale * lp;
            }
        } else
#pragma unroll
        for (int t = 0; t < FUSED_MAX_TERMS; ++t) {
            if (t >= terms.count) break;
            const FusedTerm& T = terms.t[t];
            const float x = (T.flags & 4) ? v : T.x[(int64_t)i * T.sx];
            float lp;
            if (T.kind == 2) {
                lp = x;
            } else if (T.kind == 0 || T.kind == 1) {      // two parameters, either may BE the drawn value
                const float pa_ = (T.flags & 1) ? v : T.p0[(int64_t)i * T.s0];
                const float pb_ = (T.flags & 2) ? v : T.p1[(int64_t)i * T.s1];
                if (T.kind == 0) {
                    const float d = x - pa_;
                    if (s_tcok[t]) {      // constant scale: - log b - log sqrt(2 pi) and 1 / (2 b^2) once per thread
                        lp = s_tc0[t] - d * d * s_tc1[t];
                    } else {
                        lp = -(d * d) / (2.0f * pb_ * pb_) - logf(pb_) - kHalfLog2Pi;
                    }
                } else {
                    lp = (x >= pa_ && x < pb_) ? -logf(pb_ - pa_) : -INFINITY;
                }
            } else {
                lp = term_log_prob(T.kind, T.p0, T.s0, T.p1, T.s1, x, i);
            }
            acc += T.scale * lp;
        }
        lw[i] = acc;
        if (scratch && isfinite(acc)) {      // Model._traces drops non-finite weights (model.py:65-68)
            if (acc > m_t) {
                const double r = m_t == -INFINITY ? 0.0 : (double)expf(m_t - acc);
                S[0] *= r; S[1] *= r * r; S[2] *= r; S[3] *= r;
                m_t = acc;
            }
            // fp32 exponent of an exact fp32 difference <= 0 (relative error 1e-7 per weight), float64 sums
            const double e = (double)expf(acc - m_t), xd = (double)v;
            S[0] += e; S[1] += e * e; S[2] += e * xd; S[3] += e * xd * xd; S[4] += 1.0;
        }
    }
    if (!scratch) return;
    float M = wave_max(m_t);
    if (lane == 0) shmax[wave] = M;
    __syncthreads();
    M = fmaxf(fmaxf(shmax[0], shmax[1]), fmaxf(shmax[2], shmax[3]));
    {
        const double r = (m_t == -INFINITY || M == -INFINITY) ? 0.0 : (double)expf(m_t - M);
        S[0] *= r; S[1] *= r * r; S[2] *= r; S[3] *= r;
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const double r = wave_sum(S[q]);
        if (lane == 0) sh[wave][q] = r;
    }
    __syncthreads();
    if (tid < 5) scratch[blockIdx.x * 6 + 1 + tid] = sh[0][tid] + sh[1][tid] + sh[2][tid] + sh[3][tid];
    if (tid == 0) scratch[blockIdx.x * 6] = (double)M;
    // (Folding the combine into this launch - the last workgroup to arrive behind a device-scope ticket - was tried twice: behind
    // a release fence per workgroup (round 4: each fence writes back the 8 MB of values and log-weights it shares the L2 with,
    // 18 -> 80 us per launch) and with write-through partials + agent-scope atomic loads, no fence (round 5: the posterior call
    // 70.8 -> 77.7 us at 10^6 particles, unchanged at 1 000: the last workgroup's loads queue behind the write stream);
    // profiles/r04_experiments_not_kept.txt, r05_experiments_not_kept.txt. The one-workgroup is_stats_combine_kernel follows.)
}

// ---- prior draws for vectorised trace generation (pyprob/nn/dataset.py:50-62, state.py:278-290 run n times) -----------
// out[i] ~ Normal(p0, p1) (kind 0, Box-Muller) | Uniform[p0, p1) (kind 1), parameters shared (stride 0) or per trace;
// Philox counter = offset + i, key = seed, one stream id per statement: the columns of a chunk of prior traces are
// drawn where the training step reads them, not on a host thread.
__global__ __launch_bounds__(256) void prior_draw_kernel(int kind, const float* __restrict__ p0, int s0,
                                                         const float* __restrict__ p1, int s1, int n, uint64_t seed,
                                                         uint64_t offset, uint32_t stream_id, float* __restrict__ out) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        Philox rng(seed, offset + (uint64_t)i, stream_id);
        uint32_t r[4];
        rng.next(r);
        const float a = p0[(int64_t)i * s0], b = p1[(int64_t)i * s1];
        float v;
        if (kind == 0) {
            v = a + b * sqrtf(-2.0f * logf(u01(r[0]))) * cosf(kTwoPi * u01(r[1]));
        } else {
            v = a + (b - a) * (((float)(r[0] >> 8)) * (1.0f / 16777216.0f));      // [a, b): torch.distributions.Uniform's support
            v = v < b ? v : a;
        }
        out[i] = v;
    }
}

// kinds 0, 1 read p1; 3, 4 ignore it; 5 carries the number of categories in p1_stride
static inline bool lw_kind_ok(int kind, const float* p1, int p1_stride) {
    if (kind == 0 || kind == 1) return p1 != nullptr;
    if (kind == 3 || kind == 4 || kind == 2) return true;
    return kind == 5 && p1_stride >= 1;
}

}  // namespace pp

extern "C" {

size_t pp_is_workspace_bytes(const pp_net* net, int32_t n) {
    if (!net) return 0;
    pp::IsWorkspace w;
    pp::is_carve(net, n, nullptr, w);
    return w.bytes;
}

int pp_is_init(const pp_net* net, const float* params, const float* obs, float* e_out, void* workspace,
               size_t workspace_bytes, void* stream) {
    // kernel class 6 of the in-stream timing: the device chain of a posterior call's shared first statement, from the observe
    // embedding (here) to the end of pp_is_fused (which closes the bracket) - what the call's wall time is compared with
    pp::prof_begin(6, pp::as_stream(stream));
    return pp::is_init(net, params, obs, e_out, workspace, workspace_bytes, pp::as_stream(stream));
}

int pp_is_first_statement_supported(const pp_net* net, int32_t addr_id) { return pp::is_first_statement_supported(net, addr_id) ? 1 : 0; }

int pp_is_first_statement(const pp_net* net, const float* params, const float* obs, int32_t addr_id, float* e_out, float* h, float* c,
                          void* workspace, size_t workspace_bytes, void* stream) {
    pp::prof_begin(6, pp::as_stream(stream));      // (the bracket pp_is_init opens: the device chain of a posterior call)
    return pp::is_first_statement(net, params, obs, addr_id, e_out, h, c, workspace, workspace_bytes, pp::as_stream(stream));
}

int pp_is_step(const pp_net* net, const float* params, int32_t addr_id, int32_t prev_addr_id, int32_t n,
               const float* e_obs_vec, const float* prev_value, const float* prior, int32_t prior_stride, float* h,
               float* c, int32_t state_rows, const float* value_in, float* value_out, float* logq_out, uint64_t seed,
               uint64_t offset, void* workspace, size_t workspace_bytes, void* stream) {
    return pp::is_step(net, params, addr_id, prev_addr_id, n, e_obs_vec, prev_value, prior, prior_stride, h, c, state_rows,
                       value_in,
                       value_out, logq_out, seed, offset, workspace, workspace_bytes, pp::as_stream(stream));
}

int pp_is_step_rows(const pp_net* net, const float* params, int32_t addr_id, int32_t prev_addr_id, int32_t n,
                    const float* e_obs_vec, const float* prev_value, const float* prior, int32_t prior_stride, float* h,
                    float* c, int32_t state_rows, const int64_t* rows, const float* value_in, float* value_out,
                    float* logq_out, uint64_t seed, uint64_t offset, void* workspace, size_t workspace_bytes, void* stream) {
    if (prev_addr_id < 0 && rows) {
        pp::set_error("pp_is_step_rows: the first statement of a trace has one shared state row (no index list)");
        return PP_EINVAL;
    }
    return pp::is_step(net, params, addr_id, prev_addr_id, n, e_obs_vec, prev_value, prior, prior_stride, h, c, state_rows,
                       value_in, value_out, logq_out, seed, offset, workspace, workspace_bytes, pp::as_stream(stream), false,
                       rows);
}

int pp_is_statement_rows(const pp_net* net, const float* params, int32_t addr_id, int32_t prev_addr_id, int32_t n,
                         const float* e_obs_vec, const float* prev_value_full, const float* prior, int32_t prior_stride,
                         float* h, float* c, int32_t state_rows, const int64_t* rows, float* value_full, float* lw_full,
                         int32_t prior_kind, uint64_t seed, uint64_t offset, void* workspace, size_t workspace_bytes,
                         void* stream) {
    if (prev_addr_id < 0 || !(value_full && lw_full && prev_value_full) || (prior_kind != 0 && prior_kind != 1) || !prior) {
        pp::set_error("pp_is_statement_rows: a statement after the first one, Normal (0) or Uniform (1) prior, value / log-weight / "
                      "previous-value vectors indexed by the particles' rows");
        return PP_EINVAL;
    }
    if (net && addr_id >= 0 && addr_id < net->n_addr) {
        const int kind = net->addrs[addr_id].kind;
        if (kind != PP_HEAD_NORMAL_MIXTURE && kind != PP_HEAD_TRUNCNORMAL_MIXTURE) {
            pp::set_error("pp_is_statement_rows: Normal / Uniform statements (mixture heads) only");
            return PP_EINVAL;
        }
    }
    const pp::IsStatementOut whole{value_full, lw_full, prior_kind};
    return pp::is_step(net, params, addr_id, prev_addr_id, n, e_obs_vec, prev_value_full, prior, prior_stride, h, c, state_rows,
                       nullptr, nullptr, nullptr, seed, offset, workspace, workspace_bytes, pp::as_stream(stream), false, rows, &whole);
}

int pp_is_step_fused_supported(const pp_net* net, int32_t addr_id, int32_t n) {
    // (H = 256 / 512: any n - small launches take the two-launch split statement; H = 1024: from WIDE_MIN_ROWS + 1 particles on)
    return (pp::is_step_fused_mode() != 0 && pp::is_step_fused_supported(net, addr_id) && pp::is_step_fused_pays(net, n)) ? 1 : 0;
}

int pp_prior_draw(int32_t kind, const float* p0, int32_t p0_stride, const float* p1, int32_t p1_stride, int32_t n, uint64_t seed,
                  uint64_t offset, uint32_t stream_id, float* out, void* stream) {
    if (!(p0 && p1 && out) || (kind != 0 && kind != 1)) {
        pp::set_error("pp_prior_draw: Normal (0) or Uniform (1) with two parameter vectors");
        return PP_EINVAL;
    }
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pp::prior_draw_kernel, dim3(std::min(2048, pp::cdiv(n, 256))), dim3(256), 0, pp::as_stream(stream), kind, p0,
                       p0_stride, p1, p1_stride, n, seed, offset, stream_id, out);
    PP_LAUNCH_CHECK("pp_prior_draw");
    return 0;
}

int pp_is_step_net(const pp_net* net, const float* params, int32_t addr_id, int32_t prev_addr_id, int32_t n,
                   const float* e_obs_vec, const float* prev_value, float* h, float* c, int32_t state_rows, void* workspace,
                   size_t workspace_bytes, void* stream) {
    return pp::is_step(net, params, addr_id, prev_addr_id, n, e_obs_vec, prev_value, nullptr, 0, h, c, state_rows, nullptr,
                       nullptr, nullptr, 0, 0, workspace, workspace_bytes, pp::as_stream(stream), /*net_only=*/true);
}

int pp_is_fused(const pp_net* net, int32_t addr_id, int32_t n, const float* prior, const pp_lw_term* terms,
                const int32_t* term_flags, int32_t n_terms, float* value, float* lw, int32_t overwrite, uint64_t seed,
                uint64_t offset, double* stats_out, double* stats_scratch, void* workspace, size_t workspace_bytes,
                void* stream) {
    if (!(value && lw) || n_terms < 0 || n_terms > pp::FUSED_MAX_TERMS || (n_terms && !terms) || (stats_out && !stats_scratch)) {
        pp::set_error("pp_is_fused: bad argument (at most %d terms; statistics need scratch)", pp::FUSED_MAX_TERMS);
        return PP_EINVAL;
    }
    if (n <= 0) return 0;
    pp::FusedTerms t;
    t.count = n_terms;
    bool lean = true;       // Normal terms with one scale for all particles (is_fused_kernel's LEAN term loop)
    for (int q = 0; q < n_terms; ++q) {
        const pp_lw_term& s = terms[q];
        const int fl = term_flags ? term_flags[q] : 0;
        const bool two = s.kind == 0 || s.kind == 1;
        if ((!(fl & 4) && !s.x) || (s.kind != 2 && !((fl & 1) || s.p0)) || (two && !((fl & 2) || s.p1)) ||
            ((fl & 3) && !two) || (s.kind == 5 && s.p1_stride < 1) || s.kind < 0 || s.kind > 5) {
            pp::set_error("pp_is_fused: bad term %d", q);
            return PP_EINVAL;
        }
        t.t[q] = pp::FusedTerm{s.kind, s.p0_stride, s.p1_stride, s.x_stride, fl, s.p0, s.p1, s.x, s.scale};
        lean = lean && (s.kind == 2 || (s.kind == 0 && !(fl & 2) && s.p1_stride == 0));
    }
    const int tile = 256 * pp::FUSED_PT;
    const int blocks = std::min(pp::FUSED_BLOCKS, pp::cdiv(n, tile));
    hipStream_t st = pp::as_stream(stream);
    double* scratch = stats_out ? stats_scratch : nullptr;
    pp::prof_begin(4, st);
#define PP_FUSED_LAUNCH(KIND, Y, PRIOR, KK)                                                                                  \
    do {                                                                                                                     \
        if (lean && KK == 10 && (KIND == 0 || KIND == 1))                                                                    \
            hipLaunchKernelGGL((pp::is_fused_kernel<KIND, true, (KIND == 0 || KIND == 1) ? 10 : 0>), dim3(blocks), dim3(256), 0, st, Y,  \
                               PRIOR, n, KK, t, value, lw, overwrite, seed, offset, scratch);                               \
        else if (lean)                                                                                                       \
            hipLaunchKernelGGL((pp::is_fused_kernel<KIND, true>), dim3(blocks), dim3(256), 0, st, Y, PRIOR, n, KK, t, value, lw,  \
                               overwrite, seed, offset, scratch);                                                            \
        else                                                                                                                 \
            hipLaunchKernelGGL((pp::is_fused_kernel<KIND, false>), dim3(blocks), dim3(256), 0, st, Y, PRIOR, n, KK, t, value, lw, \
                               overwrite, seed, offset, scratch);                                                            \
    } while (0)
    if (addr_id < 0) {
        PP_FUSED_LAUNCH(-1, nullptr, nullptr, 0);
    } else {
        if (!(net && prior && workspace) || addr_id >= net->n_addr) {
            pp::set_error("pp_is_fused: a draw needs the network, the prior parameters and the workspace of pp_is_step_net");
            return PP_EINVAL;
        }
        const pp_addr& ad = net->addrs[addr_id];
        if (!(ad.kind == PP_HEAD_NORMAL_MIXTURE || ad.kind == PP_HEAD_TRUNCNORMAL_MIXTURE || ad.kind == PP_HEAD_POISSON_TN_MIXTURE) ||
            ad.n_out % 3 != 0 || ad.n_out / 3 > pp::MAXK) {
            pp::set_error("pp_is_fused: mixture heads only");
            return PP_EINVAL;
        }
        pp::IsWorkspace w;
        pp::is_carve(net, 1, workspace, w);       // the shared row's head outputs, left there by pp_is_step_net
        if (w.bytes > workspace_bytes) {
            pp::set_error("pp_is_fused: workspace too small");
            return PP_ENOSPACE;
        }
        const int K = ad.n_out / 3;
        if (ad.kind == PP_HEAD_NORMAL_MIXTURE)
            PP_FUSED_LAUNCH(0, w.Y, prior, K);
        else if (ad.kind == PP_HEAD_TRUNCNORMAL_MIXTURE)
            PP_FUSED_LAUNCH(1, w.Y, prior, K);
        else
            PP_FUSED_LAUNCH(2, w.Y, prior, K);
    }
#undef PP_FUSED_LAUNCH
    pp::prof_end(4, 8.0 * n, st);
    if (stats_out)
        hipLaunchKernelGGL(pp::is_stats_combine_kernel, dim3(1), dim3(256), 0, st, stats_scratch, blocks, stats_out);
    pp::prof_end(6, 0.0, st);
    PP_LAUNCH_CHECK("pp_is_fused");
    return 0;
}

int pp_logweight_accumulate(int32_t kind, const float* p0, int32_t p0_stride, const float* p1, int32_t p1_stride,
                            const float* x, int32_t x_stride, float scale, float* lw, float* lp_out, int32_t n,
                            void* stream) {
    if (!(p0 && x) || !pp::lw_kind_ok(kind, p1, p1_stride) || kind == 2) {
        pp::set_error("pp_logweight_accumulate: bad argument");
        return PP_EINVAL;
    }
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pp::logweight_kernel, dim3(pp::cdiv(n, 256)), dim3(256), 0, pp::as_stream(stream), kind, p0,
                       p0_stride, p1, p1_stride, x, x_stride, scale, lw, lp_out, n);
    PP_LAUNCH_CHECK("pp_logweight_accumulate");
    return 0;
}

int pp_logweight_accumulate_rows(int32_t kind, const float* p0, int32_t p0_stride, const float* p1, int32_t p1_stride,
                                 const float* x, int32_t x_stride, float scale, float* lw, const int64_t* rows, int32_t m,
                                 void* stream) {
    if (!(p0 && x && lw && rows) || !pp::lw_kind_ok(kind, p1, p1_stride) || kind == 2) {
        pp::set_error("pp_logweight_accumulate_rows: bad argument");
        return PP_EINVAL;
    }
    if (m <= 0) return 0;
    hipLaunchKernelGGL(pp::logweight_rows_kernel, dim3(pp::cdiv(m, 256)), dim3(256), 0, pp::as_stream(stream), kind, p0, p0_stride,
                       p1, p1_stride, x, x_stride, scale, lw, rows, m);
    PP_LAUNCH_CHECK("pp_logweight_accumulate_rows");
    return 0;
}

int pp_copy_rows(const float* src, int32_t src_stride, float* dst, const int64_t* rows, int32_t m, void* stream) {
    if (!(src && dst && rows) || (src_stride != 0 && src_stride != 1)) {
        pp::set_error("pp_copy_rows: bad argument");
        return PP_EINVAL;
    }
    if (m <= 0) return 0;
    hipLaunchKernelGGL(pp::rows_copy_kernel, dim3(pp::cdiv(m, 256)), dim3(256), 0, pp::as_stream(stream), src, src_stride, dst, rows, m);
    PP_LAUNCH_CHECK("pp_copy_rows");
    return 0;
}

static int partition_rows(const uint8_t* cond, const int64_t* rows, int32_t m, int64_t* rows_true, int64_t* rows_false,
                          int32_t* counts, int32_t* scratch, int seq, void* stream) {
    if (!(cond && rows_true && rows_false && counts && scratch) || m < 0) {
        pp::set_error("pp_partition_rows: bad argument (scratch of PP_PARTITION_SCRATCH(m) int32 is required)");
        return PP_EINVAL;
    }
    hipStream_t st = pp::as_stream(stream);
    if (m == 0) {
        if (seq != 0) {
            pp::set_error("pp_partition_rows_polled: m = 0 needs no launch (the caller knows the counts)");
            return PP_EINVAL;
        }
        (void)hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), st);
        return 0;
    }
    const int blocks = pp::cdiv(m, pp::PART_TILE);
    hipLaunchKernelGGL(pp::partition_count_kernel, dim3(blocks), dim3(256), 0, st, cond, rows, m, scratch);
    hipLaunchKernelGGL(pp::partition_scatter_kernel, dim3(blocks), dim3(256), 0, st, cond, rows, m, scratch, rows_true, rows_false,
                       counts, seq);
    PP_LAUNCH_CHECK("pp_partition_rows");
    return 0;
}

int pp_partition_rows(const uint8_t* cond, const int64_t* rows, int32_t m, int64_t* rows_true, int64_t* rows_false,
                      int32_t* counts, int32_t* scratch, void* stream) {
    return partition_rows(cond, rows, m, rows_true, rows_false, counts, scratch, 0, stream);
}

int pp_partition_rows_polled(const uint8_t* cond, const int64_t* rows, int32_t m, int64_t* rows_true, int64_t* rows_false,
                             int32_t* counts, int32_t seq, int32_t* scratch, void* stream) {
    if (seq == 0) {
        pp::set_error("pp_partition_rows_polled: seq must be non-zero");
        return PP_EINVAL;
    }
    return partition_rows(cond, rows, m, rows_true, rows_false, counts, scratch, seq, stream);
}

int pp_axpy(float scale, const float* term, float* lw, int32_t n, void* stream) {
    if (!(term && lw)) return PP_EINVAL;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pp::axpy_kernel, dim3(pp::cdiv(n, 256)), dim3(256), 0, pp::as_stream(stream), scale, term, lw, n);
    PP_LAUNCH_CHECK("pp_axpy");
    return 0;
}

int pp_is_stats(const float* lw, const float* x, int32_t n, double* out, double* scratch, void* stream) {
    if (!(lw && out && scratch) || n <= 0) {
        pp::set_error("pp_is_stats: bad argument (scratch of PP_IS_STATS_SCRATCH doubles is required)");
        return PP_EINVAL;
    }
    const int tile = 256 * pp::STAT_PER_THREAD;
    const int blocks = std::min(pp::STAT_BLOCKS, pp::cdiv(n, tile));
    hipLaunchKernelGGL(pp::is_stats_partial_kernel, dim3(blocks), dim3(256), 0, pp::as_stream(stream), lw, x, n, scratch);
    hipLaunchKernelGGL(pp::is_stats_combine_kernel, dim3(1), dim3(256), 0, pp::as_stream(stream), scratch, blocks, out);
    PP_LAUNCH_CHECK("pp_is_stats");
    return 0;
}

int pp_logweight_terms(const pp_lw_term* terms, int32_t count, float* lw, int32_t n, int32_t overwrite, void* stream) {
    if (!(terms && lw) || count < 1 || count > 4) {
        pp::set_error("pp_logweight_terms: 1..4 terms");
        return PP_EINVAL;
    }
    if (n <= 0) return 0;
    pp::LwTerms t;
    t.count = count;
    for (int q = 0; q < count; ++q) {
        const pp_lw_term& s = terms[q];
        if (!s.x || (s.kind != 2 && !(s.p0 && pp::lw_kind_ok(s.kind, s.p1, s.p1_stride)))) {
            pp::set_error("pp_logweight_terms: bad term %d", q);
            return PP_EINVAL;
        }
        t.t[q] = pp::LwTerm{s.kind, s.p0_stride, s.p1_stride, s.x_stride, s.p0, s.p1, s.x, s.scale};
    }
    hipLaunchKernelGGL(pp::logweight_multi_kernel, dim3(pp::cdiv(n, 256)), dim3(256), 0, pp::as_stream(stream), t, lw, n,
                       overwrite);
    PP_LAUNCH_CHECK("pp_logweight_terms");
    return 0;
}

}  // extern "C"
