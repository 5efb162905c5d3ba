// Device-side proposal draws shared by the importance-sampling kernels (is_kernels.hip, is_step_fused.hip):
// Philox4x32-10 counters, the mixture draw + log q of one particle (Mixture.sample / Mixture.log_prob,
// pyprob/distributions/mixture.py:38-63; TruncatedNormal, distributions/truncated_normal.py:25-30, 94-112; proposal
// heads proposal_normal_normal_mixture.py:20-35, proposal_uniform_truncated_normal_mixture.py:24-35,
// proposal_poisson_truncated_normal_mixture.py).
#pragma once
#include "common.hpp"

#include <math.h>

namespace pp {

constexpr int MAXK = 16;
constexpr float kFp32Eps = 1.1920928955078125e-07f;
constexpr float kHalfLog2Pi = 0.91893853320467274178f;
constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kSqrt2 = 1.41421356237309504880f;
constexpr float kTwoPi = 6.28318530717958647692f;

__device__ __forceinline__ float std_cdf(float x) { return 0.5f * (1.0f + erff(x * kInvSqrt2)); }

// ---- Philox4x32-10 (Salmon et al. 2011), counter = particle index, key = seed ----------------------------
struct Philox {
    uint32_t c[4], k[2];
    __device__ __forceinline__ Philox(uint64_t seed, uint64_t ctr, uint32_t stream) {
        c[0] = (uint32_t)ctr; c[1] = (uint32_t)(ctr >> 32); c[2] = stream; c[3] = 0;
        k[0] = (uint32_t)seed; k[1] = (uint32_t)(seed >> 32);
    }
    __device__ __forceinline__ void next(uint32_t out[4]) {
        uint32_t x0 = c[0], x1 = c[1], x2 = c[2], x3 = c[3], k0 = k[0], k1 = k[1];
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x2;
            const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, y1 = (uint32_t)p1;
            const uint32_t y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1, y3 = (uint32_t)p0;
            x0 = y0; x1 = y1; x2 = y2; x3 = y3;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        out[0] = x0; out[1] = x1; out[2] = x2; out[3] = x3;
        c[3]++;  // next block of four for this particle
    }
};
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// Standard normal deviate of the SHARED-proposal kernels (the first statement of a lock-step run: is_mixture_shared_kernel and
// is_fused_kernel, one draw per particle, 10^6 particles per call): sqrt(-2 ln u1) cos(2 pi u2) on the hardware log2 / cos /
// sqrt (v_log_f32, v_cos_f32 takes its argument in revolutions, v_sqrt_f32: ~1e-6 absolute on a deviate of unit scale, ~60
// issue slots fewer than the library forms). It shapes only WHICH value is drawn; log q, log p and the likelihood terms are
// evaluated at the value that was drawn, with the accurate forms. Both kernels share this function, so the same Philox block
// gives bit-identical values on the fused and on the per-term path (tests/test_gpu_is_fused.py).
__device__ __forceinline__ float box_muller_fast(float u1, float u2) {
    return __builtin_amdgcn_sqrtf(-2.0f * __logf(u1)) * __builtin_amdgcn_cosf(u2);
}

// One particle of a mixture head. KIND 0: Normal mixture around a Normal prior (pa, pb) = (mean, stddev); KIND 1:
// TruncatedNormal mixture inside a Uniform prior (low, high); KIND 2: the Poisson head (TruncatedNormal mixture on [0, 40],
// stddev = exp(y)). y = the 3K head outputs of the particle (means | scales | logits). Draws v (Philox counter `ctr`,
// stream 0x1C) unless has_value, returns log q(v) in lp.
template <int KIND>
__device__ __forceinline__ void mixture_particle(const float* __restrict__ y, const float pa, const float pb, const int K,
                                                 const bool has_value, const float v_in, const uint64_t seed,
                                                 const uint64_t ctr, float& v_out, float& lp_out) {
    float mu[MAXK], sd[MAXK], p[MAXK];
    float zmax = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) zmax = fmaxf(zmax, y[2 * K + k]);
    float zs = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) {
            p[k] = expf(y[2 * K + k] - zmax);
            zs += p[k];
        }
    float ps = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) {
            p[k] = p[k] / zs;
            ps += p[k];
        }
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) {
            p[k] = p[k] / ps;
            if (KIND == 0) {
                mu[k] = pa + y[k] * pb;
                sd[k] = expf(y[K + k]) * pb;
            } else {
                const float rng = pb - pa;
                mu[k] = pa + sigmoidf_(y[k]) * rng;
                sd[k] = KIND == 2 ? expf(y[K + k]) : rng / 1000.0f + sigmoidf_(y[K + k]) * rng * 10.0f;
            }
        }
    float v;
    if (has_value) {
        v = v_in;
    } else {
        Philox rng(seed, ctr, 0x1C);
        v = NAN;
        for (int attempt = 0; attempt < 64; ++attempt) {
            uint32_t r[4];
            rng.next(r);
            const float u0 = u01(r[0]), u1 = u01(r[1]), u2 = u01(r[2]);
            // component index ~ Categorical(p)   (Mixture.sample, distributions/mixture.py:47-63)
            float cum = 0.0f, mk = mu[0], sk = sd[0];
            bool found = false;
#pragma unroll
            for (int k = 0; k < MAXK; ++k)
                if (k < K) {
                    cum += p[k];
                    if (!found) {
                        mk = mu[k];
                        sk = sd[k];
                        if (u0 < cum) found = true;
                    }
                }
            if (KIND == 0) {
                v = mk + sk * sqrtf(-2.0f * logf(u1)) * cosf(kTwoPi * u2);   // Box-Muller
                break;
            } else {
                // inverse-CDF draw inside [low, high) with rejection (distributions/truncated_normal.py:94-112)
                const float ca = std_cdf((pa - mk) / sk), cb = std_cdf((pb - mk) / sk);
                const float uu = ca + u1 * (cb - ca);
                v = mk + sk * kSqrt2 * erfinvf(2.0f * uu - 1.0f);
                if (isfinite(v) && v >= pa && v < pb) break;
                v = NAN;
            }
        }
    }
    // log q(v)   (Mixture.log_prob, distributions/mixture.py:42-44)
    float a[MAXK], amax = -INFINITY;
    const bool inside = (KIND == 0) || (v >= pa && v <= pb);
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) {
            const float lpk = logf(fminf(fmaxf(p[k], kFp32Eps), 1.0f - kFp32Eps));
            const float t = (v - mu[k]) / sd[k];
            float comp;
            if (KIND == 0) {
                comp = -0.5f * t * t - logf(sd[k]) - kHalfLog2Pi;
            } else {
                const float Z = std_cdf((pb - mu[k]) / sd[k]) - std_cdf((pa - mu[k]) / sd[k]);
                comp = (inside ? 0.0f : -INFINITY) + (-0.5f * t * t - kHalfLog2Pi) - logf(sd[k] * Z);
            }
            a[k] = lpk + comp;
            amax = fmaxf(amax, a[k]);
        }
    float lp = amax;
    if (amax > -INFINITY) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) s += expf(a[k] - amax);
        lp = amax + logf(s);
    }
    v_out = v;
    lp_out = lp;
}

}  // namespace pp
