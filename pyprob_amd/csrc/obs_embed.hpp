// Shared pieces of the fused observe-embedding kernels (obs_embed.hip) - also used by the row-panel kernel (panel.hip), whose
// tail walks its rows backward through the embedding stack.
#pragma once
#include "common.hpp"

namespace pp {

constexpr int OBS_EMAX = 64;    // lanes
constexpr int OBS_HIDMAX = 32;  // per-observable hidden width kept in registers
constexpr int OBS_INMAX = 8;

struct ObsLayer {
    int64_t w_off, b_off;  // offsets into the flat parameter / gradient buffers
    int rows, cols;        // weight [rows, cols]
    int lds_w, lds_b;      // offsets (floats) into the LDS image; weight rows have stride cols + 1
};

struct ObsFusedArgs {
    int n_obs, e_obs, width;
    int in[PP_MAX_OBS], hid[PP_MAX_OBS], out[PP_MAX_OBS];
    int hoff[PP_MAX_OBS];  // first lane of observable o's hidden units
    ObsLayer l0[PP_MAX_OBS], l1[PP_MAX_OBS], f0, f1;
    int lds_total;
    int64_t ohid_ld[PP_MAX_OBS], e_ld;
    float* obs_h[PP_MAX_OBS];   // [B, ohid_ld] hidden activations of observable o (saved for backward)
};

// broadcast lane `idx` (wave-uniform index) of x: v_readlane_b32, no LDS round trip (a runtime-indexed __shfl lowers to
// ds_bpermute_b32 and serialises on lgkmcnt)
__device__ __forceinline__ float bcast(float x, int idx) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), idx));
}

// ---- weights -> LDS (rows of stride cols+1) ---------------------------------------------------------------
template <int U, int NT = 256>
struct ObsStage {
    float v[U];
    // branch-free: out-of-range slots read element n-1 and are later written to a dummy LDS word
    __device__ __forceinline__ void load(const float* __restrict__ g, int n, int tid) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = g[min(tid + NT * u, n - 1)];
    }
    // row = floor(i / cols) through an exact float reciprocal (i < 2^13, cols <= 64): no integer division
    __device__ __forceinline__ void store(float* lds, int lds_w, int rows, int cols, int dummy, int tid) const {
        const int n = rows * cols;
        const float inv = 1.0f / (float)cols;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = tid + NT * u;
            const int r = (int)(((float)i + 0.5f) * inv);
            const int off = lds_w + r * (cols + 1) + (i - r * cols);
            lds[i < n ? off : dummy] = v[u];
        }
    }
};

// dx_lane = sum_j dz_j * W[j][lane] (lane < cols), dz_j held by lane z_lane0 + j
__device__ __forceinline__ float obs_dense_t(const float* lds, const ObsLayer& L, int col, bool act, float dz, int z_lane0) {
    const float* w = lds + L.lds_w + (act ? col : 0);
    const int ld = L.cols + 1;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int j = 0;
    for (; j + 8 <= L.rows; j += 8) {
        float wv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) wv[q] = w[(j + q) * ld];
        s0 += wv[0] * bcast(dz, z_lane0 + j) + wv[4] * bcast(dz, z_lane0 + j + 4);
        s1 += wv[1] * bcast(dz, z_lane0 + j + 1) + wv[5] * bcast(dz, z_lane0 + j + 5);
        s2 += wv[2] * bcast(dz, z_lane0 + j + 2) + wv[6] * bcast(dz, z_lane0 + j + 6);
        s3 += wv[3] * bcast(dz, z_lane0 + j + 3) + wv[7] * bcast(dz, z_lane0 + j + 7);
    }
    for (; j < L.rows; ++j) s0 += w[j * ld] * bcast(dz, z_lane0 + j);
    return act ? (s0 + s1) + (s2 + s3) : 0.0f;
}


// (obs_stage_all / obs_dense / obs_forward_row: shared by obs_embed.hip and the first-statement kernel of is_kernels.hip)
// ---- all layers' weights -> LDS -------------------------------------------------------------------------------
// The parameters were just rewritten by Adam on other XCDs, so every load is a long trip (~1-2 us). ALL loads of ALL
// layers are therefore issued first (one round trip), and only then the LDS stores.
template <int NOBS>
__device__ __forceinline__ void obs_stage_all(const ObsFusedArgs& a, const float* __restrict__ P, float* lds, int tid) {
    ObsStage<16> sf0, sf1;                 // 64 x 64
    ObsStage<8> s1[NOBS];                  // out x hid <= 64 x 32
    ObsStage<1> s0[NOBS];                  // hid x in  <= 32 x 8
    float bf0, bf1, b1[NOBS], b0[NOBS];
    sf0.load(P + a.f0.w_off, a.f0.rows * a.f0.cols, tid);
    sf1.load(P + a.f1.w_off, a.f1.rows * a.f1.cols, tid);
    bf0 = tid < a.f0.rows ? P[a.f0.b_off + tid] : 0.0f;
    bf1 = tid < a.f1.rows ? P[a.f1.b_off + tid] : 0.0f;
#pragma unroll
    for (int o = 0; o < NOBS; ++o) {
        if (o < a.n_obs) {
            s1[o].load(P + a.l1[o].w_off, a.l1[o].rows * a.l1[o].cols, tid);
            s0[o].load(P + a.l0[o].w_off, a.l0[o].rows * a.l0[o].cols, tid);
            b1[o] = tid < a.l1[o].rows ? P[a.l1[o].b_off + tid] : 0.0f;
            b0[o] = tid < a.l0[o].rows ? P[a.l0[o].b_off + tid] : 0.0f;
        }
    }
    const int dummy = a.lds_total;   // one spare word behind the image
    sf0.store(lds, a.f0.lds_w, a.f0.rows, a.f0.cols, dummy, tid);
    sf1.store(lds, a.f1.lds_w, a.f1.rows, a.f1.cols, dummy, tid);
    if (tid < a.f0.rows) lds[a.f0.lds_b + tid] = bf0;
    if (tid < a.f1.rows) lds[a.f1.lds_b + tid] = bf1;
#pragma unroll
    for (int o = 0; o < NOBS; ++o) {
        if (o < a.n_obs) {
            s1[o].store(lds, a.l1[o].lds_w, a.l1[o].rows, a.l1[o].cols, dummy, tid);
            s0[o].store(lds, a.l0[o].lds_w, a.l0[o].rows, a.l0[o].cols, dummy, tid);
            if (tid < a.l1[o].rows) lds[a.l1[o].lds_b + tid] = b1[o];
            if (tid < a.l0[o].rows) lds[a.l0[o].lds_b + tid] = b0[o];
        }
    }
}

// y_lane = relu(b[lane] + sum_k W[lane][k] * x_k), x_k held by lane k (+ x_lane0) of the wave
__device__ __forceinline__ float obs_dense(const float* lds, const ObsLayer& L, int row, bool act, float x, int x_lane0) {
    const float* w = lds + L.lds_w + (act ? row : 0) * (L.cols + 1);
    // four independent partial sums, eight LDS reads in flight: the loop is latency-bound otherwise
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int k = 0;
    for (; k + 8 <= L.cols; k += 8) {
        float wv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) wv[q] = w[k + q];
        s0 += wv[0] * bcast(x, x_lane0 + k) + wv[4] * bcast(x, x_lane0 + k + 4);
        s1 += wv[1] * bcast(x, x_lane0 + k + 1) + wv[5] * bcast(x, x_lane0 + k + 5);
        s2 += wv[2] * bcast(x, x_lane0 + k + 2) + wv[6] * bcast(x, x_lane0 + k + 6);
        s3 += wv[3] * bcast(x, x_lane0 + k + 3) + wv[7] * bcast(x, x_lane0 + k + 7);
    }
    for (; k < L.cols; ++k) s0 += w[k] * bcast(x, x_lane0 + k);
    const float s = (s0 + s1) + (s2 + s3);
    return act ? relu_keep_nan(s + lds[L.lds_b + row]) : 0.0f;
}


// One row (observation vector `obs`, `width` numbers) through the whole stack by ONE wave with the image in LDS: lane j < e_obs
// returns unit j of the embedding (InferenceNetwork._embed_observe, pyprob/nn/inference_network.py:132-139). The same walk and
// summation order as obs_embed_fwd_kernel: bit-identical embeddings.
template <int NOBS>
__device__ __forceinline__ float obs_forward_row(const ObsFusedArgs& a, const float* lds, const float* __restrict__ obs, int lane) {
    float h = 0.0f, c = 0.0f;
    int ci = 0, co = 0;
#pragma unroll
    for (int o = 0; o < NOBS; ++o) {
        if (o >= a.n_obs) break;
        const int jh = lane - a.hoff[o];
        const bool acth = jh >= 0 && jh < a.hid[o];
        if (acth) {
            const float* w = lds + a.l0[o].lds_w + jh * (a.in[o] + 1);
            float s = lds[a.l0[o].lds_b + jh];
            for (int i = 0; i < a.in[o]; ++i) s += w[i] * obs[ci + i];
            h = relu_keep_nan(s);
        }
        ci += a.in[o];
    }
#pragma unroll
    for (int o = 0; o < NOBS; ++o) {
        if (o >= a.n_obs) break;
        const int jc = lane - co;
        const bool actc = jc >= 0 && jc < a.out[o];
        const float v = obs_dense(lds, a.l1[o], jc, actc, h, a.hoff[o]);
        if (actc) c = v;
        co += a.out[o];
    }
    const bool acte = lane < a.e_obs;
    const float y1 = obs_dense(lds, a.f0, lane, acte, c, 0);
    return obs_dense(lds, a.f1, lane, acte, y1, 0);
}

// fills the layer descriptions (LDS image offsets included); false if the image does not fit 10 240 floats
bool obs_fused_args(const pp_net* net, float* const* obs_h, ObsFusedArgs& a);
bool obs_fused_supported(const pp_net* net);

}  // namespace pp
