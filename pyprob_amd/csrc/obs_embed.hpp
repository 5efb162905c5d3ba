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

// fills the layer descriptions (LDS image offsets included); false if the image does not fit 10 240 floats
bool obs_fused_args(const pp_net* net, float* const* obs_h, ObsFusedArgs& a);
bool obs_fused_supported(const pp_net* net);

}  // namespace pp
