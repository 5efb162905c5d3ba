// LSTM input rows x = [E | s_prev | d_prev | a_prev | d_cur | a_cur] (inference_network_lstm.py:146-181): element
// lookups shared by the stand-alone gather kernel (kernels.hip) and the observe-embedding forward kernel, which
// assembles the rows of its traces itself in the training step (obs_embed.hip).
#pragma once
#include "common.hpp"

namespace pp {

struct GatherDims {
    int e_obs, smp, dtype, addr, I;
};

__device__ __forceinline__ float sample_embed_elem(const float* __restrict__ params, const int64_t* __restrict__ at,
                                                   int a, int j, float v) {
    const int smp_in = (int)at[a * PP_ADDR_TABLE_COLS + PP_AT_SMP_IN];
    const float* w = params + at[a * PP_ADDR_TABLE_COLS + PP_AT_SMP_W];
    const float* b = params + at[a * PP_ADDR_TABLE_COLS + PP_AT_SMP_B];
    float s;
    if (smp_in == 1) {
        s = w[j] * v + b[j];
    } else {
        int c = (int)v;
        c = c < 0 ? 0 : (c >= smp_in ? smp_in - 1 : c);
        s = w[j * smp_in + c] + b[j];
    }
    return relu_keep_nan(s);
}

// column c >= e_obs of a row whose previous statement has address `ap` (< 0: none) and value v, current address `a`
__device__ __forceinline__ float gather_embedding_elem(const GatherDims& d, const float* __restrict__ params,
                                                       const int64_t* __restrict__ at, int c, int ap, float v, int a) {
    const int c1 = d.e_obs, c2 = c1 + d.smp, c3 = c2 + d.dtype, c4 = c3 + d.addr, c5 = c4 + d.dtype;
    if (c < c4) {   // previous-variable part
        if (ap < 0) return 0.0f;
        if (c < c2) return sample_embed_elem(params, at, ap, c - c1, v);
        if (c < c3) return params[at[ap * PP_ADDR_TABLE_COLS + PP_AT_DTYPE_EMB] + (c - c2)];
        return params[at[ap * PP_ADDR_TABLE_COLS + PP_AT_ADDR_EMB] + (c - c3)];
    }
    if (c < c5) return params[at[a * PP_ADDR_TABLE_COLS + PP_AT_DTYPE_EMB] + (c - c4)];
    return params[at[a * PP_ADDR_TABLE_COLS + PP_AT_ADDR_EMB] + (c - c5)];
}

// What the training step's observe-embedding kernel needs to write the LSTM input rows of its traces (X == nullptr:
// disabled) and to clear the accumulators later kernels add into.
struct RowBuild {
    GatherDims d;
    const float* params;
    const int64_t* at;
    const int32_t* row_off;   // dev [t_max + 1]
    int t_max;
    const float* value;
    const int32_t* addr;
    const int32_t* prev_row;
    float* X;
    int64_t ldx;
    float* zero_like;         // same shape as X (dX) or nullptr
    float* zero_small;        // loss slots
    int n_small;
};

}  // namespace pp
