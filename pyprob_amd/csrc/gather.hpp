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
    int xcols;                // columns of a row that are written: I (full rows) or e_obs + smp (compact rows, see AddrBias)
};

// ---- address terms of the LSTM input as a per-row bias ------------------------------------------------------------
// The columns [c2, I) of an LSTM input row are table lookups: the distribution-type and address embeddings of the
// previous and of the current statement (inference_network_lstm.py:163-176) - the same vector for every row with that
// address. Their path through W_ih is therefore a per-address vector of 4H numbers, computed ONCE per step
//     cur[a]  = b_ih + b_hh + W_ih[:, c4:I] [d_a ; a_a]        prev[a] = W_ih[:, c2:c4] [d_a ; a_a]
// and added to the rows of the product [E | s_prev] W_ih[:, :c2]^T in its epilogue: G[r] += cur[addr r] + prev[addr of
// r's previous statement]. K of the input product shrinks from I = 212 to c2 = 68, and so do N of dX = dG W_ih and of
// dW_ih = dG^T X; the gradients of the table columns follow from the column sums of dG per address group:
//     dW_ih[:, c4:I] = sum_a gsum_cur[a] (x) [d_a ; a_a]        d[d_a ; a_a] = W_ih[:, c4:I]^T gsum_cur[a] + W_ih[:, c2:c4]^T gsum_prev[a]
// (same sums as the reference's autograd, re-associated). The blocks [first_block, ...) of the step's first launch
// compute the bias vectors (and clear the group sums that later kernels add into).
struct AddrBias {
    float* AB;            // [n_addr][2][N]: [a][0] = cur (+ both LSTM biases), [a][1] = prev; nullptr = disabled
    float* gsum;          // [n_addr][2][N]: cleared here
    const float* W;       // W_ih [N][ldw]
    const float* b_ih;
    const float* b_hh;
    const float* params;  // flat parameter buffer (embedding vectors through the address table)
    const int64_t* at;
    int64_t ldw;
    int N, c2, c3, c4, c5, I, n_addr, first_block;
    uint32_t present[32];   // bit a: address a occurs in the batch (current or previous statement); n_addr <= 1024
    int all_present;
};

__device__ __forceinline__ bool addr_present(const uint32_t (&mask)[32], int all, int a) {
    return all || ((mask[(a >> 5) & 31] >> (a & 31)) & 1u);
}

// block bb of the bias job: address bb / nchunk, rows (bb % nchunk) * nt .. of W_ih; one thread per row
__device__ __forceinline__ void addr_bias_block(const AddrBias& ab, int bb, float* lds /* >= 160 floats */) {
    const int nt = blockDim.x, tid = threadIdx.x;
    const int nchunk = (ab.N + nt - 1) / nt;
    const int a = bb / nchunk, n = (bb % nchunk) * nt + tid;
    if (a >= ab.n_addr || !addr_present(ab.present, ab.all_present, a)) return;   // workgroup-uniform
    const int nd = ab.c3 - ab.c2, na = ab.c4 - ab.c3, ne = nd + na;   // [d_a ; a_a]
    const float* dt = ab.params + ab.at[a * PP_ADDR_TABLE_COLS + PP_AT_DTYPE_EMB];
    const float* ad = ab.params + ab.at[a * PP_ADDR_TABLE_COLS + PP_AT_ADDR_EMB];
    for (int k = tid; k < ne; k += nt) lds[k] = k < nd ? dt[k] : ad[k - nd];
    __syncthreads();
    if (n >= ab.N) return;
    const float* w = ab.W + (int64_t)n * ab.ldw;
    float sp0 = 0.0f, sp1 = 0.0f, sc0 = 0.0f, sc1 = 0.0f;
    int k = 0;
    if (((ab.ldw | ab.c2 | ab.c4) & 3) == 0) {
        for (; k + 8 <= ne; k += 8) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(w + ab.c2 + k), p1 = *reinterpret_cast<const f32x4*>(w + ab.c2 + k + 4);
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(w + ab.c4 + k), q1 = *reinterpret_cast<const f32x4*>(w + ab.c4 + k + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sp0 += p0[e] * lds[k + e];
                sp1 += p1[e] * lds[k + 4 + e];
                sc0 += q0[e] * lds[k + e];
                sc1 += q1[e] * lds[k + 4 + e];
            }
        }
    }
    for (; k < ne; ++k) {
        sp0 += w[ab.c2 + k] * lds[k];
        sc0 += w[ab.c4 + k] * lds[k];
    }
    float* out = ab.AB + (int64_t)a * 2 * ab.N;
    out[n] = (sc0 + sc1) + (ab.b_ih[n] + ab.b_hh[n]);
    out[ab.N + n] = sp0 + sp1;
    float* gs = ab.gsum + (int64_t)a * 2 * ab.N;
    gs[n] = 0.0f;
    gs[ab.N + n] = 0.0f;
}

// kernels.hip / obs_embed.hip (host launchers shared by engine.hip and is_kernels.hip)
int lstm_input_gather(const pp_net* net, const float* params, const float* E, int64_t e_stride, const int32_t* trace,
                      const float* value, const int32_t* addr, const int32_t* prev_row, int32_t fixed_addr,
                      int32_t fixed_prev_addr, int n_rows, float* X, int64_t ldx, hipStream_t st, float* zero_like = nullptr,
                      float* zero_small = nullptr, int n_small = 0, int xcols = 0, const AddrBias* bias = nullptr);
int obs_embed_fwd_fused(const pp_net* net, const float* P, const float* obs, int n_traces, float* const* obs_h,
                        float* cat, float* f1, float* E, hipStream_t st, const RowBuild* rows = nullptr,
                        const AddrBias* bias = nullptr);

}  // namespace pp
