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
    int* step_epoch;        // counts the calls of this workspace (the job rides in a call's FIRST launch): the tag of the
                            // cross-workgroup hand-offs of the launches that follow (handoff.hpp); nullptr: not counted
};

__device__ __forceinline__ bool addr_present(const uint32_t (&mask)[32], int all, int a) {
    return all || ((mask[(a >> 5) & 31] >> (a & 31)) & 1u);
}

constexpr int ADDR_BIAS_ROWS = 32;                       // rows of W_ih per workgroup of the bias job
constexpr int ADDR_BIAS_LDS = 32 * (2 * 128 + 1) + 128;  // floats of LDS it needs at most (ne <= 128)
static inline int addr_bias_blocks(const AddrBias& ab) { return ab.n_addr * ((ab.N + ADDR_BIAS_ROWS - 1) / ADDR_BIAS_ROWS); }

// Workgroup bb of the bias job (256 threads): address bb / nchunk, rows (bb % nchunk) * 32 .. + 31 of W_ih. The rows'
// table columns [c2, I) (2 ne contiguous floats per row) come in with coalesced 16-byte loads, all issued before the first
// LDS store (one memory round trip); thread (row, part) then takes an eighth of the two dot products from LDS and the
// eight partial sums meet through DPP. Needs ldw, c2 and ne to be multiples of 4, 2 and 2 (engine.hip checks).
__device__ __forceinline__ void addr_bias_block(const AddrBias& ab, int bb, float* lds) {
    const int tid = threadIdx.x;
    if (bb == 0 && tid == 0 && ab.step_epoch) ab.step_epoch[0] += 1;
    const int nchunk = (ab.N + ADDR_BIAS_ROWS - 1) / ADDR_BIAS_ROWS;
    const int a = bb / nchunk, n0 = (bb % nchunk) * ADDR_BIAS_ROWS;
    if (a >= ab.n_addr || !addr_present(ab.present, ab.all_present, a)) return;   // workgroup-uniform
    const int nd = ab.c3 - ab.c2, ne = ab.c4 - ab.c2;   // [d_a ; a_a]
    const int ncol = 2 * ne, q4 = ncol >> 2, ss = ncol + 1;
    float* S = lds;                          // [32][2 ne + 1]
    float* ev = lds + ADDR_BIAS_ROWS * ss;   // [ne]
    const float* dt = ab.params + ab.at[a * PP_ADDR_TABLE_COLS + PP_AT_DTYPE_EMB];
    const float* ad = ab.params + ab.at[a * PP_ADDR_TABLE_COLS + PP_AT_ADDR_EMB];
    constexpr int U = 8;   // 16-byte pieces per thread: 32 rows x 2 ne / 4 <= 2048
    f32x4 v[U];
    const int total = ADDR_BIAS_ROWS * q4;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = min(tid + 256 * u, total - 1);
        const int r = i / q4, c = i - r * q4;
        const int n = min(n0 + r, ab.N - 1);
        v[u] = *reinterpret_cast<const f32x4*>(ab.W + (int64_t)n * ab.ldw + ab.c2 + 4 * c);
    }
    float e0 = 0.0f;
    if (tid < ne) e0 = tid < nd ? dt[tid] : ad[tid - nd];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = tid + 256 * u;
        if (i < total) {
            const int r = i / q4, c = i - r * q4;
#pragma unroll
            for (int e = 0; e < 4; ++e) S[r * ss + 4 * c + e] = v[u][e];
        }
    }
    if (tid < ne) ev[tid] = e0;
    __syncthreads();
    const int row = tid >> 3, part = tid & 7;
    const int per = (ne + 7) >> 3;
    const int k0 = part * per, k1 = min(ne, k0 + per);
    const float* sr = S + row * ss;
    float sp = 0.0f, sc = 0.0f;
    for (int k = k0; k < k1; ++k) {
        sp += sr[k] * ev[k];
        sc += sr[ne + k] * ev[k];
    }
    // sum over the eight lanes of a row: xor 1, xor 2 inside the quad (DPP), then the other quad of the row
    sp += dpp_mov<0xB1>(sp); sc += dpp_mov<0xB1>(sc);
    sp += dpp_mov<0x4E>(sp); sc += dpp_mov<0x4E>(sc);
    sp += __shfl_xor(sp, 4, 64); sc += __shfl_xor(sc, 4, 64);
    const int n = n0 + row;
    if (part == 0 && n < ab.N) {
        float* out = ab.AB + (int64_t)a * 2 * ab.N;
        out[n] = sc + (ab.b_ih[n] + ab.b_hh[n]);
        out[ab.N + n] = sp;
        float* gs = ab.gsum + (int64_t)a * 2 * ab.N;
        gs[n] = 0.0f;
        gs[ab.N + n] = 0.0f;
    }
}

// kernels.hip / obs_embed.hip (host launchers shared by engine.hip and is_kernels.hip)
int lstm_input_gather(const pp_net* net, const float* params, const float* E, int64_t e_stride, const int32_t* trace,
                      const float* value, const int32_t* addr, const int32_t* prev_row, int32_t fixed_addr,
                      int32_t fixed_prev_addr, int n_rows, float* X, int64_t ldx, hipStream_t st, float* zero_like = nullptr,
                      float* zero_small = nullptr, int n_small = 0, int xcols = 0, const AddrBias* bias = nullptr);
struct PanelTranspose;   // panel.hpp
int obs_embed_fwd_fused(const pp_net* net, const float* P, const float* obs, int n_traces, float* const* obs_h,
                        float* cat, float* f1, float* E, hipStream_t st, const RowBuild* rows = nullptr,
                        const AddrBias* bias = nullptr, const PanelTranspose* transpose = nullptr);

}  // namespace pp
