// Shared device/host helpers for libpyprob_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pyprob_amd.h"

#define PP_LOSS_SLOTS_FLOATS (64 * 32 + 32)   /* 64 loss slots at a 128-byte stride + the non-finite flag's line */
#define PP_TAIL_TEAMS_MAX 16   /* lstm_tail.hip */

namespace pp {

constexpr int kWave = 64;  // CDNA wavefront

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void set_error(const char* fmt, ...);

#define PP_CHECK_ARG(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            pp::set_error(__VA_ARGS__);  \
            return PP_EINVAL;            \
        }                                \
    } while (0)

#define PP_LAUNCH_CHECK(name)                                                   \
    do {                                                                        \
        hipError_t e__ = hipGetLastError();                                     \
        if (e__ != hipSuccess) {                                                \
            pp::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return (int)e__;                                                    \
        }                                                                       \
    } while (0)

#define PP_TRY(expr)            \
    do {                        \
        int rc__ = (expr);      \
        if (rc__ != 0) return rc__; \
    } while (0)

// Blocks of an output C[M, N] = A B^T whose tiles may skip part of the summation: for rows [m0, m1) x columns [n0, n1)
// the K indices [k0, k1) multiply zeros (or the block's result is not used at all: k = [0, K)). Element units; the
// kernels skip whole 64 x 64 tiles / whole K slabs inside a block only. Up to two blocks per product.
struct GemmBlock {
    int m0, m1, n0, n1, k0, k1;
};
struct GemmHole {
    GemmBlock b[2];
};

// PP_DETERMINISTIC=1: every floating-point reduction of pp_ic_loss runs in a FIXED order - no split-K float atomics
// (one workgroup per output tile walks all of K), column sums / sample-embedding gradients / the loss by single-writer
// kernels with a fixed row order - so that the same inputs give bit-identical losses and gradients on every run (the
// reference on CPU is deterministic; the default mode is not, in the last bits: float atomics commute only in exact
// arithmetic). Slower: see DESIGN.md.
static inline bool deterministic_mode() {
    static const int on = getenv("PP_DETERMINISTIC") ? atoi(getenv("PP_DETERMINISTIC")) : 0;
    return on != 0;
}

// Epilogue extensions of the tile kernels that the C ABI does not expose (engine.hip builds them):
//   * rb: per-row address bias of the LSTM input product (gather.hpp, AddrBias): row m gets
//     rb[(2 addr[m]) N + n] + (prev[m] >= 0 ? rb[(2 addr[prev[m]] + 1) N + n] : 0); async 64x64 tile only
//     (rb_addr == nullptr: one address for all rows, none has a previous statement - rb points at its vector);
//   * cell_H > 0: the product's 64-column tiles are gate-interleaved (tile bx = hidden units [16 bx, 16 bx + 16) of all
//     four gates) and rows < cell_rows (first time steps: c_prev = 0) get the LSTM cell applied in the epilogue:
//     gate activations -> C, cell state -> cell_c, hidden state -> cell_h (what lstm_cell_fwd would have done);
//   * bw_G: the product is dH = dZ1 W1 of a batch of single-statement traces; its epilogue (direct 32x32 tile) runs the
//     LSTM cell backward on the tile instead of storing dH: gates in bw_G -> dG in place, column sums of dG added to the
//     problem's `colsum` pointer (4 bw_H wide: the per-address group sums of gather.hpp).
struct GemmExt {
    const float* rb; const int32_t* rb_addr; const int32_t* rb_prev;
    float* cell_c; float* cell_h; int cell_H, cell_rows;
    // cell_cprev != nullptr (rb == nullptr): the RECURRENT product of a later time step, G_t = pre-activations already in C
    // (input part + biases) + h_{t-1} W_hh^T; every row goes through the cell with its previous cell state cell_cprev[m]
    const float* cell_cprev;
    float* bw_G; const float* bw_C; int bw_H;
    // lean: single-statement batch whose backward runs in the dH epilogue with the zero blocks on - nobody reads the forget
    // gate's columns of G / dG nor the stored cell state (c = i g when c_prev = 0): they are not written (cell_c and bw_C
    // may be null)
    int lean;
    // split_stride > 0 (single product, async tile): exactly force_splits K splits, split z STORES its partial tile at
    // C + z * split_stride instead of adding it with float atomics - the consumer kernel adds the splits (a 64 x 64 tile
    // of atomics costs its workgroup 6-8 us, tools/wg_trace.py)
    int64_t split_stride; int force_splits;
};
struct AuxJobs;

// engine.hip: in-stream kernel timing (pp_prof_arm / pp_prof_collect)
void prof_begin(int which, hipStream_t st);
void prof_end(int which, double work, hipStream_t st);

// gemm_f32.hip
int gemm_f32(const pp_gemm_args* a, hipStream_t st, const GemmHole* hole = nullptr, const GemmExt* ext = nullptr);
// aux: small reduction jobs (aux_jobs.hpp) that ride behind the tiles of the group's first launch
int gemm_f32_grouped(const pp_gemm_args* args, int count, hipStream_t st, const GemmHole* holes = nullptr,
                     const GemmExt* ext = nullptr, const AuxJobs* aux = nullptr);
int aux_jobs_launch(const AuxJobs& jobs, hipStream_t st);   // kernels.hip: the same jobs as their own launch
// lstm_input.hip: the LSTM input product of a minibatch (K = e_obs + smp_dim) with its row biases and the first time step's cell as
// one short launch, where the shape allows it (else the async tile kernel of gemm_f32.hip takes the same arguments)
bool lstm_input_fast_ok(const pp_gemm_args& g, const GemmExt& x);
int lstm_input_fast(const pp_gemm_args& g, const GemmExt& x, hipStream_t st);

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- wave64 reductions on the VALU (DPP) -------------------------------------------------------------------
// __shfl_xor lowers to ds_bpermute_b32 (an LDS-crossbar round trip per step); a reduction that sits on the critical
// path of a latency-bound kernel is ~10x cheaper with DPP: quad_perm for xor 1/2, row_ror 4/8 for the 16-lane row,
// then v_readlane of the four row totals. Every lane ends with the total.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x124>(v);   // row_ror:4
    v += dpp_mov<0x128>(v);   // row_ror:8
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x124>(v));
    v = fmaxf(v, dpp_mov<0x128>(v));
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Touch every 64-byte line of the kernel-argument segment with ONE vector load per workgroup wave: the host wrote the
// arguments just before the launch, so each line's first scalar load is an HBM-latency miss (~1-2 us); a kernel with a
// ~1 KB by-value argument struct otherwise pays those misses one after another as its s_load's reach new lines.
__device__ __forceinline__ void warm_kernargs(int bytes) {
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63;
    if (lane * 64 < bytes) {
        int t;
        asm volatile("global_load_dword %0, %1, off" : "=v"(t) : "v"(ka + lane * 64) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(t));
    }
}

// Cooperative global -> LDS copy by a 256-thread workgroup with 8 independent loads in flight per thread (a plain
// `for (i = tid; i < n; i += 256) lds[i] = g[i]` is compiled load -> wait -> store and pays a full memory round trip
// per element: ~1 us each at low occupancy).
__device__ __forceinline__ void stage_to_lds(float* __restrict__ lds, const float* __restrict__ g, int n, int tid) {
    for (int base = tid; base < n; base += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + 256 * u;
            v[u] = i < n ? g[i] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + 256 * u;
            if (i < n) lds[i] = v[u];
        }
    }
}

struct LossFinalize {   // loss = sum of the 64 accumulator slots / B, status = non-finite flag (see loss_finalize_kernel)
    const float* acc; const int32_t* flag; float inv_b; float* loss_out; int32_t* status_out;
};
__device__ __forceinline__ void loss_finalize_inline(const LossFinalize& fin) {
    float tot = 0.0f;
    for (int k = 0; k < 64; ++k) tot += fin.acc[32 * k];
    const float l = tot * fin.inv_b;
    fin.loss_out[0] = l;
    if (fin.status_out) fin.status_out[0] = (fin.flag[0] != 0 || !isfinite(l)) ? 1 : 0;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// torch.relu keeps NaN (fmaxf would turn it into 0): a non-finite observation or activation must reach the loss so that
// the minibatch is flagged and skipped like in the reference (inference_network_lstm.py:203-217).
__device__ __forceinline__ float relu_keep_nan(float v) { return v > 0.0f ? v : (v == v ? 0.0f : v); }

}  // namespace pp
