// The LSTM input product of a ragged minibatch as ONE short launch:
//     G[r, :] = [E | s_prev][r, :K] W_ih[:, :K]^T + bias(cur address of r) + bias(previous address of r)
// (InferenceNetworkLSTM._loss, pyprob/nn/inference_network_lstm.py:147-188: the rows of lstm_input times weight_ih_l0 of nn.LSTM;
// the address / distribution-type columns enter as per-address bias vectors, gather.hpp), with the LSTM cell of a trace's FIRST
// time step (zero previous state, :186) in the epilogue.
//
// Why not the async tile kernel (gemm_f32.hip) that ran this product before: K is e_obs + smp_dim = 68 - three slabs of a ring
// built for K in the thousands. A 64 x 64 tile spent ~12 us on the latencies of its prologue (row-bias indices, ring fill) and
// epilogue for 0.6 MFLOP of work, 1 300 tiles in two rounds: 30 us for 0.78 GFLOP (profiles/r04x_ragged_step_sequence.csv).
// Here nothing is staged: a wave owns 32 rows x 16 hidden units x 4 gates as eight accumulators of v_mfma_f32_16x16x4_f32 and
// reads its operands straight from X and W_ih as 16-byte pieces - lane (c = l % 16, q = l / 16) loads k = 16 s + 4 q + j, j = 0..3
// of row / column c, and MFMA j of slab s takes one k of every group q (A[l % 16][k = l / 16], B[k = l / 16][l % 16]); a slab is in
// flight behind the MFMAs of the one before. The lane that holds gate i of (row, units) also holds f, g and o: the cell runs on
// the accumulators. 82 x 8 workgroups of four waves for 2 620 rows at H = 512: all resident at once.
#include "common.hpp"

#include <stdlib.h>

namespace pp {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LstmInputArgs {
    const float* X; int64_t ldx;      // [R][ldx]: columns [0, K) are multiplied
    const float* W; int64_t ldw;      // W_ih [4H][ldw]
    int R, H, K;
    const float* rb;                  // bias table [n_addr][2][4H] (current-address part, previous-address part), or - rb_addr
    const int32_t* rb_addr;           // null - ONE vector [4H] for every row
    const int32_t* rb_prev;           // previous row of a row (-1: none), or null
    float* G;                         // [R][4H]
    int cell_rows;                    // rows [0, cell_rows) are first time steps: cell in the epilogue
    float* C; float* Hs; int lean;    // [R][H]; lean: the forget-gate columns and C are not written (GemmExt::lean)
};

constexpr int LI_ROWS = 32;           // rows per workgroup
constexpr int LI_SMAX = 8;            // 16-k slabs held in registers (K <= 128 + 12)

template <int NS>
__global__ __launch_bounds__(256) void lstm_input_kernel(const LstmInputArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int H = a.H, N = 4 * H;
    const int m0 = blockIdx.y * LI_ROWS;
    const int ub = (blockIdx.x * 4 + wave) * 16;            // this wave's 16 hidden units
    // The product is computed TRANSPOSED: A = W_ih rows (hidden units), B = X rows (minibatch rows), so that in the D layout of
    // 16x16x4 (lane (c, q) holds rows 4 q + r, r = 0..3, of column c) a lane ends with FOUR CONSECUTIVE UNITS of ONE minibatch row:
    // the row's bias pieces are 16-byte loads, the results 16-byte stores, and a lane needs the address ids of two rows, not eight
    // (with units along the lanes the epilogue was 64 dword gathers + 40 dword stores per lane: 24 us for the launch).
    const float* xr[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) xr[hh] = a.X + (int64_t)min(m0 + 16 * hh + c, a.R - 1) * a.ldx + 4 * q;
    const float* wr[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) wr[g] = a.W + (int64_t)(g * H + ub + c) * a.ldw + 4 * q;
    // ---- K loop: slab s + 1 in flight behind the MFMAs of slab s (two register sets; the waves of a SIMD hide the rest) ----
    f32x4 av[2][2], bv[2][4];
    auto load_slab = [&](int s, int buf) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) av[buf][hh] = *reinterpret_cast<const f32x4*>(xr[hh] + 16 * s);
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[buf][g] = *reinterpret_cast<const f32x4*>(wr[g] + 16 * s);
    };
    load_slab(0, 0);
    if (NS > 1) load_slab(1, 1);
    // the address ids of this lane's two rows (behind the first operand loads: loads return in order); branch-free
    int gm[2], ia[2], ip[2], pa[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        gm[hh] = m0 + 16 * hh + c;
        const int gc = min(gm[hh], a.R - 1);
        ia[hh] = a.rb_addr ? a.rb_addr[gc] : 0;
        ip[hh] = (a.rb_addr && a.rb_prev) ? a.rb_prev[gc] : -1;      // (uniform conditions)
        pa[hh] = 0;
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[hh][g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int buf = s & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[hh][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[buf][g][j], av[buf][hh][j], acc[hh][g], 0, 0, 0);
        if (s + 2 < NS) load_slab(s + 2, buf);      // refill the set those MFMAs just read
        if (s == 0 && a.rb_addr) {                  // the previous rows' address ids: a dependent trip, hidden behind the K loop
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) pa[hh] = a.rb_addr[max(ip[hh], 0)];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // remainder K - 16 NS = 4 m (m <= 3): MFMA i takes k = 16 NS + 4 i + q
    const int rem4 = (a.K - 16 * NS) >> 2;
    for (int i = 0; i < rem4; ++i) {      // (workgroup-uniform trip count)
        const int k = 16 * NS + 4 * i - 3 * q;      // (the row pointers carry + 4 q)
        float ar[2], br[4];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) ar[hh] = xr[hh][k];
#pragma unroll
        for (int g = 0; g < 4; ++g) br[g] = wr[g][k];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[hh][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(br[g], ar[hh], acc[hh][g], 0, 0, 0);
    }
    // ---- row biases, cell of the first time step, stores: units u4 .. u4 + 3 of rows gm[0], gm[1] ----
    const int u4 = ub + 4 * q;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        f32x4 v[4];
        if (a.rb_addr) {
            const bool prev = ip[hh] >= 0;
            const float* bc = a.rb + (int64_t)(2 * ia[hh]) * N + u4;
            const float* bp = a.rb + (int64_t)(2 * (prev ? pa[hh] : 0) + 1) * N + u4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bc + g * H);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bp + g * H);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[g][e] = acc[hh][g][e] + (prev ? b0[e] + b1[e] : b0[e]);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.rb + g * H + u4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[g][e] = acc[hh][g][e] + b0[e];
            }
        }
        if (gm[hh] < a.R) {
            float* g = a.G + (int64_t)gm[hh] * N + u4;
            if (gm[hh] < a.cell_rows) {   // torch.nn.LSTM gates i, f, g, o with c_prev = 0: the forget gate multiplies zero (0 recorded)
                f32x4 gi, gg, go, cn, hn;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gi[e] = sigmoidf_(v[0][e]);
                    gg[e] = tanhf(v[2][e]);
                    go[e] = sigmoidf_(v[3][e]);
                    cn[e] = gi[e] * gg[e];
                    hn[e] = go[e] * tanhf(cn[e]);
                }
                *reinterpret_cast<f32x4*>(g) = gi;
                *reinterpret_cast<f32x4*>(g + 2 * H) = gg;
                *reinterpret_cast<f32x4*>(g + 3 * H) = go;
                if (!a.lean) {
                    *reinterpret_cast<f32x4*>(g + H) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    *reinterpret_cast<f32x4*>(a.C + (int64_t)gm[hh] * H + u4) = cn;
                }
                *reinterpret_cast<f32x4*>(a.Hs + (int64_t)gm[hh] * H + u4) = hn;
            } else {                      // a later time step: pre-activations, the recurrent product follows
#pragma unroll
                for (int gt = 0; gt < 4; ++gt) *reinterpret_cast<f32x4*>(g + gt * H) = v[gt];
            }
        }
    }
}

}  // namespace

// Does the short launch take this product? (gate-interleaved epilogue shapes only: the caller fuses the first step's cell)
bool lstm_input_fast_ok(const pp_gemm_args& g, const GemmExt& x) {
    static const int env = getenv("PP_LSTM_INPUT_FAST") ? atoi(getenv("PP_LSTM_INPUT_FAST")) : 1;
    if (!env || deterministic_mode()) return false;
    const int H = x.cell_H;
    if (H <= 0 || H % 64 != 0 || g.N != 4 * H || !x.rb || x.cell_cprev || !x.cell_h || (!x.lean && !x.cell_c)) return false;
    if (g.a_kmajor || g.b_kmajor || g.a_idx || g.b_idx || g.c_idx || g.bias || g.bias2 || g.relu || g.accumulate || g.mask || g.colsum)
        return false;
    if (g.K < 16 || g.K % 4 != 0 || g.K > 16 * LI_SMAX + 12 || g.lda % 4 != 0 || g.ldb % 4 != 0 || g.ldc != 4 * H) return false;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return al16(g.A) && al16(g.B) && al16(g.C) && al16(x.rb) && al16(x.cell_h) && (x.lean || al16(x.cell_c));
}

int lstm_input_fast(const pp_gemm_args& g, const GemmExt& x, hipStream_t st) {
    LstmInputArgs a{};
    a.X = g.A; a.ldx = g.lda; a.W = g.B; a.ldw = g.ldb;
    a.R = (int)g.M; a.H = x.cell_H; a.K = (int)g.K;
    a.rb = x.rb; a.rb_addr = x.rb_addr; a.rb_prev = x.rb_prev;
    a.G = g.C; a.cell_rows = x.cell_rows; a.C = x.cell_c; a.Hs = x.cell_h; a.lean = x.lean;
    const dim3 grid(a.H / 64, cdiv(a.R, LI_ROWS)), block(256);
    switch (a.K / 16) {
#define PP_LI_CASE(NS) case NS: hipLaunchKernelGGL(lstm_input_kernel<NS>, grid, block, 0, st, a); break
        PP_LI_CASE(1); PP_LI_CASE(2); PP_LI_CASE(3); PP_LI_CASE(4); PP_LI_CASE(5); PP_LI_CASE(6); PP_LI_CASE(7); PP_LI_CASE(8);
#undef PP_LI_CASE
        default: set_error("lstm_input_fast: K = %d", a.K); return PP_EINVAL;
    }
    PP_LAUNCH_CHECK("lstm_input_fast");
    return 0;
}

}  // namespace pp
