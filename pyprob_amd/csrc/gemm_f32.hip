// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-exact fmaf chain).
//
//   C[ix_c(m), n] = epilogue( sum_k A(m,k) * B(n,k) )
//
// Replaces every nn.Linear / nn.LSTM matmul of the reference's hot path and their autograd gradients
// (pyprob/nn/embedding_feedforward.py:40, pyprob/nn/inference_network_lstm.py:188).
// Both operands may be "k-contiguous" (row-major [rows][K]) or "k-major" ([K][rows]) so that forward (NT),
// data-gradient (NN) and weight-gradient (TN) products all run on one kernel, and either may carry a row
// gather (the address-dispatch gather of the per-address proposal heads / ragged time steps).
//
// Tiling (wave64, 4 waves / 256 threads per workgroup):
//   * workgroup tile BM x BN, K staged through LDS in slabs of BK = 32
//   * k-contiguous operands sit in LDS as [row][BK+4]: one ds_read_b128 per lane feeds four MFMAs
//     (lane l reads row l&31, k = 8*s + 4*(l>>5) .. +3); row stride 36 floats = 9 sixteen-byte slots (odd) so the
//     16-lane groups of ds_read_b128 hit 16 distinct slots -> conflict free.
//   * k-major operands sit in LDS as [k][rows+4]: four ds_read_b32 per lane (consecutive rows -> conflict free).
//   * the summation index may be visited in any order, so MFMA step j of sub-slab s uses k = 8s + j from
//     lanes 0-31 and k = 8s + 4 + j from lanes 32-63 for BOTH operands.
//   * global->register prefetch of slab i+1 overlaps the MFMAs of slab i (two barriers per slab).
#include "common.hpp"
#include "aux_jobs.hpp"

#include <algorithm>

namespace pp {

extern long long* g_timeline;   // kernels.hip (debug phase stamps)
#define GEMM_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (p.dbg && threadIdx.x == 0 && bx == 1 && by == 1 && bz == 0) p.dbg[10 + (k)] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)

constexpr int BK = 32;
constexpr int KPAD = 4;

struct GemmParams {
    const float* A; int64_t lda; const int32_t* a_idx;
    const float* B; int64_t ldb; const int32_t* b_idx;
    float* C; int64_t ldc; const int32_t* c_idx;
    int M, N, K;
    const float* bias; const float* bias2;
    const float* mask; int64_t ldmask;
    float* colsum;   // optional: colsum[n] += sum_m result[m, n] (bias gradients fused into the producing GEMM)
    int relu, accumulate;
    int vec;         // grouped launch: this problem's operands allow 16-byte loads
    int vec_c;       // C rows allow 16-byte stores (direct tiles)
    int dbg_plain;   // measurement only: split partials are stored instead of added (WRONG results)
    int64_t split_stride;   // > 0: split bz STORES its partial tile at C + bz * split_stride (the consumer adds the splits)
    // zero block (GemmHole): for output tiles inside rows [hm0, hm1) x columns [hn0, hn1) the K slabs [hs0, hs1) hold
    // nothing but zeros (or the tile is not used at all) and are skipped. hs1 <= hs0: no zero block. Two blocks per
    // product; a tile inside both skips the longer slab interval.
    int hm0[2], hm1[2], hn0[2], hn1[2], hs0[2], hs1[2];
    long long* dbg;  // debug phase stamps (NULL normally)
};

// ---- global -> register staging of one [ROWS x BK] operand slab ------------------------------------------
// KM = false: element (row, k) at P[ix(row0+row)*ld + k0+k]; KM = true: at P[ix(k0+k)*ld + row0+row].
// Everything that does not depend on the slab (row gather, base pointers, edge predicates) is resolved once in
// init(); the K loop only adds the slab offset and issues the loads.
template <int ROWS, bool KM, int VEC>
struct Stager {
    static constexpr int PER_THREAD = ROWS * BK / 256;  // floats per thread
    static constexpr int NV = PER_THREAD / VEC;         // vector slots per thread
    float r[2][PER_THREAD];      // two register sets (software pipeline, static indices only)
    const float* base[NV];       // !KM: &P[ix(row)*ld + k_slot];  KM: &P[row_slot] (k added per slab)
    int kslot[NV];               // this slot's k inside a slab
    bool ok[NV];                 // row inside the matrix
    bool vec[NV];                // KM: all VEC rows of the slot inside the matrix
    const int32_t* idx;
    int64_t ld;
    int K, nrows, grow0[NV];

    __device__ __forceinline__ void init(const float* __restrict__ P, int64_t ld_, const int32_t* __restrict__ idx_,
                                         int row0, int nrows_, int K_, int tid) {
        idx = idx_; ld = ld_; K = K_; nrows = nrows_;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int s = tid + 256 * p;
            int row, k;
            if (!KM) {
                row = s / (BK / VEC);
                k = (s % (BK / VEC)) * VEC;
            } else {
                k = s / (ROWS / VEC);
                row = (s % (ROWS / VEC)) * VEC;
            }
            const int grow = row0 + row;
            kslot[p] = k;
            grow0[p] = grow;
            ok[p] = grow < nrows;
            vec[p] = grow + VEC - 1 < nrows;
            if (!KM) {
                const int64_t rr = (ok[p] && idx) ? (int64_t)idx[grow] : (int64_t)(ok[p] ? grow : 0);
                base[p] = P + rr * ld + k;
            } else {
                base[p] = P + (ok[p] ? grow : 0);
            }
        }
    }

    template <int SET>
    __device__ __forceinline__ void load(int k0) {
        const bool full = k0 + BK <= K;   // block-uniform
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            float* dst = &r[SET][p * VEC];
            if (!KM) {
                const float* src = base[p] + k0;
                const int gk = k0 + kslot[p];
                if (ok[p] && (full || gk + VEC - 1 < K)) {
                    if (VEC == 4) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) dst[e] = v[e];
                    } else {
                        dst[0] = src[0];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) dst[e] = (ok[p] && gk + e < K) ? src[e] : 0.0f;
                }
            } else {
                const int gk = k0 + kslot[p];
                if (ok[p] && gk < K) {
                    const int64_t kk = idx ? (int64_t)idx[gk] : (int64_t)gk;
                    const float* src = base[p] + kk * ld;
                    if (VEC == 4 && vec[p]) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) dst[e] = v[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) dst[e] = (grow0[p] + e < nrows) ? src[e] : 0.0f;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) dst[e] = 0.0f;
                }
            }
        }
    }

    // LDS image: !KM -> [ROWS][BK+KPAD], KM -> [BK][ROWS+KPAD]
    template <int SET>
    __device__ __forceinline__ void store(float* __restrict__ S, int tid) const {
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int s = tid + 256 * p;
            int off;
            if (!KM) {
                const int row = s / (BK / VEC), k = (s % (BK / VEC)) * VEC;
                off = row * (BK + KPAD) + k;
            } else {
                const int k = s / (ROWS / VEC), row = (s % (ROWS / VEC)) * VEC;
                off = k * (ROWS + KPAD) + row;
            }
            if (VEC == 4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = r[SET][p * VEC + e];
                *reinterpret_cast<f32x4*>(S + off) = v;
            } else {
                S[off] = r[SET][p];
            }
        }
    }
};

// fragment of one 32-row MFMA operand for sub-slab s (8 k values): out[j] = element (row, 8s + 4h + j)
template <int ROWS, bool KM>
__device__ __forceinline__ void read_frag(const float* __restrict__ S, int row, int s, int h, float out[4]) {
    if (!KM) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(S + row * (BK + KPAD) + s * 8 + h * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = v[j];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = S[(s * 8 + h * 4 + j) * (ROWS + KPAD) + row];
    }
}

// Epilogue shared by the LDS-staged and the async tiles. acc[i][j] is the 32x32 MFMA fragment (i, j) of this wave.
template <int BM, int BN, int WM, int WN, int TM, int TN>
__device__ __forceinline__ void tile_epilogue(const GemmParams& p, f32x16 (&acc)[TM][TN], const int m0, const int n0,
                                              const int wm, const int wn, const int l31, const int h, const int bz,
                                              const bool split) {
    // epilogue: D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool lead = bz == 0;
    // Fast path (the forward products): interior tile, plain store, at most bias + ReLU. No per-element predicates,
    // one base pointer per 32x32 fragment, 16 stores at constant row strides.
    const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N);
    if (interior && !split && !p.c_idx && !p.mask && !p.colsum && !p.accumulate) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int gn = n0 + wn * WN + j * 32 + l31;
                float bsum = 0.0f;
                if (p.bias) bsum += p.bias[gn];
                if (p.bias2) bsum += p.bias2[gn];
                float* base = p.C + (int64_t)(m0 + wm * WM + i * 32 + 4 * h) * p.ldc + gn;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] + bsum;
                    if (p.relu) v = relu_keep_nan(v);
                    base[(int64_t)((r & 3) + 8 * (r >> 2)) * p.ldc] = v;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gn = n0 + wn * WN + j * 32 + l31;
            const bool ncol = gn < p.N;
            float bsum = 0.0f;
            if (ncol && lead) {
                if (p.bias) bsum += p.bias[gn];
                if (p.bias2) bsum += p.bias2[gn];
            }
            float csum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (!ncol || gm >= p.M) continue;
                const int64_t cm = p.c_idx ? (int64_t)p.c_idx[gm] : (int64_t)gm;
                float v = acc[i][j][r] + bsum;
                float* dst = p.C + cm * p.ldc + gn;
                if (split) {
                    if (p.split_stride) dst[(int64_t)bz * p.split_stride] = v;
                    else if (p.dbg_plain) *dst = v;
                    else atomicAdd(dst, v);
                } else {
                    if (p.relu) v = relu_keep_nan(v);
                    if (p.mask) v = (p.mask[cm * p.ldmask + gn] > 0.0f) ? v : 0.0f;
                    csum += v;
                    if (p.accumulate) v += *dst;
                    *dst = v;
                }
            }
            if (p.colsum && !split) {   // wave-uniform condition
                csum += __shfl_xor(csum, 32, 64);
                if (h == 0 && ncol) atomicAdd(p.colsum + gn, csum);
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM, int VEC>
__device__ __forceinline__ void gemm_tile(const GemmParams& pin, const int bx, const int by, const int bz, const int nz,
                                          float* smem) {
    // A LOCAL copy of the problem description: through the reference (kernel-argument memory, indexed by the problem id in
    // the grouped kernels) the compiler re-loaded every field at every use - ~130 scalar loads with a wait each in the
    // 16-element epilogue loop, 4 us per workgroup (tools/wg_trace.py)
    const GemmParams p = pin;
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    GEMM_STAMP(0);
    constexpr int BUF = (BM + BN) * (BK + KPAD);   // floats per LDS buffer: [A slab | B slab]

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = by * BM, n0 = bx * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // split-K: blockIdx.z owns a contiguous range of K slabs and adds its partial tile with float atomics
    const int nslab_total = (p.K + BK - 1) / BK;
    // (zero blocks: like the direct tile, this family skips a tile only when it lies inside a block that covers ALL of K)
#pragma unroll
    for (int q = 0; q < 2; ++q) {   // workgroup-uniform
        if (p.hs1[q] > p.hs0[q] && p.hs0[q] == 0 && p.hs1[q] >= nslab_total && m0 >= p.hm0[q] && min(m0 + BM, p.M) <= p.hm1[q] &&
            n0 >= p.hn0[q] && min(n0 + BN, p.N) <= p.hn1[q] && !((A_KM && p.a_idx) || (B_KM && p.b_idx)))
            return;
    }
    const int per = (nslab_total + nz - 1) / nz;
    const int s_begin = bz * per;
    const int nslab = min(nslab_total, s_begin + per) - s_begin;
    if (nslab <= 0) return;
    const bool split = nz > 1;

    // two register sets: the loads of slab i+2 are issued before the MFMAs of slab i and consumed after the MFMAs
    // of slab i+1, so ~2 slabs of matrix work cover one HBM/L2 round trip.
    Stager<BM, A_KM, VEC> sa;
    Stager<BN, B_KM, VEC> sb;
    sa.init(p.A, p.lda, p.a_idx, m0, p.M, p.K, tid);
    sb.init(p.B, p.ldb, p.b_idx, n0, p.N, p.K, tid);
    // One K slab: all eight fragment reads (4 sub-slabs x {A, B}) are issued before the first MFMA, so the LDS latency
    // is paid once per slab instead of once per sub-slab (the MFMAs of a sub-slab depend on its reads).
    auto compute = [&](const float* As, const float* Bs) {
        float a[BK / 8][TM][4], b[BK / 8][TN][4];
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) read_frag<BM, A_KM>(As, wm * WM + i * 32 + l31, s, h, a[s][i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) read_frag<BN, B_KM>(Bs, wn * WN + j * 32 + l31, s, h, b[s][j]);
        }
#pragma unroll
        for (int s = 0; s < BK / 8; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i][q], b[s][j][q], acc[i][j], 0, 0, 0);
    };
    // Pipeline: LDS is double buffered (ONE barrier per slab), global loads run two slabs ahead in two register sets.
    //   iteration i: [store slab i+1 -> other LDS buffer] [issue loads of slab i+2] [MFMAs of slab i] [barrier]
    float* const buf0 = smem;
    float* const buf1 = smem + BUF;
    const int kb = s_begin * BK;
    sa.template load<0>(kb);
    sb.template load<0>(kb);
    if (nslab > 1) {
        sa.template load<1>(kb + BK);
        sb.template load<1>(kb + BK);
    }
    GEMM_STAMP(1);
    sa.template store<0>(buf0, tid);
    sb.template store<0>(buf0 + BM * (BK + KPAD), tid);
    __syncthreads();
    GEMM_STAMP(2);
    for (int i = 0; i < nslab; i += 2) {
        // slab i is in buf0; set 1 holds slab i+1 (in flight); set 0 is free
        if (i + 1 < nslab) {
            sa.template store<1>(buf1, tid);
            sb.template store<1>(buf1 + BM * (BK + KPAD), tid);
        }
        if (i + 2 < nslab) {
            sa.template load<0>(kb + (i + 2) * BK);
            sb.template load<0>(kb + (i + 2) * BK);
        }
        compute(buf0, buf0 + BM * (BK + KPAD));
        if (i + 1 >= nslab) break;
        __syncthreads();
        // slab i+1 is in buf1; set 0 holds slab i+2 (in flight); set 1 is free
        if (i + 2 < nslab) {
            sa.template store<0>(buf0, tid);
            sb.template store<0>(buf0 + BM * (BK + KPAD), tid);
        }
        if (i + 3 < nslab) {
            sa.template load<1>(kb + (i + 3) * BK);
            sb.template load<1>(kb + (i + 3) * BK);
        }
        compute(buf1, buf1 + BM * (BK + KPAD));
        if (i + 2 >= nslab) break;
        __syncthreads();
    }

    if (p.dbg && acc[0][0][0] == 123456.789f) p.dbg[15] = 1;   // debug: the stamp below must follow the last MFMA
    GEMM_STAMP(3);
    tile_epilogue<BM, BN, WM, WN, TM, TN>(p, acc, m0, n0, wm, wn, l31, h, bz, split);
    GEMM_STAMP(4);
}

template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM, int VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * (BK + KPAD)];   // double-buffered K slabs
    gemm_tile<BM, BN, WM, WN, A_KM, B_KM, VEC>(p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z, smem);
}

// ---- async tiles: LDS-DMA software pipeline -------------------------------------------------------------------
// The register-staged tile keeps ONE slab of loads in flight per workgroup, so a K loop runs at one memory round trip
// (~1.5 us measured for operands that live in another XCD's L2 / the Infinity Cache) per slab unless several
// workgroups share the CU. Here the slabs go global -> LDS with global_load_lds_dwordx4 (no VGPR staging) into a ring
// of AS_STAGES images, three slabs ahead of the MFMAs; one barrier per slab, vmcnt-counted completion.
//   * image layout is dictated by the DMA (lane L of a wave instruction writes 16 bytes at base + 16 L), so the
//     bank-conflict padding of the staged tile is replaced by an XOR swizzle applied on the GLOBAL side:
//       k-contiguous operand: image [64 rows][8 sixteen-byte k slots]; slot p of row r holds k slot p ^ ((r>>1)&7)
//                             -> the 16 lanes of a ds_read_b128 group (rows r..r+15, same k slot) hit 16 distinct slots
//       k-major operand:      image [32 k][64 rows]; row r of k sits at r ^ (32 * ((k>>2)&1))
//                             -> the two half-waves of a fragment read (k and k+4) use disjoint bank halves
//   * edges: rows beyond the matrix are clamped to a valid row (their results are never stored); pieces beyond K (last
//     slab only) are fetched from a valid address and zeroed in the fragment registers of that slab.
//   * the loads are inline asm: the compiler orders every LDS read after all earlier LDS-DMA of the builtin
//     (s_waitcnt vmcnt(0) at the loop head), which would serialise the ring.
constexpr int AS_STAGES = 4;
constexpr int vmcnt_imm(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }   // s_waitcnt vmcnt(n) only
constexpr int AS_IMG = 64 * BK;        // floats per operand image
constexpr int AS_STAGE = 2 * AS_IMG;   // [A image | B image]
constexpr int AS_KSLABS = 32;          // a workgroup's K range may span this many slabs when an operand gathers along k
constexpr int AS_KIDX = (AS_KSLABS + 1) * BK;   // ints per staged gather list

__device__ __forceinline__ uint32_t lds_addr(const float* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)p;
}
// 64 lanes x 16 bytes: lane L's piece lands at lds_base + 16 L (lds_base wave-uniform)
__device__ __forceinline__ void dma16(const float* g, uint32_t lds_base) {
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_base), "v"(g) : "memory", "m0");
}

template <bool KM, int NP>
struct AsyncOperand {
    const float* ptr[NP];   // this lane's pieces of slab 0 (chunks NP*wave .. of the image; 8 chunks of 1 KB per image)
    const float* safe[NP];   // in-bounds stand-in for a piece beyond K
    int kpiece[NP];          // first k of the piece inside a slab
    int64_t step;           // floats between consecutive slabs
    uint32_t chunk;         // byte offset of this wave's first chunk inside an image
    // k gather of a k-major operand (rows of the batch picked by an index list): the workgroup's slice of the list is
    // staged in LDS once (kidx), the row of slab t+1 is read while slab t is issued
    const int32_t* kidx;
    int64_t ldk;
    int knext[NP];

    // gperm > 0 (k-contiguous operand only): the tile's 64 rows are rows [16 t, 16 t + 16) of four blocks of `gperm` rows
    // (t = row0 / 64): the four gates of 16 hidden units of an LSTM weight matrix
    __device__ __forceinline__ void init(const float* __restrict__ P, int64_t ld, const int32_t* __restrict__ idx, int row0,
                                         int nrows, int k_begin, int wave, int lane, const int32_t* kidx_lds, int gperm = 0) {
        chunk = (uint32_t)(NP * wave) * 1024u;
        kidx = (KM && idx) ? kidx_lds : nullptr;
        ldk = ld;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = NP * wave + j;
            if (!KM) {
                const int r = 8 * c + (lane >> 3);
                const int kslot = (lane & 7) ^ ((r >> 1) & 7);
                const int grow = gperm ? min((r >> 4) * gperm + (row0 >> 2) + (r & 15), nrows - 1) : min(row0 + r, nrows - 1);
                const int64_t rr = idx ? (int64_t)idx[grow] : (int64_t)grow;
                kpiece[j] = 4 * kslot;
                safe[j] = P + rr * ld;
                ptr[j] = safe[j] + k_begin + kpiece[j];
            } else {
                const int k = 4 * c + (lane >> 4);
                const int rs = (lane & 15) ^ (8 * (c & 1));
                int rowseg = row0 + 4 * rs;
                if (rowseg >= nrows) rowseg = row0;
                kpiece[j] = k;
                safe[j] = P + rowseg;
                ptr[j] = safe[j] + (int64_t)(k_begin + k) * ld;
                knext[j] = kidx ? kidx[k] : 0;
            }
        }
        step = KM ? (int64_t)BK * ld : (int64_t)BK;
    }

    // piece j of slab t of this workgroup (first k = k0) -> image at LDS byte address img; slabs are issued in order
    __device__ __forceinline__ void issue1(int j, int t, int k0, int K, uint32_t img) {
        const bool tail = k0 + BK > K;   // workgroup-uniform; only the last slab of the product
        const float* g;
        if (KM && kidx) {   // workgroup-uniform
            g = safe[j] + (int64_t)knext[j] * ldk;   // the staged list is clamped to valid rows: no stand-in needed
            knext[j] = kidx[(t + 1) * BK + kpiece[j]];
        } else {
            g = ptr[j] + (int64_t)t * step;
            if (tail && k0 + kpiece[j] >= K) g = safe[j];
        }
        dma16(g, img + chunk + 1024u * j);
    }
    __device__ __forceinline__ void issue(int t, int k0, int K, uint32_t img) {
#pragma unroll
        for (int j = 0; j < NP; ++j) issue1(j, t, k0, K, img);
    }

    // out[j] = element (row R of the tile, k = 8s + 4h + j) of the image
    __device__ __forceinline__ void frag(const float* __restrict__ img, int R, int s, int h, float out[4]) const {
        if (!KM) {
            const int pos = (2 * s + h) ^ ((R >> 1) & 7);
            const f32x4 v = *reinterpret_cast<const f32x4*>(img + R * BK + 4 * pos);
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = v[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = img[(8 * s + 4 * h + j) * 64 + (R ^ (32 * h))];
        }
    }
};

// Epilogue of the LSTM input product G = [E | s_prev] W_ih[:, :c2]^T (async tile, four waves, no split; GemmExt in
// common.hpp): every row gets the bias vectors of its current and its previous address; with cell_H > 0 the tile holds
// the four gates of 16 hidden units, and rows of a trace's first time step (c_prev = 0, inference_network_lstm.py:186)
// go through the LSTM cell right here - pre-activations never travel to HBM and back, lstm_cell_fwd is not launched.
// (the bias of this lane's 16 accumulator rows: two dependent memory round trips - address ids, then the vectors - which
// the tile starts BEFORE its K loop so that they overlap the operand DMA)
__device__ __forceinline__ void lstm_row_bias(const GemmParams& p, const GemmExt& x, const int m0, const int bx, const int wm,
                                              const int wn, const int l31, const int h, float (&rbv)[16]) {
    const int H = x.cell_H;
    const int nloc = wn * 32 + l31;
    const int gn = H ? (nloc >> 4) * H + bx * 16 + (nloc & 15) : bx * 64 + nloc;
    const int gnc = gn < p.N ? gn : 0;
    if (!x.rb) {        // recurrent product: the "bias" of an element is its pre-activation, already in C
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, p.M - 1);
            rbv[r] = p.C[(int64_t)gm * p.ldc + gnc];
        }
        return;
    }
    if (!x.rb_addr) {   // every row has the same current address and no previous one: rb IS its bias vector (one load)
        const float b = x.rb[gnc];
#pragma unroll
        for (int r = 0; r < 16; ++r) rbv[r] = b;
        return;
    }
    int ia[16], ip[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, p.M - 1);
        ia[r] = x.rb_addr[gm];
        ip[r] = x.rb_prev ? x.rb_prev[gm] : -1;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) ip[r] = ip[r] >= 0 ? x.rb_addr[ip[r]] : -1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float b = x.rb[(int64_t)(2 * ia[r]) * p.N + gnc];
        if (ip[r] >= 0) b += x.rb[(int64_t)(2 * ip[r] + 1) * p.N + gnc];
        rbv[r] = b;
    }
}

__device__ __forceinline__ void lstm_epilogue(const GemmParams& p, const GemmExt& x, const f32x16& acc, const int m0,
                                              const int bx, const int wm, const int wn, const int l31, const int h,
                                              float* smem, const float (&rbv)[16]) {
    const int H = x.cell_H;
    const int nloc = wn * 32 + l31;
    const int gn = H ? (nloc >> 4) * H + bx * 16 + (nloc & 15) : bx * 64 + nloc;
    const bool ncol = gn < p.N;
    const bool fused = H > 0 && m0 < x.cell_rows;   // workgroup-uniform
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r] + rbv[r];
    if (!fused) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (gm < p.M && ncol) p.C[(int64_t)gm * p.ldc + gn] = v[r];
        }
        return;
    }
    constexpr int SS = 68;   // row stride of the exchange image (16-byte rows)
    __syncthreads();         // every wave is done with the ring
#pragma unroll
    for (int r = 0; r < 16; ++r) smem[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * SS + nloc] = v[r];
    __syncthreads();
    // thread t: row t >> 2, hidden units 4 (t & 3) .. + 3 of the tile's 16, all four gates
    const int tid = threadIdx.x;
    const int row = tid >> 2, q = tid & 3;
    const int gm = m0 + row;
    if (gm >= p.M) return;
    const float* sr = smem + row * SS + 4 * q;
    const f32x4 vi = *reinterpret_cast<const f32x4*>(sr);
    const f32x4 vf = *reinterpret_cast<const f32x4*>(sr + 16);
    const f32x4 vg = *reinterpret_cast<const f32x4*>(sr + 32);
    const f32x4 vo = *reinterpret_cast<const f32x4*>(sr + 48);
    const int u0 = bx * 16 + 4 * q;
    float* g = p.C + (int64_t)gm * p.ldc + u0;
    if (x.cell_cprev) {       // a later time step: the full cell (lstm_cell_fwd_kernel)
        const f32x4 cp = *reinterpret_cast<const f32x4*>(x.cell_cprev + (int64_t)gm * H + u0);
        f32x4 gi, gf, gg, go, cn, hn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gi[e] = sigmoidf_(vi[e]);
            gf[e] = sigmoidf_(vf[e]);
            gg[e] = tanhf(vg[e]);
            go[e] = sigmoidf_(vo[e]);
            cn[e] = gf[e] * cp[e] + gi[e] * gg[e];
            hn[e] = go[e] * tanhf(cn[e]);
        }
        *reinterpret_cast<f32x4*>(g) = gi;
        *reinterpret_cast<f32x4*>(g + H) = gf;
        *reinterpret_cast<f32x4*>(g + 2 * H) = gg;
        *reinterpret_cast<f32x4*>(g + 3 * H) = go;
        *reinterpret_cast<f32x4*>(x.cell_c + (int64_t)gm * H + u0) = cn;
        *reinterpret_cast<f32x4*>(x.cell_h + (int64_t)gm * H + u0) = hn;
        return;
    }
    if (gm < x.cell_rows) {   // torch.nn.LSTM gates i, f, g, o with c_prev = 0: the forget gate multiplies zero (0 recorded)
        f32x4 gi, gg, go, cn, hn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gi[e] = sigmoidf_(vi[e]);
            gg[e] = tanhf(vg[e]);
            go[e] = sigmoidf_(vo[e]);
            cn[e] = gi[e] * gg[e];
            hn[e] = go[e] * tanhf(cn[e]);
        }
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
        *reinterpret_cast<f32x4*>(g) = gi;
        *reinterpret_cast<f32x4*>(g + 2 * H) = gg;
        *reinterpret_cast<f32x4*>(g + 3 * H) = go;
        if (!x.lean) {
            *reinterpret_cast<f32x4*>(g + H) = zero;
            *reinterpret_cast<f32x4*>(x.cell_c + (int64_t)gm * H + u0) = cn;
        }
        *reinterpret_cast<f32x4*>(x.cell_h + (int64_t)gm * H + u0) = hn;
    } else {                  // a later time step in the same tile: pre-activations, the recurrent product follows
        *reinterpret_cast<f32x4*>(g) = vi;
        *reinterpret_cast<f32x4*>(g + H) = vf;
        *reinterpret_cast<f32x4*>(g + 2 * H) = vg;
        *reinterpret_cast<f32x4*>(g + 3 * H) = vo;
    }
}

// KW = 1: four waves, one 32x32 fragment each. KW = 2: eight waves; waves w and w+4 share a fragment and split every
// slab's k range in two (in-workgroup split-K, summed through LDS at the end). A single wave cannot hide its own
// non-MFMA instructions (DMA issue, fragment reads, barrier: ~1700 cycles per slab against 1024 of MFMA, measured
// with 1 wave per SIMD); a second wave on the SIMD does, and KW = 2 provides it when the launch has only about one
// workgroup per CU.
template <bool A_KM, bool B_KM, int KW>
__device__ __forceinline__ void gemm_tile_async(const GemmParams& pin, const int bx, const int by, const int bz, const int nz,
                                                float* smem, const GemmExt* xin = nullptr, long long* tr = nullptr) {
    const GemmParams p = pin;   // local copies: see gemm_tile
    GemmExt xv{};
    if (xin) xv = *xin;
    const GemmExt* const x = xin ? &xv : nullptr;
    constexpr int ST = AS_STAGES;
#define AS_STAMP(k) do { if (tr && threadIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); tr[k] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
    constexpr int NP = 2 / KW;        // DMA pieces per thread per operand per slab
    constexpr int D = 2 * NP;         // DMA instructions per thread per slab
    constexpr int NS = 4 / KW;        // 8-wide k sub-slabs per wave per slab
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l31 = lane & 31, h = lane >> 5;
    const int w4 = wave & 3, kh = wave >> 2;
    const int wm = w4 >> 1, wn = w4 & 1;
    const int m0 = by * 64, n0 = bx * 64;
    const int nslab_all = (p.K + BK - 1) / BK;
    // Zero block: the slabs [hs0, hs0 + hlen) contribute nothing to this tile and are left out of its slab sequence
    // (the LSTM input of a trace's first time step has no previous-variable columns: inference_network_lstm.py:159-162).
    int hs0 = 0, hlen = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {   // workgroup-uniform
        if (p.hs1[q] > p.hs0[q] && m0 >= p.hm0[q] && min(m0 + 64, p.M) <= p.hm1[q] && n0 >= p.hn0[q] &&
            min(n0 + 64, p.N) <= p.hn1[q] && !((A_KM && p.a_idx) || (B_KM && p.b_idx))) {
            const int len = min(p.hs1[q], nslab_all) - p.hs0[q];
            if (len > hlen) {
                hs0 = p.hs0[q];
                hlen = len;
            }
        }
    }
    const int nslab_total = nslab_all - hlen;
    const int per = (nslab_total + nz - 1) / nz;
    const int s_begin = bz * per;
    const int T = min(nslab_total, s_begin + per) - s_begin;
    if (T <= 0) {
        if (p.split_stride && nz > 1) {   // a split without slabs still owns its partial tile: zeros (workgroup-uniform)
            for (int e = tid; e < 64 * 64; e += 256 * KW) {
                const int gm = m0 + (e >> 6), gn = n0 + (e & 63);
                if (gm < p.M && gn < p.N) p.C[(int64_t)bz * p.split_stride + (int64_t)gm * p.ldc + gn] = 0.0f;
            }
        }
        return;
    }
    const bool split = nz > 1;
    // slab t of this workgroup = entry s_begin + t of the sequence without the zero block; srel(t) counts from its first
    auto slab_of = [&](int t) { const int u = s_begin + t; return u < hs0 ? u : u + hlen; };
    const int kb = slab_of(0) * BK;
    auto srel = [&](int t) { return slab_of(t) - kb / BK; };

    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.0f;
    const bool lstm = KW == 1 && !A_KM && !B_KM && x && (x->rb || x->cell_cprev);   // LSTM input / recurrent product
    float rbv[16];
    if (lstm) lstm_row_bias(p, *x, m0, bx, wm, wn, l31, h, rbv);

    // k-gather lists of this workgroup's K range, clamped to the last valid entry, one extra slab of padding
    int32_t* kia = reinterpret_cast<int32_t*>(smem + ST * AS_STAGE);
    int32_t* kib = kia + AS_KIDX;
    if ((A_KM && p.a_idx) || (B_KM && p.b_idx)) {   // workgroup-uniform
        for (int e = tid; e < (T + 1) * BK; e += 256 * KW) {
            const int k = min(kb + e, p.K - 1);
            if (A_KM && p.a_idx) kia[e] = p.a_idx[k];
            if (B_KM && p.b_idx) kib[e] = p.b_idx[k];
        }
        __syncthreads();
    }
    AsyncOperand<A_KM, NP> oa;
    AsyncOperand<B_KM, NP> ob;
    oa.init(p.A, p.lda, p.a_idx, m0, p.M, kb, wave, lane, kia);
    ob.init(p.B, p.ldb, p.b_idx, n0, p.N, kb, wave, lane, kib, (!B_KM && x) ? x->cell_H : 0);
    const uint32_t ring = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    auto issue = [&](int t) {
        const uint32_t img = ring + (uint32_t)(t & (ST - 1)) * (AS_STAGE * 4u);
        const int sr = srel(t);
        oa.issue(sr, kb + sr * BK, p.K, img);
        ob.issue(sr, kb + sr * BK, p.K, img + AS_IMG * 4u);
    };
    // fragments of slab t: image -> registers
    auto load_frags = [&](int t, float (&a)[NS][4], float (&b)[NS][4]) {
        const float* As = smem + (t & (ST - 1)) * AS_STAGE;
        const float* Bs = As + AS_IMG;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            oa.frag(As, wm * 32 + l31, kh * NS + s, h, a[s]);
            ob.frag(Bs, wn * 32 + l31, kh * NS + s, h, b[s]);
        }
    };
    auto mma = [&](const float (&a)[NS][4], const float (&b)[NS][4]) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][q], b[s][q], acc[0][0], 0, 0, 0);
    };
    // drain version: the last slab of the product may be partial - its pieces beyond K hold stand-in data
    auto mma_edge = [&](int t, float (&a)[NS][4], float (&b)[NS][4]) {
        const int k0 = kb + srel(t) * BK;
        if (k0 + BK > p.K) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k0 + 8 * (kh * NS + s) + 4 * h + j >= p.K) { a[s][j] = 0.0f; b[s][j] = 0.0f; }
        }
        mma(a, b);
    };
    // slab t has landed when at most the younger slabs (D DMA instructions each) are outstanding
    auto wait_younger = [&](int rem) {
        switch (rem) {
            case 0: __builtin_amdgcn_s_waitcnt(vmcnt_imm(0)); break;
            case 1: __builtin_amdgcn_s_waitcnt(vmcnt_imm(D)); break;
            case 2: __builtin_amdgcn_s_waitcnt(vmcnt_imm(2 * D)); break;
            default: __builtin_amdgcn_s_waitcnt(vmcnt_imm(3 * D)); break;
        }
        __syncthreads();   // every wave's pieces are visible; every wave is done reading the previous image
    };
    // Ring of ST images. Steady state of step i: slab i sits in registers, slab i+1 is read from its image while the
    // MFMAs of slab i run (the reads are independent of the MFMA chain and issue in its shadow), slabs i+2 .. i+ST-1
    // are in flight, and slab i+ST is started into the image slab i just vacated.
    static_assert(ST == 4, "ring depth");
    AS_STAMP(4);
#pragma unroll
    for (int t = 0; t < ST; ++t)
        if (t < T) issue(t);
    float a0[NS][4], b0[NS][4], a1[NS][4], b1[NS][4];
    AS_STAMP(5);
    wait_younger(min(T - 1, ST - 1));
    AS_STAMP(6);
    load_frags(0, a0, b0);
    // One steady-state slab: the MFMAs of the slab in (ca, cb), with the fragment reads of slab tn (-> na, nb) and the
    // DMA of slab td dealt out one per MFMA (the MFMAs form a dependent chain, 64 cycles each, and the wave issues in
    // order, so only an instruction placed BETWEEN two MFMAs can run in their shadow).
    // Measured on workgroup 0, one wave per SIMD, cycles per slab: 1162 MFMAs alone, +35 barrier, +170 fragment reads,
    // +160 DMA issue = ~1430-1530 against 1024 ideal; dealing the side instructions out one per MFMA instead of the
    // compiler's clusters, a second accumulator chain, eight waves per tile and an 8-deep ring each moved it < 7 %.
    // The 32x32 wave tile pays one fragment dword per MFMA; the 64x64 wave tile of the large-M kernel pays a quarter.
    auto step = [&](int tn, int td, const float (&ca)[NS][4], const float (&cb)[NS][4], float (&na)[NS][4],
                    float (&nb)[NS][4]) {
        const float* As = smem + (tn & (ST - 1)) * AS_STAGE;
        const float* Bs = As + AS_IMG;
        const bool dma = td < T;
        const uint32_t img = ring + (uint32_t)(td & (ST - 1)) * (AS_STAGE * 4u);
        const int sd = srel(td);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[s][q], cb[s][q], acc[0][0], 0, 0, 0);
                const int slot = 4 * s + q;   // side instruction of this MFMA
                if (slot < NS) oa.frag(As, wm * 32 + l31, kh * NS + slot, h, na[slot]);
                else if (slot < 2 * NS) ob.frag(Bs, wn * 32 + l31, kh * NS + slot - NS, h, nb[slot - NS]);
                else if (slot < 2 * NS + NP) { if (dma) oa.issue1(slot - 2 * NS, sd, kb + sd * BK, p.K, img); }
                else if (slot < 2 * NS + 2 * NP) { if (dma) ob.issue1(slot - 2 * NS - NP, sd, kb + sd * BK, p.K, img + AS_IMG * 4u); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    int i = 0;
    // main loop: two slabs per trip, always ST-2 younger slabs in flight (constant wait counts, no edge handling)
    for (; i + ST + 1 <= T; i += 2) {
        if (p.dbg && tid == 0 && bx == 0 && by == 0 && bz == 0 && i < 96) p.dbg[16 + (i >> 1)] = clock64();   // debug
        __builtin_amdgcn_s_waitcnt(vmcnt_imm(D * (ST - 2)));
        __syncthreads();
        step(i + 1, i + ST, a0, b0, a1, b1);
        __builtin_amdgcn_s_waitcnt(vmcnt_imm(D * (ST - 2)));
        __syncthreads();
        step(i + 2, i + 1 + ST, a1, b1, a0, b0);
    }
    // drain: slab i is in (a0, b0), at most ST slabs remain and nothing is left to issue
    for (; i < T; i += 2) {
        if (i + 1 < T) {
            wait_younger(min(T - 2 - i, ST - 2));
            load_frags(i + 1, a1, b1);
        }
        mma_edge(i, a0, b0);
        if (i + 1 >= T) break;
        if (i + 2 < T) {
            wait_younger(min(T - 3 - i, ST - 2));
            load_frags(i + 2, a0, b0);
        }
        mma_edge(i + 1, a1, b1);
    }
    AS_STAMP(7);
    if (KW == 2) {   // waves 4..7 hand their partial fragment to waves 0..3
        __syncthreads();   // the ring is free
        float* part = smem + w4 * 1024;
        if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) part[r * 64 + lane] = acc[0][0][r];
        }
        __syncthreads();
        if (kh == 1) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += part[r * 64 + lane];
    }
    if (lstm) {
        lstm_epilogue(p, *x, acc[0][0], m0, bx, wm, wn, l31, h, smem, rbv);
        return;
    }
    tile_epilogue<64, 64, 32, 32, 1, 1>(p, acc, m0, n0, wm, wn, l31, h, bz, split);
}

extern __shared__ __attribute__((aligned(1024))) float as_ring[];   // ST x AS_STAGE floats, sized at launch

template <bool A_KM, bool B_KM, int KW>
__global__ __launch_bounds__(256 * KW) void gemm_f32_async_kernel(const GemmParams p) {
    gemm_tile_async<A_KM, B_KM, KW>(p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z, as_ring);
}

// the LSTM input product: per-row address bias, gate-interleaved tiles, fused cell (lstm_epilogue)
// tiles_per_wg > 1: a workgroup walks that many consecutive column tiles of its row panel (a ragged batch's 2 600 rows x 32
// column tiles are 1 344 workgroups of ~10 us of mostly start-up latency each for 512 resident slots: 28 us)
__global__ __launch_bounds__(256) void gemm_f32_async_lstm_kernel(const GemmParams p, const GemmExt x, const int tiles_per_wg,
                                                                  const int gx) {
    const int b0 = blockIdx.x * tiles_per_wg, b1 = min(gx, b0 + tiles_per_wg);
    for (int bx = b0; bx < b1; ++bx) {
        gemm_tile_async<false, false, 1>(p, bx, blockIdx.y, 0, 1, as_ring, &x);
        if (bx + 1 < b1) __syncthreads();   // the next tile's DMA reuses the ring / exchange image
    }
}

// ---- a handful of rows (M <= GEMV_ROWS): one wave per output column ------------------------------------------
// The first statement of a lock-step importance-sampling run evaluates the LSTM and the proposal head for ONE shared
// row. A 32x32 MFMA tile would be 97 % padding and still walk the K slabs; here wave w of the launch owns column n,
// its lanes stride over K with 16-byte loads of W[n, :] (k-contiguous B only), the A rows come from L1/L2, and a DPP
// reduction finishes the dot products. Epilogue: bias, bias2, ReLU, accumulate.
constexpr int GEMV_ROWS = 4;

template <int VEC>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const GemmParams p) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= p.N) return;   // wave-uniform
    const int64_t bn = p.b_idx ? (int64_t)p.b_idx[n] : (int64_t)n;
    const float* __restrict__ w = p.B + bn * p.ldb;
    float acc[GEMV_ROWS];
#pragma unroll
    for (int m = 0; m < GEMV_ROWS; ++m) acc[m] = 0.0f;
    const float* arow[GEMV_ROWS];
#pragma unroll
    for (int m = 0; m < GEMV_ROWS; ++m) {
        const int mm = m < p.M ? m : 0;
        arow[m] = p.A + (p.a_idx ? (int64_t)p.a_idx[mm] : (int64_t)mm) * p.lda;
    }
    if (VEC == 4) {
        const int K4 = p.K & ~3;
        for (int k = lane * 4; k < K4; k += 256) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
#pragma unroll
            for (int m = 0; m < GEMV_ROWS; ++m) {
                if (m < p.M) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(arow[m] + k);
                    acc[m] += av[0] * wv[0] + av[1] * wv[1] + av[2] * wv[2] + av[3] * wv[3];
                }
            }
        }
        for (int k = K4 + lane; k < p.K; k += 64) {
#pragma unroll
            for (int m = 0; m < GEMV_ROWS; ++m)
                if (m < p.M) acc[m] += arow[m][k] * w[k];
        }
    } else {
        for (int k = lane; k < p.K; k += 64) {
#pragma unroll
            for (int m = 0; m < GEMV_ROWS; ++m)
                if (m < p.M) acc[m] += arow[m][k] * w[k];
        }
    }
    float bsum = 0.0f;
    if (p.bias) bsum += p.bias[n];
    if (p.bias2) bsum += p.bias2[n];
#pragma unroll
    for (int m = 0; m < GEMV_ROWS; ++m) {
        if (m >= p.M) break;
        float v = wave_sum(acc[m]) + bsum;
        if (lane == 0) {
            if (p.relu) v = relu_keep_nan(v);
            float* dst = p.C + (p.c_idx ? (int64_t)p.c_idx[m] : (int64_t)m) * p.ldc + n;
            *dst = p.accumulate ? *dst + v : v;
        }
    }
}

static bool gemv_ok(const pp_gemm_args* a) {
    return a->M <= GEMV_ROWS && !a->a_kmajor && !a->b_kmajor && !a->mask && !a->colsum;
}

static int launch_gemv(const GemmParams& p, bool vec, hipStream_t st) {
    dim3 grid(cdiv(p.N, 4)), block(256);
    if (vec) hipLaunchKernelGGL(gemv_rows_kernel<4>, grid, block, 0, st, p);
    else hipLaunchKernelGGL(gemv_rows_kernel<1>, grid, block, 0, st, p);
    PP_LAUNCH_CHECK("pp_gemm_f32 (gemv)");
    return 0;
}

// Several independent products in ONE launch (the weight-gradient leaves of the backward pass, the per-address head
// products of a ragged batch): workgroup b finds its problem in the prefix table and runs the same tile code.
constexpr int GROUP_MAX = 16;
struct GroupedParams {
    GemmParams p[GROUP_MAX];
    int first[GROUP_MAX + 1];   // first workgroup of problem q; first[count] = total
    int gx[GROUP_MAX], gy[GROUP_MAX], gz[GROUP_MAX];
    int pmode[GROUP_MAX];       // XCD placement of the problem's tiles, see group_decode
    int rot[GROUP_MAX];         // pmode 0: tile t of the problem runs on XCD (t + rot) % 8
    int count;
    int xcd_aware;
    GemmExt ext;                // shared by the problems of the launch (cell backward in the dH epilogue)
    int warm;                   // touch the argument lines with one vector load first
    long long* trace;           // debug (pp_debug_wgtrace): per workgroup {start, end} wall-clock ticks (10 ns), problem, split
};
long long* g_wgtrace = nullptr;   // device buffer [8 x workgroups] or nullptr
int g_wgtrace_cap = 0, g_wgtrace_mode = 0;
__device__ __forceinline__ void wg_trace(const GroupedParams& g, int slot, long long t0, int q, int bz) {
    if (g.trace && threadIdx.x == 0) {
        g.trace[8 * slot + 0] = t0;
        g.trace[8 * slot + 1] = wall_clock64();
        g.trace[8 * slot + 2] = q;
        g.trace[8 * slot + 3] = bz;
    }
}

// Workgroup -> (problem, tile, K split). Workgroups are dealt round-robin to the 8 XCDs (id % 8) and each XCD has its
// own 4 MB L2; every problem starts at a workgroup id that is a multiple of 8, so local id & 7 IS the XCD.
//   * all K splits of one output tile get the same residue: their float atomics meet in ONE L2;
//   * pmode 1 (M-panel placement): all tiles of the row panel `by` run on XCD by % 8, column tiles and splits of a
//     panel on consecutive ids - the A panel (the large operand of dW_ih = dG^T X and of dX = dG W_ih: dG, 8 MB) comes
//     from HBM into ONE L2 once instead of once per column tile, B (X / W_ih, < 2 MB) lives in every L2;
//   * pmode 2: the same along N (B panel shared), for products whose B operand is the large one;
//   * pmode 0: tiles round-robin over the XCDs (few tiles in both directions).
// Workgroup counts are padded (panels to a multiple of 8); the padding workgroups exit.
__device__ __forceinline__ bool group_decode(const GroupedParams& g, int b, int& q, int& bx, int& by, int& bz) {
    q = 0;
    while (q + 1 < g.count && b >= g.first[q + 1]) ++q;   // workgroup-uniform
    const int l = b - g.first[q];
    const int gx = g.gx[q], gy = g.gy[q], gz = g.gz[q];
    if (g.xcd_aware && g.pmode[q] == 1) {
        const int r = l >> 3, per = gx * gz, w = r % per;
        by = (r / per) * 8 + (l & 7);
        bx = w % gx;
        bz = w / gx;
        return by < gy;
    }
    if (g.xcd_aware && g.pmode[q] == 2) {
        const int r = l >> 3, per = gy * gz, w = r % per;
        bx = (r / per) * 8 + (l & 7);
        by = w % gy;
        bz = w / gy;
        return bx < gx;
    }
    const int ntiles = gx * gy;
    int tile;
    if (g.xcd_aware) {
        // (rot continues the round robin across the problems of a group: without it tile 0 of EVERY problem - and all its K
        // splits - sat on XCD 0; the four single-tile observe-embedding leaves of the weight-gradient group shared its 32 CUs
        // with an eighth of everything else and were the last workgroups of the launch to finish, tools/wg_trace.py)
        const int r = l >> 3;
        bz = r % gz;
        tile = (r / gz) * 8 + (((l & 7) - g.rot[q]) & 7);
    } else {
        tile = l % ntiles;
        bz = l / ntiles;
    }
    if (tile >= ntiles || bz >= gz) return false;
    bx = tile % gx;
    by = tile / gx;
    return true;
}

// placement of a problem: share the panel of the LARGER operand if that direction has at least 8 tiles to spread
static inline int pick_pmode(int M, int N, int gx, int gy) {
    static const int mode = 1;
    if (!mode) return 0;
    if (M >= N) return gy >= 8 ? 1 : (gx >= 8 ? 2 : 0);
    return gx >= 8 ? 2 : (gy >= 8 ? 1 : 0);
}
static inline int group_blocks(int gx, int gy, int gz, int pmode = 0) {
    if (pmode == 1) return ((gy + 7) / 8) * 8 * gx * gz;
    if (pmode == 2) return ((gx + 7) / 8) * 8 * gy * gz;
    return ((gx * gy + 7) / 8) * 8 * gz;
}

template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM, int VEC>
__global__ __launch_bounds__(256) void gemm_f32_grouped_kernel(const GroupedParams g) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * (BK + KPAD)];   // double-buffered K slabs
    if (g.warm) warm_kernargs((int)sizeof(GroupedParams));
    int q, bx, by, bz;
    if (!group_decode(g, blockIdx.x, q, bx, by, bz)) return;
    // VEC == 4 instantiation: problems whose leading dimensions / pointers are not 16-byte friendly (e.g. the 2-wide
    // observation matrix) fall back to scalar staging individually instead of degrading the whole group
    if (VEC == 4 && !g.p[q].vec) gemm_tile<BM, BN, WM, WN, A_KM, B_KM, 1>(g.p[q], bx, by, bz, g.gz[q], smem);
    else gemm_tile<BM, BN, WM, WN, A_KM, B_KM, VEC>(g.p[q], bx, by, bz, g.gz[q], smem);
}

// ---- wave-direct tiles: small, latency-bound products ------------------------------------------------------
// The LDS-staged tile above needs >= 2 resident workgroups per CU to hide its load -> LDS -> barrier -> MFMA chain; a
// product with only ~100 64x64 tiles (the proposal-head layers, the data gradients, the weight gradients) leaves most
// CUs idle and walks its K slabs at ~1.9 us each. Here a workgroup owns a 32x32 output tile and its four waves split
// the K slabs between them (wave w: slabs w, w+4, ...): operands go global -> registers directly in MFMA operand
// layout (no LDS staging, no barrier in the K loop, two register sets in flight), the four partial tiles are summed
// through LDS and the epilogue runs row-contiguous (4 columns per thread). 4x the tiles, 1/4 of the serial chain.
constexpr int DT = 32;           // direct tile edge
constexpr int DPAD = DT + 1;     // LDS row stride of the partial tiles
constexpr int DSTR = 40;         // LDS row stride of a wave-private k-major slab image: 4*DSTR = 32 (mod 64) puts the two
                                 // half-waves of a fragment read on disjoint banks
constexpr int DSLAB = BK * DSTR; // floats per wave-private slab image

// One 32-row operand of a wave: fetch() global -> raw registers (16 floats per lane per slab), put() raw -> the wave's
// private LDS image, frag() image -> MFMA operand registers f[4s + j] = element (row l31, k0 + 8s + 4h + j).
// Lanes fetch 16-byte segments along the CONTIGUOUS direction of the operand, 8 lanes per 128-byte line (k for a
// k-contiguous operand, rows for a k-major one), and the image transposes to the fragment layout - wave-local, no
// workgroup barrier. (Fetching the fragment layout straight from global memory was tried first: 32 lines x 32 bytes
// per load instruction for k-contiguous operands, sixteen 4-byte loads per slab for k-major ones; the L1/address
// path became the bottleneck and the long-K products ran 1.6x slower than LDS-staged tiles.)
//   k-contiguous image: [row][BK+KPAD]  (one ds_read_b128 per sub-slab, conflict free as in the staged kernel)
//   k-major image:      [k][DSTR]       (ds_read_b32, consecutive rows on consecutive banks)
template <bool KM, int VEC>
struct DirectOperand {
    const float* base[4];   // !KM: row pointer of raw slot i (+ this lane's k offset); KM: base[0] = &P[row0 + rcol]
    const int32_t* idx;
    int64_t ld;
    bool okr[4];            // !KM: raw slot i's row inside the matrix
    bool ok, ok4;
    int h, l31, krow, rcol, nleft;
    float* img;

    __device__ __forceinline__ void init(const float* __restrict__ P, int64_t ld_, const int32_t* __restrict__ idx_,
                                         int row0, int nrows, int lane, float* img_) {
        h = lane >> 5;
        l31 = lane & 31;
        krow = lane >> 3;
        rcol = 4 * (lane & 7);
        ld = ld_;
        idx = idx_;
        img = img_;
        if (!KM) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + krow + 8 * i;
                okr[i] = row < nrows;
                const int rc = min(row, nrows - 1);      // a row beyond the matrix reads the last one and is zeroed
                const int64_t rr = idx ? (int64_t)idx[rc] : (int64_t)rc;
                base[i] = P + rr * ld + rcol;
            }
        } else {
            nleft = nrows - (row0 + rcol);   // rows of this lane's segment inside the matrix
            ok = nleft > 0;
            ok4 = nleft >= 4;
            base[0] = P + row0 + (ok ? rcol : 0);   // a segment beyond the matrix reads the tile's first one and is zeroed
        }
    }

    // VEC == 4 is branch-free: the load address is clamped into the matrix (rows at init, k here) and elements outside
    // are zeroed with selects. The predicated element-wise fallback this replaces cost ~900 instructions (124 exec
    // branches) for the first slab of a workgroup - 1.2 us of a 10 us kernel (tools/kernel_timeline.py).
    __device__ __forceinline__ void fetch(float g[16], int k0, int K) const {
        if (!KM) {
            const int kleft = K - (k0 + rcol);
            if (VEC == 4) {
                // a piece that starts inside K may straddle it: the row is ld >= round4(K) floats long, so the read stays
                // inside the row; a piece beyond K is read from the row start instead
                const int koff = kleft > 0 ? k0 : -rcol;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(base[i] + koff);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[4 * i + e] = (okr[i] && e < kleft) ? v[e] : 0.0f;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float* src = base[i] + k0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[4 * i + e] = (okr[i] && e < kleft) ? src[e] : 0.0f;
                }
            }
        } else {
            if (VEC == 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = k0 + krow + 8 * i;
                    const int kc = min(k, K - 1);
                    const int64_t kk = idx ? (int64_t)idx[kc] : (int64_t)kc;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(base[0] + kk * ld);   // base[0] is clamped at init
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[4 * i + e] = (k < K && e < nleft) ? v[e] : 0.0f;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = k0 + krow + 8 * i;
                    if (ok && k < K) {
                        const int64_t kk = idx ? (int64_t)idx[k] : (int64_t)k;
                        const float* src = base[0] + kk * ld;
#pragma unroll
                        for (int e = 0; e < 4; ++e) g[4 * i + e] = e < nleft ? src[e] : 0.0f;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) g[4 * i + e] = 0.0f;
                    }
                }
            }
        }
    }

    __device__ __forceinline__ void put(const float g[16]) const {
        constexpr int STR = KM ? DSTR : (BK + KPAD);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = g[4 * i + e];
            *reinterpret_cast<f32x4*>(img + (krow + 8 * i) * STR + rcol) = v;
        }
    }

    __device__ __forceinline__ void frag(float f[16]) const {
        if (!KM) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(img + l31 * (BK + KPAD) + 8 * s + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) f[4 * s + j] = v[j];
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) f[4 * s + j] = img[(8 * s + 4 * h + j) * DSTR + l31];
        }
    }
};

template <bool A_KM, bool B_KM>
constexpr int direct_lds_floats() {
    constexpr int stage = 2 * 4 * DSLAB;   // A and B images of four waves
    return stage > 4 * DT * DPAD ? stage : 4 * DT * DPAD;
}

template <bool A_KM, bool B_KM, int VEC>
__device__ __forceinline__ void gemm_tile_direct(const GemmParams& pin, const int bx, const int by, const int bz, const int nz,
                                                 float* red, const GemmExt* xin = nullptr) {
    const GemmParams p = pin;   // local copies: see gemm_tile
    GemmExt xv{};
    if (xin) xv = *xin;
    const GemmExt* const x = xin ? &xv : nullptr;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = by * DT, n0 = bx * DT;
    if (p.dbg && tid == 0 && bx == 1 && by == 1 && bz == 0) p.dbg[32] = clock64();   // debug phase stamps
    const int nslab_total = (p.K + BK - 1) / BK;
    // Zero blocks (GemmHole): this tile family honours the case that matters for correctness - a tile that lies inside a block
    // whose slab range is ALL of K contributes nothing and may be fed by columns nobody wrote (the forget gate's part of
    // dG for first-time-step rows in the lean mode, GemmExt::lean): it is skipped. Partial-K blocks are simply computed.
#pragma unroll
    for (int q = 0; q < 2; ++q) {   // workgroup-uniform
        if (p.hs1[q] > p.hs0[q] && p.hs0[q] == 0 && p.hs1[q] >= nslab_total && m0 >= p.hm0[q] && min(m0 + DT, p.M) <= p.hm1[q] &&
            n0 >= p.hn0[q] && min(n0 + DT, p.N) <= p.hn1[q] && !((A_KM && p.a_idx) || (B_KM && p.b_idx)))
            return;
    }
    const int per = (nslab_total + nz - 1) / nz;
    const int s_begin = bz * per;
    const int s_end = min(nslab_total, s_begin + per);
    if (s_end <= s_begin) return;   // workgroup-uniform
    const bool split = nz > 1;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // cell backward in the epilogue (below): this thread's gates and cell state are fetched now - row gather index, then
    // four 16-byte loads: two dependent round trips that would otherwise sit at the end of the kernel
    const bool cellbw = !A_KM && B_KM && x && x->bw_G;   // workgroup-uniform
    f32x4 pgi = {0.f, 0.f, 0.f, 0.f}, pgg = pgi, pgo = pgi, pcc = pgi;
    if (cellbw) {
        const int prow = m0 + (tid >> 3), pc4 = (tid & 7) * 4;
        if (prow < p.M && n0 + pc4 < p.N) {
            const int64_t pcm = p.c_idx ? (int64_t)p.c_idx[prow] : (int64_t)prow;
            const float* g = x->bw_G + pcm * 4 * x->bw_H + n0 + pc4;
            pgi = *reinterpret_cast<const f32x4*>(g);
            pgg = *reinterpret_cast<const f32x4*>(g + 2 * x->bw_H);
            pgo = *reinterpret_cast<const f32x4*>(g + 3 * x->bw_H);
            if (!x->lean) pcc = *reinterpret_cast<const f32x4*>(x->bw_C + pcm * x->bw_H + n0 + pc4);
            else
#pragma unroll
                for (int e = 0; e < 4; ++e) pcc[e] = pgi[e] * pgg[e];   // c = i g (c_prev = 0)
        }
    }
    DirectOperand<A_KM, VEC> oa;
    DirectOperand<B_KM, VEC> ob;
    oa.init(p.A, p.lda, p.a_idx, m0, p.M, lane, red + wave * DSLAB);
    ob.init(p.B, p.ldb, p.b_idx, n0, p.N, lane, red + (4 + wave) * DSLAB);
    // (A three-deep register ring - three slabs of loads in flight - was tried: 182 VGPRs and no gain. Phase stamps of
    // the head-layer product, tools/kernel_timeline.py: init 0.7 us, issuing the first loads 1.2 us (~900 instructions of
    // address arithmetic and edge predication), first data 2-3 us after issue (operands come from another XCD's L2 /
    // HBM), the remaining slabs < 0.2 us: the kernel is start-up latency, not the K loop.)
#define DSTAMP(k) do { if (p.dbg && tid == 0 && bx == 1 && by == 1 && bz == 0) p.dbg[32 + (k)] = clock64(); } while (0)
    DSTAMP(1);
    float ga[16], gb[16], a[16], b[16];
    int s = s_begin + wave;
    if (s < s_end) {
        oa.fetch(ga, s * BK, p.K);
        ob.fetch(gb, s * BK, p.K);
    }
    DSTAMP(2);
    bool first_slab = true;
    while (s < s_end) {
        oa.put(ga);
        ob.put(gb);
        // wave-private images: order the wave's writes before its reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        oa.frag(a);
        ob.frag(b);
        __builtin_amdgcn_wave_barrier();
        s += 4;
        if (s < s_end) {   // next slab's loads fly under this slab's MFMAs
            oa.fetch(ga, s * BK, p.K);
            ob.fetch(gb, s * BK, p.K);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
        if (first_slab) { DSTAMP(3); first_slab = false; }
    }
    if (p.dbg && acc[0] == 123456.789f) p.dbg[63] = 1;   // debug: the stamp must follow the last MFMA
    DSTAMP(4);
    __syncthreads();   // the partial tiles reuse the staging area
    // partial tiles -> LDS (D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    float* mine = red + wave * DT * DPAD;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * h) * DPAD + l31] = acc[r];
    __syncthreads();
    // epilogue: thread t owns row t>>3, columns 4*(t&7) .. +3
    const int row = tid >> 3, c4 = (tid & 7) * 4;
    const int gm = m0 + row;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int o = row * DPAD + c4 + e;
        v[e] = (red[o] + red[DT * DPAD + o]) + (red[2 * DT * DPAD + o] + red[3 * DT * DPAD + o]);
    }
    const bool rowok = gm < p.M;
    const int64_t cm = rowok ? (p.c_idx ? (int64_t)p.c_idx[gm] : (int64_t)gm) : 0;
    if (cellbw) {
        // The tile is dh of 32 single-statement rows x 32 hidden units (dH = dZ1 W1, all rows of one address): run the
        // LSTM cell backward here (no dc carry, c_prev = 0: lstm_cell_bwd_kernel with n_next = 0, c_prev = NULL) - the
        // gates in bw_G become dG in place, dH is never stored - and add the tile's column sums of dG to the address's
        // group sums (p.colsum: [4 H], the forget-gate part stays zero).
        const int H = x->bw_H;
        float d0[4] = {0.f, 0.f, 0.f, 0.f}, d2[4] = {0.f, 0.f, 0.f, 0.f}, d3[4] = {0.f, 0.f, 0.f, 0.f};
        if (rowok && n0 + c4 < p.N) {   // (N = H is a multiple of 4: all four columns or none)
            float* g = x->bw_G + cm * 4 * H + n0 + c4;
            const f32x4 gi = pgi, gg = pgg, go = pgo, cc = pcc;   // (fetched before the K loop)
            f32x4 o0, o2, o3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float tc = tanhf(cc[e]);
                const float dhv = v[e];
                const float dc = dhv * go[e] * (1.0f - tc * tc);
                d0[e] = dc * gg[e] * gi[e] * (1.0f - gi[e]);
                d2[e] = dc * gi[e] * (1.0f - gg[e] * gg[e]);
                d3[e] = dhv * tc * go[e] * (1.0f - go[e]);
                o0[e] = d0[e]; o2[e] = d2[e]; o3[e] = d3[e];
            }
            const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
            *reinterpret_cast<f32x4*>(g) = o0;
            if (!x->lean) *reinterpret_cast<f32x4*>(g + H) = zero;
            *reinterpret_cast<f32x4*>(g + 2 * H) = o2;
            *reinterpret_cast<f32x4*>(g + 3 * H) = o3;
        }
        __syncthreads();   // the partial tiles have been read
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[row * DPAD + c4 + e] = d0[e];
            red[(DT + row) * DPAD + c4 + e] = d2[e];
            red[(2 * DT + row) * DPAD + c4 + e] = d3[e];
        }
        __syncthreads();
        if (tid < 3 * DT && p.colsum) {
            const int gate = tid >> 5, col = tid & 31;
            if (n0 + col < p.N) {
                float cs = 0.0f;
#pragma unroll 8
                for (int r = 0; r < DT; ++r) cs += red[(gate * DT + r) * DPAD + col];
                atomicAdd(p.colsum + (gate == 0 ? 0 : gate + 1) * H + n0 + col, cs);
            }
        }
        return;
    }
    float* dst = p.C + cm * p.ldc + n0 + c4;
    const bool lead = bz == 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int gn = n0 + c4 + e;
        if (gn < p.N && lead) {
            if (p.bias) v[e] += p.bias[gn];
            if (p.bias2) v[e] += p.bias2[gn];
        }
    }
    if (split) {
        if (rowok) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n0 + c4 + e < p.N) atomicAdd(dst + e, v[e]);
        }
        return;
    }
    if (p.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = relu_keep_nan(v[e]);
    }
    if (p.mask && rowok) {
        const float* mk = p.mask + cm * p.ldmask + n0 + c4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n0 + c4 + e < p.N) v[e] = mk[e] > 0.0f ? v[e] : 0.0f;
    }
    if (p.colsum) {   // workgroup-uniform: column sums of the finished tile (bias gradients)
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) red[row * DPAD + c4 + e] = (rowok && n0 + c4 + e < p.N) ? v[e] : 0.0f;
        __syncthreads();
        if (tid < DT && n0 + tid < p.N) {
            float cs = 0.0f;
#pragma unroll 8
            for (int r = 0; r < DT; ++r) cs += red[r * DPAD + tid];
            atomicAdd(p.colsum + n0 + tid, cs);
        }
    }
    if (!rowok) return;
    if (p.vec_c && n0 + c4 + 3 < p.N) {
        f32x4 o;
        if (p.accumulate) {
            o = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += v[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[e];
        }
        *reinterpret_cast<f32x4*>(dst) = o;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n0 + c4 + e < p.N) dst[e] = p.accumulate ? dst[e] + v[e] : v[e];
    }
}

template <bool A_KM, bool B_KM, int VEC>
__global__ __launch_bounds__(256) void gemm_f32_direct_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float red[direct_lds_floats<A_KM, B_KM>()];
    gemm_tile_direct<A_KM, B_KM, VEC>(p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z, red);
}

template <bool A_KM, bool B_KM, int VEC>
__global__ __launch_bounds__(256) void gemm_f32_direct_grouped_kernel(const GroupedParams g) {
    __shared__ __attribute__((aligned(16))) float red[direct_lds_floats<A_KM, B_KM>()];
    if (g.warm) warm_kernargs((int)sizeof(GroupedParams));
    int q, bx, by, bz;
    if (!group_decode(g, blockIdx.x, q, bx, by, bz)) return;
    if (VEC == 4 && !g.p[q].vec) gemm_tile_direct<A_KM, B_KM, 1>(g.p[q], bx, by, bz, g.gz[q], red, &g.ext);
    else gemm_tile_direct<A_KM, B_KM, VEC>(g.p[q], bx, by, bz, g.gz[q], red, &g.ext);
}

// weight-gradient groups (both operands k-major) with the backward pass's small reduction jobs behind the tiles
__global__ __launch_bounds__(256) void gemm_f32_direct_grouped_aux_kernel(const GroupedParams g, const AuxJobs aux) {
    __shared__ __attribute__((aligned(16))) float red[direct_lds_floats<true, true>()];
    const int nb = g.first[g.count];
    if ((int)blockIdx.x >= nb) {
        aux_job_run(aux, (int)blockIdx.x - nb, red);
        return;
    }
    if (g.warm) warm_kernargs((int)sizeof(GroupedParams));
    int q, bx, by, bz;
    if (!group_decode(g, blockIdx.x, q, bx, by, bz)) return;
    if (!g.p[q].vec) gemm_tile_direct<true, true, 1>(g.p[q], bx, by, bz, g.gz[q], red);
    else gemm_tile_direct<true, true, 4>(g.p[q], bx, by, bz, g.gz[q], red);
}

template <int VEC>
static int launch_direct(const GemmParams& p, bool akm, bool bkm, int splits, hipStream_t st) {
    dim3 grid(cdiv(p.N, DT), cdiv(p.M, DT), splits);
    dim3 block(256);
    if (!akm && !bkm) hipLaunchKernelGGL((gemm_f32_direct_kernel<false, false, VEC>), grid, block, 0, st, p);
    else if (!akm && bkm) hipLaunchKernelGGL((gemm_f32_direct_kernel<false, true, VEC>), grid, block, 0, st, p);
    else if (akm && !bkm) hipLaunchKernelGGL((gemm_f32_direct_kernel<true, false, VEC>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_f32_direct_kernel<true, true, VEC>), grid, block, 0, st, p);
    PP_LAUNCH_CHECK("pp_gemm_f32 (direct)");
    return 0;
}

static int launch_direct_grouped(const GroupedParams& g, bool akm, bool bkm, hipStream_t st) {
    dim3 grid(g.first[g.count]), block(256);
    if (!akm && !bkm) hipLaunchKernelGGL((gemm_f32_direct_grouped_kernel<false, false, 4>), grid, block, 0, st, g);
    else if (!akm && bkm) hipLaunchKernelGGL((gemm_f32_direct_grouped_kernel<false, true, 4>), grid, block, 0, st, g);
    else if (akm && !bkm) hipLaunchKernelGGL((gemm_f32_direct_grouped_kernel<true, false, 4>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_direct_grouped_kernel<true, true, 4>), grid, block, 0, st, g);
    PP_LAUNCH_CHECK("pp_gemm_f32_grouped (direct)");
    return 0;
}

template <bool A_KM, bool B_KM, int KW>
__global__ __launch_bounds__(256 * KW) void gemm_f32_async_grouped_kernel(const GroupedParams g) {
    const long long t0 = g.trace ? wall_clock64() : 0;
    if (g.warm) warm_kernargs((int)sizeof(GroupedParams));
    int q, bx, by, bz;
    if (!group_decode(g, blockIdx.x, q, bx, by, bz)) return;
    gemm_tile_async<A_KM, B_KM, KW>(g.p[q], bx, by, bz, g.gz[q], as_ring, nullptr, g.trace ? g.trace + 8 * blockIdx.x : nullptr);
    wg_trace(g, blockIdx.x, t0, q, bz);
}

template <int KW>
__global__ __launch_bounds__(256 * KW) void gemm_f32_async_grouped_aux_kernel(const GroupedParams g, const AuxJobs aux) {
    const long long t0 = g.trace ? wall_clock64() : 0;
    const int nb = g.first[g.count];
    if ((int)blockIdx.x >= nb) {   // behind the tiles: the small reduction jobs (aux_jobs.hpp)
        aux_job_run(aux, (int)blockIdx.x - nb, as_ring);
        wg_trace(g, blockIdx.x, t0, 100 + (((int)blockIdx.x - nb) < aux.cs_first[aux.n_colsum] ? 0 : 1), 0);
        return;
    }
    if (g.warm) warm_kernargs((int)sizeof(GroupedParams));
    int q, bx, by, bz;
    if (!group_decode(g, blockIdx.x, q, bx, by, bz)) return;
    gemm_tile_async<true, true, KW>(g.p[q], bx, by, bz, g.gz[q], as_ring, nullptr, g.trace ? g.trace + 8 * blockIdx.x : nullptr);
    wg_trace(g, blockIdx.x, t0, q, bz);
}

// The ring is dynamic LDS (ST x 16 KB = 64 KB: two workgroups per CU).
template <typename K>
static int launch_dyn(K kernel, dim3 grid, int threads, size_t lds, hipStream_t st, const void* arg,
                      const void* arg2 = nullptr, const void* arg3 = nullptr, const void* arg4 = nullptr) {
    static thread_local const void* configured[64];
    static thread_local int nconf = 0;
    bool seen = false;
    for (int i = 0; i < nconf; ++i) seen = seen || configured[i] == (const void*)kernel;
    if (!seen) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("pp_gemm_f32: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        if (nconf < 64) configured[nconf++] = (const void*)kernel;
    }
    void* args[4] = {const_cast<void*>(arg), const_cast<void*>(arg2), const_cast<void*>(arg3), const_cast<void*>(arg4)};
    hipError_t e = hipLaunchKernel((const void*)kernel, grid, dim3(threads), args, lds, st);
    if (e != hipSuccess) {
        set_error("pp_gemm_f32 (async): launch failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

static size_t as_lds_bytes() { return (size_t)AS_STAGES * AS_STAGE * sizeof(float) + 2 * AS_KIDX * sizeof(int32_t); }

template <int KW>
static int launch_async_kw(const GemmParams& p, bool akm, bool bkm, hipStream_t st) {
    dim3 grid(cdiv(p.N, 64), cdiv(p.M, 64), 1);
    const size_t lds = as_lds_bytes();
    if (!akm && !bkm) return launch_dyn(gemm_f32_async_kernel<false, false, KW>, grid, 256 * KW, lds, st, &p);
    if (!akm && bkm) return launch_dyn(gemm_f32_async_kernel<false, true, KW>, grid, 256 * KW, lds, st, &p);
    if (akm && !bkm) return launch_dyn(gemm_f32_async_kernel<true, false, KW>, grid, 256 * KW, lds, st, &p);
    return launch_dyn(gemm_f32_async_kernel<true, true, KW>, grid, 256 * KW, lds, st, &p);
}

template <int KW>
static int launch_async_grouped_kw(const GroupedParams& g, bool akm, bool bkm, hipStream_t st) {
    dim3 grid(g.first[g.count]);
    const size_t lds = as_lds_bytes();
    if (!akm && !bkm) return launch_dyn(gemm_f32_async_grouped_kernel<false, false, KW>, grid, 256 * KW, lds, st, &g);
    if (!akm && bkm) return launch_dyn(gemm_f32_async_grouped_kernel<false, true, KW>, grid, 256 * KW, lds, st, &g);
    if (akm && !bkm) return launch_dyn(gemm_f32_async_grouped_kernel<true, false, KW>, grid, 256 * KW, lds, st, &g);
    return launch_dyn(gemm_f32_async_grouped_kernel<true, true, KW>, grid, 256 * KW, lds, st, &g);
}

// the group's tiles + the workgroups of the small reduction jobs behind them (both operands k-major only)
static int launch_async_grouped_aux(const GroupedParams& g, const AuxJobs& aux, bool kw2, hipStream_t st) {
    dim3 grid(g.first[g.count] + aux.n_blocks);
    const size_t lds = as_lds_bytes();
    if (kw2) return launch_dyn(gemm_f32_async_grouped_aux_kernel<2>, grid, 512, lds, st, &g, &aux);
    return launch_dyn(gemm_f32_async_grouped_aux_kernel<1>, grid, 256, lds, st, &g, &aux);
}
static int launch_direct_grouped_aux(const GroupedParams& g, const AuxJobs& aux, hipStream_t st) {
    dim3 grid(g.first[g.count] + aux.n_blocks), block(256);
    hipLaunchKernelGGL(gemm_f32_direct_grouped_aux_kernel, grid, block, 0, st, g, aux);
    PP_LAUNCH_CHECK("pp_gemm_f32_grouped (direct + jobs)");
    return 0;
}

// eight waves per tile when the launch cannot put four workgroups on every CU anyway
static bool eight_waves(int64_t blocks) {
    static const int mode = 0;
    if (mode == 1) return false;
    if (mode == 2) return true;
    return blocks <= 512;
}

static int launch_async(const GemmParams& p, bool akm, bool bkm, hipStream_t st) {
    const int64_t blocks = (int64_t)cdiv(p.N, 64) * cdiv(p.M, 64);
    return eight_waves(blocks) ? launch_async_kw<2>(p, akm, bkm, st) : launch_async_kw<1>(p, akm, bkm, st);
}

static int launch_async_grouped(const GroupedParams& g, bool akm, bool bkm, hipStream_t st, int64_t active_blocks = -1) {
    return eight_waves(active_blocks >= 0 ? active_blocks : g.first[g.count]) ? launch_async_grouped_kw<2>(g, akm, bkm, st)
                                                                              : launch_async_grouped_kw<1>(g, akm, bkm, st);
}

template <int BM, int BN, int WM, int WN, int VEC>
static int launch_layout(const GemmParams& p, bool akm, bool bkm, int splits, hipStream_t st) {
    dim3 grid(cdiv(p.N, BN), cdiv(p.M, BM), splits);
    dim3 block(256);
    if (!akm && !bkm) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, false, VEC>), grid, block, 0, st, p);
    else if (!akm && bkm) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, true, VEC>), grid, block, 0, st, p);
    else if (akm && !bkm) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, false, VEC>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, true, VEC>), grid, block, 0, st, p);
    PP_LAUNCH_CHECK("pp_gemm_f32");
    return 0;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// element ranges -> the slab interval the tiles may skip: only whole slabs inside [k0, k1) (k1 >= K: up to the last slab)
static void clear_holes(GemmParams& p) {
    for (int q = 0; q < 2; ++q) p.hm0[q] = p.hm1[q] = p.hn0[q] = p.hn1[q] = p.hs0[q] = p.hs1[q] = 0;
}
static void set_hole(GemmParams& p, const GemmHole* h) {
    clear_holes(p);
    static const int enabled = getenv("PP_GEMM_HOLES") ? atoi(getenv("PP_GEMM_HOLES")) : 1;
    if (!h || !enabled) return;
    for (int q = 0; q < 2; ++q) {
        const GemmBlock& b = h->b[q];
        if (b.k1 <= b.k0 || (q == 1 && enabled == 2)) continue;      // (PP_GEMM_HOLES=2: first block only, A/B measurements)
        p.hm0[q] = b.m0; p.hm1[q] = b.m1; p.hn0[q] = b.n0; p.hn1[q] = b.n1;
        p.hs0[q] = cdiv(b.k0, BK);
        p.hs1[q] = b.k1 >= p.K ? cdiv(p.K, BK) : b.k1 / BK;
    }
}

static void fill_params(const pp_gemm_args* a, GemmParams& p) {
    p.A = a->A; p.lda = a->lda; p.a_idx = a->a_idx;
    p.B = a->B; p.ldb = a->ldb; p.b_idx = a->b_idx;
    p.C = a->C; p.ldc = a->ldc; p.c_idx = a->c_idx;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.bias = a->bias; p.bias2 = a->bias2; p.mask = a->mask; p.ldmask = a->ldmask;
    p.relu = a->relu; p.accumulate = a->accumulate;
    p.colsum = a->colsum;
    static const int plain = 0;
    p.dbg_plain = plain;
    p.split_stride = 0;
    clear_holes(p);
    p.vec = 1;
    p.vec_c = (a->ldc % 4 == 0 && aligned16(a->C)) ? 1 : 0;
    static const int stamp_all = getenv("PP_DBG_STAMP") ? 1 : 0;   // debug: stamp every product (tools/kernel_timeline.py)
    p.dbg = (stamp_all || (a->M == 1024 && a->N == 2048)) ? g_timeline : nullptr;   // debug: stamp the forward input GEMM only
}

static bool vec_ok(const pp_gemm_args* a) {
    return (a->lda % 4 == 0) && (a->ldb % 4 == 0) && aligned16(a->A) && aligned16(a->B);
}

// The async tile needs 16-byte pieces and cannot gather along k (the k-major row pointer would change every slab).
static bool async_ok(const pp_gemm_args* a) {
    static const int mode = 1;
    return mode && vec_ok(a) && a->K >= 1;
}
// a k-major operand that gathers along k stages its index list in LDS: the workgroup's K range must fit
static bool async_split_ok(const pp_gemm_args* a, int splits) {
    static const int allow = 1;
    const bool kgather = (a->a_kmajor && a->a_idx) || (a->b_kmajor && a->b_idx);
    if (kgather && !allow) return false;
    return !kgather || cdiv(cdiv(a->K, BK), splits) <= AS_KSLABS;
}

// Split-K: the weight-gradient products have K = rows of the batch and only a handful of output tiles; one
// workgroup per tile would walk dozens of slabs serially while most CUs idle. Spread the slabs over
// ~2 workgroups per CU and combine with float atomics (the gradient buffers are zero-initialised accumulators).
// Only for linear epilogues into a dense or pre-zeroed destination, and only when the caller opts in (split_k).
// Which tile code runs a product (or a group) of `tiles64` 64x64 tiles: the wave-direct 32x32 tiles when the LDS-staged
// kernel could not put ~2 workgroups on every CU and the K range is short (`slabs` = sum over tiles of their K slabs).
// PP_GEMM_DIRECT=0/1 forces one of them (A/B measurements).
static bool use_direct(int64_t tiles64, int64_t slabs) {
    static const int mode = -1;
    // (128: the per-address head weight gradients of a ragged 12-address step - 24 problems, ~300 tiles - run 25 us faster
    // on the async tiles than on 2500 direct workgroups; the single products that need the direct tile stay below)
    static const int limit = 128;
    static const int kmax = 16;
    if (mode == 0) return false;
    if (mode == 1) return tiles64 < 4096;
    // long-K products would need cross-workgroup split-K to occupy the chip; their float atomics cost more than the
    // shorter chain saves (measured: dX 1024x212x2048 26.6 us staged / split 8, 28-41 us direct)
    return tiles64 < limit && slabs <= kmax * tiles64;
}

static bool split_allowed(const pp_gemm_args* a) {
    if (deterministic_mode()) return false;   // no float atomics: one workgroup per tile walks all of K
    const int64_t tiles64 = (int64_t)cdiv(a->M, 64) * cdiv(a->N, 64);
    const bool linear = !a->relu && !a->mask && !a->colsum;
    return a->split_k && linear && cdiv(a->K, BK) >= 4 && tiles64 < 384 && (a->accumulate || !a->c_idx);
}

static int pick_splits(const pp_gemm_args* a, int64_t budget_blocks) {
    const int64_t tiles64 = (int64_t)cdiv(a->M, 64) * cdiv(a->N, 64);
    const int nslab = cdiv(a->K, BK);
    if (!split_allowed(a)) return 1;
    int splits = (int)std::min<int64_t>(std::min<int64_t>((budget_blocks + tiles64 - 1) / tiles64, nslab / 2), 32);
    return std::max(splits, 1);
}

// Grouped launch: every workgroup should walk about the same number of K slabs, whichever problem it belongs to
// (an equal per-problem block budget left the largest product un-split and 4x slower than the rest of the group).
static int pick_splits_by_work(const pp_gemm_args* a, int slabs_per_block) {
    const int nslab = cdiv(a->K, BK);
    if (!split_allowed(a)) return 1;
    return std::max(1, std::min(std::min(cdiv(nslab, slabs_per_block), nslab / 2), 32));
}

static int zero_for_split(const pp_gemm_args* a, hipStream_t st) {   // partial tiles are added atomically: start from zero
    hipError_t e = hipMemset2DAsync(a->C, (size_t)a->ldc * sizeof(float), 0, (size_t)a->N * sizeof(float), a->M, st);
    if (e != hipSuccess) {
        set_error("pp_gemm_f32: hipMemset2DAsync failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

template <int VEC>
static int launch_grouped(const GroupedParams& g, bool akm, bool bkm, hipStream_t st) {
    dim3 grid(g.first[g.count]), block(256);
    if (!akm && !bkm) hipLaunchKernelGGL((gemm_f32_grouped_kernel<64, 64, 32, 32, false, false, VEC>), grid, block, 0, st, g);
    else if (!akm && bkm) hipLaunchKernelGGL((gemm_f32_grouped_kernel<64, 64, 32, 32, false, true, VEC>), grid, block, 0, st, g);
    else if (akm && !bkm) hipLaunchKernelGGL((gemm_f32_grouped_kernel<64, 64, 32, 32, true, false, VEC>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_grouped_kernel<64, 64, 32, 32, true, true, VEC>), grid, block, 0, st, g);
    PP_LAUNCH_CHECK("pp_gemm_f32_grouped");
    return 0;
}

// A split product as a one-problem group: the grouped kernels own the XCD-aware workgroup -> (tile, split) mapping.
static int launch_split(const GemmParams& p, bool vec, bool akm, bool bkm, int tile, int splits, int kind, hipStream_t st) {
    static const int xcd = 1;
    GroupedParams g;
    g.ext = GemmExt{};
    g.trace = nullptr;
    g.warm = 1;
    g.xcd_aware = xcd;
    g.count = 1;
    g.p[0] = p;
    g.p[0].vec = vec ? 1 : 0;
    g.gx[0] = cdiv(p.N, tile); g.gy[0] = cdiv(p.M, tile); g.gz[0] = splits;
    g.pmode[0] = pick_pmode(p.M, p.N, g.gx[0], g.gy[0]);
    g.rot[0] = 0;
    g.first[0] = 0;
    g.first[1] = group_blocks(g.gx[0], g.gy[0], splits, g.pmode[0]);
    if (g_wgtrace && ((g_wgtrace_mode == 1 && akm && bkm) || (g_wgtrace_mode == 2 && !akm && bkm)) && g.first[1] <= g_wgtrace_cap)
        g.trace = g_wgtrace;
    return kind == 1 ? launch_direct_grouped(g, akm, bkm, st)
         : kind == 2 ? launch_async_grouped(g, akm, bkm, st) : launch_grouped<4>(g, akm, bkm, st);
}

int gemm_f32(const pp_gemm_args* a, hipStream_t st, const GemmHole* hole, const GemmExt* ext) {
    PP_CHECK_ARG(a && a->A && a->B && a->C, "pp_gemm_f32: null operand");
    PP_CHECK_ARG(a->M >= 0 && a->N >= 0 && a->K >= 0, "pp_gemm_f32: negative dimension");
    if (a->M == 0 || a->N == 0) return 0;
    GemmParams p;
    fill_params(a, p);
    set_hole(p, hole);
    const bool vec = vec_ok(a);
    if (ext && (ext->rb || ext->cell_cprev)) {   // the LSTM input / recurrent product: epilogue in the async 64x64 tile only
        PP_CHECK_ARG((ext->rb_addr || !ext->rb_prev) && vec && a->K >= 1 && !a->a_kmajor && !a->b_kmajor && !a->a_idx && !a->b_idx && !a->c_idx &&
                         !a->mask && !a->colsum && !a->accumulate && !a->relu && !a->bias && !a->bias2,
                     "pp_gemm_f32: unsupported LSTM input product");
        PP_CHECK_ARG(!ext->cell_cprev || (!ext->rb && ext->cell_H > 0 && ext->cell_rows >= a->M && !ext->lean),
                     "pp_gemm_f32: bad recurrent-product arguments");
        PP_CHECK_ARG(ext->cell_H == 0 || (ext->cell_H % 16 == 0 && a->N == 4 * ext->cell_H && a->ldc % 4 == 0 && aligned16(a->C) &&
                                          (ext->cell_c || ext->lean) && ext->cell_h),
                     "pp_gemm_f32: bad fused-cell arguments");
        const int gx = cdiv(a->N, 64), gy = cdiv(a->M, 64);
        // (measured on the ragged step, 1 344 tiles: 1, 2 or 4 column tiles per workgroup give the same 28 us - the tile,
        // not the workgroup start, is what costs)
        int tpw = 1;
        dim3 grid(cdiv(gx, tpw), gy, 1);
        return launch_dyn(gemm_f32_async_lstm_kernel, grid, 256, as_lds_bytes(), st, &p, ext, &tpw, &gx);
    }
    if (ext && ext->split_stride > 0) {   // K splits that STORE their partial tiles (async tile; the consumer adds them)
        PP_CHECK_ARG(ext->force_splits >= 1 && ext->force_splits <= 32 && async_ok(a) && async_split_ok(a, ext->force_splits) &&
                         !a->c_idx && !a->relu && !a->mask && !a->colsum && !a->bias && !a->bias2 && !a->accumulate,
                     "pp_gemm_f32: unsupported split-store product");
        p.split_stride = ext->force_splits > 1 ? ext->split_stride : 0;
        return launch_split(p, vec, a->a_kmajor, a->b_kmajor, 64, ext->force_splits, 2, st);
    }
    if (gemv_ok(a)) return launch_gemv(p, vec, st);
    // Tile choice: the hot-path GEMMs are small (<= a few thousand rows); 64x64 tiles give >= 2 workgroups per CU
    // on the 1024x2048x212 input GEMM. Very tall problems (batched IS) use 128x128 tiles.
    const int64_t tiles64 = (int64_t)cdiv(a->M, 64) * cdiv(a->N, 64);
    const bool big = tiles64 >= 4096 && a->N >= 128;
    if (use_direct(tiles64, tiles64 * cdiv(a->K, BK))) {
        // 32x32 tiles, 4 waves share the K slabs: split further across workgroups only when the tiles alone do not
        // give every CU a few workgroups
        const int64_t tiles32 = (int64_t)cdiv(a->M, DT) * cdiv(a->N, DT);
        const int nslab = cdiv(a->K, BK);
        int splits = 1;
        static const int maxsplit = 16;
        if (split_allowed(a)) splits = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(cdiv(1024, tiles32), nslab / 8), maxsplit));
        if (splits > 1 && !a->accumulate) PP_TRY(zero_for_split(a, st));
        if (splits > 1) return launch_split(p, vec, a->a_kmajor, a->b_kmajor, DT, splits, 1, st);
        return vec ? launch_direct<4>(p, a->a_kmajor, a->b_kmajor, splits, st)
                   : launch_direct<1>(p, a->a_kmajor, a->b_kmajor, splits, st);
    }
    static const int budget = 256;
    static const int force = 0;
    const int splits = big ? 1 : (force && split_allowed(a) ? force : pick_splits(a, budget));
    if (splits > 1 && !a->accumulate) PP_TRY(zero_for_split(a, st));
    const bool as = !big && async_ok(a) && async_split_ok(a, splits);
    if (splits > 1) return launch_split(p, vec, a->a_kmajor, a->b_kmajor, 64, splits, as ? 2 : 0, st);
    if (as) return launch_async(p, a->a_kmajor, a->b_kmajor, st);
    if (big) {
        return vec ? launch_layout<128, 128, 64, 64, 4>(p, a->a_kmajor, a->b_kmajor, 1, st)
                   : launch_layout<128, 128, 64, 64, 1>(p, a->a_kmajor, a->b_kmajor, 1, st);
    }
    return vec ? launch_layout<64, 64, 32, 32, 4>(p, a->a_kmajor, a->b_kmajor, splits, st)
               : launch_layout<64, 64, 32, 32, 1>(p, a->a_kmajor, a->b_kmajor, splits, st);
}

// 64 x 64 tile-slabs a product really walks once its zero blocks are left out (what the launch's work is split by)
static int64_t effective_work(const pp_gemm_args* a, const GemmHole* h, int* active_tiles = nullptr) {
    const int gx = cdiv(a->N, 64), gy = cdiv(a->M, 64), nslab = cdiv(a->K, BK);
    int64_t work = (int64_t)gx * gy * nslab;
    if (active_tiles) *active_tiles = gx * gy;
    static const int enabled = getenv("PP_GEMM_HOLES") ? atoi(getenv("PP_GEMM_HOLES")) : 1;
    if (!h || !enabled || (a->a_kmajor && a->a_idx) || (a->b_kmajor && a->b_idx)) return work;
    for (int by = 0; by < gy; ++by)
        for (int bx = 0; bx < gx; ++bx) {
            int best = 0;
            for (int q = 0; q < 2; ++q) {
                const GemmBlock& b = h->b[q];
                if (b.k1 <= b.k0) continue;
                const int m0 = by * 64, n0 = bx * 64;
                if (m0 < b.m0 || std::min(m0 + 64, a->M) > b.m1 || n0 < b.n0 || std::min(n0 + 64, a->N) > b.n1) continue;
                const int s0 = cdiv(b.k0, BK), s1 = b.k1 >= a->K ? nslab : b.k1 / BK;
                best = std::max(best, std::min(s1, nslab) - s0);
            }
            work -= best;
            if (best >= nslab && active_tiles) --*active_tiles;
        }
    return work;
}

// `count` independent products with the same operand layouts in as few launches as possible (GROUP_MAX per launch).
int gemm_f32_grouped(const pp_gemm_args* args, int count, hipStream_t st, const GemmHole* holes, const GemmExt* ext,
                     const AuxJobs* aux) {
    PP_CHECK_ARG(count >= 0 && (count == 0 || args), "pp_gemm_f32_grouped: bad argument");
    int i = 0;
    bool aux_pending = aux && aux->n_blocks > 0;
    // kernel arguments beyond 4 KB (GroupedParams + AuxJobs): if the runtime refuses the launch once, the jobs run as
    // their own launch from then on
    static bool aux_fused_ok = !(getenv("PP_AUX_FUSED") && atoi(getenv("PP_AUX_FUSED")) == 0);
    while (i < count) {
        GroupedParams g;
        g.ext = ext ? *ext : GemmExt{};
        g.trace = nullptr;
        g.warm = 1;
        static const int xcd = 1;
        g.xcd_aware = xcd;
        g.count = 0;
        g.first[0] = 0;
        const int akm = args[i].a_kmajor, bkm = args[i].b_kmajor;
        // slabs each workgroup walks so that the launch has ~target workgroups
        static const int target_staged = 768;
        // (with the zero blocks left out of the work estimate: 128-256 active workgroups measured equal, 0.152 ms per GUM
        // step; 320+ costs 4 us in split-K atomics, one split per tile 3 us in idle CUs - profiles/r02_e/f_ab_*.json)
        static const int target_async = 256;
        static const int target_direct = 1536;
        int64_t work = 0, work32 = 0, tiles = 0;
        bool as = true;
        for (int k = i, c = 0; k < count && c < GROUP_MAX; ++k) {
            const pp_gemm_args* a = &args[k];
            if (a->a_kmajor != akm || a->b_kmajor != bkm) break;
            if (a->M <= 0 || a->N <= 0) continue;
            static const int trace = 0;
            if (trace && !async_ok(a))
                fprintf(stderr, "[pp_gemm] group problem not async: M=%d N=%d K=%d lda=%lld ldb=%lld A%%16=%d B%%16=%d\n", a->M, a->N,
                        a->K, (long long)a->lda, (long long)a->ldb, (int)((uintptr_t)a->A & 15), (int)((uintptr_t)a->B & 15));
            as = as && async_ok(a);
            tiles += (int64_t)cdiv(a->M, 64) * cdiv(a->N, 64);
            static const int effw = 1;
            work += effw ? effective_work(a, holes ? &holes[k] : nullptr)
                         : (int64_t)cdiv(a->M, 64) * cdiv(a->N, 64) * cdiv(a->K, BK);
            work32 += (int64_t)cdiv(a->M, DT) * cdiv(a->N, DT) * cdiv(a->K, BK);
            ++c;
        }
        // (the cell-backward epilogue exists in the direct tile only)
        const bool direct = (ext && ext->bw_G) ? true : use_direct(tiles, work);
        // the async tiles stream a long K range at full rate, so they want fewer, longer workgroups (fewer atomics)
        const int target = as ? target_async : target_staged;
        // direct tiles: a workgroup's four waves share its slabs, so it should own >= 8 of them
        // ... but no workgroup should walk more than ~24 slabs (~10 us of K loop): a ragged batch's dW_ih / dW_hh have
        // K = 1600-2600 rows, and 82 slabs per workgroup made the weight-gradient launches of the GUMM step 98 + 89 us
        static const int spb_max = 24;
        const int spb = direct ? (int)std::max<int64_t>(8, (work32 + target_direct - 1) / target_direct)
                               : (int)std::min<int64_t>(spb_max, std::max<int64_t>(2, (work + target - 1) / target));
        const int tile = direct ? DT : 64;
        int64_t active_blocks = 0;   // workgroups that have slabs to walk (tiles inside a zero block exit at once)
        int xcd_cursor = 0;          // next XCD of the round robin over the tiles of the round-robin-placed problems
        int j = i;
        for (; j < count && g.count < GROUP_MAX; ++j) {
            const pp_gemm_args* a = &args[j];
            PP_CHECK_ARG(a->A && a->B && a->C && a->M >= 0 && a->N >= 0 && a->K >= 0, "pp_gemm_f32_grouped: bad problem");
            if (a->a_kmajor != akm || a->b_kmajor != bkm) break;   // next launch
            if (a->M == 0 || a->N == 0) continue;
            const int q = g.count++;
            fill_params(a, g.p[q]);
            set_hole(g.p[q], holes ? &holes[j] : nullptr);
            g.p[q].vec = vec_ok(a) ? 1 : 0;
            int splits = direct ? (split_allowed(a) ? std::max(1, std::min(cdiv(cdiv(a->K, BK), spb), 32)) : 1)
                                : pick_splits_by_work(a, spb);
            // (One- and two-tile problems of an async group - the observe-embedding leaves, whose operands were written by the
            // kernel just before: first slab 7 us after the start, then a dozen slabs at 0.9 us - are the last workgroups of
            // the weight-gradient launch to finish, tools/wg_trace.py. Giving them 3x / 6x the splits made the launch SLOWER,
            // 24.5 -> 26.7 / 29 us: the launch is bound by its total number of float atomics, not by its longest workgroup.
            // PP_GROUP_SMALL_DIV > 1 re-enables the experiment.)
            static const int small_div = 1;
            if (!direct && as && small_div > 1 && split_allowed(a) && (int64_t)cdiv(a->M, 64) * cdiv(a->N, 64) <= 2)
                splits = std::max(splits, pick_splits_by_work(a, std::max(2, spb / small_div)));
            // a gathered k range must fit the LDS index list of the async tile
            if (as && !direct && split_allowed(a) && !async_split_ok(a, splits))
                splits = std::min(cdiv(cdiv(a->K, BK), AS_KSLABS), 32);
            if (splits > 1 && !a->accumulate) PP_TRY(zero_for_split(a, st));
            as = as && async_split_ok(a, splits);
            g.gx[q] = cdiv(a->N, tile); g.gy[q] = cdiv(a->M, tile); g.gz[q] = splits;
            {
                int act = g.gx[q] * g.gy[q];
                if (tile == 64) (void)effective_work(a, holes ? &holes[j] : nullptr, &act);
                active_blocks += (int64_t)act * splits;
            }
            g.pmode[q] = pick_pmode(a->M, a->N, g.gx[q], g.gy[q]);
            static const int rotate = 1;
            g.rot[q] = (g.pmode[q] == 0 && rotate) ? (xcd_cursor & 7) : 0;
            if (g.pmode[q] == 0) xcd_cursor += g.gx[q] * g.gy[q];
            g.first[q + 1] = g.first[q] + group_blocks(g.gx[q], g.gy[q], splits, g.pmode[q]);
        }
        if (g_wgtrace && ((g_wgtrace_mode == 1 && akm && bkm) || (g_wgtrace_mode == 2 && !akm && bkm)) &&
            g.first[g.count] + (aux_pending ? aux->n_blocks : 0) <= g_wgtrace_cap)
            g.trace = g_wgtrace;
        bool rode = false;
        if (g.count > 0 && aux_pending && aux_fused_ok && akm && bkm && (direct || as)) {
            const int rc = direct ? launch_direct_grouped_aux(g, *aux, st)
                                  : launch_async_grouped_aux(g, *aux, eight_waves(active_blocks), st);
            if (rc == 0) {
                rode = true;
                aux_pending = false;
            } else {
                (void)hipGetLastError();
                aux_fused_ok = false;
            }
        }
        if (rode) { i = j; continue; }
        if (g.count > 0 && direct) PP_TRY(launch_direct_grouped(g, akm, bkm, st));
        else if (g.count > 0 && as) PP_TRY(launch_async_grouped(g, akm, bkm, st, active_blocks));
        else
        if (g.count > 0) PP_TRY(launch_grouped<4>(g, akm, bkm, st));
        i = j;
    }
    if (aux_pending) PP_TRY(aux_jobs_launch(*aux, st));
    return 0;
}

}  // namespace pp

// debug: per-workgroup start / end stamps of the grouped async launches (mode 1: weight-gradient groups, 2: data-gradient
// products); buf = device int64 [8 * cap] or NULL
extern "C" int pp_debug_wgtrace(long long* buf, int32_t cap, int32_t mode) {
    pp::g_wgtrace = buf;
    pp::g_wgtrace_cap = cap;
    pp::g_wgtrace_mode = mode;
    return 0;
}

extern "C" int pp_gemm_f32(const pp_gemm_args* args, void* stream) { return pp::gemm_f32(args, pp::as_stream(stream), nullptr, nullptr); }
extern "C" int pp_gemm_f32_grouped(const pp_gemm_args* args, int32_t count, void* stream) {
    return pp::gemm_f32_grouped(args, count, pp::as_stream(stream), nullptr, nullptr, nullptr);
}
