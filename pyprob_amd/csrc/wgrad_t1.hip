// Weight gradients of the backward pass in one launch:
//   dW_ih = dG^T X,  dW_hh = dG_t^T h_{t-1},  dW1 = dz1^T h,  dW2 = dy^T a1,  the observe-embedding's dWf1 = dE^T f1, dWf0 = dF1^T cat,
//   dW1_o = dCat_o^T h_o
// (inference_network_lstm.py:186-220 backward: torch.autograd's weight gradients of nn.LSTM / nn.Linear), every product
// with K = rows of the minibatch and both operands stored row = k by the kernels before it (the second one optionally
// through a row gather: h_{t-1} by prev_row, the rows of an address group).
//
// Why not the grouped tile kernels (gemm_f32.hip): those stage K slabs through an LDS ring, three slabs ahead - a workgroup
// streams ~45 GB/s, so 1 024 rows cost ~13 us per tile however small the problem (measured: the four one-tile embedding
// problems ALONE take the launch 13 us, profiles/r03g_wgrad_composition.txt). Here both operands are k-major, which is the
// MFMA operand layout of v_mfma_f32_32x32x2_f32 as it lies in memory (lane l supplies A[k0 + l / 32][m0 + l % 32]): the
// fragments are plain coalesced dword loads into a register ring, eight row pairs ahead of the MFMAs - no LDS, no barrier
// in the K loop. A workgroup (8 waves) owns a 64 x 64 output tile over one of S row ranges; every wave multiplies the whole
// tile over its own eighth of the rows; the waves meet in LDS, the S ranges in float atomics on dW.
// Single-statement batches (behind the panel kernel): 23.2 -> 15.2 us. Ragged batches (K up to 2 600 rows, 33 problems):
// one launch instead of two, step 0.555 -> 0.519 ms; what bounds it there: DESIGN.md 8.2 (the float atomics).
//
// The reduction jobs of the backward pass (aux_jobs.hpp: column sums, table-column gradients, bias gradients, the loss) ride
// behind the tiles as before.
#include "wgrad_t1.hpp"

#include <stdlib.h>

#include <algorithm>

namespace pp {

extern long long* g_wgtrace;   // gemm_f32.hip (pp_debug_wgtrace)
extern int g_wgtrace_cap, g_wgtrace_mode;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_RING = 8;    // row pairs in the ring: nine in flight (36 loads per wave) while one multiplies. The operands
                              // were written by the previous kernel on other XCDs - every load is a ~2 us trip to the
                              // Infinity Cache - and a CU needs ~100 KB in flight to stream at its L1 fill rate

// two rows (k + lane / 32) of the tile's 64 A columns and 64 B columns: the operands of four MFMAs
struct Pair {
    float a0, a1, b0, b1;
};

// K loop of one wave: the whole 64 x 64 tile over rows [kb, ke).
template <bool GATHER>
__device__ __forceinline__ void wgrad_kloop(f32x16 (&acc)[2][2], const float* __restrict__ A, const float* __restrict__ B,
                                            const int32_t* __restrict__ bidx, const int lda, const int ldb, const int M, const int N,
                                            const int m0, const int n0, const int kb, const int ke, const int lane,
                                            const bool hm, const bool hn) {
    const int l31 = lane & 31, h = lane >> 5;
    // columns beyond M / N are clamped: their products land in accumulator rows / columns the epilogue never stores.
    // Rows beyond the range are clamped too and enter as a * 0 (a multiply, not a select: a select lets the compiler
    // predicate the LOAD, one branch per load)
    const float* pa0 = A + min(m0 + l31, M - 1);
    const float* pa1 = A + min(m0 + 32 + l31, M - 1);
    const float* pb0 = B + min(n0 + l31, N - 1);
    const float* pb1 = B + min(n0 + 32 + l31, N - 1);
    // Row gather on the second operand: an index loaded next to its row would have to be WAITED for, and loads return in
    // order - s_waitcnt on the youngest load drains the whole ring, every pair (measured: the ragged step's launch ran its
    // MFMAs a third of the time). So a lane keeps the indices of 64 rows (one coalesced load per 32 pairs, issued a block
    // ahead: by the time it is read more than 63 younger loads have been issued) and a pair takes its two with v_readlane.
    int idxA = 0, idxB = 0;
    if (GATHER) idxA = bidx[min(kb + lane, ke - 1)];
    auto load = [&](int pidx, Pair& p) {      // row pair pidx of this wave: rows kb + 2 pidx + h
        const int k = kb + 2 * pidx + h;
        const int kc = min(k, ke - 1);
        const float keep = k < ke ? 1.0f : 0.0f;
        p.a0 = pa0[(int64_t)kc * lda] * keep;
        p.a1 = pa1[(int64_t)kc * lda] * keep;
        int kb2 = kc;
        if (GATHER) {
            if ((pidx & 31) == 0) {      // first pair of a 64-row block (wave-uniform): fetch the NEXT block's indices
                const int nb = (pidx >> 5) + 1;
                const int v = bidx[min(kb + 64 * nb + lane, ke - 1)];
                idxB = (nb & 1) ? v : idxB;
                idxA = (nb & 1) ? idxA : v;
            }
            const int v = ((pidx >> 5) & 1) ? idxB : idxA;
            const int sel = 2 * (pidx & 31);
            const int r0 = __builtin_amdgcn_readlane(v, sel), r1 = __builtin_amdgcn_readlane(v, sel + 1);
            kb2 = h ? r1 : r0;
        }
        p.b0 = pb0[(int64_t)kb2 * ldb];
        p.b1 = pb1[(int64_t)kb2 * ldb];
    };
    auto mma = [&](const Pair& p) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.a0, p.b0, acc[0][0], 0, 0, 0);
        if (hn) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.a0, p.b1, acc[0][1], 0, 0, 0);
        if (hm) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.a1, p.b0, acc[1][0], 0, 0, 0);
        if (hm && hn) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.a1, p.b1, acc[1][1], 0, 0, 0);
    };
    Pair f[WG_RING];
    const int nfr = (ke - kb + 1) >> 1;      // row pairs of this wave
    // (loads are unconditional so that the compiler counts them with s_waitcnt vmcnt(N) instead of draining the queue at
    // a branch)
#pragma unroll
    for (int i = 0; i < WG_RING - 1; ++i) load(i, f[i]);
    int i0 = 0;
    for (; i0 + WG_RING <= nfr; i0 += WG_RING) {
#pragma unroll
        for (int j = 0; j < WG_RING; ++j) {
            load(i0 + j + WG_RING - 1, f[(j + WG_RING - 1) % WG_RING]);
            mma(f[j]);
        }
    }
    const int nrem = nfr - i0;      // < WG_RING: already in the ring
#pragma unroll
    for (int j = 0; j < WG_RING - 1; ++j)
        if (j < nrem) mma(f[j]);
}

__global__ __launch_bounds__(512, 4) void wgrad_t1_kernel(const WgradT1Args g, const AuxJobs aux) {
    __shared__ float lds[8192];      // [8 waves][16][64] partial quadrants; the reduction jobs use 2 048 floats
    const long long t_start = g.trace ? wall_clock64() : 0;
    // The reduction jobs ride in front of the tiles when the tiles need more than one round of the chip's workgroup slots
    // (ragged batches: they would otherwise start when the last tile has been dispatched and BE the tail of the launch,
    // ~15 us, tools/wg_trace_wgrad.py), else behind them (one round: the tiles are the long workgroups and start at once).
    const int aux_lo = g.aux_first ? 0 : g.n_blocks;
    const int bx = (int)blockIdx.x - (g.aux_first ? aux.n_blocks : 0);
    if ((int)blockIdx.x >= aux_lo && (int)blockIdx.x < aux_lo + aux.n_blocks) {
        aux_job_run(aux, (int)blockIdx.x - aux_lo, lds);
        if (g.trace && threadIdx.x == 0) {
            long long* tr = g.trace + 8 * blockIdx.x;
            tr[0] = t_start; tr[1] = wall_clock64(); tr[2] = 100; tr[3] = 0;
        }
        return;
    }
    int pi = 0;
#pragma unroll
    for (int q = 1; q < WGRAD_T1_MAX; ++q)
        if (q < g.n_prob && bx >= g.p[q].first) pi = q;
    // (scalar loads with a run-time index: one trip to the kernel arguments)
    const float* const A = g.p[pi].A;
    const float* const B = g.p[pi].B;
    float* const C = g.p[pi].C;
    const int32_t* const bidx = g.p[pi].bidx;
    const int lda = g.p[pi].lda, ldb = g.p[pi].ldb, ldc = g.p[pi].ldc, M = g.p[pi].M, N = g.p[pi].N, nt = g.p[pi].nt;
    const int K = g.p[pi].K, ks = g.p[pi].ks;
    const int local = bx - g.p[pi].first;
    const int mtg = g.p[pi].mtg;
    int s, tm, tn;
    if (mtg > 0) {
        // XCD-aware order (workgroups go to the 8 XCDs round-robin by block index, each XCD has its own L2): block r (mod 8)
        // of the problem owns the row tiles tm = r (mod 8) - all their column tiles and row splits. An XCD then fetches ITS
        // M / 8 columns of A once and all of B; with the column tile fastest (below) every XCD owned one column tile of B and
        // streamed ALL of A - the larger operand (dG: 4H columns) - from the Infinity Cache: 8 x |A| + |B| instead of
        // |A| + 8 x |B| per launch. Row tiles past the end of a ragged last group are empty workgroups.
        const int r = local & 7, u = local >> 3;
        const int tmg = u % mtg, v = u / mtg;
        tn = v % nt;
        s = v / nt;
        tm = tmg * 8 + r;
        if (tm * 64 >= M) return;
    } else {
        const int tiles = ((M + 63) >> 6) * nt;
        s = local / tiles;
        const int t = local - s * tiles;
        tm = t / nt;
        tn = t - tm * nt;
    }
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = tm * 64, n0 = tn * 64;
    // Every wave multiplies the WHOLE 64 x 64 tile over its own eighth of the split's rows: each row of the operands enters
    // the CU once (quadrant-per-wave layouts fetch every line twice, and the L1 fill rate is what bounds the stream), and
    // the four accumulators are independent MFMA chains.
    const int k0 = s * ks, k1 = min(K, k0 + ks);
    const int per = ((k1 - k0 + 15) >> 4) * 2;
    const int kb = k0 + wave * per, ke = min(k1, kb + per);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const bool hm = m0 + 32 < M, hn = n0 + 32 < N;      // second half of the tile's rows / columns exists (wave-uniform)
    if (kb < ke) {
        if (bidx) wgrad_kloop<true>(acc, A, B, bidx, lda, ldb, M, N, m0, n0, kb, ke, lane, hm, hn);
        else wgrad_kloop<false>(acc, A, B, bidx, lda, ldb, M, N, m0, n0, kb, ke, lane, hm, hn);
    }
    if (g.trace && threadIdx.x == 0) g.trace[8 * blockIdx.x + 4] = wall_clock64();
    // the eight waves' partial tiles meet in LDS, one 32 x 32 quadrant at a time; the S row ranges meet in float atomics
    // D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    // (launches with several rounds of workgroups issue the atomics after the LAST LDS round: a barrier waits for the
    // workgroup's outstanding memory operations and the memory side is saturated with atomics there, tools/wg_trace_wgrad.py)
    float out[2][2][2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
#pragma unroll
        for (int qj = 0; qj < 2; ++qj) {
            out[qi][qj][0] = out[qi][qj][1] = 0.0f;
            if ((qi && !hm) || (qj && !hn)) continue;      // (block-uniform)
            if (qi + qj > 0) __syncthreads();
            float* mine = lds + wave * 1024 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r * 64] = acc[qi][qj][r];
            __syncthreads();
            // 512 threads: element (r, lane) with r = 2 (tid >> 6) + {0, 1}
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * wave + e;
                float v = 0.0f;
#pragma unroll
                for (int w = 0; w < 8; ++w) v += lds[w * 1024 + r * 64 + lane];
                out[qi][qj][e] = v;
                if (!g.aux_first) {      // one round of workgroups: the atomics of a round overlap the next round's LDS traffic
                    const int gm = m0 + 32 * qi + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int gn = n0 + 32 * qj + l31;
                    if (gm < M && gn < N) atomicAdd(C + (int64_t)gm * ldc + gn, v);
                }
            }
        }
    }
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
        for (int qj = 0; qj < 2; ++qj) {
            if (!g.aux_first || (qi && !hm) || (qj && !hn)) continue;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * wave + e;
                const int gm = m0 + 32 * qi + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int gn = n0 + 32 * qj + l31;
                if (gm < M && gn < N) atomicAdd(C + (int64_t)gm * ldc + gn, out[qi][qj][e]);
            }
        }
    if (g.trace && threadIdx.x == 0) {
        long long* tr = g.trace + 8 * blockIdx.x;
        tr[0] = t_start; tr[1] = wall_clock64(); tr[2] = pi; tr[3] = s;
    }
}

}  // namespace

bool wgrad_t1_build(const pp_gemm_args* q, const GemmHole* holes, int n, WgradT1Args& out) {
    static const int env = getenv("PP_WGRAD_T1") ? atoi(getenv("PP_WGRAD_T1")) : 1;
    static const int env_s = 0;
    static const int env_xmap = getenv("PP_WGRAD_XMAP") ? atoi(getenv("PP_WGRAD_XMAP")) : 1;      // A/B: 0 = column tiles fastest
    static const bool env_wgs_set = false;
    static const int env_wgs = 240;
    if (!env || deterministic_mode() || n <= 0) return false;
    out = WgradT1Args{};
    int np = 0;
    // a cut [m_lo, m_hi) x [n_lo, n_hi) of problem a whose first k_lo rows contribute nothing
    auto add = [&](const pp_gemm_args& a, int m_lo, int m_hi, int n_lo, int n_hi, int k_lo) -> bool {
        if (m_hi <= m_lo || n_hi <= n_lo || k_lo >= a.K) return true;
        if (np >= WGRAD_T1_MAX) return false;
        WgradT1Prob& p = out.p[np++];
        p.A = a.A + (int64_t)k_lo * a.lda + m_lo;
        p.bidx = a.b_idx ? a.b_idx + k_lo : nullptr;
        p.B = a.B + (a.b_idx ? 0 : (int64_t)k_lo * a.ldb) + n_lo;
        p.C = a.C + (int64_t)m_lo * a.ldc + n_lo;
        p.lda = (int)a.lda; p.ldb = (int)a.ldb; p.ldc = (int)a.ldc;
        p.M = m_hi - m_lo; p.N = n_hi - n_lo; p.nt = cdiv(p.N, 64);
        p.K = a.K - k_lo;
        return true;
    };
    for (int i = 0; i < n; ++i) {
        const pp_gemm_args& a = q[i];
        if (!a.a_kmajor || !a.b_kmajor || a.a_idx || a.c_idx || a.bias || a.bias2 || a.mask || a.relu || a.colsum ||
            !a.accumulate || a.K <= 0 || a.M <= 0 || a.N <= 0 || a.lda >= (1 << 30) || a.ldb >= (1 << 30) || a.ldc >= (1 << 30))
            return false;
        // Zero blocks (GemmHole: the product is zero for m, n, k inside the block): only blocks that start at k = 0 are used
        // (a trace's first time step has no previous variable and a zero cell state). The tensor is cut along the blocks'
        // edges; a cell starts at the largest k1 of the blocks that cover it.
        int mc[6] = {0, (int)a.M, 0, 0, 0, 0}, nc[6] = {0, (int)a.N, 0, 0, 0, 0};
        int nm = 2, nn = 2;
        GemmBlock z[2];
        int nz = 0;
        if (holes) {
            for (int b = 0; b < 2; ++b) {
                GemmBlock h = holes[i].b[b];
                h.m0 = std::max(h.m0, 0); h.n0 = std::max(h.n0, 0);
                h.m1 = std::min(h.m1, (int)a.M); h.n1 = std::min(h.n1, (int)a.N); h.k1 = std::min(h.k1, (int)a.K);
                if (h.m1 <= h.m0 || h.n1 <= h.n0 || h.k1 <= h.k0 || h.k0 > 0) continue;      // (k0 > 0: not used, still correct)
                z[nz++] = h;
                mc[nm++] = h.m0; mc[nm++] = h.m1;
                nc[nn++] = h.n0; nc[nn++] = h.n1;
            }
        }
        std::sort(mc, mc + nm);
        std::sort(nc, nc + nn);
        nm = (int)(std::unique(mc, mc + nm) - mc);
        nn = (int)(std::unique(nc, nc + nn) - nc);
        for (int jn = 0; jn + 1 < nn; ++jn) {
            int run_lo = -1, run_hi = -1, run_k = 0;      // consecutive row intervals with the same first row merge
            for (int jm = 0; jm + 1 < nm; ++jm) {
                int k_lo = 0;
                for (int b = 0; b < nz; ++b)
                    if (z[b].m0 <= mc[jm] && z[b].m1 >= mc[jm + 1] && z[b].n0 <= nc[jn] && z[b].n1 >= nc[jn + 1])
                        k_lo = std::max(k_lo, z[b].k1);
                if (run_lo >= 0 && k_lo == run_k) {
                    run_hi = mc[jm + 1];
                } else {
                    if (run_lo >= 0 && !add(a, run_lo, run_hi, nc[jn], nc[jn + 1], run_k)) return false;
                    run_lo = mc[jm]; run_hi = mc[jm + 1]; run_k = k_lo;
                }
            }
            if (run_lo >= 0 && !add(a, run_lo, run_hi, nc[jn], nc[jn + 1], run_k)) return false;
        }
    }
    if (np == 0) return false;
    // row splits: ~240 workgroups in the launch when the work is small (one per CU: fewer, longer row ranges mean fewer
    // atomics), one workgroup per tile when there are more tiles than that; at least 64 rows per split
    int64_t tile_rows = 0;
    for (int i = 0; i < np; ++i) tile_rows += (int64_t)cdiv(out.p[i].M, 64) * out.p[i].nt * out.p[i].K;
    // (large launches - ragged batches: 10^6 tile rows - want ~600 rows per workgroup: balance over 2 x 256 slots matters more
    // than the atomics of the extra splits)
    const int64_t blocks = env_wgs_set ? env_wgs : std::min<int64_t>(1536, std::max<int64_t>(env_wgs, tile_rows / 600));
    const int target_rows = (int)std::max<int64_t>(64, cdiv(tile_rows, std::max<int64_t>(blocks, 1)));
    out.n_prob = np;
    // the longest row ranges first: their workgroups start first and the short ones fill the tail
    std::stable_sort(out.p, out.p + np, [](const WgradT1Prob& x, const WgradT1Prob& y) { return x.K > y.K; });
    int first = 0;
    for (int i = 0; i < np; ++i) {
        WgradT1Prob& p = out.p[i];
        int S = env_s > 0 ? env_s : (p.K + target_rows / 2) / target_rows;
        S = std::max(1, std::min(S, std::min(16, std::max(1, p.K / 64))));
        p.S = S;
        p.ks = ((cdiv(p.K, S) + 3) / 4) * 4;
        p.first = first;
        const int mt = cdiv(p.M, 64);
        p.mtg = (env_xmap && mt >= 8) ? cdiv(mt, 8) : 0;
        first += (p.mtg ? p.mtg * 8 : mt) * p.nt * S;
    }
    out.n_blocks = first;
    return true;
}

int wgrad_t1(const WgradT1Args& a, const AuxJobs* aux, hipStream_t st) {
    static const AuxJobs none{};
    const AuxJobs& j = aux ? *aux : none;
    const int blocks = a.n_blocks + (aux ? j.n_blocks : 0);
    WgradT1Args t = a;
    t.aux_first = (aux && a.n_blocks > 512) ? 1 : 0;      // 2 workgroups x 256 CUs = one round
    if (g_wgtrace && g_wgtrace_mode == 1 && blocks <= g_wgtrace_cap) t.trace = g_wgtrace;
    hipLaunchKernelGGL(wgrad_t1_kernel, dim3(blocks), dim3(512), 0, st, t, j);
    PP_LAUNCH_CHECK("wgrad_t1");
    return 0;
}

}  // namespace pp

// Host-side plan of the weight-gradient launch for `count` queued products (no device work): which cuts of the tensors become
// problems, their row ranges and splits. For tests of the host logic (tests/test_host.py). zero_blocks: [count][2][6] =
// {m0, m1, n0, n1, k0, k1} per product or NULL; out: [cap][10] = {M, N, K, S, ks, gather, first, C offset, A offset, B offset}
// (offsets in floats from the product's own pointers). Returns the number of problems, 0 if the tile kernels take the flush.
extern "C" int pp_debug_wgrad_plan(const pp_gemm_args* q, const int32_t* zero_blocks, int32_t count, int64_t* out, int32_t cap,
                                   int32_t* n_blocks) {
    std::vector<pp::GemmHole> holes((size_t)std::max(count, 0));
    for (int i = 0; i < count && zero_blocks; ++i)
        for (int b = 0; b < 2; ++b) {
            const int32_t* z = zero_blocks + ((int64_t)i * 2 + b) * 6;
            holes[i].b[b] = pp::GemmBlock{z[0], z[1], z[2], z[3], z[4], z[5]};
        }
    pp::WgradT1Args a;
    if (!pp::wgrad_t1_build(q, zero_blocks ? holes.data() : nullptr, count, a)) return 0;
    if (n_blocks) *n_blocks = a.n_blocks;
    for (int i = 0; i < a.n_prob && i < cap; ++i) {
        const pp::WgradT1Prob& p = a.p[i];
        // which product does the problem belong to: the one whose C range contains its C
        int src = -1;
        for (int k = 0; k < count; ++k)
            if (p.C >= q[k].C && p.C < q[k].C + (int64_t)q[k].M * q[k].ldc) src = k;
        int64_t* o = out + (int64_t)i * 10;
        o[0] = p.M; o[1] = p.N; o[2] = p.K; o[3] = p.S; o[4] = p.ks; o[5] = p.bidx ? 1 : 0; o[6] = p.first;
        o[7] = src >= 0 ? p.C - q[src].C : -1;
        o[8] = src >= 0 ? p.A - q[src].A : -1;
        o[9] = src >= 0 ? (p.bidx ? p.bidx - q[src].b_idx : (p.B - q[src].B)) : -1;
    }
    return a.n_prob;
}

