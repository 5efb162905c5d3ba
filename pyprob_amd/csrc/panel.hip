// Row-panel kernel for batches of single-statement traces (T = 1: GaussianUnknownMean, BASELINE.json configs[1]).
//
// With one controlled statement per trace the whole DATA path of InferenceNetworkLSTM._loss + backward is row-local
// (pyprob/nn/inference_network_lstm.py:136-220 with h0 = c0 = 0, :186-187):
//     E row -> G = E W_ih[:, :e]^T + bias(address) -> LSTM cell -> h -> z1 = relu(h W1^T + b1) -> y = z1 W2^T + b2
//       -> mixture log_prob / loss / dy -> dz1 = (dy W2) * [z1 > 0] -> dh = dz1 W1 -> cell backward -> dG -> dX = dG W_ih[:, :e]
// and only the weight gradients reduce over rows. Five launches of the tile kernels (input product + cell | head layer 1 |
// head tails | dH + cell backward | dX: 55 us of a 114 us step, each mostly fill / drain / first-slab latency) become ONE
// launch in which a workgroup owns a panel of 8 rows for the whole chain and the weights stream past it.
//
// Arithmetic: v_mfma_f32_4x4x1_16B_f32 - sixteen independent 4x4x1 blocks per instruction. With the activations of 4 batch
// rows broadcast to every block (lane l supplies row l % 4) and ONE weight per lane (lane l supplies output column l),
// register i of lane l accumulates out[row i][column l]: 4 rows x 64 columns x 1 k per instruction (tools/micro/
// mfma_4x4_probe.hip: one wave issues one every 11 cycles and up to four waves per SIMD do so concurrently) - a 16- or
// 32-row MFMA tile would leave 3/4 of the chip idle at 1024 rows. Two row groups share every weight register.
//
// Weights: every product reads its weight matrix K-MAJOR (row = summation index, the 64 output columns of a tile
// contiguous), so that the B operand of an MFMA is ONE coalesced dword load per lane (256 B per wave instruction, row
// address in scalar registers) straight into a register, two 16-k units ahead of its use. dh = dz1 W1 and dX = dG W_ih
// have that layout in the parameter tensors themselves; the forward products read TRANSPOSED COPIES (WihT [e][4H],
// W1T [H][64 ceil(hid / 64)]) written by extra workgroups of the step's first launch (panel_transpose_block).
// What the first version taught (tools/panel_timeline.py, profiles/r03_panel_*): weights through an LDS ring filled by
// LDS-DMA cost ~100 cycles of wave time per 1 KB DMA instruction (400 per 16-k unit, more than its 32 MFMAs), and per-lane
// 64-bit address arithmetic for direct loads cost as much again - instruction issue, not memory, bounds this kernel.
//
// Ownership: wave w owns hidden units [64 UT w, 64 UT (w + 1)) in the input product (gates i, g, o of those units; the
// forget gate multiplies c0 = 0 and is never formed) AND in dh = dz1 W1, so the gate activations stay in its registers
// from the forward cell to the backward cell. Head layer 1 (271 columns = 4.2 tiles), the tail product and dX (one tile)
// split K over the waves instead and meet through LDS float atomics.
//
// Not bit-reproducible (LDS / global float atomics): PP_DETERMINISTIC=1 keeps the tile path. Parity: tests/test_gpu_panel.py
// (every buffer and gradient against the tile path and the oracle).
#include "common.hpp"
#include "panel.hpp"
#include "handoff.hpp"

#include <algorithm>

namespace pp {

namespace {

constexpr int PANEL_ROWS = 8;
constexpr int PANEL_WAVES = 8;
constexpr float kFp32Eps = 1.1920928955078125e-07f;
constexpr float kHalfLog2Pi = 0.91893853320467274178f;
constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kInvSqrt2Pi = 0.39894228040143267794f;
constexpr float kLogEps = -18.420680743952367f;

// sigmoid / tanh on the hardware exp2 and reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each; absolute error of the results ~1e-7)
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ float std_cdf_(float x) { return 0.5f * (1.0f + erff(x * kInvSqrt2)); }
__device__ __forceinline__ float std_pdf_(float x) { return kInvSqrt2Pi * expf(-0.5f * x * x); }

__device__ __forceinline__ void wave_sync_lds() {   // order this wave's LDS writes before its LDS reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One 16-k unit of a k-major weight matrix for this lane's output column: w[kk] = W[(k0 + kk)][n0 + lane]. `row0` is the
// wave-uniform address of element (k0, n0), ld the row pitch, voff = 4 * lane: scalar base + 32-bit lane offset, so the
// row addresses are computed on the scalar unit (no per-lane 64-bit address arithmetic).
__device__ __forceinline__ void load_unit(const float* __restrict__ row0, int64_t ld, uint32_t voff, float (&w)[16]) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
        w[kk] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(row0 + kk * ld) + voff);
}
// the same with the k rows clamped to kmax - 1 (a unit that reaches past the last row; its products meet zero activations)
__device__ __forceinline__ void load_unit_clamped(const float* __restrict__ base, int64_t ld, int k0, int kmax, uint32_t voff,
                                                  float (&w)[16]) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
        w[kk] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + (int64_t)min(k0 + kk, kmax - 1) * ld) + voff);
}
// The 16 k of a unit for both row groups. The A operand of v_mfma_f32_4x4x1 is per block (lanes 4 b .. 4 b + 3 = rows 0..3
// of block b); with CBSZ = 4 the instruction broadcasts block ABID's four values to all sixteen blocks. So ONE register per
// row group holds the activations of 16 k - lane l: act[row l % 4][k0 + l / 4] (one ds_read_b32, pitches = 8 mod 32 keep
// the 32-lane halves conflict-free) - and MFMA j of the unit selects k0 + j with ABID = j. (A first version read the
// activations as broadcast ds_read_b128: 8 LDS reads and 32 registers per unit; the LDS was busy half of the kernel.)
#define PP_MMA_STEP(J)                                                   \
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, w[J], c0, 4, J, 0);      \
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, w[J], c1, 4, J, 0);
__device__ __forceinline__ void mma_unit(const float* act /* + row * pitch + k0 + lane / 4 applied by the caller */, int pitch,
                                         const float (&w)[16], f32x4& c0, f32x4& c1) {
    const float a0 = act[0], a1 = act[4 * pitch];
    PP_MMA_STEP(0) PP_MMA_STEP(1) PP_MMA_STEP(2) PP_MMA_STEP(3) PP_MMA_STEP(4) PP_MMA_STEP(5) PP_MMA_STEP(6) PP_MMA_STEP(7)
    PP_MMA_STEP(8) PP_MMA_STEP(9) PP_MMA_STEP(10) PP_MMA_STEP(11) PP_MMA_STEP(12) PP_MMA_STEP(13) PP_MMA_STEP(14) PP_MMA_STEP(15)
}

// A stream of n 16-k units through NB register sets, loads NB - 1 units ahead of the MFMAs (a wave's stream is latency
// bound: ~2 000 cycles from issue to use under load, ~350 cycles of MFMAs per unit; vmcnt holds at most 63 loads, so
// NB = 4 = 48 + the unit being consumed is the deepest that fits):
//   load(i, w)  issues the 16 loads of unit i into w;  use(i, w)  consumes unit i.
template <int NB, typename Load, typename Use>
__device__ __forceinline__ void stream_units(int n, Load load, Use use) {
    float w[NB][16];
#pragma unroll
    for (int b = 0; b < NB - 1; ++b)
        if (b < n) load(b, w[b]);
    for (int i = 0; i < n; i += NB) {
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            if (i + s < n) {
                if (i + s + NB - 1 < n) load(i + s + NB - 1, w[(s + NB - 1) % NB]);
                use(i + s, w[s]);
            }
        }
    }
}

__host__ __device__ inline int round4i(int x) { return (x + 3) & ~3; }

struct PanelLds {   // offsets in floats
    int PE, PH, PZ, zk, ws;
    int sE, sH, sZ, sDZ, sDY, sW2, sP, PP, total;
};
__host__ __device__ inline PanelLds panel_lds(int H, int hid, int n_out, int e) {
    PanelLds L;
    L.zk = (hid + 15) & ~15;
    L.PE = e + 8; L.PH = H + 8; L.PZ = L.zk + 8;      // = 8 (or 24) mod 32: see mma_unit
    L.ws = hid | 1;
    int o = 0;
    L.sE = o; o += PANEL_ROWS * L.PE;
    // sH, later (with sZ behind it) the four tiles' dG staging images [tile][gate][8][72] of phases 5 / 6
    {
        const int stage = 4 * 3 * PANEL_ROWS * 72 - PANEL_ROWS * L.PZ;
        L.sH = o; o += (PANEL_ROWS * L.PH > stage ? PANEL_ROWS * L.PH : stage);
    }
    L.sZ = o; o += PANEL_ROWS * L.PZ;
    L.sDZ = o; o += PANEL_ROWS * L.PZ;
    L.sDY = o; o += PANEL_ROWS * 72;
    L.sW2 = o; o += round4i(n_out * L.ws);
    // per-wave partial tiles of the K-split products (head layer 1: [wave][row][hid]; tail product and dX: [wave][row][64]):
    // plain stores + a reduction pass after a barrier. (LDS float atomics were the first version's bottleneck: ~450 ds_add_f32
    // wave instructions per workgroup kept the LDS busy for half of the kernel, profiles/r03_panel_pmc_*.csv.)
    L.PP = ((hid + 3) & ~3) > 64 ? ((hid + 3) & ~3) : 64;
    L.sP = o; o += PANEL_WAVES * PANEL_ROWS * L.PP;
    L.total = o;
    return L;
}

// KIND: head kind (0 Normal mixture, 1 TruncatedNormal mixture in a Uniform prior, 2 Poisson head), kernels.hip.
// SPLIT: workgroups per panel. With SPLIT = 2 the two workgroups of a pair
// own one half of the hidden units each: a workgroup streams HALF of every weight matrix (the stream of the weights through
// one CU's L1 at ~25 B/clk is what bounds the kernel, profiles/r03_panel_v3_pmc_*.csv), the K-split partial sums of head
// layer 1 and of dX cross between the pair through memory (two hand-offs of 8.7 KB and 2 KB), the small tail phases run
// redundantly on both, each writes 4 of the 8 rows of the shared outputs. 1 024 rows then fill all 256 CUs.
template <int KIND, int SPLIT, bool OBS>
__global__ __launch_bounds__(512) void panel_t1_kernel(const PanelArgs ain, const PanelObs oin) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const PanelArgs a = ain;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = blockIdx.x;
    const int half = SPLIT == 2 ? ((bx >> 3) & 1) : 0;
    const int panel = SPLIT == 2 ? ((bx >> 4) * 8 + (bx & 7)) : bx;
    const int m0 = panel * PANEL_ROWS;
    if (m0 >= a.B) return;                       // (both workgroups of a pair)
    constexpr int NOWN = PANEL_WAVES / SPLIT;    // waves that own a 64-unit tile of hidden units
    constexpr int NB = 4;                        // register sets of the weight streams
    // SPLIT = 2: the workgroup's hidden units are 4 tiles; wave w < 4 OWNS tile w (gates in its registers), wave w + 4 HELPS
    // it: in the products that run over an owner's columns (input product, dh, dX) the two split the K range and the
    // helper's partial sums cross through LDS.
    const int tw = wave % NOWN;                  // tile of this wave
    const int hp = wave / NOWN;                  // 0 owner, 1 helper
    constexpr int NH = SPLIT;                    // waves per tile
    const bool own = hp == 0;
    const int H = a.H, hid = a.hid, e = a.e, n_out = a.n_out, K = a.K;
    const PanelLds L = panel_lds(H, hid, n_out, e);
    float* const sE = smem + L.sE;
    float* const sH = smem + L.sH;
    float* const sZ = smem + L.sZ;
    float* const sDZ = smem + L.sDZ;
    float* const sDY = smem + L.sDY;
    float* const sW2 = smem + L.sW2;
    float* const sP = smem + L.sP;           // partial tiles, this wave's block at sP + wave * 8 * PP
    const int PP = L.PP;
    const int PE = L.PE, PH = L.PH, PZ = L.PZ, zk = L.zk, ws = L.ws;
    const int rl = lane & 3;                 // this lane's row inside a row group (A operand)
    const int kl = lane >> 2;                // ... and its k inside a 16-k unit
    const uint32_t voff = 4u * (uint32_t)lane;
    const int HS = H / SPLIT;                // hidden units of this workgroup
    const int U0 = half * HS + tw * 64;      // first hidden unit of this wave's tile
    const int KB = half * HS + wave * (HS / PANEL_WAVES);      // this wave's k range of head layer 1: [KB, KB + HS / 8)
    const int epoch = *a.epoch;
    const int NT5 = (hid + 63) >> 6;         // 64-column tiles of the head's hidden layer
    const int KU = e >> 4;                   // 16-k units of the input product
    const int64_t ldT = 4 * (int64_t)H + 64; // row pitch of WihT (+ 256 B: an 8 KB pitch would put all 16 rows of a unit on one L2 channel)
    const int64_t ld1T = 64 * (int64_t)NT5;  // row pitch of W1T
    const int dbg_slot = (bx == 0 ? 0 : (bx == 77 ? 1 : -1));
#define PANEL_STAMP(k)                                                                                          \
    do {                                                                                                        \
        if (a.dbg && dbg_slot >= 0 && lane == 0 && (wave == 0 || wave == 5))                                    \
            a.dbg[(dbg_slot * 2 + (wave == 5 ? 1 : 0)) * 16 + (k)] = clock64();                                 \
    } while (0)
    PANEL_STAMP(0);

    // ---------------- staging: E rows now; W2 travels in registers until the tail needs it ----------------
    {
        const int e4 = e >> 2;
        if (tid < PANEL_ROWS * e4) {
            const int r = tid / e4, c = tid - r * e4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(a.X + (int64_t)min(m0 + r, a.B - 1) * a.ldx + 4 * c);
            *reinterpret_cast<f32x4*>(sE + r * PE + 4 * c) = v;
        }
        for (int i = tid; i < PANEL_ROWS * PZ; i += 512) { sZ[i] = 0.0f; sDZ[i] = 0.0f; }      // (the zero pads matter)
        for (int i = tid; i < PANEL_ROWS * 72; i += 512) sDY[i] = 0.0f;
    }
    // bias of this lane's hidden units (gates i, g, o), fetched now, used in the cell
    float bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bias[g] = a.AB[(g == 0 ? 0 : g + 1) * H + U0 + lane];
    __syncthreads();
    PANEL_STAMP(1);

    // ---------------- phase 1: G = E W_ih[:, :e]^T + bias, LSTM cell (c0 = 0) ----------------
    float gi[8], gg[8], go[8], tc[8];   // gate activations and tanh(c) of (row, unit U0 + lane): owners only
    {
        f32x4 acc[3][2];
#pragma unroll
        for (int g = 0; g < 3; ++g) { acc[g][0] = f32x4{0, 0, 0, 0}; acc[g][1] = f32x4{0, 0, 0, 0}; }
        // units (gate, q): columns gate_row(g) H + U0 .., k = 16 q of WihT; owner and helper take half of the k units each
        const int KQ1 = KU / NH, q0 = hp * KQ1;
        stream_units<NB>(
            3 * KQ1,
            [&](int i, float (&w)[16]) {
                const int g = i / KQ1, q = q0 + i - g * KQ1;
                load_unit(a.WihT + (int64_t)(16 * q) * ldT + (g == 0 ? 0 : g + 1) * H + U0, ldT, voff, w);
            },
            [&](int i, const float (&w)[16]) {
                const int g = i / KQ1, q = q0 + i - g * KQ1;
                const float* act = sE + rl * PE + 16 * q + kl;
                if (g == 0) mma_unit(act, PE, w, acc[0][0], acc[0][1]);
                else if (g == 1) mma_unit(act, PE, w, acc[1][0], acc[1][1]);
                else mma_unit(act, PE, w, acc[2][0], acc[2][1]);
            });
        if (NH == 2) {      // helper -> owner: 24 partial sums per lane through sP (not in use before phase 2)
            float* hx = sP + tw * (24 * 64) + lane;
            if (!own) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                        for (int i = 0; i < 4; ++i) hx[((g * 2 + rg) * 4 + i) * 64] = acc[g][rg][i];
            }
            __syncthreads();
            if (own) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[g][rg][i] += hx[((g * 2 + rg) * 4 + i) * 64];
            }
        }
        // cell: rows 4 rg + i, unit u = U0 + lane. sigmoid and tanh through v_exp_f32 / v_rcp_f32 (absolute error ~1e-7:
        // the accurate library forms cost ~1 000 instructions per wave here)
        if (own) {
            const int u = U0 + lane;
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * rg + i;
                    const float vi = fast_sigmoid(acc[0][rg][i] + bias[0]);
                    const float vg = fast_tanh(acc[1][rg][i] + bias[1]);
                    const float vo = fast_sigmoid(acc[2][rg][i] + bias[2]);
                    const float c = vi * vg;
                    const float tcv = fast_tanh(c);
                    const float h = vo * tcv;
                    gi[r] = vi; gg[r] = vg; go[r] = vo; tc[r] = tcv;
                    sH[r * PH + u] = h;
                    if (m0 + r < a.B) a.Hs[(int64_t)(m0 + r) * H + u] = h;
                }
        }
    }
    PANEL_STAMP(2);
    __syncthreads();     // sH complete
    PANEL_STAMP(3);
    // W2 [n_out][hid] -> LDS image [n_out][ws]: the loads are issued here, the LDS stores happen after phase 2's units
    const int wtot = n_out * hid;
    const bool w2flat = (ws == hid) && (reinterpret_cast<uintptr_t>(a.W2) & 15) == 0 && wtot <= 512 * 4 * 5;
    f32x4 w2v[5];
    if (w2flat) {
        const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(a.W2);
        const int n4 = wtot >> 2;
#pragma unroll
        for (int u = 0; u < 5; ++u) w2v[u] = src[min(tid + 512 * u, n4 - 1)];
    }
    // ---------------- phase 2: z1 = relu(h W1^T + b1); K split over the waves, partial tiles meet in sZ ----------------
    // this wave's k range is an eighth of the workgroup's hidden units; items = column tiles, KQ units each; one accumulator
    {
        const int KQ = HS / (PANEL_WAVES * 16);
        f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        stream_units<NB>(
            NT5 * KQ,
            [&](int i, float (&w)[16]) {
                const int t = i / KQ, q = i - t * KQ;
                load_unit(a.W1T + (int64_t)(KB + 16 * q) * ld1T + 64 * t, ld1T, voff, w);
            },
            [&](int i, const float (&w)[16]) {
                const int t = i / KQ, q = i - t * KQ;
                mma_unit(sH + rl * PH + KB + 16 * q + kl, PH, w, c0, c1);
                if (q == KQ - 1) {      // the tile's partial sums over this wave's k range are complete
                    const int n = 64 * t + lane;
                    if (n < hid) {
                        float* pw = sP + wave * (PANEL_ROWS * PP) + n;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            pw[r * PP] = c0[r];
                            pw[(4 + r) * PP] = c1[r];
                        }
                    }
                    c0 = f32x4{0, 0, 0, 0}; c1 = f32x4{0, 0, 0, 0};
                }
            });
    }
    if (w2flat) {       // W2 -> LDS (loaded at the top of the kernel)
        const int n4 = wtot >> 2;
#pragma unroll
        for (int u = 0; u < 5; ++u)
            if (tid + 512 * u < n4) *reinterpret_cast<f32x4*>(sW2 + 4 * (tid + 512 * u)) = w2v[u];
        for (int i = 4 * n4 + tid; i < wtot; i += 512) sW2[i] = a.W2[i];
    } else {
        const float inv = 1.0f / (float)hid;
        for (int base = tid; base < wtot; base += 512 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = a.W2[min(base + 512 * u, wtot - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + 512 * u;
                if (i < wtot) {
                    const int o = (int)(((float)i + 0.5f) * inv);
                    sW2[o * ws + (i - o * hid)] = v[u];
                }
            }
        }
    }
    PANEL_STAMP(4);
    __syncthreads();
    PANEL_STAMP(5);
    {   // z1 = relu(b1 + this half's partial sums + the partner's); A1 rows to memory (pad columns zero)
        unsigned long long* const xz_own = a.xz + (int64_t)(panel * 2 + half) * (PANEL_ROWS * a.lda1);
        const unsigned long long* const xz_other = a.xz + (int64_t)(panel * 2 + (half ^ 1)) * (PANEL_ROWS * a.lda1);
        constexpr int EPT = 5;      // elements per thread: 8 rows x lda1 <= 512 x 5 (hid <= 320)
        float mine[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = tid + 512 * k;
            mine[k] = 0.0f;
            if (i < PANEL_ROWS * a.lda1) {
                const int r = i / a.lda1, n = i - r * a.lda1;
                if (n < hid) {
                    const float* pr = sP + r * PP + n;
#pragma unroll
                    for (int wv = 0; wv < PANEL_WAVES; ++wv) mine[k] += pr[wv * (PANEL_ROWS * PP)];
                    gput(xz_own + i, mine[k], epoch);
                }
            }
        }
        PANEL_STAMP(14);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = tid + 512 * k;
            if (i < PANEL_ROWS * a.lda1) {
                const int r = i / a.lda1, n = i - r * a.lda1;
                float z = 0.0f;
                if (n < hid) {      // (half 0) + (half 1), the same order on both sides: identical z1 in both workgroups
                    const float theirs = gget(xz_other + i, epoch);
                    z = relu_keep_nan((half == 0 ? mine[k] + theirs : theirs + mine[k]) + a.b1[n]);
                    sZ[r * PZ + n] = z;
                }
                if (m0 + r < a.B && (r >> 2) == half) a.A1[(int64_t)(m0 + r) * a.lda1 + n] = z;
            }
        }
    }
    __syncthreads();
    PANEL_STAMP(6);
    // ---------------- phase 3: y = z1 W2^T (+ b2 below); 16-k units dealt to the waves, lanes = outputs ----------------
    {
        const float* wr = sW2 + min(lane, n_out - 1) * ws;
        f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        for (int u = wave; u < (zk >> 4); u += PANEL_WAVES) {
            float w[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) w[j] = wr[min(16 * u + j, hid - 1)];   // (k >= hid multiplies the zero pad of z1)
            mma_unit(sZ + rl * PZ + 16 * u + kl, PZ, w, c0, c1);
        }
        // (sP is free again: the fix-up pass consumed the partial tiles of head layer 1 before the barrier above)
        float* pw = sP + wave * (PANEL_ROWS * PP) + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pw[i * PP] = c0[i];
            pw[(4 + i) * PP] = c1[i];
        }
    }
    __syncthreads();
    PANEL_STAMP(7);
    // ---------------- mixture log_prob, loss, d lp / d y: wave w = row w, one mixture component per lane ----------------
    {
        const int r = wave;
        const bool rowok = m0 + r < a.B;
        const bool mine = rowok && (SPLIT == 1 || (r >> 2) == half);     // this workgroup writes the row's shared outputs
        const int gr = min(m0 + r, a.B - 1);
        const bool comp = lane < K;
        float ymu = 0.0f, ysd = 0.0f, yz = -INFINITY;
        if (comp) {      // y = b2 + the eight waves' partial sums
            ymu = a.b2[lane]; ysd = a.b2[K + lane]; yz = a.b2[2 * K + lane];
            const float* pr = sP + r * PP + lane;
#pragma unroll
            for (int wv = 0; wv < PANEL_WAVES; ++wv) {
                ymu += pr[wv * (PANEL_ROWS * PP)];
                ysd += pr[wv * (PANEL_ROWS * PP) + K];
                yz += pr[wv * (PANEL_ROWS * PP) + 2 * K];
            }
        }
        const float v = a.value[gr], pa = a.prior[2 * gr], pb = a.prior[2 * gr + 1];
        const float zmax = wave_max(yz);
        const float ex = comp ? expf(yz - zmax) : 0.0f;
        const float pi = ex / wave_sum(ex);
        const float ps = wave_sum(pi);
        const float p = pi / ps;
        float mu, sd, sm = 0.f, ss = 0.f;
        const float rng = pb - pa;
        if (KIND == 0) {
            mu = pa + ymu * pb;
            sd = expf(ysd) * pb;
        } else {
            sm = sigmoidf_(ymu);
            ss = sigmoidf_(ysd);
            mu = pa + sm * rng;
            sd = KIND == 2 ? expf(ysd) : rng / 1000.0f + ss * rng * 10.0f;
        }
        const float tt = (v - mu) / sd;
        float cl, alpha = 0.f, beta = 0.f, Z = 1.f;
        if (KIND == 0) {
            cl = -0.5f * tt * tt - logf(sd) - kHalfLog2Pi;
        } else {
            alpha = (pa - mu) / sd;
            beta = (pb - mu) / sd;
            Z = std_cdf_(beta) - std_cdf_(alpha);
            const bool inside = v >= pa && v <= pb;
            cl = (inside ? 0.0f : -INFINITY) + (-0.5f * tt * tt - kHalfLog2Pi) - logf(sd * Z);
        }
        const float al = comp ? logf(fminf(fmaxf(p, kFp32Eps), 1.0f - kFp32Eps)) + cl : -INFINITY;
        const float amax = wave_max(al);
        float lp = amax;
        if (amax > -INFINITY) lp = amax + logf(wave_sum(comp ? expf(al - amax) : 0.0f));
        if (wave_sum((comp && al != al) ? 1.0f : 0.0f) > 0.0f) lp = NAN;   // NaN in a component poisons the logsumexp
        const bool rescued = (lp == -INFINITY);
        const bool bad = !rescued && !isfinite(lp);
        if (mine && lane == 0) {
            if (a.lp_out) a.lp_out[gr] = lp;
            atomicAdd(a.loss_acc + 32 * ((blockIdx.x * PANEL_WAVES + wave) & 63), rescued ? -kLogEps : -lp);
            if (bad) atomicOr(a.flag, 1);
        }
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
        const bool live = rowok && !(rescued || bad);
        {
            const float resp = (comp && live) ? expf(al - lp) : 0.0f;
            const bool in = (p >= kFp32Eps) && (p <= 1.0f - kFp32Eps);
            float dp = (comp && in) ? resp / p : 0.0f;
            const float dpp = wave_sum(dp * p);
            dp = comp ? (dp - dpp) / ps : 0.0f;
            const float dpipi = wave_sum(dp * pi);
            if (comp && live) {
                if (KIND == 0) {
                    d0 = a.grad_scale * resp * tt / sd * pb;
                    d1 = a.grad_scale * resp * (tt * tt - 1.0f);
                } else {
                    const float fa = std_pdf_(alpha), fb = std_pdf_(beta);
                    const float dmu = resp * (tt / sd - (fa - fb) / (sd * Z));
                    const float dsd = resp * ((tt * tt - 1.0f) / sd - (alpha * fa - beta * fb) / (sd * Z));
                    d0 = a.grad_scale * dmu * rng * sm * (1.0f - sm);
                    d1 = a.grad_scale * dsd * (KIND == 2 ? sd : rng * 10.0f * ss * (1.0f - ss));
                }
                d2 = a.grad_scale * pi * (dp - dpipi);
            }
        }
        if (comp) {
            sDY[r * 72 + lane] = d0; sDY[r * 72 + K + lane] = d1; sDY[r * 72 + 2 * K + lane] = d2;
            if (mine) {
                float* dy = a.DY + (int64_t)gr * a.lddy;
                dy[lane] = d0; dy[K + lane] = d1; dy[2 * K + lane] = d2;
            }
        }
        if (mine && lane >= n_out && lane < a.lddy) a.DY[(int64_t)gr * a.lddy + lane] = 0.0f;   // pad columns
    }
    __syncthreads();
    PANEL_STAMP(8);
    // ---------------- phase 4: dz1 = (dy W2) * [z1 > 0]; lanes = hidden columns, k = outputs ----------------
    {
        const int ku = (n_out + 15) >> 4;        // 16-k units of dy (zero beyond n_out)
        for (int t = wave; t < NT5; t += PANEL_WAVES) {
            const int j = 64 * t + lane, jc = min(j, hid - 1);
            f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
            for (int u = 0; u < ku; ++u) {
                float w[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) w[q] = sW2[min(16 * u + q, n_out - 1) * ws + jc];
                mma_unit(sDY + rl * 72 + 16 * u + kl, 72, w, c0, c1);
            }
            if (j < zk) {
#pragma unroll
                for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * rg + i;
                        const float d = (j < hid && sZ[r * PZ + j] > 0.0f) ? (rg ? c1[i] : c0[i]) : 0.0f;
                        sDZ[r * PZ + j] = d;
                        if (j < a.lda1 && m0 + r < a.B && (SPLIT == 1 || rg == half)) a.dZ1[(int64_t)(m0 + r) * a.lda1 + j] = d;
                    }
            }
        }
    }
    PANEL_STAMP(9);
    // ---------------- phase 5: dh = dz1 W1 for this wave's hidden units, cell backward in registers ----------------
    const int KZ = zk >> 4;            // 16-k units over the head's hidden layer (rows beyond hid meet the zero pad of dz1)
    __syncthreads();     // sDZ complete (and sH is free: the dG staging of phase 6 lives there)
    PANEL_STAMP(10);
    float gs_i = 0.f, gs_g = 0.f, gs_o = 0.f;      // column sums of dG over this panel's rows (owners)
    // [gate][8 rows][64 k + 8]: the tile's dG, read by owner AND helper (the images run from sH into sZ: h and z1 are dead)
    float* const stg3 = sH + tw * (3 * PANEL_ROWS * 72);
    {
        f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        const float* wcol = a.W1 + U0;
        const int KZh = NH == 2 ? (KZ + 1) >> 1 : KZ;          // owner: units [0, KZh), helper: [KZh, KZ)
        const int qb = hp * KZh, qn = hp == 0 ? KZh : KZ - KZh;
        stream_units<NB>(
            qn,
            [&](int i, float (&w)[16]) {
                const int q = qb + i;
                if (16 * q + 16 <= hid) load_unit(wcol + (int64_t)(16 * q) * H, H, voff, w);
                else load_unit_clamped(wcol, H, 16 * q, hid, voff, w);
            },
            [&](int i, const float (&w)[16]) { mma_unit(sDZ + rl * PZ + 16 * (qb + i) + kl, PZ, w, c0, c1); });
        if (NH == 2) {      // helper -> owner through sP (free between the mixture phase and the partial tiles of dX)
            float* hx = sP + tw * (8 * 64) + lane;
            if (!own) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { hx[i * 64] = c0[i]; hx[(4 + i) * 64] = c1[i]; }
            }
            __syncthreads();
            if (own) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { c0[i] += hx[i * 64]; c1[i] += hx[(4 + i) * 64]; }
            }
        }
        if (own) {
            const int u = U0 + lane;
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * rg + i;
                    const float dh = rg ? c1[i] : c0[i];
                    const float vi = gi[r], vg = gg[r], vo = go[r], tcv = tc[r];
                    const float dc = dh * vo * (1.0f - tcv * tcv);
                    float d_i = dc * vg * vi * (1.0f - vi);
                    float d_g = dc * vi * (1.0f - vg * vg);
                    float d_o = dh * tcv * vo * (1.0f - vo);
                    if (m0 + r >= a.B) { d_i = 0.0f; d_g = 0.0f; d_o = 0.0f; }
                    gs_i += d_i; gs_g += d_g; gs_o += d_o;
                    // dG: to memory (the weight-gradient launch reads it) and into the tile's staging image (A operand of dX)
                    stg3[(0 * 8 + r) * 72 + lane] = d_i;
                    stg3[(1 * 8 + r) * 72 + lane] = d_g;
                    stg3[(2 * 8 + r) * 72 + lane] = d_o;
                    if (m0 + r < a.B) {
                        float* g = a.G + (int64_t)(m0 + r) * 4 * H;
                        g[u] = d_i; g[2 * H + u] = d_g; g[3 * H + u] = d_o;
                    }
                }
        }
    }
    PANEL_STAMP(11);
    __syncthreads();     // the staging images are complete
    // observe-embedding backward (tail): its weights and this wave's row inputs are fetched now, behind phase 6's streams
    constexpr int ONB = 2;                       // observables of the fused tail (static loops, see obs_embed.hip)
    ObsStage<8, 512> of1, of0;                   // e x e <= 64 x 64
    ObsStage<4, 512> ol1[ONB];                   // out x hid <= 64 x 32
    float t_f1 = 0.0f, t_cat = 0.0f, t_h = 0.0f;
    int t_oh = -1, t_jh = 0;
    int64_t t_hld = 0;
    if (OBS) {
        const ObsFusedArgs& oa = oin.a;
        of1.load(oin.P + oa.f1.w_off, oa.f1.rows * oa.f1.cols, tid);
        of0.load(oin.P + oa.f0.w_off, oa.f0.rows * oa.f0.cols, tid);
#pragma unroll
        for (int o = 0; o < ONB; ++o)
            if (o < oa.n_obs) ol1[o].load(oin.P + oa.l1[o].w_off, oa.l1[o].rows * oa.l1[o].cols, tid);
        const float* my_h = nullptr;
#pragma unroll
        for (int o = 0; o < ONB; ++o)
            if (o < oa.n_obs && lane >= oa.hoff[o] && lane < oa.hoff[o] + oa.hid[o]) {
                t_oh = o; t_jh = lane - oa.hoff[o];
                my_h = oa.obs_h[o]; t_hld = oa.ohid_ld[o];
            }
        const int tb = m0 + half * 4 + wave;      // waves 0..3 walk rows 4 half + wave
        if (wave < 4 && tb < a.B) {
            if (lane < oa.e_obs) {
                t_f1 = oin.f1[(int64_t)tb * oa.e_ld + lane];
                t_cat = oin.cat[(int64_t)tb * oa.e_ld + lane];
            }
            if (t_oh >= 0) t_h = my_h[(int64_t)tb * t_hld + t_jh];
        }
    }
    // ---------------- phase 6: dX[:, :e] = dG W_ih[:, :e]; K = this tile's 3 x 64 gate rows, split owner / helper ----------------
    {
        f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        // units (gate, q): rows gate_row(g) H + U0 + 16 q .. of W_ih, columns [0, 64); owner and helper take two q each
        const int KQ6 = 4 / NH, q0 = hp * KQ6;
        stream_units<NB>(
            3 * KQ6,
            [&](int i, float (&w)[16]) {
                const int g = i / KQ6, q = q0 + i - g * KQ6;
                load_unit(a.Wih + (int64_t)((g == 0 ? 0 : g + 1) * H + U0 + 16 * q) * a.ldw, a.ldw, voff, w);
            },
            [&](int i, const float (&w)[16]) {
                const int g = i / KQ6, q = q0 + i - g * KQ6;
                mma_unit(stg3 + (g * 8 + rl) * 72 + 16 * q + kl, 72, w, c0, c1);
            });
        float* pw = sP + wave * (PANEL_ROWS * PP) + lane;      // (free since the mixture phase read the tail's partials)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pw[i * PP] = c0[i];
            pw[(4 + i) * PP] = c1[i];
        }
    }
    PANEL_STAMP(12);
    // group sums of dG (this address's slot of gsum: LSTM bias and table-column gradients follow from them, aux_jobs.hpp):
    // one atomic per (gate, unit) per workgroup, issued last so that nothing in the kernel waits for them
    if (own) {
        const int u = U0 + lane;
        atomicAdd(a.gsum + u, gs_i);
        atomicAdd(a.gsum + 2 * H + u, gs_g);
        atomicAdd(a.gsum + 3 * H + u, gs_o);
    }
    __syncthreads();
    float* const oimg = sH;                               // LDS image of the embedding weights: sH .. sW2 are dead now
    float* const sDX = sH + 10240 + 8;                    // this workgroup's four dX rows [4][64]
    if (OBS) {
        const ObsFusedArgs& oa = oin.a;
        const int dummy = oa.lds_total;
        of1.store(oimg, oa.f1.lds_w, oa.f1.rows, oa.f1.cols, dummy, tid);
        of0.store(oimg, oa.f0.lds_w, oa.f0.rows, oa.f0.cols, dummy, tid);
#pragma unroll
        for (int o = 0; o < ONB; ++o)
            if (o < oa.n_obs) ol1[o].store(oimg, oa.l1[o].lds_w, oa.l1[o].rows, oa.l1[o].cols, dummy, tid);
    }
    {   // dX rows: the eight waves' partial tiles + the partner's half of the K range
        unsigned long long* const xd_own = a.xd + (int64_t)(panel * 2 + half) * (PANEL_ROWS * 64);
        const unsigned long long* const xd_other = a.xd + (int64_t)(panel * 2 + (half ^ 1)) * (PANEL_ROWS * 64);
        const int i = tid, r = i / e, c = i - r * e;      // 8 e <= 512: one element per thread
        if (i < PANEL_ROWS * e) {
            const float* pr = sP + r * PP + c;
            float sum = 0.0f;
#pragma unroll
            for (int wv = 0; wv < PANEL_WAVES; ++wv) sum += pr[wv * (PANEL_ROWS * PP)];
            gput(xd_own + i, sum, epoch);
            if (m0 + r < a.B && (r >> 2) == half) {      // (the other rows' sums are the partner's to write)
                const float theirs = gget(xd_other + i, epoch);
                const float dx = half == 0 ? sum + theirs : theirs + sum;
                a.dX[(int64_t)(m0 + r) * a.ldx + c] = dx;
                if (OBS) sDX[(r & 3) * 64 + c] = dx;
            }
        }
    }
    PANEL_STAMP(13);
    if (OBS) {
        // dz2 = dX * [E > 0]; dz1 = (Wf1^T dz2) * [f1 > 0]; dzc = (Wf0^T dz1) * [cat > 0]; dh_o = (W1_o^T dzc_o) * [h_o > 0]
        // (inference_network.py:132-139 backward; one wave per row, lane = unit, obs_embed.hip's walk)
        __syncthreads();
        const ObsFusedArgs& oa = oin.a;
        const int r = half * 4 + wave, tb = m0 + r;
        if (wave < 4 && tb < a.B) {
            const bool acte = lane < oa.e_obs;
            float dz2 = 0.0f;
            if (acte) dz2 = sE[r * PE + lane] > 0.0f ? sDX[wave * 64 + lane] : 0.0f;
            if (acte) oin.dE[(int64_t)tb * oa.e_ld + lane] = dz2;
            float dz1 = obs_dense_t(oimg, oa.f1, lane, acte, dz2, 0);
            dz1 = t_f1 > 0.0f ? dz1 : 0.0f;
            if (acte) oin.dF1[(int64_t)tb * oa.e_ld + lane] = dz1;
            float dzc = obs_dense_t(oimg, oa.f0, lane, acte, dz1, 0);
            dzc = t_cat > 0.0f ? dzc : 0.0f;
            if (acte) oin.dCat[(int64_t)tb * oa.e_ld + lane] = dzc;
            float dh = 0.0f;
            int co = 0;
#pragma unroll
            for (int o = 0; o < ONB; ++o) {
                if (o >= oa.n_obs) break;
                const bool acth = (t_oh == o);
                const float d = obs_dense_t(oimg, oa.l1[o], t_jh, acth, dzc, co);
                if (acth) dh = d;
                co += oa.out[o];
            }
            if (t_oh >= 0) oin.dHo0[(int64_t)t_oh * oin.dh_stride + (int64_t)tb * t_hld + t_jh] = t_h > 0.0f ? dh : 0.0f;
        }
        PANEL_STAMP(14);
    }
#undef PANEL_STAMP
}

}  // namespace

size_t panel_lds_bytes(int H, int hid, int n_out, int e) { return (size_t)panel_lds(H, hid, n_out, e).total * sizeof(float); }

// Which single-statement batches the panel kernel takes (everything else stays on the tile kernels).
bool panel_t1_supported(int kind, int H, int hid, int n_out, int e) {
    static const int env = getenv("PP_PANEL") ? atoi(getenv("PP_PANEL")) : 1;
    if (!env || deterministic_mode()) return false;
    if (kind != PP_HEAD_NORMAL_MIXTURE && kind != PP_HEAD_TRUNCNORMAL_MIXTURE && kind != PP_HEAD_POISSON_TN_MIXTURE) return false;
    if (H != 512) return false;      // (H = 1024: the LDS images do not fit; the tile kernels take it)
    // the pair hand-off spins on its partner workgroup (b, b + 8): both must become resident together. One 160 KB workgroup
    // per CU and in-order dispatch give that on a device with more than 16 free CUs; a CU-masked or partitioned device with
    // fewer takes the tile kernels instead of risking the bounded spin's trap
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        return n;
    }();
    if (cus < 32) return false;
    if (n_out % 3 != 0 || n_out < 3 || n_out > 48) return false;
    if (hid < 16 || hid > 576) return false;
    if (e < 16 || e > 64 || e % 16 != 0) return false;
    return panel_lds_bytes(H, hid, n_out, e) <= 160 * 1024;
}

template <int KIND, int SPLIT, bool OBS>
static int panel_launch(const PanelArgs& a, const PanelObs& po, size_t lds, hipStream_t st) {
    static thread_local bool configured = false;   // > 64 KB of dynamic LDS needs the opt-in once per kernel
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)panel_t1_kernel<KIND, SPLIT, OBS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) {
            set_error("panel_t1: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        configured = true;
    }
    const int panels = cdiv(a.B, PANEL_ROWS);
    const int grid = SPLIT == 2 ? 16 * cdiv(panels, 8) : panels;      // pairs are blocks (b, b + 8) of a 16-block window
    hipLaunchKernelGGL((panel_t1_kernel<KIND, SPLIT, OBS>), dim3(grid), dim3(512), lds, st, a, po);
    return 0;
}

// The launch uses two workgroups per panel (pairs are blocks (b, b + 8) of a 16-block window); up to 256 panels.
int panel_t1_split(int B, int H) { return (H == 512 && cdiv(B, PANEL_ROWS) <= 256) ? 2 : 0; }

// The embedding-backward tail: the fused embedding kernels' shapes with at most two observables, and the dead LDS buffers
// (sH .. sW2) must hold the weight image and four dX rows.
bool panel_obs_tail_ok(const pp_net* net, int H, int hid, int n_out, int e) {
    static const int env = 1;
    if (!env || !obs_fused_supported(net) || net->n_obs > 2 || net->e_obs != e) return false;
    const PanelLds L = panel_lds(H, hid, n_out, e);
    return L.sP - L.sH >= PANEL_OBS_LDS;
}

int panel_t1(int kind, const PanelArgs& a, hipStream_t st, const PanelObs* obs) {
    PP_CHECK_ARG(panel_t1_supported(kind, a.H, a.hid, a.n_out, a.e), "panel_t1: unsupported shape");
    PP_CHECK_ARG(a.ldx % 4 == 0 && a.ldw % 4 == 0 && a.lda1 % 4 == 0 && a.lda1 >= a.hid && a.lda1 <= ((a.hid + 15) & ~15) &&
                     a.lddy <= 64 && a.lddy >= a.n_out && a.K * 3 == a.n_out && a.e * PANEL_ROWS <= 512 && PANEL_ROWS * a.lda1 <= 512 * 5,
                 "panel_t1: bad leading dimensions");
    const size_t lds = panel_lds_bytes(a.H, a.hid, a.n_out, a.e);
    PP_CHECK_ARG(panel_t1_split(a.B, a.H) == 2 && a.xz && a.xd && a.epoch, "panel_t1: too many rows, or no hand-off buffers");
    if (obs) {
        const PanelLds L = panel_lds(a.H, a.hid, a.n_out, a.e);
        PP_CHECK_ARG(obs->a.n_obs <= 2 && obs->a.e_obs == a.e && obs->a.lds_total + 1 <= 10240 + 8 && L.sP - L.sH >= PANEL_OBS_LDS,
                     "panel_t1: the observe-embedding tail does not fit");
    }
    static const PanelObs none{};
#define PP_PANEL_GO(KIND)                                                             \
    do {                                                                              \
        if (obs) PP_TRY((panel_launch<KIND, 2, true>(a, *obs, lds, st)));             \
        else PP_TRY((panel_launch<KIND, 2, false>(a, none, lds, st)));                \
    } while (0)
    if (kind == PP_HEAD_NORMAL_MIXTURE) PP_PANEL_GO(0);
    else if (kind == PP_HEAD_TRUNCNORMAL_MIXTURE) PP_PANEL_GO(1);
    else PP_PANEL_GO(2);
#undef PP_PANEL_GO
    PP_LAUNCH_CHECK("panel_t1");
    return 0;
}

}  // namespace pp
