// 16-row panel kernel for batches of single-statement traces (T = 1: GaussianUnknownMean, BASELINE.json configs[1]) - the
// successor of panel.hip's 8-row kernel on the benchmark shape (H = 512, e = 64, head hidden width in (256, 272], <= 32 head
// outputs); everything else stays on panel.hip / the tile kernels.
//
// The data path of InferenceNetworkLSTM._loss + backward is row-local when every trace has one controlled statement
// (pyprob/nn/inference_network_lstm.py:136-220 with h0 = c0 = 0, :186-187):
//     E row -> G = E W_ih[:, :e]^T + bias(address) -> LSTM cell -> h -> z1 = relu(h W1^T + b1) -> y = z1 W2^T + b2
//       -> mixture log_prob / loss / dy -> dz1 = (dy W2) * [z1 > 0] -> dh = dz1 W1 -> cell backward -> dG -> dX = dG W_ih[:, :e]
// What bounded the 8-row kernel (profiles/r04p_panel_pmc_*.csv, r05a_panel_timeline.txt): every workgroup pulled 1.1 MB of
// weights through ONE CU's L1 miss queue (~25 B/clk) for 8 rows of reuse - 59 k of its 92 k cycles sat in the four weight
// streams - on v_mfma_f32_4x4x1 (11 cycles per 512 FLOP as issued), with per-k dword loads (16 VMEM + 32 MFMA instructions per
// 16-k unit and ~3 900 VALU instructions per wave of address and broadcast work around them).
//
// Geometry here: a PANEL is 16 rows (the M of v_mfma_f32_16x16x4_f32: full-rate fp32 MFMA with no padded rows) and is owned by
// FOUR workgroups (1 024 rows = 64 panels x 4 = 256 workgroups = one per CU); workgroup q owns hidden units [128 q, 128 q + 128)
// - a quarter of every weight matrix: 0.55 MB per CU - and wave w of it the 16 units [128 q + 16 w, + 16): the three gates of
// those units are three accumulator tiles of the wave, so the gate activations stay in its registers from the forward cell to
// the backward cell, and dh = dz1 W1 lands in the same registers. The B operands are FRAGMENT IMAGES (panel16_images.hpp):
// one coalesced 1 KB load feeds four MFMAs; a wave's 69 fragments are one static stream through a 16-deep register ring that
// runs AHEAD across the phases (the weights do not depend on the rows), so only the first fragments of the kernel wait for
// memory. Per wave: 280 MFMAs (9 k cycles of one SIMD's matrix pipe, two waves per SIMD), 69 loads, ~30 LDS reads.
//   phase 1  G = E W^T: 3 gate tiles x 4 units (48 MFMAs) -> cell on the accumulators -> h tile to LDS + memory
//   phase 2  partial z1 = h[:, own 128] W1^T[own 128, :]: 17 column tiles of K = 128; wave w takes tiles w and w + 8 over all K
//            and its own 16-k unit of tile 16 (68 MFMAs each); the four workgroups' partial sums cross through memory as
//            {value, tag} granules (handoff.hpp), every granule load of a wave in flight together, summed in ONE order on all
//            four sides -> identical z1 everywhere
//   phase 3  y = z1 W2^T (K split over the waves, 16-20 MFMAs), redundantly in the four workgroups (each needs dy of all rows)
//   mixture  sixteen lanes per row (lane = component; row reductions are DPP row operations), four rows per wave
//   phase 4  dz1 = (dy W2) * [z1 > 0] on the wave's tiles of phase 2 (z1 still in registers)
//   phase 5  dh = dz1 W1 for the wave's 16 units (68 MFMAs) -> cell backward in registers -> dG to memory, column sums
//   phase 6  partial dX = dG[:, own] W_ih[own, :e]: K = the wave's own 3 x 16 gate rows (A operand from a wave-private LDS
//            tile, no workgroup barrier), the eight waves' tiles meet in LDS, the four workgroups' in memory; every workgroup
//            finishes 4 of the panel's rows and walks them backward through the observe embedding (panel.hip's tail).
// The four workgroups of a panel are blocks b, b + 8, b + 16, b + 24 of a 32-block window: the same XCD under round-robin
// placement (speed only). Not bit-reproducible (float atomics of the loss and column sums): PP_DETERMINISTIC=1 keeps the tile
// path. Parity: tests/test_gpu_panel.py (every buffer and gradient against the tile path and the oracle, both panel kernels).
#include "common.hpp"
#include "panel16.hpp"
#include "handoff.hpp"

#include <algorithm>
#include <type_traits>

namespace pp {

namespace {

constexpr int PR = 16;                 // rows of a panel
constexpr int NWV = 8;                 // waves of a workgroup
constexpr int SS = 4;                  // workgroups per panel
constexpr int EE = 64, KE = EE / 16;   // observe-embedding width (K of the input product), its 16-k units
constexpr int NO = 32;                 // padded head outputs (two tiles)
// LDS pitches (floats) = 8 mod 16: the A operand's ds_read_b128 (lane: row l & 15, four k at 4 (l >> 4)) is conflict-free
constexpr int PE = EE + 8, PDY = NO + 8, PG = 16 + 8, PX = EE + 8;

// Shapes that depend on the LSTM width H (512: the benchmark network; 1024: BASELINE.json configs[4]'s per-rank network). A
// workgroup owns H / 4 hidden units, a wave UTW = H / 512 unit tiles of them; the head's hidden layer has NT 16-column tiles
// (hid in (16 (NT - 1), 16 NT]: 17 | 33 for K <= 10 mixture components), TPW = (NT - 1) / 8 of them per wave in phases 2 and
// 4 plus the K-split (phase 2) / wave-0 (phase 4) last tile.
template <int HH_>
struct P16 {
    static constexpr int HH = HH_;
    static constexpr int UT = HH / 16;            // unit tiles
    static constexpr int UW = HH / SS;            // hidden units of a workgroup
    static constexpr int UTW = UW / 128;          // unit tiles of a wave
    static constexpr int KL = UW / 16;            // 16-k units of the workgroup's h tile (K of phase 2)
    static constexpr int NT = HH == 512 ? 17 : 33;
    static constexpr int TPW = (NT - 1) / NWV;
    static constexpr int ZK = 16 * NT;            // padded hidden width of the head
    static constexpr int ZT = 16 * (NT - 1);      // first column of the last tile
    static constexpr int PH = UW + 8, PZ = ZK + 8;
    static constexpr int L_E = 0;
    static constexpr int L_H = L_E + PR * PE;
    static constexpr int L_Z = L_H + PR * PH;
    static constexpr int L_YP = L_Z + PR * PZ;                 // [wave][16][16] partial y tiles
    static constexpr int L_T = L_YP + NWV * PR * 16;           // [wave][16][16] partial last tile of head layer 1
    static constexpr int L_DY = L_T + NWV * PR * 16;
    static constexpr int L_DZ = L_DY + PR * PDY;
    static constexpr int L_DG = L_DZ + PR * PZ;                // [wave][3][16][PG] dG tiles of ONE unit tile (wave-private)
    static constexpr int L_END = L_DG + NWV * 3 * PR * PG;
    static constexpr int L_XP = L_Z;                           // [wave][16][PX] partial dX tiles: over sZ .. sDY (dead by then)
    static_assert(L_XP + NWV * PR * PX <= L_DZ, "the partial dX tiles must not reach the live dz1 tile");
    static constexpr int L_OIMG = L_DZ;                        // observe-embedding weight image of the tail: over sDZ + sDG
    static_assert(L_END - L_OIMG >= 10240 + 8, "the embedding image must fit the dead buffers");
    static_assert(L_END * 4 <= 160 * 1024, "LDS");
    // the fragment stream of a wave: phase 1 | 2 | 3 | 4 | per unit tile: 5, 6
    static constexpr int F1 = 3 * KE * UTW, F2 = KL * TPW + UTW, F3 = (2 * NT + NWV - 1) / NWV, F4 = 2 * TPW + 2, F5 = NT, F6 = 3 * KE;
    static constexpr int B1 = 0, B2 = B1 + F1, B3 = B2 + F2, B4 = B3 + F3, B5 = B4 + F4, FU = F5 + F6, FT = B5 + UTW * FU;
    static constexpr int RING = HH == 512 ? 24 : 16;
};

constexpr float kFp32Eps = 1.1920928955078125e-07f;
constexpr float kHalfLog2Pi = 0.91893853320467274178f;
constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kInvSqrt2Pi = 0.39894228040143267794f;
constexpr float kLogEps = -18.420680743952367f;

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ float std_cdf_(float x) { return 0.5f * (1.0f + erff(x * kInvSqrt2)); }
__device__ __forceinline__ float std_pdf_(float x) { return kInvSqrt2Pi * expf(-0.5f * x * x); }

__device__ __forceinline__ void wave_sync_lds() {   // order this wave's LDS writes before its LDS reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// reductions over the 16 lanes of a DPP row (every lane of the row ends with the result)
__device__ __forceinline__ float row_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x124>(v);
    v += dpp_mov<0x128>(v);
    return v;
}
__device__ __forceinline__ float row_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x124>(v));
    v = fmaxf(v, dpp_mov<0x128>(v));
    return v;
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ f32x4 mma4(const f32x4& a, const f32x4& b, f32x4 c) {      // the four K steps of one fragment
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
}
#define PP_MMA(S, A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x4f32((A)[S], (B)[S], C, 0, 0, 0)

// ---- the fragment stream of a wave ------------------------------------------------------------------------------------
template <int HH_>
struct FragPtrs {
    using T = P16<HH_>;
    const f32x4 *p1, *p2, *p3, *p4, *p5, *p6;      // per-phase bases, this lane's float4 of fragment 0
    int wave;
    template <int G>
    __device__ __forceinline__ const f32x4* at() const {
        if constexpr (G < T::B2) {                 // (16-k unit, unit tile of the wave, gate)
            constexpr int f = G - T::B1;
            return p1 + ((f / (3 * T::UTW)) * (T::UT * 3) + (f % (3 * T::UTW))) * 64;
        } else if constexpr (G < T::B3) {          // tiles wave + 8 t per 16-k unit, then the last tile for the wave's own units
            constexpr int f = G - T::B2;
            if constexpr (f < T::KL * T::TPW) return p2 + ((f / T::TPW) * T::NT + 8 * (f % T::TPW)) * 64;
            else return p2 + ((T::UTW * wave + (f - T::KL * T::TPW)) * T::NT + (T::NT - 1) - wave) * 64;
        } else if constexpr (G < T::B4) {          // items wave + 8 f of the 2 NT (unit, tile) pairs of head layer 2
            constexpr int f = G - T::B3;
            return p3 + min(8 * f, 2 * T::NT - 1 - wave) * 64;
        } else if constexpr (G < T::B5) {          // (tile wave + 8 t | last, unit f & 1)
            constexpr int f = G - T::B4;
            return p4 + ((f & 1) * T::NT + ((f >> 1) < T::TPW ? 8 * (f >> 1) : (T::NT - 1) - wave)) * 64;
        } else {                                   // per unit tile j of the wave: NT units of dh, then (gate, column tile) of dX
            constexpr int j = (G - T::B5) / T::FU, f = (G - T::B5) % T::FU;
            if constexpr (f < T::F5) return p5 + (f * T::UT + j) * 64;
            else return p6 + (j * 3 * KE + (f - T::F5)) * 64;
        }
    }
};

// KIND: head kind (0 Normal mixture, 1 TruncatedNormal mixture in a Uniform prior, 2 Poisson head), kernels.hip.
// FLAGS (PP_PANEL_HANDOFF=flag, A/B of the hand-off protocol INSIDE this kernel; default: the {value, tag} granules of handoff.hpp):
// the partial sums cross as 4-byte payloads - the first half of a workgroup's slot of a.xz / a.xd - behind ONE flag word per producer
// wave (s_waitcnt vmcnt(0), then a relaxed system-scope store of the step's tag; the flag words lie in the second half of the
// workgroup's xz slot: [0, 8) the tile sums of phase 2, [8, 12) the last tile's elements, [16, 24) the dX sums). A consumer wave
// polls the flags of the partner waves of ITS index (they wrote what it needs), then loads the payloads: half the bytes, two
// dependent round trips instead of one (tools/micro/handoff_probe.hip: +0.2 us median per exchange in isolation).
template <int HH_, int KIND, bool OBS, bool FLAGS = false>
__global__ __launch_bounds__(512) void panel16_kernel(const Panel16Args ain, const PanelObs oin) {
    using T = P16<HH_>;
    constexpr int HH = T::HH, UT = T::UT, UTW = T::UTW, NT = T::NT, TPW = T::TPW, ZK = T::ZK, ZT = T::ZT, PH = T::PH, PZ = T::PZ;
    constexpr int RING = T::RING, FT = T::FT, B1 = T::B1, B2 = T::B2, B3 = T::B3, B4 = T::B4, B5 = T::B5;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const PanelArgs a = ain.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = blockIdx.x;
    const int q = (bx >> 3) & (SS - 1);                   // this workgroup's quarter of the hidden units
    const int panel = (bx >> 5) * 8 + (bx & 7);
    const int m0 = panel * PR;
    if (m0 >= a.B) return;                                // (all four workgroups of the panel)
    const int c = lane & 15, g = lane >> 4;               // C / B operand: column c, rows 4 g + i; A operand: row c, k group g
    const int hid = a.hid, n_out = a.n_out, K = a.K;
    const int ut0 = UTW * (8 * q + wave);                 // this wave's first unit tile (it owns UTW consecutive ones)
    const int epoch = *a.epoch;
    auto put32 = [](float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    auto get32 = [](const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    auto flag_set = [&](unsigned* f) {      // (the whole wave: its payload stores are acknowledged before the flag leaves)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(f, gtag(epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    auto flags_wait = [&](const unsigned* f0, const unsigned* f1, const unsigned* f2) {
        const unsigned tag = gtag(epoch);
        int spins = 0;
        while (true) {
            const unsigned x0 = __hip_atomic_load(f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned x1 = __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned x2 = __hip_atomic_load(f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (x0 == tag && x1 == tag && x2 == tag) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 20)) __builtin_trap();
        }
    };
    // (flag words of workgroup (panel, quarter): behind the PR * ZK payload floats of its xz slot)
    auto flags_of = [&](int quarter) {
        return reinterpret_cast<unsigned*>(reinterpret_cast<float*>(a.xz + (int64_t)(panel * SS + quarter) * (PR * ZK)) + PR * ZK);
    };
    float* const sE = smem + T::L_E;
    float* const sH = smem + T::L_H;
    float* const sZ = smem + T::L_Z;
    float* const sYP = smem + T::L_YP;
    float* const sT = smem + T::L_T;
    float* const sDY = smem + T::L_DY;
    float* const sDZ = smem + T::L_DZ;
    float* const sDGw = smem + T::L_DG + wave * (3 * PR * PG);
    float* const sXP = smem + T::L_XP;
    const int dbg_slot = (bx == 0 ? 0 : (bx == 77 ? 1 : -1));
#define P16_STAMP(k)                                                                                            \
    do {                                                                                                        \
        if (a.dbg && dbg_slot >= 0 && lane == 0 && (wave == 0 || wave == 5))                                    \
            a.dbg[(dbg_slot * 2 + (wave == 5 ? 1 : 0)) * 16 + (k)] = clock64();                                 \
    } while (0)
    P16_STAMP(0);

    FragPtrs<HH_> fp;
    fp.wave = wave;
    fp.p1 = reinterpret_cast<const f32x4*>(ain.img[0]) + (ut0 * 3) * 64 + lane;
    fp.p2 = reinterpret_cast<const f32x4*>(ain.img[1]) + ((T::KL * q) * NT + wave) * 64 + lane;
    fp.p3 = reinterpret_cast<const f32x4*>(ain.img[2]) + wave * 64 + lane;
    fp.p4 = reinterpret_cast<const f32x4*>(ain.img[3]) + wave * 64 + lane;
    fp.p5 = reinterpret_cast<const f32x4*>(ain.img[4]) + ut0 * 64 + lane;
    fp.p6 = reinterpret_cast<const f32x4*>(ain.img[5]) + (ut0 * 3 * KE) * 64 + lane;
    // ---------------- staging: E rows (their loads go out first), then the head of the fragment stream ----------------
    f32x4 ev = {0, 0, 0, 0};
    if (tid < PR * (EE / 4))
        ev = *reinterpret_cast<const f32x4*>(a.X + (int64_t)min(m0 + (tid >> 4), a.B - 1) * a.ldx + 4 * (tid & 15));
    f32x4 ring[RING];
#define P16_ISSUE(G)                                                           \
    do {                                                                       \
        if constexpr ((G) < FT) ring[(G) % RING] = *fp.template at<(G)>();     \
    } while (0)
    static_for<0, RING>([&](auto GG) {
        constexpr int g0 = decltype(GG)::value;
        P16_ISSUE(g0);
    });
    // small vectors of later phases, fetched now (a first touch from memory costs ~2 000 cycles where it is needed): the LSTM
    // bias of this lane's units, b1 of its head columns, and - waves 0..3, row 4 wave + g - the row's value, prior and b2
    float bias[UTW][3];
#pragma unroll
    for (int j = 0; j < UTW; ++j)
#pragma unroll
        for (int y = 0; y < 3; ++y) bias[j][y] = a.AB[(y == 0 ? 0 : y + 1) * HH + 16 * (ut0 + j) + c];
    float b1v[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) b1v[t] = a.b1[16 * (wave + 8 * t) + c];
    const float b16 = a.b1[min(ZT + (tid & 15), hid - 1)];      // (threads < 256: column ZT + tid % 16 of the last tile)
    const int mrow = min(m0 + 4 * (wave & 3) + g, a.B - 1);
    const float m_v = a.value[mrow], m_pa = a.prior[2 * mrow], m_pb = a.prior[2 * mrow + 1];
    const int mcol = min(c, K - 1);
    const float m_b2mu = a.b2[mcol], m_b2sd = a.b2[K + mcol], m_b2z = a.b2[2 * K + mcol];
    if (tid < PR * (EE / 4)) *reinterpret_cast<f32x4*>(sE + (tid >> 4) * PE + 4 * (tid & 15)) = ev;
    for (int i = tid; i < PR * PDY; i += 512) sDY[i] = 0.0f;
    __syncthreads();
    P16_STAMP(1);

    // ---------------- phase 1: G = E W_ih[:, :e]^T + bias, LSTM cell (c0 = 0) ----------------
    float gi[UTW][4], gg[UTW][4], go[UTW][4], tc[UTW][4];        // gate activations and tanh(c) of (row 4 g + i, unit)
    {
        f32x4 acc[UTW][3];
#pragma unroll
        for (int j = 0; j < UTW; ++j)
#pragma unroll
            for (int y = 0; y < 3; ++y) acc[j][y] = f32x4{0, 0, 0, 0};
        static_for<0, KE>([&](auto KQ) {
            constexpr int kq = decltype(KQ)::value;
            constexpr int G0 = B1 + 3 * UTW * kq;
            const f32x4 av = *reinterpret_cast<const f32x4*>(sE + c * PE + 16 * kq + 4 * g);
            static_for<0, 4>([&](auto SQ) {
                constexpr int sq = decltype(SQ)::value;
                static_for<0, 3 * UTW>([&](auto FF) {
                    constexpr int ff = decltype(FF)::value;
                    acc[ff / 3][ff % 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sq], ring[(G0 + ff) % RING][sq], acc[ff / 3][ff % 3], 0, 0, 0);
                });
            });
            static_for<0, 3 * UTW>([&](auto FF) {
                constexpr int ff = decltype(FF)::value;
                P16_ISSUE(G0 + ff + RING);
            });
        });
        // cell: sigmoid and tanh through v_exp_f32 / v_rcp_f32 (absolute error ~1e-7, asserted in tests/test_gpu_panel.py)
#pragma unroll
        for (int j = 0; j < UTW; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * g + i;
                const float vi = fast_sigmoid(acc[j][0][i] + bias[j][0]);
                const float vg = fast_tanh(acc[j][1][i] + bias[j][1]);
                const float vo = fast_sigmoid(acc[j][2][i] + bias[j][2]);
                const float tcv = fast_tanh(vi * vg);
                const float h = vo * tcv;
                gi[j][i] = vi; gg[j][i] = vg; go[j][i] = vo; tc[j][i] = tcv;
                sH[r * PH + 16 * (UTW * wave + j) + c] = h;
                if (m0 + r < a.B) a.Hs[(int64_t)(m0 + r) * HH + 16 * (ut0 + j) + c] = h;
            }
    }
    P16_STAMP(2);
    __syncthreads();     // the workgroup's h tile [16][H / 4] is complete
    P16_STAMP(3);

    // ---------------- phase 2: partial z1 = h[:, own units] W1^T; tiles wave + 8 t (all K), the last tile (own units) ----------------
    float zt[TPW][4];    // z1 of (row 4 g + i, column 16 (wave + 8 t) + c); the last tile's z1 lives in sZ
    {
        f32x4 acc[TPW + 1];
#pragma unroll
        for (int t = 0; t <= TPW; ++t) acc[t] = f32x4{0, 0, 0, 0};
        static_for<0, T::KL>([&](auto KLL) {
            constexpr int kl = decltype(KLL)::value;
            constexpr int G0 = B2 + TPW * kl;
            const f32x4 av = *reinterpret_cast<const f32x4*>(sH + c * PH + 16 * kl + 4 * g);
            static_for<0, 4>([&](auto SQ) {
                constexpr int sq = decltype(SQ)::value;
                static_for<0, TPW>([&](auto TT) {
                    constexpr int t = decltype(TT)::value;
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sq], ring[(G0 + t) % RING][sq], acc[t], 0, 0, 0);
                });
            });
            static_for<0, TPW>([&](auto TT) {
                constexpr int t = decltype(TT)::value;
                P16_ISSUE(G0 + t + RING);
            });
        });
        static_for<0, UTW>([&](auto JJ) {
            constexpr int j = decltype(JJ)::value;
            const f32x4 av = *reinterpret_cast<const f32x4*>(sH + c * PH + 16 * (UTW * wave + j) + 4 * g);
            acc[TPW] = mma4(av, ring[(B2 + T::KL * TPW + j) % RING], acc[TPW]);
            P16_ISSUE(B2 + T::KL * TPW + j + RING);
        });
        P16_STAMP(4);
        // the partial sums leave for the three partner workgroups: {value, tag} granules [panel][quarter][row][ZK]
        unsigned long long* const xz_own = a.xz + (int64_t)(panel * SS + q) * (PR * ZK);
        float* const xzf_own = reinterpret_cast<float*>(xz_own);      // FLAGS: [row][ZK] payloads
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sT[(wave * PR + 4 * g + i) * 16 + c] = acc[TPW][i];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                if constexpr (FLAGS) put32(xzf_own + (4 * g + i) * ZK + 16 * (wave + 8 * t) + c, acc[t][i]);
                else gput(xz_own + (4 * g + i) * ZK + 16 * (wave + 8 * t) + c, acc[t][i], epoch);
            }
        }
        if constexpr (FLAGS) flag_set(flags_of(q) + wave);
        __syncthreads();     // the last tile's eight partial tiles
        // the last tile (the K split over the waves): one element per thread of the first four waves - sum of the eight partial
        // tiles, published like the others; its partner sums ride in the first batch of loads below
        const int er = tid >> 4, ec = tid & 15;          // (tid < 256: row er, column ZT + ec)
        float own16 = 0.0f;
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < NWV; ++w) own16 += sT[(w * PR + er) * 16 + ec];
            if constexpr (FLAGS) put32(xzf_own + er * ZK + ZT + ec, own16);
            else gput(xz_own + er * ZK + ZT + ec, own16, epoch);
        }
        if constexpr (FLAGS) {
            if (wave < 4) flag_set(flags_of(q) + 8 + wave);      // (tid < 256: the first four waves)
        }
        P16_STAMP(5);
        // the three partners' partial sums, two tiles (+ the last tile's element) per batch: every granule load of a batch is in
        // flight together; a pass is repeated until all of a lane's granules carry this step's tag. z1 = relu(b1 + quarter 0 +
        // quarter 1 + quarter 2 + quarter 3): the SAME order in all four workgroups
        const unsigned tag = gtag(epoch);
        const unsigned long long* pb[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) pb[s] = a.xz + (int64_t)(panel * SS + (s + (s >= q ? 1 : 0))) * (PR * ZK);
        auto four = [&](float own, float x0, float x1, float x2) {
            const float v0 = q == 0 ? own : x0;
            const float v1 = q == 1 ? own : (q > 1 ? x1 : x0);
            const float v2 = q == 2 ? own : (q > 2 ? x2 : x1);
            const float v3 = q == 3 ? own : x2;
            return ((v0 + v1) + v2) + v3;
        };
        if constexpr (FLAGS) {      // the partner waves of this wave's index have published everything the batches below load
            const int q0 = q == 0 ? 1 : 0, q1 = q <= 1 ? 2 : 1, q2 = q <= 2 ? 3 : 2;
            flags_wait(flags_of(q0) + wave, flags_of(q1) + wave, flags_of(q2) + wave);
            if (wave < 4) flags_wait(flags_of(q0) + 8 + wave, flags_of(q1) + 8 + wave, flags_of(q2) + 8 + wave);
        }
        static_for<0, TPW / 2>([&](auto BB) {
            constexpr int bt = 2 * decltype(BB)::value;       // tiles bt and bt + 1 of the wave
            // (waves 4..7 and the later batches have no element of the last tile: a duplicate of a granule waited for anyway)
            const int o16 = (bt == 0 && tid < 256) ? er * ZK + ZT + ec : (4 * g) * ZK + 16 * (wave + 8 * bt) + c;
            float xs[3][9];
            int spins = 0;
            if constexpr (FLAGS) {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const float* pf = reinterpret_cast<const float*>(pb[s]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ro = (4 * g + i) * ZK + c;
                        xs[s][i] = get32(pf + ro + 16 * (wave + 8 * bt));
                        xs[s][4 + i] = get32(pf + ro + 16 * (wave + 8 * (bt + 1)));
                    }
                    xs[s][8] = get32(pf + o16);
                }
            } else
            while (true) {
                unsigned long long x[3][9];
#pragma unroll
                for (int s = 0; s < 3; ++s) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ro = (4 * g + i) * ZK + c;
                        x[s][i] = __hip_atomic_load(pb[s] + ro + 16 * (wave + 8 * bt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        x[s][4 + i] = __hip_atomic_load(pb[s] + ro + 16 * (wave + 8 * (bt + 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    x[s][8] = __hip_atomic_load(pb[s] + o16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                bool ok = true;
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int e = 0; e < 9; ++e) {
                        ok = ok && ((unsigned)(x[s][e] >> 32) == tag);
                        xs[s][e] = __uint_as_float((unsigned)x[s][e]);
                    }
                if (ok) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 20)) __builtin_trap();
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int j = 16 * (wave + 8 * (bt + tt)) + c;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 4 * tt + i, r = 4 * g + i;
                    const float z = relu_keep_nan(four(acc[bt + tt][i], xs[0][e], xs[1][e], xs[2][e]) + b1v[bt + tt]);      // (j < ZT < hid)
                    zt[bt + tt][i] = z;
                    sZ[r * PZ + j] = z;
                    if (g == q && m0 + r < a.B) a.A1[(int64_t)(m0 + r) * a.lda1 + j] = z;
                }
            }
            if (bt == 0 && tid < 256) {
                const int j = ZT + ec;
                const float z = j < hid ? relu_keep_nan(four(own16, xs[0][8], xs[1][8], xs[2][8]) + b16) : 0.0f;
                sZ[er * PZ + j] = z;
                if ((er >> 2) == q && m0 + er < a.B && j < a.lda1) a.A1[(int64_t)(m0 + er) * a.lda1 + j] = z;
            }
        });
        P16_STAMP(6);
    }
    __syncthreads();     // z1 [16][ZK] complete
    P16_STAMP(7);

    // ---------------- phase 3: y = z1 W2^T; item wave + 8 f of the 2 NT (unit, tile) pairs: this wave's tile is wave & 1 ----------------
    {
        f32x4 y0 = {0, 0, 0, 0}, y1 = {0, 0, 0, 0};
        static_for<0, T::F3>([&](auto F) {
            constexpr int f = decltype(F)::value;
            const int unit = (wave >> 1) + 4 * f;
            if (unit < NT) {         // (wave-uniform)
                const f32x4 av = *reinterpret_cast<const f32x4*>(sZ + c * PZ + 16 * unit + 4 * g);
                if constexpr (f & 1) y1 = mma4(av, ring[(B3 + f) % RING], y1);
                else y0 = mma4(av, ring[(B3 + f) % RING], y0);
            }
            P16_ISSUE(B3 + f + RING);
        });
#pragma unroll
        for (int i = 0; i < 4; ++i) sYP[(wave * PR + 4 * g + i) * 16 + c] = y0[i] + y1[i];
    }
    __syncthreads();
    P16_STAMP(8);

    // ---------------- mixture log_prob, loss, d lp / d y: waves 0..3, row 4 wave + g, one mixture component per lane ----------------
    if (wave < 4) {
        const int r = 4 * wave + g;
        const bool rowok = m0 + r < a.B;
        const bool mine = rowok && wave == q;               // this workgroup writes the row's shared outputs
        const int gr = min(m0 + r, a.B - 1);
        const bool comp = c < K;
        // y[o] = b2[o] + the four partial tiles of column tile o >> 4 (waves of that parity)
        auto yof = [&](int o) {
            const float* pr = sYP + (((o >> 4) & 1) * PR + r) * 16 + (o & 15);
            return ((pr[0] + pr[2 * PR * 16]) + pr[4 * PR * 16]) + pr[6 * PR * 16];
        };
        float ymu = 0.0f, ysd = 0.0f, yz = -INFINITY;
        if (comp) {
            ymu = m_b2mu + yof(c);
            ysd = m_b2sd + yof(K + c);
            yz = m_b2z + yof(2 * K + c);
        }
        const float v = m_v, pa = m_pa, pb = m_pb;
        const float zmax = row_max(yz);
        const float ex = comp ? expf(yz - zmax) : 0.0f;
        const float pi = ex / row_sum(ex);
        const float ps = row_sum(pi);
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
        const float amax = row_max(al);
        float lp = amax;
        if (amax > -INFINITY) lp = amax + logf(row_sum(comp ? expf(al - amax) : 0.0f));
        if (row_sum((comp && al != al) ? 1.0f : 0.0f) > 0.0f) lp = NAN;   // NaN in a component poisons the logsumexp
        const bool rescued = (lp == -INFINITY);
        const bool bad = !rescued && !isfinite(lp);
        if (mine && c == 0) {
            if (a.lp_out) a.lp_out[gr] = lp;
            atomicAdd(a.loss_acc + 32 * ((blockIdx.x * 4 + g) & 63), rescued ? -kLogEps : -lp);
            if (bad) atomicOr(a.flag, 1);
        }
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
        const bool live = rowok && !(rescued || bad);
        {
            const float resp = (comp && live) ? expf(al - lp) : 0.0f;
            const bool in = (p >= kFp32Eps) && (p <= 1.0f - kFp32Eps);
            float dp = (comp && in) ? resp / p : 0.0f;
            const float dpp = row_sum(dp * p);
            dp = comp ? (dp - dpp) / ps : 0.0f;
            const float dpipi = row_sum(dp * pi);
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
            sDY[r * PDY + c] = d0; sDY[r * PDY + K + c] = d1; sDY[r * PDY + 2 * K + c] = d2;
            if (mine) {
                float* dy = a.DY + (int64_t)gr * a.lddy;
                dy[c] = d0; dy[K + c] = d1; dy[2 * K + c] = d2;
            }
        }
        if (mine)      // pad columns of the row's dy
            for (int o = n_out + c; o < a.lddy; o += 16) a.DY[(int64_t)gr * a.lddy + o] = 0.0f;
    }
    __syncthreads();
    P16_STAMP(9);

    // ---------------- phase 4: dz1 = (dy W2) * [z1 > 0] on the tiles of phase 2 ----------------
    {
        f32x4 acc[TPW + 1];
#pragma unroll
        for (int t = 0; t <= TPW; ++t) acc[t] = f32x4{0, 0, 0, 0};
        static_for<0, T::F4>([&](auto F) {
            constexpr int f = decltype(F)::value;
            if ((f >> 1) < TPW || wave == 0) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(sDY + c * PDY + 16 * (f & 1) + 4 * g);
                acc[f >> 1] = mma4(av, ring[(B4 + f) % RING], acc[f >> 1]);
            }
            P16_ISSUE(B4 + f + RING);
        });
#pragma unroll
        for (int t = 0; t <= TPW; ++t) {
            if (t == TPW && wave != 0) break;
            const int j = t < TPW ? 16 * (wave + 8 * t) + c : ZT + c;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * g + i;
                const float zv = t < TPW ? zt[t < TPW ? t : 0][i] : sZ[r * PZ + j];
                const float d = (j < hid && zv > 0.0f) ? acc[t][i] : 0.0f;
                sDZ[r * PZ + j] = d;
                if (g == q && m0 + r < a.B && j < a.lda1) a.dZ1[(int64_t)(m0 + r) * a.lda1 + j] = d;
            }
        }
    }
    __syncthreads();     // dz1 [16][ZK] complete
    P16_STAMP(10);

    // observe-embedding backward (tail): its weights and the masks of this lane's outputs are fetched behind phase 6
    constexpr int ONB = 2;
    ObsStage<8, 512> of1, of0;
    ObsStage<4, 512> ol1[ONB];
    // masks of the tail's three layers for this lane's outputs: waves 0..3 = column tile `wave` of the two 64-wide layers, lanes
    // 0..15 (row group 0 of the C layout) = the workgroup's four rows; waves o < n_obs: the hidden units of observable o
    float m_f1[4] = {0, 0, 0, 0}, m_cat[4] = {0, 0, 0, 0}, m_h[4] = {0, 0, 0, 0};
    f32x4 xacc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};      // partial dX tiles of this wave
    // ---------------- phases 5 and 6, one unit tile of the wave at a time ----------------
    // phase 5: dh = dz1 W1 for the tile's 16 units, cell backward in registers; phase 6: partial dX[:, :e] += dG[:, tile's gate rows]
    // W_ih[those rows, :e] with the A operand from a wave-private LDS tile (no workgroup barrier between the two)
    static_for<0, UTW>([&](auto JJ) {
        constexpr int j = decltype(JJ)::value;
        constexpr int G5 = B5 + j * T::FU, G6 = G5 + T::F5;
        f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
        static_for<0, T::F5 / 2>([&](auto FP) {          // two fragments at a time: their MFMA chains interleave
            constexpr int f = 2 * decltype(FP)::value;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(sDZ + c * PZ + 16 * f + 4 * g);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(sDZ + c * PZ + 16 * (f + 1) + 4 * g);
            const f32x4 b0 = ring[(G5 + f) % RING], b1 = ring[(G5 + f + 1) % RING];
            PP_MMA(0, a0, b0, d0); PP_MMA(0, a1, b1, d1);
            PP_MMA(1, a0, b0, d0); PP_MMA(1, a1, b1, d1);
            PP_MMA(2, a0, b0, d0); PP_MMA(2, a1, b1, d1);
            PP_MMA(3, a0, b0, d0); PP_MMA(3, a1, b1, d1);
            P16_ISSUE(G5 + f + RING); P16_ISSUE(G5 + f + 1 + RING);
        });
        static_assert(T::F5 % 2 == 1, "one fragment left");
        {
            const f32x4 av = *reinterpret_cast<const f32x4*>(sDZ + c * PZ + 16 * (T::F5 - 1) + 4 * g);
            d0 = mma4(av, ring[(G5 + T::F5 - 1) % RING], d0);
            P16_ISSUE(G5 + T::F5 - 1 + RING);
        }
        const int u = 16 * (ut0 + j) + c;                // this lane's hidden unit (C layout)
        float gs_i = 0.f, gs_g = 0.f, gs_o = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * g + i;
            const float dh = d0[i] + d1[i];
            const float vi = gi[j][i], vg = gg[j][i], vo = go[j][i], tcv = tc[j][i];
            const float dc = dh * vo * (1.0f - tcv * tcv);
            float d_i = dc * vg * vi * (1.0f - vi);
            float d_g = dc * vi * (1.0f - vg * vg);
            float d_o = dh * tcv * vo * (1.0f - vo);
            if (m0 + r >= a.B) { d_i = 0.0f; d_g = 0.0f; d_o = 0.0f; }
            gs_i += d_i; gs_g += d_g; gs_o += d_o;
            sDGw[(0 * PR + r) * PG + c] = d_i;
            sDGw[(1 * PR + r) * PG + c] = d_g;
            sDGw[(2 * PR + r) * PG + c] = d_o;
            if (m0 + r < a.B) {
                float* gp = a.G + (int64_t)(m0 + r) * 4 * HH;
                gp[u] = d_i; gp[2 * HH + u] = d_g; gp[3 * HH + u] = d_o;
            }
        }
        // column sums of dG over the panel's rows (this address's slot of gsum: LSTM bias and table-column gradients follow
        // from them, aux_jobs.hpp): the four row groups of a column meet through the LDS crossbar, one atomic per (gate, unit)
        gs_i += __shfl_xor(gs_i, 16, 64); gs_g += __shfl_xor(gs_g, 16, 64); gs_o += __shfl_xor(gs_o, 16, 64);
        gs_i += __shfl_xor(gs_i, 32, 64); gs_g += __shfl_xor(gs_g, 32, 64); gs_o += __shfl_xor(gs_o, 32, 64);
        if (g == 0) {
            atomicAdd(a.gsum + u, gs_i);
            atomicAdd(a.gsum + 2 * HH + u, gs_g);
            atomicAdd(a.gsum + 3 * HH + u, gs_o);
        }
        wave_sync_lds();     // this wave's dG tiles (A operand of phase 6) are its own
        if constexpr (j == 0) {
            P16_STAMP(11);
            if (OBS) {
                const ObsFusedArgs& oa = oin.a;
                of1.load(oin.P + oa.f1.w_off, oa.f1.rows * oa.f1.cols, tid);
                of0.load(oin.P + oa.f0.w_off, oa.f0.rows * oa.f0.cols, tid);
#pragma unroll
                for (int o = 0; o < ONB; ++o)
                    if (o < oa.n_obs) ol1[o].load(oin.P + oa.l1[o].w_off, oa.l1[o].rows * oa.l1[o].cols, tid);
                if (wave < 4 && g == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int tb = min(m0 + 4 * q + i, a.B - 1);
                        m_f1[i] = oin.f1[(int64_t)tb * oa.e_ld + 16 * wave + c];
                        m_cat[i] = oin.cat[(int64_t)tb * oa.e_ld + 16 * wave + c];
                    }
#pragma unroll
                    for (int o = 0; o < ONB; ++o)
                        if (o == wave && o < oa.n_obs && c < oa.hid[o]) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) m_h[i] = oa.obs_h[o][(int64_t)min(m0 + 4 * q + i, a.B - 1) * oa.ohid_ld[o] + c];
                        }
                }
            }
        }
        static_for<0, 3>([&](auto Y) {
            constexpr int y = decltype(Y)::value;
            const f32x4 av = *reinterpret_cast<const f32x4*>(sDGw + (y * PR + c) * PG + 4 * g);
            const f32x4 b0 = ring[(G6 + 4 * y) % RING], b1 = ring[(G6 + 4 * y + 1) % RING], b2 = ring[(G6 + 4 * y + 2) % RING],
                        b3 = ring[(G6 + 4 * y + 3) % RING];
            PP_MMA(0, av, b0, xacc[0]); PP_MMA(0, av, b1, xacc[1]); PP_MMA(0, av, b2, xacc[2]); PP_MMA(0, av, b3, xacc[3]);
            PP_MMA(1, av, b0, xacc[0]); PP_MMA(1, av, b1, xacc[1]); PP_MMA(1, av, b2, xacc[2]); PP_MMA(1, av, b3, xacc[3]);
            PP_MMA(2, av, b0, xacc[0]); PP_MMA(2, av, b1, xacc[1]); PP_MMA(2, av, b2, xacc[2]); PP_MMA(2, av, b3, xacc[3]);
            PP_MMA(3, av, b0, xacc[0]); PP_MMA(3, av, b1, xacc[1]); PP_MMA(3, av, b2, xacc[2]); PP_MMA(3, av, b3, xacc[3]);
            P16_ISSUE(G6 + 4 * y + RING); P16_ISSUE(G6 + 4 * y + 1 + RING); P16_ISSUE(G6 + 4 * y + 2 + RING); P16_ISSUE(G6 + 4 * y + 3 + RING);
        });
    });
    {
        float* pw = sXP + wave * (PR * PX);       // (sZ .. sDY are dead: every wave passed the barrier behind phase 4)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) pw[(4 * g + i) * PX + 16 * t + c] = xacc[t][i];
    }
    P16_STAMP(12);
    __syncthreads();
    float* const oimg = smem + T::L_OIMG;         // sDZ and the dG tiles are dead now
    float* const sT0 = smem + T::L_H;                // A operands of the tail's layers: [16][PX], rows 0..3 = this workgroup's rows
    float* const sT1 = smem + T::L_Z;                // (over the partial dX tiles, behind a barrier)
    float* const sT2 = smem + T::L_Z + PR * PX;
    if (OBS) {
        const ObsFusedArgs& oa = oin.a;
        const int dummy = oa.lds_total;
        of1.store(oimg, oa.f1.lds_w, oa.f1.rows, oa.f1.cols, dummy, tid);
        of0.store(oimg, oa.f0.lds_w, oa.f0.rows, oa.f0.cols, dummy, tid);
#pragma unroll
        for (int o = 0; o < ONB; ++o)
            if (o < oa.n_obs) ol1[o].store(oimg, oa.l1[o].lds_w, oa.l1[o].rows, oa.l1[o].cols, dummy, tid);
    }
    {   // dX: the eight waves' partial tiles, then the three partner workgroups' sums for this workgroup's four rows
        unsigned long long* const xd_own = a.xd + (int64_t)(panel * SS + q) * (PR * EE);
        const int col = tid & 63;
        float mine = 0.0f;
        int myrow = -1;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int r = (tid >> 6) + 8 * jj;
            const float* pr = sXP + r * PX + col;
            float sum = 0.0f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) sum += pr[w * (PR * PX)];
            if constexpr (FLAGS) put32(reinterpret_cast<float*>(xd_own) + r * EE + col, sum);
            else gput(xd_own + r * EE + col, sum, epoch);
            if ((r >> 2) == q) { mine = sum; myrow = r; }
        }
        if constexpr (FLAGS) flag_set(flags_of(q) + 16 + wave);
        if (myrow >= 0) {      // (wave-uniform: the four waves whose row r = wave (+ 8) lies in this workgroup's quarter)
            const unsigned tag = gtag(epoch);
            float xs[3];
            int spins = 0;
            if constexpr (FLAGS) {      // element (myrow, col) of a partner was written by ITS wave of this index
                const int q0 = q == 0 ? 1 : 0, q1 = q <= 1 ? 2 : 1, q2 = q <= 2 ? 3 : 2;
                flags_wait(flags_of(q0) + 16 + wave, flags_of(q1) + 16 + wave, flags_of(q2) + 16 + wave);
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    xs[s] = get32(reinterpret_cast<const float*>(a.xd + (int64_t)(panel * SS + (s + (s >= q ? 1 : 0))) * (PR * EE)) + myrow * EE + col);
                (void)tag; (void)spins;
            } else
            while (true) {
                unsigned long long x[3];
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    x[s] = __hip_atomic_load(a.xd + (int64_t)(panel * SS + (s + (s >= q ? 1 : 0))) * (PR * EE) + myrow * EE + col,
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                bool ok = true;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    ok = ok && ((unsigned)(x[s] >> 32) == tag);
                    xs[s] = __uint_as_float((unsigned)x[s]);
                }
                if (ok) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 20)) __builtin_trap();
            }
            const float v0 = q == 0 ? mine : xs[0];
            const float v1 = q == 1 ? mine : (q > 1 ? xs[1] : xs[0]);
            const float v2 = q == 2 ? mine : (q > 2 ? xs[2] : xs[1]);
            const float v3 = q == 3 ? mine : xs[2];
            const float dx = ((v0 + v1) + v2) + v3;
            if (m0 + myrow < a.B) a.dX[(int64_t)(m0 + myrow) * a.ldx + col] = dx;
            if (OBS) {      // dz2 = dX * [E > 0] (inference_network.py:132-139 backward, the last ReLU of the final stack)
                const float dz2 = sE[myrow * PE + col] > 0.0f ? dx : 0.0f;
                sT0[(myrow & 3) * PX + col] = dz2;
                if (m0 + myrow < a.B) oin.dE[(int64_t)(m0 + myrow) * oin.a.e_ld + col] = dz2;
            }
        }
    }
    P16_STAMP(13);
    if (OBS) {
        // dz1 = (dz2 Wf1) * [f1 > 0]; dzc = (dz1 Wf0) * [cat > 0]; dh_o = (dzc_o W1_o) * [h_o > 0]: the workgroup's four rows as
        // rows 0..3 of a 16-row MFMA tile (the other rows of the A operand are whatever the buffer holds: rows are independent),
        // B fragments from the weights' LDS image (row = k, stride cols + 1); wave t < 4 owns column tile t of the 64-wide layers
        const ObsFusedArgs& oa = oin.a;
        auto layer64 = [&](const float* T, const ObsLayer& Lw, const float (&mask)[4], float* dst, int64_t dld, float* Tn) {
            f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
            const float* wb = oimg + Lw.lds_w + 16 * wave + c;
            const int ld = Lw.cols + 1;
#pragma unroll
            for (int un = 0; un < 4; un += 2) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(T + c * PX + 16 * un + 4 * g);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(T + c * PX + 16 * (un + 1) + 4 * g);
                const float* w0 = wb + (16 * un + 4 * g) * ld;
                const float* w1 = w0 + 16 * ld;
                const f32x4 b0 = {w0[0], w0[ld], w0[2 * ld], w0[3 * ld]}, b1 = {w1[0], w1[ld], w1[2 * ld], w1[3 * ld]};
                PP_MMA(0, a0, b0, c0); PP_MMA(0, a1, b1, c1);
                PP_MMA(1, a0, b0, c0); PP_MMA(1, a1, b1, c1);
                PP_MMA(2, a0, b0, c0); PP_MMA(2, a1, b1, c1);
                PP_MMA(3, a0, b0, c0); PP_MMA(3, a1, b1, c1);
            }
            if (g == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d = mask[i] > 0.0f ? c0[i] + c1[i] : 0.0f;
                    Tn[i * PX + 16 * wave + c] = d;
                    if (m0 + 4 * q + i < a.B) dst[(int64_t)(m0 + 4 * q + i) * dld + 16 * wave + c] = d;
                }
            }
        };
        __syncthreads();     // dz2 rows; the weight image
        if (wave < 4) layer64(sT0, oa.f1, m_f1, oin.dF1, oa.e_ld, sT1);
        __syncthreads();
        if (wave < 4) layer64(sT1, oa.f0, m_cat, oin.dCat, oa.e_ld, sT2);
        __syncthreads();
        // per observable o (wave o): dh_o = dzc[:, co .. co + out_o) W1_o, out_o = 16 units of K, one column tile of hidden units
        int co = 0;
#pragma unroll
        for (int o = 0; o < ONB; ++o) {
            if (o >= oa.n_obs) break;
            if (wave == o) {
                f32x4 c0 = {0, 0, 0, 0};
                const int ld = oa.l1[o].cols + 1;
                const float* wb = oimg + oa.l1[o].lds_w + min(c, oa.hid[o] - 1);
                for (int un = 0; un < (oa.out[o] >> 4); ++un) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(sT2 + c * PX + co + 16 * un + 4 * g);
                    const float* w0 = wb + (16 * un + 4 * g) * ld;
                    const f32x4 b0 = {w0[0], w0[ld], w0[2 * ld], w0[3 * ld]};
                    c0 = mma4(a0, b0, c0);
                }
                if (g == 0 && c < oa.hid[o]) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (m0 + 4 * q + i < a.B)
                            oin.dHo0[(int64_t)o * oin.dh_stride + (int64_t)(m0 + 4 * q + i) * oa.ohid_ld[o] + c] = m_h[i] > 0.0f ? c0[i] : 0.0f;
                }
            }
            co += oa.out[o];
        }
        P16_STAMP(14);
    }
#undef P16_STAMP
#undef P16_ISSUE
}

template <int HH_, int KIND, bool OBS, bool FLAGS = false>
static int panel16_launch(const Panel16Args& a, const PanelObs& po, hipStream_t st) {
    constexpr size_t lds = (size_t)P16<HH_>::L_END * sizeof(float);
    static thread_local bool configured = false;   // > 64 KB of dynamic LDS needs the opt-in once per kernel
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)panel16_kernel<HH_, KIND, OBS, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("panel16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        configured = true;
    }
    const int panels = cdiv(a.a.B, PR);
    hipLaunchKernelGGL((panel16_kernel<HH_, KIND, OBS, FLAGS>), dim3(8 * SS * cdiv(panels, 8)), dim3(512), lds, st, a, po);
    return 0;
}

template <int HH_>
static int panel16_slots() {      // workgroups of this kernel the device holds at once (the occupancy gate of panel16_supported)
    int dev = 0, cus = 0, per_cu = 0;
    constexpr size_t lds = (size_t)P16<HH_>::L_END * sizeof(float);
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipFuncSetAttribute((const void*)panel16_kernel<HH_, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)panel16_kernel<HH_, 0, true>, 512, lds) != hipSuccess) return 0;
    return cus * per_cu;
}

}  // namespace

static inline int p16_zk(int H) { return H == 1024 ? P16<1024>::ZK : P16<512>::ZK; }
int64_t panel16_xz_granules(int B, int H) { return (int64_t)cdiv(B, PR) * SS * PR * p16_zk(H); }
int64_t panel16_xd_granules(int B) { return (int64_t)cdiv(B, PR) * SS * PR * EE; }

// Which single-statement batches the 16-row kernel takes. PP_PANEL=1 keeps the 8-row kernel (A/B), 0 the tile kernels.
bool panel16_supported(int kind, int H, int hid, int n_out, int e, int B) {
    static const int env = getenv("PP_PANEL") ? atoi(getenv("PP_PANEL")) : 2;
    if (env < 2 || deterministic_mode()) return false;
    if (kind != PP_HEAD_NORMAL_MIXTURE && kind != PP_HEAD_TRUNCNORMAL_MIXTURE && kind != PP_HEAD_POISSON_TN_MIXTURE) return false;
    if ((H != 512 && H != 1024) || e != EE || hid <= p16_zk(H) - 16 || hid > p16_zk(H)) return false;
    if (n_out % 3 != 0 || n_out < 3 || n_out > 30) return false;           // K <= 10 components: one DPP row per row, two output tiles
    if (B < 1 || cdiv(B, PR) > 1024) return false;
    // The four workgroups of a panel wait for each other through memory: a 32-block window of the grid must be resident
    // together. One workgroup per CU (103 | 146 KB of LDS) and in-order dispatch give that on a device whose occupancy for this
    // kernel covers at least two windows; a CU-masked or partitioned device with fewer slots takes the other kernels instead of
    // risking the bounded spin's trap (ADVICE r03 / VERDICT r04 1d).
    static const int slots512 = panel16_slots<512>(), slots1024 = panel16_slots<1024>();
    return (H == 512 ? slots512 : slots1024) >= 2 * 8 * SS;
}

// the MFMA tail's shapes: two 64 x 64 layers of the final stack, per observable a 16 k-multiple of outputs and <= 16 hidden units
bool panel16_obs_ok(const ObsFusedArgs& oa) {
    if (oa.n_obs < 1 || oa.n_obs > 2 || oa.e_obs != EE || oa.lds_total + 1 > 10240 + 8) return false;
    if (oa.f1.rows != EE || oa.f1.cols != EE || oa.f0.rows != EE || oa.f0.cols != EE) return false;
    int tot = 0;
    for (int o = 0; o < oa.n_obs; ++o) {
        if (oa.out[o] % 16 != 0 || oa.out[o] < 16 || oa.hid[o] < 1 || oa.hid[o] > 16) return false;
        if (oa.l1[o].rows != oa.out[o] || oa.l1[o].cols != oa.hid[o]) return false;
        tot += oa.out[o];
    }
    return tot == EE;
}

int panel16(int kind, const Panel16Args& a, hipStream_t st, const PanelObs* obs) {
    const PanelArgs& p = a.a;
    PP_CHECK_ARG(panel16_supported(kind, p.H, p.hid, p.n_out, p.e, p.B), "panel16: unsupported shape");
    PP_CHECK_ARG(p.ldx % 4 == 0 && p.ldw % 4 == 0 && p.lda1 >= p.hid && p.lddy >= p.n_out && p.K * 3 == p.n_out && p.xz && p.xd &&
                     p.epoch && a.img[0] && a.img[5],
                 "panel16: bad leading dimensions or missing buffers");
    if (obs) PP_CHECK_ARG(panel16_obs_ok(obs->a), "panel16: the observe-embedding tail does not fit");
    static const PanelObs none{};
    // PP_PANEL_HANDOFF=flag (read per call: the A/B tests flip it inside one process): 4-byte payloads behind one flag per producer
    // wave instead of {value, tag} granules - the kernel with the observe-embedding tail, H = 512 only (the benchmarked instantiations)
    const char* hv = getenv("PP_PANEL_HANDOFF");
    const bool flags = hv && hv[0] == 'f' && obs && p.H == 512;
#define PP_P16_GO(HH_, KIND)                                                                       \
    do {                                                                                           \
        if (obs && flags && HH_ == 512) PP_TRY((panel16_launch<512, KIND, true, true>(a, *obs, st)));  \
        else if (obs) PP_TRY((panel16_launch<HH_, KIND, true>(a, *obs, st)));                      \
        else PP_TRY((panel16_launch<HH_, KIND, false>(a, none, st)));                              \
    } while (0)
#define PP_P16_KIND(HH_)                                                  \
    do {                                                                  \
        if (kind == PP_HEAD_NORMAL_MIXTURE) PP_P16_GO(HH_, 0);            \
        else if (kind == PP_HEAD_TRUNCNORMAL_MIXTURE) PP_P16_GO(HH_, 1);  \
        else PP_P16_GO(HH_, 2);                                           \
    } while (0)
    if (p.H == 512) PP_P16_KIND(512);
    else PP_P16_KIND(1024);
#undef PP_P16_KIND
#undef PP_P16_GO
    PP_LAUNCH_CHECK("panel16");
    return 0;
}

}  // namespace pp
