// LSTM recurrence over the LATE time steps of a ragged minibatch, one launch per direction (gfx950 only).
//
// In the step-major layout (packed.py) the traces are sorted by length, so the rows of time step t are the first n_t
// traces and n_t falls quickly (GaussianUnknownMeanMarsaglia, batch 1024: 1024, 1024, 220, 220, 47, 47, 10, 10, 2, 2, ...).
// The reference runs nn.LSTM over the padded batch (inference_network_lstm.py:186-188); the engine's per-step path is
// one recurrent GEMM + one cell kernel per time step and direction - for a step with a few dozen rows that is two
// dependent launches of pure start-up latency (12 + 5 us measured) around 0.1 us of arithmetic.
//
// Here all time steps t >= t0 (n_t0 <= teams * 8 rows) of one layer run inside ONE kernel per direction:
//   * workgroup = (team, slot). A slot owns U = 8192 / H hidden units for the whole kernel and keeps ITS slice of W_hh -
//     the 4U gate rows x H, 128 floats per thread - in REGISTERS: the weights are read once per launch, not per step.
//   * a team owns the traces i = team (mod teams): at most 8 rows per step. Traces are independent, so teams never talk
//     to each other; the S = H / U slots of ONE team exchange once per time step through memory:
//       forward   all-gather of h_t: a slot publishes its U units of the team's rows, every slot reads all H;
//       backward  dh_{t-1} = dG_t W_hh is split over K: a slot multiplies ITS 4U columns of dG_t (which it has just
//                 produced - no exchange) with its weight rows and publishes a partial [rows, H]; the slot owning a
//                 unit adds the S partials (reduce-scatter: rows x H values in per slot instead of rows x 4H).
//     The exchanged words are 8-byte GRANULES {value, tag} stored and loaded with agent-scope relaxed atomics (one
//     write-through store, L1-bypassing loads; MI355X_MICROARCH.md, inter-workgroup visibility): the consumer polls the
//     data itself - no flag, no drain, no fence: one memory hop per time step. tag = launch epoch : step, so a granule
//     of an earlier step or launch is never mistaken for the awaited one; two buffers alternate by step parity (a slot
//     can be at most one step ahead of a team mate). team = blockIdx % teams puts a team on one XCD when teams == 8.
//   * residency: teams * S <= number of CUs is checked by the host; every spin is bounded and raises the step's
//     non-finite flag on time-out (the minibatch is then skipped and reported like a non-finite loss).
// The arithmetic per time step is VALU fp32 (<= 8 rows x 64 gate columns x H per workgroup): an MFMA tile would be
// > 75 % padding at these row counts.
#include <atomic>

#include "common.hpp"

namespace pp {

constexpr int TAIL_RMAX = 8;             // rows of one team per time step
constexpr int TAIL_SPIN_MAX = 1 << 19;   // polls (>= 0.5 us each) before a wait gives up
constexpr int TAIL_STEP_BITS = 8;        // tag = epoch << 8 | (t - t0)

typedef unsigned long long u64;

struct TailDims {
    int t0, T, teams;
    unsigned tag_base;
    int probe;   // 1: a wait first polls one granule per producer, then reads everything (0: polls with the bulk read)
};
struct TailFwdArgs {
    float* G;                 // [R, 4H] pre-activations (input part + biases already there) -> gate activations
    float* C;                 // [R, H]
    float* Hs;                // [R, H]
    const float* Whh;         // [4H, H]
    const int32_t* row_off;   // device [T + 1]
    u64* xch;                 // granules [2][teams][RMAX][H]
    int32_t* flag;            // the step's non-finite flag (raised on a spin time-out)
    TailDims d;
};
struct TailBwdArgs {
    float* G;                 // gate activations -> dG
    const float* C;
    float* dH;                // [R, H] gradient into the hidden states (from the heads / the layer above); rows of step
                              // t0 - 1 receive dG_t0 W_hh
    float* dC;                // [B, H] cell-state gradient carried to step t0 - 1
    const float* Whh;
    const int32_t* row_off;
    u64* xch;                 // granules [2][teams][S dest][S src * RMAX * U]
    int32_t* flag;
    float* db;                // bias gradients (b_ih, b_hh get the same sums), or null
    float* db2;
    LossFinalize fin;
    TailDims d;
};

__device__ __forceinline__ u64 ld_agent64(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_granule(u64* p, float v, unsigned tag) {
    __hip_atomic_store(p, (u64)__float_as_uint(v) | ((u64)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float lo32(u64 v) { return __uint_as_float((unsigned)(v & 0xffffffffull)); }
__device__ __forceinline__ unsigned hi32(u64 v) { return (unsigned)(v >> 32); }

// Wait for `total` consecutive granules carrying `want` and copy their values to LDS in the same order. Lanes < n_probe
// first poll ONE granule each (probe[i * probe_stride]: the last one a producer stores), so that the bulk read - 8-byte
// loads, NQ in flight per thread - normally runs once; it is repeated while any tag is still old. Ends with a barrier.
template <int NQ>
__device__ __forceinline__ bool tail_gather(const u64* src, int total, unsigned want, float* dst, int n_probe, int probe_first,
                                            int probe_stride, int32_t* flag, bool alive) {
    const int tid = threadIdx.x;
    if (alive && tid < n_probe) {
        const u64* p = src + probe_first + (int64_t)tid * probe_stride;
        int it = 0;
        while (hi32(ld_agent64(p)) != want && ++it < TAIL_SPIN_MAX) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    for (int it = 0; it < TAIL_SPIN_MAX; ++it) {
        u64 v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = tid + 256 * q;
            v[q] = i < total ? ld_agent64(src + i) : 0ull;
        }
        int ok = 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = tid + 256 * q;
            if (i < total) {
                if (hi32(v[q]) == want) dst[i] = lo32(v[q]);
                else ok = 0;
            }
        }
        if (__syncthreads_and(ok)) return alive;
        if (!alive) return false;   // (an earlier wait of this launch timed out: do not wait again)
        __builtin_amdgcn_s_sleep(1);
    }
    if (tid == 0) atomicOr(reinterpret_cast<int*>(flag), 1);
    return false;
}

// ---------------------------------------------------------------------------------------------------------------
// forward: for t = t0 .. T-1:  G_t += h_{t-1} W_hh^T ; (c_t, h_t) = cell(G_t, c_{t-1})
// thread = (gate column gc of the slot's 4U, K segment ks of 128): w[] = W_hh[row(gc), ks*128 .. +128)
// ---------------------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(256, 1) void lstm_tail_fwd_kernel(const TailFwdArgs a) {
    constexpr int U = 8192 / H, GC = 4 * U, KS = 256 / GC, S = H / U, WS = H + 4, NQ = TAIL_RMAX * H / 256;
    static_assert(KS * 128 == H, "K segments of 128");
    extern __shared__ float lds[];
    float* hl = lds;                       // [RMAX][H] h_{t-1} of the team's rows
    float* red = lds + TAIL_RMAX * H;      // [KS][RMAX][GC] partial sums
    const int tid = threadIdx.x;
    const int teams = a.d.teams, team = blockIdx.x % teams, slot = blockIdx.x / teams;
    const int gc = tid % GC, ks = tid / GC;
    float w[128];
    // the slot's weight rows (4 gates x U rows of H floats, each gate's block contiguous in memory): coalesced global -> LDS
    // with 8 loads in flight per thread, then every thread picks its 128 floats (row stride H + 4: no bank conflicts)
    for (int base = tid; base < GC * H / 4; base += 256 * 8) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = base + 256 * q, r = (4 * i) / H, c = (4 * i) % H;   // r = gate * U + unit
            v[q] = *reinterpret_cast<const float4*>(a.Whh + (int64_t)((r / U) * H + slot * U + r % U) * H + c);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = base + 256 * q, r = (4 * i) / H, c = (4 * i) % H;
            *reinterpret_cast<float4*>(lds + r * WS + c) = v[q];
        }
    }
    __syncthreads();
    {
        const float4* wp = reinterpret_cast<const float4*>(lds + gc * WS + ks * 128);
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const float4 v = wp[q];
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
    }
    __syncthreads();
    bool alive = true;
    const int cm = tid / (U / 2), cu = 2 * (tid % (U / 2));   // cell thread: (row m of the team, units cu, cu + 1), every step
    const int j = slot * U + cu;
    for (int t = a.d.t0; t < a.d.T; ++t) {
        const int r0 = a.row_off[t], n = a.row_off[t + 1] - r0, rp = a.row_off[t - 1];
        const int cnt = n > team ? (n - team + teams - 1) / teams : 0;   // rows of this team: traces team, team + teams, ...
        if (cnt == 0) break;
        const bool cell = tid < cnt * (U / 2);
        const int64_t row = r0 + team + teams * cm, prow = rp + team + teams * cm;
        // the cell's own inputs do not depend on the team mates: in flight while the hidden state is awaited
        float2 pre[4], cp;
        if (cell) {
#pragma unroll
            for (int q = 0; q < 4; ++q) pre[q] = *reinterpret_cast<const float2*>(a.G + row * 4 * H + q * H + j);
            cp = *reinterpret_cast<const float2*>(a.C + prow * H + j);
        }
        if (t == a.d.t0) {   // h_{t0-1}: written by the kernels before this one
            for (int i = tid; i < cnt * (H / 4); i += 256) {
                const int m = i / (H / 4), k4 = i % (H / 4);
                *reinterpret_cast<float4*>(hl + m * H + 4 * k4) =
                    *reinterpret_cast<const float4*>(a.Hs + (int64_t)(rp + team + teams * m) * H + 4 * k4);
            }
            __syncthreads();
        } else {             // h_{t-1}: the team's granules of the previous step ([m][H], last stored: row cnt-1, unit U-1 of a slot)
            const u64* src = a.xch + ((int64_t)(((t - 1) & 1) * teams + team) * TAIL_RMAX) * H;
            alive = tail_gather<NQ>(src, cnt * H, a.d.tag_base + (unsigned)(t - 1 - a.d.t0), hl, a.d.probe ? S : 0, (cnt - 1) * H + U - 1, U, a.flag,
                                    alive);
        }
        for (int m0 = 0; m0 < cnt; m0 += 4) {   // (rows beyond cnt: stale LDS, computed and never read)
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            const float4* h0 = reinterpret_cast<const float4*>(hl + (m0 + 0) * H + ks * 128);
            const float4* h1 = reinterpret_cast<const float4*>(hl + (m0 + 1) * H + ks * 128);
            const float4* h2 = reinterpret_cast<const float4*>(hl + (m0 + 2) * H + ks * 128);
            const float4* h3 = reinterpret_cast<const float4*>(hl + (m0 + 3) * H + ks * 128);
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const float4 x0 = h0[q], x1 = h1[q], x2 = h2[q], x3 = h3[q];
                a0 = fmaf(w[4 * q], x0.x, a0); a0 = fmaf(w[4 * q + 1], x0.y, a0); a0 = fmaf(w[4 * q + 2], x0.z, a0); a0 = fmaf(w[4 * q + 3], x0.w, a0);
                a1 = fmaf(w[4 * q], x1.x, a1); a1 = fmaf(w[4 * q + 1], x1.y, a1); a1 = fmaf(w[4 * q + 2], x1.z, a1); a1 = fmaf(w[4 * q + 3], x1.w, a1);
                a2 = fmaf(w[4 * q], x2.x, a2); a2 = fmaf(w[4 * q + 1], x2.y, a2); a2 = fmaf(w[4 * q + 2], x2.z, a2); a2 = fmaf(w[4 * q + 3], x2.w, a2);
                a3 = fmaf(w[4 * q], x3.x, a3); a3 = fmaf(w[4 * q + 1], x3.y, a3); a3 = fmaf(w[4 * q + 2], x3.z, a3); a3 = fmaf(w[4 * q + 3], x3.w, a3);
            }
            red[(ks * TAIL_RMAX + m0 + 0) * GC + gc] = a0;
            red[(ks * TAIL_RMAX + m0 + 1) * GC + gc] = a1;
            red[(ks * TAIL_RMAX + m0 + 2) * GC + gc] = a2;
            red[(ks * TAIL_RMAX + m0 + 3) * GC + gc] = a3;
        }
        __syncthreads();
        if (cell) {
            float act[4][2], cn[2], hn[2], p[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float s0 = pre[q].x, s1 = pre[q].y;
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    s0 += red[(k * TAIL_RMAX + cm) * GC + q * U + cu];
                    s1 += red[(k * TAIL_RMAX + cm) * GC + q * U + cu + 1];
                }
                p[q][0] = s0; p[q][1] = s1;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float gi = sigmoidf_(p[0][e]), gf = sigmoidf_(p[1][e]), gg = tanhf(p[2][e]), go = sigmoidf_(p[3][e]);
                const float c = gi * gg + gf * (e ? cp.y : cp.x);
                act[0][e] = gi; act[1][e] = gf; act[2][e] = gg; act[3][e] = go;
                cn[e] = c;
                hn[e] = go * tanhf(c);
            }
            if (t + 1 < a.d.T) {   // first of all: what the team mates wait for
                u64* dst = a.xch + ((int64_t)((t & 1) * teams + team) * TAIL_RMAX + cm) * H + j;
                const unsigned tag = a.d.tag_base + (unsigned)(t - a.d.t0);
                st_granule(dst, hn[0], tag);
                st_granule(dst + 1, hn[1], tag);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float2*>(a.G + row * 4 * H + q * H + j) = make_float2(act[q][0], act[q][1]);
            *reinterpret_cast<float2*>(a.C + row * H + j) = make_float2(cn[0], cn[1]);
            *reinterpret_cast<float2*>(a.Hs + row * H + j) = make_float2(hn[0], hn[1]);
        }
        // (hl / red are rewritten behind the barriers of the next step's gather)
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward: for t = T-1 .. t0:  dG_t = cell'(dh_t, dc_t) ; dh_{t-1} += dG_t W_hh ; dc_{t-1} = dc_t f_t
// product thread = KP = H / 256 columns k of the slot's partial [rows, H]: w[] = W_hh[the slot's 4U gate rows, its k's]
// ---------------------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(256, 1) void lstm_tail_bwd_kernel(const TailBwdArgs a) {
    constexpr int U = 8192 / H, GC = 4 * U, S = H / U, KP = H / 256, NQ = TAIL_RMAX * H / 256;
    static_assert(GC * KP == 128, "128 weights per thread");
    extern __shared__ float lds[];
    float* gl = lds;                                  // [RMAX][GC] the slot's columns of dG_t
    float* part = gl + TAIL_RMAX * GC;                // [S][cnt][U] partials received for the slot's units
    float* sbl = part + S * TAIL_RMAX * U;            // [RMAX][4][U] bias-gradient partials
    const int tid = threadIdx.x;
    if (a.fin.acc && blockIdx.x == 0 && tid == 0) loss_finalize_inline(a.fin);
    const int teams = a.d.teams, team = blockIdx.x % teams, slot = blockIdx.x / teams;
    const int k0 = tid * KP;
    float w[128];
#pragma unroll
    for (int g = 0; g < GC; ++g) {   // gate column g of the slot = gate g / U, unit g % U; lanes run along k: coalesced
        const float* wp = a.Whh + (int64_t)((g / U) * H + slot * U + (g % U)) * H + k0;
#pragma unroll
        for (int e = 0; e < KP; ++e) w[g * KP + e] = wp[e];
    }
    bool alive = true;
    // cell thread (row m, units cu, cu + 1): state carried over the time steps (row sets are nested: n_t >= n_{t+1})
    const int cm = tid / (U / 2), cu = 2 * (tid % (U / 2));
    const int j = slot * U + cu;
    float dc[2] = {0.f, 0.f}, rec[2] = {0.f, 0.f}, sb[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    // the cell's own inputs of a step (gates, c_t, c_{t-1}, dh from above) do not depend on the team mates: those of the
    // NEXT step to process are requested before the wait of the current one
    float2 gt[4], cv, cp, dhv;
    auto team_rows = [&](int t) {
        const int n = a.row_off[t + 1] - a.row_off[t];
        return n > team ? (n - team + teams - 1) / teams : 0;
    };
    auto fetch = [&](int t) {
        if (t < a.d.t0 || tid >= team_rows(t) * (U / 2)) return;
        const int64_t row = a.row_off[t] + team + teams * cm, prow = a.row_off[t - 1] + team + teams * cm;
        const float* g = a.G + row * 4 * H + j;
#pragma unroll
        for (int q = 0; q < 4; ++q) gt[q] = *reinterpret_cast<const float2*>(g + q * H);
        cv = *reinterpret_cast<const float2*>(a.C + row * H + j);
        cp = *reinterpret_cast<const float2*>(a.C + prow * H + j);
        dhv = *reinterpret_cast<const float2*>(a.dH + row * H + j);
    };
    int t_first = a.d.T - 1;
    while (t_first >= a.d.t0 && team_rows(t_first) == 0) --t_first;   // the team's longest trace ends here
    fetch(t_first);
    for (int t = t_first; t >= a.d.t0; --t) {
        const int r0 = a.row_off[t], rp = a.row_off[t - 1];
        const int cnt = team_rows(t);
        const unsigned tag = a.d.tag_base + (unsigned)(t - a.d.t0);
        const bool cell = tid < cnt * (U / 2);
        if (cell) {   // the cell's backward for the slot's units -> its 4U columns of dG_t
            const int64_t row = r0 + team + teams * cm;
            float* g = a.G + row * 4 * H + j;
            float d[4][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float gi = e ? gt[0].y : gt[0].x, gf = e ? gt[1].y : gt[1].x, gg = e ? gt[2].y : gt[2].x,
                            go = e ? gt[3].y : gt[3].x;
                const float tc = tanhf(e ? cv.y : cv.x);
                const float dh = (e ? dhv.y : dhv.x) + rec[e];
                const float dcv = dc[e] + dh * go * (1.0f - tc * tc);
                d[0][e] = dcv * gg * gi * (1.0f - gi);
                d[1][e] = dcv * (e ? cp.y : cp.x) * gf * (1.0f - gf);
                d[2][e] = dcv * gi * (1.0f - gg * gg);
                d[3][e] = dh * tc * go * (1.0f - go);
                dc[e] = dcv * gf;
                sb[0][e] += d[0][e]; sb[1][e] += d[1][e]; sb[2][e] += d[2][e]; sb[3][e] += d[3][e];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gl[cm * GC + q * U + cu] = d[q][0];
                gl[cm * GC + q * U + cu + 1] = d[q][1];
                *reinterpret_cast<float2*>(g + q * H) = make_float2(d[q][0], d[q][1]);   // (read by the weight-gradient launch)
            }
        }
        fetch(t - 1);
        __syncthreads();
        // partial[m][k0 ..] = sum over the slot's gate columns; published to the slot that owns unit k: dest = k / U
        {
            u64* xb = a.xch + ((int64_t)((t & 1) * teams + team) * S) * ((int64_t)S * TAIL_RMAX * U);
            for (int m0 = 0; m0 < cnt; m0 += 4) {
                float acc[4][KP];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int e = 0; e < KP; ++e) acc[r][e] = 0.f;
#pragma unroll
                for (int g4 = 0; g4 < GC / 4; ++g4) {
                    float4 x[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[r] = *reinterpret_cast<const float4*>(gl + (m0 + r) * GC + 4 * g4);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int e = 0; e < KP; ++e) {
                            acc[r][e] = fmaf(x[r].x, w[(4 * g4 + 0) * KP + e], acc[r][e]);
                            acc[r][e] = fmaf(x[r].y, w[(4 * g4 + 1) * KP + e], acc[r][e]);
                            acc[r][e] = fmaf(x[r].z, w[(4 * g4 + 2) * KP + e], acc[r][e]);
                            acc[r][e] = fmaf(x[r].w, w[(4 * g4 + 3) * KP + e], acc[r][e]);
                        }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + r;
                    if (m < cnt) {
#pragma unroll
                        for (int e = 0; e < KP; ++e) {
                            const int k = k0 + e, dest = k / U;
                            st_granule(xb + (int64_t)dest * (S * TAIL_RMAX * U) + (int64_t)(slot * cnt + m) * U + k % U, acc[r][e], tag);
                        }
                    }
                }
            }
        }
        // the S partials for the slot's units ([src][m][U]; a source's last store: row cnt-1, its highest k of this slot)
        {
            const u64* src = a.xch + ((int64_t)((t & 1) * teams + team) * S + slot) * ((int64_t)S * TAIL_RMAX * U);
            alive = tail_gather<NQ>(src, S * cnt * U, tag, part, a.d.probe ? S : 0, (cnt - 1) * U + U - 1, cnt * U, a.flag, alive);
        }
        if (cell) {   // dh_{t-1}[row m, units] = the sum over the sources
            float s0 = 0.f, s1 = 0.f;
            for (int k = 0; k < S; ++k) {
                s0 += part[(k * cnt + cm) * U + cu];
                s1 += part[(k * cnt + cm) * U + cu + 1];
            }
            rec[0] = s0; rec[1] = s1;
            if (t == a.d.t0) {   // hand over to the per-step path: dh and dc of step t0 - 1 (these elements are this thread's)
                const int64_t prow = rp + team + teams * cm;
                float2* dhp = reinterpret_cast<float2*>(a.dH + prow * H + j);
                const float2 o = *dhp;
                *dhp = make_float2(o.x + s0, o.y + s1);
                *reinterpret_cast<float2*>(a.dC + (int64_t)(team + teams * cm) * H + j) = make_float2(dc[0], dc[1]);
            }
        }
        __syncthreads();   // gl / part are rewritten by the next step
    }
    if (!a.db) return;
    // bias gradients: sum over the team's rows in LDS, one atomic per (gate, unit) and workgroup
    __syncthreads();
    if (tid < TAIL_RMAX * (U / 2)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sbl[(cm * 4 + q) * U + cu] = sb[q][0];
            sbl[(cm * 4 + q) * U + cu + 1] = sb[q][1];
        }
    }
    __syncthreads();
    if (tid < 4 * U) {
        const int q = tid / U, uu = tid % U;
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < TAIL_RMAX; ++m) s += sbl[(m * 4 + q) * U + uu];
        if (s != 0.0f) {
            atomicAdd(a.db + q * H + slot * U + uu, s);
            if (a.db2) atomicAdd(a.db2 + q * H + slot * U + uu, s);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static int device_cus() {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        else cus = 0;
        (void)hipGetLastError();
    }
    return cus;
}

static bool tail_dim_ok(int H) { return H == 256 || H == 512 || H == 1024; }

// teams of a network with hidden size H on this device (0: the tail kernels do not apply)
int lstm_tail_teams(int H) {
    static const int enabled = getenv("PP_LSTM_TAIL") ? atoi(getenv("PP_LSTM_TAIL")) : 1;
    if (!enabled || !tail_dim_ok(H)) return 0;
    const int U = 8192 / H, S = H / U;
    int teams = std::min(PP_TAIL_TEAMS_MAX, device_cus() / S);
    if (teams > 8 && teams < 16) teams = 8;   // a whole number of teams per XCD
    return std::max(teams, 0);
}

// bytes of the two granule exchange areas (forward, backward) at the START of the pp_ic_loss workspace
void lstm_tail_exchange_bytes(int H, size_t* fwd, size_t* bwd) {
    const int teams = lstm_tail_teams(H);
    *fwd = *bwd = 0;
    if (!teams) return;
    const int U = 8192 / H, S = H / U;
    *fwd = sizeof(u64) * 2 * (size_t)teams * TAIL_RMAX * H;
    *bwd = sizeof(u64) * 2 * (size_t)teams * S * S * TAIL_RMAX * U;
}

// How a batch uses the tail kernels: teams (0 = not at all) and the first time step they take.
// n_active: host [T]. PP_LSTM_TAIL=0 switches the path off; PP_LSTM_TAIL_MIN_STEPS (default 2) = fewest steps worth a launch.
int lstm_tail_plan(const int32_t* n_active, int T, int H, int* t0_out, int* teams_out) {
    static const int min_steps = 2;
    *t0_out = T;
    *teams_out = 0;
    const int teams = lstm_tail_teams(H);
    if (teams < 1 || T < 2) return 0;
    int t0 = T;
    for (int t = 1; t < T; ++t)
        if (n_active[t] <= teams * TAIL_RMAX) { t0 = t; break; }
    if (T - t0 < min_steps || T - t0 > (1 << TAIL_STEP_BITS)) return 0;
    *t0_out = t0;
    *teams_out = teams;
    return 1;
}

// A tag no earlier launch of this process has used. When the epoch counter wraps, the exchange areas of the workspace
// in use are cleared (tag 0 is never awaited).
static unsigned next_tag_base(void* xch_f, size_t f_bytes, void* xch_b, size_t b_bytes, hipStream_t st) {
    static std::atomic<unsigned> epoch{0};
    unsigned e = ++epoch;
    if ((e & ((1u << (32 - TAIL_STEP_BITS)) - 1)) == 0) {
        (void)hipMemsetAsync(xch_f, 0, f_bytes, st);
        (void)hipMemsetAsync(xch_b, 0, b_bytes, st);
        e = ++epoch;
    }
    return e << TAIL_STEP_BITS;
}

static int tail_probe() {
    static const int probe = 1;
    return probe;
}

template <int H>
static int launch_fwd(const TailFwdArgs& a, hipStream_t st) {
    constexpr int U = 8192 / H, GC = 4 * U, KS = 256 / GC;
    const size_t lds = sizeof(float) * std::max(TAIL_RMAX * H + KS * TAIL_RMAX * GC, GC * (H + 4));
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_tail_fwd_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            set_error("lstm_tail_fwd: cannot reserve %zu bytes of LDS", lds);
            return PP_EHIP;
        }
        attr = true;
    }
    hipLaunchKernelGGL(lstm_tail_fwd_kernel<H>, dim3(a.d.teams * (H / U)), dim3(256), lds, st, a);
    PP_LAUNCH_CHECK("lstm_tail_fwd");
    return 0;
}
template <int H>
static int launch_bwd(const TailBwdArgs& a, hipStream_t st) {
    constexpr int U = 8192 / H, GC = 4 * U, S = H / U;
    const size_t lds = sizeof(float) * (TAIL_RMAX * GC + S * TAIL_RMAX * U + TAIL_RMAX * 4 * U);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_tail_bwd_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            set_error("lstm_tail_bwd: cannot reserve %zu bytes of LDS", lds);
            return PP_EHIP;
        }
        attr = true;
    }
    hipLaunchKernelGGL(lstm_tail_bwd_kernel<H>, dim3(a.d.teams * (H / U)), dim3(256), lds, st, a);
    PP_LAUNCH_CHECK("lstm_tail_bwd");
    return 0;
}

int lstm_tail_fwd(float* G, float* C, float* Hs, const float* Whh, const int32_t* row_off_dev, int t0, int T, int H, int teams,
                  void* xch_f, void* xch_b, int32_t* flag, hipStream_t st) {
    PP_CHECK_ARG(G && C && Hs && Whh && row_off_dev && xch_f && flag && t0 >= 1 && t0 < T && teams >= 1 && teams <= PP_TAIL_TEAMS_MAX,
                 "lstm_tail_fwd: bad argument");
    size_t fb, bb;
    lstm_tail_exchange_bytes(H, &fb, &bb);
    TailFwdArgs a{G, C, Hs, Whh, row_off_dev, static_cast<u64*>(xch_f), flag, TailDims{t0, T, teams, next_tag_base(xch_f, fb, xch_b, bb, st), tail_probe()}};
    switch (H) {
        case 256: return launch_fwd<256>(a, st);
        case 512: return launch_fwd<512>(a, st);
        case 1024: return launch_fwd<1024>(a, st);
    }
    set_error("lstm_tail_fwd: lstm_dim %d not supported", H);
    return PP_EINVAL;
}

int lstm_tail_bwd(float* G, const float* C, float* dH, float* dC, const float* Whh, const int32_t* row_off_dev, int t0, int T,
                  int H, int teams, void* xch_f, void* xch_b, int32_t* flag, float* db, float* db2, const LossFinalize& fin,
                  hipStream_t st) {
    PP_CHECK_ARG(G && C && dH && dC && Whh && row_off_dev && xch_b && flag && t0 >= 1 && t0 < T && teams >= 1 &&
                     teams <= PP_TAIL_TEAMS_MAX, "lstm_tail_bwd: bad argument");
    size_t fb, bb;
    lstm_tail_exchange_bytes(H, &fb, &bb);
    TailBwdArgs a{G, C, dH, dC, Whh, row_off_dev, static_cast<u64*>(xch_b), flag, db, db2, fin,
                  TailDims{t0, T, teams, next_tag_base(xch_f, fb, xch_b, bb, st), tail_probe()}};
    switch (H) {
        case 256: return launch_bwd<256>(a, st);
        case 512: return launch_bwd<512>(a, st);
        case 1024: return launch_bwd<1024>(a, st);
    }
    set_error("lstm_tail_bwd: lstm_dim %d not supported", H);
    return PP_EINVAL;
}

}  // namespace pp
