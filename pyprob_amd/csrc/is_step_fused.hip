// One importance-sampling statement of N particles with per-particle LSTM state as ONE kernel
// (InferenceNetworkLSTM._infer_step pyprob/nn/inference_network_lstm.py:82-134 + state.sample's IC branch
// pyprob/state.py:203-219 + Mixture.sample / Mixture.log_prob pyprob/distributions/mixture.py:38-63):
//
//     gates = [s_prev | h] [W_s | W_hh]^T + bias        bias = b_ih + b_hh + W_ih[:, shared columns] x_shared
//     (c, h) <- LSTM cell                                in registers: the gate matrix never exists in memory
//     a1 = relu(h W1^T + b1),  y = a1 W2^T + b2          from the workgroup's own fresh h tile (LDS)
//     v ~ mixture(y, prior),  log q(v)                   Philox draw in the tail
//
// Every particle of a statement has the same observe embedding and the same (previous, current) address, so of the
// LSTM input row [E | s_prev | d_prev | a_prev | d_cur | a_cur] (inference_network_lstm.py:116-121) only the sample
// embedding s_prev (smp_dim = 4 columns) differs between particles: the other 208 columns are ONE row times W_ih, a
// bias vector computed once per call (is_prep_kernel). K of the per-particle product is H + smp_dim instead of H + 212.
//
// Geometry: a workgroup of eight waves owns 32 particles and ALL 4H gate columns; wave w owns the four gates of the
// hidden units [w H/8, (w+1) H/8): 4 UB blocks of v_mfma_f32_32x32x2_f32 accumulators (UB = H / 256; 128 registers at
// H = 512), so the cell runs on the accumulators without any exchange. 32 x 4H fp32 accumulators are half of a CU's
// register file: 32 rows is the largest panel that keeps the whole gate row on chip.
// Operands: the weights are re-tiled ONCE per call (is_prep_kernel) into "fragment images": for k-slab s (8 k), wave w,
// block b one contiguous KB holding, for lane l = (column c = l & 31, half hh = l >> 5), the four k values 8 s + 4 hh + j of
// its column - exactly the B operand of four consecutive MFMAs. A wave streams ITS blocks with one coalesced 16-byte load
// per lane and block straight into VGPRs (no LDS for the weights, a two-slab register ring refilled behind the MFMAs;
// the one s_barrier per two slabs of the K loop orders no data, it keeps the two waves of a SIMD in step); the A fragment
// (four k of the lane's particle row) is one 16-byte load per slab from the particle's h row (an optional row index list
// gathers the rows of a diverged control-flow path in place). h and c are updated in place: a barrier separates the last
// read of the old h rows from the first store.
#include "is_step_fused.hpp"

#include "gather.hpp"
#include "is_draw.hpp"

#include <algorithm>

namespace pp {

extern long long* g_timeline;   // kernels.hip (pp_debug_timeline)

namespace {

constexpr int FR = 32;   // particles per workgroup
constexpr int FW = 8;    // waves per workgroup

// sigmoid / tanh on v_exp_f32 / v_rcp_f32 (~1 ulp each; absolute error of the results ~1e-7, asserted by
// tests/test_gpu_is_step_fused.py against the float64 oracle)
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// reductions over the 16 lanes of a DPP row (lanes 16 q .. 16 q + 15): every lane ends with the result
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0x128>(v));   // row_ror:8
    v = fmaxf(v, dpp_mov<0x124>(v));   // row_ror:4
    v = fmaxf(v, dpp_mov<0x122>(v));   // row_ror:2
    v = fmaxf(v, dpp_mov<0x121>(v));   // row_ror:1
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0x128>(v);
    v += dpp_mov<0x124>(v);
    v += dpp_mov<0x122>(v);
    v += dpp_mov<0x121>(v);
    return v;
}
__device__ __forceinline__ int row16_min_int(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xF, 0xF, false));
    return v;
}
// inclusive prefix sum over the 16 lanes of a row (row_shr:n; lanes without a source add zero)
__device__ __forceinline__ float row16_prefix(float v) {
    v += dpp_mov<0x111>(v);
    v += dpp_mov<0x112>(v);
    v += dpp_mov<0x114>(v);
    v += dpp_mov<0x118>(v);
    return v;
}

struct PrepArgs {
    const float* P;
    const int64_t* at;
    int64_t w_ih, w_hh, b_ih, b_hh, w1, w2;
    int H, I, ub, nsh;              // nsh = H / 8 slabs of W_hh (slab nsh = the sample-embedding columns of W_ih)
    int hu, halves;                 // hidden units per gate image (H; the wide LSTM launch: H / halves, one image per part)
    GatherDims d;
    int addr_id, prev_addr;
    const float* e_obs_vec;
    const float* h0;                // shared state row (its recurrent product joins the bias) or nullptr
    const float* c0;
    int hid, n_out, nb16, ns2;
    float* whh_img; float* w1_img; float* w2_img; float* bias; float* c0_copy;
    int64_t q_whh, q_w1, q_w2;      // 16-byte pieces of the three images
    int img_blocks;
};

// blocks [0, img_blocks): the fragment images (one 16-byte piece = four k of one column per thread);
// blocks [img_blocks, ...): the bias row, one wave per gate column.
__global__ __launch_bounds__(256) void is_prep_kernel(const PrepArgs a) {
    __shared__ float sx[1024 + 1024];
    const int tid = threadIdx.x;
    const int H = a.H;
    if ((int)blockIdx.x < a.img_blocks) {
        const int64_t total = a.q_whh + a.q_w1 + a.q_w2;
        for (int64_t q = (int64_t)blockIdx.x * 256 + tid; q < total; q += (int64_t)a.img_blocks * 256) {
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            float* dst;
            if (q < a.q_whh) {
                const int lane = (int)(q & 63);
                const int64_t t = q >> 6;
                const int nb = 4 * a.ub;
                const int blk = (int)(t % nb), w = (int)((t / nb) % FW);
                const int sp = (int)(t / (nb * FW)), part = sp / (a.nsh + 1), s = sp - part * (a.nsh + 1);
                const int g = blk / a.ub, ub = blk - g * a.ub;
                const int col = g * H + part * a.hu + (w * a.ub + ub) * 32 + (lane & 31);
                const int k0 = 4 * (lane >> 5);
                if (s < a.nsh) {
                    v = *reinterpret_cast<const f32x4*>(a.P + a.w_hh + (int64_t)col * H + 8 * s + k0);
                } else {
                    const float* wi = a.P + a.w_ih + (int64_t)col * a.I + a.d.e_obs;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (k0 + j < a.d.smp) ? wi[k0 + j] : 0.0f;
                }
                dst = a.whh_img + q * 4;
            } else if (q < a.q_whh + a.q_w1) {
                // W1 as B fragments of v_mfma_f32_16x16x4_f32: 16-k slab s, 16-column block cb, lane (column l % 16, k group
                // l / 16) holds k = 16 s + 4 (l / 16) + j, j = 0..3 (MFMA j of the slab takes one k of every group)
                const int64_t r = q - a.q_whh;
                const int lane = (int)(r & 63);
                const int64_t t = r >> 6;
                const int cb = (int)(t % a.nb16), s = (int)(t / a.nb16);
                const int col = cb * 16 + (lane & 15);
                if (col < a.hid) v = *reinterpret_cast<const f32x4*>(a.P + a.w1 + (int64_t)col * H + 16 * s + 4 * (lane >> 4));
                dst = a.w1_img + r * 4;
            } else {
                const int64_t r = q - a.q_whh - a.q_w1;
                const int lane = (int)(r & 63);
                const int s = (int)(r >> 6);
                const int col = lane & 31, k0 = 8 * s + 4 * (lane >> 5);
                if (col < a.n_out) {
                    const float* wr = a.P + a.w2 + (int64_t)col * a.hid;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (k0 + j < a.hid) ? wr[k0 + j] : 0.0f;
                }
                dst = a.w2_img + r * 4;
            }
            *reinterpret_cast<f32x4*>(dst) = v;
        }
        return;
    }
    // ---- bias row: b_ih + b_hh + W_ih x_shared (sample-embedding columns left out) [+ W_hh h0] ----
    // (the input row passes through LDS in chunks of 1 024 columns: any lstm_in - pyprob's default observe embedding is 256 wide
    // per observable; one chunk, i.e. the same order of additions as before, up to 1 024)
    const int bb = blockIdx.x - a.img_blocks;
    const int c1 = a.d.e_obs, c2 = c1 + a.d.smp;
    const int wave = tid >> 6, lane = tid & 63;
    const int n = bb * 4 + wave;
    const bool own = n < 4 * H;
    const float* wi = a.P + a.w_ih + (int64_t)(own ? n : 0) * a.I;
    float acc = 0.0f;
    if (a.h0)
        for (int k = tid; k < H; k += 256) sx[1024 + k] = a.h0[k];
    if (bb == 0 && a.c0)
        for (int k = tid; k < H; k += 256) a.c0_copy[k] = a.c0[k];
    for (int base = 0; base < a.I; base += 1024) {
        const int cnt = min(1024, a.I - base);
        if (base) __syncthreads();      // the previous chunk has been consumed
        for (int kk = tid; kk < cnt; kk += 256) {
            const int k = base + kk;
            float x;
            if (k < c1) x = a.e_obs_vec[k];
            else if (k < c2) x = 0.0f;
            else x = gather_embedding_elem(a.d, a.P, a.at, k, a.prev_addr, 0.0f, a.addr_id);
            sx[kk] = x;
        }
        __syncthreads();
        if (own)
            for (int kk = lane; kk < cnt; kk += 64) acc += wi[base + kk] * sx[kk];
    }
    if (!own) return;
    if (a.h0) {
        const float* wh = a.P + a.w_hh + (int64_t)n * H;
        for (int k = lane; k < H; k += 64) acc += wh[k] * sx[1024 + k];
    }
    acc = wave_sum(acc);
    if (lane == 0) a.bias[n] = acc + (a.P[a.b_ih + n] + a.P[a.b_hh + n]);
}

struct FusedArgs {
    const float* whh_img;
    const float* bias;
    float* h;
    float* c;
    const float* c0;            // state_shared: the shared previous cell state row (a copy: row 0 of c is rewritten)
    const int64_t* rows;        // particle -> state row, or nullptr (identity)
    int state_shared;
    const float* prev_value;    // [n]
    const float* smp_w;         // sample embedding of the previous address: [smp_dim, smp_in], [smp_dim]
    const float* smp_b;
    int smp_in, smp;
    const float* w1_img; const float* b1; int hid, nb16;
    const float* w2_img; const float* b2; int n_out, ns2;
    float* y_out; int64_t ldy;  // optional copy of the head outputs (heads that are sampled by their own kernel)
    const float* prior; int prior_stride;
    const float* value_in; float* value_out; float* logq_out;
    uint64_t seed, offset;
    int K, n;
    // whole-statement mode (pp_is_statement_rows): the previous values are read at the particle's row, the drawn value goes to
    // value_full[row] and the log-weight takes + log p(v) - log q(v) there (state.py:211-217) - no gather / scatter / axpy
    // launches around the statement. prior_kind: 0 Normal(pa, pb), 1 Uniform[pa, pb)
    int prev_indexed;
    float* value_full;
    float* lw_full;
    int prior_kind;
    long long* dbg;             // debug: s_memtime stamps [2 workgroups][2 waves][16] (pp_debug_timeline) or nullptr
    // split statement (small launches): the new hidden rows [n][H] written by is_small_lstm_kernel - the HEADONLY instantiation
    // starts from them instead of running the K loop and the cell
    const float* hn;
    // wide LSTM launch (KM > 1): the new hidden rows go to hout [n][H K-extent] (compact); panels = workgroups per part
    float* hout;
    int panels;
};

extern __shared__ __attribute__((aligned(1024))) float fused_lds[];

// KIND 0 / 1 / 2: mixture heads (is_draw.hpp), drawn in the tail; 3: head outputs only. SHARED: every particle's previous state is
// one shared row (no per-particle recurrent product) - a template parameter, not a runtime branch: with both K-loop variants in
// one function the 128 accumulator registers met at a control-flow join in different physical registers and the compiler
// shuffled all of them and spilled 32 (8 KB of scratch traffic per particle in the PMC passes of the first version)
// HEADONLY: the second launch of a SPLIT statement (see is_small_lstm_kernel below): the panel's new hidden rows come from a.hn,
// are stored to the state rows and laid out in LDS; everything from head layer 1 on is the same code.
// KM > 1: the WIDE LSTM launch (LSTM of 256 UB KM hidden units: H = 1024 as UB = 2, KM = 2). The gate row of 32 particles no
// longer fits a CU's accumulators, so a workgroup owns 32 particles x ONE PART of the hidden units (256 UB of them, all four
// gates): the same K loop over the whole previous state (K extent HK = 256 UB KM), one gate image per part, the cell on the
// accumulators; workgroups [part * panels, (part + 1) * panels) take part `part`. c is updated in place (a part reads and writes
// only its own units); the new h goes to a.hout ([n][HK], compact) because the other parts still read the old rows; the kernel
// ends after the cell (the head layers and the draw run as the chain's launches, is_kernels.hip is_step).
template <int UB, int KIND, bool SHARED, bool HEADONLY = false, int KM = 1>
__global__ __launch_bounds__(512) void is_step_fused_kernel(const FusedArgs a) {
    constexpr int H = 256 * UB;                    // hidden units of this workgroup
    constexpr int HK = H * KM;                     // hidden units of the LSTM: K extent of the recurrent product, state row pitch
    constexpr int NSH = H / 8;
    constexpr int NSK = HK / 8;                    // k-slabs of W_hh (slab NSK of an image: the sample-embedding columns of W_ih)
    constexpr int NB = 4 * UB;
    constexpr int SLAB = FW * NB * 256;            // floats of one k-slab of the gate image
    float* sH = fused_lds;                         // [H / 16][2][64][4]: the fresh hidden tile as 16-row A fragments (H * 32 floats)
    // (UB = 4, the head-only launch of the H = 1024 statement: the tile alone is 128 KB - the activations of head layer 1 go OVER it,
    // behind the K-split partials' 32 KB, once every wave has finished reading it)
    float* sA1 = UB == 4 ? sH + 8192 : sH + NSH * 256;   // [ns2][64][4]: head layer 1 activations as A fragments
    float* sY = UB == 4 ? sH + NSH * 256 : sA1 + a.ns2 * 256;   // [32][33] head outputs
    int* sRow = reinterpret_cast<int*>(KM > 1 ? fused_lds : sY + FR * 33);   // [32] state row of every particle of the panel
    float* sPart = sH;                             // [8][32][32] K-split partials of head layer 2 (the tile is dead by then)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c31 = lane & 31, hh = lane >> 5;
    const int part = KM > 1 ? (int)blockIdx.x / a.panels : 0;
    const int u0 = part * H;                       // first hidden unit of this workgroup
    const int m0 = ((int)blockIdx.x - part * (KM > 1 ? a.panels : 0)) * FR;
    const int dbg_slot = a.dbg ? (blockIdx.x == 0 ? 0 : ((int)blockIdx.x == (int)gridDim.x / 2 ? 1 : -1)) : -1;
#define FUSED_STAMP(k)                                                                          \
    do {                                                                                        \
        if (dbg_slot >= 0 && lane == 0 && (wave == 0 || wave == 5))                             \
            a.dbg[(dbg_slot * 2 + (wave == 5 ? 1 : 0)) * 16 + (k)] = clock64();                 \
    } while (0)
    FUSED_STAMP(0);
    if (tid < FR) {
        const int gr = min(m0 + tid, a.n - 1);
        sRow[tid] = a.rows ? (int)a.rows[gr] : gr;
    }
    if constexpr (HEADONLY) {
        __syncthreads();   // sRow
        for (int e = tid; e < FR * H; e += 512) {
            const int row = e / H, u = e - row * H;
            const bool live = m0 + row < a.n;
            const float hv = a.hn[(int64_t)min(m0 + row, a.n - 1) * H + u];
            if (live) a.h[(int64_t)sRow[row] * H + u] = hv;      // (64-bit: n H reaches 2^32 at 8.4 M particles of H = 512)
            const int hslot = ((u >> 4) * 128 + ((u >> 2) & 3) * 16) * 4 + (u & 3);
            sH[hslot + (row >> 4) * 256 + (row & 15) * 4] = hv;
        }
    } else {
    const int gr = min(m0 + c31, a.n - 1);              // this lane's particle (A operand row)
    const int64_t ridx = a.rows ? a.rows[gr] : (int64_t)gr;
    const float* arow = a.h + ridx * HK + 4 * hh;
    const float* bimg = a.whh_img + (size_t)part * ((size_t)(NSK + 1) * SLAB) + (size_t)wave * (NB * 256) + lane * 4;

    f32x16 acc[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        const int g = blk / UB, ub = blk % UB;
        const float b = a.bias[g * HK + u0 + (wave * UB + ub) * 32 + c31];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[blk][r] = b;
    }
    auto load_blk = [&](int s, int blk) { return *reinterpret_cast<const f32x4*>(bimg + (size_t)s * SLAB + blk * 256); };
    auto load_a = [&](int s) { return *reinterpret_cast<const f32x4*>(arow + 8 * s); };
    // One slab: block by block, the four MFMAs of a block (k pairs j = 0..3) and then - REFILL - the load of the SAME block of
    // the slab two steps ahead into the registers those MFMAs just read. The loads are spread one per four MFMAs (a burst of
    // nine loads per slab stalled the wave in the memory pipeline's issue queue while its MFMAs could have run) and every
    // fragment has almost two slab times to arrive: the register ring is two slabs deep without holding more than two slabs.
#define FUSED_SLAB(AV, BUF, REFILL, SN)                                                                         \
    _Pragma("unroll") for (int blk = 0; blk < NB; ++blk) {                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                           \
            acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[j], BUF[blk][j], acc[blk], 0, 0, 0);             \
        if (REFILL) BUF[blk] = load_blk(SN, blk);                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    }
    constexpr int NS = SHARED ? 0 : NSK;   // shared state: h W_hh^T is one row for everybody, part of the bias
    f32x4 b0[NB], b1[NB], a0, a1, a2, a3;
    // item 0 of the stream: the sample embedding of the previous value, k = 4 hh + j < smp_dim (embedding_feedforward.py: one
    // Linear + ReLU; gather.hpp sample_embed_elem: a Linear(1, smp_dim) of the value, or a row of the one-hot
    // Linear(C, smp_dim)); its weights are slab NSH of the image. Items 1 .. NS: the slabs 0 .. NS - 1 of h W_hh^T.
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) b0[blk] = load_blk(NSK, blk);
    {
        const float pv = a.prev_value[a.prev_indexed ? ridx : (int64_t)gr];
        int cat = (int)pv;
        cat = cat < 0 ? 0 : (cat >= a.smp_in ? a.smp_in - 1 : cat);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = min(4 * hh + j, a.smp - 1);
            const float e = a.smp_in == 1 ? a.smp_w[k] * pv + a.smp_b[k] : a.smp_w[k * a.smp_in + cat] + a.smp_b[k];
            a0[j] = (4 * hh + j < a.smp) ? relu_keep_nan(e) : 0.0f;
        }
    }
    if constexpr (NS != 0) {
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) b1[blk] = load_blk(0, blk);
        a1 = load_a(0);
        __builtin_amdgcn_sched_barrier(0);
        int it = 0;      // item `it` sits in b0 / a0, item it + 1 in b1 / a1
        for (; it + 3 <= NS; it += 2) {
            a2 = load_a(it + 1);
            __builtin_amdgcn_sched_barrier(0);
            FUSED_SLAB(a0, b0, true, it + 1)      // consumes item it, refills with item it + 2 = slab it + 1
            a3 = load_a(it + 2);
            __builtin_amdgcn_sched_barrier(0);
            FUSED_SLAB(a1, b1, true, it + 2)
            a0 = a2;
            a1 = a3;
            // Not for data (the K loop shares nothing): the barrier keeps the two waves of a SIMD in step. Left alone, the older
            // wave wins every issue arbitration, finishes its K loop at 0.55 of the pair's time and the younger one runs the rest
            // by itself with nothing to hide its load latency behind (K loop 312 k cycles; with the barrier both take 287 k of
            // 266 k of MFMA time; one barrier per slab instead of per two: the same; s_setprio per slab: no effect)
            __builtin_amdgcn_s_barrier();
        }
        a2 = load_a(NS - 1);                      // it == NS - 2: items NS - 2, NS - 1, NS remain
        __builtin_amdgcn_sched_barrier(0);
        FUSED_SLAB(a0, b0, true, NS - 1)
        FUSED_SLAB(a1, b1, false, 0)
        FUSED_SLAB(a2, b0, false, 0)
    } else {
        __builtin_amdgcn_sched_barrier(0);
        FUSED_SLAB(a0, b0, false, 0)
    }
#undef FUSED_SLAB
    // Gate activations in place on the accumulators, before the barrier (a wave that is early does them while it would wait):
    // acc[i] <- sigmoid(i) tanh(g), acc[f] <- sigmoid(f), acc[o] <- sigmoid(o); the g accumulators are dead afterwards, which is
    // what keeps the cell phase below the register budget
#pragma unroll
    for (int ub = 0; ub < UB; ++ub)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float gi = fast_sigmoid(acc[0 * UB + ub][r]);
            const float gg = fast_tanh(acc[2 * UB + ub][r]);
            acc[0 * UB + ub][r] = gi * gg;
            acc[1 * UB + ub][r] = fast_sigmoid(acc[1 * UB + ub][r]);
            acc[3 * UB + ub][r] = fast_sigmoid(acc[3 * UB + ub][r]);
        }
    FUSED_STAMP(1);    // K loop done
    __syncthreads();   // every wave has read the old h rows of the panel (and sRow is visible)
    FUSED_STAMP(2);

    // ---- LSTM cell on the accumulators (torch.nn.LSTM gate order i, f, g, o) ----
    // Two passes over the accumulator rows (r < 8, r >= 8): a pass's loads of the old cell state are issued together and BEFORE
    // its stores (which go through the same pointer and would pin every later load behind them); 32-bit element offsets keep
    // the address registers at one per row (state buffers of up to 2^32 elements: 8 M particles at H = 512).
    const float* cprev = SHARED ? a.c0 : a.c;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        uint32_t off[8];
        float cp[UB][8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = 8 * half + q;
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            off[q] = (uint32_t)sRow[row] * (uint32_t)HK;
        }
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) {
            const uint32_t u = (uint32_t)(u0 + (wave * UB + ub) * 32 + c31);
#pragma unroll
            for (int q = 0; q < 8; ++q) cp[ub][q] = cprev[SHARED ? u : off[q] + u];
        }
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) {
            const int u = u0 + (wave * UB + ub) * 32 + c31;
            // A fragments of v_mfma_f32_16x16x4_f32 for head layer 1: [16-k slab u / 16][row block][k group (u / 4) % 4][row % 16][u % 4]
            const int hslot = ((u >> 4) * 128 + ((u >> 2) & 3) * 16) * 4 + (u & 3);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = 8 * half + q;
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float cn = acc[1 * UB + ub][r] * cp[ub][q] + acc[0 * UB + ub][r];
                const float hn = acc[3 * UB + ub][r] * fast_tanh(cn);
                if constexpr (KM > 1) {
                    if (m0 + row < a.n) {
                        a.c[off[q] + (uint32_t)u] = cn;
                        a.hout[(uint32_t)(m0 + row) * (uint32_t)HK + (uint32_t)u] = hn;
                    }
                } else {
                    if (m0 + row < a.n) {
                        a.c[off[q] + (uint32_t)u] = cn;
                        a.h[off[q] + (uint32_t)u] = hn;
                    }
                    sH[hslot + (row >> 4) * 256 + (row & 15) * 4] = hn;
                }
            }
        }
    }
    }
    FUSED_STAMP(3);    // cell done
    if constexpr (KM > 1) return;
    __syncthreads();
    FUSED_STAMP(4);

    // ---- head layer 1: a1 = relu(h W1^T + b1) on v_mfma_f32_16x16x4_f32: 2 x nb16 tiles of 16 rows x 16 columns, wave w
    // takes the contiguous tile range [w T / 8, (w + 1) T / 8) (hid = 271: 34 tiles, 4 or 5 per wave; with 32-column blocks one
    // wave had two of nine blocks and the other seven waited for it, profiles/r04e_is_step_timeline.txt). Two accumulators
    // per tile (even / odd k groups) keep the dependent-accumulate latency out of the chain; the B fragments (one KB per
    // 16-k slab and tile, W1's second image) run eight slabs ahead in registers.
    if constexpr (UB == 4) {
        // (H = 1024: the tiles' results wait in registers for a barrier - see sA1 above; up to MAXT tiles per wave: hid <= 576;
        // pyprob's proposal layer is (H + 3 K) / 2 = 527 wide at H = 1024, K = 10)
        constexpr int MAXT = 9;
        const int T16 = 2 * a.nb16;
        const int t0 = (wave * T16) / FW, t1 = ((wave + 1) * T16) / FW;
        constexpr int NS16 = H / 16;
        constexpr int HR = 16;
        const int i16 = lane & 15, kq = lane >> 4;
        f32x4 res[MAXT];
#pragma unroll
        for (int ti = 0; ti < MAXT; ++ti) {
            res[ti] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            const int t = t0 + ti;
            if (t < t1) {      // (wave-uniform)
                const int cb = t >> 1, rb = t & 1;
                f32x4 e0 = {0.0f, 0.0f, 0.0f, 0.0f}, e1 = {0.0f, 0.0f, 0.0f, 0.0f};
                const float* wimg = a.w1_img + (size_t)cb * 256 + lane * 4;
                const size_t sstride = (size_t)a.nb16 * 256;
                const float* aimg = sH + rb * 256 + lane * 4;
                f32x4 bq[HR];
#pragma unroll
                for (int u = 0; u < HR; ++u) bq[u] = *reinterpret_cast<const f32x4*>(wimg + u * sstride);
                __builtin_amdgcn_sched_barrier(0);
                for (int s16 = 0; s16 < NS16; s16 += HR) {
#pragma unroll
                    for (int u = 0; u < HR; ++u) {
                        const f32x4 av = *reinterpret_cast<const f32x4*>(aimg + (s16 + u) * 512);
                        const f32x4 bv = bq[u];
                        if (s16 + u + HR < NS16) bq[u] = *reinterpret_cast<const f32x4*>(wimg + (size_t)(s16 + u + HR) * sstride);
                        e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], e0, 0, 0, 0);
                        e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], e1, 0, 0, 0);
                        e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], e0, 0, 0, 0);
                        e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], e1, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) res[ti][r] = e0[r] + e1[r];
            }
        }
        __syncthreads();      // every wave has read the hidden tile: the activations go over it
#pragma unroll
        for (int ti = 0; ti < MAXT; ++ti) {
            const int t = t0 + ti;
            if (t < t1) {
                const int cb = t >> 1, rb = t & 1;
                const int col = cb * 16 + i16;
                const float bias1 = col < a.hid ? a.b1[col] : 0.0f;
                if ((col >> 3) < a.ns2) {
                    const int slot = ((col >> 3) * 64 + ((col >> 2) & 1) * 32) * 4 + (col & 3);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rb * 16 + 4 * kq + r;
                        sA1[slot + row * 4] = col < a.hid ? relu_keep_nan(res[ti][r] + bias1) : 0.0f;
                    }
                }
            }
        }
    } else
    {
        const int T16 = 2 * a.nb16;
        const int t0 = (wave * T16) / FW, t1 = ((wave + 1) * T16) / FW;
        constexpr int NS16 = H / 16;
        const int i16 = lane & 15, kq = lane >> 4;
        for (int t = t0; t < t1; ++t) {
            const int cb = t >> 1, rb = t & 1;
            f32x4 e0 = {0.0f, 0.0f, 0.0f, 0.0f}, e1 = {0.0f, 0.0f, 0.0f, 0.0f};
            const float* wimg = a.w1_img + (size_t)cb * 256 + lane * 4;
            const size_t sstride = (size_t)a.nb16 * 256;
            const float* aimg = sH + rb * 256 + lane * 4;
            // (the head-only launch of a split statement has the registers for a ring of 16: its tiles are a latency chain -
            // one small workgroup per 32 particles, nothing else on the CU to hide it)
            constexpr int HR = HEADONLY ? 16 : 8;
            f32x4 bq[HR];
#pragma unroll
            for (int u = 0; u < HR; ++u) bq[u] = *reinterpret_cast<const f32x4*>(wimg + u * sstride);
            // (head-only: the scheduler otherwise sinks every load next to its use - fewer registers, the latency of all 32
            // slabs in a row; the one-kernel statement is left as measured)
            if constexpr (HEADONLY) __builtin_amdgcn_sched_barrier(0);
            for (int s16 = 0; s16 < NS16; s16 += HR) {
#pragma unroll
                for (int u = 0; u < HR; ++u) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(aimg + (s16 + u) * 512);
                    const f32x4 bv = bq[u];
                    if (s16 + u + HR < NS16) bq[u] = *reinterpret_cast<const f32x4*>(wimg + (size_t)(s16 + u + HR) * sstride);
                    e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], e0, 0, 0, 0);
                    e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], e1, 0, 0, 0);
                    e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], e0, 0, 0, 0);
                    e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], e1, 0, 0, 0);
                    if constexpr (HEADONLY) __builtin_amdgcn_sched_barrier(0);
                }
            }
            // D of 16x16x4: lane (column i16, row group kq) holds rows 4 kq + r of the tile
            const int col = cb * 16 + i16;
            const float bias1 = col < a.hid ? a.b1[col] : 0.0f;
            if ((col >> 3) < a.ns2) {
                const int slot = ((col >> 3) * 64 + ((col >> 2) & 1) * 32) * 4 + (col & 3);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rb * 16 + 4 * kq + r;
                    sA1[slot + row * 4] = col < a.hid ? relu_keep_nan(e0[r] + e1[r] + bias1) : 0.0f;
                }
            }
        }
    }
    FUSED_STAMP(5);    // head layer 1 done (this wave)
    __syncthreads();
    FUSED_STAMP(6);

    // ---- head layer 2: y = a1 W2^T + b2, K split over the waves, partials summed in a fixed order ----
    {
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.0f;
        if constexpr (HEADONLY) {      // this wave's (at most 8) slabs of W2 in flight together instead of one after the other
            f32x4 bw[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) bw[q] = *reinterpret_cast<const f32x4*>(a.w2_img + (size_t)min(wave + q * FW, a.ns2 - 1) * 256 + lane * 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int s = wave + q * FW;
                if (s < a.ns2) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(sA1 + s * 256 + lane * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bw[q][j], acc2, 0, 0, 0);
                }
            }
            for (int s = wave + 8 * FW; s < a.ns2; s += FW) {      // (heads wider than 512 hidden units)
                const f32x4 av = *reinterpret_cast<const f32x4*>(sA1 + s * 256 + lane * 4);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.w2_img + (size_t)s * 256 + lane * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc2, 0, 0, 0);
            }
        } else
        for (int s = wave; s < a.ns2; s += FW) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(sA1 + s * 256 + lane * 4);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.w2_img + (size_t)s * 256 + lane * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            sPart[(wave * 32 + row) * 32 + c31] = acc2[r];
        }
    }
    __syncthreads();
    for (int e = tid; e < FR * 32; e += 512) {
        const int row = e >> 5, col = e & 31;
        float y = col < a.n_out ? a.b2[col] : 0.0f;
#pragma unroll
        for (int w = 0; w < FW; ++w) y += sPart[(w * 32 + row) * 32 + col];
        sY[row * 33 + col] = y;
        if (a.y_out && m0 + row < a.n && col < a.n_out) a.y_out[(int64_t)(m0 + row) * a.ldy + col] = y;
    }
    FUSED_STAMP(7);    // head layer 2 + combine
    if (KIND == 3) return;
    __syncthreads();

    // ---- draw + log q (Mixture.sample / log_prob, is_draw.hpp mixture_particle re-arranged): sixteen lanes per particle,
    // lane k owns component k; maxima, sums and the inclusive prefix of the component probabilities cross the sixteen lanes
    // through DPP row operations. (One lane per particle ran ~2 500 instructions on half a wave while seven waves and the
    // CU's LDS waited: 22 000 cycles per panel.) ----
    {
        const int row = tid >> 4, k = tid & 15;
        const int64_t i = m0 + row;
        const bool live = i < a.n;
        const int64_t ic = live ? i : a.n - 1;
        const int K = a.K;
        const float* y = sY + row * 33;
        const float pa = a.prior[ic * 2 * a.prior_stride], pb = a.prior[ic * 2 * a.prior_stride + 1];
        const bool comp = k < K;
        const float z = comp ? y[2 * K + k] : -INFINITY;
        const float zmax = row16_max(z);
        float p = comp ? expf(z - zmax) : 0.0f;
        p = p / row16_sum(p);
        p = p / row16_sum(p);
        float mu, sd;
        if (KIND == 0) {
            mu = pa + (comp ? y[k] : 0.0f) * pb;
            sd = expf(comp ? y[K + k] : 0.0f) * pb;
        } else {
            const float rng = pb - pa;
            mu = pa + sigmoidf_(comp ? y[k] : 0.0f) * rng;
            sd = KIND == 2 ? expf(comp ? y[K + k] : 0.0f) : rng / 1000.0f + sigmoidf_(comp ? y[K + k] : 0.0f) * rng * 10.0f;
        }
        float ca = 0.0f, cb = 1.0f;
        if (KIND != 0) {
            ca = std_cdf((pa - mu) / sd);
            cb = std_cdf((pb - mu) / sd);
        }
        float v;
        if (a.value_in) {
            v = a.value_in[ic];
        } else {
            const float cum = row16_prefix(p);                       // inclusive prefix over the components
            Philox rng(a.seed, a.offset + (uint64_t)ic, 0x1C);
            v = NAN;
            bool done = false;
            for (int attempt = 0; attempt < 64; ++attempt) {
                if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
                uint32_t r[4];
                rng.next(r);
                const float u0 = u01(r[0]), u1 = u01(r[1]), u2 = u01(r[2]);
                // component ~ Categorical(p): the first k with u0 < cum_k, else the last one (mixture.py:47-63)
                const int kk = row16_min_int((comp && u0 < cum) ? k : K - 1);
                const int src = (lane & ~15) + kk;
                const float mk = __shfl(mu, src, 64), sk = __shfl(sd, src, 64);
                float cand;
                bool ok;
                if (KIND == 0) {
                    cand = mk + sk * sqrtf(-2.0f * logf(u1)) * cosf(kTwoPi * u2);   // Box-Muller
                    ok = true;
                } else {      // inverse-CDF draw inside [low, high) with rejection (truncated_normal.py:94-112)
                    const float cak = __shfl(ca, src, 64), cbk = __shfl(cb, src, 64);
                    const float uu = cak + u1 * (cbk - cak);
                    cand = mk + sk * kSqrt2 * erfinvf(2.0f * uu - 1.0f);
                    ok = isfinite(cand) && cand >= pa && cand < pb;
                }
                if (!done && ok) {
                    v = cand;
                    done = true;
                }
                if (KIND == 0) break;
            }
        }
        // log q(v) = logsumexp_k (log p_k + log f_k(v))   (mixture.py:42-44)
        const bool inside = (KIND == 0) || (v >= pa && v <= pb);
        const float lpk = logf(fminf(fmaxf(p, kFp32Eps), 1.0f - kFp32Eps));
        const float tt = (v - mu) / sd;
        float term;
        if (KIND == 0) term = -0.5f * tt * tt - logf(sd) - kHalfLog2Pi;
        else term = (inside ? 0.0f : -INFINITY) + (-0.5f * tt * tt - kHalfLog2Pi) - logf(sd * (cb - ca));
        const float ak = comp ? lpk + term : -INFINITY;
        const float amax = row16_max(ak);
        float lp = amax;
        if (amax > -INFINITY) lp = amax + logf(row16_sum(comp ? expf(ak - amax) : 0.0f));
        if (live && k == 0) {
            if (a.value_out) a.value_out[i] = v;
            if (a.logq_out) a.logq_out[i] = lp;
            if (a.value_full) {
                const int64_t ri = sRow[row];
                a.value_full[ri] = v;
                // + log p(v) of the program's own prior, then - log q(v): two fp32 additions in the order of the separate
                // log-weight kernels (pp_logweight_accumulate, pp_axpy)
                float plp;
                if (a.prior_kind == 0) {
                    const float d = v - pa;
                    plp = -(d * d) / (2.0f * pb * pb) - logf(pb) - kHalfLog2Pi;
                } else {
                    plp = (v >= pa && v < pb) ? -logf(pb - pa) : -INFINITY;
                }
                float l = a.lw_full[ri];
                l += plp;
                l += -1.0f * lp;
                a.lw_full[ri] = l;
            }
        }
    }
    FUSED_STAMP(8);    // draw + log q
#undef FUSED_STAMP
}

// ---- split statement: the LSTM step of a SMALL launch, split over the gate columns ----------------------------------------------
// A launch of a few thousand particles is one generation of workgroups of the kernel above, each streaming ALL 4.7 MB of weights
// through one CU for its 32 particles: 0.23 ms however few they are (profiles/r04d_is_step_small_n.jsonl) - the late iterations of
// a program with stochastic control flow run 2 000, 400, 100 ... particles per statement. Here a workgroup owns 32 particles x
// ONE column group (the 4 x 64 gate columns the big kernel gives to one wave), its waves one 32-column block each: 8 x as many
// workgroups, each streaming an eighth of the weights - the same fragment images, the same k order per accumulator (bit-identical
// gate pre-activations). The gates meet in LDS for the cell; c is updated in place, the new h goes to a.hn ([n][H], compact):
// other column groups still read the old h rows, so the state rows are written by the second launch (is_step_fused_kernel
// <.., HEADONLY>: head layers, draw, log q, whole-statement tail - the code above from head layer 1 on).
struct SmallLstmArgs {
    const float* whh_img; const float* bias;
    const float* h; float* c;
    const int64_t* rows;
    const float* prev_value; int prev_indexed;
    const float* smp_w; const float* smp_b; int smp_in, smp;
    float* hn;
    int n;
};

template <int UB>
__global__ __launch_bounds__(256 * UB) void is_small_lstm_kernel(const SmallLstmArgs a) {
    constexpr int H = 256 * UB;
    constexpr int NSH = H / 8;
    constexpr int NB = 4 * UB;
    constexpr int SLAB = FW * NB * 256;
    constexpr int RING = 16;     // items in flight per wave: 65 items of ~0.1 us of MFMAs behind ~2 us of load latency (ring 4: 28 us per launch)
    __shared__ float sG[NB][32][33];
    __shared__ int sRow[32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int blk = __builtin_amdgcn_readfirstlane(tid >> 6);      // this wave's block: gate blk / UB, unit block blk % UB
    const int c31 = lane & 31, hh = lane >> 5;
    const int cg = blockIdx.x & (FW - 1);                          // column group = the big kernel's wave index
    const int m0 = (blockIdx.x / FW) * FR;
    if (tid < FR) {
        const int gr = min(m0 + tid, a.n - 1);
        sRow[tid] = a.rows ? (int)a.rows[gr] : gr;
    }
    const int gr = min(m0 + c31, a.n - 1);
    const int64_t ridx = a.rows ? a.rows[gr] : (int64_t)gr;
    const float* arow = a.h + ridx * H + 4 * hh;
    const float* bimg = a.whh_img + (size_t)cg * (NB * 256) + blk * 256 + lane * 4;
    const int g = blk / UB, ub = blk % UB;
    f32x16 acc;
    {
        const float b = a.bias[g * H + (cg * UB + ub) * 32 + c31];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b;
    }
    auto load_b = [&](int s) { return *reinterpret_cast<const f32x4*>(bimg + (size_t)s * SLAB); };
    auto load_a = [&](int s) { return *reinterpret_cast<const f32x4*>(arow + 8 * s); };
    f32x4 ra[RING], rb[RING];
    const f32x4 bs = load_b(NSH);
#pragma unroll
    for (int i = 0; i < RING; ++i) {
        ra[i] = load_a(i);
        rb[i] = load_b(i);
    }
    f32x4 as;      // item 0: the sample embedding of the previous value (the big kernel's a0)
    {
        const float pv = a.prev_value[a.prev_indexed ? ridx : (int64_t)gr];
        int cat = (int)pv;
        cat = cat < 0 ? 0 : (cat >= a.smp_in ? a.smp_in - 1 : cat);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = min(4 * hh + j, a.smp - 1);
            const float e = a.smp_in == 1 ? a.smp_w[k] * pv + a.smp_b[k] : a.smp_w[k * a.smp_in + cat] + a.smp_b[k];
            as[j] = (4 * hh + j < a.smp) ? relu_keep_nan(e) : 0.0f;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[j], bs[j], acc, 0, 0, 0);
    for (int s = 0; s < NSH; s += RING) {
#pragma unroll
        for (int i = 0; i < RING; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[i][j], rb[i][j], acc, 0, 0, 0);
            // REFILL into the registers those MFMAs just read (loading first needs a second register set and a copy per
            // item); unconditional loads: the last ones are repeats nobody uses
            const int sn = min(s + i + RING, NSH - 1);
            ra[i] = load_a(sn);
            rb[i] = load_b(sn);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
        sG[blk][row][c31] = g == 2 ? fast_tanh(acc[r]) : fast_sigmoid(acc[r]);
    }
    __syncthreads();
    // cell (torch.nn.LSTM gate order i, f, g, o): 32 rows x 32 UB units of this column group
    for (int e = tid; e < FR * 32 * UB; e += 256 * UB) {
        const int row = e / (32 * UB), uu = e - row * (32 * UB);
        const int ub2 = uu >> 5, cc = uu & 31;
        if (m0 + row >= a.n) continue;
        const int u = (cg * UB + ub2) * 32 + cc;
        const uint32_t off = (uint32_t)sRow[row] * (uint32_t)H + (uint32_t)u;
        const float ig = sG[0 * UB + ub2][row][cc] * sG[2 * UB + ub2][row][cc];
        const float cn = sG[1 * UB + ub2][row][cc] * a.c[off] + ig;
        a.c[off] = cn;
        a.hn[(int64_t)(m0 + row) * H + u] = sG[3 * UB + ub2][row][cc] * fast_tanh(cn);
    }
}

template <int UB, int KIND, bool SHARED, bool HEADONLY = false>
int launch_fused_s(const FusedArgs& a, size_t lds, hipStream_t st) {
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&is_step_fused_kernel<UB, KIND, SHARED, HEADONLY>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            set_error("pp_is_step: cannot raise the LDS limit of the fused statement kernel: %s", hipGetErrorString(e));
            return (int)e;
        }
        raised = true;
    }
    hipLaunchKernelGGL((is_step_fused_kernel<UB, KIND, SHARED, HEADONLY>), dim3(cdiv(a.n, FR)), dim3(512), lds, st, a);
    return 0;
}
template <int UB, int KIND>
int launch_fused(const FusedArgs& a, size_t lds, hipStream_t st) {
    if (a.hn) return launch_fused_s<UB, KIND, false, true>(a, lds, st);
    return a.state_shared ? launch_fused_s<UB, KIND, true>(a, lds, st) : launch_fused_s<UB, KIND, false>(a, lds, st);
}

static inline int64_t round256(int64_t x) { return (x + 255) & ~int64_t(255); }

}  // namespace

bool is_step_fused_supported(const pp_net* net, int addr_id) {
    if (is_step_small_supported(net, addr_id)) return true;
    if (!net || net->lstm_dim == 0 || std::max(1, (int)net->lstm_depth) != 1) return false;
    if (net->lstm_dim != 256 && net->lstm_dim != 512 && net->lstm_dim != 1024) return false;
    if (net->smp_dim < 1 || net->smp_dim > 8 || !net->addr_table) return false;
    if (addr_id < 0 || addr_id >= net->n_addr) return false;
    const pp_addr& ad = net->addrs[addr_id];
    if (ad.n_out < 1 || ad.n_out > 32 || ad.hid < 1) return false;
    // H = 1024: the wide LSTM launch + the head-only launch (its layer-1 tiles wait in registers: at most 9 per wave, and the
    // activations go over the hidden tile behind the 32 KB of K-split partials)
    if (net->lstm_dim == 1024) return ad.hid <= 576;
    const int ns2 = (ad.hid + 7) / 8;
    const size_t lds = ((size_t)(net->lstm_dim / 8) * 256 + (size_t)ns2 * 256 + FR * 33 + FR) * sizeof(float);
    return lds <= 150 * 1024;
}

bool is_lstm_wide_supported(const pp_net* net) {
    if (!net || net->lstm_dim != 1024 || std::max(1, (int)net->lstm_depth) != 1) return false;
    return net->smp_dim >= 1 && net->smp_dim <= 8 && net->addr_table;
}

void is_fused_carve_sizes(const pp_net* net, IsFusedBuffers& f) {
    f = IsFusedBuffers{};
    if (is_small_network(net)) {
        is_small_carve_sizes(net, f);
        return;
    }
    if (is_lstm_wide_supported(net)) {      // two gate images of 512 hidden units each, K extent 1024; the head images
        const int H = net->lstm_dim;
        int64_t hid = 1;
        for (int a = 0; a < net->n_addr; ++a) hid = std::max<int64_t>(hid, net->addrs[a].hid);
        f.n_whh = (int64_t)2 * (H / 8 + 1) * FW * 8 * 256;
        f.n_w1 = (int64_t)(H / 16) * ((hid + 15) / 16) * 256;
        f.n_w2 = (int64_t)((hid + 7) / 8) * 256;
        f.n_bias = 4 * H;
        return;
    }
    if (!net || (net->lstm_dim != 256 && net->lstm_dim != 512)) return;
    const int H = net->lstm_dim, ub = H / 256, nsh = H / 8;
    int64_t hid = 1;
    for (int a = 0; a < net->n_addr; ++a) hid = std::max<int64_t>(hid, net->addrs[a].hid);
    f.n_whh = (int64_t)(nsh + 1) * FW * 4 * ub * 256;
    f.n_w1 = (int64_t)(H / 16) * ((hid + 15) / 16) * 256;
    f.n_w2 = (int64_t)((hid + 7) / 8) * 256;
    f.n_bias = 4 * H;
}

int is_step_fused(const pp_net* net, const float* P, int addr_id, int prev_addr_id, int n, const float* e_obs_vec,
                  const float* prev_value, const float* prior, int prior_stride, float* h, float* c, int state_rows,
                  const int64_t* rows, const float* value_in, float* value_out, float* logq_out, uint64_t seed, uint64_t offset,
                  const IsFusedBuffers& f, float* c0_copy, float* y_out, int64_t ldy, bool net_only, bool* sampled, hipStream_t st,
                  const IsStatementOut* whole, float* hn_split, int64_t layer_rows) {
    if (is_step_small_supported(net, addr_id))
        return is_step_small(net, P, addr_id, prev_addr_id, n, e_obs_vec, prev_value, prior, prior_stride, h, c, state_rows,
                             layer_rows > 0 ? layer_rows : n, rows, value_in, value_out, logq_out, seed, offset, f, y_out, ldy, net_only,
                             sampled, st, whole);
    const pp_addr& ad = net->addrs[addr_id];
    const bool wide = net->lstm_dim == 1024;      // the LSTM step as the wide launch, then the head-only launch (hn_split required)
    const int H = net->lstm_dim, ub = wide ? 2 : H / 256, nsh = H / 8;
    const bool shared = state_rows == 1;
    if (wide && !hn_split) {
        set_error("pp_is_step: the H = 1024 statement needs the scratch rows of its two launches");
        return PP_EINVAL;
    }
    PrepArgs p{};
    p.P = P; p.at = net->addr_table;
    p.w_ih = net->w_ih; p.w_hh = net->w_hh; p.b_ih = net->b_ih; p.b_hh = net->b_hh; p.w1 = ad.w1; p.w2 = ad.w2;
    p.H = H; p.I = net->lstm_in; p.ub = ub; p.nsh = nsh; p.hu = wide ? 512 : H; p.halves = wide ? 2 : 1;
    p.d = GatherDims{net->e_obs, net->smp_dim, net->dtype_dim, net->addr_dim, net->lstm_in};
    p.addr_id = addr_id; p.prev_addr = prev_addr_id;
    p.e_obs_vec = e_obs_vec;
    p.h0 = shared ? h : nullptr;
    p.c0 = shared ? c : nullptr;
    p.hid = ad.hid; p.n_out = ad.n_out; p.nb16 = (ad.hid + 15) / 16; p.ns2 = (ad.hid + 7) / 8;
    p.whh_img = f.whh; p.w1_img = f.w1; p.w2_img = f.w2; p.bias = f.bias; p.c0_copy = c0_copy;
    p.q_whh = (int64_t)p.halves * (nsh + 1) * FW * 4 * ub * 64;
    p.q_w1 = (int64_t)(H / 16) * p.nb16 * 64;
    p.q_w2 = (int64_t)p.ns2 * 64;
    p.img_blocks = (int)std::min<int64_t>(wide ? 2048 : 1024, (p.q_whh + p.q_w1 + p.q_w2 + 255) / 256);
    hipLaunchKernelGGL(is_prep_kernel, dim3(p.img_blocks + H), dim3(256), 0, st, p);
    PP_LAUNCH_CHECK("pp_is_step(prepare)");

    FusedArgs a{};
    a.whh_img = f.whh; a.bias = f.bias;
    a.h = h; a.c = c; a.c0 = c0_copy; a.rows = rows; a.state_shared = shared ? 1 : 0;
    const pp_addr& pad = net->addrs[prev_addr_id];
    a.prev_value = prev_value; a.smp_w = P + pad.smp_w; a.smp_b = P + pad.smp_b; a.smp_in = pad.smp_in; a.smp = net->smp_dim;
    a.w1_img = f.w1; a.b1 = P + ad.b1; a.hid = ad.hid; a.nb16 = p.nb16;
    a.w2_img = f.w2; a.b2 = P + ad.b2; a.n_out = ad.n_out; a.ns2 = p.ns2;
    a.prior = prior; a.prior_stride = prior_stride;
    a.value_in = value_in; a.value_out = value_out; a.logq_out = logq_out;
    a.seed = seed; a.offset = offset; a.K = ad.n_out / 3; a.n = n;
    a.dbg = g_timeline;
    if (whole) {
        a.prev_indexed = 1;
        a.value_full = whole->value_full;
        a.lw_full = whole->lw_full;
        a.prior_kind = whole->prior_kind;
    }
    int kind = 3;
    if (!net_only && ad.n_out % 3 == 0 && ad.n_out / 3 <= MAXK) {
        if (ad.kind == PP_HEAD_NORMAL_MIXTURE) kind = 0;
        else if (ad.kind == PP_HEAD_TRUNCNORMAL_MIXTURE) kind = 1;
        else if (ad.kind == PP_HEAD_POISSON_TN_MIXTURE) kind = 2;
    }
    if (kind == 3) { a.y_out = y_out; a.ldy = ldy; }
    const size_t lds = wide ? ((size_t)nsh * 256 + FR * 33 + FR) * sizeof(float)      // (the activations lie over the hidden tile)
                            : ((size_t)nsh * 256 + (size_t)p.ns2 * 256 + FR * 33 + FR) * sizeof(float);
    // kernel class 5 of the in-stream timing: the fused statement (work = FLOPs of the reference's algorithm, SURVEY.md 8d:
    // input + recurrent product, both head layers)
    const double flops = (double)n * (2.0 * (net->lstm_in + (shared ? 0 : H)) * 4.0 * H + 2.0 * ((double)H * ad.hid + (double)ad.hid * ad.n_out));
    prof_begin(5, st);
    int rc = 0;
    if (wide) {      // H = 1024: the LSTM step as the wide launch (KM = 2; shared state: no recurrent product), then the head-only launch
        FusedArgs q = a;
        q.hout = hn_split; q.panels = cdiv(n, FR);
        if (shared) hipLaunchKernelGGL((is_step_fused_kernel<2, 3, true, false, 2>), dim3(2 * q.panels), dim3(512), 256, st, q);
        else hipLaunchKernelGGL((is_step_fused_kernel<2, 3, false, false, 2>), dim3(2 * q.panels), dim3(512), 256, st, q);
        PP_LAUNCH_CHECK("pp_is_step(wide LSTM)");
        a.hn = hn_split;
        a.state_shared = 0;      // (the head-only launch writes every particle's new row)
        if (kind == 0) rc = launch_fused_s<4, 0, false, true>(a, lds, st);
        else if (kind == 1) rc = launch_fused_s<4, 1, false, true>(a, lds, st);
        else if (kind == 2) rc = launch_fused_s<4, 2, false, true>(a, lds, st);
        else rc = launch_fused_s<4, 3, false, true>(a, lds, st);
        prof_end(5, flops, st);
        if (rc) return rc;
        PP_LAUNCH_CHECK("pp_is_step(fused statement, H = 1024)");
        *sampled = kind != 3;
        return 0;
    }
    if (hn_split && !shared) {      // split statement: the LSTM step over 8 x as many workgroups, then the head-only launch
        SmallLstmArgs q{};
        q.whh_img = f.whh; q.bias = f.bias; q.h = h; q.c = c; q.rows = rows;
        q.prev_value = prev_value; q.prev_indexed = a.prev_indexed;
        q.smp_w = a.smp_w; q.smp_b = a.smp_b; q.smp_in = a.smp_in; q.smp = a.smp;
        q.hn = hn_split; q.n = n;
        if (ub == 1) hipLaunchKernelGGL(is_small_lstm_kernel<1>, dim3(cdiv(n, FR) * FW), dim3(256), 0, st, q);
        else hipLaunchKernelGGL(is_small_lstm_kernel<2>, dim3(cdiv(n, FR) * FW), dim3(512), 0, st, q);
        PP_LAUNCH_CHECK("pp_is_step(split statement, LSTM)");
        a.hn = hn_split;
    }
#define PP_FUSED_CASE(U, KD) rc = launch_fused<U, KD>(a, lds, st)
    if (ub == 1) {
        if (kind == 0) PP_FUSED_CASE(1, 0); else if (kind == 1) PP_FUSED_CASE(1, 1); else if (kind == 2) PP_FUSED_CASE(1, 2); else PP_FUSED_CASE(1, 3);
    } else {
        if (kind == 0) PP_FUSED_CASE(2, 0); else if (kind == 1) PP_FUSED_CASE(2, 1); else if (kind == 2) PP_FUSED_CASE(2, 2); else PP_FUSED_CASE(2, 3);
    }
#undef PP_FUSED_CASE
    prof_end(5, flops, st);
    if (rc) return rc;
    PP_LAUNCH_CHECK("pp_is_step(fused statement)");
    *sampled = kind != 3;   // false: the head outputs are in y_out, the caller's sampling kernel follows
    return 0;
}

// The LSTM step of a statement on the H = 1024 network (one layer): is_step_fused_kernel<2, 3, SHARED, false, 2>, 2 cdiv(n, 32)
// workgroups. c in place, the new hidden rows to hn [n][H] (state_rows == 1: hn may be h itself - the old state is one row, read by
// the prepare launch only).
int is_lstm_wide(const pp_net* net, const float* P, int addr_id, int prev_addr_id, int n, const float* e_obs_vec,
                 const float* prev_value, float* h, float* c, int state_rows, const IsFusedBuffers& f, float* c0_copy, float* hn,
                 hipStream_t st) {
    const int H = net->lstm_dim, hu = 512, ub = hu / 256, nsh = H / 8;
    const bool shared = state_rows == 1;
    PrepArgs p{};
    p.P = P; p.at = net->addr_table;
    p.w_ih = net->w_ih; p.w_hh = net->w_hh; p.b_ih = net->b_ih; p.b_hh = net->b_hh;
    p.H = H; p.I = net->lstm_in; p.ub = ub; p.nsh = nsh; p.hu = hu; p.halves = H / hu;
    p.d = GatherDims{net->e_obs, net->smp_dim, net->dtype_dim, net->addr_dim, net->lstm_in};
    p.addr_id = addr_id; p.prev_addr = prev_addr_id;
    p.e_obs_vec = e_obs_vec;
    p.h0 = shared ? h : nullptr;
    p.c0 = shared ? c : nullptr;
    p.whh_img = f.whh; p.bias = f.bias; p.c0_copy = c0_copy;
    p.q_whh = (int64_t)p.halves * (nsh + 1) * FW * 4 * ub * 64;
    p.img_blocks = (int)std::min<int64_t>(2048, (p.q_whh + 255) / 256);
    hipLaunchKernelGGL(is_prep_kernel, dim3(p.img_blocks + H), dim3(256), 0, st, p);
    PP_LAUNCH_CHECK("pp_is_step(prepare, wide LSTM)");

    FusedArgs a{};
    a.whh_img = f.whh; a.bias = f.bias;
    a.h = h; a.c = c; a.c0 = c0_copy; a.state_shared = shared ? 1 : 0;
    const pp_addr& pad = net->addrs[prev_addr_id];
    a.prev_value = prev_value; a.smp_w = P + pad.smp_w; a.smp_b = P + pad.smp_b; a.smp_in = pad.smp_in; a.smp = net->smp_dim;
    a.n = n; a.hout = hn; a.panels = cdiv(n, FR);
    a.dbg = g_timeline;
    // kernel class 5 of the in-stream timing (work = FLOPs of the reference's algorithm, SURVEY.md 8d: input + recurrent product)
    const double flops = (double)n * 2.0 * (net->lstm_in + (shared ? 0 : H)) * 4.0 * H;
    prof_begin(5, st);
    if (shared) hipLaunchKernelGGL((is_step_fused_kernel<2, 3, true, false, 2>), dim3(2 * a.panels), dim3(512), 256, st, a);
    else hipLaunchKernelGGL((is_step_fused_kernel<2, 3, false, false, 2>), dim3(2 * a.panels), dim3(512), 256, st, a);
    prof_end(5, flops, st);
    PP_LAUNCH_CHECK("pp_is_step(wide LSTM)");
    return 0;
}

}  // namespace pp
