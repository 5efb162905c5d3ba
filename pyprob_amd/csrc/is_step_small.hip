// The fused importance-sampling statement for SMALL LSTMs: H = 32, 64 .. 256 hidden units (multiples of 32; tuned instantiations for
// 32, 64 and 128, one with run-time indices for the rest), 1 .. PP_MAX_LSTM_DEPTH layers
// (InferenceNetworkLSTM._infer_step pyprob/nn/inference_network_lstm.py:82-134 with nn.LSTM(I, H, depth) :31, state.sample's IC
// branch pyprob/state.py:203-219, Mixture.sample / log_prob pyprob/distributions/mixture.py:38-63) - the networks of the
// reference's own tests (lstm_dim 32 / 64, tests/test_inference.py) and BASELINE.json configs[0]. Same entry points and the same
// modes as is_step_fused.hip (row index list, shared first state, whole-statement tail, re-scoring, head-outputs-only); what
// differs is the geometry, because at these widths a wave cannot own the four gates of H / 8 units as whole 32-column blocks:
//
//   * a workgroup owns 64 particles (two 32-row blocks); wave (rb, ub) owns row block rb x the 32 hidden units [32 ub, 32 ub + 32)
//     x ALL FOUR gates: four v_mfma_f32_32x32x2_f32 accumulators, so the cell still runs on the accumulators of one lane;
//   * every A operand comes from LDS: the old hidden rows of EVERY layer are staged once at the start (one exposed memory
//     latency per workgroup, [depth][64][H + 4]); the fresh rows of layer l - 1 (the input of layer l, and at the end the
//     input of the proposal layer) live in one tile [64][H + 4] that layer l overwrites behind a barrier;
//   * the weights are small (H = 128, two layers: 0.4 MB) and stay in L2: a wave streams its four gate blocks per 8-k slab
//     from a fragment image (is_small_prep_kernel) through a register ring four slabs deep;
//   * layer 0's input is what it is in the big kernel: only the sample embedding of the previous value differs between the
//     particles of a statement - one 8-k slab; everything else of [E | s | d | a | d | a] W_ih^T is a bias row per call;
//   * the draw is is_draw.hpp's mixture_particle, one lane per particle (the chain's own tail: same Philox counters, same
//     arithmetic) on ONE wave of the workgroup - 64 particles fill it -, a different one from workgroup to workgroup;
//   * LDS: the tile, and the staged old rows with the head activations / outputs over them once the last layer has read them
//     (H = 64, one layer: 39 KB - four workgroups per CU).
#include "is_step_fused.hpp"

#include "gather.hpp"
#include "is_draw.hpp"

#include <algorithm>

namespace pp {

extern long long* g_timeline;   // kernels.hip (pp_debug_timeline)

namespace {

constexpr int RING = 3;     // k-slabs of gate fragments in flight per wave
// particles per workgroup: 32 RB - two row blocks (the drawing wave is full), one at H = 128 (four waves per workgroup: three
// workgroups per CU overlap their phases, where one workgroup of eight waves ran them one after the other)
constexpr int rows_per_block(int ubk) { return (ubk == 1 || ubk == 2) ? 64 : 32; }

__device__ __forceinline__ float fast_sigmoid_s(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh_s(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

struct SmallPrepArgs {
    const float* P;
    const int64_t* at;
    int H, I, L, ubk, nsh;
    int64_t w_ih[PP_MAX_LSTM_DEPTH], w_hh[PP_MAX_LSTM_DEPTH], b_ih[PP_MAX_LSTM_DEPTH], b_hh[PP_MAX_LSTM_DEPTH];
    GatherDims d;
    int addr_id, prev_addr;
    const float* e_obs_vec;
    const float* h0;                // shared state: row 0 of every layer (layer l at l * layer_stride), or nullptr
    const float* c0;
    int64_t layer_stride;
    int64_t w1, w2;
    int hid, n_out, nb1, ns2;
    float* gimg; float* w1_img; float* w2_img; float* bias; float* c0_copy;
    int64_t q_g, q_w1, q_w2;        // 16-byte pieces of the three images
    int img_blocks;
};

// Gate image: items of 4 H x 8 floats, [item][unit block][gate][64 lanes][4]; lane (column c = l & 31, half hh = l >> 5) holds
// k = 8 s + 4 hh + j, j = 0..3, of gate column g H + 32 ub + c - the B operands of four consecutive MFMAs.
//   layer 0: item 0 = the sample-embedding columns of W_ih (k < smp_dim, zero-padded to 8), items 1 .. H / 8 = W_hh;
//   layer l >= 1: items 0 .. H / 8 - 1 = W_ih_l ([4H, H]: the input is the hidden row of layer l - 1), then H / 8 items of W_hh_l.
// blocks [0, img_blocks): images; then L * H blocks of four waves: one gate column's bias each.
__global__ __launch_bounds__(256) void is_small_prep_kernel(const SmallPrepArgs a) {
    __shared__ float sx[1024 + 256];
    const int tid = threadIdx.x;
    const int H = a.H, nsh = a.nsh;
    if ((int)blockIdx.x < a.img_blocks) {
        const int64_t total = a.q_g + a.q_w1 + a.q_w2;
        for (int64_t q = (int64_t)blockIdx.x * 256 + tid; q < total; q += (int64_t)a.img_blocks * 256) {
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            float* dst;
            if (q < a.q_g) {
                const int lane = (int)(q & 63);
                const int t = (int)(q >> 6);
                const int g = t & 3, ub = (t >> 2) % a.ubk, item = (t >> 2) / a.ubk;
                const int col = g * H + ub * 32 + (lane & 31);
                const int k0 = 4 * (lane >> 5);
                if (item == 0) {
                    const float* wi = a.P + a.w_ih[0] + (int64_t)col * a.I + a.d.e_obs;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (k0 + j < a.d.smp) ? wi[k0 + j] : 0.0f;
                } else if (item <= nsh) {
                    v = *reinterpret_cast<const f32x4*>(a.P + a.w_hh[0] + (int64_t)col * H + 8 * (item - 1) + k0);
                } else {
                    const int r = item - 1 - nsh, l = 1 + r / (2 * nsh), ri = r % (2 * nsh);
                    const int64_t w = ri < nsh ? a.w_ih[l] : a.w_hh[l];
                    v = *reinterpret_cast<const f32x4*>(a.P + w + (int64_t)col * H + 8 * (ri < nsh ? ri : ri - nsh) + k0);
                }
                dst = a.gimg + q * 4;
            } else if (q < a.q_g + a.q_w1) {      // W1 [hid, H]: [slab][32-column block][64][4]
                const int64_t r = q - a.q_g;
                const int lane = (int)(r & 63);
                const int t = (int)(r >> 6);
                const int cb = t % a.nb1, s = t / a.nb1;
                const int col = cb * 32 + (lane & 31);
                if (col < a.hid) v = *reinterpret_cast<const f32x4*>(a.P + a.w1 + (int64_t)col * H + 8 * s + 4 * (lane >> 5));
                dst = a.w1_img + r * 4;
            } else {                              // W2 [n_out, hid]: [slab][64][4]
                const int64_t r = q - a.q_g - a.q_w1;
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
    // ---- bias rows: layer 0: b_ih + b_hh + W_ih x_shared (sample-embedding columns left out); layer l >= 1: b_ih + b_hh;
    //      shared previous state: + W_hh_l h0_l (every particle's recurrent product is this one row) ----
    const int bb = blockIdx.x - a.img_blocks;
    const int l = bb / H, nb = bb - l * H;      // H blocks of four gate columns per layer
    const int wave = tid >> 6, lane = tid & 63;
    const int n = nb * 4 + wave;
    float acc = 0.0f;
    if (a.h0)
        for (int k = tid; k < H; k += 256) sx[1024 + k] = a.h0[l * a.layer_stride + k];
    if (nb == 0 && a.c0)
        for (int k = tid; k < H; k += 256) a.c0_copy[l * H + k] = a.c0[l * a.layer_stride + k];
    if (l == 0) {      // (the input row in chunks of 1 024 columns: any lstm_in)
        const int c1 = a.d.e_obs, c2 = c1 + a.d.smp;
        const float* wi = a.P + a.w_ih[0] + (int64_t)n * a.I;
        for (int base = 0; base < a.I; base += 1024) {
            const int cnt = min(1024, a.I - base);
            if (base) __syncthreads();
            for (int kk = tid; kk < cnt; kk += 256) {
                const int k = base + kk;
                float x;
                if (k < c1) x = a.e_obs_vec[k];
                else if (k < c2) x = 0.0f;
                else x = gather_embedding_elem(a.d, a.P, a.at, k, a.prev_addr, 0.0f, a.addr_id);
                sx[kk] = x;
            }
            __syncthreads();
            for (int kk = lane; kk < cnt; kk += 64) acc += wi[base + kk] * sx[kk];
        }
    } else {
        __syncthreads();
    }
    if (a.h0) {
        const float* wh = a.P + a.w_hh[l] + (int64_t)n * H;
        for (int k = lane; k < H; k += 64) acc += wh[k] * sx[1024 + k];
    }
    acc = wave_sum(acc);
    if (lane == 0) a.bias[l * 4 * H + n] = acc + (a.P[a.b_ih[l] + n] + a.P[a.b_hh[l] + n]);
}

struct SmallArgs {
    const float* gimg;
    const float* bias;          // [L][4 H]
    float* h;
    float* c;
    int64_t layer_stride;       // floats between the layers of (h, c)
    const float* c0;            // shared state: [L][H] copy of the previous cell rows
    const int64_t* rows;
    const float* prev_value;
    const float* smp_w; const float* smp_b; int smp_in, smp;
    const float* w1_img; const float* b1; int hid, nb1;
    const float* w2_img; const float* b2; int n_out, ns2;
    float* y_out; int64_t ldy;
    const float* prior; int prior_stride;
    const float* value_in; float* value_out; float* logq_out;
    uint64_t seed, offset;
    int K, n, L, ubk;
    int prev_indexed;
    float* value_full; float* lw_full; int prior_kind;
    long long* dbg;             // debug: clock64 stamps [2 workgroups][2 waves][16] (pp_debug_timeline, tools/is_small_timeline.py)
};

extern __shared__ __attribute__((aligned(16))) float small_lds[];

// UBK = H / 32 unit blocks; KIND 0 / 1 / 2: mixture heads drawn in the tail, 3: head outputs only; SHARED: one previous state row
// for every particle (its recurrent products are in the bias rows).
// Registers: four accumulators (64) + the ring (48) + the previous cell values (16) fit 168 = three waves per SIMD; the compiler
// left alone takes ~180 (two waves per SIMD), and every phase of a workgroup outside the K loop (staging, cell, head layers, the
// ~20 000-cycle draw of one wave) leaves the MFMA pipe to the OTHER workgroups of the CU.
// UBK = 0: the unit-block count is a.ubk (H = 96, 160 .. 256: every index below is a run-time value; 32 particles per workgroup)
template <int UBK, int KIND, bool SHARED>
__global__ __launch_bounds__(UBK ? 2 * rows_per_block(UBK) * UBK : 512) __attribute__((amdgpu_waves_per_eu(3, 3)))
void is_step_small_kernel(const SmallArgs a) {
    const int ubk = UBK ? UBK : a.ubk;
    constexpr int SR = rows_per_block(UBK);
    const int H = 32 * ubk;
    const int HP = H + 4;                           // row pitch of the hidden tiles (16-byte aligned rows, rows 4 banks apart)
    const int NSH = H / 8;
    const int NT = 2 * SR * ubk;
    const int ITEMF = ubk * 4 * 256;                // floats of one item of the gate image
    const int AP = a.ns2 * 8 + 4;                   // row pitch of the head activations
    float* sH = small_lds;                          // [64][HP] fresh hidden rows of the layer below (first: the sample embedding)
    int* sRow = reinterpret_cast<int*>(sH + SR * HP);   // [64]
    float* sHold = reinterpret_cast<float*>(sRow + SR); // [L][64][HP] old hidden rows of every layer (not SHARED); dead after the
    float* sA1 = sHold;                             // last layer's K loop: [64][AP] head activations and
    float* sY = sA1 + SR * AP;                      // [64][33] head outputs lie over them

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ub = wave % ubk, rb = wave / ubk;
    const int c31 = lane & 31, hh = lane >> 5;
    const int m0 = (int)blockIdx.x * SR;
    const int u = ub * 32 + c31;                    // this lane's hidden unit
    const int dbg_slot = a.dbg ? (blockIdx.x == 0 ? 0 : ((int)blockIdx.x == (int)gridDim.x / 2 ? 1 : -1)) : -1;
    int dbg_k = 0;
#define SMALL_STAMP()                                                                              \
    do {                                                                                           \
        if (dbg_slot >= 0 && lane == 0 && (wave == 0 || wave == NT / 64 - 1) && dbg_k < 16)          \
            a.dbg[(dbg_slot * 2 + (wave == 0 ? 0 : 1)) * 16 + dbg_k] = clock64();                  \
        ++dbg_k;                                                                                   \
    } while (0)
    SMALL_STAMP();      // 0 start
    if (tid < SR) {
        const int gr = min(m0 + tid, a.n - 1);
        const int64_t ri = a.rows ? a.rows[gr] : (int64_t)gr;
        sRow[tid] = (int)ri;
        // slab 0 of the tile: relu(sample embedding) of the previous value, zero-padded to 8 (embedding_feedforward.py: one
        // Linear + ReLU; a Linear(1, smp_dim) of the value, or a row of the one-hot Linear(C, smp_dim))
        const float pv = a.prev_value[a.prev_indexed ? ri : (int64_t)gr];
        int cat = (int)pv;
        cat = cat < 0 ? 0 : (cat >= a.smp_in ? a.smp_in - 1 : cat);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int kk = min(k, a.smp - 1);
            const float e = a.smp_in == 1 ? a.smp_w[kk] * pv + a.smp_b[kk] : a.smp_w[kk * a.smp_in + cat] + a.smp_b[kk];
            sH[tid * HP + k] = k < a.smp ? relu_keep_nan(e) : 0.0f;
        }
    }
    SMALL_STAMP();      // 1 rows + sample embedding (this wave's share)
    if constexpr (!SHARED) {
        // the old hidden rows of every layer -> LDS, 16 bytes per thread and load, four loads in flight. Every thread looks its
        // rows up itself (the list is read again from L1 / L2): no barrier between the row list and these loads - the two
        // dependent round trips of the sample embedding (row -> previous value) and of the staging (row -> hidden row) overlap
        const int per_layer = SR * (H / 4);
        const int total = a.L * per_layer;
        for (int e0 = tid; e0 < total; e0 += 4 * NT) {
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e0 + q * NT;
                if (e < total) {
                    const int l = e / per_layer, r = e - l * per_layer;
                    const int row = r / (H / 4), p = r - row * (H / 4);
                    const int gr = min(m0 + row, a.n - 1);
                    const int64_t ri = a.rows ? a.rows[gr] : (int64_t)gr;
                    v[q] = *reinterpret_cast<const f32x4*>(a.h + l * a.layer_stride + ri * H + 4 * p);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e0 + q * NT;
                if (e < total) {
                    const int l = e / per_layer, r = e - l * per_layer;
                    const int row = r / (H / 4), p = r - row * (H / 4);
                    *reinterpret_cast<f32x4*>(sHold + (l * SR + row) * HP + 4 * p) = v[q];
                }
            }
        }
    }
    __syncthreads();
    SMALL_STAMP();      // 2 old rows staged

    const int arow = (rb * 32 + c31) * HP + 4 * hh;   // this lane's A row (+ its k half) inside a tile
    const float* gim = a.gimg + (size_t)ub * (4 * 256) + lane * 4;
    for (int l = 0; l < a.L; ++l) {
        f32x16 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float b = a.bias[(l * 4 + g) * H + u];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = b;
        }
        // the previous cell state of this lane's 16 (row, unit) pairs: issued now, used after the K loop
        float cp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            cp[r] = SHARED ? a.c0[l * H + u] : a.c[l * a.layer_stride + (int64_t)sRow[row] * H + u];
        }
        const int n_in = l == 0 ? 1 : NSH;                  // items of the input product (A from the tile)
        const int T = n_in + (SHARED ? 0 : NSH);            // + the recurrent product (A from the staged old rows)
        const float* img = gim + (size_t)(l == 0 ? 0 : (1 + NSH) + (l - 1) * 2 * NSH) * ITEMF;
        const float* aold = sHold + l * SR * HP + arow;
        f32x4 ring[RING][4];
#pragma unroll
        for (int q = 0; q < RING; ++q)
            if (q < T) {
#pragma unroll
                for (int g = 0; g < 4; ++g) ring[q][g] = *reinterpret_cast<const f32x4*>(img + (size_t)q * ITEMF + g * 256);
            }
        for (int i0 = 0; i0 < T; i0 += RING) {
#pragma unroll
            for (int q = 0; q < RING; ++q) {
                const int i = i0 + q;
                if (i < T) {      // (wave-uniform)
                    const f32x4 av = *reinterpret_cast<const f32x4*>(i < n_in ? sH + arow + 8 * i : aold + 8 * (i - n_in));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], ring[q][g][j], acc[g], 0, 0, 0);
                    if (i + RING < T) {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            ring[q][g] = *reinterpret_cast<const f32x4*>(img + (size_t)(i + RING) * ITEMF + g * 256);
                    }
                }
            }
        }
        // gates (torch.nn.LSTM order i, f, g, o) in place, then the barrier: every wave has read the tile and the old rows
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float gi = fast_sigmoid_s(acc[0][r]);
            const float gg = fast_tanh_s(acc[2][r]);
            acc[0][r] = gi * gg;
            acc[1][r] = fast_sigmoid_s(acc[1][r]);
            acc[3][r] = fast_sigmoid_s(acc[3][r]);
        }
        SMALL_STAMP();      // 3 + 4 l: K loop + gates
        __syncthreads();
        SMALL_STAMP();      // 4 + 4 l: barrier
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float cn = acc[1][r] * cp[r] + acc[0][r];
            const float hn = acc[3][r] * fast_tanh_s(cn);
            if (m0 + row < a.n) {
                const int64_t off = l * a.layer_stride + (int64_t)sRow[row] * H + u;
                a.c[off] = cn;
                a.h[off] = hn;
            }
            sH[row * HP + u] = hn;
        }
        SMALL_STAMP();      // 5 + 4 l: cell
        __syncthreads();
        SMALL_STAMP();      // 6 + 4 l: barrier
    }

    // ---- head layer 1: a1 = relu(h W1^T + b1): wave (rb, ub) takes the 32-column blocks ub, ub + UBK, ... of its row block ----
    for (int cb = ub; cb < a.nb1; cb += ubk) {
        f32x16 e;
#pragma unroll
        for (int r = 0; r < 16; ++r) e[r] = 0.0f;
        const float* wimg = a.w1_img + (size_t)cb * 256 + lane * 4;
        const size_t sstride = (size_t)a.nb1 * 256;
        for (int s0 = 0; s0 < NSH; s0 += 8) {      // eight slabs of W1 in flight together
            f32x4 bq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) bq[q] = *reinterpret_cast<const f32x4*>(wimg + (size_t)min(s0 + q, NSH - 1) * sstride);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (s0 + q < NSH) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(sH + arow + 8 * (s0 + q));
#pragma unroll
                    for (int j = 0; j < 4; ++j) e = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bq[q][j], e, 0, 0, 0);
                }
        }
        const int col = cb * 32 + c31;
        if (col < a.ns2 * 8) {
            const float bias1 = col < a.hid ? a.b1[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                sA1[row * AP + col] = col < a.hid ? relu_keep_nan(e[r] + bias1) : 0.0f;
            }
        }
    }
    SMALL_STAMP();          // head layer 1
    __syncthreads();
    SMALL_STAMP();          // barrier

    // ---- head layer 2: y = a1 W2^T + b2 (at most 32 columns): the first unit-block wave of each row block ----
    if (ub == 0) {
        f32x16 e;
#pragma unroll
        for (int r = 0; r < 16; ++r) e[r] = 0.0f;
        const float* arow1 = sA1 + (rb * 32 + c31) * AP + 4 * hh;
        for (int s0 = 0; s0 < a.ns2; s0 += 8) {      // eight slabs of W2 in flight together (loaded one by one inside the loop,
            f32x4 bw[8];                              // each MFMA group waited a full L2 round trip: 9 000 of the phase's 11 000 cycles)
#pragma unroll
            for (int q = 0; q < 8; ++q)
                bw[q] = *reinterpret_cast<const f32x4*>(a.w2_img + (size_t)min(s0 + q, a.ns2 - 1) * 256 + lane * 4);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (s0 + q < a.ns2) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(arow1 + 8 * (s0 + q));
#pragma unroll
                    for (int j = 0; j < 4; ++j) e = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bw[q][j], e, 0, 0, 0);
                }
        }
        const float bias2 = c31 < a.n_out ? a.b2[c31] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float y = c31 < a.n_out ? e[r] + bias2 : 0.0f;
            sY[row * 33 + c31] = y;
            if (a.y_out && m0 + row < a.n && c31 < a.n_out) a.y_out[(int64_t)(m0 + row) * a.ldy + c31] = y;
        }
    }
    SMALL_STAMP();          // head layer 2
    if (KIND == 3) return;
    __syncthreads();
    SMALL_STAMP();          // barrier

    // ---- draw + log q: one lane per particle (is_draw.hpp mixture_particle - the tail of the chain's own launches) ----
    // (the drawing wave rotates with the workgroup: the ~2 500 instructions of a draw are as long as a wave's whole MFMA phase, and
    // the first wave of every workgroup lands on the same SIMD)
    if (wave == (int)(blockIdx.x % (NT / 64))) {
        const int64_t i = m0 + lane;
        if (lane < SR && i < a.n) {
            const float pa = a.prior[i * 2 * a.prior_stride], pb = a.prior[i * 2 * a.prior_stride + 1];
            float v, lp;
            mixture_particle<KIND == 3 ? 0 : KIND>(sY + lane * 33, pa, pb, a.K, a.value_in != nullptr, a.value_in ? a.value_in[i] : 0.0f,
                                                   a.seed, a.offset + (uint64_t)i, v, lp);
            if (a.value_out) a.value_out[i] = v;
            if (a.logq_out) a.logq_out[i] = lp;
            if (a.value_full) {
                // whole-statement mode (pp_is_statement_rows): + log p(v) of the program's own prior, then - log q(v): two fp32
                // additions in the order of the separate log-weight kernels (pp_logweight_accumulate, pp_axpy)
                const int64_t ri = sRow[lane];
                a.value_full[ri] = v;
                float plp;
                if (a.prior_kind == 0) {
                    const float d = v - pa;
                    plp = -(d * d) / (2.0f * pb * pb) - logf(pb) - kHalfLog2Pi;
                } else {
                    plp = (v >= pa && v < pb) ? -logf(pb - pa) : -INFINITY;
                }
                float lw = a.lw_full[ri];
                lw += plp;
                lw += -1.0f * lp;
                a.lw_full[ri] = lw;
            }
        }
    }
    SMALL_STAMP();          // draw (the drawing wave: workgroup 0 -> wave 0, the middle workgroup -> wave (grid / 2) % waves)
#undef SMALL_STAMP
}

size_t small_lds_bytes(int H, int L, int ns2, bool shared) {
    const int HP = H + 4, AP = ns2 * 8 + 4, SR = rows_per_block(H / 32);
    const size_t head = (size_t)SR * AP + SR * 33, old = shared ? 0 : (size_t)L * SR * HP;
    return ((size_t)SR * HP + SR + std::max(head, old)) * sizeof(float);
}

template <int UBK, int KIND, bool SHARED>
int launch_small(const SmallArgs& a, size_t lds, hipStream_t st) {
    static bool raised = false;
    if (!raised && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&is_step_small_kernel<UBK, KIND, SHARED>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            set_error("pp_is_step: cannot raise the LDS limit of the small-network statement kernel: %s", hipGetErrorString(e));
            return (int)e;
        }
        raised = true;
    }
    constexpr int SR = rows_per_block(UBK);
    hipLaunchKernelGGL((is_step_small_kernel<UBK, KIND, SHARED>), dim3(cdiv(a.n, SR)), dim3(2 * SR * a.ubk), lds, st, a);
    return 0;
}
template <int UBK>
int launch_small_kind(const SmallArgs& a, int kind, bool shared, size_t lds, hipStream_t st) {
#define PP_SMALL_CASE(KD) (shared ? launch_small<UBK, KD, true>(a, lds, st) : launch_small<UBK, KD, false>(a, lds, st))
    switch (kind) {
        case 0: return PP_SMALL_CASE(0);
        case 1: return PP_SMALL_CASE(1);
        case 2: return PP_SMALL_CASE(2);
        default: return PP_SMALL_CASE(3);
    }
#undef PP_SMALL_CASE
}

}  // namespace

// every multiple of 32 up to 256 (H = 256 with one layer is the big kernel's: is_step_fused.hip)
bool is_small_network(const pp_net* net) {
    if (!net) return false;
    const int H = net->lstm_dim, L = std::max(1, (int)net->lstm_depth);
    return H >= 32 && H <= 256 && (H % 32) == 0 && !(H == 256 && L == 1) && L <= PP_MAX_LSTM_DEPTH;
}

bool is_step_small_supported(const pp_net* net, int addr_id) {
    if (!is_small_network(net)) return false;
    const int H = net->lstm_dim, L = std::max(1, (int)net->lstm_depth);
    if (net->smp_dim < 1 || net->smp_dim > 8 || !net->addr_table) return false;
    if (addr_id < 0 || addr_id >= net->n_addr) return false;
    const pp_addr& ad = net->addrs[addr_id];
    if (ad.n_out < 1 || ad.n_out > 32 || ad.hid < 1 || ad.hid > 512) return false;
    return small_lds_bytes(H, L, (ad.hid + 7) / 8, false) <= 150 * 1024;
}

void is_small_carve_sizes(const pp_net* net, IsFusedBuffers& f) {
    f = IsFusedBuffers{};
    const int H = net->lstm_dim, L = std::max(1, (int)net->lstm_depth), nsh = H / 8;
    int64_t hid = 1;
    for (int a = 0; a < net->n_addr; ++a) hid = std::max<int64_t>(hid, net->addrs[a].hid);
    f.n_whh = (int64_t)((1 + nsh) + (L - 1) * 2 * nsh) * 4 * H * 8;
    f.n_w1 = (int64_t)nsh * ((hid + 31) / 32) * 256;
    f.n_w2 = (int64_t)((hid + 7) / 8) * 256;
    f.n_bias = (int64_t)L * 4 * H + (int64_t)L * H;      // bias rows, then the copy of the shared cell rows
}

int is_step_small(const pp_net* net, const float* P, int addr_id, int prev_addr_id, int n, const float* e_obs_vec,
                  const float* prev_value, const float* prior, int prior_stride, float* h, float* c, int state_rows,
                  int64_t layer_rows, const int64_t* rows, const float* value_in, float* value_out, float* logq_out, uint64_t seed,
                  uint64_t offset, const IsFusedBuffers& f, float* y_out, int64_t ldy, bool net_only, bool* sampled, hipStream_t st,
                  const IsStatementOut* whole) {
    const pp_addr& ad = net->addrs[addr_id];
    const int H = net->lstm_dim, L = std::max(1, (int)net->lstm_depth), nsh = H / 8, ubk = H / 32;
    const bool shared = state_rows == 1;
    SmallPrepArgs p{};
    p.P = P; p.at = net->addr_table;
    p.H = H; p.I = net->lstm_in; p.L = L; p.ubk = ubk; p.nsh = nsh;
    for (int l = 0; l < L; ++l) {
        p.w_ih[l] = l == 0 ? net->w_ih : net->lstm_w_ih[l];
        p.w_hh[l] = l == 0 ? net->w_hh : net->lstm_w_hh[l];
        p.b_ih[l] = l == 0 ? net->b_ih : net->lstm_b_ih[l];
        p.b_hh[l] = l == 0 ? net->b_hh : net->lstm_b_hh[l];
    }
    p.d = GatherDims{net->e_obs, net->smp_dim, net->dtype_dim, net->addr_dim, net->lstm_in};
    p.addr_id = addr_id; p.prev_addr = prev_addr_id;
    p.e_obs_vec = e_obs_vec;
    p.layer_stride = layer_rows * H;
    p.h0 = shared ? h : nullptr;
    p.c0 = shared ? c : nullptr;
    p.w1 = ad.w1; p.w2 = ad.w2;
    p.hid = ad.hid; p.n_out = ad.n_out; p.nb1 = (ad.hid + 31) / 32; p.ns2 = (ad.hid + 7) / 8;
    float* c0_copy = f.bias + (int64_t)L * 4 * H;
    p.gimg = f.whh; p.w1_img = f.w1; p.w2_img = f.w2; p.bias = f.bias; p.c0_copy = c0_copy;
    p.q_g = (int64_t)((1 + nsh) + (L - 1) * 2 * nsh) * ubk * 4 * 64;
    p.q_w1 = (int64_t)nsh * p.nb1 * 64;
    p.q_w2 = (int64_t)p.ns2 * 64;
    p.img_blocks = (int)std::min<int64_t>(512, (p.q_g + p.q_w1 + p.q_w2 + 255) / 256);
    hipLaunchKernelGGL(is_small_prep_kernel, dim3(p.img_blocks + L * H), dim3(256), 0, st, p);
    PP_LAUNCH_CHECK("pp_is_step(prepare, small network)");

    SmallArgs a{};
    a.ubk = ubk;
    a.gimg = f.whh; a.bias = f.bias;
    a.h = h; a.c = c; a.layer_stride = p.layer_stride; a.c0 = c0_copy; a.rows = rows;
    const pp_addr& pad = net->addrs[prev_addr_id];
    a.prev_value = prev_value; a.smp_w = P + pad.smp_w; a.smp_b = P + pad.smp_b; a.smp_in = pad.smp_in; a.smp = net->smp_dim;
    a.w1_img = f.w1; a.b1 = P + ad.b1; a.hid = ad.hid; a.nb1 = p.nb1;
    a.w2_img = f.w2; a.b2 = P + ad.b2; a.n_out = ad.n_out; a.ns2 = p.ns2;
    a.prior = prior; a.prior_stride = prior_stride;
    a.value_in = value_in; a.value_out = value_out; a.logq_out = logq_out;
    a.seed = seed; a.offset = offset; a.K = ad.n_out / 3; a.n = n; a.L = L;
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
    const size_t lds = small_lds_bytes(H, L, p.ns2, shared);
    // kernel class 5 of the in-stream timing (work = FLOPs of the reference's algorithm, SURVEY.md 8d: input + recurrent
    // product of every layer, both head layers)
    double flops = 2.0 * (net->lstm_in + (shared ? 0 : H)) * 4.0 * H + 2.0 * ((double)H * ad.hid + (double)ad.hid * ad.n_out);
    flops += (L - 1) * 2.0 * (H + (shared ? 0 : H)) * 4.0 * H;
    prof_begin(5, st);
    int rc;
    if (ubk == 1) rc = launch_small_kind<1>(a, kind, shared, lds, st);
    else if (ubk == 2) rc = launch_small_kind<2>(a, kind, shared, lds, st);
    else if (ubk == 4) rc = launch_small_kind<4>(a, kind, shared, lds, st);
    else rc = launch_small_kind<0>(a, kind, shared, lds, st);
    prof_end(5, flops * n, st);
    if (rc) return rc;
    PP_LAUNCH_CHECK("pp_is_step(fused statement, small network)");
    *sampled = kind != 3;
    return 0;
}

}  // namespace pp
