// Row-panel kernel for single-statement batches (panel.hip): arguments and host entry points used by engine.hip.
#pragma once
#include "common.hpp"
#include "obs_embed.hpp"
#include "panel16_images.hpp"

namespace pp {

struct PanelArgs {
    int B, H, hid, K, n_out, e;          // rows, LSTM hidden, head hidden, mixture components, 3 K, observe-embedding width
    int ldx, lda1, lddy, ldw;            // leading dimensions of X / dX, A1 / dZ1, DY, W_ih (= lstm_in)
    const float* X;                      // [B][ldx] LSTM input rows (columns [0, e) = observe embedding)
    const float* Wih; const float* AB;   // W_ih [4H][ldw]; per-address bias vector [4H] (b_ih + b_hh + table columns, gather.hpp)
    const float* WihT; const float* W1T; // k-major copies: WihT [e][4H] = W_ih[:, :e]^T, W1T [H][64 ceil(hid / 64)] = W1^T (PanelTranspose)
    const float* W1; const float* b1;    // [hid][H], [hid]
    const float* W2; const float* b2;    // [n_out][hid], [n_out]
    const float* value; const float* prior;
    float* Hs; float* G; float* A1; float* DY; float* dZ1; float* dX;
    float* gsum;                         // [4H] column sums of dG for this address (atomics)
    float* lp_out; float* loss_acc; int32_t* flag;
    float grad_scale;
    // pair hand-off (two workgroups per panel): 8-byte {value, epoch} granules of the partial sums of head layer 1
    // [panels][2][8][lda1] and of dX [panels][2][8][64]; *epoch is incremented once per step by the first launch
    unsigned long long* xz; unsigned long long* xd; const int* epoch;
    long long* dbg;                      // debug: clock64() stamps [2 workgroups][2 waves][16] (pp_debug_timeline) or nullptr
};

// Backward of the observe embedding as the TAIL of the panel kernel (obs_embed.hip's dgrad kernel, one launch less): a
// workgroup finishes dX for 4 of its panel's rows; four of its waves then walk one row each through the final stack and the
// per-observable layers (weights in an LDS image over the kernel's dead buffers) and write dE, dF1, dCat, dH_o - the
// operands of the embedding's weight gradients in the grouped launch that follows.
struct PanelObs {
    ObsFusedArgs a;
    const float* P;                      // flat parameters
    const float* cat; const float* f1;   // saved activations [B][e_ld]
    float* dE; float* dF1; float* dCat; float* dHo0;
    int64_t dh_stride;
};
constexpr int PANEL_OBS_LDS = 10240 + 8 + 4 * 64;      // the LDS image (+ its dummy word) and four dX rows, floats

// ---- k-major copies of the two forward weight matrices --------------------------------------------------------------
// The panel kernel reads every weight matrix with the summation index as the row (one coalesced dword per lane and k). The
// backward products have that layout in the parameter tensors; the forward ones read transposed copies, rewritten every step
// (Adam just changed the weights) by extra workgroups of the step's FIRST launch (obs_embed_fwd_kernel):
//   block b < tiles_ih : rows [n0, n0 + 64) of W_ih (gates i, g, o only), columns [0, e)  ->  WihT[k][n0 ..]
//   the rest           : tile (ti, tj) of W1 [hid][H]  ->  W1T[64 tj ..][64 ti ..]; rows beyond hid are written as zeros
// (the epoch of the pair hand-off is advanced by the bias job of the same launch, gather.hpp AddrBias::step_epoch)
struct PanelTranspose {
    const float* Wih; int64_t ldw; float* WihT;   // WihT [e][4H + 64]
    const float* W1; float* W1T; int64_t ld1T;    // W1 [hid][H], W1T [H][ld1T], ld1T = 64 ceil(hid / 64)
    int H, hid, e;
    int tiles_ih, first_block, n_blocks;           // 3 H / 64 tiles of W_ih; first workgroup of the job in its launch
    int mode16;                                    // 1: the job writes the 16-row kernel's fragment images (p16) instead
    Panel16Prep p16;
};
static inline int panel_transpose_blocks(int H, int hid) { return 3 * H / 64 + ((hid + 63) / 64) * (H / 64); }

__device__ __forceinline__ void panel_transpose_block(const PanelTranspose& tr, int b, float* lds /* >= 64 * 65 floats */) {
    if (tr.mode16) {
        panel16_image_block(tr.Wih, tr.ldw, tr.W1, tr.H, tr.hid, tr.e, tr.p16, b);
        return;
    }
    const int tid = threadIdx.x;       // 256 threads
    const int tx = tid & 63, ty = tid >> 6;
    const bool ih = b < tr.tiles_ih;
    const float* src; int64_t ld_src; int r0, c0, rmax, cmax;
    float* dst; int64_t ld_dst; int dr0, dc0, out_rows;
    if (ih) {
        const int n0 = (b * 64 < tr.H) ? b * 64 : b * 64 + tr.H;       // skip the forget gate's rows [H, 2H)
        src = tr.Wih; ld_src = tr.ldw; r0 = n0; c0 = 0; rmax = 4 * tr.H; cmax = tr.e;
        dst = tr.WihT; ld_dst = 4 * (int64_t)tr.H + 64; dr0 = 0; dc0 = n0; out_rows = tr.e < 64 ? tr.e : 64;
    } else {
        const int q = b - tr.tiles_ih, tjn = tr.H >> 6;
        const int ti = q / tjn, tj = q - ti * tjn;
        src = tr.W1; ld_src = tr.H; r0 = 64 * ti; c0 = 64 * tj; rmax = tr.hid; cmax = tr.H;
        dst = tr.W1T; ld_dst = tr.ld1T; dr0 = 64 * tj; dc0 = 64 * ti; out_rows = 64;
    }
    // 64 x 64 tile: 16 row loads per thread, all in flight, then LDS, then 16 coalesced stores of the transposed rows
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int r = r0 + ty + 4 * u, c = c0 + tx;
        v[u] = (r < rmax && c < cmax) ? src[(int64_t)r * ld_src + c] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) lds[(ty + 4 * u) * 65 + tx] = v[u];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int rr = ty + 4 * u;        // row of the transposed tile = source column
        if (rr < out_rows) dst[(int64_t)(dr0 + rr) * ld_dst + dc0 + tx] = lds[tx * 65 + rr];
    }
}

// single-statement batch, one address, one LSTM layer, mixture head: does the panel kernel take this shape?
bool panel_t1_supported(int kind, int H, int hid, int n_out, int e);
size_t panel_lds_bytes(int H, int hid, int n_out, int e);
// one launch: input product + cell, head layer 1, tail + loss + dy, dz1, dh + cell backward (dG, group sums), dX
// obs != nullptr: the observe-embedding backward of the rows rides in the tail (panel_obs_tail_ok must hold)
int panel_t1(int kind, const PanelArgs& a, hipStream_t st, const PanelObs* obs = nullptr);
bool panel_obs_tail_ok(const pp_net* net, int H, int hid, int n_out, int e);
int panel_t1_split(int B, int H);                   // 2 when the (two-workgroups-per-panel) launch takes this many rows, else 0

}  // namespace pp
