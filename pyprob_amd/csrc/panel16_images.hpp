// Fragment images of the 16-row panel kernel's weight matrices (panel16.hip) and the workgroup job that writes them in the
// training step's first launch (obs_embed_fwd_kernel, through PanelTranspose of panel.hpp).
#pragma once
#include "common.hpp"

namespace pp {

// ---- fragment images --------------------------------------------------------------------------------------------------
// Every product of the panel kernel runs on v_mfma_f32_16x16x4_f32 (A: lane l holds A[row l & 15][k = l >> 4], B: lane l
// holds B[k = l >> 4][column l & 15], MI355X guide 3). A wave consumes its B operand as FRAGMENTS: the four K steps of one
// 16-k unit of one 16-column tile, lane l = (c = l & 15, g = l >> 4) holding
//     frag[s] = B[k = 16 unit + 4 g + s][n = 16 tile + c],      s = 0 .. 3
// (K step s multiplies the k set {4 g + s}: a permutation of the unit's 16 k that the A operand - one ds_read_b128 of
// A[row][16 unit + 4 g .. + 3] - follows). An IMAGE stores a matrix fragment by fragment, 1 KB each (lane-major float4s), so
// a wave's operand stream is coalesced 1 KB loads straight into registers, one per four MFMAs. Adam has just changed the
// weights, so the images are rewritten every step by extra workgroups of the step's FIRST launch (obs_embed_fwd_kernel),
// like the k-major copies of the 8-row kernel (panel.hpp) before them.
//   img 0  input product   B[k][n] = W_ih[gate(y) H + 16 ut + c][k]       index ((unit (H/16) + ut) 3 + y)      k < e
//   img 1  head layer 1    B[k][n] = W1[n][k]  (n < hid)                   index (unit NT + tile)                unit over H / 16
//   img 2  head layer 2    B[k][n] = W2[n][k]  (k < hid, n < n_out)        index (unit 2 + tile)                 unit over NT
//   img 3  dz1 = dy W2     B[k][n] = W2[k][n]                              index (unit NT + tile)                unit over 2
//   img 4  dh = dz1 W1     B[k][n] = W1[k][n]  (k < hid)                   index (unit (H/16) + ut)              unit over NT
//   img 5  dX = dG W_ih    B[k][n] = W_ih[gate(y) H + 16 ut + 4 g + s][n]  index ((ut 3 + y) (e/16) + tile)
// gate(y) = rows of gates i, g, o (y = 0, 1, 2 -> 0, 2, 3: the forget gate multiplies c0 = 0 and is never formed);
// NT = ceil(hid / 16).
struct Panel16Images {
    float* img[6];
    int64_t frags[6];       // fragments (1 KB each) per image
};
static inline void panel16_image_sizes(int H, int hid, int e, int64_t (&frags)[6]) {
    const int NT = (hid + 15) / 16, UT = H / 16, KE = e / 16;
    frags[0] = (int64_t)KE * UT * 3;
    frags[1] = (int64_t)UT * NT;
    frags[2] = (int64_t)NT * 2;
    frags[3] = (int64_t)2 * NT;
    frags[4] = (int64_t)NT * UT;
    frags[5] = (int64_t)UT * 3 * KE;
}

struct Panel16Prep {      // rides in PanelTranspose (mode 16): the source matrices of the images
    const float* W2; int n_out;
    Panel16Images im;
    int blocks_before[7];   // first workgroup (of the job) of image i; [6] = total
};

// one thread per (fragment, lane): a float4 of four K steps
__device__ __forceinline__ void panel16_image_block(const float* __restrict__ Wih, int64_t ldw, const float* __restrict__ W1,
                                                    int H, int hid, int e, const Panel16Prep& p, int b) {
    int im = 0;
#pragma unroll
    for (int i = 1; i < 6; ++i)
        if (b >= p.blocks_before[i]) im = i;
    const int q = (b - p.blocks_before[im]) * 256 + (int)threadIdx.x;      // (fragment, lane)
    const int64_t f = q >> 6;
    if (f >= p.im.frags[im]) return;
    const int lane = q & 63, c = lane & 15, g = lane >> 4;
    const int NT = (hid + 15) >> 4, UT = H >> 4, KE = e >> 4;
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (im == 0) {
        const int y = (int)(f % 3), ut = (int)((f / 3) % UT), unit = (int)(f / (3 * UT));
        const int row = (y == 0 ? 0 : y + 1) * H + 16 * ut + c;
        v = *reinterpret_cast<const f32x4*>(Wih + (int64_t)row * ldw + 16 * unit + 4 * g);          // (ldw % 4 == 0)
    } else if (im == 1) {
        const int tile = (int)(f % NT), unit = (int)(f / NT);
        const int n = 16 * tile + c;
        if (n < hid) v = *reinterpret_cast<const f32x4*>(W1 + (int64_t)n * H + 16 * unit + 4 * g);
    } else if (im == 2) {
        const int tile = (int)(f & 1), unit = (int)(f >> 1);
        const int n = 16 * tile + c;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 16 * unit + 4 * g + s;
            if (n < p.n_out && k < hid) v[s] = p.W2[(int64_t)n * hid + k];
        }
    } else if (im == 3) {
        const int tile = (int)(f % NT), unit = (int)(f / NT);
        const int n = 16 * tile + c;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 16 * unit + 4 * g + s;
            if (k < p.n_out && n < hid) v[s] = p.W2[(int64_t)k * hid + n];
        }
    } else if (im == 4) {
        const int ut = (int)(f % UT), unit = (int)(f / UT);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 16 * unit + 4 * g + s;
            if (k < hid) v[s] = W1[(int64_t)k * H + 16 * ut + c];
        }
    } else {
        const int tile = (int)(f % KE), y = (int)((f / KE) % 3), ut = (int)(f / (3 * KE));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = (y == 0 ? 0 : y + 1) * H + 16 * ut + 4 * g + s;
            v[s] = Wih[(int64_t)row * ldw + 16 * tile + c];
        }
    }
    reinterpret_cast<f32x4*>(p.im.img[im])[q] = v;
}

}  // namespace pp
