// Fused observe-embedding kernels (InferenceNetwork._embed_observe, pyprob/nn/inference_network.py:132-139, built
// at :80-130: per observable FF(in -> hid -> out, ReLU, ReLU); concat; FF(e -> e -> e, ReLU, ReLU)) for SMALL
// embeddings (e_obs <= 64, sum of hidden widths <= 64, inputs <= 8 wide): the whole stack is a few thousand MACs per
// trace, so six GEMM launches forward and ~18 launches backward were pure launch/latency cost. Here one wavefront walks
// a trace through all layers with the weights resident in LDS: lane j owns output unit j, inputs are broadcast
// with v_readlane, weight rows are read conflict-free (row stride cols+1). The backward kernel keeps each lane's
// weight-gradient rows in registers across its traces, combines the four waves with LDS float atomics and flushes one
// atomic per parameter per workgroup. Larger embeddings take the generic GEMM path (engine.hip).
#include "common.hpp"
#include "gather.hpp"
#include "obs_embed.hpp"
#include "panel.hpp"

#include <stdlib.h>

#include <algorithm>

namespace pp {

// NOBS: compile-time bound of the observable loops (1, 2, 4, 8 >= n_obs). With the loops unrolled every index into the
// argument structs is static and the LOCAL copies below live in scalar registers; through the kernel-argument memory and
// run-time indices the compiler re-loaded the layer descriptions at every use (136 scalar loads, each with a wait, on
// the critical path of a kernel that runs one trace per wave).
template <int NOBS>
__global__ __launch_bounds__(256) void obs_embed_fwd_kernel(const ObsFusedArgs ain, const float* __restrict__ P,
                                                            const float* __restrict__ obs, int n_traces,
                                                            int traces_per_wave, float* __restrict__ cat,
                                                            float* __restrict__ f1, float* __restrict__ E,
                                                            const RowBuild rbin, const AddrBias ab,
                                                            const PanelTranspose tr) {
    __shared__ float lds[10240];
    const ObsFusedArgs a = ain;
    const RowBuild rb = rbin;
    if (tr.n_blocks && (int)blockIdx.x >= tr.first_block) {   // extra workgroups: k-major weight copies for the panel kernel
        panel_transpose_block(tr, (int)blockIdx.x - tr.first_block, lds);
        return;
    }
    if (ab.AB && (int)blockIdx.x >= ab.first_block) {   // extra workgroups: per-address bias vectors of the LSTM input
        addr_bias_block(ab, (int)blockIdx.x - ab.first_block, lds);
        return;
    }
    warm_kernargs((int)(sizeof(ObsFusedArgs) + sizeof(RowBuild) + sizeof(AddrBias) + sizeof(PanelTranspose)) + 64);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // training step: this is the first kernel - clear the loss slots that later kernels add into
    if (rb.zero_small && blockIdx.x == 0)
        for (int q = tid; q < rb.n_small; q += 256) rb.zero_small[q] = 0.0f;
    obs_stage_all<NOBS>(a, P, lds, tid);
    __syncthreads();
    const int b0 = (blockIdx.x * 4 + wave) * traces_per_wave;
    for (int t = 0; t < traces_per_wave; ++t) {
        const int b = b0 + t;
        if (b >= n_traces) break;   // wave-uniform
        float h = 0.0f, c = 0.0f;
        int ci = 0, co = 0;
#pragma unroll
        for (int o = 0; o < NOBS; ++o) {
            if (o >= a.n_obs) break;
            const int jh = lane - a.hoff[o];
            const bool acth = jh >= 0 && jh < a.hid[o];
            if (acth) {   // layer 0 reads the raw observation directly (no cross-lane traffic)
                const float* w = lds + a.l0[o].lds_w + jh * (a.in[o] + 1);
                float s = lds[a.l0[o].lds_b + jh];
                for (int i = 0; i < a.in[o]; ++i) s += w[i] * obs[(int64_t)b * a.width + ci + i];
                h = relu_keep_nan(s);
                a.obs_h[o][(int64_t)b * a.ohid_ld[o] + jh] = h;
            }
            ci += a.in[o];
        }
#pragma unroll
        for (int o = 0; o < NOBS; ++o) {
            if (o >= a.n_obs) break;
            const int jc = lane - co;
            const bool actc = jc >= 0 && jc < a.out[o];
            const float v = obs_dense(lds, a.l1[o], jc, actc, h, a.hoff[o]);
            if (actc) c = v;
            co += a.out[o];
        }
        const bool acte = lane < a.e_obs;
        if (acte) cat[(int64_t)b * a.e_ld + lane] = c;
        const float y1 = obs_dense(lds, a.f0, lane, acte, c, 0);
        if (acte) f1[(int64_t)b * a.e_ld + lane] = y1;
        const float y2 = obs_dense(lds, a.f1, lane, acte, y1, 0);
        if (acte) E[(int64_t)b * a.e_ld + lane] = y2;
        if (rb.X) {
            // LSTM input rows of this trace, one per time step it is alive in (step-major rows: row_off[t] + b): the
            // embedding just computed sits in lane c < e_obs, the address / previous-sample columns come from the tables
            // (a separate gather launch did this before: one launch and one read of E less). Also clears dX.
            {   // t = 0 needs no index loads: row_off[0] = 0, every trace is alive, no previous statement (pr = -1) - the
                // columns [e_obs, c4) are zero; only full-width rows look the current address up
                const int c4 = rb.d.e_obs + rb.d.smp + rb.d.dtype + rb.d.addr;
                const int ad = rb.xcols > c4 ? rb.addr[b] : 0;
                float* xr = rb.X + (int64_t)b * rb.ldx;
                if (acte) xr[lane] = y2;
                for (int c = rb.d.e_obs + lane; c < rb.xcols; c += 64) xr[c] = gather_embedding_elem(rb.d, rb.params, rb.at, c, -1, 0.0f, ad);
                if (rb.zero_like) {
                    float* zr = rb.zero_like + (int64_t)b * rb.ldx;
                    for (int c = lane; c < rb.xcols; c += 64) zr[c] = 0.0f;
                }
            }
            for (int t = 1; t < rb.t_max; ++t) {
                const int r0 = rb.row_off[t];
                if (b >= rb.row_off[t + 1] - r0) break;   // wave-uniform
                const int r = r0 + b;
                const int pr = rb.prev_row[r];
                const int ap = pr < 0 ? -1 : rb.addr[pr];
                const float v = pr < 0 ? 0.0f : rb.value[pr];
                const int ad = rb.addr[r];
                float* xr = rb.X + (int64_t)r * rb.ldx;
                if (acte) xr[lane] = y2;
                for (int c = rb.d.e_obs + lane; c < rb.xcols; c += 64) xr[c] = gather_embedding_elem(rb.d, rb.params, rb.at, c, ap, v, ad);
                if (rb.zero_like) {
                    float* zr = rb.zero_like + (int64_t)r * rb.ldx;
                    for (int c = lane; c < rb.xcols; c += 64) zr[c] = 0.0f;
                }
            }
        }
    }
}

// Backward of the observe embedding, DATA gradients only: per trace (one wave)
//   dz1 = (Wf1^T dz2) * [f1 > 0],  dzc = (Wf0^T dz1) * [cat > 0],  dh_o = (W1_o^T dzc_o) * [h_o > 0]
// written to dF1 / dCat / dH_o. The weight gradients (dz^T x over the batch) are MFMA products with K = batch rows and
// join the grouped weight-gradient launch of the backward pass; the bias gradients are column sums of the same
// buffers. (A first version accumulated the weight gradients in registers and flushed ~10k atomics per workgroup:
// 57 us; this split is 3x cheaper.)
template <int NOBS>
__global__ __launch_bounds__(256) void obs_embed_dgrad_kernel(const ObsFusedArgs ain, const float* __restrict__ P,
                                                              int n_traces, int traces_per_wave,
                                                              const float* __restrict__ cat, const float* __restrict__ f1,
                                                              const float* __restrict__ dX, int64_t ldx,
                                                              const int32_t* __restrict__ row_off, int t_max,
                                                              const float* __restrict__ E, float* __restrict__ dE,
                                                              float* __restrict__ dF1, float* __restrict__ dCat,
                                                              float* const dHo0, int64_t dh_stride, int n_split,
                                                              int64_t split_stride) {
    __shared__ float lds[10240];
    warm_kernargs((int)sizeof(ObsFusedArgs) + 96);
    const ObsFusedArgs a = ain;   // local copy, static indices: see obs_embed_fwd_kernel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // which observable this lane belongs to as a hidden unit (and that observable's hidden-activation buffer)
    int oh = -1, jh = 0;
    const float* my_h = nullptr;
    int64_t my_hld = 0;
#pragma unroll
    for (int o = 0; o < NOBS; ++o)
        if (o < a.n_obs && lane >= a.hoff[o] && lane < a.hoff[o] + a.hid[o]) {
            oh = o; jh = lane - a.hoff[o];
            my_h = a.obs_h[o]; my_hld = a.ohid_ld[o];
        }
    const bool acte = lane < a.e_obs;
    const int b0 = (blockIdx.x * 4 + wave) * traces_per_wave;
    // Gradient into the embedding output of trace b: every time step of the trace consumed E[b], so
    // dE[b, c] = (E[b, c] > 0) * sum_t dX[row_off[t] + b, c] (deterministic, no atomics); it is also written out as
    // the operand of the final layer's weight / bias gradients.
    auto load_dz2 = [&](int b) {
        float acc = 0.0f;
        if (acte) {
            acc = dX[(int64_t)b * ldx + lane];   // t = 0: row b (row_off[0] = 0, every trace is alive) - no index load
            // (single-statement batches: dX arrives as the partial tiles of its K splits, stored instead of added)
            // (eight loads in flight per trip: a load-add loop would pay one memory round trip per split)
            for (int z0 = 1; z0 < n_split; z0 += 8) {
                float pv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    pv[u] = (z0 + u < n_split) ? dX[(int64_t)(z0 + u) * split_stride + (int64_t)b * ldx + lane] : 0.0f;
                acc += ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
            }
            for (int t = 1; t < t_max; ++t) {
                const int r0 = row_off[t];
                if (b >= row_off[t + 1] - r0) break;
                acc += dX[(int64_t)(r0 + b) * ldx + lane];
            }
            acc = E[(int64_t)b * a.e_ld + lane] > 0.0f ? acc : 0.0f;
        }
        return acc;
    };
    // first trace's inputs: issued BEFORE the weight staging so both memory round trips overlap
    float nx_dz2 = 0.f, nx_f1 = 0.f, nx_cat = 0.f, nx_h = 0.f;
    if (b0 < n_traces) {
        nx_dz2 = load_dz2(b0);
        nx_f1 = acte ? f1[(int64_t)b0 * a.e_ld + lane] : 0.0f;
        nx_cat = acte ? cat[(int64_t)b0 * a.e_ld + lane] : 0.0f;
        nx_h = oh >= 0 ? my_h[(int64_t)b0 * my_hld + jh] : 0.0f;
    }
    obs_stage_all<NOBS>(a, P, lds, tid);
    __syncthreads();
    for (int t = 0; t < traces_per_wave; ++t) {
        const int b = b0 + t;
        if (b >= n_traces) break;   // wave-uniform
        const float dz2 = nx_dz2, f1v = nx_f1, catv = nx_cat, hv = nx_h;
        if (t + 1 < traces_per_wave && b + 1 < n_traces) {
            const int bn = b + 1;
            nx_dz2 = load_dz2(bn);
            nx_f1 = acte ? f1[(int64_t)bn * a.e_ld + lane] : 0.0f;
            nx_cat = acte ? cat[(int64_t)bn * a.e_ld + lane] : 0.0f;
            nx_h = oh >= 0 ? my_h[(int64_t)bn * my_hld + jh] : 0.0f;
        }
        if (acte) dE[(int64_t)b * a.e_ld + lane] = dz2;
        float dz1 = obs_dense_t(lds, a.f1, lane, acte, dz2, 0);
        dz1 = f1v > 0.0f ? dz1 : 0.0f;
        if (acte) dF1[(int64_t)b * a.e_ld + lane] = dz1;
        float dzc = obs_dense_t(lds, a.f0, lane, acte, dz1, 0);
        dzc = catv > 0.0f ? dzc : 0.0f;
        if (acte) dCat[(int64_t)b * a.e_ld + lane] = dzc;
        float dh = 0.0f;
        int co = 0;
#pragma unroll
        for (int o = 0; o < NOBS; ++o) {
            if (o >= a.n_obs) break;
            const bool acth = (oh == o);
            const float d = obs_dense_t(lds, a.l1[o], jh, acth, dzc, co);   // dh_k = sum_j dzc_j W1[j][k]
            if (acth) dh = d;
            co += a.out[o];
        }
        if (oh >= 0) dHo0[(int64_t)oh * dh_stride + (int64_t)b * my_hld + jh] = hv > 0.0f ? dh : 0.0f;
    }
}

// ---- host side -----------------------------------------------------------------------------------------
static inline int64_t round4(int64_t x) { return (x + 3) & ~int64_t(3); }

bool obs_fused_supported(const pp_net* net) {
    if (net->e_obs > OBS_EMAX) return false;
    for (int o = 0; o < net->n_obs; ++o)
        if (net->obs_depth[o] != 0 && net->obs_depth[o] != 2) return false;   // other depths: generic GEMM path (engine.hip)
    int hsum = 0;
    for (int o = 0; o < net->n_obs; ++o) {
        if (net->obs_in[o] > OBS_INMAX || net->obs_hid[o] > OBS_HIDMAX || net->obs_hid[o] < 1) return false;
        hsum += net->obs_hid[o];
    }
    return hsum <= 64;
}

bool obs_fused_args(const pp_net* net, float* const* obs_h, ObsFusedArgs& a) {
    for (int o = 0; o < PP_MAX_OBS; ++o) a.obs_h[o] = o < net->n_obs ? obs_h[o] : nullptr;
    a.n_obs = net->n_obs;
    a.e_obs = net->e_obs;
    a.e_ld = round4(net->e_obs);
    int lds = 0, hoff = 0, width = 0;
    auto layer = [&](ObsLayer& L, int64_t w, int64_t b, int rows, int cols) {
        L.w_off = w; L.b_off = b; L.rows = rows; L.cols = cols;
        L.lds_w = lds; lds += rows * (cols + 1);
        L.lds_b = lds; lds += rows;
    };
    for (int o = 0; o < net->n_obs; ++o) {
        a.in[o] = net->obs_in[o]; a.hid[o] = net->obs_hid[o]; a.out[o] = net->obs_out[o];
        a.hoff[o] = hoff; hoff += a.hid[o];
        a.ohid_ld[o] = round4(a.hid[o]);
        width += a.in[o];
        layer(a.l0[o], net->obs_w0[o], net->obs_b0[o], a.hid[o], a.in[o]);
        layer(a.l1[o], net->obs_w1[o], net->obs_b1[o], a.out[o], a.hid[o]);
    }
    a.width = width;
    layer(a.f0, net->fin_w0, net->fin_b0, net->e_obs, net->e_obs);
    layer(a.f1, net->fin_w1, net->fin_b1, net->e_obs, net->e_obs);
    a.lds_total = lds;
    return lds + 1 <= 10240;   // + the dummy word of the branch-free staging
}

static int pick_traces_per_wave(int n, int target_blocks) {
    return std::max((n + 4 * target_blocks - 1) / (4 * target_blocks), 1);
}

// obs_h: host array of n_obs device pointers; E/cat/f1 leading dim = round4(e_obs)
int obs_embed_fwd_fused(const pp_net* net, const float* P, const float* obs, int n_traces, float* const* obs_h,
                        float* cat, float* f1, float* E, hipStream_t st, const RowBuild* rows, const AddrBias* bias,
                        const PanelTranspose* transpose) {
    ObsFusedArgs a;
    if (!obs_fused_supported(net) || !obs_fused_args(net, obs_h, a)) return PP_EINVAL;
    const int tpw = pick_traces_per_wave(n_traces, 256);   // forward: staging is cheap, spread the traces
    RowBuild rb{};
    if (rows) rb = *rows;
    if (rb.X && rb.xcols <= 0) rb.xcols = rb.d.I;
    AddrBias ab{};
    int blocks = cdiv(n_traces, 4 * tpw);
    if (bias && bias->AB) {   // the bias job rides in this launch as extra workgroups
        ab = *bias;
        ab.first_block = blocks;
        blocks += addr_bias_blocks(ab);
    }
    PanelTranspose tr{};
    if (transpose && transpose->n_blocks > 0) {   // behind the bias job
        tr = *transpose;
        tr.first_block = blocks;
        blocks += tr.n_blocks;
    }
#define PP_OBS_FWD(N) hipLaunchKernelGGL(obs_embed_fwd_kernel<N>, dim3(blocks), dim3(256), 0, st, a, P, obs, n_traces, tpw, cat, f1, E, rb, ab, tr)
    if (a.n_obs <= 1) PP_OBS_FWD(1);
    else if (a.n_obs <= 2) PP_OBS_FWD(2);
    else if (a.n_obs <= 4) PP_OBS_FWD(4);
    else PP_OBS_FWD(8);
#undef PP_OBS_FWD
    PP_LAUNCH_CHECK("obs_embed_fwd_fused");
    return 0;
}

// dHo: n_obs buffers of [B, round4(hid_o)] laid out `dh_stride` floats apart, starting at dHo0
int obs_embed_dgrad_fused(const pp_net* net, const float* P, int n_traces, float* const* obs_h, const float* cat,
                          const float* f1, const float* dX, int64_t ldx, const int32_t* row_off_dev, int t_max,
                          const float* E, float* dE, float* dF1, float* dCat, float* dHo0, int64_t dh_stride,
                          hipStream_t st, int n_split, int64_t split_stride) {
    ObsFusedArgs a;
    if (!obs_fused_supported(net) || !obs_fused_args(net, obs_h, a)) return PP_EINVAL;
    const int tpw = pick_traces_per_wave(n_traces, 256);
    const int nsp = n_split > 1 && t_max == 1 ? n_split : 1;
#define PP_OBS_BWD(N) hipLaunchKernelGGL(obs_embed_dgrad_kernel<N>, dim3(cdiv(n_traces, 4 * tpw)), dim3(256), 0, st, a, P, n_traces, tpw, cat, f1, dX, ldx, row_off_dev, t_max, E, dE, dF1, dCat, dHo0, dh_stride, nsp, split_stride)
    if (a.n_obs <= 1) PP_OBS_BWD(1);
    else if (a.n_obs <= 2) PP_OBS_BWD(2);
    else if (a.n_obs <= 4) PP_OBS_BWD(4);
    else PP_OBS_BWD(8);
#undef PP_OBS_BWD
    PP_LAUNCH_CHECK("obs_embed_dgrad_fused");
    return 0;
}

}  // namespace pp
