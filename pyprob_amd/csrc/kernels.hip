// HBM-bound kernels of the inference-compilation path: ragged LSTM-input gather (address dispatch), LSTM cell,
// proposal transforms + mixture log-prob with a wavefront-reduced loss accumulator, column reductions for bias /
// embedding-table gradients, flat Adam. All fp32, gfx950 wave64.
#include "common.hpp"
#include "gather.hpp"
#include "aux_jobs.hpp"

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

namespace pp {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

// ------------------------------------------------------------------------------------------------------
// out[c * out_stride] += sum_i X[ix(i)*ldx + c] * (wgt ? wgt[i * ldw] : 1)
// (the weighted form is the gradient of a layer with a handful of inputs: dW[:, k] = sum_i dz[i, :] x[i, k])
// 256 threads = 64 columns x 4 row lanes; each workgroup reduces ROWS_PER_BLOCK rows, one atomic per column.
// ------------------------------------------------------------------------------------------------------
constexpr int COLSUM_ROWS = 64;
constexpr int COLSUM_MAX_JOBS = 48;   // (48 x 72 bytes: the by-value kernel arguments stay under 4 KB; 16 made 5 launches of a 12-address step)

int loss_finalize(const float* acc, const int32_t* flag, int n_traces, float* loss_out, int32_t* status_out, hipStream_t st);

// Deterministic variant (PP_DETERMINISTIC): workgroup (column block, group) owns its 64 destination columns - ONE
// writer - and walks the jobs of its group (all jobs with the same destination) in order; each thread sums its rows
// sequentially, the four row lanes are combined in a fixed order.
struct ColsumGroups {
    ColsumJob j[COLSUM_MAX_JOBS];
    int first[COLSUM_MAX_JOBS + 1];   // jobs [first[g], first[g+1]) form group g
};

__global__ __launch_bounds__(256) void colsum_det_kernel(const ColsumGroups g) {
    __shared__ float part[4][64];
    const int tid = threadIdx.x, cl = tid & 63, rl = tid >> 6;
    const int col = blockIdx.x * 64 + cl;
    const int j0 = g.first[blockIdx.y], j1 = g.first[blockIdx.y + 1];
    float total = 0.0f;
    for (int q = j0; q < j1; ++q) {
        const ColsumJob& jb = g.j[q];
        float acc = 0.0f;
        if (col < jb.n_cols) {
            for (int i = rl; i < jb.n_rows; i += 4) {
                const int64_t r = jb.idx ? (int64_t)jb.idx[i] : (int64_t)i;
                float v = jb.X[r * jb.ldx + col];
                if (jb.wgt) v *= jb.wgt[(int64_t)i * jb.ldw];
                acc += v;
            }
        }
        part[rl][cl] = acc;
        __syncthreads();
        if (rl == 0) total += (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
        __syncthreads();
    }
    const ColsumJob& j = g.j[j0];
    if (rl == 0 && col < j.n_cols) {
        float* dst = j.out + (int64_t)col * (j.out_stride ? j.out_stride : 1);
        *dst += total;
        if (j.out2) j.out2[col] += total;
    }
}

static int colsum_multi_det(const ColsumJob* jobs, int count, hipStream_t st) {
    std::vector<char> done(count, 0);
    int left = 0;
    for (int i = 0; i < count; ++i) {
        if (jobs[i].n_rows <= 0 || jobs[i].n_cols <= 0) done[i] = 1; else ++left;
    }
    while (left > 0) {
        ColsumGroups pack;
        int n = 0, groups = 0, max_cols = 0;
        pack.first[0] = 0;
        for (int i = 0; i < count && n < COLSUM_MAX_JOBS; ++i) {
            if (done[i]) continue;
            // a destination already packed in this launch but not as the group being built: wait for the next launch
            bool clash = false;
            for (int q = 0; q < n; ++q) clash = clash || pack.j[q].out == jobs[i].out;
            if (clash) continue;
            // group: this job and every later job with the same destination (same width), in job order
            const int g0 = n;
            for (int k = i; k < count && n < COLSUM_MAX_JOBS; ++k) {
                if (done[k] || jobs[k].out != jobs[i].out) continue;
                PP_CHECK_ARG(jobs[k].X && jobs[k].n_cols == jobs[i].n_cols && jobs[k].out2 == jobs[i].out2 &&
                                 jobs[k].out_stride == jobs[i].out_stride, "pp_colsum_f32: inconsistent jobs for one destination");
                pack.j[n++] = jobs[k];
                done[k] = 1;
                --left;
            }
            if (n > g0) {
                pack.first[++groups] = n;
                max_cols = std::max(max_cols, jobs[i].n_cols);
            }
        }
        if (groups == 0) break;
        hipLaunchKernelGGL(colsum_det_kernel, dim3(cdiv(max_cols, 64), groups), dim3(256), 0, st, pack);
        PP_LAUNCH_CHECK("pp_colsum_f32 (deterministic)");
    }
    return 0;
}

int colsum_multi(const ColsumJob* jobs, int count, hipStream_t st, const float* fin_acc, const int32_t* fin_flag,
                 int fin_traces, float* fin_loss, int32_t* fin_status) {
    if (deterministic_mode()) {
        PP_TRY(colsum_multi_det(jobs, count, st));
        if (fin_acc) return loss_finalize(fin_acc, fin_flag, fin_traces, fin_loss, fin_status, st);
        return 0;
    }
    // One workgroup per (job, 64 columns, 64 rows) that exists - the job list of aux_jobs.hpp as its own launch. (A launch
    // grid shaped by the LARGEST job, as before, started 24 000 workgroups for the 48 jobs of a ragged 12-address step,
    // most of which only found out that their job was smaller: 23-29 us per launch.)
    LossFinalize fin{fin_acc, fin_flag, fin_traces > 0 ? 1.0f / (float)fin_traces : 0.0f, fin_loss, fin_status};
    int i = 0;
    bool launched = false;
    while (i < count) {
        AuxJobs pack{};
        for (; i < count && pack.n_colsum < AUX_MAX_COLSUM; ++i) {
            if (jobs[i].n_rows <= 0 || jobs[i].n_cols <= 0) continue;
            PP_CHECK_ARG(jobs[i].X && jobs[i].out, "pp_colsum_f32: null pointer");
            pack.cs[pack.n_colsum++] = jobs[i];
        }
        if (pack.n_colsum == 0) continue;
        if (!launched) pack.fin = fin;      // only the first launch finalises
        aux_layout(pack, false);
        PP_TRY(aux_jobs_launch(pack, st));
        launched = true;
    }
    if (!launched && fin.acc) return loss_finalize(fin_acc, fin_flag, fin_traces, fin_loss, fin_status, st);   // (no job)
    return 0;
}

// the jobs of aux_jobs.hpp as their own launch (when they cannot ride behind the weight-gradient tiles)
__global__ __launch_bounds__(256) void aux_jobs_kernel(const AuxJobs jobs) {
    __shared__ float lds[2048];
    aux_job_run(jobs, (int)blockIdx.x, lds);
}

int aux_jobs_launch(const AuxJobs& jobs, hipStream_t st) {
    if (jobs.n_blocks <= 0) return 0;
    hipLaunchKernelGGL(aux_jobs_kernel, dim3(jobs.n_blocks), dim3(256), 0, st, jobs);
    PP_LAUNCH_CHECK("pp_aux_jobs");
    return 0;
}

int colsum_f32(const float* X, int64_t ldx, const int32_t* idx, int n_rows, int n_cols, float* out, float* out2,
               hipStream_t st) {
    PP_CHECK_ARG(X && out, "pp_colsum_f32: null pointer");
    ColsumJob j{X, ldx, idx, n_rows, n_cols, out, out2, nullptr, 0, 0};
    return colsum_multi(&j, 1, st, nullptr, nullptr, 0, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------------
// LSTM input rows: x = [E | s_prev | d_prev | a_prev | d_cur | a_cur]   (inference_network_lstm.py:146-181)
// One thread per output element; consecutive lanes write consecutive columns (coalesced row writes); the
// embedding rows are looked up through the per-address offset table (address dispatch).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_input_gather_kernel(
    GatherDims d, const float* __restrict__ params, const int64_t* __restrict__ at, const float* __restrict__ E,
    int64_t e_stride, const int32_t* __restrict__ trace, const float* __restrict__ value,
    const int32_t* __restrict__ addr, const int32_t* __restrict__ prev_row, int32_t fixed_addr,
    int32_t fixed_prev_addr, int n_rows, float* __restrict__ X, int64_t ldx, float* __restrict__ zero_like,
    float* __restrict__ zero_small, int n_small, int xcols, const AddrBias ab) {
    __shared__ float ab_lds[ADDR_BIAS_LDS];
    if (ab.AB && (int)blockIdx.x >= ab.first_block) {   // extra workgroups: per-address bias vectors (gather.hpp)
        addr_bias_block(ab, (int)blockIdx.x - ab.first_block, ab_lds);
        return;
    }
    // The training step's first kernel also clears two accumulators that later kernels add into (saves two memset
    // launches, ~5 us each): dX (same shape as X; the split-K data-gradient product accumulates into it) and the loss
    // slots.
    if (zero_small && blockIdx.x == 0)
        for (int q = threadIdx.x; q < n_small; q += 256) zero_small[q] = 0.0f;
    const int nb = ab.AB ? ab.first_block : (int)gridDim.x;   // workgroups that gather
    // xcols < I: compact rows [E | s_prev]; the table columns enter the product as a per-address bias instead
    const int64_t total = (int64_t)n_rows * xcols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)nb * 256) {
        const int r = (int)(e / xcols);
        const int c = (int)(e - (int64_t)r * xcols);
        float out;
        if (c < d.e_obs) {
            const int64_t b = trace ? (int64_t)trace[r] : (int64_t)r;
            out = E[b * e_stride + c];
        } else {
            int ap;
            float v = 0.0f;
            if (prev_row) {
                const int pr = prev_row[r];
                ap = pr < 0 ? -1 : addr[pr];
                if (pr >= 0) v = value[pr];
            } else {  // lock-step IS: every row has the same previous address, value[r] is the previous value
                ap = fixed_prev_addr;
                if (ap >= 0) v = value[r];
            }
            out = gather_embedding_elem(d, params, at, c, ap, v, addr ? addr[r] : fixed_addr);
        }
        X[(int64_t)r * ldx + c] = out;
        if (zero_like) zero_like[(int64_t)r * ldx + c] = 0.0f;
    }
}

int lstm_input_gather(const pp_net* net, const float* params, const float* E, int64_t e_stride, const int32_t* trace,
                      const float* value, const int32_t* addr, const int32_t* prev_row, int32_t fixed_addr,
                      int32_t fixed_prev_addr, int n_rows, float* X, int64_t ldx, hipStream_t st, float* zero_like,
                      float* zero_small, int n_small, int xcols, const AddrBias* bias) {
    PP_CHECK_ARG(net && params && E && X && net->addr_table, "pp_lstm_input_gather: null pointer");
    if (n_rows <= 0) return 0;
    GatherDims d{net->e_obs, net->smp_dim, net->dtype_dim, net->addr_dim, net->lstm_in};
    PP_CHECK_ARG(d.I == d.e_obs + d.smp + 2 * (d.dtype + d.addr), "pp_lstm_input_gather: lstm_in mismatch");
    if (xcols <= 0) xcols = d.I;
    const int64_t total = (int64_t)n_rows * xcols;
    int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
    AddrBias ab{};
    if (bias && bias->AB) {
        ab = *bias;
        ab.first_block = blocks;
        blocks += addr_bias_blocks(ab);
    }
    hipLaunchKernelGGL(lstm_input_gather_kernel, dim3(blocks), dim3(256), 0, st, d, params, net->addr_table, E, e_stride,
                       trace, value, addr, prev_row, fixed_addr, fixed_prev_addr, n_rows, X, ldx, zero_like, zero_small, n_small,
                       xcols, ab);
    PP_LAUNCH_CHECK("pp_lstm_input_gather");
    return 0;
}

// FeedForward network (pyprob/nn/inference_network_feedforward.py:72,85): the proposal layer of every time step reads
// the observe embedding of its trace, so the "hidden state" rows of the heads are copies of E; the same launch clears
// the loss slots (it is the first kernel after the observe embedding).
__global__ __launch_bounds__(256) void embedding_rows_kernel(const float* __restrict__ E, int64_t lde,
                                                             const int32_t* __restrict__ trace, int n_rows, int e_obs,
                                                             float* __restrict__ Hs, int64_t ldh,
                                                             float* __restrict__ zero_small, int n_small) {
    if (zero_small && blockIdx.x == 0)
        for (int q = threadIdx.x; q < n_small; q += 256) zero_small[q] = 0.0f;
    const int64_t total = (int64_t)n_rows * e_obs;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / e_obs), c = (int)(e - (int64_t)r * e_obs);
        Hs[(int64_t)r * ldh + c] = E[(int64_t)trace[r] * lde + c];
    }
}

int embedding_rows(const float* E, int64_t lde, const int32_t* trace, int n_rows, int e_obs, float* Hs, int64_t ldh,
                   float* zero_small, int n_small, hipStream_t st) {
    const int64_t total = (int64_t)n_rows * e_obs;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 4096));
    hipLaunchKernelGGL(embedding_rows_kernel, dim3(blocks), dim3(256), 0, st, E, lde, trace, n_rows, e_obs, Hs, ldh,
                       zero_small, n_small);
    PP_LAUNCH_CHECK("embedding_rows");
    return 0;
}

// Gradient of the sample-embedding layers (one Linear + ReLU per address) from dX[:, e_obs : e_obs+smp].
// lane -> row, wave -> embedding unit j (a workgroup = 64 rows x 4 units: one thread per row walked the units one after the
// other - four dependent chains of table loads, reductions and atomics, 10.7 us for the 1 800 rows of a ragged minibatch); when a
// wave's rows share the previous address (the common case in step-major order) the contributions are wave-reduced and one lane
// issues the atomics.
__global__ __launch_bounds__(256) void sample_embed_bwd_kernel(GatherDims d, const float* __restrict__ params,
                                                               const int64_t* __restrict__ at,
                                                               const float* __restrict__ value,
                                                               const int32_t* __restrict__ addr,
                                                               const int32_t* __restrict__ prev_row, int row_begin,
                                                               int n_rows, const float* __restrict__ dX, int64_t ldx,
                                                               float* __restrict__ grads) {
    const int r = row_begin + blockIdx.x * 64 + (threadIdx.x & 63);
    const int j0 = threadIdx.x >> 6;
    const bool live = r < n_rows;
    int ap = -1;
    float v = 0.0f;
    if (live) {
        const int pr = prev_row[r];
        if (pr >= 0) {
            ap = addr[pr];
            v = value[pr];
        }
    }
    const int ap0 = __builtin_amdgcn_readfirstlane(ap);
    const bool uniform = __all(ap == ap0);
    if (uniform && ap0 < 0) return;
    for (int j = j0; j < d.smp; j += 4) {
        float ds = 0.0f;
        int smp_in = 1, cat = 0;
        if (ap >= 0) {
            smp_in = (int)at[ap * PP_ADDR_TABLE_COLS + PP_AT_SMP_IN];
            const float s = sample_embed_elem(params, at, ap, j, v);
            ds = s > 0.0f ? dX[(int64_t)r * ldx + d.e_obs + j] : 0.0f;
            if (smp_in > 1) {
                cat = (int)v;
                cat = cat < 0 ? 0 : (cat >= smp_in ? smp_in - 1 : cat);
            }
        }
        const int smp_in0 = __builtin_amdgcn_readfirstlane(smp_in);
        if (uniform && smp_in0 == 1) {
            const float sw = wave_sum(ds * v), sb = wave_sum(ds);
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(grads + at[ap0 * PP_ADDR_TABLE_COLS + PP_AT_SMP_W] + j, sw);
                atomicAdd(grads + at[ap0 * PP_ADDR_TABLE_COLS + PP_AT_SMP_B] + j, sb);
            }
        } else if (ap >= 0 && ds != 0.0f) {
            float* gw = grads + at[ap * PP_ADDR_TABLE_COLS + PP_AT_SMP_W];
            float* gb = grads + at[ap * PP_ADDR_TABLE_COLS + PP_AT_SMP_B];
            if (smp_in == 1) atomicAdd(gw + j, ds * v);
            else atomicAdd(gw + j * smp_in + cat, ds);
            atomicAdd(gb + j, ds);
        }
    }
}

// Deterministic variant: thread o of workgroup (address a) owns ONE output of the address's sample-embedding layer -
// weight element (j, k) for o < smp * smp_in, bias j after that - and scans the rows whose PREVIOUS variable has address
// a (nxt_rows[q0 .. q0 + m), fixed order) on its own. No atomics, one writer per gradient element.
struct SmpDetGroups {
    int addr[32], q0[32], m[32];
};
__global__ __launch_bounds__(64) void sample_embed_bwd_det_kernel(GatherDims d, const float* __restrict__ params,
                                                                  const int64_t* __restrict__ at,
                                                                  const float* __restrict__ value,
                                                                  const int32_t* __restrict__ prev_row,
                                                                  const int32_t* __restrict__ nxt_rows, SmpDetGroups g,
                                                                  const float* __restrict__ dX, int64_t ldx,
                                                                  float* __restrict__ grads) {
    const int a = g.addr[blockIdx.y], q0 = g.q0[blockIdx.y], m = g.m[blockIdx.y];
    const int smp_in = (int)at[a * PP_ADDR_TABLE_COLS + PP_AT_SMP_IN];
    const int n_w = d.smp * smp_in, o = blockIdx.x * 64 + threadIdx.x;
    if (o >= n_w + d.smp) return;
    const bool is_bias = o >= n_w;
    const int j = is_bias ? o - n_w : o / smp_in, k = is_bias ? 0 : o % smp_in;
    float acc = 0.0f;
    for (int q = 0; q < m; ++q) {
        const int r = nxt_rows[q0 + q];
        const float v = value[prev_row[r]];
        const float s = sample_embed_elem(params, at, a, j, v);
        const float ds = s > 0.0f ? dX[(int64_t)r * ldx + d.e_obs + j] : 0.0f;
        if (is_bias) acc += ds;
        else if (smp_in == 1) acc += ds * v;
        else {
            int cat = (int)v;
            cat = cat < 0 ? 0 : (cat >= smp_in ? smp_in - 1 : cat);
            if (cat == k) acc += ds;
        }
    }
    float* dst = grads + at[a * PP_ADDR_TABLE_COLS + (is_bias ? PP_AT_SMP_B : PP_AT_SMP_W)] + (is_bias ? j : o);
    *dst += acc;
}

int sample_embed_bwd_det(const pp_net* net, const float* params, const float* value, const int32_t* prev_row,
                         const int32_t* nxt_rows, const int32_t* nxt_off, const float* dX, int64_t ldx, float* grads,
                         hipStream_t st) {
    GatherDims d{net->e_obs, net->smp_dim, net->dtype_dim, net->addr_dim, net->lstm_in};
    int a = 0;
    while (a < net->n_addr) {
        SmpDetGroups g;
        int n = 0, max_out = 0;
        for (; a < net->n_addr && n < 32; ++a) {
            const int m = nxt_off[a + 1] - nxt_off[a];
            if (m <= 0) continue;
            g.addr[n] = a; g.q0[n] = nxt_off[a]; g.m[n] = m;
            max_out = std::max(max_out, net->smp_dim * (net->addrs[a].smp_in + 1));
            ++n;
        }
        if (n == 0) continue;
        hipLaunchKernelGGL(sample_embed_bwd_det_kernel, dim3(cdiv(max_out, 64), n), dim3(64), 0, st, d, params, net->addr_table,
                           value, prev_row, nxt_rows, g, dX, ldx, grads);
        PP_LAUNCH_CHECK("sample_embed_bwd (deterministic)");
    }
    return 0;
}

int sample_embed_bwd(const pp_net* net, const float* params, const float* value, const int32_t* addr,
                     const int32_t* prev_row, int row_begin, int n_rows, const float* dX, int64_t ldx, float* grads,
                     hipStream_t st) {
    if (n_rows - row_begin <= 0) return 0;
    GatherDims d{net->e_obs, net->smp_dim, net->dtype_dim, net->addr_dim, net->lstm_in};
    hipLaunchKernelGGL(sample_embed_bwd_kernel, dim3(cdiv(n_rows - row_begin, 64)), dim3(256), 0, st, d, params,
                       net->addr_table, value, addr, prev_row, row_begin, n_rows, dX, ldx, grads);
    PP_LAUNCH_CHECK("sample_embed_bwd");
    return 0;
}

// dE[b, c] = (E[b,c] > 0) * sum_t dX[row_off[t] + b, c]: gradient into the observe embedding output, with the
// ReLU mask of the final embedding layer applied (deterministic: no atomics).
__global__ __launch_bounds__(256) void obs_grad_kernel(const float* __restrict__ dX, int64_t ldx,
                                                       const int32_t* __restrict__ row_off, int t_max, int n_traces,
                                                       int e_obs, const float* __restrict__ E, int64_t lde,
                                                       float* __restrict__ dE, int64_t ldde) {
    const int64_t total = (int64_t)n_traces * e_obs;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int b = (int)(e / e_obs), c = (int)(e - (int64_t)b * e_obs);
        float acc = 0.0f;
        for (int t = 0; t < t_max; ++t) {
            const int r0 = row_off[t], n = row_off[t + 1] - r0;
            if (b >= n) break;
            acc += dX[(int64_t)(r0 + b) * ldx + c];
        }
        dE[(int64_t)b * ldde + c] = E[(int64_t)b * lde + c] > 0.0f ? acc : 0.0f;
    }
}

int obs_grad(const float* dX, int64_t ldx, const int32_t* row_off_dev, int t_max, int n_traces, int e_obs,
             const float* E, int64_t lde, float* dE, int64_t ldde, hipStream_t st) {
    const int64_t total = (int64_t)n_traces * e_obs;
    if (total <= 0) return 0;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(obs_grad_kernel, dim3(blocks), dim3(256), 0, st, dX, ldx, row_off_dev, t_max, n_traces, e_obs, E,
                       lde, dE, ldde);
    PP_LAUNCH_CHECK("obs_grad");
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// LSTM cell (torch.nn.LSTM gate order i, f, g, o)
// ------------------------------------------------------------------------------------------------------
constexpr int CELL_ROWS = 8;    // rows per workgroup; 256 threads = 64 hidden units x 4 row lanes
constexpr int CELL_ROWS_BWD = 8;    // (32 rows per workgroup - 4x fewer bias-gradient atomics per line - was slower: 8.5 -> 11 us)

__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(float* __restrict__ G, const float* __restrict__ c_prev,
                                                            float* __restrict__ c, float* __restrict__ h, int n,
                                                            int H, int c_prev_shared) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    if (j >= H) return;
#pragma unroll
    for (int q = 0; q < CELL_ROWS / 4; ++q) {
        const int r = blockIdx.y * CELL_ROWS + rl + 4 * q;
        if (r >= n) break;
        float* g = G + (int64_t)r * 4 * H;
        const int64_t e = (int64_t)r * H + j;
        const float gi = sigmoidf_(g[j]);
        const float gg = tanhf(g[2 * H + j]);
        const float go = sigmoidf_(g[3 * H + j]);
        // first time step (h0 = c0 = 0, inference_network_lstm.py:186-187): the forget gate multiplies zero; its
        // pre-activation is not even computed (engine.hip, zero blocks of the input GEMM) and 0 is recorded for backward
        float gf = 0.0f, cn = gi * gg;
        if (c_prev) {
            gf = sigmoidf_(g[H + j]);
            cn += gf * c_prev[c_prev_shared ? j : e];
        }
        g[j] = gi;
        g[H + j] = gf;
        g[2 * H + j] = gg;
        g[3 * H + j] = go;
        c[e] = cn;
        h[e] = go * tanhf(cn);
    }
}

int lstm_cell_fwd(float* G, const float* c_prev, float* c, float* h, int n, int H, hipStream_t st, int c_prev_shared) {
    PP_CHECK_ARG(G && c && h && H > 0, "pp_lstm_cell_fwd: bad argument");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(cdiv(H, 64), cdiv(n, CELL_ROWS)), dim3(256), 0, st, G, c_prev, c, h, n, H,
                       c_prev_shared);
    PP_LAUNCH_CHECK("pp_lstm_cell_fwd");
    return 0;
}

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(float* __restrict__ G, const float* __restrict__ c_prev,
                                                            const float* __restrict__ c,
                                                            const float* __restrict__ dh,
                                                            float* __restrict__ dc_carry, int n, int n_next, int H,
                                                            float* __restrict__ db, float* __restrict__ db2,
                                                            LossFinalize fin, const float* __restrict__ dh_parts,
                                                            int n_parts, int64_t part_stride) {
    __shared__ float part[4][4][64];
    // the backward pass's first cell launch also turns the loss slots into the loss (one launch less per step)
    if (fin.acc && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) loss_finalize_inline(fin);
    const int jl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < H) {
#pragma unroll
        for (int q = 0; q < CELL_ROWS_BWD / 4; ++q) {
            const int r = blockIdx.y * CELL_ROWS_BWD + rl + 4 * q;
            if (r >= n) break;
            float* g = G + (int64_t)r * 4 * H;
            const int64_t e = (int64_t)r * H + j;
            const float gi = g[j], gf = g[H + j], gg = g[2 * H + j], go = g[3 * H + j];
            const float tc = tanhf(c[e]);
            float dhv = dh[e];
            // dh_t of the rows that have a next time step also holds dG_{t+1} W_hh: stored as the partial tiles of that
            // product's K splits (no float atomics there), added here - up to eight loads in flight
            if (n_parts > 0 && r < n_next) {
                float pv[8];
#pragma unroll
                for (int z = 0; z < 8; ++z) pv[z] = z < n_parts ? dh_parts[(int64_t)z * part_stride + e] : 0.0f;
                dhv += ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
            }
            const float dc = (r < n_next ? dc_carry[e] : 0.0f) + dhv * go * (1.0f - tc * tc);
            const float cp = c_prev ? c_prev[e] : 0.0f;
            const float d0 = dc * gg * gi * (1.0f - gi);
            const float d1 = dc * cp * gf * (1.0f - gf);
            const float d2 = dc * gi * (1.0f - gg * gg);
            const float d3 = dhv * tc * go * (1.0f - go);
            g[j] = d0; g[H + j] = d1; g[2 * H + j] = d2; g[3 * H + j] = d3;
            dc_carry[e] = dc * gf;
            s0 += d0; s1 += d1; s2 += d2; s3 += d3;
        }
    }
    if (!db) return;
    // fused bias gradient: db[gate*H + j] += sum over this workgroup's rows
    part[rl][0][jl] = s0; part[rl][1][jl] = s1; part[rl][2][jl] = s2; part[rl][3][jl] = s3;
    __syncthreads();
    if (j < H) {
        const int gate = rl;
        const float t = part[0][gate][jl] + part[1][gate][jl] + part[2][gate][jl] + part[3][gate][jl];
        atomicAdd(db + gate * H + j, t);
        if (db2) atomicAdd(db2 + gate * H + j, t);
    }
}

int lstm_cell_bwd(float* G, const float* c_prev, const float* c, const float* dh, float* dc_carry, int n, int n_next,
                  int H, float* db, float* db2, hipStream_t st, const float* fin_acc, const int32_t* fin_flag,
                  int fin_traces, float* fin_loss, int32_t* fin_status, const float* dh_parts, int n_parts,
                  int64_t part_stride) {
    PP_CHECK_ARG(G && c && dh && dc_carry && H > 0 && n_next <= n, "pp_lstm_cell_bwd: bad argument");
    if (n <= 0) return 0;
    LossFinalize fin{fin_acc, fin_flag, fin_traces > 0 ? 1.0f / (float)fin_traces : 0.0f, fin_loss, fin_status};
    PP_CHECK_ARG(n_parts >= 0 && n_parts <= 8 && (n_parts == 0 || dh_parts), "pp_lstm_cell_bwd: bad split partials");
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(cdiv(H, 64), cdiv(n, CELL_ROWS_BWD)), dim3(256), 0, st, G, c_prev, c, dh,
                       dc_carry, n, n_next, H, db, db2, fin, dh_parts, n_parts, part_stride);
    PP_LAUNCH_CHECK("pp_lstm_cell_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Proposal heads: transforms + log_prob + d log_prob / d y, one row per lane, loss reduced per wavefront.
// ------------------------------------------------------------------------------------------------------
constexpr int MAXK = 16;
constexpr float kFp32Eps = 1.1920928955078125e-07f;   // torch.finfo(float32).eps (util.clamp_probs)
constexpr float kHalfLog2Pi = 0.91893853320467274178f;
constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kInvSqrt2Pi = 0.39894228040143267794f;
constexpr float kLogEps = -18.420680743952367f;        // log(1e-8), pyprob/util.py:35

__device__ __forceinline__ float std_cdf(float x) { return 0.5f * (1.0f + erff(x * kInvSqrt2)); }
__device__ __forceinline__ float std_pdf(float x) { return kInvSqrt2Pi * expf(-0.5f * x * x); }

struct MixtureParams {
    float mu[MAXK], sd[MAXK], pi[MAXK], p[MAXK];
    float pisum;
};

// KIND 0: Normal components around a Normal prior; KIND 1: TruncatedNormal components inside a Uniform prior
// (stddev = range/1000 + sigmoid(y) 10 range); KIND 2: the Poisson head - TruncatedNormal components on the fixed
// interval [pa, pb] = [0, 40] with stddev = exp(y) (proposal_poisson_truncated_normal_mixture.py:19-37).
template <int KIND>
__device__ __forceinline__ void mixture_params(const float* __restrict__ y, int K, float pa, float pb,
                                               MixtureParams& m, float sm[MAXK], float ss[MAXK]) {
    float zmax = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) zmax = fmaxf(zmax, y[2 * K + k]);
    float zs = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) {
            m.pi[k] = expf(y[2 * K + k] - zmax);
            zs += m.pi[k];
        }
    float ps = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) {
            m.pi[k] = m.pi[k] / zs;
            ps += m.pi[k];
        }
    m.pisum = ps;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) {
            m.p[k] = m.pi[k] / ps;
            if (KIND == 0) {
                m.mu[k] = pa + y[k] * pb;
                m.sd[k] = expf(y[K + k]) * pb;
            } else {
                const float rng = pb - pa;
                sm[k] = sigmoidf_(y[k]);
                ss[k] = sigmoidf_(y[K + k]);
                m.mu[k] = pa + sm[k] * rng;
                m.sd[k] = KIND == 2 ? expf(y[K + k]) : rng / 1000.0f + ss[k] * rng * 10.0f;
            }
        }
}

// log q(v) and responsibilities; returns lp (may be -inf for TruncatedNormal outside [low, high])
template <int KIND>
__device__ __forceinline__ float mixture_logprob(const MixtureParams& m, int K, float v, float low, float high,
                                                 float a[MAXK]) {
    float amax = -INFINITY;
    const bool inside = (KIND == 0) || (v >= low && v <= high);
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) {
            const float lpk = logf(fminf(fmaxf(m.p[k], kFp32Eps), 1.0f - kFp32Eps));
            const float t = (v - m.mu[k]) / m.sd[k];
            float comp;
            if (KIND == 0) {
                comp = -0.5f * t * t - logf(m.sd[k]) - kHalfLog2Pi;
            } else {
                const float alpha = (low - m.mu[k]) / m.sd[k], beta = (high - m.mu[k]) / m.sd[k];
                const float Z = std_cdf(beta) - std_cdf(alpha);
                comp = (inside ? 0.0f : -INFINITY) + (-0.5f * t * t - kHalfLog2Pi) - logf(m.sd[k] * Z);
            }
            a[k] = lpk + comp;
            amax = fmaxf(amax, a[k]);
        }
    if (!(amax > -INFINITY)) return amax;  // -inf (or NaN)
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) s += expf(a[k] - amax);
    return amax + logf(s);
}

template <int KIND>
__global__ __launch_bounds__(64) void head_mixture_kernel(const float* __restrict__ Y, int64_t ldy,
                                                           const int32_t* __restrict__ rows,
                                                           const float* __restrict__ value,
                                                           const float* __restrict__ prior, int n, int K,
                                                           float grad_scale, float* __restrict__ lp_out,
                                                           float* __restrict__ DY, float* __restrict__ loss_acc,
                                                           int32_t* __restrict__ nonfinite) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    float contrib = 0.0f;
    bool bad = false;
    if (i < n) {
        const int r = rows ? rows[i] : i;
        const float* y = Y + (int64_t)i * ldy;
        const float v = value[r], pa = prior[2 * r], pb = prior[2 * r + 1];
        MixtureParams m;
        float sm[MAXK], ss[MAXK], a[MAXK];
        mixture_params<KIND>(y, K, pa, pb, m, sm, ss);
        float lp = mixture_logprob<KIND>(m, K, v, pa, pb, a);
        if (lp_out) lp_out[r] = lp;
        const bool rescued = (lp == -INFINITY);
        bad = !rescued && !isfinite(lp);
        contrib = rescued ? -kLogEps : -lp;
        if (DY) {
            float* dy = DY + (int64_t)i * ldy;
            if (rescued || bad) {
                for (int k = 0; k < 3 * K; ++k) dy[k] = 0.0f;
            } else {
                // responsibilities and the softmax/normalisation chain (see oracle/ic_oracle.py head_*_mixture)
                float resp[MAXK], dp[MAXK];
                float dpp = 0.0f;
#pragma unroll
                for (int k = 0; k < MAXK; ++k)
                    if (k < K) {
                        resp[k] = expf(a[k] - lp);
                        const bool in = (m.p[k] >= kFp32Eps) && (m.p[k] <= 1.0f - kFp32Eps);
                        dp[k] = in ? resp[k] / m.p[k] : 0.0f;
                        dpp += dp[k] * m.p[k];
                    }
                float dpipi = 0.0f;
#pragma unroll
                for (int k = 0; k < MAXK; ++k)
                    if (k < K) {
                        dp[k] = (dp[k] - dpp) / m.pisum;  // d lp / d pi_k
                        dpipi += dp[k] * m.pi[k];
                    }
#pragma unroll
                for (int k = 0; k < MAXK; ++k)
                    if (k < K) {
                        const float t = (v - m.mu[k]) / m.sd[k];
                        float dmu, dsd;
                        if (KIND == 0) {
                            dmu = resp[k] * t / m.sd[k];
                            dsd = resp[k] * (t * t - 1.0f) / m.sd[k];
                            dy[k] = grad_scale * dmu * pb;
                            dy[K + k] = grad_scale * dsd * m.sd[k];
                        } else {
                            const float rng = pb - pa;
                            const float alpha = (pa - m.mu[k]) / m.sd[k], beta = (pb - m.mu[k]) / m.sd[k];
                            const float Z = std_cdf(beta) - std_cdf(alpha);
                            const float fa = std_pdf(alpha), fb = std_pdf(beta);
                            dmu = resp[k] * (t / m.sd[k] - (fa - fb) / (m.sd[k] * Z));
                            dsd = resp[k] * ((t * t - 1.0f) / m.sd[k] - (alpha * fa - beta * fb) / (m.sd[k] * Z));
                            dy[k] = grad_scale * dmu * rng * sm[k] * (1.0f - sm[k]);
                            dy[K + k] = grad_scale * dsd * (KIND == 2 ? m.sd[k] : rng * 10.0f * ss[k] * (1.0f - ss[k]));
                        }
                        dy[2 * K + k] = grad_scale * m.pi[k] * (dp[k] - dpipi);
                    }
            }
        }
    }
    // wavefront-reduced loss accumulator: one atomic per wave
    const float ws = wave_sum(contrib);
    if ((threadIdx.x & 63) == 0 && loss_acc && ws != 0.0f) atomicAdd(loss_acc, ws);
    if (bad && nonfinite) atomicOr(nonfinite, 1);
}

// Categorical head: probs = softmax(y) + 1e-8, renormalised and clamped by torch.distributions.Categorical.
__global__ __launch_bounds__(64) void head_categorical_kernel(const float* __restrict__ Y, int64_t ldy,
                                                               const int32_t* __restrict__ rows,
                                                               const float* __restrict__ value, int n, int C,
                                                               float grad_scale, float* __restrict__ lp_out,
                                                               float* __restrict__ DY, float* __restrict__ loss_acc,
                                                               int32_t* __restrict__ nonfinite) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    float contrib = 0.0f;
    bool bad = false;
    if (i < n) {
        const int r = rows ? rows[i] : i;
        const float* y = Y + (int64_t)i * ldy;
        int vi = (int)value[r];
        vi = vi < 0 ? 0 : (vi >= C ? C - 1 : vi);
        float zmax = -INFINITY;
        for (int k = 0; k < C; ++k) zmax = fmaxf(zmax, y[k]);
        float zs = 0.0f;
        for (int k = 0; k < C; ++k) zs += expf(y[k] - zmax);
        float S = 0.0f, pisum = 0.0f;
        for (int k = 0; k < C; ++k) {
            const float pik = expf(y[k] - zmax) / zs;
            S += pik + 1e-8f;
            pisum += pik;
        }
        const float piv = expf(y[vi] - zmax) / zs;
        const float pv = (piv + 1e-8f) / S;
        const float lp = pv == pv ? logf(fminf(fmaxf(pv, kFp32Eps), 1.0f - kFp32Eps)) : pv;   // (clamp would drop a NaN)
        if (lp_out) lp_out[r] = lp;
        bad = !isfinite(lp);
        contrib = -lp;
        if (DY) {
            float* dy = DY + (int64_t)i * ldy;
            const bool in = (pv >= kFp32Eps) && (pv <= 1.0f - kFp32Eps);
            const float dpv = in ? 1.0f / pv : 0.0f;      // d lp / d p_v
            const float dpp = dpv * pv;                   // sum_k dp_k p_k
            // dq_k = (dp_k - dpp) / S ; sum_k dq_k pi_k = (dpv*piv - dpp*pisum) / S
            const float dqpi = (dpv * piv - dpp * pisum) / S;
            for (int k = 0; k < C; ++k) {
                const float pik = expf(y[k] - zmax) / zs;
                const float dq = ((k == vi ? dpv : 0.0f) - dpp) / S;
                dy[k] = bad ? 0.0f : grad_scale * pik * (dq - dqpi);
            }
        }
    }
    const float ws = wave_sum(contrib);
    if ((threadIdx.x & 63) == 0 && loss_acc && ws != 0.0f) atomicAdd(loss_acc, ws);
    if (bad && nonfinite) atomicOr(nonfinite, 1);
}

// Bernoulli head as the reference's `_loss` evaluates it (PP_HEAD_BERNOULLI, pyprob_amd.h): the row's proposal
// p = sigmoid(y) + 1e-8 is scored against all n values of its sub-batch step, of which n1 are 1 (prior = (n, n1)):
// lp_row = n1 log p + (n - n1) log(1 - p), probs clamped to [eps, 1 - eps] like torch's probs_to_logits.
__global__ __launch_bounds__(64) void head_bernoulli_kernel(const float* __restrict__ Y, int64_t ldy,
                                                             const int32_t* __restrict__ rows,
                                                             const float* __restrict__ prior, int n, float grad_scale,
                                                             float* __restrict__ lp_out, float* __restrict__ DY,
                                                             float* __restrict__ loss_acc,
                                                             int32_t* __restrict__ nonfinite) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    float contrib = 0.0f;
    bool bad = false;
    if (i < n) {
        const int r = rows ? rows[i] : i;
        const float y = Y[(int64_t)i * ldy];
        const float cnt = prior[2 * (int64_t)r], ones = prior[2 * (int64_t)r + 1];
        const float sg = sigmoidf_(y);
        const float p = sg + 1e-8f;
        const bool in = p >= kFp32Eps && p <= 1.0f - kFp32Eps;
        const float pc = fminf(fmaxf(p, kFp32Eps), 1.0f - kFp32Eps);
        const float lp = p == p ? ones * logf(pc) + (cnt - ones) * log1pf(-pc) : p;
        if (lp_out) lp_out[r] = lp;
        bad = !isfinite(lp);
        contrib = -lp;
        if (DY) {
            const float dp = in ? ones / pc - (cnt - ones) / (1.0f - pc) : 0.0f;
            DY[(int64_t)i * ldy] = bad ? 0.0f : grad_scale * dp * sg * (1.0f - sg);
        }
    }
    const float ws = wave_sum(contrib);
    if ((threadIdx.x & 63) == 0 && loss_acc && ws != 0.0f) atomicAdd(loss_acc, ws);
    if (bad && nonfinite) atomicOr(nonfinite, 1);
}

int head_logprob(int kind, const float* y, int64_t ldy, const int32_t* rows, const float* value, const float* prior,
                 int n, int n_out, float grad_scale, float* lp_out, float* dy, float* loss_acc, int32_t* nonfinite,
                 hipStream_t st) {
    PP_CHECK_ARG(y && value, "pp_head_logprob: null pointer");
    if (n <= 0) return 0;
    dim3 grid(cdiv(n, 64)), block(64);   // one wavefront per workgroup: spread the transcendental-heavy rows over CUs
    if (kind == PP_HEAD_CATEGORICAL) {
        hipLaunchKernelGGL(head_categorical_kernel, grid, block, 0, st, y, ldy, rows, value, n, n_out, grad_scale,
                           lp_out, dy, loss_acc, nonfinite);
    } else if (kind == PP_HEAD_BERNOULLI) {
        PP_CHECK_ARG(prior && n_out == 1, "pp_head_logprob: the Bernoulli head needs (n, n1) per row and n_out = 1");
        hipLaunchKernelGGL(head_bernoulli_kernel, grid, block, 0, st, y, ldy, rows, prior, n, grad_scale, lp_out, dy,
                           loss_acc, nonfinite);
    } else {
        PP_CHECK_ARG(prior, "pp_head_logprob: mixture heads need prior parameters");
        PP_CHECK_ARG(n_out % 3 == 0 && n_out / 3 <= MAXK && n_out > 0,
                     "pp_head_logprob: mixture heads support 1..%d components (got n_out=%d)", MAXK, n_out);
        if (kind == PP_HEAD_NORMAL_MIXTURE)
            hipLaunchKernelGGL(head_mixture_kernel<0>, grid, block, 0, st, y, ldy, rows, value, prior, n, n_out / 3,
                               grad_scale, lp_out, dy, loss_acc, nonfinite);
        else if (kind == PP_HEAD_TRUNCNORMAL_MIXTURE)
            hipLaunchKernelGGL(head_mixture_kernel<1>, grid, block, 0, st, y, ldy, rows, value, prior, n, n_out / 3,
                               grad_scale, lp_out, dy, loss_acc, nonfinite);
        else if (kind == PP_HEAD_POISSON_TN_MIXTURE)
            hipLaunchKernelGGL(head_mixture_kernel<2>, grid, block, 0, st, y, ldy, rows, value, prior, n, n_out / 3,
                               grad_scale, lp_out, dy, loss_acc, nonfinite);
        else
            PP_CHECK_ARG(false, "pp_head_logprob: unknown head kind %d", kind);
    }
    PP_LAUNCH_CHECK("pp_head_logprob");
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Fused proposal-head tail for the mixture heads: second FF layer (y = a1 W2^T + b2), proposal transforms, mixture
// log_prob, loss, d lp/d y, bias gradients and the masked data gradient dz1 = (dy W2) * [a1 > 0] in ONE kernel.
// One wavefront per row: lanes split the hidden units (coalesced a1 / dz1 rows, conflict-free LDS reads of W2),
// wave reductions give y; lanes 0..K-1 then own one mixture component each, so the exp/log/erf work is spread over
// lanes instead of serialised in one thread. W2 is staged once per workgroup in LDS.
// Replaces: Y GEMM (unaligned W2 rows -> scalar loads), head_mixture_kernel, dz1 GEMM, colsum(db2) = 4 launches.
// ------------------------------------------------------------------------------------------------------
constexpr int TAIL_LDS_FLOATS = 32768;   // dynamic LDS budget of the fused head tail (128 KB)

// NQ4 = float4 groups per lane: lane l owns hidden units j = 256 q + 4 l + e (q < NQ4, e < 4), so a1 / dz1 rows move as
// 16-byte accesses and every W2 row (LDS stride hid4 = round4(hid)) is read with one ds_read_b128 per 4 FMAs.
constexpr int TAIL_MAX_JOBS = 16;
struct TailJob {   // one address group: rows [0, n) of the group-compact head buffers
    const float* A1; const float* W2; const float* b2; const int32_t* rows; float* DY; float* dZ1; int n;
};
struct TailJobs {
    TailJob j[TAIL_MAX_JOBS];
};

__device__ __forceinline__ void wave_lds_sync() {   // order this wave's LDS writes before its LDS reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS: W2 as [n_out][ws] (ws = hid made odd: lane o of the layer-2 products reads row o, an odd stride spreads the 32
// rows over the 32 banks; when hid is odd already the image is the parameter tensor itself and is staged with one
// round of 16-byte loads), then per wave the activation row a1 [hid4] and the 64 + 64 floats of y and dy.
template <int KIND, int NE>
__global__ __launch_bounds__(256) void head_tail_kernel(const TailJobs jobs, int64_t lda1, int hid, int K,
                                                        const float* __restrict__ value,
                                                        const float* __restrict__ prior, int rows_per_wave,
                                                        float grad_scale, float* __restrict__ lp_out, int64_t lddy,
                                                        int64_t lddz, float* __restrict__ loss_acc,
                                                        int32_t* __restrict__ nonfinite, long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) float tail_lds[];
    const TailJob& jb = jobs.j[blockIdx.y];   // blockIdx.y = address group (same head kind and shape in one launch)
    const int n = jb.n;
    if ((int)blockIdx.x * 4 * rows_per_wave >= n) return;   // this group has fewer rows than the launch grid
    const float* __restrict__ A1 = jb.A1;
    const float* __restrict__ W2 = jb.W2;
    const float* __restrict__ b2 = jb.b2;
    const int32_t* __restrict__ rows = jb.rows;
    float* __restrict__ DY = jb.DY;
    float* __restrict__ dZ1 = jb.dZ1;
#define PP_STAMP(k) do { if (dbg && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x == 0) dbg[(k)] = clock64(); } while (0)
    PP_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_out = 3 * K;
    const int hid4 = (hid + 3) & ~3;
    const int ws = hid | 1;
    const int wtot = n_out * ws, wtot4 = (wtot + 3) & ~3;
    float* const w2s = tail_lds;
    float* const a1s = tail_lds + wtot4 + wave * hid4;
    float* const ys = tail_lds + wtot4 + 4 * hid4 + wave * 128;
    float* const dys = ys + 64;
    if (ws == hid && (reinterpret_cast<uintptr_t>(W2) & 15) == 0) {
        // the image IS the tensor: flat 16-byte copy, every load of a thread in flight at once (one round trip)
        const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(W2);
        const int n4 = wtot >> 2;
        for (int base = tid; base < n4; base += 256 * 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[min(base + 256 * u, n4 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (base + 256 * u < n4) *reinterpret_cast<f32x4*>(w2s + 4 * (base + 256 * u)) = v[u];
        }
        for (int i = 4 * n4 + tid; i < wtot; i += 256) w2s[i] = W2[i];
    } else {
        // row stride differs (even hid) or unaligned tensor: branch-free scalar copy, 16 loads in flight per thread;
        // row = floor(i / ws) through an exact float reciprocal (i < 2^15)
        const float inv = 1.0f / (float)ws;
        for (int base = tid; base < wtot; base += 256 * 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = min(base + 256 * u, wtot - 1);
                const int o = (int)(((float)i + 0.5f) * inv);
                const int j = i - o * ws;
                v[u] = W2[o * hid + min(j, hid - 1)];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (base + 256 * u < wtot) w2s[base + 256 * u] = v[u];
        }
    }
    // bias of the three outputs this lane owns as mixture component `lane`
    const bool comp = lane < K;
    const float b2mu = comp ? b2[lane] : 0.0f, b2sd = comp ? b2[K + lane] : 0.0f, b2z = comp ? b2[2 * K + lane] : 0.0f;
    __syncthreads();
    PP_STAMP(1);
    const bool bwd = DY != nullptr;
    float loss_local = 0.f;
    bool bad_any = false;
    const int half = lane >> 5, l31 = lane & 31;
    const int jh = ((hid + 7) >> 3) << 2;                 // k range of a half-wave in the layer-2 products (multiple of 4)
    const int jbeg = half * jh, jend = min(hid, jbeg + jh);
    int jc[NE];                                           // this lane's columns j = lane + 64 e, clamped
#pragma unroll
    for (int e = 0; e < NE; ++e) jc[e] = min(lane + 64 * e, hid - 1);
    const int i0 = (blockIdx.x * 4 + wave) * rows_per_wave;
    for (int t = 0; t < rows_per_wave; ++t) {
        const int i = i0 + t;
        if (i >= n) break;   // wave-uniform
        const int r = rows ? rows[i] : i;
        float a1v[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int j = lane + 64 * e;
            a1v[e] = j < hid ? A1[(int64_t)i * lda1 + j] : 0.0f;   // pad columns of the workspace row are not initialised
            if (j < hid4) a1s[j] = a1v[e];
        }
        wave_lds_sync();
        PP_STAMP(2);
        // second layer: lane (o, half) owns output o over half of the k range: no cross-lane reduction except the
        // final exchange between the half-waves; a1 is read as 16-byte broadcasts, W2 row o with an odd stride
        for (int ob = 0; ob < n_out; ob += 32) {
            const int o = ob + l31;
            const float* __restrict__ wr = w2s + min(o, n_out - 1) * ws;
            float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
            int j = jbeg;
#pragma unroll 4
            for (; j + 3 < jend; j += 4) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(a1s + j);
                p0 += av[0] * wr[j];
                p1 += av[1] * wr[j + 1];
                p2 += av[2] * wr[j + 2];
                p3 += av[3] * wr[j + 3];
            }
            for (; j < jend; ++j) p0 += a1s[j] * wr[j];
            float acc = (p0 + p1) + (p2 + p3);
            acc += __shfl_xor(acc, 32, 64);
            if (half == 0 && o < n_out) ys[o] = acc;
        }
        wave_lds_sync();
        const float ymu = comp ? ys[lane] + b2mu : 0.0f;
        const float ysd = comp ? ys[K + lane] + b2sd : 0.0f;
        const float yz = comp ? ys[2 * K + lane] + b2z : -INFINITY;
        PP_STAMP(3);
        // one mixture component per lane
        const float v = value[r], pa = prior[2 * r], pb = prior[2 * r + 1];
        const float zmax = wave_max(yz);
        const float e = comp ? expf(yz - zmax) : 0.0f;
        const float pi = e / wave_sum(e);
        const float ps = wave_sum(pi);
        const float p = pi / ps;
        float mu, sd, sm = 0.f, ss = 0.f, rng = pb - pa;
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
            Z = std_cdf(beta) - std_cdf(alpha);
            const bool inside = v >= pa && v <= pb;
            cl = (inside ? 0.0f : -INFINITY) + (-0.5f * tt * tt - kHalfLog2Pi) - logf(sd * Z);
        }
        const float a = comp ? logf(fminf(fmaxf(p, kFp32Eps), 1.0f - kFp32Eps)) + cl : -INFINITY;
        const float amax = wave_max(a);
        float lp = amax;
        if (amax > -INFINITY) lp = amax + logf(wave_sum(comp ? expf(a - amax) : 0.0f));
        // NaN in any component poisons the result like the reference's logsumexp would
        if (wave_sum((comp && a != a) ? 1.0f : 0.0f) > 0.0f) lp = NAN;
        PP_STAMP(4);
        if (lp_out && lane == 0) lp_out[r] = lp;
        const bool rescued = (lp == -INFINITY);
        const bool bad = !rescued && !isfinite(lp);
        bad_any |= bad;
        if (lane == 0) loss_local += rescued ? -kLogEps : -lp;
        if (!bwd) continue;
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
        const bool live = !(rescued || bad);
        {
            const float resp = (comp && live) ? expf(a - lp) : 0.0f;
            const bool in = (p >= kFp32Eps) && (p <= 1.0f - kFp32Eps);
            float dp = (comp && in) ? resp / p : 0.0f;
            const float dpp = wave_sum(dp * p);
            dp = comp ? (dp - dpp) / ps : 0.0f;
            const float dpipi = wave_sum(dp * pi);
            if (comp && live) {
                if (KIND == 0) {
                    d0 = grad_scale * resp * tt / sd * pb;
                    d1 = grad_scale * resp * (tt * tt - 1.0f);
                } else {
                    const float fa = std_pdf(alpha), fb = std_pdf(beta);
                    const float dmu = resp * (tt / sd - (fa - fb) / (sd * Z));
                    const float dsd = resp * ((tt * tt - 1.0f) / sd - (alpha * fa - beta * fb) / (sd * Z));
                    d0 = grad_scale * dmu * rng * sm * (1.0f - sm);
                    d1 = grad_scale * dsd * (KIND == 2 ? sd : rng * 10.0f * ss * (1.0f - ss));
                }
                d2 = grad_scale * pi * (dp - dpipi);
            }
        }
        if (comp) {
            float* dy = DY + (int64_t)i * lddy;
            dy[lane] = d0; dy[K + lane] = d1; dy[2 * K + lane] = d2;
            dys[lane] = d0; dys[K + lane] = d1; dys[2 * K + lane] = d2;
        }
        wave_lds_sync();
        PP_STAMP(5);
        // dz1_j = [a1_j > 0] * sum_o dy_o W2[o][j]: lane owns columns j = lane + 64 e (consecutive lanes, consecutive
        // banks), dy_o is an LDS broadcast
        float dz[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) dz[e] = 0.0f;
#pragma unroll 2
        for (int o = 0; o < n_out; ++o) {
            const float d = dys[o];
            const float* __restrict__ wr = w2s + o * ws;
#pragma unroll
            for (int e = 0; e < NE; ++e) dz[e] += d * wr[jc[e]];
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int j = lane + 64 * e;
            if (j < hid4) dZ1[(int64_t)i * lddz + j] = (j < hid && a1v[e] > 0.0f) ? dz[e] : 0.0f;
        }
        wave_lds_sync();   // a1s / ys / dys are rewritten by the next row
    }
    PP_STAMP(6);
    // loss: one atomic per wave, spread over 64 accumulator slots on 64 different 128-byte lines (same-address float
    // atomics serialise at ~40 ns each in L2 - a thousand waves on one word cost ~40 us - and atomics to different
    // words of ONE line still queue in that line's L2 channel). loss_finalize sums the slots.
    if (lane == 0 && loss_acc && loss_local != 0.0f) atomicAdd(loss_acc + 32 * ((blockIdx.x * 4 + wave + 7 * blockIdx.y) & 63), loss_local);
    if (bad_any && nonfinite && lane == 0) atomicOr(nonfinite, 1);
    // The bias gradients db1 = colsum(dz1), db2 = colsum(dy) are NOT accumulated here: ~1000 waves adding to the
    // same few hundred addresses serialise in L2 (measured: +40 us); the caller runs the 16-way colsum kernel instead.
}

static size_t head_tail_lds_floats(int hid, int n_out) {
    const int hid4 = (hid + 3) & ~3;
    return (size_t)((n_out * (hid | 1) + 3) & ~3) + 4 * (size_t)hid4 + 4 * 128;
}

bool head_tail_supported(int kind, int hid, int n_out) {
    static const bool disabled = false;
    if (disabled) return false;
    if (kind != PP_HEAD_NORMAL_MIXTURE && kind != PP_HEAD_TRUNCNORMAL_MIXTURE && kind != PP_HEAD_POISSON_TN_MIXTURE) return false;
    if (n_out % 3 != 0 || n_out / 3 > MAXK || n_out / 3 < 1) return false;
    return hid >= 1 && hid <= 1024 && head_tail_lds_floats(hid, n_out) <= TAIL_LDS_FLOATS;
}

long long* g_timeline = nullptr;   // debug: per-phase clock64() stamps of workgroups 0 and 100 (pp_debug_timeline)

template <int KIND, int NE>
static int head_tail_launch_ne(dim3 grid, size_t lds, hipStream_t st, const TailJobs& jobs, int64_t lda1, int hid, int K,
                               const float* value, const float* prior, int rpw, float gs, float* lp_out, int64_t lddy,
                               int64_t lddz, float* loss_acc, int32_t* nonfinite) {
    static thread_local bool configured = false;   // > 64 KB of dynamic LDS needs the opt-in once per kernel
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)head_tail_kernel<KIND, NE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(TAIL_LDS_FLOATS * sizeof(float)));
        if (e != hipSuccess) {
            set_error("head_tail: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        configured = true;
    }
    hipLaunchKernelGGL((head_tail_kernel<KIND, NE>), grid, dim3(256), lds, st, jobs, lda1, hid, K, value, prior, rpw, gs, lp_out,
                       lddy, lddz, loss_acc, nonfinite, g_timeline);
    return 0;
}

template <int KIND>
static int head_tail_launch(int ne, dim3 grid, size_t lds, hipStream_t st, const TailJobs& jobs, int64_t lda1, int hid, int K,
                            const float* value, const float* prior, int rpw, float gs, float* lp_out, int64_t lddy,
                            int64_t lddz, float* loss_acc, int32_t* nonfinite) {
#define PP_TAIL(N) return head_tail_launch_ne<KIND, N>(grid, lds, st, jobs, lda1, hid, K, value, prior, rpw, gs, lp_out, lddy, \
                                                       lddz, loss_acc, nonfinite)
    if (ne <= 2) PP_TAIL(2);
    if (ne <= 4) PP_TAIL(4);
    if (ne <= 6) PP_TAIL(6);
    if (ne <= 8) PP_TAIL(8);
    if (ne <= 12) PP_TAIL(12);
    PP_TAIL(16);
#undef PP_TAIL
}

// Address groups with the same head kind and shape (kind, hid, n_out) share launches (TAIL_MAX_JOBS per launch).
int head_tail_multi(int kind, const TailJob* jobs, int count, int64_t lda1, int hid, int n_out, const float* value,
                    const float* prior, float grad_scale, float* lp_out, int64_t lddy, int64_t lddz, float* loss_acc,
                    int32_t* nonfinite, hipStream_t st) {
    PP_CHECK_ARG(head_tail_supported(kind, hid, n_out), "head_tail: unsupported head shape");
    PP_CHECK_ARG(lda1 % 4 == 0 && lddz % 4 == 0, "head_tail: leading dimensions must be multiples of 4");
    const int ne = (hid + 63) / 64;   // columns per lane
    const size_t lds = head_tail_lds_floats(hid, n_out) * sizeof(float);
    int i = 0;
    while (i < count) {
        TailJobs pack;
        int nj = 0, max_n = 0;
        for (; i < count && nj < TAIL_MAX_JOBS; ++i) {
            if (jobs[i].n <= 0) continue;
            PP_CHECK_ARG(jobs[i].A1 && jobs[i].W2 && jobs[i].b2 && (!jobs[i].DY || jobs[i].dZ1), "head_tail: bad job");
            pack.j[nj++] = jobs[i];
            max_n = std::max(max_n, jobs[i].n);
        }
        if (nj == 0) continue;
        // ~256 workgroups for the largest group: W2 staging (n_out*hid floats) is amortised over rows_per_wave rows
        const int rpw = std::max((max_n + 4 * 256 - 1) / (4 * 256), 1);
        dim3 grid(cdiv(max_n, 4 * rpw), nj);
        if (kind == PP_HEAD_NORMAL_MIXTURE)
            PP_TRY(head_tail_launch<0>(ne, grid, lds, st, pack, lda1, hid, n_out / 3, value, prior, rpw, grad_scale, lp_out, lddy,
                                       lddz, loss_acc, nonfinite));
        else if (kind == PP_HEAD_TRUNCNORMAL_MIXTURE)
            PP_TRY(head_tail_launch<1>(ne, grid, lds, st, pack, lda1, hid, n_out / 3, value, prior, rpw, grad_scale, lp_out, lddy,
                                       lddz, loss_acc, nonfinite));
        else
            PP_TRY(head_tail_launch<2>(ne, grid, lds, st, pack, lda1, hid, n_out / 3, value, prior, rpw, grad_scale, lp_out, lddy,
                                       lddz, loss_acc, nonfinite));
        PP_LAUNCH_CHECK("head_tail");
    }
    return 0;
}

// loss = acc / B, status = non-finite flag
__global__ void loss_finalize_kernel(const float* __restrict__ acc, const int32_t* __restrict__ flag, float inv_b,
                                     float* __restrict__ loss_out, int32_t* __restrict__ status_out) {
    float tot = 0.0f;
    for (int k = 0; k < 64; ++k) tot += acc[32 * k];   // accumulator slots (128-byte stride)
    const float l = tot * inv_b;
    loss_out[0] = l;
    if (status_out) status_out[0] = (flag[0] != 0 || !isfinite(l)) ? 1 : 0;
}

// Deterministic loss (PP_DETERMINISTIC): the head kernels only write the per-row proposal log_prob; ONE workgroup sums
// the rows in a fixed order, with the reference's -inf -> log(1e-8) rescue (inference_network_lstm.py:207-213) and the
// non-finite check (:214-217).
__global__ __launch_bounds__(256) void loss_rows_kernel(const float* __restrict__ lp, int n_rows, float inv_b,
                                                        const int32_t* __restrict__ flag, float* __restrict__ loss_out,
                                                        int32_t* __restrict__ status_out) {
    __shared__ double part[256];
    __shared__ int bad[256];
    double acc = 0.0;
    int b = 0;
    for (int r = threadIdx.x; r < n_rows; r += 256) {
        float l = lp[r];
        if (l == -INFINITY) l = kLogEps;
        if (!isfinite(l)) b = 1;
        acc -= (double)l;
    }
    part[threadIdx.x] = acc;
    bad[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            part[threadIdx.x] += part[threadIdx.x + s];
            bad[threadIdx.x] |= bad[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float l = (float)(part[0] * (double)inv_b);
        loss_out[0] = l;
        if (status_out) status_out[0] = (bad[0] || (flag && flag[0] != 0) || !isfinite(l)) ? 1 : 0;
    }
}

int loss_from_rows(const float* lp, int n_rows, int n_traces, const int32_t* flag, float* loss_out, int32_t* status_out,
                   hipStream_t st) {
    hipLaunchKernelGGL(loss_rows_kernel, dim3(1), dim3(256), 0, st, lp, n_rows, 1.0f / (float)n_traces, flag, loss_out,
                       status_out);
    PP_LAUNCH_CHECK("loss_from_rows");
    return 0;
}

int loss_finalize(const float* acc, const int32_t* flag, int n_traces, float* loss_out, int32_t* status_out,
                  hipStream_t st) {
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, st, acc, flag, 1.0f / (float)n_traces, loss_out,
                       status_out);
    PP_LAUNCH_CHECK("loss_finalize");
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Adam over the flat parameter buffer. Tensors are padded to 1024-float chunks; a chunk -> tensor map gives each
// workgroup its tensor's presence flag and bias corrections (per-tensor step counts, as torch.optim.Adam keeps).
// ------------------------------------------------------------------------------------------------------
// One launch: the workgroup of a 1024-float chunk looks up its tensor, derives the bias corrections from the
// tensor's step count (thread 0, double precision like torch.optim.Adam's Python floats), updates its chunk and -
// as the LAST workgroup of that tensor to finish - advances the step count. Every workgroup reads the count before it
// signals arrival, so all chunks of a tensor see the same step (a separate "prepare" launch did this before: 4.7 us of
// a 170 us step). Arrival is counted in two levels, at most 32 workgroups per counter: same-address atomics
// serialise at ~40 ns each and W_hh alone has 1024 chunks (a single counter per tensor made this kernel 68 us).
// Optionally clears the gradient chunk it consumed (zero_grad of the NEXT step: saves a 6.5 MB memset launch).
//
// scratch, PP_ADAM_SCRATCH = 1056 ints per tensor: 32 sub-counters (chunk index mod 32) on separate 128-byte lines
// (atomics to one line serialise in its L2 channel whatever the address), then the top counter, first chunk + 1
// (0 = not cached yet) and the number of chunks. All counters return to zero after every call.
constexpr int ADAM_TOP = 1024, ADAM_FIRST = 1025, ADAM_CHUNKS = 1026;   // after 32 sub-counters, one per 128-byte line
// ADAM_SEEN: set once any chunk of the tensor had a non-zero gradient. While it is 0 both moments of the WHOLE tensor are
// still zero (they start at zero and only a non-zero gradient moves them), so a chunk whose gradient is all zero is a
// no-op of Adam (m = v = 0 -> update 0; no weight decay) and touches nothing but its gradient chunk: the recurrent
// weights of a single-statement program (GaussianUnknownMean: dL/dW_hh = 0 forever) are 63 % of the parameters and cost
// 1/8 of their Adam traffic this way. The host sets the flag when it writes moments itself (checkpoint load).
constexpr int ADAM_SEEN = 1027;

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ P, float* __restrict__ Gr, float* __restrict__ M,
                                                   float* __restrict__ V, const int32_t* __restrict__ chunk_tensor,
                                                   int n_chunks, const float* __restrict__ active,
                                                   int32_t* __restrict__ tensor_step, int32_t* __restrict__ scratch, float lr,
                                                   float beta1, float beta2, float eps, float wd, float gscale,
                                                   int zero_grads, const int32_t* __restrict__ skip) {
    __shared__ float s_corr[2];
    const int b = blockIdx.x;
    // the loss of this step was not finite: the reference skips the batch (inference_network_lstm.py:216-217, no
    // optimizer step); checked on the device so that the host does not have to synchronise every iteration
    const bool skipped = skip && skip[0] != 0;
    const int t = chunk_tensor[b];
    if (t < 0 || !(active[t] > 0.0f)) return;   // workgroup-uniform; every chunk of a tensor takes the same branch
    if (skipped) {   // no update, no step count; the (non-finite) gradients are still cleared for the next step
        if (zero_grads) *reinterpret_cast<f32x4*>(Gr + (int64_t)b * 1024 + threadIdx.x * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    int32_t* const sc = scratch + (int64_t)t * PP_ADAM_SCRATCH;
    // the chunk's loads are issued before thread 0 walks its dependent chain (step count, corrections)
    const int64_t o = (int64_t)b * 1024 + threadIdx.x * 4;
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(Gr + o);
    // a tensor that never had a non-zero gradient: wait for the gradient chunk first; all zero -> nothing else is read.
    // ONE thread reads the flag (other workgroups of the tensor may be setting it right now: every thread reading it for
    // itself could split the workgroup around the barrier below)
    __shared__ int s_seen;
    if (threadIdx.x == 0) s_seen = __atomic_load_n(sc + ADAM_SEEN, __ATOMIC_RELAXED);
    __syncthreads();
    const bool seen = s_seen != 0;   // workgroup-uniform
    bool idle = false;
    if (!seen && wd == 0.0f) {
        const bool zero = g0[0] == 0.0f && g0[1] == 0.0f && g0[2] == 0.0f && g0[3] == 0.0f;
        idle = __syncthreads_and(zero ? 1 : 0) != 0;
        if (!idle && threadIdx.x == 0) __atomic_store_n(sc + ADAM_SEEN, 1, __ATOMIC_RELAXED);
    } else if (!seen && threadIdx.x == 0) {
        __atomic_store_n(sc + ADAM_SEEN, 1, __ATOMIC_RELAXED);
    }
    f32x4 p = f32x4{0.f, 0.f, 0.f, 0.f}, m = p, v = p;
    if (!idle) {
        p = *reinterpret_cast<f32x4*>(P + o);
        m = *reinterpret_cast<f32x4*>(M + o);
        v = *reinterpret_cast<f32x4*>(V + o);
    }
    int step_old = 0, first = 0, chunks = 0;
    if (threadIdx.x == 0) {
        first = __atomic_load_n(sc + ADAM_FIRST, __ATOMIC_RELAXED) - 1;
        chunks = __atomic_load_n(sc + ADAM_CHUNKS, __ATOMIC_RELAXED);
        if (first < 0) {
            // chunks of one tensor are contiguous and the ids ascend: two binary searches (~20 dependent loads), done by
            // every workgroup of the FIRST call only; the last one to arrive caches the run
            int lo = 0, hi = b;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (chunk_tensor[mid] < t) lo = mid + 1; else hi = mid;
            }
            int lo2 = b, hi2 = n_chunks;
            while (lo2 < hi2) {
                const int mid = (lo2 + hi2) >> 1;
                if (chunk_tensor[mid] <= t) lo2 = mid + 1; else hi2 = mid;
            }
            first = lo;
            chunks = lo2 - lo;
        }
        step_old = __atomic_load_n(tensor_step + t, __ATOMIC_RELAXED);
        const int step = step_old + 1;
        // beta^step by repeated squaring (integer exponent): ~40 double multiplies instead of two generic pow() calls,
        // which cost ~6 us on the single active lane every workgroup waits for
        double p1 = 1.0, p2 = 1.0, q1 = (double)beta1, q2 = (double)beta2;
        for (int e = step; e > 0; e >>= 1) {
            if (e & 1) { p1 *= q1; p2 *= q2; }
            q1 *= q1;
            q2 *= q2;
        }
        const double bc1 = 1.0 - p1, bc2 = 1.0 - p2;
        s_corr[0] = (float)((double)lr / bc1);
        s_corr[1] = (float)(1.0 / sqrt(bc2));
    }
    __syncthreads();
    const float step_size = s_corr[0], inv_sqrt_bc2 = s_corr[1];
    // Arrival is signalled NOW (every workgroup has read the old step count by the time it gets here) and the returned
    // ticket is only looked at after the update below, so the atomic's round trip overlaps the chunk's arithmetic and
    // stores. No fence: the step count was CONSUMED (bias corrections) before this point, so that load has completed;
    // a device-scope __threadfence() writes back / invalidates L2 in every workgroup (measured: 9 -> 66 us).
    const int k = (b - first) & 31;                         // sub-counter of this chunk
    const int quota = (chunks - k + 31) >> 5;               // chunks of the tensor that share it
    int ticket = -1;
    if (threadIdx.x == 0) ticket = atomicAdd(sc + 32 * k, 1);
    if (!idle) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float g = g0[e] * gscale;
            if (wd != 0.0f) g += wd * p[e];
            m[e] = beta1 * m[e] + (1.0f - beta1) * g;
            v[e] = beta2 * v[e] + (1.0f - beta2) * g * g;
            const float denom = sqrtf(v[e]) * inv_sqrt_bc2 + eps;
            p[e] -= step_size * (m[e] / denom);
        }
        *reinterpret_cast<f32x4*>(P + o) = p;
        *reinterpret_cast<f32x4*>(M + o) = m;
        *reinterpret_cast<f32x4*>(V + o) = v;
        if (zero_grads) *reinterpret_cast<f32x4*>(Gr + o) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (threadIdx.x == 0 && ticket == quota - 1) {
        __atomic_store_n(sc + 32 * k, 0, __ATOMIC_RELAXED);
        if (atomicAdd(sc + ADAM_TOP, 1) == min(chunks, 32) - 1) {   // last chunk of the tensor
            __atomic_store_n(sc + ADAM_TOP, 0, __ATOMIC_RELAXED);
            __atomic_store_n(sc + ADAM_FIRST, first + 1, __ATOMIC_RELAXED);
            __atomic_store_n(sc + ADAM_CHUNKS, chunks, __ATOMIC_RELAXED);
            __atomic_store_n(tensor_step + t, step_old + 1, __ATOMIC_RELAXED);
        }
    }
}

int adam_step(float* params, float* grads, float* m, float* v, int64_t n_params, const int32_t* chunk_tensor,
              const float* active, int32_t* tensor_step, int32_t* scratch, int n_tensors, float lr, float beta1, float beta2,
              float eps, float wd, float gscale, int flags, const int32_t* skip, hipStream_t st) {
    PP_CHECK_ARG(params && grads && m && v && chunk_tensor && active && tensor_step && scratch, "pp_adam_step: null pointer");
    PP_CHECK_ARG(n_params % 1024 == 0, "pp_adam_step: n_params must be a multiple of 1024 (padded tensors)");
    if (n_tensors <= 0 || n_params == 0) return 0;
    const int n_chunks = (int)(n_params / 1024);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)n_chunks), dim3(256), 0, st, params, grads, m, v, chunk_tensor, n_chunks,
                       active, tensor_step, scratch, lr, beta1, beta2, eps, wd, gscale, (flags & PP_ADAM_ZERO_GRADS) ? 1 : 0, skip);
    PP_LAUNCH_CHECK("pp_adam_step");
    return 0;
}

}  // namespace pp
