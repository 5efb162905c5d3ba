// Small reduction jobs of the backward pass that ride in the workgroups BEHIND the tiles of the grouped weight-gradient
// launch (or run as their own launch): column sums (bias / embedding-table gradients), the parameter gradients that
// follow from the per-address column sums of dG (gather.hpp, AddrBias), the loss finalisation. Every job reads buffers
// that are complete before the launch starts and writes gradient words nothing else in the launch touches, so the
// workgroups need no ordering among themselves. Written for any workgroup size that is a multiple of 64 (the host
// tile kernels run 256 or 512 threads).
#pragma once
#include "common.hpp"

namespace pp {

// out[c * out_stride] += sum_i X[ix(i)*ldx + c] * (wgt ? wgt[i * ldw] : 1)
struct ColsumJob {
    const float* X; int64_t ldx; const int32_t* idx; int n_rows, n_cols; float* out; float* out2;
    const float* wgt; int64_t ldw; int out_stride;   // optional per-row weight; out_stride 0 = dense
};

constexpr int AUX_MAX_COLSUM = 48;   // (48 x 72 bytes of kernel arguments; with GroupedParams ~7.5 KB per launch)
constexpr int AUX_COLSUM_ROWS = 64;    // rows per column-sum workgroup
constexpr int AUX_OUTER_ROWS = 16;     // rows of dW_ih per workgroup of the outer-product job
constexpr int AUX_TABLE_ROWS = 128;    // rows of W_ih per workgroup of the table-gradient job
constexpr int AUX_DBSUM_ROWS = 256;

struct AuxJobs {
    int n_blocks;                        // workgroups of all jobs (0: nothing to do)
    int n_colsum;
    int cs_first[AUX_MAX_COLSUM + 1];    // first workgroup of column-sum job j
    ColsumJob cs[AUX_MAX_COLSUM];
    int outer_first, table_first, dbsum_first, fin_block;   // first workgroup of each derived job (-1: absent)
    // gradients that follow from gsum[a][0 | 1][N] = column sums of dG over the rows whose current | previous statement
    // has address a (gather.hpp):
    //   outer:  dW_ih[n, c2 + k] += sum_a gsum[a][1][n] e_a[k],  dW_ih[n, c4 + k] += sum_a gsum[a][0][n] e_a[k]   (e_a = [d_a ; a_a])
    //   table:  d e_a[k] += sum_n W_ih[n, c2 + k] gsum[a][1][n] + W_ih[n, c4 + k] gsum[a][0][n]
    //   dbsum:  db_ih[n] += sum_a gsum[a][0][n]   (and db_hh)
    const float* gsum;
    const float* W;        // W_ih [N][ldw]
    float* dW;             // its gradient, same layout
    int64_t ldw;
    const float* params;
    float* grads;
    const int64_t* at;     // device address table
    float* db_ih; float* db_hh;
    int N, c2, c4, nd, ne, n_addr;
    uint32_t present[32];
    int all_present;
    LossFinalize fin;      // fin.acc == nullptr: absent
};

__device__ __forceinline__ bool aux_present(const AuxJobs& j, int a) {
    return j.all_present || ((j.present[(a >> 5) & 31] >> (a & 31)) & 1u);
}

// Wide jobs (the per-address column sums of dG: 4H columns) take 256 columns per workgroup with 16-byte loads: a quarter of
// the workgroups, four times the bytes in flight per thread (2864 workgroups of 64 x 64 floats made the column-sum launch of a
// ragged 12-address step 27 us for 43 MB).
__host__ __device__ __forceinline__ bool aux_colsum_wide(const ColsumJob& j) {
    return j.n_cols >= 256 && (j.n_cols & 3) == 0 && (j.ldx & 3) == 0 && !j.wgt && j.out_stride <= 1 &&
           (reinterpret_cast<uintptr_t>(j.X) & 15) == 0;
}
__host__ __device__ __forceinline__ int aux_colsum_blocks(const ColsumJob& j) {
    const int cw = aux_colsum_wide(j) ? 256 : 64;
    return ((j.n_cols + cw - 1) / cw) * ((j.n_rows + AUX_COLSUM_ROWS - 1) / AUX_COLSUM_ROWS);
}

__device__ __forceinline__ void aux_colsum_wide_block(const ColsumJob& jb, int local, float* lds) {
    const int nt = blockDim.x, tid = threadIdx.x;
    const int nrl = nt >> 6;                        // row lanes
    const int cl = tid & 63, rl = tid >> 6;
    const int ncc = (jb.n_cols + 255) / 256;
    const int bx = local % ncc, by = local / ncc;
    const int col = bx * 256 + 4 * cl;              // this thread's four columns
    const int r0 = by * AUX_COLSUM_ROWS;
    const float* __restrict__ X = jb.X;
    const int32_t* __restrict__ idx = jb.idx;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    if (col < jb.n_cols) {
        // (four 16-byte loads in flight per trip: this code shares its kernel with the MFMA tiles, whose two workgroups per
        // CU need the kernel to stay within 128 VGPRs - eight in flight pushed it to 169 and the weight-gradient launch of the
        // config-2 step from 23 to 80 us)
#pragma unroll 1
        for (int q0 = 0; q0 < 16; q0 += 4) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + u;
                const int i = r0 + rl + nrl * q;
                v[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                if (q * nrl < AUX_COLSUM_ROWS && i < jb.n_rows) {
                    const int64_t r = idx ? (int64_t)idx[i] : (int64_t)i;
                    v[u] = *reinterpret_cast<const f32x4*>(X + r * jb.ldx + col);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u];
        }
    }
    // row lanes meet in LDS: [nrl][256]
#pragma unroll
    for (int e = 0; e < 4; ++e) lds[rl * 256 + 4 * cl + e] = acc[e];
    __syncthreads();
    for (int c = tid; c < 256; c += nt) {
        const int gc = bx * 256 + c;
        if (gc < jb.n_cols) {
            float sum = 0.0f;
            for (int q = 0; q < nrl; ++q) sum += lds[q * 256 + c];
            atomicAdd(jb.out + gc, sum);
            if (jb.out2) atomicAdd(jb.out2 + gc, sum);
        }
    }
}

__device__ __forceinline__ void aux_colsum_block(const ColsumJob& jbin, int local, float* lds) {
    const ColsumJob jb = jbin;   // local copy: fields read through the kernel-argument table are re-loaded at every use
    if (aux_colsum_wide(jb)) {   // workgroup-uniform
        aux_colsum_wide_block(jb, local, lds);
        return;
    }
    const int nt = blockDim.x, tid = threadIdx.x;
    const int nrl = nt >> 6;                        // row lanes
    const int cl = tid & 63, rl = tid >> 6;
    const int ncc = (jb.n_cols + 63) / 64;
    const int bx = local % ncc, by = local / ncc;
    const int col = bx * 64 + cl;
    const int r0 = by * AUX_COLSUM_ROWS;
    const float* __restrict__ X = jb.X;
    const int32_t* __restrict__ idx = jb.idx;
    const float* __restrict__ wgt = jb.wgt;
    float acc = 0.0f;
    if (col < jb.n_cols) {
        float v[16];   // all loads issued before the adds (independent addresses: the latencies overlap)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = r0 + rl + nrl * q;
            v[q] = 0.0f;
            if (q * nrl < AUX_COLSUM_ROWS && i < jb.n_rows) {
                const int64_t r = idx ? (int64_t)idx[i] : (int64_t)i;
                v[q] = X[r * jb.ldx + col];
                if (wgt) v[q] *= wgt[(int64_t)i * jb.ldw];
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += v[q];
    }
    lds[rl * 64 + cl] = acc;
    __syncthreads();
    if (rl == 0 && col < jb.n_cols) {
        float s = 0.0f;
        for (int q = 0; q < nrl; ++q) s += lds[q * 64 + cl];
        atomicAdd(jb.out + (int64_t)col * (jb.out_stride ? jb.out_stride : 1), s);
        if (jb.out2) atomicAdd(jb.out2 + col, s);
    }
}

__device__ __forceinline__ void aux_outer_block(const AuxJobs& j, int local) {
    const int tid = threadIdx.x;
    if (tid >= 256) return;
    const int n = local * AUX_OUTER_ROWS + (tid >> 4), kq = tid & 15;
    if (n >= j.N) return;
    constexpr int MAXJ = 16;   // 2 ne <= 256 columns
    float acc[MAXJ];
#pragma unroll
    for (int q = 0; q < MAXJ; ++q) acc[q] = 0.0f;
    const int ne = j.ne, nd = j.nd, N = j.N, n_addr = j.n_addr, ncol = 2 * ne;
    const float* const gsum = j.gsum;
    const float* const params = j.params;
    const int64_t* const at = j.at;
    for (int a = 0; a < n_addr; ++a) {
        if (!aux_present(j, a)) continue;
        const float* gs = gsum + (int64_t)a * 2 * N;
        const float g0 = gs[n], g1 = gs[N + n];
        const float* dt = params + at[a * PP_ADDR_TABLE_COLS + PP_AT_DTYPE_EMB];
        const float* ad = params + at[a * PP_ADDR_TABLE_COLS + PP_AT_ADDR_EMB];
#pragma unroll
        for (int q = 0; q < MAXJ; ++q) {
            const int col = kq + 16 * q;
            if (col < ncol) {
                const int k = col < ne ? col : col - ne;
                const float e = k < nd ? dt[k] : ad[k - nd];
                acc[q] += (col < ne ? g1 : g0) * e;
            }
        }
    }
    // columns [c2, c2 + ne) and [c4, c4 + ne) are adjacent (c4 == c2 + ne): one run of 2 ne columns, single writer
    float* row = j.dW + (int64_t)n * j.ldw + j.c2;
#pragma unroll
    for (int q = 0; q < MAXJ; ++q) {
        const int col = kq + 16 * q;
        if (col < ncol) row[col] += acc[q];
    }
}

__device__ __forceinline__ void aux_table_block(const AuxJobs& j, int local, float* lds) {
    const int nt = blockDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int nchunk = (j.N + AUX_TABLE_ROWS - 1) / AUX_TABLE_ROWS;
    const int a = local / nchunk, chunk = local % nchunk;
    if (a >= j.n_addr || !aux_present(j, a)) return;   // workgroup-uniform
    const int ncol = 2 * j.ne;
    for (int k = tid; k < ncol; k += nt) lds[k] = 0.0f;
    __syncthreads();
    const float* gs = j.gsum + (int64_t)a * 2 * j.N;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // columns lane, lane + 64, ... of the run [c2, c2 + 2 ne)
    const int n1 = min(j.N, (chunk + 1) * AUX_TABLE_ROWS);
    // eight rows per trip, every load of the trip issued before the first multiply (a row at a time would pay one memory
    // round trip per row: measured 40 us for this job)
    for (int nb = chunk * AUX_TABLE_ROWS + wave; nb < n1; nb += 8 * nw) {
        float g0[8], g1[8], wv[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int n = nb + u * nw;
            const int nn = min(n, n1 - 1);
            g0[u] = gs[nn];
            g1[u] = gs[j.N + nn];
            if (n >= n1) { g0[u] = 0.0f; g1[u] = 0.0f; }
            const float* w = j.W + (int64_t)nn * j.ldw + j.c2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = lane + 64 * q;
                wv[u][q] = col < ncol ? w[col] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += wv[u][q] * ((lane + 64 * q) < j.ne ? g1[u] : g0[u]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = lane + 64 * q;
        if (col < ncol) atomicAdd(&lds[col], acc[q]);
    }
    __syncthreads();
    float* dt = j.grads + j.at[a * PP_ADDR_TABLE_COLS + PP_AT_DTYPE_EMB];
    float* ad = j.grads + j.at[a * PP_ADDR_TABLE_COLS + PP_AT_ADDR_EMB];
    for (int col = tid; col < ncol; col += nt) {
        const int k = col < j.ne ? col : col - j.ne;
        atomicAdd(k < j.nd ? dt + k : ad + (k - j.nd), lds[col]);   // (addresses of one distribution type share d_a)
    }
}

__device__ __forceinline__ void aux_dbsum_block(const AuxJobs& j, int local) {
    const int tid = threadIdx.x;
    if (tid >= 256) return;
    const int n = local * AUX_DBSUM_ROWS + tid;
    if (n >= j.N) return;
    float s = 0.0f;
    for (int a = 0; a < j.n_addr; ++a)
        if (aux_present(j, a)) s += j.gsum[(int64_t)a * 2 * j.N + n];
    j.db_ih[n] += s;
    j.db_hh[n] += s;
}

// workgroup `b` (0 <= b < n_blocks) of the job list; lds: >= 2048 floats
__device__ __forceinline__ void aux_job_run(const AuxJobs& j, int b, float* lds) {
    if (b < j.cs_first[j.n_colsum]) {
        int q = 0;
        while (q + 1 < j.n_colsum && b >= j.cs_first[q + 1]) ++q;   // workgroup-uniform
        aux_colsum_block(j.cs[q], b - j.cs_first[q], lds);
        return;
    }
    if (j.fin_block >= 0 && b == j.fin_block) {
        if (threadIdx.x == 0) loss_finalize_inline(j.fin);
        return;
    }
    if (j.dbsum_first >= 0 && b >= j.dbsum_first) { aux_dbsum_block(j, b - j.dbsum_first); return; }
    if (j.table_first >= 0 && b >= j.table_first) { aux_table_block(j, b - j.table_first, lds); return; }
    if (j.outer_first >= 0 && b >= j.outer_first) { aux_outer_block(j, b - j.outer_first); return; }
}

// host: lay the jobs out (colsum jobs first, then outer, table, dbsum, finalize)
static inline void aux_layout(AuxJobs& j, bool derived) {
    int b = 0;
    j.cs_first[0] = 0;
    for (int q = 0; q < j.n_colsum; ++q) {
        b += aux_colsum_blocks(j.cs[q]);
        j.cs_first[q + 1] = b;
    }
    j.outer_first = j.table_first = j.dbsum_first = -1;
    if (derived) {
        j.outer_first = b;
        b += (j.N + AUX_OUTER_ROWS - 1) / AUX_OUTER_ROWS;
        j.table_first = b;
        b += j.n_addr * ((j.N + AUX_TABLE_ROWS - 1) / AUX_TABLE_ROWS);
        if (j.db_ih) {
            j.dbsum_first = b;
            b += (j.N + AUX_DBSUM_ROWS - 1) / AUX_DBSUM_ROWS;
        }
    }
    j.fin_block = -1;
    if (j.fin.acc) j.fin_block = b++;
    j.n_blocks = b;
}

}  // namespace pp
