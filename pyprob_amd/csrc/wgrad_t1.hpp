// Weight gradients of a single-statement batch (wgrad_t1.hip): arguments and the host entry point used by engine.hip.
#pragma once
#include "aux_jobs.hpp"
#include "common.hpp"

namespace pp {

constexpr int WGRAD_T1_MAX = 40;

// dW[m][n] += sum_k A[k][m] B[ix(k)][n], k = rows of the minibatch: both operands are activations stored row = k (the way the
// panel kernel / the backward pass wrote them; ix = an optional row gather: h_{t-1} rows by prev_row, the rows of an address
// group), dW is a row-major weight-gradient tensor (ldc = its row length).
struct WgradT1Prob {
    const float* A; const float* B; float* C;
    const int32_t* bidx; // row gather of B, or nullptr
    int lda, ldb, ldc;
    int M, N;            // extent of this problem (a cut of the tensor: zero blocks are left out or start later, see the host)
    int nt;              // 64-column tiles
    int K, S, ks;        // rows, row splits, rows per split (multiple of 4)
    int first;           // first workgroup of the problem in the launch
    int mtg;             // > 0: XCD-aware tile order (groups of 8 row tiles, see wgrad_t1_kernel); 0: column tiles fastest
};

struct WgradT1Args {
    WgradT1Prob p[WGRAD_T1_MAX];
    int n_prob, n_blocks;     // problems; workgroups of all tiles
    int aux_first;            // workgroups of the reduction jobs in front of the tiles (launches with more than one round of tiles) or behind
    long long* trace;         // debug (pp_debug_wgtrace, mode 1): per workgroup {start, end, problem, split, K loop done} wall-clock ticks
};

// Takes the queued weight-gradient products (queue_wgrad: k-major operands, accumulate) when every one of them fits the
// kernel (k-major operands, at most a row gather on the second one, zero blocks that are whole-K or a prefix of the rows);
// false: use the grouped tile kernels.
bool wgrad_t1_build(const pp_gemm_args* q, const GemmHole* holes, int n, WgradT1Args& out);
int wgrad_t1(const WgradT1Args& a, const AuxJobs* aux, hipStream_t st);

}  // namespace pp
