// Host-side orchestration behind the C ABI: InferenceNetworkLSTM._loss (+ backward) as a chain of HIP kernels over
// the step-major packed trace batch. No allocation, no host synchronisation: every launch goes to the caller's stream.
#include "common.hpp"
#include "gather.hpp"
#include "aux_jobs.hpp"
#include "panel.hpp"
#include "panel16.hpp"
#include "wgrad_t1.hpp"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace pp {

// kernels.hip / gemm_f32.hip
const char* last_error();
int colsum_multi(const ColsumJob* jobs, int count, hipStream_t st, const float* fin_acc = nullptr,
                 const int32_t* fin_flag = nullptr, int fin_traces = 0, float* fin_loss = nullptr,
                 int32_t* fin_status = nullptr);
int colsum_f32(const float* X, int64_t ldx, const int32_t* idx, int n_rows, int n_cols, float* out, float* out2,
               hipStream_t st);
int embedding_rows(const float* E, int64_t lde, const int32_t* trace, int n_rows, int e_obs, float* Hs, int64_t ldh,
                   float* zero_small, int n_small, hipStream_t st);
int sample_embed_bwd(const pp_net* net, const float* params, const float* value, const int32_t* addr,
                     const int32_t* prev_row, int row_begin, int n_rows, const float* dX, int64_t ldx, float* grads,
                     hipStream_t st);
// lstm_tail.hip: the late time steps of a ragged batch in one launch per direction
bool dp_overlap_hull(int64_t* lo, int64_t* hi);          // dp.hip
int dp_bucket0_issue(float* grads_full, hipStream_t st);
int lstm_tail_plan(const int32_t* n_active, int T, int H, int* t0_out, int* teams_out);
void lstm_tail_exchange_bytes(int H, size_t* fwd, size_t* bwd);
int lstm_tail_fwd(float* G, float* C, float* Hs, const float* Whh, const int32_t* row_off_dev, int t0, int T, int H, int teams,
                  void* xch_f, void* xch_b, int32_t* flag, hipStream_t st);
int lstm_tail_bwd(float* G, const float* C, float* dH, float* dC, const float* Whh, const int32_t* row_off_dev, int t0, int T,
                  int H, int teams, void* xch_f, void* xch_b, int32_t* flag, float* db, float* db2, const LossFinalize& fin,
                  hipStream_t st);
int obs_grad(const float* dX, int64_t ldx, const int32_t* row_off_dev, int t_max, int n_traces, int e_obs,
             const float* E, int64_t lde, float* dE, int64_t ldde, hipStream_t st);
int lstm_cell_fwd(float* G, const float* c_prev, float* c, float* h, int n, int H, hipStream_t st, int c_prev_shared = 0);
int lstm_cell_bwd(float* G, const float* c_prev, const float* c, const float* dh, float* dc_carry, int n, int n_next,
                  int H, float* db, float* db2, hipStream_t st, const float* fin_acc = nullptr,
                  const int32_t* fin_flag = nullptr, int fin_traces = 0, float* fin_loss = nullptr,
                  int32_t* fin_status = nullptr, const float* dh_parts = nullptr, int n_parts = 0, int64_t part_stride = 0);
int head_logprob(int kind, const float* y, int64_t ldy, const int32_t* rows, const float* value, const float* prior,
                 int n, int n_out, float grad_scale, float* lp_out, float* dy, float* loss_acc, int32_t* nonfinite,
                 hipStream_t st);
bool head_tail_supported(int kind, int hid, int n_out);
struct TailJob {
    const float* A1; const float* W2; const float* b2; const int32_t* rows; float* DY; float* dZ1; int n;
};
int head_tail_multi(int kind, const TailJob* jobs, int count, int64_t lda1, int hid, int n_out, const float* value,
                    const float* prior, float grad_scale, float* lp_out, int64_t lddy, int64_t lddz, float* loss_acc,
                    int32_t* nonfinite, hipStream_t st);
int loss_finalize(const float* acc, const int32_t* flag, int n_traces, float* loss_out, int32_t* status_out,
                  hipStream_t st);
int loss_from_rows(const float* lp, int n_rows, int n_traces, const int32_t* flag, float* loss_out, int32_t* status_out,
                   hipStream_t st);
int sample_embed_bwd_det(const pp_net* net, const float* params, const float* value, const int32_t* prev_row,
                         const int32_t* nxt_rows, const int32_t* nxt_off, const float* dX, int64_t ldx, float* grads,
                         hipStream_t st);
int adam_step(float* params, float* grads, float* m, float* v, int64_t n_params, const int32_t* chunk_tensor,
              const float* active, int32_t* tensor_step, int32_t* arrived, int n_tensors, float lr, float beta1, float beta2,
              float eps, float wd, float gscale, int flags, const int32_t* skip, hipStream_t st);

extern long long* g_timeline;   // kernels.hip

// obs_embed.hip (obs_embed.hpp: obs_fused_supported, obs_fused_args)
int obs_embed_dgrad_fused(const pp_net* net, const float* P, int n_traces, float* const* obs_h, const float* cat,
                          const float* f1, const float* dX, int64_t ldx, const int32_t* row_off_dev, int t_max,
                          const float* E, float* dE, float* dF1, float* dCat, float* dHo0, int64_t dh_stride,
                          hipStream_t st, int n_split = 1, int64_t split_stride = 0);

static inline int64_t round4(int64_t x) { return (x + 3) & ~int64_t(3); }

// ---- in-stream kernel timing (bench.py roofline leg) ---------------------------------------------------
struct Prof {
    int which = -1;
    std::vector<hipEvent_t> ev0, ev1;
    std::vector<double> flops;
    int used = 0;
    int stride = 1, seen = 0;      // an event pair around every stride-th launch of the class (pp_prof_stride)
    bool cur = false;
};
static Prof g_prof;

void prof_begin(int which, hipStream_t st) {
    if (g_prof.which != which) return;
    g_prof.cur = (g_prof.seen++ % g_prof.stride) == 0 && g_prof.used < (int)g_prof.ev0.size();
    if (g_prof.cur) (void)hipEventRecord(g_prof.ev0[g_prof.used], st);
}
void prof_end(int which, double flops, hipStream_t st) {
    if (g_prof.which == which && g_prof.cur) {
        (void)hipEventRecord(g_prof.ev1[g_prof.used], st);
        g_prof.flops[g_prof.used] = flops;
        g_prof.used++;
        g_prof.cur = false;
    }
}

// ---- workspace carving ------------------------------------------------------------------------------
struct Carver {
    char* base;
    size_t off = 0, cap;
    Carver(void* p, size_t c) : base(static_cast<char*>(p)), cap(c) {}
    template <typename T>
    T* take(int64_t count) {
        off = (off + 255) & ~size_t(255);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += (size_t)std::max<int64_t>(count, 1) * sizeof(T);
        return p;
    }
};

struct Workspace {
    // forward activations
    float* obs_h[PP_MAX_OBS];  // [B, round4(obs_hid)] (first hidden layer; depth 2: the only one)
    float* obs_hl[PP_MAX_OBS][PP_MAX_OBS_DEPTH];   // hidden activations of layer l < depth - 1 (obs_hl[o][0] == obs_h[o])
    float* dObsH2;             // second [B, maxhid4] scratch of the generic-depth backward
    float* cat;                // [B, e4] concatenated per-observable embeddings
    float* f1;                 // [B, e4] hidden layer of the final observe embedding
    float* E;                  // [B, e4] observe embedding
    float* X;                  // [R, i4] LSTM input rows
    float* G;                  // [R, 4H] gate pre-activations -> gates -> dG   (layer 0; layer k: Gl[k] ...)
    float* C;                  // [R, H]
    float* Hs;                 // [R, H] hidden states of the TOP layer (what the proposal heads read)
    float* Gl[PP_MAX_LSTM_DEPTH];   // per layer of nn.LSTM(I, H, depth): Gl[0] == G, Hl[depth - 1] == Hs
    float* Cl[PP_MAX_LSTM_DEPTH];
    float* Hl[PP_MAX_LSTM_DEPTH];
    float* dH2;                // [R, H] gradient into the hidden states of the layer below (depth > 1)
    float* A1;                 // [R, hid4] head hidden activations, group-compact row order
    float* Y;                  // [R, out4] head outputs
    float* DY;                 // [R, out4]
    // backward
    float* dZ1;                // [R, hid4]
    float* dH;                 // [R, H]
    float* dC;                 // [B, H]
    float* dHp;                // [DH_SPLITS][B, H] partial tiles of the recurrent data-gradient product (see ic_loss)
    float* dX;                 // [R, i4]
    float* dE;                 // [B, e4]
    float* dF1;                // [B, e4]
    float* dCat;               // [B, e4]
    float* dObsH;              // n_obs x [B, maxhid4] (observable o at dObsH + o * B * maxhid4)
    float* loss_acc;           // [1]
    int32_t* flag;             // [1]
    void* xch_f;               // granule exchange areas of the LSTM tail kernels: at the START of the workspace (a fixed
    void* xch_b;               // place whatever the batch size), zero when the workspace was allocated (see the header)
    float* lp_rows;            // [R] per-row proposal log_prob (deterministic mode: the loss is reduced from it)
    float* AB;                 // [n_addr][2][4H] per-address bias vectors of the LSTM input (gather.hpp, AddrBias)
    float* gsum;               // [n_addr][2][4H] column sums of dG per address group (current | previous statement)
    float* WihT;               // [e_obs][4H] k-major copy of W_ih[:, :e_obs] (panel.hpp), rewritten every step
    float* W1T;                // [H][64 ceil(maxhid / 64)] k-major copy of the present address's first head layer
    unsigned long long* xz; unsigned long long* xd; int32_t* epoch;   // pair hand-off of the panel launch (panel.hpp)
    Panel16Images p16;         // fragment images of the 16-row panel kernel (panel16_images.hpp), rewritten every step
    bool compact;              // LSTM input rows are [E | s_prev] (i4 = round4(e_obs + smp_dim)), the table columns a bias
    int xc;                    // columns of an LSTM input row: e_obs + smp_dim (compact) or lstm_in
    int64_t e4, i4, hid4, out4, ohid4[PP_MAX_OBS], maxohid4;
    size_t bytes;
};

// The address terms of the LSTM input as a per-address bias (gather.hpp): the default for LSTM networks whose dimensions
// allow 16-byte pieces; PP_ADDR_BIAS=0 and the deterministic mode keep the full-width rows (A/B measurements, tests).
static bool compact_rows(const pp_net* net) {
    static const int env = getenv("PP_ADDR_BIAS") ? atoi(getenv("PP_ADDR_BIAS")) : 1;
    if (!env || deterministic_mode() || net->lstm_dim == 0) return false;
    const int c2 = net->e_obs + net->smp_dim, ne = net->dtype_dim + net->addr_dim;
    return net->lstm_in % 4 == 0 && c2 % 4 == 0 && net->lstm_dim % 16 == 0 && ne >= 2 && ne % 2 == 0 && ne <= 128 && net->n_addr >= 1 &&
           net->n_addr <= 1024 && net->addr_table != nullptr;
}
constexpr int DX_SPLITS = 16;
constexpr int DH_SPLITS = 8;
static int env_flag(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

static void carve(const pp_net* net, int B, int R, void* p, size_t cap, Workspace& w) {
    Carver c(p, cap);
    const bool ff = net->lstm_dim == 0;   // FeedForward network: no LSTM buffers, the heads read rows of width e_obs
    const int H = ff ? net->e_obs : net->lstm_dim;
    {   // first, so that their place does not depend on (B, R): the tail kernels' tags outlive a call
        size_t fb = 0, bb = 0;
        if (!ff) lstm_tail_exchange_bytes(H, &fb, &bb);
        w.xch_f = c.take<char>((int64_t)fb);
        w.xch_b = c.take<char>((int64_t)bb);
        // epoch of the panel launch's pair hand-off: a fixed place too (it counts the steps of this workspace)
        w.epoch = c.take<int32_t>(16);
    }
    w.e4 = round4(net->e_obs);
    w.compact = compact_rows(net);
    w.xc = w.compact ? net->e_obs + net->smp_dim : net->lstm_in;
    w.i4 = round4(w.xc);
    int64_t hid = 1, out = 1;
    for (int a = 0; a < net->n_addr; ++a) {
        hid = std::max<int64_t>(hid, net->addrs[a].hid);
        out = std::max<int64_t>(out, net->addrs[a].n_out);
    }
    w.hid4 = round4(hid);
    w.out4 = round4(out);
    w.maxohid4 = 4;
    for (int o = 0; o < net->n_obs; ++o) {
        w.ohid4[o] = round4(net->obs_hid[o]);
        w.maxohid4 = std::max(w.maxohid4, w.ohid4[o]);
        w.obs_h[o] = c.take<float>((int64_t)B * w.ohid4[o]);
        const int depth = net->obs_depth[o] ? net->obs_depth[o] : 2;
        for (int l = 0; l < PP_MAX_OBS_DEPTH; ++l) w.obs_hl[o][l] = nullptr;
        w.obs_hl[o][0] = w.obs_h[o];
        for (int l = 1; l + 1 < depth; ++l) w.obs_hl[o][l] = c.take<float>((int64_t)B * w.ohid4[o]);
    }
    w.cat = c.take<float>((int64_t)B * w.e4);
    w.f1 = c.take<float>((int64_t)B * w.e4);
    w.E = c.take<float>((int64_t)B * w.e4);
    w.X = c.take<float>(ff ? 0 : (int64_t)R * w.i4);
    const int L = ff ? 1 : std::max(1, std::min((int)net->lstm_depth, PP_MAX_LSTM_DEPTH));
    for (int l = 0; l < PP_MAX_LSTM_DEPTH; ++l) w.Gl[l] = w.Cl[l] = w.Hl[l] = nullptr;
    for (int l = 0; l < L; ++l) {
        w.Gl[l] = c.take<float>(ff ? 0 : (int64_t)R * 4 * H);
        w.Cl[l] = c.take<float>(ff ? 0 : (int64_t)R * H);
        w.Hl[l] = c.take<float>((int64_t)R * H);
    }
    w.G = w.Gl[0];
    w.C = w.Cl[0];
    w.Hs = w.Hl[L - 1];
    w.dH2 = c.take<float>(L > 1 ? (int64_t)R * H : 0);
    w.A1 = c.take<float>((int64_t)R * w.hid4);
    w.Y = c.take<float>((int64_t)R * w.out4);
    w.DY = c.take<float>((int64_t)R * w.out4);
    w.dZ1 = c.take<float>((int64_t)R * w.hid4);
    w.dH = c.take<float>((int64_t)R * H);
    w.dC = c.take<float>(ff ? 0 : (int64_t)B * H);
    w.dHp = c.take<float>(ff ? 0 : (int64_t)B * H * DH_SPLITS);   // stored K-split partials of dh_{t-1} += dG_t W_hh
    // (compact rows: room for the partial tiles of up to DX_SPLITS K splits of dX, see ic_loss)
    w.dX = c.take<float>(ff ? 0 : (int64_t)R * w.i4 * (w.compact ? DX_SPLITS : 1));
    w.dE = c.take<float>((int64_t)B * w.e4);
    w.dF1 = c.take<float>((int64_t)B * w.e4);
    w.dCat = c.take<float>((int64_t)B * w.e4);
    w.dObsH = c.take<float>((int64_t)net->n_obs * B * w.maxohid4);
    w.dObsH2 = c.take<float>((int64_t)B * w.maxohid4);
    // 64 loss accumulator slots, one per 128-byte line (atomics to one line serialise in its L2 channel), then the
    // non-finite flag; cleared by the step's first kernel
    w.loss_acc = c.take<float>(PP_LOSS_SLOTS_FLOATS);
    w.flag = reinterpret_cast<int32_t*>(w.loss_acc ? w.loss_acc + 64 * 32 : nullptr);
    w.lp_rows = c.take<float>(deterministic_mode() ? R : 0);
    w.AB = c.take<float>(w.compact ? (int64_t)net->n_addr * 2 * 4 * H : 0);
    w.gsum = c.take<float>(w.compact ? (int64_t)net->n_addr * 2 * 4 * H : 0);
    const bool panel_shape = w.compact && (H == 512 || H == 1024);
    w.WihT = c.take<float>(panel_shape ? (int64_t)net->e_obs * (4 * H + 64) : 0);
    w.W1T = c.take<float>(panel_shape ? (int64_t)H * 64 * ((hid + 63) / 64) : 0);
    // pair hand-off of the split panel launch: partial sums [panels][2][8][hid4] and [panels][2][8][64], flags, epoch
    // (sized for whichever panel kernel takes the batch: two workgroups per 8-row panel / four per 16-row panel)
    const int64_t n_pan = (B + 7) / 8;
    w.xz = c.take<unsigned long long>(panel_shape ? std::max<int64_t>(n_pan * 2 * 8 * w.hid4, panel16_xz_granules(B, H)) : 0);
    w.xd = c.take<unsigned long long>(panel_shape ? std::max<int64_t>(n_pan * 2 * 8 * 64, panel16_xd_granules(B)) : 0);
    w.p16 = Panel16Images{};
    if (panel_shape && net->e_obs == 64) {
        panel16_image_sizes(H, hid, net->e_obs, w.p16.frags);
        for (int i = 0; i < 6; ++i) w.p16.img[i] = c.take<float>(w.p16.frags[i] * 256);
    }
    w.bytes = c.off + 256;
}

static int check_net(const pp_net* net) {
    PP_CHECK_ARG(net, "null pp_net");
    PP_CHECK_ARG(net->n_obs >= 1 && net->n_obs <= PP_MAX_OBS, "pp_net: n_obs=%d out of range", net->n_obs);
    PP_CHECK_ARG(net->lstm_dim >= 0 && net->e_obs > 0, "pp_net: bad dimensions");
    if (net->lstm_dim == 0) {   // InferenceNetworkFeedForward: no LSTM, no address / sample embeddings
        PP_CHECK_ARG(net->lstm_in == 0, "pp_net: a FeedForward network (lstm_dim 0) has lstm_in 0");
    } else {
        PP_CHECK_ARG(net->lstm_in > 0 && net->lstm_in == net->e_obs + net->smp_dim + 2 * (net->addr_dim + net->dtype_dim),
                     "pp_net: lstm_in != e_obs + smp_dim + 2*(addr_dim+dtype_dim)");
    }
    PP_CHECK_ARG(net->n_addr == 0 || net->addrs, "pp_net: addrs is null");
    PP_CHECK_ARG(net->lstm_depth >= 0 && net->lstm_depth <= PP_MAX_LSTM_DEPTH, "pp_net: lstm_depth %d out of range", net->lstm_depth);
    return 0;
}

// y = act(x W^T + b): x [n, in] (ldx), W [out, in], y [n, out] (ldy)
static int linear_fwd(const float* x, int64_t ldx, const int32_t* x_idx, const float* W, const float* b, float* y,
                      int64_t ldy, int n, int in, int out, bool relu, const float* bias2, hipStream_t st,
                      const GemmHole* hole = nullptr) {
    pp_gemm_args g{};
    g.A = x; g.lda = ldx; g.a_idx = x_idx;
    g.B = W; g.ldb = in;
    g.C = y; g.ldc = ldy;
    g.M = n; g.N = out; g.K = in;
    g.bias = b; g.bias2 = bias2; g.relu = relu ? 1 : 0;
    return gemm_f32(&g, st, hole);
}

// dW[out, in] += dz^T x  (dz [n, out] (lddz), x [n, in] (ldx, optional k-gather x_idx)): queued; all weight-gradient
// products of a backward pass are leaves of the dependency graph and run as one grouped launch (gemm_f32_grouped)
// the weight-gradient leaves of the backward pass, one grouped launch (timed as kernel class 1 when armed)
static int launch_wgrads(std::vector<pp_gemm_args>& wq, hipStream_t st, const std::vector<GemmHole>* holes = nullptr,
                         bool timed = true, const AuxJobs* aux = nullptr, bool t1 = false) {
    if (wq.empty()) return aux ? aux_jobs_launch(*aux, st) : 0;
    double flops = 0.0;
    for (const auto& g : wq) flops += 2.0 * (double)g.M * (double)g.N * (double)g.K;
    if (timed) prof_begin(1, st);
    WgradT1Args wa;
    // k-major operands stream straight into MFMA fragments (wgrad_t1.hip) unless a product needs what only the tile kernels do
    if (t1 && wgrad_t1_build(wq.data(), holes && holes->size() == wq.size() ? holes->data() : nullptr, (int)wq.size(), wa))
        PP_TRY(wgrad_t1(wa, aux, st));
    else
    PP_TRY(gemm_f32_grouped(wq.data(), (int)wq.size(), st, holes ? holes->data() : nullptr, nullptr, aux));
    if (timed) prof_end(1, flops, st);
    return 0;
}

static void queue_wgrad(std::vector<pp_gemm_args>& q, const float* dz, int64_t lddz, const float* x, int64_t ldx,
                        const int32_t* x_idx, float* dW, int n, int in, int out, std::vector<GemmHole>* holes = nullptr,
                        GemmHole hole = GemmHole{}) {
    if (holes) {
        holes->resize(q.size(), GemmHole{});
        holes->push_back(hole);
    }
    pp_gemm_args g{};
    g.A = dz; g.lda = lddz; g.a_kmajor = 1;
    g.B = x; g.ldb = ldx; g.b_kmajor = 1; g.b_idx = x_idx;
    g.C = dW; g.ldc = in;
    g.M = out; g.N = in; g.K = n;
    g.accumulate = 1;
    g.split_k = 1;
    q.push_back(g);
}

static int linear_wgrad(const float* dz, int64_t lddz, const float* x, int64_t ldx, const int32_t* x_idx, float* dW,
                        float* db, float* db2, int n, int in, int out, hipStream_t st) {
    std::vector<pp_gemm_args> q;
    queue_wgrad(q, dz, lddz, x, ldx, x_idx, dW, n, in, out);
    PP_TRY(gemm_f32(&q[0], st));
    if (db) PP_TRY(colsum_f32(dz, lddz, nullptr, n, out, db, db2, st));
    return 0;
}

// dx[n, in] (lddx, optional scatter idx) = (dz W) (* relu mask); W [out, in]
static int linear_dgrad(const float* dz, int64_t lddz, const float* W, float* dx, int64_t lddx, const int32_t* dx_idx,
                        const float* mask, int64_t ldmask, int n, int in, int out, bool accumulate, hipStream_t st,
                        float* colsum = nullptr, const GemmHole* hole = nullptr) {
    pp_gemm_args g{};
    g.A = dz; g.lda = lddz;
    g.B = W; g.ldb = in; g.b_kmajor = 1;
    g.C = dx; g.ldc = lddx; g.c_idx = dx_idx;
    g.M = n; g.N = in; g.K = out;
    g.mask = mask; g.ldmask = ldmask;
    g.accumulate = accumulate ? 1 : 0;
    g.colsum = colsum;
    g.split_k = 1;
    return gemm_f32(&g, st, hole);
}

static int observe_embedding_fwd(const pp_net* net, const float* P, const float* obs, int64_t ldobs, int B, Workspace& w,
                                 hipStream_t st) {
    if (obs_fused_supported(net))   // small embeddings: one fused launch (obs_embed.hip)
        return obs_embed_fwd_fused(net, P, obs, B, w.obs_h, w.cat, w.f1, w.E, st);
    int ci = 0, co = 0;
    for (int o = 0; o < net->n_obs; ++o) {
        // EmbeddingFeedForward(num_layers = depth), ReLU after every layer (embedding_feedforward.py:35-48)
        const int depth = net->obs_depth[o] ? net->obs_depth[o] : 2;
        const float* x = obs + ci;
        int64_t ldx = ldobs;
        int in = net->obs_in[o];
        for (int l = 0; l < depth; ++l) {
            const bool last = l == depth - 1;
            const int out = last ? net->obs_out[o] : net->obs_hid[o];
            float* y = last ? w.cat + co : w.obs_hl[o][l];
            const int64_t ldy = last ? w.e4 : w.ohid4[o];
            const int64_t wl = net->obs_depth[o] ? net->obs_w[o][l] : (l == 0 ? net->obs_w0[o] : net->obs_w1[o]);
            const int64_t bl = net->obs_depth[o] ? net->obs_b[o][l] : (l == 0 ? net->obs_b0[o] : net->obs_b1[o]);
            PP_TRY(linear_fwd(x, ldx, nullptr, P + wl, P + bl, y, ldy, B, in, out, true, nullptr, st));
            x = y; ldx = ldy; in = out;
        }
        ci += net->obs_in[o];
        co += net->obs_out[o];
    }
    PP_TRY(linear_fwd(w.cat, w.e4, nullptr, P + net->fin_w0, P + net->fin_b0, w.f1, w.e4, B, net->e_obs, net->e_obs, true,
                      nullptr, st));
    PP_TRY(linear_fwd(w.f1, w.e4, nullptr, P + net->fin_w1, P + net->fin_b1, w.E, w.e4, B, net->e_obs, net->e_obs, true,
                      nullptr, st));
    return 0;
}

int ic_loss(const pp_net* net, const pp_batch* bt, const float* P, float* grads, void* ws,
            size_t ws_bytes, float* loss_out, int32_t* status_out, float* lp_out, int flags, hipStream_t st) {
    PP_TRY(check_net(net));
    PP_CHECK_ARG(bt && P && ws && loss_out, "pp_ic_loss: null pointer");
    PP_CHECK_ARG(bt->n_traces > 0 && bt->n_rows >= bt->n_traces && bt->t_max >= 1, "pp_ic_loss: empty batch");
    PP_CHECK_ARG(bt->n_active && bt->row_off && bt->grp_off && bt->obs && bt->value && bt->prior && bt->addr &&
                     bt->prev_row && bt->grp_rows && bt->trace && bt->row_off_dev && bt->nxt_off && bt->nxt_rows,
                 "pp_ic_loss: incomplete pp_batch");
    {
        int obs_in = 0;
        for (int o = 0; o < net->n_obs; ++o) obs_in += net->obs_in[o];
        PP_CHECK_ARG(bt->obs_width == obs_in, "pp_ic_loss: batch obs_width %d != the network's observable width %d",
                     bt->obs_width, obs_in);
    }
    const bool bwd = flags & PP_LOSS_BACKWARD;
    PP_CHECK_ARG(!bwd || grads, "pp_ic_loss: PP_LOSS_BACKWARD needs a gradient buffer");
    const bool ff = net->lstm_dim == 0;   // FeedForward network (inference_network_feedforward.py:68-98)
    const int B = bt->n_traces, R = bt->n_rows, T = bt->t_max, H = ff ? net->e_obs : net->lstm_dim, I = net->lstm_in;
    Workspace w;
    carve(net, B, R, ws, ws_bytes, w);
    if (w.bytes > ws_bytes) {
        set_error("pp_ic_loss: workspace too small (%zu < %zu bytes)", ws_bytes, w.bytes);
        return PP_ENOSPACE;
    }
    if (bwd && (flags & PP_LOSS_ZERO_GRADS)) (void)hipMemsetAsync(grads, 0, (size_t)net->n_params * sizeof(float), st);

    // ---------------- forward ----------------
    // First kernel: observe embedding; for small embeddings the same launch assembles the LSTM input rows of its traces
    // and clears the loss slots and (backward) dX. Otherwise embedding GEMMs + the stand-alone gather kernel.
    const bool fused_obs = obs_fused_supported(net);
    // ragged batches: the time steps t >= tail_t0 (few rows each) of every LSTM layer run in one launch per direction
    int tail_t0 = T, tail_teams = 0;
    if (!ff) lstm_tail_plan(bt->n_active, T, H, &tail_t0, &tail_teams);
    const int n_clear = PP_LOSS_SLOTS_FLOATS;
    const float* heads_in = w.Hs;     // input rows of the proposal layers: LSTM outputs, or observe embeddings (FF)
    int64_t heads_ld = H;
    // kernel class 2 of the in-stream timing: observe embedding + LSTM input rows (the gather path). Algorithmic bytes:
    // observations and per-row (value, address, previous row) in; X rows, E / cat / f1 and the observables' hidden
    // activations out (4 I per trace-step written, 4 n_obs per trace read)
    double gather_bytes = 4.0 * B * bt->obs_width + 12.0 * R + 4.0 * (ff ? 0.0 : (double)R * w.xc) + 3.0 * 4.0 * B * net->e_obs;
    if (!ff && (flags & PP_LOSS_BACKWARD)) gather_bytes += 4.0 * R * w.xc;   // the same launch clears dX
    // compact rows: the same launch computes the per-address bias vectors of the LSTM input (reads W_ih[:, c2:I] per
    // present address, writes 2 x 4H, clears the group sums)
    const bool compact = w.compact;
    const int c2x = net->e_obs + net->smp_dim, ne_x = net->dtype_dim + net->addr_dim;
    // single-statement batches have no previous statement at all: the sample-embedding columns are zero in every row
    const int nx = (T == 1) ? net->e_obs : c2x;
    // Single-statement batch: dX = dG W_ih[:, :e_obs] has ONE consumer, the observe-embedding backward kernel; its K splits
    // store their partial tiles and that kernel adds them - no float atomics (6-8 us per 64 x 64 tile, tools/wg_trace.py)
    // and no cleared dX
    static const int dx_partials_env = env_flag("PP_DX_PARTIALS", 1);
    const bool dx_partials = compact && bwd && T == 1 && dx_partials_env && obs_fused_supported(net);
    // (see GemmExt::lean) the cell backward of a single-statement, single-layer batch runs in the dH epilogue, and with the
    // zero blocks on neither dX nor dW_ih read the forget gate's columns of first-step rows
    // (H a multiple of 64: the zero blocks then cover the forget gate's tiles and slabs exactly)
    static const bool lean_env = env_flag("PP_FUSE_CELL_BWD", 1) && env_flag("PP_FUSE_CELL", 1) && env_flag("PP_GEMM_HOLES", 1) == 1 &&
                                 env_flag("PP_CELL_LEAN", 1);
    // (and dX as stored split partials: that product then runs on the async tiles, which skip the forget gate's K range)
    const bool lean_cell = compact && bwd && T == 1 && std::max(1, (int)net->lstm_depth) == 1 && H % 64 == 0 && lean_env &&
                           dx_partials;
    AddrBias abias{};
    int n_present = 0, only_addr = 0;   // addresses that occur in the batch
    if (compact) {
        abias.AB = w.AB; abias.gsum = w.gsum; abias.step_epoch = w.epoch;
        abias.W = P + net->w_ih; abias.b_ih = P + net->b_ih; abias.b_hh = P + net->b_hh;
        abias.params = P; abias.at = net->addr_table; abias.ldw = I;
        abias.N = 4 * H; abias.c2 = c2x; abias.c3 = c2x + net->dtype_dim; abias.c4 = c2x + ne_x;
        abias.c5 = abias.c4 + net->dtype_dim; abias.I = I; abias.n_addr = net->n_addr;
        for (int q = 0; q < 32; ++q) abias.present[q] = 0u;
        for (int a = 0; a < net->n_addr; ++a)
            if (bt->grp_off[a + 1] > bt->grp_off[a] || bt->nxt_off[a + 1] > bt->nxt_off[a]) {
                abias.present[a >> 5] |= 1u << (a & 31);
                ++n_present;
                only_addr = a;
            }
        gather_bytes += (double)n_present * (4.0 * 4 * H * (2.0 * ne_x) + 4.0 * 4.0 * 4 * H);
    }
    for (int o = 0; o < net->n_obs; ++o) gather_bytes += 4.0 * B * net->obs_hid[o];
    // Row-panel kernel (panel.hip): a single-statement batch with ONE address keeps input product + cell, head layer 1, the
    // head tail, dz1, dH + cell backward and dX in one launch (rows of the one address group are the batch rows in order)
    bool panel = false, obs_tail = false;
    bool panel16_go = false;      // the 16-row kernel (panel16.hip) instead of the 8-row one (panel.hip)
    if (lean_cell && n_present == 1 && fused_obs && bt->grp_off[only_addr] == 0 && bt->grp_off[only_addr + 1] == R) {
        const pp_addr& ad = net->addrs[only_addr];
        const bool shape_ok = head_tail_supported(ad.kind, ad.hid, ad.n_out) &&
                              w.hid4 <= ((ad.hid + 15) & ~15) && w.out4 <= 64 && (flags & PP_LOSS_KEEP_LP ? lp_out != nullptr : true);
        panel16_go = shape_ok && w.p16.img[0] && panel16_supported(ad.kind, H, ad.hid, ad.n_out, net->e_obs, R);
        if (panel16_go && bwd) {
            ObsFusedArgs oa;
            panel16_go = obs_fused_supported(net) && obs_fused_args(net, w.obs_h, oa) && panel16_obs_ok(oa);
        }
        panel = panel16_go || (shape_ok && panel_t1_supported(ad.kind, H, ad.hid, ad.n_out, net->e_obs) && panel_t1_split(R, H) == 2);
    }
    prof_begin(2, st);
    if (ff) {
        // every time step's proposal layer reads the observe embedding of its trace (:72,85): Hs rows = E[trace]
        // (single-statement batches: row r IS trace r, the heads read E in place and the embedding kernel clears the
        // loss slots - one launch less)
        const bool in_place = fused_obs && T == 1;
        if (fused_obs) {
            RowBuild rb{};
            rb.zero_small = in_place ? reinterpret_cast<float*>(w.loss_acc) : nullptr;
            rb.n_small = n_clear;
            PP_TRY(obs_embed_fwd_fused(net, P, bt->obs, B, w.obs_h, w.cat, w.f1, w.E, st, &rb));
        } else {
            PP_TRY(observe_embedding_fwd(net, P, bt->obs, bt->obs_width, B, w, st));
        }
        if (in_place) {
            heads_in = w.E;
            heads_ld = w.e4;
        } else {
            PP_TRY(embedding_rows(w.E, w.e4, bt->trace, R, net->e_obs, w.Hs, H, reinterpret_cast<float*>(w.loss_acc),
                                  n_clear, st));
        }
    } else if (fused_obs && T <= 2) {   // (long traces: a wave would write all rows of its trace serially - separate gather)
        RowBuild rb{};
        rb.d = GatherDims{net->e_obs, net->smp_dim, net->dtype_dim, net->addr_dim, net->lstm_in};
        rb.params = P; rb.at = net->addr_table; rb.row_off = bt->row_off_dev; rb.t_max = T;
        rb.value = bt->value; rb.addr = bt->addr; rb.prev_row = bt->prev_row;
        rb.X = w.X; rb.ldx = w.i4; rb.xcols = w.xc;
        rb.zero_like = (bwd && !dx_partials) ? w.dX : nullptr;
        rb.zero_small = reinterpret_cast<float*>(w.loss_acc); rb.n_small = n_clear;
        PanelTranspose ptr{};
        if (panel) {
            const pp_addr& ad = net->addrs[only_addr];
            ptr.Wih = P + net->w_ih; ptr.ldw = I; ptr.WihT = w.WihT;
            ptr.W1 = P + ad.w1; ptr.W1T = w.W1T; ptr.ld1T = 64 * ((ad.hid + 63) / 64);
            ptr.H = H; ptr.hid = ad.hid; ptr.e = net->e_obs;
            ptr.tiles_ih = 3 * H / 64;
            ptr.n_blocks = panel_transpose_blocks(H, ad.hid);
            if (panel16_go) {      // the job writes the six fragment images instead (one thread per fragment lane)
                ptr.mode16 = 1;
                ptr.p16.W2 = P + ad.w2; ptr.p16.n_out = ad.n_out;
                ptr.p16.im = w.p16;
                panel16_image_sizes(H, ad.hid, net->e_obs, ptr.p16.im.frags);      // (this address's head; the buffers hold the widest)
                int nb = 0;
                for (int i = 0; i < 6; ++i) {
                    ptr.p16.blocks_before[i] = nb;
                    nb += cdiv(ptr.p16.im.frags[i] * 64, 256);
                }
                ptr.p16.blocks_before[6] = nb;
                ptr.n_blocks = nb;
            }
        }
        PP_TRY(obs_embed_fwd_fused(net, P, bt->obs, B, w.obs_h, w.cat, w.f1, w.E, st, &rb, compact ? &abias : nullptr,
                                   panel ? &ptr : nullptr));
    } else {
        PP_TRY(observe_embedding_fwd(net, P, bt->obs, bt->obs_width, B, w, st));
        // (also clears the loss slots and, for a backward pass, dX: see the kernel)
        PP_TRY(lstm_input_gather(net, P, w.E, w.e4, bt->trace, bt->value, bt->addr, bt->prev_row, -1, -1, R, w.X, w.i4, st,
                                 bwd ? w.dX : nullptr, reinterpret_cast<float*>(w.loss_acc), n_clear, w.xc,
                                 compact ? &abias : nullptr));
    }
    prof_end(2, gather_bytes, st);
    // nn.LSTM(I, H, depth), inference_network_lstm.py:31,186-188: layer k reads the hidden states of layer k - 1
    const int L = ff ? 0 : std::max(1, (int)net->lstm_depth);
    const bool compact_ok_dims = !ff && !deterministic_mode();   // (fused epilogues use float4 stores and no fixed order is at stake)
    auto lw_ih = [&](int l) { return l == 0 ? net->w_ih : net->lstm_w_ih[l]; };
    auto lw_hh = [&](int l) { return l == 0 ? net->w_hh : net->lstm_w_hh[l]; };
    auto lb_ih = [&](int l) { return l == 0 ? net->b_ih : net->lstm_b_ih[l]; };
    auto lb_hh = [&](int l) { return l == 0 ? net->b_hh : net->lstm_b_hh[l]; };
    for (int l = 0; l < L; ++l) {
        const float* in = l == 0 ? w.X : w.Hl[l - 1];
        const int64_t in_ld = l == 0 ? w.i4 : H;
        const int in_w = l == 0 ? I : H;
        // a trace's first time step has no previous variable: columns [e_obs, c4) of its LSTM input row are zero
        // (inference_network_lstm.py:159-162) - rows [0, B) of the step-major layout, layer 0 -
        // ... and no previous cell state in any layer: its forget gate multiplies c_{-1} = 0, so columns [H, 2H) of its
        // pre-activations are never looked at (lstm_cell_fwd/bwd with c_prev == NULL) - not computed at all
        GemmHole zero{};
        zero.b[1] = GemmBlock{0, B, H, 2 * H, 0, in_w};
        if (l == 0) zero.b[0] = GemmBlock{0, B, 0, 4 * H, net->e_obs, net->e_obs + net->smp_dim + net->dtype_dim + net->addr_dim};
        if (l == 0) prof_begin(0, st);
        bool cell_done = false;   // the first time step's cell ran in the product's epilogue
        if (l == 0 && compact) {
            // G = [E | s_prev] W_ih[:, :c2]^T + cur[addr] + prev[previous addr] (gather.hpp); first-step rows have no previous
            // statement: their sample-embedding columns are zero too
            pp_gemm_args g{};
            g.A = w.X; g.lda = w.i4;
            g.B = P + net->w_ih; g.ldb = I;
            g.C = w.Gl[0]; g.ldc = 4 * H;
            g.M = R; g.N = 4 * H; g.K = nx;
            GemmExt x{};
            x.rb = w.AB; x.rb_addr = bt->addr; x.rb_prev = T > 1 ? bt->prev_row : nullptr;
            if (T == 1 && n_present == 1) {   // one address in a single-statement batch: the bias is one vector
                x.rb = w.AB + (int64_t)only_addr * 2 * 4 * H;
                x.rb_addr = nullptr;
            }
            static const int fuse_cell = env_flag("PP_FUSE_CELL", 1);
            if (fuse_cell) {   // gate-interleaved tiles, LSTM cell of the first time step in the epilogue
                x.cell_H = H; x.cell_rows = B; x.cell_c = w.Cl[0]; x.cell_h = w.Hl[0];
                x.lean = lean_cell ? 1 : 0;
                cell_done = true;
            }
            GemmHole zc{};
            zc.b[0] = GemmBlock{0, B, 0, 4 * H, net->e_obs, nx};
            if (panel) {
                const pp_addr& ad = net->addrs[only_addr];
                PanelArgs pa{};
                pa.B = R; pa.H = H; pa.hid = ad.hid; pa.n_out = ad.n_out; pa.K = ad.n_out / 3; pa.e = net->e_obs;
                pa.ldx = (int)w.i4; pa.lda1 = (int)w.hid4; pa.lddy = (int)w.out4; pa.ldw = I;
                pa.X = w.X; pa.Wih = P + net->w_ih; pa.AB = w.AB + (int64_t)only_addr * 2 * 4 * H;
                pa.WihT = w.WihT; pa.W1T = w.W1T;
                pa.xz = w.xz; pa.xd = w.xd; pa.epoch = w.epoch;
                pa.W1 = P + ad.w1; pa.b1 = P + ad.b1; pa.W2 = P + ad.w2; pa.b2 = P + ad.b2;
                pa.value = bt->value; pa.prior = bt->prior;
                pa.Hs = w.Hl[0]; pa.G = w.Gl[0]; pa.A1 = w.A1; pa.DY = w.DY; pa.dZ1 = w.dZ1; pa.dX = w.dX;
                pa.gsum = w.gsum + (int64_t)only_addr * 2 * 4 * H;
                pa.lp_out = (flags & PP_LOSS_KEEP_LP) ? lp_out : nullptr;
                pa.loss_acc = w.loss_acc; pa.flag = w.flag; pa.grad_scale = -1.0f / (float)B;
                pa.dbg = g_timeline;
                // training: the observe-embedding backward of the rows rides in the kernel's tail (one launch less)
                PanelObs po{};
                obs_tail = bwd && (panel16_go || panel_obs_tail_ok(net, H, ad.hid, ad.n_out, net->e_obs)) && obs_fused_args(net, w.obs_h, po.a);
                if (obs_tail) {
                    po.P = P; po.cat = w.cat; po.f1 = w.f1;
                    po.dE = w.dE; po.dF1 = w.dF1; po.dCat = w.dCat; po.dHo0 = w.dObsH;
                    po.dh_stride = (int64_t)B * w.maxohid4;
                }
                if (panel16_go) {
                    Panel16Args p16{};
                    p16.a = pa;
                    for (int i = 0; i < 6; ++i) p16.img[i] = w.p16.img[i];
                    PP_TRY(panel16(ad.kind, p16, st, obs_tail ? &po : nullptr));
                } else {
                    PP_TRY(panel_t1(ad.kind, pa, st, obs_tail ? &po : nullptr));
                }
                cell_done = true;
                // executed data-path FLOPs of the launch: forward + backward products of the 8-row panels
                prof_end(0, 2.0 * R * (2.0 * 3.0 * H * net->e_obs + 2.0 * (double)H * ad.hid + 2.0 * (double)ad.hid * ad.n_out), st);
            } else {
            if (lstm_input_fast_ok(g, x)) PP_TRY(lstm_input_fast(g, x, st));      // (lstm_input.hip; the zero block is zeros in X)
            else PP_TRY(gemm_f32(&g, st, &zc, &x));
            prof_end(0, 2.0 * R * (double)nx * 4.0 * H, st);
            }
        } else {
        PP_TRY(linear_fwd(in, in_ld, nullptr, P + lw_ih(l), P + lb_ih(l), w.Gl[l], 4 * H, R, in_w, 4 * H, false, P + lb_hh(l), st,
                          &zero));
        if (l == 0) prof_end(0, 2.0 * R * (double)I * 4.0 * H, st);
        }
        for (int t = 0; t < T; ++t) {
            if (tail_teams && t == tail_t0) {   // all remaining time steps of this layer: one launch (lstm_tail.hip)
                PP_TRY(lstm_tail_fwd(w.Gl[l], w.Cl[l], w.Hl[l], P + lw_hh(l), bt->row_off_dev, tail_t0, T, H, tail_teams,
                                     w.xch_f, w.xch_b, w.flag, st));
                break;
            }
            const int n = bt->n_active[t], r0 = bt->row_off[t];
            float* Gt = w.Gl[l] + (int64_t)r0 * 4 * H;
            const float* c_prev = nullptr;
            if (t > 0) {
                const int rp = bt->row_off[t - 1];
                pp_gemm_args g{};
                g.A = w.Hl[l] + (int64_t)rp * H; g.lda = H;
                g.B = P + lw_hh(l); g.ldb = H;
                g.C = Gt; g.ldc = 4 * H;
                g.M = n; g.N = 4 * H; g.K = H;
                c_prev = w.Cl[l] + (int64_t)rp * H;
                // Recurrent product with the cell in its epilogue (gate-interleaved tiles; one workgroup per tile walks all of
                // K = H, the pre-activations are read instead of accumulated into): no lstm_cell_fwd launch, no round trip of
                // G. Needs enough tiles to fill the chip without a K split: n >= 64 rows x 4H / 64 column tiles.
                static const int fuse_rec = env_flag("PP_FUSE_CELL_REC", 1);
                if (fuse_rec && compact_ok_dims && H % 16 == 0 && n >= 64) {
                    GemmExt x{};
                    x.cell_H = H; x.cell_rows = n; x.cell_c = w.Cl[l] + (int64_t)r0 * H; x.cell_h = w.Hl[l] + (int64_t)r0 * H;
                    x.cell_cprev = c_prev;
                    PP_TRY(gemm_f32(&g, st, nullptr, &x));
                    continue;
                }
                g.accumulate = 1;
                g.split_k = 1;   // few rows late in a ragged batch: spread K over workgroups (accumulation into G)
                PP_TRY(gemm_f32(&g, st));
            }
            if (t == 0 && cell_done) continue;
            PP_TRY(lstm_cell_fwd(Gt, c_prev, w.Cl[l] + (int64_t)r0 * H, w.Hl[l] + (int64_t)r0 * H, n, H, st));
        }
    }
    const float gscale = -1.0f / (float)B;
    std::vector<ColsumJob> cs;
    // deterministic mode: the heads write the per-row log_prob only; the loss is a fixed-order sum over the rows
    const bool det = deterministic_mode();
    float* const lp_rows = det ? (((flags & PP_LOSS_KEEP_LP) && lp_out) ? lp_out : w.lp_rows)
                               : ((flags & PP_LOSS_KEEP_LP) ? lp_out : nullptr);
    float* const loss_slots = det ? nullptr : w.loss_acc;
    // heads: first FF layer of EVERY address group in one grouped launch (rows gathered by address: the dispatch
    // gather), then the fused tails, grouped by (kind, shape)
    {
        std::vector<pp_gemm_args> hq;
        for (int a = 0; a < net->n_addr; ++a) {
            const int g0 = bt->grp_off[a], n = bt->grp_off[a + 1] - g0;
            if (n <= 0) continue;
            const pp_addr& ad = net->addrs[a];
            pp_gemm_args g{};
            g.A = heads_in; g.lda = heads_ld; g.a_idx = bt->grp_rows + g0;
            g.B = P + ad.w1; g.ldb = H;
            g.C = w.A1 + (int64_t)g0 * w.hid4; g.ldc = w.hid4;
            g.M = n; g.N = ad.hid; g.K = H;
            g.bias = P + ad.b1; g.relu = 1;
            hq.push_back(g);
        }
        if (!panel) PP_TRY(gemm_f32_grouped(hq.data(), (int)hq.size(), st));
    }
    std::vector<char> done(net->n_addr, 0);
    for (int a = 0; a < net->n_addr; ++a) {
        const int g0 = bt->grp_off[a], n = bt->grp_off[a + 1] - g0;
        if (n <= 0 || done[a]) continue;
        const pp_addr& ad = net->addrs[a];
        if (head_tail_supported(ad.kind, ad.hid, ad.n_out)) {
            // fused tail (layer 2 + log_prob + loss [+ dy, dz1]) for this and every later group of the same head shape
            std::vector<TailJob> tj;
            for (int b = a; b < net->n_addr; ++b) {
                const pp_addr& bd = net->addrs[b];
                const int h0 = bt->grp_off[b], m = bt->grp_off[b + 1] - h0;
                if (m <= 0 || done[b] || bd.kind != ad.kind || bd.hid != ad.hid || bd.n_out != ad.n_out) continue;
                done[b] = 1;
                tj.push_back(TailJob{w.A1 + (int64_t)h0 * w.hid4, P + bd.w2, P + bd.b2, bt->grp_rows + h0,
                                     bwd ? w.DY + (int64_t)h0 * w.out4 : nullptr, w.dZ1 + (int64_t)h0 * w.hid4, m});
                if (bwd) {   // bias gradients by the low-contention column-sum kernel (multi-job launches below)
                    cs.push_back(ColsumJob{w.DY + (int64_t)h0 * w.out4, w.out4, nullptr, m, bd.n_out, grads + bd.b2, nullptr});
                    cs.push_back(ColsumJob{w.dZ1 + (int64_t)h0 * w.hid4, w.hid4, nullptr, m, bd.hid, grads + bd.b1, nullptr});
                }
            }
            if (!panel)
                PP_TRY(head_tail_multi(ad.kind, tj.data(), (int)tj.size(), w.hid4, ad.hid, ad.n_out, bt->value, bt->prior, gscale,
                                       lp_rows, w.out4, w.hid4, loss_slots, w.flag, st));
            continue;
        }
        done[a] = 1;
        float* A1 = w.A1 + (int64_t)g0 * w.hid4;
        float* Y = w.Y + (int64_t)g0 * w.out4;
        PP_TRY(linear_fwd(A1, w.hid4, nullptr, P + ad.w2, P + ad.b2, Y, w.out4, n, ad.hid, ad.n_out, false, nullptr, st));
        PP_TRY(head_logprob(ad.kind, Y, w.out4, bt->grp_rows + g0, bt->value, bt->prior, n, ad.n_out, gscale,
                            lp_rows, bwd ? w.DY + (int64_t)g0 * w.out4 : nullptr, loss_slots, w.flag, st));
    }
    if (det) PP_TRY(loss_from_rows(lp_rows, R, B, w.flag, loss_out, status_out, st));
    if (!bwd) {
        if (!det) PP_TRY(loss_finalize(w.loss_acc, w.flag, B, loss_out, status_out, st));
        return 0;
    }
    // (deterministic mode: the loss is final; the kernels below get no loss slots to fold)
    const float* const fin_acc = det ? nullptr : w.loss_acc;
    // (with a backward pass the loss slots are folded by the first LSTM-cell launch below)
    // (the bias / table column sums queued in `cs` are launched once, at the end of the backward pass)

    // ---------------- backward ----------------
    // Weight-gradient leaves: queued and flushed as ONE grouped launch at the end of the backward pass.
    // (a second stream for these leaves with fork / join events was measured twice and lost both times - 0.157 -> 0.166 ms
    // on config 2, 183 vs 170 us with two grouped launches: a cross-queue event wait costs ~5 us - and is gone)
    std::vector<pp_gemm_args> wq;
    std::vector<GemmHole> wholes;
    auto flush_wgrads = [&](hipStream_t stream, bool timed, const AuxJobs* aux = nullptr) -> int {
        wholes.resize(wq.size(), GemmHole{});
        // data parallel with an overlap range (dp.hip): the products that complete the range - the LSTM layer's weight
        // gradients - and the reduction jobs go first; the range's all-reduce starts on the side stream behind them and the
        // remaining products (proposal layers, observe embedding) run under it
        int64_t lo = 0, hi = 0;
        if (dp_overlap_hull(&lo, &hi) && wq.size() > 1) {
            std::vector<pp_gemm_args> qa, qb;
            std::vector<GemmHole> ha, hb;
            for (size_t i = 0; i < wq.size(); ++i) {
                const bool in = wq[i].C >= grads + lo && wq[i].C < grads + hi;
                (in ? qa : qb).push_back(wq[i]);
                (in ? ha : hb).push_back(wholes[i]);
            }
            if (!qa.empty() && !qb.empty()) {
                PP_TRY(launch_wgrads(qa, stream, &ha, timed, aux, true));
                PP_TRY(dp_bucket0_issue(grads, stream));
                PP_TRY(launch_wgrads(qb, stream, &hb, false, nullptr, true));
                wq.clear();
                wholes.clear();
                return 0;
            }
        }
        PP_TRY(launch_wgrads(wq, stream, &wholes, timed, aux, true));
        wq.clear();
        wholes.clear();
        return 0;
    };
    // Single-statement batch, one LSTM layer: every row is a trace's only time step, so the cell backward needs nothing
    // but dh from the heads - it runs in the epilogue of the dH product (gemm_tile_direct), which also adds each tile's
    // column sums of dG to its address's group sums; the loss is finalised by the jobs behind the weight-gradient tiles.
    static const int fuse_cell_bwd = env_flag("PP_FUSE_CELL_BWD", 1);
    const bool fused_bwd = compact && T == 1 && L == 1 && fuse_cell_bwd;
    int dx_splits = 1;
    std::vector<pp_gemm_args> dq;   // per-address data gradients into dH
    for (int a = 0; a < net->n_addr; ++a) {
        const int g0 = bt->grp_off[a], n = bt->grp_off[a + 1] - g0;
        if (n <= 0) continue;
        const pp_addr& ad = net->addrs[a];
        const float* A1 = w.A1 + (int64_t)g0 * w.hid4;
        const float* DY = w.DY + (int64_t)g0 * w.out4;
        float* dZ1 = w.dZ1 + (int64_t)g0 * w.hid4;
        const bool fused = head_tail_supported(ad.kind, ad.hid, ad.n_out);   // dz1 already produced by the forward tail
        queue_wgrad(wq, DY, w.out4, A1, w.hid4, nullptr, grads + ad.w2, n, ad.hid, ad.n_out);
        if (!fused) {
            PP_TRY(colsum_f32(DY, w.out4, nullptr, n, ad.n_out, grads + ad.b2, nullptr, st));
            PP_TRY(linear_dgrad(DY, w.out4, P + ad.w2, dZ1, w.hid4, nullptr, A1, w.hid4, n, ad.hid, ad.n_out, false, st,
                                det ? nullptr : grads + ad.b1));   // db1 = colsum(dZ1) fused into the epilogue
            if (det) cs.push_back(ColsumJob{dZ1, w.hid4, nullptr, n, ad.hid, grads + ad.b1, nullptr});
        }
        // (panel kernel: one group that covers all rows in order - no gather, the product can take the streaming kernel)
        queue_wgrad(wq, dZ1, w.hid4, heads_in, heads_ld, panel ? nullptr : bt->grp_rows + g0, grads + ad.w1, n, H, ad.hid);
        {   // dH[rows of this address] = dZ1 W1: queued, every address group in one grouped launch
            pp_gemm_args g{};
            g.A = dZ1; g.lda = w.hid4;
            g.B = P + ad.w1; g.ldb = H; g.b_kmajor = 1;
            g.C = w.dH; g.ldc = H; g.c_idx = bt->grp_rows + g0;
            g.M = n; g.N = H; g.K = ad.hid;
            if (fused_bwd) g.colsum = w.gsum + (int64_t)a * 2 * 4 * H;   // group sums of dG (current-address slot)
            dq.push_back(g);
        }
    }
    if (panel) {
        // (dG, the group sums and dX came out of the panel launch)
    } else if (fused_bwd) {
        GemmExt x{};
        x.bw_G = w.Gl[0]; x.bw_C = w.Cl[0]; x.bw_H = H; x.lean = lean_cell ? 1 : 0;
        PP_TRY(gemm_f32_grouped(dq.data(), (int)dq.size(), st, nullptr, &x));
    } else {
        PP_TRY(gemm_f32_grouped(dq.data(), (int)dq.size(), st));
    }
    // the gradient of the observe embedding is summed over the time steps from dX[:, :e_obs] (LSTM) / from dH (FF)
    const float* dXs = ff ? w.dH : w.dX;
    const int64_t ldxs = ff ? H : w.i4;
    const int cz0 = net->e_obs, cz1 = net->e_obs + net->smp_dim + net->dtype_dim + net->addr_dim;   // previous-variable columns
    float* dH_cur = w.dH;          // gradient into the hidden states of the layer being processed (top: from the heads)
    float* dH_other = w.dH2;
    for (int l = L - 1; l >= 0; --l) {
        int dh_parts = 0;   // K splits of dG_{t+1} W_hh waiting in w.dHp for the cell backward of step t
        for (int t = T - 1; t >= 0; --t) {
            if (fused_bwd) break;   // dG is already in place
            if (tail_teams && t >= tail_t0) {   // steps T-1 .. tail_t0 in one launch; it leaves dh / dc of step tail_t0 - 1
                const LossFinalize fin{l == L - 1 ? fin_acc : nullptr, w.flag, B > 0 ? 1.0f / (float)B : 0.0f, loss_out, status_out};
                PP_TRY(lstm_tail_bwd(w.Gl[l], w.Cl[l], dH_cur, w.dC, P + lw_hh(l), bt->row_off_dev, tail_t0, T, H, tail_teams,
                                     w.xch_f, w.xch_b, w.flag, det ? nullptr : grads + lb_ih(l), det ? nullptr : grads + lb_hh(l),
                                     fin, st));
                t = tail_t0;
                continue;
            }
            const int n = bt->n_active[t], r0 = bt->row_off[t];
            const int n_next = (t + 1 < T) ? bt->n_active[t + 1] : 0;
            float* Gt = w.Gl[l] + (int64_t)r0 * 4 * H;
            const float* c_prev = t > 0 ? w.Cl[l] + (int64_t)bt->row_off[t - 1] * H : nullptr;
            // the backward pass's first cell launch folds the loss slots (the tail launch when there is one)
            const bool fin = l == L - 1 && t == T - 1;
            PP_TRY(lstm_cell_bwd(Gt, c_prev, w.Cl[l] + (int64_t)r0 * H, dH_cur + (int64_t)r0 * H, w.dC, n, n_next, H,
                                 det ? nullptr : grads + lb_ih(l), det ? nullptr : grads + lb_hh(l), st,
                                 fin ? fin_acc : nullptr, w.flag, B, loss_out, status_out, w.dHp, dh_parts,
                                 (int64_t)B * H));   // bias gradients fused; + the stored partials of dG_{t+1} W_hh
            dh_parts = 0;
            if (t > 0) {  // dh_{t-1} += dG_t W_hh
                // As K-split partial tiles that the NEXT cell-backward launch adds (rows [0, n) of step t - 1 are the same
                // traces): no float atomics (6 us per 64 x 64 tile), no read-modify-write of dH
                static const int dh_partials_env = env_flag("PP_DH_PARTIALS", 1);
                const int tiles = cdiv(n, 64) * cdiv(H, 64), nslab = 4 * H / 32;
                const int splits = std::max(1, std::min({DH_SPLITS, cdiv(256, tiles), nslab / 2}));
                if (dh_partials_env && !det && splits > 1 && H % 4 == 0) {
                    pp_gemm_args g{};
                    g.A = Gt; g.lda = 4 * H;
                    g.B = P + lw_hh(l); g.ldb = H; g.b_kmajor = 1;
                    g.C = w.dHp; g.ldc = H;
                    g.M = n; g.N = H; g.K = 4 * H;
                    GemmExt x{};
                    x.split_stride = (int64_t)B * H;
                    x.force_splits = splits;
                    PP_TRY(gemm_f32(&g, st, nullptr, &x));
                    dh_parts = splits;
                } else {
                    PP_TRY(linear_dgrad(Gt, 4 * H, P + lw_hh(l), dH_cur + (int64_t)bt->row_off[t - 1] * H, H, nullptr, nullptr, 0,
                                        n, H, 4 * H, true, st));
                }
            }
        }
        if (det)   // bias gradients = column sums of the complete dG of this layer, by the single-writer kernel
            cs.push_back(ColsumJob{w.Gl[l], 4 * H, nullptr, R, 4 * H, grads + lb_ih(l), grads + lb_hh(l)});
        // parameter gradients of this layer (leaves, grouped with every head's weight gradients). First-time-step rows give
        // nothing to the forget-gate rows (dG[:, H:2H] = 0 where c_{t-1} = 0) and, in layer 0, nothing to the
        // previous-variable columns of dW_ih (their inputs are zero there)
        const float* in = l == 0 ? w.X : w.Hl[l - 1];
        const int64_t in_ld = l == 0 ? w.i4 : H;
        const int in_w = l == 0 ? I : H;
        GemmHole wh{};
        if (l == 0 && compact) {   // dW_ih[:, :c2] = dG^T [E | s_prev]; the table columns follow from the group sums (aux jobs)
            wh.b[1] = GemmBlock{H, 2 * H, 0, nx, 0, B};
            wh.b[0] = GemmBlock{0, 4 * H, cz0, nx, 0, B};
            queue_wgrad(wq, w.Gl[0], 4 * H, w.X, w.i4, nullptr, grads + net->w_ih, R, nx, 4 * H, &wholes, wh);
            wq.back().ldc = I;
        } else {
        wh.b[1] = GemmBlock{H, 2 * H, 0, in_w, 0, B};
        if (l == 0) wh.b[0] = GemmBlock{0, 4 * H, cz0, cz1, 0, B};
        queue_wgrad(wq, w.Gl[l], 4 * H, in, in_ld, nullptr, grads + lw_ih(l), R, in_w, 4 * H, &wholes, wh);
        }
        if (T > 1) {
            const int r1 = bt->row_off[1];
            queue_wgrad(wq, w.Gl[l] + (int64_t)r1 * 4 * H, 4 * H, w.Hl[l], H, bt->prev_row + r1, grads + lw_hh(l), R - r1, H,
                        4 * H);
        }
        if (l > 0) {   // gradient into the hidden states of the layer below: dH_{l-1} = dG_l W_ih_l (the forget-gate part of
            GemmHole dh{};                                         // the sum is zero for first-time-step rows)
            dh.b[0] = GemmBlock{0, B, 0, H, H, 2 * H};
            PP_TRY(linear_dgrad(w.Gl[l], 4 * H, P + lw_ih(l), dH_other, H, nullptr, nullptr, 0, R, H, 4 * H, false, st, nullptr,
                                &dh));
            std::swap(dH_cur, dH_other);
        }
    }
    // LSTM parameter gradients are queued; layer 0's data gradient follows
    if (!ff) {
    // (wq is flushed at the very end, together with the observe-embedding weight gradients)
    // dX = dG W_ih, then scatter into the embedding tables / sample embeddings / observe embedding. Nobody reads the
    // previous-variable columns of first-time-step rows (no previous variable, no parameter behind them).
    // The forget-gate part of the summation over the gates is zero for those rows.
    if (compact) {
        // dX[:, :c2] = dG W_ih[:, :c2] (observe-embedding and sample-embedding columns; the table columns need no per-row
        // gradient: their parameter gradients follow from the column sums of dG per address group)
        const GemmHole dx_unused{{{0, B, cz0, nx, 0, 4 * H}, {0, B, 0, nx, H, 2 * H}}};
        pp_gemm_args g{};
        g.A = w.G; g.lda = 4 * H;
        g.B = P + net->w_ih; g.ldb = I; g.b_kmajor = 1;
        g.C = w.dX; g.ldc = w.i4;
        g.M = R; g.N = nx; g.K = 4 * H;
        if (panel) {
            dx_splits = 1;      // complete rows, written by the panel launch
        } else if (dx_partials) {
            // ~3 slabs per split (the K loop is short either way; more splits = more workgroups streaming dG)
            const int nslab = 4 * H / 32;
            dx_splits = std::max(1, std::min(DX_SPLITS, nslab / 4));
            GemmExt x{};
            x.split_stride = (int64_t)R * w.i4;
            x.force_splits = dx_splits;
            PP_TRY(gemm_f32(&g, st, &dx_unused, &x));
        } else {
        g.accumulate = 1;   // dX was cleared by the gather kernel
        g.split_k = 1;
        PP_TRY(gemm_f32(&g, st, &dx_unused));
        }
        if (!fused_bwd) {   // group sums of dG by current / previous address (the fused dH epilogue produced the former)
            for (int a = 0; a < net->n_addr; ++a) {
                const int g0 = bt->grp_off[a], n = bt->grp_off[a + 1] - g0;
                if (n > 0) cs.push_back(ColsumJob{w.G, 4 * H, bt->grp_rows + g0, n, 4 * H, w.gsum + (int64_t)a * 2 * 4 * H, nullptr});
                const int q0 = bt->nxt_off[a], m = bt->nxt_off[a + 1] - q0;
                if (m > 0) cs.push_back(ColsumJob{w.G, 4 * H, bt->nxt_rows + q0, m, 4 * H, w.gsum + ((int64_t)a * 2 + 1) * 4 * H, nullptr});
            }
        }
    } else {
    const GemmHole dx_unused{{{0, B, cz0, cz1, 0, 4 * H}, {0, B, 0, I, H, 2 * H}}};
    PP_TRY(linear_dgrad(w.G, 4 * H, P + net->w_ih, w.dX, w.i4, nullptr, nullptr, 0, R, I, 4 * H, true, st, nullptr,
                        &dx_unused));   // dX was cleared by the gather kernel
    }
    const int c1 = net->e_obs, c2 = c1 + net->smp_dim, c3 = c2 + net->dtype_dim, c4 = c3 + net->addr_dim,
              c5 = c4 + net->dtype_dim;
    for (int a = 0; a < net->n_addr && !compact; ++a) {
        const pp_addr& ad = net->addrs[a];
        const int g0 = bt->grp_off[a], n = bt->grp_off[a + 1] - g0;
        if (n > 0) {  // rows where `a` is the current address
            cs.push_back(ColsumJob{w.dX + c4, w.i4, bt->grp_rows + g0, n, net->dtype_dim, grads + ad.dtype_emb, nullptr});
            cs.push_back(ColsumJob{w.dX + c5, w.i4, bt->grp_rows + g0, n, net->addr_dim, grads + ad.addr_emb, nullptr});
        }
        const int q0 = bt->nxt_off[a], m = bt->nxt_off[a + 1] - q0;
        if (m > 0) {  // rows whose previous variable has address `a`
            cs.push_back(ColsumJob{w.dX + c2, w.i4, bt->nxt_rows + q0, m, net->dtype_dim, grads + ad.dtype_emb, nullptr});
            cs.push_back(ColsumJob{w.dX + c3, w.i4, bt->nxt_rows + q0, m, net->addr_dim, grads + ad.addr_emb, nullptr});
        }
    }
    if (T > 1 && det)
        PP_TRY(sample_embed_bwd_det(net, P, bt->value, bt->prev_row, bt->nxt_rows, bt->nxt_off, w.dX, w.i4, grads, st));
    else if (T > 1)
        PP_TRY(sample_embed_bwd(net, P, bt->value, bt->addr, bt->prev_row, bt->row_off[1], R, w.dX, w.i4, grads, st));
    }   // !ff
    // Column sums + weight-gradient leaves. Compact rows: the jobs that turn the group sums of dG into the table-column
    // gradients (and, after the fused cell backward, the LSTM bias gradients and the loss) ride behind the tiles of the
    // grouped weight-gradient launch; for a single-statement batch so do the column sums themselves (nothing in the
    // launch depends on anything else in it), which removes the separate column-sum launch.
    auto reduce_and_flush = [&]() -> int {
        if (!compact && ff && !det) {
            // FeedForward network: no column sum depends on another launch's group sums, so they all ride behind the
            // weight-gradient tiles (and the loss is finalised there): one launch less per step
            static const int ff_ride = env_flag("PP_AUX_COLSUM", 1);
            int n_live = 0;
            for (const auto& j : cs) n_live += (j.n_rows > 0 && j.n_cols > 0) ? 1 : 0;
            if (ff_ride && n_live <= AUX_MAX_COLSUM && !wq.empty()) {
                AuxJobs aux{};
                for (const auto& j : cs)
                    if (j.n_rows > 0 && j.n_cols > 0) aux.cs[aux.n_colsum++] = j;
                aux.fin = LossFinalize{fin_acc, w.flag, B > 0 ? 1.0f / (float)B : 0.0f, loss_out, status_out};
                aux_layout(aux, false);
                return flush_wgrads(st, true, &aux);
            }
        }
        if (!compact) {
            if (ff)   // (LSTM: the first cell launch of the backward pass finalises the loss)
                PP_TRY(colsum_multi(cs.data(), (int)cs.size(), st, fin_acc, w.flag, B, loss_out, status_out));
            else
                PP_TRY(colsum_multi(cs.data(), (int)cs.size(), st));
            return flush_wgrads(st, true);
        }
        AuxJobs aux{};
        static const int cs_ride_env = env_flag("PP_AUX_COLSUM", 1);
        int n_live = 0;
        for (const auto& j : cs) n_live += (j.n_rows > 0 && j.n_cols > 0) ? 1 : 0;
        if (fused_bwd && cs_ride_env && n_live <= AUX_MAX_COLSUM) {
            for (const auto& j : cs)
                if (j.n_rows > 0 && j.n_cols > 0) aux.cs[aux.n_colsum++] = j;
        } else {
            PP_TRY(colsum_multi(cs.data(), (int)cs.size(), st));
        }
        aux.gsum = w.gsum; aux.W = P + net->w_ih; aux.dW = grads + net->w_ih; aux.ldw = I;
        aux.params = P; aux.grads = grads; aux.at = net->addr_table;
        aux.N = 4 * H; aux.c2 = c2x; aux.c4 = c2x + ne_x; aux.nd = net->dtype_dim; aux.ne = ne_x; aux.n_addr = net->n_addr;
        for (int q = 0; q < 32; ++q) aux.present[q] = abias.present[q];
        aux.all_present = 0;
        if (fused_bwd) {
            aux.db_ih = grads + net->b_ih; aux.db_hh = grads + net->b_hh;
            aux.fin = LossFinalize{fin_acc, w.flag, B > 0 ? 1.0f / (float)B : 0.0f, loss_out, status_out};
        }
        aux_layout(aux, true);
        return flush_wgrads(st, true, &aux);
    };
    // observe embedding backward
    if (obs_fused_supported(net)) {
        // dE (sum over the trace's time steps of dX, masked by the last ReLU) and the data gradients of the whole stack
        // in one fused launch; weight gradients join the grouped MFMA launch; bias gradients are column sums of the same
        // buffers
        const int64_t dhs = (int64_t)B * w.maxohid4;
        if (!obs_tail)      // (single-statement batches on the panel kernel: done in its tail)
            PP_TRY(obs_embed_dgrad_fused(net, P, B, w.obs_h, w.cat, w.f1, dXs, ldxs, bt->row_off_dev, T, w.E, w.dE, w.dF1, w.dCat,
                                         w.dObsH, dhs, st, dx_splits, (int64_t)R * w.i4));
        const int e = net->e_obs;
        queue_wgrad(wq, w.dE, w.e4, w.f1, w.e4, nullptr, grads + net->fin_w1, B, e, e);
        queue_wgrad(wq, w.dF1, w.e4, w.cat, w.e4, nullptr, grads + net->fin_w0, B, e, e);
        cs.push_back(ColsumJob{w.dE, w.e4, nullptr, B, e, grads + net->fin_b1, nullptr});
        cs.push_back(ColsumJob{w.dF1, w.e4, nullptr, B, e, grads + net->fin_b0, nullptr});
        int ci = 0, co = 0;
        for (int o = 0; o < net->n_obs; ++o) {
            const int in = net->obs_in[o], hid = net->obs_hid[o], out = net->obs_out[o];
            float* dHo = w.dObsH + (int64_t)o * dhs;
            queue_wgrad(wq, w.dCat + co, w.e4, w.obs_h[o], w.ohid4[o], nullptr, grads + net->obs_w1[o], B, hid, out);
            // dW0[:, k] = sum_b dh[b, :] obs[b, k]: a handful of input columns -> weighted column sums, not a GEMM
            for (int k = 0; k < in; ++k)
                cs.push_back(ColsumJob{dHo, w.ohid4[o], nullptr, B, hid, grads + net->obs_w0[o] + k, nullptr,
                                       bt->obs + ci + k, bt->obs_width, in});
            cs.push_back(ColsumJob{w.dCat + co, w.e4, nullptr, B, out, grads + net->obs_b1[o], nullptr});
            cs.push_back(ColsumJob{dHo, w.ohid4[o], nullptr, B, hid, grads + net->obs_b0[o], nullptr});
            ci += in;
            co += out;
        }
        return reduce_and_flush();
    }
    PP_TRY(obs_grad(dXs, ldxs, bt->row_off_dev, T, B, net->e_obs, w.E, w.e4, w.dE, w.e4, st));   // dE, ReLU mask applied
    PP_TRY(reduce_and_flush());
    const int e = net->e_obs;
    PP_TRY(linear_wgrad(w.dE, w.e4, w.f1, w.e4, nullptr, grads + net->fin_w1, grads + net->fin_b1, nullptr, B, e, e, st));
    PP_TRY(linear_dgrad(w.dE, w.e4, P + net->fin_w1, w.dF1, w.e4, nullptr, w.f1, w.e4, B, e, e, false, st,
                        det ? nullptr : grads + net->fin_b0));
    if (det) PP_TRY(colsum_f32(w.dF1, w.e4, nullptr, B, e, grads + net->fin_b0, nullptr, st));
    PP_TRY(linear_wgrad(w.dF1, w.e4, w.cat, w.e4, nullptr, grads + net->fin_w0, nullptr, nullptr, B, e, e, st));
    PP_TRY(linear_dgrad(w.dF1, w.e4, P + net->fin_w0, w.dCat, w.e4, nullptr, w.cat, w.e4, B, e, e, false, st));
    int ci = 0, co = 0;
    for (int o = 0; o < net->n_obs; ++o) {
        // backward through EmbeddingFeedForward(num_layers = depth): dz of layer l (ReLU mask already applied) gives
        // dW_l = dz^T x_l, db_l = colsum dz, and dz of layer l - 1 = (dz W_l) * [x_l > 0]
        const int depth = net->obs_depth[o] ? net->obs_depth[o] : 2;
        const float* dz = w.dCat + co;
        int64_t lddz = w.e4;
        float* scratch[2] = {w.dObsH, w.dObsH2};
        for (int l = depth - 1; l >= 0; --l) {
            const int out = l == depth - 1 ? net->obs_out[o] : net->obs_hid[o];
            const int in = l == 0 ? net->obs_in[o] : net->obs_hid[o];
            const float* x = l == 0 ? bt->obs + ci : w.obs_hl[o][l - 1];
            const int64_t ldx = l == 0 ? bt->obs_width : w.ohid4[o];
            const int64_t wl = net->obs_depth[o] ? net->obs_w[o][l] : (l == 0 ? net->obs_w0[o] : net->obs_w1[o]);
            const int64_t bl = net->obs_depth[o] ? net->obs_b[o][l] : (l == 0 ? net->obs_b0[o] : net->obs_b1[o]);
            PP_TRY(linear_wgrad(dz, lddz, x, ldx, nullptr, grads + wl, grads + bl, nullptr, B, in, out, st));
            if (l > 0) {
                float* dprev = scratch[l & 1];
                PP_TRY(linear_dgrad(dz, lddz, P + wl, dprev, w.ohid4[o], nullptr, x, ldx, B, in, out, false, st));
                dz = dprev;
                lddz = w.ohid4[o];
            }
        }
        ci += net->obs_in[o];
        co += net->obs_out[o];
    }
    return 0;
}

}  // namespace pp

// ---------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------
extern "C" {

int pp_abi_version(void) { return PP_ABI_VERSION; }
const char* pp_last_error(void) { return pp::last_error(); }

int pp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

size_t pp_ic_workspace_bytes(const pp_net* net, int32_t n_traces, int32_t n_rows) {
    if (!net || pp::check_net(net) != 0) return 0;
    pp::Workspace w;
    pp::carve(net, n_traces, n_rows, nullptr, 0, w);
    return w.bytes;
}

int pp_ic_loss(const pp_net* net, const pp_batch* batch, const float* params, float* grads, void* workspace,
               size_t workspace_bytes, float* loss_out, int32_t* status_out, float* lp_out, int32_t flags, void* stream) {
    return pp::ic_loss(net, batch, params, grads, workspace, workspace_bytes, loss_out, status_out, lp_out, flags,
                       pp::as_stream(stream));
}

int pp_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n_params,
                 const int32_t* chunk_tensor, const float* active, int32_t* tensor_step, int32_t* arrived, int32_t n_tensors,
                 float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, int32_t flags,
                 const int32_t* skip, void* stream) {
    // kernel class 3 of the in-stream timing: the optimizer pass over the flat buffers (HBM bound): reads of params, grads
    // and both moments, writes of params, both moments and the cleared gradients - 8 x 4 bytes per (padded) parameter
    pp::prof_begin(3, pp::as_stream(stream));
    const int rc = pp::adam_step(params, grads, exp_avg, exp_avg_sq, n_params, chunk_tensor, active, tensor_step, arrived,
                                 n_tensors, lr, beta1, beta2, eps, weight_decay, grad_scale, flags, skip, pp::as_stream(stream));
    pp::prof_end(3, ((flags & PP_ADAM_ZERO_GRADS) ? 32.0 : 28.0) * (double)n_params, pp::as_stream(stream));
    return rc;
}

int pp_colsum_f32(const float* X, int64_t ldx, const int32_t* row_idx, int32_t n_rows, int32_t n_cols, float* out,
                  float* out2, void* stream) {
    return pp::colsum_f32(X, ldx, row_idx, n_rows, n_cols, out, out2, pp::as_stream(stream));
}

int pp_lstm_input_gather(const pp_net* net, const float* params, const float* E, const int32_t* trace_of_row,
                         const float* value, const int32_t* addr, const int32_t* prev_row, int32_t n_rows, float* X,
                         int64_t ldx, void* stream) {
    if (!net) return PP_EINVAL;
    return pp::lstm_input_gather(net, params, E, pp::round4(net->e_obs), trace_of_row, value, addr, prev_row, -1, -1, n_rows,
                                 X, ldx, pp::as_stream(stream));
}

int pp_lstm_cell_fwd(float* G, const float* c_prev, float* c, float* h, int32_t n, int32_t H, void* stream) {
    return pp::lstm_cell_fwd(G, c_prev, c, h, n, H, pp::as_stream(stream));
}

int pp_lstm_cell_bwd(float* G, const float* c_prev, const float* c, const float* dh, float* dc_carry, int32_t n,
                     int32_t n_next, int32_t H, void* stream) {
    return pp::lstm_cell_bwd(G, c_prev, c, dh, dc_carry, n, n_next, H, nullptr, nullptr, pp::as_stream(stream));
}

int pp_head_logprob(int32_t kind, const float* y, int64_t ldy, const int32_t* rows, const float* value, const float* prior,
                    int32_t n, int32_t n_out, float grad_scale, float* lp_out, float* dy, float* loss_acc,
                    int32_t* nonfinite, void* stream) {
    return pp::head_logprob(kind, y, ldy, rows, value, prior, n, n_out, grad_scale, lp_out, dy, loss_acc, nonfinite,
                            pp::as_stream(stream));
}

// Shader-clock probe: one wave spins for `iters` dependent FMAs and records s_memtime (shader cycles) and
// s_memrealtime (100 MHz wall clock) before and after. out[0] = shader cycles, out[1] = wall ticks (10 ns).
__global__ void clock_probe_kernel(int iters, long long* out, float* sink) {
    float x = (float)threadIdx.x;
    const long long c0 = clock64();
    const long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) x = x * 1.0000001f + 0.5f;
    const long long c1 = clock64();
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = w1 - w0;
    }
    if (x == 123.456f) sink[0] = x;
}

int pp_debug_clock_probe(int32_t iters, long long* out /*dev [2]*/, float* sink /*dev [1]*/, void* stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, pp::as_stream(stream), iters, out, sink);
    PP_LAUNCH_CHECK("pp_debug_clock_probe");
    return 0;
}

int pp_debug_timeline(long long* buf /*dev [16] or NULL*/) {
    pp::g_timeline = buf;
    return 0;
}

int pp_prof_arm(int32_t which, int32_t max_samples) {
    for (auto e : pp::g_prof.ev0) (void)hipEventDestroy(e);
    for (auto e : pp::g_prof.ev1) (void)hipEventDestroy(e);
    pp::g_prof.ev0.clear();
    pp::g_prof.ev1.clear();
    pp::g_prof.flops.clear();
    pp::g_prof.used = 0;
    pp::g_prof.seen = 0;
    pp::g_prof.cur = false;
    pp::g_prof.which = -1;
    if (max_samples <= 0) return 0;
    pp::g_prof.ev0.resize(max_samples);
    pp::g_prof.ev1.resize(max_samples);
    pp::g_prof.flops.assign(max_samples, 0.0);
    for (int i = 0; i < max_samples; ++i) {
        if (hipEventCreate(&pp::g_prof.ev0[i]) != hipSuccess || hipEventCreate(&pp::g_prof.ev1[i]) != hipSuccess) {
            pp::set_error("pp_prof_arm: hipEventCreate failed");
            return PP_ENODEV;
        }
    }
    pp::g_prof.which = which;
    return 0;
}

int pp_prof_stride(int32_t stride) {
    pp::g_prof.stride = stride > 0 ? stride : 1;
    return 0;
}

int pp_prof_collect(float* ms_out, int32_t cap, int32_t* n_out, double* flops_out) {
    int n = std::min<int>(pp::g_prof.used, cap);
    for (int i = 0; i < n; ++i) {
        (void)hipEventSynchronize(pp::g_prof.ev1[i]);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, pp::g_prof.ev0[i], pp::g_prof.ev1[i]);
        ms_out[i] = ms;
        if (flops_out) flops_out[i] = pp::g_prof.flops[i];
    }
    if (n_out) *n_out = n;
    pp::g_prof.used = 0;
    return 0;
}

}  // extern "C"
