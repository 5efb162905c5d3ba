// Data-parallel gradient exchange from inside the C loop: RCCL called directly on the flat gradient buffer
// (_distributed_sync_grad, pyprob/nn/inference_network.py:296-333, is one all-reduce per tensor plus a presence map and the
// loss; here it is ONE grouped launch over [flat grads | presence map | loss | non-finite flag] minus the ranges that are
// zero on every rank by construction). librccl is dlopen'ed from the path the host passes (the one torch already loaded),
// so the library has no link-time dependency on it; the communicator is this library's own (bootstrap: rank 0's
// ncclUniqueId travels to the other ranks over torch.distributed, pyprob_amd/parallel.py).
#include "common.hpp"

#include <dlfcn.h>
#include <string.h>

#include <rccl/rccl.h>

namespace pp {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclComm_t comm = nullptr;
    int world = 0, rank = -1;
};
static Rccl g_rccl;

static int rccl_load(const char* path) {
    if (g_rccl.lib) return 0;
    PP_CHECK_ARG(path && path[0], "pp_dp: the path of librccl.so is required");
    void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        set_error("pp_dp: dlopen(%s) failed: %s", path, dlerror());
        return PP_ENODEV;
    }
#define PP_SYM(field, name)                                                       \
    *reinterpret_cast<void**>(&g_rccl.field) = dlsym(h, name);                    \
    if (!g_rccl.field) {                                                          \
        set_error("pp_dp: %s has no symbol %s", path, name);                      \
        return PP_ENODEV;                                                         \
    }
    PP_SYM(GetUniqueId, "ncclGetUniqueId")
    PP_SYM(CommInitRank, "ncclCommInitRank")
    PP_SYM(CommDestroy, "ncclCommDestroy")
    PP_SYM(AllReduce, "ncclAllReduce")
    PP_SYM(GroupStart, "ncclGroupStart")
    PP_SYM(GroupEnd, "ncclGroupEnd")
    PP_SYM(GetErrorString, "ncclGetErrorString")
#undef PP_SYM
    g_rccl.lib = h;
    return 0;
}

#define PP_HIP(call)                                                                           \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess) {                                                               \
            set_error("pp_dp: %s failed: %s", #call, hipGetErrorString(e__));                  \
            return PP_EHIP;                                                                    \
        }                                                                                      \
    } while (0)

#define PP_NCCL(call, what)                                                                    \
    do {                                                                                       \
        ncclResult_t r__ = (call);                                                             \
        if (r__ != ncclSuccess) {                                                              \
            set_error("pp_dp: %s failed: %s", what, g_rccl.GetErrorString(r__));               \
            return PP_EHIP;                                                                    \
        }                                                                                      \
    } while (0)

int dp_world() { return g_rccl.comm ? g_rccl.world : 0; }

// in-place sum over the ranks of the pieces [off[i], off[i] + cnt[i]) of `base`: one grouped launch
int dp_allreduce_pieces(float* base, const int64_t* off, const int64_t* cnt, int n, hipStream_t st) {
    PP_CHECK_ARG(g_rccl.comm, "pp_dp: no communicator (pp_dp_init)");
    PP_CHECK_ARG(base && off && cnt && n >= 1, "pp_dp_allreduce: bad argument");
    if (n > 1) PP_NCCL(g_rccl.GroupStart(), "ncclGroupStart");
    for (int i = 0; i < n; ++i) {
        if (cnt[i] <= 0) continue;
        PP_NCCL(g_rccl.AllReduce(base + off[i], base + off[i], (size_t)cnt[i], ncclFloat32, ncclSum, g_rccl.comm, st), "ncclAllReduce");
    }
    if (n > 1) PP_NCCL(g_rccl.GroupEnd(), "ncclGroupEnd");
    return 0;
}

// ---- bucket 0 under the rest of the weight-gradient launch ------------------------------------------------
// The reference reduces its gradients bucket by bucket (`distributed_num_buckets`, inference_network.py:300-325) so that a
// framework with an asynchronous backward can overlap; here every gradient of a step used to complete in ONE launch - the last one
// before the exchange - so nothing overlapped. With an overlap range set (pp_dp_overlap: the LSTM layer's gradient tensors) the
// backward pass issues the weight-gradient launch in two parts (engine.hip flush_wgrads): part A completes the range, its
// all-reduce starts on a SIDE stream behind an event, part B (the heads' and the observe embedding's products) runs under it on
// the main stream; dp_reduce_grads then reduces everything BUT the range and makes the main stream wait for the side stream
// before Adam. Same collectives in the same order on every rank (the decision is host-side state set identically by all ranks).
struct Overlap {
    int64_t off[2] = {0, 0}, cnt[2] = {0, 0};
    int n = 0;
    hipStream_t side = nullptr;
    hipEvent_t ready = nullptr, done = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};      // stats: side begin / end, main begin / end of the rest, join
    bool pending = false, stats = false, stats_valid = false;
};
static Overlap g_ov;

bool dp_overlap_hull(int64_t* lo, int64_t* hi) {
    if (!g_rccl.comm || g_ov.n <= 0) return false;
    *lo = g_ov.off[0];
    *hi = g_ov.off[g_ov.n - 1] + g_ov.cnt[g_ov.n - 1];
    return true;
}

static int overlap_objects() {
    if (g_ov.side) return 0;
    PP_HIP(hipStreamCreateWithFlags(&g_ov.side, hipStreamNonBlocking));
    PP_HIP(hipEventCreateWithFlags(&g_ov.ready, hipEventDisableTiming));
    PP_HIP(hipEventCreateWithFlags(&g_ov.done, hipEventDisableTiming));
    for (auto& e : g_ov.ev) PP_HIP(hipEventCreate(&e));
    return 0;
}

// part A of the weight-gradient launch is queued on `st`: start the all-reduce of the range behind it, on the side stream
int dp_bucket0_issue(float* grads_full, hipStream_t st) {
    if (!g_rccl.comm || g_ov.n <= 0 || g_ov.pending) return 0;
    PP_TRY(overlap_objects());
    PP_HIP(hipEventRecord(g_ov.ready, st));
    PP_HIP(hipStreamWaitEvent(g_ov.side, g_ov.ready, 0));
    if (g_ov.stats) PP_HIP(hipEventRecord(g_ov.ev[0], g_ov.side));
    PP_TRY(dp_allreduce_pieces(grads_full, g_ov.off, g_ov.cnt, g_ov.n, g_ov.side));
    if (g_ov.stats) PP_HIP(hipEventRecord(g_ov.ev[1], g_ov.side));
    PP_HIP(hipEventRecord(g_ov.done, g_ov.side));
    g_ov.pending = true;
    return 0;
}

// tail of the flat gradient buffer before / after the exchange: [n_tensors presence flags | loss | non-finite flag]
__global__ void dp_tail_pre_kernel(float* __restrict__ tail, const float* __restrict__ presence, int n_tensors,
                                   const int32_t* __restrict__ status) {
    for (int t = threadIdx.x; t < n_tensors; t += blockDim.x)
        if (presence) tail[t] = presence[t];
    if (threadIdx.x == 0) tail[n_tensors + 1] = status[0] != 0 ? 1.0f : 0.0f;   // every rank skips a step that ANY rank flagged
}
__global__ void dp_tail_post_kernel(const float* __restrict__ tail, int n_tensors, float inv_world, float* __restrict__ loss_out,
                                    int32_t* __restrict__ status_out) {
    if (threadIdx.x == 0) {
        if (loss_out) loss_out[0] = tail[n_tensors] * inv_world;               // mean loss (inference_network.py:327-333)
        if (status_out) status_out[0] = tail[n_tensors + 1] > 0.0f ? 1 : 0;
    }
}

// grads_full = [n_params gradients | n_tensors presence flags | loss | flag]; skip ranges (sorted, inside the gradients)
// stay out of the exchange. presence: this step's local presence map, or nullptr when the tail already holds it.
int dp_reduce_grads(float* grads_full, int64_t n_params, int n_tensors, const float* presence, const int32_t* status,
                    const int64_t* skip_off, const int64_t* skip_cnt, int n_skip, float* loss_out, int32_t* status_out,
                    hipStream_t st) {
    PP_CHECK_ARG(g_rccl.comm && grads_full && status && n_params > 0 && n_tensors >= 0 && n_skip >= 0 && n_skip <= 6 && (!g_ov.pending || n_skip <= 4),
                 "pp_dp_reduce_grads: bad argument");
    float* tail = grads_full + n_params;
    hipLaunchKernelGGL(dp_tail_pre_kernel, dim3(1), dim3(256), 0, st, tail, presence, n_tensors, status);
    PP_LAUNCH_CHECK("pp_dp_reduce_grads (tail)");
    // what stays out of THIS collective: the caller's ranges (zero on every rank) and, when its all-reduce is already running
    // on the side stream, the overlap range - merged in ascending order
    int64_t so[8], sc[8];
    int ns = 0;
    {
        int i = 0, j = 0;
        const int nj = g_ov.pending ? g_ov.n : 0;
        while (i < n_skip || j < nj) {
            const bool take_i = j >= nj || (i < n_skip && skip_off[i] <= g_ov.off[j]);
            so[ns] = take_i ? skip_off[i] : g_ov.off[j];
            sc[ns] = take_i ? skip_cnt[i] : g_ov.cnt[j];
            ++ns;
            if (take_i) ++i; else ++j;
        }
    }
    int64_t off[10], cnt[10];
    int n = 0;
    int64_t pos = 0;
    const int64_t total = n_params + n_tensors + 2;
    for (int i = 0; i < ns; ++i) {
        PP_CHECK_ARG(so[i] >= pos && so[i] + sc[i] <= n_params, "pp_dp_reduce_grads: bad skip range (or one that overlaps the overlap range)");
        if (so[i] > pos) { off[n] = pos; cnt[n] = so[i] - pos; ++n; }
        pos = so[i] + sc[i];
    }
    off[n] = pos; cnt[n] = total - pos; ++n;
    const bool stats = g_ov.pending && g_ov.stats;
    if (stats) PP_HIP(hipEventRecord(g_ov.ev[2], st));
    PP_TRY(dp_allreduce_pieces(grads_full, off, cnt, n, st));
    if (stats) PP_HIP(hipEventRecord(g_ov.ev[3], st));
    if (g_ov.pending) {      // Adam reads the overlap range too: the main stream waits for the side stream's collective
        PP_HIP(hipStreamWaitEvent(st, g_ov.done, 0));
        if (stats) {
            PP_HIP(hipEventRecord(g_ov.ev[4], st));
            g_ov.stats_valid = true;
        }
        g_ov.pending = false;
    }
    hipLaunchKernelGGL(dp_tail_post_kernel, dim3(1), dim3(64), 0, st, tail, n_tensors, 1.0f / (float)g_rccl.world, loss_out,
                       status_out);
    PP_LAUNCH_CHECK("pp_dp_reduce_grads (tail)");
    return 0;
}

}  // namespace pp

extern "C" {

using pp::set_error;

int pp_dp_unique_id(const char* rccl_path, void* id_out) {
    PP_TRY(pp::rccl_load(rccl_path));
    PP_CHECK_ARG(id_out, "pp_dp_unique_id: null pointer");
    ncclUniqueId id;
    if (pp::g_rccl.GetUniqueId(&id) != ncclSuccess) {
        pp::set_error("pp_dp_unique_id: ncclGetUniqueId failed");
        return PP_EHIP;
    }
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int pp_dp_init(const char* rccl_path, const void* unique_id, int32_t rank, int32_t world) {
    PP_TRY(pp::rccl_load(rccl_path));
    PP_CHECK_ARG(unique_id && world >= 1 && rank >= 0 && rank < world, "pp_dp_init: bad argument");
    if (pp::g_rccl.comm) {
        pp::set_error("pp_dp_init: a communicator already exists (pp_dp_destroy first)");
        return PP_EINVAL;
    }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    ncclResult_t r = pp::g_rccl.CommInitRank(&comm, world, id, rank);   // collective over the ranks; the current HIP device
    if (r != ncclSuccess) {
        pp::set_error("pp_dp_init: ncclCommInitRank failed: %s", pp::g_rccl.GetErrorString(r));
        return PP_EHIP;
    }
    pp::g_rccl.comm = comm;
    pp::g_rccl.world = world;
    pp::g_rccl.rank = rank;
    return 0;
}

int pp_dp_world(void) { return pp::dp_world(); }

int pp_dp_destroy(void) {
    if (pp::g_rccl.comm) {
        (void)pp::g_rccl.CommDestroy(pp::g_rccl.comm);
        pp::g_rccl.comm = nullptr;
        pp::g_rccl.world = 0;
        pp::g_rccl.rank = -1;
    }
    pp::g_ov.n = 0;
    pp::g_ov.pending = false;
    pp::g_ov.stats = pp::g_ov.stats_valid = false;
    return 0;
}

int pp_dp_overlap(const int64_t* off, const int64_t* cnt, int32_t n) {
    PP_CHECK_ARG(n >= 0 && n <= 2 && (n == 0 || (off && cnt)), "pp_dp_overlap: at most two ranges");
    if (pp::g_ov.pending) {
        pp::set_error("pp_dp_overlap: an overlapped all-reduce is in flight (call between steps)");
        return PP_EINVAL;
    }
    for (int i = 0; i < n; ++i) {
        PP_CHECK_ARG(off[i] >= 0 && cnt[i] > 0 && (i == 0 || off[i] >= off[i - 1] + cnt[i - 1]), "pp_dp_overlap: ranges must ascend");
        pp::g_ov.off[i] = off[i];
        pp::g_ov.cnt[i] = cnt[i];
    }
    pp::g_ov.n = n;
    return 0;
}

int pp_dp_overlap_stats(int32_t arm, float* us_out) {
    // arm = 1 / 0: HIP event pairs around the collectives of the following steps on / off (an event record costs the stream
    // ~1.5 us: bench.py arms it for a short pass of its own). us_out [3] (or NULL): the LAST overlapped step's
    // {range's all-reduce on the side stream, the rest's all-reduce on the main stream, what the main stream then still waited for
    // the side stream}; the call synchronises on those events. Returns 1 when us_out was filled.
    int filled = 0;
    if (us_out && pp::g_ov.stats_valid) {
        PP_HIP(hipEventSynchronize(pp::g_ov.ev[4]));
        PP_HIP(hipEventSynchronize(pp::g_ov.ev[1]));
        float a = 0, b = 0, c = 0;
        PP_HIP(hipEventElapsedTime(&a, pp::g_ov.ev[0], pp::g_ov.ev[1]));
        PP_HIP(hipEventElapsedTime(&b, pp::g_ov.ev[2], pp::g_ov.ev[3]));
        PP_HIP(hipEventElapsedTime(&c, pp::g_ov.ev[3], pp::g_ov.ev[4]));
        us_out[0] = a * 1e3f; us_out[1] = b * 1e3f; us_out[2] = c * 1e3f;
        filled = 1;
    }
    pp::g_ov.stats = arm != 0;
    if (!arm) pp::g_ov.stats_valid = false;
    return filled;
}

int pp_dp_allreduce(float* base, const int64_t* off, const int64_t* cnt, int32_t n, void* stream) {
    return pp::dp_allreduce_pieces(base, off, cnt, n, pp::as_stream(stream));
}

int pp_dp_reduce_grads(float* grads_full, int64_t n_params, int32_t n_tensors, const float* presence, const int32_t* status,
                       const int64_t* skip_off, const int64_t* skip_cnt, int32_t n_skip, float* loss_out,
                       int32_t* status_out, void* stream) {
    return pp::dp_reduce_grads(grads_full, n_params, n_tensors, presence, status, skip_off, skip_cnt, n_skip, loss_out, status_out,
                               pp::as_stream(stream));
}

}  // extern "C"
