// Native training loop: the body of InferenceNetwork.optimize (pyprob/nn/inference_network.py:461-499) for a run of
// minibatches, entirely on the host side of the C ABI - per step: pack the minibatch straight from the dataset columns
// into a pinned staging slot (pp_pack_indexed), ONE asynchronous upload, pp_ic_loss (loss + backward), pp_adam_step. The
// Python host only decides which traces form which minibatch, the learning rates, and reads the losses back once per
// run. Measured motivation: a training step is 0.16 ms on the GPU, while Python spent 0.25-0.38 ms
// per step on packing calls, torch tensor wrappers and ctypes marshalling.
// The uploads share the compute stream on purpose: a dedicated copy stream with event hand-offs (upload i+1 under
// step i) was measured SLOWER on this runtime - 224 instead of 180 us per LSTM step, 191 instead of 104 us per
// FeedForward step (tools/train_loop_bench.py) - the cross-stream waits cost more than the ~10 us copy they hide.
// Packing in a second host thread (pack group g+1 while this thread enqueues group g) was measured too: +-2 us per
// step in an A/B on one box - packing (50 us) hides behind the GPU's step either way - so the loop stays single-threaded.
#include "common.hpp"

#include <algorithm>
#include <vector>

namespace pp {
void set_error(const char* fmt, ...);
int ic_loss(const pp_net* net, const pp_batch* bt, const float* P, float* grads, void* ws, size_t ws_bytes, float* loss_out,
            int32_t* status_out, float* lp_out, int flags, hipStream_t st);
int adam_step(float* params, float* grads, float* m, float* v, int64_t n_params, const int32_t* chunk_tensor,
              const float* active, int32_t* tensor_step, int32_t* arrived, int n_tensors, float lr, float beta1, float beta2,
              float eps, float wd, float gscale, int flags, const int32_t* skip, hipStream_t st);
int dp_world();
int dp_reduce_grads(float* grads_full, int64_t n_params, int n_tensors, const float* presence, const int32_t* status,
                    const int64_t* skip_off, const int64_t* skip_cnt, int n_skip, float* loss_out, int32_t* status_out,
                    hipStream_t st);
}  // namespace pp

// upload-complete events of the two staging halves; one training run at a time per process (like the reference's loop)
static hipEvent_t g_half_event[2];
static bool g_have_event[2] = {false, false};

extern "C" {

int64_t pp_train_slot_words(int32_t n_traces, int64_t n_rows, int32_t t_max, int32_t obs_width, int32_t n_addr,
                            int32_t n_tensors) {
    return ((pp_pack_words(n_traces, n_rows, t_max, obs_width, n_addr) + n_tensors + 63) / 64) * 64;
}

int pp_train_steps(const pp_net* net, const pp_train_buffers* tb, const pp_tensor_roles* roles,
                   const pp_shard_columns* shards, int32_t n_shards, const int64_t* first, int32_t obs_width,
                   const int64_t* ids, const int64_t* step_off, int32_t n_steps, const float* lr, float beta1, float beta2,
                   float eps, float weight_decay, int32_t grads_clean, int64_t* addr_iterations, void* stream) {
    if (!(net && tb && roles && shards && first && ids && step_off && lr) || n_steps < 0) {
        pp::set_error("pp_train_steps: null pointer");
        return PP_EINVAL;
    }
    if (!(tb->params && tb->grads && tb->exp_avg && tb->exp_avg_sq && tb->chunk_tensor && tb->tensor_step && tb->adam_scratch &&
          tb->workspace && tb->staging && tb->device_batch && tb->loss_ring && tb->status_ring) ||
        tb->n_slots < 1 || tb->n_slots > 64 || tb->slot_words <= 0 || tb->n_tensors <= 0) {
        pp::set_error("pp_train_steps: incomplete pp_train_buffers");
        return PP_EINVAL;
    }
    hipStream_t st = pp::as_stream(stream);
    const int n_addr = net->n_addr, n_tensors = tb->n_tensors;
    // data parallel: every step all-reduces [grads | presence | loss | flag] between backward and Adam (csrc/dp.hip)
    const bool dp = tb->dp_world >= 1;
    if (dp && (pp::dp_world() != tb->dp_world || tb->dp_n_skip < 0 || tb->dp_n_skip > 4)) {
        pp::set_error("pp_train_steps: dp_world %d but the communicator has %d ranks (pp_dp_init)", tb->dp_world, pp::dp_world());
        return PP_EINVAL;
    }
    float* const dp_tail = tb->grads + net->n_params;   // [n_tensors presence | loss | flag] (data parallel only)
    // Minibatches are uploaded in GROUPS: the slots form two halves; a group of up to n_slots/2 steps is packed into one
    // half and crosses PCIe as ONE copy, then its steps are enqueued back to back. A copy on the compute stream costs
    // ~15-20 us of idle GPU per occurrence (it waits for the previous step, then pays the DMA latency): per step that
    // was 11 % of a 160 us step, per group of 8 it is 1.5 %. Group sizes ramp 1, 2, 4, ... so the GPU starts at once.
    const int G = std::max(1, tb->n_slots / 2);
    hipEvent_t* ev = g_half_event;
    bool* have_ev = g_have_event;
    struct Packed { pp_pack_info info; int64_t words; };
    std::vector<Packed> pk(G);
    int rc = 0;
    for (int i = 0, gi = 0; i < n_steps && rc == 0; ++gi) {
        const int half = tb->n_slots >= 2 ? (gi & 1) : 0;
        const int gs = std::min({G, 1 << std::min(gi, 16), n_steps - i});
        if (!have_ev[half]) {
            if (hipEventCreateWithFlags(&ev[half], hipEventDisableTiming) != hipSuccess) {
                pp::set_error("pp_train_steps: hipEventCreate failed");
                rc = PP_EHIP;
                break;
            }
            have_ev[half] = true;
        } else {
            (void)hipEventSynchronize(ev[half]);   // the copy that last read this staging half has completed
        }
        float* host0 = static_cast<float*>(tb->staging) + (int64_t)half * G * tb->slot_words;
        float* dev0 = static_cast<float*>(tb->device_batch) + (int64_t)half * G * tb->slot_words;
        for (int k = 0; k < gs && rc == 0; ++k) {
            float* host = host0 + (int64_t)k * tb->slot_words;
            const int32_t* hosti = reinterpret_cast<const int32_t*>(host);
            const int64_t n = step_off[i + k + 1] - step_off[i + k];
            if (n <= 0 || n > INT32_MAX) {
                pp::set_error("pp_train_steps: step %d has %lld traces", i + k, (long long)n);
                rc = PP_EINVAL;
                break;
            }
            pp_pack_info& info = pk[k].info;
            rc = pp_pack_indexed(shards, n_shards, first, ids + step_off[i + k], (int32_t)n, obs_width, n_addr, host,
                                 tb->slot_words - n_tensors, &info);
            if (rc != 0) break;
            pk[k].words = info.src_row + info.n_rows;   // end of the packed buffer (pp_pack_words)
            // presence map of this minibatch (which tensors have grad != None in the reference): after the packed words
            const int32_t* goff = hosti + info.grp_off;
            const int32_t* noff = hosti + info.nxt_off;
            float* act = host + pk[k].words;
            for (int t = 0; t < n_tensors; ++t) {
                const int role = roles->role[t];
                bool on = role & 4;
                for (int q = roles->off[t]; q < roles->off[t + 1] && !on; ++q) {
                    const int a = roles->addr[q];
                    on = ((role & 1) && goff[a + 1] > goff[a]) || ((role & 2) && noff[a + 1] > noff[a]);
                }
                act[t] = on ? 1.0f : 0.0f;
            }
            if (addr_iterations)   // inference_network_lstm.py:198
                for (int a = 0; a < n_addr; ++a) addr_iterations[a] += goff[a + 1] > goff[a];
        }
        if (rc != 0) break;
        const size_t bytes = ((size_t)(gs - 1) * tb->slot_words + pk[gs - 1].words + n_tensors) * 4;
        if (hipMemcpyAsync(dev0, host0, bytes, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipEventRecord(ev[half], st) != hipSuccess) {
            pp::set_error("pp_train_steps: upload failed: %s", hipGetErrorString(hipGetLastError()));
            rc = PP_EHIP;
            break;
        }
        for (int k = 0; k < gs && rc == 0; ++k, ++i) {
            const pp_pack_info& info = pk[k].info;
            const int32_t* hosti = reinterpret_cast<const int32_t*>(host0 + (int64_t)k * tb->slot_words);
            float* dev = dev0 + (int64_t)k * tb->slot_words;
            const int32_t* devi = reinterpret_cast<const int32_t*>(dev);
            pp_batch bt{};
            bt.n_traces = (int32_t)info.n_traces; bt.n_rows = (int32_t)info.n_rows; bt.t_max = (int32_t)info.t_max;
            bt.obs_width = obs_width;
            bt.n_active = hosti + info.n_active; bt.row_off = hosti + info.row_off;
            bt.grp_off = hosti + info.grp_off; bt.nxt_off = hosti + info.nxt_off;
            bt.obs = dev + info.obs; bt.value = dev + info.value; bt.prior = dev + info.prior;
            bt.addr = devi + info.addr; bt.prev_row = devi + info.prev_row; bt.grp_rows = devi + info.grp_rows;
            bt.trace = devi + info.trace; bt.row_off_dev = devi + info.row_off_dev; bt.nxt_rows = devi + info.nxt_rows;
            const int flags = PP_LOSS_BACKWARD | ((i == 0 && !grads_clean) ? PP_LOSS_ZERO_GRADS : 0);
            rc = pp::ic_loss(net, &bt, tb->params, tb->grads, tb->workspace, tb->workspace_bytes,
                             dp ? dp_tail + n_tensors : tb->loss_ring + i, tb->status_ring + i, nullptr, flags, st);
            if (rc != 0) break;
            if (dp) {   // presence map + flag into the tail, ONE grouped all-reduce, mean loss / any-rank flag back to the rings
                rc = pp::dp_reduce_grads(tb->grads, net->n_params, n_tensors, dev + pk[k].words, tb->status_ring + i,
                                         tb->dp_skip_off, tb->dp_skip_cnt, tb->dp_n_skip, tb->loss_ring + i, tb->status_ring + i,
                                         st);
                if (rc != 0) break;
            }
            // Adam checks the step's non-finite flag itself (`skip`) and clears the gradients it consumed (the next
            // step's zero_grad, :486); data parallel: the merged presence map, the summed gradients / world
            rc = pp::adam_step(tb->params, tb->grads, tb->exp_avg, tb->exp_avg_sq, net->n_params, tb->chunk_tensor,
                               dp ? dp_tail : dev + pk[k].words, tb->tensor_step, tb->adam_scratch, n_tensors, lr[i], beta1, beta2,
                               eps, weight_decay, dp ? 1.0f / (float)tb->dp_world : 1.0f, PP_ADAM_ZERO_GRADS,
                               tb->status_ring + i, st);
        }
    }
    // Returns with the last groups still queued: the caller plans its next run meanwhile. The staging halves stay
    // guarded by the two (process-wide) events - the next call waits on them before rewriting a half, and
    // pp_train_sync() waits for both (before the staging memory is freed or reused for something else).
    return rc;
}

// The same loop body for minibatches that are ALREADY resident in HBM (pp_batch structs whose device arrays are in place):
// per step zero_grad -> _loss -> backward -> [all-reduce] -> Adam (pyprob/nn/inference_network.py:486-496), n_steps steps
// in one C call. active[i] = device presence map [n_tensors] of step i's minibatch. Losses / flags go to the rings.
int pp_train_resident(const pp_net* net, const pp_train_buffers* tb, const pp_batch* const* batches, const float* const* active,
                      int32_t n_steps, const float* lr, float beta1, float beta2, float eps, float weight_decay,
                      int32_t grads_clean, void* stream) {
    if (!(net && tb && batches && active && lr) || n_steps < 0) {
        pp::set_error("pp_train_resident: null pointer");
        return PP_EINVAL;
    }
    if (!(tb->params && tb->grads && tb->exp_avg && tb->exp_avg_sq && tb->chunk_tensor && tb->tensor_step && tb->adam_scratch &&
          tb->workspace && tb->loss_ring && tb->status_ring) || tb->n_tensors <= 0) {
        pp::set_error("pp_train_resident: incomplete pp_train_buffers");
        return PP_EINVAL;
    }
    hipStream_t st = pp::as_stream(stream);
    const int n_tensors = tb->n_tensors;
    const bool dp = tb->dp_world >= 1;
    if (dp && (pp::dp_world() != tb->dp_world || tb->dp_n_skip < 0 || tb->dp_n_skip > 4)) {
        pp::set_error("pp_train_resident: dp_world %d but the communicator has %d ranks (pp_dp_init)", tb->dp_world, pp::dp_world());
        return PP_EINVAL;
    }
    float* const dp_tail = tb->grads + net->n_params;
    for (int i = 0; i < n_steps; ++i) {
        if (!batches[i] || !active[i]) {
            pp::set_error("pp_train_resident: step %d has no batch / presence map", i);
            return PP_EINVAL;
        }
        const int flags = PP_LOSS_BACKWARD | ((i == 0 && !grads_clean) ? PP_LOSS_ZERO_GRADS : 0);
        int rc = pp::ic_loss(net, batches[i], tb->params, tb->grads, tb->workspace, tb->workspace_bytes,
                             dp ? dp_tail + n_tensors : tb->loss_ring + i, tb->status_ring + i, nullptr, flags, st);
        if (rc != 0) return rc;
        if (dp) {
            rc = pp::dp_reduce_grads(tb->grads, net->n_params, n_tensors, active[i], tb->status_ring + i, tb->dp_skip_off,
                                     tb->dp_skip_cnt, tb->dp_n_skip, tb->loss_ring + i, tb->status_ring + i, st);
            if (rc != 0) return rc;
        }
        rc = pp::adam_step(tb->params, tb->grads, tb->exp_avg, tb->exp_avg_sq, net->n_params, tb->chunk_tensor,
                           dp ? dp_tail : active[i], tb->tensor_step, tb->adam_scratch, n_tensors, lr[i], beta1, beta2, eps,
                           weight_decay, dp ? 1.0f / (float)tb->dp_world : 1.0f, PP_ADAM_ZERO_GRADS, tb->status_ring + i, st);
        if (rc != 0) return rc;
    }
    return 0;
}

int pp_train_sync(void) {
    for (int h = 0; h < 2; ++h)
        if (g_have_event[h] && hipEventSynchronize(g_half_event[h]) != hipSuccess) {
            pp::set_error("pp_train_sync: %s", hipGetErrorString(hipGetLastError()));
            return PP_EHIP;
        }
    return 0;
}

}  // extern "C"
