/*
 * pyprob_amd.h -- C ABI of libpyprob_amd.so: the MI355X (gfx950) inference-compilation engine for pyprob.
 *
 * This is the drop-in boundary (SURVEY.md §8b seam B3). pyprob itself is 100% Python and has no FFI for this
 * path, so every entry point below cites the reference *Python* code it replaces (paths relative to the pyprob
 * v1.5.0 tree). The reference-side binding a maintainer would add is a ctypes stub; see INTEGRATION.md.
 *
 * Conventions
 *   - plain C: pointers, sizes, POD structs; no torch/HIP types. `stream` is a hipStream_t passed as void*.
 *   - every pointer marked "dev" is device (HBM) memory owned by the caller (the host uses the PyTorch-ROCm
 *     caching allocator as plumbing); the library never allocates or frees device memory and never
 *     synchronises the stream unless the function says so.
 *   - all arithmetic is fp32 (reference: util._dtype = torch.float, pyprob/util.py:29). GEMMs use the exact
 *     fp32 MFMA (v_mfma_f32_32x32x2_f32), everything else fp32 VALU.
 *   - return value: 0 on success, a hipError_t (>0) from a failed launch, or a PP_E* code (<0). No C++
 *     exceptions cross this ABI. pp_last_error() returns a static, thread-local description.
 *   - threading: call from one host thread per stream (pyprob's trace runtime is single-threaded,
 *     pyprob/state.py:13-27).
 *
 * Packed trace batch ("step-major ragged"): the B traces of a minibatch are sorted by controlled length,
 * longest first; row r = row_off[t] + b holds time step t of trace b (b < n_active[t]). Sub-batches of the
 * reference (pyprob/nn/dataset.py:21-37: traces with an identical address sequence) are runs of traces in this
 * order; because nn.LSTM weights are shared and h0=c0=0 for every trace, running all sub-batches through one
 * step-major pass is arithmetically the per-sub-batch loop of InferenceNetworkLSTM._loss
 * (pyprob/nn/inference_network_lstm.py:138-219).
 */
#ifndef PYPROB_AMD_H
#define PYPROB_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_ABI_VERSION 14
#define PP_MAX_OBS 8
#define PP_MAX_LSTM_DEPTH 4
#define PP_MAX_OBS_DEPTH 4

/* error codes (negative; positive values are hipError_t) */
#define PP_EINVAL   (-1)   /* bad argument (shape, alignment, null pointer) */
#define PP_ENOSPACE (-2)   /* workspace too small */
#define PP_ENODEV   (-3)   /* no gfx950 device / kernels not loadable */
#define PP_EHIP     (-4)   /* a HIP runtime call failed (pp_last_error() has the text) */

/* proposal head kinds (pyprob/nn/inference_network_lstm.py:52-66) */
#define PP_HEAD_NORMAL_MIXTURE       0  /* ProposalNormalNormalMixture: prior Normal(mean, stddev) */
#define PP_HEAD_TRUNCNORMAL_MIXTURE  1  /* ProposalUniformTruncatedNormalMixture: prior Uniform(low, high) */
#define PP_HEAD_CATEGORICAL          2  /* ProposalCategoricalCategorical: prior Categorical(C) */
#define PP_HEAD_BERNOULLI            4  /* ProposalBernoulliBernoulli (proposal_bernoulli_bernoulli.py:16-20), n_out = 1. In the
                                           training loss the reference broadcasts probs [n, 1] against values [n] to an
                                           [n, n] log_prob matrix per sub-batch step and sums it (inference_network_lstm.py:
                                           195-202); the row's `prior` pair carries (n, sum of the n values) of its sub-batch
                                           step so that the row term sum_j log Bernoulli(v_j; p_row) is computed in place */
#define PP_HEAD_POISSON_TN_MIXTURE   3  /* ProposalPoissonTruncatedNormalMixture: prior Poisson; (p0, p1) = (low, high) = (0, 40) */

int         pp_abi_version(void);
const char* pp_last_error(void);
/* number of visible HIP devices whose arch is gfx950; 0 if none (never throws) */
int         pp_device_count(void);

/* ------------------------------------------------------------------------------------------------------
 * Network description. Offsets are in floats into one flat fp32 parameter buffer (and the identically laid
 * out gradient / Adam-moment buffers). Tensor shapes are the reference's.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct pp_addr {
    int32_t kind;        /* PP_HEAD_* */
    int32_t n_out;       /* head output width: 3K (mixtures) or C (categorical) */
    int32_t hid;         /* head hidden width int((H+n_out)/2), embedding_feedforward.py:26 */
    int32_t smp_in;      /* sample-embedding input width: 1, or C (one-hot) for categorical */
    int32_t dtype_id;    /* index of the distribution-type embedding this address uses */
    int32_t _pad;
    int64_t addr_emb;    /* [addr_dim]            _layers_address_embedding.<addr> */
    int64_t dtype_emb;   /* [dtype_dim]           _layers_distribution_type_embedding.<DistName> */
    int64_t smp_w;       /* [smp_dim, smp_in]     _layers_sample_embedding.<addr>._layers.0.weight */
    int64_t smp_b;       /* [smp_dim] */
    int64_t w1, b1;      /* [hid, H], [hid]       _layers_proposal.<addr>._ff._layers.0 (FeedForward network: [hid, e_obs]) */
    int64_t w2, b2;      /* [n_out, hid], [n_out] _layers_proposal.<addr>._ff._layers.1 */
} pp_addr;

typedef struct pp_net {
    int32_t n_obs;                    /* observables with a FEEDFORWARD depth-2 embedding (inference_network.py:110-118) */
    int32_t obs_in[PP_MAX_OBS];       /* flattened input width of observable o */
    int32_t obs_hid[PP_MAX_OBS];      /* int((in+out)/2) */
    int32_t obs_out[PP_MAX_OBS];      /* embedding dim of observable o */
    int64_t obs_w0[PP_MAX_OBS], obs_b0[PP_MAX_OBS], obs_w1[PP_MAX_OBS], obs_b1[PP_MAX_OBS];
    int32_t e_obs;                    /* sum of obs_out */
    int32_t smp_dim, addr_dim, dtype_dim;
    int64_t fin_w0, fin_b0, fin_w1, fin_b1;   /* _layers_observe_embedding_final (e_obs -> e_obs -> e_obs) */
    int32_t lstm_in;                  /* I = e_obs + smp_dim + 2*(addr_dim+dtype_dim), inference_network_lstm.py:30 */
    int32_t lstm_dim;                 /* H; 0 = InferenceNetworkFeedForward (inference_network_feedforward.py): no LSTM and
                                         no address / sample embeddings (lstm_in = 0, their offsets unused); the proposal
                                         layers [hid, e_obs] read the observe embedding of the trace */
    int64_t w_ih, w_hh, b_ih, b_hh;   /* _layers_lstm.{weight_ih,weight_hh,bias_ih,bias_hh}_l0 */
    int32_t n_addr;
    int32_t n_dtype;
    const pp_addr* addrs;             /* host array [n_addr] */
    const int64_t* addr_table;        /* dev  [n_addr, PP_ADDR_TABLE_COLS] same records for per-row dispatch */
    int64_t n_params;                 /* floats in the flat buffer (incl. padding) */
    int32_t lstm_depth;               /* layers of nn.LSTM(I, H, depth) (inference_network_lstm.py:31); 0 is read as 1 */
    int32_t _pad2;
    /* _layers_lstm.{weight_ih,weight_hh,bias_ih,bias_hh}_l<k> of every layer; entry 0 repeats w_ih .. b_hh above. Layer
     * k >= 1 reads the hidden states of layer k-1: weight_ih_l<k> is [4H, H] */
    int64_t lstm_w_ih[PP_MAX_LSTM_DEPTH], lstm_w_hh[PP_MAX_LSTM_DEPTH], lstm_b_ih[PP_MAX_LSTM_DEPTH], lstm_b_hh[PP_MAX_LSTM_DEPTH];
    /* observe embeddings of a depth other than 2 (EmbeddingFeedForward(num_layers = depth), embedding_feedforward.py:22-33,
     * inference_network.py:110-118): depth 1 = Linear(in, out); depth d >= 2 = Linear(in, hid), (d - 2) x Linear(hid, hid),
     * Linear(hid, out), ReLU after every layer. obs_depth[o] = 0 is read as 2 (obs_w0 .. obs_b1 above). For other depths
     * layer l of observable o is obs_w[o][l] / obs_b[o][l] (_layers_observe_embedding.<name>._layers.<l>). */
    int32_t obs_depth[PP_MAX_OBS];
    int64_t obs_w[PP_MAX_OBS][PP_MAX_OBS_DEPTH], obs_b[PP_MAX_OBS][PP_MAX_OBS_DEPTH];
} pp_net;

/* columns of the device address table */
#define PP_ADDR_TABLE_COLS 8
#define PP_AT_KIND 0
#define PP_AT_SMP_IN 1
#define PP_AT_ADDR_EMB 2
#define PP_AT_DTYPE_EMB 3
#define PP_AT_SMP_W 4
#define PP_AT_SMP_B 5
#define PP_AT_N_OUT 6
#define PP_AT_RESERVED 7

typedef struct pp_batch {
    int32_t n_traces;            /* B  (Batch.size, pyprob/nn/dataset.py:24) */
    int32_t n_rows;              /* R = sum of controlled trace lengths */
    int32_t t_max;               /* longest controlled trace */
    int32_t obs_width;           /* sum of obs_in */
    const int32_t* n_active;     /* host [t_max]   traces with length > t */
    const int32_t* row_off;      /* host [t_max+1] prefix sum of n_active */
    const int32_t* grp_off;      /* host [n_addr+1] rows grouped by address: group a = grp_rows[grp_off[a]:grp_off[a+1]] */
    const float*   obs;          /* dev [B, obs_width] observed values, trace order = packed order */
    const float*   value;        /* dev [R] sampled value of every controlled variable (category index as float) */
    const float*   prior;        /* dev [R, 2] (mean, stddev) | (low, high) | Poisson: (0, 40) | Bernoulli: (n, n1) of the
                                    row's sub-batch step (PP_HEAD_BERNOULLI) | Categorical: unused */
    const int32_t* addr;         /* dev [R] address id of the row */
    const int32_t* prev_row;     /* dev [R] row of the previous time step of the same trace, -1 at t = 0 */
    const int32_t* grp_rows;     /* dev [R] row ids sorted by address id */
    const int32_t* trace;        /* dev [R] trace index b of the row */
    const int32_t* row_off_dev;  /* dev [t_max+1] copy of row_off */
    const int32_t* nxt_off;      /* host [n_addr+1] rows grouped by the address of their PREVIOUS variable */
    const int32_t* nxt_rows;     /* dev [R - B] row ids (t >= 1) sorted by previous address id */
} pp_batch;

/* ------------------------------------------------------------------------------------------------------
 * Host-side minibatch packing (no device access): ragged trace-major columns -> the step-major pp_batch layout.
 * Replaces Batch.__init__ (pyprob/nn/dataset.py:21-37) and the per-trace torch.stack / torch.cat of _loss
 * (pyprob/nn/inference_network_lstm.py:146-196). `out` receives ONE buffer of 4-byte words: the first
 * info->device_words words are what the device needs (upload them with one copy), the rest are the host-side arrays of
 * pp_batch; every info field below is a word offset into `out`.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct pp_pack_info {
    int64_t n_traces, n_rows, t_max, device_words;
    int64_t obs, value, prior, addr, prev_row, grp_rows, trace, row_off_dev, nxt_rows;   /* device part */
    int64_t n_active, row_off, grp_off, nxt_off;                                         /* host part of pp_batch */
    int64_t order;      /* [B] packed trace position -> input trace index (traces are sorted longest-first) */
    int64_t src_row;    /* [R] packed row -> row of the trace-major input */
} pp_pack_info;

/* words `out` must hold; t_max = longest trace */
int64_t pp_pack_words(int32_t n_traces, int64_t n_rows, int32_t t_max, int32_t obs_width, int32_t n_addr);

/* trace_len [B]; addr_ids / values [R] and prior [R, prior_width] trace-major; obs [B, obs_width]. Errors: a trace of
 * length zero ("Trace of length zero.", dataset.py:28-29), an address id outside [0, n_addr), a short buffer. */
int pp_pack_ragged(const int32_t* trace_len, const int32_t* addr_ids, const float* values, const float* prior,
                   int32_t prior_width, const float* obs, int32_t n_traces, int32_t obs_width, int32_t n_addr,
                   void* out, int64_t out_words, pp_pack_info* info);

/* The same packing straight from the columns of a packed on-disk trace dataset (pyprob_amd/dataset.py: memory-mapped
 * .npy columns, one struct per shard): minibatch = the traces with global indices ids[0..n_ids). `first` [n_shards+1]
 * is the global index of every shard's first trace. addr_remap (or NULL) maps a shard's address ids to the network's;
 * an unmapped address (-1) is an error (the caller polymorphs first, inference_network_lstm.py:150-152). Replaces
 * OfflineDataset.__getitem__ + the DataLoader collate (pyprob/nn/dataset.py:197-205, 262). */
typedef struct pp_shard_columns {
    const int32_t* trace_len;   /* [n] */
    const int64_t* row_off;     /* [n+1] */
    const float*   obs;         /* [n, obs_width] */
    const float*   value;       /* [rows] */
    const float*   prior;       /* [rows, 2] */
    const int32_t* addr;        /* [rows] shard-local address ids */
    const int32_t* addr_remap;  /* [shard addresses] -> network address id, or NULL for identity */
} pp_shard_columns;

int pp_pack_indexed(const pp_shard_columns* shards, int32_t n_shards, const int64_t* first, const int64_t* ids,
                    int32_t n_ids, int32_t obs_width, int32_t n_addr, void* out, int64_t out_words, pp_pack_info* info);

/* ------------------------------------------------------------------------------------------------------
 * Whole-path entry points
 * ---------------------------------------------------------------------------------------------------- */

/* Bytes of scratch HBM pp_ic_loss needs for a batch of at most n_traces traces / n_rows rows.
 * The workspace must be ZERO-FILLED once when it is allocated and must not be written by the caller afterwards: its
 * first bytes (a size that depends on the network only) hold the exchange areas of the LSTM tail kernels
 * (csrc/lstm_tail.hip), whose tagged words must never be mistaken for fresh ones. Everything else in it is scratch
 * of a single call. */
size_t pp_ic_workspace_bytes(const pp_net* net, int32_t n_traces, int32_t n_rows);

#define PP_LOSS_BACKWARD   1   /* also write dLoss/dparams into `grads` */
#define PP_LOSS_ZERO_GRADS 2   /* zero `grads` (n_params floats) first: optimizer.zero_grad(), inference_network.py:486 */
#define PP_LOSS_KEEP_LP    4   /* write the per-row proposal log_prob (row order) to lp_out */

/*
 * InferenceNetworkLSTM._loss(batch) (+ loss.backward()):  pyprob/nn/inference_network_lstm.py:136-220,
 * pyprob/nn/inference_network.py:487,493. Computes
 *     loss = -(1/B) * sum_rows log q(value | lstm state)          (written to loss_out[0], dev)
 * with -inf log-probs replaced by log(1e-8) (:207-213); status_out[0] (dev int32) is set non-zero if the loss
 * is still non-finite (the reference then skips the batch, :216-217). With PP_LOSS_BACKWARD, `grads` receives
 * the gradient of every parameter in the layout of `params` (tensors that do not participate stay zero).
 */
int pp_ic_loss(const pp_net* net, const pp_batch* batch, const float* params /*dev*/, float* grads /*dev or NULL*/,
               void* workspace /*dev*/, size_t workspace_bytes, float* loss_out /*dev [1]*/,
               int32_t* status_out /*dev [1]*/, float* lp_out /*dev [R] or NULL*/, int32_t flags, void* stream);

/*
 * torch.optim.Adam.step() over the flat buffer (pyprob/nn/inference_network.py:348,496), per-tensor skipping
 * of parameters whose grad is None (tensors that did not take part in the loss) and per-tensor step counts.
 *   chunk_tensor  dev [n_params/1024]  tensor id owning each 1024-float chunk (tensors are padded to 1024 and laid
 *                                      out in id order: the ids ascend)
 *   active        dev [n_tensors]      float >0 -> tensor has a gradient this step (DP: all-reduced presence map,
 *                                      pyprob/nn/inference_network.py:300-315)
 *   tensor_step   dev [n_tensors]      int32 Adam step count per tensor, incremented here when active
 *   scratch       dev [PP_ADAM_SCRATCH * n_tensors] int32, opaque, owned by the optimizer state: ZERO before the
 *                                      first call and whenever chunk_tensor changes (two-level arrival counters - the
 *                                      last chunk of a tensor to finish advances its step count - the cached
 *                                      chunk run of each tensor, and word PP_ADAM_SEEN: non-zero once the tensor had a
 *                                      non-zero gradient. While it is 0 the tensor's moments are taken to be zero and
 *                                      all-zero gradient chunks are skipped: a caller that writes exp_avg / exp_avg_sq
 *                                      itself (checkpoint load) sets the word of every tensor to 1)
 *   grad_scale    1/world_size for data-parallel averaging (:324-325), else 1
 *   flags         PP_ADAM_ZERO_GRADS: clear every consumed gradient chunk (optimizer.zero_grad() of the next step,
 *                 inference_network.py:486); the next pp_ic_loss can then run without PP_LOSS_ZERO_GRADS
 *   skip          dev int32[1] or NULL: when non-zero the call does nothing - pass pp_ic_loss's status_out to skip a
 *                 batch whose loss is not finite (inference_network_lstm.py:216-217) without reading it on the host
 * One launch (bias corrections are derived per chunk from the tensor's step count).
 */
#define PP_ADAM_ZERO_GRADS 1
#define PP_ADAM_SCRATCH 1056
#define PP_ADAM_SEEN 1027
int pp_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n_params,
                 const int32_t* chunk_tensor, const float* active, int32_t* tensor_step, int32_t* scratch,
                 int32_t n_tensors, float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                 int32_t flags, const int32_t* skip, void* stream);

/*
 * optimizer.step() for torch.optim.SGD(lr, momentum, nesterov, weight_decay) - Optimizer.SGD of
 * InferenceNetwork._create_optimizer (pyprob/nn/inference_network.py:349-350, nesterov=True there) - over the same flat
 * buffers as pp_adam_step (chunk_tensor / active / grad_scale / flags / skip mean the same):
 *   g = grad * grad_scale + weight_decay * p;  buf = momentum * buf + g;  p -= lr * (nesterov ? g + momentum * buf : buf)
 * momentum_buf (dev [n_params], may be NULL when momentum == 0) must be ZERO before a parameter's first step: that is
 * torch's `buf = clone(grad)` of a first step. Tensors with active[t] == 0 are not touched (no decay, buffer unchanged).
 * One launch.
 */
int pp_sgd_step(float* params, float* grads, float* momentum_buf, int64_t n_params, const int32_t* chunk_tensor,
                const float* active, int32_t n_tensors, float lr, float momentum, int32_t nesterov, float weight_decay,
                float grad_scale, int32_t flags, const int32_t* skip, void* stream);

/*
 * The LARC wrapper of Optimizer.ADAM_LARC / SGD_LARC (pyprob/nn/optimizer_larc.py:72-107; inference_network.py:351-352
 * constructs it with the defaults trust_coefficient 0.002, clip = 1, eps 1e-8, epsilon 1/16000): per tensor with a
 * gradient,  local = (|p| != 0 && |g| != 0) ? trust * |p| / (|g| + weight_decay * |p| + eps) : epsilon,
 * adaptive = clip ? min(local / lr, 1) : local,  and IN PLACE  grad = (grad * grad_scale + weight_decay * p) * adaptive
 * (|g| is the norm of grad * grad_scale: the reference divides by the world size before the optimizer runs, :324-325).
 * The wrapped optimizer then steps with weight_decay = 0 and grad_scale = 1 (optimizer_larc.py:81,105-107):
 *   pp_larc_scale(..., lr, wd, 1/world, ...);  pp_adam_step / pp_sgd_step(..., lr, ..., 0.0f, 1.0f, ...).
 * scratch: dev, PP_LARC_SCRATCH_FLOATS(n_params, n_tensors) floats, contents irrelevant. Three launches (chunk partial sums
 * of squares, stored; per-tensor norms in fp64 in a fixed order; the rescaling pass): bit-reproducible. skip as above.
 */
#define PP_LARC_SCRATCH_FLOATS(n_params, n_tensors) (2 * ((n_params) / 1024) + (n_tensors))
int pp_larc_scale(const float* params, float* grads, int64_t n_params, const int32_t* chunk_tensor, const float* active,
                  int32_t n_tensors, float lr, float weight_decay, float grad_scale, float trust_coefficient, float eps,
                  float epsilon, int32_t clip, float* scratch, const int32_t* skip, void* stream);

/*
 * The body of the training loop (InferenceNetwork.optimize, pyprob/nn/inference_network.py:461-499: next minibatch ->
 * zero_grad -> _loss -> backward -> optimizer.step) for a RUN of minibatches whose traces come from packed dataset
 * columns, without returning to the caller between steps. Per step i: the traces ids[step_off[i] .. step_off[i+1]) are
 * packed (pp_pack_indexed) into a pinned staging slot together with the minibatch's presence map, uploaded with one
 * asynchronous copy, and pp_ic_loss(PP_LOSS_BACKWARD) + pp_adam_step(lr[i], PP_ADAM_ZERO_GRADS, skip = the step's own
 * non-finite flag) are enqueued on `stream`. loss_ring[i] / status_ring[i] receive the step's loss and non-finite flag
 * (a flagged step leaves the parameters untouched, :493-499). Single rank (data parallel runs keep the per-step
 * all-reduce on the caller's side). The caller polymorphs beforehand: an address unknown to `net` is PP_EINVAL.
 *   pp_tensor_roles  which tensors a minibatch gives a gradient to (grad is not None in the reference): tensor t is
 *                    present if role[t] & 4, or if one of its addresses addr[off[t] .. off[t+1]) occurs as a current
 *                    (role & 1) / previous (role & 2) variable of the minibatch (inference_network_lstm.py:168-171).
 *   staging          pinned host memory, device_batch device memory: n_slots slots of slot_words 4-byte words each,
 *                    slot_words >= pp_train_slot_words(...) of the largest step. The slots form two halves: a group of
 *                    up to n_slots/2 consecutive steps is packed into one half and uploaded with ONE copy (group
 *                    sizes ramp 1, 2, 4, ...); a half is rewritten only after its upload completed.
 *   workspace        pp_ic_workspace_bytes(net, most traces, most rows of any step)
 *   grads_clean      non-zero: `grads` is all zero on entry (left so by PP_ADAM_ZERO_GRADS); it is on return.
 *   addr_iterations  host [n_addr] or NULL: += 1 per step in which the address occurs (proposal_layer.
 *                    _total_train_iterations, inference_network_lstm.py:198)
 * Returns as soon as the last step is enqueued (the last uploads may still be pending: the next call waits for them
 * before it rewrites a staging half; pp_train_sync() waits for them explicitly - call it before freeing or reusing the
 * staging memory). One training run at a time per process. The losses are read by the caller.
 */
typedef struct pp_train_buffers {
    float* params; float* grads; float* exp_avg; float* exp_avg_sq;             /* dev [n_params] */
    const int32_t* chunk_tensor; int32_t* tensor_step; int32_t* adam_scratch;   /* dev, as in pp_adam_step */
    void* workspace; size_t workspace_bytes;                                     /* dev */
    void* staging; void* device_batch; int64_t slot_words;                       /* pinned host / dev */
    float* loss_ring; int32_t* status_ring;                                      /* dev [n_steps] */
    int32_t n_tensors; int32_t n_slots;
    /* data parallel (ABI 6): dp_world = size of the communicator of pp_dp_init (0: single rank). `grads` is then the flat
     * buffer [n_params | n_tensors presence flags | loss | non-finite flag]; between backward and Adam of every step the
     * loop all-reduces it (minus the dp_n_skip ranges [dp_skip_off, + dp_skip_cnt) that are zero on every rank), Adam divides
     * by dp_world and every rank skips a step that any rank flagged (pyprob/nn/inference_network.py:296-333, 448) */
    int32_t dp_world; int32_t dp_n_skip;
    int64_t dp_skip_off[4], dp_skip_cnt[4];
} pp_train_buffers;

typedef struct pp_tensor_roles {
    const int32_t* off;    /* host [n_tensors + 1] */
    const int32_t* addr;   /* host [off[n_tensors]] address ids */
    const int32_t* role;   /* host [n_tensors] bit 0: current variable, bit 1: previous variable, bit 2: always */
} pp_tensor_roles;

int64_t pp_train_slot_words(int32_t n_traces, int64_t n_rows, int32_t t_max, int32_t obs_width, int32_t n_addr,
                            int32_t n_tensors);

int pp_train_sync(void);
/* The same loop body (optimize, inference_network.py:486-496: zero_grad -> _loss -> backward -> [all-reduce] -> Adam) for
 * minibatches that are ALREADY resident in HBM: batches[i] = pp_batch of step i (device arrays in place, host arrays valid
 * for the duration of the call), active[i] = dev [n_tensors] presence map of that minibatch, lr[i] its learning rate; losses
 * and flags go to tb->loss_ring / status_ring [n_steps] (staging / device_batch of tb are not used). One C call enqueues
 * n_steps steps; tb->dp_world selects the data-parallel branch like in pp_train_steps. */
int pp_train_resident(const pp_net* net, const pp_train_buffers* tb, const pp_batch* const* batches,
                      const float* const* active, int32_t n_steps, const float* lr, float beta1, float beta2, float eps,
                      float weight_decay, int32_t grads_clean, void* stream);

int pp_train_steps(const pp_net* net, const pp_train_buffers* buffers, const pp_tensor_roles* roles,
                   const pp_shard_columns* shards, int32_t n_shards, const int64_t* first, int32_t obs_width,
                   const int64_t* ids, const int64_t* step_off, int32_t n_steps, const float* lr /*host [n_steps]*/,
                   float beta1, float beta2, float eps, float weight_decay, int32_t grads_clean,
                   int64_t* addr_iterations, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Importance sampling with the inference network, lock-step over N particles
 * (pyprob/state.py:203-219, pyprob/nn/inference_network_lstm.py:82-134, pyprob/trace.py:123-125)
 * ---------------------------------------------------------------------------------------------------- */

/* Bytes of scratch for pp_is_step with n particles. The workspace must be ZERO-FILLED once when it is allocated (it holds an
 * arrival counter of the first-statement kernels, which every launch leaves at zero); everything else in it is scratch of a
 * single call. */
size_t pp_is_workspace_bytes(const pp_net* net, int32_t n);

/*
 * InferenceNetwork._infer_init(observe): observe embedding of the single observation, batch 1
 * (pyprob/nn/inference_network.py:141-148). obs: dev [obs_width]; e_out: dev [e_obs].
 */
int pp_is_init(const pp_net* net, const float* params, const float* obs, float* e_out, void* workspace,
               size_t workspace_bytes, void* stream);

/*
 * One controlled `pyprob.sample` statement for n particles at once: _infer_step (LSTM step with carried
 * (h, c)) -> proposal head -> value ~ q -> log q(value).
 *   addr_id / prev_addr_id   address of this / the previous controlled variable (-1: first variable of the trace)
 *   e_obs_vec  dev [e_obs]   output of pp_is_init (shared by all particles)
 *   prev_value dev [n]       values sampled at the previous statement (ignored when prev_addr_id < 0)
 *   prior      dev [n,2] or [2] (prior_stride 0 = same prior parameters for every particle)
 *   h, c       dev [depth, n, H] LSTM state of every layer (layer k at offset k * n * H), updated in place
 *   state_rows 1 or n: rows of (h, c) that are valid on entry. The first statement of a trace (prev_addr_id < 0) ignores
 *              the state, evaluates the network for ONE row (every particle has the same input and zero state) and
 *              writes row 0 only: the caller passes state_rows = 1 to the next call, which turns the shared recurrent
 *              term into a bias row and writes all n rows; afterwards state_rows = n.
 *              FeedForward network (lstm_dim 0; inference_network_feedforward.py:52-66): no state - h, c, prev_value,
 *              prev_addr_id and state_rows are ignored (h, c may be NULL); the layer of addr_id is applied to e_obs_vec.
 *   value_in   dev [n] or NULL: if given, score these values instead of sampling (re-scoring / parity tests)
 *   value_out  dev [n]       sampled (or copied) values
 *   logq_out   dev [n]       proposal log_prob of value (Mixture.log_prob / Categorical.log_prob)
 *   seed, offset             Philox4x32-10 counter-based RNG: particle i uses counter (offset + i)
 */
int pp_is_step(const pp_net* net, const float* params, int32_t addr_id, int32_t prev_addr_id, int32_t n,
               const float* e_obs_vec, const float* prev_value, const float* prior, int32_t prior_stride,
               float* h, float* c, int32_t state_rows, const float* value_in, float* value_out, float* logq_out,
               uint64_t seed, uint64_t offset, void* workspace, size_t workspace_bytes, void* stream);

/* pp_is_step for the particles of a diverged control-flow path (pyprob/state.py:203-219 on a subset of the traces of
 * pyprob/model.py:59): particle i of the call owns row rows[i] of (h, c) (dev int64 [n], distinct rows; NULL = row i) - the
 * state is read and written in place through the list, every other per-particle array (prev_value, prior, value_in / out,
 * logq_out) is compact [n]. Statements after the first one only (prev_addr_id >= 0), and only where the fused statement
 * kernel exists (csrc/is_step_fused.hip: one-layer LSTM, H = 256, 512 or 1024; csrc/is_step_small.hip: H a multiple of 32 up to 256 with 1 ..
 * PP_MAX_LSTM_DEPTH layers; head at most 32 outputs wide); PP_EINVAL otherwise (the caller gathers / scatters the rows itself).
 * (ABI 14) state_rows with a row list: 1 = row 0 is everybody's previous state (one-layer LSTMs only), otherwise the ROW COUNT OF
 * ONE LAYER of the state buffer - (h, c) are [depth, state_rows, H], layer k at offset k * state_rows * H, and rows[i] < state_rows.
 * (Up to ABI 13 the callers passed n here, which made a path of exactly one particle read row 0 as a shared state.) pp_is_step_fused_supported(net, addr_id, n) != 0: the kernel exists AND is
 * the faster path for n particles (below ~3 000 rows a launch is one generation of latency-bound workgroups and pp_is_step's
 * chain of small launches wins; pp_is_step makes the same choice).
 * With that kernel a statement is ONE launch (+ one preparation launch): gates, LSTM cell, both head layers, the draw and
 * log q; the gate pre-activations never reach memory, (h, c) are read once and written once.
 * H = 32, 64 .. 256 (csrc/is_step_small.hip, one kernel at every n; H = 256 from two layers on): a workgroup owns 64 particles, wave (row block, unit block)
 * the four gates of 32 hidden units; the old hidden rows of every layer are staged in LDS once, layer l reads the fresh rows of
 * layer l - 1 from an LDS tile; the draw is the chain's own one-lane-per-particle tail.
 * H = 1024 (one layer): the statement is TWO launches - the LSTM step as one wide launch (two workgroups per 32 particles, half of
 * the hidden units each; gates on the accumulators, c in place, the new hidden rows through a scratch) and the head-only launch
 * (state rows, both head layers, draw, log q, whole-statement tail) - from 2 049 particles on (pp_is_step_fused_supported says so
 * per n); below that pp_is_step takes the chain of GEMM launches, pp_is_step_rows / pp_is_statement_rows still the two launches. */
int pp_is_step_rows(const pp_net* net, const float* params, int32_t addr_id, int32_t prev_addr_id, int32_t n,
                    const float* e_obs_vec, const float* prev_value, const float* prior, int32_t prior_stride,
                    float* h, float* c, int32_t state_rows, const int64_t* rows, const float* value_in, float* value_out,
                    float* logq_out, uint64_t seed, uint64_t offset, void* workspace, size_t workspace_bytes, void* stream);
int pp_is_step_fused_supported(const pp_net* net, int32_t addr_id, int32_t n);
/* The WHOLE statement of state.sample's IC branch (pyprob/state.py:203-219) for the particles of one control-flow path in ONE
 * launch of the fused statement kernel: particle i owns row r = rows[i] (NULL: r = i) of the state AND of the full-width
 * per-particle vectors - its previous value is prev_value_full[r], the drawn value goes to value_full[r], and
 * lw_full[r] += log p(v) - log q(v) with p the program's own prior (prior_kind 0: Normal(prior[0], prior[1]), 1: Uniform[prior[0],
 * prior[1]); prior_stride as in pp_is_step, rows of `prior` are compact). Replaces gather of the previous values + pp_is_step_rows
 * + scatter of the values + pp_logweight_accumulate + pp_axpy. Statements after the first one, mixture heads, where
 * pp_is_step_fused_supported says so; PP_EINVAL otherwise. */
int pp_is_statement_rows(const pp_net* net, const float* params, int32_t addr_id, int32_t prev_addr_id, int32_t n,
                         const float* e_obs_vec, const float* prev_value_full, const float* prior, int32_t prior_stride,
                         float* h, float* c, int32_t state_rows, const int64_t* rows, float* value_full, float* lw_full,
                         int32_t prior_kind, uint64_t seed, uint64_t offset, void* workspace, size_t workspace_bytes,
                         void* stream);

/* log p(value) of prior and likelihood terms, accumulated into the per-particle log-weight:
 *     lw[i] += scale * log_prob(dist(params_i); x_i)
 * (state.py:211-217: +prior, -proposal; state.py:147-149: +likelihood_importance * likelihood), evaluated in fp32 like
 * the torch.distributions classes behind pyprob/distributions/{normal,uniform,poisson,bernoulli,categorical}.py.
 *   kind: 0 Normal(p0=mean, p1=stddev), 1 Uniform(p0=low, p1=high) with support [low, high),
 *         3 Poisson(p0=rate), 4 Bernoulli(p0=probs) (p1 unused, may be NULL),
 *         5 Categorical: p0 = probs row(s) of p1_stride categories, row i at p0 + i * p0_stride (0 = one shared row);
 *           x = category index as float; p1 unused
 *   p0/p1/x strides: 0 broadcasts a single value, 1 reads per particle. lw and lp_out are optional (dev [n]). */
int pp_logweight_accumulate(int32_t kind, const float* p0, int32_t p0_stride, const float* p1, int32_t p1_stride,
                            const float* x, int32_t x_stride, float scale, float* lw /*dev [n]*/, float* lp_out,
                            int32_t n, void* stream);

/* The same term for the particles of ONE control-flow path of a lock-step run (the reference scores one trace at a time,
 * state.py:147-149, 211-217; a path is the set of particles that took the same branches):
 *     lw[r] += scale * log_prob(dist(params_r); x_r)   for r = rows[j], j < m
 * rows: dev int64 [m], ascending particle indices; strides as above (1 = indexed by the particle, not by j). */
int pp_logweight_accumulate_rows(int32_t kind, const float* p0, int32_t p0_stride, const float* p1, int32_t p1_stride,
                                 const float* x, int32_t x_stride, float scale, float* lw /*dev [n]*/,
                                 const int64_t* rows, int32_t m, void* stream);

/* dst[r] = src[r * src_stride] for r = rows[j], j < m (src_stride 0: one shared value): what a path's execution of the program
 * returned, into the call's per-particle result vector (model.py:66-74 collects one trace's result at a time). */
int pp_copy_rows(const float* src, int32_t src_stride, float* dst /*dev [n]*/, const int64_t* rows, int32_t m, void* stream);

/* A branch of the program taken per particle (`while s >= 1:` - the reference runs the Python condition per trace): stable
 * partition of a path's rows by a condition byte per particle.
 *   cond: dev uint8 [n] (torch.bool storage), indexed by the PARTICLE; rows: dev int64 [m] ascending, or NULL = particles 0..m-1
 *   rows_true / rows_false: dev int64 [m] each; the first counts[0] / counts[1] entries are written, ascending
 *   counts: dev int32 [2] = { rows whose condition is non-zero, the others };  scratch: dev int32 [PP_PARTITION_SCRATCH(m)]
 * Two launches, no host synchronisation (the caller reads `counts` when it needs the decision). */
#define PP_PARTITION_SCRATCH(m) (((m) + 1023) / 1024 + 1)
int pp_partition_rows(const uint8_t* cond, const int64_t* rows, int32_t m, int64_t* rows_true, int64_t* rows_false,
                      int32_t* counts, int32_t* scratch, void* stream);
/* (ABI 13) The same partition with the decision POLLED instead of copied back: counts = int32 [3] in host-mapped (pinned)
 * memory; the kernel stores counts[0], counts[1] and then, behind a system-scope fence, counts[2] = seq (non-zero; the caller
 * cleared the word before the call and spins on it). A branch of a lock-step run costs the host a poll of ~10 us instead of a
 * blocking 8-byte device-to-host copy of ~45 us (profiles/r04x_gumm_call_timeline_after.csv: the gap behind every
 * __amd_rocclr_copyBuffer). m must be > 0. */
int pp_partition_rows_polled(const uint8_t* cond, const int64_t* rows, int32_t m, int64_t* rows_true, int64_t* rows_false,
                             int32_t* counts /*host-mapped [3]*/, int32_t seq, int32_t* scratch, void* stream);

/* Up to four log-weight terms in ONE pass over the particles:
 *   lw[i] (+)= sum_t scale_t * term_t(i);  kind as in pp_logweight_accumulate, or 2: the value x itself (e.g. -log q)
 * overwrite != 0 starts from 0 instead of the current lw (saves the zero-fill). */
typedef struct pp_lw_term {
    int32_t kind, p0_stride, p1_stride, x_stride;
    const float *p0, *p1, *x;
    float scale;
} pp_lw_term;
int pp_logweight_terms(const pp_lw_term* terms, int32_t count, float* lw /*dev [n]*/, int32_t n, int32_t overwrite,
                       void* stream);

/* One posterior statement of a lock-step run in ONE pass over the particles (state.sample IC branch pyprob/state.py:203-219 +
 * the state.observe terms that follow it :118-155 + Trace.end's sum pyprob/trace.py:123-125 + the Empirical reductions
 * pyprob/distributions/empirical.py:298-309, 451-466, 758-766):
 *   pp_is_step_net  the network part of pp_is_step (LSTM step + proposal layer); the head outputs stay in the workspace.
 *   pp_is_fused     per particle: draw v ~ q (addr_id >= 0: the SHARED proposal of a trace's first statement, left in the
 *                   workspace by pp_is_step_net; mixture heads; same Philox stream as pp_is_step), lw (+)= -log q(v) +
 *                   sum_t scale_t term_t(i); value[i] = v. addr_id < 0: no draw, `value` is read. term_flags[t] bit 0 / 1 / 2:
 *                   p0 / p1 / x of term t IS the particle's value (e.g. the mean of the likelihood Normal(mu, s) after
 *                   mu = sample(...)). stats_out != NULL: the statistics of pp_is_stats over (lw, value) in the same pass.
 *                   At most 8 terms. 8 bytes per particle reach memory. */
int pp_is_step_net(const pp_net* net, const float* params, int32_t addr_id, int32_t prev_addr_id, int32_t n,
                   const float* e_obs_vec, const float* prev_value, float* h, float* c, int32_t state_rows, void* workspace,
                   size_t workspace_bytes, void* stream);
/* (ABI 12) pp_is_init + pp_is_step_net(addr_id, prev_addr_id = -1) of a trace's FIRST statement as ONE launch: the observe
 * embedding of the one shared row (InferenceNetwork._infer_init, pyprob/nn/inference_network.py:141-148) is computed inside the
 * launch of the LSTM step (_infer_step with prev_variable None, pyprob/nn/inference_network_lstm.py:82-134), the proposal layer's
 * workgroups ride behind the LSTM's in the same launch (environment PP_IS_FIRST=2: as a second launch), its outputs stay in the
 * workspace for pp_is_fused. `obs` (the observation vector, as for pp_is_init) may be host-mapped (pinned)
 * memory: the kernel reads it in place, a posterior call needs no copy launch. e_out: [round4(e_obs) + 8]: the embedding, then the
 * first 8 observation values (device copies: the x of the call's observe terms). (h, c): row 0 is written (state_rows = 1).
 * Bit-identical to the two calls it replaces. pp_is_first_statement_supported: one-layer LSTM, lstm_in <= 256, an embedding
 * the fused embedding kernels take. A statistics buffer of pp_is_fused / pp_is_stats may likewise be host-mapped memory: its
 * element [5] (the count) is stored last, behind a system-scope fence - a caller that set it negative may poll it. */
int pp_is_first_statement_supported(const pp_net* net, int32_t addr_id);
int pp_is_first_statement(const pp_net* net, const float* params, const float* obs, int32_t addr_id, float* e_out, float* h, float* c,
                          void* workspace, size_t workspace_bytes, void* stream);
int pp_is_fused(const pp_net* net, int32_t addr_id, int32_t n, const float* prior /*dev [2]*/, const pp_lw_term* terms,
                const int32_t* term_flags, int32_t n_terms, float* value /*dev [n]*/, float* lw /*dev [n]*/, int32_t overwrite,
                uint64_t seed, uint64_t offset, double* stats_out /*dev [6] or NULL*/, double* stats_scratch, void* workspace,
                size_t workspace_bytes, void* stream);

/* Prior draws of vectorised trace generation (the per-trace generator: pyprob/nn/dataset.py:50-62 with state.sample's
 * prior branch pyprob/state.py:278-290): out[i] ~ Normal(p0, p1) (kind 0) | Uniform[p0, p1) (kind 1), parameters shared
 * (stride 0) or per trace (stride 1); Philox4x32-10, counter offset + i, key seed, `stream_id` distinguishes statements. */
int pp_prior_draw(int32_t kind, const float* p0, int32_t p0_stride, const float* p1, int32_t p1_stride, int32_t n, uint64_t seed,
                  uint64_t offset, uint32_t stream_id, float* out /*dev [n]*/, void* stream);

/* lw[i] += scale * term[i] (e.g. -log q). */
int pp_axpy(float scale, const float* term, float* lw, int32_t n, void* stream);

/* Wavefront-reduced importance statistics over n particles (pyprob/distributions/empirical.py:298-309,
 * 451-466, 758-766): out (dev, double[6]) = { max lw, sum w, sum w^2, sum w*x, sum w*x^2, count finite } with
 * w = exp(lw - max lw) evaluated in fp64. ESS = (sum w)^2 / sum w^2. One pass over the particles (per-workgroup maxima, rescaled by a
 * one-workgroup combine); `scratch` dev >= PP_IS_STATS_SCRATCH doubles. */
#define PP_IS_STATS_SCRATCH 6144
int pp_is_stats(const float* lw, const float* x, int32_t n, double* out, double* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Individual kernels (used by the whole-path entry points; exported for unit parity tests and profiling)
 * ---------------------------------------------------------------------------------------------------- */
typedef struct pp_gemm_args {
    const float* A; int64_t lda; const int32_t* a_idx;  /* A(m,k) = a_kmajor ? A[ix(k)*lda + m] : A[ix(m)*lda + k] */
    const float* B; int64_t ldb; const int32_t* b_idx;  /* B(n,k) = b_kmajor ? B[ix(k)*ldb + n] : B[ix(n)*ldb + k] */
    float*       C; int64_t ldc; const int32_t* c_idx;  /* C[ix(m)*ldc + n] */
    int32_t M, N, K;
    int32_t a_kmajor, b_kmajor;
    const float* bias;   /* [N] or NULL, added to every row */
    const float* bias2;  /* [N] or NULL */
    const float* mask; int64_t ldmask;  /* optional: result = mask[ix_c(m)*ldmask + n] > 0 ? result : 0 (ReLU backward) */
    int32_t relu;        /* result = max(result, 0) */
    int32_t accumulate;  /* C += result instead of C = result */
    float*       colsum; /* optional [N]: colsum[n] += sum_m result[m,n] (after relu/mask): fused bias gradient */
    int32_t split_k;     /* 0: one workgroup per tile walks all of K (bit-reproducible); 1: the library may spread K over
                            several workgroups and combine with float atomics (used for the gradient products) */
    int32_t _pad;
} pp_gemm_args;

/* C[M,N] = epilogue( sum_k A(m,k) * B(n,k) ) on the fp32 matrix cores (nn.Linear / nn.LSTM GEMMs and their
 * gradients: embedding_feedforward.py:40, inference_network_lstm.py:188). *_idx are optional dev row-index
 * (gather/scatter) arrays: the "address-dispatch gather" of the proposal heads. */
int pp_gemm_f32(const pp_gemm_args* args, void* stream);
/* `count` independent products with identical operand layouts (a_kmajor/b_kmajor) in as few launches as possible
 * (up to 16 problems per launch): the weight-gradient leaves of a backward pass, the per-address head products. */
int pp_gemm_f32_grouped(const pp_gemm_args* args, int32_t count, void* stream);

/* out[c] += sum_i X[ix(i)*ldx + c] for c < n_cols (bias and embedding-table gradients). out2 optional. */
int pp_colsum_f32(const float* X, int64_t ldx, const int32_t* row_idx, int32_t n_rows, int32_t n_cols,
                  float* out, float* out2, void* stream);

/* LSTM input rows  x = [E | s_{t-1} | d_{t-1} | a_{t-1} | d_t | a_t]  (inference_network_lstm.py:146-181).
 * E: dev [B, e_obs] (trace b of row r = r - row_off[t]; given as trace_of_row dev [R]). X: dev [R, ldx]. */
int pp_lstm_input_gather(const pp_net* net, const float* params, const float* E, const int32_t* trace_of_row,
                         const float* value, const int32_t* addr, const int32_t* prev_row, int32_t n_rows,
                         float* X, int64_t ldx, void* stream);

/* Pointwise LSTM cell, gate order i,f,g,o (torch.nn.LSTM). G: [n,4H] pre-activations in, activated gates out.
 * c_prev NULL = first time step of a trace (c_{-1} = 0, inference_network_lstm.py:186-187): the forget gate multiplies zero,
 * its columns [H, 2H) of G are NOT read (pp_ic_loss leaves them uncomputed) and 0 is stored there for the backward pass. */
int pp_lstm_cell_fwd(float* G, const float* c_prev, float* c, float* h, int32_t n, int32_t H, void* stream);
/* Backward of the cell for one time step. G: gates in, dG (pre-activation grads) out. dc_carry: dev [n,H];
 * rows < n_next hold dL/dc_t from step t+1 on entry; on exit rows < n hold dL/dc_{t-1}. */
int pp_lstm_cell_bwd(float* G, const float* c_prev, const float* c, const float* dh, float* dc_carry,
                     int32_t n, int32_t n_next, int32_t H, void* stream);

/* Proposal transforms + Mixture/Categorical log_prob + its gradient w.r.t. the head output y
 * (proposal_normal_normal_mixture.py:20-35, proposal_uniform_truncated_normal_mixture.py:20-36,
 *  proposal_categorical_categorical.py:16-20, distributions/mixture.py:14-16,42-44,
 *  distributions/truncated_normal.py:25-30,40-54).
 *   y  dev [n, ldy]  head outputs, compact group order; rows dev [n] -> row id (value/prior/lp_out index) or NULL
 *   dy dev [n, ldy]  (NULL: forward only) receives grad_scale * d lp / d y  (0 for rows whose lp is -inf)
 *   loss_acc dev [1] += -sum lp (after the -inf -> log(1e-8) rescue); nonfinite dev [1] set if any lp is NaN/+inf */
int pp_head_logprob(int32_t kind, const float* y, int64_t ldy, const int32_t* rows, const float* value,
                    const float* prior, int32_t n, int32_t n_out, float grad_scale, float* lp_out, float* dy,
                    float* loss_acc, int32_t* nonfinite, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * In-stream kernel timing for bench.py's roofline leg: when armed, pp_ic_loss / pp_adam_step record a hipEvent pair
 * around the kernel class `which` on their stream, once per call:
 *   0  forward input GEMM X*W_ih^T                                   work = FLOPs
 *   1  the grouped weight-gradient launch of the backward pass       work = FLOPs
 *   2  observe embedding + LSTM input rows (the gather path)         work = algorithmic bytes
 *   3  pp_adam_step (optimizer pass over the flat buffers)           work = algorithmic bytes
 *   4  the draw + log q kernel of pp_is_step                         work = algorithmic bytes
 *   5  the fused statement kernel of pp_is_step / pp_is_step_rows    work = FLOPs (input + recurrent
 *                                                                    product and both head layers per particle)
 *   6  the device chain of a posterior call's first statement: every launch from pp_is_init to the end of pp_is_fused
 *      (observe embedding, the one-row network, the pass over the particles with its statistics)   work = 0
 * pp_prof_collect returns the elapsed milliseconds and the work of every recorded launch (flops_out).
 * ---------------------------------------------------------------------------------------------------- */
/* Host-side plan of the streaming weight-gradient launch (csrc/wgrad_t1.hip) for `count` queued products dW += A^T B (both
 * operands k-major: torch.autograd's weight gradients of nn.Linear / nn.LSTM, inference_network_lstm.py:186-220 backward) -
 * no device work, for tests of the host logic. zero_blocks: [count][2][6] = {m0, m1, n0, n1, k0, k1} or NULL; out: [cap][10] =
 * {M, N, K, S, ks, gather, first workgroup, C offset, A offset, B (or index) offset}. Returns the number of problems (0: the
 * tile kernels take the flush). */
int pp_debug_wgrad_plan(const pp_gemm_args* products, const int32_t* zero_blocks, int32_t count, int64_t* out, int32_t cap,
                        int32_t* n_blocks);
int pp_prof_arm(int32_t which, int32_t max_samples);          /* allocate event pairs; 0 disarms */
int pp_prof_stride(int32_t stride);                           /* time every stride-th launch of the class (default 1):
                                                                 an event pair costs the stream ~3 us, a stride keeps the
                                                                 timed region of a benchmark within ~1 % of an untimed one */
int pp_prof_collect(float* ms_out, int32_t cap, int32_t* n_out, double* flops_out); /* syncs the events */

/* Diagnostic: one wave runs `iters` dependent FMAs; out[0] = elapsed shader cycles (s_memtime), out[1] = elapsed
 * 100 MHz wall ticks (s_memrealtime). Effective shader clock = out[0] / (out[1] * 10 ns): used by bench.py to report the
 * DVFS state the timed region ran in (MI355X_MICROARCH.md "DVFS give-back"). */
/* Diagnostic: when set, the fused head-tail kernel writes per-phase s_memtime stamps of workgroups 0 and 100 to buf. */
/* ---- data parallel: RCCL from the C side (csrc/dp.hip) --------------------------------------------------------------
 * Replaces _distributed_sync_grad (pyprob/nn/inference_network.py:296-333: one all-reduce per tensor + presence map + loss)
 * with ONE grouped launch over the flat buffer. librccl.so is dlopen'ed from `rccl_path` (the copy torch loaded); the
 * communicator belongs to this library: rank 0 creates a 128-byte id (pp_dp_unique_id), the host hands it to every rank
 * (torch.distributed broadcast, pyprob_amd/parallel.py), all ranks call pp_dp_init (collective, current HIP device). */
int pp_dp_unique_id(const char* rccl_path, void* id_out /*host [128]*/);
int pp_dp_init(const char* rccl_path, const void* unique_id /*host [128]*/, int32_t rank, int32_t world);
int pp_dp_world(void);       /* size of the communicator, 0 when there is none */
int pp_dp_destroy(void);
/* in-place sum over the ranks of the pieces [off[i], off[i] + cnt[i]) of base (floats): one grouped launch */
int pp_dp_allreduce(float* base /*dev*/, const int64_t* off /*host*/, const int64_t* cnt /*host*/, int32_t n, void* stream);
/* The exchange of one training step: grads_full = dev [n_params | n_tensors | loss | flag]. Copies `presence` (dev
 * [n_tensors], or NULL when the tail already holds the step's presence map) and the step's non-finite flag `status` (dev
 * int32) into the tail, all-reduces everything but the skip ranges, and writes the mean loss / the any-rank flag to
 * loss_out / status_out (dev, may be NULL). Adam then runs with grad_scale = 1 / world on the reduced tail. */
int pp_dp_reduce_grads(float* grads_full, int64_t n_params, int32_t n_tensors, const float* presence, const int32_t* status,
                       const int64_t* skip_off /*host*/, const int64_t* skip_cnt /*host*/, int32_t n_skip, float* loss_out,
                       int32_t* status_out, void* stream);

/* (ABI 13) Bucket 0 under the rest of the backward pass - what `distributed_num_buckets` is for in the reference
 * (pyprob/nn/inference_network.py:300-325: gradients reduced bucket by bucket). off / cnt (host, n <= 2 ascending ranges of the
 * flat gradient buffer, in floats; n = 0: off) name gradient tensors that the backward pass completes EARLY: pp_ic_loss then
 * issues its weight-gradient launch in two parts - the products (and reduction jobs) that write into the hull of the ranges
 * first, the remaining ones after - and starts the ranges' all-reduce on a side stream between the two; pp_dp_reduce_grads
 * leaves the ranges out of its own collective and makes its stream wait for the side stream before it returns the step to
 * Adam. Host-side state: every rank must set the same ranges (pyprob_amd.engine.ICEngine.enable_dp_overlap: the first LSTM
 * layer's weight and bias gradients, minus a skipped W_hh; PP_DP_OVERLAP=1 - measured on a one-rank group the two-part launch and
 * its two cross-stream waits cost a 67 us step ~26 us, so it is not the default). Same result as the unsplit exchange. */
int pp_dp_overlap(const int64_t* off /*host*/, const int64_t* cnt /*host*/, int32_t n);
/* arm = 1 / 0: HIP event pairs around the collectives of the following overlapped steps on / off. us_out (host [3] or NULL):
 * the LAST overlapped step's {ranges' all-reduce on the side stream, the rest's all-reduce, what the step's stream then still
 * waited for the side stream} in microseconds (synchronises on those events). Returns 1 when us_out was filled, else 0. */
int pp_dp_overlap_stats(int32_t arm, float* us_out);

int pp_debug_timeline(long long* buf /*dev [16] or NULL*/);
/* debug: per-workgroup {start, end, problem, split, operands ready, loads issued, first slab landed, K loop done} stamps (10 ns ticks) of the grouped async GEMM launches
 * (mode 1: weight-gradient groups, 2: data-gradient products); buf = dev int64 [8 * cap] or NULL (tools/wg_trace.py) */
int pp_debug_wgtrace(long long* buf, int32_t cap, int32_t mode);
int pp_debug_clock_probe(int32_t iters, long long* out /*dev [2]*/, float* sink /*dev [1]*/, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PYPROB_AMD_H */
