// Host-side packing of a ragged minibatch into the step-major layout pp_ic_loss consumes. Pure host code (no device
// access, callable without a GPU): the native replacement of what the reference does per minibatch in Python -
// Batch.__init__'s grouping (pyprob/nn/dataset.py:21-37) and the per-trace torch.stack / torch.cat calls of _loss
// (pyprob/nn/inference_network_lstm.py:146-196). The numpy version of the same algorithm (packed.py, kept as the
// test oracle of this function) costs 130-190 us per 1024-trace minibatch in ~40 small array operations; this is one
// pass over the rows.
#include "common.hpp"

#include <algorithm>
#include <numeric>
#include <vector>

namespace pp {
void set_error(const char* fmt, ...);
}

extern "C" {

// Layout of the packed buffer, in 4-byte words (floats and int32 share it; one H2D copy uploads the device part):
//   device part: obs [B*W] | value [R] | prior [2R] | addr [R] | prev_row [R] | grp_rows [R] | trace [R] |
//                row_off [T+1] | nxt_rows [max(R-B, 1)]
//   host part:   n_active [T] | row_off [T+1] | grp_off [n_addr+1] | nxt_off [n_addr+1] | order [B] | src_row [R]
int64_t pp_pack_words(int32_t n_traces, int64_t n_rows, int32_t t_max, int32_t obs_width, int32_t n_addr) {
    const int64_t B = n_traces, R = n_rows, T = t_max;
    const int64_t dev = B * obs_width + 3 * R + 4 * R + (T + 1) + std::max<int64_t>(R - B, 1);
    const int64_t host = T + (T + 1) + 2 * ((int64_t)n_addr + 1) + B + R;
    return dev + host;
}

}  // extern "C" (reopened below)

namespace {

// One pass over the rows. SRC tells, for input trace b: its length, and for its statement t: address id, value, the two
// prior parameters; and its observation row.
template <class SRC>
int pack_core(const SRC& src, int B, int obs_width, int n_addr, void* out, int64_t out_words, pp_pack_info* info) {
    std::vector<int32_t> len(B), first_addr(B);
    int T = 0;
    int64_t R = 0;
    for (int b = 0; b < B; ++b) {
        len[b] = src.length(b);
        if (len[b] <= 0) {
            pp::set_error("Trace of length zero.");     // pyprob/nn/dataset.py:28-29
            return PP_EINVAL;
        }
        R += len[b];
        T = std::max(T, len[b]);
        first_addr[b] = src.addr(b, 0);
    }
    if (out_words < pp_pack_words(B, R, T, obs_width, n_addr)) {
        pp::set_error("pp_pack: output buffer too small");
        return PP_ENOSPACE;
    }
    std::vector<int64_t> off(B + 1, 0);
    for (int b = 0; b < B; ++b) off[b + 1] = off[b] + len[b];
    // longest first; ties keep traces with the same first address adjacent, then input order (stable)
    std::vector<int32_t> order(B);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        if (len[a] != len[b]) return len[a] > len[b];
        return first_addr[a] < first_addr[b];
    });
    float* w = static_cast<float*>(out);
    int32_t* wi = static_cast<int32_t*>(out);
    int64_t p = 0;
    const int64_t o_obs = p; p += (int64_t)B * obs_width;
    const int64_t o_val = p; p += R;
    const int64_t o_pri = p; p += 2 * R;
    const int64_t o_addr = p; p += R;
    const int64_t o_prev = p; p += R;
    const int64_t o_grp = p; p += R;
    const int64_t o_trace = p; p += R;
    const int64_t o_roffd = p; p += T + 1;
    const int64_t o_nxt = p; p += std::max<int64_t>(R - B, 1);
    const int64_t dev_words = p;
    const int64_t o_nact = p; p += T;
    const int64_t o_roff = p; p += T + 1;
    const int64_t o_goff = p; p += n_addr + 1;
    const int64_t o_noff = p; p += n_addr + 1;
    const int64_t o_order = p; p += B;
    const int64_t o_src = p; p += R;

    for (int i = 0; i < B; ++i) {
        wi[o_order + i] = order[i];
        src.obs_row(order[i], w + o_obs + (int64_t)i * obs_width);
    }
    {   // n_active[t] = traces longer than t (lengths are sorted descending)
        int i = B;
        for (int t = 0; t < T; ++t) {
            while (i > 0 && len[order[i - 1]] <= t) --i;
            wi[o_nact + t] = i;
        }
    }
    wi[o_roff] = 0;
    for (int t = 0; t < T; ++t) wi[o_roff + t + 1] = wi[o_roff + t] + wi[o_nact + t];
    for (int t = 0; t <= T; ++t) wi[o_roffd + t] = wi[o_roff + t];
    std::vector<int32_t> cur(n_addr + 1, 0), prv(n_addr + 1, 0);
    for (int t = 0; t < T; ++t) {
        const int n = wi[o_nact + t], r0 = wi[o_roff + t];
        const int rp = t > 0 ? wi[o_roff + t - 1] : 0;
        for (int i = 0; i < n; ++i) {
            const int b = order[i];
            const int r = r0 + i;
            const int a = src.addr(b, t);
            if (a < 0 || a >= n_addr) {
                pp::set_error("pp_pack: address id %d out of range (trace %d, statement %d): address unknown by the network?", a, b, t);
                return PP_EINVAL;
            }
            w[o_val + r] = src.value(b, t);
            w[o_pri + 2 * (int64_t)r] = src.prior(b, t, 0);
            w[o_pri + 2 * (int64_t)r + 1] = src.prior(b, t, 1);
            wi[o_addr + r] = a;
            wi[o_trace + r] = i;
            wi[o_prev + r] = t > 0 ? rp + i : -1;
            wi[o_src + r] = (int32_t)(off[b] + t);
            ++cur[a + 1];
            if (t > 0) ++prv[wi[o_addr + rp + i] + 1];
        }
    }
    // rows grouped by address (stable in row order), and rows t >= 1 grouped by the address of their previous variable
    for (int a = 0; a < n_addr; ++a) { cur[a + 1] += cur[a]; prv[a + 1] += prv[a]; }
    for (int a = 0; a <= n_addr; ++a) { wi[o_goff + a] = cur[a]; wi[o_noff + a] = prv[a]; }
    std::vector<int32_t> gpos(cur.begin(), cur.end() - 1), npos(prv.begin(), prv.end() - 1);
    if (R - B <= 0) wi[o_nxt] = 0;
    for (int64_t r = 0; r < R; ++r) {
        wi[o_grp + gpos[wi[o_addr + r]]++] = (int32_t)r;
        const int pr = wi[o_prev + r];
        if (pr >= 0) wi[o_nxt + npos[wi[o_addr + pr]]++] = (int32_t)r;
    }
    info->n_traces = B;
    info->n_rows = R;
    info->t_max = T;
    info->device_words = dev_words;
    info->obs = o_obs; info->value = o_val; info->prior = o_pri; info->addr = o_addr; info->prev_row = o_prev;
    info->grp_rows = o_grp; info->trace = o_trace; info->row_off_dev = o_roffd; info->nxt_rows = o_nxt;
    info->n_active = o_nact; info->row_off = o_roff; info->grp_off = o_goff; info->nxt_off = o_noff;
    info->order = o_order; info->src_row = o_src;
    return 0;
}

struct RaggedSource {   // trace-major arrays of one minibatch
    const int32_t* trace_len; const int32_t* addr_ids; const float* values; const float* prior_; int prior_width;
    const float* obs; int obs_width; std::vector<int64_t> off;
    int length(int b) const { return trace_len[b]; }
    int addr(int b, int t) const { return addr_ids[off[b] + t]; }
    float value(int b, int t) const { return values[off[b] + t]; }
    float prior(int b, int t, int k) const { return k < prior_width ? prior_[(off[b] + t) * prior_width + k] : 0.0f; }
    void obs_row(int b, float* dst) const { std::copy(obs + (int64_t)b * obs_width, obs + (int64_t)(b + 1) * obs_width, dst); }
};

struct IndexedSource {   // traces picked by global index out of the column shards of a packed on-disk dataset
    const pp_shard_columns* shards; int obs_width; std::vector<int32_t> shard; std::vector<int64_t> local;
    int length(int b) const { return shards[shard[b]].trace_len[local[b]]; }
    int64_t row(int b, int t) const { return shards[shard[b]].row_off[local[b]] + t; }
    int addr(int b, int t) const {
        const pp_shard_columns& s = shards[shard[b]];
        const int a = s.addr[row(b, t)];
        return s.addr_remap ? s.addr_remap[a] : a;
    }
    float value(int b, int t) const { return shards[shard[b]].value[row(b, t)]; }
    float prior(int b, int t, int k) const { return shards[shard[b]].prior[2 * row(b, t) + k]; }
    void obs_row(int b, float* dst) const {
        const float* o = shards[shard[b]].obs + local[b] * obs_width;
        std::copy(o, o + obs_width, dst);
    }
};

}  // namespace

extern "C" {

int pp_pack_ragged(const int32_t* trace_len, const int32_t* addr_ids, const float* values, const float* prior,
                   int32_t prior_width, const float* obs, int32_t n_traces, int32_t obs_width, int32_t n_addr,
                   void* out, int64_t out_words, pp_pack_info* info) {
    if (!(trace_len && addr_ids && values && obs && out && info) || n_traces <= 0 || n_addr <= 0 || obs_width < 0 ||
        (prior_width > 0 && !prior)) {
        pp::set_error("pp_pack_ragged: bad argument");
        return PP_EINVAL;
    }
    RaggedSource src{trace_len, addr_ids, values, prior, prior_width, obs, obs_width, std::vector<int64_t>(n_traces + 1, 0)};
    for (int b = 0; b < n_traces; ++b) src.off[b + 1] = src.off[b] + std::max(trace_len[b], 0);
    return pack_core(src, n_traces, obs_width, n_addr, out, out_words, info);
}

int pp_pack_indexed(const pp_shard_columns* shards, int32_t n_shards, const int64_t* first, const int64_t* ids,
                    int32_t n_ids, int32_t obs_width, int32_t n_addr, void* out, int64_t out_words, pp_pack_info* info) {
    if (!(shards && first && ids && out && info) || n_shards <= 0 || n_ids <= 0 || n_addr <= 0 || obs_width < 0) {
        pp::set_error("pp_pack_indexed: bad argument");
        return PP_EINVAL;
    }
    IndexedSource src{shards, obs_width, std::vector<int32_t>(n_ids), std::vector<int64_t>(n_ids)};
    for (int b = 0; b < n_ids; ++b) {
        if (ids[b] < 0 || ids[b] >= first[n_shards]) {
            pp::set_error("pp_pack_indexed: trace index %lld out of range", (long long)ids[b]);
            return PP_EINVAL;
        }
        const int s = (int)(std::upper_bound(first, first + n_shards + 1, ids[b]) - first) - 1;
        src.shard[b] = s;
        src.local[b] = ids[b] - first[s];
    }
    return pack_core(src, n_ids, obs_width, n_addr, out, out_words, info);
}

}  // extern "C"
