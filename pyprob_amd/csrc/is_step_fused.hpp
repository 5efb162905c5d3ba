// The fused importance-sampling statement (is_step_fused.hip): interface towards is_kernels.hip.
#pragma once
#include "common.hpp"

namespace pp {

// per-call operand images in the pp_is_step workspace (sizes in floats; 0 when the network's shape has no fused kernel)
struct IsFusedBuffers {
    float *whh = nullptr, *w1 = nullptr, *w2 = nullptr, *bias = nullptr;
    int64_t n_whh = 0, n_w1 = 0, n_w2 = 0, n_bias = 0;
};

// one-layer LSTM with H in {256, 512, 1024} (1024: two launches, head at most 576 hidden units) or an LSTM of up to four layers with
// H a multiple of 32 up to 256 (is_step_small.hip); head of `addr_id` at most 32 outputs wide
bool is_step_fused_supported(const pp_net* net, int addr_id);
void is_fused_carve_sizes(const pp_net* net, IsFusedBuffers& f);

// whole-statement mode: previous values indexed by the particle's row, value and log-weight written at that row
struct IsStatementOut {
    float* value_full;
    float* lw_full;
    int prior_kind;      // 0 Normal, 1 Uniform
};

// A statement AFTER the first one of a trace (prev_addr_id >= 0) for n particles. state_rows == 1: row 0 of (h, c) is
// everybody's previous state. rows (or nullptr): state row of particle i (h, c are read and written at rows[i]; all other
// per-particle arrays are compact). *sampled: values and log q are written; false when only the head outputs were produced
// (y_out [n, ldy]: categorical / Bernoulli heads and net_only calls - the caller samples).
// hn_split: [n][H] scratch - the statement runs as TWO launches (LSTM step split over the gate columns, then head + draw): the
// faster shape for launches of a few thousand particles and fewer (ignored for the shared-state second statement)
int is_step_fused(const pp_net* net, const float* P, int addr_id, int prev_addr_id, int n, const float* e_obs_vec,
                  const float* prev_value, const float* prior, int prior_stride, float* h, float* c, int state_rows,
                  const int64_t* rows, const float* value_in, float* value_out, float* logq_out, uint64_t seed, uint64_t offset,
                  const IsFusedBuffers& f, float* c0_copy, float* y_out, int64_t ldy, bool net_only, bool* sampled, hipStream_t st,
                  const IsStatementOut* whole = nullptr, float* hn_split = nullptr, int64_t layer_rows = 0);
// (layer_rows: rows of one layer of (h, c) - the state of an LSTM of depth > 1 is [depth, layer_rows, H]; 0 = n)

// is_step_small.hip: the same statement for H = 32, 64 .. 256 (multiples of 32; H = 256 from two layers on) and 1 .. PP_MAX_LSTM_DEPTH
// layers (one kernel, every mode above)
bool is_small_network(const pp_net* net);
bool is_step_small_supported(const pp_net* net, int addr_id);
void is_small_carve_sizes(const pp_net* net, IsFusedBuffers& f);
int is_step_small(const pp_net* net, const float* P, int addr_id, int prev_addr_id, int n, const float* e_obs_vec,
                  const float* prev_value, const float* prior, int prior_stride, float* h, float* c, int state_rows,
                  int64_t layer_rows, const int64_t* rows, const float* value_in, float* value_out, float* logq_out, uint64_t seed,
                  uint64_t offset, const IsFusedBuffers& f, float* y_out, int64_t ldy, bool net_only, bool* sampled, hipStream_t st,
                  const IsStatementOut* whole);

// H = 1024 (one layer): the LSTM step of a statement as ONE launch (two workgroups per 32 particles, half of the hidden units
// each); the head layers and the draw stay with the caller's launches. c is updated in place, the new hidden rows go to hn [n][H]
bool is_lstm_wide_supported(const pp_net* net);
int is_lstm_wide(const pp_net* net, const float* P, int addr_id, int prev_addr_id, int n, const float* e_obs_vec,
                 const float* prev_value, float* h, float* c, int state_rows, const IsFusedBuffers& f, float* c0_copy, float* hn,
                 hipStream_t st);

}  // namespace pp
