// The reference's other optimizers over the flat buffers (InferenceNetwork._create_optimizer,
// pyprob/nn/inference_network.py:343-355): optim.SGD(momentum, nesterov=True, weight_decay) and the LARC wrapper
// (pyprob/nn/optimizer_larc.py:72-103, default arguments: trust_coefficient 0.002, clip, eps 1e-8, epsilon 1/16000) that
// rescales each tensor's gradient by min(trust * |p| / (|g| + wd |p| + eps) / lr, 1) before the wrapped optimizer steps.
// Same buffer layout as pp_adam_step: tensors padded to 1024-float chunks (padding is zero in params and grads, so norms
// over the chunks are the tensors' norms), chunk_tensor names each chunk's tensor, `active` is the presence map
// (grad is not None), `skip` the device flag of a non-finite loss. HBM-bound elementwise passes: one workgroup per chunk,
// one 16-byte load / store per thread and buffer.
#include "common.hpp"

namespace pp {

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ P, float* __restrict__ Gr, float* __restrict__ Mb,
                                                  const int32_t* __restrict__ chunk_tensor, const float* __restrict__ active,
                                                  float lr, float momentum, int nesterov, float wd, float gscale,
                                                  int zero_grads, const int32_t* __restrict__ skip) {
    const int b = blockIdx.x;
    const int t = chunk_tensor[b];
    if (t < 0 || !(active[t] > 0.0f)) return;     // no gradient this step: torch skips the parameter (no decay either)
    const int64_t o = (int64_t)b * 1024 + threadIdx.x * 4;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    if (skip && skip[0] != 0) {                   // non-finite loss: no step (inference_network_lstm.py:216-217)
        if (zero_grads) *reinterpret_cast<f32x4*>(Gr + o) = zero;
        return;
    }
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(Gr + o);
    f32x4 p = *reinterpret_cast<const f32x4*>(P + o);
    f32x4 m = zero;
    const bool with_momentum = momentum != 0.0f;
    if (with_momentum) m = *reinterpret_cast<const f32x4*>(Mb + o);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float g = g0[e] * gscale;
        if (wd != 0.0f) g += wd * p[e];
        if (with_momentum) {
            // a zero buffer before a parameter's first step is torch's `buf = clone(grad)` (dampening 0)
            m[e] = momentum * m[e] + g;
            g = nesterov ? g + momentum * m[e] : m[e];
        }
        p[e] -= lr * g;
    }
    *reinterpret_cast<f32x4*>(P + o) = p;
    if (with_momentum) *reinterpret_cast<f32x4*>(Mb + o) = m;
    if (zero_grads) *reinterpret_cast<f32x4*>(Gr + o) = zero;
}

int sgd_step(float* params, float* grads, float* mbuf, int64_t n_params, const int32_t* chunk_tensor, const float* active,
             int n_tensors, float lr, float momentum, int nesterov, float wd, float gscale, int flags, const int32_t* skip,
             hipStream_t st) {
    PP_CHECK_ARG(params && grads && chunk_tensor && active, "pp_sgd_step: null pointer");
    PP_CHECK_ARG(mbuf || momentum == 0.0f, "pp_sgd_step: momentum needs its buffer");
    PP_CHECK_ARG(n_params % 1024 == 0, "pp_sgd_step: n_params must be a multiple of 1024 (padded tensors)");
    PP_CHECK_ARG(momentum >= 0.0f && (!nesterov || momentum > 0.0f), "pp_sgd_step: nesterov needs momentum > 0 (torch.optim.SGD)");
    if (n_tensors <= 0 || n_params == 0) return 0;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)(n_params / 1024)), dim3(256), 0, st, params, grads, mbuf, chunk_tensor, active,
                       lr, momentum, nesterov ? 1 : 0, wd, gscale, (flags & PP_ADAM_ZERO_GRADS) ? 1 : 0, skip);
    PP_LAUNCH_CHECK("pp_sgd_step");
    return 0;
}

// ---- LARC ---------------------------------------------------------------------------------------------------------
// pass 1: per chunk (sum p^2, sum g^2), stored (no float atomics: the per-tensor totals are summed in a fixed order)
__global__ __launch_bounds__(256) void larc_partial_kernel(const float* __restrict__ P, const float* __restrict__ Gr,
                                                           const int32_t* __restrict__ chunk_tensor,
                                                           const float* __restrict__ active, float* __restrict__ partial,
                                                           const int32_t* __restrict__ skip) {
    __shared__ float s_red[8];
    const int b = blockIdx.x;
    const int t = chunk_tensor[b];
    if (t < 0 || !(active[t] > 0.0f) || (skip && skip[0] != 0)) return;
    const int64_t o = (int64_t)b * 1024 + threadIdx.x * 4;
    const f32x4 p = *reinterpret_cast<const f32x4*>(P + o);
    const f32x4 g = *reinterpret_cast<const f32x4*>(Gr + o);
    float sp = (p[0] * p[0] + p[1] * p[1]) + (p[2] * p[2] + p[3] * p[3]);
    float sg = (g[0] * g[0] + g[1] * g[1]) + (g[2] * g[2] + g[3] * g[3]);
    sp = wave_sum(sp);
    sg = wave_sum(sg);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_red[w] = sp;
        s_red[4 + w] = sg;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * (int64_t)b] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        partial[2 * (int64_t)b + 1] = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    }
}

// pass 2: one workgroup per tensor: norms from the chunk partials (fp64, fixed order), the adaptive factor
__global__ __launch_bounds__(256) void larc_ratio_kernel(const int32_t* __restrict__ chunk_tensor, int n_chunks,
                                                         const float* __restrict__ active, const float* __restrict__ partial,
                                                         float* __restrict__ ratio, float lr, float wd, float gscale, float trust,
                                                         float eps, float epsilon, int clip, const int32_t* __restrict__ skip) {
    __shared__ int s_run[2];
    __shared__ double s_red[8];
    const int t = blockIdx.x;
    if (!(active[t] > 0.0f) || (skip && skip[0] != 0)) return;
    if (threadIdx.x == 0) {   // chunks of one tensor are contiguous and the ids ascend
        int lo = 0, hi = n_chunks;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (chunk_tensor[mid] < t) lo = mid + 1; else hi = mid;
        }
        int lo2 = lo, hi2 = n_chunks;
        while (lo2 < hi2) {
            const int mid = (lo2 + hi2) >> 1;
            if (chunk_tensor[mid] <= t) lo2 = mid + 1; else hi2 = mid;
        }
        s_run[0] = lo;
        s_run[1] = lo2;
    }
    __syncthreads();
    double sp = 0.0, sg = 0.0;
    for (int c = s_run[0] + (int)threadIdx.x; c < s_run[1]; c += 256) {
        sp += (double)partial[2 * (int64_t)c];
        sg += (double)partial[2 * (int64_t)c + 1];
    }
    sp = wave_sum(sp);
    sg = wave_sum(sg);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_red[w] = sp;
        s_red[4 + w] = sg;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double pn = sqrt((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
        const double gn = sqrt((s_red[4] + s_red[5]) + (s_red[6] + s_red[7])) * (double)gscale;   // norm of the averaged gradient
        // optimizer_larc.py:88-99
        double local = (double)epsilon;
        if (pn != 0.0 && gn != 0.0) local = (double)trust * pn / (gn + pn * (double)wd + (double)eps);
        const double adaptive = clip ? fmin(local / (double)lr, 1.0) : local;
        ratio[t] = (float)adaptive;
    }
}

// pass 3: grad = (grad * gscale + wd * p) * ratio[tensor]   (optimizer_larc.py:101-102; the wrapped optimizer then runs
// with weight_decay 0, :81,105-107)
__global__ __launch_bounds__(256) void larc_apply_kernel(const float* __restrict__ P, float* __restrict__ Gr,
                                                         const int32_t* __restrict__ chunk_tensor,
                                                         const float* __restrict__ active, const float* __restrict__ ratio,
                                                         float wd, float gscale, const int32_t* __restrict__ skip) {
    const int b = blockIdx.x;
    const int t = chunk_tensor[b];
    if (t < 0 || !(active[t] > 0.0f) || (skip && skip[0] != 0)) return;
    const int64_t o = (int64_t)b * 1024 + threadIdx.x * 4;
    f32x4 g = *reinterpret_cast<const f32x4*>(Gr + o);
    const float r = ratio[t];
    if (wd != 0.0f) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(P + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = (g[e] * gscale + wd * p[e]) * r;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = (g[e] * gscale) * r;
    }
    *reinterpret_cast<f32x4*>(Gr + o) = g;
}

int larc_scale(const float* params, float* grads, int64_t n_params, const int32_t* chunk_tensor, const float* active,
               int n_tensors, float lr, float wd, float gscale, float trust, float eps, float epsilon, int clip, float* scratch,
               const int32_t* skip, hipStream_t st) {
    PP_CHECK_ARG(params && grads && chunk_tensor && active && scratch, "pp_larc_scale: null pointer");
    PP_CHECK_ARG(n_params % 1024 == 0, "pp_larc_scale: n_params must be a multiple of 1024 (padded tensors)");
    PP_CHECK_ARG(lr > 0.0f || !clip, "pp_larc_scale: the clipping variant divides by the learning rate");
    if (n_tensors <= 0 || n_params == 0) return 0;
    const int n_chunks = (int)(n_params / 1024);
    float* partial = scratch;
    float* ratio = scratch + 2 * (int64_t)n_chunks;
    hipLaunchKernelGGL(larc_partial_kernel, dim3((unsigned)n_chunks), dim3(256), 0, st, params, grads, chunk_tensor, active, partial,
                       skip);
    hipLaunchKernelGGL(larc_ratio_kernel, dim3((unsigned)n_tensors), dim3(256), 0, st, chunk_tensor, n_chunks, active, partial, ratio,
                       lr, wd, gscale, trust, eps, epsilon, clip ? 1 : 0, skip);
    hipLaunchKernelGGL(larc_apply_kernel, dim3((unsigned)n_chunks), dim3(256), 0, st, params, grads, chunk_tensor, active, ratio, wd,
                       gscale, skip);
    PP_LAUNCH_CHECK("pp_larc_scale");
    return 0;
}

}  // namespace pp

extern "C" {

int pp_sgd_step(float* params, float* grads, float* momentum_buf, int64_t n_params, const int32_t* chunk_tensor,
                const float* active, int32_t n_tensors, float lr, float momentum, int32_t nesterov, float weight_decay,
                float grad_scale, int32_t flags, const int32_t* skip, void* stream) {
    // kernel class 3 of the in-stream timing (the optimizer pass): params, grads, buffer read; params, buffer (, grads) written
    pp::prof_begin(3, pp::as_stream(stream));
    const int rc = pp::sgd_step(params, grads, momentum_buf, n_params, chunk_tensor, active, n_tensors, lr, momentum, nesterov,
                                weight_decay, grad_scale, flags, skip, pp::as_stream(stream));
    const double words = (momentum != 0.0f ? 5.0 : 3.0) + ((flags & PP_ADAM_ZERO_GRADS) ? 1.0 : 0.0);
    pp::prof_end(3, 4.0 * words * (double)n_params, pp::as_stream(stream));
    return rc;
}

int pp_larc_scale(const float* params, float* grads, int64_t n_params, const int32_t* chunk_tensor, const float* active,
                  int32_t n_tensors, float lr, float weight_decay, float grad_scale, float trust_coefficient, float eps,
                  float epsilon, int32_t clip, float* scratch, const int32_t* skip, void* stream) {
    return pp::larc_scale(params, grads, n_params, chunk_tensor, active, n_tensors, lr, weight_decay, grad_scale,
                          trust_coefficient, eps, epsilon, clip, scratch, skip, pp::as_stream(stream));
}

}  // extern "C"
