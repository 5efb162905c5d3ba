// Micro-benchmark: issue interval of v_mfma_f32_32x32x2_f32 with 1, 2 or 4 independent accumulator chains per wave,
// and with 1..4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ __launch_bounds__(1024) void k(float* out, int iters, long long* cyc) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16 / CH; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CH>
void run(int threads, int blocks = 1) {
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    const int iters = 2000;
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 16.0 * iters;   // MFMAs per wave
    printf("blocks=%d chains=%d waves/SIMD=%d : %.1f clock64 ticks per MFMA per wave, %.1f ns per MFMA per wave (event)\n", blocks, CH, threads / 256,
           c / n, ms * 1e6 / n);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int t : {256, 512, 1024}) { run<1>(t); run<2>(t); run<4>(t); }
    for (int b : {8, 64, 256, 512, 1024}) run<1>(256, b);
    return 0;
}
