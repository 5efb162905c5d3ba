// Lane layout and issue rate of v_mfma_f32_4x4x1_16B_f32 (sixteen independent 4x4x1 blocks per instruction) on gfx950:
// the row-panel kernel (csrc/panel.hip) relies on
//   A: lane l supplies A[i = l % 4] of block l / 4        B: lane l supplies B[j = l % 4] of block l / 4
//   D: register i of lane l = D[i][j = l % 4] of block l / 4
// i.e. with A = the activations of 4 batch rows broadcast to every block and B = one weight per lane, register i of lane l
// accumulates out[row i][column l]: a 4-row x 64-column x 1-k product per 8-cycle instruction (64 FLOP/clk/SIMD, the fp32 peak).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_4x4_probe.hip -o /tmp/mfma_4x4_probe && /tmp/mfma_4x4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const float* act /*[4][K]*/, const float* w /*[64][K]*/, int K, float* out /*[4][64]*/) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
        const float a = act[(l & 3) * K + k];
        const float b = w[l * K + k];
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) out[i * 64 + l] = acc[i];
}

__global__ void rate_kernel(int iters, float* sink, long long* cycles) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const float a = (float)threadIdx.x, b = 1.0f / (1.0f + threadIdx.x);
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
    f32x4 s = c0 + c1 + c2 + c3;
    if (s[0] == 123.456f) sink[0] = s[1] + s[2] + s[3];
}

template <int NACC>
__global__ void rate_n_kernel(int iters, float* sink, long long* cycles) {
    f32x4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = f32x4{0, 0, 0, 0};
    const float a = (float)threadIdx.x, b = 1.0f / (1.0f + threadIdx.x);
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8 / NACC; ++rep)
#pragma unroll
            for (int k = 0; k < NACC; ++k) c[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[k], 0, 0, 0);
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
    f32x4 s = c[0];
    for (int k = 1; k < NACC; ++k) s += c[k];
    if (s[0] == 123.456f) sink[0] = s[1] + s[2] + s[3];
}

template <int NACC>
void run_rate(float* dout, long long* dc) {
    const int iters = 5000;
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(rate_n_kernel<NACC>, dim3(1), dim3(threads), 0, 0, iters, dout, dc);
        long long cyc;
        hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
        printf("  %d accumulator chain(s), %2d waves on the CU: %.2f cycles per instruction per wave\n", NACC, threads / 64,
               (double)cyc / (8.0 * iters));
    }
}

int main() {
    const int K = 37;
    float ha[4 * K], hw[64 * K], ho[256];
    for (int i = 0; i < 4 * K; ++i) ha[i] = sinf(0.37f * i) + 0.1f * (i % 4);
    for (int i = 0; i < 64 * K; ++i) hw[i] = cosf(0.11f * i) * (1 + (i % 7));
    float *da, *dw, *dout;
    long long* dc;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&dw, sizeof(hw)); hipMalloc(&dout, sizeof(ho)); hipMalloc(&dc, 8);
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
    hipMemcpy(dw, hw, sizeof(hw), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, da, dw, K, dout);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 4; ++i)
        for (int n = 0; n < 64; ++n) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)ha[i * K + k] * hw[n * K + k];
            worst = fmax(worst, fabs(ref - ho[i * 64 + n]));
        }
    printf("layout: max |out[row i][col l] - reference| = %.3g  (%s)\n", worst, worst < 1e-3 ? "OK: D reg i, lane l = row i, column l" : "MISMATCH");
    const int iters = 20000;
    hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(64), 0, 0, iters, dout, dc);
    long long cyc;
    hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
    printf("rate: %.2f cycles per v_mfma_f32_4x4x1_16B_f32 (one wave, four accumulators) = %.1f FLOP/clk/SIMD\n",
           (double)cyc / (4.0 * iters), 512.0 * 4.0 * iters / (double)cyc);
    hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(512), 0, 0, iters, dout, dc);
    hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
    printf("rate, 8 waves on one CU: %.2f cycles per instruction per wave\n", (double)cyc / (4.0 * iters));
    run_rate<1>(dout, dc);
    run_rate<2>(dout, dc);
    run_rate<4>(dout, dc);
    run_rate<8>(dout, dc);
    return worst < 1e-3 ? 0 : 1;
}
