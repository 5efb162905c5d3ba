#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { float v[2000]; };
__global__ void k(Big b, float* out) { out[threadIdx.x] = b.v[threadIdx.x * 7 % 2000] + b.v[1999]; }
int main() {
  Big b; for (int i = 0; i < 2000; ++i) b.v[i] = (float)i;
  float* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, b, d);
  hipError_t e = hipDeviceSynchronize();
  float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  printf("err=%d h[3]=%f (expect %f)\n", (int)e, h[3], 21.0f + 1999.0f);
  return 0;
}
