// Kernel-level A/B of the cross-workgroup hand-off of csrc/panel16.hip (VERDICT r05 item 5 ii: "measure tag-free payloads behind
// one flag per tile against the current scheme at the kernel level"), isolated from the panel kernel's arithmetic:
//   scheme A (csrc/handoff.hpp, what ships): every value is an 8-byte {value, tag} granule, one system-scope relaxed store; the
//            consumer polls the granules themselves - payload and "ready" arrive in ONE memory round trip;
//   scheme B: 4-byte payloads (system-scope relaxed stores), s_waitcnt vmcnt(0), then ONE flag per producer wave; the consumer
//            polls the three partner waves' flags and only then loads the payloads - half the bytes, two dependent round trips.
// Geometry of the panel kernel's first exchange: 256 workgroups of 512 threads = 64 panels x 4 quarters (quarters of a panel are
// blocks 8 apart: the same XCD); a thread publishes 9 values and needs the 9 values of the SAME thread index of its three partners
// (27 loads in flight together). EXCH exchanges per launch (epoch = exchange index), a short dependent ALU delay between them.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/handoff_probe.hip -o /tmp/handoff_probe && /tmp/handoff_probe
// prints per scheme: cycles per exchange (clock64 of wave 0 of every workgroup: median / p90 over workgroups), kernel time per
// exchange (hipEvents), bytes written per exchange.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
#include <vector>

constexpr int NV = 9;        // values per thread and exchange
constexpr int NT = 512;
constexpr int WGS = 256;
constexpr int EXCH = 64;

__device__ __forceinline__ unsigned gtag(int epoch) { return 0x7FC00001u + ((unsigned)epoch % 0x3FFFFFu); }

template <int SCHEME>
__global__ __launch_bounds__(NT) void exchange_kernel(unsigned long long* gran /*[WGS][NV][NT]*/, float* pay /*[WGS][NV][NT]*/,
                                                      unsigned* flags /*[WGS][8]*/, int epoch0, float* sink, long long* cyc /*[WGS][EXCH]*/, float* check /*[WGS][NT]*/) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x;
    const int q = (b >> 3) & 3, panel = (b & 7) + 8 * (b >> 5);
    auto block_of = [&](int qq) { return (panel & 7) + 8 * qq + 32 * (panel >> 3); };
    int partner[3];
    for (int s = 0; s < 3; ++s) partner[s] = block_of(s + (s >= q ? 1 : 0));
    float v[NV];
    for (int i = 0; i < NV; ++i) v[i] = (float)(tid * 3 + i + b);
    float total = 0.0f;
    for (int e = 0; e < EXCH; ++e) {
        const int epoch = epoch0 + e;
        const unsigned tag = gtag(epoch);
        // two alternating buffers: a slot is rewritten two exchanges later, when every partner has published the exchange in between
        // and therefore finished reading this one (one exchange per launch in the product kernel: the launch boundary does this)
        unsigned long long* const gr = gran + (size_t)(epoch & 1) * WGS * NV * NT;
        float* const py = pay + (size_t)(epoch & 1) * WGS * NV * NT;
        unsigned* const fl = flags + (epoch & 1) * WGS * 8;
        const long long t0 = clock64();
        if (SCHEME == 0) {
            for (int i = 0; i < NV; ++i) {
                const unsigned long long x = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v[i]);
                __hip_atomic_store(gr + ((size_t)b * NV + i) * NT + tid, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            float xs[3][NV];
            int spins = 0;
            while (true) {
                unsigned long long x[3][NV];
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int i = 0; i < NV; ++i)
                        x[s][i] = __hip_atomic_load(gr + ((size_t)partner[s] * NV + i) * NT + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                bool ok = true;
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        ok = ok && ((unsigned)(x[s][i] >> 32) == tag);
                        xs[s][i] = __uint_as_float((unsigned)x[s][i]);
                    }
                if (ok) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 22)) __builtin_trap();
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = 0.25f * (v[i] + xs[0][i] + xs[1][i] + xs[2][i]);
        } else {
            for (int i = 0; i < NV; ++i)
                __hip_atomic_store(py + ((size_t)b * NV + i) * NT + tid, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's payload stores are acknowledged (write-through)
            if (lane == 0) __hip_atomic_store(fl + b * 8 + wave, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            int spins = 0;
            while (true) {      // the partner WAVES of the same index wrote what this wave needs
                unsigned f0 = __hip_atomic_load(fl + partner[0] * 8 + wave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                unsigned f1 = __hip_atomic_load(fl + partner[1] * 8 + wave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                unsigned f2 = __hip_atomic_load(fl + partner[2] * 8 + wave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (f0 == tag && f1 == tag && f2 == tag) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 22)) __builtin_trap();
            }
            float xs[3][NV];
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    xs[s][i] = __hip_atomic_load(py + ((size_t)partner[s] * NV + i) * NT + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = 0.25f * (v[i] + xs[0][i] + xs[1][i] + xs[2][i]);
        }
        const long long t1 = clock64();
        if (tid == 0) cyc[(size_t)b * EXCH + e] = t1 - t0;
        // ~1 us of dependent arithmetic before the next exchange
        for (int k = 0; k < 200; ++k) v[k % NV] = v[k % NV] * 1.0000001f + 1e-9f;
        __syncthreads();
        for (int i = 0; i < NV; ++i) total += v[i];
    }
    if (total == 123.456f) sink[0] = total;
    check[(size_t)b * NT + tid] = total;      // (both schemes run the same arithmetic in the same order: bit-identical sums)
}

template <int SCHEME>
static double run(const char* name, unsigned long long* gran, float* pay, unsigned* flags, float* sink, long long* cyc, int& epoch, float* check) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) {
        hipLaunchKernelGGL(exchange_kernel<SCHEME>, dim3(WGS), dim3(NT), 0, 0, gran, pay, flags, epoch, sink, cyc, check);
        epoch += EXCH;
    }
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(exchange_kernel<SCHEME>, dim3(WGS), dim3(NT), 0, 0, gran, pay, flags, epoch, sink, cyc, check);
        epoch += EXCH;
    }
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h((size_t)WGS * EXCH);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<long long> s;
    for (int b = 0; b < WGS; ++b)
        for (int e = 8; e < EXCH; ++e) s.push_back(h[(size_t)b * EXCH + e]);      // (the first exchanges of a launch start skewed)
    std::sort(s.begin(), s.end());
    const double bytes = (double)WGS * NT * NV * (SCHEME == 0 ? 8 : 4) + (SCHEME == 0 ? 0 : WGS * 8 * 4);
    printf("%-28s exchange: median %lld cycles, p90 %lld, p99 %lld (clock64, 100 MHz x 24 on this part: 2400 = 1 us) | kernel %.3f us per "
           "exchange (incl. the ~1 us delay) | %.2f MB written per exchange\n",
           name, s[s.size() / 2], s[s.size() * 9 / 10], s[s.size() * 99 / 100], ms * 1e3 / (reps * EXCH), bytes / 1e6);
    hipEventDestroy(e0); hipEventDestroy(e1);
    std::vector<float> hc((size_t)WGS * NT);
    hipMemcpy(hc.data(), check, hc.size() * sizeof(float), hipMemcpyDeviceToHost);
    double sum = 0.0;
    for (float x : hc) sum += (double)x;
    return sum;
}

int main() {
    unsigned long long* gran; float* pay; unsigned* flags; float* sink; long long* cyc;
    hipMalloc(&gran, (size_t)2 * WGS * NV * NT * 8);
    hipMalloc(&pay, (size_t)2 * WGS * NV * NT * 4);
    hipMalloc(&flags, 2 * WGS * 8 * 4);
    hipMalloc(&sink, 4);
    hipMalloc(&cyc, (size_t)WGS * EXCH * 8);
    hipMemset(gran, 0, (size_t)2 * WGS * NV * NT * 8);
    hipMemset(pay, 0, (size_t)2 * WGS * NV * NT * 4);
    hipMemset(flags, 0, 2 * WGS * 8 * 4);
    float* check;
    hipMalloc(&check, (size_t)WGS * NT * 4);
    int epoch = 1;
    for (int round = 0; round < 2; ++round) {
        const double ca = run<0>("A {value, tag} granules", gran, pay, flags, sink, cyc, epoch, check);
        const double cb = run<1>("B payload + flag per wave", gran, pay, flags, sink, cyc, epoch, check);
        printf("checksum of the last launch's values: A %.9e  B %.9e  %s\n", ca, cb, ca == cb ? "(identical)" : "(DIFFERENT)");
    }
    return 0;
}
