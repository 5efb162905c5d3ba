// Cross-workgroup hand-off of partial sums through memory, without fences and without clearing anything (panel.hip: the two
// workgroups of a row panel; wgrad_t1.hip: the row ranges of a weight-gradient tile).
#pragma once
#include "common.hpp"

namespace pp {

// Every value crosses as ONE naturally aligned 8-byte granule {value, tag} written by one system-scope relaxed atomic store
// (global_store_dwordx2 ... sc0 sc1: write-through) and polled by the thread that needs it with system-scope loads (no L1 /
// L2 hit on the reading side): payload and "ready" arrive together, one memory round trip instead of payload + flag + payload
// read. The tag is the step's epoch (incremented once per step by the first launch), so nothing is ever cleared. The two
// workgroups of a pair are blocks b and b + 8 (the same XCD under round-robin placement; correctness does not depend on it).
// A bounded spin traps instead of hanging the device if a partner never arrives.
// (the tag word is a NaN bit pattern that no arithmetic produces - 0x7FC00001 + epoch mod (2^22 - 1): a quiet NaN WITH a
// payload, never the plain 0x7FC00000 - so stale floats or integers of another batch shape's workspace layout can never look
// like a ready granule; a slot would have to stay untouched for exactly 2^22 - 1 steps to alias)
__device__ __forceinline__ unsigned gtag(int epoch) { return 0x7FC00001u + ((unsigned)epoch % 0x3FFFFFu); }
__device__ __forceinline__ void gput(unsigned long long* p, float v, int epoch) {
    const unsigned long long x = ((unsigned long long)gtag(epoch) << 32) | (unsigned long long)__float_as_uint(v);
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float gget(const unsigned long long* p, int epoch) {
    int spins = 0;
    while (true) {
        const unsigned long long x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(x >> 32) == gtag(epoch)) return __uint_as_float((unsigned)x);
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) __builtin_trap();
    }
}

}  // namespace pp
