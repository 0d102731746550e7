// Does hipExtAnyOrderLaunch drop the barrier between two kernels of one stream on gfx950?
// k_slow spins ~40 us then raises a flag; k_next records when it STARTED (wall clock) and
// whether the flag was already up.  Ordered launch: always starts after k_slow ended.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k_slow(unsigned long long *ts, int *flag, long long spin) {
    if (threadIdx.x == 0) {
        ts[0] = wall_clock64();
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
        ts[1] = wall_clock64();
        __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void k_next(unsigned long long *ts, int *flag) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ts[2] = wall_clock64();
        ts[3] = (unsigned long long)__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        long long t0 = wall_clock64();
        while (!__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) && wall_clock64() - t0 < 100000000) __builtin_amdgcn_s_sleep(8);
        ts[4] = wall_clock64();
    }
}
int main() {
    unsigned long long *ts; int *flag;
    hipMalloc(&ts, 64); hipMalloc(&flag, 4);
    hipStream_t s; hipStreamCreate(&s);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(flag, 0, 4, s); hipMemsetAsync(ts, 0, 64, s);
            hipStreamSynchronize(s);
            hipLaunchKernelGGL(k_slow, dim3(1), dim3(64), 0, s, ts, flag, 4000LL);   // 100 MHz clock: 40 us
            if (mode == 0) hipLaunchKernelGGL(k_next, dim3(1), dim3(64), 0, s, ts, flag);
            else hipExtLaunchKernelGGL(k_next, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ts, flag);
            hipError_t e = hipStreamSynchronize(s);
            unsigned long long h[5]; hipMemcpy(h, ts, 40, hipMemcpyDeviceToHost);
            printf("mode %s rep %d: err %d slow [%llu..%llu] next start +%lld ticks after slow START, flag seen at start %llu, waited %llu ticks\n",
                   mode ? "anyorder" : "ordered ", rep, (int)e, 0ull, h[1] - h[0], (long long)(h[2] - h[0]), h[3], h[4] - h[2]);
        }
    }
    return 0;
}
