// latlab.hip -- dependent-chain latencies of the instructions kernel A's step is made of,
// one wave per SIMD (the regime the sweep runs in).  hipcc --offload-arch=gfx950 -O3 -o latlab latlab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int CTRL>
__device__ __forceinline__ float dpp(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xF, 0xF, false));
}

constexpr int ITER = 4096;

// mode: 0 fma chain, 1 exp2 chain, 2 log2 chain, 3 bpermute chain, 4 wave_shr dpp chain, 5 row_shr dpp chain,
// 6 full lse step (dpp + fma + lse), 7 s_barrier only, 8 lse step + barrier each 8, 9 readlane chain,
// 10 ds_read chain, 11 lse step with gathers (bpermute of an independent register)
__global__ void lat(int mode, float *out, long long *cyc, float seed) {
    __shared__ float buf[1024];
    const int lane = threadIdx.x & 63;
    float x = seed + lane * 1e-3f, y = seed * 0.5f;
    buf[threadIdx.x] = x;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {
#pragma unroll 16
        for (int i = 0; i < ITER; ++i) x = fmaf(x, 0.999f, y);
    } else if (mode == 1) {
#pragma unroll 16
        for (int i = 0; i < ITER; ++i) x = __builtin_amdgcn_exp2f(x) - 1.0f;
    } else if (mode == 2) {
#pragma unroll 16
        for (int i = 0; i < ITER; ++i) x = __builtin_amdgcn_logf(x) + 3.0f;
    } else if (mode == 3) {
        int a = (lane * 4 + 4) & 255;
#pragma unroll 16
        for (int i = 0; i < ITER; ++i) x = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(x)));
    } else if (mode == 4) {
#pragma unroll 16
        for (int i = 0; i < ITER; ++i) x = dpp<0x138>(y, x);
    } else if (mode == 5) {
#pragma unroll 16
        for (int i = 0; i < ITER; ++i) x = dpp<0x111>(y, x);
    } else if (mode == 6 || mode == 8 || mode == 11) {
        int a = (lane * 4 + 8) & 255;
        for (int i = 0; i < ITER; i += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float ls = y, lm = y * 0.5f;
                if (mode == 11) {
                    ls = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(y + i + k)));
                    lm = __int_as_float(__builtin_amdgcn_ds_bpermute(a + 4, __float_as_int(y + i + k)));
                }
                const float left = dpp<0x138>(y, x);
                const float av = fmaf(ls, 1.44f, x), bv = fmaf(lm, 1.44f, left);
                const float mx = fmaxf(av, bv), d = -fabsf(av - bv);
                x = mx + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(d));
            }
            if (mode == 8) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else if (mode == 7) {
        for (int i = 0; i < ITER; ++i) asm volatile("s_barrier" ::: "memory");
    } else if (mode == 9) {
#pragma unroll 16
        for (int i = 0; i < ITER; ++i) {
            const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 17));
            x = x * 0.5f + s;
        }
    } else if (mode == 12) {
        float a = x, b = x + 1, c2 = x + 2, d = x + 3;
#pragma unroll 4
        for (int i = 0; i < ITER; i += 4) {
            a = fmaf(a, 0.999f, y); b = fmaf(b, 0.999f, y); c2 = fmaf(c2, 0.999f, y); d = fmaf(d, 0.999f, y);
        }
        x = a + b + c2 + d;
    } else if (mode == 13) {
        float a = x, b = x + 1, c2 = x + 2, d = x + 3;
#pragma unroll 4
        for (int i = 0; i < ITER; i += 4) {
            a = __builtin_amdgcn_exp2f(a) - 1.0f; b = __builtin_amdgcn_exp2f(b) - 1.0f;
            c2 = __builtin_amdgcn_exp2f(c2) - 1.0f; d = __builtin_amdgcn_exp2f(d) - 1.0f;
        }
        x = a + b + c2 + d;
    } else if (mode == 14) {
        int a0 = (lane * 4 + 4) & 255;
        float a = x, b = x + 1, c2 = x + 2, d = x + 3;
#pragma unroll 4
        for (int i = 0; i < ITER; i += 4) {
            a = __int_as_float(__builtin_amdgcn_ds_bpermute(a0, __float_as_int(a)));
            b = __int_as_float(__builtin_amdgcn_ds_bpermute(a0, __float_as_int(b)));
            c2 = __int_as_float(__builtin_amdgcn_ds_bpermute(a0, __float_as_int(c2)));
            d = __int_as_float(__builtin_amdgcn_ds_bpermute(a0, __float_as_int(d)));
        }
        x = a + b + c2 + d;
    } else if (mode == 10) {
        int idx = lane;
#pragma unroll 16
        for (int i = 0; i < ITER; ++i) idx = __float_as_int(buf[idx & 1023]) & 1023;
        x = (float)idx;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float *out;
    long long *cyc;
    CHECK(hipMalloc(&out, 1 << 20));
    CHECK(hipMalloc(&cyc, 4096 * 8));
    const char *names[] = {"fma chain", "exp2 chain (+sub)", "log2 chain (+add)", "ds_bpermute chain", "dpp wave_shr chain",
                           "dpp row_shr chain", "lse step (dpp+2fma+lse)", "s_barrier", "lse step + barrier/8", "readlane+fma chain",
                           "ds_read_b32 chain", "lse step + 2 bpermute gathers", "4 independent fma chains (per instr)", "4 independent exp2+sub chains (per pair)", "4 independent bpermute chains (per instr)"};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int waves = 1; waves <= 1; waves += 4) {
        for (int mode = 0; mode < 15; ++mode) {
            for (int blocks : {1, 256}) {
                hipLaunchKernelGGL(lat, dim3(blocks), dim3(64 * waves), 0, 0, mode, out, cyc, 1.5f);
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(lat, dim3(blocks), dim3(64 * waves), 0, 0, mode, out, cyc, 1.5f);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                long long c;
                CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
                printf("waves/block %d blocks %3d  %-32s %8.1f memtime-ticks/iter  %8.1f ns/iter (wall)\n", waves, blocks,
                       names[mode], (double)c / ITER, ms * 1e6 / ITER);
            }
        }
    }
    return 0;
}
