// bwprobe.hip -- what can a hand-written gfx950 streaming kernel reach with the
// access pattern of the logZ kernels?  (tools only; not part of the library)
//
//   hipcc --offload-arch=gfx950 -O3 tools/bwprobe.hip -o tools/bwprobe && tools/bwprobe
//
// Pattern: scores (T, N, 40) fp32; a "row-set" = 64 reads x 40 floats = 10 KB contiguous,
// consecutive rows of one column are N*160 B apart.  Block (col, chunk) streams CH rows
// with W waves (CH/W consecutive rows per wave), DEPTH row-sets in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int W, int DEPTH, bool WRITE, bool NT>
__global__ __launch_bounds__(W * 64) void rowset_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                       int T, int N, int CH, float *sink) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = blockIdx.x, chunk = blockIdx.y;
    const int rows = CH / W;
    const int t0 = chunk * CH + w * rows;
    const size_t rowstride = (size_t)N * 40;
    const float *base = in + (size_t)col * 64 * 40;
    float *obase = out + (size_t)col * 64 * 40;
    f4 acc = {0, 0, 0, 0};
    f4 buf[DEPTH][10];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const f4 *p = reinterpret_cast<const f4 *>(base + (size_t)min(t0 + d, T - 1) * rowstride);
#pragma unroll
        for (int q = 0; q < 10; ++q) buf[d][q] = NT ? __builtin_nontemporal_load(p + lane + 64 * q) : p[lane + 64 * q];
    }
    for (int r0 = 0; r0 < rows; r0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int t = t0 + r0 + d;
            f4 cur[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) cur[q] = buf[d][q];
            const f4 *p = reinterpret_cast<const f4 *>(base + (size_t)min(t + DEPTH, T - 1) * rowstride);
#pragma unroll
            for (int q = 0; q < 10; ++q) buf[d][q] = NT ? __builtin_nontemporal_load(p + lane + 64 * q) : p[lane + 64 * q];
            if (WRITE) {
                f4 *o = reinterpret_cast<f4 *>(obase + (size_t)min(t, T - 1) * rowstride);
#pragma unroll
                for (int q = 0; q < 10; ++q) {
                    const f4 v = cur[q] * 2.0f;
                    if (NT) __builtin_nontemporal_store(v, o + lane + 64 * q); else o[lane + 64 * q] = v;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 10; ++q) acc += cur[q];
            }
        }
    }
    if (!WRITE && acc[0] + acc[1] + acc[2] + acc[3] == 1234.5678f) sink[0] = acc[0];
}

// fully linear: block b streams a contiguous span, all waves interleaved at 1 KB granularity
template <int W, int DEPTH, bool NT>
__global__ __launch_bounds__(W * 64) void linear_kernel(const f4 *__restrict__ in, size_t n4, float *sink) {
    const size_t per_block = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per_block, hi = min(lo + per_block, n4);
    f4 acc = {0, 0, 0, 0};
    for (size_t i = lo + threadIdx.x; i < hi; i += (size_t)W * 64 * DEPTH) {
        f4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const size_t j = min(i + (size_t)d * W * 64, n4 - 1);
            v[d] = NT ? __builtin_nontemporal_load(in + j) : in[j];
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5678f) sink[0] = acc[0];
}

template <typename F>
static void timeit(const char *name, double bytes, F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < 15; ++i) {
        CK(hipEventRecord(a, 0)); launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-44s median %8.1f us  min %8.1f us  -> %5.2f TB/s (min %5.2f)\n", name, ms[7] * 1e3, ms[0] * 1e3,
           bytes / (ms[7] * 1e-3) / 1e12, bytes / (ms[0] * 1e-3) / 1e12);
    fflush(stdout);
}

template <int W, int DEPTH, bool WRITE, bool NT>
static void run_rowset(const float *in, float *out, float *sink, int T, int N, int CH) {
    char name[128];
    snprintf(name, sizeof name, "rowset %s W=%d depth=%d CH=%d %s N=%d", WRITE ? "r+w " : "read", W, DEPTH, CH, NT ? "nt" : "  ", N);
    const double bytes = (double)T * N * 160 * (WRITE ? 2 : 1);
    timeit(name, bytes, [&] {
        hipLaunchKernelGGL((rowset_kernel<W, DEPTH, WRITE, NT>), dim3(N / 64, T / CH), dim3(W * 64), 0, 0, in, out, T, N, CH, sink);
    });
}

int main() {
    const int T = 4000;
    for (int N : {64, 128, 256, 1024}) {
        const size_t n = (size_t)T * N * 40;
        float *in, *out, *sink;
        CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&sink, 64));
        CK(hipMemset(in, 0, n * 4)); CK(hipMemset(out, 0, n * 4));
        printf("---- T=%d N=%d (%.1f MB) ----\n", T, N, n * 4 / 1e6);
        for (int blocks : {1024, 2048, 4096, 8192}) {
            char name[96];
            snprintf(name, sizeof name, "linear read W=4 depth=8 blocks=%d", blocks);
            timeit(name, n * 4.0, [&] { hipLaunchKernelGGL((linear_kernel<4, 8, false>), dim3(blocks), dim3(256), 0, 0, (const f4 *)in, n / 4, sink); });
        }
        timeit("linear read W=4 depth=8 blocks=2048 nt", n * 4.0, [&] { hipLaunchKernelGGL((linear_kernel<4, 8, true>), dim3(2048), dim3(256), 0, 0, (const f4 *)in, n / 4, sink); });
        timeit("linear read W=8 depth=4 blocks=2048", n * 4.0, [&] { hipLaunchKernelGGL((linear_kernel<8, 4, false>), dim3(2048), dim3(512), 0, 0, (const f4 *)in, n / 4, sink); });
        timeit("linear read W=16 depth=4 blocks=1024", n * 4.0, [&] { hipLaunchKernelGGL((linear_kernel<16, 4, false>), dim3(1024), dim3(1024), 0, 0, (const f4 *)in, n / 4, sink); });
        run_rowset<4, 1, false, false>(in, out, sink, T, N, 32);
        run_rowset<4, 2, false, false>(in, out, sink, T, N, 32);
        run_rowset<4, 2, false, true>(in, out, sink, T, N, 32);
        run_rowset<4, 4, false, false>(in, out, sink, T, N, 32);
        run_rowset<4, 2, false, false>(in, out, sink, T, N, 16);
        run_rowset<4, 2, false, false>(in, out, sink, T, N, 8);
        run_rowset<8, 2, false, false>(in, out, sink, T, N, 32);
        run_rowset<8, 4, false, false>(in, out, sink, T, N, 32);
        run_rowset<8, 1, true, false>(in, out, sink, T, N, 32);
        run_rowset<8, 2, true, false>(in, out, sink, T, N, 32);
        run_rowset<8, 2, true, true>(in, out, sink, T, N, 32);
        run_rowset<8, 4, true, false>(in, out, sink, T, N, 32);
        run_rowset<4, 2, true, false>(in, out, sink, T, N, 32);
        run_rowset<4, 2, true, false>(in, out, sink, T, N, 8);
        run_rowset<4, 2, true, true>(in, out, sink, T, N, 8);
        CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(sink));
    }
    return 0;
}
