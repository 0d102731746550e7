// clip_kernels.hip -- per-parameter gradient maxima and clip-by-value on the device (gfx950).
//
// Replaces apply_clipping (bin/train_flipflop.py:201-212): the reference takes
// float(torch.max(torch.abs(p.grad))) for every parameter tensor -- one reduction kernel and
// one host synchronisation per tensor, ~45 per step -- and clamps the tensors whose maximum
// exceeds its threshold.  Here all gradients live in one flat arena (parallel.FlatGradArena):
//   pass 1  max |g| per segment           (grid = segments x chunks, one atomicMax per block;
//                                          non-negative floats order like their bit patterns)
//   pass 2  clamp to +-thresh[segment]    (only launched once thresholds exist)
// The maxima stay on the device and are copied to the host asynchronously for the rolling
// MAD statistics of the NEXT step's thresholds.  HBM-bound: 4 B (pass 1) + 8 B (pass 2) per
// gradient element.
#include "ff_common.h"

namespace tk {

constexpr int CLIP_THREADS = 256;
constexpr int CLIP_PER_THREAD = 16;          // elements per thread and block: 4096 per block

__global__ __launch_bounds__(CLIP_THREADS) void grad_maxabs_kernel(const float *__restrict__ g,
                                                                   const int64_t *__restrict__ seg_off,
                                                                   float *__restrict__ maxs) {
    const int s = blockIdx.x;
    const int64_t lo = seg_off[s], hi = seg_off[s + 1];
    const int64_t base = lo + (int64_t)blockIdx.y * (CLIP_THREADS * CLIP_PER_THREAD);
    if (base >= hi) return;
    // NaN gradients must surface as NaN maxima (the reference's float(max) would): fmaxf drops
    // them, so carry a flag
    float m = 0.f;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < CLIP_PER_THREAD; ++k) {
        const int64_t i = base + (int64_t)k * CLIP_THREADS + threadIdx.x;
        const float v = g[min(i, hi - 1)];                      // clamped, unconditional load
        m = fmaxf(m, fabsf(v));
        bad |= !(v == v);
    }
    m = wave_allmax_dpp(m);
    __shared__ float part[CLIP_THREADS / WAVE];
    __shared__ int anybad;
    if (threadIdx.x == 0) anybad = 0;
    __syncthreads();
    if (bad) anybad = 1;
    if (lane_id() == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = part[0];
#pragma unroll
        for (int w = 1; w < CLIP_THREADS / WAVE; ++w) r = fmaxf(r, part[w]);
        if (anybad) r = __builtin_nanf("");
        // r >= 0 (or NaN, whose bit pattern 0x7fc00000 exceeds every finite float's)
        atomicMax(reinterpret_cast<unsigned int *>(maxs + s), __float_as_uint(r));
    }
}

__global__ __launch_bounds__(CLIP_THREADS) void grad_clamp_kernel(float *__restrict__ g,
                                                                  const int64_t *__restrict__ seg_off,
                                                                  const float *__restrict__ thresh) {
    const int s = blockIdx.x;
    const float th = thresh[s];
    if (!(th >= 0.f) || !(th < __builtin_huge_valf())) return;        // NaN / inf / negative: no clipping
    const int64_t lo = seg_off[s], hi = seg_off[s + 1];
    const int64_t base = lo + (int64_t)blockIdx.y * (CLIP_THREADS * CLIP_PER_THREAD);
    if (base >= hi) return;
#pragma unroll
    for (int k = 0; k < CLIP_PER_THREAD; ++k) {
        const int64_t i = base + (int64_t)k * CLIP_THREADS + threadIdx.x;
        if (i < hi) g[i] = fminf(fmaxf(g[i], -th), th);          // torch.clamp_(min=-t, max=t)
    }
}

__global__ void maxs_zero_kernel(float *__restrict__ maxs, int nseg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nseg) maxs[i] = 0.f;
}

int grad_clip_dispatch(float *grads, const int64_t *seg_off, size_t nseg, size_t max_seg_len,
                       const float *thresh, float *maxs, hipStream_t stream) {
    if (nseg == 0) return 0;
    const unsigned chunks = (unsigned)((max_seg_len + CLIP_THREADS * CLIP_PER_THREAD - 1) /
                                       (CLIP_THREADS * CLIP_PER_THREAD));
    // (a kernel, not hipMemsetAsync: replayed from a hipGraph on ROCm 7.2 the memset node of this
    // small buffer left garbage in some of its words, which atomicMax then kept for ever)
    hipLaunchKernelGGL(maxs_zero_kernel, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, stream, maxs, (int)nseg);
    hipLaunchKernelGGL(grad_maxabs_kernel, dim3((unsigned)nseg, chunks ? chunks : 1), dim3(CLIP_THREADS), 0,
                       stream, grads, seg_off, maxs);
    if (thresh != nullptr)
        hipLaunchKernelGGL(grad_clamp_kernel, dim3((unsigned)nseg, chunks ? chunks : 1), dim3(CLIP_THREADS), 0,
                           stream, grads, seg_off, thresh);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

}  // namespace tk
