// viterbi_kernels.hip -- flip-flop Viterbi decode for gfx950.
//
// Replaces taiyaki/cupy_extensions/flipflop.py:387-518 and the torch path
// taiyaki/decode.py:75-115, whose tie rule (first / lowest index wins, flop
// ties go to the flip source) and arithmetic (one fp32 add per candidate) are
// reproduced exactly, so fwd, traceback and path are bit-identical.
//
// lane = read (coalesced 10 KiB row-sets, LDS transpose as in logz_kernels.hip);
// the max-plus recursion is serial in T by construction (a time-parallel form
// would re-associate the fp32 adds and break bit-exactness).  The traceback
// pointers of one (t, read) are also packed 4 bits each into one 32-bit word
// of the workspace so the backward pass streams 4 B/lane/step instead of
// chasing pointers through the int64 traceback tensor.
#include "ff_common.h"

namespace tk {

constexpr int VIT_PF = 16;      // traceback words prefetched per batch

template <int NB>
__global__ __launch_bounds__(WAVE) void viterbi_kernel(const float *__restrict__ scores, int T,
                                                      int N, float *__restrict__ fwd_out,
                                                      int64_t *__restrict__ tb_out,
                                                      int64_t *__restrict__ path_out,
                                                      uint32_t *__restrict__ packed, int Npad) {
    using F = FF<NB>;
    static_assert(F::NS <= 8, "4-bit packed traceback holds at most 8 states");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f4 *buf = reinterpret_cast<f4 *>(smem);
    const int lane = lane_id();
    const int n0 = blockIdx.x * WAVE;
    const int nvalid = min(WAVE, N - n0) * F::PIECES;
    const size_t n = (size_t)n0 + lane;
    const bool live = n < (size_t)N;
    const size_t rowstride = (size_t)N * F::S;
    const float *base = scores + (size_t)n0 * F::S;

    float f[F::NS];
#pragma unroll
    for (int s = 0; s < F::NS; ++s) f[s] = (s < NB) ? 0.f : NEG_LARGE;   // decode.py:93-95
    if (fwd_out != nullptr && live) {
#pragma unroll
        for (int s = 0; s < F::NS; ++s) fwd_out[n * F::NS + s] = f[s];
    }

    RowSet<NB> cur, nxt;
    if (T > 0) cur.issue(base, nvalid, lane);
    for (int t = 0; t < T; ++t) {
        nxt.issue(base + (size_t)min(t + 1, T - 1) * rowstride, nvalid, lane);
        cur.to_rows(buf, lane);
        float g[F::NS];
        uint32_t word = 0;
#pragma unroll
        for (int to = 0; to < NB; ++to) {
            // decode.py:99-101: max over `from`, first index wins
            float best = f[0] + cur.get(to * F::NS);
            uint32_t arg = 0;
#pragma unroll
            for (int from = 1; from < F::NS; ++from) {
                const float v = f[from] + cur.get(to * F::NS + from);
                if (v > best) {
                    best = v;
                    arg = from;
                }
            }
            g[to] = best;
            word |= arg << (4 * to);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            // decode.py:102-105: index 0 = from flip b, 1 = flop stay; tie -> flip
            const float u = f[b] + cur.get(F::FLOP0 + b);
            const float v = f[NB + b] + cur.get(F::FLOP0 + NB + b);
            const bool stay = v > u;
            g[NB + b] = stay ? v : u;
            word |= (uint32_t)(stay ? NB + b : b) << (4 * (NB + b));
        }
#pragma unroll
        for (int s = 0; s < F::NS; ++s) f[s] = g[s];
        packed[(size_t)t * Npad + n] = word;
        if (live) {
            if (fwd_out != nullptr) {
#pragma unroll
                for (int s = 0; s < F::NS; ++s) fwd_out[((size_t)(t + 1) * N + n) * F::NS + s] = f[s];
            }
            if (tb_out != nullptr) {
#pragma unroll
                for (int s = 0; s < F::NS; ++s)
                    tb_out[((size_t)t * N + n) * F::NS + s] = (int64_t)((word >> (4 * s)) & 0xFu);
            }
        }
        cur = nxt;
    }

    // traceback (decode.py:108-113); argmax = first maximal index
    uint32_t st = 0;
    float best = f[0];
#pragma unroll
    for (int s = 1; s < F::NS; ++s) {
        if (f[s] > best) {
            best = f[s];
            st = s;
        }
    }
    if (live) path_out[(size_t)T * N + n] = (int64_t)st;
    for (int thi = T; thi > 0; thi -= VIT_PF) {
        uint32_t wd[VIT_PF];
#pragma unroll
        for (int k = 0; k < VIT_PF; ++k) {
            const int t = max(thi - 1 - k, 0);      // clamped, never branched: one straight load run
            wd[k] = packed[(size_t)t * Npad + n];
        }
#pragma unroll
        for (int k = 0; k < VIT_PF; ++k) {
            const int t = thi - 1 - k;
            if (t >= 0) {
                st = (wd[k] >> (4 * st)) & 0xFu;
                if (live) path_out[(size_t)t * N + n] = (int64_t)st;
            }
        }
    }
}

size_t viterbi_workspace_bytes(size_t T, size_t N, size_t nbase) {
    (void)nbase;
    const size_t Npad = (N + WAVE - 1) / WAVE * WAVE;
    return (T > 0 ? T : 1) * Npad * sizeof(uint32_t);
}

template <int NB>
static int viterbi_launch(const float *scores, size_t T, size_t N, float *fwd, int64_t *tb,
                          int64_t *path, void *workspace, hipStream_t stream) {
    using F = FF<NB>;
    const int ncols = (int)((N + WAVE - 1) / WAVE);
    hipLaunchKernelGGL(viterbi_kernel<NB>, dim3(ncols), dim3(WAVE),
                       (size_t)WAVE * F::PIECES * sizeof(f4), stream, scores, (int)T, (int)N, fwd,
                       tb, path, static_cast<uint32_t *>(workspace), ncols * WAVE);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

int viterbi_dispatch(const float *scores, size_t T, size_t N, size_t nbase, float *fwd,
                     int64_t *tb, int64_t *path, void *workspace, size_t workspace_bytes,
                     hipStream_t stream) {
    if (workspace_bytes < viterbi_workspace_bytes(T, N, nbase)) return 3;
    switch (nbase) {
        case 1: return viterbi_launch<1>(scores, T, N, fwd, tb, path, workspace, stream);
        case 2: return viterbi_launch<2>(scores, T, N, fwd, tb, path, workspace, stream);
        case 3: return viterbi_launch<3>(scores, T, N, fwd, tb, path, workspace, stream);
        case 4: return viterbi_launch<4>(scores, T, N, fwd, tb, path, workspace, stream);
        default: return 2;
    }
}

}  // namespace tk
