// viterbi_kernels.hip -- flip-flop Viterbi decode for gfx950.
//
// Replaces taiyaki/cupy_extensions/flipflop.py:387-518 and the torch path
// taiyaki/decode.py:75-115, whose tie rule (first / lowest index wins, flop
// ties go to the flip source) and arithmetic (one fp32 add per candidate) are
// reproduced exactly, so fwd, traceback and path are bit-identical.
//
// EIGHT LANES PER READ: lane (r, j) owns destination state j of read r (8 reads per
// wave), so one step is one 8-way max chain per lane instead of ten per lane, and
// N = 128 reads give 16 waves instead of 2.  The max-plus recursion itself stays
// serial in T by construction (a time-parallel form would re-associate the fp32
// adds and break bit-exactness).  The running vector is all-gathered inside the
// 8-lane group after every step; score rows are prefetched VIT_PF steps ahead.
// The traceback pointers of one (t, read) are packed 4 bits each into one 32-bit
// word of the workspace so the backward pass streams 4 B/read/step instead of
// chasing pointers through the int64 traceback tensor.
#include "ff_common.h"

namespace tk {

constexpr int VIT_GRP = 8;      // lanes per read
constexpr int VIT_PF = 8;       // score rows in flight per lane
constexpr int VIT_TB = 16;      // traceback words prefetched per batch

template <int NB>
__global__ __launch_bounds__(WAVE) void viterbi_kernel(const float *__restrict__ scores, int T,
                                                      int N, float *__restrict__ fwd_out,
                                                      int64_t *__restrict__ tb_out,
                                                      int64_t *__restrict__ path_out,
                                                      uint32_t *__restrict__ packed, int Npad) {
    using F = FF<NB>;
    static_assert(F::NS <= VIT_GRP, "one lane per state, 4-bit packed traceback");
    const int lane = lane_id();
    const int j = lane & (VIT_GRP - 1), rloc = lane >> 3;
    const size_t nreal = (size_t)blockIdx.x * VIT_GRP + rloc;
    const bool live = nreal < (size_t)N && j < F::NS;
    const size_t n = min(nreal, (size_t)N - 1);
    const int jc = min(j, F::NS - 1);
    const bool flip = jc < NB;
    const size_t rowstride = (size_t)N * F::S;
    // every lane reads NS contiguous floats of a score row from its own base (no divergent
    // branch around the loads): flip lane j its candidate block s[j*NS ..], flop lanes the
    // flop block s[FLOP0 ..] (= [from-flip scores | flop-stay scores])
    const float *mine = scores + n * F::S + (flip ? jc * F::NS : F::FLOP0);

    float f[F::NS];                 // the read's full forward vector, replicated in its 8 lanes
#pragma unroll
    for (int s = 0; s < F::NS; ++s) f[s] = (s < NB) ? 0.f : NEG_LARGE;      // decode.py:93-95
    if (fwd_out != nullptr && live) fwd_out[nreal * F::NS + j] = f[jc];

    float sc[VIT_PF][F::NS];
    auto fetch = [&](int t, float (&dst)[F::NS]) {
        const float *row = mine + (size_t)min(t, T - 1) * rowstride;       // clamped, unconditional
        if constexpr (F::NS % 4 == 0) {
            const f4 *rv = reinterpret_cast<const f4 *>(row);               // 16-byte aligned
#pragma unroll
            for (int q = 0; q < F::NS / 4; ++q) {
                const f4 x = rv[q];
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[4 * q + r] = x[r];
            }
        } else {
#pragma unroll
            for (int k = 0; k < F::NS; ++k) dst[k] = row[k];
        }
    };
#pragma unroll
    for (int k = 0; k < VIT_PF; ++k) fetch(k, sc[k]);

    for (int t0 = 0; t0 < T; t0 += VIT_PF) {
#pragma unroll
        for (int k = 0; k < VIT_PF; ++k) {
            const int t = t0 + k;
            if (t < T) {
                float best;
                uint32_t arg;
                if (flip) {
                    // decode.py:99-101: max over `from`, first index wins
                    best = f[0] + sc[k][0];
                    arg = 0;
#pragma unroll
                    for (int from = 1; from < F::NS; ++from) {
                        const float v = f[from] + sc[k][from];
                        if (v > best) {
                            best = v;
                            arg = from;
                        }
                    }
                } else {
                    // decode.py:102-105: index 0 = from flip b, 1 = flop stay; tie -> flip
                    const int bb = jc - NB;
                    float fb = f[0], fs = f[NB], su = sc[k][0], sv = sc[k][NB];
#pragma unroll
                    for (int q = 1; q < NB; ++q) {
                        if (bb == q) {
                            fb = f[q];
                            fs = f[NB + q];
                            su = sc[k][q];
                            sv = sc[k][NB + q];
                        }
                    }
                    const float u = fb + su;
                    const float v = fs + sv;
                    const bool stay = v > u;
                    best = stay ? v : u;
                    arg = stay ? (uint32_t)jc : (uint32_t)bb;
                }
                fetch(t + VIT_PF, sc[k]);
                // all-gather the new vector inside the 8-lane group
#pragma unroll
                for (int s = 0; s < F::NS; ++s) f[s] = __shfl(best, (lane & ~(VIT_GRP - 1)) | s, WAVE);
                // packed traceback word: OR of (arg << 4 j) over the group
                uint32_t word = (j < F::NS) ? (arg << (4 * j)) : 0u;
                word |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)word, 0xB1, 0xF, 0xF, false);
                word |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)word, 0x4E, 0xF, 0xF, false);
                word |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)word, 0x141, 0xF, 0xF, false);
                if (j == 0 && nreal < (size_t)Npad) packed[(size_t)t * Npad + nreal] = word;
                if (live) {
                    if (fwd_out != nullptr) fwd_out[((size_t)(t + 1) * N + nreal) * F::NS + j] = best;
                    if (tb_out != nullptr) tb_out[((size_t)t * N + nreal) * F::NS + j] = (int64_t)arg;
                }
            }
        }
    }

    // traceback (decode.py:108-113); argmax = first maximal index.  One lane per read.
    uint32_t st = 0;
    float best = f[0];
#pragma unroll
    for (int s = 1; s < F::NS; ++s) {
        if (f[s] > best) {
            best = f[s];
            st = s;
        }
    }
    const bool tracer = j == 0 && nreal < (size_t)N;
    if (tracer) path_out[(size_t)T * N + nreal] = (int64_t)st;
    const size_t np = min(nreal, (size_t)Npad - 1);
    for (int thi = T; thi > 0; thi -= VIT_TB) {
        uint32_t wd[VIT_TB];
#pragma unroll
        for (int k = 0; k < VIT_TB; ++k) {
            const int t = max(thi - 1 - k, 0);      // clamped, never branched: one straight load run
            wd[k] = packed[(size_t)t * Npad + np];
        }
#pragma unroll
        for (int k = 0; k < VIT_TB; ++k) {
            const int t = thi - 1 - k;
            if (t >= 0) {
                st = (wd[k] >> (4 * st)) & 0xFu;
                if (tracer) path_out[(size_t)t * N + nreal] = (int64_t)st;
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
    const int ngrp = (int)((N + VIT_GRP - 1) / VIT_GRP);
    const int Npad = (int)((N + WAVE - 1) / WAVE) * WAVE;
    hipLaunchKernelGGL(viterbi_kernel<NB>, dim3(ngrp), dim3(WAVE), 0, stream, scores, (int)T,
                       (int)N, fwd, tb, path, static_cast<uint32_t *>(workspace), Npad);
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
