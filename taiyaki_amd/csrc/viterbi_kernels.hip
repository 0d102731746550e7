// viterbi_kernels.hip -- flip-flop Viterbi decode for gfx950.
//
// Replaces taiyaki/cupy_extensions/flipflop.py:387-518 and the torch path
// taiyaki/decode.py:75-115, whose tie rule (first / lowest index wins, flop
// ties go to the flip source) and arithmetic (one fp32 add per candidate) are
// reproduced exactly, so fwd, traceback and path are bit-identical.
//
// EIGHT LANES PER READ: lane (r, j) owns destination state j of read r (8 reads per
// wave).  The max-plus recursion stays serial in T by construction (a time-parallel form
// would re-associate the fp32 adds and break bit-exactness), so the work is in making one
// step short:
//   * every lane evaluates the same eight candidates f[from] + s[from]; a flop lane's
//     invalid sources are masked to -inf (when the row is used, not when it is fetched: the
//     loads stay in flight for VIT_PF steps), so there is no flip / flop divergence;
//   * the all-gather of the eight state values is folded into the adds: the value of lane
//     4q+k of the lane's own quad is a quad_perm DPP operand (v_add_f32_dpp), the other
//     quad comes from one lane^4 exchange (two bank-masked row shifts) -- no ds_bpermute;
//   * "first index wins" is kept by scanning each quad's four candidates in order and
//     letting the lower quad win ties;
//   * every lane drops its source index as ONE BYTE (a single 64-byte store per wave and
//     step); the eight bytes of a read are one 64-bit word, so the path pass reads them
//     with prefetched, address-independent loads instead of chasing the int64 tensor.
// Score rows are prefetched VIT_PF steps ahead.
#include <type_traits>

#include "ff_common.h"

namespace tk {

constexpr int VIT_GRP = 8;      // lanes per read
constexpr int VIT_PF = 8;       // score rows in flight per lane
constexpr int VIT_TB = 16;      // traceback steps prefetched per batch
constexpr float VIT_NEG_INF = -__builtin_huge_valf();

// lane l <- lane l ^ 4 (inside every 8-lane group): two bank-masked row shifts
__device__ __forceinline__ float xor4_f32(float x) {
    int r = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x104, 0xF, 0x5, false);   // row_shl:4 -> lanes 0-3, 8-11
    r = __builtin_amdgcn_update_dpp(r, __float_as_int(x), 0x114, 0xF, 0xA, false);       // row_shr:4 -> lanes 4-7, 12-15
    return __int_as_float(r);
}

// c[k] = s[k] + x[lane 4q + k] (k < 4), c[4 + k] = s[4 + k] + xo[lane 4q + k]
__device__ __forceinline__ void vit_candidates(const float (&s)[8], float x, float xo, float (&c)[8]) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %10 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %8, %11 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %8, %12 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %8, %13 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %9, %14 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %9, %15 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %9, %16 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %7, %9, %17 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
        : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5]), "=&v"(c[6]),
          "=&v"(c[7])
        : "v"(x), "v"(xo), "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]),
          "v"(s[7]));
}

template <int NB>
__global__ __launch_bounds__(WAVE) void viterbi_kernel(const float *__restrict__ scores, int T,
                                                      int N, float *__restrict__ fwd_out,
                                                      int64_t *__restrict__ tb_out,
                                                      int64_t *__restrict__ path_out,
                                                      unsigned char *__restrict__ packed) {
    using F = FF<NB>;
    static_assert(F::NS <= VIT_GRP, "one lane per state");
    const int lane = lane_id();
    const int j = lane & (VIT_GRP - 1), rloc = lane >> 3, q = j >> 2;
    const size_t nreal = (size_t)blockIdx.x * VIT_GRP + rloc;
    const bool live = nreal < (size_t)N && j < F::NS;
    const size_t n = min(nreal, (size_t)N - 1);
    const int jc = min(j, F::NS - 1);
    const bool flip = jc < NB;
    const size_t rowstride = (size_t)N * F::S;
    // every lane reads the NS contiguous floats of its destination's block (no divergent
    // branch around the loads): flip lane j the block s[j*NS ..], flop lanes the flop block
    // s[FLOP0 ..] = [from-flip scores | flop-stay scores].  All global accesses are a
    // wave-uniform row pointer (scalar registers) plus a 32-bit lane offset.
    const int mine = (int)n * F::S + (flip ? jc * F::NS : F::FLOP0);
    const int slot = (int)min(nreal, (size_t)N - 1) * F::NS + jc;       // lane's element of an (N, NS) row
    // register word w pairs with source state src[w]: the lane's own quad first (4q + w),
    // then the other quad (4 (1 - q) + w - 4)
    int src[8];
    bool ok[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        src[w] = (w < 4) ? 4 * q + w : 4 * (1 - q) + (w - 4);
        // decode.py:99-105: a flip state is reached from every state, flop b only from flip b
        // and from itself
        ok[w] = src[w] < F::NS && (flip || src[w] == jc - NB || src[w] == jc);
    }

    const int base_own = 4 * q, base_other = 4 * (1 - q);
    float best = (j < NB) ? 0.f : ((j < F::NS) ? NEG_LARGE : VIT_NEG_INF);      // decode.py:93-95
    if (fwd_out != nullptr && live) fwd_out[slot] = best;

    float sc[VIT_PF][8];
    // running wave-uniform pointers (scalar adds per step, no index multiplications)
    const float *frow = scores;                 // row being fetched (stops at row T - 1)
    int tfetch = 0;
    // the fetch only issues loads: anything computed on the loaded words here would make the
    // wave wait for the row it has just requested (one L2 round trip per step)
    auto fetch = [&](float (&dst)[8]) {
        if constexpr (F::NS == 8) {
            // the two float4 halves of the block, own quad's sources first
            const f4 a = *reinterpret_cast<const f4 *>(frow + (mine + 4 * q));
            const f4 b = *reinterpret_cast<const f4 *>(frow + (mine + 4 * (1 - q)));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dst[k] = a[k];
                dst[4 + k] = b[k];
            }
        } else {
#pragma unroll
            for (int w = 0; w < 8; ++w) dst[w] = frow[mine + min(src[w], F::NS - 1)];
        }
        // clamped, never branched on: past the end the last row is fetched again
        frow += (tfetch < T - 1) ? rowstride : 0;
        ++tfetch;
    };
#pragma unroll
    for (int k = 0; k < VIT_PF; ++k) fetch(sc[k]);

    const int wave_row = blockIdx.x;            // 8 reads per wave
    const size_t nwaves = gridDim.x;
    // traceback bytes: one per lane (= destination state) and step, [t][wave][64]
    unsigned char *prow = packed + (size_t)wave_row * WAVE;
    float *fout = fwd_out != nullptr ? fwd_out + (size_t)N * F::NS : nullptr;        // row t + 1
    int64_t *tout = tb_out;
    // The loop body is instantiated per (outputs wanted, every lane live): the per-step
    // "is this pointer null / is this lane live" tests are loop invariants, but left inside
    // they cost a scalar compare + branch + exec-mask round trip each, every step.
    auto steps = [&](auto want_full, auto all_live) {
        constexpr bool FULLOUT = decltype(want_full)::value, ALLLIVE = decltype(all_live)::value;
        for (int t0 = 0; t0 < T; t0 += VIT_PF) {
#pragma unroll
            for (int k = 0; k < VIT_PF; ++k) {
                const int t = t0 + k;
                if (t < T) {
                    // a flop lane's invalid sources are masked when the row is USED, VIT_PF
                    // steps after it was requested
                    float sm[8], c[8];
#pragma unroll
                    for (int w = 0; w < 8; ++w) sm[w] = ok[w] ? sc[k][w] : VIT_NEG_INF;
                    vit_candidates(sm, best, xor4_f32(best), c);
                    fetch(sc[k]);
                    // first maximum of each quad's four candidates, in source order
                    float bo = c[0], bt = c[4];
                    int ao = 0, at = 0;
#pragma unroll
                    for (int i = 1; i < 4; ++i) {
                        if (c[i] > bo) {
                            bo = c[i];
                            ao = i;
                        }
                        if (c[4 + i] > bt) {
                            bt = c[4 + i];
                            at = i;
                        }
                    }
                    // the quad holding the lower source indices wins ties (own quad for q = 0)
                    const bool take_other = (q == 0) ? (bt > bo) : !(bo > bt);
                    best = take_other ? bt : bo;
                    const int arg = (take_other ? at : ao) + (take_other ? base_other : base_own);
                    prow[lane] = (unsigned char)arg;        // one 64-byte store per wave and step
                    prow += nwaves * WAVE;
                    if constexpr (FULLOUT) {
                        if (ALLLIVE || live) {
                            fout[slot] = best;
                            tout[slot] = (int64_t)arg;
                        }
                        fout += (size_t)N * F::NS;
                        tout += (size_t)N * F::NS;
                    }
                }
            }
        }
    };
    const bool every_lane_live = __all(live);
    if (fwd_out != nullptr && tb_out != nullptr) {
        if (every_lane_live) steps(std::true_type{}, std::true_type{});
        else steps(std::true_type{}, std::false_type{});
    } else {
        steps(std::false_type{}, std::true_type{});
    }

    // traceback (decode.py:108-113); argmax = first maximal index.  One lane per read; the
    // group's eight final values are gathered once.
    float f[F::NS];
#pragma unroll
    for (int s = 0; s < F::NS; ++s) f[s] = __shfl(best, (lane & ~(VIT_GRP - 1)) | s, WAVE);
    uint32_t st = 0;
    float top = f[0];
#pragma unroll
    for (int s = 1; s < F::NS; ++s) {
        if (f[s] > top) {
            top = f[s];
            st = s;
        }
    }
    const bool tracer = j == 0 && nreal < (size_t)N;
    if (tracer) path_out[(size_t)T * N + nreal] = (int64_t)st;
    // the eight bytes of this read's group at a step are one 64-bit word: no dependent address
    const unsigned long long *words = reinterpret_cast<const unsigned long long *>(packed) +
                                      (size_t)wave_row * VIT_GRP + rloc;
    const size_t wstride = nwaves * VIT_GRP;
    for (int thi = T; thi > 0; thi -= VIT_TB) {
        unsigned long long wd[VIT_TB];
#pragma unroll
        for (int k = 0; k < VIT_TB; ++k) {
            const int t = max(thi - 1 - k, 0);      // clamped, never branched: one straight load run
            wd[k] = words[(size_t)t * wstride];
        }
#pragma unroll
        for (int k = 0; k < VIT_TB; ++k) {
            const int t = thi - 1 - k;
            if (t >= 0) {
                st = (uint32_t)(wd[k] >> (8 * st)) & 7u;
                if (tracer) path_out[(size_t)t * N + nreal] = (int64_t)st;
            }
        }
    }
}

size_t viterbi_workspace_bytes(size_t T, size_t N, size_t nbase) {
    (void)nbase;
    const size_t nwaves = (N + VIT_GRP - 1) / VIT_GRP;
    return (T > 0 ? T : 1) * nwaves * WAVE;         // one byte per lane and step
}

template <int NB>
static int viterbi_launch(const float *scores, size_t T, size_t N, float *fwd, int64_t *tb,
                          int64_t *path, void *workspace, hipStream_t stream) {
    const int ngrp = (int)((N + VIT_GRP - 1) / VIT_GRP);
    hipLaunchKernelGGL(viterbi_kernel<NB>, dim3(ngrp), dim3(WAVE), 0, stream, scores, (int)T,
                       (int)N, fwd, tb, path, static_cast<unsigned char *>(workspace));
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
