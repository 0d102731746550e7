// viterbi_kernels.hip -- flip-flop Viterbi decode for gfx950.
//
// Replaces taiyaki/cupy_extensions/flipflop.py:387-518 and the torch path
// taiyaki/decode.py:75-115, whose tie rule (first / lowest index wins, flop
// ties go to the flip source) and arithmetic (one fp32 add per candidate) are
// reproduced exactly, so fwd, traceback and path are bit-identical.
//
// The max-plus recursion is serial in T by construction (a time-parallel form would
// re-associate the fp32 adds and break bit-exactness) and has 2 nbase <= 8 states: it is a
// pure latency problem, so the design minimises the DEPENDENT instructions of a step.
//
// ONE WAVEFRONT PER READ, lane = (to, from) = (lane / 8, lane % 8): every candidate
// f[from] + s[to, from] of a step is one lane's single v_add_f32.  Then
//   * the maximum over `from` is three DPP steps inside the 8-lane group (quad_perm x 2,
//     row_half_mirror), result in all eight lanes;
//   * the new state vector is transposed back (lane (to, from) needs f[from], which group
//     `from` now holds) by ONE ds_bpermute with a constant address;
//   * off the chain: "first index wins" = the lowest set bit of the group's byte of the ballot
//     (candidate == maximum); that index is the traceback byte.
// A step is add -> 3 x v_max_f32_dpp -> ds_bpermute: ~60 ns, against ~270 ns for the
// round-1 layout (8 lanes per read, every lane scanning eight candidates: ~80 instructions
// per step on the chain's wave).  Invalid sources of a flop state are masked in the SCORE
// (-inf), one step ahead of its use; rows are requested VIT_PF steps ahead.
// The path pass decodes 64 steps per batch: every lane fetches the 8-byte traceback word of
// one step (next batch in flight while this one is decoded).  A word IS the table state ->
// previous state, and v_perm_b32 composes two such tables (four entries per instruction), so
// the batch is an inclusive SCAN over the lanes (6 levels) instead of a 64-step walk; the 64
// states go out in one store.
#include <type_traits>

#include "ff_common.h"

namespace tk {

constexpr int VIT_GRP = 8;      // lanes per destination state = source states
constexpr int VIT_PF = 12;      // score rows in flight
constexpr float VIT_NEG_INF = -__builtin_huge_valf();

// maximum over the 8-lane group, in every lane of the group
__device__ __forceinline__ float vit_grp_max(float x) {
    float r;
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(r) : "v"(x));
    return r;
}

template <int NB, bool FULLOUT>
__global__ __launch_bounds__(WAVE) void viterbi_kernel(const float *__restrict__ scores, int T,
                                                      int N, float *__restrict__ fwd_out,
                                                      int64_t *__restrict__ tb_out,
                                                      int64_t *__restrict__ path_out,
                                                      unsigned char *__restrict__ packed, int npad) {
    using F = FF<NB>;
    static_assert(F::NS <= VIT_GRP, "one lane group per state");
    const int lane = lane_id();
    const int to = lane >> 3, from = lane & 7;
    const int n = blockIdx.x;
    // decode.py:99-105: a flip state is reached from every state, flop b only from flip b and
    // from itself
    const bool flip = to < NB;
    const bool valid = to < F::NS && from < F::NS && (flip || from == to - NB || from == to);
    const int sidx = flip ? to * F::NS + min(from, F::NS - 1) : F::FLOP0 + min(from, F::NS - 1);
    const bool leader = from == 0 && to < F::NS;
    const size_t rowstride = (size_t)N * F::S;
    const int tr4 = 4 * ((from << 3) | to);                     // transpose: the lane of group `from`

    float f = (from < NB) ? 0.f : ((from < F::NS) ? NEG_LARGE : VIT_NEG_INF);      // decode.py:93-95
    if (FULLOUT && leader) fwd_out[(size_t)n * F::NS + to] = (to < NB) ? 0.f : NEG_LARGE;

    // Memory goes through buffer instructions: a descriptor per group of VIT_PF steps (scalar
    // ALU), a constant per-lane offset, a scalar per-step offset -- no vector address arithmetic
    // among the ~15 instructions of a step.
    constexpr int RSRC3 = 0x00027000;
    const unsigned rs4 = 4u * (unsigned)rowstride;
    const unsigned lane_ld4 = 4u * (unsigned)((size_t)n * F::S + min(sidx, F::S - 1));
    float sc[VIT_PF];
    auto fetch_group = [&](int t0, int k0, int k1) {            // rows t0 + k0 .. t0 + k1 - 1 -> sc[k]
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(scores + (size_t)min(t0, max(T - 1, 0)) * rowstride), 0, 0x7fffffff, RSRC3);
        const int last = max(T - 1 - min(t0, max(T - 1, 0)), 0);        // rows past the end re-read the last one
#pragma unroll
        for (int k = 0; k < VIT_PF; ++k)
            if (k >= k0 && k < k1)
                sc[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, lane_ld4, rs4 * (unsigned)min(k, last), 0));
    };
    fetch_group(0, 0, VIT_PF);

    const size_t pstride = (size_t)npad * VIT_GRP;               // traceback bytes [t][npad][8]
    const unsigned lane_tb = (unsigned)((size_t)n * VIT_GRP + to);
    const unsigned lane_o4 = 4u * (unsigned)((size_t)n * F::NS + min(to, F::NS - 1));
    const size_t ostride = (size_t)N * F::NS;

    float m = VIT_NEG_INF;
    float snext = valid ? sc[0] : VIT_NEG_INF;
    // every lane of a group holds the group's result: all eight store it (same byte, same address)
    // instead of one leader lane behind an exec-mask branch; only alphabets with dead groups
    // (2 nbase < 8) need the mask
    constexpr bool ALL_GROUPS_LIVE = F::NS == VIT_GRP;
    // One group of up to VIT_PF steps, straight-line: with a branch between the steps the
    // compiler can no longer count the loads in flight and waits for the newest but one -- a
    // memory round trip per step.  The rows of the NEXT group are requested as their registers
    // fall free.
    auto group = [&](int t0, auto full) {
        constexpr bool FULL = decltype(full)::value;
        const __amdgpu_buffer_rsrc_t rnext = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(scores + (size_t)min(t0 + VIT_PF, max(T - 1, 0)) * rowstride), 0, 0x7fffffff, RSRC3);
        const int lastn = max(T - 1 - min(t0 + VIT_PF, max(T - 1, 0)), 0);
        const __amdgpu_buffer_rsrc_t rtb = __builtin_amdgcn_make_buffer_rsrc(
            packed + (size_t)t0 * pstride, 0, 0x7fffffff, RSRC3);
        const __amdgpu_buffer_rsrc_t rfo = __builtin_amdgcn_make_buffer_rsrc(
            FULLOUT ? fwd_out + (size_t)(t0 + 1) * ostride : nullptr, 0, 0x7fffffff, RSRC3);
        int64_t *tout = FULLOUT ? tb_out + (size_t)t0 * ostride + (size_t)n * F::NS + min(to, F::NS - 1) : nullptr;
#pragma unroll
        for (int k = 0; k < VIT_PF; ++k) {
            if (FULL || t0 + k < T) {
                const float cand = f + snext;
                m = vit_grp_max(cand);
                f = __int_as_float(__builtin_amdgcn_ds_bpermute(tr4, __float_as_int(m)));
                // ---- off the chain
                sc[k] = __uint_as_float(
                    __builtin_amdgcn_raw_buffer_load_b32(rnext, lane_ld4, rs4 * (unsigned)min(k, lastn), 0));
                // the next step's scores, masked one step ahead of their use (requested VIT_PF - 1
                // steps ago; at the last step of a group: row 0 of the next, requested at step 0)
                snext = valid ? sc[(k + 1) % VIT_PF] : VIT_NEG_INF;
                // "first index wins": the lowest set bit of the group's byte of the ballot
                // (candidate == maximum) is the traceback byte
                const unsigned long long eq = __ballot(cand == m);
                const unsigned bits = (unsigned)(eq >> (8 * to));
                const unsigned arg = (unsigned)__builtin_ctz((bits & 0xffu) | 0x100u) & 7u;
                if (ALL_GROUPS_LIVE || to < F::NS) {
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)arg, rtb, lane_tb,
                                                         (unsigned)(k * pstride), 0);
                    if (FULLOUT) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m), rfo, lane_o4,
                                                              4u * (unsigned)(k * ostride), 0);
                        tout[(size_t)k * ostride] = (int64_t)arg;
                    }
                }
            }
        }
    };
    const int tfull = T / VIT_PF * VIT_PF;
    for (int t0 = 0; t0 < tfull; t0 += VIT_PF) group(t0, std::true_type{});
    if (tfull < T) group(tfull, std::false_type{});

#ifdef TK_VIT_NOPATH
    return;                                     // lab: forward pass alone
#endif
    // ---- traceback (decode.py:108-113); argmax = first maximal state
    unsigned st = 0;
    {
        float top = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 0));
        if (T == 0) top = 0.f;
#pragma unroll
        for (int s = 1; s < F::NS; ++s) {
            float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), s * VIT_GRP));
            if (T == 0) v = (s < NB) ? 0.f : NEG_LARGE;
            if (v > top) {
                top = v;
                st = s;
            }
        }
    }
    if (lane == 0) path_out[(size_t)T * N + n] = (int64_t)st;
    // the eight bytes of a read at a step are one 64-bit word: no dependent address
    const unsigned long long *words = reinterpret_cast<const unsigned long long *>(packed) + n;
    auto load_batch = [&](int thi) {                            // lane k: the word of step thi - 1 - k
        const int t = max(thi - 1 - lane, 0);                   // clamped, never branched
        return words[(size_t)t * npad];
    };
    // four batches in flight (a batch's scan is ~0.3 us, a load round trip a multiple of that);
    // the ring rotates by NAME -- the loop is unrolled four times -- because a register copy of a
    // word in flight would wait for its load
    constexpr int VIT_TBQ = 4;
    unsigned long long q[VIT_TBQ];
#pragma unroll
    for (int b = 0; b < VIT_TBQ; ++b) q[b] = load_batch(T - b * WAVE);
    // The eight bytes of a step's word are the table  state -> previous state,  and v_perm_b32
    // with a table as byte selector COMPOSES two tables, four entries at a time (selectors 0-3
    // pick bytes of the low dword, 4-7 of the high one).  So the 64 steps of a batch are not walked
    // one after the other: an inclusive scan over the lanes (six levels, two v_perm_b32 + two
    // ds_bpermute each) leaves in lane k the composition of steps 0..k, and one more look-up with
    // the incoming state gives every lane its own state.  The scans of the four batches in flight
    // do not depend on each other (only that last look-up chains them): they are issued together
    // so that one's ds_bpermute round trips hide behind the others'.
    auto scan = [&](unsigned long long cur, unsigned &lo, unsigned &hi) {
        lo = (unsigned)cur;
        hi = (unsigned)(cur >> 32);
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const int src4 = 4 * (lane - d);                    // (wraps for lane < d: not used there)
            const unsigned blo = (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)lo);
            const unsigned bhi = (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)hi);
            // (this step's table) o (the steps before): entry s = mine[theirs[s]]
            const unsigned nlo = __builtin_amdgcn_perm(hi, lo, blo), nhi = __builtin_amdgcn_perm(hi, lo, bhi);
            if (lane >= d) {
                lo = nlo;
                hi = nhi;
            }
        }
    };
    for (int thi = T; thi > 0; thi -= VIT_TBQ * WAVE) {
        unsigned lo[VIT_TBQ], hi[VIT_TBQ];
#pragma unroll
        for (int b = 0; b < VIT_TBQ; ++b) {
            const unsigned long long cur = q[b];
            q[b] = load_batch(thi - (b + VIT_TBQ) * WAVE);
            scan(cur, lo[b], hi[b]);
        }
#pragma unroll
        for (int b = 0; b < VIT_TBQ; ++b) {
            const int th = thi - b * WAVE;
            if (th > 0) {                                        // wave-uniform
                const unsigned mine = __builtin_amdgcn_perm(hi[b], lo[b], st) & 0xffu;     // st: the batch's start state
                const int t = th - 1 - lane;
                if (t >= 0) path_out[(size_t)t * N + n] = (int64_t)mine;
                // (lanes before the start of the read re-read step 0; the walk ends with this batch then)
                st = (unsigned)__builtin_amdgcn_readlane((int)mine, WAVE - 1);
            }
        }
    }
}

size_t viterbi_workspace_bytes(size_t T, size_t N, size_t nbase) {
    (void)nbase;
    const size_t npad = (N + VIT_GRP - 1) / VIT_GRP * VIT_GRP;
    return (T > 0 ? T : 1) * npad * VIT_GRP;        // one byte per state and step
}

template <int NB>
static int viterbi_launch(const float *scores, size_t T, size_t N, float *fwd, int64_t *tb,
                          int64_t *path, void *workspace, hipStream_t stream) {
    const int npad = (int)((N + VIT_GRP - 1) / VIT_GRP * VIT_GRP);
    if (N == 0) return 0;
    if (fwd != nullptr && tb != nullptr)
        hipLaunchKernelGGL((viterbi_kernel<NB, true>), dim3((unsigned)N), dim3(WAVE), 0, stream, scores, (int)T,
                           (int)N, fwd, tb, path, static_cast<unsigned char *>(workspace), npad);
    else
        hipLaunchKernelGGL((viterbi_kernel<NB, false>), dim3((unsigned)N), dim3(WAVE), 0, stream, scores, (int)T,
                           (int)N, fwd, tb, path, static_cast<unsigned char *>(workspace), npad);
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
