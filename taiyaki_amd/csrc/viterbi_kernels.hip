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
    // (round 6, tried: the three levels through __builtin_amdgcn_update_dpp + fmaxf so that the compiler may fill the wait slots --
    // it does not fuse them: mov, mov_dpp, canonicalising max, max per level, 475 more instructions per kernel.  The asm stays.)
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

// vit_grp_max with the two wait slots in front of its first two DPP reads FILLED (round 6): the invalid-source masks of the
// next two steps' scores (v_cndmask: -inf where the lane's (to, from) pair does not exist) take the place of an s_nop each --
// two issue slots less per pair of steps on a chain that is bound by its instruction count.  Same values, same order.
// (used by the five-wave form only: beside ONE trace wave per parity the faster chain gains nothing -- a tile's sixteen steps then
// take it less than that trace wave needs for its eight, and the three-wave kernel measures 196 against 189 us with it)
__device__ __forceinline__ float vit_grp_max_fill(float x, float s1, unsigned long long m1, float s2, unsigned long long m2,
                                                  float &o1, float &o2) {
    float r;
    const float ninf = VIT_NEG_INF;
    asm("v_cndmask_b32_e64 %1, %4, %5, %7\n\t"
        "s_nop 0\n\t"
        "v_max_f32_dpp %0, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %2, %4, %6, %8\n\t"
        "s_nop 0\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(r), "=&v"(o1), "=&v"(o2)
        : "v"(x), "v"(ninf), "v"(s1), "v"(s2), "s"(m1), "s"(m2));
    return r;
}

// ---- traceback (decode.py:108-113); argmax = first maximal state.  `m`: the last step's maxima as the
// forward pass left them -- state s in group s (one-wave kernel, and the alternating layout after an even
// last step, i.e. T odd) or in lane s of every group (alternating layout, T even).
template <int NB>
__device__ __forceinline__ void viterbi_path_pass(float m, int T, int N, int n, int lane, int64_t *__restrict__ path_out,
                                                  const unsigned char *__restrict__ packed, int npad, bool alternating) {
    using F = FF<NB>;
    // ---- traceback (decode.py:108-113); argmax = first maximal state
    unsigned st = 0;
    {
        const int per_state = (alternating && (T & 1) == 0) ? 1 : VIT_GRP;
        float top = __shfl(m, 0, WAVE);
        if (T == 0) top = 0.f;
#pragma unroll
        for (int s = 1; s < F::NS; ++s) {
            float v = __shfl(m, s * per_state, WAVE);
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
    viterbi_path_pass<NB>(m, T, N, n, lane, path_out, packed, npad, /*alternating=*/false);
}


// ---- the traceback walk on THREE waves (round 6).  The one-wave walk above scans batches of 64 steps one after the
// other (four in flight): 22 of the kernel's 206 us at T 4000.  A batch's scan needs nothing from the batches before it --
// only the LOOK-UP with the batch's start state chains them.  So: (A) the waves take the batches of a segment (64 batches =
// 4096 steps) in turn, scan them and leave every lane's table in LDS (the forward pass's ring, free by now: 64 x 64 x 8 B);
// (B) wave 0 scans the 64 WHOLE-batch tables once more -- the same composition, one level up -- which gives every batch its
// start state; (C) the waves look their batches' states up and store them.  Integer table composition throughout: the same
// path, bit for bit.  `tabs`: 64 x 64 eight-byte words of LDS; `stsh`: 66 words.
template <int NB>
__device__ __forceinline__ void viterbi_path_pass3(float m, int T, int N, int n, int lane, int wave, int nwaves,
                                                   int64_t *__restrict__ path_out, const unsigned char *__restrict__ packed,
                                                   int npad, unsigned long long *tabs, unsigned *stsh) {
    using F = FF<NB>;
    if (wave == 0) {
        // ---- decode.py:108-113: argmax of the last column = first maximal state (alternating layout: see viterbi_path_pass)
        unsigned st = 0;
        const int per_state = ((T & 1) == 0) ? 1 : VIT_GRP;
        float top = __shfl(m, 0, WAVE);
        if (T == 0) top = 0.f;
#pragma unroll
        for (int s = 1; s < F::NS; ++s) {
            float v = __shfl(m, s * per_state, WAVE);
            if (T == 0) v = (s < NB) ? 0.f : NEG_LARGE;
            if (v > top) {
                top = v;
                st = s;
            }
        }
        if (lane == 0) {
            path_out[(size_t)T * N + n] = (int64_t)st;
            stsh[0] = st;
        }
    }
    const unsigned long long *words = reinterpret_cast<const unsigned long long *>(packed) + n;
    auto load_batch = [&](int thi) {                            // lane k: the word of step thi - 1 - k
        const int t = max(thi - 1 - lane, 0);                   // clamped, never branched
        return words[(size_t)t * npad];
    };
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
    constexpr int SEG = WAVE;                                   // batches per segment
    constexpr int PIPE = 3;                                     // batches of a wave in flight
    for (int thi_seg = T; thi_seg > 0; thi_seg -= SEG * WAVE) {
        const int nb = min(SEG, (thi_seg + WAVE - 1) / WAVE);
        __syncthreads();                                        // stsh[0] is there; the segment before is done with `tabs`
        // (A) every lane's table of this wave's batches
        for (int b0 = wave; b0 < nb; b0 += PIPE * nwaves) {
            unsigned long long q[PIPE];
#pragma unroll
            for (int u = 0; u < PIPE; ++u) q[u] = load_batch(thi_seg - min(b0 + u * nwaves, nb - 1) * WAVE);
            unsigned lo[PIPE], hi[PIPE];
#pragma unroll
            for (int u = 0; u < PIPE; ++u) scan(q[u], lo[u], hi[u]);
#pragma unroll
            for (int u = 0; u < PIPE; ++u)
                if (b0 + u * nwaves < nb)                       // (wave-uniform)
                    tabs[(size_t)(b0 + u * nwaves) * WAVE + lane] = ((unsigned long long)hi[u] << 32) | lo[u];
        }
        __syncthreads();
        // (B) the start state of every batch: a scan over the whole-batch tables (lane = batch)
        if (wave == 0) {
            const unsigned long long whole = (lane < nb) ? tabs[(size_t)lane * WAVE + (WAVE - 1)] : 0x0706050403020100ull;
            unsigned lo, hi;
            scan(whole, lo, hi);
            const unsigned st_in = stsh[0];
            const unsigned after = __builtin_amdgcn_perm(hi, lo, st_in) & 0xffu;   // the state the batch hands to the next one
            stsh[1 + lane] = after;                             // batch lane + 1 starts there
        }
        __syncthreads();
        // (C) every step's state
        for (int b = wave; b < nb; b += nwaves) {
            const unsigned long long tb = tabs[(size_t)b * WAVE + lane];
            const unsigned st = stsh[b];
            const unsigned mine = __builtin_amdgcn_perm((unsigned)(tb >> 32), (unsigned)tb, st) & 0xffu;
            const int t = thi_seg - b * WAVE - 1 - lane;
            if (t >= 0) path_out[(size_t)t * N + n] = (int64_t)mine;
        }
        __syncthreads();
        if (wave == 0 && lane == 0) stsh[0] = stsh[nb];         // the next segment starts where this one ended
    }
}

// ===========================================================================
// Round 5: THREE WAVES PER READ.  The one-wave kernel above spends ~30 instructions per step on one
// wave's issue stream, a third of them the traceback byte (ballot, shifts, find-first, store) and the
// full-output stores -- none of which the recursion depends on.  Here wave 0 runs ONLY the recursion,
// in the alternating layout (LABNOTES R4.5: even steps lane = (to, from) with the maximum inside the
// 8-lane group, odd steps lane = (from, to) with the maximum across the groups -- row_ror:8 and gfx950's
// v_permlane16_swap / v_permlane32_swap -- so every lane already HOLDS the state it needs next and no
// ds_bpermute round trip sits on the chain; alone that change measured nothing, because the step was
// bound by its instruction count), and leaves each step's state vector in an LDS ring.  Waves 1 and 2
// follow a tile of 16 steps behind (even / odd steps, a fixed layout each): they take the step's candidates
// and maxima from the ring (two LDS words per lane and step -- the chain wave's own values, so the same
// bits), compare, and write the traceback byte (and fwd / traceback of the reference's full-output form);
// they touch no score row.  One s_barrier per 16 steps hands a tile over; the score rows are
// requested 64 steps ahead (one wave per SIMD: registers are free).  Arithmetic, tie rule and outputs
// are those of the one-wave kernel, bit for bit.
// ===========================================================================
#ifndef TK_VIT_TILE
#define TK_VIT_TILE 16
#endif
#ifndef TK_VIT_GROUP
#define TK_VIT_GROUP 64
#endif
constexpr int VIT_TILE = TK_VIT_TILE;       // steps per hand-over (one s_barrier)
// trace waves per step parity (round 6): 1 = three waves per read (round 5); 2 = five -- with the chain wave's wait slots filled
// (vit_grp_max_fill) a tile's sixteen steps take it ~1260 cycles, less than ONE trace wave needs for its eight steps of the tile
// (a template parameter of the kernel: five waves per read oversubscribe the SIMDs once several reads share a CU -- viterbi_launch)
__host__ __device__ constexpr int vit_waves(int split) { return 1 + 2 * split; }
// (round 6, measured and not kept: a SIXTH wave that ends at once, so that -- if a workgroup's waves go to the CU's four SIMDs in
// turn -- the fifth does not share the chain wave's SIMD: 181.4 against 180.5 us at T 4000 / N 256, 221 against 215 at N 512;
// tiles of 32 steps instead of 16: 179.6 against 180.3; of 8: 187; the ODD step's wait slots filled with the scalar row offsets of
// the next loads (s_min_i32 + s_mul_i32 in place of two s_nop 1; mind the SCC clobber): 178.5 against 179.9 -- within the noise:
// with five waves the kernel is no longer bound by the chain wave alone -- profiles/r6_viterbi_fill_split_ab.txt)
#ifndef TK_VIT_RING2
#define TK_VIT_RING2 1
#endif
constexpr int VIT_GROUP = TK_VIT_GROUP;     // steps of straight-line code = score rows in flight = the LDS ring
constexpr int VIT_RING = VIT_GROUP / VIT_TILE;  // tiles in the ring: at least the one being written and the one being read

// maximum over the eight lanes that share lane % 8 (one per 8-lane group), in all of them: within a
// 16-lane row by DPP, across the rows and halves by the lane-swap instructions of gfx950.  A swap
// exchanges halves of TWO registers in place, so max(x, swapped x) needs x twice: the copies are made by
// issuing the level's maximum twice (independent instructions) instead of a v_mov behind it -- two
// dependent hops less on the recursion's chain (the intrinsic form: max, mov, swap, max, mov, swap, max).
#ifndef TK_VIT_DUPMAX
#define TK_VIT_DUPMAX 1
#endif
__device__ __forceinline__ float vit_cross_max(float x) {
#if TK_VIT_DUPMAX
    float a, b, c;
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %1\n\t"
        "v_max_f32 %2, %0, %1\n\t"
        "v_max_f32 %0, %0, %1\n\t"
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 %2, %0\n\t"
        "v_max_f32 %0, %2, %0"
        : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(x));
    return a;
#else
    float r;
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf"
        : "=&v"(r) : "v"(x));
    {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(r), __float_as_uint(r), false, false);
        asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(__uint_as_float(sw[0])), "v"(__uint_as_float(sw[1])));
    }
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(r), __float_as_uint(r), false, false);
        asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(__uint_as_float(sw[0])), "v"(__uint_as_float(sw[1])));
    }
    return r;
#endif
}

__device__ __forceinline__ void vit_handover() {
    // (LDS traffic only: the chain wave's score loads and the trace waves' stores stay in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NB, bool FULLOUT, int VIT_TRACE_SPLIT>
__global__ __launch_bounds__(vit_waves(VIT_TRACE_SPLIT) * WAVE) void viterbi3_kernel(const float *__restrict__ scores, int T,
                                                           int N, float *__restrict__ fwd_out,
                                                           int64_t *__restrict__ tb_out,
                                                           int64_t *__restrict__ path_out,
                                                           unsigned char *__restrict__ packed, int npad) {
    using F = FF<NB>;
    static_assert(F::NS <= VIT_GRP, "one lane group per state");
    static_assert(VIT_GROUP == VIT_RING * VIT_TILE && VIT_RING >= 2 && VIT_TILE % 2 == 0, "the ring holds one group; tiles start with an even step");
    static_assert(VIT_RING * VIT_TILE * 2 * WAVE * sizeof(float) >= (size_t)WAVE * WAVE * sizeof(unsigned long long), "the traceback walk keeps a segment's tables in the ring");
    __shared__ __attribute__((aligned(16))) float ring[VIT_RING * VIT_TILE * 2 * WAVE];      // per step: the maxima (the new state vector) and the candidates
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = lane >> 3, sub = lane & 7;          // even steps: (to, from) = (grp, sub); odd steps: (sub, grp)
    const int n = blockIdx.x;
    // decode.py:99-105: a flip state is reached from every state, flop b only from flip b and from itself
    auto valid_of = [&](int to, int from) {
        return to < F::NS && from < F::NS && (to < NB || from == to - NB || from == to);
    };
    auto sidx_of = [&](int to, int from) {
        return (to < NB) ? to * F::NS + min(from, F::NS - 1) : F::FLOP0 + min(from, F::NS - 1);
    };
    const bool validA = valid_of(grp, sub), validB = valid_of(sub, grp);
    const size_t rowstride = (size_t)N * F::S;
    constexpr int RSRC3 = 0x00027000;
    const unsigned rs4 = 4u * (unsigned)rowstride;
    const unsigned ld4A = 4u * (unsigned)((size_t)n * F::S + min(sidx_of(grp, sub), F::S - 1));
    const unsigned ld4B = 4u * (unsigned)((size_t)n * F::S + min(sidx_of(sub, grp), F::S - 1));
    const int NT = (T + VIT_TILE - 1) / VIT_TILE;       // hand-overs: every wave passes exactly NT barriers
    const int Tm1 = max(T - 1, 0);
    // descriptor of the rows from t0 on; rows past the end re-read the last one (clamped, never branched)
    auto rows_from = [&](int t0) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(scores + (size_t)min(t0, Tm1) * rowstride), 0,
                                                 0x7fffffff, RSRC3);
    };
    float *const ring_lane = ring + lane;
    f2 *const ring2_lane = reinterpret_cast<f2 *>(ring) + lane;
    (void)ring_lane;
    (void)ring2_lane;
    float m = VIT_NEG_INF;

    if (wave == 0) {
        // ---------------------------------------------------------------- the recursion, nothing else
        float f = (sub < NB) ? 0.f : ((sub < F::NS) ? NEG_LARGE : VIT_NEG_INF);        // decode.py:93-95 (step 0 is even: from = sub)
        float sc[VIT_GROUP];
        {
            const __amdgpu_buffer_rsrc_t rs = rows_from(0);
#pragma unroll
            for (int k = 0; k < VIT_GROUP; ++k)
                sc[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (k & 1) ? ld4B : ld4A, rs4 * (unsigned)min(k, Tm1), 0));
        }
        float snext = validA ? sc[0] : VIT_NEG_INF;
        constexpr bool FILL = VIT_TRACE_SPLIT > 1;
        const unsigned long long maskA = __ballot(validA), maskB = __ballot(validB);
        float snext2 = VIT_NEG_INF;             // (FILL) the masked score of the step after the next one
        (void)maskA;
        (void)maskB;
        auto group = [&](int t0, auto full) {
            constexpr bool FULL = decltype(full)::value;
            const __amdgpu_buffer_rsrc_t rnext = rows_from(t0 + VIT_GROUP);
            const int lastn = max(Tm1 - min(t0 + VIT_GROUP, Tm1), 0);
#pragma unroll
            for (int k = 0; k < VIT_GROUP; ++k) {
                if (FULL || t0 + k < T) {                           // (wave-uniform)
                    const bool odd = (k & 1) != 0;                  // (compile time: unrolled, t0 is even)
                    const float cand = f + snext;
                    // (FILL, even step: the masks of the next two steps' scores ride in its wait slots -- sc[k + 1] and sc[k + 2] were
                    // requested a group ago; at the group's last pair sc[0] is the row this group's step 0 has just re-requested)
                    float sn1 = 0.f, sn2 = 0.f;
                    if (FILL && !odd) m = vit_grp_max_fill(cand, sc[(k + 1) % VIT_GROUP], maskB, sc[(k + 2) % VIT_GROUP], maskA, sn1, sn2);
                    else m = odd ? vit_cross_max(cand) : vit_grp_max(cand);
                    f = m;                                          // the next step's lane holds what it needs
                    // ---- off the chain: the vector for the trace waves, the row 64 steps ahead, the next mask
#if TK_VIT_RING2
                    // (the ring is exactly one group: slot = step inside the group; a step's maximum and candidate are ONE
                    // 8-byte store per lane -- one issue slot of the chain wave instead of two, one ds_read_b64 for the trace waves)
                    ring2_lane[k * WAVE] = f2{m, cand};
#else
                    ring_lane[(2 * k) * WAVE] = m;                  // (the ring is exactly one group: slot = step inside the group)
                    ring_lane[(2 * k + 1) * WAVE] = cand;
#endif
                    sc[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rnext, odd ? ld4B : ld4A, rs4 * (unsigned)min(k, lastn), 0));
                    if (FILL) {
                        if (!odd) {
                            snext = sn1;
                            snext2 = sn2;
                        } else {
                            snext = snext2;
                        }
                    } else {
                        snext = (odd ? validA : validB) ? sc[(k + 1) % VIT_GROUP] : VIT_NEG_INF;
                    }
                }
                if (k % VIT_TILE == VIT_TILE - 1 && (FULL || t0 + k - (VIT_TILE - 1) < T)) vit_handover();
            }
        };
        const int tfull = T / VIT_GROUP * VIT_GROUP;
        for (int t0 = 0; t0 < tfull; t0 += VIT_GROUP) group(t0, std::true_type{});
        if (tfull < T) group(tfull, std::false_type{});
    } else {
        // ---------------------------------------------------------------- traceback bytes (+ full outputs) of the even / odd steps
        const int par = (wave - 1) & 1;                         // this wave's steps: t = par (mod 2) ...
        const int part = VIT_TRACE_SPLIT == 1 ? 0 : (wave - 1) >> 1;    // ... of this part of every tile
        const bool odd = par != 0;
        const int st_to = odd ? sub : grp;                      // the state a lane's candidate belongs to
        const unsigned lane_tb = (unsigned)((size_t)n * VIT_GRP + st_to);
        const unsigned lane_o4 = 4u * (unsigned)((size_t)n * F::NS + min(st_to, F::NS - 1));
        const size_t pstride = (size_t)npad * VIT_GRP;           // traceback bytes [t][npad][8]
        const size_t ostride = (size_t)N * F::NS;
        constexpr bool ALL_GROUPS_LIVE = F::NS == VIT_GRP;
        constexpr int TH = VIT_TILE / 2 / VIT_TRACE_SPLIT;      // this wave's steps per tile
        static_assert(VIT_TILE % (2 * VIT_TRACE_SPLIT) == 0, "a tile's steps of one parity are shared out evenly");
        if (FULLOUT && par == 0 && sub == 0 && grp < F::NS) fwd_out[(size_t)n * F::NS + grp] = (grp < NB) ? 0.f : NEG_LARGE;
        auto group = [&](int t0, auto full) {
            constexpr bool FULL = decltype(full)::value;
            const __amdgpu_buffer_rsrc_t rtb = __builtin_amdgcn_make_buffer_rsrc(packed + (size_t)t0 * pstride, 0, 0x7fffffff, RSRC3);
            const __amdgpu_buffer_rsrc_t rfo = __builtin_amdgcn_make_buffer_rsrc(
                FULLOUT ? fwd_out + (size_t)(t0 + 1) * ostride : nullptr, 0, 0x7fffffff, RSRC3);
            int64_t *tout = FULLOUT ? tb_out + (size_t)t0 * ostride + (size_t)n * F::NS + min(st_to, F::NS - 1) : nullptr;
#pragma unroll
            for (int tile = 0; tile < VIT_GROUP / VIT_TILE; ++tile) {
                if (FULL || t0 + tile * VIT_TILE < T) {             // (wave-uniform: the chain wave passes this barrier too)
                    vit_handover();                                 // the tile's vectors are in the ring
#ifdef TK_VIT_NOTRACE
                    continue;                                       // lab: the chain wave alone
#endif
                    const int slot0 = tile * VIT_TILE;              // (the ring is exactly one group)
                    float mm[TH], cd[TH];
#pragma unroll
                    for (int q = 0; q < TH; ++q) {
                        const int k = 2 * (part * TH + q) + par;    // step inside the tile
#if defined(TK_VIT_TRACE_NOLDS)
                        mm[q] = (float)(slot0 + k + lane);          // lab (timing only, wrong bytes): no ring reads
                        cd[q] = (float)(slot0 + k + (lane & 56));
#elif TK_VIT_RING2
                        const f2 mc = ring2_lane[(slot0 + k) * WAVE];
                        mm[q] = mc.x;
                        cd[q] = mc.y;
#else
                        mm[q] = ring_lane[(2 * (slot0 + k)) * WAVE];
                        cd[q] = ring_lane[(2 * (slot0 + k) + 1) * WAVE];
#endif
                    }
#pragma unroll
                    for (int q = 0; q < TH; ++q) {
                        const int kk = tile * VIT_TILE + 2 * (part * TH + q) + par;   // step inside the group
                        if (FULL || t0 + kk < T) {
                            const float cand = cd[q];
                            // "first index wins": the lowest set bit among the ballot bits (candidate == maximum)
                            // of the state's eight candidates is the traceback byte
                            const unsigned long long eq = __ballot(cand == mm[q]);
                            unsigned arg;
                            if (!odd) {
                                const unsigned bits = (unsigned)(eq >> (8 * grp));
                                arg = (unsigned)__builtin_ctz((bits & 0xffu) | 0x100u) & 7u;
                            } else {
                                const unsigned long long bits = (eq >> sub) & 0x0101010101010101ull;
                                arg = ((unsigned)__builtin_ctzll(bits | (1ull << 63)) >> 3) & 7u;
                            }
#ifdef TK_VIT_TRACE_NOSTORE
                            asm volatile("" :: "v"(arg));           // lab (timing only): the bytes are formed and dropped
                            continue;
#endif
                            if (ALL_GROUPS_LIVE || st_to < F::NS) {
                                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)arg, rtb, lane_tb, (unsigned)(kk * pstride), 0);
                                if (FULLOUT) {
                                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mm[q]), rfo, lane_o4, 4u * (unsigned)(kk * ostride), 0);
                                    tout[(size_t)kk * ostride] = (int64_t)arg;
                                }
                            }
                        }
                    }
                }
            }
        };
        const int tfull = T / VIT_GROUP * VIT_GROUP;
        for (int t0 = 0; t0 < tfull; t0 += VIT_GROUP) group(t0, std::true_type{});
        if (tfull < T) group(tfull, std::false_type{});
    }
    (void)NT;
    // the trace waves' bytes must have landed before the walk
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef TK_VIT_NOPATH
    return;                                     // lab: forward pass alone
#endif
#ifdef TK_VIT_PATH1
    if (wave != 0) return;                      // lab: round 5's one-wave walk
    viterbi_path_pass<NB>(m, T, N, n, lane, path_out, packed, npad, /*alternating=*/true);
#else
    // (the bytes were written by other waves of this workgroup and have landed: vmcnt(0) before the barrier above; no line of
    // them was read before, so no stale copy can answer.  The ring is free by now.)
    __shared__ unsigned stsh[WAVE + 2];
    viterbi_path_pass3<NB>(m, T, N, n, lane, wave, vit_waves(VIT_TRACE_SPLIT), path_out, packed, npad, reinterpret_cast<unsigned long long *>(ring), stsh);
#endif
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
    // Three waves per read while the reads do not fill the chip's wave slots anyway (a read is bound by its serial
    // chain: T 4000 / N 256 342 -> 217 us path only, 404 -> 224 with the full outputs; N 1024 529 -> 345); at 2048
    // reads the two forms measure the same (743 / 733 us) and beyond the launch is bound by throughput, where
    // three waves per read cost more slots than the chain saves (tools/vitbench.py, profiles/r5_vitbench.txt)
    const char *v1 = TK_LAB_ENV("TK_VIT_V1");                   // lab: 1 = the one-wave kernel of rounds 1-4, 0 = three waves, for A/B
    const bool three = v1 ? v1[0] != '1' : N <= 1536;
    if (three) {
        // trace waves per step parity: 2 (five waves per read) while a read has a CU nearly to itself -- T 4000 / N 256 193 -> 180 us
        // path only, 233 -> 184 with the full outputs, N 512 230 -> 215; beyond ~2 reads per CU the extra waves share SIMDs with chain
        // waves (N 1024: 326 -> 427 us) and the three-wave form stays (profiles/r6_viterbi_fill_split_ab.txt).  TK_VIT_SPLIT = 1 | 2 | 4 (lab)
        int split = N <= 640 ? 2 : 1;
        if (const char *e = TK_LAB_ENV("TK_VIT_SPLIT")) split = atoi(e);
        auto go = [&](auto sp) {
            constexpr int SP = decltype(sp)::value;
            if (fwd != nullptr && tb != nullptr)
                hipLaunchKernelGGL((viterbi3_kernel<NB, true, SP>), dim3((unsigned)N), dim3(vit_waves(SP) * WAVE), 0, stream, scores,
                                   (int)T, (int)N, fwd, tb, path, static_cast<unsigned char *>(workspace), npad);
            else
                hipLaunchKernelGGL((viterbi3_kernel<NB, false, SP>), dim3((unsigned)N), dim3(vit_waves(SP) * WAVE), 0, stream, scores,
                                   (int)T, (int)N, fwd, tb, path, static_cast<unsigned char *>(workspace), npad);
        };
        if (split == 4) go(std::integral_constant<int, 4>{});
        else if (split == 2) go(std::integral_constant<int, 2>{});
        else go(std::integral_constant<int, 1>{});
        return hipGetLastError() == hipSuccess ? 0 : 4;
    }
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
