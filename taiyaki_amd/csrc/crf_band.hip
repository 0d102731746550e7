// crf_band.hip -- kernel A, band mode: sequence-constrained flip-flop CRF score + gradient
// (plain and cat-mod) as a BANDED, SKEWED sweep in the LINEAR domain for gfx950.
//
// Replaces taiyaki/ctc/c_crf_flipflop.c:43-516 and c_cat_mod_flipflop.c:37-582 (forward,
// backward, posterior scatter).  tests/helpers/crf_linear_model.py states the arithmetic in numpy
// (tests/helpers/crf_skew_model.py the schedule) and is checked against the oracle on the CPU.
//
//   * The L lattice positions of a read are cut into chunks of PW = 64 R cells, one wavefront
//     each.  Time is cut into blocks of BK = 8 steps.  Chunk w runs time block j in phase j + w
//     (forward sweep) resp. (NB-1-j) + (W-1-w) (backward sweep): the boundary cell a chunk needs
//     from its neighbour for every step of a block was written to a two-slot LDS ring one phase
//     earlier, so a workgroup executes ONE s_barrier per BK time steps.
//   * A cell is m * 2^f: float mantissa, integer FRAME per cell (int32 in registers, a 16-bit offset from
//     a per-chunk base in the checkpoint columns), fixed for the steps of a block.
//     The score row is exponentiated once per row and wave (one VGPR, lane = transition id); a
//     step is two ds_bpermute gathers and  m' = m es + m_up (em 2^(f_up - f))  -- an fma and a
//     DPP-fed v_fmac, no exp / log on the serial chain (round 2: max + log2(1 + 2^-|d|), ~15
//     instructions and two quarter-rate transcendentals per cell-step).
//   * Frames are set at block start as the K-Lipschitz envelope of the cells' own exponents
//     along the flow:  f[p] = max(exponent(cell p), f[upstream] - KLIP)  (a decayed prefix maximum:
//     six DPP steps per wave, the neighbouring chunk's edge frame comes in through the ring).
//     The lattice has CLIFFS near its diagonal front (cells a few positions apart differ by
//     2^60 .. 2^1000: few forced paths against combinatorially many free ones), so a cell's own
//     exponent says nothing about what flows into it during the block; with Lipschitz frames a
//     cell receives at most 2^KLIP of its frame unit per step and the growth inside a block is
//     bounded by the weights alone:  (1 + 2^KLIP)^8 2^(8 * 7.2) < 2^127 for |sharp * score| <= 5
//     (the network's 5 tanh).  A cell far below its upstream neighbours keeps a small mantissa and
//     is flushed to zero beyond 2^-126 of its frame: it is about to be overwritten by their inflow.
//   * Only the BAND is computed: cell (t, p) lies on a complete path iff p <= t and
//     L-1-p <= T-t, so chunk w is live for T-L+PW+2 of the T steps.
//   * The gradient pass does not read stored lattices (round 2: 10.8x / 43x the algorithmic
//     bytes): the sweeps leave ONE checkpoint column (m, f) per block plus the 8 boundary cells
//     per (chunk, block), and crf_band_posterior_kernel recomputes a block's columns itself
//     (7 cheap steps each way).  The posteriors of row t are products of the forward step's own
//     two terms with the same lane's backward cell,  (F_t[p] es) B_{t+1}[p]  (stay) and
//     (F_t[p-1] em) B_{t+1}[p]  (move INTO p), scaled by 2^(fF + fB - floor(log2 Z)).  They are
//     written to wave-private LDS in position order and read back in an order SORTED by transition
//     id (the permutation is computed once per read by "rank" workgroups that ride along in the
//     sweep launch), so the per-id sums of the reference's scatter-add (c_crf_flipflop.c:403-412)
//     are differences of ONE prefix scan held in registers: no atomics, no cross-wave reduction,
//     fixed summation order -> bitwise reproducible.
//   * Round 3 also: the frames' own-cells half runs ahead of the barrier (band_frames_own / _finish); cat-mod with
//     per-COLUMN factors exponentiates a row once per wave like the plain CRF (CW); the gradient pass runs two
//     waves per workgroup and interleaves its prefix scans.
//   * Round 4: the block length is a template parameter and the step weights carry a bias (see BK_MAX); ONE
//     row-maker wave per sweep workgroup exponentiates each score row once and leaves it in an LDS ring from which
//     every chunk wave gathers its weights (band_rowmaker: -12 % on the op against round 3's helper waves, which
//     shipped gathered weights per chunk pair and are gone).  The gradient pass decides which of a block's chunks
//     carry posterior mass in ONE batched pass before it computes any (frame loads of 8 / 16 chunks in flight together,
//     a ballot per chunk, the COLUMN test on the cell posteriors at the block's first column: see there) and issues
//     what it needs from memory before it can decide anything at its top -- it was bound by memory round trips in
//     front of its VALU work, not by the work (LABNOTES R4.11).
//   * The linear path is exact or says so: a read whose sweeps end non-finite (overflow: scores
//     beyond the bound above, e.g. sharpening factors > 1; underflow of everything: no complete
//     path, log-probabilities far below zero), whose two sweeps disagree, or ANY of whose rows'
//     posterior totals differs from the score (mass lost to a flush that mattered: bands a few
//     cells wide) sets gate[n] and is redone by the log-domain checkpoint kernel of
//     crf_kernels.hip, which takes any input the reference takes.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "crf_log.h"

namespace tk {

// Time steps per block (= per workgroup barrier, = per set of frames): a TEMPLATE parameter `BK` of everything
// below.  What a block costs besides its steps -- frames, checkpoint column, ring, barrier: ~75 of a
// phase's ~135 VALU instructions at BK = 8 -- is amortised over its steps (measured: 12-step blocks take
// the train step's sweep from 80 to 77 us and cost the gradient pass 3.5; -6 % at T = 1600: DESIGN.md
// section 4, LABNOTES.md round 4); what limits the length is the
// mantissas' growth between two frame updates, (1 + 2^KLIP) x the largest step weight per step inside
// fp32's exponent range.  band_pick_block() chooses:
//    BK = 12, weights biased by 2^-3   plain CRF, |sharp x score| <= 5.18 (the network's 5 tanh, unsharpened); cat-mod with
//                                      per-column factors from 705 bases on (round 5)
//    BK = 8,  no bias                  round 3's arithmetic: cat-mod on shorter reads; plain CRF sharpened up to 1.36
//    BK = 8,  weights biased by 2^-3   sharpening factors up to 1.76
//    BK = 4,  no bias                  sharpening factors up to 3.5
// The BIAS (BandArgs::wbias): every step weight carries a factor 2^-wbias, folded into the argument of its
// exponential (an fma instead of a multiply).  It centres the weights' range 2^(+-7.2 sharp) in fp32's
// exponent range, so that growth AND decay over a block both fit; every lattice value at time t is scaled by
// 2^(-wbias t), which posteriors (ratios to Z) do not see and the two scores get back as wbias x T.
constexpr int BK_MAX = 12;          // (the longest; sizes nothing -- every array is sized by the template parameter)
static_assert(BK_MAX % 4 == 0, "blocks move through the ring as float4");
constexpr int KLIP = 6;             // frame slope along the flow (bits per cell)
constexpr int ROW_PITCH = 48;       // shared-rows feed (band_rowmaker): floats per row image in LDS (S <= 46)
// layout of a block's exponentiated rows in the shared-rows image (lab builds: 0 = [row][id], round 4; 1 = [row pair][id][2];
// 2 = [id][row], the default: four rows at one id are one ds_read_b128 -- profiles/r5_rows_pair_ab.txt, r5_rows_quad_ab.txt)
#ifndef TK_ROWS_PAIR
#define TK_ROWS_PAIR 2
#endif
#ifndef TK_ROWS_VOL
#define TK_ROWS_VOL 0
#endif
constexpr int BAND_MAXW = 16;       // waves per workgroup
constexpr int POST_WAVES = 2;       // waves (= time blocks) per gradient-pass workgroup (8: +1.5 % in the step, +4 % at row K: coarser tail)
constexpr int KEY_DEAD = 63;        // sort key of padding instances
constexpr int NOFRAME = -(1 << 28); // "no live cell upstream"
constexpr float ROWZ_TOL = 1e-3f;   // bits: posterior row total vs score; sweep vs sweep
#ifndef TK_POST_SKIP_BELOW
#define TK_POST_SKIP_BELOW -160
#endif
// gradient pass: chunks whose cells all have log2 posterior bounds below this are skipped (see there); the
// longer block's mantissas may grow 2^14 further than the 8-step block's
template <int BK>
constexpr int POST_SKIP_BELOW = (BK > 8) ? TK_POST_SKIP_BELOW - 16 : TK_POST_SKIP_BELOW;

struct Win {
    int j0, j1;                     // first / last live time block (j0 > j1: never live)
};

// Live time blocks of chunk w (tests/helpers/crf_skew_model.py: windows()).
template <int BK>
__device__ __forceinline__ Win band_window(int w, int PW, int L, int T) {
    const int a = w * PW;
    if (w < 0 || a >= L) return {1, 0};
    const int NB = (T + BK - 1) / BK;
    if (L > T + 1) return {0, NB - 1};          // no complete path: nothing to trim by
    const int b = min(a + PW - 1, L - 1);
    const int tlo = max(0, a - 1), thi = min(T - 1, b + T - L + 1);
    return {tlo / BK, thi / BK};
}

#ifndef TK_BARRIER_BUILTIN
#define TK_BARRIER_BUILTIN 1
#endif
__device__ __forceinline__ void band_barrier() {
    // LDS traffic only: the checkpoint stores stay in flight across the barrier
#if TK_BARRIER_BUILTIN
    // (as builtins between two compiler fences, not as one asm string: the compiler's wait-count bookkeeping then KNOWS
    // that no LDS operation is pending behind the barrier -- with the asm it assumed the ring writes still were, and
    // the first phase of every loop trip waited for the ring reads with lgkmcnt(0) BEFORE it issued its gathers: two
    // LDS round trips one after the other at the head of the phase)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0), vmcnt / expcnt untouched
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// ... and the same with a register pinned in front of it: what computes `v` is issued BEFORE the barrier
// (hipcc is free to sink register-only work past an asm with a memory clobber, and did)
__device__ __forceinline__ void band_barrier_after(int &v) {
#if TK_BARRIER_BUILTIN
    asm volatile("" : "+v"(v) : : "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(v) : : "memory");
#endif
}

__device__ __forceinline__ float bperm(int byteaddr, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byteaddr, __float_as_int(v)));
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_iadd_masked(int x) {
    return x + __builtin_amdgcn_update_dpp(0, x, CTRL, ROWMASK, 0xF, false);
}
__device__ __forceinline__ int wave_inclusive_scan_int(int x) {
    x = dpp_iadd_masked<0x111, 0xF>(x);     // row_shr:1
    x = dpp_iadd_masked<0x112, 0xF>(x);     // row_shr:2
    x = dpp_iadd_masked<0x114, 0xF>(x);     // row_shr:4
    x = dpp_iadd_masked<0x118, 0xF>(x);     // row_shr:8
    x = dpp_iadd_masked<0x142, 0xA>(x);     // row_bcast:15 -> rows 1, 3
    x = dpp_iadd_masked<0x143, 0xC>(x);     // row_bcast:31 -> rows 2, 3
    return x;
}

// Inclusive prefix maximum over the 64 lanes in six fused v_max_i32_dpp (lanes without a source,
// and the rows a row_bcast does not apply to, are not written).
__device__ __forceinline__ int wave_prefix_max_fused(int x) {
    asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
    asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(x));
    asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(x));
    asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(x));
    asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(x));
    asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(x));
    return x;
}

constexpr int BUF_WORD3 = 0x00027000;       // raw buffer descriptor, 32-bit data (gfx9 family)

template <int R>
__device__ __forceinline__ void band_buffer_store(__amdgpu_buffer_rsrc_t rs, unsigned voff, const unsigned (&x)[R],
                                                  unsigned soff = 0) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    if constexpr (R == 4) {
        // two 8-byte stores, not one 16-byte store: a buffer store of more than 64 bits whose data
        // registers are overwritten by the next VALU instruction stores the NEW value of a dword on
        // gfx950 (observed in round 2; hipcc only guards the immediate-soffset form of that hazard)
        __builtin_amdgcn_raw_buffer_store_b64(u2{x[0], x[1]}, rs, voff, soff, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u2{x[2], x[3]}, rs, voff + 8u, soff, 0);
    } else if constexpr (R == 2) {
        __builtin_amdgcn_raw_buffer_store_b64(u2{x[0], x[1]}, rs, voff, soff, 0);
    } else {
        __builtin_amdgcn_raw_buffer_store_b32(x[0], rs, voff, soff, 0);
    }
}

constexpr int LEM_MIN = -100;       // cat-mod: the frame slope follows move weights down to 2^-100
// R 16-bit values per lane (the frames of a checkpoint column as offsets from the chunk's base)
template <int R>
__device__ __forceinline__ void band_buffer_store16(__amdgpu_buffer_rsrc_t rs, unsigned voff, const int (&x)[R],
                                                    unsigned soff = 0) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    // (v_perm_b32: the low halves of two registers in one instruction)
    if constexpr (R == 4) {
        __builtin_amdgcn_raw_buffer_store_b64(u2{__builtin_amdgcn_perm((unsigned)x[1], (unsigned)x[0], 0x05040100u),
                                                 __builtin_amdgcn_perm((unsigned)x[3], (unsigned)x[2], 0x05040100u)},
                                              rs, voff, soff, 0);
    } else if constexpr (R == 2) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm((unsigned)x[1], (unsigned)x[0], 0x05040100u), rs, voff, soff, 0);
    } else {
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)x[0], rs, voff, soff, 0);
    }
}

__device__ __forceinline__ int clamp_shift(int d, int klip) { return min(max(d, -160), klip - LEM_MIN + 1); }

// Frames of a block from the cells' own exponents: see the file header.  Works on the wave's cells
// in FLOW order (index q = lane R + j, upstream = q - 1); fb = the frame of the cell upstream of
// q = 0 (NOFRAME: none).  The decayed prefix maximum  f[q] = max_k (own[q - k] - k KLIP)  is a PLAIN
// prefix maximum of  z[q] = own[q] + KLIP q  (one ramp add, six fused DPP maxima, one ramp
// subtract).  Rescales m to the new frames; sc[j] = 2^(f[upstream] - f[j]) (0 where no move exists).
// SLOPE = false: the envelope falls by KLIP per cell.  SLOPE = true (cat-mod): by KLIP - lem[q] into
// cell q, lem[q] <= 0 the exponent of the LARGEST weight the move into q has in this block: a move
// that is weak in every row of the block (a modification the network does not believe in: modfact 8 x
// a log-probability of -10 is 2^-115) makes a true cliff in the lattice which the frames must follow
// -- the cells behind it hold ALL the mass of the paths that have passed it.  The inflow a cell can
// receive per step stays bounded by 2^KLIP of its frame unit (weight <= 2^lem, scale <= 2^(KLIP - lem)).
// The part of band_frames that needs only the wave's OWN cells (constant slope): their ramp-domain
// exponents and the exclusive prefix maximum over the lanes.  It runs at the END of a block, between the
// ring write and the barrier, so that its ~25 dependent instructions sit in the shadow of the LDS
// round trip and the barrier instead of behind them; band_frames_finish folds in the neighbour's edge
// frame (one maximum: the cell at q = -1 precedes every lane) when the ring has delivered it.
template <int R>
__device__ __forceinline__ void band_frames_own(const float (&m)[R], const int (&f)[R], int (&z)[R], int &run_excl, int lane, int klip) {
    const int q0 = lane * R;
    int zl = NOFRAME;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const bool live = m[j] > 0.f && m[j] < __builtin_huge_valf();
        z[j] = live ? f[j] + __builtin_amdgcn_frexp_expf(m[j]) + klip * (q0 + j + 1) : NOFRAME;
        zl = max(zl, z[j]);
    }
    const int zi = wave_prefix_max_fused(zl);
    run_excl = __builtin_amdgcn_update_dpp(NOFRAME, zi, 0x138, 0xF, 0xF, false);       // wave_shr:1; lane 0: nothing
}
template <int R>
__device__ __forceinline__ void band_frames_finish(float (&m)[R], int (&f)[R], float (&sc)[R], const bool (&has)[R],
                                                   const int (&z)[R], int run_excl, int fb, int lane, int klip) {
    const int q0 = lane * R;
    int run = max(run_excl, (fb > NOFRAME / 2) ? fb : NOFRAME);
    int fn[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        run = max(run, z[j]);
        fn[j] = (run > NOFRAME / 2) ? run - klip * (q0 + j + 1) : 0;
        m[j] = __builtin_amdgcn_ldexpf(m[j], max(f[j] - fn[j], -300));
        f[j] = fn[j];
    }
    int fup = __builtin_amdgcn_update_dpp(0, fn[R - 1], 0x138, 0xF, 0xF, false);       // wave_shr:1
    if (lane == 0) fup = (fb > NOFRAME / 2) ? fb : fn[0];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int d = clamp_shift((j == 0 ? fup : fn[j > 0 ? j - 1 : 0]) - fn[j], klip);
        sc[j] = has[j] ? __builtin_amdgcn_ldexpf(1.f, d) : 0.f;
    }
}

template <int R, bool SLOPE>
__device__ __forceinline__ void band_frames(float (&m)[R], int (&f)[R], float (&sc)[R], const bool (&has)[R],
                                            const int (&lem)[R], int fb, int lane, int klip) {
    const int q0 = lane * R;
    // ramp[j] = sum over the cells up to (q0 + j) of their slope allowance
    int ramp[R];
    if constexpr (SLOPE) {
        int own = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            own += klip - lem[j];
            ramp[j] = own;
        }
        const int before = wave_inclusive_scan_int(own) - own;
#pragma unroll
        for (int j = 0; j < R; ++j) ramp[j] += before;
    } else {
#pragma unroll
        for (int j = 0; j < R; ++j) ramp[j] = klip * (q0 + j + 1);
    }
    int z[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        // live = positive and finite (a NaN / inf cell keeps its frame and poisons the score)
        const bool live = m[j] > 0.f && m[j] < __builtin_huge_valf();
        z[j] = live ? f[j] + __builtin_amdgcn_frexp_expf(m[j]) + ramp[j] : NOFRAME;
    }
    const int zb = (lane == 0 && fb > NOFRAME / 2) ? fb : NOFRAME;     // the cell at q = -1 (ramp 0)
    int zl = zb;
#pragma unroll
    for (int j = 0; j < R; ++j) zl = max(zl, z[j]);
    const int zi = wave_prefix_max_fused(zl);
    int run = __builtin_amdgcn_update_dpp(zb, zi, 0x138, 0xF, 0xF, false);     // wave_shr:1; lane 0 keeps zb
    int fn[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        run = max(run, z[j]);
        fn[j] = (run > NOFRAME / 2) ? run - ramp[j] : 0;    // dead with nothing upstream: any frame does
        m[j] = __builtin_amdgcn_ldexpf(m[j], max(f[j] - fn[j], -300));
        f[j] = fn[j];
    }
    int fup = __builtin_amdgcn_update_dpp(0, fn[R - 1], 0x138, 0xF, 0xF, false);       // wave_shr:1
    if (lane == 0) fup = (fb > NOFRAME / 2) ? fb : fn[0];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int d = clamp_shift((j == 0 ? fup : fn[j > 0 ? j - 1 : 0]) - fn[j], klip);
        sc[j] = has[j] ? __builtin_amdgcn_ldexpf(1.f, d) : 0.f;
    }
}

// ---------------------------------------------------------------------------
// Ids from flip-flop codes (BandArgs::codes): ff_common.h lbl_*.  No launch of the linear path reads an index ARRAY
// when the call brings its labels: the sweeps, the rank workgroups and the gradient pass form ids from the codes
// (read-only inputs that stay in every XCD's L2 from call to call; the arrays a build kernel has just written are
// misses in seven of eight L2s).  A first version let the rank workgroups write the arrays for the gradient pass:
// that alone cost the sweep launch 7.6 us (profiles/r5_index_build_ab.txt).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int band_code(const BandArgs &a, int64_t i) { return lbl_code(a, i); }
__device__ __forceinline__ int band_stay_id(const BandArgs &a, int cp) { return lbl_stay(a, cp); }
__device__ __forceinline__ int band_move_id(const BandArgs &a, int cp, int cn) { return lbl_move(a, cp, cn); }
__device__ __forceinline__ int band_mod_seq(const BandArgs &a, int cn, int64_t i_next, bool *bad) {
    return lbl_mod_seq(a, cn, a.mod_cats[i_next], bad);
}

// This read's offset into the label arrays = the sum of the lengths before it, clamped to the label array; every
// workgroup sums for itself (a batch is a few hundred reads).  All threads of the workgroup take part; `sh`:
// 3 x BAND_MAXW shared 64-bit words.  One round trip (the lengths -- this read's own among them) and ONE barrier:
// per-wave partial sums, no atomics.  Returns (offset, total announced, this read's length).
__device__ __forceinline__ void band_offset_of(const BandArgs &a, int n, long long *sh, long long *off_out, long long *all_out,
                                               int *len_out) {
    long long mine = 0, all = 0, own = 0;
    for (int i = threadIdx.x; i < a.N; i += blockDim.x) {
        const long long v = a.seqlen[i];
        all += v;
        if (i < n) mine += v;
        if (i == n) own = v;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        mine += __shfl_xor(mine, m, WAVE);
        all += __shfl_xor(all, m, WAVE);
        own += __shfl_xor(own, m, WAVE);
    }
    const int w = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
    if ((threadIdx.x & (WAVE - 1)) == 0) {
        sh[3 * w] = mine;
        sh[3 * w + 1] = all;
        sh[3 * w + 2] = own;
    }
    __syncthreads();
    mine = all = own = 0;
    for (int k = 0; k < nw; ++k) {
        mine += sh[3 * k];
        all += sh[3 * k + 1];
        own += sh[3 * k + 2];
    }
    *off_out = mine;
    *all_out = all;
    *len_out = (int)own;
}

template <int R, bool MOD, bool FWD, bool GRAD, bool ROWS, bool CW, int BK, bool PRE4 = false>
__device__ __forceinline__ void band_sweep(const BandArgs &a, int n, int ws, int L, int64_t off, float *E, int *Ef, const float *Ezero,
                                           const f4 *Wt, const int wofs = 0) {
    // `n`: the read (scores, labels); `ws`: its slot in the workspace arrays -- n for the batch's launch, the
    // workgroup's own slot for the retry launch (crf_band_retry_kernel), whose arrays hold a few reads only
    constexpr int PW = R * WAVE;
    // the wave index is wave-uniform: keep everything derived from it in SGPRs
    // (`wofs`: the workgroup's first wave of THIS sweep -- 0 but in the retry launch, whose two sweeps share a workgroup)
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) - wofs, lane = threadIdx.x & (WAVE - 1);
    const int N = a.N, T = a.T, S = a.S, W = a.W;
    const int a0 = w * PW;
    const int NB = (T + BK - 1) / BK, NPH = NB + W - 1;
    const int src = FWD ? w - 1 : w + 1;                        // the chunk our boundary cell comes from
    const Win win = band_window<BK>(w, PW, L, T);
    const Win wsrc = (src >= 0 && src < W) ? band_window<BK>(src, PW, L, T) : Win{1, 0};
    if (win.j0 > win.j1) {
        // a chunk past the end of this read: keep the workgroup's barriers company
        for (int ph = 0; ph < NPH + (ROWS ? 1 : 0); ++ph) band_barrier();
        return;
    }
    const size_t rowstride = (size_t)N * S;
    const float *lpn = a.lp + (size_t)n * S;
    const unsigned col4 = 4u * (unsigned)min(lane, S - 1);
    const float c = a.c_can;
    // cat-mod with per-COLUMN factors (BandArgs::colw): the row is exponentiated once per wave with a
    // multiplier per lane = column (sharp log2 e on the canonical columns, factor x sharp_mod x log2 e on a
    // modification's), and a move's weight is the product of two gathers from it
    constexpr bool colw_mode = MOD && CW;       // (a template parameter: a per-step runtime branch cost both forms 10 %)
    const float cw_lane = colw_mode ? ((int)lane < a.ncan ? c : a.colw[min(max((int)lane - a.ncan, 0), S - a.ncan - 1)] * a.c_mod) : c;
    // the weights' bias (see BK_MAX): once per step weight -- on the canonical columns' exponentials, not on
    // a modification column's factor of a per-column product
    const float wbias = a.wbias;
    const float wb_lane = (colw_mode && (int)lane >= a.ncan) ? 0.f : wbias;
    const int klip = a.klip;            // the frames' slope along the flow, bits per cell (KLIP; more for narrow bands: crf_band_pick_block)

    // The wave's cells in FLOW order: index q = lane R + j, upstream = q - 1, i.e. position
    // a0 + q forward and a0 + PW - 1 - q backward (the backward sweep runs on mirrored lanes, so
    // both directions shift with wave_shr and scan with row_shr / row_bcast).  Per cell: the
    // bpermute byte addresses (lane = transition id) of its stay and of the move INTO it (forward)
    // / OUT of it (backward); cat-mod: that move's mod column and weight.
    int st4[R], mv4[R], md4[MOD ? R : 1];
    float fw[MOD ? R : 1];
    bool has[R];
    float m[R], sc[R];
    int f[R];
    bool bad_label = false;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int q = lane * R + j, p = FWD ? a0 + q : a0 + PW - 1 - q;
        const int ms = FWD ? p - 1 : p;                         // the move's source position
        has[j] = ms >= 0 && ms < L - 1;
        if (a.codes != nullptr) {
            // (the launch builds its indices itself: ids straight from the flip-flop codes.  The FORWARD sweep's
            // waves see every label of the read here -- position p and the move into it -- so they are also the
            // label check of tk_flipflop_build_indices_dev, TK_STATUS_BAD_LABEL, at no load of its own)
            const int craw = (p < L) ? a.codes[off + p] : 0;
            const int cp = min(max(craw, 0), 2 * a.nbase - 1);
            if (FWD) bad_label |= craw != cp;
            st4[j] = 4 * ((p < L) ? band_stay_id(a, cp) : 0);
            const int c0 = has[j] ? band_code(a, off + ms) : 0, c1 = has[j] ? band_code(a, off + ms + 1) : 0;
            mv4[j] = 4 * (has[j] ? band_move_id(a, c0, c1) : 0);
            if (MOD) {
                const int mseq = has[j] ? band_mod_seq(a, c1, off + ms + 1, FWD ? &bad_label : nullptr) : 0;
                md4[MOD ? j : 0] = 4 * (has[j] ? a.ncan + mseq : 0);
                fw[MOD ? j : 0] = has[j] ? a.mcw[mseq] * a.c_mod : 0.f;
            }
        } else {
            st4[j] = 4 * ((p < L) ? a.stay[off + p] : 0);       // (a padding cell is 0 and stays 0)
            mv4[j] = 4 * (has[j] ? a.move[off + ms] : 0);
            if (MOD) {
                md4[MOD ? j : 0] = 4 * (has[j] ? a.mod[off + ms] : 0);
                fw[MOD ? j : 0] = has[j] ? a.modfact[off + ms] * a.c_mod : 0.f;
            }
        }
        m[j] = (p == (FWD ? 0 : L - 1)) ? 1.f : 0.f;            // c_crf_flipflop.c:113-116, 216-220
        f[j] = 0;
        sc[j] = 0.f;
    }

    if (FWD && bad_label && a.status) atomicOr(a.status, 8u);
    const unsigned rs4 = 4u * (unsigned)rowstride;
    float *ckm = GRAD ? (FWD ? a.ckFm : a.ckBm) + (size_t)ws * NB * a.LP + a0 : nullptr;
    int16_t *ckf = GRAD ? (FWD ? a.ckFf : a.ckBf) + (size_t)ws * NB * a.LP + a0 : nullptr;
    int *ckb = GRAD ? (FWD ? a.ckFb : a.ckBb) + (size_t)ws * NB * W + w : nullptr;
    const __amdgpu_buffer_rsrc_t rm_all = __builtin_amdgcn_make_buffer_rsrc(ckm, 0, 0x7fffffff, BUF_WORD3);
    const __amdgpu_buffer_rsrc_t rf_all = __builtin_amdgcn_make_buffer_rsrc(ckf, 0, 0x7fffffff, BUF_WORD3);
    const __amdgpu_buffer_rsrc_t rb_all = __builtin_amdgcn_make_buffer_rsrc(ckb, 0, 0x7fffffff, BUF_WORD3);
    const unsigned lp4 = 4u * (unsigned)a.LP, w4 = 4u * (unsigned)W;
    int fbase_prev = 0;
    bool have_base = false;
    // the gradient pass works on 64-cell chunks whatever R is: the lanes that hold the last cell (in
    // flow order) of a 64-cell run hand their cell of every step over.  Forward that is the LAST
    // cell of sub-chunk w R + k - 1, backward the FIRST cell of sub-chunk w R + R - k  (k = 1 .. R).
    constexpr int SUBL = WAVE / R;                              // lanes per 64 cells
    const bool sub_lane = ((lane + 1) % SUBL) == 0;
    const int sub_k = (lane + 1) / SUBL;
    float *bnd = GRAD ? (FWD ? a.bndF : a.bndB) + ((size_t)ws * NB * a.Wp + w * R + (FWD ? sub_k - 1 : R - sub_k)) * BK
                      : nullptr;
    // byte offset of the lane's R cells inside a checkpoint row of the chunk
    const unsigned lane_cell4 = 4u * (unsigned)(FWD ? lane * R : PW - (lane + 1) * R);
    const bool edge_lane = lane == WAVE - 1;                    // holds the most downstream cell (j = R-1)

    // Score rows: three register sets rotate through "current block", "next" and "the one after"
    // (in sweep direction).  A set is loaded two phases before it is consumed and nothing reads
    // it in between, so the load latency never sits in a phase; the live loop is unrolled three
    // times to keep the rotation in register NAMES (a copy would have to wait for the load).
    float row0[BK], row1[BK], row2[BK];
    // (row offsets inside a block are loop-invariant scalars; the descriptor covers exactly the rows
    // of the block that exist, so a row past the end of the tensor is out of range and reads 0 --
    // its step is never taken -- without a clamp per row.  The SCALAR offset takes part in the range
    // check on gfx950: tools/bufrange_probe.hip, profiles/r3_bufrange_probe.txt.  Clamping instead
    // costs 3 % as selects and 6 % at T = 4000 as a branch around the loads.)
    unsigned rowoff[BK];
#pragma unroll
    for (int i = 0; i < BK; ++i) rowoff[i] = rs4 * (unsigned)i;
    auto load_block = [&](int j, float (&dst)[BK]) {
        j = min(max(j, 0), NB - 1);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(lpn + (size_t)(j * BK) * rowstride), 0, (int)(rs4 * (unsigned)min(BK, T - j * BK)), BUF_WORD3);
#pragma unroll
        for (int i = 0; i < BK; ++i) {
            dst[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, col4, rowoff[i], 0));
        }
    };

    // the frames' own-cells half ahead of the barrier: constant slope only (cat-mod's slope needs the NEXT block's
    // move weights), and not at four cells per lane, whose sweep is bound by registers and VALU throughput, not by
    // this chain (measured: -2 % at the train step's shape, +3 % at T = 4000 / N = 256 with it)
    constexpr bool SPLIT_FRAMES = !MOD && R <= 2;
    int zown[R], zrun_excl = NOFRAME;           // band_frames_own's results, carried from block to block
#pragma unroll
    for (int j = 0; j < R; ++j) zown[j] = NOFRAME;
    // ROWS: the LDS slot of the block this chunk runs next (block j lives in slot j mod (W + 1))
    int rslot = 0;
    int stamp_k = 0;
#ifdef TK_LAB_STAMPS
#define STAMP(q)                                                                            \
    if (a.dbg && blockIdx.x == (GRAD ? a.N : 0) && w == 0 && lane == 0 && stamp_k < 64)        \
        a.dbg[stamp_k * 8 + (q)] = __builtin_amdgcn_s_memtime();
#else
#define STAMP(q)
#endif
    (void)stamp_k;

    // One live phase = one time block of this chunk:
    //   1. the weights of all BK steps, gathered by transition id from the exponentiated rows (2 R
    //      ds_bpermute per step; they depend on neither the cells nor the frames, so their LDS
    //      round trips overlap the frame computation);
    //   2. the frames of the block (band_frames) and the checkpoint column;
    //   3. BK steps of pure VALU: per cell an fma, for the lane's first cell a DPP-fed v_fmac that
    //      takes the upstream lane's last cell (lane 0's source is out of range and reads 0 -- its
    //      upstream cell is the ring's, folded in by the fma);
    //   4. the edge lane hands its BK boundary cells to the ring (and to HBM for the gradient pass).
    auto body = [&](int j, const float (&cur)[BK], float (&fill)[BK]) {
        STAMP(0);
        if constexpr (!ROWS) load_block(FWD ? j + 2 : j - 2, fill);
        const bool pl = j >= wsrc.j0 && j <= wsrc.j1;           // the neighbour ran this block one phase ago
        const int slot = j & 1;
        const int srcc = min(max(src, 0), W - 1);
        // the boundary cells of the whole block: lane 0 reads the ring, the others a row of zeros
        const f4 *Ein = reinterpret_cast<const f4 *>((pl && lane == 0) ? E + (srcc * 2 + slot) * BK : Ezero);
        float ein[BK];
#pragma unroll
        for (int q4 = 0; q4 < BK / 4; ++q4) {
            const f4 e = Ein[q4];
#pragma unroll
            for (int q = 0; q < 4; ++q) ein[q4 * 4 + q] = e[q];
        }
        const int fb = pl ? Ef[srcc * 2 + slot] : NOFRAME;
        const int nvalid = min(BK, T - j * BK);                 // rows of this block that exist
        // (R = 4: four steps' worth of weights at a time -- 16 waves leave 128 registers per lane)
        constexpr int GH = (R == 4) ? 4 : BK;
        float es[GH][R], em[GH][R];
        auto gather_group = [&](int ii0) {
            static_assert(!ROWS || !MOD || CW, "shared rows hold exponentials: cat-mod needs per-column factors");
            if constexpr (ROWS) {
                // SHARED ROWS (band_rowmaker): the block's exponentiated rows are in LDS, one image per
                // workgroup; a weight is one LDS read at (row, transition id) -- the value ds_bpermute would
                // have fetched from this wave's own copy of the row, bit for bit
                const float *rp = reinterpret_cast<const float *>(Wt) + (size_t)rslot * (BK * ROW_PITCH);
#if TK_ROWS_PAIR == 2
                // (the image holds a block's rows PER ID, [id][row]: four rows at one id are one ds_read_b128 -- 4 LDS cycles
                // per wave-instruction for 16 bytes per lane, the rate of the MI355X guide's table; the compiler had merged
                // the pair form's two ds_read_b64 into ds_read2_b64 at 8 cycles)
#pragma unroll
                for (int g = 0; g < GH; g += 4) {
                    const int ib = FWD ? ii0 + g : BK - 4 - (ii0 + g);      // the four rows of steps g .. g + 3 (backward: descending)
#pragma unroll
                    for (int jj = 0; jj < R; ++jj) {
                        const f4 s4 = *reinterpret_cast<const f4 *>(rp + (st4[jj] >> 2) * BK + ib);
                        f4 m4 = *reinterpret_cast<const f4 *>(rp + (mv4[jj] >> 2) * BK + ib);
                        if constexpr (MOD) m4 = m4 * *reinterpret_cast<const f4 *>(rp + (md4[MOD ? jj : 0] >> 2) * BK + ib);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            es[g + k][jj] = s4[FWD ? k : 3 - k];
                            em[g + k][jj] = m4[FWD ? k : 3 - k];
                        }
                    }
                }
#elif TK_ROWS_PAIR
                // (the image holds the rows in PAIRS, [pair][id][2]: both rows of a pair at one id are one ds_read_b64 --
                // 2 LDS cycles per wave-instruction where ds_read2_b32 takes 4; same banks per lane group, same values)
                typedef float f2a __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int g = 0; g < GH; g += 2) {
                    const int i0 = FWD ? ii0 + g : BK - 1 - (ii0 + g);     // the row of step g; the pair's other row is step g + 1's
                    const int pr = i0 >> 1;
                    const int ga = FWD ? g : g + 1, gb = FWD ? g + 1 : g;   // .x = the even row, .y = the odd row
#if TK_ROWS_VOL
                    const volatile f2a *pp = reinterpret_cast<const volatile f2a *>(rp + pr * (2 * ROW_PITCH));
#else
                    const f2a *pp = reinterpret_cast<const f2a *>(rp + pr * (2 * ROW_PITCH));
#endif
#pragma unroll
                    for (int jj = 0; jj < R; ++jj) {
                        const f2a s2 = pp[st4[jj] >> 2];
                        es[ga][jj] = s2.x;
                        es[gb][jj] = s2.y;
                        f2a m2 = pp[mv4[jj] >> 2];
                        if constexpr (MOD) m2 = m2 * pp[md4[MOD ? jj : 0] >> 2];
                        em[ga][jj] = m2.x;
                        em[gb][jj] = m2.y;
                    }
                }
#else
#pragma unroll
                for (int g = 0; g < GH; ++g) {
                    const int i = FWD ? ii0 + g : BK - 1 - (ii0 + g);
#pragma unroll
                    for (int jj = 0; jj < R; ++jj) {
                        es[g][jj] = rp[i * ROW_PITCH + (st4[jj] >> 2)];
                        if constexpr (MOD)
                            em[g][jj] = rp[i * ROW_PITCH + (mv4[jj] >> 2)] * rp[i * ROW_PITCH + (md4[MOD ? jj : 0] >> 2)];
                        else
                            em[g][jj] = rp[i * ROW_PITCH + (mv4[jj] >> 2)];
                    }
                }
#endif
                return;
            }
#pragma unroll
            for (int g = 0; g < GH; ++g) {
                const int i = FWD ? ii0 + g : BK - 1 - (ii0 + g);
                const float er = fast_exp2(fmaf(cur[i], cw_lane, -wb_lane));
#pragma unroll
                for (int jj = 0; jj < R; ++jj) {
                    es[g][jj] = bperm(st4[jj], er);
                    if constexpr (MOD) {
                        // c_cat_mod_flipflop.c:64-66: move score + modfact * mod score
                        if constexpr (colw_mode)
                            em[g][jj] = bperm(mv4[jj], er) * bperm(md4[MOD ? jj : 0], er);
                        else
                            em[g][jj] = fast_exp2(fmaf(bperm(md4[MOD ? jj : 0], cur[i]), fw[MOD ? jj : 0], fmaf(bperm(mv4[jj], cur[i]), c, -wbias)));
                    } else {
                        em[g][jj] = bperm(mv4[jj], er);
                    }
                }
            }
        };
        float edge[BK];
        auto step_group = [&](int ii0, auto full_tag) {
            constexpr bool FULLBLK = decltype(full_tag)::value;
            // everything that does not depend on the cells first: move weights in the cells' frames,
            // the boundary lane's inflow
            // (R = 4 has the registers for that in launches of up to 12 waves -- 170 per lane -- and not at 16 waves: PRE4)
            constexpr bool PRE = R < 4 || PRE4;
            float mt[PRE ? GH : 1][R], u[PRE ? GH : 1];
            if constexpr (PRE && (!MOD || CW)) {
                // (two steps per instruction: v_pk_mul_f32 -- -1 % for the plain CRF, -2 % for cat-mod with
                // per-column factors; cat-mod's general form, whose weights come out of a v_exp_f32 each,
                // measured +2 % with the pairing and keeps single multiplies)
                typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int g = 0; g < GH; g += 2) {
                    const int i0 = FWD ? ii0 + g : BK - 1 - (ii0 + g), i1 = FWD ? i0 + 1 : i0 - 1;
                    // (the pairs in the order the gathers and the ring leave them in registers -- ascending row: the
                    // backward sweep walks the rows downwards, and built the other way round every pair costs two
                    // v_mov to swap, 12 issue slots of a phase's ~150.  Bit-identical either way; measured per form on
                    // one box (profiles/r4_sweep_pair_order_ab.txt): cat-mod 133.6 -> 130.2 us, two cells per lane
                    // 182.9 -> 176.6 -- and the plain one-cell form 100.3 -> 102.3, so that one keeps its swaps)
                    constexpr bool ASC = FWD || MOD || R > 1;
                    constexpr int ga = ASC && !FWD ? 1 : 0, gb = 1 - ga;
#pragma unroll
                    for (int jj = 0; jj < R; ++jj) {
                        const f2 p = f2{em[g + ga][jj], em[g + gb][jj]} * f2{sc[jj], sc[jj]};
                        mt[g + ga][jj] = p.x;
                        mt[g + gb][jj] = p.y;
                    }
                    const f2 q = f2{ein[ga ? i1 : i0], ein[ga ? i0 : i1]} * f2{mt[g + ga][0], mt[g + gb][0]};
                    u[g + ga] = q.x;
                    u[g + gb] = q.y;
                }
            } else if constexpr (PRE) {
#pragma unroll
                for (int g = 0; g < GH; ++g) {
                    const int i = FWD ? ii0 + g : BK - 1 - (ii0 + g);
#pragma unroll
                    for (int jj = 0; jj < R; ++jj) mt[g][jj] = em[g][jj] * sc[jj];
                    u[g] = ein[i] * mt[g][0];
                }
            }
#pragma unroll
            for (int g = 0; g < GH; ++g) {
                const int i = FWD ? ii0 + g : BK - 1 - (ii0 + g);
                if (!FULLBLK) {
                    edge[i] = 0.f;
                    if (i >= nvalid) continue;                  // (wave-uniform; only in the last block)
                }
                edge[i] = m[R - 1];
                float nw[R], mtg[R];
#pragma unroll
                for (int jj = 0; jj < R; ++jj) mtg[jj] = PRE ? mt[PRE ? g : 0][jj] : em[g][jj] * sc[jj];
                nw[0] = fmaf(m[0], es[g][0], PRE ? u[PRE ? g : 0] : ein[i] * mtg[0]);
                asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                    : "+v"(nw[0]) : "v"(m[R - 1]), "v"(mtg[0]));
#pragma unroll
                for (int jj = 1; jj < R; ++jj) nw[jj] = fmaf(m[jj], es[g][jj], m[jj - 1] * mtg[jj]);
#pragma unroll
                for (int jj = 0; jj < R; ++jj) m[jj] = nw[jj];
            }
        };
        gather_group(0);
        STAMP(1);
        int lem[R];
#pragma unroll
        for (int jj = 0; jj < R; ++jj) {
            lem[jj] = 0;
            if constexpr (MOD) {
                float emx = em[0][jj];
#pragma unroll
                for (int g = 1; g < GH; ++g) emx = fmaxf(emx, em[g][jj]);
                // (R = 4 knows the first four steps' weights here: a later, larger one can only make
                // the block overflow, which the score check catches)
                lem[jj] = (has[jj] && emx > 0.f) ? min(max(__builtin_amdgcn_frexp_expf(emx), LEM_MIN), 0) : 0;
            }
        }
        if constexpr (!SPLIT_FRAMES) {
            band_frames<R, MOD>(m, f, sc, has, lem, fb, lane, klip);
        } else {
            // (the own-cells half ran at the end of the previous block, or before the first one)
            band_frames_finish<R>(m, f, sc, has, zown, zrun_excl, fb, lane, klip);
        }
        if (edge_lane) Ef[w * 2 + slot] = f[R - 1];
        if (GRAD) {
            // checkpoint column: forward column 8 j, backward column 8 j + nvalid (positions ascending)
            // (one descriptor per array for the whole read, built before the loop; the block is a scalar offset:
            // NB LP 4 bytes stay far below 2^31)
            const unsigned soff_m = (unsigned)j * lp4;
            // frames as 16-bit offsets from a per-(chunk, block) base (the envelope falls by KLIP per
            // cell and rises with the cells' own exponents: a few thousand across a chunk at most.  An
            // offset that does not fit -- scores far outside the network's range -- is NOT checked here:
            // a wrapped frame puts its cell 2^65536 off, and the gradient pass verifies every row's
            // total against the partition function anyway: such a read is disowned there)
            // (the base is the PREVIOUS block's first frame -- frames drift by a few bits per block -- so
            // that no cross-lane read sits between this block's frames and its stores; a chunk's first
            // block starts from the neighbour's edge frame)
            const int fbase = have_base ? fbase_prev : (pl ? __builtin_amdgcn_readfirstlane(fb) : 0);
            fbase_prev = __builtin_amdgcn_readfirstlane(f[0]);
            have_base = true;
            unsigned xm[R];
            int xf[R];
#pragma unroll
            for (int jj = 0; jj < R; ++jj) {
                xm[jj] = __float_as_uint(m[FWD ? jj : R - 1 - jj]);
                xf[jj] = f[FWD ? jj : R - 1 - jj] - fbase;      // (kept to 16 bits: see the base's comment)
            }
            band_buffer_store<R>(rm_all, lane_cell4, xm, soff_m);
            band_buffer_store16<R>(rf_all, lane_cell4 / 2, xf, soff_m / 2);
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)fbase, rb_all, 0, (unsigned)j * w4, 0);   // (every lane, the same word)
        }
        STAMP(2);
        if (nvalid == BK) {
            step_group(0, std::true_type{});
#pragma unroll
            for (int ii0 = GH; ii0 < BK; ii0 += GH) {
                gather_group(ii0);
                step_group(ii0, std::true_type{});
            }
        } else {
            step_group(0, std::false_type{});
#pragma unroll
            for (int ii0 = GH; ii0 < BK; ii0 += GH) {
                gather_group(ii0);
                step_group(ii0, std::false_type{});
            }
        }
        STAMP(3);
        if (edge_lane) {
            f4 *Eo = reinterpret_cast<f4 *>(E + (w * 2 + slot) * BK);
#pragma unroll
            for (int q4 = 0; q4 < BK / 4; ++q4) Eo[q4] = f4{edge[4 * q4], edge[4 * q4 + 1], edge[4 * q4 + 2], edge[4 * q4 + 3]};
        }
        if (GRAD && sub_lane) {
            // the boundary cells of this block, for the gradient pass
            f4 *Bo = reinterpret_cast<f4 *>(bnd + (size_t)j * a.Wp * BK);
#pragma unroll
            for (int q4 = 0; q4 < BK / 4; ++q4) Bo[q4] = f4{edge[4 * q4], edge[4 * q4 + 1], edge[4 * q4 + 2], edge[4 * q4 + 3]};
        }
        if constexpr (SPLIT_FRAMES) {
            band_frames_own<R>(m, f, zown, zrun_excl, lane, klip);    // (for the next block)
            band_barrier_after(zrun_excl);
        } else {
            band_barrier();
        }
        STAMP(4);
        ++stamp_k;
        if constexpr (ROWS) {
            rslot += FWD ? 1 : -1;
            rslot = (rslot > W) ? 0 : ((rslot < 0) ? W : rslot);
        }
    };

    // phases: chunk w runs block j in phase j + w (forward) / (NB-1-j) + (W-1-w) (backward);
    // before and after its live blocks it only takes part in the barriers
    const int jfirst = FWD ? win.j0 : win.j1, nlive = win.j1 - win.j0 + 1;
    const int ph0 = FWD ? win.j0 + w : (NB - 1 - win.j1) + (W - 1 - w);
    const int dj = FWD ? 1 : -1;
    if constexpr (SPLIT_FRAMES) band_frames_own<R>(m, f, zown, zrun_excl, lane, klip);
    if constexpr (!ROWS) {
        load_block(jfirst, row0);
        load_block(jfirst + dj, row1);
    } else {
        if constexpr (ROWS) rslot = jfirst % (W + 1);
        band_barrier();                                         // the row maker's lead phase
    }
    for (int ph = 0; ph < ph0; ++ph) band_barrier();
    for (int k = 0; k < nlive; k += 3) {
        body(jfirst + dj * k, row0, row2);
        if (k + 1 < nlive) body(jfirst + dj * (k + 1), row1, row0);
        if (k + 2 < nlive) body(jfirst + dj * (k + 2), row2, row1);
    }
    for (int ph = ph0 + nlive; ph < NPH; ++ph) band_barrier();

    // score = fwd[T][L-1] (c_crf_flipflop.c:131) / bwd[0][0] (:234), in bits
    const int pend = FWD ? L - 1 : 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int q = lane * R + j, p = FWD ? a0 + q : a0 + PW - 1 - q;
        if (p == pend) {
            // (a cost-only call, too: round 5 -- its forward sweep used to write the cost by itself whenever the score
            // was finite, and a read outside the linear path's range came back silently WRONG, e.g. cat-mod with five
            // modifications per base: costs off by 0.02 .. 0.19.  Now both sweeps run, and the launch behind them
            // (the tail launch's vote pass, crf_band_tail_kernel) writes the cost where they agree and retries / redoes the read where they do not.)
            const double sc2 = (double)f[j] + log2((double)m[j]);
            (FWD ? a.scoreF : a.scoreB)[ws] = sc2;
        }
    }
}

// ===========================================================================
// SHARED ROWS (round 4).  Every chunk of a read gathers its step weights from the SAME exponentiated score
// rows -- chunk w needs block j's rows in phase j + w.  Without this (ROWS = false: launches whose workgroup
// has no room for a 17th wave) each chunk wave loads and exponentiates the rows itself and gathers with
// ds_bpermute; round 3's helper waves did it per chunk PAIR and shipped 2 R BK weights per chunk through LDS
// (ds_write_b128 at 13 LDS cycles a piece).  Here ONE extra wave per sweep workgroup, the row maker, loads
// and exponentiates block ph + 1's rows during phase ph and leaves them in an LDS ring of W + 1 blocks (a
// block is last read W - 1 phases after its first use); the chunk waves read a weight with one LDS read at
// (row, transition id) -- the compiler pairs them into ds_read2_b32.  The values are the ones ds_bpermute
// would have fetched: results are bit for bit those of the other feeds (tools/crf_bitcmp.py,
// test_crf_weight_feeds_change_no_bit).  Measured (profiles/r4_feed_modes.txt): the op at the train step's
// shape 110 us against 125 with helper waves and 134 with neither; -15 .. -22 % at every other shape tried.
// ===========================================================================
template <bool MOD, bool FWD, bool CW, int BK>
__device__ __forceinline__ void band_rowmaker(const BandArgs &a, int n, float *Er) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int N = a.N, T = a.T, S = a.S, W = a.W;
    const int NB = (T + BK - 1) / BK, NPH = NB + W - 1;
    const size_t rowstride = (size_t)N * S;
    const float *lpn = a.lp + (size_t)n * S;
    const unsigned col4 = 4u * (unsigned)min(lane, S - 1);
    const unsigned rs4 = 4u * (unsigned)rowstride;
    const float c = a.c_can;
    const float cw_lane = (MOD && CW) ? ((int)lane < a.ncan ? c : a.colw[min(max((int)lane - a.ncan, 0), S - a.ncan - 1)] * a.c_mod) : c;
    const float wb_lane = (MOD && CW && (int)lane >= a.ncan) ? 0.f : a.wbias;
    unsigned rowoff[BK];
#pragma unroll
    for (int i = 0; i < BK; ++i) rowoff[i] = rs4 * (unsigned)i;
    auto load_block = [&](int j, float (&dst)[BK]) {
        j = min(max(j, 0), NB - 1);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(lpn + (size_t)(j * BK) * rowstride), 0, (int)(rs4 * (unsigned)min(BK, T - j * BK)), BUF_WORD3);
#pragma unroll
        for (int i = 0; i < BK; ++i)
            dst[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, col4, rowoff[i], 0));
    };
    // the block the LEADING chunk (forward: chunk 0, backward: chunk W - 1) runs in phase ph
    auto block_of = [&](int ph) { return FWD ? ph : NB - 1 - ph; };
    auto emit = [&](int j, const float (&row)[BK]) {
        if (j < 0 || j >= NB) return;                           // (wave-uniform)
#if TK_ROWS_PAIR == 2
        f4 *dst = reinterpret_cast<f4 *>(Er + (size_t)(j % (W + 1)) * (BK * ROW_PITCH) + (size_t)lane * BK);
        if (lane < ROW_PITCH) {
#pragma unroll
            for (int i = 0; i < BK; i += 4)
                dst[i >> 2] = f4{fast_exp2(fmaf(row[i], cw_lane, -wb_lane)), fast_exp2(fmaf(row[i + 1], cw_lane, -wb_lane)),
                                 fast_exp2(fmaf(row[i + 2], cw_lane, -wb_lane)), fast_exp2(fmaf(row[i + 3], cw_lane, -wb_lane))};
        }
#elif TK_ROWS_PAIR
        typedef float f2a __attribute__((ext_vector_type(2)));
        f2a *dst = reinterpret_cast<f2a *>(Er + (size_t)(j % (W + 1)) * (BK * ROW_PITCH)) + lane;
        if (lane < ROW_PITCH) {
#pragma unroll
            for (int i = 0; i < BK; i += 2)
                dst[(i >> 1) * ROW_PITCH] = f2a{fast_exp2(fmaf(row[i], cw_lane, -wb_lane)), fast_exp2(fmaf(row[i + 1], cw_lane, -wb_lane))};
        }
#else
        float *dst = Er + (size_t)(j % (W + 1)) * (BK * ROW_PITCH) + lane;
        if (lane < ROW_PITCH) {
#pragma unroll
            for (int i = 0; i < BK; ++i) dst[i * ROW_PITCH] = fast_exp2(fmaf(row[i], cw_lane, -wb_lane));
        }
#endif
    };
    float r0[BK], r1[BK];
    load_block(block_of(0), r0);
    load_block(block_of(1), r1);
    // lead phase: block_of(0); then during phase ph: block_of(ph + 1)
    emit(block_of(0), r0);
    load_block(block_of(2), r0);
    band_barrier();
    for (int ph = 0; ph < NPH; ph += 2) {
        emit(block_of(ph + 1), r1);
        load_block(block_of(ph + 3), r1);
        band_barrier();
        if (ph + 1 < NPH) {
            emit(block_of(ph + 2), r0);
            load_block(block_of(ph + 4), r0);
            band_barrier();
        }
    }
}

// ---------------- sorted transition instances of a read's 64-cell chunks (for the gradient pass) ----------------
// `ws`: the read's slot in rec / segend; waves w0, w0 + nw, ... of the caller take the chunks in turn.
template <bool MOD>
__device__ __forceinline__ void band_rank(const BandArgs &a, int ws, int L, int64_t off, int w, int nwaves) {
    constexpr int KINDS = MOD ? 3 : 2;
    const int lane = threadIdx.x & (WAVE - 1), S = a.S;
    // ---------------- sorted transition instances of the read's 64-cell chunks ----------------
    // instance = (cell, kind): stay at p | move INTO p | (cat-mod) the mod term of that move,
    // held by the lane that owns p in the gradient pass (position order).  Sort key = transition
    // id (the three kinds use disjoint id ranges), padding last.  Ranks come from ballots in a
    // fixed order, so the permutation -- and with it every floating-point sum of the gradient
    // pass -- is the same from run to run.
    for (int ck = w; ck * WAVE < L; ck += nwaves) {
        int key[KINDS];
        const int p = ck * WAVE + lane;
        const bool has = p >= 1 && p < L;
        if (a.codes != nullptr) {
            const int cp = (p < L) ? band_code(a, off + p) : 0, cb = has ? band_code(a, off + p - 1) : 0;
            key[0] = (p < L) ? band_stay_id(a, cp) : KEY_DEAD;
            key[1] = has ? band_move_id(a, cb, cp) : KEY_DEAD;
            if (MOD) key[MOD ? 2 : 0] = has ? a.ncan + band_mod_seq(a, cp, off + p, nullptr) : KEY_DEAD;
        } else {
            key[0] = (p < L) ? a.stay[off + p] : KEY_DEAD;
            key[1] = has ? a.move[off + p - 1] : KEY_DEAD;
            if (MOD) key[MOD ? 2 : 0] = has ? a.mod[off + p - 1] : KEY_DEAD;
        }
        int cnt = 0;                    // lane b: instances with key b ranked so far
        int rank[KINDS];
#pragma unroll
        for (int e = 0; e < KINDS; ++e) {
            int r = 0;
            for (int b = 0; b < WAVE; ++b) {
                if (b == S + 2) b = KEY_DEAD;               // keys S+2 .. 62 do not occur
                const unsigned long long mask = __ballot(key[e] == b);
                if (key[e] == b)
                    r = __builtin_amdgcn_readlane(cnt, b) + __popcll(mask & ((1ull << lane) - 1ull));
                if (lane == b) cnt += __popcll(mask);
            }
            rank[e] = r;
        }
        const int incl = wave_inclusive_scan_int(cnt);      // lane b: end of key b's segment
        const int start = incl - cnt;
        a.segend[((size_t)ws * a.Wp + ck) * WAVE + lane] = incl;
        uint32_t *recn = a.rec + ((size_t)ws * a.Wp + ck) * KINDS * WAVE;
#pragma unroll
        for (int e = 0; e < KINDS; ++e) {
            const int pos = __builtin_amdgcn_ds_bpermute(key[e] * 4, start) + rank[e];
            // position pos of the sorted order lives in lane pos / KINDS, register pos % KINDS;
            // it reads the LDS word the owner of (cell, kind) writes
            recn[(pos % KINDS) * WAVE + pos / KINDS] = (uint32_t)((lane * KINDS + e) * 4);
        }
    }
}

// ===========================================================================
// sweep + rank launch.  blockIdx.x in [0, N): sorted-instance records for the gradient pass;
// [N, 2N): forward sweep of read n; [2N, 3N): backward sweep.  Cost-only calls launch 2 N
// workgroups: the two sweeps (their scores are compared by the launch behind them: the tail launch's vote pass).
// ===========================================================================
// WCAP = the most waves a launch of this instantiation may have: the register budget of a lane is
// 512 / ceil(WCAP / 4) (R = 4 wants more than the 128 that 16 waves leave).
template <int R, bool MOD, int WCAP, bool ROWS, bool CW, int BK>
__global__ __launch_bounds__(WCAP *WAVE) void crf_band_sweep_kernel(BandArgs a) {
    extern __shared__ __attribute__((aligned(16))) char band_dyn_lds[];     // ROWS: the exponentiated rows (band_rowmaker)
    constexpr int PW = R * WAVE;
    __shared__ __attribute__((aligned(16))) float E[BAND_MAXW * 2 * BK];
    __shared__ int Ef[BAND_MAXW * 2];
    __shared__ __attribute__((aligned(16))) float Ezero[BK];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, W = a.W;
    const bool want_grad = a.grad != nullptr;
#ifndef TK_PRE4
#define TK_PRE4 1
#endif
    // four cells per lane: the step weights' frame factors two steps per v_pk_mul_f32 ahead of the steps (what the
    // narrower forms do) where the launch bound leaves registers for them
    constexpr bool PRE4 = TK_PRE4 && R == 4 && WCAP <= 12;
    const int slot3 = blockIdx.x / N;
    // dispatch order (round 5): the two sweeps first -- 2 N workgroups onto the chip's CUs, one each at the train step's
    // shape --, the rank workgroups behind them (short; they share a CU with a sweep for a few microseconds).  Rank
    // workgroups first (rounds 3-4) left it to their lifetime where the backward sweeps landed: 67.0 against 69.6 us
    // for the launch at the train step's shape, cat-mod 81.4 against 85.0 (profiles/r5_index_build_ab.txt).
#ifndef TK_RANK_LAST
#define TK_RANK_LAST 1
#endif
    const int role = (want_grad && !TK_RANK_LAST) ? (slot3 + 2) % 3 : slot3;       // 0 forward, 1 backward, 2 rank (cost only: 2 N workgroups, the two sweeps)
    const int n = blockIdx.x - slot3 * N;
    __shared__ long long offsh[3 * BAND_MAXW];
    long long off_raw = 0, all_raw = 0;
    int L;
    if (a.codes != nullptr) {
        // the launch builds its indices itself: this read's offset and length from the lengths (see band_offset_of)
        int len_n = 0;
        band_offset_of(a, n, offsh, &off_raw, &all_raw, &len_n);
        L = (int)max(0ll, min((long long)len_n, a.total_len - min(off_raw, a.total_len)));
    } else {
        L = min(a.seqlen[n], (int)(a.seqoff[n + 1] - a.seqoff[n]));            // (offsets are clamped to the label array)
    }
    // (what is left of build_indices_kernel's outputs: the read's offset, for the launches behind -- the gradient pass
    // and the tail launch take a read's offset and length from seqoff; its label checks ride in the forward sweep's set-up)
    if (a.codes != nullptr && role == 0 && tid == 0) {
        int64_t *seqoff = const_cast<int64_t *>(a.seqoff);
        seqoff[n] = min(off_raw, a.total_len);
        if (n == N - 1) {
            seqoff[N] = min(all_raw, a.total_len);
            if (all_raw > a.total_len && a.status) atomicOr(a.status, 8u);      // more labels announced than handed over
        }
    }
    if (role == 2 && tid == 0) a.gate[n] = 0;
    if (!want_grad && role == 0 && tid == 0) a.gate[n] = 2;     // cost only: pending -- the tail launch compares the two sweep scores
    if (a.gate2 != nullptr && role == (want_grad ? 2 : 0) && tid == 0) a.gate2[n] = -1;     // not retried (yet)
    if (a.anygate != nullptr && role == 2 && n == 0 && tid == 0) *a.anygate = 0;
    if (role == 2 && tid < 16) const_cast<float *>(a.zeros)[tid] = 0.f;   // (every rank workgroup: the same zeros)
    if (L == 0 || L > W * PW) {
        // c_crf_flipflop.c:269-272: cost 0 for an empty read (the gradient pass does it when
        // there is one); too long for the launch: flagged
        if (!want_grad && role == 0 && tid == 0) {
            a.cost[n] = (L == 0) ? crf_add_cost(a, n, 0.f) : __builtin_nanf("");
            a.gate[n] = 0;
            if (L != 0 && a.status) atomicOr(a.status, 16u);
        }
        return;
    }
    const int64_t off = a.codes != nullptr ? (int64_t)min(off_raw, a.total_len) : a.seqoff[n];
    (void)PW;

    if (role == 2) {
        band_rank<MOD>(a, n, L, off, w, (int)(blockDim.x >> 6));
        return;
    }

    if (tid < BK) Ezero[tid] = 0.f;
    __syncthreads();
    f4 *Wt = reinterpret_cast<f4 *>(band_dyn_lds);
    if constexpr (ROWS) {
        if (w >= W) {
            // the row maker: the workgroup's exponentiated rows, a phase ahead
            if (role == 0)
                band_rowmaker<MOD, true, CW, BK>(a, n, reinterpret_cast<float *>(band_dyn_lds));
            else
                band_rowmaker<MOD, false, CW, BK>(a, n, reinterpret_cast<float *>(band_dyn_lds));
            return;
        }
    }
    if (!want_grad && role == 0)
        band_sweep<R, MOD, true, false, ROWS, CW, BK, PRE4>(a, n, n, L, off, E, Ef, Ezero, Wt);
    else if (!want_grad)
        band_sweep<R, MOD, false, false, ROWS, CW, BK, PRE4>(a, n, n, L, off, E, Ef, Ezero, Wt);
    else if (role == 0)
        band_sweep<R, MOD, true, true, ROWS, CW, BK, PRE4>(a, n, n, L, off, E, Ef, Ezero, Wt);
    else
        band_sweep<R, MOD, false, true, ROWS, CW, BK, PRE4>(a, n, n, L, off, E, Ef, Ezero, Wt);
}

// Inclusive wave prefix sum in six fused DPP adds (the row_bcast steps write only the rows they
// apply to; hipcc's expansion spends a v_mov_dpp + v_add on each of those and re-zeroes a register).
__device__ __forceinline__ float wave_scan_fused(float x) {
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x));
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x));
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x));
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x));
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(x));
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(x));
    return x;
}

// The same for the RG independent rows of a gradient-pass row group at once (RG = 4, 8 or 12 = the block
// length), level by level: the adds of a level do not depend on each other, so the two wait states a DPP
// read needs after the write of its source are filled with the other rows' adds instead of an s_nop per add
// (48 issue slots per chunk-block at RG = 8; the pass is bound by its instruction count).  One s_nop in front
// covers whatever wrote the inputs.
#define TK_SCAN_ROW(k, CTRL) "v_add_f32_dpp %" #k ", %" #k ", %" #k " " CTRL "\n\t"
#define TK_SCAN4_LEVEL(PRE, CTRL, x)                                                                       \
    asm(PRE TK_SCAN_ROW(0, CTRL) TK_SCAN_ROW(1, CTRL) TK_SCAN_ROW(2, CTRL) TK_SCAN_ROW(3, CTRL)             \
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]))
#define TK_SCAN8_LEVEL(PRE, CTRL, x)                                                                       \
    asm(PRE TK_SCAN_ROW(0, CTRL) TK_SCAN_ROW(1, CTRL) TK_SCAN_ROW(2, CTRL) TK_SCAN_ROW(3, CTRL)             \
            TK_SCAN_ROW(4, CTRL) TK_SCAN_ROW(5, CTRL) TK_SCAN_ROW(6, CTRL) TK_SCAN_ROW(7, CTRL)             \
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]))
#define TK_SCAN12_LEVEL(PRE, CTRL, x)                                                                      \
    asm(PRE TK_SCAN_ROW(0, CTRL) TK_SCAN_ROW(1, CTRL) TK_SCAN_ROW(2, CTRL) TK_SCAN_ROW(3, CTRL)             \
            TK_SCAN_ROW(4, CTRL) TK_SCAN_ROW(5, CTRL) TK_SCAN_ROW(6, CTRL) TK_SCAN_ROW(7, CTRL)             \
            TK_SCAN_ROW(8, CTRL) TK_SCAN_ROW(9, CTRL) TK_SCAN_ROW(10, CTRL) TK_SCAN_ROW(11, CTRL)           \
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),   \
          "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]))
#define TK_SCAN_ALL_LEVELS(LEVEL, x)                                          \
    LEVEL("s_nop 1\n\t", "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", x); \
    LEVEL("", "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1", x);          \
    LEVEL("", "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1", x);          \
    LEVEL("", "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1", x);          \
    LEVEL("", "row_bcast:15 row_mask:0xa bank_mask:0xf", x);                    \
    LEVEL("", "row_bcast:31 row_mask:0xc bank_mask:0xf", x)
__device__ __forceinline__ void wave_scan_fused_rows(float (&x)[4]) { TK_SCAN_ALL_LEVELS(TK_SCAN4_LEVEL, x); }
__device__ __forceinline__ void wave_scan_fused_rows(float (&x)[8]) { TK_SCAN_ALL_LEVELS(TK_SCAN8_LEVEL, x); }
__device__ __forceinline__ void wave_scan_fused_rows(float (&x)[12]) { TK_SCAN_ALL_LEVELS(TK_SCAN12_LEVEL, x); }
#undef TK_SCAN_ALL_LEVELS
#undef TK_SCAN12_LEVEL
#undef TK_SCAN8_LEVEL
#undef TK_SCAN4_LEVEL
#undef TK_SCAN_ROW

// ===========================================================================
// gradient pass: grid (N, ceil(NB / POST_WAVES)), wave = one time block (BK rows) of read n,
// looping over the block's live chunks.  Per chunk: the two checkpoint columns, the boundary
// cells, the backward columns of the block (7 steps), then row by row the forward step whose two
// terms times the backward cell ARE the posteriors; written to wave-private LDS in position order,
// read back sorted by transition id, one DPP prefix scan, segment-end look-ups.
// ===========================================================================
__host__ __device__ inline size_t band_post_lds_bytes(bool mod, int bk) {
    return (size_t)POST_WAVES * bk * (mod ? 3 : 2) * WAVE * 4;
}

// One wave, one time block `jb` of read `n` (workspace slot `ws`: the tail launch's retry, crf_band_tail_kernel); `sP`: the
// wave's own RG x EPL x 64 floats of LDS.  Returns 0, or why the linear path disowns the read: 1 a sweep score is not
// finite, 4 the sweeps disagree, 2 a row of this block lost mass (the caller records it).
template <bool MOD, bool CW, int BK>
__device__ __forceinline__ int band_posterior_block(const BandArgs &a, const int n, const int ws, const int jb, float *sP) {
#define TK_POST_PREAMBLE const int lane = threadIdx.x & (WAVE - 1); const bool head = jb == 0 && lane == 0;
#define TK_POST_WS ws
#define TK_POST_HEAD head
#define TK_POST_LEAVE return 0
#define TK_POST_SWEEPS_DISAGREE(why) return (why)
#define TK_POST_ROWS_LOST return lost ? 2 : 0
#include "crf_band_posterior.inc"
#undef TK_POST_PREAMBLE
#undef TK_POST_WS
#undef TK_POST_HEAD
#undef TK_POST_LEAVE
#undef TK_POST_SWEEPS_DISAGREE
#undef TK_POST_ROWS_LOST
}

// The batch's launch: grid (N, ceil(NB / POST_WAVES)), wave = one time block (BK rows) of read n.
// (reason codes in gate[n], lab dump: 1 non-finite sweep score, 4 sweeps disagree, 2 a row lost mass)
template <bool MOD, bool CW, int BK>
__global__ __launch_bounds__(POST_WAVES *WAVE) __attribute__((amdgpu_waves_per_eu(5))) void crf_band_posterior_kernel(BandArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: SGPR
    const int n = blockIdx.x;
    const int jb = blockIdx.y * POST_WAVES + wave;                  // this wave's time block
    float *sP = reinterpret_cast<float *>(smem) + (size_t)wave * ((MOD && BK > 8) ? 4 : BK) * (MOD ? 3 : 2) * WAVE;
#define TK_POST_PREAMBLE const int lane = tid & (WAVE - 1);
#define TK_POST_WS n
#define TK_POST_HEAD (blockIdx.y == 0 && tid == 0)
#define TK_POST_LEAVE return
#define TK_POST_SWEEPS_DISAGREE(why) do { if (blockIdx.y == 0 && tid == 0) { a.gate[n] = (why); if (a.anygate) *a.anygate = 1; } return; } while (0)
#define TK_POST_ROWS_LOST do { if (lost && lane == 0) { a.gate[n] = 2; if (a.anygate) *a.anygate = 1; } } while (0)
#include "crf_band_posterior.inc"
#undef TK_POST_PREAMBLE
#undef TK_POST_WS
#undef TK_POST_HEAD
#undef TK_POST_LEAVE
#undef TK_POST_SWEEPS_DISAGREE
#undef TK_POST_ROWS_LOST
}

// ===========================================================================
// Round 6 -- THE TAIL LAUNCH of the linear path: the per-read second chance (BandRetry, crf_band.h), and behind it, in
// the same workgroup, the log domain.  Launched behind the batch's gradient pass with a few workgroups of 16 waves.
// Usually nothing is disowned and a workgroup leaves after one pass over the gate array (a cost-only call: one wave
// writes the costs of the reads whose two sweeps agree on the way).  Otherwise the k-th disowned read goes to workgroup
// k mod gridDim.x, which -- one read after the other, in its own slot of the retry workspace --
//   1. ranks the read's transition instances, runs the forward and the backward sweep with 4-step blocks and steep
//      frames (a.klip, a.wbias: crf_band_pick_retry; both sweeps at once where 2 W waves fit the workgroup), then the
//      gradient pass's blocks, a wave per block in turn: the device functions of the batch's launches, one workgroup
//      instead of 3 + NB / 2.  ~300 us for a read of 720 bases at T 800; the batch keeps its fast configuration;
//   2. if that disowns the read as well (or the call has no retry configuration): crf_read (crf_log.h), the log-domain
//      form that takes any input the reference takes, ~1 ms at T 800.
// One launch does what round 5's gated crf_kernel launch did and what a retry launch of its own would: the op's fourth
// launch is gone (each costs ~3 us on the stream whatever it finds: profiles/r6_tail_launches_ab.txt).
// Status word: bits 20-31 count the reads retried, bits 8-19 the reads redone in the log domain.
// ===========================================================================
__device__ __forceinline__ int band_first_verdict(const BandRetry &r, int n, double *score2) {
    const int g = r.gate[n];
    if (r.firstF == nullptr || g != 2) return g;
    const double F = r.firstF[n], B = r.firstB[n], d = F - B;
    if (!(F - F == 0.0 && B - B == 0.0)) return 1;              // overflow / nothing left: not representable
    if (!(d > -1e-3 && d < 1e-3)) return 4;                     // mass lost on the way in one of them
    *score2 = 0.5 * (F + B);
    return 0;
}

template <int R, bool MOD, bool CW>
__global__ __launch_bounds__(BAND_MAXW *WAVE) void crf_band_tail_kernel(BandArgs a, BandRetry r, CrfArgs ca) {
    constexpr int BK = 4, PW = R * WAVE, KINDS = MOD ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) char retry_dyn_lds[];    // the gradient pass's rows: 16 waves x BK x KINDS x 64 floats
                                                                            // (crf_read lays its own image over the same bytes)
    __shared__ __attribute__((aligned(16))) float E[2][BAND_MAXW * 2 * BK];    // (a ring per sweep)
    __shared__ int Ef[2][BAND_MAXW * 2];
    __shared__ __attribute__((aligned(16))) float Ezero[BK];
    __shared__ int why_sh;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = (int)(blockDim.x >> 6);
    const bool want_grad = a.grad != nullptr;
    const int T = a.T, NB = (T + BK - 1) / BK;
    // the common path of a gradient call: the batch's gradient pass disowned nobody -- one word says so
    if (r.anygate != nullptr && *r.anygate == 0) return;
    {
        const bool writer = blockIdx.x == 0 && w == 0;          // (one wave writes the cost-only calls' costs)
        unsigned long long any = 0;
        for (int n0 = 0; n0 < a.N; n0 += WAVE) {
            const int n = n0 + lane;
            double score2 = 0.0;
            const int g = n < a.N ? band_first_verdict(r, n, &score2) : 0;
            if (writer && n < a.N && g == 0 && r.firstF != nullptr && r.gate[n] == 2) {
                // score = mean of the two sweeps (c_crf_flipflop.c:482-491 does the same), cost = -score / T; the
                // bias comes back: every one of the T step weights on a path carried 2^-wbias
                const float cst = crf_add_cost(a, n, (float)(-((score2 + (double)r.first_wbias * (double)T) * 0.6931471805599453) / (double)T) * a.out_scale);
                a.cost[n] = cst;
                if (a.status && !isfinite(cst)) atomicOr(a.status, 1u);
            }
            any |= __ballot(g != 0);
        }
        if (any == 0) return;
    }
    if (tid < BK) Ezero[tid] = 0.f;
    const int ws = (int)blockIdx.x;
    int seen = 0, redone = 0;
    for (int n = 0; n < a.N; ++n) {
        double unused;
        if (band_first_verdict(r, n, &unused) == 0) continue;
        const bool mine = seen % (int)gridDim.x == (int)blockIdx.x;
        ++seen;
        if (!mine) continue;
        const int64_t off = a.seqoff[n];
        const int L = min(a.seqlen[n], (int)(a.seqoff[n + 1] - off));
        __syncthreads();                                        // (the read before: its verdict is taken, its LDS free)
        if (tid == 0) why_sh = (r.retry && L > 0 && L <= a.W * PW) ? 0 : 16;
        if (r.retry && L > 0 && L <= a.W * PW) {
            band_rank<MOD>(a, ws, L, off, w, nwaves);
            __syncthreads();
            if (2 * a.W <= nwaves) {
                // both sweeps at once: waves [0, W) forward, [W, 2 W) backward -- the same number of phases, hence of barriers;
                // the rest of the workgroup keeps them company
                if (w < a.W) band_sweep<R, MOD, true, true, false, CW, BK>(a, n, ws, L, off, E[0], Ef[0], Ezero, nullptr);
                else if (w < 2 * a.W) band_sweep<R, MOD, false, true, false, CW, BK>(a, n, ws, L, off, E[1], Ef[1], Ezero, nullptr, a.W);
                else
                    for (int ph = 0; ph < NB + a.W - 1; ++ph) band_barrier();
            } else {
                band_sweep<R, MOD, true, true, false, CW, BK>(a, n, ws, L, off, E[0], Ef[0], Ezero, nullptr);
                __syncthreads();
                band_sweep<R, MOD, false, true, false, CW, BK>(a, n, ws, L, off, E[1], Ef[1], Ezero, nullptr);
            }
            // what the sweeps and the ranking left in the workspace is read by OTHER waves of this workgroup below: release,
            // barrier, acquire (a slot's lines may sit in this CU's vector cache from the read before)
            __threadfence();
            __syncthreads();
            __threadfence();
            if (want_grad) {
                float *sP = reinterpret_cast<float *>(retry_dyn_lds) + (size_t)w * BK * KINDS * WAVE;
                int why = 0;
                for (int jb = w; jb < NB && (why & 5) == 0; jb += nwaves)       // (the sweeps' verdict, 1 / 4, is every block's)
                    why |= band_posterior_block<MOD, CW, BK>(a, n, ws, jb, sP);
                if (why != 0 && lane == 0) atomicOr(&why_sh, why);
            } else if (tid == 0) {
                // cost only: both sweeps finite and agreeing, or not the linear path's
                const double F = a.scoreF[ws], B = a.scoreB[ws], d = F - B;
                if (!(F - F == 0.0 && B - B == 0.0)) why_sh = 1;
                else if (!(d > -1e-3 && d < 1e-3)) why_sh = 4;
                else {
                    const double score2 = 0.5 * (F + B) + (double)a.wbias * (double)T;
                    const float cst = crf_add_cost(a, n, (float)(-(score2 * 0.6931471805599453) / (double)T) * a.out_scale);
                    a.cost[n] = cst;
                    if (a.status && !isfinite(cst)) atomicOr(a.status, 1u);
                }
            }
        }
        __syncthreads();
        const int why_all = why_sh;
        if (tid == 0 && r.gate2 != nullptr) r.gate2[n] = why_all;
        if (why_all != 0 && r.log_domain) {
            // nobody on the linear path: the log domain, in this workgroup (its cost and gradient rows overwrite whatever
            // the linear attempts left)
            __syncthreads();
            crf_read<crf_tail_log_R(R), BAND_MAXW, MOD>(ca, n, (int)blockIdx.x);
            ++redone;
        }
    }
    if (tid == 0 && a.status) {
        // reads retried (bits 20-31; counted once, by workgroup 0) and reads redone in the log domain (bits 8-19; every
        // workgroup its own) -- include/taiyaki_amd_flipflop.h: TK_STATUS_RETRIED_SHIFT, TK_STATUS_GATED_SHIFT
        uint32_t add = (uint32_t)min(redone, 0xfff) << 8;
        if (blockIdx.x == 0 && r.retry) add += (uint32_t)min(seen, 0xfff) << 20;
        if (add) atomicAdd(a.status, add);
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// Cells per lane: the smallest R whose chunks fit the 16 waves of a workgroup.
int crf_band_pick_R(size_t max_seqlen, bool mod) {
    int R = 1;
    // The plain CRF takes two cells per lane from 513 bases on (nine chunk waves and a row maker at one cell per lane
    // are ten waves per workgroup: every wave shares its SIMD, and no second workgroup fits a CU beside it -- batches of
    // more than 128 reads run their 2 N sweeps in two rounds.  Measured, round 5, profiles/r5_cells_per_lane.txt: reads up
    // to 533 bases at T 800: N 128 102.2 -> 100.4 us, N 192 156.5 -> 125.2, N 256 179.6 -> 144.7; reads up to 399 bases
    // (seven chunk waves) and every cat-mod shape tried are faster at one cell per lane.)
    if (!mod && max_seqlen > (size_t)8 * WAVE) R = 2;
    if (const char *e = TK_LAB_ENV("TK_CRF_BAND_R")) {
        R = atoi(e);
        if (R != 1 && R != 2 && R != 4) R = 1;
    }
    // (one wave of the workgroup is the row maker: 15 chunk waves below four cells per lane -- a read of
    // 961 .. 1024 bases takes two cells per lane and keeps the shared rows: 204 us against 247 at T 1600 / N 64)
    while (R < 4 && (size_t)R * WAVE * (BAND_MAXW - 1) < max_seqlen) R *= 2;
    return R;
}

bool crf_band_fits(size_t max_seqlen) { return max_seqlen <= (size_t)4 * WAVE * BAND_MAXW; }

// Block length and weight bias for a call (see BK_MAX): `sharp` = the sharpening factor of the canonical
// columns.  bk = 0: the linear path does not take this call (the log-domain kernel does every read).
// TK_CRF_BK = 4 | 8 | 12 forces a block length, TK_CRF_WBIAS a bias (lab: tools/crf_gate_probe.py).
BandBlock crf_band_pick_block(float sharp, bool mod, size_t max_seqlen, bool colw, size_t nblk, size_t bulk_seqlen) {
    BandBlock b{8, 0.f, KLIP};
    const float x = sharp > 0.f ? sharp : 1.f;
    // NARROW BANDS (round 5).  A band of T - L + 2 cells carries its mass along its two fronts, where a cell's value is a
    // product of FORCED moves and falls by 15 .. 30 bits per cell (cat-mod: a modification penalty per move) -- faster than
    // frames of slope KLIP = 6 can follow, so the front's cells were flushed, the two sweeps disagreed and the read went to
    // the log-domain kernel (~1 ms at T 800): the plain CRF from L ~ 0.85 T on, cat-mod from 0.70 T on under iid scores
    // (profiles/r5_gate_by_band_width.txt; found with tests/helpers/crf_linear_model.py, which loses the same mass at the
    // same cell).  A steeper slope needs shorter blocks: growth per step is (1 + 2^slope) x the largest weight, and
    // 8 x (7.2 - 3 + 11.0) = 121.6 bits fit fp32 where 12 x (7.2 - 3 + 6.02) = 122.6 did.  A batch whose longest read may
    // be narrower than that takes 8-step blocks, bias 3 and slope 11: ~11 % slower sweeps for the plain CRF, the same
    // block length for cat-mod below 705 bases.
    // ROUND 6: these are rules for the batch's BULK now, not for its longest read: `bulk_seqlen` is a length all but a few
    // reads stay below (the caller's host-side knowledge; 0 = unknown = no such rule).  The few beyond it run the fast
    // configuration with everybody else, and whichever of them the linear path then disowns is retried alone at
    // 4 steps / slope 20 (crf_band_retry_kernel) -- one long read no longer moves 127 others to slower blocks.
    const bool narrow = nblk > 0 && (double)bulk_seqlen > (mod ? 0.62 : 0.78) * (double)nblk;
    // (cat-mod beyond 0.78 T: 4-step blocks carry slope 20 -- 4 x (7.2 + 20) = 108.8 bits; the model keeps L = 0.9 T with it)
    if (x <= 1.03f && mod && nblk > 0 && (double)bulk_seqlen > 0.78 * (double)nblk) b = {4, 0.f, 20};
    else if (x <= 1.03f && narrow && (!mod || colw)) b = {8, 3.f, 11};
    else if (!mod && x <= 1.03f) b = {12, 3.f, KLIP};
    // cat-mod with per-column factors (round 5): the same 12-step blocks and bias from 705 bases on -- measured,
    // profiles/r5_catmod_bk12.txt: reads up to 799 bases 215 -> 199 us, T 1600 (two cells per lane) 255 -> 230, T 4000 / N 256
    // 1997 -> 1815, and fewer reads disowned at long T; reads up to 533 / 666 bases are faster at 8 (128.6 / 162.2 against
    // 132.4 / 164.8 us; N 256: two ten-wave workgroups no longer share a CU at 12 steps' registers).  The general
    // per-position form has no 12-step instantiation.
    else if (mod && colw && x <= 1.03f && max_seqlen > (size_t)11 * WAVE) b = {12, 3.f, KLIP};
    else if (x <= 1.36f) b = {8, 0.f, KLIP};
    else if (x <= 1.76f) b = {8, 3.f, KLIP};
    else if (x <= 3.5f) b = {4, 0.f, KLIP};
    else b = {0, 0.f, KLIP};
    if (const char *e = TK_LAB_ENV("TK_CRF_BK")) {
        const int v = atoi(e);
        if (v == 4 || v == 8 || (v == 12 && (!mod || colw))) b.bk = v;
    }
    if (const char *e = TK_LAB_ENV("TK_CRF_WBIAS")) b.wbias = (float)atof(e);
    if (const char *e = TK_LAB_ENV("TK_CRF_KLIP")) b.klip = atoi(e);
    return b;
}

// The second chance's configuration: 4-step blocks with the steepest frames their growth bound allows,
//   4 x (7.2 sharp - bias + slope) <= 125.6 bits  and, for the decay,  7.2 sharp + bias <= 30.5 bits per step:
// slope 20 without a bias up to a sharpening factor of 1.58 (what round 5's cat-mod rule for narrow bands ran: "keeps everything
// tried"), slope 20 with a bias up to 2.8, 18 at 3.0, 11 at 3.5.
BandBlock crf_band_pick_retry(float sharp, BandBlock fast) {
    const float x = sharp > 0.f ? sharp : 1.f;
    if (fast.bk == 0 || x > 3.5f || TK_LAB_ENV("TK_CRF_NO_RETRY")) return {0, 0.f, KLIP};
    const float g = 7.2f * x;
    float bias = 0.f;
    int klip = 20;
    if (g + 20.f > 31.4f) {
        const float need = ceilf(2.f * (g + 20.f - 31.4f)) * 0.5f, bmax = floorf(2.f * fmaxf(0.f, 30.5f - g)) * 0.5f;
        bias = fminf(need, bmax);
        klip = min(20, (int)floorf(31.4f - g + bias));
    }
    if (const char *e = TK_LAB_ENV("TK_CRF_RETRY_KLIP")) klip = atoi(e);
    if (const char *e = TK_LAB_ENV("TK_CRF_RETRY_WBIAS")) bias = (float)atof(e);
    if (klip <= KLIP || (fast.bk == 4 && fast.klip >= klip)) return {0, 0.f, KLIP};     // (the batch's launch ran that already)
    return {4, bias, klip};
}

// Slots of the retry workspace = workgroups of the retry launch: a sixteenth of the batch, at least 4.
size_t crf_band_retry_slots(size_t nbatch) {
    size_t s = (nbatch + 15) / 16;
    if (s < 4) s = 4;
    return s < nbatch ? s : nbatch;
}

int crf_band_retry_R(size_t max_seqlen) {
    int R = 1;
    while (R < 4 && (size_t)R * WAVE * (BAND_MAXW / 2) < max_seqlen) R *= 2;
    return R;
}

BandLayout crf_band_layout(size_t ntrans, size_t nblk, size_t nbatch, size_t max_seqlen, bool mod,
                           bool want_grad, int bk, int force_R) {
    (void)ntrans;
    BandLayout l;
    const size_t BK = (size_t)(bk > 0 ? bk : 8);
    l.BK = (int)BK;
    l.R = force_R > 0 ? force_R : crf_band_pick_R(max_seqlen, mod);
    const size_t PW = (size_t)l.R * WAVE;
    l.W = (int)((max_seqlen + PW - 1) / PW);
    if (l.W < 1) l.W = 1;
    l.LP = (size_t)l.W * PW;
    const size_t KINDS = mod ? 3 : 2, NB = (nblk + BK - 1) / BK, Wp = l.LP / WAVE;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t r = off;
        if (want_grad) off += (bytes + 255) / 256 * 256;        // (a cost-only call needs the gate alone)
        return r;
    };
    l.ckFm = take(nbatch * NB * l.LP * sizeof(float));
    l.ckBm = take(nbatch * NB * l.LP * sizeof(float));
    l.ckFf = take(nbatch * NB * l.LP * sizeof(int16_t));
    l.ckBf = take(nbatch * NB * l.LP * sizeof(int16_t));
    l.ckFb = take(nbatch * NB * l.W * sizeof(int));
    l.ckBb = take(nbatch * NB * l.W * sizeof(int));
    l.bndF = take(nbatch * NB * Wp * BK * sizeof(float));
    l.bndB = take(nbatch * NB * Wp * BK * sizeof(float));
    // (the two sweep scores also in a cost-only call: it runs both sweeps and is only believed where they agree)
    l.scoreF = off;
    off += (nbatch * sizeof(double) + 255) / 256 * 256;
    l.scoreB = off;
    off += (nbatch * sizeof(double) + 255) / 256 * 256;
    l.rec = take(nbatch * Wp * KINDS * WAVE * sizeof(uint32_t));
    l.segend = take(nbatch * Wp * WAVE * sizeof(int));
    l.gate = off;
    off += (nbatch * sizeof(int) + 255) / 256 * 256;
    l.gate2 = off;              // the retry launch's verdicts, per read of the batch
    off += (nbatch * sizeof(int) + 255) / 256 * 256;
    l.anygate = off;
    off += 256;
    l.zeros = off;
    off += 256;
    l.total = off + 256;
    return l;
}

// lab knob (tools/overlap_probe.py): 0 both passes, 1 the sweep launch only, 2 the gradient pass only
#ifdef TK_LAB
static int g_band_lab_phase = 0;
void crf_band_lab_phase(int phase) { g_band_lab_phase = phase; }
#else
constexpr int g_band_lab_phase = 0;
#endif

// Does this launch run with a row maker (band_rowmaker)?  Whenever W + 1 waves fit a workgroup and the weights
// are gathers from an exponentiated row (the plain CRF; cat-mod with per-column factors); 4-step blocks (sharpened
// calls) keep the plain feed.  TK_CRF_FEED = self | rows forces one (lab, tests).
static bool band_use_rows(const BandArgs &a, bool mod, int bk) {
    if (bk < 8 || a.W + 1 > BAND_MAXW || (mod && a.colw == nullptr)) return false;
    if (a.S > ROW_PITCH) return false;      // a row image holds ROW_PITCH columns: wider rows (cat-mod with >= 5
                                            // modifications, nbase 5) gather from their own exponentiated rows
    if (const char *e = TK_LAB_ENV("TK_CRF_FEED")) return e[0] == 'r';
    return true;
}

template <int R, bool MOD, bool CW, int BK, bool ROWS>
static int band_launch_sweep(const BandArgs &a, hipStream_t stream) {
    const bool want_grad = a.grad != nullptr;
    const int nw = a.W + (ROWS ? 1 : 0);
    const size_t lds = ROWS ? (size_t)(a.W + 1) * BK * ROW_PITCH * sizeof(float) : 0;
    const dim3 grid((want_grad ? 3 : 2) * a.N), block(nw * WAVE);
    // (R = 4 is compiled per wave-count class: 155 registers where the launch bounds allow them)
    auto go = [&](auto cap) {
        constexpr int WCAP = decltype(cap)::value;
        hipLaunchKernelGGL((crf_band_sweep_kernel<R, MOD, WCAP, ROWS, CW, BK>), grid, block, lds, stream, a);
    };
    if constexpr (R == 4 && BK >= 8) {
        if (nw <= 8) go(std::integral_constant<int, 8>{});
        else if (nw <= 12) go(std::integral_constant<int, 12>{});
        else go(std::integral_constant<int, BAND_MAXW>{});
    } else {
        go(std::integral_constant<int, BAND_MAXW>{});
    }
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

template <int R, bool MOD, bool CW, int BK>
static int band_launch(const BandArgs &a, hipStream_t stream) {
    const bool want_grad = a.grad != nullptr;
    if (g_band_lab_phase != 2) {
        int rc;
        bool rows = false;
        if constexpr (BK >= 8 && (!MOD || CW)) rows = band_use_rows(a, MOD, BK);
        if constexpr (BK >= 8 && (!MOD || CW)) {
            rc = rows ? band_launch_sweep<R, MOD, CW, BK, true>(a, stream) : band_launch_sweep<R, MOD, CW, BK, false>(a, stream);
        } else {
            rc = band_launch_sweep<R, MOD, CW, BK, false>(a, stream);
        }
        if (rc != 0) return rc;
    }
    if (!want_grad || g_band_lab_phase == 1) return 0;
    if (a.before_gradient != nullptr && hipStreamWaitEvent(stream, a.before_gradient, 0) != hipSuccess) return 4;
    const int NB = (a.T + BK - 1) / BK;
    hipLaunchKernelGGL((crf_band_posterior_kernel<MOD, CW, BK>), dim3(a.N, (NB + POST_WAVES - 1) / POST_WAVES),
                       dim3(POST_WAVES * WAVE), band_post_lds_bytes(MOD, BK), stream, a);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// block lengths per form: the plain CRF and cat-mod with per-column factors have all three (crf_band_pick_block says which
// a call takes), the general per-position cat-mod form 8 and 4
template <int R, bool MOD, bool CW>
static int band_launch_bk(const BandArgs &a, int bk, hipStream_t stream) {
    if (bk == 4) return band_launch<R, MOD, CW, 4>(a, stream);
    if (bk == 8) return band_launch<R, MOD, CW, 8>(a, stream);
    // (12-step blocks: the plain CRF and cat-mod with per-column factors; the general per-position cat-mod form stays at 8)
    if constexpr (!MOD || CW) {
        if (bk == 12) return band_launch<R, MOD, CW, 12>(a, stream);
    }
    return 2;
}

int crf_band_dispatch(const BandArgs &a0, int R, bool mod, int bk, hipStream_t stream) {
    BandArgs a = a0;
#ifdef TK_LAB_STAMPS
    static unsigned long long *dbg = nullptr;
    if (TK_LAB_ENV("TK_CRF_STAMPS")) {
        if (!dbg) (void)hipMalloc(&dbg, 1024 * 8);
        (void)hipMemsetAsync(dbg, 0, 1024 * 8, stream);
        a.dbg = dbg;
    }
    struct Printer {
        unsigned long long *d;
        hipStream_t s;
        ~Printer() {
            if (!d) return;
            static unsigned long long h[1024];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            fprintf(stderr, "gradient pass: %llu of %llu chunk-blocks skipped; per wave (%llu waves, %.2f live chunks): prologue %.0f, "
                    "mask pass %.0f, bodies %.0f, epilogue %.0f clocks\n", h[600], h[601], h[620], (double)h[621] / (double)(h[620] ? h[620] : 1),
                    (double)h[610] / (double)(h[620] ? h[620] : 1), (double)h[611] / (double)(h[620] ? h[620] : 1),
                    (double)h[612] / (double)(h[620] ? h[620] : 1), (double)h[613] / (double)(h[620] ? h[620] : 1));
            for (int k = 2; k < 12; ++k)
                fprintf(stderr, "phase %2d: loads+frames %5llu  steps %5llu  boundary %5llu  barrier %5llu  next-phase-gap %5llu\n", k,
                        h[k * 8 + 1] - h[k * 8 + 0], h[k * 8 + 2] - h[k * 8 + 1], h[k * 8 + 3] - h[k * 8 + 2],
                        h[k * 8 + 4] - h[k * 8 + 3], h[(k + 1) * 8 + 0] - h[k * 8 + 4]);
        }
    } printer{a.dbg, stream};
#endif
    if (a.W < 1 || a.W > BAND_MAXW) return 2;
    switch (R * 2 + (mod ? 1 : 0)) {
        case 2: return band_launch_bk<1, false, false>(a, bk, stream);
        case 3: return a.colw ? band_launch_bk<1, true, true>(a, bk, stream) : band_launch_bk<1, true, false>(a, bk, stream);
        case 4: return band_launch_bk<2, false, false>(a, bk, stream);
        case 5: return a.colw ? band_launch_bk<2, true, true>(a, bk, stream) : band_launch_bk<2, true, false>(a, bk, stream);
        case 8: return band_launch_bk<4, false, false>(a, bk, stream);
        case 9: return a.colw ? band_launch_bk<4, true, true>(a, bk, stream) : band_launch_bk<4, true, false>(a, bk, stream);
        default: return 2;
    }
}

template <int R, bool MOD, bool CW>
static int band_tail_launch(const BandArgs &a, const BandRetry &r, const CrfArgs &ca, size_t nslots, hipStream_t stream) {
    // dynamic LDS: the gradient pass's rows of the retry, or the log-domain form's image, whichever is larger
    const size_t lds_retry = (size_t)BAND_MAXW * 4 * (MOD ? 3 : 2) * WAVE * sizeof(float);
    const size_t lds_log = crf_lds_bytes(crf_tail_log_R(R), BAND_MAXW, a.S, MOD ? 3 : 2);
    const size_t lds = lds_retry > lds_log ? lds_retry : lds_log;
    if (lds > 150 * 1024) return 2;
    if (raise_dynamic_lds(reinterpret_cast<const void *>(&crf_band_tail_kernel<R, MOD, CW>), 150 * 1024)) return 4;    // (+ ~1.5 KB static)
    hipLaunchKernelGGL((crf_band_tail_kernel<R, MOD, CW>), dim3((unsigned)nslots), dim3(BAND_MAXW * WAVE), lds, stream, a, r, ca);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

int crf_band_tail_dispatch(const BandArgs &a, const BandRetry &r, const CrfArgs &ca, int R, bool mod, size_t nslots, hipStream_t stream) {
    if (a.W < 1 || a.W > BAND_MAXW || nslots == 0) return 2;
    switch (R * 2 + (mod ? 1 : 0)) {
        case 2: return band_tail_launch<1, false, false>(a, r, ca, nslots, stream);
        case 3: return a.colw ? band_tail_launch<1, true, true>(a, r, ca, nslots, stream) : band_tail_launch<1, true, false>(a, r, ca, nslots, stream);
        case 4: return band_tail_launch<2, false, false>(a, r, ca, nslots, stream);
        case 5: return a.colw ? band_tail_launch<2, true, true>(a, r, ca, nslots, stream) : band_tail_launch<2, true, false>(a, r, ca, nslots, stream);
        case 8: return band_tail_launch<4, false, false>(a, r, ca, nslots, stream);
        case 9: return a.colw ? band_tail_launch<4, true, true>(a, r, ca, nslots, stream) : band_tail_launch<4, true, false>(a, r, ca, nslots, stream);
        default: return 2;
    }
}

}  // namespace tk
