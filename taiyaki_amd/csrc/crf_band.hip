// crf_band.hip -- kernel A, band mode: sequence-constrained flip-flop CRF score + gradient
// (plain and cat-mod) as a BANDED, SKEWED sweep for gfx950.
//
// Replaces taiyaki/ctc/c_crf_flipflop.c:43-516 and c_cat_mod_flipflop.c:37-582 (forward,
// backward, posterior scatter).  tests/helpers/crf_skew_model.py states the schedule in numpy and is
// checked against the oracle on the CPU.
//
//   * The L lattice positions of a read are cut into chunks of PW = 64 R cells, one wavefront
//     each (lane l owns cells [l R, (l+1) R) of its chunk in registers).  Time is cut into
//     blocks of BK = 8 steps.  Chunk w runs time block j in phase j + w (forward sweep) resp.
//     (NB-1-j) + (W-1-w) (backward sweep): the boundary cell a chunk needs from its neighbour
//     for every step of a block was written to a two-slot LDS ring one phase earlier, so a
//     workgroup executes ONE s_barrier per BK time steps (the per-step version spent ~320 ns
//     per step on the barrier, an LDS round trip and a vmcnt(0) drain of the lattice store).
//     Inside a block a step is R cells per lane, one DPP wave shift and 2R ds_bpermute
//     gathers from the score row, which the wave holds in ONE VGPR (lane = transition id):
//     no LDS tile, no cross-wave dependency, rows are prefetched two blocks ahead.
//   * Every chunk carries its own INTEGER log2 offset and renormalises every BNORM = 8 steps
//     by floor(max(own cells, incoming boundary cells)): subtracting an integer is exact in
//     fp32 and the offsets add exactly in int32 -- no fp64 on the path.  (The reference
//     subtracts the column maximum every step, c_crf_flipflop.c:73-77; any common offset is
//     exact to account for.)
//   * Only the BAND is computed and stored: cell (t, p) lies on a complete path iff p <= t and
//     L-1-p <= T-t, so chunk w is live for T-L+PW+2 of the T steps.  Dead (chunk, block) pairs
//     cost one s_barrier; roughly half of the lattice traffic of the full rectangle goes away.
//   * crf_band_posterior_kernel: one wave per ROW, looping over the row's live chunks.  The
//     2 PW (3 PW for cat-mod) transition instances of a chunk are evaluated in an order SORTED
//     by transition id (the permutation is computed once per read by "rank" workgroups that
//     ride along in the sweep launch), so the per-id sums of the reference's scatter-add
//     (c_crf_flipflop.c:403-412) are differences of ONE prefix scan held in registers (DPP
//     scan + ds_bpermute look-ups): no atomics, no cross-wave reduction, no barrier, and the
//     summation order is fixed -> bitwise reproducible.
#include <stdio.h>
#include <stdlib.h>

#include "crf_band.h"
#include "ff_common.h"

namespace tk {

#ifndef TK_BAND_BK
#define TK_BAND_BK 8
#endif
constexpr int BK = TK_BAND_BK;               // time steps per block (= per workgroup barrier)
constexpr int BNORM = 8;            // steps between renormalisations (a multiple of 4 that divides BK)
constexpr int BSUB = BK / BNORM;
constexpr int BAND_MAXW = 16;       // waves per workgroup
constexpr int POST_WAVES = 8;       // waves per posterior workgroup
constexpr int POST_ROWS = 8;        // consecutive rows per posterior wave
constexpr int KEY_DEAD = 63;        // sort key of padding instances

struct Win {
    int j0, j1;                     // first / last live time block (j0 > j1: never live)
};

// Live time blocks of chunk w (tests/helpers/crf_skew_model.py: windows()).
__device__ __forceinline__ Win band_window(int w, int PW, int L, int T) {
    const int a = w * PW;
    if (w < 0 || a >= L) return {1, 0};
    const int NB = (T + BK - 1) / BK;
    if (L > T + 1) return {0, NB - 1};          // no complete path: nothing to trim by
    const int b = min(a + PW - 1, L - 1);
    const int tlo = max(0, a - 1), thi = min(T - 1, b + T - L + 1);
    return {tlo / BK, thi / BK};
}

__device__ __forceinline__ bool want_grad_launch(const BandArgs &a) { return a.grad != nullptr; }

__device__ __forceinline__ void band_barrier() {
    // LDS traffic only: the lattice stores stay in flight across the barrier
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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

template <int R>
__device__ __forceinline__ void band_store_cells(float *dst, const float (&x)[R]) {
    if constexpr (R == 4) {
        *reinterpret_cast<f4 *>(dst) = f4{x[0], x[1], x[2], x[3]};
    } else if constexpr (R == 2) {
        *reinterpret_cast<f2 *>(dst) = f2{x[0], x[1]};
    } else {
        dst[0] = x[0];
    }
}

// One score row of read n as ONE register per wave: lane s < S holds lp[t][n][s], lane S the
// -LARGE sentinel (padding positions), lane S+1 the 0.0 sentinel (absent mod term).
__device__ __forceinline__ float band_row(const float *lpn, size_t rowstride, int t, int col, float sent,
                                          bool is_col) {
    const float x = lpn[(size_t)t * rowstride + col];
    return is_col ? x : sent;
}

// Wave-wide maximum as ONE scalar: six v_max_f32_dpp (butterflies inside the 16-lane rows, then
// row_bcast 15 / 31) leave it in lane 63.  hipcc's own expansion of the same reduction is ~25
// instructions (v_mov_dpp + two v_max per stage); the sweep is issue-bound, so this matters.
__device__ __forceinline__ float wave_max_scalar(float x) {
    // one statement per instruction (each carries its own DPP-hazard wait states) so that the
    // scheduler may interleave the reduction with the steps it runs beside
#define TK_ASM asm
    TK_ASM("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
    TK_ASM("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(x));
    TK_ASM("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(x));
    TK_ASM("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(x));
    TK_ASM("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(x));
    TK_ASM("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(x));
#undef TK_ASM
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

constexpr int BUF_WORD3 = 0x00027000;       // raw buffer descriptor, 32-bit data (gfx9 family)

template <int R>
__device__ __forceinline__ void band_buffer_store(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff,
                                                  const float (&x)[R]) {
    if constexpr (R == 4) {
        // two 8-byte stores, not one 16-byte store: a buffer store of more than 64 bits whose data
        // registers are overwritten by the next VALU instruction stores the NEW value of a dword on
        // gfx950 also when soffset is an SGPR (observed: dword 1 of the column replaced by the
        // following v_sub's result) -- hipcc only guards the immediate-soffset form of that hazard
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(u2{__float_as_uint(x[0]), __float_as_uint(x[1])}, rs, voff, soff, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u2{__float_as_uint(x[2]), __float_as_uint(x[3])}, rs, voff + 8u, soff, 0);
    } else if constexpr (R == 2) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(u2{__float_as_uint(x[0]), __float_as_uint(x[1])}, rs, voff, soff, 0);
    } else {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x[0]), rs, voff, soff, 0);
    }
}

template <int R, bool MOD, bool FWD, bool GRAD>
__device__ __forceinline__ void band_sweep(const BandArgs &a, int n, int L, float *E, int *Eoff,
                                           float *Escratch) {
    constexpr int PW = R * WAVE;
    // the wave index is wave-uniform: keep everything derived from it in SGPRs
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & (WAVE - 1);
    const int N = a.N, T = a.T, S = a.S, W = a.W;
    const int64_t off = a.seqoff[n];
    const int p0 = w * PW + lane * R;
    const float neg = NEG_LARGE * LOG2E;
    const int NB = (T + BK - 1) / BK, NPH = NB + W - 1;
    const int src = FWD ? w - 1 : w + 1;                        // the chunk our boundary cell comes from
    const Win win = band_window(w, PW, L, T);
    const Win wsrc = (src >= 0 && src < W) ? band_window(src, PW, L, T) : Win{1, 0};
    if (win.j0 > win.j1) {
        // a chunk past the end of this read: keep the workgroup's barriers company
        for (int ph = 0; ph < NPH; ++ph) band_barrier();
        return;
    }
    const size_t rowstride = (size_t)N * S;
    const float *lpn = a.lp + (size_t)n * S;
    const unsigned col4 = 4u * (unsigned)min(lane, S - 1);
    const bool is_col = lane < S;
    const float sent = (lane == S) ? NEG_LARGE : 0.f;
    const float c = a.c_can;

    // transition ids as ds_bpermute byte addresses.  Forward: cell p takes the move INTO p
    // (from p-1); backward: the move OUT of p.
    int st4[R], mv4[R], md4[MOD ? R : 1];
    float fw[MOD ? R : 1];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int p = p0 + j, q = FWD ? p - 1 : p;              // the move's source position
        st4[j] = 4 * ((p < L) ? a.stay[off + p] : S);
        const bool has = q >= 0 && q < L - 1;
        mv4[j] = 4 * (has ? a.move[off + q] : S);
        if (MOD) {
            md4[j] = 4 * (has ? a.mod[off + q] : S + 1);
            fw[j] = has ? a.modfact[off + q] * a.c_mod : 0.f;
        }
    }

    float x[R];                                                 // this lane's lattice cells
#pragma unroll
    for (int j = 0; j < R; ++j) x[j] = (p0 + j == (FWD ? 0 : L - 1)) ? 0.f : neg;     // :113-116, :216-220
    int offacc = 0;                                             // log2 offset of this chunk
    float *latn = GRAD ? (FWD ? a.latF : a.latB) + (size_t)n * T * a.LP + w * PW : nullptr;
    const unsigned lane_cell4 = 4u * (unsigned)(lane * R);
    const unsigned lp4 = 4u * (unsigned)a.LP, rs4 = 4u * (unsigned)rowstride;
    const int NSUB = (T + BNORM - 1) / BNORM;
    int *offn = GRAD ? (FWD ? a.offF : a.offB) + (size_t)n * NSUB * W + w : nullptr;
    const int edge_lane = FWD ? WAVE - 1 : 0;

    // Memory goes through buffer instructions: a per-block descriptor (SALU), a constant
    // per-lane offset and a scalar row offset -- no vector address arithmetic in the steps.
    // Score rows: three register sets rotate through "current block", "next" and "the one after"
    // (in sweep direction).  A set is loaded two phases before it is consumed and nothing reads
    // it in between, so the load latency never sits in a phase; the live loop is unrolled three
    // times to keep the rotation in register NAMES (a copy would have to wait for the load).
    float row0[BK], row1[BK], row2[BK];
    auto load_block = [&](int j, float (&dst)[BK]) {
        j = min(max(j, 0), NB - 1);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(lpn + (size_t)(j * BK) * rowstride), 0, 0x7fffffff, BUF_WORD3);
        const int last = T - 1 - j * BK;                        // rows past the end re-read the last one
#pragma unroll
        for (int i = 0; i < BK; ++i) {
            dst[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, col4, rs4 * (unsigned)min(i, last), 0));
        }
    };

    // Scores of ONE time step gathered by transition id (raw): issued a step ahead of their use.
    struct Gath {
        float ls[R], lm[R], ld[MOD ? R : 1];
    };
    auto gather = [&](float rowraw) {
        const float row = is_col ? rowraw : sent;
        Gath g;
#pragma unroll
        for (int jj = 0; jj < R; ++jj) {
            g.ls[jj] = bperm(st4[jj], row);
            g.lm[jj] = bperm(mv4[jj], row);
            if (MOD) g.ld[MOD ? jj : 0] = bperm(md4[MOD ? jj : 0], row);
        }
        return g;
    };
    const bool edge_in = lane == (FWD ? 0 : WAVE - 1);
    // One time step on the cells: forward consumes row t (column t -> t+1), backward t+1 -> t.
    // Everything that does not depend on the cells is computed first (sharpened stay / move terms;
    // the boundary cell coming in from the neighbouring chunk is folded into the boundary lane's
    // move term, whose DPP source is out of range and reads 0).  On the serial chain:
    // v_add_f32_dpp (neighbour cell + move term), sub, exp, add, log, add.
    const float emask = edge_in ? 1.f : 0.f;
    auto advance = [&](const Gath &g, float ein_i, float dmask) {
        // (ein_i + delta) on the boundary lane, 0 elsewhere: fma(ein_i, emask, delta * emask)
        const float eterm = fmaf(ein_i, emask, dmask);
        float sc[R], mc[R];
#pragma unroll
        for (int jj = 0; jj < R; ++jj) {
            sc[jj] = g.ls[jj] * c;
            float m = (jj == (FWD ? 0 : R - 1)) ? fmaf(g.lm[jj], c, eterm) : g.lm[jj] * c;
            if (MOD) m = fmaf(g.ld[MOD ? jj : 0], fw[MOD ? jj : 0], m);
            mc[jj] = m;
        }
        float nb;       // the neighbouring lane's boundary cell + this lane's boundary move term
        if constexpr (FWD)
            asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                : "=v"(nb) : "v"(x[R - 1]), "v"(mc[0]));
        else
            asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                : "=v"(nb) : "v"(x[0]), "v"(mc[R - 1]));
        if constexpr (FWD) {
#pragma unroll
            for (int jj = R - 1; jj >= 0; --jj) {
                const float bv = (jj == 0) ? nb : x[jj > 0 ? jj - 1 : 0] + mc[jj];
                x[jj] = lse2(x[jj] + sc[jj], bv);
            }
        } else {
#pragma unroll
            for (int jj = 0; jj < R; ++jj) {
                const float bv = (jj == R - 1) ? nb : x[jj < R - 1 ? jj + 1 : 0] + mc[jj];
                x[jj] = lse2(x[jj] + sc[jj], bv);
            }
        }
    };

    int stamp_k = 0;
#ifdef TK_LAB_STAMPS
#define STAMP(q)                                                                            \
    if (a.dbg && blockIdx.x == (GRAD ? a.N : 0) && w == 0 && lane == 0 && stamp_k < 64)        \
        a.dbg[stamp_k * 8 + (q)] = __builtin_amdgcn_s_memtime();
#else
#define STAMP(q)
#endif
    (void)stamp_k;

    // One live phase = one time block of this chunk.  The sweep is bound by instruction ISSUE on
    // the read's CU (all of a read's waves share one CU) and, per wave, by the dependent chain
    // dpp -> fma -> sub -> exp -> add -> log -> add of a step (~105 cycles, tools/latlab.hip).
    // Everything that does not depend on the cells is kept off that chain:
    //   * the gathers of step i+1 are issued before the arithmetic of step i;
    //   * one renormalisation per block (BNORM = BK): a six-instruction DPP maximum;
    //   * the incoming boundary cells and offsets of the whole block are read at its start;
    //   * a full block is straight-line code (no exec-mask or scalar branch between the steps).
    auto body = [&](int j, const float (&cur)[BK], float (&fill)[BK]) {
        STAMP(0);
        load_block(FWD ? j + 2 : j - 2, fill);
        const bool pl = j >= wsrc.j0 && j <= wsrc.j1;           // the neighbour ran this block one phase ago
        const int slot = j & 1;
        const int srcc = min(max(src, 0), W - 1);
        // every lane stores its candidate boundary cell: the boundary lane into the ring, the
        // others into a scratch word of their own (no exec-mask branch inside a step)
        float *Ew = (lane == edge_lane) ? E + (w * 2 + slot) * BK : Escratch + w * (WAVE + BK) + lane;
        float ein[BK];
        int osrc[BSUB], omine[BSUB];
#pragma unroll
        for (int i = 0; i < BK; ++i) ein[i] = neg;
#pragma unroll
        for (int ss = 0; ss < BSUB; ++ss) osrc[ss] = omine[ss] = 0;
        if (pl) {
            const f4 *Ein = reinterpret_cast<const f4 *>(E + (srcc * 2 + slot) * BK);
#pragma unroll
            for (int ss = 0; ss < BSUB; ++ss) osrc[ss] = Eoff[(srcc * 2 + slot) * BSUB + ss];
#pragma unroll
            for (int q4 = 0; q4 < BK / 4; ++q4) {
                const f4 e = Ein[q4];
#pragma unroll
                for (int q = 0; q < 4; ++q) ein[q4 * 4 + q] = e[q];
            }
        }
        __amdgpu_buffer_rsrc_t lat_rs = __builtin_amdgcn_make_buffer_rsrc(
            GRAD ? latn + (size_t)(j * BK) * a.LP : nullptr, 0, 0x7fffffff, BUF_WORD3);
        const int nvalid = min(BK, T - j * BK);                 // rows of this block that exist
        STAMP(1);
        float delta = 0.f;
        auto group_start = [&](int sub, int qlo, int qhi) {
            // renormalise by floor(max(this column, the boundary cells that will come in during the
            // group [qlo, qhi])): an integer, so the subtraction is exact and the offsets add exactly
            float mx = x[0];
#pragma unroll
            for (int jj = 1; jj < R; ++jj) mx = fmaxf(mx, x[jj]);
            mx = wave_max_scalar(mx);
            const int d0 = (pl ? osrc[sub] : offacc) - offacc;
            const float delta0 = (float)d0;
#pragma unroll
            for (int q = 0; q < BNORM; ++q)
                if (q >= qlo && q <= qhi) mx = fmaxf(mx, ein[sub * BNORM + q] + delta0);
            const float m = (mx > -1e29f) ? floorf(mx) : 0.f;
#pragma unroll
            for (int jj = 0; jj < R; ++jj) x[jj] -= m;
            const int mi = (int)m;
            offacc += mi;
            omine[sub] = offacc;
            delta = (float)(d0 - mi);
        };
        if (nvalid == BK) {
            Gath g = gather(cur[FWD ? 0 : BK - 1]);
#pragma unroll
            for (int ii = 0; ii < BK; ++ii) {
                const int i = FWD ? ii : BK - 1 - ii;
                if ((ii % BNORM) == 0) group_start(i / BNORM, 0, BNORM - 1);
                Gath gn = g;
                if (ii + 1 < BK) gn = gather(cur[FWD ? i + 1 : i - 1]);
                // forward: column t (before row t is consumed); backward: column t+1
                if (GRAD) band_buffer_store<R>(lat_rs, lane_cell4, lp4 * (unsigned)i, x);
                Ew[i] = FWD ? x[R - 1] : x[0];                 // the boundary lane's word is the real one
                advance(g, ein[i], delta * emask);
                g = gn;
            }
        } else {
            // the last, partial block of a T that is not a multiple of BK
            for (int ii = 0; ii < BK; ++ii) {
                const int i = FWD ? ii : BK - 1 - ii;
                if (i >= nvalid) continue;
                const int sub = i / BNORM;
                const bool first = FWD ? (i % BNORM == 0) : (i % BNORM == BNORM - 1 || i == nvalid - 1);
                if (first) {
                    const int qhi = min(BNORM - 1, nvalid - 1 - sub * BNORM);
#pragma unroll
                    for (int ss = 0; ss < BSUB; ++ss)
                        if (ss == sub) group_start(ss, 0, qhi);
                }
                float rowraw = 0.f, e = neg;
#pragma unroll
                for (int k = 0; k < BK; ++k)
                    if (k == i) {
                        rowraw = cur[k];
                        e = ein[k];
                    }
                const Gath g = gather(rowraw);
                if (GRAD) band_buffer_store<R>(lat_rs, lane_cell4, lp4 * (unsigned)i, x);
                Ew[i] = FWD ? x[R - 1] : x[0];
                advance(g, e, delta * emask);
            }
        }
        STAMP(2);
        // this block's offsets: to the ring (the neighbour reads them next phase) and to HBM
        if (lane < BSUB) {
            int o = omine[0];
#pragma unroll
            for (int ss = 1; ss < BSUB; ++ss)
                if (lane == ss) o = omine[ss];
            Eoff[(w * 2 + slot) * BSUB + lane] = o;
            if (GRAD && j * BK + lane * BNORM < T) offn[(size_t)(j * BSUB + lane) * W] = o;
        }
        STAMP(3);
        band_barrier();
        STAMP(4);
        ++stamp_k;
    };

    // phases: chunk w runs block j in phase j + w (forward) / (NB-1-j) + (W-1-w) (backward);
    // before and after its live blocks it only takes part in the barriers
    const int jfirst = FWD ? win.j0 : win.j1, nlive = win.j1 - win.j0 + 1;
    const int ph0 = FWD ? win.j0 + w : (NB - 1 - win.j1) + (W - 1 - w);
    const int dj = FWD ? 1 : -1;
    load_block(jfirst, row0);
    load_block(jfirst + dj, row1);
    for (int ph = 0; ph < ph0; ++ph) band_barrier();
    for (int k = 0; k < nlive; k += 3) {
        body(jfirst + dj * k, row0, row2);
        if (k + 1 < nlive) body(jfirst + dj * (k + 1), row1, row0);
        if (k + 2 < nlive) body(jfirst + dj * (k + 2), row2, row1);
    }
    for (int ph = ph0 + nlive; ph < NPH; ++ph) band_barrier();

    // score = sum of factors + fwd[T][L-1] (c_crf_flipflop.c:131) / bwd[0][0] (:234)
    const int pend = FWD ? L - 1 : 0;
    if (pend >= p0 && pend < p0 + R) {
        float last = 0.f;
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (p0 + j == pend) last = x[j];
        const double sc2 = (double)offacc + (double)last;
        if (GRAD) {
            (FWD ? a.scoreF : a.scoreB)[n] = sc2;
        } else {
            const float cst = (float)(-(sc2 * 0.6931471805599453) / (double)T) * a.out_scale;
            a.cost[n] = cst;
            if (a.status && !isfinite(cst)) atomicOr(a.status, 1u);
        }
    }
}

// ===========================================================================
// sweep + rank launch.  blockIdx.x in [0, N): forward sweep of read n; [N, 2N): backward
// sweep; [2N, 3N): sorted-instance records for the posterior pass.  Cost-only calls launch
// the first N workgroups.
// ===========================================================================
template <int R, bool MOD>
__global__ __launch_bounds__(BAND_MAXW *WAVE) void crf_band_sweep_kernel(BandArgs a) {
    constexpr int PW = R * WAVE;
    constexpr int KINDS = MOD ? 3 : 2;
    constexpr int EPL = KINDS * R;
    __shared__ __attribute__((aligned(16))) float E[BAND_MAXW * 2 * BK];
    __shared__ int Eoff[BAND_MAXW * 2 * BSUB];
    __shared__ float Escratch[BAND_MAXW * (WAVE + BK)];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, S = a.S, W = a.W;
    // dispatch order: rank workgroups first (short; they leave their CUs within microseconds),
    // then the forward, then the backward sweeps
    const int slot3 = blockIdx.x / N;
    const int role = want_grad_launch(a) ? (slot3 + 2) % 3 : 0;            // 0 forward, 1 backward, 2 rank
    const int n = blockIdx.x - slot3 * N;
    const int L = a.seqlen[n];
    const bool want_grad = a.grad != nullptr;
    if (L == 0 || L > W * PW) {
        // c_crf_flipflop.c:269-272: cost 0 for an empty read (the posterior pass does it when
        // there is one); too long for the launch: flagged
        if (!want_grad && tid == 0) {
            a.cost[n] = (L == 0) ? 0.f : __builtin_nanf("");
            if (L != 0 && a.status) atomicOr(a.status, 4u);
        }
        return;
    }
    const int64_t off = a.seqoff[n];
    const int p0 = w * PW + lane * R;

    if (role == 2) {
        // ---------------- sorted transition instances of chunk w -----------------------
        // instance = (kind, cell): stay at p | move p -> p+1 | (cat-mod) the mod term of that
        // move.  Sort key = transition id (the three kinds use disjoint id ranges), padding
        // last.  Ranks come from ballots in a fixed order, so the permutation -- and with it
        // every floating-point sum of the posterior pass -- is the same from run to run.
        if (w * PW >= L) return;
        int key[EPL], word[EPL];
        float wgt[MOD ? EPL : 1];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int p = p0 + j;
            const int idx = lane * R + j;
            const int st = (p < L) ? a.stay[off + p] : S;
            const int mv = (p < L - 1) ? a.move[off + p] : S;
            key[j] = (p < L) ? st : KEY_DEAD;
            word[j] = idx | (idx << 8) | (st << 17) | ((S + 1) << 23);
            key[R + j] = (p < L - 1) ? mv : KEY_DEAD;
            if (MOD) {
                const int md = (p < L - 1) ? a.mod[off + p] : S + 1;
                const float mf = (p < L - 1) ? a.modfact[off + p] : 0.f;
                word[R + j] = idx | ((idx + 1) << 8) | (mv << 17) | (md << 23) | (1 << 29);
                key[2 * R + j] = (p < L - 1) ? md : KEY_DEAD;
                word[2 * R + j] = idx | ((idx + 1) << 8) | (mv << 17) | (md << 23) | (2 << 29);
                wgt[j] = 0.f;
                wgt[R + j] = mf;
                wgt[2 * R + j] = mf;
            } else {
                word[R + j] = idx | ((idx + 1) << 8) | (mv << 17) | ((S + 1) << 23) | (1 << 29);
            }
        }
        int cnt = 0;                        // lane b: instances with key b ranked so far
        int rank[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            int r = 0;
            for (int b = 0; b < WAVE; ++b) {
                if (b == S + 2) b = KEY_DEAD;                   // keys S+2 .. 62 do not occur
                const unsigned long long mask = __ballot(key[e] == b);
                if (key[e] == b)
                    r = __builtin_amdgcn_readlane(cnt, b) + __popcll(mask & ((1ull << lane) - 1ull));
                if (lane == b) cnt += __popcll(mask);
            }
            rank[e] = r;
        }
        const int incl = wave_inclusive_scan_int(cnt);          // lane b: end of key b's segment
        const int start = incl - cnt;
        a.segend[((size_t)n * W + w) * WAVE + lane] = incl;
        uint32_t *recn = a.rec + ((size_t)n * W + w) * EPL * WAVE;
        float *recwn = MOD ? a.recw + ((size_t)n * W + w) * EPL * WAVE : nullptr;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int pos = __builtin_amdgcn_ds_bpermute(key[e] * 4, start) + rank[e];
            // position pos of the sorted order lives in lane pos / EPL, register pos % EPL
            const int slot = (pos % EPL) * WAVE + pos / EPL;
            recn[slot] = (uint32_t)word[e];
            if (MOD) recwn[slot] = wgt[MOD ? e : 0];
        }
        return;
    }

    if (!want_grad)
        band_sweep<R, MOD, true, false>(a, n, L, E, Eoff, Escratch);
    else if (role == 0)
        band_sweep<R, MOD, true, true>(a, n, L, E, Eoff, Escratch);
    else
        band_sweep<R, MOD, false, true>(a, n, L, E, Eoff, Escratch);
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

// ===========================================================================
// posterior pass, rows outside (the round-2 first form): one row at a time, looping over its live
// chunks.  Kept for R = 4 (T = 4000: the pass is HBM-bound there, 17 GB of lattice, and wants
// the occupancy of its 52 registers and a row's chunks read as one contiguous run; the
// chunks-outside form below needs 141 registers at R = 4 and measured 2.59 ms against 2.24).
// ===========================================================================
__host__ __device__ inline size_t band_post_rows_lds_bytes(int R, int W, bool mod) {
    const int EPL = (mod ? 3 : 2) * R, PW = R * WAVE;
    size_t words = (size_t)W * EPL * WAVE * (mod ? 2 : 1) + (size_t)W * WAVE +
                   (size_t)POST_WAVES * (2 * PW + 4);
    return words * 4;
}

template <int R, bool MOD>
__global__ __launch_bounds__(POST_WAVES *WAVE) void crf_band_posterior_rows_kernel(BandArgs a) {
    constexpr int PW = R * WAVE;
    constexpr int KINDS = MOD ? 3 : 2;
    constexpr int EPL = KINDS * R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: SGPR
    const int n = blockIdx.x;
    const int N = a.N, T = a.T, S = a.S, W = a.W;
    const int L = a.seqlen[n];
    const size_t rowstride = (size_t)N * S;
    const int t0 = (blockIdx.y * POST_WAVES + wave) * POST_ROWS;

    if (L == 0 || L > W * PW) {
        if (blockIdx.y == 0 && tid == 0) {
            a.cost[n] = (L == 0) ? 0.f : __builtin_nanf("");     // c_crf_flipflop.c:269-272, 458-464
            if (L != 0 && a.status) atomicOr(a.status, 4u);
        }
        if (L == 0 && lane < S)
            for (int t = t0; t < min(t0 + POST_ROWS, T); ++t)
                a.grad[(size_t)t * rowstride + (size_t)n * S + lane] = 0.f;
        return;
    }
    const int Wn = (L + PW - 1) / PW;                           // chunks this read has

    uint32_t *recL = reinterpret_cast<uint32_t *>(smem);        // [Wn][EPL][64]
    float *recwL = reinterpret_cast<float *>(recL + (size_t)W * EPL * WAVE);     // (cat-mod) same shape
    int *segq = reinterpret_cast<int *>(recwL + (MOD ? (size_t)W * EPL * WAVE : 0));   // [Wn][64]
    float *sF = reinterpret_cast<float *>(segq + (size_t)W * WAVE) + (size_t)wave * (2 * PW + 4);
    float *sB = sF + PW;                                        // PW + 1 cells

    for (int e = tid; e < Wn * EPL * WAVE; e += POST_WAVES * WAVE) {
        recL[e] = a.rec[(size_t)n * W * EPL * WAVE + e];
        if (MOD) recwL[e] = a.recw[(size_t)n * W * EPL * WAVE + e];
    }
    for (int e = tid; e < Wn * WAVE; e += POST_WAVES * WAVE) {
        // where the inclusive prefix at the END of key `lane`'s segment lives: lane q, register r
        const int idx = a.segend[(size_t)n * W * WAVE + e] - 1;
        segq[e] = idx < 0 ? -1 : (((idx / EPL) * 4) | ((idx % EPL) << 16));
    }
    __syncthreads();
    if (t0 >= T) return;

    const double scoreF = a.scoreF[n];
    if (blockIdx.y == 0 && tid == 0) {
        // score = mean of the two sweeps (c_crf_flipflop.c:482-491), cost = -score / T
        const double score2 = 0.5 * (scoreF + a.scoreB[n]);
        const float cst = (float)(-(score2 * 0.6931471805599453) / (double)T) * a.out_scale;
        a.cost[n] = cst;
        if (a.status && !isfinite(cst)) atomicOr(a.status, 1u);
    }
    const float *lpn = a.lp + (size_t)n * S;
    const int col = min(lane, S - 1);
    const bool is_col = lane < S;
    const float sent = (lane == S) ? NEG_LARGE : 0.f;
    const float neg = NEG_LARGE * LOG2E;
    const float c = a.c_can;
    const bool trim = L <= T + 1;
    const int NSUB = (T + BNORM - 1) / BNORM;
    const float *Fn = a.latF + (size_t)n * T * a.LP + lane * R;
    const float *Bn = a.latB + (size_t)n * T * a.LP + lane * R;
    const int *oFn = a.offF + (size_t)n * NSUB * W;
    const int *oBn = a.offB + (size_t)n * NSUB * W;
    bool bad = false;

    for (int t = t0; t < min(t0 + POST_ROWS, T); ++t) {
        const float row = band_row(lpn, rowstride, t, col, sent, is_col);
        // log2 offsets of this row's chunks: lane c holds chunk c's
        const int cl = min(lane, W - 1);
        const int oF = oFn[(size_t)(t / BNORM) * W + cl], oB = oBn[(size_t)(t / BNORM) * W + cl];
        int c_lo = 0, c_hi = Wn - 1;
        if (trim) {
            // chunk [a, b] holds a cell of some complete path through row t:
            // a <= t  and  b + 1 >= L - T + t
            c_hi = min(c_hi, t / PW);
            const int need = L - T + t;
            if (need > 0) c_lo = max(0, (need + PW - 1) / PW - 1);
        }
        float colacc = 0.f, total = 0.f;
        for (int ck = c_lo; ck <= c_hi; ++ck) {
            const int apos = ck * PW;
            const int oFc = __builtin_amdgcn_readlane(oF, ck), oBc = __builtin_amdgcn_readlane(oB, ck);
            const float ct = (float)(scoreF - (double)(oFc + oBc));
            float fv[R], bv[R];
            if constexpr (R == 4) {
                const f4 f = *reinterpret_cast<const f4 *>(Fn + (size_t)t * a.LP + apos);
                const f4 b = *reinterpret_cast<const f4 *>(Bn + (size_t)t * a.LP + apos);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    fv[j] = f[j];
                    bv[j] = b[j];
                }
            } else if constexpr (R == 2) {
                const f2 f = *reinterpret_cast<const f2 *>(Fn + (size_t)t * a.LP + apos);
                const f2 b = *reinterpret_cast<const f2 *>(Bn + (size_t)t * a.LP + apos);
                fv[0] = f[0];
                fv[1] = f[1];
                bv[0] = b[0];
                bv[1] = b[1];
            } else {
                fv[0] = Fn[(size_t)t * a.LP + apos];
                bv[0] = Bn[(size_t)t * a.LP + apos];
            }
            // first cell of the next chunk (the target of this chunk's last move); it was stored
            // iff that chunk had started by row t
            const bool right_live = ck + 1 < Wn && (!trim || t >= apos + PW - 1);
            float bright = neg;
            if (right_live) {
                const int oBr = __builtin_amdgcn_readlane(oB, min(ck + 1, W - 1));
                bright = a.latB[((size_t)n * T + t) * a.LP + apos + PW] + (float)(oBr - oBc);
            }
            if constexpr (R == 4) {
                *reinterpret_cast<f4 *>(sF + lane * 4) = f4{fv[0] - ct, fv[1] - ct, fv[2] - ct, fv[3] - ct};
                *reinterpret_cast<f4 *>(sB + lane * 4) = f4{bv[0], bv[1], bv[2], bv[3]};
            } else if constexpr (R == 2) {
                *reinterpret_cast<f2 *>(sF + lane * 2) = f2{fv[0] - ct, fv[1] - ct};
                *reinterpret_cast<f2 *>(sB + lane * 2) = f2{bv[0], bv[1]};
            } else {
                sF[lane] = fv[0] - ct;
                sB[lane] = bv[0];
            }
            if (lane == 0) sB[PW] = bright;
            wave_lds_fence();
            // ---- the chunk's instances in sorted order: lane l holds sorted positions
            //      l*EPL .. l*EPL + EPL-1; running (inclusive) prefix in v[]
            float v[EPL];
#pragma unroll
            for (int r = 0; r < EPL; ++r) {
                const uint32_t word = recL[((size_t)ck * EPL + r) * WAVE + lane];
                const float vF = sF[word & 0xffu];
                const float vB = sB[(word >> 8) & 0x1ffu];
                float xx = fmaf(bperm((int)((word >> 17) & 63u) * 4, row), c, vF + vB);
                float scale = 1.f;
                if (MOD) {
                    const float mf = recwL[((size_t)ck * EPL + r) * WAVE + lane];
                    xx = fmaf(bperm((int)((word >> 23) & 63u) * 4, row), mf * a.c_mod, xx);
                    scale = ((word >> 29) == 2u) ? mf : 1.f;    // d/d(mod score) = posterior * modfact
                }
                const float pr = fast_exp2(xx) * scale;
                v[r] = (r == 0) ? pr : v[r > 0 ? r - 1 : 0] + pr;
            }
            const float incl = wave_inclusive_scan_dpp(v[EPL - 1]);
            const float base = incl - v[EPL - 1];
            // prefix at the end of key `lane`'s segment
            const int sq = segq[ck * WAVE + lane];
            float P = 0.f;
#pragma unroll
            for (int r = 0; r < EPL; ++r) {
                const float cand = bperm(sq & 0xffff, v[r] + base);
                if ((sq >> 16) == r) P = cand;
            }
            if (sq < 0) P = 0.f;
            const float prev = wave_shift_up1(P, 0.f);
            colacc += P - prev;
            // row normaliser: all stay and move instances (the reference's softmax over the
            // 2L-1 transitions, c_crf_flipflop.c:400-401); mod ids sort after them
            total += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(P), a.ncan - 1));
            wave_lds_fence();
        }
        // gradient of -score / T  (ctc.pyx:113)
        const float g = colacc * (-1.0f / (total * (float)T));
        if (lane < S) {
            bad |= !isfinite(g);
            a.grad[(size_t)t * rowstride + (size_t)n * S + lane] = g;
        }
    }
    if (a.status && bad) atomicOr(a.status, 2u);
}

// ===========================================================================
// posterior pass: grid (N, ceil(T / (POST_WAVES * POST_ROWS))), wave = POST_ROWS rows.
// Chunks outside, the wave's rows inside: what belongs to a chunk -- its sorted instance
// records as LDS / bpermute addresses, the end-of-segment look-ups, its log2 offsets (one
// sub-block of BNORM = POST_ROWS rows) -- is set up once per chunk and serves all the rows,
// and the lattice cells of all the rows of a chunk are requested before the first is used.
// The pass is bound by VALU throughput at the train step's shape (four SIMD cycles per
// instruction; ~47 instructions per (row, chunk) at R = 1, 81 with the rows outside) and by
// HBM at T = 4000 (17 GB of lattice).
// ===========================================================================
static_assert(POST_ROWS == BNORM, "a wave's rows share one sub-block of log2 offsets");
__host__ __device__ inline size_t band_post_lds_bytes(int R, int W, bool mod) {
    (void)W;
    (void)mod;
    return (size_t)POST_WAVES * (2 * R * WAVE + WAVE) * 4;
}

template <int R, bool MOD>
__global__ __launch_bounds__(POST_WAVES *WAVE) void crf_band_posterior_kernel(BandArgs a) {
    constexpr int PW = R * WAVE;
    constexpr int KINDS = MOD ? 3 : 2;
    constexpr int EPL = KINDS * R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: SGPR
    const int n = blockIdx.x;
    const int N = a.N, T = a.T, S = a.S, W = a.W;
    const int L = a.seqlen[n];
    const size_t rowstride = (size_t)N * S;
    const int t0 = (blockIdx.y * POST_WAVES + wave) * POST_ROWS;

    if (L == 0 || L > W * PW) {
        if (blockIdx.y == 0 && tid == 0) {
            a.cost[n] = (L == 0) ? 0.f : __builtin_nanf("");     // c_crf_flipflop.c:269-272, 458-464
            if (L != 0 && a.status) atomicOr(a.status, 4u);
        }
        if (L == 0 && lane < S)
            for (int t = t0; t < min(t0 + POST_ROWS, T); ++t)
                a.grad[(size_t)t * rowstride + (size_t)n * S + lane] = 0.f;
        return;
    }
    const int Wn = (L + PW - 1) / PW;                           // chunks this read has
    float *sF = reinterpret_cast<float *>(smem) + (size_t)wave * (2 * PW + WAVE);
    float *sB = sF + PW;                                        // PW cells + the next chunk's first (x 64)
    if (t0 >= T) return;

    const double scoreF = a.scoreF[n];
    if (blockIdx.y == 0 && tid == 0) {
        // score = mean of the two sweeps (c_crf_flipflop.c:482-491), cost = -score / T
        const double score2 = 0.5 * (scoreF + a.scoreB[n]);
        const float cst = (float)(-(score2 * 0.6931471805599453) / (double)T) * a.out_scale;
        a.cost[n] = cst;
        if (a.status && !isfinite(cst)) atomicOr(a.status, 1u);
    }
    const float *lpn = a.lp + (size_t)n * S;
    const int col = min(lane, S - 1);
    const bool is_col = lane < S;
    const float sent = (lane == S) ? NEG_LARGE : 0.f;
    const float neg = NEG_LARGE * LOG2E;
    const float c = a.c_can;
    const bool trim = L <= T + 1;
    const int NSUB = (T + BNORM - 1) / BNORM;
    const int nrows = min(POST_ROWS, T - t0);

    // the wave's score rows, one register each (lane = transition id)
    float row[POST_ROWS];
#pragma unroll
    for (int k = 0; k < POST_ROWS; ++k) row[k] = band_row(lpn, rowstride, min(t0 + k, T - 1), col, sent, is_col);
    // log2 offsets of the chunks for these rows (one sub-block): lane c holds chunk c's
    const int cl = min(lane, W - 1);
    const int oF = a.offF[((size_t)n * NSUB + t0 / BNORM) * W + cl];
    const int oB = a.offB[((size_t)n * NSUB + t0 / BNORM) * W + cl];
    const float ctv = (float)(scoreF - (double)(oF + oB));

    // live chunks of row t: chunk [a, b] holds a cell of some complete path through row t iff
    // a <= t  and  b + 1 >= L - T + t
    auto chunk_lo = [&](int t) {
        const int need = L - T + t;
        return (trim && need > 0) ? max(0, (need + PW - 1) / PW - 1) : 0;
    };
    auto chunk_hi = [&](int t) { return trim ? min(Wn - 1, t / PW) : Wn - 1; };
    const int cmin = chunk_lo(t0), cmax = chunk_hi(t0 + nrows - 1);       // both bounds grow with t
    // lane k < nrows: the live chunk range of row t0 + k (one ballot per chunk gives the rows' mask)
    const int tl = t0 + min(lane, POST_ROWS - 1);
    const int lo_l = chunk_lo(tl), hi_l = (lane < nrows) ? chunk_hi(tl) : -1;

    float colacc[POST_ROWS], total[POST_ROWS];
#pragma unroll
    for (int k = 0; k < POST_ROWS; ++k) colacc[k] = total[k] = 0.f;
    const float *Fn = a.latF + (size_t)n * T * a.LP + lane * R;
    const float *Bn = a.latB + (size_t)n * T * a.LP + lane * R;
    const float *Bfirst = a.latB + (size_t)n * T * a.LP;
    const uint32_t *recn = a.rec + (size_t)n * W * EPL * WAVE + lane;
    const float *recwn = MOD ? a.recw + (size_t)n * W * EPL * WAVE + lane : nullptr;

    for (int ck = cmin; ck <= cmax; ++ck) {
        const int apos = ck * PW;
        // ---- per chunk: instance records -> LDS / bpermute addresses, end-of-segment look-up
        const float *pF[EPL], *pB[EPL];
        int aS[EPL], aM[MOD ? EPL : 1];
        float mfw[MOD ? EPL : 1], scl[MOD ? EPL : 1];
#pragma unroll
        for (int r = 0; r < EPL; ++r) {
            const uint32_t word = recn[((size_t)ck * EPL + r) * WAVE];
            pF[r] = sF + (word & 0xffu);
            pB[r] = sB + ((word >> 8) & 0x1ffu);
            aS[r] = (int)((word >> 17) & 63u) * 4;
            if (MOD) {
                const float mf = recwn[((size_t)ck * EPL + r) * WAVE];
                aM[MOD ? r : 0] = (int)((word >> 23) & 63u) * 4;
                mfw[MOD ? r : 0] = mf * a.c_mod;
                scl[MOD ? r : 0] = ((word >> 29) == 2u) ? mf : 1.f;      // d/d(mod score) = posterior * modfact
            }
        }
        // where the inclusive prefix at the END of key `lane`'s segment lives: lane q, register r
        const int sidx = a.segend[((size_t)n * W + ck) * WAVE + lane] - 1;
        const int sq_addr = (sidx / EPL) * 4, sq_reg = sidx % EPL;
        const float ct = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ctv), ck));
        const int oBc = __builtin_amdgcn_readlane(oB, ck);
        const int oBr = __builtin_amdgcn_readlane(oB, min(ck + 1, W - 1));
        const float dright = (float)(oBr - oBc);
        const int right_col = min(apos + PW, (int)a.LP - 1);
        const unsigned long long live = __ballot(ck >= lo_l && ck <= hi_l);
        // the next chunk's first cell (the target of this chunk's last move) was stored iff that
        // chunk had started by row t: rows t >= apos + PW - 1 when the band is trimmed
        const bool has_right = ck + 1 < Wn;
        const int right_from = trim ? apos + PW - 1 : 0;

        // ---- the cells of every row of this chunk, requested before the first is used.  A row that
        //      is not live in this chunk asks for the first live row's cells again (a scalar select
        //      of the address: no branch in the load stream, no extra HBM traffic) and drops them
        const int kfirst = __builtin_ctzll(live | (1ull << (POST_ROWS - 1)));
        float fv[POST_ROWS][R], bv[POST_ROWS][R], br[POST_ROWS];
#pragma unroll
        for (int k = 0; k < POST_ROWS; ++k) {
            const size_t trow = (size_t)min(t0 + (((live >> k) & 1ull) ? k : kfirst), T - 1) * a.LP;
            if constexpr (R == 4) {
                const f4 f = *reinterpret_cast<const f4 *>(Fn + trow + apos);
                const f4 b = *reinterpret_cast<const f4 *>(Bn + trow + apos);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    fv[k][j] = f[j];
                    bv[k][j] = b[j];
                }
            } else if constexpr (R == 2) {
                const f2 f = *reinterpret_cast<const f2 *>(Fn + trow + apos);
                const f2 b = *reinterpret_cast<const f2 *>(Bn + trow + apos);
                fv[k][0] = f[0];
                fv[k][1] = f[1];
                bv[k][0] = b[0];
                bv[k][1] = b[1];
            } else {
                fv[k][0] = Fn[trow + apos];
                bv[k][0] = Bn[trow + apos];
            }
            br[k] = Bfirst[trow + right_col];
        }
#pragma unroll
        for (int k = 0; k < POST_ROWS; ++k) {
            if (!((live >> k) & 1ull)) continue;                // wave-uniform
            const float bright = (has_right && t0 + k >= right_from) ? br[k] + dright : neg;
            if constexpr (R == 4) {
                *reinterpret_cast<f4 *>(sF + lane * 4) =
                    f4{fv[k][0] - ct, fv[k][1] - ct, fv[k][2] - ct, fv[k][3] - ct};
                *reinterpret_cast<f4 *>(sB + lane * 4) = f4{bv[k][0], bv[k][1], bv[k][2], bv[k][3]};
            } else if constexpr (R == 2) {
                *reinterpret_cast<f2 *>(sF + lane * 2) = f2{fv[k][0] - ct, fv[k][1] - ct};
                *reinterpret_cast<f2 *>(sB + lane * 2) = f2{bv[k][0], bv[k][1]};
            } else {
                sF[lane] = fv[k][0] - ct;
                sB[lane] = bv[k][0];
            }
            sB[PW + lane] = bright;                             // every lane: no exec-mask detour; word PW is read
            wave_lds_fence();
            // ---- the chunk's instances in sorted order: lane l holds sorted positions
            //      l*EPL .. l*EPL + EPL-1; running (inclusive) prefix in v[]
            float v[EPL];
#pragma unroll
            for (int r = 0; r < EPL; ++r) {
                float xx = fmaf(bperm(aS[r], row[k]), c, *pF[r] + *pB[r]);
                if (MOD) xx = fmaf(bperm(aM[MOD ? r : 0], row[k]), mfw[MOD ? r : 0], xx);
                float pr = fast_exp2(xx);
                if (MOD) pr *= scl[MOD ? r : 0];
                v[r] = (r == 0) ? pr : v[r > 0 ? r - 1 : 0] + pr;
            }
            const float incl = wave_scan_fused(v[EPL - 1]);
            const float base = incl - v[EPL - 1];
            float P = 0.f;
#pragma unroll
            for (int r = 0; r < EPL; ++r) {
                const float cand = bperm(sq_addr, v[r] + base);
                if (sq_reg == r) P = cand;
            }
            if (sidx < 0) P = 0.f;
            const float prev = wave_shift_up1(P, 0.f);
            colacc[k] += P - prev;
            // row normaliser: all stay and move instances (the reference's softmax over the
            // 2L-1 transitions, c_crf_flipflop.c:400-401); mod ids sort after them
            total[k] += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(P), a.ncan - 1));
            wave_lds_fence();
        }
    }
    bool bad = false;
#pragma unroll
    for (int k = 0; k < POST_ROWS; ++k) {
        if (k < nrows) {
            // gradient of -score / T  (ctc.pyx:113)
            const float g = colacc[k] * (-1.0f / (total[k] * (float)T));
            if (lane < S) {
                bad |= !isfinite(g);
                a.grad[(size_t)(t0 + k) * rowstride + (size_t)n * S + lane] = g;
            }
        }
    }
    if (a.status && bad) atomicOr(a.status, 2u);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// Cells per lane: the smallest R whose chunks fit the 16 waves of a workgroup.  The sweep is
// bound by instruction issue (one VALU instruction per ~5 cycles and wave, tools/latlab.hip), a
// read's live chunks sit on one CU, and the band keeps about half of them live at a time: small
// chunks spread those over the CU's four SIMDs (cfg 2: R=1 108 us, R=2 115 us, R=4 158 us).
int crf_band_pick_R(size_t max_seqlen) {
    int R = 1;
    if (const char *e = getenv("TK_CRF_BAND_R")) {
        R = atoi(e);
        if (R != 1 && R != 2 && R != 4) R = 1;
    }
    while (R < 4 && (size_t)R * WAVE * BAND_MAXW < max_seqlen) R *= 2;
    return R;
}

bool crf_band_fits(size_t max_seqlen) { return max_seqlen <= (size_t)4 * WAVE * BAND_MAXW; }

BandLayout crf_band_layout(size_t ntrans, size_t nblk, size_t nbatch, size_t max_seqlen, bool mod) {
    (void)ntrans;
    BandLayout l;
    l.R = crf_band_pick_R(max_seqlen);
    const size_t PW = (size_t)l.R * WAVE;
    l.W = (int)((max_seqlen + PW - 1) / PW);
    if (l.W < 1) l.W = 1;
    l.LP = (size_t)l.W * PW;
    const size_t EPL = (mod ? 3 : 2) * (size_t)l.R, NSUB = (nblk + BNORM - 1) / BNORM;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t r = off;
        off += (bytes + 255) / 256 * 256;
        return r;
    };
    l.latF = take(nbatch * nblk * l.LP * sizeof(float));
    l.latB = take(nbatch * nblk * l.LP * sizeof(float));
    l.offF = take(nbatch * NSUB * l.W * sizeof(int));
    l.offB = take(nbatch * NSUB * l.W * sizeof(int));
    l.scoreF = take(nbatch * sizeof(double));
    l.scoreB = take(nbatch * sizeof(double));
    l.rec = take(nbatch * l.W * EPL * WAVE * sizeof(uint32_t));
    l.recw = take(mod ? nbatch * l.W * EPL * WAVE * sizeof(float) : 0);
    l.segend = take(nbatch * l.W * WAVE * sizeof(int));
    l.total = off + 256;
    return l;
}

// lab knob (tools/overlap_probe.py): 0 both passes, 1 the sweep launch only, 2 the posterior pass only
static int g_band_lab_phase = 0;
void crf_band_lab_phase(int phase) { g_band_lab_phase = phase; }

template <int R, bool MOD>
static int band_launch(const BandArgs &a, hipStream_t stream) {
    const bool want_grad = a.grad != nullptr;
    if (g_band_lab_phase != 2)
        hipLaunchKernelGGL((crf_band_sweep_kernel<R, MOD>), dim3((want_grad ? 3 : 1) * a.N), dim3(a.W * WAVE), 0,
                           stream, a);
    if (hipGetLastError() != hipSuccess) return 4;
    if (!want_grad || g_band_lab_phase == 1) return 0;
    const int rows = POST_WAVES * POST_ROWS;
    if constexpr (R == 4) {
        const size_t lds = band_post_rows_lds_bytes(R, a.W, MOD);
        if (lds > 160 * 1024) return 2;
        if (lds > 64 * 1024 &&
            raise_dynamic_lds(reinterpret_cast<const void *>(&crf_band_posterior_rows_kernel<R, MOD>)))
            return 4;
        hipLaunchKernelGGL((crf_band_posterior_rows_kernel<R, MOD>), dim3(a.N, (a.T + rows - 1) / rows),
                           dim3(POST_WAVES * WAVE), lds, stream, a);
    } else {
        hipLaunchKernelGGL((crf_band_posterior_kernel<R, MOD>), dim3(a.N, (a.T + rows - 1) / rows),
                           dim3(POST_WAVES * WAVE), band_post_lds_bytes(R, a.W, MOD), stream, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

int crf_band_dispatch(const BandArgs &a0, int R, bool mod, hipStream_t stream) {
    BandArgs a = a0;
#ifdef TK_LAB_STAMPS
    static unsigned long long *dbg = nullptr;
    if (getenv("TK_CRF_STAMPS")) {
        if (!dbg) (void)hipMalloc(&dbg, 64 * 8 * 8);
        (void)hipMemsetAsync(dbg, 0, 64 * 8 * 8, stream);
        a.dbg = dbg;
    }
    struct Printer {
        unsigned long long *d;
        hipStream_t s;
        ~Printer() {
            if (!d) return;
            static unsigned long long h[64 * 8];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            for (int k = 2; k < 12; ++k)
                fprintf(stderr, "phase %2d: loads+ring %5llu  steps %5llu  offsets %5llu  barrier %5llu  next-phase-gap %5llu\n", k,
                        h[k * 8 + 1] - h[k * 8 + 0], h[k * 8 + 2] - h[k * 8 + 1], h[k * 8 + 3] - h[k * 8 + 2],
                        h[k * 8 + 4] - h[k * 8 + 3], h[(k + 1) * 8 + 0] - h[k * 8 + 4]);
        }
    } printer{a.dbg, stream};
#endif
    if (a.W < 1 || a.W > BAND_MAXW) return 2;
    switch (R * 2 + (mod ? 1 : 0)) {
        case 2: return band_launch<1, false>(a, stream);
        case 3: return band_launch<1, true>(a, stream);
        case 4: return band_launch<2, false>(a, stream);
        case 5: return band_launch<2, true>(a, stream);
        case 8: return band_launch<4, false>(a, stream);
        case 9: return band_launch<4, true>(a, stream);
        default: return 2;
    }
}

}  // namespace tk
