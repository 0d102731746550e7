// crf_log.h -- kernel A's LOG-DOMAIN form as device code: one read, all three passes, in one workgroup (crf_read).
// Used by crf_kernel (crf_kernels.hip: batches the linear path does not take, TK_CRF_MODE=ckpt) and by the tail
// launch of the linear path (crf_band.hip: crf_band_tail_kernel), whose workgroups redo in the log domain what the
// linear path disowned twice.  Replaces taiyaki/ctc/c_crf_flipflop.c:43-516 and c_cat_mod_flipflop.c:37-582.
#pragma once
#include "crf_band.h"
#include "ff_common.h"

namespace tk {

struct CrfArgs {
    const float *lp;            // (T, N, S) scores (unsharpened)
    int T, N, S;
    int ncan;                   // canonical transition columns (== S for the plain CRF)
    const int32_t *stay;        // padded per-position layout, see header
    const int32_t *move;
    const int32_t *mod;         // nullable
    const float *modfact;       // nullable
    const int32_t *seqlen;      // (N)
    const int64_t *seqoff;      // (N + 1)
    float c_can;                // sharp_can * log2(e)
    float c_mod;                // sharp_mod * log2(e)
    float out_scale;            // cost multiplier (1 / sharpfact)
    float *cost;                // (N)
    float *grad;                // (T, N, S) or null
    float *ckpt;                // workspace: checkpoint columns
    double *ckoff;              // workspace: checkpoint offsets
    uint32_t *status;
    // behind a band launch that built its indices from the caller's labels (crf_band.h: BandArgs::codes): no index
    // array was written -- the reads redone here form their ids from the codes, too (null: the arrays above are inputs)
    const int32_t *codes, *mod_cats, *cmo;
    const float *mcw;
    int nbase;
    float grad_scale;           // gradient multiplier (1 for the reference's operators)
    const float *grad_scale_vec;    // nullable; (N): a further per-read multiplier    // fused cat-mod loss: kernel B ran first into a compact buffer; this operator adds
    // add_scale * add_cost[n] to the cost and add_scale * (gradient multiplier) * add_grad[t][n][s]
    // (s < add_S) to the gradient it writes.  Null: nothing to add.
    const float *add_grad;      // (T, N, add_S)
    const float *add_cost;      // (N)
    int add_S;
    float add_scale;
};

__host__ __device__ inline int crf_ck(int R, int W, int kinds) {
    // recompute tile + sorted-posterior tile <= 112 KiB of LDS
    const int c = 28672 / (R * W * WAVE * (kinds + 1));
    return c >= 16 ? 16 : (c >= 8 ? 8 : (c >= 4 ? 4 : 2));
}

template <int R, int W, bool MOD>
struct CrfCfg {
    static constexpr int NT = W * WAVE;                 // threads per read
    static constexpr int LPAD = R * NT;                 // lattice positions covered
    static constexpr int KINDS = MOD ? 3 : 2;           // stay, move(, mod) posterior streams
    static constexpr int CK0 = 28672 / (LPAD * (KINDS + 1));
    static constexpr int CK = CK0 >= 16 ? 16 : (CK0 >= 8 ? 8 : (CK0 >= 4 ? 4 : 2));
    static constexpr int MAXK = (CK + W - 1) / W;       // tile rows moved per wave (S <= 64)
    static constexpr int EPL = LPAD / WAVE;             // sorted elements per lane in the flush
};

// LDS carve (floats): tile[CK][SP] | Psort[CK][KINDS][LPAD] | Fblk[CK][R][NT] | offs (2*CK) |
// segstart[KINDS][SP+1] | lanebase[W][64] | edgeF[2][W] | edgeB[2][W] | red[W] | misc[8]
// (the ranking scratch wcnt[KINDS][W][SP] overlays Fblk during set-up)
__host__ __device__ inline size_t crf_lds_bytes(int R, int W, int S, int kinds) {
    const int SP = S + 2, CK = crf_ck(R, W, kinds), NT = W * WAVE;
    size_t f = 0;
    f += (size_t)CK * SP;
    f += (size_t)CK * kinds * R * NT;
    size_t fb = (size_t)CK * R * NT, scratch = (size_t)kinds * W * SP + kinds * SP;
    f += fb > scratch ? fb : scratch;
    f += (size_t)2 * CK + 2;
    f += (size_t)kinds * (SP + 1);
    f += (size_t)W * WAVE;
    f += (size_t)9 * W + 10;            // edgeF, edgeB: [2][W] doubles each; red [W]; misc; alignment
    return (f * 4 + 15) / 16 * 16;
}

// The lattice state of this kernel is kept in DOUBLE (round 5).  In fp32 -- the reference's own arithmetic -- a cell
// carries one rounding of its magnitude (tens to hundreds of bits below the column maximum) per step, and over
// T in the thousands with raw cat-mod logits x 8 the posteriors sit 5e-3 .. 1e-2 from a float64 evaluation: the
// reference's level, but two fp32 algorithms' noise is two different samples, and round 4's fuzz sweep drew one
// at twice the reference's (the criterion was widened for it).  With the cells in double the only fp32 left on
// the chain is the correction term log2(1 + 2^-|d|) in [0, 1] (absolute error ~1e-7 per step): 1e-5 .. 1e-4 from
// float64 on the same cases.  fp64 adds run at the fp32 rate on this chip; the kernel redoes disowned reads and
// serves the fallback modes, it is not the fast path.
__device__ __forceinline__ double lse2d(double a, double b) {
    const double mx = fmax(a, b);
    const float d = (float)(fmin(a, b) - mx);           // <= 0 (the two are finite: "nothing" is -1.44e30)
    return mx + (double)fast_log2(1.0f + fast_exp2(d));
}
__device__ __forceinline__ double wave_shift_up1(double src, double fill) {
    return __hiloint2double(wave_shift_up1(__double2hiint(src), __double2hiint(fill)),
                            wave_shift_up1(__double2loint(src), __double2loint(fill)));
}
__device__ __forceinline__ double wave_shift_down1(double src, double fill) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(src), 0x130, 0xF, 0xF, false),
                            __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(src), 0x130, 0xF, 0xF, false));
}

// One read, all three passes.  `ckslot`: which set of checkpoint columns of the workspace this workgroup uses.
template <int R, int W, bool MOD>
__device__ __forceinline__ void crf_read(const CrfArgs &a, const int n, const int ckslot) {
    using Cfg = CrfCfg<R, W, MOD>;
    constexpr int CK = Cfg::CK, NT = Cfg::NT, MAXK = Cfg::MAXK, LPAD = Cfg::LPAD;
    constexpr int KINDS = Cfg::KINDS, EPL = Cfg::EPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & (WAVE - 1);
    const int T = a.T, N = a.N, S = a.S, SP = S + 2;
    const int L = min(a.seqlen[n], (int)(a.seqoff[n + 1] - a.seqoff[n]));      // (offsets are clamped to the label array)
    const bool want_grad = a.grad != nullptr;
    const float gsc = a.grad_scale * (a.grad_scale_vec != nullptr ? a.grad_scale_vec[n] : 1.0f);

    float *tile = reinterpret_cast<float *>(smem);              // [CK][SP]
    float *Psort = tile + CK * SP;                              // [CK][KINDS][LPAD]
    float *Fblk = Psort + (size_t)CK * KINDS * LPAD;            // [CK][R][NT]
    size_t fbsz = (size_t)CK * R * NT;
    {
        const size_t scratch = (size_t)KINDS * W * SP + KINDS * SP;
        if (scratch > fbsz) fbsz = scratch;
    }
    float *after = Fblk + fbsz;
    after += ((after - tile) & 1);                              // 8-byte alignment for the doubles
    double *offs = reinterpret_cast<double *>(after);           // [CK]
    int *segstart = reinterpret_cast<int *>(offs + CK);         // [KINDS][SP + 1]
    float *lanebase = reinterpret_cast<float *>(segstart + KINDS * (SP + 1));   // [W][64]
    float *after2 = lanebase + W * WAVE;
    after2 += ((after2 - tile) & 1);                            // 8-byte alignment for the doubles
    double *edgeF = reinterpret_cast<double *>(after2);         // [2][W]
    double *edgeB = edgeF + 2 * W;                              // [2][W]
    float *red = reinterpret_cast<float *>(edgeB + 2 * W);      // [W]
    double *misc = reinterpret_cast<double *>(red + W + (W & 1));   // [2]

    const size_t rowstride = (size_t)N * S;
    const float *lpn = a.lp + (size_t)n * S;

    if (L == 0) {
        // c_crf_flipflop.c:269-272 / 458-464: cost 0, zero gradient rows
        if (tid == 0) a.cost[n] = crf_add_cost(a, n, 0.f);
        if (want_grad && lane < S) {
            for (int t = wave; t < T; t += W)
                a.grad[(size_t)t * rowstride + (size_t)n * S + lane] = crf_add_grad(a, (size_t)t, n, lane, 0.f, gsc);
        }
        return;
    }
    if (L > R * NT) {
        if (tid == 0) {
            a.cost[n] = __builtin_nanf("");
            if (a.status) atomicOr(a.status, 16u);
        }
        if (want_grad && lane < S) {        // NaN rows, not uninitialised memory (see crf_band_posterior_kernel)
            for (int t = wave; t < T; t += W)
                a.grad[(size_t)t * rowstride + (size_t)n * S + lane] = __builtin_nanf("");
        }
        return;
    }

    // ---- tile movers: rows t0 .. t0+nrows-1 of this read <-> LDS.  Wave w moves
    //      rows w, w+W, ...; lane = column (S <= 64): no index arithmetic, and the
    //      loads are unconditional (indices clamped) so they pipeline freely. -----------
    auto tile_fetch = [&](int t0, float (&pre)[MAXK]) {
        const int nrows = min(CK, T - t0);
        const int col = min(lane, S - 1);
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            const int row = min(wave + W * k, nrows - 1);
            pre[k] = lpn[(size_t)(t0 + row) * rowstride + col];
        }
    };
    auto tile_commit = [&](int t0, const float (&pre)[MAXK]) {
        const int nrows = min(CK, T - t0);
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            const int row = wave + W * k;
            if (row < nrows && lane < S) tile[row * SP + lane] = pre[k];
        }
    };

    // ---- per-position transition ids -> registers ---------------------------
    const int64_t off = a.seqoff[n];
    const int p0 = tid * R;
    int st[R], mv[R], md[MOD ? R : 1];
    float fw[MOD ? R : 1];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int p = p0 + j;
        if (a.codes != nullptr) {
            const int cp = (p < L) ? lbl_code(a, off + p) : 0, cn = (p < L - 1) ? lbl_code(a, off + p + 1) : 0;
            st[j] = (p < L) ? lbl_stay(a, cp) : S;
            mv[j] = (p < L - 1) ? lbl_move(a, cp, cn) : S;
            if (MOD) {
                const int mq = (p < L - 1) ? lbl_mod_seq(a, cn, a.mod_cats[off + p + 1], nullptr) : 0;
                md[j] = (p < L - 1) ? a.ncan + mq : S + 1;
                fw[j] = (p < L - 1) ? a.mcw[mq] * a.c_mod : 0.f;
            }
        } else {
            st[j] = (p < L) ? a.stay[off + p] : S;            // S   = -LARGE sentinel slot
            mv[j] = (p < L - 1) ? a.move[off + p] : S;
            if (MOD) {
                md[j] = (p < L - 1) ? a.mod[off + p] : S + 1; // S+1 = 0.0 sentinel slot
                fw[j] = (p < L - 1) ? a.modfact[off + p] * a.c_mod : 0.f;
            }
        }
    }
    // transition INTO this thread's first position (from position p0 - 1)
    const bool has_in = (p0 >= 1) && (p0 - 1 < L - 1);
    int mvin0 = S, mdin0 = S + 1;
    float fwin0 = 0.f;
    if (has_in && a.codes != nullptr) {
        const int cb = lbl_code(a, off + p0 - 1), cp = lbl_code(a, off + p0);
        mvin0 = lbl_move(a, cb, cp);
        if (MOD) {
            const int mq = lbl_mod_seq(a, cp, a.mod_cats[off + p0], nullptr);
            mdin0 = a.ncan + mq;
            fwin0 = a.mcw[mq] * a.c_mod;
        }
    } else if (has_in) {
        mvin0 = a.move[off + p0 - 1];
        if (MOD) {
            mdin0 = a.mod[off + p0 - 1];
            fwin0 = a.modfact[off + p0 - 1] * a.c_mod;
        }
    }
    // sentinel slots of every LDS row (tile loads never touch them)
    for (int r = tid; r < CK; r += NT) {
        tile[r * SP + S] = NEG_LARGE;
        tile[r * SP + S + 1] = 0.f;
    }
    const float c = a.c_can;
    const double neg = (double)(NEG_LARGE * LOG2E);

    // ---- sorted slots for the posterior streams (gradient path only) -------------------
    // Every (position, kind) gets a slot such that slots with the same transition id are
    // contiguous: key-major, then (wave, j, lane).  Ranks come from ballots, so the layout
    // (and therefore every floating-point sum) is identical from run to run.
    int slot[KINDS][R];
    if (want_grad) {
        const int K = SP;                                       // keys 0 .. S+1
        int *wcnt = reinterpret_cast<int *>(Fblk);              // [KINDS][W][K]   (set-up scratch)
        int *ktot = wcnt + KINDS * W * K;                       // [KINDS][K]
#pragma unroll
        for (int kind = 0; kind < KINDS; ++kind) {
            int cnt = 0;            // lane b: occurrences of key b seen so far in this wave
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int key = (kind == 0) ? st[j] : ((kind == 1) ? mv[j] : md[MOD ? j : 0]);
                int rank = 0;
                for (int b = 0; b < K; ++b) {
                    const unsigned long long mask = __ballot(key == b);
                    if (key == b)
                        rank = __builtin_amdgcn_readlane(cnt, b) +
                               __popcll(mask & ((1ull << lane) - 1ull));
                    if (lane == b) cnt += __popcll(mask);
                }
                slot[kind][j] = rank;                           // rank within (wave, key) for now
            }
            if (lane < K) wcnt[(kind * W + wave) * K + lane] = cnt;
        }
        __syncthreads();
        for (int e = tid; e < KINDS * K; e += NT) {
            const int kind = e / K, b = e - kind * K;
            int tot = 0;
            for (int w = 0; w < W; ++w) tot += wcnt[(kind * W + w) * K + b];
            ktot[e] = tot;
        }
        __syncthreads();
        for (int e = tid; e < KINDS * K; e += NT) {
            const int kind = e / K, b = e - kind * K;
            int start = 0;
            for (int bb = 0; bb < b; ++bb) start += ktot[kind * K + bb];
            segstart[kind * (SP + 1) + b] = start;
            if (b == K - 1) segstart[kind * (SP + 1) + K] = start + ktot[e];
            // per-wave base of this key: overwrite the counts with exclusive prefix + start
            int run = start;
            for (int w = 0; w < W; ++w) {
                const int cwb = wcnt[(kind * W + w) * K + b];
                wcnt[(kind * W + w) * K + b] = run;
                run += cwb;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kind = 0; kind < KINDS; ++kind)
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int key = (kind == 0) ? st[j] : ((kind == 1) ? mv[j] : md[MOD ? j : 0]);
                slot[kind][j] += wcnt[(kind * W + wave) * K + key];
            }
        __syncthreads();            // the scratch region becomes Fblk again
    }

    // block-wide column max of step t rides on the step barrier: every wave drops
    // its max into red[] at the end of step t, everybody folds it in at step t+1
    auto fold_norm = [&](double (&x)[R], double &edge_val, double &offacc) {
        float mx = red[0];
#pragma unroll
        for (int w = 1; w < W; ++w) mx = fmaxf(mx, red[w]);
        if (!(mx > -1e29f)) mx = 0.f;           // nothing reachable yet: keep the scale
        // (the fold is a float -- the column maximum rounded down to fp32 -- taken off doubles exactly)
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] -= (double)mx;
        edge_val -= (double)mx;
        offacc += (double)mx;
    };
    auto post_max = [&](const double (&x)[R]) {
        float mx = (float)x[0];
#pragma unroll
        for (int j = 1; j < R; ++j) mx = fmaxf(mx, (float)x[j]);
        mx = wave_allmax_dpp(mx);
        if (lane == 0) red[wave] = mx;
    };

    // ---- one forward column update (c_crf_flipflop.c:43-78); t = index of the row
    //      consumed; ends with the step barrier ------------------------------------------
    auto fwd_step = [&](double (&f)[R], const float *row, int t, bool norm_in, double &offacc) {
        double ein = (W > 1 && wave > 0) ? edgeF[((t - 1) & 1) * W + wave - 1] : neg;
        if (norm_in) fold_norm(f, ein, offacc);
        double left0 = wave_shift_up1(f[R - 1], neg);
        if (W > 1 && lane == 0) left0 = ein;
#pragma unroll
        for (int j = R - 1; j >= 0; --j) {
            const float ls = row[st[j]];
            const int mi = (j == 0) ? mvin0 : mv[j > 0 ? j - 1 : 0];
            const float lm = row[mi];
            const double left = (j == 0) ? left0 : f[j > 0 ? j - 1 : 0];
            const double av = fma((double)ls, (double)c, f[j]);
            double bv = fma((double)lm, (double)c, left);
            if (MOD) {
                const int di = (j == 0) ? mdin0 : md[j > 0 ? j - 1 : 0];
                const float dw = (j == 0) ? fwin0 : fw[j > 0 ? j - 1 : 0];
                bv = fma((double)row[di], (double)dw, bv);
            }
            f[j] = lse2d(av, bv);
        }
        if (W > 1 && lane == WAVE - 1) edgeF[(t & 1) * W + wave] = f[R - 1];
        if (((t + 1) & 3) == 0) post_max(f);
        __syncthreads();
    };
    // publish the column's wave-boundary values before the first step from it
    auto fwd_edge_init = [&](const double (&f)[R], int t0) {
        if (W > 1 && lane == WAVE - 1) edgeF[((t0 - 1) & 1) * W + wave] = f[R - 1];
        __syncthreads();
    };

    const int NK = (T + CK - 1) / CK;
    float *ck_n = a.ckpt + (size_t)ckslot * NK * (R * NT);
    double *ckoff_n = a.ckoff + (size_t)ckslot * NK;

    // ======================= forward sweep ===================================
    double f[R];
#pragma unroll
    for (int j = 0; j < R; ++j) f[j] = (p0 + j == 0) ? 0.0 : neg;           // :113-116
    double offF = 0.0;
    fwd_edge_init(f, 0);
    {
        float pre[MAXK];
        tile_fetch(0, pre);
        for (int k = 0; k < NK; ++k) {
            const int t0 = k * CK, nrows = min(CK, T - t0);
            tile_commit(t0, pre);           // (the previous tile's last step ended with a barrier)
            __syncthreads();
            if (k + 1 < NK) tile_fetch(t0 + CK, pre);
            if (want_grad) {
#pragma unroll
                // (a checkpoint column is a float snapshot of the double chain: ONE rounding per tile, which the
                // tile's recompute starts from -- it does not accumulate from tile to tile)
                for (int j = 0; j < R; ++j) ck_n[((size_t)k * R + j) * NT + tid] = (float)f[j];
                if (tid == 0) ckoff_n[k] = offF;
            }
            for (int i = 0; i < nrows; ++i) {
                const int t = t0 + i;
                fwd_step(f, tile + i * SP, t, t > 0 && (t & 3) == 0, offF);
            }
        }
    }
    // a column max published by the very last step is never folded in: harmless.
    // score = sum of factors + fwd[T][L-1]  (c_crf_flipflop.c:131)
    if (tid == (L - 1) / R) {
        const int jj = (L - 1) % R;
        double last = 0.0;
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (j == jj) last = f[j];
        misc[0] = last;
    }
    __syncthreads();
    const double fwd_score2 = offF + misc[0];
    if (!want_grad) {
        if (tid == 0) {
            const float cst = crf_add_cost(a, n, (float)(-(fwd_score2 * 0.6931471805599453) / (double)T) * a.out_scale);
            a.cost[n] = cst;
            if (a.status && !isfinite(cst)) atomicOr(a.status, 1u);
        }
        return;
    }

    // ======================= backward sweep + posterior =======================
    double b[R];
#pragma unroll
    for (int j = 0; j < R; ++j) b[j] = (p0 + j == L - 1) ? 0.0 : neg;       // :216-220
    double offB = 0.0;
    bool bad = false;
    int nbwd = 0;                       // backward steps done so far
    bool bnorm_pending = false;
    const float inv_cmod = MOD ? (1.0f / a.c_mod) : 0.f;
    if (W > 1 && lane == 0) edgeB[1 * W + wave] = b[0];        // slot (nbwd-1)&1 with nbwd = 0
    __syncthreads();
    {
        float pre[MAXK];
        tile_fetch((NK - 1) * CK, pre);
        for (int k = NK - 1; k >= 0; --k) {
            const int t0 = k * CK, nrows = min(CK, T - t0);
            tile_commit(t0, pre);
            if (k > 0) tile_fetch(t0 - CK, pre);
            // -- recompute the forward columns of this tile from its checkpoint
#pragma unroll
            for (int j = 0; j < R; ++j) f[j] = (double)ck_n[((size_t)k * R + j) * NT + tid];
            offF = ckoff_n[k];
            fwd_edge_init(f, t0);       // barrier: tile and edges are visible
            for (int i = 0; i < nrows; ++i) {
                const int t = t0 + i;
                // The checkpoint holds the column BEFORE the fold that was pending at
                // the tile boundary (CK % 4 == 0): re-post its column max so the step
                // folds exactly what the forward sweep folded.
                if (i == 0 && t > 0 && (t & 3) == 0) {
                    post_max(f);
                    __syncthreads();
                }
                // (column, offset) are stored pre-fold: a consistent pair
#pragma unroll
                for (int j = 0; j < R; ++j) Fblk[((size_t)i * R + j) * NT + tid] = (float)f[j];
                if (tid == 0) offs[i] = offF;
                fwd_step(f, tile + i * SP, t, t > 0 && (t & 3) == 0, offF);
            }
            // -- walk the tile backwards (c_crf_flipflop.c:150-182 fused with 372-413)
            for (int i = nrows - 1; i >= 0; --i) {
                const float *row = tile + i * SP;
                float *prow = Psort + (size_t)i * KINDS * LPAD;
                double ein = (W > 1 && wave < W - 1) ? edgeB[((nbwd - 1) & 1) * W + wave + 1] : neg;
                if (bnorm_pending) fold_norm(b, ein, offB);
                const double ct = fwd_score2 - offs[i] - offB;
                double right0 = wave_shift_down1(b[0], neg);
                if (W > 1 && lane == WAVE - 1) right0 = ein;
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const float ls = row[st[j]];
                    const float lm = row[mv[j]];
                    const double br = (j == R - 1) ? right0 : b[j < R - 1 ? j + 1 : 0];
                    const double as = fma((double)ls, (double)c, b[j]);
                    double am = fma((double)lm, (double)c, br);
                    if (MOD) am = fma((double)row[md[j]], (double)fw[j], am);
                    const double fc = (double)Fblk[((size_t)i * R + j) * NT + tid] - ct;
                    const float ps = fast_exp2((float)(fc + as));
                    const float pm = fast_exp2((float)(fc + am));
                    prow[slot[0][j]] = ps;
                    prow[LPAD + slot[1][j]] = pm;
                    if (MOD) prow[2 * LPAD + slot[MOD ? 2 : 0][j]] = pm * (fw[j] * inv_cmod);
                    b[j] = lse2d(as, am);
                }
                if (W > 1 && lane == 0) edgeB[(nbwd & 1) * W + wave] = b[0];
                ++nbwd;
                bnorm_pending = (nbwd & 3) == 0;
                if (bnorm_pending) post_max(b);
                __syncthreads();
            }
            // -- flush: one wave per row.  For each posterior stream the row's sorted
            //    array becomes lane-local inclusive prefixes (+ a per-lane base from a DPP
            //    wave scan); lane = transition id then takes the difference of the prefixes
            //    at its segment boundaries.  The row total (stay + move streams) is the
            //    reference's per-column softmax normaliser (c_crf_flipflop.c:400-401); output
            //    scaling -1/T (ctc.pyx:113).
            for (int row = wave; row < nrows; row += W) {
                float colval = 0.f, total = 0.f;
                float *lb = lanebase + wave * WAVE;
#pragma unroll
                for (int kind = 0; kind < KINDS; ++kind) {
                    float *arr = Psort + ((size_t)row * KINDS + kind) * LPAD;
                    float run = 0.f;
                    if constexpr (EPL % 4 == 0) {
                        // 16-byte LDS accesses (a scalar walk at stride EPL is 16-way bank-conflicted)
                        f4 *av = reinterpret_cast<f4 *>(arr + lane * EPL);
#pragma unroll
                        for (int e = 0; e < EPL / 4; ++e) {
                            f4 x = av[e];
                            x[0] += run;
                            x[1] += x[0];
                            x[2] += x[1];
                            x[3] += x[2];
                            run = x[3];
                            av[e] = x;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < EPL; ++e) {
                            run += arr[lane * EPL + e];
                            arr[lane * EPL + e] = run;
                        }
                    }
                    const float inc = wave_inclusive_scan_dpp(run);
                    lb[lane] = inc - run;
                    wave_lds_fence();
                    if (kind < 2) total += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inc), 63));
                    if (lane < S) {
                        const int s0 = segstart[kind * (SP + 1) + lane];
                        const int s1 = segstart[kind * (SP + 1) + lane + 1];
                        const float p1 = (s1 > 0) ? arr[s1 - 1] + lb[(s1 - 1) / EPL] : 0.f;
                        const float p0s = (s0 > 0) ? arr[s0 - 1] + lb[(s0 - 1) / EPL] : 0.f;
                        colval += (s1 > s0) ? (p1 - p0s) : 0.f;
                    }
                    wave_lds_fence();
                }
                const float g = crf_add_grad(a, (size_t)(t0 + row), n, lane, colval * (-gsc / (total * (float)T)), gsc);
                if (lane < S) {
                    bad |= !isfinite(g);
                    a.grad[(size_t)(t0 + row) * rowstride + (size_t)n * S + lane] = g;
                }
            }
            __syncthreads();
        }
    }
    // bwd score = bwd[0][0] + sum of factors (c_crf_flipflop.c:234); score = mean (:482-491)
    if (bnorm_pending) {
        double ein = 0.0;
        fold_norm(b, ein, offB);
    }
    if (tid == 0) {
        const double bwd_score2 = offB + b[0];
        const double score2 = 0.5 * (fwd_score2 + bwd_score2);
        const float cst = crf_add_cost(a, n, (float)(-(score2 * 0.6931471805599453) / (double)T) * a.out_scale);
        a.cost[n] = cst;
        if (a.status && !isfinite(cst)) atomicOr(a.status, 1u);
    }
    if (a.status && bad) atomicOr(a.status, 2u);
}

// the tail launch of the linear path (crf_band.hip): `a` = the retry's arguments (its own workspace arrays, wbias / klip;
// klip <= 0: no retry configuration for this call -- disowned reads go straight to the log domain), `ca` = the log-domain
// form's arguments (checkpoint columns for `nslots` workgroups of 16 waves, R = crf_tail_log_R(band R))
int crf_band_tail_dispatch(const BandArgs &a, const BandRetry &r, const CrfArgs &ca, int R, bool mod, size_t nslots, hipStream_t stream);
__host__ __device__ constexpr int crf_tail_log_R(int band_R) { return band_R == 4 ? 4 : 2; }

}  // namespace tk
