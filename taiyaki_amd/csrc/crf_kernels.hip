// crf_kernels.hip -- sequence-constrained flip-flop CRF score + gradient (plain and
// cat-mod) for gfx950.
//
// Replaces taiyaki/ctc/c_crf_flipflop.c:43-516 and c_cat_mod_flipflop.c:37-582
// (forward, backward, posterior scatter) and the index algebra of
// taiyaki/flipflopfings.py:6-31 / ctc.pyx:127-134,282-292.
//
// Design (see DESIGN.md "Kernel A"):
//   * one wavefront per read; lane l owns the R consecutive lattice positions
//     [l*R, (l+1)*R) in registers, so one time step is R independent cells per
//     lane plus ONE neighbour exchange; no workgroup barrier anywhere.
//   * arithmetic is log2-space (v_exp_f32 / v_log_f32 are base-2 natively):
//     cell = max(a,b) + log2(1 + 2^-|a-b|), the reference's logaddexp.  The
//     per-column max-subtraction of the reference (c_crf_flipflop.c:73-77) is
//     applied every 4th column (any common offset is exact to account for).
//   * score rows are staged CK rows at a time in LDS and gathered by
//     transition id; stay / move / mod ids live in registers.
//   * the (T+1) x L forward lattice is never written out: the forward sweep
//     stores one checkpoint column every CK steps; the backward sweep
//     recomputes each CK-column tile into LDS, walks it backwards fused with the
//     backward recursion and scatter-adds the posteriors into an LDS tile of
//     per-row transition bins (ds_add_f32, deterministic within one wave),
//     which is normalised per row (the reference's per-column softmax,
//     c_crf_flipflop.c:400-401) and streamed out once.
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
};

template <int R>
struct CrfCfg {
    static constexpr int CK = (256 / R) > 32 ? 32 : ((256 / R) < 4 ? 4 : (256 / R));
    static constexpr int MAXK = CK;     // tile prefetch registers: CK*64 floats >= CK*S
};

__host__ __device__ inline size_t crf_lds_bytes(int R, int CK, int S) {
    const int SP = S + 2;
    size_t b = 0;
    b += (size_t)CK * SP * 4;           // score tile
    b += (size_t)CK * SP * 4;           // gradient bins
    b += (size_t)CK * R * WAVE * 4;     // recomputed forward columns
    b += (size_t)CK * 8;                // per-row forward offsets (double)
    b += (size_t)CK * 4;                // per-row scale
    return (b + 15) / 16 * 16;
}

template <int R, bool MOD>
__global__ __launch_bounds__(WAVE) void crf_kernel(CrfArgs a) {
    constexpr int CK = CrfCfg<R>::CK;
    constexpr int MAXK = CrfCfg<R>::MAXK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id();
    const int n = blockIdx.x;
    const int T = a.T, N = a.N, S = a.S, SP = S + 2;
    const int L = a.seqlen[n];
    const bool want_grad = a.grad != nullptr;

    float *tile = reinterpret_cast<float *>(smem);                    // [CK][SP]
    float *gt = tile + CK * SP;                                       // [CK][SP]
    float *Fblk = gt + CK * SP;                                       // [CK][R][64]
    double *offs = reinterpret_cast<double *>(
        smem + (((size_t)(2 * CK * SP + CK * R * WAVE) * 4 + 7) / 8) * 8);  // [CK]
    float *rscale = reinterpret_cast<float *>(offs + CK);             // [CK]

    const size_t rowstride = (size_t)N * S;
    const float *lpn = a.lp + (size_t)n * S;

    // ---- tile movers: rows t0 .. t0+nrows-1 of this read <-> LDS ------------
    auto tile_fetch = [&](int t0, float (&pre)[MAXK]) {
        // unconditional loads (index clamped): no exec-mask branches, no vmcnt(0) stalls
        const int total = min(CK, T - t0) * S;
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            const int e = min(lane + WAVE * k, total - 1);
            const int row = e / S, col = e - row * S;
            pre[k] = lpn[(size_t)(t0 + row) * rowstride + col];
        }
    };
    auto tile_commit = [&](int t0, const float (&pre)[MAXK]) {
        const int total = min(CK, T - t0) * S;
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            const int e = lane + WAVE * k;
            if (e < total) {
                const int row = e / S, col = e - row * S;
                tile[row * SP + col] = pre[k];
            }
        }
        wave_lds_fence();
    };

    if (L == 0) {
        // c_crf_flipflop.c:269-272 / 458-464: cost 0, zero gradient rows
        if (lane == 0) a.cost[n] = 0.f;
        if (want_grad) {
            for (int t = 0; t < T; ++t)
                for (int col = lane; col < S; col += WAVE)
                    a.grad[(size_t)t * rowstride + (size_t)n * S + col] = 0.f;
        }
        return;
    }
    if (L > R * WAVE) {
        if (lane == 0) {
            a.cost[n] = __builtin_nanf("");
            if (a.status) atomicOr(a.status, 4u);
        }
        return;
    }

    // ---- per-position transition ids -> registers ---------------------------
    const int64_t off = a.seqoff[n];
    int st[R], mv[R], md[MOD ? R : 1];
    float fw[MOD ? R : 1];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int p = lane * R + j;
        st[j] = (p < L) ? a.stay[off + p] : S;                // S   = -LARGE sentinel slot
        mv[j] = (p < L - 1) ? a.move[off + p] : S;
        if (MOD) {
            md[j] = (p < L - 1) ? a.mod[off + p] : S + 1;     // S+1 = 0.0 sentinel slot
            fw[j] = (p < L - 1) ? a.modfact[off + p] * a.c_mod : 0.f;
        }
    }
    // transition INTO this lane's first position comes from the left neighbour
    int mvin0 = __shfl_up(mv[R - 1], 1, WAVE);
    int mdin0 = MOD ? __shfl_up(md[R - 1], 1, WAVE) : 0;
    float fwin0 = MOD ? __shfl_up(fw[R - 1], 1, WAVE) : 0.f;
    if (lane == 0) {
        mvin0 = S;
        mdin0 = S + 1;
        fwin0 = 0.f;
    }
    // sentinel slots of every LDS row (tile loads never touch them)
    for (int r = lane; r < CK; r += WAVE) {
        tile[r * SP + S] = NEG_LARGE;
        tile[r * SP + S + 1] = 0.f;
    }
    const float c = a.c_can;
    const float neg = NEG_LARGE * LOG2E;

    // ---- one forward column update (c_crf_flipflop.c:43-78) ------------------
    auto fwd_step = [&](float (&f)[R], const float *row) {
        float left0 = __shfl_up(f[R - 1], 1, WAVE);
        if (lane == 0) left0 = neg;
#pragma unroll
        for (int j = R - 1; j >= 0; --j) {
            const float ls = row[st[j]];
            const int mi = (j == 0) ? mvin0 : mv[j > 0 ? j - 1 : 0];
            const float lm = row[mi];
            const float left = (j == 0) ? left0 : f[j > 0 ? j - 1 : 0];
            const float av = fmaf(ls, c, f[j]);
            float bv = fmaf(lm, c, left);
            if (MOD) {
                const int di = (j == 0) ? mdin0 : md[j > 0 ? j - 1 : 0];
                const float dw = (j == 0) ? fwin0 : fw[j > 0 ? j - 1 : 0];
                bv = fmaf(row[di], dw, bv);
            }
            f[j] = lse2(av, bv);
        }
    };
    auto normalise = [&](float (&f)[R], double &offacc) {
        float mx = f[0];
#pragma unroll
        for (int j = 1; j < R; ++j) mx = fmaxf(mx, f[j]);
        mx = wave_allmax(mx);
#pragma unroll
        for (int j = 0; j < R; ++j) f[j] -= mx;
        offacc += (double)mx;
    };

    const int NK = (T + CK - 1) / CK;
    float *ck_n = a.ckpt + (size_t)n * NK * (R * WAVE);
    double *ckoff_n = a.ckoff + (size_t)n * NK;

    // ======================= forward sweep ===================================
    float f[R];
#pragma unroll
    for (int j = 0; j < R; ++j) f[j] = (lane * R + j == 0) ? 0.f : neg;     // :113-116
    double offF = 0.0;
    {
        float pre[MAXK];
        tile_fetch(0, pre);
        for (int k = 0; k < NK; ++k) {
            const int t0 = k * CK, nrows = min(CK, T - t0);
            tile_commit(t0, pre);
            if (k + 1 < NK) tile_fetch(t0 + CK, pre);
            if (want_grad) {
#pragma unroll
                for (int j = 0; j < R; ++j) ck_n[((size_t)k * R + j) * WAVE + lane] = f[j];
                if (lane == 0) ckoff_n[k] = offF;
            }
            for (int i = 0; i < nrows; ++i) {
                fwd_step(f, tile + i * SP);
                if (((t0 + i + 1) & 3) == 0) normalise(f, offF);
            }
        }
    }
    // score = sum of factors + fwd[T][L-1]  (c_crf_flipflop.c:131)
    float last = 0.f;
    {
        const int jj = (L - 1) % R;
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (j == jj) last = f[j];
        last = __shfl(last, (L - 1) / R, WAVE);
    }
    const double fwd_score2 = offF + (double)last;
    if (!want_grad) {
        if (lane == 0) {
            const float cst = (float)(-(fwd_score2 * 0.6931471805599453) / (double)T) * a.out_scale;
            a.cost[n] = cst;
            if (a.status && !isfinite(cst)) atomicOr(a.status, 1u);
        }
        return;
    }

    // ======================= backward sweep + posterior =======================
    float b[R];
#pragma unroll
    for (int j = 0; j < R; ++j) b[j] = (lane * R + j == L - 1) ? 0.f : neg;  // :216-220
    double offB = 0.0;
    bool bad = false;
    const float inv_cmod = MOD ? (1.0f / a.c_mod) : 0.f;
    {
        float pre[MAXK];
        tile_fetch((NK - 1) * CK, pre);
        for (int k = NK - 1; k >= 0; --k) {
            const int t0 = k * CK, nrows = min(CK, T - t0);
            tile_commit(t0, pre);
            if (k > 0) tile_fetch(t0 - CK, pre);
            // -- recompute the forward columns of this tile from its checkpoint
#pragma unroll
            for (int j = 0; j < R; ++j) f[j] = ck_n[((size_t)k * R + j) * WAVE + lane];
            offF = ckoff_n[k];
            for (int i = 0; i < nrows; ++i) {
#pragma unroll
                for (int j = 0; j < R; ++j) Fblk[(i * R + j) * WAVE + lane] = f[j];
                if (lane == 0) offs[i] = offF;
                fwd_step(f, tile + i * SP);
                if (((t0 + i + 1) & 3) == 0) normalise(f, offF);
            }
            for (int e = lane; e < CK * SP; e += WAVE) gt[e] = 0.f;
            wave_lds_fence();
            // -- walk the tile backwards (c_crf_flipflop.c:150-182 fused with 372-413)
            for (int i = nrows - 1; i >= 0; --i) {
                const int t = t0 + i;
                const float *row = tile + i * SP;
                float *grow = gt + i * SP;
                const float ct = (float)(fwd_score2 - offs[i] - offB);
                float right0 = __shfl_down(b[0], 1, WAVE);
                if (lane == WAVE - 1) right0 = neg;
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const float ls = row[st[j]];
                    const float lm = row[mv[j]];
                    const float br = (j == R - 1) ? right0 : b[j < R - 1 ? j + 1 : 0];
                    const float as = fmaf(ls, c, b[j]);
                    float am = fmaf(lm, c, br);
                    if (MOD) am = fmaf(row[md[j]], fw[j], am);
                    const float fc = Fblk[(i * R + j) * WAVE + lane] - ct;
                    const float ps = fast_exp2(fc + as);
                    const float pm = fast_exp2(fc + am);
                    atomicAdd(grow + st[j], ps);
                    atomicAdd(grow + mv[j], pm);
                    if (MOD) atomicAdd(grow + md[j], pm * (fw[j] * inv_cmod));
                    b[j] = lse2(as, am);
                }
                if (((T - t) & 3) == 0) normalise(b, offB);
            }
            wave_lds_fence();
            // -- per-row normalisation (the reference's softmax over the column's
            //    2L-1 transitions, c_crf_flipflop.c:400-401) and output scaling
            //    -1/T (ctc.pyx:113)
            if (lane < nrows) {
                // every posterior lands in exactly one stay/move bin (ids < ncan);
                // for cat-mod the mod bins hold p * fact ON TOP and are not summed
                float sum = 0.f;
                for (int col = 0; col < a.ncan; ++col) sum += gt[lane * SP + col];
                rscale[lane] = sum;
            }
            wave_lds_fence();
            {
                const int total = nrows * S;
                for (int e = lane; e < total; e += WAVE) {
                    const int row = e / S, col = e - row * S;
                    const float sc = -1.0f / (rscale[row] * (float)T);
                    const float g = gt[row * SP + col] * sc;
                    bad |= !isfinite(g);
                    a.grad[(size_t)(t0 + row) * rowstride + (size_t)n * S + col] = g;
                }
            }
            wave_lds_fence();
        }
    }
    // bwd score = bwd[0][0] + sum of factors (c_crf_flipflop.c:234); score = mean (:482-491)
    const float first = __shfl(b[0], 0, WAVE);
    const double bwd_score2 = offB + (double)first;
    if (lane == 0) {
        const double score2 = 0.5 * (fwd_score2 + bwd_score2);
        const float cst = (float)(-(score2 * 0.6931471805599453) / (double)T) * a.out_scale;
        a.cost[n] = cst;
        if (a.status && !isfinite(cst)) atomicOr(a.status, 1u);
    }
    if (a.status && bad) atomicOr(a.status, 2u);
}

// ---------------------------------------------------------------------------
// index construction (flipflopfings.py:6-31, ctc.pyx:127-134, 282-292)
// ---------------------------------------------------------------------------
__global__ void seqoff_kernel(const int32_t *__restrict__ seqlen, int nbatch,
                              int64_t *__restrict__ seqoff) {
    // single block; chunked serial prefix sum (nbatch is a few thousand at most)
    __shared__ long long part[256];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int per = (nbatch + nt - 1) / nt;
    const int lo = min(nbatch, tid * per), hi = min(nbatch, lo + per);
    long long s = 0;
    for (int i = lo; i < hi; ++i) s += seqlen[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        long long acc = 0;
        for (int i = 0; i < nt; ++i) {
            const long long v = part[i];
            part[i] = acc;
            acc += v;
        }
    }
    __syncthreads();
    long long acc = part[tid];
    for (int i = lo; i < hi; ++i) {
        seqoff[i] = acc;
        acc += seqlen[i];
    }
    if (hi == nbatch && lo <= nbatch) seqoff[nbatch] = acc;     // identical value from every writer
}

__global__ void build_indices_kernel(const int32_t *__restrict__ seqs,
                                     const int32_t *__restrict__ seqlen,
                                     const int64_t *__restrict__ seqoff, int nbase,
                                     const int32_t *__restrict__ mod_cats,
                                     const int32_t *__restrict__ can_mods_offsets,
                                     const float *__restrict__ mod_cat_weights,
                                     int32_t *__restrict__ stay, int32_t *__restrict__ move,
                                     int32_t *__restrict__ mod, float *__restrict__ fact) {
    const int n = blockIdx.x;
    const int L = seqlen[n];
    const int64_t off = seqoff[n];
    const int ns = 2 * nbase, ncan = ns * (nbase + 1);
    for (int p = threadIdx.x; p < L; p += blockDim.x) {
        const int cp = seqs[off + p];
        stay[off + p] = cp + min(cp, nbase) * ns;                 // flipflopfings.py:20-31
        if (p + 1 < L) {
            const int cn = seqs[off + p + 1];
            move[off + p] = cp + min(cn, nbase) * ns;             // flipflopfings.py:6-17
            if (mod_cats != nullptr) {
                // ctc.pyx:288-292
                const int mseq = can_mods_offsets[cn % nbase] + mod_cats[off + p + 1];
                mod[off + p] = ncan + mseq;
                fact[off + p] = mod_cat_weights[mseq];
            }
        } else {
            move[off + p] = 0;
            if (mod_cats != nullptr) {
                mod[off + p] = ncan;
                fact[off + p] = 0.f;
            }
        }
    }
}

int build_indices_dispatch(const int32_t *seqs, const int32_t *seqlen, size_t nbatch,
                           size_t nbase, const int32_t *mod_cats,
                           const int32_t *can_mods_offsets, const float *mod_cat_weights,
                           int64_t *seqoff, int32_t *stay, int32_t *move, int32_t *mod,
                           float *fact, hipStream_t stream) {
    hipLaunchKernelGGL(seqoff_kernel, dim3(1), dim3(256), 0, stream, seqlen, (int)nbatch, seqoff);
    hipLaunchKernelGGL(build_indices_kernel, dim3((unsigned)nbatch), dim3(128), 0, stream, seqs,
                       seqlen, seqoff, (int)nbase, mod_cats, can_mods_offsets, mod_cat_weights,
                       stay, move, mod, fact);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// ---------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------
static int crf_pick_R(size_t max_seqlen) {
    int R = 1;
    while ((size_t)R * WAVE < max_seqlen) R *= 2;
    return R;
}

static int crf_ck(int R) {
    const int c = 256 / R;
    return c > 32 ? 32 : (c < 4 ? 4 : c);
}

size_t crf_workspace_bytes(size_t ntrans, size_t nblk, size_t nbatch, size_t max_seqlen,
                           int want_grad) {
    (void)ntrans;
    if (!want_grad) return 256;
    if (max_seqlen == 0) max_seqlen = nblk + 1;
    const int R = crf_pick_R(max_seqlen);
    const size_t NK = (nblk + crf_ck(R) - 1) / crf_ck(R);
    const size_t ck = nbatch * NK * (size_t)R * WAVE * sizeof(float);
    const size_t co = nbatch * NK * sizeof(double);
    return (ck + 255) / 256 * 256 + (co + 255) / 256 * 256 + 256;
}

template <int R, bool MOD>
static int crf_launch_one(const CrfArgs &a, hipStream_t stream) {
    const size_t lds = crf_lds_bytes(R, CrfCfg<R>::CK, a.S);
    if (lds > 160 * 1024) return 2;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&crf_kernel<R, MOD>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return 4;
    }
    hipLaunchKernelGGL((crf_kernel<R, MOD>), dim3(a.N), dim3(WAVE), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

template <bool MOD>
static int crf_launch_mod(int R, const CrfArgs &a, hipStream_t stream) {
    switch (R) {
        case 1: return crf_launch_one<1, MOD>(a, stream);
        case 2: return crf_launch_one<2, MOD>(a, stream);
        case 4: return crf_launch_one<4, MOD>(a, stream);
        case 8: return crf_launch_one<8, MOD>(a, stream);
        case 16: return crf_launch_one<16, MOD>(a, stream);
        case 32: return crf_launch_one<32, MOD>(a, stream);
        case 64: return crf_launch_one<64, MOD>(a, stream);
        default: return 2;
    }
}

int crf_dispatch(const float *logprob, size_t ntrans, size_t nblk, size_t nbatch,
                 const int32_t *stayidx, const int32_t *moveidx, const int32_t *modidx,
                 const float *modfact, const int32_t *seqlen, const int64_t *seqoff,
                 size_t max_seqlen, size_t ncan, float sharp_can, float sharp_mod,
                 float out_scale, float *cost, float *grad, void *workspace,
                 size_t workspace_bytes, uint32_t *status, hipStream_t stream) {
    if (ntrans > 62 || ncan > ntrans || ncan == 0) return 2;
    if (max_seqlen == 0) max_seqlen = nblk + 1;
    const int R = crf_pick_R(max_seqlen);
    if (R > 64) return 2;
    const size_t need = crf_workspace_bytes(ntrans, nblk, nbatch, max_seqlen, grad != nullptr);
    if (need > workspace_bytes) return 3;
    CrfArgs a;
    a.lp = logprob;
    a.T = (int)nblk;
    a.N = (int)nbatch;
    a.S = (int)ntrans;
    a.ncan = (int)ncan;
    a.stay = stayidx;
    a.move = moveidx;
    a.mod = modidx;
    a.modfact = modfact;
    a.seqlen = seqlen;
    a.seqoff = seqoff;
    a.c_can = sharp_can * LOG2E;
    a.c_mod = sharp_mod * LOG2E;
    a.out_scale = out_scale;
    a.cost = cost;
    a.grad = grad;
    const size_t NK = (nblk + crf_ck(R) - 1) / crf_ck(R);
    const size_t ckb = (nbatch * NK * (size_t)R * WAVE * sizeof(float) + 255) / 256 * 256;
    a.ckpt = static_cast<float *>(workspace);
    a.ckoff = reinterpret_cast<double *>(static_cast<char *>(workspace) + (grad ? ckb : 0));
    a.status = status;
    return modidx != nullptr ? crf_launch_mod<true>(R, a, stream)
                             : crf_launch_mod<false>(R, a, stream);
}

}  // namespace tk
