// beam_kernels.hip -- hash beam search over the flip-flop lattice for gfx950.
//
// Replaces taiyaki/decodeutil/c_hashdecode.c:346-507 (`flipflop_beamsearch`), its guiding
// backward pass c_flipflopfwdbwd.c:55-91 and the wrapper decodeutil.pyx:9-51.
//
// One wavefront per read; the beam (<= 12 elements for the 4-base alphabet; the reference's
// default is 5) lives in lanes 0..W-1 (hash, score, last state), a block's candidate records
// -- W x nbase extensions, then W stays -- one per lane.  What the reference does with two
// quicksorts per block is done without sorting:
//   * merge by hash: a record has at most ONE partner with the same hash (the stay of prefix
//     s and the extension of s[:-1] by its last base); every lane compares its hash with the
//     other candidates' (v_readlane sweep), the later one folds into the earlier one with the
//     reference's logsumexpf;
//   * top-k by score: rank = number of records that beat this one; rank r < width becomes beam
//     element r.  The order of records with EXACTLY equal scores is, in the reference, whatever
//     its quicksort (qsort.h) leaves, and it decides who stays in the beam; such ties turn up in
//     about one block in a hundred.  A block that has one is redone the reference's way: the
//     records are laid out in hash order (ranked in parallel) and lane 0 runs the reference's
//     sort procedure on them (`beam_qsort`), so the beam follows the reference through ties too.
// Sequences are never copied: every block stores one byte per beam element (parent slot,
// appended state) and the best element's sequence is walked back at the end (through LDS when
// the table fits).  The running beam cut (`beam_cut` > 0) is the reference's: a record is
// dropped if it is worse than the best seen BEFORE it by more than log(beam_cut) -- an exclusive
// prefix maximum in candidate order.
// expf / log1pf of the reference's logsumexpf are evaluated in double and rounded to float
// (correctly rounded results, what glibc's float functions return in all but rare cases), so
// scores agree with the reference to the last bit almost everywhere and the beam takes the
// same decisions.
#include "ff_common.h"

namespace tk {

constexpr int BEAM_MAXW = 12;

__device__ __forceinline__ unsigned long long beam_mix(unsigned long long h) {
    h ^= h >> 23;
    h *= 0x2127599bf4325c37ULL;
    h ^= h >> 47;
    return h;
}
// fasthash.c:95-103
__device__ __forceinline__ unsigned long long beam_chain(unsigned long long h, unsigned long long v) {
    h ^= beam_mix(v);
    h *= 0x880355f21e6d1965ULL;
    return beam_mix(h);
}
// expf(-a) for a in [0, 17), correctly rounded: exp in double to ~2^-50 (range reduction by ln 2 in two
// parts, Taylor polynomial of degree 13 on |r| <= 0.347), rounded once to float.  ocml's generic
// double exp + log1p made a log-sum-exp ~165 instructions, most of them half-rate; these two are ~60.
__device__ __forceinline__ float beam_expf_neg(float a) {
    const double x = -(double)a;
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(k, -6.93147180369123816490e-01, x);
    r = __builtin_fma(k, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;                  // 1 / 13!
    p = __builtin_fma(p, r, 2.08767569878681e-09);      // 1 / 12!
    p = __builtin_fma(p, r, 2.505210838544172e-08);     // 1 / 11!
    p = __builtin_fma(p, r, 2.755731922398589e-07);     // 1 / 10!
    p = __builtin_fma(p, r, 2.7557319223985893e-06);    // 1 / 9!
    p = __builtin_fma(p, r, 2.48015873015873e-05);      // 1 / 8!
    p = __builtin_fma(p, r, 1.984126984126984e-04);     // 1 / 7!
    p = __builtin_fma(p, r, 1.388888888888889e-03);     // 1 / 6!
    p = __builtin_fma(p, r, 8.333333333333333e-03);     // 1 / 5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02);    // 1 / 4!
    p = __builtin_fma(p, r, 1.6666666666666666e-01);    // 1 / 3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return (float)__builtin_ldexp(p, (int)k);
}
// log1pf(e) for e in (0, 1], correctly rounded: 1 + e is exact in double, folded to (1/sqrt 2, sqrt 2],
// log t = 2 atanh((t - 1) / (t + 1)) with the series to s^21 (|s| <= 0.1716: remainder < 2^-60).
__device__ __forceinline__ float beam_log1pf_unit(float e) {
    const double t0 = 1.0 + (double)e;
    const bool fold = t0 > 1.4142135623730951;
    const double t = fold ? 0.5 * t0 : t0;
    const double s = (t - 1.0) / (t + 1.0);
    const double z = s * s;
    double q = 1.0 / 21.0;
    q = __builtin_fma(q, z, 1.0 / 19.0);
    q = __builtin_fma(q, z, 1.0 / 17.0);
    q = __builtin_fma(q, z, 1.0 / 15.0);
    q = __builtin_fma(q, z, 1.0 / 13.0);
    q = __builtin_fma(q, z, 1.0 / 11.0);
    q = __builtin_fma(q, z, 1.0 / 9.0);
    q = __builtin_fma(q, z, 1.0 / 7.0);
    q = __builtin_fma(q, z, 1.0 / 5.0);
    q = __builtin_fma(q, z, 1.0 / 3.0);
    q = __builtin_fma(q, z, 1.0);
    const double l = 2.0 * s * q;
    return (float)(fold ? l + 6.93147180559945309417e-01 : l);
}
// c_hashdecode.c:50-54
__device__ __forceinline__ float beam_lse(float x, float y) {
    const float absdif = fabsf(x - y);
    // (clamped argument: both functions run unconditionally, a lane outside the range drops the result)
    const float tail = beam_log1pf_unit(beam_expf_neg(fminf(absdif, 17.0f)));
    return fmaxf(x, y) + ((absdif < 17.0f) ? tail : 0.0f);
}
__device__ __forceinline__ float rdl(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float bpf(float v, int l) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(l * 4, __float_as_int(v)));
}
__device__ __forceinline__ int bpi(int v, int l) { return __builtin_amdgcn_ds_bpermute(l * 4, v); }

// The sort procedure of taiyaki/decodeutil/qsort.h:39-186 with LESS(i, j) = key[i] > key[j]
// (c_hashdecode.c:156-158), on n <= 64 records in LDS: median of (second, middle, last), Sedgewick
// partition, insertion sort below 16 records, the smaller subfile first -- the same comparisons
// and exchanges in the same order, so equal keys end where the reference leaves them.
__device__ void beam_qsort(float *key, int *id, int n) {
    auto less = [&](int i, int j) { return key[i] > key[j]; };
    auto swap = [&](int i, int j) {
        const float k = key[i];
        key[i] = key[j];
        key[j] = k;
        const int t = id[i];
        id[i] = id[j];
        id[j] = t;
    };
    if (n <= 1) return;
    int lo = 0, hi = n - 1, sp = 0;
    int st_lo[8], st_hi[8];
    while (true) {
        if (hi - lo + 1 >= 16) {
            const int m = lo + ((hi - lo) >> 1);
            const int a1 = lo + 1, a2 = m, a3 = hi;
            if (less(a2, a1)) {
                if (less(a3, a2)) swap(a1, a3);
                else {
                    swap(a1, a2);
                    if (less(a3, a2)) swap(a2, a3);
                }
            } else if (less(a3, a2)) {
                swap(a2, a3);
                if (less(a2, a1)) swap(a1, a2);
            }
            swap(lo, m);
            int i = lo + 1, j = hi;
            while (true) {
                do ++i; while (less(i, lo));
                do --j; while (less(lo, j));
                if (i >= j) break;
                swap(i, j);
            }
            i = j + 1;
            swap(lo, j);
            --j;
            int bl, bh, sl, sh;
            if (j - lo >= hi - i) { bl = lo; bh = j; sl = i; sh = hi; }
            else { bl = i; bh = hi; sl = lo; sh = j; }
            if (sl == sh) { lo = bl; hi = bh; }
            else {
                st_lo[sp] = bl;
                st_hi[sp] = bh;
                ++sp;
                lo = sl;
                hi = sh;
            }
        } else {
            for (int q = lo + 1; q <= hi; ++q)
                for (int k = q; k > lo && less(k, k - 1); --k) swap(k, k - 1);
            if (sp == 0) break;
            --sp;
            lo = st_lo[sp];
            hi = st_hi[sp];
        }
    }
}

struct BeamArgs {
    const float *scores;        // (T, N, S)
    int T, N, nbase;
    int width;                  // max_beam_width
    float logcut;               // log(beam_cut); -inf = no cutting
    int guided;
    float *bwd;                 // workspace [N][T + 1][2 nbase]
    unsigned char *bp;          // workspace [N][T][16]: (parent slot << 4) | (appended state + 1)
    signed char *seq;           // out (N, T): flip-flop states, -1 padded
    int *seqlen;                // out (N)
    float *score;               // out (N)
    int lds_rows;               // rows of the back-pointer table kept in LDS (0: walk global memory)
};

__global__ __launch_bounds__(WAVE) void beam_kernel(BeamArgs a) {
    extern __shared__ unsigned char lds_bp[];           // [lds_rows][16]
    __shared__ unsigned long long nh[16];
    __shared__ float nsc[16];
    __shared__ int nlast[16], nbp[16];
    __shared__ float qkey[WAVE];
    __shared__ int qid[WAVE], qrank[WAVE];
    const int n = blockIdx.x, lane = threadIdx.x;
    const int T = a.T, nb = a.nbase, ns = 2 * nb, S = ns * (nb + 1);
    const size_t rowstride = (size_t)a.N * S;
    const float *sc = a.scores + (size_t)n * S;
    float *bwd = a.bwd + (size_t)n * (T + 1) * ns;
    unsigned char *bpn = a.bp + (size_t)n * T * 16;
    const int col = min(lane, S - 1);

    // ---- guiding backward pass (c_flipflopfwdbwd.c:55-91): lane = from-state ----------------
    if (lane < ns) bwd[(size_t)T * ns + lane] = 0.f;
    // (score rows are requested PF blocks ahead of their use: a block is a serial chain of nbase + 1
    // log-sum-exps, and a load issued where it is needed adds a memory round trip to every one)
    constexpr int PF = 4;
    auto load_row = [&](int blk) { return sc[(size_t)min(max(blk, 0), T - 1) * rowstride + col]; };
    if (a.guided) {
        float p = 0.f;                                  // pbwd[lane]
        float cur[PF], nxt[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) cur[q] = load_row(T - 1 - q);
        for (int b0 = T; b0 > 0; b0 -= PF) {
#pragma unroll
            for (int q = 0; q < PF; ++q) nxt[q] = load_row(b0 - 1 - PF - q);
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int blk = b0 - q;
                if (blk <= 0) break;
                const float row = cur[q];
                const int fr = min(lane, ns - 1);
                // to the flop of this state's base
                float c = bpf(row, ns * nb + fr) + bpf(p, nb + fr % nb);
                for (int to = 0; to < nb; ++to) c = beam_lse(c, bpf(row, to * ns + fr) + rdl(p, to));
                p = c;
                if (lane < ns) bwd[(size_t)(blk - 1) * ns + lane] = c;
            }
#pragma unroll
            for (int q = 0; q < PF; ++q) cur[q] = nxt[q];
        }
    } else {
        for (int i = lane; i < T * ns; i += WAVE) bwd[i] = 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();

    // ---- beam search --------------------------------------------------------------------
    // beam element i in lane i
    unsigned long long eh = beam_chain(0x880355f21e6d1965ULL, (unsigned long long)min(lane, nb - 1));
    float es = 0.f;
    int el = min(lane, nb - 1);
    int W = nb;
    auto load_bwd = [&](int blk) { return bwd[(size_t)min(blk + 1, T) * ns + min(lane, ns - 1)]; };
    float rowc[PF], bsvc[PF], rown[PF], bsvn[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) {
        rowc[q] = load_row(q);
        bsvc[q] = load_bwd(q);
    }
    for (int b0 = 0; b0 < T; b0 += PF) {
      // (this group's rows and backward scores were requested a group ago)
#pragma unroll
      for (int q = 0; q < PF; ++q) {
          rown[q] = load_row(b0 + PF + q);
          bsvn[q] = load_bwd(b0 + PF + q);
      }
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        const int blk = b0 + q;
        if (blk >= T) break;
        const float row = rowc[q];
        const float bsv = (lane < ns) ? bsvc[q] : 0.f;                                  // bwdscore[lane]
        const int next = W * nb, ncand = next + W;
        // candidate of this lane
        const bool is_ext = lane < next, is_cand = lane < ncand;
        const int i = is_ext ? lane / nb : min(max(lane - next, 0), W - 1);
        const int base = lane % nb;
        const unsigned pl = (unsigned)i;
        const float pscore = bpf(es, pl);
        const int plast = bpi(el, pl);
        const unsigned hlo = (unsigned)bpi((int)(unsigned)eh, pl), hhi = (unsigned)bpi((int)(unsigned)(eh >> 32), pl);
        const unsigned long long phash = ((unsigned long long)hhi << 32) | hlo;
        const int newstate = is_ext ? ((base != plast) ? base : plast + nb) : plast;
        const int tidx = plast + ns * min(newstate, nb);       // MOVE_IDX / STAY_IDX
        const float bnew = bpf(bsv, newstate);              // bwdscore[new last state]
        float cscore = (pscore + bpf(row, tidx)) + bnew;
        const unsigned long long chash = is_ext ? beam_chain(phash, (unsigned long long)newstate) : phash;
        bool valid = is_cand;
        if (a.logcut > -1e30f) {
            // lower bound from the best element (c_hashdecode.c:385-396), then the running maximum
            // over the records BEFORE this one in candidate order
            const int pb = __builtin_amdgcn_readlane(el, 0);
            float mx = rdl(row, nb * ns + pb) + rdl(bsv, pb < nb ? pb + nb : pb);
            for (int k = 0; k < nb; ++k) mx = fmaxf(mx, rdl(row, k * ns + pb) + rdl(bsv, k));
            mx += rdl(es, 0);
            float run = is_cand ? cscore : -__builtin_huge_valf();
            // inclusive prefix max over lanes, then shift by one lane
            for (int d = 1; d < WAVE; d <<= 1) {
                const float o = bpf(run, max(lane - d, 0));
                if (lane >= d) run = fmaxf(run, o);
            }
            float before = bpf(run, max(lane - 1, 0));
            if (lane == 0) before = -__builtin_huge_valf();
            valid = valid && !(cscore < fmaxf(mx, before) + a.logcut);
        }
        // ---- merge records with the same hash (= the same sequence)
        int partner = -1;
        for (int j = 0; j < ncand; ++j) {
            const unsigned jlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)chash, j);
            const unsigned jhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(chash >> 32), j);
            const bool jvalid = __builtin_amdgcn_readlane((int)valid, j) != 0;
            if (jvalid && j != lane && jlo == (unsigned)chash && jhi == (unsigned)(chash >> 32)) partner = j;
        }
        const float pscore2 = bpf(cscore, max(partner, 0));
        bool uniq = valid;
        if (valid && partner >= 0) {
            if (partner > lane) cscore = beam_lse(pscore2, cscore);     // keep the earlier record
            else uniq = false;
        }
        // ---- rank among the unique records by score
        int rank = 0, nuniq = 0;
        bool tie = false;
        for (int j = 0; j < ncand; ++j) {
            const float js = rdl(cscore, j);
            const bool ju = __builtin_amdgcn_readlane((int)uniq, j) != 0;
            nuniq += ju;
            rank += (ju && js > cscore) ? 1 : 0;
            tie = tie || (ju && uniq && j != lane && js == cscore);
        }
        if (__builtin_amdgcn_ballot_w64(tie) != 0ull) {
            // equal scores: the reference's order.  Its score sort starts from the records in
            // descending hash order (c_hashdecode.c:440), a merged pair as (sum, -inf).
            int hpos = (valid && !uniq) ? 1 : 0;
            for (int j = 0; j < ncand; ++j) {
                const unsigned jlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)chash, j);
                const unsigned jhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(chash >> 32), j);
                const bool jvalid = __builtin_amdgcn_readlane((int)valid, j) != 0;
                hpos += (jvalid && (((unsigned long long)jhi << 32) | jlo) > chash) ? 1 : 0;
            }
            const int nrec = __builtin_popcountll(__builtin_amdgcn_ballot_w64(valid));
            if (valid) {
                qkey[hpos] = uniq ? cscore : -__builtin_huge_valf();
                qid[hpos] = lane;
            }
            __syncthreads();
            if (lane == 0) beam_qsort(qkey, qid, nrec);
            __syncthreads();
            if (lane < nrec) qrank[qid[lane]] = lane;
            __syncthreads();
            if (valid) rank = qrank[lane];
        }
        const int newW = min(a.width, nuniq);              // c_hashdecode.c:474
        if (uniq && rank < newW) {
            nh[rank] = chash;
            nsc[rank] = cscore - bnew;                      // remove the backward contribution (:485)
            nlast[rank] = newstate;
            nbp[rank] = (i << 4) | (is_ext ? newstate + 1 : 0);
        }
        __syncthreads();
        if (lane < newW) {
            eh = nh[lane];
            es = nsc[lane];
            el = nlast[lane];
            const unsigned char b = (unsigned char)nbp[lane];
            if (blk < a.lds_rows) lds_bp[blk * 16 + lane] = b;
            bpn[(size_t)blk * 16 + lane] = b;
        }
        W = newW;
        __syncthreads();
      }
#pragma unroll
      for (int q = 0; q < PF; ++q) {
          rowc[q] = rown[q];
          bsvc[q] = bsvn[q];
      }
    }
    // ---- walk the best element's sequence back -----------------------------------------------
    if (lane == 0) {
        a.score[n] = es;
        int slot = 0, len = 1;
        for (int blk = T - 1; blk >= 0; --blk) {
            const unsigned char b = (blk < a.lds_rows) ? lds_bp[blk * 16 + slot] : bpn[(size_t)blk * 16 + slot];
            len += (b & 15) ? 1 : 0;
            slot = b >> 4;
        }
        const int first = slot;                             // the initial one-state sequence
        signed char *out = a.seq + (size_t)n * T;
        const int keep = min(len, T);                       // (the reference's buffer holds nblock states)
        a.seqlen[n] = keep;
        slot = 0;
        int pos = len - 1;
        for (int blk = T - 1; blk >= 0; --blk) {
            const unsigned char b = (blk < a.lds_rows) ? lds_bp[blk * 16 + slot] : bpn[(size_t)blk * 16 + slot];
            if (b & 15) {
                if (pos < T) out[pos] = (signed char)((b & 15) - 1);
                --pos;
            }
            slot = b >> 4;
        }
        if (T > 0) out[0] = (signed char)first;
        for (int k = keep; k < T; ++k) out[k] = -1;
    }
}

// ---------------------------------------------------------------------------------------------
// The single-read lattice passes of the decoder on their own: flipflop_forward / flipflop_backward
// (c_flipflopfwdbwd.c:55-152) with the wrappers' optional initial vector (decodeutil.pyx:54-108).
// One wavefront per read, lane = state; every log-sum-exp in the reference's order, so the
// matrices agree with the reference's to the rounding of expf / log1pf.
//   out (N, T + 1, 2 nbase); total (N) = logsumexp over the last (forward) / first (backward) row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void lattice_kernel(const float *__restrict__ scores, int T, int N, int nb,
                                                       int forward, const float *__restrict__ init,
                                                       float *__restrict__ out, float *__restrict__ total) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const int ns = 2 * nb, S = ns * (nb + 1);
    const size_t rowstride = (size_t)N * S;
    const float *sc = scores + (size_t)n * S;
    float *mat = out + (size_t)n * (T + 1) * ns;
    const int col = min(lane, S - 1), st = min(lane, ns - 1);
    float p = (init != nullptr) ? init[(size_t)n * ns + st] : 0.f;
    if (forward) {
        if (lane < ns) mat[lane] = p;
        for (int blk = 0; blk < T; ++blk) {
            const float row = sc[(size_t)blk * rowstride + col];
            // every lane runs both recurrences (ds_bpermute returns 0 for a source lane that is
            // masked off, so the gathers must not sit in divergent code) and keeps its own
            const int b = st % nb;
            // to the flop of base b: from its flip, then from itself (:129-134)
            const float cflop = beam_lse(bpf(row, ns * nb + b) + bpf(p, b), bpf(row, ns * nb + b + nb) + bpf(p, b + nb));
            // to the flip of base b: from every state, in order (:136-143)
            float cflip = bpf(row, b * ns) + rdl(p, 0);
            for (int fr = 1; fr < ns; ++fr) cflip = beam_lse(cflip, bpf(row, b * ns + fr) + rdl(p, fr));
            const float c = (st >= nb) ? cflop : cflip;
            p = c;
            if (lane < ns) mat[(size_t)(blk + 1) * ns + lane] = c;
        }
    } else {
        if (lane < ns) mat[(size_t)T * ns + lane] = p;
        for (int blk = T; blk > 0; --blk) {
            const float row = sc[(size_t)(blk - 1) * rowstride + col];
            float c = bpf(row, ns * nb + st) + bpf(p, nb + st % nb);
            for (int to = 0; to < nb; ++to) c = beam_lse(c, bpf(row, to * ns + st) + rdl(p, to));
            p = c;
            if (lane < ns) mat[(size_t)(blk - 1) * ns + lane] = c;
        }
    }
    float tot = rdl(p, 0);
    for (int i = 1; i < ns; ++i) tot = beam_lse(tot, rdl(p, i));
    if (lane == 0) total[n] = tot;
}

int lattice_dispatch(const float *scores, size_t T, size_t N, size_t nbase, int forward, const float *init,
                     float *out, float *total, hipStream_t stream) {
    if (nbase < 1 || nbase > 4) return 2;
    if (N == 0) return 0;
    hipLaunchKernelGGL(lattice_kernel, dim3((unsigned)N), dim3(WAVE), 0, stream, scores, (int)T, (int)N, (int)nbase,
                       forward, init, out, total);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

size_t beam_workspace_bytes(size_t T, size_t N, size_t nbase) {
    return N * (T + 1) * 2 * nbase * sizeof(float) + N * T * 16 + 512;
}

int beam_dispatch(const float *scores, size_t T, size_t N, size_t nbase, int width, float beam_cut, int guided,
                  signed char *seq, int *seqlen, float *score, void *workspace, size_t workspace_bytes,
                  hipStream_t stream) {
    if (nbase < 1 || nbase > 4 || width < 1 || width > BEAM_MAXW || (size_t)width * (nbase + 1) > WAVE) return 2;
    if (!(beam_cut >= 0.f) || beam_cut > 1.f) return 1;
    if (workspace_bytes < beam_workspace_bytes(T, N, nbase)) return 3;
    BeamArgs a;
    a.scores = scores;
    a.T = (int)T;
    a.N = (int)N;
    a.nbase = (int)nbase;
    a.width = width;
    a.logcut = beam_cut > 0.f ? logf(beam_cut) : -__builtin_huge_valf();
    a.guided = guided;
    const size_t bwd_bytes = (N * (T + 1) * 2 * nbase * sizeof(float) + 255) / 256 * 256;
    a.bwd = static_cast<float *>(workspace);
    a.bp = reinterpret_cast<unsigned char *>(static_cast<char *>(workspace) + bwd_bytes);
    a.seq = seq;
    a.seqlen = seqlen;
    a.score = score;
    a.lds_rows = (int)(T <= 3584 ? T : 3584);              // 56 KiB of dynamic LDS at most
    hipLaunchKernelGGL(beam_kernel, dim3((unsigned)N), dim3(WAVE), (size_t)a.lds_rows * 16, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

}  // namespace tk
