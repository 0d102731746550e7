// crf_kernels.hip -- sequence-constrained flip-flop CRF score + gradient (plain and
// cat-mod) for gfx950.
//
// Replaces taiyaki/ctc/c_crf_flipflop.c:43-516 and c_cat_mod_flipflop.c:37-582
// (forward, backward, posterior scatter) and the index algebra of
// taiyaki/flipflopfings.py:6-31 / ctc.pyx:127-134,282-292.
//
// Two forms (see DESIGN.md "Kernel A"):
//   * crf_band.hip (default): banded skewed sweep + row-parallel posterior pass, both lattices
//     of the band in HBM.  This file holds its launcher, the index construction and
//   * crf_kernel below, the single-launch CHECKPOINT form for batches whose lattices would not
//     fit the workspace cap: one workgroup of W wavefronts per read, position-parallel (thread
//     g owns R consecutive lattice positions in registers, one DPP / LDS neighbour exchange and
//     one s_barrier per time step, log2-space LSE, column max every 4th step tracked in fp64);
//     the forward sweep stores one checkpoint column every CK steps, the backward sweep
//     recomputes each CK-column tile into LDS, walks it backwards fused with the backward
//     recursion and writes every posterior to a slot PRE-SORTED by transition id; at tile flush
//     a wave turns a row into per-id sums with a DPP prefix scan and boundary differences (no
//     atomics), normalises the row (c_crf_flipflop.c:400-401) and streams it out once.
#include <stdio.h>
#include <stdlib.h>

#include "crf_log.h"

namespace tk {

// Workgroup n does read n (TK_CRF_MODE=ckpt, batches the band path does not take).  Behind a band launch the disowned
// reads are redone by that path's own tail launch (crf_band.hip: crf_band_tail_kernel), which calls crf_read itself.
template <int R, int W, bool MOD>
__global__ __launch_bounds__(W *WAVE) void crf_kernel(CrfArgs a) {
    crf_read<R, W, MOD>(a, (int)blockIdx.x, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------
// index construction (flipflopfings.py:6-31, ctc.pyx:127-134, 282-292)
// ---------------------------------------------------------------------------
__global__ void seqoff_kernel(const int32_t *__restrict__ seqlen, int nbatch,
                              int64_t *__restrict__ seqoff, long long total_len, uint32_t *status) {
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
    // Offsets are CLAMPED to the label array: a batch that announces more labels than it hands over
    // is flagged, and its reads end where the array ends (the CRF kernels take
    // min(seqlen, seqoff[n+1] - seqoff[n]) for a read's length): no kernel indexes past `total_len`.
    for (int i = lo; i < hi; ++i) {
        seqoff[i] = min(acc, total_len);
        acc += seqlen[i];
    }
    if (hi == nbatch && lo <= nbatch) {
        seqoff[nbatch] = min(acc, total_len);       // identical value from every writer
        if (acc > total_len && status) atomicOr(status, 8u);    // more labels announced than handed over
    }
}

__global__ void build_indices_kernel(const int32_t *__restrict__ seqs,
                                     const int32_t *__restrict__ seqlen,
                                     int64_t *__restrict__ seqoff, int nbatch, int nbase,
                                     const int32_t *__restrict__ mod_cats,
                                     const int32_t *__restrict__ can_mods_offsets,
                                     const float *__restrict__ mod_cat_weights,
                                     int32_t *__restrict__ stay, int32_t *__restrict__ move,
                                     int32_t *__restrict__ mod, float *__restrict__ fact,
                                     long long total_len, uint32_t *__restrict__ status) {
    const int n = blockIdx.x;
    // this read's offset = the sum of the lengths before it, clamped to the label array (see
    // seqoff_kernel): every workgroup sums for itself -- a batch is a few hundred reads, and it saves
    // the separate prefix-sum launch (~5 us of a 170 us loss path)
    __shared__ long long off_sh;
    __shared__ long long wsum[2][2];    // [wave][mine | all]: the block is two wavefronts
    if (nbatch == 0) {                  // (offsets already there: seqoff_kernel ran)
        if (threadIdx.x == 0) off_sh = seqoff[n];
        __syncthreads();
    } else {
        long long mine = 0, all = 0;
        for (int i = threadIdx.x; i < nbatch; i += blockDim.x) {
            const long long v = seqlen[i];
            all += v;
            if (i < n) mine += v;
        }
        // wave sums by butterfly, one hand-over through LDS -- the tree of fourteen barriers this replaces was
        // most of the kernel
        auto wave_sum = [&](long long v) {
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE);
            return v;
        };
        const long long wm = wave_sum(mine), wa = wave_sum(all);
        if ((threadIdx.x & (WAVE - 1)) == 0) {
            wsum[threadIdx.x >> 6][0] = wm;
            wsum[threadIdx.x >> 6][1] = wa;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            long long m = 0, t = 0;
            for (unsigned w = 0; w < (blockDim.x + WAVE - 1) / WAVE && w < 2; ++w) {
                m += wsum[w][0];
                t += wsum[w][1];
            }
            off_sh = m;
            seqoff[n] = min(m, total_len);
            if (n == nbatch - 1) {
                seqoff[nbatch] = min(t, total_len);
                if (t > total_len && status) atomicOr(status, 8u);      // more labels announced than handed over
            }
        }
        __syncthreads();
    }
    const int64_t off = min(off_sh, total_len);
    // (a batch that announces more labels than were handed over is flagged by seqoff_kernel;
    // here its reads are cut at the end of the label array)
    const int L = (int)max(0ll, min((long long)seqlen[n], total_len - (long long)off));
    const int ns = 2 * nbase, ncan = ns * (nbase + 1);
    bool bad = false;
    // the reference asserts 0 <= move / stay index < ntrans (ctc.pyx:127-134); a bad label is
    // clamped here, so that no kernel downstream gathers out of range, and reported
    auto label = [&](int64_t i) {
        const int c = seqs[i];
        bad |= c < 0 || c >= ns;
        return min(max(c, 0), ns - 1);
    };
    for (int p = threadIdx.x; p < L; p += blockDim.x) {
        const int cp = label(off + p);
        stay[off + p] = cp + min(cp, nbase) * ns;                 // flipflopfings.py:20-31
        if (p + 1 < L) {
            const int cn = label(off + p + 1);
            move[off + p] = cp + min(cn, nbase) * ns;             // flipflopfings.py:6-17
            if (mod_cats != nullptr) {
                // ctc.pyx:288-292
                const int lo = can_mods_offsets[cn % nbase], hi = can_mods_offsets[cn % nbase + 1];
                int mseq = lo + mod_cats[off + p + 1];
                bad |= mseq < lo || mseq >= hi;
                mseq = min(max(mseq, lo), hi - 1);
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
    if (bad && status) atomicOr(status, 8u);
}

int build_indices_dispatch(const int32_t *seqs, const int32_t *seqlen, size_t nbatch,
                           size_t nbase, const int32_t *mod_cats,
                           const int32_t *can_mods_offsets, const float *mod_cat_weights,
                           int64_t *seqoff, int32_t *stay, int32_t *move, int32_t *mod,
                           float *fact, size_t total_len, uint32_t *status, hipStream_t stream) {
    // (one launch: every workgroup computes its read's offset itself; seqoff_kernel is kept for
    // batches of more than 8192 reads, where the redundant sums would cost more than a launch)
    if (nbatch > 8192)
        hipLaunchKernelGGL(seqoff_kernel, dim3(1), dim3(256), 0, stream, seqlen, (int)nbatch, seqoff,
                           (long long)total_len, status);
    hipLaunchKernelGGL(build_indices_kernel, dim3((unsigned)nbatch), dim3(128), 0, stream, seqs,
                       seqlen, seqoff, (int)(nbatch > 8192 ? 0 : nbatch), (int)nbase, mod_cats, can_mods_offsets, mod_cat_weights,
                       stay, move, mod, fact, (long long)total_len, status);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// ---------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------
struct CrfShape {
    int R, W;
};

// (R, W) from the longest sequence: 2 cells per lane, up to 16 waves, then deeper strips
static CrfShape crf_pick_shape(size_t max_seqlen) {
    if (max_seqlen <= WAVE) return {1, 1};
    int W = 1;
    while ((size_t)2 * W * WAVE < max_seqlen && W < 16) W *= 2;
    int R = 2;
    while ((size_t)R * W * WAVE < max_seqlen) R *= 2;
    return {R, W};
}

// (a cost-only call keeps no checkpoint column: crf_kernel stores them under want_grad only)
static size_t crf_ckpt_bytes(size_t nblk, size_t nbatch, CrfShape sh, bool want_grad = true) {
    if (!want_grad) return 256;
    const int CK = crf_ck(sh.R, sh.W, 3);   // the cat-mod tile is the smaller one: upper bound
    const size_t NK = (nblk + CK - 1) / CK;
    const size_t ck = nbatch * NK * (size_t)sh.R * sh.W * WAVE * sizeof(float);
    const size_t co = nbatch * NK * sizeof(double);
    return (ck + 255) / 256 * 256 + (co + 255) / 256 * 256 + 256;
}

// Which form of kernel A runs (TK_CRF_MODE overrides: band | ckpt):
//   band     crf_band.hip -- linear-domain banded skewed sweep + recomputing gradient pass, followed
//            by ONE tail launch (crf_band.hip: crf_band_tail_kernel) that retries the reads the band path disowned alone
//            and redoes in the log domain (crf_read, crf_log.h) what it disowns twice
//            (default whenever the sequences fit 16 waves x 256 cells and the checkpoint columns
//            fit the workspace cap)
//   ckpt     the single-launch log-domain checkpoint/recompute kernel of this file on every read:
//            three serial passes per read (workspace-bound batches, and the band path's safety net)
enum CrfMode { CRF_BAND, CRF_CKPT };
static size_t crf_lattice_cap_bytes() {
    // TK_CRF_LATTICE_MB, else a quarter of THIS device's memory (72 GB of an MI355X's 288; round 3's checkpoint
    // columns are 3.9 GB at T = 4000 / N = 256, so the cap is about smaller devices and partitions, and about
    // callers that do not know their longest sequence).  No device (the build container): 40 GiB.
    if (const char *e = TK_LAB_ENV("TK_CRF_LATTICE_MB")) return (size_t)atoll(e) * 1024 * 1024;
    static size_t cap[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        (void)hipGetLastError();
        return (size_t)40960 * 1024 * 1024;
    }
    if (cap[dev] == 0) {
        size_t total = 0;
        if (hipDeviceTotalMem(&total, dev) != hipSuccess || total == 0) {
            (void)hipGetLastError();
            total = (size_t)160 * 1024 * 1024 * 1024;
        }
        cap[dev] = total / 4;
    }
    return cap[dev];
}
// `bk`: the block length the linear path would use for this call (crf_band_pick_block; 0 = it does not take it)
// the band layout's size for workspace queries: the plain CRF and cat-mod may pick different cells per lane
// (crf_band_pick_R), hence different padded read lengths -- the bound covers both
static size_t crf_band_total_bound(size_t ntrans, size_t nblk, size_t nbatch, size_t max_seqlen, bool want_grad, int bk) {
    const size_t a = crf_band_layout(ntrans, nblk, nbatch, max_seqlen, true, want_grad, bk).total;
    const size_t b = crf_band_layout(ntrans, nblk, nbatch, max_seqlen, false, want_grad, bk).total;
    return a > b ? a : b;
}
static CrfMode crf_pick_mode(size_t ntrans, size_t nblk, size_t nbatch, size_t max_seqlen, bool want_grad, int bk) {
    const char *e = TK_LAB_ENV("TK_CRF_MODE");
    const bool force_ckpt = e && e[0] == 'c';
    if (!force_ckpt && bk > 0 && crf_band_fits(max_seqlen) &&
        crf_band_total_bound(ntrans, nblk, nbatch, max_seqlen, want_grad, bk) <= crf_lattice_cap_bytes())
        return CRF_BAND;
    return CRF_CKPT;
}

// the retry's workspace (round 6): the band layout of 4-step blocks for crf_band_retry_slots(nbatch) reads (gradient
// form whatever the call: a cost-only retry runs the same sweeps)
static size_t crf_retry_bytes(size_t ntrans, size_t nblk, size_t nbatch, size_t max_seqlen) {
    const int R = crf_band_retry_R(max_seqlen);
    const size_t a = crf_band_layout(ntrans, nblk, crf_band_retry_slots(nbatch), max_seqlen, true, true, 4, R).total;
    const size_t b = crf_band_layout(ntrans, nblk, crf_band_retry_slots(nbatch), max_seqlen, false, true, 4, R).total;
    return a > b ? a : b;
}
// the log-domain form's checkpoint columns behind the band path: one set per workgroup of the TAIL launch (crf_band.hip:
// crf_band_tail_kernel -- 16 waves, cells per lane by the band launch's)
static CrfShape crf_tail_shape(int band_R) { return {crf_tail_log_R(band_R), 16}; }
static size_t crf_tail_ckpt_bound(size_t ntrans, size_t nblk, size_t nbatch, size_t max_seqlen, bool want_grad) {
    (void)ntrans;
    return crf_ckpt_bytes(nblk, crf_band_retry_slots(nbatch), crf_tail_shape(crf_band_retry_R(max_seqlen)), want_grad);
}

// workspace = [band layout (band mode only)] [the retry's band layout] [checkpoint columns + offsets of the log-domain form:
// behind the band path one set per workgroup of the tail launch -- a sixteenth of the batch, at least 4: it redoes only what
// the linear path disowned twice, none on the inputs a network produces (round 3 sized them for the whole batch: 8.4 of
// 12.2 GB at T = 4000 / N = 256) --, else one per read]
// `sharp`: the call's sharpening factor -- it picks the linear path's block length, and short blocks keep
// more checkpoint columns.  The block lengths of the plain CRF and of cat-mod differ; the bound covers both.
size_t crf_workspace_bytes_sharp(size_t ntrans, size_t nblk, size_t nbatch, size_t max_seqlen,
                                 int want_grad, float sharp) {
    if (max_seqlen == 0) max_seqlen = nblk + 1;
    const CrfShape sh = crf_pick_shape(max_seqlen);
    // (every block length either form may take for this shape: with and without per-column factors; a batch with narrow
    // bands takes 8 steps where wider ones take 12)
    // (... and whatever the batch's bulk is: between "unknown" and "every read as long as the longest")
    auto shortest = [](int x, int y) { return (x > 0 && y > 0) ? (x < y ? x : y) : (x > 0 ? x : y); };
    int bk = 0;
    for (int bulk = 0; bulk < 2; ++bulk)
        for (int form = 0; form < 3; ++form)
            bk = shortest(bk, crf_band_pick_block(sharp, form > 0, max_seqlen, form == 2, nblk, bulk ? max_seqlen : 0).bk);
    // (the cat-mod layout is the larger one: an upper bound for both)
    // (a call whose own block choice differs from the one assumed here -- another sharpening factor than the
    // query's -- needs what ITS factor's query returns; with less, crf_dispatch falls back to the log-domain
    // kernel on every read if the workspace holds that kernel's whole-batch columns, and returns 3 otherwise:
    // sizing every workspace for that case would be 8.0 instead of 4.7 GB at T = 4000 / N = 256)
    if (crf_pick_mode(ntrans, nblk, nbatch, max_seqlen, want_grad != 0, bk) == CRF_BAND)
        return crf_tail_ckpt_bound(ntrans, nblk, nbatch, max_seqlen, want_grad != 0) +
               crf_band_total_bound(ntrans, nblk, nbatch, max_seqlen, want_grad != 0, bk) + crf_retry_bytes(ntrans, nblk, nbatch, max_seqlen);
    return crf_ckpt_bytes(nblk, nbatch, sh, want_grad != 0);
}

size_t crf_workspace_bytes(size_t ntrans, size_t nblk, size_t nbatch, size_t max_seqlen,
                           int want_grad) {
    return crf_workspace_bytes_sharp(ntrans, nblk, nbatch, max_seqlen, want_grad, 1.0f);
}

template <int R, int W, bool MOD>
static int crf_launch_one(const CrfArgs &a, hipStream_t stream, size_t nwg) {
    const size_t lds = crf_lds_bytes(R, W, a.S, MOD ? 3 : 2);
    if (lds > 160 * 1024) return 2;
    if (raise_dynamic_lds(reinterpret_cast<const void *>(&crf_kernel<R, W, MOD>))) return 4;
    hipLaunchKernelGGL((crf_kernel<R, W, MOD>), dim3((unsigned)nwg), dim3(W * WAVE), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

template <bool MOD>
static int crf_launch_mod(CrfShape sh, const CrfArgs &a, hipStream_t stream, size_t nwg) {
    const int key = sh.R * 100 + sh.W;
#define TK_CRF_CASE(R_, W_)                                                           \
    case R_ * 100 + W_:                                                               \
        return crf_launch_one<R_, W_, MOD>(a, stream, nwg);
    switch (key) {
        TK_CRF_CASE(1, 1)
        TK_CRF_CASE(2, 1)
        TK_CRF_CASE(2, 2)
        TK_CRF_CASE(2, 4)
        TK_CRF_CASE(2, 8)
        TK_CRF_CASE(2, 16)
        TK_CRF_CASE(4, 16)
        default: return 2;
    }
#undef TK_CRF_CASE
}

int crf_dispatch(const float *logprob, size_t ntrans, size_t nblk, size_t nbatch,
                 const int32_t *stayidx, const int32_t *moveidx, const int32_t *modidx,
                 const float *modfact, const int32_t *seqlen, const int64_t *seqoff,
                 size_t max_seqlen, size_t ncan, float sharp_can, float sharp_mod,
                 float out_scale, float grad_scale, const float *grad_scale_vec, float *cost, float *grad,
                 void *workspace, size_t workspace_bytes, uint32_t *status, hipStream_t stream,
                 const float *add_grad, const float *add_cost, int add_S, float add_scale,
                 hipEvent_t add_ready, const float *mod_col_weights, const SeqLabels *labels) {
    if (ntrans > 62 || ncan > ntrans || ncan == 0) return 2;
    if (labels != nullptr && ((labels->seqs == nullptr && labels->total_len != 0) || labels->nbase == 0 ||
                              2 * labels->nbase * (labels->nbase + 1) != ncan ||
                              ((modidx != nullptr) != (labels->mod_cats != nullptr)) ||
                              (labels->mod_cats != nullptr && (labels->can_mods_offsets == nullptr || labels->mod_cat_weights == nullptr))))
        return 1;
    if (max_seqlen == 0) max_seqlen = nblk + 1;
    const CrfShape sh = crf_pick_shape(max_seqlen);
    if ((size_t)sh.R * sh.W * WAVE < max_seqlen || sh.R > 4) return 2;
    const bool mod = modidx != nullptr;
    // the linear path's block length for this sharpening factor; when the workspace the caller brought is
    // too small for it (sized without the factor: tk_crf_flipflop_workspace_bytes) but large enough for the
    // log-domain kernel on every read, that kernel does the call
    BandBlock blk = crf_band_pick_block(sharp_can, mod, max_seqlen, mod && mod_col_weights != nullptr, nblk,
                                        labels != nullptr ? labels->bulk_seqlen : 0);
    bool band = crf_pick_mode(ntrans, nblk, nbatch, max_seqlen, grad != nullptr, blk.bk) == CRF_BAND;
    // the second chance for what the batch's launch disowns (round 6); left out when the workspace the caller brought has no
    // room for it (sized by an older query): such reads go straight to the log-domain kernel, as in round 5
    BandBlock rblk = band ? crf_band_pick_retry(sharp_can, blk) : BandBlock{0, 0.f, 0};
    const size_t retry_slots = crf_band_retry_slots(nbatch);
    const BandLayout bl = crf_band_layout(ntrans, nblk, nbatch, max_seqlen, mod, grad != nullptr, blk.bk > 0 ? blk.bk : 8);
    const size_t band_bytes = band ? bl.total : 0;
    // (the tail launch has its own cells per lane: the retry's sweeps run side by side when 2 W waves fit its 16)
    const int tailR = crf_band_retry_R(max_seqlen);
    const CrfShape tsh = crf_tail_shape(tailR);
    const size_t tail_ckpt = crf_ckpt_bytes(nblk, retry_slots, tsh, grad != nullptr);
    size_t retry_bytes = (band && rblk.bk > 0) ? crf_band_layout(ntrans, nblk, retry_slots, max_seqlen, mod, true, rblk.bk, tailR).total : 0;
    if (band && tail_ckpt + band_bytes + retry_bytes > workspace_bytes) {
        rblk.bk = 0;
        retry_bytes = 0;
    }
    if (band && tail_ckpt + band_bytes > workspace_bytes)
        band = false;
    if (!band && crf_ckpt_bytes(nblk, nbatch, sh, grad != nullptr) > workspace_bytes) return 3;
    if (labels != nullptr && !band) {
        // the index arrays are OUTPUTS of this call; the band launch builds them itself, this path takes the
        // stand-alone kernel (tk_flipflop_build_indices_dev's)
        const int rc = build_indices_dispatch(labels->seqs, seqlen, nbatch, labels->nbase, labels->mod_cats,
                                              labels->can_mods_offsets, labels->mod_cat_weights, const_cast<int64_t *>(seqoff),
                                              const_cast<int32_t *>(stayidx), const_cast<int32_t *>(moveidx),
                                              const_cast<int32_t *>(modidx), const_cast<float *>(modfact), labels->total_len,
                                              status, stream);
        if (rc != 0) return rc;
    }
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
    a.grad_scale = grad_scale;
    a.grad_scale_vec = grad_scale_vec;
    a.add_grad = add_grad;
    a.add_cost = add_cost;
    a.add_S = add_S;
    a.add_scale = add_scale;
    a.cost = cost;
    a.grad = grad;
    a.codes = a.mod_cats = a.cmo = nullptr;
    a.mcw = nullptr;
    a.nbase = 0;
    a.status = status;
    char *wb = static_cast<char *>(workspace);
    // (what add_grad / add_cost hold may come from another stream: the band path waits between its sweeps
    // and its gradient pass, the single-launch form before it starts)
    if (add_ready != nullptr && !(band && grad != nullptr) && hipStreamWaitEvent(stream, add_ready, 0) != hipSuccess) return 4;
    if (band) {
        const bool g = grad != nullptr;
        const BandLayout l = bl;
        BandArgs b;
        b.lp = logprob;
        b.T = (int)nblk;
        b.N = (int)nbatch;
        b.S = (int)ntrans;
        b.ncan = (int)ncan;
        b.stay = stayidx;
        b.move = moveidx;
        b.mod = modidx;
        b.modfact = modfact;
        b.seqlen = seqlen;
        b.seqoff = seqoff;
        b.c_can = sharp_can * LOG2E;
        b.c_mod = sharp_mod * LOG2E;
        b.out_scale = out_scale;
        b.grad_scale = grad_scale;
        b.grad_scale_vec = grad_scale_vec;
        b.add_grad = add_grad;
        b.add_cost = add_cost;
        b.add_S = add_S;
        b.add_scale = add_scale;
        b.cost = cost;
        b.grad = grad;
        b.status = status;
        b.W = l.W;
        b.LP = (int)l.LP;
        b.Wp = (int)(l.LP / WAVE);
        b.ckFm = g ? reinterpret_cast<float *>(wb + l.ckFm) : nullptr;
        b.ckBm = g ? reinterpret_cast<float *>(wb + l.ckBm) : nullptr;
        b.ckFf = g ? reinterpret_cast<int16_t *>(wb + l.ckFf) : nullptr;
        b.ckBf = g ? reinterpret_cast<int16_t *>(wb + l.ckBf) : nullptr;
        b.ckFb = g ? reinterpret_cast<int *>(wb + l.ckFb) : nullptr;
        b.ckBb = g ? reinterpret_cast<int *>(wb + l.ckBb) : nullptr;
        b.bndF = g ? reinterpret_cast<float *>(wb + l.bndF) : nullptr;
        b.bndB = g ? reinterpret_cast<float *>(wb + l.bndB) : nullptr;
        b.scoreF = reinterpret_cast<double *>(wb + l.scoreF);
        b.scoreB = reinterpret_cast<double *>(wb + l.scoreB);
        b.rec = g ? reinterpret_cast<uint32_t *>(wb + l.rec) : nullptr;
        b.segend = g ? reinterpret_cast<int *>(wb + l.segend) : nullptr;
        b.gate = reinterpret_cast<int *>(wb + l.gate);
        b.gate2 = reinterpret_cast<int *>(wb + l.gate2);
        b.anygate = g ? reinterpret_cast<int *>(wb + l.anygate) : nullptr;
        b.zeros = reinterpret_cast<const float *>(wb + l.zeros);
        b.dbg = nullptr;
        b.before_gradient = add_ready;
        b.colw = mod ? mod_col_weights : nullptr;
        b.wbias = blk.wbias;
        b.klip = blk.klip;
        // (a batch of empty reads has no label array: any non-null pointer says "build here", nothing reads it)
        b.codes = labels != nullptr ? (labels->seqs != nullptr ? labels->seqs : stayidx) : nullptr;
        b.mod_cats = labels != nullptr ? labels->mod_cats : nullptr;
        b.cmo = labels != nullptr ? labels->can_mods_offsets : nullptr;
        b.mcw = labels != nullptr ? labels->mod_cat_weights : nullptr;
        b.total_len = labels != nullptr ? (long long)labels->total_len : 0;
        b.nbase = labels != nullptr ? (int)labels->nbase : 0;
        const int rc = crf_band_dispatch(b, l.R, mod, blk.bk, stream);
        if (rc != 0) return rc;
        if (TK_LAB_ENV("TK_CRF_GATE_DUMP")) {                       // lab: how many reads did the band path disown?
            (void)hipStreamSynchronize(stream);
            static int hostg[1 << 16];
            const size_t ng = nbatch < (1u << 16) ? nbatch : (1u << 16);
            (void)hipMemcpy(hostg, b.gate, ng * sizeof(int), hipMemcpyDeviceToHost);
            size_t cnt = 0, why[8] = {0};
            for (size_t i = 0; i < ng; ++i) {
                cnt += hostg[i] != 0;
                ++why[hostg[i] & 7];
            }
            fprintf(stderr, "crf band: %zu of %zu reads gated (non-finite score %zu, sweeps disagree %zu, row lost mass %zu)\n",
                    cnt, ng, why[1], why[4], why[2]);
            for (size_t i = 0, shown = 0; i < ng && shown < 16; ++i)
                if (hostg[i]) fprintf(stderr, "crf band:   read %zu (reason %d)\n", i, hostg[i]), ++shown;
        }
        // THE TAIL LAUNCH (round 6; crf_band.hip: crf_band_tail_kernel): the reads the batch's launch disowned, once more on
        // the linear path -- alone, 4-step blocks, steep frames -- and what that disowns too, redone in the log domain by the
        // same workgroup.  One launch; it finds nothing to do on the inputs a network produces.
        BandArgs c = b;
        c.gate = nullptr;
        c.gate2 = nullptr;
        c.anygate = nullptr;
        if (rblk.bk > 0) {
            const BandLayout q = crf_band_layout(ntrans, nblk, retry_slots, max_seqlen, mod, true, rblk.bk, tailR);
            char *wr = wb + l.total;
            c.W = q.W;
            c.LP = (int)q.LP;
            c.Wp = (int)(q.LP / WAVE);
            c.ckFm = reinterpret_cast<float *>(wr + q.ckFm);
            c.ckBm = reinterpret_cast<float *>(wr + q.ckBm);
            c.ckFf = reinterpret_cast<int16_t *>(wr + q.ckFf);
            c.ckBf = reinterpret_cast<int16_t *>(wr + q.ckBf);
            c.ckFb = reinterpret_cast<int *>(wr + q.ckFb);
            c.ckBb = reinterpret_cast<int *>(wr + q.ckBb);
            c.bndF = reinterpret_cast<float *>(wr + q.bndF);
            c.bndB = reinterpret_cast<float *>(wr + q.bndB);
            c.scoreF = reinterpret_cast<double *>(wr + q.scoreF);
            c.scoreB = reinterpret_cast<double *>(wr + q.scoreB);
            c.rec = reinterpret_cast<uint32_t *>(wr + q.rec);
            c.segend = reinterpret_cast<int *>(wr + q.segend);
            c.wbias = rblk.wbias;
            c.klip = rblk.klip;
        }
        // (the offsets and -- a call that brought index arrays -- the ids are the batch launch's; a launch that built its
        // ids from the labels left seqoff behind, and the tail forms its ids from the codes as well)
        BandRetry r;
        r.gate = b.gate;
        r.gate2 = b.gate2;
        r.firstF = g ? nullptr : b.scoreF;
        r.firstB = g ? nullptr : b.scoreB;
        r.first_wbias = blk.wbias;
        r.anygate = b.anygate;
        r.retry = rblk.bk > 0 ? 1 : 0;
        r.log_domain = 1;
        if (const char *e = TK_LAB_ENV("TK_CRF_NO_FALLBACK"))       // lab: time / test the linear path alone
            if (e[0] == '1') r.log_domain = 0;
        a.codes = b.codes;      // (ids from the labels wherever the band launch took them: it wrote no index array)
        a.mod_cats = b.mod_cats;
        a.cmo = b.cmo;
        a.mcw = b.mcw;
        a.nbase = b.nbase;
        wb += l.total + retry_bytes;
        {
            const int CK = crf_ck(tsh.R, tsh.W, mod ? 3 : 2);
            const size_t NK = (nblk + CK - 1) / CK;
            const size_t ckb = (retry_slots * NK * (size_t)tsh.R * tsh.W * WAVE * sizeof(float) + 255) / 256 * 256;
            a.ckpt = reinterpret_cast<float *>(wb);
            a.ckoff = reinterpret_cast<double *>(wb + (grad ? ckb : 0));
        }
        const int rr = crf_band_tail_dispatch(c, r, a, tailR, mod, retry_slots, stream);
        if (rr != 0) return rr;
        if (TK_LAB_ENV("TK_CRF_GATE_DUMP")) {
            (void)hipStreamSynchronize(stream);
            static int h1[1 << 16], h2[1 << 16];
            const size_t ng = nbatch < (1u << 16) ? nbatch : (1u << 16);
            (void)hipMemcpy(h1, b.gate, ng * sizeof(int), hipMemcpyDeviceToHost);
            (void)hipMemcpy(h2, b.gate2, ng * sizeof(int), hipMemcpyDeviceToHost);
            size_t tried = 0, kept = 0;
            for (size_t i = 0; i < ng; ++i) {
                tried += h2[i] != -1;
                kept += h2[i] == 0;
            }
            fprintf(stderr, "crf band tail (retry: bk %d, bias %.1f, slope %d): %zu reads taken, %zu kept on the linear path\n", rblk.bk, rblk.wbias, rblk.klip, tried, kept);
            for (size_t i = 0, shown = 0; i < ng && shown < 16; ++i)
                if (h2[i] > 0) fprintf(stderr, "crf band tail:   read %zu first %d retry %d\n", i, h1[i], h2[i]), ++shown;
        }
        return 0;
    }
    {
        const int CK = crf_ck(sh.R, sh.W, mod ? 3 : 2);
        const size_t NK = (nblk + CK - 1) / CK;
        const size_t ckb = (nbatch * NK * (size_t)sh.R * sh.W * WAVE * sizeof(float) + 255) / 256 * 256;
        a.ckpt = reinterpret_cast<float *>(wb);
        a.ckoff = reinterpret_cast<double *>(wb + (grad ? ckb : 0));
    }
    return mod ? crf_launch_mod<true>(sh, a, stream, nbatch) : crf_launch_mod<false>(sh, a, stream, nbatch);
}

}  // namespace tk
