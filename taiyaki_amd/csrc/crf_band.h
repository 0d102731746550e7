// crf_band.h -- internal interface of kernel A's band mode (crf_band.hip), used by the
// launcher in crf_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace tk {

struct BandArgs {
    const float *lp;            // (T, N, S) scores (unsharpened)
    int T, N, S;
    int ncan;                   // canonical transition columns (== S for the plain CRF)
    const int32_t *stay;        // padded per-position ids (tk_flipflop_build_indices_dev)
    const int32_t *move;
    const int32_t *mod;         // nullable
    const float *modfact;       // nullable
    const int32_t *seqlen;      // (N)
    const int64_t *seqoff;      // (N + 1)
    float c_can;                // sharp_can * log2(e)
    float c_mod;                // sharp_mod * log2(e)
    float out_scale;            // cost multiplier (1 / sharpfact)
    float grad_scale;           // gradient multiplier (1 for the reference's operators)
    const float *grad_scale_vec;    // nullable; (N): a further per-read multiplier
    // fused cat-mod loss: kernel B ran first into a compact buffer; this operator adds
    // add_scale * add_cost[n] to the cost and add_scale * (gradient multiplier) * add_grad[t][n][s]
    // (s < add_S) to the gradient it writes.  Null: nothing to add.
    const float *add_grad;      // (T, N, add_S)
    const float *add_cost;      // (N)
    int add_S;
    float add_scale;
    float *cost;                // (N)
    float *grad;                // (T, N, S) or null (cost only)
    uint32_t *status;
    int W;                      // chunks (= waves) per read
    int LP;                     // checkpoint row pitch = W * 64 * R
    int Wp;                     // 64-cell chunks per checkpoint row = LP / 64 (the gradient pass's chunks)
    // one checkpoint column per time block and sweep: cell = m * 2^f
    float *ckFm, *ckBm;         // [N][NB][LP]  mantissas (forward: column 8 j; backward: column 8 j + nvalid)
    int16_t *ckFf, *ckBf;       // [N][NB][LP]  frames (exponents), fixed for the block: 16-bit offsets from ...
    int *ckFb, *ckBb;           // [N][NB][W]   ... a base per (sweep chunk, block)
    float *bndF, *bndB;         // [N][NB][Wp][8]  forward: the LAST cell of a 64-cell chunk before every step of the
                                // block; backward: its FIRST cell
    double *scoreF, *scoreB;    // [N]  log2 scores of the two sweeps
    uint32_t *rec;              // [N][Wp][KINDS][64]  sorted transition instances of every 64-cell chunk
    int *segend;                // [N][Wp][64]  end of every transition id's segment
    const float *zeros;         // 64 B of zeros (the boundary row of lanes that take none)
    int *gate;                  // [N]  != 0: the batch's launch disowns this read (retried, and if need be redone in the log domain, by crf_band_tail_kernel)
    int *anygate;               // one word, nullable: != 0 iff the batch's gradient pass disowned ANY read (the tail launch looks
                                // at this word first: on the common path it leaves without a pass over the gate array)
    int *gate2;                 // [N]  nullable; the retry launch's verdicts: the batch's launch sets -1 ("not retried"), the retry
                                // launch 0 (it owns the read) or why not
    unsigned long long *dbg;    // lab builds only (TK_LAB_STAMPS)
    const float *colw;          // nullable (cat-mod): (S - ncan) per-COLUMN factors; promise that modfact[p] = colw[mod[p] - ncan]
    float wbias;                // every step weight carries 2^-wbias (crf_band.hip: BK_MAX); the scores get wbias T back
    int klip;                   // the frames' slope along the flow, bits per cell (crf_band.hip: KLIP; BandBlock::klip)
    hipEvent_t before_gradient; // host side: the stream waits for this event between the sweeps and the gradient
                                // pass (what `add_grad` / `add_cost` hold was produced on another stream); null: none
    // Round 5 -- the index build INSIDE the sweep launch (codes != null): stay / move / mod / modfact / seqoff above
    // are then OUTPUTS of this launch, not inputs.  Every sweep workgroup forms its read's offset and its cells'
    // ids from the flip-flop codes itself; the rank workgroups (a cost-only call: the forward sweeps) also write
    // the arrays for the launches behind (gradient pass, tail launch).  Saves tk_flipflop_build_indices_dev's
    // launch: ~5 us of the op's ~99 at the train step's shape.
    const int32_t *codes;       // (total_len) flip-flop codes 0 .. 2 nbase - 1, reads concatenated; null: ids are inputs
    const int32_t *mod_cats;    // (total_len) or null
    const int32_t *cmo;         // can_mods_offsets (nbase + 1)
    const float *mcw;           // mod_cat_weights (nbase + nmod)
    long long total_len;
    int nbase;
};

// what tk_crf_flipflop_labels_dev / tk_flipflop_loss_fused_labels_dev hand down (include/taiyaki_amd_flipflop.h: tk_seq_labels)
struct SeqLabels {
    const int32_t *seqs;
    size_t total_len;
    size_t nbase;
    const int32_t *mod_cats;
    const int32_t *can_mods_offsets;
    const float *mod_cat_weights;
    size_t bulk_seqlen;         // 0: unknown.  A length that all but a few reads of the batch stay below (crf_band_pick_block)
};

// Round 6 -- the per-read SECOND CHANCE on the linear path.  The batch's launch runs the fast configuration for everybody;
// the reads it disowns are swept again by crf_band_retry_kernel -- one workgroup per read, the conservative configuration
// (4-step blocks, frames of slope up to 20: crf_band_pick_retry), sweeps and gradient pass one after the other in that
// workgroup, in a workspace of a few slots -- and only what THAT disowns goes to the log-domain kernel.
struct BandRetry {
    const int *gate;            // [N]  the batch launch's verdicts (cost-only calls: 2 = pending, see firstF)
    int *gate2;                 // [N]  out: 0 the retry owns the read (cost / gradient rows written), else why not
    const double *firstF, *firstB;  // cost-only calls: the batch launch's two sweep scores (a pending read is retried iff they
                                    // are not finite or disagree); null for gradient calls
    float first_wbias;              // ... and that launch's weight bias: a pending read whose scores agree gets its cost here
    const int *anygate;             // gradient calls: see BandArgs::anygate; null: look at the gate array
    int retry;                      // 0: no retry configuration for this call -- disowned reads go straight to the log domain
    int log_domain;                 // 1; lab builds: 0 = leave what the linear path disowned alone (TK_CRF_NO_FALLBACK)
};

struct BandBlock {
    int bk;                     // time steps per block: 4, 8 or 12 (0: not for the linear path)
    float wbias;
    int klip;                   // frame slope (bits per cell): 6; 11 on 8-step biased blocks for batches with narrow bands
};

struct BandLayout {
    int R, W, BK;
    size_t LP;
    size_t ckFm, ckBm, ckFf, ckBf, ckFb, ckBb, bndF, bndB, scoreF, scoreB, rec, segend, gate, gate2, anygate, zeros, total;
};

bool crf_band_fits(size_t max_seqlen);
// `bulk_seqlen`: what picks the configuration -- a length all but a few reads stay below (0: unknown = the fast configuration;
// the few beyond it are retried one by one); `max_seqlen` sizes the launch
BandBlock crf_band_pick_block(float sharp, bool mod, size_t max_seqlen, bool colw = false, size_t nblk = 0, size_t bulk_seqlen = 0);
// the retry launch's configuration for this call (bk = 0: none -- the batch's launch already ran it, or nothing milder exists)
BandBlock crf_band_pick_retry(float sharp, BandBlock fast);
size_t crf_band_retry_slots(size_t nbatch);
// `force_R` > 0: that many cells per lane instead of the batch launch's choice (the retry's own layout: crf_band_retry_R)
BandLayout crf_band_layout(size_t ntrans, size_t nblk, size_t nbatch, size_t max_seqlen, bool mod,
                           bool want_grad, int bk, int force_R = 0);
// cells per lane of the RETRY: the smallest that leaves 2 W <= 16 waves, so that the tail launch's workgroup runs both sweeps at once
int crf_band_retry_R(size_t max_seqlen);
int crf_band_dispatch(const BandArgs &a, int R, bool mod, int bk, hipStream_t stream);
// (the tail launch -- retry, then the log domain: crf_log.h: crf_band_tail_dispatch)

}  // namespace tk
