/*
 * taiyaki_amd_flipflop.h -- C ABI of the MI355X (gfx950) flip-flop CRF hot path.
 *
 * Drop-in boundary for the loss / decode operators of nanoporetech/taiyaki:
 *
 *   reference interface                                   replaced by
 *   ----------------------------------------------------  --------------------------------
 *   taiyaki/ctc/c_crf_flipflop.h:3-11   (libctc.pxd:5-13)   crf_flipflop_cost / _grad      (host ptrs, exact prototype)
 *   taiyaki/ctc/c_cat_mod_flipflop.h:3-13 (libctc.pxd:16-25) cat_mod_flipflop_cost / _grad  (host ptrs, exact prototype)
 *   taiyaki/ctc/ctc.pyx:116-153 FlipFlopCRF                 tk_crf_flipflop_dev            (device ptrs + stream)
 *   taiyaki/ctc/ctc.pyx:258-312 CatModFlipFlop              tk_crf_flipflop_dev (mod args) (device ptrs + stream)
 *   taiyaki/flipflopfings.py:6-31 + ctc.pyx:127-134,282-292 tk_flipflop_build_indices_dev
 *   taiyaki/cupy_extensions/flipflop.py:88-368 (fwd/bwd/    tk_flipflop_logz_dev
 *      make_trans/LogZ), taiyaki/layers.py:1277-1299
 *   taiyaki/cupy_extensions/flipflop.py:470-518,            tk_flipflop_viterbi_dev
 *      taiyaki/decode.py:75-115
 *   taiyaki/qscores.py:88-142 errprobs_from_trans           tk_flipflop_errprobs_dev
 *   bin/train_flipflop.py:201-212 apply_clipping            tk_grad_maxabs_clip_dev
 *   taiyaki/signal_mapping.py:515-557,676-716 + chunk_selection.py:29-95 +
 *   bin/train_flipflop.py:103-140 (chunk extraction, filters,
 *   batch stacking, flip-flop coding)                       tk_chunks_{locate,select,gather}_dev
 *   taiyaki/flipflop_remap.py:6-88 map_to_crf_viterbi       tk_flipflop_remap_dev
 *   taiyaki/signal_mapping.py:202-316 from_remapping_path   tk_remap_path_to_ref_to_signal_dev
 *
 * Conventions
 *  - plain C: pointers and sizes only, no torch / HIP types in the signatures
 *    (`stream` is a hipStream_t passed as void*; NULL = the default stream).
 *  - `_dev` entry points take DEVICE pointers, enqueue work on `stream` and return
 *    without synchronising.  The exact-prototype entry points take HOST pointers
 *    (like the reference), stage through the device and synchronise.
 *  - all tensors are C-contiguous fp32 (nblk, nbatch, ntrans) like the reference
 *    (c_crf_flipflop.c:434-516); base pointers of (nblk,nbatch,ntrans) tensors must
 *    be 16-byte aligned.
 *  - every `_dev` function returns an int status (TK_OK or TK_ERR_*); a launch
 *    failure is reported, never swallowed.  Non-finite results are reported
 *    through the optional device-side `status` word (bit flags below), which the
 *    Python shim turns into the reference's AssertionError (ctc.pyx:48,62-65,107-112).
 *  - ownership: the caller allocates every buffer including the workspace (size
 *    from the matching *_workspace_bytes); the library allocates nothing on the
 *    `_dev` path.
 *  - thread-safety: re-entrant; no global state.
 */
#ifndef TAIYAKI_AMD_FLIPFLOP_H
#define TAIYAKI_AMD_FLIPFLOP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TK_OK 0
#define TK_ERR_BAD_ARG 1        /* NULL pointer, bad shape, misaligned tensor          */
#define TK_ERR_UNSUPPORTED 2    /* nbase / ntrans / sequence length outside the build */
#define TK_ERR_WORKSPACE 3      /* workspace too small                                */
#define TK_ERR_LAUNCH 4         /* HIP launch / runtime failure                       */

/* bits of the device-side status word */
#define TK_STATUS_NONFINITE_SCORE 1u
#define TK_STATUS_NONFINITE_GRAD 2u
#define TK_STATUS_SEQS_OVERFLOW 4u      /* tk_chunks_gather_dev: seqs buffer too small */
#define TK_STATUS_SEQ_TOO_LONG 16u      /* CRF: a sequence longer than the max_seqlen the launch was sized for */
#define TK_STATUS_BAD_LABEL 8u          /* tk_flipflop_build_indices_dev: a flip-flop code outside
                                           [0, 2 nbase), a mod category outside its base's range, or
                                           sum(seqlen) > total_len (the reference asserts that move /
                                           stay indices lie in [0, ntrans), ctc.pyx:127-134) */
/* bits 8-31 of the status word COUNT, two fields of 12 bits (round 6; one of 24 before):
 *   bits  8-19  the reads the CRF's linear-domain path handed to its LOG-DOMAIN kernel (a read whose sweeps overflow,
 *               disagree, or whose posterior rows lose mass, also at the second try: right answers at ~1000x the cost
 *               per read);
 *   bits 20-31  the reads the batch's launch disowned and the RETRY launch swept again on the linear path, alone and at
 *               its conservative configuration (~2 reads' worth of a launch each); the first field counts those of them
 *               that failed there too.
 * Each call adds at most 4095 to a field.  They accumulate over calls like the flags do (a field that overflows carries
 * upwards: read them every few hundred calls); a trainer that sees the first grow is losing time, not accuracy
 * (taiyaki_amd.ctc.last_gate_count / last_retry_count, train.Trainer). */
#define TK_STATUS_FLAG_MASK 0xffu
#define TK_STATUS_GATED_SHIFT 8
#define TK_STATUS_RETRIED_SHIFT 20
#define TK_STATUS_COUNT_MASK 0xfffu

/* library / build identification: returns e.g. "taiyaki_amd flipflop gfx950 r1" */
const char *tk_version(void);

/* ------------------------------------------------------------------------- *
 * Index construction on device (flipflopfings.py:6-31, ctc.pyx:127-134, 282-292)
 *   seqs      (sum L)  int32 flip-flop codes 0..2nb-1, reads concatenated
 *   seqlen    (nbatch) int32
 *   mod_cats  (sum L)  int32 or NULL; can_mods_offsets (nbase+1) int32; mod_cat_weights (nbase+nmod) f32
 * Outputs (all in the PADDED per-position layout: entry off[n]+p belongs to
 * position p of read n; the move/mod entry of a read's last position is a
 * sentinel):
 *   seqoff (nbatch+1) int64 prefix offsets ; stayidx, moveidx (sum L) int32 ;
 *   modidx (sum L) int32, modfact (sum L) f32 (only when mod_cats != NULL)
 * Labels are range-checked: an offending code is clamped (so that no later kernel indexes out
 * of range) and TK_STATUS_BAD_LABEL is OR-ed into *status (nullable).
 * ------------------------------------------------------------------------- */
int tk_flipflop_build_indices_dev(const int32_t *seqs, const int32_t *seqlen,
                                  size_t nbatch, size_t total_len, size_t nbase,
                                  const int32_t *mod_cats,
                                  const int32_t *can_mods_offsets,
                                  const float *mod_cat_weights,
                                  int64_t *seqoff, int32_t *stayidx,
                                  int32_t *moveidx, int32_t *modidx,
                                  float *modfact, uint32_t *status, void *stream);

/* ------------------------------------------------------------------------- *
 * (A) sequence-constrained flip-flop CRF score and gradient
 *     (c_crf_flipflop.c:434-516, c_cat_mod_flipflop.c:493-582 semantics).
 *
 *   lp[t,n,i] = logprob[t,n,i] * (i < ncan ? sharp_can : sharp_mod)
 *   cost[n]   = -score(lp)/nblk * out_scale          (ctc.pyx:66,113,145)
 *   grad[t,n,i] = d(-score(lp)/nblk)/d lp[t,n,i]     (ctc.pyx:113; NULL => cost only,
 *                 in which case score is the forward score, c_crf_flipflop.c:255-290)
 *   seqlen[n]==0 => cost 0, zero gradient rows        (c_crf_flipflop.c:269-272,458-464)
 *   max_seqlen: an upper bound of seqlen (0 = unknown => nblk+1 is assumed).  It SIZES the launch -- cells per lane, waves
 *               per read, the workspace -- so a tight bound is also the smaller launch.  Round 6: the block length and the
 *               slope of the linear path's frames no longer follow it (round 5: one read beyond 0.78 nblk moved the whole
 *               batch to shorter blocks) but the batch's BULK (tk_seq_labels.bulk_seqlen below; unknown for the entry points
 *               that take index arrays): everybody runs the fast configuration, and the few reads it disowns are swept again
 *               alone by the TAIL launch -- 4-step blocks, frames of slope 20 -- and, only if that fails too, redone in the
 *               log domain by the same workgroup.  Results of a read are bit-for-bit the same in any batch launched with
 *               the same max_seqlen and configuration.
 *   mod_col_weights (nullable, cat-mod only; (ntrans - ncan) floats on the device): the caller's PROMISE
 *     that modfact[p] == mod_col_weights[modidx[p] - ncan] for every move, i.e. that the factor is a property
 *     of the modification column -- which is what `mod_cat_weights` of the reference's operator is
 *     (ctc.pyx:288-292; tk_flipflop_build_indices_dev fills modfact from it).  A move weight
 *     exp(sharp s[move] + modfact s[mod]) then factors into two GATHERS from one row that was exponentiated
 *     once per wave (lane = column, exponent multiplier per column), as in the plain CRF, instead of one
 *     exponential per lattice cell and step.  NULL: the general form (any per-position factors).
 * ------------------------------------------------------------------------- */
size_t tk_crf_flipflop_workspace_bytes(size_t ntrans, size_t nblk, size_t nbatch,
                                       size_t max_seqlen, int want_grad);
/* ... for a call with sharpening factor `sharpfact` (ctc.pyx:116-153, the --sharpen schedule of
 * bin/_bin_argparse.py:58-62): the linear-domain path takes sharpened scores with shorter time blocks
 * (factors up to 3.5), which keep more checkpoint columns.  Size the workspace with the factor the call
 * will carry.  A call whose workspace is too small for its factor's layout is done by the log-domain kernel on
 * every read IF the workspace holds that kernel's whole-batch checkpoint columns (it does at the train step's
 * shapes; at T = 4000 / N = 256 those are 8.0 GB against the linear path's 4.6) and returns TK_ERR 3
 * (workspace too small) otherwise -- never a wrong result.  (Round 6: the size includes the tail launch's slots --
 * a 4-step layout and the log-domain form's checkpoint columns for a sixteenth of the batch; a workspace that
 * holds the linear path's layout but not the retry's sends disowned reads straight to the log domain.) */
size_t tk_crf_flipflop_workspace_bytes_sharp(size_t ntrans, size_t nblk, size_t nbatch,
                                             size_t max_seqlen, int want_grad, float sharpfact);

/* The same operator WITHOUT the index-build launch (round 5): instead of the arrays tk_flipflop_build_indices_dev
 * made, hand over what that call takes -- the flip-flop codes (and, cat-mod, the modification categories and
 * tables) in a tk_seq_labels.  Every launch of the linear path then forms its ids from the codes itself (the
 * arithmetic of the build kernel, per cell; the codes are read-only inputs that stay in the L2s from call to call);
 * labels are range-checked as there (TK_STATUS_BAD_LABEL, offending codes clamped).  seqoff (nbatch + 1) is
 * written; stayidx / moveidx / modidx / modfact (total_len entries each) are SCRATCH: written only by calls the
 * linear path does not take (sharpening beyond 3.5, workspace-bound batches), which run the stand-alone build
 * kernel first -- same results either way.  Saves a launch per call: -1.5 .. -4 us of the op's ~100 at the train
 * step's shape (profiles/r5_index_build_ab.txt).  Cat-mod: a move's factor is mod_cat_weights BY COLUMN, so the
 * per-column form of the kernels (mod_col_weights above) applies. */
typedef struct tk_seq_labels {
    const int32_t *seqs;                /* (total_len) flip-flop codes 0 .. 2 nbase - 1, reads concatenated (device) */
    size_t total_len;
    size_t nbase;
    const int32_t *mod_cats;            /* (total_len) or NULL -- with the two tables below (device) */
    const int32_t *can_mods_offsets;    /* (nbase + 1) */
    const float *mod_cat_weights;       /* (nbase + nmod) */
    size_t bulk_seqlen;                 /* round 6.  0 = unknown.  Otherwise a length that all but a few reads of the batch
                                         * (a sixteenth, say) stay below: it picks the launch's block configuration where
                                         * max_seqlen only sizes it -- a batch whose BULK has narrow bands (beyond 0.78 nblk,
                                         * cat-mod 0.62) runs shorter blocks with steeper frames; a few long reads among
                                         * ordinary ones do not: they run the fast configuration with everybody else, and
                                         * whichever of them the linear path disowns is retried alone (TK_STATUS_RETRIED_SHIFT). */
} tk_seq_labels;
int tk_crf_flipflop_labels_dev(const float *logprob, size_t ntrans, size_t nblk, size_t nbatch,
                               const tk_seq_labels *labels, const int32_t *seqlen,
                               int64_t *seqoff, int32_t *stayidx, int32_t *moveidx, int32_t *modidx, float *modfact,
                               size_t max_seqlen, size_t ncan, float sharp_can, float sharp_mod,
                               float out_scale, float *cost, float *grad, void *workspace,
                               size_t workspace_bytes, uint32_t *status, void *stream);

int tk_crf_flipflop_dev(const float *logprob, size_t ntrans, size_t nblk,
                        size_t nbatch, const int32_t *stayidx,
                        const int32_t *moveidx, const int32_t *modidx,
                        const float *modfact, const int32_t *seqlen,
                        const int64_t *seqoff, size_t max_seqlen, size_t ncan,
                        float sharp_can, float sharp_mod, float out_scale,
                        float *cost, float *grad, void *workspace,
                        size_t workspace_bytes, uint32_t *status, void *stream,
                        const float *mod_col_weights);

/* ------------------------------------------------------------------------- *
 * Fused train-step loss (replaces the assembly in `calculate_loss`,
 * bin/train_flipflop.py:172-182: crf_flipflop_loss(outputs, ...) +
 * flipflop_logpartition(outputs) / nblk, two operators whose gradients autograd
 * then adds).  One call, plain CRF (ntrans = 2 nbase (nbase + 1)):
 *   lossvector (nbatch) = -score_A / (nblk sharp) + logZ / nblk
 *   grad (nblk, nbatch, ntrans) = d lossvector[n] / d scores[:, n, :]
 * Kernel A runs first (costs -> lossvector, its gradient -> grad); kernel B then
 * ADDS logZ / nblk and (d logZ / d scores) / nblk in place, the latter inside its
 * posterior kernel's coalesced row-set stores (one gradient tensor, no autograd
 * add).  `logz` (nbatch) receives the log-partition values.  Workspaces as for
 * the two separate entry points.
 * `grad_scale` x `grad_scale_per_read[n]` (nullable vector on the device: 1)
 * multiplies the gradient -- not the loss values: a caller that reduces
 * lossvector with known weights (`lossvector.mean()` of train_flipflop.py:182:
 * 1 / nbatch) receives the final d loss / d scores and needs no elementwise pass
 * over the tensor in its backward.  1, NULL = d lossvector[n] / d scores[:, n, :].
 * Cat-mod (`modidx`, `modfact` non-NULL, ntrans = 2 nbase (nbase + 1) + the mod
 * columns; ctc.pyx:258-312 + logZ of the canonical columns, train_flipflop.py:
 * 165-176): kernel B runs FIRST on a compact copy of the canonical columns (`aux`,
 * tk_flipflop_loss_fused_aux_bytes), kernel A folds logZ / nblk and its gradient
 * into what it writes.  Plain CRF: modidx = modfact = NULL, ntrans = 2 nbase
 * (nbase + 1), aux may be NULL (the one-queue form: A, then B adds in place).
 * TWO QUEUES: when `aux` has tk_flipflop_loss_fused_aux_bytes() bytes (non-zero for the plain CRF too unless
 * the overlap is off) and `stream` is not capturing, kernel B runs on a second hardware queue of the device
 * beside kernel A's sweeps -- forked from and joined to `stream`, so the call's ordering on `stream` is what
 * it is without -- and A's gradient pass folds B's results in.  A capturing `stream` gets the one-queue form.
 * tk_flipflop_loss_overlap(mode): 0 off, 1 on outside captures (default; environment TK_LOSS_OVERLAP=0|1|2
 * sets the initial mode), 2 lab (also inside captures); returns the previous mode, any other `mode` only queries.
 * ------------------------------------------------------------------------- */
int tk_flipflop_loss_overlap(int mode);
size_t tk_flipflop_loss_fused_aux_bytes(size_t nblk, size_t nbatch, size_t nbase, size_t ntrans);
int tk_flipflop_loss_fused_dev(const float *scores, size_t nblk, size_t nbatch, size_t nbase, size_t ntrans,
                               const int32_t *stayidx, const int32_t *moveidx,
                               const int32_t *modidx, const float *modfact,
                               const int32_t *seqlen, const int64_t *seqoff,
                               size_t max_seqlen, float sharpfact, float grad_scale,
                               const float *grad_scale_per_read, float *lossvector,
                               float *grad, float *logz, void *crf_workspace,
                               size_t crf_workspace_bytes, void *logz_workspace,
                               size_t logz_workspace_bytes, void *aux, size_t aux_bytes,
                               uint32_t *status, void *stream,
                               const float *mod_col_weights);
/* ... with the index build inside its first launch (see tk_crf_flipflop_labels_dev): nbase comes with the labels */
int tk_flipflop_loss_fused_labels_dev(const float *scores, size_t nblk, size_t nbatch, size_t ntrans,
                                      const tk_seq_labels *labels, const int32_t *seqlen,
                                      int64_t *seqoff, int32_t *stayidx, int32_t *moveidx, int32_t *modidx, float *modfact,
                                      size_t max_seqlen, float sharpfact, float grad_scale,
                                      const float *grad_scale_per_read, float *lossvector,
                                      float *grad, float *logz, void *crf_workspace, size_t crf_workspace_bytes,
                                      void *logz_workspace, size_t logz_workspace_bytes, void *aux, size_t aux_bytes,
                                      uint32_t *status, void *stream);

/* ------------------------------------------------------------------------- *
 * Hash beam search (replaces taiyaki/decodeutil/c_hashdecode.h:10
 * `flipflop_beamsearch` + the guiding backward pass c_flipflopfwdbwd.h
 * `flipflop_backward` + the wrapper decodeutil.pyx:9-51), a batch of reads per
 * launch, one wavefront per read:
 *   scores (nblk, nbatch, ntrans) device; beam_width <= 12 (reference default 5);
 *   beam_cut in [0, 1] (0 = no cutting); guided != 0 uses the backward scores.
 *   seq (nbatch, nblk) int8 flip-flop states of the best sequence, -1 padded;
 *   seqlen (nbatch); score (nbatch) = the reference's return value.
 * Records of exactly equal score are ordered as the reference's sort procedure
 * (decodeutil/qsort.h) orders them.  Returns 0, 1 (beam_cut outside [0, 1]),
 * 2 (nbase > 4, beam_width outside 1..12), 3 (workspace too small), 4 (launch).
 * ------------------------------------------------------------------------- */
size_t tk_flipflop_beamsearch_workspace_bytes(size_t nblk, size_t nbatch, size_t nbase);
int tk_flipflop_beamsearch_dev(const float *scores, size_t nblk, size_t nbatch, size_t nbase,
                               int beam_width, float beam_cut, int guided, int8_t *seq,
                               int32_t *seqlen, float *score, void *workspace,
                               size_t workspace_bytes, void *stream);

/* The decoder's lattice passes on their own (replace c_flipflopfwdbwd.h
 * `flipflop_forward` / `flipflop_backward` + decodeutil.pyx:54-108): unnormalised
 * forward (forward != 0) or backward scores of every block,
 *   out (nbatch, nblk + 1, 2 nbase), first (forward) resp. last (backward) row =
 *   init (nbatch, 2 nbase) or zeros when init is NULL; total (nbatch) = the
 *   reference's return value (log-sum-exp over the final row). */
int tk_flipflop_lattice_dev(const float *scores, size_t nblk, size_t nbatch, size_t nbase,
                            int forward, const float *init, float *out, float *total,
                            void *stream);

/* ------------------------------------------------------------------------- *
 * (B) log-partition over the 2*nbase state lattice and its gradient
 *     logz[n] = log sum_{all flip-flop paths starting in a flip state} exp(sum_t s)
 *     grad[t,n,:] = d logz[n] / d scores[t,n,:]  == posterior transition
 *     probabilities (decode.flipflop_make_trans); NULL => logZ only.
 * ------------------------------------------------------------------------- */
size_t tk_flipflop_logz_workspace_bytes(size_t nblk, size_t nbatch, size_t nbase);

int tk_flipflop_logz_dev(const float *scores, size_t nblk, size_t nbatch,
                         size_t nbase, float *logz, float *grad, void *workspace,
                         size_t workspace_bytes, uint32_t *status, void *stream);

/* ------------------------------------------------------------------------- *
 * Viterbi decode (decode.py:75-115: first-index tie rule, bit-exact fp32 adds)
 *   fwd (nblk+1, nbatch, 2nb) f32 ; traceback (nblk, nbatch, 2nb) int64 ;
 *   path (nblk+1, nbatch) int64.   fwd and traceback may BOTH be NULL (path only: what
 *   bin/basecall.py:222 consumes); one without the other is TK_ERR_BAD_ARG.
 * ------------------------------------------------------------------------- */
size_t tk_flipflop_viterbi_workspace_bytes(size_t nblk, size_t nbatch, size_t nbase);

int tk_flipflop_viterbi_dev(const float *scores, size_t nblk, size_t nbatch,
                            size_t nbase, float *fwd, int64_t *traceback,
                            int64_t *path, void *workspace,
                            size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- *
 * Per-block base error probabilities of a decoded path (qscores.py:88-142)
 *   trans (nblk, nbatch, 2nb(nb+1)) f32 posterior transition weights (what
 *   tk_flipflop_logz_dev writes as `grad`); path (nblk+1, nbatch) int64 flip-flop
 *   states; errprobs (nblk+1, nbatch) f32, row 0 = -1.
 *     p[t+1, n] = sum(trans into base path[t+1, n] % nb) / (sum(trans into any base) + 1e-10)
 *     errprobs  = 1 - p
 * ------------------------------------------------------------------------- */
int tk_flipflop_errprobs_dev(const float *trans, const int64_t *path, size_t nblk,
                             size_t nbatch, size_t nbase, float *errprobs, void *stream);

/* ------------------------------------------------------------------------- *
 * Gradient maxima + clip by value over a flat gradient arena
 * (bin/train_flipflop.py:201-212 apply_clipping, without its per-tensor host syncs)
 *   grads: all trainable gradients, contiguous; seg_off (nseg+1) int64 element offsets of
 *   the parameter tensors; max_seg_len = longest segment (grid sizing);
 *   maxs[s] = max |grads[seg s]| BEFORE clipping (NaN if the segment holds a NaN);
 *   thresh (nseg) nullable: where given (finite, >= 0) segment s is clamped to
 *   [-thresh[s], thresh[s]] -- identical to the reference's "clamp if max > thresh".
 * ------------------------------------------------------------------------- */
int tk_grad_maxabs_clip_dev(float *grads, const int64_t *seg_off, size_t nseg,
                            size_t max_seg_len, const float *thresh, float *maxs,
                            void *stream);

/* ------------------------------------------------------------------------- *
 * Training-chunk extraction from a mapped-signal set RESIDENT in device memory.
 * The arrays are the per-read datasets of the reference's mapped-signal format
 * (signal_mapping.py:26-33, docs/FILE_FORMATS.md:43-75), concatenated over reads.
 * ------------------------------------------------------------------------- */
typedef struct tk_mapped_store {
    const int16_t *dacs;            /* Dacs of all reads                                      */
    const int64_t *dacs_off;        /* (nreads + 1) element offsets into dacs                 */
    const int32_t *ref_to_signal;   /* Ref_to_signal of all reads (reflen_r + 1 entries each) */
    const int64_t *rts_off;         /* (nreads + 1) element offsets into ref_to_signal        */
    const int16_t *reference;       /* Reference of all reads; read r starts at rts_off[r] - r */
    const double *scaling;          /* nreads x 5: offset, range, digitisation, shift_frompA,
                                       scale_frompA                                           */
    const int32_t *mapped;          /* nreads x 2: get_mapped_dacs_region (signal_mapping.py:366-380) */
    size_t nreads;
} tk_mapped_store;

/* chunk_selection.py:9-26 FILTER_PARAMETERS; enabled = 0 is the reference's "median/mad/stride/
 * path_buffer is None" short circuit (signal_mapping.py:688-695) */
typedef struct tk_chunk_filter {
    int enabled;
    int model_stride;
    double filter_mean_dwell, filter_max_dwell, median_meandwell, mad_meandwell, path_buffer;
} tk_chunk_filter;

#define TK_CHUNK_NREASON 8  /* pass, emptysequence, emptysignal, tooshort, nullmapping, pathbuffer,
                               meandwell, maxdwell (signal_mapping.py:611-623)               */

/* SignalMapping.get_chunk_with_sample_length + Chunk.apply_filters for ncand candidates
 * (cand_read[c], start): start = cand_start[c] samples into the mapped region, or -- when
 * cand_start is NULL -- floor(cand_frac[c] * spare_length), cand_frac uniform in [0, 1)
 * (np.random.randint(spare_length), signal_mapping.py:541-542).  Outputs per candidate:
 * reason code, first sample of the chunk within the read, first base and number of bases of its
 * reference slice (0 unless accepted), maximum dwell.  All pointers are device pointers. */
int tk_chunks_locate_dev(const tk_mapped_store *store, const int32_t *cand_read,
                         const int32_t *cand_start, const double *cand_frac, size_t ncand,
                         size_t chunk_len, const tk_chunk_filter *filter, uint8_t *reason,
                         int32_t *dacstart, int32_t *seqstart, int32_t *seqlen,
                         int32_t *maxdwell, void *stream);

/* chunk_selection.sample_chunks' accept loop: sel[k] = index of the k-th accepted candidate in
 * draw order (k < nwant; -1 when fewer pass), seqoff (nwant + 1) = offsets of the chunks'
 * sequences in the concatenated sequence array, counts (TK_CHUNK_NREASON + 2) = rejection
 * histogram over the attempts the reference's loop makes, then number accepted, attempts. */
int tk_chunks_select_dev(const uint8_t *reason, const int32_t *seqlen, size_t ncand, size_t nwant,
                         int32_t *sel, int64_t *seqoff, int32_t *counts, void *stream);

/* bin/train_flipflop.py:103-140: indata (chunk_len, nwant, 1) float32 (columns beyond the
 * accepted count are zero), seqs = concatenated flip-flop coded sequences (int32, capacity
 * seqs_cap; TK_STATUS_SEQS_OVERFLOW in *status if that is too small), seqlens (nwant; 0 beyond
 * the accepted count).  reverse = the network reads the signal backwards (np.flip of signal
 * and labels).  can_labels / mod_labels (nullable, together with mod_cats) are the cat-mod
 * label maps of layers.py:1441-1460; ncan = number of canonical bases. */
int tk_chunks_gather_dev(const tk_mapped_store *store, const int32_t *cand_read,
                         const int32_t *dacstart, const int32_t *seqstart, const int32_t *seqlen,
                         const int32_t *sel, const int64_t *seqoff, const int32_t *counts,
                         size_t nwant, size_t chunk_len, int reverse, int standardize, size_t ncan,
                         const int32_t *can_labels, const int32_t *mod_labels, float *indata,
                         int32_t *seqs, size_t seqs_cap, int32_t *seqlens, int32_t *mod_cats,
                         uint32_t *status, void *stream);

/* ------------------------------------------------------------------------- *
 * Best path through a score matrix that spells a given sequence, for nread reads at once
 * (taiyaki/flipflop_remap.py:6-88 map_to_crf_viterbi; float64 like the reference).
 *   scores     rows of ntrans float32, all reads concatenated; read i owns rows
 *              row_off[i] .. row_off[i+1] (T_i of them)
 *   stay_index concatenated, M_i = seq_off[i+1] - seq_off[i] per read, starting at seq_off[i]
 *   step_index concatenated, M_i - 1 per read, starting at seq_off[i] - i
 *   localpen   (nread) score paid per block spent in the start / end state
 *   score      (nread) float64;  path: T_i + 1 int64 per read starting at row_off[i] + i,
 *              -1 = start / end state
 *   traceback  scratch, T_i * ceil(M_i / 64) 64-bit words per read starting at tb_off[i]
 * M_i >= 1; max_seqlen = max M_i <= 16384 (TK_ERR_UNSUPPORTED beyond). */
int tk_flipflop_remap_dev(const float *scores, const int64_t *row_off, size_t ntrans,
                          const int32_t *stay_index, const int32_t *step_index,
                          const int64_t *seq_off, const double *localpen, size_t nread,
                          size_t max_seqlen, double *score, int64_t *path, uint64_t *traceback,
                          const int64_t *tb_off, void *stream);

/* SignalMapping.from_remapping_path + get_reftosignal (taiyaki/signal_mapping.py:202-316):
 * Ref_to_signal of nread reads from their remapping paths.  path: concatenated, read i owns
 * entries path_off[i] .. path_off[i+1] (tk_flipflop_remap_dev's output layout: path_off[i] =
 * row_off[i] + i), -1 at the clipped ends and non-decreasing in between; entry k sits at signal
 * position k * stride - 1 + signalstart[i]; siglen[i] = length of the read's Dacs; reference
 * lengths from ref_off (nread + 1).  Output: reflen_i + 1 int32 per read starting at
 * ref_off[i] + i -- the tk_mapped_store layout. */
int tk_remap_path_to_ref_to_signal_dev(const int64_t *path, const int64_t *path_off,
                                       const int64_t *ref_off, const int64_t *signalstart,
                                       const int64_t *siglen, size_t stride, size_t nread,
                                       int32_t *ref_to_signal, void *stream);

/* ------------------------------------------------------------------------- *
 * Exact reference prototypes (HOST pointers; taiyaki/ctc/c_crf_flipflop.h:3-11,
 * c_cat_mod_flipflop.h:3-13).  Index arrays use the reference layout
 * (moves: max(L - 1, 0) entries per read, concatenated, as ctc.pyx:127-134 builds
 * them = sum(seqlen) - nbatch entries when no read is empty).  A batch with an
 * EMPTY read that is not last is where this entry point departs from the
 * reference C on purpose: the C indexes read b's moves at seqidx[b] - b
 * (c_crf_flipflop.c:479-480), one slot early per empty read in front, i.e. into
 * its neighbour's ids; here read b's moves start where the caller put them.
 * These stage through device memory, run the same kernels and synchronise; an
 * internal failure yields NAN scores (the reference's own out-of-memory
 * behaviour, c_crf_flipflop.c:278-282).
 * ------------------------------------------------------------------------- */
void crf_flipflop_grad(float const *logprob, size_t ntrans, size_t nblk,
                       size_t nbatch, size_t const *moveidxs,
                       size_t const *stayidxs, int32_t const *seqlen,
                       float *score, float *grad);

void crf_flipflop_cost(float const *logprob, size_t ntrans, size_t nblk,
                       size_t nbatch, size_t const *moveidxs,
                       size_t const *stayidxs, int32_t const *seqlen,
                       float *score);

void cat_mod_flipflop_grad(float const *logprob, size_t ntrans, size_t nblk,
                           size_t nbatch, size_t const *moveidxs,
                           size_t const *stayidxs, size_t const *modmoveidxs,
                           float const *modmovefacts, int32_t const *seqlen,
                           float *score, float *grad);

void cat_mod_flipflop_cost(float const *logprob, size_t ntrans, size_t nblk,
                           size_t nbatch, size_t const *moveidxs,
                           size_t const *stayidxs, size_t const *modmoveidxs,
                           float const *modmovefacts, int32_t const *seqlen,
                           float *score);

/* Measurement helper, no reference counterpart: a float4 streaming device-to-device copy of n floats (n % 4 == 0,
 * 16-byte aligned), the device-copy ceiling SURVEY.md 8(d) asks the roofline fraction to be quoted against
 * (bench.py: `roofline.copy_ceiling`). */
int tk_devcopy_f32_dev(float *dst, const float *src, size_t n, void *stream);

/* ------------------------------------------------------------------------- *
 * Multi-GPU: the data-parallel gradient all-reduce on RCCL over xGMI
 * (libtaiyaki_amd_rccl.so -- a library of its own, see csrc/rccl_api.cpp).
 * Replaces the collective inside the reference's DistributedDataParallel wrap
 * (bin/train_flipflop.py:255-268 process group, 384-397 all-reduce in backward)
 * and its checkpoint-file + barrier parameter hand-shake (380-392).
 * One process per GPU; rank 0 makes the unique id and hands its
 * tk_rccl_unique_id_bytes() bytes to the other ranks; every rank then calls
 * tk_rccl_comm_init (collective).  tk_allreduce_f32_dev: SUM in place over the
 * ranks, enqueued on `stream` (the 1 / nranks factor is the caller's);
 * tk_broadcast_f32_dev: rank `root`'s buffer to every rank.
 * ------------------------------------------------------------------------- */
size_t tk_rccl_unique_id_bytes(void);
int tk_rccl_unique_id(void *id_out, size_t bytes);
int tk_rccl_comm_init(void **comm_out, int nranks, const void *id_bytes, int rank);
/* The rendezvous of its own (round 5): what the reference gets from torch's TCP store at
 * MASTER_ADDR:MASTER_PORT (bin/train_flipflop.py:255-268), on plain sockets, no torch.distributed.
 * tk_rendezvous_bytes: rank 0 listens on addr:port and hands the `bytes` bytes at `buf` to each of its
 *   nranks - 1 peers; a peer connects (retrying until rank 0 is up) and receives them into `buf`.  Every
 *   rank returns once all peers have the bytes; TK_ERR_BAD_ARG for bad arguments, TK_ERR_LAUNCH (4) when
 *   `timeout_ms` passes first, a peer announces another nranks, or a rank shows up twice.  No GPU involved.
 * tk_rccl_comm_init_rendezvous: unique id on rank 0 + tk_rendezvous_bytes + tk_rccl_comm_init -- the one
 *   call a C host makes per process. */
int tk_rendezvous_bytes(const char *addr, int port, int rank, int nranks, void *buf, size_t bytes, int timeout_ms);
int tk_rccl_comm_init_rendezvous(void **comm_out, const char *addr, int port, int rank, int nranks, int timeout_ms);
int tk_allreduce_f32_dev(void *comm, float *buf, size_t n, void *stream);
int tk_broadcast_f32_dev(void *comm, float *buf, size_t n, int root, void *stream);
int tk_rccl_comm_destroy(void *comm);

#ifdef __cplusplus
}
#endif
#endif /* TAIYAKI_AMD_FLIPFLOP_H */
