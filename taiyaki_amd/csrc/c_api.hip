// c_api.hip -- the extern "C" boundary declared in include/taiyaki_amd_flipflop.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/taiyaki_amd_flipflop.h"
#include "crf_band.h"

namespace tk {
size_t logz_workspace_bytes(size_t T, size_t N, size_t nbase);
int logz_dispatch(const float *scores, size_t T, size_t N, size_t nbase, float *logz,
                  float *grad, void *workspace, size_t workspace_bytes, uint32_t *status,
                  hipStream_t stream, float *loss_acc = nullptr, float acc_scale = 0.f, float grad_scale = 1.f,
                  const float *grad_scale_vec = nullptr);
size_t viterbi_workspace_bytes(size_t T, size_t N, size_t nbase);
int viterbi_dispatch(const float *scores, size_t T, size_t N, size_t nbase, float *fwd,
                     int64_t *tb, int64_t *path, void *workspace, size_t workspace_bytes,
                     hipStream_t stream);
size_t crf_workspace_bytes(size_t ntrans, size_t nblk, size_t nbatch, size_t max_seqlen,
                           int want_grad);
size_t crf_workspace_bytes_sharp(size_t ntrans, size_t nblk, size_t nbatch, size_t max_seqlen,
                                 int want_grad, float sharp);
int crf_dispatch(const float *logprob, size_t ntrans, size_t nblk, size_t nbatch,
                 const int32_t *stayidx, const int32_t *moveidx, const int32_t *modidx,
                 const float *modfact, const int32_t *seqlen, const int64_t *seqoff,
                 size_t max_seqlen, size_t ncan, float sharp_can, float sharp_mod,
                 float out_scale, float grad_scale, const float *grad_scale_vec, float *cost, float *grad,
                 void *workspace, size_t workspace_bytes, uint32_t *status, hipStream_t stream,
                 const float *add_grad = nullptr, const float *add_cost = nullptr, int add_S = 0,
                 float add_scale = 0.f, hipEvent_t add_ready = nullptr, const float *mod_col_weights = nullptr,
                 const SeqLabels *labels = nullptr);
bool logz_side_stream(hipStream_t *s, hipEvent_t *fork, hipEvent_t *join);
#ifdef TK_LAB
void crf_band_lab_phase(int phase);
#endif
size_t beam_workspace_bytes(size_t T, size_t N, size_t nbase);
int lattice_dispatch(const float *scores, size_t T, size_t N, size_t nbase, int forward, const float *init,
                     float *out, float *total, hipStream_t stream);
int beam_dispatch(const float *scores, size_t T, size_t N, size_t nbase, int width, float beam_cut, int guided,
                  signed char *seq, int *seqlen, float *score, void *workspace, size_t workspace_bytes,
                  hipStream_t stream);
int grad_clip_dispatch(float *grads, const int64_t *seg_off, size_t nseg, size_t max_seg_len,
                       const float *thresh, float *maxs, hipStream_t stream);
int errprobs_dispatch(const float *trans, const int64_t *path, size_t T, size_t N, size_t nbase,
                      float *out, hipStream_t stream);
int build_indices_dispatch(const int32_t *seqs, const int32_t *seqlen, size_t nbatch,
                           size_t nbase, const int32_t *mod_cats,
                           const int32_t *can_mods_offsets, const float *mod_cat_weights,
                           int64_t *seqoff, int32_t *stay, int32_t *move, int32_t *mod,
                           float *fact, size_t total_len, uint32_t *status, hipStream_t stream);
int chunks_locate_dispatch(const tk_mapped_store *st, const int32_t *cand_read, const int32_t *cand_start,
                           const double *cand_frac, size_t ncand, size_t chunk_len,
                           const tk_chunk_filter *fp, uint8_t *reason, int32_t *dacstart,
                           int32_t *seqstart, int32_t *seqlen, int32_t *maxdwell, hipStream_t stream);
int chunks_select_dispatch(const uint8_t *reason, const int32_t *seqlen, size_t ncand, size_t nwant,
                           int32_t *sel, int64_t *seqoff, int32_t *counts, hipStream_t stream);
int chunks_gather_dispatch(const tk_mapped_store *st, const int32_t *cand_read, const int32_t *dacstart,
                           const int32_t *seqstart, const int32_t *seqlen, const int32_t *sel,
                           const int64_t *seqoff, const int32_t *counts, size_t nwant, size_t chunk_len,
                           int reverse, int standardize, size_t ncan, const int32_t *can_labels,
                           const int32_t *mod_labels, float *indata, int32_t *seqs, size_t seqs_cap,
                           int32_t *seqlens_out, int32_t *mod_cats, uint32_t *status, hipStream_t stream);
struct RemapArgs {
    const float *scores;
    const int64_t *row_off;
    const int32_t *stay_index;
    const int32_t *step_index;
    const int64_t *seq_off;
    const double *localpen;
    int K;
    double *score;
    int64_t *path;
    uint64_t *tb;
    const int64_t *tb_off;
};
int remap_dispatch(const RemapArgs &a, size_t nread, size_t max_M, hipStream_t stream);
int path_to_reftosignal_dispatch(const int64_t *path, const int64_t *path_off, const int64_t *ref_off,
                                 const int64_t *signalstart, const int64_t *siglen, int stride, size_t nread,
                                 int32_t *rts, hipStream_t stream);
}  // namespace tk

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" {

const char *tk_version(void) { return "taiyaki_amd flipflop gfx950 r3"; }

int tk_flipflop_build_indices_dev(const int32_t *seqs, const int32_t *seqlen, size_t nbatch,
                                  size_t total_len, size_t nbase, const int32_t *mod_cats,
                                  const int32_t *can_mods_offsets,
                                  const float *mod_cat_weights, int64_t *seqoff,
                                  int32_t *stayidx, int32_t *moveidx, int32_t *modidx,
                                  float *modfact, uint32_t *status, void *stream) {
    if (!seqlen || !seqoff || !stayidx || !moveidx || nbatch == 0 || nbase == 0) return TK_ERR_BAD_ARG;
    if (total_len > 0 && !seqs) return TK_ERR_BAD_ARG;
    if (mod_cats && (!can_mods_offsets || !mod_cat_weights || !modidx || !modfact)) return TK_ERR_BAD_ARG;
    return tk::build_indices_dispatch(seqs, seqlen, nbatch, nbase, mod_cats, can_mods_offsets,
                                      mod_cat_weights, seqoff, stayidx, moveidx, modidx,
                                      modfact, total_len, status, static_cast<hipStream_t>(stream));
}

int tk_grad_maxabs_clip_dev(float *grads, const int64_t *seg_off, size_t nseg, size_t max_seg_len,
                            const float *thresh, float *maxs, void *stream) {
    if (!grads || !seg_off || !maxs) return TK_ERR_BAD_ARG;
    const int rc = tk::grad_clip_dispatch(grads, seg_off, nseg, max_seg_len, thresh, maxs,
                                          static_cast<hipStream_t>(stream));
    return rc == 0 ? TK_OK : TK_ERR_LAUNCH;
}

int tk_flipflop_remap_dev(const float *scores, const int64_t *row_off, size_t ntrans,
                          const int32_t *stay_index, const int32_t *step_index, const int64_t *seq_off,
                          const double *localpen, size_t nread, size_t max_seqlen, double *score,
                          int64_t *path, uint64_t *traceback, const int64_t *tb_off, void *stream) {
    if (!row_off || !stay_index || !seq_off || !localpen || !score || !path || !tb_off || ntrans == 0)
        return TK_ERR_BAD_ARG;
    if (nread == 0) return TK_OK;
    if (max_seqlen > 1 && !step_index) return TK_ERR_BAD_ARG;
    if (nread > (size_t)INT32_MAX || ntrans > 4096) return TK_ERR_UNSUPPORTED;
    tk::RemapArgs a{scores, row_off, stay_index, step_index, seq_off, localpen, (int)ntrans, score, path,
                    traceback, tb_off};
    return tk::remap_dispatch(a, nread, max_seqlen, static_cast<hipStream_t>(stream));
}

int tk_remap_path_to_ref_to_signal_dev(const int64_t *path, const int64_t *path_off, const int64_t *ref_off,
                                       const int64_t *signalstart, const int64_t *siglen, size_t stride,
                                       size_t nread, int32_t *ref_to_signal, void *stream) {
    if (!path || !path_off || !ref_off || !signalstart || !siglen || !ref_to_signal || stride == 0)
        return TK_ERR_BAD_ARG;
    if (nread == 0) return TK_OK;
    if (nread > (size_t)INT32_MAX || stride > (size_t)INT32_MAX) return TK_ERR_UNSUPPORTED;
    return tk::path_to_reftosignal_dispatch(path, path_off, ref_off, signalstart, siglen, (int)stride, nread,
                                            ref_to_signal, static_cast<hipStream_t>(stream));
}

static bool store_ok(const tk_mapped_store *s) {
    return s && s->dacs && s->dacs_off && s->ref_to_signal && s->rts_off && s->reference && s->scaling &&
           s->mapped && s->nreads > 0;
}

int tk_chunks_locate_dev(const tk_mapped_store *store, const int32_t *cand_read, const int32_t *cand_start,
                         const double *cand_frac, size_t ncand, size_t chunk_len,
                         const tk_chunk_filter *filter, uint8_t *reason, int32_t *dacstart,
                         int32_t *seqstart, int32_t *seqlen, int32_t *maxdwell, void *stream) {
    if (!store_ok(store) || !filter || !cand_read || (!cand_start && !cand_frac) || !reason || !dacstart ||
        !seqstart || !seqlen || !maxdwell)
        return TK_ERR_BAD_ARG;
    if (ncand == 0) return TK_OK;
    if (ncand > (size_t)INT32_MAX || chunk_len > (size_t)INT32_MAX) return TK_ERR_UNSUPPORTED;
    return tk::chunks_locate_dispatch(store, cand_read, cand_start, cand_frac, ncand, chunk_len, filter, reason,
                                      dacstart, seqstart, seqlen, maxdwell, static_cast<hipStream_t>(stream));
}

int tk_chunks_select_dev(const uint8_t *reason, const int32_t *seqlen, size_t ncand, size_t nwant,
                         int32_t *sel, int64_t *seqoff, int32_t *counts, void *stream) {
    if (!seqoff || !counts || (nwant > 0 && !sel) || (ncand > 0 && (!reason || !seqlen))) return TK_ERR_BAD_ARG;
    if (ncand > (size_t)INT32_MAX || nwant > (size_t)INT32_MAX) return TK_ERR_UNSUPPORTED;
    return tk::chunks_select_dispatch(reason, seqlen, ncand, nwant, sel, seqoff, counts,
                                      static_cast<hipStream_t>(stream));
}

int tk_chunks_gather_dev(const tk_mapped_store *store, const int32_t *cand_read, const int32_t *dacstart,
                         const int32_t *seqstart, const int32_t *seqlen, const int32_t *sel,
                         const int64_t *seqoff, const int32_t *counts, size_t nwant, size_t chunk_len,
                         int reverse, int standardize, size_t ncan, const int32_t *can_labels,
                         const int32_t *mod_labels, float *indata, int32_t *seqs, size_t seqs_cap,
                         int32_t *seqlens, int32_t *mod_cats, uint32_t *status, void *stream) {
    if (!store_ok(store) || !cand_read || !dacstart || !seqstart || !seqlen || !sel || !seqoff || !counts ||
        !indata || !seqs || !seqlens || ncan == 0)
        return TK_ERR_BAD_ARG;
    if ((can_labels == nullptr) != (mod_labels == nullptr) || (mod_cats && !mod_labels)) return TK_ERR_BAD_ARG;
    if (nwant == 0 || chunk_len == 0) return TK_OK;
    if (nwant > 65535u * 16u || chunk_len > (size_t)INT32_MAX) return TK_ERR_UNSUPPORTED;
    return tk::chunks_gather_dispatch(store, cand_read, dacstart, seqstart, seqlen, sel, seqoff, counts, nwant,
                                      chunk_len, reverse, standardize, ncan, can_labels, mod_labels, indata,
                                      seqs, seqs_cap, seqlens, mod_cats, status,
                                      static_cast<hipStream_t>(stream));
}

int tk_flipflop_errprobs_dev(const float *trans, const int64_t *path, size_t nblk, size_t nbatch,
                             size_t nbase, float *errprobs, void *stream) {
    if (!trans || !path || !errprobs || nblk == 0 || nbatch == 0) return TK_ERR_BAD_ARG;
    if (!aligned16(trans)) return TK_ERR_BAD_ARG;
    const int rc = tk::errprobs_dispatch(trans, path, nblk, nbatch, nbase, errprobs,
                                         static_cast<hipStream_t>(stream));
    return rc == 0 ? TK_OK : (rc == 2 ? TK_ERR_UNSUPPORTED : TK_ERR_LAUNCH);
}

size_t tk_crf_flipflop_workspace_bytes(size_t ntrans, size_t nblk, size_t nbatch,
                                       size_t max_seqlen, int want_grad) {
    return tk::crf_workspace_bytes(ntrans, nblk, nbatch, max_seqlen, want_grad);
}

size_t tk_crf_flipflop_workspace_bytes_sharp(size_t ntrans, size_t nblk, size_t nbatch,
                                             size_t max_seqlen, int want_grad, float sharpfact) {
    return tk::crf_workspace_bytes_sharp(ntrans, nblk, nbatch, max_seqlen, want_grad, sharpfact);
}

static bool labels_ok(const tk_seq_labels *labels, tk::SeqLabels *out) {
    if (labels == nullptr || labels->nbase == 0 || (labels->seqs == nullptr && labels->total_len != 0)) return false;
    *out = tk::SeqLabels{labels->seqs, labels->total_len, labels->nbase, labels->mod_cats, labels->can_mods_offsets,
                         labels->mod_cat_weights, labels->bulk_seqlen};
    return true;
}

static int crf_flipflop_impl(const float *logprob, size_t ntrans, size_t nblk, size_t nbatch,
                             const int32_t *stayidx, const int32_t *moveidx, const int32_t *modidx,
                             const float *modfact, const int32_t *seqlen, const int64_t *seqoff,
                             size_t max_seqlen, size_t ncan, float sharp_can, float sharp_mod,
                             float out_scale, float *cost, float *grad, void *workspace,
                             size_t workspace_bytes, uint32_t *status, void *stream, const float *mod_col_weights,
                             const tk::SeqLabels *labels) {
    if (!logprob || !stayidx || !moveidx || !seqlen || !seqoff || !cost || !workspace) return TK_ERR_BAD_ARG;
    if (ntrans == 0 || nblk == 0 || nbatch == 0) return TK_ERR_BAD_ARG;
    if ((modidx == nullptr) != (modfact == nullptr)) return TK_ERR_BAD_ARG;
    return tk::crf_dispatch(logprob, ntrans, nblk, nbatch, stayidx, moveidx, modidx, modfact,
                            seqlen, seqoff, max_seqlen, ncan, sharp_can, sharp_mod, out_scale, 1.0f, nullptr,
                            cost, grad, workspace, workspace_bytes, status,
                            static_cast<hipStream_t>(stream), nullptr, nullptr, 0, 0.f, nullptr,
                            modidx != nullptr ? mod_col_weights : nullptr, labels);
}

int tk_crf_flipflop_dev(const float *logprob, size_t ntrans, size_t nblk, size_t nbatch,
                        const int32_t *stayidx, const int32_t *moveidx, const int32_t *modidx,
                        const float *modfact, const int32_t *seqlen, const int64_t *seqoff,
                        size_t max_seqlen, size_t ncan, float sharp_can, float sharp_mod,
                        float out_scale, float *cost, float *grad, void *workspace,
                        size_t workspace_bytes, uint32_t *status, void *stream, const float *mod_col_weights) {
    return crf_flipflop_impl(logprob, ntrans, nblk, nbatch, stayidx, moveidx, modidx, modfact, seqlen, seqoff, max_seqlen,
                             ncan, sharp_can, sharp_mod, out_scale, cost, grad, workspace, workspace_bytes, status, stream,
                             mod_col_weights, nullptr);
}

int tk_crf_flipflop_labels_dev(const float *logprob, size_t ntrans, size_t nblk, size_t nbatch,
                               const tk_seq_labels *labels, const int32_t *seqlen,
                               int64_t *seqoff, int32_t *stayidx, int32_t *moveidx, int32_t *modidx, float *modfact,
                               size_t max_seqlen, size_t ncan, float sharp_can, float sharp_mod,
                               float out_scale, float *cost, float *grad, void *workspace,
                               size_t workspace_bytes, uint32_t *status, void *stream) {
    tk::SeqLabels lab;
    if (!labels_ok(labels, &lab)) return TK_ERR_BAD_ARG;
    if ((labels->mod_cats != nullptr) != (modidx != nullptr)) return TK_ERR_BAD_ARG;
    // (modfact is filled from mod_cat_weights by column: the per-column form of the cat-mod kernels applies)
    return crf_flipflop_impl(logprob, ntrans, nblk, nbatch, stayidx, moveidx, modidx, modfact, seqlen, seqoff, max_seqlen,
                             ncan, sharp_can, sharp_mod, out_scale, cost, grad, workspace, workspace_bytes, status, stream,
                             labels->mod_cat_weights, &lab);
}

// rows of S floats -> their first S0 columns, contiguous (the canonical block of a cat-mod tensor)
__global__ void slice_cols_kernel(const float *__restrict__ src, float *__restrict__ dst, size_t nrows, int S, int S0) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * (size_t)S0) return;
    const size_t r = i / (size_t)S0;
    dst[i] = src[r * (size_t)S + (i - r * (size_t)S0)];
}

// Kernel B beside kernel A's sweeps.  The sweeps are one workgroup per CU issuing one instruction every ~5
// cycles: they leave the chip's memory system and most of its issue slots idle, and kernel B's three launches
// (~30 us at the train step's shape) do not depend on them.  With an aux buffer for kernel B's gradient, B runs
// on a second hardware queue (the side stream of logz_kernels.hip) while the sweeps run, and kernel A's
// gradient pass, which waits for it, folds logZ / nblk and (d logZ) / nblk into the rows it writes anyway.
// Measured (round 4, LABNOTES R4.6, profiles/r4_overlap_timeline.txt): the sweeps do not slow down (57.4 ->
// 58.7 us), the fork and the join cost ~7 + ~6 us of queue idle, the fold 3 us in A's gradient pass: loss path
// 136 -> 127-129 us (plain), 180 -> 157 us (cat-mod), 226 -> 210 us (T=1600).  Mode 1 (the default; TK_LOSS_OVERLAP=0
// or tk_flipflop_loss_overlap(0) turn it off): on unless the caller's stream is CAPTURING -- a captured
// train step replays the one-queue form: a captured fork / join replays correctly but 150 us SLOWER than the
// one-queue graph (384 against 234 us for the operator with its index preparation, tools/overlap_capture_probe.py,
// profiles/r4_overlap_capture_probe.txt; mode 2 keeps that experiment reachable).
static std::atomic<int> &loss_overlap_mode() {
    static std::atomic<int> mode([] {
        const char *e = getenv("TK_LOSS_OVERLAP");      // (the one variable the release build reads: once, here)
        return (e != nullptr && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1;
    }());
    return mode;
}
int tk_flipflop_loss_overlap(int mode) {
    std::atomic<int> &m = loss_overlap_mode();
    return (mode >= 0 && mode <= 2) ? m.exchange(mode) : m.load();
}
static bool loss_overlap_enabled(hipStream_t st) {
    const int mode = loss_overlap_mode().load();
    if (mode == 0) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (mode == 1 && st != nullptr && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return false;
    return true;
}
// (the side queue and its two events are one per device: calls from several host threads take turns enqueueing)
static std::mutex loss_overlap_mu;

size_t tk_flipflop_loss_fused_aux_bytes(size_t nblk, size_t nbatch, size_t nbase, size_t ntrans) {
    const size_t ncan = 2 * nbase * (nbase + 1);
    const size_t one = (nblk * nbatch * ncan * sizeof(float) + 255) / 256 * 256;
    if (ntrans <= ncan) return loss_overlap_enabled(nullptr) ? one : 0;    // plain CRF: in place (the experiment: kernel B's gradient)
    return 2 * one;                                             // canonical scores + their logZ gradient
}

static int loss_fused_impl(const float *scores, size_t nblk, size_t nbatch, size_t nbase, size_t ntrans,
                           const int32_t *stayidx, const int32_t *moveidx, const int32_t *modidx,
                           const float *modfact, const int32_t *seqlen,
                           const int64_t *seqoff, size_t max_seqlen, float sharpfact, float grad_scale,
                           const float *grad_scale_per_read, float *lossvector,
                           float *grad, float *logz, void *crf_workspace, size_t crf_workspace_bytes,
                           void *logz_workspace, size_t logz_workspace_bytes, void *aux, size_t aux_bytes,
                           uint32_t *status, void *stream, const float *mod_col_weights, const tk::SeqLabels *labels) {
    if (!scores || !stayidx || !moveidx || !seqlen || !seqoff || !lossvector || !grad || !logz || !crf_workspace ||
        !logz_workspace || nblk == 0 || nbatch == 0 || nbase == 0 || !(sharpfact > 0.f))
        return TK_ERR_BAD_ARG;
    if (!aligned16(scores) || !aligned16(grad)) return TK_ERR_BAD_ARG;
    if ((modidx == nullptr) != (modfact == nullptr)) return TK_ERR_BAD_ARG;
    const size_t ncan = 2 * nbase * (nbase + 1);
    const size_t one = (nblk * nbatch * ncan * sizeof(float) + 255) / 256 * 256;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool catmod = modidx != nullptr;
    if (!catmod && ntrans != ncan) return TK_ERR_BAD_ARG;
    if (catmod && (ntrans <= ncan || ntrans > 62)) return TK_ERR_BAD_ARG;
    if (catmod && (aux == nullptr || aux_bytes < 2 * one)) return TK_ERR_WORKSPACE;

    if (!catmod && (aux == nullptr || aux_bytes < one || !loss_overlap_enabled(st))) {
        // (A) first: per-read costs into `lossvector`, its gradient into `grad`; (B) then ADDS
        // logZ / nblk and (d logZ / d scores) / nblk in place -- in its posterior kernel, whose stores
        // are whole coalesced row sets
        int rc = tk::crf_dispatch(scores, ntrans, nblk, nbatch, stayidx, moveidx, nullptr, nullptr, seqlen, seqoff,
                                  max_seqlen, ntrans, sharpfact, sharpfact, 1.0f / sharpfact, grad_scale,
                                  grad_scale_per_read, lossvector, grad, crf_workspace, crf_workspace_bytes, status, st,
                                  nullptr, nullptr, 0, 0.f, nullptr, nullptr, labels);
        if (rc != 0) return rc;
        return tk::logz_dispatch(scores, nblk, nbatch, nbase, logz, grad, logz_workspace, logz_workspace_bytes, status,
                                 st, lossvector, 1.0f / (float)nblk, grad_scale, grad_scale_per_read);
    }
    // (B) into a compact gradient of its own -- for cat-mod (bin/train_flipflop.py:165-176) on a compact
    // copy of the canonical columns, which are not contiguous in the (T, N, ntrans) tensor --, (A) folds
    // logZ / nblk into its costs and (d logZ) / nblk, times the gradient multiplier, into the canonical
    // columns of the rows it writes anyway: one gradient tensor, no elementwise pass outside these kernels.
    float *x40 = catmod ? static_cast<float *>(aux) : nullptr;
    float *g40 = reinterpret_cast<float *>(static_cast<char *>(aux) + (catmod ? one : 0));
    hipStream_t sb = st;
    hipEvent_t fork = nullptr, join = nullptr;
    const bool side = loss_overlap_enabled(st) && tk::logz_side_stream(&sb, &fork, &join);
    std::unique_lock<std::mutex> turn(loss_overlap_mu, std::defer_lock);
    if (side) turn.lock();
    if (side) {
        if (hipEventRecord(fork, st) != hipSuccess || hipStreamWaitEvent(sb, fork, 0) != hipSuccess) return TK_ERR_LAUNCH;
    } else {
        sb = st;
    }
    if (catmod) {
        const size_t n = nblk * nbatch * ncan;
        hipLaunchKernelGGL(slice_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sb, scores, x40,
                           nblk * nbatch, (int)ntrans, (int)ncan);
        if (hipGetLastError() != hipSuccess) return TK_ERR_LAUNCH;
    }
    int rc = tk::logz_dispatch(catmod ? x40 : scores, nblk, nbatch, nbase, logz, g40, logz_workspace, logz_workspace_bytes,
                               status, sb);
    if (side && hipEventRecord(join, sb) != hipSuccess) rc = rc != 0 ? rc : TK_ERR_LAUNCH;
    if (rc != 0) {
        if (side) (void)hipStreamWaitEvent(st, join, 0);        // (never leave the side queue dangling in a capture)
        return rc;
    }
    // ctc.pyx:258-303: only the canonical columns are sharpened; cost / sharp
    rc = tk::crf_dispatch(scores, ntrans, nblk, nbatch, stayidx, moveidx, modidx, modfact, seqlen, seqoff, max_seqlen,
                          ncan, sharpfact, catmod ? 1.0f : sharpfact, 1.0f / sharpfact, grad_scale, grad_scale_per_read,
                          lossvector, grad, crf_workspace, crf_workspace_bytes, status, st, g40, logz, (int)ncan,
                          1.0f / (float)nblk, side ? join : nullptr, catmod ? mod_col_weights : nullptr, labels);
    if (rc != 0 && side) (void)hipStreamWaitEvent(st, join, 0);
    return rc;
}

int tk_flipflop_loss_fused_dev(const float *scores, size_t nblk, size_t nbatch, size_t nbase, size_t ntrans,
                               const int32_t *stayidx, const int32_t *moveidx, const int32_t *modidx,
                               const float *modfact, const int32_t *seqlen,
                               const int64_t *seqoff, size_t max_seqlen, float sharpfact, float grad_scale,
                               const float *grad_scale_per_read, float *lossvector,
                               float *grad, float *logz, void *crf_workspace, size_t crf_workspace_bytes,
                               void *logz_workspace, size_t logz_workspace_bytes, void *aux, size_t aux_bytes,
                               uint32_t *status, void *stream, const float *mod_col_weights) {
    return loss_fused_impl(scores, nblk, nbatch, nbase, ntrans, stayidx, moveidx, modidx, modfact, seqlen, seqoff, max_seqlen,
                           sharpfact, grad_scale, grad_scale_per_read, lossvector, grad, logz, crf_workspace,
                           crf_workspace_bytes, logz_workspace, logz_workspace_bytes, aux, aux_bytes, status, stream,
                           mod_col_weights, nullptr);
}

int tk_flipflop_loss_fused_labels_dev(const float *scores, size_t nblk, size_t nbatch, size_t ntrans,
                                      const tk_seq_labels *labels, const int32_t *seqlen,
                                      int64_t *seqoff, int32_t *stayidx, int32_t *moveidx, int32_t *modidx, float *modfact,
                                      size_t max_seqlen, float sharpfact, float grad_scale,
                                      const float *grad_scale_per_read, float *lossvector,
                                      float *grad, float *logz, void *crf_workspace, size_t crf_workspace_bytes,
                                      void *logz_workspace, size_t logz_workspace_bytes, void *aux, size_t aux_bytes,
                                      uint32_t *status, void *stream) {
    tk::SeqLabels lab;
    if (!labels_ok(labels, &lab)) return TK_ERR_BAD_ARG;
    if ((labels->mod_cats != nullptr) != (modidx != nullptr)) return TK_ERR_BAD_ARG;
    return loss_fused_impl(scores, nblk, nbatch, labels->nbase, ntrans, stayidx, moveidx, modidx, modfact, seqlen, seqoff,
                           max_seqlen, sharpfact, grad_scale, grad_scale_per_read, lossvector, grad, logz, crf_workspace,
                           crf_workspace_bytes, logz_workspace, logz_workspace_bytes, aux, aux_bytes, status, stream,
                           labels->mod_cat_weights, &lab);
}

typedef float tk_f4 __attribute__((ext_vector_type(4)));
// Measurement helper ("report the fraction against a measured device-copy ceiling"): a plain float4
// streaming copy, four independent 16-byte loads in flight per lane, one workgroup per 16 KiB -- the form the
// MI355X guide measured at 6.29 TB/s.  bench.py times it on the score tensor beside the logZ op.
__global__ __launch_bounds__(256) void devcopy_f4_kernel(const tk_f4 *__restrict__ src, tk_f4 *__restrict__ dst, size_t n4) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    tk_f4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + (size_t)k * 256 < n4) v[k] = __builtin_nontemporal_load(src + base + (size_t)k * 256);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + (size_t)k * 256 < n4) __builtin_nontemporal_store(v[k], dst + base + (size_t)k * 256);
}
int tk_devcopy_f32_dev(float *dst, const float *src, size_t n, void *stream) {
    if (dst == nullptr || src == nullptr || n % 4 != 0 || ((uintptr_t)dst | (uintptr_t)src) % 16 != 0) return TK_ERR_BAD_ARG;
    if (n == 0) return TK_OK;
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(devcopy_f4_kernel, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const tk_f4 *>(src), reinterpret_cast<tk_f4 *>(dst), n4);
    return hipGetLastError() == hipSuccess ? TK_OK : TK_ERR_LAUNCH;
}

#ifdef TK_LAB
// lab hook (lab build only; declared in tools/lab_api.h, not in the public header): see crf_band.hip
extern "C" void tk_lab_crf_band_phase(int phase) { tk::crf_band_lab_phase(phase); }
#endif

int tk_flipflop_lattice_dev(const float *scores, size_t nblk, size_t nbatch, size_t nbase, int forward,
                            const float *init, float *out, float *total, void *stream) {
    if (scores == nullptr || out == nullptr || total == nullptr) return 1;
    return tk::lattice_dispatch(scores, nblk, nbatch, nbase, forward, init, out, total,
                                static_cast<hipStream_t>(stream));
}

size_t tk_flipflop_beamsearch_workspace_bytes(size_t nblk, size_t nbatch, size_t nbase) {
    return tk::beam_workspace_bytes(nblk, nbatch, nbase);
}

int tk_flipflop_beamsearch_dev(const float *scores, size_t nblk, size_t nbatch, size_t nbase, int beam_width,
                               float beam_cut, int guided, int8_t *seq, int32_t *seqlen, float *score,
                               void *workspace, size_t workspace_bytes, void *stream) {
    if (!scores || !seq || !seqlen || !score || !workspace || nblk == 0 || nbatch == 0) return TK_ERR_BAD_ARG;
    return tk::beam_dispatch(scores, nblk, nbatch, nbase, beam_width, beam_cut, guided,
                             reinterpret_cast<signed char *>(seq), seqlen, score, workspace, workspace_bytes,
                             static_cast<hipStream_t>(stream));
}

size_t tk_flipflop_logz_workspace_bytes(size_t nblk, size_t nbatch, size_t nbase) {
    return tk::logz_workspace_bytes(nblk, nbatch, nbase);
}

int tk_flipflop_logz_dev(const float *scores, size_t nblk, size_t nbatch, size_t nbase,
                         float *logz, float *grad, void *workspace, size_t workspace_bytes,
                         uint32_t *status, void *stream) {
    if (!scores || !logz || !workspace || nblk == 0 || nbatch == 0) return TK_ERR_BAD_ARG;
    if (!aligned16(scores) || (grad && !aligned16(grad))) return TK_ERR_BAD_ARG;
    return tk::logz_dispatch(scores, nblk, nbatch, nbase, logz, grad, workspace,
                             workspace_bytes, status, static_cast<hipStream_t>(stream));
}

size_t tk_flipflop_viterbi_workspace_bytes(size_t nblk, size_t nbatch, size_t nbase) {
    return tk::viterbi_workspace_bytes(nblk, nbatch, nbase);
}

int tk_flipflop_viterbi_dev(const float *scores, size_t nblk, size_t nbatch, size_t nbase,
                            float *fwd, int64_t *traceback, int64_t *path, void *workspace,
                            size_t workspace_bytes, void *stream) {
    if (!scores || !path || !workspace || nbatch == 0) return TK_ERR_BAD_ARG;
    if (!aligned16(scores)) return TK_ERR_BAD_ARG;
    if ((fwd == nullptr) != (traceback == nullptr)) return TK_ERR_BAD_ARG;      // both or neither
    return tk::viterbi_dispatch(scores, nblk, nbatch, nbase, fwd, traceback, path, workspace,
                                workspace_bytes, static_cast<hipStream_t>(stream));
}

/* ------------------------------------------------------------------------- *
 * exact reference prototypes on HOST pointers
 * ------------------------------------------------------------------------- */
namespace {

struct DevBuf {
    void *p = nullptr;
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess; }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};

bool host_seq_call(float const *logprob, size_t ntrans, size_t nblk, size_t nbatch,
                   size_t const *moveidxs, size_t const *stayidxs, size_t const *modmoveidxs,
                   float const *modmovefacts, int32_t const *seqlen, float *score, float *grad) {
    // reference layout -> padded per-position layout (moves: seqidx[b] - b, c_crf_flipflop.c:479)
    std::vector<int64_t> off(nbatch + 1, 0);
    int32_t maxlen = 0;
    for (size_t b = 0; b < nbatch; ++b) {
        off[b + 1] = off[b] + seqlen[b];
        if (seqlen[b] > maxlen) maxlen = seqlen[b];
    }
    const size_t total = (size_t)off[nbatch];
    std::vector<int32_t> stay(total ? total : 1), move(total ? total : 1), mod;
    std::vector<float> fact;
    if (modmoveidxs) {
        mod.assign(total ? total : 1, 0);
        fact.assign(total ? total : 1, 0.f);
    }
    // Read b's moves start after the moves of the reads before it.  The caller (ctc.pyx:127-134)
    // concatenates max(L - 1, 0) move ids per read, i.e. the start is off[b] minus the number of
    // NON-EMPTY reads before b.  The reference C indexes off[b] - b (c_crf_flipflop.c:479-480),
    // which is the same number when no read is empty and otherwise reads its neighbours' slots
    // (before the array, even): this entry point follows the array the caller actually built.
    std::vector<size_t> mstart(nbatch + 1, 0);
    for (size_t b = 0; b < nbatch; ++b) mstart[b + 1] = mstart[b] + (seqlen[b] > 0 ? (size_t)seqlen[b] - 1 : 0);
    auto mbase = [&](size_t b) { return mstart[b]; };
    for (size_t b = 0; b < nbatch; ++b) {
        const size_t L = (size_t)seqlen[b];
        for (size_t p = 0; p < L; ++p) {
            stay[off[b] + p] = (int32_t)stayidxs[off[b] + p];
            if (p + 1 < L) {
                move[off[b] + p] = (int32_t)moveidxs[mbase(b) + p];
                if (modmoveidxs) {
                    mod[off[b] + p] = (int32_t)modmoveidxs[mbase(b) + p];
                    fact[off[b] + p] = modmovefacts[mbase(b) + p];
                }
            } else {
                move[off[b] + p] = 0;
            }
        }
    }
    const size_t nelt = nblk * nbatch * ntrans;
    const size_t wsb = tk::crf_workspace_bytes(ntrans, nblk, nbatch, (size_t)maxlen, grad != nullptr);
    DevBuf d_lp, d_stay, d_move, d_mod, d_fact, d_len, d_off, d_cost, d_grad, d_ws;
    if (!d_lp.alloc(nelt * 4) || !d_stay.alloc(stay.size() * 4) || !d_move.alloc(move.size() * 4) ||
        !d_len.alloc(nbatch * 4) || !d_off.alloc((nbatch + 1) * 8) || !d_cost.alloc(nbatch * 4) ||
        !d_ws.alloc(wsb))
        return false;
    if (grad && !d_grad.alloc(nelt * 4)) return false;
    if (modmoveidxs && (!d_mod.alloc(mod.size() * 4) || !d_fact.alloc(fact.size() * 4))) return false;
    bool ok = true;
    ok &= hipMemcpy(d_lp.p, logprob, nelt * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok &= hipMemcpy(d_stay.p, stay.data(), stay.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok &= hipMemcpy(d_move.p, move.data(), move.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok &= hipMemcpy(d_len.p, seqlen, nbatch * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok &= hipMemcpy(d_off.p, off.data(), (nbatch + 1) * 8, hipMemcpyHostToDevice) == hipSuccess;
    if (modmoveidxs) {
        ok &= hipMemcpy(d_mod.p, mod.data(), mod.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
        ok &= hipMemcpy(d_fact.p, fact.data(), fact.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) return false;
    // The C layer of the reference works on lp directly (no sharpening, no -1/nblk):
    // score = +path score, grad = +posterior (c_crf_flipflop.c:434-516).  The device
    // entry point returns cost = -score/nblk and d cost, so undo that here.
    // canonical columns = everything below the first mod column (ids of stay/move < ncan <= mod ids)
    size_t ncan = ntrans;
    if (modmoveidxs) {
        for (size_t b = 0; b < nbatch; ++b)
            for (size_t p = 0; p + 1 < (size_t)seqlen[b]; ++p)
                if (modmoveidxs[mbase(b) + p] < ncan) ncan = modmoveidxs[mbase(b) + p];
    }
    const int rc = tk::crf_dispatch(
        static_cast<const float *>(d_lp.p), ntrans, nblk, nbatch,
        static_cast<const int32_t *>(d_stay.p), static_cast<const int32_t *>(d_move.p),
        modmoveidxs ? static_cast<const int32_t *>(d_mod.p) : nullptr,
        modmoveidxs ? static_cast<const float *>(d_fact.p) : nullptr,
        static_cast<const int32_t *>(d_len.p), static_cast<const int64_t *>(d_off.p),
        (size_t)maxlen, ncan, 1.0f, 1.0f, 1.0f, 1.0f, nullptr, static_cast<float *>(d_cost.p),
        grad ? static_cast<float *>(d_grad.p) : nullptr, d_ws.p, wsb, nullptr, nullptr);
    if (rc != 0 || hipDeviceSynchronize() != hipSuccess) return false;
    std::vector<float> cost(nbatch);
    if (hipMemcpy(cost.data(), d_cost.p, nbatch * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
    for (size_t b = 0; b < nbatch; ++b) {
        if (grad != nullptr && seqlen[b] == 0) continue;     // score untouched (:458-464)
        score[b] = -cost[b] * (float)nblk;
    }
    if (grad) {
        if (hipMemcpy(grad, d_grad.p, nelt * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
        const float s = -(float)nblk;
        for (size_t i = 0; i < nelt; ++i) grad[i] *= s;
    }
    return true;
}

void fail_scores(float *score, size_t nbatch) {
    for (size_t b = 0; b < nbatch; ++b) score[b] = NAN;     // c_crf_flipflop.c:278-282
}

}  // namespace

void crf_flipflop_grad(float const *logprob, size_t ntrans, size_t nblk, size_t nbatch,
                       size_t const *moveidxs, size_t const *stayidxs, int32_t const *seqlen,
                       float *score, float *grad) {
    if (!host_seq_call(logprob, ntrans, nblk, nbatch, moveidxs, stayidxs, nullptr, nullptr, seqlen,
                       score, grad))
        fail_scores(score, nbatch);
}

void crf_flipflop_cost(float const *logprob, size_t ntrans, size_t nblk, size_t nbatch,
                       size_t const *moveidxs, size_t const *stayidxs, int32_t const *seqlen,
                       float *score) {
    if (!host_seq_call(logprob, ntrans, nblk, nbatch, moveidxs, stayidxs, nullptr, nullptr, seqlen,
                       score, nullptr))
        fail_scores(score, nbatch);
}

void cat_mod_flipflop_grad(float const *logprob, size_t ntrans, size_t nblk, size_t nbatch,
                           size_t const *moveidxs, size_t const *stayidxs,
                           size_t const *modmoveidxs, float const *modmovefacts,
                           int32_t const *seqlen, float *score, float *grad) {
    if (!host_seq_call(logprob, ntrans, nblk, nbatch, moveidxs, stayidxs, modmoveidxs,
                       modmovefacts, seqlen, score, grad))
        fail_scores(score, nbatch);
}

void cat_mod_flipflop_cost(float const *logprob, size_t ntrans, size_t nblk, size_t nbatch,
                           size_t const *moveidxs, size_t const *stayidxs,
                           size_t const *modmoveidxs, float const *modmovefacts,
                           int32_t const *seqlen, float *score) {
    if (!host_seq_call(logprob, ntrans, nblk, nbatch, moveidxs, stayidxs, modmoveidxs,
                       modmovefacts, seqlen, score, nullptr))
        fail_scores(score, nbatch);
}

}  // extern "C"
