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
    float *cost;                // (N)
    float *grad;                // (T, N, S) or null (cost only)
    uint32_t *status;
    int W;                      // chunks (= waves) per read
    int LP;                     // lattice row pitch = W * 64 * R
    float *latF, *latB;         // [N][T][LP]  forward column t / backward column t + 1 (band only)
    int *offF, *offB;           // [N][ceil(T/4)][W]  integer log2 offsets of (row group, chunk)
    double *scoreF, *scoreB;    // [N]  log2 scores of the two sweeps
    uint32_t *rec;              // [N][W][EPL][64]  sorted transition instances of every chunk
    float *recw;                // (cat-mod) their mod weights
    int *segend;                // [N][W][64]  end of every transition id's segment
    unsigned long long *dbg;    // lab builds only (TK_LAB_STAMPS): s_memtime stamps of workgroup 0, wave 0
};

struct BandLayout {
    int R, W;
    size_t LP;
    size_t latF, latB, offF, offB, scoreF, scoreB, rec, recw, segend, total;
};

bool crf_band_fits(size_t max_seqlen);
BandLayout crf_band_layout(size_t ntrans, size_t nblk, size_t nbatch, size_t max_seqlen, bool mod);
int crf_band_dispatch(const BandArgs &a, int R, bool mod, hipStream_t stream);

}  // namespace tk
