// chunk_kernels.hip -- training-chunk extraction and batch assembly on the device (gfx950).
//
// Replaces the per-batch host work of the reference's trainer, one Python object per chunk:
//   taiyaki/signal_mapping.py:515-557  SignalMapping.get_chunk_with_sample_length
//   taiyaki/signal_mapping.py:382-423  get_reference_locations (two searchsorted calls)
//   taiyaki/signal_mapping.py:463-513  get_current, _get_chunk (max dwell)
//   taiyaki/signal_mapping.py:676-716  Chunk.apply_filters
//   taiyaki/chunk_selection.py:29-95   sample_chunks (accept until enough, rejection counts)
//   bin/train_flipflop.py:103-140      np.vstack(...).T, flipflop_code, concatenation, H2D copy
// MI355X-first: the whole mapped-signal training set (int16 Dacs, int32 Ref_to_signal, int16
// Reference, five floats per read) stays RESIDENT in HBM (288 GB holds ~10^11 samples), a batch
// is three small launches, and what comes out is already the loss path's input: the
// (chunk_len, nbatch, 1) float32 signal tensor and the concatenated flip-flop coded sequences.
//   1. chunk_locate   one wave per candidate (read, start): mapped region, the two binary
//                     searches, wave-parallel max dwell, the three filters -> reason code
//   2. chunk_select   one block: the first `nwant` accepted candidates in draw order, the
//                     sequence offsets (exclusive scan), the rejection histogram over the
//                     attempts the reference's loop would have made
//   3. chunk_gather   signal: 64 x 16 (sample, chunk) tiles through LDS -- int16 reads along the
//                     signal, float32 writes along the batch, both coalesced; the float64
//                     arithmetic of get_current in the reference's order, rounded once to
//                     float32 like torch.tensor(..., dtype=float32).  Sequence: one block per
//                     chunk, flip/flop by the parity of the run position.
// HBM-bound and tiny: 6 B per sample (2 in, 4 out).
#include "ff_common.h"
#include "../../include/taiyaki_amd_flipflop.h"

#pragma clang fp contract(off)

namespace tk {

// reason codes, in the order of oracle/chunks.py REASONS (signal_mapping.py:611-623)
enum : uint8_t { CH_PASS = 0, CH_EMPTY_SEQ, CH_EMPTY_SIG, CH_SHORT, CH_NULL_MAP, CH_PATH_BUFFER,
                 CH_MEAN_DWELL, CH_MAX_DWELL, CH_NREASON };
constexpr double CH_TINY = 0.00000001;      // signal_mapping.py:609

// np.searchsorted(a[0:n], v, side): first index with a[i] > v (right) / a[i] >= v (left)
template <bool RIGHT>
__device__ __forceinline__ int search_sorted(const int32_t *__restrict__ a, int n, int v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int x = a[mid];
        if (RIGHT ? (x <= v) : (x < v)) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, WAVE));
    return v;
}

constexpr int LOC_WAVES = 4;

__global__ __launch_bounds__(LOC_WAVES *WAVE) void chunk_locate_kernel(
    tk_mapped_store st, const int32_t *__restrict__ cand_read, const int32_t *__restrict__ cand_start,
    const double *__restrict__ cand_frac, int ncand, int chunk_len, tk_chunk_filter fp,
    uint8_t *__restrict__ reason, int32_t *__restrict__ dacstart, int32_t *__restrict__ seqstart,
    int32_t *__restrict__ seqlen, int32_t *__restrict__ maxdwell) {
    const int c = blockIdx.x * LOC_WAVES + (threadIdx.x >> 6);
    if (c >= ncand) return;                     // wave-uniform
    const int lane = lane_id();
    const int r = cand_read[c];
    const int32_t *rts = st.ref_to_signal + st.rts_off[r];
    const int nrts = (int)(st.rts_off[r + 1] - st.rts_off[r]);
    const int lo = st.mapped[2 * r], hi = st.mapped[2 * r + 1];
    const int spare = hi - lo - chunk_len;      // signal_mapping.py:534-536
    uint8_t why = CH_PASS;
    int a = 0, r0 = 0, r1 = 0, mx = 1;
    int start = 0;
    if (spare > 0) {
        // start_sample: given, or the fraction of the spare length a uniform draw selects
        start = cand_start ? cand_start[c] : min((int)(cand_frac[c] * (double)spare), spare - 1);
    }
    if (spare <= 0 || start >= spare || start < 0) {
        why = CH_SHORT;
    } else {
        a = start + lo;
        const int b = a + chunk_len;
        if (a < lo || b > hi) {                 // :402-405 -> :553-556
            why = CH_NULL_MAP;
        } else {
            r0 = search_sorted<true>(rts, nrts, a) - 1;         // :413-415
            r1 = search_sorted<false>(rts, nrts, b);            // :418-419
            if (r1 == r0) why = CH_EMPTY_SEQ;                   // :500-503
            else if (chunk_len == 0) why = CH_EMPTY_SIG;
        }
    }
    if (why == CH_PASS) {
        // max of np.diff(Ref_to_signal[r0:r1]) (:507-513), 1 when the region holds one base
        int m = INT32_MIN;
        for (int i = r0 + lane; i + 1 < r1; i += WAVE) m = max(m, rts[i + 1] - rts[i]);
        m = wave_max_i32(m);
        mx = (r1 - r0 > 1) ? m : 1;
        if (fp.enabled) {                       // :688-716, Python float arithmetic = double
            const double sig_len = (double)chunk_len, seq_len = (double)(r1 - r0);
            const double mean_dwell = sig_len / (seq_len + CH_TINY);
            if (sig_len / (seq_len * (double)fp.model_stride) <= fp.path_buffer) why = CH_PATH_BUFFER;
            else if (fabs(mean_dwell - fp.median_meandwell) > fp.filter_mean_dwell * fp.mad_meandwell)
                why = CH_MEAN_DWELL;
            else if ((double)mx > fp.filter_max_dwell * fp.median_meandwell) why = CH_MAX_DWELL;
        }
    }
    if (lane == 0) {
        reason[c] = why;
        dacstart[c] = a;
        seqstart[c] = r0;
        seqlen[c] = (why == CH_PASS) ? r1 - r0 : 0;
        maxdwell[c] = mx;
    }
}

// ---------------------------------------------------------------------------
// 2. first `nwant` accepted candidates, sequence offsets, rejection histogram
// ---------------------------------------------------------------------------
constexpr int SEL_THREADS = 1024;

// exclusive prefix sum of one value per thread over the block; returns the prefix, *total = sum
__device__ __forceinline__ long long block_exclusive_scan(long long v, long long *lds, long long *total) {
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    long long inc = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const long long o = __shfl_up(inc, d, WAVE);
        if (lane >= d) inc += o;
    }
    __syncthreads();                            // lds may still be read from the previous call
    if (lane == WAVE - 1) lds[wave] = inc;
    __syncthreads();
    long long before = 0, all = 0;
    for (int w = 0; w < SEL_THREADS / WAVE; ++w) {
        const long long t = lds[w];
        if (w < wave) before += t;
        all += t;
    }
    *total = all;
    return before + inc - v;
}

__global__ __launch_bounds__(SEL_THREADS) void chunk_select_kernel(
    const uint8_t *__restrict__ reason, const int32_t *__restrict__ seqlen, int ncand, int nwant,
    int32_t *__restrict__ sel, int64_t *__restrict__ seqoff, int32_t *__restrict__ counts) {
    __shared__ long long part[SEL_THREADS / WAVE];
    __shared__ int hist[CH_NREASON];
    __shared__ int attempts_s;
    const int tid = threadIdx.x;
    if (tid < CH_NREASON) hist[tid] = 0;
    if (tid == 0) attempts_s = ncand;
    for (int k = tid; k < nwant; k += SEL_THREADS) sel[k] = -1;
    __syncthreads();
    // accepted candidates, in draw order; the loop of chunk_selection.py:77-93 stops after the
    // attempt that yields the nwant-th accepted chunk
    long long running = 0;
    for (int base = 0; base < ncand; base += SEL_THREADS) {
        const int c = base + tid;
        const int ok = (c < ncand && reason[c] == CH_PASS) ? 1 : 0;
        long long tot;
        const long long rank = running + block_exclusive_scan(ok, part, &tot);
        if (ok && rank < nwant) {
            sel[rank] = c;
            if (rank == nwant - 1) attempts_s = c + 1;
        }
        running += tot;
        if (running >= nwant) break;            // block-uniform
    }
    __syncthreads();
    const int attempts = (nwant > 0) ? attempts_s : 0;
    for (int c = tid; c < attempts; c += SEL_THREADS) atomicAdd(&hist[reason[c]], 1);
    // sequence offsets of the selected chunks
    const int nsel = (int)min(running, (long long)nwant);
    long long off = 0;
    for (int base = 0; base <= nwant; base += SEL_THREADS) {
        const int k = base + tid;
        const long long L = (k < nsel) ? seqlen[sel[k]] : 0;
        long long tot;
        const long long pre = off + block_exclusive_scan(L, part, &tot);
        if (k <= nwant) seqoff[k] = pre;
        off += tot;
    }
    __syncthreads();
    if (tid < CH_NREASON) counts[tid] = hist[tid];
    if (tid == 0) {
        counts[CH_NREASON] = nsel;
        counts[CH_NREASON + 1] = attempts;
    }
}

// ---------------------------------------------------------------------------
// 3a. signal tiles
// ---------------------------------------------------------------------------
constexpr int GT = 64;              // tile: 64 samples ...
constexpr int GN = 16;              // ... x 16 chunks (504 blocks for 128 chunks of 4000 samples)
constexpr int GATHER_THREADS = 256;

__global__ __launch_bounds__(GATHER_THREADS) void chunk_signal_kernel(
    tk_mapped_store st, const int32_t *__restrict__ cand_read, const int32_t *__restrict__ dacstart,
    const int32_t *__restrict__ sel, const int32_t *__restrict__ counts, int nwant, int chunk_len,
    int reverse, int standardize, float *__restrict__ indata) {
    __shared__ float tile[GN][GT + 1];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int t0 = blockIdx.x * GT, n0 = blockIdx.y * GN;
    const int nsel = counts[CH_NREASON];
    // read: wave w takes chunks n0 + w, w + 4, ...; lane = sample (2-byte loads, 128 B per wave)
#pragma unroll
    for (int j = wave; j < GN; j += GATHER_THREADS / WAVE) {
        const int n = n0 + j, t = t0 + lane;
        float v = 0.f;
        if (n < nsel && t < chunk_len) {
            const int c = sel[n], r = cand_read[c];
            const int pos = reverse ? (chunk_len - 1 - t) : t;          // np.flip of the chunk
            const int16_t d = st.dacs[st.dacs_off[r] + dacstart[c] + pos];
            const double *sc = st.scaling + 5 * (size_t)r;
            // signal_mapping.py:476-480, float64 like numpy's int16 array (+) Python float
            double cur = ((double)d + sc[0]) * sc[1] / sc[2];
            if (standardize) cur = (cur - sc[3]) / sc[4];
            v = (float)cur;
        }
        tile[j][lane] = v;
    }
    __syncthreads();
    // write: a wave stores 4 samples x 16 chunks per pass (64 B runs along the batch)
    const int jn = lane & (GN - 1), it = lane / GN;
#pragma unroll
    for (int i = wave * (WAVE / GN) + it; i < GT; i += GATHER_THREADS / GN) {
        const int t = t0 + i, n = n0 + jn;
        if (t < chunk_len && n < nwant) indata[(size_t)t * nwant + n] = tile[jn][i];
    }
}

// ---------------------------------------------------------------------------
// 3b. sequences: flip-flop code of the (optionally reversed) reference slice
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chunk_sequence_kernel(
    tk_mapped_store st, const int32_t *__restrict__ cand_read, const int32_t *__restrict__ seqstart,
    const int32_t *__restrict__ seqlen, const int32_t *__restrict__ sel,
    const int64_t *__restrict__ seqoff, const int32_t *__restrict__ counts, int reverse, int ncan,
    const int32_t *__restrict__ can_labels, const int32_t *__restrict__ mod_labels,
    int64_t seqs_cap, int32_t *__restrict__ seqs, int32_t *__restrict__ seqlens_out,
    int32_t *__restrict__ mod_cats, uint32_t *__restrict__ status) {
    const int n = blockIdx.x;
    const int nsel = counts[CH_NREASON];
    if (n >= nsel) {
        if (threadIdx.x == 0) seqlens_out[n] = 0;
        return;
    }
    const int c = sel[n], r = cand_read[c];
    const int L = seqlen[c];
    // the Reference of read r starts rts_off[r] - r entries in (one fewer entry per read)
    const int16_t *ref = st.reference + (st.rts_off[r] - r) + seqstart[c];
    const int64_t off = seqoff[n];
    if (threadIdx.x == 0) {
        seqlens_out[n] = L;
        if (off + L > seqs_cap && status) atomicOr(status, TK_STATUS_SEQS_OVERFLOW);
    }
    auto label = [&](int k) {               // k-th label in output order
        const int v = ref[reverse ? (L - 1 - k) : k];
        return can_labels ? can_labels[v] : v;
    };
    for (int k = threadIdx.x; k < L; k += blockDim.x) {
        if (off + k >= seqs_cap) break;
        const int me = label(k);
        // flipflopfings.py:34-53: flop at odd positions within a run of equal labels
        int run = 0;
        while (run < k && label(k - 1 - run) == me) ++run;
        seqs[off + k] = me + ((run & 1) ? ncan : 0);
        if (mod_cats) mod_cats[off + k] = mod_labels[ref[reverse ? (L - 1 - k) : k]];
    }
}

// ---------------------------------------------------------------------------
int chunks_locate_dispatch(const tk_mapped_store *st, const int32_t *cand_read, const int32_t *cand_start,
                           const double *cand_frac, size_t ncand, size_t chunk_len,
                           const tk_chunk_filter *fp, uint8_t *reason, int32_t *dacstart,
                           int32_t *seqstart, int32_t *seqlen, int32_t *maxdwell, hipStream_t stream) {
    const int blocks = (int)((ncand + LOC_WAVES - 1) / LOC_WAVES);
    hipLaunchKernelGGL(chunk_locate_kernel, dim3(blocks), dim3(LOC_WAVES * WAVE), 0, stream, *st, cand_read,
                       cand_start, cand_frac, (int)ncand, (int)chunk_len, *fp, reason, dacstart, seqstart,
                       seqlen, maxdwell);
    return hipGetLastError() == hipSuccess ? TK_OK : TK_ERR_LAUNCH;
}

int chunks_select_dispatch(const uint8_t *reason, const int32_t *seqlen, size_t ncand, size_t nwant,
                           int32_t *sel, int64_t *seqoff, int32_t *counts, hipStream_t stream) {
    hipLaunchKernelGGL(chunk_select_kernel, dim3(1), dim3(SEL_THREADS), 0, stream, reason, seqlen, (int)ncand,
                       (int)nwant, sel, seqoff, counts);
    return hipGetLastError() == hipSuccess ? TK_OK : TK_ERR_LAUNCH;
}

int chunks_gather_dispatch(const tk_mapped_store *st, const int32_t *cand_read, const int32_t *dacstart,
                           const int32_t *seqstart, const int32_t *seqlen, const int32_t *sel,
                           const int64_t *seqoff, const int32_t *counts, size_t nwant, size_t chunk_len,
                           int reverse, int standardize, size_t ncan, const int32_t *can_labels,
                           const int32_t *mod_labels, float *indata, int32_t *seqs, size_t seqs_cap,
                           int32_t *seqlens_out, int32_t *mod_cats, uint32_t *status, hipStream_t stream) {
    const dim3 grid((unsigned)((chunk_len + GT - 1) / GT), (unsigned)((nwant + GN - 1) / GN));
    hipLaunchKernelGGL(chunk_signal_kernel, grid, dim3(GATHER_THREADS), 0, stream, *st, cand_read, dacstart, sel,
                       counts, (int)nwant, (int)chunk_len, reverse, standardize, indata);
    hipLaunchKernelGGL(chunk_sequence_kernel, dim3((unsigned)nwant), dim3(256), 0, stream, *st, cand_read,
                       seqstart, seqlen, sel, seqoff, counts, reverse, (int)ncan, can_labels, mod_labels,
                       (int64_t)seqs_cap, seqs, seqlens_out, mod_cats, status);
    return hipGetLastError() == hipSuccess ? TK_OK : TK_ERR_LAUNCH;
}

}  // namespace tk
