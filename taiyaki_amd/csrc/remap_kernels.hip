// remap_kernels.hip -- highest-scoring path through a flip-flop score matrix that spells a given
// sequence (gfx950).
//
// Replaces taiyaki/flipflop_remap.py:6-88 (map_to_crf_viterbi), the alignment step of
// prepare_mapped_reads (prepare_mapping_funcs.py:88): a Python loop over the T time steps with
// ~12 numpy calls on length-M arrays per step, then a Python traceback.  It is the max-plus
// twin of the sequence CRF (crf_kernels.hip) with "start" and "end" states that may swallow
// signal at `localpen` per block (glocal mapping).
//
// One workgroup per read, position-parallel: thread g owns the R consecutive sequence
// positions [g R, (g+1) R) in registers.  A time step is R independent cells per lane and ONE
// neighbour exchange (DPP wave_shr inside a wave, a double-buffered LDS slot and one s_barrier
// between waves; a sequence that fits one wave runs without any barrier).  Score rows are
// staged 32 at a time in LDS (the next tile is already in registers while this one is
// consumed) and gathered by transition id.  Everything is float64, in the reference's order of
// operations: the float32 scores enter exactly, so scores, ties (strict '<') and therefore
// paths are bit-identical to the reference's.
//
// Traceback: one bit per (step, position), packed along the sequence so that byte b of a row
// holds positions 8b..8b+7.  A position moves back by at most one per step, so wave 0 walks 64
// steps per round trip: lane l fetches the 64-bit window [m-63, m] of row n-l (two aligned
// 8-byte loads), then the walk itself is scalar (v_readlane + s_lshr), and the 64 path entries
// leave in one coalesced store.
//
// HBM traffic is small (T K 4 bytes of scores, T M / 8 bytes of traceback each way); the
// kernel is bound by the latency of the serial time loop, like the reference -- only per step
// it costs ~0.1 us instead of ~60 us of numpy dispatch.
#include "ff_common.h"
#include "../../include/taiyaki_amd_flipflop.h"

#pragma clang fp contract(off)

namespace tk {

constexpr double REMAP_LARGE = 1e30;        // taiyaki/constants.py LARGE_VAL
constexpr int RM_ROWS = 32;                 // score rows per LDS tile

struct RemapArgs {
    const float *scores;        // concatenated (sum T_i, K)
    const int64_t *row_off;     // (nread + 1) row offsets
    const int32_t *stay_index;  // concatenated, M_i per read
    const int32_t *step_index;  // concatenated, M_i - 1 per read (read i starts at seq_off[i] - i)
    const int64_t *seq_off;     // (nread + 1)
    const double *localpen;     // per read
    int K;
    double *score;              // (nread)
    int64_t *path;              // concatenated, T_i + 1 per read (read i starts at row_off[i] + i)
    uint64_t *tb;               // traceback bits
    const int64_t *tb_off;      // (nread) offsets into tb, in 64-bit words
};

__device__ __forceinline__ double wave_shift_up1_f64(double src, double fill) {
    const int lo = wave_shift_up1(__double2loint(src), __double2loint(fill));
    const int hi = wave_shift_up1(__double2hiint(src), __double2hiint(fill));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane) {
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

// scores a thread carries from one tile to the next: RM_ROWS * K / threads.  R = 2 blocks can
// be as small as one wave (20 for K = 40); R = 8 / 16 blocks have at least 257 / 513 threads.
__host__ __device__ constexpr int remap_prefetch_regs(int R) { return R == 2 ? 24 : 8; }

// NT: the launch bound.  Sixteen float64 cells per thread do not fit the 128 registers a 1024-thread
// workgroup leaves (54 spills, 2 us per step); sequences up to 12288 bases need at most 768 threads,
// which leaves 170: no spills (round 3).
template <int R, int NT>
__global__ __launch_bounds__(NT) void remap_kernel(RemapArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int read = blockIdx.x;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int64_t row0 = a.row_off[read];
    const int T = (int)(a.row_off[read + 1] - row0);
    const int64_t s0 = a.seq_off[read];
    const int M = (int)(a.seq_off[read + 1] - s0);
    const int K = a.K;
    const double localpen = a.localpen[read];
    const float *scores = a.scores + row0 * K;
    int64_t *path = a.path + row0 + read;
    const int pitchw = (M + 63) >> 6;                       // 64-bit words per traceback row
    uint64_t *tb = a.tb + a.tb_off[read];

    float *tile = reinterpret_cast<float *>(smem);                              // [2][RM_ROWS * K]
    double *slot = reinterpret_cast<double *>(tile + 2 * RM_ROWS * K);          // [2][16]
    int *start_nm = reinterpret_cast<int *>(slot + 2 * 16);                     // [2]
    const bool multi = nwaves > 1;                          // block-uniform
    auto sync = [&] {
        if (multi) __syncthreads();
        else wave_lds_fence();
    };

    for (int i = tid; i <= T; i += nthreads) path[i] = -1;  // flipflop_remap.py:69

    // my positions and their transition ids; positions >= M are inert (they only receive)
    const int m_first = tid * R;
    uint32_t ids[R];                                        // stay id | step id << 16
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int m = m_first + r;
        const uint32_t st = (m < M) ? (uint32_t)a.stay_index[s0 + m] : 0u;
        const uint32_t sp = (m < M - 1) ? (uint32_t)a.step_index[s0 - read + m] : 0u;
        ids[r] = st | (sp << 16);
    }
    double p[R];                                            // :30-32
#pragma unroll
    for (int r = 0; r < R; ++r) p[r] = (m_first + r == 0) ? 0.0 : -REMAP_LARGE;
    double start_score = 0.0, end_score = -REMAP_LARGE;     // thread 0 / the owner of M - 1
    int alignment_end = 0;
    const int last_owner = (M - 1) / R, last_r = (M - 1) - last_owner * R;

    const int tile_elems = RM_ROWS * K;
    const int per_thread = (tile_elems + nthreads - 1) / nthreads;     // <= 20 with >= 64 threads, K = 40
    constexpr int PRE_MAX = remap_prefetch_regs(R);
    float pre[PRE_MAX];
    const int64_t total_elems = (int64_t)T * K;
    auto fetch_tile = [&](int j) {                          // into registers
#pragma unroll
        for (int k = 0; k < PRE_MAX; ++k) {
            if (k < per_thread) {
                const int64_t e = (int64_t)j * tile_elems + (int64_t)k * nthreads + tid;
                pre[k] = scores[min(e, total_elems - 1)];   // clamped, unconditional
            }
        }
    };
    auto store_tile = [&](int j) {
        float *dst = tile + (j & 1) * tile_elems;
#pragma unroll
        for (int k = 0; k < PRE_MAX; ++k) {
            if (k < per_thread) {
                const int e = k * nthreads + tid;
                if (e < tile_elems) dst[e] = pre[k];
            }
        }
    };
    const int ntiles = (T + RM_ROWS - 1) / RM_ROWS;
    if (ntiles > 0) {
        fetch_tile(0);
        store_tile(0);
    }
    sync();

    for (int j = 0; j < ntiles; ++j) {
        if (j + 1 < ntiles) fetch_tile(j + 1);
        const float *rows = tile + (j & 1) * tile_elems;
        const int n_lo = j * RM_ROWS, n_hi = min(T, n_lo + RM_ROWS);
        for (int n = n_lo; n < n_hi; ++n) {
            const float *row = rows + (n - n_lo) * K;
            // what my last position offers its right-hand neighbour (:48-49)
            const double out = p[R - 1] + (double)row[ids[R - 1] >> 16];
            double cin = wave_shift_up1_f64(out, 0.0);
            if (multi) {
                double *sl = slot + (n & 1) * 16;
                if (lane == WAVE - 1) sl[wave] = out;
                __syncthreads();
                if (lane == 0 && wave > 0) cin = sl[wave - 1];
            }
            // the start state feeds position 0, the end state drains position M - 1: one thread
            // each, the other waves skip these blocks
            double leave_start = 0.0;
            if (tid == 0) {
                const double stay0 = (double)row[ids[0] & 0xffffu];
                leave_start = start_score - localpen;                      // :52
                start_score = start_score + fmax(stay0, -localpen);        // :53
            }
            if (tid == last_owner) {
                double p_last = p[0];
                uint32_t id_last = ids[0];
#pragma unroll
                for (int r = 1; r < R; ++r) {
                    p_last = (r == last_r) ? p[r] : p_last;
                    id_last = (r == last_r) ? ids[r] : id_last;
                }
                const double stay_l = (double)row[id_last & 0xffffu];
                const double remain = end_score + fmax(stay_l, -localpen); // :63
                const double into_end = p_last - localpen;                 // :64
                end_score = fmax(remain, into_end);
                if (into_end > remain) alignment_end = n;
            }
            uint32_t bits = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double stay_s = (double)row[ids[r] & 0xffffu];
                const double cstay = p[r] + stay_s;                         // :45-46
                const double next_out = p[r] + (double)row[ids[r] >> 16];
                double cand = cin;
                bool bit = cstay < cin;                                     // :59
                if (r == 0) {
                    cand = (tid == 0) ? start_score : cand;                 // :58 (the updated start score)
                    bit = (tid == 0) ? (leave_start > cstay) : bit;         // :60
                }
                p[r] = fmax(cstay, cand);                                   // :56-58
                bits |= (bit ? 1u : 0u) << r;
                cin = next_out;
                // 16 cells of float64 state leave no room for hoisting every cell's gathers
                if (R > 8 && (r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            // traceback row n (row 0 of the reference's table is all zero and not stored)
            uint8_t *trow = reinterpret_cast<uint8_t *>(tb + (size_t)n * pitchw);
            if (R == 2) {
                uint32_t v = bits;
                v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xF, 0xF, false) << 2;   // row_shl:1
                v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x102, 0xF, 0xF, false) << 4;   // row_shl:2
                if ((tid & 3) == 0 && (tid >> 2) < pitchw * 8) trow[tid >> 2] = (uint8_t)v;
            } else if (R == 8) {
                if (tid < pitchw * 8) trow[tid] = (uint8_t)bits;
            } else {
                static_assert(R == 2 || R == 8 || R == 16, "traceback packing");
                if (tid * 2 < pitchw * 8) reinterpret_cast<uint16_t *>(trow)[tid] = (uint16_t)bits;
            }
        }
        sync();                                 // every wave is done with buffer (j + 1) & 1 (tile j - 1)
        if (j + 1 < ntiles) store_tile(j + 1);
        sync();
    }

    // ---- where the traceback starts (:71-77), the score (:88)
    if (tid == last_owner) {
        double c_last = p[0];
#pragma unroll
        for (int r = 1; r < R; ++r) c_last = (r == last_r) ? p[r] : c_last;
        const bool from_seq_end = c_last > end_score;
        start_nm[0] = from_seq_end ? T : alignment_end;
        start_nm[1] = M - 1;
        a.score[read] = fmax(c_last, end_score);
    }
    __threadfence();
    __syncthreads();
    if (wave != 0) return;

    // ---- traceback (:79-86): 64 steps per round trip
    int n = __builtin_amdgcn_readfirstlane(start_nm[0]);
    int m = __builtin_amdgcn_readfirstlane(start_nm[1]);
    while (n >= 0 && m >= 0) {
        const int n0 = n, base = m - 63;        // the window holds positions base .. base + 63
        const int row_l = n0 - lane;
        uint64_t mask = 0;
        if (row_l >= 1) {
            const uint64_t *trow = tb + (size_t)(row_l - 1) * pitchw;
            const int q = base >> 6, sh = base & 63;            // arithmetic shift: floor
            const uint64_t w0 = (q >= 0 && q < pitchw) ? trow[q] : 0;
            const uint64_t w1 = (q + 1 >= 0 && q + 1 < pitchw) ? trow[q + 1] : 0;
            mask = sh ? ((w0 >> sh) | (w1 << (64 - sh))) : w0;
        }
        int64_t mine = -1;
#pragma unroll
        for (int l = 0; l < WAVE; ++l) {
            if (n >= 0 && m >= 0) {             // wave-uniform
                const uint64_t mk = readlane_u64(mask, l);
                if (lane == l) mine = m;
                m -= (int)((mk >> (m - base)) & 1);
                n -= 1;
            }
        }
        if (mine >= 0) path[row_l] = mine;
    }
}

// ---------------------------------------------------------------------------
// Ref_to_signal from remapping paths (signal_mapping.py:268-316 from_remapping_path, :202-265
// get_reftosignal): entry k of a path sits at signal position k stride - 1 + signalstart;
// reference position r starts at the signal position of the first path entry >= r.  A path is
// -1 (clipped) at its two ends and non-decreasing in between, so that entry is a binary search.
// One workgroup per read; output in the mapped-signal store's layout (read i at ref_off[i] + i).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void path_to_reftosignal_kernel(
    const int64_t *__restrict__ path, const int64_t *__restrict__ path_off, const int64_t *__restrict__ ref_off,
    const int64_t *__restrict__ signalstart, const int64_t *__restrict__ siglen_all, int stride,
    int32_t *__restrict__ rts_all) {
    __shared__ int klo_s, khi_s;
    const int read = blockIdx.x;
    const int64_t p0 = path_off[read];
    const int npath = (int)(path_off[read + 1] - p0);
    const int reflen = (int)(ref_off[read + 1] - ref_off[read]);
    const int64_t ss = signalstart[read], siglen = siglen_all[read];
    const int64_t *pth = path + p0;
    int32_t *rts = rts_all + ref_off[read] + read;
    if (threadIdx.x == 0) {
        klo_s = INT32_MAX;
        khi_s = -1;
    }
    __syncthreads();
    auto sigloc = [&](int k) { return (int64_t)k * stride - 1 + ss; };
    int lo = INT32_MAX, hi = -1;
    for (int k = threadIdx.x; k < npath; k += blockDim.x) {
        const int64_t sl = sigloc(k);
        if (pth[k] != -1 && sl >= 0 && sl < siglen) {
            lo = min(lo, k);
            hi = max(hi, k);
        }
    }
    if (hi >= 0) {
        atomicMin(&klo_s, lo);
        atomicMax(&khi_s, hi);
    }
    __syncthreads();
    const int klo = klo_s, khi = khi_s;
    if (khi < 0) {                                      // nothing mapped: :232-234
        for (int r = threadIdx.x; r <= reflen; r += blockDim.x) rts[r] = -1;
        return;
    }
    const int64_t v0 = pth[klo], vl = pth[khi];
    for (int r = threadIdx.x; r <= reflen; r += blockDim.x) {
        int32_t out;
        if (r < v0) {
            out = -1;                                   // :249-252 start of the reference not mapped
        } else if (r <= vl) {
            int a = klo, b = khi;                       // first k with path[k] >= r
            while (a < b) {
                const int mid = (a + b) >> 1;
                if (pth[mid] < r) a = mid + 1;
                else b = mid;
            }
            out = (int32_t)sigloc(a);                   // :236-238 np.repeat(valid idx, moves)
        } else if (r == vl + 1) {
            out = (int32_t)(sigloc(khi) + 1);           // :240-242 end of the last mapped position
        } else {
            out = (int32_t)(siglen + 1);                // :253-255 end of the reference not mapped
        }
        rts[r] = out;
    }
}

int path_to_reftosignal_dispatch(const int64_t *path, const int64_t *path_off, const int64_t *ref_off,
                                 const int64_t *signalstart, const int64_t *siglen, int stride, size_t nread,
                                 int32_t *rts, hipStream_t stream) {
    hipLaunchKernelGGL(path_to_reftosignal_kernel, dim3((unsigned)nread), dim3(256), 0, stream, path, path_off,
                       ref_off, signalstart, siglen, stride, rts);
    return hipGetLastError() == hipSuccess ? TK_OK : TK_ERR_LAUNCH;
}

size_t remap_lds_bytes(int K) { return 2 * (size_t)RM_ROWS * K * sizeof(float) + 2 * 16 * sizeof(double) + 16; }

int remap_dispatch(const RemapArgs &a, size_t nread, size_t max_M, hipStream_t stream) {
    if (max_M == 0 || max_M > 16384 || a.K <= 0) return TK_ERR_UNSUPPORTED;
    const int R = max_M <= 2048 ? 2 : (max_M <= 8192 ? 8 : 16);
    int threads = (int)((max_M + R - 1) / R);
    threads = ((threads + WAVE - 1) / WAVE) * WAVE;
    // the prefetch registers hold RM_ROWS * K / threads scores per thread
    if ((RM_ROWS * a.K + threads - 1) / threads > remap_prefetch_regs(R)) return TK_ERR_UNSUPPORTED;
    const size_t lds = remap_lds_bytes(a.K);
    switch (R) {
    case 2: hipLaunchKernelGGL((remap_kernel<2, 1024>), dim3((unsigned)nread), dim3(threads), lds, stream, a); break;
    case 8: hipLaunchKernelGGL((remap_kernel<8, 1024>), dim3((unsigned)nread), dim3(threads), lds, stream, a); break;
    default:
        if (threads <= 768)
            hipLaunchKernelGGL((remap_kernel<16, 768>), dim3((unsigned)nread), dim3(threads), lds, stream, a);
        else
            hipLaunchKernelGGL((remap_kernel<16, 1024>), dim3((unsigned)nread), dim3(threads), lds, stream, a);
        break;
    }
    return hipGetLastError() == hipSuccess ? TK_OK : TK_ERR_LAUNCH;
}

}  // namespace tk
