// qscore_kernels.hip -- per-block base error probabilities for the basecaller (gfx950).
//
// Replaces taiyaki/qscores.py:88-142 (errprobs_from_trans), which runs nbase masked
// matmuls over the (T, N, S) posterior tensor plus a normalise / gather chain of torch
// kernels (five passes over the tensor).  Here it is ONE pass: a wave loads the 64 reads'
// rows of a time step as one coalesced 10 KiB segment (lane = read after an LDS
// transpose, the same row-set mover as the logZ kernels), sums the transitions into each
// base, normalises, picks the base of the Viterbi path and writes 4 bytes per (block,
// read).  HBM-bound: T*N*S*4 bytes in, (T+1)*N*(8+4) bytes of path / output.
#include "ff_common.h"

namespace tk {

constexpr int QS_WAVES = 4;         // waves per block
constexpr int QS_ROWS = 2;          // time steps per wave, both row-sets in flight at once
constexpr float QS_SMALL = 1e-10f;  // taiyaki/constants.py:7 SMALL_VAL

template <int NB>
__global__ __launch_bounds__(QS_WAVES *WAVE) void errprobs_kernel(const float *__restrict__ trans,
                                                                  const int64_t *__restrict__ path,
                                                                  int T, int N,
                                                                  float *__restrict__ out) {
    using F = FF<NB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    f4 *buf = reinterpret_cast<f4 *>(smem) + wave * (WAVE * F::PIECES);
    const int n0 = blockIdx.x * WAVE;
    const int nlive = min(WAVE, N - n0);
    const int nvalid = nlive * F::PIECES;
    const int t0 = (blockIdx.y * QS_WAVES + wave) * QS_ROWS;
    const size_t rowstride = (size_t)N * F::S;
    const float *base = trans + (size_t)n0 * F::S;
    const int n = n0 + lane;
    const bool live = lane < nlive;

    // row 0 of the output has no transition into it: 1 - 2.0 (qscores.py:138-140)
    if (blockIdx.y == 0 && wave == 0 && live) out[n] = -1.0f;
    if (t0 >= T) return;

    RowSet<NB> w[QS_ROWS];
    int64_t st[QS_ROWS];
#pragma unroll
    for (int j = 0; j < QS_ROWS; ++j) {
        const int t = min(t0 + j, T - 1);
        w[j].issue_nt(base + (size_t)t * rowstride, nvalid, lane);      // read once
        st[j] = path[(size_t)(t + 1) * N + min(n, N - 1)];
    }
#pragma unroll
    for (int j = 0; j < QS_ROWS; ++j) {
        if (t0 + j < T) {
            w[j].to_rows(buf, lane);
            // total posterior weight of all transitions into base b, flip or flop
            // (qscores.py:58-85: 2nb transitions into b_flip, b_flip -> b_flop, b_flop stay)
            float bp[NB], tot = 0.f;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float acc = w[j].get(b * F::NS);
#pragma unroll
                for (int from = 1; from < F::NS; ++from) acc += w[j].get(b * F::NS + from);
                acc += w[j].get(F::FLOP0 + b);
                acc += w[j].get(F::FLOP0 + NB + b);
                bp[b] = acc;
                tot += acc;
            }
            const int pb = (int)(st[j] % NB);
            float mine = bp[0];
#pragma unroll
            for (int b = 1; b < NB; ++b) mine = (pb == b) ? bp[b] : mine;
            // qscores.py:131-141: normalise by (sum + SMALL_VAL), error prob = 1 - p
            if (live) out[(size_t)(t0 + j + 1) * N + n] = 1.0f - mine / (tot + QS_SMALL);
        }
    }
}

template <int NB>
static int errprobs_launch(const float *trans, const int64_t *path, size_t T, size_t N,
                           float *out, hipStream_t stream) {
    const int ncols = (int)((N + WAVE - 1) / WAVE);
    const int per_block = QS_WAVES * QS_ROWS;
    const size_t lds = (size_t)QS_WAVES * WAVE * FF<NB>::PIECES * sizeof(f4);
    hipLaunchKernelGGL(errprobs_kernel<NB>, dim3(ncols, (unsigned)((T + per_block - 1) / per_block)),
                       dim3(QS_WAVES * WAVE), lds, stream, trans, path, (int)T, (int)N, out);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

int errprobs_dispatch(const float *trans, const int64_t *path, size_t T, size_t N, size_t nbase,
                      float *out, hipStream_t stream) {
    switch (nbase) {
        case 1: return errprobs_launch<1>(trans, path, T, N, out, stream);
        case 2: return errprobs_launch<2>(trans, path, T, N, out, stream);
        case 3: return errprobs_launch<3>(trans, path, T, N, out, stream);
        case 4: return errprobs_launch<4>(trans, path, T, N, out, stream);
        default: return 2;
    }
}

}  // namespace tk
