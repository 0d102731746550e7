// logz_kernels.hip -- log-partition function of the 2*nbase-state flip-flop CRF
// and its gradient (posterior transition probabilities) for gfx950.
//
// Replaces taiyaki/cupy_extensions/flipflop.py:10-368 (flipflop_fwd / _bwd /
// _make_trans / LogZ) and the torch fallback taiyaki/layers.py:1277-1299.
//
// Design (MI355X-first, see DESIGN.md "Kernel B"):
//   * lane = read: 64 consecutive reads' rows at one time step are ONE contiguous
//     10 KiB segment of the (T, N, S) tensor, so every HBM access is a fully
//     coalesced 16-byte-per-lane stream; a wave-private LDS buffer transposes
//     pieces -> rows.  No cross-lane arithmetic anywhere.
//   * the time axis is parallelised exactly with (sum,*)-semiring transfer
//     matrices: K1 computes the 2nb x 2nb transfer matrix of every 32-row chunk
//     (one wave per chunk, matrix in registers), K2 scans the chunk matrices
//     (forward and backward boundary vectors + logZ), K3 re-reads each chunk
//     ONCE, rows held in registers, runs the in-chunk forward/backward and
//     writes the normalised posterior.  HBM traffic = 2 reads + 1 write of the
//     score tensor = the algorithmic minimum 3*T*N*S*4 bytes (+ ~6% workspace).
//   * arithmetic is linear-space fp32 with exact power-of-two renormalisation
//     (integer exponents are accumulated exactly; row maxima in fp64), so no
//     transcendental sits on the serial dependency chain.
#include "ff_common.h"

namespace tk {

constexpr int LOGZ_CH = 32;             // rows per chunk
constexpr int K1_WAVES = 4;             // independent chunks per K1 block
constexpr int K3_WAVES = 8;             // waves per K3 block: 8 x 4 rows = 1 chunk
constexpr int K3_ROWS = LOGZ_CH / K3_WAVES;

// per-wave LDS buffer of K3 in f4 units: the row-set transpose buffer, which
// doubles as storage for the wave's K3_ROWS forward vectors
template <int NB>
__host__ __device__ constexpr int k3_buf_f4() {
    constexpr int a = WAVE * FF<NB>::PIECES;
    constexpr int b = K3_ROWS * FF<NB>::NS * WAVE / 4;
    return a > b ? a : b;
}

struct LogzWs {
    float *P;        // [C][NS*NS][Npad]  chunk transfer matrices (row-scaled mantissas)
    int32_t *E;      // [C][NS][Npad]     per-row binary exponents
    double *M;       // [C][Npad]         sum of row maxima of the chunk
    float *Vin;      // [C][NS][Npad]     forward vector entering chunk c
    float *Uout;     // [C][NS][Npad]     backward vector leaving chunk c
};

// ---------------------------------------------------------------------------
// K1: chunk transfer matrices.  grid = (ncols, ceil(C / K1_WAVES)), block = 256.
// ---------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(K1_WAVES *WAVE) void logz_transfer_kernel(
    const float *__restrict__ scores, int T, int N, int C, int Npad, LogzWs ws) {
    using F = FF<NB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int c = blockIdx.y * K1_WAVES + wave;
    if (c >= C) return;
    f4 *buf = reinterpret_cast<f4 *>(smem) + wave * (WAVE * F::PIECES);
    const int n0 = blockIdx.x * WAVE;
    const int nvalid = min(WAVE, N - n0) * F::PIECES;
    const int t0 = c * LOGZ_CH, t1 = min(T, t0 + LOGZ_CH);
    const size_t rowstride = (size_t)N * F::S;
    const float *base = scores + (size_t)n0 * F::S;

    float P[F::NS][F::NS];
    int e[F::NS];
    double msum = 0.0;
#pragma unroll
    for (int i = 0; i < F::NS; ++i) {
        e[i] = 0;
#pragma unroll
        for (int j = 0; j < F::NS; ++j) P[i][j] = (i == j) ? 1.f : 0.f;
    }

    RowSet<NB> cur, nxt;
    cur.issue(base + (size_t)t0 * rowstride, nvalid, lane);
    for (int t = t0; t < t1; ++t) {
        if (t + 1 < t1) nxt.issue(base + (size_t)(t + 1) * rowstride, nvalid, lane);
        cur.to_rows(buf, lane);
        msum += (double)cur.exp_normalise();
#pragma unroll
        for (int i = 0; i < F::NS; ++i) {
            float out[F::NS];
            ff_fwd_step<NB>(P[i], cur, out);
#pragma unroll
            for (int j = 0; j < F::NS; ++j) P[i][j] = out[j];
        }
        if (((t - t0) & 3) == 3) {
#pragma unroll
            for (int i = 0; i < F::NS; ++i) e[i] += pow2_normalise(P[i]);
        }
        cur = nxt;
    }
#pragma unroll
    for (int i = 0; i < F::NS; ++i) e[i] += pow2_normalise(P[i]);

    const size_t n = (size_t)n0 + lane;     // < Npad always
    float *Pout = ws.P + (size_t)c * (F::NS * F::NS) * Npad + n;
#pragma unroll
    for (int i = 0; i < F::NS; ++i)
#pragma unroll
        for (int j = 0; j < F::NS; ++j) Pout[(size_t)(i * F::NS + j) * Npad] = P[i][j];
    int32_t *Eout = ws.E + (size_t)c * F::NS * Npad + n;
#pragma unroll
    for (int i = 0; i < F::NS; ++i) Eout[(size_t)i * Npad] = e[i];
    ws.M[(size_t)c * Npad + n] = msum;
}

// ---------------------------------------------------------------------------
// K2: scan over chunk matrices.  grid = ncols, block = 128: wave 0 runs the
// forward scan (Vin[c], logZ), wave 1 the backward scan (Uout[c]).
// ---------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(2 * WAVE) void logz_scan_kernel(int N, int C, int Npad, LogzWs ws,
                                                          float *__restrict__ logz,
                                                          int want_bwd,
                                                          uint32_t *__restrict__ status) {
    using F = FF<NB>;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const size_t n = (size_t)blockIdx.x * WAVE + lane;
    const size_t cstrideP = (size_t)(F::NS * F::NS) * Npad, cstrideV = (size_t)F::NS * Npad;

    if (wave == 0) {
        // paths start in any flip state with weight 1 (layers.py:1289-1295,
        // cupy flipflop.py:115-118)
        float v[F::NS];
#pragma unroll
        for (int s = 0; s < F::NS; ++s) v[s] = (s < NB) ? 1.f : 0.f;
        double macc = 0.0;
        long long eacc = 0;
        for (int c = 0; c < C; ++c) {
            float *vin = ws.Vin + (size_t)c * cstrideV + n;
#pragma unroll
            for (int s = 0; s < F::NS; ++s) vin[(size_t)s * Npad] = v[s];
            const float *Pc = ws.P + (size_t)c * cstrideP + n;
            const int32_t *Ec = ws.E + (size_t)c * cstrideV + n;
            // bring v[i] * 2^e[i] to a common exponent
            int te[F::NS], emax = INT32_MIN;
#pragma unroll
            for (int i = 0; i < F::NS; ++i) {
                const int ei = Ec[(size_t)i * Npad];
                te[i] = ei;
                if (v[i] > 0.f) emax = max(emax, ei + __builtin_amdgcn_frexp_expf(v[i]));
            }
            if (emax == INT32_MIN) emax = 0;
            float vs[F::NS], out[F::NS];
#pragma unroll
            for (int i = 0; i < F::NS; ++i) vs[i] = __builtin_amdgcn_ldexpf(v[i], te[i] - emax);
#pragma unroll
            for (int j = 0; j < F::NS; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < F::NS; ++i) acc = fmaf(vs[i], Pc[(size_t)(i * F::NS + j) * Npad], acc);
                out[j] = acc;
            }
            const int ex = pow2_normalise(out);
#pragma unroll
            for (int j = 0; j < F::NS; ++j) v[j] = out[j];
            eacc += (long long)emax + ex;
            macc += ws.M[(size_t)c * Npad + n];
        }
        float tot = 0.f;
#pragma unroll
        for (int s = 0; s < F::NS; ++s) tot += v[s];
        const double lz = macc + (double)eacc * 0.6931471805599453 + (double)logf(tot);
        if (n < (size_t)N) {
            const float lzf = (float)lz;
            logz[n] = lzf;
            if (status != nullptr && !isfinite(lzf)) atomicOr(status, 1u);
        }
    } else if (want_bwd) {
        // paths may end in any state (cupy flipflop.py:163-166); scale is free
        float u[F::NS];
#pragma unroll
        for (int s = 0; s < F::NS; ++s) u[s] = 1.f;
        for (int c = C - 1; c >= 0; --c) {
            float *uo = ws.Uout + (size_t)c * cstrideV + n;
#pragma unroll
            for (int s = 0; s < F::NS; ++s) uo[(size_t)s * Npad] = u[s];
            const float *Pc = ws.P + (size_t)c * cstrideP + n;
            const int32_t *Ec = ws.E + (size_t)c * cstrideV + n;
            float y[F::NS];
            int te[F::NS], emax = INT32_MIN;
#pragma unroll
            for (int i = 0; i < F::NS; ++i) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < F::NS; ++j) acc = fmaf(Pc[(size_t)(i * F::NS + j) * Npad], u[j], acc);
                y[i] = acc;
                te[i] = Ec[(size_t)i * Npad];
                if (acc > 0.f) emax = max(emax, te[i] + __builtin_amdgcn_frexp_expf(acc));
            }
            if (emax == INT32_MIN) emax = 0;
#pragma unroll
            for (int i = 0; i < F::NS; ++i) u[i] = __builtin_amdgcn_ldexpf(y[i], te[i] - emax);
        }
    }
}

// ---------------------------------------------------------------------------
// K3: posterior.  grid = (ncols, C), block = 512 (8 waves x 4 rows = one chunk).
// Rows live in registers; forward / backward boundary vectors are chained
// between the waves through LDS; posteriors overwrite the rows in place and
// are streamed out through the same coalescing transpose.
// ---------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(K3_WAVES *WAVE, 2) void logz_posterior_kernel(
    const float *__restrict__ scores, float *__restrict__ grad, int T, int N, int Npad,
    LogzWs ws, uint32_t *__restrict__ status) {
    using F = FF<NB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    constexpr int BUF_F4 = k3_buf_f4<NB>();
    f4 *buf = reinterpret_cast<f4 *>(smem) + wave * BUF_F4;
    // between the load phase and the store phase the transpose buffer is idle:
    // it keeps this wave's forward vectors (lane-private slots, no sync needed)
    float *fvb = reinterpret_cast<float *>(buf);            // [K3_ROWS][NS][64]
    float *chainF = reinterpret_cast<float *>(reinterpret_cast<f4 *>(smem) + K3_WAVES * BUF_F4);
    float *chainB = chainF + F::NS * WAVE;
    const int c = blockIdx.y;
    const int n0 = blockIdx.x * WAVE;
    const int nvalid = min(WAVE, N - n0) * F::PIECES;
    const size_t rowstride = (size_t)N * F::S;
    const int tw = c * LOGZ_CH + wave * K3_ROWS;        // first row of this wave
    const float *base = scores + (size_t)n0 * F::S;

    // 1. rows -> registers (weights w = exp(s - rowmax))
    RowSet<NB> w[K3_ROWS];
#pragma unroll
    for (int j = 0; j < K3_ROWS; ++j) {
        if (tw + j < T) w[j].issue(base + (size_t)(tw + j) * rowstride, nvalid, lane);
    }
#pragma unroll
    for (int j = 0; j < K3_ROWS; ++j) {
        if (tw + j < T) {
            w[j].to_rows(buf, lane);
            (void)w[j].exp_normalise();
        }
    }

    // 2. chain the boundary vectors through the 8 waves
    const size_t n = (size_t)n0 + lane;
    const size_t cstrideV = (size_t)F::NS * Npad;
    float bexit[F::NS];             // backward vector AFTER this wave's last row
#pragma unroll 1
    for (int s = 0; s < K3_WAVES; ++s) {
        if (wave == s) {
            float f[F::NS];
            if (s == 0) {
                const float *vin = ws.Vin + (size_t)c * cstrideV + n;
#pragma unroll
                for (int k = 0; k < F::NS; ++k) f[k] = vin[(size_t)k * Npad];
            } else {
#pragma unroll
                for (int k = 0; k < F::NS; ++k) f[k] = chainF[k * WAVE + lane];
            }
#pragma unroll
            for (int j = 0; j < K3_ROWS; ++j) {
#pragma unroll
                for (int k = 0; k < F::NS; ++k) fvb[(j * F::NS + k) * WAVE + lane] = f[k];
                if (tw + j < T) {
                    float out[F::NS];
                    ff_fwd_step<NB>(f, w[j], out);
                    (void)pow2_normalise(out);
#pragma unroll
                    for (int k = 0; k < F::NS; ++k) f[k] = out[k];
                }
            }
#pragma unroll
            for (int k = 0; k < F::NS; ++k) chainF[k * WAVE + lane] = f[k];
        }
        if (wave == K3_WAVES - 1 - s) {
            float b[F::NS];
            if (s == 0) {
                const float *uo = ws.Uout + (size_t)c * cstrideV + n;
#pragma unroll
                for (int k = 0; k < F::NS; ++k) b[k] = uo[(size_t)k * Npad];
            } else {
#pragma unroll
                for (int k = 0; k < F::NS; ++k) b[k] = chainB[k * WAVE + lane];
            }
#pragma unroll
            for (int k = 0; k < F::NS; ++k) bexit[k] = b[k];
#pragma unroll
            for (int j = K3_ROWS - 1; j >= 0; --j) {
                if (tw + j < T) {
                    float out[F::NS];
                    ff_bwd_step<NB>(b, w[j], out);
                    (void)pow2_normalise(out);
#pragma unroll
                    for (int k = 0; k < F::NS; ++k) b[k] = out[k];
                }
            }
#pragma unroll
            for (int k = 0; k < F::NS; ++k) chainB[k * WAVE + lane] = b[k];
        }
        __syncthreads();
    }

    // 3. posterior rows, last row first:
    //    g[uv] = fwd[t][u] * w[t][uv] * bwd[t+1][v] / sum_uv(...)
    //    (cupy flipflop.py:280-291 + softmax at 351-354)
    float b[F::NS];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < F::NS; ++k) b[k] = bexit[k];
#pragma unroll
    for (int j = K3_ROWS - 1; j >= 0; --j) {
        if (tw + j < T) {
            float bn[F::NS], fv[F::NS];
            ff_bwd_step<NB>(b, w[j], bn);
            (void)pow2_normalise(bn);
#pragma unroll
            for (int k = 0; k < F::NS; ++k) fv[k] = fvb[(j * F::NS + k) * WAVE + lane];
            float sum = 0.f;
#pragma unroll
            for (int to = 0; to < NB; ++to) {
#pragma unroll
                for (int from = 0; from < F::NS; ++from) {
                    const float g = fv[from] * w[j].get(to * F::NS + from) * b[to];
                    w[j].set(to * F::NS + from, g);
                    sum += g;
                }
            }
#pragma unroll
            for (int from = 0; from < F::NS; ++from) {
                const float g = fv[from] * w[j].get(F::FLOP0 + from) * b[NB + (from % NB)];
                w[j].set(F::FLOP0 + from, g);
                sum += g;
            }
            const float inv = 1.0f / sum;
            bad |= !isfinite(inv);
#pragma unroll
            for (int i = 0; i < F::S; ++i) w[j].set(i, w[j].get(i) * inv);
#pragma unroll
            for (int k = 0; k < F::NS; ++k) b[k] = bn[k];
        }
    }
    if (status != nullptr && bad && n < (size_t)N) atomicOr(status, 2u);

    // 4. stream the posterior out (rows -> pieces -> coalesced stores)
    wave_lds_fence();
    float *gbase = grad + (size_t)n0 * F::S;
#pragma unroll
    for (int j = 0; j < K3_ROWS; ++j) {
        if (tw + j < T) {
            w[j].to_pieces(buf, lane);
            w[j].store(gbase + (size_t)(tw + j) * rowstride, nvalid, lane);
        }
    }
}

// ---------------------------------------------------------------------------
// host-side launcher
// ---------------------------------------------------------------------------
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

template <int NB>
static size_t logz_ws_layout(size_t T, size_t N, void *base, LogzWs *ws) {
    using F = FF<NB>;
    const size_t C = (T + LOGZ_CH - 1) / LOGZ_CH, Npad = align_up(N, WAVE);
    size_t off = 0;
    char *p = static_cast<char *>(base);
    auto take = [&](size_t bytes) {
        char *r = p ? p + off : nullptr;
        off += align_up(bytes, 256);
        return r;
    };
    float *P = reinterpret_cast<float *>(take(C * F::NS * F::NS * Npad * sizeof(float)));
    int32_t *E = reinterpret_cast<int32_t *>(take(C * F::NS * Npad * sizeof(int32_t)));
    double *M = reinterpret_cast<double *>(take(C * Npad * sizeof(double)));
    float *Vin = reinterpret_cast<float *>(take(C * F::NS * Npad * sizeof(float)));
    float *Uout = reinterpret_cast<float *>(take(C * F::NS * Npad * sizeof(float)));
    if (ws) *ws = LogzWs{P, E, M, Vin, Uout};
    return off;
}

template <int NB>
static int logz_launch(const float *scores, size_t T, size_t N, float *logz, float *grad,
                       void *workspace, size_t workspace_bytes, uint32_t *status,
                       hipStream_t stream) {
    using F = FF<NB>;
    LogzWs ws;
    const size_t need = logz_ws_layout<NB>(T, N, workspace, &ws);
    if (need > workspace_bytes) return 3;
    const int C = (int)((T + LOGZ_CH - 1) / LOGZ_CH);
    const int ncols = (int)((N + WAVE - 1) / WAVE), Npad = ncols * WAVE;
    const size_t bufbytes = (size_t)WAVE * F::PIECES * sizeof(f4);
    {
        dim3 grid(ncols, (C + K1_WAVES - 1) / K1_WAVES), block(K1_WAVES * WAVE);
        hipLaunchKernelGGL(logz_transfer_kernel<NB>, grid, block, K1_WAVES * bufbytes, stream,
                           scores, (int)T, (int)N, C, Npad, ws);
    }
    {
        dim3 grid(ncols), block(2 * WAVE);
        hipLaunchKernelGGL(logz_scan_kernel<NB>, grid, block, 0, stream, (int)N, C, Npad, ws,
                           logz, grad != nullptr ? 1 : 0, status);
    }
    if (grad != nullptr) {
        dim3 grid(ncols, C), block(K3_WAVES * WAVE);
        const size_t lds = K3_WAVES * (size_t)k3_buf_f4<NB>() * sizeof(f4) +
                           2 * F::NS * WAVE * sizeof(float);
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(&logz_posterior_kernel<NB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return 4;
        hipLaunchKernelGGL(logz_posterior_kernel<NB>, grid, block, lds, stream, scores, grad,
                           (int)T, (int)N, Npad, ws, status);
    }
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

size_t logz_workspace_bytes(size_t T, size_t N, size_t nbase) {
    switch (nbase) {
        case 1: return logz_ws_layout<1>(T, N, nullptr, nullptr);
        case 2: return logz_ws_layout<2>(T, N, nullptr, nullptr);
        case 3: return logz_ws_layout<3>(T, N, nullptr, nullptr);
        case 4: return logz_ws_layout<4>(T, N, nullptr, nullptr);
        default: return 0;
    }
}

int logz_dispatch(const float *scores, size_t T, size_t N, size_t nbase, float *logz,
                  float *grad, void *workspace, size_t workspace_bytes, uint32_t *status,
                  hipStream_t stream) {
    switch (nbase) {
        case 1: return logz_launch<1>(scores, T, N, logz, grad, workspace, workspace_bytes, status, stream);
        case 2: return logz_launch<2>(scores, T, N, logz, grad, workspace, workspace_bytes, status, stream);
        case 3: return logz_launch<3>(scores, T, N, logz, grad, workspace, workspace_bytes, status, stream);
        case 4: return logz_launch<4>(scores, T, N, logz, grad, workspace, workspace_bytes, status, stream);
        default: return 2;
    }
}

}  // namespace tk
