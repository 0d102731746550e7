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
//     pieces -> rows.
//   * the time axis is parallelised exactly with (sum,*)-semiring transfer
//     matrices, three launches:
//       transfer  : one WAVE per 16-row chunk folds the rows into a 2nb x 2nb matrix
//                   kept in registers as row pairs (v_pk_fma_f32); no cross-wave
//                   products, no barriers.  (Small tensors: four waves per chunk.)
//       middle    : one block per read, eight lanes per chain: super-chunk products,
//                   serial scan over the supers only, expansion back to chunk
//                   boundary vectors; logZ.  Everything from one LDS image.
//       posterior : one 512-thread block per chunk, two rows per wave in registers,
//                   in-chunk forward/backward chained through LDS, normalised
//                   posterior streamed out.  128 VGPRs = two blocks per CU, so one
//                   block's serial chain hides behind the other's loads and stores.
//     HBM traffic = 2 reads + 1 write of the score tensor = the algorithmic
//     minimum 3*T*N*S*4 bytes (+ ~12 % workspace that stays L2/MALL resident).
//   * arithmetic is linear-space fp32 with exact power-of-two renormalisation
//     (integer exponents are accumulated exactly; row maxima in fp64), so no
//     transcendental sits on a serial dependency chain.
#include <stdlib.h>

#include "ff_common.h"

namespace tk {

// rows per chunk (CH) is a template parameter of K1 / K3, picked per problem so that
// the grid fills the chip: 32 for big tensors, 16 / 8 for small ones
// chunks per super-chunk (template parameter SUP of the middle kernel): the serial part is
// SUP + C/SUP + SUP steps, so 8 up to 128 chunks and 16 beyond
__host__ __device__ constexpr int logz_super(int C) { return C > 128 ? 16 : 8; }
constexpr int K1_WAVES = 4;             // waves per K1 block = 4 independent chunks
#ifndef TK_K3_WAVES
#define TK_K3_WAVES 8
#endif
constexpr int K3_WAVES = TK_K3_WAVES;   // waves per K3 block: 8 x CH/8 rows = 1 chunk
constexpr int ZERO_ROW_EXP = -(1 << 28);    // exponent of an all-zero matrix row

// tuning knobs (tools/logz_lab.hip rebuilds this file with -D overrides)
#ifndef TK_K1_NT_LOAD
#define TK_K1_NT_LOAD 0
#endif
#ifdef TK_LAB_TIMING
__device__ long long tk_dbg[64];
#define TK_STAMP(k) do { if (blockIdx.x == 1 && blockIdx.y == 20 && threadIdx.x == 0) tk_dbg[k] = clock64(); } while (0)
#else
#define TK_STAMP(k)
#endif
#ifndef TK_K3_NT_STORE
#define TK_K3_NT_STORE 1
#endif
#ifndef TK_K3_REV
#define TK_K3_REV 0
#endif
#ifndef TK_K1_REV
#define TK_K1_REV 0
#endif

// ---------------------------------------------------------------------------
// XMat: a 2nb x 2nb transfer matrix  value[i][j] = m[i][j] * 2^e[i] * exp(M)
// stored per read as NF4 float4 (lane = read => [mat][q][Npad] float4 layout,
// every access is a coalesced 1 KiB wave transaction).
// ---------------------------------------------------------------------------
template <int NB>
struct XMat {
    static constexpr int NS = 2 * NB;
    static constexpr int NW = NS * NS + NS + 2;         // payload dwords
    static constexpr int NF4 = (NW + 3) / 4;
    float m[NS][NS];
    int e[NS];
    double M;

    __device__ __forceinline__ void set_identity() {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            e[i] = 0;
#pragma unroll
            for (int j = 0; j < NS; ++j) m[i][j] = (i == j) ? 1.f : 0.f;
        }
        M = 0.0;
    }
    __device__ __forceinline__ float word(int k) const {
        if (k < NS * NS) return m[k / NS][k % NS];
        if (k < NS * NS + NS) return __int_as_float(e[k - NS * NS]);
        if (k == NS * NS + NS) return __int_as_float(__double2loint(M));
        if (k == NS * NS + NS + 1) return __int_as_float(__double2hiint(M));
        return 0.f;
    }
    // exact power-of-two renormalisation of every row
    __device__ __forceinline__ void renorm() {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            float mx = m[i][0];
#pragma unroll
            for (int j = 1; j < NS; ++j) mx = fmaxf(mx, m[i][j]);
            if (mx > 0.f) {
                const int ex = __builtin_amdgcn_frexp_expf(mx);
#pragma unroll
                for (int j = 0; j < NS; ++j) m[i][j] = __builtin_amdgcn_ldexpf(m[i][j], -ex);
                e[i] += ex;
            } else {
                e[i] = ZERO_ROW_EXP;
            }
        }
    }
};

// out = normalise(v (x) A); returns the binary exponent taken out (v is a row
// vector of weights with max ~1)
template <int NB>
__device__ __forceinline__ int xvec_mat(const float (&v)[2 * NB], const XMat<NB> &A,
                                        float (&out)[2 * NB]) {
    constexpr int NS = 2 * NB;
    int emax = ZERO_ROW_EXP;
#pragma unroll
    for (int i = 0; i < NS; ++i)
        if (v[i] > 0.f && A.e[i] != ZERO_ROW_EXP)
            emax = max(emax, A.e[i] + __builtin_amdgcn_frexp_expf(v[i]));
    if (emax == ZERO_ROW_EXP) emax = 0;
    float vs[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) vs[i] = __builtin_amdgcn_ldexpf(v[i], max(A.e[i] - emax, -300));
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        float acc = vs[0] * A.m[0][j];
#pragma unroll
        for (int i = 1; i < NS; ++i) acc = fmaf(vs[i], A.m[i][j], acc);
        out[j] = acc;
    }
    return emax + pow2_normalise(out);
}

// out = normalise(A (x) u)  (scale is irrelevant for backward vectors)
template <int NB>
__device__ __forceinline__ void xmat_vec(const XMat<NB> &A, const float (&u)[2 * NB],
                                         float (&out)[2 * NB]) {
    constexpr int NS = 2 * NB;
    float y[NS];
    int emax = ZERO_ROW_EXP;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        float acc = A.m[i][0] * u[0];
#pragma unroll
        for (int j = 1; j < NS; ++j) acc = fmaf(A.m[i][j], u[j], acc);
        y[i] = acc;
        if (acc > 0.f && A.e[i] != ZERO_ROW_EXP)
            emax = max(emax, A.e[i] + __builtin_amdgcn_frexp_expf(acc));
    }
    if (emax == ZERO_ROW_EXP) emax = 0;
#pragma unroll
    for (int i = 0; i < NS; ++i) out[i] = __builtin_amdgcn_ldexpf(y[i], max(A.e[i] - emax, -300));
}

template <int NB>
__device__ __forceinline__ void xmat_lds_put(const XMat<NB> &A, float *region, int lane) {
#pragma unroll
    for (int k = 0; k < XMat<NB>::NW; ++k) region[k * WAVE + lane] = A.word(k);
}

template <int NB>
__device__ __forceinline__ void xmat_lds_get(XMat<NB> &A, const float *region, int lane) {
    constexpr int NS = 2 * NB;
#pragma unroll
    for (int k = 0; k < NS * NS; ++k) A.m[k / NS][k % NS] = region[k * WAVE + lane];
#pragma unroll
    for (int k = 0; k < NS; ++k) A.e[k] = __float_as_int(region[(NS * NS + k) * WAVE + lane]);
    A.M = __hiloint2double(__float_as_int(region[(NS * NS + NS + 1) * WAVE + lane]),
                           __float_as_int(region[(NS * NS + NS) * WAVE + lane]));
}

struct LogzWs {
    f4 *Pc;         // [Npad][C][NF4] chunk transfer matrices, read-major (XMat words)
    float *Vin;     // [Npad][C][NS]    forward vector entering chunk c   (read-major)
    float *Uout;    // [Npad][C][NS]    backward vector leaving chunk c
    // fused train-step loss (tk_flipflop_loss_fused_dev): kernel A ran first; its per-read costs
    // and its gradient are already in place, this operator ADDS acc_scale * logZ resp.
    // acc_scale * posterior (acc_scale = 1 / nblk).  Null: the plain operator.
    float *loss_acc;
    float acc_scale;
    // gradient multiplier of the fused operator: scalar x optional per-read vector (a train step
    // that hands over d mean(lossvector) / d lossvector gets the FINAL gradient: no elementwise
    // pass over the (T, N, S) tensor in backward).  logZ / the loss values are not scaled.
    float grad_scale;
    const float *grad_scale_vec;
    // reads in the whole tensor: the row stride.  (A launch may cover a sub-range of the reads --
    // every pointer then starts at its first read -- see logz_launch.)
    int nstride;
};

// per-wave LDS buffer of K3 in f4 units: the row-set transpose buffer, which
// doubles as storage for the wave's K3_ROWS forward vectors
template <int NB, int CH>
__host__ __device__ constexpr int k3_buf_f4() {
    constexpr int a = WAVE * FF<NB>::PIECES;
    constexpr int b = (CH / K3_WAVES) * FF<NB>::NS * WAVE / 4;
    return a > b ? a : b;
}

// ---------------------------------------------------------------------------
// K1: chunk transfer matrices.  ONE WAVE PER CHUNK: the wave folds all CH rows of its
// chunk into one 2nb x 2nb matrix in registers, so there is no cross-wave product at
// all (combining per-wave partial matrices cost 15-25 % of the kernel, LDS-bound) and
// no block barrier; the four waves of a block are four consecutive chunks of one
// column.  The stream is HBM-bound: a row costs ~2.5k cycles of issue against the
// ~4k cycles the memory system needs to deliver it with every SIMD loading.
// grid = (ncols, ceil(C / 4)), block = 256.
// ---------------------------------------------------------------------------
// RING > 0: the score rows go from HBM straight into a wave-private LDS ring of RING row-sets
// (global_load_lds_dwordx4, gfx950) instead of through registers, RING - 1 rows in flight behind
// the one being consumed.  It costs no registers (the register path cannot afford a second
// row-set in flight next to the 128-register matrix ping-pong) and no ds_write pass; it costs
// LDS -- RING x 10 KiB per wave -- so it is the form for launches that fill the chip with one
// wave per SIMD anyway.
typedef __attribute__((address_space(3))) void tk_lds_void;
typedef const __attribute__((address_space(1))) void tk_global_void;

template <int NB, int CH, int RING, bool NT>
__global__ __launch_bounds__(K1_WAVES *WAVE, 2) void logz_transfer_kernel(
    const float *__restrict__ scores, int T, int N, int C, int Npad, LogzWs ws) {
    using F = FF<NB>;
    using X = XMat<NB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int c = (TK_K1_REV ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y) * K1_WAVES + wave;
    if (c >= C) return;                         // wave-uniform; the kernel has no barriers
    // wave-private LDS: the row-set transpose buffer, later the matrix image
    constexpr int IMG_F4 = (X::NF4 > F::PIECES * (RING > 0 ? RING : 1) ? X::NF4 : F::PIECES * (RING > 0 ? RING : 1));
    constexpr int IMG_WORDS = IMG_F4 * 4 * WAVE;
    float *img = reinterpret_cast<float *>(smem) + (size_t)wave * IMG_WORDS;
    f4 *buf = reinterpret_cast<f4 *>(img);
    const int n0 = blockIdx.x * WAVE;
    const int nvalid = min(WAVE, N - n0) * F::PIECES;
    const int t0 = c * CH, t1 = min(T, t0 + CH);
    const size_t rowstride = (size_t)ws.nstride * F::S;
    const float *base = scores + (size_t)n0 * F::S;

    // The running product is kept as PAIRS OF ROWS (Pp[ip][j] = rows 2ip, 2ip+1 at column
    // j): both rows of a pair take the same step, so every multiply-add is one
    // v_pk_fma_f32 with the score broadcast, and the outputs land in the same layout --
    // no operand shuffles between steps (hipcc finds the pairing for a float[NS][NS] too,
    // but then moves ~130 registers per row to feed it).
    constexpr int NS = F::NS, NP = NS / 2;
    f2 Pp[NP][NS], Pq[NP][NS];      // ping-pong: a step reads one set and writes the other
    int pe[NS];
    double pM;
    // Rows are renormalised every fourth step; a row whose maximum fell below 2^-60 in
    // between (per-step score ranges of tens of nats -- no tanh-bounded network output does
    // that) has lost low-order entries to underflow: `shrunk` makes the wave redo its chunk
    // renormalising after every step.
    bool shrunk = false;
    auto renorm = [&](f2 (&A)[NP][NS]) {           // exact power-of-two renormalisation of every row
#pragma unroll
        for (int ip = 0; ip < NP; ++ip) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float mx = A[ip][0][h];
#pragma unroll
                for (int j = 1; j < NS; ++j) mx = fmaxf(mx, A[ip][j][h]);
                shrunk |= mx < 0x1p-60f;
                if (mx > 0.f) {
                    const int ex = __builtin_amdgcn_frexp_expf(mx);
#pragma unroll
                    for (int j = 0; j < NS; ++j) A[ip][j][h] = __builtin_amdgcn_ldexpf(A[ip][j][h], -ex);
                    pe[2 * ip + h] += ex;
                } else {
                    pe[2 * ip + h] = ZERO_ROW_EXP;
                }
            }
        }
    };
    // B = A (x) row: the NP row pairs are NP independent accumulator chains, issued
    // interleaved (a packed op's result cannot feed the very next instruction)
    auto step = [&](const f2 (&A)[NP][NS], f2 (&B)[NP][NS], const RowSet<NB> &cur) {
#pragma unroll
        for (int to = 0; to < NB; ++to) {
            f2 acc[NP];
            const float w0 = cur.get(to * NS);
#pragma unroll
            for (int ip = 0; ip < NP; ++ip) acc[ip] = A[ip][0] * f2{w0, w0};
#pragma unroll
            for (int from = 1; from < NS; ++from) {
                const float w = cur.get(to * NS + from);
#pragma unroll
                for (int ip = 0; ip < NP; ++ip) acc[ip] = __builtin_elementwise_fma(A[ip][from], f2{w, w}, acc[ip]);
            }
#pragma unroll
            for (int ip = 0; ip < NP; ++ip) B[ip][to] = acc[ip];
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float wst = cur.get(F::FLOP0 + NB + b), wf = cur.get(F::FLOP0 + b);
#pragma unroll
            for (int ip = 0; ip < NP; ++ip)
                B[ip][NB + b] = __builtin_elementwise_fma(A[ip][b], f2{wf, wf}, A[ip][NB + b] * f2{wst, wst});
        }
    };
    pM = 0.0;
#pragma unroll
    for (int ip = 0; ip < NP; ++ip) {
        pe[2 * ip] = pe[2 * ip + 1] = 0;
#pragma unroll
        for (int j = 0; j < NS; ++j) Pp[ip][j] = f2{(2 * ip == j) ? 1.f : 0.f, (2 * ip + 1 == j) ? 1.f : 0.f};
    }
    TK_STAMP(8);
    auto rowptr = [&](int t) { return base + (size_t)min(t, t1 - 1) * rowstride; };
    if constexpr (RING > 0) {
        constexpr int SLOT = WAVE * F::PIECES;          // f4 per row-set
        auto issue = [&](int slot, int t) {
            const f4 *src = reinterpret_cast<const f4 *>(rowptr(t));
            f4 *dst = buf + slot * SLOT;
#pragma unroll
            for (int q = 0; q < F::PIECES; ++q)
                __builtin_amdgcn_global_load_lds((tk_global_void *)(src + min(q * WAVE + lane, nvalid - 1)),
                                                 (tk_lds_void *)(dst + q * WAVE), 16, 0, 0);
        };
        // rows issued after row t are still allowed in flight when row t is read: vmcnt counts
        // loads in order, PIECES per row
        auto landed = [&](int t) {
            const int behind = min(RING - 1, t1 - 1 - t);
            if (behind >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * F::PIECES) : "memory");
            else if (behind == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(F::PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        auto take = [&](RowSet<NB> &r, int slot) {     // pieces -> own row, straight from the ring
#pragma unroll
            for (int q = 0; q < F::PIECES; ++q) r.v[q] = buf[slot * SLOT + lane * F::PIECES + q];
        };
        static_assert(RING == 0 || RING == 3, "landed() is written for two rows in flight");
#pragma unroll
        for (int k = 0; k < RING; ++k)
            if (t0 + k < t1) issue(k, t0 + k);
        int slot = 0;
        RowSet<NB> r0, r1;
        for (int t = t0; t < t1; t += 2) {
            landed(t);
            take(r0, slot);
            pM += (double)r0.exp_normalise();           // every piece of the slot is in registers now
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t + RING < t1) issue(slot, t + RING);
            slot = (slot + 1 == RING) ? 0 : slot + 1;
            step(Pp, Pq, r0);
            if (t == t0) TK_STAMP(9);
            if (t + 1 >= t1) {              // odd tail: the product sits in the other set
#pragma unroll
                for (int ip = 0; ip < NP; ++ip)
#pragma unroll
                    for (int j = 0; j < NS; ++j) Pp[ip][j] = Pq[ip][j];
                break;
            }
            landed(t + 1);
            take(r1, slot);
            pM += (double)r1.exp_normalise();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t + 1 + RING < t1) issue(slot, t + 1 + RING);
            slot = (slot + 1 == RING) ? 0 : slot + 1;
            step(Pq, Pp, r1);
            if (((t - t0) & 3) == 2) renorm(Pp);
            if (t == t0 + 8) TK_STAMP(10);
        }
        renorm(Pp);
    } else {
        // one row-set in flight ahead of the one being consumed; row indices are clamped
        // (never branched on) so the load stream has no control flow
        RowSet<NB> r0, r1;
        auto fetch = [&](RowSet<NB> &r, int t) {
            if (TK_K1_NT_LOAD || NT) r.issue_nt(rowptr(t), nvalid, lane);      // compile-time: no branch in the load stream
            else r.issue(rowptr(t), nvalid, lane);
        };
        fetch(r0, t0);
        for (int t = t0; t < t1; t += 2) {
            fetch(r1, t + 1);
            r0.to_rows(buf, lane);
            pM += (double)r0.exp_normalise();
            step(Pp, Pq, r0);
            if (t == t0) TK_STAMP(9);
            if (t + 1 >= t1) {              // odd tail: the product sits in the other set
#pragma unroll
                for (int ip = 0; ip < NP; ++ip)
#pragma unroll
                    for (int j = 0; j < NS; ++j) Pp[ip][j] = Pq[ip][j];
                break;
            }
            fetch(r0, t + 2);
            r1.to_rows(buf, lane);
            pM += (double)r1.exp_normalise();
            step(Pq, Pp, r1);
            if (((t - t0) & 3) == 2) renorm(Pp);
            if (t == t0 + 8) TK_STAMP(10);
        }
        renorm(Pp);
    }
    X P;
#pragma unroll
    for (int ip = 0; ip < NP; ++ip) {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            P.m[2 * ip][j] = Pp[ip][j][0];
            P.m[2 * ip + 1][j] = Pp[ip][j][1];
        }
        P.e[2 * ip] = pe[2 * ip];
        P.e[2 * ip + 1] = pe[2 * ip + 1];
    }
    P.M = pM;
    if (__any(shrunk)) {            // wave-uniform, rare: redo the chunk one careful step at a time
        P.set_identity();
        RowSet<NB> r;
        for (int t = t0; t < t1; ++t) {
            r.issue(rowptr(t), nvalid, lane);
            r.to_rows(buf, lane);
            P.M += (double)r.exp_normalise();
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                float out[NS];
                ff_fwd_step<NB>(P.m[i], r, out);
#pragma unroll
                for (int j = 0; j < NS; ++j) P.m[i][j] = out[j];
            }
            P.renorm();
        }
    }
    // Pc is READ-major ([read][chunk][NF4] float4): the middle kernel streams one
    // read's matrices as a single contiguous run (a [chunk][q][read] layout put
    // all of a read's pieces 16*Npad bytes apart = on ONE L2 channel).  The wave turns
    // its 64 matrices through LDS in exactly that order: lane r writes its NF4 float4
    // contiguously (lane stride 304 B = 12 banks mod 64: conflict-free ds_write_b128),
    // then float4 number p of the image is piece (read p / NF4, q p % NF4): a linear
    // ds_read_b128.  (A [word][read] image made every gather read ~19-way conflicted.)
    TK_STAMP(11);
    wave_lds_fence();
    f4 *img4 = reinterpret_cast<f4 *>(img);
#pragma unroll
    for (int q = 0; q < X::NF4; ++q) {
        f4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = P.word(4 * q + k);
        img4[lane * X::NF4 + q] = o;
    }
    wave_lds_fence();
    f4 *dst = ws.Pc + ((size_t)n0 * C + c) * X::NF4;
#pragma unroll
    for (int k = 0; k < X::NF4; ++k) {
        const int p = lane + k * WAVE;
        const int r = p / X::NF4, q = p - r * X::NF4;
        dst[(size_t)r * C * X::NF4 + q] = img4[p];
    }
    TK_STAMP(12);
}

// ---------------------------------------------------------------------------
// K1, cooperative form for SMALL problems (fewer chunks than SIMDs): four waves share
// one chunk, wave w folds rows [w CH/4, (w+1) CH/4) and the four partial matrices are
// multiplied row-parallel through LDS.  The products cost 15-25 % of the kernel, but a
// small tensor has no other way to put more than C * ncols waves on the chip.
// grid = (ncols, C), block = 256 = one chunk.
// ---------------------------------------------------------------------------
template <int NB, int CH>
__global__ __launch_bounds__(K1_WAVES *WAVE, 2) void logz_transfer_coop_kernel(
    const float *__restrict__ scores, int T, int N, int C, int Npad, LogzWs ws) {
    using F = FF<NB>;
    using X = XMat<NB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int c = blockIdx.y;
    f4 *buf = reinterpret_cast<f4 *>(smem) + wave * (WAVE * F::PIECES);
    const int n0 = blockIdx.x * WAVE;
    const int nvalid = min(WAVE, N - n0) * F::PIECES;
    constexpr int K1_ROWS = CH / K1_WAVES;
    const int t0 = c * CH + wave * K1_ROWS, t1 = min(T, t0 + K1_ROWS);
    const size_t rowstride = (size_t)ws.nstride * F::S;
    const float *base = scores + (size_t)n0 * F::S;

    X P;
    P.set_identity();
    if (t0 < t1) {
        // one row-set in flight ahead of the one being consumed (2000 waves at
        // T=4000/N=256 give the memory system its depth); row indices are clamped
        // (never branched on) so the load stream has no control flow
        RowSet<NB> r0, r1;
        auto rowptr = [&](int t) { return base + (size_t)min(t, t1 - 1) * rowstride; };
        auto consume = [&](RowSet<NB> &cur, int t) {
            cur.to_rows(buf, lane);
            P.M += (double)cur.exp_normalise();
#pragma unroll
            for (int i = 0; i < F::NS; ++i) {
                float out[F::NS];
                ff_fwd_step<NB>(P.m[i], cur, out);
#pragma unroll
                for (int j = 0; j < F::NS; ++j) P.m[i][j] = out[j];
            }
            P.renorm();         // every step: this form serves small problems, the cost is noise
        };
        auto fetch = [&](RowSet<NB> &r, int t) {
            if (TK_K1_NT_LOAD) r.issue_nt(rowptr(t), nvalid, lane);
            else r.issue(rowptr(t), nvalid, lane);
        };
        fetch(r0, t0);
        for (int t = t0; t < t1; t += 2) {
            fetch(r1, t + 1);
            consume(r0, t);
            if (t + 1 >= t1) break;
            fetch(r0, t + 2);
            consume(r1, t + 1);
        }
        if (t0 >= t1) P.renorm();       // no rows (ragged last chunk): the identity, normalised like the rest
    }
    // combine P0 P1 P2 P3 row-parallel: row r of the product is (row r of P0) (x) P1
    // (x) P2 (x) P3, three cheap mat-vecs, and the rows are spread over the 4 waves
    // (a serial 8x8 x 8x8 tree kept three waves idle for ~30 % of the block's life).
    float *mats = reinterpret_cast<float *>(smem);          // [K1_WAVES][NW][64], overlays the buffers
    __syncthreads();
    xmat_lds_put<NB>(P, mats + (size_t)wave * X::NW * WAVE, lane);
    __syncthreads();
    constexpr int RPW = (F::NS + K1_WAVES - 1) / K1_WAVES;  // rows per wave
    float rows[RPW][F::NS];
    int rexp[RPW];
    double msum = 0.0;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = wave * RPW + q;
        if (r < F::NS) {
            float v[F::NS];
#pragma unroll
            for (int k = 0; k < F::NS; ++k) v[k] = mats[(r * F::NS + k) * WAVE + lane];
            long long eacc = __float_as_int(mats[(F::NS * F::NS + r) * WAVE + lane]);
            const bool zero_row = eacc == ZERO_ROW_EXP;
#pragma unroll
            for (int m = 1; m < K1_WAVES; ++m) {
                X A;
                xmat_lds_get<NB>(A, mats + (size_t)m * X::NW * WAVE, lane);
                float out[F::NS];
                eacc += xvec_mat<NB>(v, A, out);
                if (q == 0) msum += A.M;
#pragma unroll
                for (int k = 0; k < F::NS; ++k) v[k] = out[k];
            }
            float mx = v[0];
#pragma unroll
            for (int k = 1; k < F::NS; ++k) mx = fmaxf(mx, v[k]);
            rexp[q] = (zero_row || !(mx > 0.f)) ? ZERO_ROW_EXP : (int)eacc;
#pragma unroll
            for (int k = 0; k < F::NS; ++k) rows[q][k] = v[k];
        }
    }
    if (wave == 0) msum += P.M;         // M of P0 (each wave added P1..P3 once)
    __syncthreads();                    // all reads of `mats` done
    // assemble the chunk matrix in LDS in OUTPUT order: float4 number r * NF4 + q is piece q
    // of read r, so the store below is a linear, conflict-free ds_read_b128 (a [word][read]
    // image made the gather ~19-way bank-conflicted)
    float *tot = mats;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = wave * RPW + q;
        if (r < F::NS) {
#pragma unroll
            for (int k = 0; k < F::NS; ++k) tot[lane * (4 * X::NF4) + r * F::NS + k] = rows[q][k];
            tot[lane * (4 * X::NF4) + F::NS * F::NS + r] = __int_as_float(rexp[q]);
        }
    }
    if (wave == 0) {
        tot[lane * (4 * X::NF4) + F::NS * F::NS + F::NS] = __int_as_float(__double2loint(msum));
        tot[lane * (4 * X::NF4) + F::NS * F::NS + F::NS + 1] = __int_as_float(__double2hiint(msum));
#pragma unroll
        for (int k = X::NW; k < 4 * X::NF4; ++k) tot[lane * (4 * X::NF4) + k] = 0.f;
    }
    __syncthreads();
    {
        // Pc is READ-major ([read][chunk][NF4] float4), see logz_transfer_kernel
        const f4 *tot4 = reinterpret_cast<const f4 *>(tot);
        f4 *dst = ws.Pc + ((size_t)n0 * C + c) * X::NF4;
        for (int p = threadIdx.x; p < WAVE * X::NF4; p += K1_WAVES * WAVE) {
            const int r = p / X::NF4, q = p - r * X::NF4;
            dst[(size_t)r * C * X::NF4 + q] = tot4[p];
        }
    }
}

// ---------------------------------------------------------------------------
// The three small "middle" kernels work with EIGHT LANES PER READ (lane g of the
// group owns row g / column g of the 2nb x 2nb matrices; groups of 8 are
// half-rows of the DPP network, so group max is 3 DPP ops and a broadcast is one
// ds_bpermute).  A serial step is ~40 instructions instead of a 64-FMA mat-vec
// per lane, and a whole chain's matrices fit in registers up front.
// Word k of read n lives at f4 index (k / 4) * Npad + n, component k % 4.
// ---------------------------------------------------------------------------
constexpr int GRP = 8;

__device__ __forceinline__ float grp_bcast(float x, int i) {
    return __shfl(x, (lane_id() & ~(GRP - 1)) | i, WAVE);
}
__device__ __forceinline__ int grp_bcast(int x, int i) {
    return __shfl(x, (lane_id() & ~(GRP - 1)) | i, WAVE);
}
// 8-lane group maximum: three v_max_*_dpp (the DPP operand is folded into the max; hipcc
// emits mov + s_nop + mov_dpp + max for the intrinsic form, and these sit on the serial
// chain of every step).  s_nop 1 = the two wait states a DPP read needs after a VALU write.
__device__ __forceinline__ int grp_max_i(int x) {
    int r;
    asm("s_nop 1\n\t"
        "v_max_i32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(r) : "v"(x));
    return r;
}
// 8-lane group sum, result in every lane of the group (same three DPP steps as the maximum)
__device__ __forceinline__ float grp_sum_f(float x) {
    float r;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ float grp_max_f(float x) {
    float r;
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(r) : "v"(x));
    return r;
}
// Shares keep their NS words in "DPP order": the in-group dot product broadcasts lane
// 4q+k of the lane's own quad q (quad_perm [k,k,k,k]) and lane 7-4q-k of the other quad
// (the same quad_perm applied to the half-mirrored value), so word k of `a` pairs with
// state 4q+k and word 4+k with state 7-4q-k.  Eight broadcasts = 1 + 8 DPP movs on the
// VALU; no ds_bpermute round trip through the LDS crossbar on the serial chain.
template <int NS>
__device__ __forceinline__ int dpp_state(int g, int w) {           // state paired with word w
    const int q4 = (g >> 2) << 2;
    return (w < 4) ? q4 + w : 7 - q4 - (w - 4);
}
// sum_w a[w] * x[state(w)] over the group (x = 0 in lanes >= NS): one mirror + eight
// multiply-adds whose first operand is the DPP-broadcast lane (v_fmac_f32_dpp)
__device__ __forceinline__ float grp_dot(const float (&a)[8], float x) {
    float acc, xm;
    asm("s_nop 1\n\t"
        "v_mov_b32_dpp %1, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %0, %2, %3 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %5 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %6 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %7 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %9 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %10 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
        : "=&v"(acc), "=&v"(xm)
        : "v"(x), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
    return acc;
}

// one lane's share of a matrix for v (x) A: column g + exponent of row g
template <int NB>
struct ColShare {
    float a[8];
    int e;
    __device__ __forceinline__ void load(const float *img, int g) {     // img: one read's words
        constexpr int NS = 2 * NB;
        const int gc = min(g, NS - 1);
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int i = dpp_state<NS>(g, w);
            const float x = img[min(i, NS - 1) * NS + gc];
            a[w] = (i < NS) ? x : 0.f;
        }
        e = (g < NS) ? __float_as_int(img[NS * NS + gc]) : ZERO_ROW_EXP;
    }
};

// one lane's share for A (x) u: row g + exponent of row g
template <int NB>
struct RowShare {
    float a[8];
    int e;
    __device__ __forceinline__ void load(const float *img, int g) {
        constexpr int NS = 2 * NB;
        const int gc = min(g, NS - 1);
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int j = dpp_state<NS>(g, w);
            const float x = img[gc * NS + min(j, NS - 1)];
            a[w] = (j < NS) ? x : 0.f;
        }
        e = (g < NS) ? __float_as_int(img[NS * NS + gc]) : ZERO_ROW_EXP;
    }
};

// the fp64 log-scale M of a matrix image
template <int NB>
__device__ __forceinline__ double img_M(const float *img) {
    constexpr int NS = 2 * NB;
    return __hiloint2double(__float_as_int(img[NS * NS + NS + 1]), __float_as_int(img[NS * NS + NS]));
}

// (Round 5, measured and not kept: a second, TRANSPOSED copy of every matrix in the LDS image, both halves 16-byte
// aligned, so that a column / row share is two ds_read_b128 instead of nine scalar reads -- a third of a step's ~70
// instructions.  Stamps at T = 4000 / N = 256: scan 5212 -> 5016 cycles, expand 4608 -> 4408 (-4 %): the steps are
// bound by the latency of their dependent 8-lane chain, not by their instruction count -- and the scatter that
// writes the transposed half took stage + combine from 13.0 to 32.7 thousand cycles: middle kernel 14.3 -> 25.6 us.)
//
// The serial steps below are issued by ONE wave in order, so every instruction in the
// loop body is latency: bookkeeping that is not on the dependency chain (the fp64 M sums,
// 64-bit exponent sums, register copies of the prefetched share) is kept out of them.
//
// v (lane g holds v[g]) <- v (x) A up to a power of two; returns the exponent taken out
// (group-uniform).  The rows of A have maxima in [1/2, 1) and the aligned weights have
// their maximum in [1/2, 1), so the result lies in [2^-2, 2^3): no second normalisation
// is needed -- the next step aligns on the exponents of whatever v holds.  A zero row
// (e = ZERO_ROW_EXP) keeps t hugely negative by itself; if every lane is zero the step
// returns an arbitrary finite exponent with v = 0, which the callers flag.
template <int NB>
__device__ __forceinline__ int grp_vec_mat(float &v, const ColShare<NB> &A, int g) {
    constexpr int NS = 2 * NB;
    const int t = (v > 0.f) ? A.e + __builtin_amdgcn_frexp_expf(v) : ZERO_ROW_EXP;
    const int emax = max(grp_max_i(t), -(1 << 20));
    const float vs = __builtin_amdgcn_ldexpf(v, max(A.e - emax, -300));     // row g's scaled weight
    const float acc = grp_dot(A.a, vs);
    v = (g < NS) ? acc : 0.f;
    return emax;
}

// u (lane g holds u[g]) <- normalise(A (x) u)
template <int NB>
__device__ __forceinline__ void grp_mat_vec(float &u, const RowShare<NB> &A, int g) {
    constexpr int NS = 2 * NB;
    float acc = grp_dot(A.a, u);
    if (g >= NS) acc = 0.f;
    const int t = (acc > 0.f) ? A.e + __builtin_amdgcn_frexp_expf(acc) : ZERO_ROW_EXP;
    const int emax = max(grp_max_i(t), -(1 << 20));
    u = __builtin_amdgcn_ldexpf(acc, max(A.e - emax, -300));
}

__device__ __forceinline__ double wave_sum_f64(double x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, WAVE);
    return x;
}

// LDS image of the middle kernel: [C] chunk matrices, [NSUP] super totals, 2 x [NSUP][GRP]
// boundary vectors, and one matrix of slack (the one-ahead prefetches run past the end)
template <int NB>
__host__ __device__ constexpr size_t logz_middle_lds_bytes(int C, int NSUP) {
    constexpr size_t NFW = XMat<NB>::NW + (XMat<NB>::NW & 1);
    return ((size_t)(C + NSUP + 1) * NFW + 2 * (size_t)NSUP * 8) * sizeof(float);
}

// ---------------------------------------------------------------------------
// K2 (fused middle): combine + scan + expand in ONE launch, one 4-wave block per
// read.  The read's chunk matrices (C x 304 B, 38 KB at T = 4000) are staged in
// LDS once; then
//   combine : super s -> wave s % 4; the 8 groups of the wave are the 8 rows of the
//             running product (row_i(AB) = row_i(A) B), 8 chained mat-vecs each
//   scan    : forward by group 0 of wave 0 (-> logZ), backward by group 0 of wave 1
//   expand  : 2*NSUP independent 8-step chains over the 32 groups of the block
// Everything between the stage-in and the final stores runs out of LDS/registers,
// and the three launch/ramp/latency floors (~5 us each) collapse into one.
// grid = N, block = 256.
// ---------------------------------------------------------------------------
constexpr int K2_WAVES = 16;


template <int NB, int SUP>
__global__ __launch_bounds__(K2_WAVES *WAVE, 8) void logz_middle_kernel(int N, int C, int NSUP, int Npad,
                                                                  LogzWs ws,
                                                                  float *__restrict__ logz,
                                                                  int want_grad,
                                                                  uint32_t *__restrict__ status) {
    using F = FF<NB>;
    using X = XMat<NB>;
    // LDS stride of one matrix image: the NW payload words, not the NF4 padded float4s -- at
    // 250 chunks that is what lets TWO blocks share a CU's 160 KB (2 x 80 KB), so one
    // read's serial scan overlaps another's work-bound combine (<= 64 VGPRs for the same)
    constexpr int NFW = X::NW + (X::NW & 1);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *pcimg = reinterpret_cast<float *>(smem);             // [C][NFW]
    float *totimg = pcimg + (size_t)C * NFW;                    // [NSUP][NFW]
    float *vsl = totimg + (size_t)NSUP * NFW;                   // [NSUP][GRP]
    float *usl = vsl + (size_t)NSUP * GRP;                      // [NSUP][GRP]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & (WAVE - 1);
    const int g = lane & (GRP - 1), grp = lane >> 3;
    const size_t n = blockIdx.x;

#ifdef TK_LAB_TIMING
#undef TK_STAMP
#define TK_STAMP(k) do { if (blockIdx.x == 100 && threadIdx.x == 0) tk_dbg[k] = clock64(); } while (0)
#endif
    TK_STAMP(0);
    // ---- 1. stage the read's chunk matrices, SUPER BY SUPER: wave s % K2_WAVES loads the SUP matrices of
    //         super s (one contiguous run of <= SUP NF4 float4) and goes straight on to combine them -- nothing
    //         but this wave reads them before the barrier in front of the scan, so there is no workgroup-wide
    //         wait between the loads and the first products (round 4; before, all 16 waves staged the whole
    //         image together and met at a barrier: 7200 of the kernel's 25800 cycles at T = 4000).
    //         The image is repacked from 4 NF4 to NFW words per matrix (8-byte aligned: two 64-bit LDS stores
    //         per float4, the padding words are dropped).
    auto stage_super = [&](int s) {
        constexpr int PASS = (SUP * X::NF4 + WAVE - 1) / WAVE;     // float4 per lane: every load in flight at once
        const int c0 = s * SUP, cnt = min(C - c0, SUP), nf = cnt * X::NF4;
        const f4 *src = ws.Pc + (n * (size_t)C + c0) * X::NF4;
        f4 tmp[PASS];
#pragma unroll
        for (int k = 0; k < PASS; ++k) tmp[k] = src[min(k * WAVE + lane, nf - 1)];
#pragma unroll
        for (int k = 0; k < PASS; ++k) {
            const int idx = k * WAVE + lane;
            if (idx < nf) {
                const int c = idx / X::NF4, q = idx - c * X::NF4;
                float *dst = pcimg + (size_t)(c0 + c) * NFW + 4 * q;
                *reinterpret_cast<f2 *>(dst) = f2{tmp[k][0], tmp[k][1]};
                if (4 * q + 2 < NFW) *reinterpret_cast<f2 *>(dst + 2) = f2{tmp[k][2], tmp[k][3]};
            }
        }
        wave_lds_fence();       // (this wave's stores before this wave's reads; other waves wait for the barrier below)
    };
    TK_STAMP(1);

    // ---- 2. combine: group `grp` of the wave carries row `grp` of the super's product.
    //         Fully unrolled over the super's chunks: LDS offsets are immediates, the
    //         shares ping-pong between two register sets one matrix ahead.
    for (int s = wave; s < NSUP; s += K2_WAVES) {
        stage_super(s);
        const int c0 = s * SUP, cnt = min(C - c0, SUP);
        const float *img0 = pcimg + (size_t)c0 * NFW;
        // row `grp` of the first matrix IS the running product after one chunk
        const int rr = min(grp, F::NS - 1);
        float v = (g < F::NS && grp < F::NS) ? img0[rr * F::NS + min(g, F::NS - 1)] : 0.f;
        int eacc = __float_as_int(img0[F::NS * F::NS + rr]);
        if (eacc == ZERO_ROW_EXP) {
            v = 0.f;
            eacc = 0;
        }
        ColShare<NB> A[2];
        A[1].load(img0 + NFW, g);
        // the fp64 log-scales ride along (one broadcast LDS read and one add per step, off the
        // dependency chain) instead of a six-stage cross-lane sum after it
        double macc = img_M<NB>(img0);
#pragma unroll
        for (int i = 1; i < SUP; ++i) {
            if (i < cnt) {
                A[(i + 1) & 1].load(img0 + (i + 1) * NFW, g);      // may run one matrix past the super
                macc += img_M<NB>(img0 + i * NFW);
                eacc += grp_vec_mat<NB>(v, A[i & 1], g);
            }
        }
        const float mx = grp_max_f(v);
        {   // rows of the stored totals are normalised like K1's (maximum in [1/2, 1))
            const int ex = (mx > 0.f) ? __builtin_amdgcn_frexp_expf(mx) : 0;
            v = __builtin_amdgcn_ldexpf(v, -ex);
            eacc += ex;
        }
        float *t = totimg + (size_t)s * NFW;
        if (grp < F::NS && g < F::NS) t[grp * F::NS + g] = v;
        if (grp < F::NS && g == 0) t[F::NS * F::NS + grp] = __int_as_float((mx > 0.f) ? eacc : ZERO_ROW_EXP);
        if (lane == 0) {
            t[F::NS * F::NS + F::NS] = __int_as_float(__double2loint(macc));
            t[F::NS * F::NS + F::NS + 1] = __int_as_float(__double2hiint(macc));
        }
    }
    __syncthreads();
    TK_STAMP(2);

    // ---- 3. scan over the super totals (every group of the wave runs the same chain, so
    //         the boundary-vector stores need no predicate)
    if (wave == 0) {
        // forward: paths start in any flip state with weight 1 (layers.py:1289-1295,
        // cupy flipflop.py:115-118)
        float v = (g < NB) ? 1.f : 0.f;
        int eacc = 0;
        ColShare<NB> A[2];
        A[0].load(totimg, g);
        double macc = 0.0;                      // the supers' fp64 log-scales ride along, off the chain
        for (int s = 0; s < NSUP; s += 2) {
            A[1].load(totimg + (size_t)(s + 1) * NFW, g);
            vsl[s * GRP + g] = v;
            macc += img_M<NB>(totimg + (size_t)s * NFW);
            eacc += grp_vec_mat<NB>(v, A[0], g);
            if (s + 1 >= NSUP) break;
            A[0].load(totimg + (size_t)(s + 2) * NFW, g);
            vsl[(s + 1) * GRP + g] = v;
            macc += img_M<NB>(totimg + (size_t)(s + 1) * NFW);
            eacc += grp_vec_mat<NB>(v, A[1], g);
        }
        const float tot = grp_sum_f(v);         // lanes >= NS of the group hold 0
        if (lane == 0) {
            const float lzf = (float)(macc + (double)eacc * 0.6931471805599453 + (double)logf(tot));
            logz[n] = lzf;
            if (ws.loss_acc != nullptr) ws.loss_acc[n] += ws.acc_scale * lzf;       // lossvector = (A) + logZ / nblk
            if (status != nullptr && !isfinite(lzf)) atomicOr(status, 1u);
        }
    } else if (wave == 1 && want_grad) {
        // backward: paths may end in any state (cupy flipflop.py:163-166); scale is free
        float u = (g < F::NS) ? 1.f : 0.f;
        RowShare<NB> A[2];
        A[0].load(totimg + (size_t)(NSUP - 1) * NFW, g);
        for (int s = NSUP - 1; s >= 0; s -= 2) {
            A[1].load(totimg + (size_t)max(s - 1, 0) * NFW, g);
            usl[s * GRP + g] = u;
            grp_mat_vec<NB>(u, A[0], g);
            if (s - 1 < 0) break;
            A[0].load(totimg + (size_t)max(s - 2, 0) * NFW, g);
            usl[(s - 1) * GRP + g] = u;
            grp_mat_vec<NB>(u, A[1], g);
        }
    }
    if (!want_grad) return;
    __syncthreads();
    TK_STAMP(3);

    // ---- 4. expand to chunk granularity: lower half of the waves forward chains, upper
    //         half backward chains; one 8-lane group per chain
    constexpr int HALF = K2_WAVES / 2;
    const bool fwd = wave < HALF;
    const int slot = (wave % HALF) * GRP + grp;
    float *dstbase = (fwd ? ws.Vin : ws.Uout) + n * (size_t)C * F::NS;     // [read][c][NS]
    for (int s = slot; s < NSUP; s += HALF * GRP) {
        const int c0 = s * SUP, cnt = min(C - c0, SUP);
        float v = (fwd ? vsl : usl)[s * GRP + g];
        if (fwd) {
            const float *img0 = pcimg + (size_t)c0 * NFW;
            float *dst = dstbase + (size_t)c0 * F::NS;
            ColShare<NB> A[2];
            A[0].load(img0, g);
#pragma unroll
            for (int i = 0; i < SUP; ++i) {
                if (i < cnt) {
                    A[(i + 1) & 1].load(img0 + (i + 1) * NFW, g);
                    if (g < F::NS) dst[i * F::NS + g] = v;
                    (void)grp_vec_mat<NB>(v, A[i & 1], g);
                }
            }
        } else {
            const float *img1 = pcimg + (size_t)(c0 + cnt - 1) * NFW;      // last chunk of the super
            float *dst = dstbase + (size_t)(c0 + cnt - 1) * F::NS;
            RowShare<NB> A[2];
            A[0].load(img1, g);
#pragma unroll
            for (int i = 0; i < SUP; ++i) {
                if (i < cnt) {
                    A[(i + 1) & 1].load(img1 - (i + 1 < cnt ? i + 1 : i) * NFW, g);
                    if (g < F::NS) dst[-i * F::NS + g] = v;
                    grp_mat_vec<NB>(v, A[i & 1], g);
                }
            }
        }
    }
    TK_STAMP(4);
}

// ---------------------------------------------------------------------------
// K3: posterior.  grid = (ncols, C), block = 512 (8 waves x 4 rows = one chunk).
// Rows live in registers; forward / backward boundary vectors are chained
// between the waves through LDS; posteriors overwrite the rows in place and
// are streamed out through the same coalescing transpose.
// ---------------------------------------------------------------------------
template <int NB, int CH, bool ACC = false>
__global__ __launch_bounds__(K3_WAVES *WAVE, (CH / K3_WAVES <= 2 ? 4 : 2)) void logz_posterior_kernel(
    const float *__restrict__ scores, float *__restrict__ grad, int T, int N, int Npad,
    LogzWs ws, uint32_t *__restrict__ status, int nt_load) {
    // ACC: add acc_scale * posterior to the gradient kernel A left in `grad` (fused loss)
    using F = FF<NB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    constexpr int K3_ROWS = CH / K3_WAVES;
    constexpr int BUF_F4 = k3_buf_f4<NB, CH>();
    f4 *buf = reinterpret_cast<f4 *>(smem) + wave * BUF_F4;
    // between the load phase and the store phase the transpose buffer is idle:
    // it keeps this wave's forward vectors (lane-private slots, no sync needed)
    float *fvb = reinterpret_cast<float *>(buf);            // [K3_ROWS][NS][64]
    // Hand-off slots of the two chains.  Big chunks: behind the buffers.  Small chunks (so
    // that two blocks fit in one CU's LDS): in the unused tail of the PRODUCING wave's own
    // buffer -- a wave writes its slot only after its own transposes are done, and reads of
    // it are over (stage barriers) long before that wave transposes its output.  (A slot in
    // another wave's buffer would race with that wave's load-phase transposes, which no
    // barrier separates from the first chain stage.)
    constexpr bool CHAIN_IN_BUF = (K3_ROWS + 2) * F::NS * WAVE <= BUF_F4 * 4;
    float *const chain_tail = reinterpret_cast<float *>(reinterpret_cast<f4 *>(smem) + K3_WAVES * BUF_F4);
    auto chainF_of = [&](int w) {       // slot written by wave w for wave w + 1
        return CHAIN_IN_BUF ? reinterpret_cast<float *>(reinterpret_cast<f4 *>(smem) + w * BUF_F4) +
                                  K3_ROWS * F::NS * WAVE
                            : chain_tail;
    };
    auto chainB_of = [&](int w) {       // slot written by wave w for wave w - 1
        return CHAIN_IN_BUF ? reinterpret_cast<float *>(reinterpret_cast<f4 *>(smem) + w * BUF_F4) +
                                  (K3_ROWS + 1) * F::NS * WAVE
                            : chain_tail + F::NS * WAVE;
    };
    const int c = TK_K3_REV ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
    const int n0 = blockIdx.x * WAVE;
    const int nvalid = min(WAVE, N - n0) * F::PIECES;
    const size_t rowstride = (size_t)ws.nstride * F::S;
    const int tw = c * CH + wave * K3_ROWS;             // first row of this wave
    const float *base = scores + (size_t)n0 * F::S;
    const size_t n = (size_t)n0 + lane;
    const float acc_g = ACC ? ws.acc_scale * ws.grad_scale *
                                  (ws.grad_scale_vec != nullptr ? ws.grad_scale_vec[min((int)n, N - 1)] : 1.0f)
                            : 1.0f;

    // 1. rows -> registers (weights w = exp(s - rowmax))
    RowSet<NB> w[K3_ROWS];
#pragma unroll
    for (int j = 0; j < K3_ROWS; ++j)
        // last use of the scores.  Tensors that fit the 256 MB Infinity Cache are read with
        // plain loads (part of this second read then hits L2 / MALL: 105 -> 100 us at
        // T=4000 N=256); bigger ones stream past the caches (non-temporal)
        if (nt_load) w[j].issue_nt(base + (size_t)min(tw + j, T - 1) * rowstride, nvalid, lane);
        else w[j].issue(base + (size_t)min(tw + j, T - 1) * rowstride, nvalid, lane);
    // chain heads (wave 0: forward vector entering the chunk, last wave: backward
    // vector leaving it), loaded while the rows are in flight
    float head[F::NS];
    {
        const float *src = ((wave == 0) ? ws.Vin : ws.Uout) + (n * gridDim.y + c) * F::NS;    // [read][c][NS]
#pragma unroll
        for (int k = 0; k < F::NS; ++k) head[k] = src[k];
    }
#pragma unroll
    for (int j = 0; j < K3_ROWS; ++j) {
        if (tw + j < T) {
            w[j].to_rows(buf, lane);
            (void)w[j].exp_normalise();
        }
    }

    // 2. chain the boundary vectors through the 8 waves
    float bexit[F::NS];             // backward vector AFTER this wave's last row
#pragma unroll 1
    for (int st = 0; st < K3_WAVES; ++st) {
        if (wave == st) {
            float f[F::NS];
#pragma unroll
            for (int k = 0; k < F::NS; ++k) f[k] = (st == 0) ? head[k] : chainF_of(st - 1)[k * WAVE + lane];
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
            for (int k = 0; k < F::NS; ++k) chainF_of(st)[k * WAVE + lane] = f[k];
        }
        if (wave == K3_WAVES - 1 - st) {
            float b[F::NS];
#pragma unroll
            for (int k = 0; k < F::NS; ++k) b[k] = (st == 0) ? head[k] : chainB_of(K3_WAVES - st)[k * WAVE + lane];
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
            for (int k = 0; k < F::NS; ++k) chainB_of(K3_WAVES - 1 - st)[k * WAVE + lane] = b[k];
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
            const float inv = (ACC ? acc_g : 1.0f) / sum;
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
            if constexpr (ACC) {
                // read-modify-write in the coalesced piece layout: + kernel A's gradient row set
                const f4 *src = reinterpret_cast<const f4 *>(gbase + (size_t)(tw + j) * rowstride);
#pragma unroll
                for (int q = 0; q < F::PIECES; ++q) w[j].v[q] += src[min(q * WAVE + lane, nvalid - 1)];
            }
            if (TK_K3_NT_STORE) w[j].store_nt(gbase + (size_t)(tw + j) * rowstride, nvalid, lane);
            else w[j].store(gbase + (size_t)(tw + j) * rowstride, nvalid, lane);
        }
    }
}

// ---------------------------------------------------------------------------
// host-side launcher
// ---------------------------------------------------------------------------
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// LDS ring form of the transfer kernel (120 KiB per block = one block per CU): for launches with
// fewer waves than SIMDs, where the extra row in flight is worth 3-9 % of the kernel (T=4000 x
// N=192: 28.1 vs 30.9 us); at a wave per SIMD (N=256) it measures the same and above that the
// second block per CU matters more.  TK_K1_RING=0/1 overrides
constexpr int K1_RING = 3;
static bool logz_use_ring(size_t nchunks) {
    if (const char *e = TK_LAB_ENV("TK_K1_RING")) return atoi(e) != 0;     // tuning / test override
    return nchunks <= 900;
}

// chunk size = rows per K3 block.  16 rows (two per wave) keep K3 at 128 VGPRs = two
// blocks per CU, so one block's serial chain overlaps the other's loads and stores;
// 8 when a 16-row grid would leave CUs without a block.  32 is kept for the env override.
static int logz_pick_ch(size_t T, size_t N) {
    const size_t ncols = (N + WAVE - 1) / WAVE;
    return ncols * ((T + 15) / 16) >= 384 ? 16 : 8;
}

template <int NB>
static size_t logz_ws_layout(size_t T, size_t N, void *base, LogzWs *ws) {
    using F = FF<NB>;
    using X = XMat<NB>;
    const size_t C = (T + 8 - 1) / 8, Npad = align_up(N, WAVE);     // sized for the smallest chunk
    size_t off = 0;
    char *p = static_cast<char *>(base);
    auto take = [&](size_t bytes) {
        char *r = p ? p + off : nullptr;
        off += align_up(bytes, 256);
        return r;
    };
    const size_t mbytes = (size_t)X::NF4 * Npad * sizeof(f4);
    f4 *Pc = reinterpret_cast<f4 *>(take(C * mbytes));
    float *Vin = reinterpret_cast<float *>(take(C * F::NS * Npad * sizeof(float)));
    float *Uout = reinterpret_cast<float *>(take(C * F::NS * Npad * sizeof(float)));
    if (ws) *ws = LogzWs{Pc, Vin, Uout, nullptr, 0.f};
    return off;
}

constexpr int PH_TRANSFER = 1, PH_MIDDLE = 2, PH_POSTERIOR = 4, PH_ALL = 7;

// `phases`: which of the three launches to enqueue (the two-queue pipeline of logz_launch issues
// them separately); `nchunks_all`: chunk count of the WHOLE operator call, which picks the form of
// the transfer kernel (a sub-range of the reads runs the form the whole call would).
template <int NB, int CH>
static int logz_launch_ch(const float *scores, size_t T, size_t N, float *logz, float *grad,
                          LogzWs ws, uint32_t *status, hipStream_t stream, int phases = PH_ALL,
                          size_t nchunks_all = 0) {
    using F = FF<NB>;
    const int C = (int)((T + CH - 1) / CH);
    const int SUP = logz_super(C), NSUP = (C + SUP - 1) / SUP;
    const int ncols = (int)((N + WAVE - 1) / WAVE), Npad = ncols * WAVE;
    if (nchunks_all == 0) nchunks_all = (size_t)ncols * C;
    if (phases & PH_TRANSFER) {
        const size_t bufwords = 4 * (size_t)WAVE * F::PIECES, imgwords = 4 * (size_t)XMat<NB>::NF4 * WAVE;
        const size_t lds = K1_WAVES * (imgwords > bufwords ? imgwords : bufwords) * sizeof(float);
        // (the ring form's LDS image can exceed the plain one's: raise for every form, whatever `lds` is)
        if (raise_dynamic_lds(reinterpret_cast<const void *>(&logz_transfer_kernel<NB, CH, 0, false>)) ||
            raise_dynamic_lds(reinterpret_cast<const void *>(&logz_transfer_kernel<NB, CH, 0, true>)) ||
            raise_dynamic_lds(reinterpret_cast<const void *>(&logz_transfer_kernel<NB, CH, K1_RING, false>)) ||
            raise_dynamic_lds(reinterpret_cast<const void *>(&logz_transfer_coop_kernel<NB, CH>)))
            return 4;
        // one wave per chunk needs about a wave per SIMD to stream at full rate; below that
        // the cooperative form (4 waves per chunk) is faster
        if (nchunks_all >= 640) {
            if (logz_use_ring((size_t)ncols * C)) {
                const size_t ringlds = K1_WAVES * (size_t)K1_RING * WAVE * F::PIECES * sizeof(f4);
                hipLaunchKernelGGL((logz_transfer_kernel<NB, CH, K1_RING, false>), dim3(ncols, (C + K1_WAVES - 1) / K1_WAVES),
                                   dim3(K1_WAVES * WAVE), ringlds > lds ? ringlds : lds, stream, scores, (int)T,
                                   (int)N, C, Npad, ws);
            } else {
                // a score tensor that fits the Infinity Cache is read with plain loads: the
                // posterior kernel's second read then hits (T=4000 x N=256: 99.5 us for the op
                // against 110 with streaming loads here); a bigger one is streamed (N=1024:
                // transfer 130 us instead of 157, the op 385 instead of 412; break-even ~300 MB).
                // Two instantiations, not a runtime flag: a branch in the load stream costs the
                // whole gain
                bool nt_load = (size_t)T * ws.nstride * F::S * sizeof(float) > ((size_t)300 << 20);
                if (const char *e = TK_LAB_ENV("TK_K1_NT")) nt_load = atoi(e) != 0;     // tuning / test override
                const dim3 grid(ncols, (C + K1_WAVES - 1) / K1_WAVES);
                if (nt_load)
                    hipLaunchKernelGGL((logz_transfer_kernel<NB, CH, 0, true>), grid, dim3(K1_WAVES * WAVE), lds, stream,
                                       scores, (int)T, (int)N, C, Npad, ws);
                else
                    hipLaunchKernelGGL((logz_transfer_kernel<NB, CH, 0, false>), grid, dim3(K1_WAVES * WAVE), lds, stream,
                                       scores, (int)T, (int)N, C, Npad, ws);
            }
        } else
            hipLaunchKernelGGL((logz_transfer_coop_kernel<NB, CH>), dim3(ncols, C), dim3(K1_WAVES * WAVE),
                               lds, stream, scores, (int)T, (int)N, C, Npad, ws);
    }
    if (phases & PH_MIDDLE) {
        const size_t lds = logz_middle_lds_bytes<NB>(C, NSUP);
        if (lds > 160 * 1024) return 2;         // too many chunks for one LDS image
        if (raise_dynamic_lds(reinterpret_cast<const void *>(&logz_middle_kernel<NB, 8>)) ||
            raise_dynamic_lds(reinterpret_cast<const void *>(&logz_middle_kernel<NB, 16>)))
            return 4;
        if (SUP == 8)
            hipLaunchKernelGGL((logz_middle_kernel<NB, 8>), dim3((unsigned)N), dim3(K2_WAVES * WAVE), lds, stream,
                               (int)N, C, NSUP, Npad, ws, logz, grad != nullptr ? 1 : 0, status);
        else
            hipLaunchKernelGGL((logz_middle_kernel<NB, 16>), dim3((unsigned)N), dim3(K2_WAVES * WAVE), lds, stream,
                               (int)N, C, NSUP, Npad, ws, logz, grad != nullptr ? 1 : 0, status);
    }
    if (grad != nullptr && (phases & PH_POSTERIOR)) {
        dim3 grid(ncols, C), block(K3_WAVES * WAVE);
        constexpr bool chain_in_buf =
            ((CH / K3_WAVES) + 2) * F::NS * WAVE <= k3_buf_f4<NB, CH>() * 4;
        const size_t lds = K3_WAVES * (size_t)k3_buf_f4<NB, CH>() * sizeof(f4) +
                           (chain_in_buf ? 0 : 2 * F::NS * WAVE * sizeof(float));
        if (raise_dynamic_lds(reinterpret_cast<const void *>(&logz_posterior_kernel<NB, CH, false>)) ||
            raise_dynamic_lds(reinterpret_cast<const void *>(&logz_posterior_kernel<NB, CH, true>)))
            return 4;
        const int nt_load = (size_t)T * ws.nstride * F::S * sizeof(float) > ((size_t)200 << 20);
        if (ws.loss_acc != nullptr)
            hipLaunchKernelGGL((logz_posterior_kernel<NB, CH, true>), grid, block, lds, stream, scores, grad,
                               (int)T, (int)N, Npad, ws, status, nt_load);
        else
            hipLaunchKernelGGL((logz_posterior_kernel<NB, CH, false>), grid, block, lds, stream, scores, grad,
                               (int)T, (int)N, Npad, ws, status, nt_load);
    }
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// A second hardware queue for the pipeline below: a HIGH-PRIORITY stream per device (HIP keeps a
// hardware queue per priority level, so it runs beside the caller's stream even when
// GPU_MAX_HW_QUEUES = 1 keeps the train step's compute on one queue, tools/queue_probe.py).
struct LogzSide {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, t1 = nullptr, join = nullptr;
    bool ok = false;
};
static LogzSide *logz_side_for_current_device() {
    static std::mutex mu;
    static LogzSide side[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(mu);
    LogzSide &sd = side[dev];
    if (sd.s == nullptr) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);       // hi = numerically lowest = highest priority
        const char *e = TK_LAB_ENV("TK_SIDE_PRIO");                 // lab: "lo" puts the side queue BELOW the caller's
        const int prio = (e != nullptr && e[0] == 'l') ? lo : hi;
        sd.ok = hipStreamCreateWithPriority(&sd.s, hipStreamNonBlocking, prio) == hipSuccess &&
                hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&sd.t1, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&sd.join, hipEventDisableTiming) == hipSuccess;
        if (!sd.ok) (void)hipGetLastError();
    }
    return sd.ok ? &sd : nullptr;
}

// (for the fused loss, c_api.hip: kernel B on this queue beside kernel A's sweeps)
bool logz_side_stream(hipStream_t *s, hipEvent_t *fork, hipEvent_t *join) {
    LogzSide *sd = logz_side_for_current_device();
    if (sd == nullptr) return false;
    *s = sd->s;
    *fork = sd->fork;
    *join = sd->join;
    return true;
}

// the workspace and the per-read vectors of a sub-range of the reads starting at read n0 (every
// per-read array is read-major)
template <int NB>
static LogzWs logz_ws_from(const LogzWs &ws, size_t n0, size_t C) {
    using F = FF<NB>;
    LogzWs o = ws;
    o.Pc = ws.Pc + n0 * C * XMat<NB>::NF4;
    o.Vin = ws.Vin + n0 * C * F::NS;
    o.Uout = ws.Uout + n0 * C * F::NS;
    if (ws.loss_acc != nullptr) o.loss_acc = ws.loss_acc + n0;
    if (ws.grad_scale_vec != nullptr) o.grad_scale_vec = ws.grad_scale_vec + n0;
    return o;
}

// TWO-QUEUE PIPELINE over the reads (big tensors): transfer -> middle -> posterior is a strict chain
// and every launch is a single round of workgroups, so nothing overlaps inside one queue; the middle
// kernel's ~14 us are pure dependent-step latency with the memory system idle.  The reads are cut in
// two halves (whole 64-read column groups) and the second half runs one stage behind the first in a
// second hardware queue:
//      queue A:  transfer(h1) | middle(h1)   | posterior(h1)              | (join)
//      queue B:       (wait)  | transfer(h2) | middle(h2) | posterior(h2) |
// -- middle(h1) runs beside transfer(h2), middle(h2) beside posterior(h1).
template <int NB, int CH>
static int logz_launch_split(const float *scores, size_t T, size_t N, float *logz, float *grad, LogzWs ws,
                             uint32_t *status, hipStream_t stream, LogzSide *sd) {
    using F = FF<NB>;
    const size_t C = (T + CH - 1) / CH;
    const size_t ncols = (N + WAVE - 1) / WAVE, n1 = (ncols / 2) * WAVE, n2 = N - n1;
    const size_t all = ncols * C;
    const LogzWs w1 = logz_ws_from<NB>(ws, 0, C), w2 = logz_ws_from<NB>(ws, n1, C);
    const float *s2 = scores + n1 * F::S;
    float *g2 = grad + n1 * F::S, *z2 = logz + n1;
    int rc = 0;
    if (hipEventRecord(sd->fork, stream) != hipSuccess || hipStreamWaitEvent(sd->s, sd->fork, 0) != hipSuccess) return 4;
    rc = logz_launch_ch<NB, CH>(scores, T, n1, logz, grad, w1, status, stream, PH_TRANSFER, all);
    if (rc) return rc;
    const char *mode = TK_LAB_ENV("TK_LOGZ_SPLIT");
    const bool stagger = !(mode && mode[0] == '2');             // lab: 2 = both halves side by side, no stagger
    if (stagger && (hipEventRecord(sd->t1, stream) != hipSuccess || hipStreamWaitEvent(sd->s, sd->t1, 0) != hipSuccess))
        return 4;
    rc = logz_launch_ch<NB, CH>(s2, T, n2, z2, g2, w2, status, sd->s, PH_ALL, all);
    if (rc) return rc;
    if (hipEventRecord(sd->join, sd->s) != hipSuccess) return 4;
    rc = logz_launch_ch<NB, CH>(scores, T, n1, logz, grad, w1, status, stream, PH_MIDDLE | PH_POSTERIOR, all);
    if (rc) return rc;
    return hipStreamWaitEvent(stream, sd->join, 0) == hipSuccess ? 0 : 4;
}

template <int NB>
static int logz_launch(const float *scores, size_t T, size_t N, float *logz, float *grad,
                       void *workspace, size_t workspace_bytes, uint32_t *status,
                       hipStream_t stream, float *loss_acc, float acc_scale, float grad_scale,
                       const float *grad_scale_vec) {
    LogzWs ws;
    const size_t need = logz_ws_layout<NB>(T, N, workspace, &ws);
    if (need > workspace_bytes) return 3;
    ws.loss_acc = loss_acc;
    ws.acc_scale = acc_scale;
    ws.grad_scale = grad_scale;
    ws.grad_scale_vec = grad_scale_vec;
    ws.nstride = (int)N;
    int ch = logz_pick_ch(T, N);
    if (const char *e = TK_LAB_ENV("TK_LOGZ_CH")) ch = atoi(e);         // tuning override
    // the middle kernel keeps one read's chunk matrices in LDS: fall back to bigger chunks
    auto middle_lds = [&](int c) {
        const int C = (int)((T + c - 1) / c);
        return logz_middle_lds_bytes<NB>(C, (C + logz_super(C) - 1) / logz_super(C));
    };
    while (ch < 32 && middle_lds(ch) > 160 * 1024) ch *= 2;
    {
        // OFF by default -- measured (round 3, T=4000): N=256 134.8 us staggered / 119.4 us side by side
        // against 101.0 us for the plain chain, N=512 229 / 221 against 204.  A half's kernels do not
        // take half the time (transfer 25.9 us against 30.6, middle 17.3 against 14.0, posterior 34.6
        // against 53.0: each wave's work is the same, only the rounds of workgroups shrink) and every
        // cross-queue event costs ~10 us.  TK_LOGZ_SPLIT=1 (2: no stagger) keeps the experiment
        // reproducible (tools/logz_sweep.py).
        const size_t ncols = (N + WAVE - 1) / WAVE;
        bool split = false;
        if (const char *e = TK_LAB_ENV("TK_LOGZ_SPLIT")) split = atoi(e) != 0 && grad != nullptr && ncols >= 2 && ch == 16;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (split && (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) split = false;
        if (split) {
            if (LogzSide *sd = logz_side_for_current_device())
                return logz_launch_split<NB, 16>(scores, T, N, logz, grad, ws, status, stream, sd);
        }
    }
    switch (ch) {
        case 8: return logz_launch_ch<NB, 8>(scores, T, N, logz, grad, ws, status, stream);
        case 16: return logz_launch_ch<NB, 16>(scores, T, N, logz, grad, ws, status, stream);
        default: return logz_launch_ch<NB, 32>(scores, T, N, logz, grad, ws, status, stream);
    }
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
                  hipStream_t stream, float *loss_acc, float acc_scale, float grad_scale,
                  const float *grad_scale_vec) {
    if (loss_acc != nullptr && grad == nullptr) return 1;
    switch (nbase) {
        case 1: return logz_launch<1>(scores, T, N, logz, grad, workspace, workspace_bytes, status, stream, loss_acc, acc_scale, grad_scale, grad_scale_vec);
        case 2: return logz_launch<2>(scores, T, N, logz, grad, workspace, workspace_bytes, status, stream, loss_acc, acc_scale, grad_scale, grad_scale_vec);
        case 3: return logz_launch<3>(scores, T, N, logz, grad, workspace, workspace_bytes, status, stream, loss_acc, acc_scale, grad_scale, grad_scale_vec);
        case 4: return logz_launch<4>(scores, T, N, logz, grad, workspace, workspace_bytes, status, stream, loss_acc, acc_scale, grad_scale, grad_scale_vec);
        default: return 2;
    }
}

}  // namespace tk
