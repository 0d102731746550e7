// ff_common.h -- device-side building blocks shared by the gfx950 flip-flop kernels.
//
// CDNA4 only: 64-lane wavefronts, LDS staging, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <set>
#include <utility>

// Lab switches.  Environment overrides of the dispatch (block length, cells per lane, feed mode, forced kernels,
// the switch that suppresses the log-domain redo, chunk sizes of kernel B ...) exist only in the LAB build
// (-DTK_LAB: libtaiyaki_amd_flipflop_lab.so, which tests/ and tools/ load when they flip one).  The release
// library reads no environment variable on a launch path: TK_LAB_ENV is a null pointer there and the branches
// behind it (and their strings) are compiled out -- tests/test_host_logic.py looks at `strings` and `nm -D`.
#ifdef TK_LAB
#include <stdlib.h>
#define TK_LAB_ENV(name) getenv(name)
#else
#define TK_LAB_ENV(name) (static_cast<const char *>(nullptr))
#endif

namespace tk {

// Dynamic LDS beyond 64 KiB needs hipFuncAttributeMaxDynamicSharedMemorySize on the kernel, PER
// DEVICE: raised once per (kernel, device) the first time a process launches it there (outside
// the steady-state launch path, so launches stay capturable into a hipGraph).
inline int raise_dynamic_lds(const void *fn, int bytes = 160 * 1024) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 4;
    std::lock_guard<std::mutex> g(mu);
    if (done.count({fn, dev})) return 0;
    // (a kernel that also has static LDS asks for less: the two together must fit the CU's 160 KiB)
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 4;
    done.insert({fn, dev});
    return 0;
}

constexpr int WAVE = 64;
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float NEG_LARGE = -1e30f;     // the reference's LARGE_VAL (c_crf_flipflop.c:11)

// Flip-flop transition layout (taiyaki/layers.py:1253-1274):
//   s[to*NS + from]      to < NB (to-flip), any from
//   s[NS*NB + from]      from < NB: flip -> own flop ; from >= NB: flop stay
template <int NB>
struct FF {
    static constexpr int NS = 2 * NB;
    static constexpr int S = NS * (NB + 1);
    static constexpr int PIECES = S / 4;    // float4 pieces per row; S % 4 == 0 always
    static constexpr int FLOP0 = NS * NB;
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// Compiler-level ordering for wave-private LDS traffic.  The LDS unit executes
// one wave's DS instructions in order, so no hardware barrier is needed; this
// only stops hipcc from moving DS ops across the point.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * LOG2E); }

// log2(2^a + 2^b) -- the reference's logaddexp (vect_mathfun.h:79-102:
// max + log(1 + exp(-|d|)), NOT log1p) in base 2: one v_exp_f32 + one v_log_f32.
__device__ __forceinline__ float lse2(float a, float b) {
    const float mx = fmaxf(a, b);
    const float d = -fabsf(a - b);
    return mx + fast_log2(1.0f + fast_exp2(d));
}

// ---- ids from flip-flop codes (flipflopfings.py:6-31, ctc.pyx:127-134, 282-292): the arithmetic of
//      build_indices_kernel (crf_kernels.hip), per cell, for the launches that build their indices themselves
//      (`A`: BandArgs / CrfArgs with codes, mod_cats, cmo, mcw, nbase, ncan)
template <class A>
__device__ __forceinline__ int lbl_code(const A &a, int64_t i) {
    return min(max(a.codes[i], 0), 2 * a.nbase - 1);            // (a bad label is clamped here and REPORTED by the checker)
}
template <class A>
__device__ __forceinline__ int lbl_stay(const A &a, int cp) { return cp + min(cp, a.nbase) * (2 * a.nbase); }
template <class A>
__device__ __forceinline__ int lbl_move(const A &a, int cp, int cn) { return cp + min(cn, a.nbase) * (2 * a.nbase); }
// the modification column (minus ncan) of the move INTO the position whose code is `cn` and whose category is `cat`
template <class A>
__device__ __forceinline__ int lbl_mod_seq(const A &a, int cn, int cat, bool *bad) {
    const int lo = a.cmo[cn % a.nbase], hi = a.cmo[cn % a.nbase + 1];
    const int mseq = lo + cat;
    if (bad != nullptr) *bad |= mseq < lo || mseq >= hi;
    return min(max(mseq, lo), hi - 1);
}

// ---- DPP cross-lane primitives (GFX9 data-parallel-primitive controls) ------
// quad_perm[1,0,3,2] = 0xB1, quad_perm[2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140, wave_shl:1 = 0x130, wave_shr:1 = 0x138.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src),
                                                      CTRL, 0xF, 0xF, false));
}

// lane l receives src of lane l-1 (lane 0 keeps `fill`): one VALU op, no LDS crossbar
__device__ __forceinline__ float wave_shift_up1(float src, float fill) {
    return dpp_f32<0x138>(fill, src);
}
// lane l receives src of lane l+1 (lane 63 keeps `fill`)
__device__ __forceinline__ float wave_shift_down1(float src, float fill) {
    return dpp_f32<0x130>(fill, src);
}
__device__ __forceinline__ int wave_shift_up1(int src, int fill) {
    return __builtin_amdgcn_update_dpp(fill, src, 0x138, 0xF, 0xF, false);
}

// max over the 64 lanes, result uniform: 4 DPP butterflies inside each 16-lane
// row, then 4 v_readlane across the rows (max is idempotent, so mirrors suffice)
__device__ __forceinline__ float wave_allmax_dpp(float x) {
    x = fmaxf(x, dpp_f32<0xB1>(x, x));
    x = fmaxf(x, dpp_f32<0x4E>(x, x));
    x = fmaxf(x, dpp_f32<0x141>(x, x));
    x = fmaxf(x, dpp_f32<0x140>(x, x));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Inclusive prefix sum over the 64 lanes with DPP only (row_shr 1/2/4/8 inside each
// 16-lane row, then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3).
// Lanes without a source keep the `old` operand = 0.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_add_masked(float x) {
    const float y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROWMASK, 0xF, false));
    return x + y;
}
__device__ __forceinline__ float wave_inclusive_scan_dpp(float x) {
    x = dpp_add_masked<0x111, 0xF>(x);      // row_shr:1
    x = dpp_add_masked<0x112, 0xF>(x);      // row_shr:2
    x = dpp_add_masked<0x114, 0xF>(x);      // row_shr:4
    x = dpp_add_masked<0x118, 0xF>(x);      // row_shr:8
    x = dpp_add_masked<0x142, 0xA>(x);      // row_bcast:15 -> rows 1, 3
    x = dpp_add_masked<0x143, 0xC>(x);      // row_bcast:31 -> rows 2, 3
    return x;
}

__device__ __forceinline__ float wave_allmax(float x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        x = fmaxf(x, __shfl_xor(x, m, WAVE));
    }
    return x;
}

__device__ __forceinline__ float wave_allsum(float x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        x += __shfl_xor(x, m, WAVE);
    }
    return x;
}

// ---------------------------------------------------------------------------
// "Row-set" = the S scores of 64 consecutive reads at one time step: one
// contiguous 64*S*4-byte segment of the (T, N, S) tensor (10 KiB for S = 40).
// A wave loads it with perfectly coalesced 16-byte pieces (lane l takes pieces
// l, l+64, ...), then transposes through a wave-private LDS buffer so that
// lane l ends up with the S scores of read n0 + l in registers.
// ---------------------------------------------------------------------------
template <int NB>
struct RowSet {
    using F = FF<NB>;
    f4 v[F::PIECES];

    // issue the coalesced global loads (nvalid = number of valid float4 pieces,
    // 64*PIECES for a full column of reads).  Loads are UNCONDITIONAL: the piece
    // index is clamped, so lanes past the end re-read the last valid piece
    // (finite filler for reads that do not exist) and hipcc emits one straight
    // run of global_load_dwordx4 with no exec-mask branches or vmcnt(0) waits.
    __device__ __forceinline__ void issue(const float *__restrict__ base, int nvalid, int lane) {
        const f4 *src = reinterpret_cast<const f4 *>(base);
#pragma unroll
        for (int q = 0; q < F::PIECES; ++q) v[q] = src[min(q * WAVE + lane, nvalid - 1)];
    }

    // streaming variants: last use of the data / write-once output
    __device__ __forceinline__ void issue_nt(const float *__restrict__ base, int nvalid, int lane) {
        const f4 *src = reinterpret_cast<const f4 *>(base);
#pragma unroll
        for (int q = 0; q < F::PIECES; ++q)
            v[q] = __builtin_nontemporal_load(src + min(q * WAVE + lane, nvalid - 1));
    }
    __device__ __forceinline__ void store_nt(float *__restrict__ base, int nvalid, int lane) const {
        f4 *dst = reinterpret_cast<f4 *>(base);
#pragma unroll
        for (int q = 0; q < F::PIECES; ++q) {
            const int idx = q * WAVE + lane;
            if (idx < nvalid) __builtin_nontemporal_store(v[q], dst + idx);
        }
    }

    // pieces -> own row (in place).  buf: wave-private LDS, 64*PIECES f4.
    __device__ __forceinline__ void to_rows(f4 *buf, int lane) {
#pragma unroll
        for (int q = 0; q < F::PIECES; ++q) buf[q * WAVE + lane] = v[q];
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < F::PIECES; ++q) v[q] = buf[lane * F::PIECES + q];
        wave_lds_fence();
    }

    // own row -> pieces (in place), the inverse shuffle for coalesced stores
    __device__ __forceinline__ void to_pieces(f4 *buf, int lane) {
#pragma unroll
        for (int q = 0; q < F::PIECES; ++q) buf[lane * F::PIECES + q] = v[q];
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < F::PIECES; ++q) v[q] = buf[q * WAVE + lane];
        wave_lds_fence();
    }

    __device__ __forceinline__ void store(float *__restrict__ base, int nvalid, int lane) const {
        f4 *dst = reinterpret_cast<f4 *>(base);
#pragma unroll
        for (int q = 0; q < F::PIECES; ++q) {
            const int idx = q * WAVE + lane;
            if (idx < nvalid) dst[idx] = v[q];
        }
    }

    __device__ __forceinline__ void set(int i, float x) { v[i >> 2][i & 3] = x; }
    __device__ __forceinline__ float get(int i) const { return v[i >> 2][i & 3]; }

    // w = exp(s - rowmax); returns rowmax.  Keeps every weight in (0, 1].
    __device__ __forceinline__ float exp_normalise() {
        float m = get(0);
#pragma unroll
        for (int i = 1; i < F::S; ++i) m = fmaxf(m, get(i));
        const float m2 = m * LOG2E;
#pragma unroll
        for (int i = 0; i < F::S; ++i) set(i, fast_exp2(fmaf(get(i), LOG2E, -m2)));
        return m;
    }
};

// One linear-space forward step of the 2*NB-state flip-flop lattice:
//   out[to]     = sum_from in[from] * w[to*NS + from]         (to < NB)
//   out[NB + b] = in[b] * w[FLOP0 + b] + in[NB + b] * w[FLOP0 + NB + b]
template <int NB>
__device__ __forceinline__ void ff_fwd_step(const float (&in)[2 * NB], const RowSet<NB> &w,
                                            float (&out)[2 * NB]) {
    using F = FF<NB>;
#pragma unroll
    for (int to = 0; to < NB; ++to) {
        float acc = in[0] * w.get(to * F::NS);
#pragma unroll
        for (int from = 1; from < F::NS; ++from) acc = fmaf(in[from], w.get(to * F::NS + from), acc);
        out[to] = acc;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        out[NB + b] = fmaf(in[b], w.get(F::FLOP0 + b), in[NB + b] * w.get(F::FLOP0 + NB + b));
    }
}

typedef float f2 __attribute__((ext_vector_type(2)));

// kernel A's "add" terms (fused cat-mod loss: kernel B's logZ and gradient, computed first on the
// canonical columns, folded into kernel A's own writes)
template <class Args>
__device__ __forceinline__ float crf_add_cost(const Args &a, int n, float cst) {
    return a.add_cost != nullptr ? cst + a.add_scale * a.add_cost[n] : cst;
}
template <class Args>
__device__ __forceinline__ float crf_add_grad(const Args &a, size_t t, int n, int lane, float g, float gsc) {
    if (a.add_grad != nullptr && lane < a.add_S)
        g = fmaf(a.add_grad[(t * (size_t)a.N + (size_t)n) * a.add_S + lane], a.add_scale * gsc, g);
    return g;
}

// One linear-space backward step:
//   out[from] = sum_{to<NB} w[to*NS + from] * in[to] + w[FLOP0 + from] * in[flop(from)]
// with flop(from) = NB + (from mod NB).
template <int NB>
__device__ __forceinline__ void ff_bwd_step(const float (&in)[2 * NB], const RowSet<NB> &w,
                                            float (&out)[2 * NB]) {
    using F = FF<NB>;
#pragma unroll
    for (int from = 0; from < F::NS; ++from) {
        const int fl = NB + (from % NB);
        float acc = w.get(F::FLOP0 + from) * in[fl];
#pragma unroll
        for (int to = 0; to < NB; ++to) acc = fmaf(w.get(to * F::NS + from), in[to], acc);
        out[from] = acc;
    }
}

// Exact power-of-two renormalisation of a short vector: x *= 2^-e with
// e = exponent(max x); returns e (0 if the vector is all zero).
template <int K>
__device__ __forceinline__ int pow2_normalise(float (&x)[K]) {
    float m = x[0];
#pragma unroll
    for (int i = 1; i < K; ++i) m = fmaxf(m, x[i]);
    const int e = (m > 0.f) ? __builtin_amdgcn_frexp_expf(m) : 0;
#pragma unroll
    for (int i = 0; i < K; ++i) x[i] = __builtin_amdgcn_ldexpf(x[i], -e);
    return e;
}

}  // namespace tk
