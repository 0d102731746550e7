// beam_kernels.hip -- hash beam search over the flip-flop lattice for gfx950.
//
// Replaces taiyaki/decodeutil/c_hashdecode.c:346-507 (`flipflop_beamsearch`), its guiding
// backward pass c_flipflopfwdbwd.c:55-91 and the wrapper decodeutil.pyx:9-51.
//
// One wavefront per read; the beam (<= 12 elements for the 4-base alphabet; the reference's
// default is 5) lives in lanes 0..W-1 (hash, score, last state), a block's candidate records
// -- W x nbase extensions, then W stays -- one per lane.  What the reference does with two
// quicksorts per block is done without sorting:
//   * merge by hash: a record has at most ONE partner with the same hash (the stay of prefix
//     s and the extension of s[:-1] by its last base); every lane compares its hash with the
//     other candidates' (v_readlane sweep), the later one folds into the earlier one with the
//     reference's logsumexpf;
//   * top-k by score: rank = number of records that beat this one; rank r < width becomes beam
//     element r.  The order of records with EXACTLY equal scores is, in the reference, whatever
//     its quicksort (qsort.h) leaves, and it decides who stays in the beam; such ties turn up in
//     about one block in a hundred.  A block that has one is redone the reference's way: the
//     records are laid out in hash order (ranked in parallel) and lane 0 runs the reference's
//     sort procedure on them (`beam_qsort`), so the beam follows the reference through ties too.
// Sequences are never copied: every block stores one byte per beam element (parent slot,
// appended state) and the best element's sequence is walked back at the end (through LDS when
// the table fits).  The running beam cut (`beam_cut` > 0) is the reference's: a record is
// dropped if it is worse than the best seen BEFORE it by more than log(beam_cut) -- an exclusive
// prefix maximum in candidate order.
// expf / log1pf of the reference's logsumexpf are evaluated in double and rounded to float
// (correctly rounded results, what glibc's float functions return in all but rare cases), so
// scores agree with the reference to the last bit almost everywhere and the beam takes the
// same decisions.
#include "ff_common.h"

namespace tk {

constexpr int BEAM_MAXW = 12;

__device__ __forceinline__ unsigned long long beam_mix(unsigned long long h) {
    h ^= h >> 23;
    h *= 0x2127599bf4325c37ULL;
    h ^= h >> 47;
    return h;
}
// fasthash.c:95-103
__device__ __forceinline__ unsigned long long beam_chain(unsigned long long h, unsigned long long v) {
    h ^= beam_mix(v);
    h *= 0x880355f21e6d1965ULL;
    return beam_mix(h);
}
// The reference's logsumexpf (c_hashdecode.c:50-54) is  max + log1pf(expf(-|x - y|)):  two float
// functions, each rounded to float.  Both are evaluated in double to better than 2^-52 of the result and
// rounded once -- on every third float in [0, 17) (366 M arguments) the pair gives bit for bit what
// glibc's double exp / log1p give -- by TABLE-driven reductions, because what a block of the search
// costs is the length of the DEPENDENT double-precision chain (a half-rate fma every ~40 cycles for a
// lone wave): ocml's generic exp + log1p were ~165 instructions per log-sum-exp, a table-free
// Taylor / atanh form ~50 dependent operations, this one ~24.
//   expf(-a):  n = rint(-a 32 / ln 2), r = -a - n ln2/32 (two-part constant), |r| <= 0.0109:
//              2^(n >> 5) * EXP2[n & 31] * (1 + r + ... + r^6 / 720)
//   log1pf(e): t = 1 + e (exact in double), i = rint(64 (t - 1)), c = 1 + i / 64, r = t / c - 1
//              (|r| <= 1 / 128):  LOGC[i] + r - r^2 / 2 + ... + r^7 / 7
// (degree 5 for the exponential mis-rounds 2 of those 366 M arguments; these degrees none)
constexpr int TAB_EXP2 = 0, TAB_INVC = 32, TAB_LOGC = 97, TAB_N = 162;
__constant__ double BEAM_TAB[TAB_N] = {
    0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0,
    0x1.172b83c7d517bp+0, 0x1.1d4873168b9aap+0, 0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0,
    0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0, 0x1.3dea64c123422p+0, 0x1.44e086061892dp+0,
    0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0, 0x1.6247eb03a5585p+0,
    0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0, 0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0,
    0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0,
    0x1.ae89f995ad3adp+0, 0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0,
    0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0, 0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0,
    0x1.0000000000000p+0, 0x1.f81f81f81f820p-1, 0x1.f07c1f07c1f08p-1, 0x1.e9131abf0b767p-1,
    0x1.e1e1e1e1e1e1ep-1, 0x1.dae6076b981dbp-1, 0x1.d41d41d41d41dp-1, 0x1.cd85689039b0bp-1,
    0x1.c71c71c71c71cp-1, 0x1.c0e070381c0e0p-1, 0x1.bacf914c1bad0p-1, 0x1.b4e81b4e81b4fp-1,
    0x1.af286bca1af28p-1, 0x1.a98ef606a63bep-1, 0x1.a41a41a41a41ap-1, 0x1.9ec8e951033d9p-1,
    0x1.999999999999ap-1, 0x1.948b0fcd6e9e0p-1, 0x1.8f9c18f9c18fap-1, 0x1.8acb90f6bf3aap-1,
    0x1.8618618618618p-1, 0x1.8181818181818p-1, 0x1.7d05f417d05f4p-1, 0x1.78a4c8178a4c8p-1,
    0x1.745d1745d1746p-1, 0x1.702e05c0b8170p-1, 0x1.6c16c16c16c17p-1, 0x1.6816816816817p-1,
    0x1.642c8590b2164p-1, 0x1.6058160581606p-1, 0x1.5c9882b931057p-1, 0x1.58ed2308158edp-1,
    0x1.5555555555555p-1, 0x1.51d07eae2f815p-1, 0x1.4e5e0a72f0539p-1, 0x1.4afd6a052bf5bp-1,
    0x1.47ae147ae147bp-1, 0x1.446f86562d9fbp-1, 0x1.4141414141414p-1, 0x1.3e22cbce4a902p-1,
    0x1.3b13b13b13b14p-1, 0x1.3813813813814p-1, 0x1.3521cfb2b78c1p-1, 0x1.323e34a2b10bfp-1,
    0x1.2f684bda12f68p-1, 0x1.2c9fb4d812ca0p-1, 0x1.29e4129e4129ep-1, 0x1.27350b8812735p-1,
    0x1.2492492492492p-1, 0x1.21fb78121fb78p-1, 0x1.1f7047dc11f70p-1, 0x1.1cf06ada2811dp-1,
    0x1.1a7b9611a7b96p-1, 0x1.1811811811812p-1, 0x1.15b1e5f75270dp-1, 0x1.135c81135c811p-1,
    0x1.1111111111111p-1, 0x1.0ecf56be69c90p-1, 0x1.0c9714fbcda3bp-1, 0x1.0a6810a6810a7p-1,
    0x1.0842108421084p-1, 0x1.0624dd2f1a9fcp-1, 0x1.0410410410410p-1, 0x1.0204081020408p-1,
    0x1.0000000000000p-1, 0x0.0p+0, 0x1.fc0a8b0fc03e4p-7, 0x1.f829b0e783300p-6,
    0x1.77458f632dcfcp-5, 0x1.f0a30c01162a6p-5, 0x1.341d7961bd1d1p-4, 0x1.6f0d28ae56b4cp-4,
    0x1.a926d3a4ad563p-4, 0x1.e27076e2af2e6p-4, 0x1.0d77e7cd08e59p-3, 0x1.29552f81ff523p-3,
    0x1.44d2b6ccb7d1ep-3, 0x1.5ff3070a793d4p-3, 0x1.7ab890210d909p-3, 0x1.9525a9cf456b4p-3,
    0x1.af3c94e80bff3p-3, 0x1.c8ff7c79a9a22p-3, 0x1.e27076e2af2e6p-3, 0x1.fb9186d5e3e2bp-3,
    0x1.0a324e27390e3p-2, 0x1.1675cababa60ep-2, 0x1.22941fbcf7966p-2, 0x1.2e8e2bae11d31p-2,
    0x1.3a64c556945eap-2, 0x1.4618bc21c5ec2p-2, 0x1.51aad872df82dp-2, 0x1.5d1bdbf5809cap-2,
    0x1.686c81e9b14afp-2, 0x1.739d7f6bbd007p-2, 0x1.7eaf83b82afc3p-2, 0x1.89a3386c1425bp-2,
    0x1.947941c2116fbp-2, 0x1.9f323ecbf984cp-2, 0x1.a9cec9a9a084ap-2, 0x1.b44f77bcc8f63p-2,
    0x1.beb4d9da71b7cp-2, 0x1.c8ff7c79a9a22p-2, 0x1.d32fe7e00ebd5p-2, 0x1.dd46a04c1c4a1p-2,
    0x1.e744261d68788p-2, 0x1.f128f5faf06edp-2, 0x1.faf588f78f31fp-2, 0x1.02552a5a5d0ffp-1,
    0x1.0723e5c1cdf40p-1, 0x1.0be72e4252a83p-1, 0x1.109f39e2d4c97p-1, 0x1.154c3d2f4d5eap-1,
    0x1.19ee6b467c96fp-1, 0x1.1e85f5e7040d0p-1, 0x1.23130d7bebf43p-1, 0x1.2795e1289b11bp-1,
    0x1.2c0e9ed448e8cp-1, 0x1.307d7334f10bep-1, 0x1.34e289d9ce1d3p-1, 0x1.393e0d3562a1ap-1,
    0x1.3d9026a7156fbp-1, 0x1.41d8fe84672aep-1, 0x1.4618bc21c5ec2p-1, 0x1.4a4f85db03ebbp-1,
    0x1.4e7d811b75bb1p-1, 0x1.52a2d265bc5abp-1, 0x1.56bf9d5b3f399p-1, 0x1.5ad404c359f2dp-1,
    0x1.5ee02a9241675p-1, 0x1.62e42fefa39efp-1,
};
constexpr double LN2_32_HI = 0x1.62e42fefa3000p-6, LN2_32_LO = 0x1.3de6af278ece6p-47, INV_LN2_32 = 0x1.71547652b82fep+5;

// tables into LDS (per-lane look-ups; a wave's own DS instructions execute in order)
__device__ __forceinline__ void beam_load_tables(double *tab, int lane) {
    for (int i = lane; i < TAB_N; i += WAVE) tab[i] = BEAM_TAB[i];
    wave_lds_fence();
}
__device__ __forceinline__ float beam_expf_neg(float a, const double *tab) {
    const double x = -(double)a;
    const double nf = __builtin_rint(x * INV_LN2_32);
    double r = __builtin_fma(nf, -LN2_32_HI, x);
    r = __builtin_fma(nf, -LN2_32_LO, r);
    const int n = (int)nf;
    const double scale = tab[TAB_EXP2 + (n & 31)];
    double p = 1.0 / 720.0;
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return (float)__builtin_ldexp(scale * p, n >> 5);
}
__device__ __forceinline__ float beam_log1pf_unit(float e, const double *tab) {
    const double t = 1.0 + (double)e;
    const int i = (int)__builtin_rint((t - 1.0) * 64.0);
    const double r = __builtin_fma(t, tab[TAB_INVC + i], -1.0);
    double q = 1.0 / 7.0;
    q = __builtin_fma(q, -r, 1.0 / 6.0);
    q = __builtin_fma(q, -r, 1.0 / 5.0);
    q = __builtin_fma(q, -r, 1.0 / 4.0);
    q = __builtin_fma(q, -r, 1.0 / 3.0);
    q = __builtin_fma(q, -r, 0.5);
    q = __builtin_fma(q, -r, 1.0);
    return (float)__builtin_fma(q, r, tab[TAB_LOGC + i]);
}
// c_hashdecode.c:50-54
__device__ __forceinline__ float beam_lse(float x, float y, const double *tab) {
    const float absdif = fabsf(x - y);
    // (clamped argument: both functions run unconditionally, a lane outside the range drops the result)
    const float tail = beam_log1pf_unit(beam_expf_neg(fminf(absdif, 17.0f), tab), tab);
    return fmaxf(x, y) + ((absdif < 17.0f) ? tail : 0.0f);
}
__device__ __forceinline__ float rdl(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float bpf(float v, int l) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(l * 4, __float_as_int(v)));
}
__device__ __forceinline__ int bpi(int v, int l) { return __builtin_amdgcn_ds_bpermute(l * 4, v); }

// The sort procedure of taiyaki/decodeutil/qsort.h:39-186 with LESS(i, j) = key[i] > key[j]
// (c_hashdecode.c:156-158), on n <= 64 records: median of (second, middle, last), Sedgewick
// partition, insertion sort below 16 records, the smaller subfile first -- the same comparisons
// and exchanges in the same order, so equal keys end where the reference leaves them.
// The records live ACROSS THE LANES of two registers (lane p = position p) and the whole wave runs
// the procedure with uniform control flow: an element access is a v_readlane / v_writelane (a few
// cycles, the indices are scalars), not a 100-cycle LDS round trip from one lane -- the first version
// ran this on lane 0 over LDS arrays and a block with a tie (1-3 % of them) cost ~100 us, more than
// all the other blocks together.  The subfile stack sits in the lanes of a third / fourth register.
__device__ __forceinline__ void beam_qsort_lanes(float &keyv, int &idv, int n, int lane) {
    // (v_writelane has no builtin in this hipcc: a compare + select on the lane id does the same)
    auto wrl = [&](int val, int at, int old) { return lane == at ? val : old; };
    auto K = [&](int i) { return rdl(keyv, i); };
    auto less = [&](int i, int j) { return K(i) > K(j); };
    auto swap = [&](int i, int j) {
        const int ki = __builtin_amdgcn_readlane(__float_as_int(keyv), i);
        const int kj = __builtin_amdgcn_readlane(__float_as_int(keyv), j);
        const int ti = __builtin_amdgcn_readlane(idv, i), tj = __builtin_amdgcn_readlane(idv, j);
        int kv = __float_as_int(keyv);
        kv = wrl(kj, i, kv);
        kv = wrl(ki, j, kv);
        keyv = __int_as_float(kv);
        idv = wrl(tj, i, idv);
        idv = wrl(ti, j, idv);
    };
    if (n <= 1) return;
    int lo = 0, hi = n - 1, sp = 0;
    int st_lo = 0, st_hi = 0;                       // lane s = stack slot s
    while (true) {
        if (hi - lo + 1 >= 16) {
            const int m = lo + ((hi - lo) >> 1);
            const int a1 = lo + 1, a2 = m, a3 = hi;
            if (less(a2, a1)) {
                if (less(a3, a2)) swap(a1, a3);
                else {
                    swap(a1, a2);
                    if (less(a3, a2)) swap(a2, a3);
                }
            } else if (less(a3, a2)) {
                swap(a2, a3);
                if (less(a2, a1)) swap(a1, a2);
            }
            swap(lo, m);
            int i = lo + 1, j = hi;
            while (true) {
                do ++i; while (less(i, lo));
                do --j; while (less(lo, j));
                if (i >= j) break;
                swap(i, j);
            }
            i = j + 1;
            swap(lo, j);
            --j;
            int bl, bh, sl, sh;
            if (j - lo >= hi - i) { bl = lo; bh = j; sl = i; sh = hi; }
            else { bl = i; bh = hi; sl = lo; sh = j; }
            if (sl == sh) { lo = bl; hi = bh; }
            else {
                st_lo = wrl(bl, sp, st_lo);
                st_hi = wrl(bh, sp, st_hi);
                ++sp;
                lo = sl;
                hi = sh;
            }
        } else {
            for (int q = lo + 1; q <= hi; ++q)
                for (int k = q; k > lo && less(k, k - 1); --k) swap(k, k - 1);
            if (sp == 0) break;
            --sp;
            lo = __builtin_amdgcn_readlane(st_lo, sp);
            hi = __builtin_amdgcn_readlane(st_hi, sp);
        }
    }
}

struct BeamArgs {
    const float *scores;        // (T, N, S)
    int T, N, nbase;
    int width;                  // max_beam_width
    float logcut;               // log(beam_cut); -inf = no cutting
    int guided;
    float *bwd;                 // workspace [N][T + 1][2 nbase]
    unsigned char *bp;          // workspace [N][T][16]: (parent slot << 4) | (appended state + 1)
    signed char *seq;           // out (N, T): flip-flop states, -1 padded
    int *seqlen;                // out (N)
    float *score;               // out (N)
    int lds_rows;               // rows of the back-pointer table kept in LDS (0: walk global memory)
};

// NBT: the alphabet size as a compile-time constant (4: DNA / RNA, everything the reference ships) so that
// the loops over bases unroll and their gathers leave the serial chains; 0: read it from the arguments.
template <int NBT>
__global__ __launch_bounds__(WAVE) void beam_kernel(BeamArgs a) {
    extern __shared__ unsigned char lds_bp[];           // [lds_rows][16]
    __shared__ unsigned long long nh[16];
    __shared__ float nsc[16];
    __shared__ int nlast[16], nbp[16];
    __shared__ float qkey[WAVE];
    __shared__ int qid[WAVE], qrank[WAVE];
    __shared__ double tab[TAB_N];
    const int n = blockIdx.x, lane = threadIdx.x;
    beam_load_tables(tab, lane);
    const int T = a.T, nb = NBT ? NBT : a.nbase, ns = 2 * nb, S = ns * (nb + 1);
    const size_t rowstride = (size_t)a.N * S;
    const float *sc = a.scores + (size_t)n * S;
    float *bwd = a.bwd + (size_t)n * (T + 1) * ns;
    unsigned char *bpn = a.bp + (size_t)n * T * 16;
    const int col = min(lane, S - 1);

    // ---- guiding backward pass (c_flipflopfwdbwd.c:55-91): lane = from-state ----------------
    if (lane < ns) bwd[(size_t)T * ns + lane] = 0.f;
    // (score rows are requested PF blocks ahead of their use: a block is a serial chain of nbase + 1
    // log-sum-exps, and a load issued where it is needed adds a memory round trip to every one)
    constexpr int PF = 4;
    auto load_row = [&](int blk) { return sc[(size_t)min(max(blk, 0), T - 1) * rowstride + col]; };
    if (a.guided) {
        float p = 0.f;                                  // pbwd[lane]
        float cur[PF], nxt[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) cur[q] = load_row(T - 1 - q);
        for (int b0 = T; b0 > 0; b0 -= PF) {
#pragma unroll
            for (int q = 0; q < PF; ++q) nxt[q] = load_row(b0 - 1 - PF - q);
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int blk = b0 - q;
                if (blk <= 0) break;
                const float row = cur[q];
                const int fr = min(lane, ns - 1);
                // to the flop of this state's base
                float c = bpf(row, ns * nb + fr) + bpf(p, nb + fr % nb);
                float term[NBT ? NBT : 1];
                if constexpr (NBT != 0) {
#pragma unroll
                    for (int to = 0; to < NBT; ++to) term[to] = bpf(row, to * ns + fr) + rdl(p, to);
#pragma unroll
                    for (int to = 0; to < NBT; ++to) c = beam_lse(c, term[to], tab);
                } else {
                    for (int to = 0; to < nb; ++to) c = beam_lse(c, bpf(row, to * ns + fr) + rdl(p, to), tab);
                }
                p = c;
                if (lane < ns) bwd[(size_t)(blk - 1) * ns + lane] = c;
            }
#pragma unroll
            for (int q = 0; q < PF; ++q) cur[q] = nxt[q];
        }
    } else {
        for (int i = lane; i < T * ns; i += WAVE) bwd[i] = 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();

    // ---- beam search --------------------------------------------------------------------
    // beam element i in lane i.  (The workgroup is ONE wavefront: its LDS instructions execute in order, so
    // the hand-offs through LDS below need no s_barrier -- and must not have __syncthreads(), whose
    // s_waitcnt vmcnt(0) would wait for the block's back-pointer store and the prefetched rows every time.)
    unsigned long long eh = beam_chain(0x880355f21e6d1965ULL, (unsigned long long)min(lane, nb - 1));
    float es = 0.f;
    int el = min(lane, nb - 1);
    int W = nb;
    auto load_bwd = [&](int blk) { return bwd[(size_t)min(blk + 1, T) * ns + min(lane, ns - 1)]; };
    float rowc[PF], bsvc[PF], rown[PF], bsvn[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) {
        rowc[q] = load_row(q);
        bsvc[q] = load_bwd(q);
    }
    for (int b0 = 0; b0 < T; b0 += PF) {
      // (this group's rows and backward scores were requested a group ago)
#pragma unroll
      for (int q = 0; q < PF; ++q) {
          rown[q] = load_row(b0 + PF + q);
          bsvn[q] = load_bwd(b0 + PF + q);
      }
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        const int blk = b0 + q;
        if (blk >= T) break;
        const float row = rowc[q];
        const float bsv = (lane < ns) ? bsvc[q] : 0.f;                                  // bwdscore[lane]
        const int next = W * nb, ncand = next + W;
        // candidate of this lane
        const bool is_ext = lane < next, is_cand = lane < ncand;
        const int i = is_ext ? lane / nb : min(max(lane - next, 0), W - 1);
        const int base = lane % nb;
        const unsigned pl = (unsigned)i;
        const float pscore = bpf(es, pl);
        const int plast = bpi(el, pl);
        const unsigned hlo = (unsigned)bpi((int)(unsigned)eh, pl), hhi = (unsigned)bpi((int)(unsigned)(eh >> 32), pl);
        const unsigned long long phash = ((unsigned long long)hhi << 32) | hlo;
        const int newstate = is_ext ? ((base != plast) ? base : plast + nb) : plast;
        const int tidx = plast + ns * min(newstate, nb);       // MOVE_IDX / STAY_IDX
        const float bnew = bpf(bsv, newstate);              // bwdscore[new last state]
        float cscore = (pscore + bpf(row, tidx)) + bnew;
        const unsigned long long chash = is_ext ? beam_chain(phash, (unsigned long long)newstate) : phash;
        bool valid = is_cand;
        if (a.logcut > -1e30f) {
            // lower bound from the best element (c_hashdecode.c:385-396), then the running maximum
            // over the records BEFORE this one in candidate order
            const int pb = __builtin_amdgcn_readlane(el, 0);
            float mx = rdl(row, nb * ns + pb) + rdl(bsv, pb < nb ? pb + nb : pb);
            for (int k = 0; k < nb; ++k) mx = fmaxf(mx, rdl(row, k * ns + pb) + rdl(bsv, k));
            mx += rdl(es, 0);
            float run = is_cand ? cscore : -__builtin_huge_valf();
            // inclusive prefix max over lanes, then shift by one lane
            for (int d = 1; d < WAVE; d <<= 1) {
                const float o = bpf(run, max(lane - d, 0));
                if (lane >= d) run = fmaxf(run, o);
            }
            float before = bpf(run, max(lane - 1, 0));
            if (lane == 0) before = -__builtin_huge_valf();
            valid = valid && !(cscore < fmaxf(mx, before) + a.logcut);
        }
        // ---- merge records with the same hash (= the same sequence).  The only pairs there can be
        //      are (extension of s[:-1] by its last base, stay of s): two extensions with the same
        //      sequence would have the same parent, two stays are two beam elements.  So the W stay
        //      records are broadcast one by one and the extension lanes compare -- W steps, not W (nbase + 1).
        int partner = -1;
        for (int j = next; j < ncand; ++j) {
            const unsigned jlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)chash, j);
            const unsigned jhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(chash >> 32), j);
            const bool jvalid = __builtin_amdgcn_readlane((int)valid, j) != 0;
            const bool hit = jvalid && valid && is_ext && jlo == (unsigned)chash && jhi == (unsigned)(chash >> 32);
            const unsigned long long hits = __builtin_amdgcn_ballot_w64(hit);
            if (hit) partner = j;                                   // the extension: keeps the record
            if (lane == j && hits != 0ull) partner = __builtin_ctzll(hits);     // the stay: folds into it
        }
        const float pscore2 = bpf(cscore, max(partner, 0));
        bool uniq = valid;
        if (valid && partner >= 0) {
            if (partner > lane) cscore = beam_lse(pscore2, cscore, tab);     // keep the earlier record
            else uniq = false;
        }
        // ---- rank among the unique records by score: every lane counts the records that beat its own
        //      and those that equal it (itself included).  Records that are not in the running -- merged
        //      away, cut, lanes past the candidates -- carry -inf and beat nobody, so the sweep needs no
        //      validity test and may run past the last candidate (four records per trip).
        const float rs = uniq ? cscore : -__builtin_huge_valf();
        const int nuniq = __builtin_popcountll(__builtin_amdgcn_ballot_w64(uniq));
        int rank = 0, same = 0;
        for (int j0 = 0; j0 < ncand; j0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float js = rdl(rs, j0 + u);                   // (ncand <= 60: lanes 60 .. 63 hold -inf)
                rank += (js > rs) ? 1 : 0;
                same += (js == rs) ? 1 : 0;
            }
        }
        const bool tie = uniq && same > 1;
        if (__builtin_amdgcn_ballot_w64(tie) != 0ull) {
            // equal scores: the reference's order.  Its score sort starts from the records in
            // descending hash order (c_hashdecode.c:440), a merged pair as (sum, -inf).
            int hpos = (valid && !uniq) ? 1 : 0;
            for (int j = 0; j < ncand; ++j) {
                const unsigned jlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)chash, j);
                const unsigned jhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(chash >> 32), j);
                const bool jvalid = __builtin_amdgcn_readlane((int)valid, j) != 0;
                hpos += (jvalid && (((unsigned long long)jhi << 32) | jlo) > chash) ? 1 : 0;
            }
            const int nrec = __builtin_popcountll(__builtin_amdgcn_ballot_w64(valid));
            if (valid) {
                qkey[hpos] = uniq ? cscore : -__builtin_huge_valf();
                qid[hpos] = lane;
            }
            wave_lds_fence();
            float keyv = qkey[min(lane, max(nrec - 1, 0))];
            int idv = qid[min(lane, max(nrec - 1, 0))];
            beam_qsort_lanes(keyv, idv, nrec, lane);
            if (lane < nrec) qrank[idv] = lane;
            wave_lds_fence();
            if (valid) rank = qrank[lane];
        }
        const int newW = min(a.width, nuniq);              // c_hashdecode.c:474
        if (uniq && rank < newW) {
            nh[rank] = chash;
            nsc[rank] = cscore - bnew;                      // remove the backward contribution (:485)
            nlast[rank] = newstate;
            nbp[rank] = (i << 4) | (is_ext ? newstate + 1 : 0);
        }
        wave_lds_fence();
        if (lane < newW) {
            eh = nh[lane];
            es = nsc[lane];
            el = nlast[lane];
            const unsigned char b = (unsigned char)nbp[lane];
            if (blk < a.lds_rows) lds_bp[blk * 16 + lane] = b;
            bpn[(size_t)blk * 16 + lane] = b;
        }
        W = newW;
        wave_lds_fence();
      }
#pragma unroll
      for (int q = 0; q < PF; ++q) {
          rowc[q] = rown[q];
          bsvc[q] = bsvn[q];
      }
    }
    // ---- walk the best element's sequence back -----------------------------------------------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (the back-pointer stores of every lane, once)
    __syncthreads();
    if (lane == 0) {
        a.score[n] = es;
        int slot = 0, len = 1;
        for (int blk = T - 1; blk >= 0; --blk) {
            const unsigned char b = (blk < a.lds_rows) ? lds_bp[blk * 16 + slot] : bpn[(size_t)blk * 16 + slot];
            len += (b & 15) ? 1 : 0;
            slot = b >> 4;
        }
        const int first = slot;                             // the initial one-state sequence
        signed char *out = a.seq + (size_t)n * T;
        const int keep = min(len, T);                       // (the reference's buffer holds nblock states)
        a.seqlen[n] = keep;
        slot = 0;
        int pos = len - 1;
        for (int blk = T - 1; blk >= 0; --blk) {
            const unsigned char b = (blk < a.lds_rows) ? lds_bp[blk * 16 + slot] : bpn[(size_t)blk * 16 + slot];
            if (b & 15) {
                if (pos < T) out[pos] = (signed char)((b & 15) - 1);
                --pos;
            }
            slot = b >> 4;
        }
        if (T > 0) out[0] = (signed char)first;
        for (int k = keep; k < T; ++k) out[k] = -1;
    }
}

// ---------------------------------------------------------------------------------------------
// The single-read lattice passes of the decoder on their own: flipflop_forward / flipflop_backward
// (c_flipflopfwdbwd.c:55-152) with the wrappers' optional initial vector (decodeutil.pyx:54-108).
// One wavefront per read, lane = state; every log-sum-exp in the reference's order, so the
// matrices agree with the reference's to the rounding of expf / log1pf.
//   out (N, T + 1, 2 nbase); total (N) = logsumexp over the last (forward) / first (backward) row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void lattice_kernel(const float *__restrict__ scores, int T, int N, int nb,
                                                       int forward, const float *__restrict__ init,
                                                       float *__restrict__ out, float *__restrict__ total) {
    __shared__ double tab[TAB_N];
    const int n = blockIdx.x, lane = threadIdx.x;
    beam_load_tables(tab, lane);
    const int ns = 2 * nb, S = ns * (nb + 1);
    const size_t rowstride = (size_t)N * S;
    const float *sc = scores + (size_t)n * S;
    float *mat = out + (size_t)n * (T + 1) * ns;
    const int col = min(lane, S - 1), st = min(lane, ns - 1);
    float p = (init != nullptr) ? init[(size_t)n * ns + st] : 0.f;
    if (forward) {
        if (lane < ns) mat[lane] = p;
        for (int blk = 0; blk < T; ++blk) {
            const float row = sc[(size_t)blk * rowstride + col];
            // every lane runs both recurrences (ds_bpermute returns 0 for a source lane that is
            // masked off, so the gathers must not sit in divergent code) and keeps its own
            const int b = st % nb;
            // to the flop of base b: from its flip, then from itself (:129-134)
            const float cflop = beam_lse(bpf(row, ns * nb + b) + bpf(p, b), bpf(row, ns * nb + b + nb) + bpf(p, b + nb), tab);
            // to the flip of base b: from every state, in order (:136-143)
            float cflip = bpf(row, b * ns) + rdl(p, 0);
            for (int fr = 1; fr < ns; ++fr) cflip = beam_lse(cflip, bpf(row, b * ns + fr) + rdl(p, fr), tab);
            const float c = (st >= nb) ? cflop : cflip;
            p = c;
            if (lane < ns) mat[(size_t)(blk + 1) * ns + lane] = c;
        }
    } else {
        if (lane < ns) mat[(size_t)T * ns + lane] = p;
        for (int blk = T; blk > 0; --blk) {
            const float row = sc[(size_t)(blk - 1) * rowstride + col];
            float c = bpf(row, ns * nb + st) + bpf(p, nb + st % nb);
            for (int to = 0; to < nb; ++to) c = beam_lse(c, bpf(row, to * ns + st) + rdl(p, to), tab);
            p = c;
            if (lane < ns) mat[(size_t)(blk - 1) * ns + lane] = c;
        }
    }
    float tot = rdl(p, 0);
    for (int i = 1; i < ns; ++i) tot = beam_lse(tot, rdl(p, i), tab);
    if (lane == 0) total[n] = tot;
}

int lattice_dispatch(const float *scores, size_t T, size_t N, size_t nbase, int forward, const float *init,
                     float *out, float *total, hipStream_t stream) {
    if (nbase < 1 || nbase > 4) return 2;
    if (N == 0) return 0;
    hipLaunchKernelGGL(lattice_kernel, dim3((unsigned)N), dim3(WAVE), 0, stream, scores, (int)T, (int)N, (int)nbase,
                       forward, init, out, total);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

size_t beam_workspace_bytes(size_t T, size_t N, size_t nbase) {
    return N * (T + 1) * 2 * nbase * sizeof(float) + N * T * 16 + 512;
}

int beam_dispatch(const float *scores, size_t T, size_t N, size_t nbase, int width, float beam_cut, int guided,
                  signed char *seq, int *seqlen, float *score, void *workspace, size_t workspace_bytes,
                  hipStream_t stream) {
    if (nbase < 1 || nbase > 4 || width < 1 || width > BEAM_MAXW || (size_t)width * (nbase + 1) > WAVE) return 2;
    if (!(beam_cut >= 0.f) || beam_cut > 1.f) return 1;
    if (workspace_bytes < beam_workspace_bytes(T, N, nbase)) return 3;
    BeamArgs a;
    a.scores = scores;
    a.T = (int)T;
    a.N = (int)N;
    a.nbase = (int)nbase;
    a.width = width;
    a.logcut = beam_cut > 0.f ? logf(beam_cut) : -__builtin_huge_valf();
    a.guided = guided;
    const size_t bwd_bytes = (N * (T + 1) * 2 * nbase * sizeof(float) + 255) / 256 * 256;
    a.bwd = static_cast<float *>(workspace);
    a.bp = reinterpret_cast<unsigned char *>(static_cast<char *>(workspace) + bwd_bytes);
    a.seq = seq;
    a.seqlen = seqlen;
    a.score = score;
    a.lds_rows = (int)(T <= 3584 ? T : 3584);              // 56 KiB of dynamic LDS at most
    if (nbase == 4)
        hipLaunchKernelGGL(beam_kernel<4>, dim3((unsigned)N), dim3(WAVE), (size_t)a.lds_rows * 16, stream, a);
    else
        hipLaunchKernelGGL(beam_kernel<0>, dim3((unsigned)N), dim3(WAVE), (size_t)a.lds_rows * 16, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

}  // namespace tk
