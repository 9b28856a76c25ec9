// zring.hip — exact integer polynomial products on top of the NTT engine, for the
// reference's two schoolbook callers either side of the NTT path:
//
//   N1  BFV ciphertext multiply      bfv/src/lib.rs:59-90 (tensor), :251-271 (relinearize_204)
//         arith::ring_n::naive_mul     ring_n.rs:307-320  linear convolution over Z, `as i64` wrap
//         arith::ring_n::mul_div_round ring_n.rs:130-138  f64 scale, round, Zq::from_f64, fold
//   N2  TFHE                         arith/src/ring_torus.rs:266-298 (Tn x Tn mod 2^64, X^n+1)
//         TGGSW x TGLWE                tfhe/src/tggsw.rs:45-62,139-149, tglwe.rs:182-194
//
// The reference does these with O(n^2) loops in i128 / wrapping u128.  Here a product over Z
// is computed EXACTLY as K <= 3 products modulo NTT-friendly 61-bit primes (the engine's
// kernels, with one plan per prime), recombined by Garner's CRT and only then reduced
// mod 2^64 — which is what the reference's truncation keeps.  A linear convolution of two
// length-n inputs is the negacyclic product of their zero-padded length-2n images (degree
// < 2n: nothing wraps); the torus product is negacyclic at length n with a centred lift.
// K is chosen from a caller-supplied bound on the true coefficient magnitude.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "capi_internal.hpp"
#include "digit_mac.hpp"
#include "digit32.hpp"
#include "bfv32.hpp"
#include "ntt_rounds.hpp"
#include "zq_device.hpp"
#include "mac_kernel.hpp"

using fhe::Mod;
using fhe::Tw;
using fhe::u32;
using fhe::u64;
typedef unsigned __int128 u128;

// ---------------------------------------------------------------------------
// CRT primes: p = 1 (mod 2^21), p < 2^61 (the engine's wide lazy range), pairwise coprime.
// ---------------------------------------------------------------------------
static const u64 kCrtPrimes[3] = {
    2305843009211596801ull,  // 2^61 - 2^21 + 1  (the engine's headline modulus)
    2305843009196916737ull,  // 0x1fffffffff000001
    2305843009146585089ull,  // 0x1ffffffffc000001
};

namespace fhe {

struct CrtConsts {
    Mod m[3];
    Tw inv1_mod2;    // P1^-1            mod P2
    Tw inv12_mod3;   // (P1*P2)^-1       mod P3
    Tw p1_mod3;      // P1               mod P3
    u64 p1;          // P1
    u64 p12_lo;      // P1*P2            mod 2^64
    u64 prod_lo[3];  // P1, P1*P2, P1*P2*P3  mod 2^64  (what a negative lift subtracts)
    u64 half1;       // ceil(P1 / 2)
    u64 half2;       // (P2 - 1) / 2
    u64 half3;       // (P3 - 1) / 2
};

__global__ __launch_bounds__(256) void zr_reduce_pad_kernel(const u64 *__restrict__ in,
                                                            u64 *__restrict__ out, u64 rows, u32 n,
                                                            u32 n2, Mod m) {
    const u64 total = rows * n2, stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const u64 r = i / n2;
        const u32 j = (u32)(i - r * n2);
        out[i] = j < n ? reduce_any(in[r * n + j], m) : 0ull;
    }
}

// Tn::decompose(beta = 2, l), ring_torus.rs:67-77 / torus.rs:43-52: digit d of coefficient x is
// bit l-1-d.  out[(row*l + d)][j]
__global__ __launch_bounds__(256) void zr_digits_kernel(const u64 *__restrict__ in,
                                                        u64 *__restrict__ out, u64 rows, u32 n, u32 l) {
    const u64 total = rows * l * n, stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const u64 rd = i / n;
        const u32 j = (u32)(i - rd * n);
        const u64 r = rd / l;
        const u32 d = (u32)(rd - r * l);
        out[i] = (in[r * n + j] >> (l - 1 - d)) & 1ull;
    }
}

// BFV tensor in the NTT domain (bfv/src/lib.rs:71-77): c0 = a0*b0, c1 = a0*b1 + a1*b0, c2 = a1*b1.
// ab = [a0 | a1 | b0 | b1], each `count` words; c = [c0 | c1 | c2].
__global__ __launch_bounds__(256) void zr_tensor_kernel(const u64 *__restrict__ ab, u64 *__restrict__ c,
                                                        u64 count, Mod m) {
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        const u64 a0 = ab[i], a1 = ab[count + i], b0 = ab[2 * count + i], b1 = ab[3 * count + i];
        c[i] = mul_mod_var(a0, b0, m);
        const u64 s = mul_mod_var(a0, b1, m) + mul_mod_var(a1, b0, m);
        c[count + i] = canon2(s, m);
        c[2 * count + i] = mul_mod_var(a1, b1, m);
    }
}

// c[r][b][j] = x[b][j] * y[r][j]   (key polynomials y[r] against the whole batch; x is read once for all rows)
__global__ __launch_bounds__(256) void zr_mul_bcast_kernel(const u64 *__restrict__ x,
                                                           const u64 *__restrict__ y, u64 *__restrict__ c,
                                                           u64 batch, u32 n2, u32 rows, Mod m) {
    const u64 per = batch * n2, stride = (u64)gridDim.x * 256;
    for (u64 bj = (u64)blockIdx.x * 256 + threadIdx.x; bj < per; bj += stride) {
        const u32 j = (u32)(bj % n2);
        const u64 xv = x[bj];
        for (u32 r = 0; r < rows; r++) c[(u64)r * per + bj] = mul_mod_var(xv, y[(u64)r * n2 + j], m);
    }
}

// External product, one-prime form: the key words are split into 32-bit halves laid out as
// [t][half][c][n] (t = TGLev and level, c = component), so that one multiply-accumulate over the
// digit transforms yields both half-sums; each is below (k+1)*l*n*2^32 in magnitude and is lifted
// exactly from its residue mod P1, then  S = S_lo + (S_hi << 32)  mod 2^64.
__global__ __launch_bounds__(256) void zr_split32_kernel(const u64 *__restrict__ g, u64 *__restrict__ out, u64 T,
                                                         u32 k1, u32 n) {
    const u64 per = (u64)k1 * n, total = T * per, stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const u64 t = i / per, r = i - t * per;
        const u64 v = g[i];
        out[(t * 2) * per + r] = v & 0xffffffffull;
        out[(t * 2 + 1) * per + r] = v >> 32;
    }
}
__global__ __launch_bounds__(256) void zr_combine32_kernel(const u64 *__restrict__ r, u64 *__restrict__ out, u64 batch,
                                                           u32 k1, u32 n, u64 p1, u64 half1) {
    const u64 per = (u64)k1 * n, total = batch * per, stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const u64 b = i / per, cj = i - b * per;
        u64 lo = r[b * 2 * per + cj], hi = r[b * 2 * per + per + cj];
        if (lo >= half1) lo -= p1;          // centred lift, two's complement
        if (hi >= half1) hi -= p1;
        out[i] = lo + (hi << 32);
    }
}

// Garner: residues r1 (mod P1), r2 (mod P2), r3 (mod P3) of an integer V with |V| < P/2
// (SIGNED) or 0 <= V < P  ->  V mod 2^64.
template <int K, bool SIGNED>
__device__ __forceinline__ u64 crt_value(u64 x1, u64 x2, u64 x3, const CrtConsts &cc) {
    u64 v = x1;
    bool neg = false;
    if (K == 1) {
        neg = x1 >= cc.half1;
    } else {
        const Mod &m2 = cc.m[1];
        // d2 = (r2 - r1) * P1^-1 mod P2      (r1 < P1 may exceed P2: reduce first)
        const u64 x1m2 = canon2(x1, m2);                       // P1 < 2*P2
        const u64 diff = canon2(x2 + m2.q - x1m2, m2);
        const u64 d2 = canon2(mul_shoup_lazy(diff, cc.inv1_mod2.w, cc.inv1_mod2.wp, m2), m2);
        v = x1 + cc.p1 * d2;                                    // mod 2^64
        if (K == 2) {
            neg = d2 > cc.half2 || (d2 == cc.half2 && x1 >= cc.half1);
        } else {
            const Mod &m3 = cc.m[2];
            // d3 = (r3 - r1 - P1*d2) * (P1*P2)^-1 mod P3
            const u64 x1m3 = canon2(x1, m3);
            const u64 d2m3 = canon2(d2, m3);
            const u64 t = canon2(mul_shoup_lazy(d2m3, cc.p1_mod3.w, cc.p1_mod3.wp, m3), m3);
            u64 diff3 = canon2(x3 + m3.q - x1m3, m3);
            diff3 = canon2(diff3 + m3.q - t, m3);
            const u64 d3 = canon2(mul_shoup_lazy(diff3, cc.inv12_mod3.w, cc.inv12_mod3.wp, m3), m3);
            v += cc.p12_lo * d3;
            neg = d3 > cc.half3 ||
                  (d3 == cc.half3 && (d2 > cc.half2 || (d2 == cc.half2 && x1 >= cc.half1)));
        }
    }
    if (SIGNED && neg) v -= cc.prod_lo[K - 1];
    return v;
}

template <int K, bool SIGNED>
__global__ __launch_bounds__(256) void zr_crt_kernel(const u64 *__restrict__ r1,
                                                     const u64 *__restrict__ r2,
                                                     const u64 *__restrict__ r3, u64 *__restrict__ out,
                                                     u64 count, CrtConsts cc) {
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += stride)
        out[i] = crt_value<K, SIGNED>(r1[i], K > 1 ? r2[i] : 0ull, K > 2 ? r3[i] : 0ull, cc);
}

// mul_div_round (ring_n.rs:130-138) + Rq::from_vec_f64 (ring_nq.rs:160-163) + the X^n+1 fold
// (ring_nq.rs:132-141): v holds the 2n-1 convolution words (slot 2n-1 = 0) as i64;
// out[j] = z[j] - z[j+n] in Z_q with z[i] = from_f64(round((num as f64 * v[i] as f64) / den as f64)).
__global__ __launch_bounds__(256) void zr_mul_div_round_kernel(const u64 *__restrict__ v,
                                                               u64 *__restrict__ out, u64 rows, u32 n,
                                                               u64 q, u64 num, u64 den) {
    const u64 total = rows * n, stride = (u64)gridDim.x * 256;
    const double numf = (double)num, denf = (double)den;
    const u64 qmu = ~0ull / q;                                  // Zq::from_f64's remainder by multiplication (zq_device.hpp)
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const u64 r = i / n;
        const u32 j = (u32)(i - r * n);
        const long long lo = (long long)v[r * 2 * n + j], hi = (long long)v[r * 2 * n + n + j];
        const u64 zl = zq_from_f64_mu(q, qmu, round((numf * (double)lo) / denf));
        const u64 zh = (j == n - 1) ? 0ull : zq_from_f64_mu(q, qmu, round((numf * (double)hi) / denf));
        out[i] = zl >= zh ? zl - zh : (q + zl) - zh;   // Zq::sub, zq.rs:259-276
    }
}

// The same with the Garner recombination in front and an optional Zq addend behind
// (c0 + &r0, bfv/src/lib.rs:269): residue arrays of `rows` x 2n words in, `rows` x n words out,
// so the recombined integers are never written.  Unsigned lift (operands are non-negative).
template <int K>
__global__ __launch_bounds__(256) void zr_crt_mdr_kernel(const u64 *__restrict__ r1, const u64 *__restrict__ r2,
                                                         const u64 *__restrict__ r3, const u64 *__restrict__ addend,
                                                         u64 *__restrict__ out, u64 rows, u32 n, u64 q, u64 num,
                                                         u64 den, CrtConsts cc) {
    const u64 total = rows * n, stride = (u64)gridDim.x * 256;
    const double numf = (double)num, denf = (double)den;
    const u64 qmu = ~0ull / q;                                  // Zq::from_f64's remainder by multiplication (zq_device.hpp)
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const u64 r = i / n;
        const u32 j = (u32)(i - r * n);
        const u64 il = r * 2 * n + j, ih = il + n;
        const long long lo = (long long)crt_value<K, false>(r1[il], K > 1 ? r2[il] : 0ull, K > 2 ? r3[il] : 0ull, cc);
        const u64 zl = zq_from_f64_mu(q, qmu, round((numf * (double)lo) / denf));
        u64 zh = 0;
        if (j != n - 1) {
            const long long hi = (long long)crt_value<K, false>(r1[ih], K > 1 ? r2[ih] : 0ull, K > 2 ? r3[ih] : 0ull, cc);
            zh = zq_from_f64_mu(q, qmu, round((numf * (double)hi) / denf));
        }
        u64 v = zl >= zh ? zl - zh : (q + zl) - zh;   // Zq::sub, zq.rs:259-276
        if (addend) {
            v += addend[i];
            if (v >= q) v -= q;                        // Zq::add, zq.rs:219-231
        }
        out[i] = v;
    }
}

// Relinearisation with ONE prime: the key words (< pq < 2^63) are split at bit h into halves below 2^h, so that a product
// c2 * half summed over n terms stays below the prime and both half-products are exact integers; the product modulo 2^64
// — all the reference's `as i64` keeps — is lo + (hi << h).  Key rows laid out [rlk0_lo | rlk0_hi | rlk1_lo | rlk1_hi].
__global__ __launch_bounds__(256) void zr_split_h_kernel(const u64 *__restrict__ key, u64 *__restrict__ out, u64 rows, u32 n, u32 h) {
    const u64 total = rows * n, stride = (u64)gridDim.x * 256;
    const u64 mask = (1ull << h) - 1ull;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const u64 r = i / n, j = i - r * n, v = key[i];
        out[(2 * r) * n + j] = v & mask;
        out[(2 * r + 1) * n + j] = v >> h;
    }
}
// out[o][b][j] = addend[o][b][j] + fold(from_f64(round(num * V / den))),  V = (lo + (hi << h)) as i64 taken from the
// half-product rows R[2o][b] (lo) and R[2o+1][b] (hi), each 2n words: mul_div_round + from_vec_f64 + fold + Zq::add as
// zr_crt_mdr_kernel does them, with the recombination of the halves in place of Garner's.
__global__ __launch_bounds__(256) void zr_split_mdr_kernel(const u64 *__restrict__ R, const u64 *__restrict__ addend,
                                                           u64 *__restrict__ out, u64 batch, u32 n, u32 h, u64 q, u64 num, u64 den) {
    const u64 per = batch * n, total = 2 * per, stride = (u64)gridDim.x * 256;
    const double numf = (double)num, denf = (double)den;
    const u64 qmu = ~0ull / q;                                  // Zq::from_f64's remainder by multiplication (zq_device.hpp)
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const u64 o = i / per, bj = i - o * per, b = bj / n;
        const u32 j = (u32)(bj - b * n);
        const u64 il = ((2 * o) * batch + b) * 2 * n + j, ih = ((2 * o + 1) * batch + b) * 2 * n + j;
        const long long lo = (long long)(R[il] + (R[ih] << h));
        const u64 zl = zq_from_f64_mu(q, qmu, round((numf * (double)lo) / denf));
        u64 zh = 0;
        if (j != n - 1) {
            const long long hi = (long long)(R[il + n] + (R[ih + n] << h));
            zh = zq_from_f64_mu(q, qmu, round((numf * (double)hi) / denf));
        }
        u64 v = zl >= zh ? zl - zh : (q + zl) - zh;   // Zq::sub, zq.rs:259-276
        v += addend[i];
        if (v >= q) v -= q;                            // Zq::add, zq.rs:219-231
        out[i] = v;
    }
}

// The LAST pass of a two-pass inverse transform modulo ONE prime with mul_div_round + from_vec_f64 + the X^n+1 fold
// as its epilogue (single-prime products: the residue IS the integer): a thread of the strided pass ends up holding
// rows f and f + F/2 of its column, i.e. coefficients j and j + n of the 2n-word convolution — exactly the pair the
// fold subtracts — so the scaled, rounded, folded Z_q words are formed in registers and the 2n-word integers are
// never written (BFV tensor at N = 8192: one kernel and 0.8 GB of traffic less per 2048 ciphertext pairs).
// the strided last pass of an inverse transform on one column tile: load (lazy values of the contiguous pass), rounds;
// leaves register k = row (k << A0) | tf of column c, below 2q.  FIRST: nobody has touched the LDS tile yet.
template <int LA, int CW, bool FIRST>
__device__ __forceinline__ void inv_strided_tile(u64 (&v)[16], const u64 *__restrict__ pin, u64 *lds, const Tw *ltw, u32 lb, u32 c, u32 tf,
                                                 const Mod &m, const Tw ninv, const Tw s_ninv) {
    using C = StridedCfg<LA, CW>;
    constexpr int ALAST = C::a_of(C::NR - 1);
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = ld_at<u64>(pin, ((field_of<ALAST>(tf, k) << lb) + c) * 8u);
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        round_inv_sel<4, false, true, 4>(v, ltw, (1u << LS) + (tf >> A), m, ninv, s_ninv);
        exchange_strided<CW, A, C::a_of(1), FIRST>(v, lds, c, tf);
    }
    if constexpr (C::NR > 1) {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        round_inv_sel<4, false, true, 4>(v, ltw, (1u << LS) + (tf >> A), m, ninv, s_ninv);
        exchange_strided<CW, A, C::A0, FIRST && (C::NR <= 2)>(v, lds, c, tf);
    }
    round_inv_sel<C::R0, true, true, 4>(v, ltw, 1u, m, ninv, s_ninv);
}

// output polynomial `poly` = fold(from_f64(round(num * intt(in[poly]) / den))): see above
template <int LA, int CW>
__global__ __launch_bounds__((StridedCfg<LA, CW>::TH)) void zr_inv_strided_mdr_kernel(PassArgs a, u64 *__restrict__ outq, u64 q,
                                                                                      double numf, double denf) {
    using C = StridedCfg<LA, CW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const u32 tid = threadIdx.x, c = tid % CW, tf = tid / CW;
    const u32 lb = a.log_n - LA;
    const u32 lcg = lb - __builtin_ctz(CW);
    const u32 cg = blockIdx.x & ((1u << lcg) - 1u);
    const u64 poly = (u64)(blockIdx.x >> lcg);
    const Mod &m = a.mod;
    const u64 qmu = ~0ull / q;
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + C::DATA_BYTES);
    for (u32 li = tid; li < (u32)C::F; li += C::TH) ltw[li] = a.tw[li];
    __syncthreads();
    u64 v[16];
    inv_strided_tile<LA, CW, true>(v, a.in + (poly << a.log_n) + (u64)cg * CW, lds, ltw, lb, c, tf, m, a.ninv, a.s_ninv);
    // register k holds row (k << A0) | tf: k and k + 8 are rows f and f + F/2 = coefficients j and j + n
    const u32 nq = 1u << (a.log_n - 1);                            // n: words per output polynomial
    u64 *__restrict__ po = outq + poly * nq + (u64)cg * CW + c;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const u32 f = ((u32)k << C::A0) | tf;
        const long long lo = (long long)canon2(v[k], m), hi = (long long)canon2(v[k + 8], m);
        const u64 zl = zq_from_f64_mu(q, qmu, round((numf * (double)lo) / denf));
        const u64 zh = zq_from_f64_mu(q, qmu, round((numf * (double)hi) / denf));   // slot 2n-1 of a (2n-1)-term convolution is 0
        po[(u64)f << lb] = zl >= zh ? zl - zh : (q + zl) - zh;    // Zq::sub, zq.rs:259-276
    }
}

// Zq::add over whole polynomials (c0 + &r0, bfv/src/lib.rs:269)
__global__ __launch_bounds__(256) void zr_rq_add_kernel(const u64 *__restrict__ a, const u64 *__restrict__ b,
                                                        u64 *__restrict__ c, u64 count, u64 q) {
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        u64 s = a[i] + b[i];
        if (s >= q) s -= q;
        c[i] = s;
    }
}

}  // namespace fhe

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static inline u64 hmulmod(u64 a, u64 b, u64 p) { return (u64)(((u128)a * b) % p); }
static u64 hpow(u64 a, u64 e, u64 p) {
    u64 r = 1;
    a %= p;
    while (e) { if (e & 1) r = hmulmod(r, a, p); a = hmulmod(a, a, p); e >>= 1; }
    return r;
}
static inline Tw htw(u64 w, u64 p) { return Tw{w, (u64)((((u128)w) << 64) / p)}; }

// number of primes for a bound of `bits` on |true value| (signed lift needs one more bit)
static int primes_for_bits(unsigned bits, bool is_signed) {
    const unsigned need = bits + (is_signed ? 1u : 0u);
    if (need <= 60) return 1;
    if (need <= 121) return 2;
    if (need <= 182) return 3;
    return 0;
}

struct ZCtx {
    int K = 0;
    const fhe_ntt_plan *plan[3] = {nullptr, nullptr, nullptr};
    fhe::DevicePlan dp[3];
    fhe::CrtConsts cc{};
};

static int zctx_init(ZCtx *z, u64 n2, int K) {
    if (K < 1 || K > 3) return fhe_fail(FHE_E_INVALID, "coefficient bound too large for 3 CRT primes");
    z->K = K;
    for (int k = 0; k < K; k++) {
        int rc = fhe_ntt_plan_get(kCrtPrimes[k], n2, &z->plan[k]);
        if (rc != FHE_OK) return rc;
        rc = fhe_device_plan(z->plan[k], &z->dp[k]);
        if (rc != FHE_OK) return rc;
        z->cc.m[k] = z->plan[k]->mod;
    }
    const u64 P1 = kCrtPrimes[0], P2 = kCrtPrimes[1], P3 = kCrtPrimes[2];
    z->cc.p1 = P1;
    z->cc.p12_lo = P1 * P2;
    z->cc.prod_lo[0] = P1;
    z->cc.prod_lo[1] = P1 * P2;
    z->cc.prod_lo[2] = P1 * P2 * P3;
    z->cc.half1 = (P1 + 1) / 2;
    z->cc.half2 = (P2 - 1) / 2;
    z->cc.half3 = (P3 - 1) / 2;
    z->cc.inv1_mod2 = htw(hpow(P1 % P2, P2 - 2, P2), P2);
    z->cc.p1_mod3 = htw(P1 % P3, P3);
    z->cc.inv12_mod3 = htw(hpow(hmulmod(P1 % P3, P2 % P3, P3), P3 - 2, P3), P3);
    if (K < 2) z->cc.m[1] = z->cc.m[0];
    if (K < 3) z->cc.m[2] = z->cc.m[0];
    return FHE_OK;
}


static int z_forward(const ZCtx &z, int k, const u64 *in, u64 *out, u64 rows, hipStream_t st) {
    hipError_t e = fhe::launch_ntt_forward(z.dp[k], in, out, rows, fhe_batch_tile_for(z.plan[k]), st);
    return e == hipSuccess ? FHE_OK : fhe_hip_fail(e, "zring forward NTT");
}
// out[r] = NTT_k(src[r] mod P_k, zero-padded from n_src to the plan's n): the reduction and the
// padding happen in the transform's load; sizes without such a kernel (n < 16) stage through `out`.
static int z_forward_src(const ZCtx &z, int k, const u64 *src, u64 *out, u64 rows, u64 n_src, hipStream_t st) {
    const u64 n = z.plan[k]->n;
    hipError_t e = fhe::launch_ntt_forward_reduce(z.dp[k], src, out, rows, (uint32_t)__builtin_ctzll(n_src),
                                                  fhe_batch_tile_for(z.plan[k]), st);
    if (e == hipSuccess) return FHE_OK;
    if (e != hipErrorNotSupported) return fhe_hip_fail(e, "zring reducing forward NTT");
    { fhe::KernelTimer kt_("zr_reduce_pad", 0, st);
    hipLaunchKernelGGL(fhe::zr_reduce_pad_kernel, dim3(fhe_ew_grid(rows * n)), dim3(256), 0, st, src, out, rows, (u32)n_src, (u32)n, z.cc.m[k]);
    }
    LAUNCH_OK("zr_reduce_pad_kernel");
    return z_forward(z, k, out, out, rows, st);
}
static int z_inverse(const ZCtx &z, int k, const u64 *in, u64 *out, u64 rows, hipStream_t st) {
    hipError_t e = fhe::launch_ntt_inverse(z.dp[k], in, nullptr, nullptr, out, rows,
                                           fhe_batch_tile_for(z.plan[k]), st);
    return e == hipSuccess ? FHE_OK : fhe_hip_fail(e, "zring inverse NTT");
}
static int z_crt(const ZCtx &z, bool is_signed, const u64 *r1, const u64 *r2, const u64 *r3, u64 *out,
                 u64 count, hipStream_t st) {
    const unsigned g = fhe_ew_grid(count);
    fhe::KernelTimer kt_("zr_crt", z.K, st);
#define CRT_CASE(K_, S_) hipLaunchKernelGGL((fhe::zr_crt_kernel<K_, S_>), dim3(g), dim3(256), 0, st, r1, r2, r3, out, count, z.cc)
    if (z.K == 1) { if (is_signed) CRT_CASE(1, true); else CRT_CASE(1, false); }
    else if (z.K == 2) { if (is_signed) CRT_CASE(2, true); else CRT_CASE(2, false); }
    else { if (is_signed) CRT_CASE(3, true); else CRT_CASE(3, false); }
#undef CRT_CASE
    LAUNCH_OK("zr_crt_kernel");
    return FHE_OK;
}

// out[row][j] = (addend[row][j] +) fold(from_f64(round(num * crt(residues)[.] / den))): rows x 2n residues in
static int z_crt_mdr(const ZCtx &z, const u64 *r1, const u64 *r2, const u64 *r3, const u64 *addend, u64 *out,
                     u64 rows, u64 n, u64 q, u64 num, u64 den, hipStream_t st) {
    const unsigned g = fhe_ew_grid(rows * n);
    fhe::KernelTimer kt_("zr_crt_mdr", z.K, st);
#define MDR_CASE(K_) hipLaunchKernelGGL((fhe::zr_crt_mdr_kernel<K_>), dim3(g), dim3(256), 0, st, r1, r2, r3, addend, out, rows, (u32)n, q, num, den, z.cc)
    if (z.K == 1) MDR_CASE(1); else if (z.K == 2) MDR_CASE(2); else MDR_CASE(3);
#undef MDR_CASE
    LAUNCH_OK("zr_crt_mdr_kernel");
    return FHE_OK;
}

// inverse transform (two-pass sizes, one prime) whose last pass scales, rounds and folds: rows x 2n residues in
// (NTT domain), rows x n words of Z_q out.  hipErrorNotSupported: the caller takes z_inverse + z_crt_mdr.
// (For the split-key relinearisation a dual form of the kernel — both halves' last passes in one kernel that also
// recombines and adds (c0, c1) — was measured slower than inverse + zr_split_mdr_kernel, 833 vs 382 + 411 us per
// 2048 ciphertexts: the f64 division sits badly in a VALU-bound pass when it serves two transforms; not kept.)
static hipError_t z_inverse_mdr(const ZCtx &z, u64 *r, u64 *out, u64 rows, u64 q, u64 num, u64 den, hipStream_t st) {
    const fhe::DevicePlan &dp = z.dp[0];
    const int L = dp.log_n;
    if (z.K != 1 || !dp.wide || L <= fhe::kMaxSinglePassLog) return hipErrorNotSupported;
    hipError_t e = fhe::launch_ntt_inverse_first_pass(dp, r, r, rows, st);
    if (e != hipSuccess) return e;
    fhe::PassArgs a{};
    a.tw = dp.tw_inv; a.mod = dp.mod; a.ninv = dp.ninv; a.s_ninv = dp.s_ninv; a.log_n = dp.log_n; a.in = r; a.batch = rows;
    const int LA = L - fhe::contig_bits(L);
    fhe::KernelTimer kt_("zr_inv_strided_mdr", LA, st);
#define MDR_PASS(LA_, CW_)                                                                                              \
    {                                                                                                                   \
        using C = fhe::StridedCfg<LA_, CW_>;                                                                            \
        const u64 grid = ((1ull << (L - LA_)) / CW_) * rows;                                                            \
        if (grid > 0x7fffffffull) return hipErrorInvalidValue;                                                          \
        if ((e = fhe::allow_big_lds((const void *)fhe::zr_inv_strided_mdr_kernel<LA_, CW_>, C::LDS_BYTES)) != hipSuccess) return e; \
        hipLaunchKernelGGL((fhe::zr_inv_strided_mdr_kernel<LA_, CW_>), dim3((unsigned)grid), dim3(C::TH), C::LDS_BYTES, st, a, out, q, \
                           (double)num, (double)den);                                                                   \
    }
    switch (LA) {
        case 6: MDR_PASS(6, 128) break;
        case 7: MDR_PASS(7, 64) break;
        case 8: MDR_PASS(8, 32) break;
        default: return hipErrorNotSupported;
    }
#undef MDR_PASS
    return hipGetLastError();
}

static unsigned bits_of(u64 x) { unsigned b = 0; while (x) { b++; x >>= 1; } return b; }
static unsigned ceil_log2(u64 x) { return x <= 1 ? 0 : bits_of(x - 1); }

static int check_pow2_n(u64 n, const char *who) {
    if (n < 2 || (n & (n - 1)) != 0 || n > (1ull << 19))
        return fhe_fail(FHE_E_BAD_N, "%s: n=%llu must be a power of two in [2, 2^19]", who, (unsigned long long)n);
    return FHE_OK;
}

// ---- arith::ring_n::naive_mul ---------------------------------------------------------------
extern "C" int fhe_r_naive_mul_dev(uint64_t n, const void *d_a, const void *d_b, void *d_out, size_t batch,
                                   unsigned a_bits, unsigned b_bits, void *hip_stream) {
    int rc = check_pow2_n(n, "fhe_r_naive_mul_dev");
    if (rc != FHE_OK) return rc;
    if (batch == 0) return FHE_OK;
    if (!d_a || !d_b || !d_out) return fhe_fail(FHE_E_NULL, "fhe_r_naive_mul_dev: NULL buffer");
    REQUIRE_ALIGNED(d_a); REQUIRE_ALIGNED(d_b); REQUIRE_ALIGNED(d_out);
    if (a_bits == 0 || a_bits > 64) a_bits = 64;
    if (b_bits == 0 || b_bits > 64) b_bits = 64;
    const u64 n2 = 2 * n;
    ZCtx z;
    rc = zctx_init(&z, n2, primes_for_bits(a_bits + b_bits + ceil_log2(n), false));
    if (rc != FHE_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    const u64 words = batch * n2;
    void *wsv = nullptr;
    rc = fhe_workspace_get(1, (2 + (size_t)z.K) * words * 8, st, &wsv);
    if (rc != FHE_OK) return rc;
    u64 *A = (u64 *)wsv, *B = A + words, *R = B + words;   // R: K residue arrays
    for (int k = 0; k < z.K; k++) {
        if ((rc = z_forward_src(z, k, (const u64 *)d_a, A, batch, n, st)) != FHE_OK) return rc;
        if ((rc = z_forward_src(z, k, (const u64 *)d_b, B, batch, n, st)) != FHE_OK) return rc;
        hipError_t e = fhe::launch_ntt_inverse(z.dp[k], A, B, nullptr, R + (u64)k * words, batch,
                                               fhe_batch_tile_for(z.plan[k]), st);
        if (e != hipSuccess) return fhe_hip_fail(e, "zring inverse(A.*B)");
    }
    return z_crt(z, false, R, R + words, R + 2 * words, (u64 *)d_out, words, st);
}

// ---- arith::ring_n::mul_div_round -----------------------------------------------------------
extern "C" int fhe_mul_div_round_dev(uint64_t q, uint64_t n, const void *d_v, uint64_t num, uint64_t den,
                                     void *d_out, size_t batch, void *hip_stream) {
    if (n < 1) return fhe_fail(FHE_E_BAD_N, "fhe_mul_div_round_dev: n = 0");
    if (q == 0 || den == 0 || (q >> 63)) return fhe_fail(FHE_E_BAD_Q, "fhe_mul_div_round_dev: q must be in [1, 2^63), den > 0");
    if (batch == 0) return FHE_OK;
    if (!d_v || !d_out) return fhe_fail(FHE_E_NULL, "fhe_mul_div_round_dev: NULL buffer");
    int dev;
    int rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    { fhe::KernelTimer kt_("zr_mul_div_round", 0, (hipStream_t)hip_stream);
    hipLaunchKernelGGL(fhe::zr_mul_div_round_kernel, dim3(fhe_ew_grid(batch * n)), dim3(256), 0, (hipStream_t)hip_stream,
                       (const u64 *)d_v, (u64 *)d_out, (u64)batch, (u32)n, (u64)q, (u64)num, (u64)den);
    }
    LAUNCH_OK("zr_mul_div_round_kernel");
    return FHE_OK;
}

// ---- BFV on two / three 27-bit primes (bfv32.hip): small q, 1024 <= n <= 8192 -----------------------------------------
static bool bfv32_on(uint64_t q, uint64_t n, uint64_t pq) { return fhe_ext32_enabled() && fhe::bfv32_shape_supported(q, n, pq); }
static int bfv32_args(uint64_t n, fhe::Bfv32Args *a) {
    int rc = fhe_ext32_tables(2 * n, &a->t);
    if (rc != FHE_OK) return rc;
    a->log_n2 = ceil_log2(2 * n);
    for (int i = 0; i < 3; i++) {
        const uint32_t p = a->t.p[i];
        uint32_t inv = p;                                       // Newton: p^-1 mod 2^32 (p odd: 3 correct bits, doubled per step)
        for (int it = 0; it < 5; it++) inv *= 2u - p * inv;
        a->pinv_neg[i] = 0u - inv;
        const u64 w = ((u64)a->t.ninv[i].w << 32) % p;          // (2n)^-1 * 2^32 mod p
        a->ninv_mont[i] = fhe::Tw32{(uint32_t)w, (uint32_t)((w << 32) / p)};
        const fhe_ntt_plan *plan = nullptr;                     // (cached: fhe_ext32_tables built the tables from it)
        if ((rc = fhe_ntt_plan_get(p, 2 * n, &plan)) != FHE_OK) return rc;
        const u64 w1 = (plan->roots_inv[1] * w) % p;            // roots_inv[1] * (2n)^-1 * 2^32 mod p: the last stage and the scaling in one
        a->w1ninv_mont[i] = fhe::Tw32{(uint32_t)w1, (uint32_t)((w1 << 32) / p)};
    }
    return FHE_OK;
}
// Zq::from_f64 in f64 alone where every scaled coefficient stays below 2^50 (bfv32.hip: zq_from_f64_small): bit-identical
// (tests/test_round3_gpu.py) and ~90 instructions shorter per coefficient.  Round 3 measured it SLOWER inside the block
// kernels (tensor 954 vs 854 us per 2048 pairs: the shorter epilogue made kernels that were already spilling spill more)
// and left it opt-in; with the residues parked outside the registers (round 4: no scratch in either form) it is the
// faster one — tensor 867 -> 821 us, relinearisation 815 -> 796 us, VALU instructions per wave 8120 -> 6946 / 11297 -> 10133
// (gpurun_out/bfv/small*.json) — and the default.  FHE_BFV_SMALL_F64=0 selects the general conversion.
static bool bfv32_small_f64_on() {
    static const bool on = [] { const char *e = getenv("FHE_BFV_SMALL_F64"); return !(e && e[0] == '0'); }();
    return on;
}
// the epilogue's quotient as reciprocal + two fma (bfv32.hip: exact_quotient) needs an ODD integer denominator; the argument
// written there covers denominators below 2^48 (what is used: q < 2^21, p = q^2 < 2^42) — the gate is that, not 2^53
// (FHE_BFV_FAST_DIV=0: the IEEE division sequence — the A/B)
static double bfv32_rden(uint64_t den) {
    static const bool on = [] { const char *e = getenv("FHE_BFV_FAST_DIV"); return !(e && e[0] == '0'); }();
    return (on && (den & 1ull) && den < (1ull << 48)) ? 1.0 / (double)den : 0.0;
}
// q <= every prime in use: canonical source words need no reduction (FHE_BFV_BELOW_P=0: reduce anyway)
static uint32_t bfv32_below_p(uint64_t q, const fhe::Bfv32Args &a, int primes) {
    static const bool on = [] { const char *e = getenv("FHE_BFV_BELOW_P"); return !(e && e[0] == '0'); }();
    for (int i = 0; i < primes; i++) if (q > a.t.p[i]) return 0u;
    return on ? 1u : 0u;
}
static int bfv32_tensor(uint64_t q, uint64_t n, uint64_t t, const void *d_ab, void *d_c, size_t batch, hipStream_t st) {
    fhe::Bfv32Args a{};
    int rc = bfv32_args(n, &a);
    if (rc != FHE_OK) return rc;
    void *wsv = nullptr;
    // transforms of [a0 | a1 | b0 | b1] x batch modulo two primes: 2 * 4 * batch rows of 2n u32
    if ((rc = fhe_workspace_get(1, (u64)2 * 4 * batch * 2 * n * 4, st, &wsv)) != FHE_OK) return rc;
    a.src = (const u64 *)d_ab; a.fw = (uint32_t *)wsv; a.rows = 4 * (u64)batch; a.primes = 2; a.word32 = 1;   // q < 2^21
    a.below_p = bfv32_below_p(q, a, 2);
    // below_p reads the source words as their own residues: the v < q contract, verified on request
    if (a.below_p && (rc = fhe_check_canonical_words(q, d_ab, 4 * batch * n, st, "fhe_bfv_tensor_dev")) != FHE_OK) return rc;
    hipError_t e = fhe::launch_bfv32_forward(a, st);
    if (e != hipSuccess) return fhe_hip_fail(e, "bfv32_forward_kernel");
    a.batch = batch; a.out = (u64 *)d_c; a.q = q; a.qmu = ~0ull / q; a.numf = (double)t; a.denf = (double)q;
    // (quotients below 2^52 in magnitude: an integer quotient is then a double, never the midpoint of two)
    a.rdenf = ((unsigned __int128)2 * n * (q - 1) * (q - 1) * t / q < ((unsigned __int128)1 << 52)) ? bfv32_rden(q) : 0.0;
    {   // the integer epilogue (bfv32.hip: zq_scale_round_int) where t * v < 2^52 for every coefficient v <= 2 n (q - 1)^2
        // — opt-in with FHE_BFV_INT_ROUND=1: bit-identical (tests/test_round3_gpu.py) but measured SLOWER than the f64 form
        // on MI355X (951 vs 855 us per 2048 pairs: f64 runs at full rate here, two 64-bit quotients cost more than one division)
        static const bool on = [] { const char *e = getenv("FHE_BFV_INT_ROUND"); return e && e[0] == '1'; }();
        const unsigned __int128 vmax = (unsigned __int128)2 * n * (q - 1) * (q - 1);
        a.int_num = (on && t != 0 && vmax * t < ((unsigned __int128)1 << 52)) ? t : 0;
        // Zq::from_f64 in f64 alone (bfv32.hip: zq_from_f64_small) where every scaled coefficient stays below 2^50
        // (the default since round 4, see bfv32_small_f64_on)
        a.small_f64 = (bfv32_small_f64_on() && q < (1ull << 30) && vmax * t / q < ((unsigned __int128)1 << 50)) ? 1u : 0u;
        a.qinvf = 1.0 / (double)q;
    }
    e = fhe::launch_bfv32_tensor_inverse(a, st);
    return e == hipSuccess ? FHE_OK : fhe_hip_fail(e, "bfv32_tensor_inverse_kernel");
}
// key: [rlk0 | rlk1] x n words -> prep[prime (3)][2][2n] u32 = 6n 64-bit words
static int bfv32_rlk_prepare(uint64_t n, const u64 *d_rlk, void *prep, hipStream_t st) {
    fhe::Bfv32Args a{};
    int rc = bfv32_args(n, &a);
    if (rc != FHE_OK) return rc;
    a.src = d_rlk; a.fw = (uint32_t *)prep; a.rows = 2; a.primes = 3;
    hipError_t e = fhe::launch_bfv32_forward(a, st);
    return e == hipSuccess ? FHE_OK : fhe_hip_fail(e, "bfv32_forward_kernel");
}
static int bfv32_relinearize(uint64_t q, uint64_t n, uint64_t pq, const void *d_prep, const void *d_c, void *d_out, size_t batch,
                             hipStream_t st) {
    fhe::Bfv32Args a{};
    int rc = bfv32_args(n, &a);
    if (rc != FHE_OK) return rc;
    void *wsv = nullptr;
    // transforms of c2 modulo three primes (3 * batch rows of 2n u32), then the two planes where Garner's digits are parked
    const u64 x_bytes = (u64)3 * batch * 2 * n * 4, park_bytes = (u64)2 * 2 * batch * n * 8;
    if ((rc = fhe_workspace_get(1, x_bytes + park_bytes, st, &wsv)) != FHE_OK) return rc;
    a.park = (u64 *)((unsigned char *)wsv + x_bytes);
    a.src = (const u64 *)d_c + 2 * (u64)batch * n; a.fw = (uint32_t *)wsv; a.rows = batch; a.primes = 3; a.word32 = 1;
    a.below_p = bfv32_below_p(q, a, 3);
    if (a.below_p && (rc = fhe_check_canonical_words(q, a.src, batch * n, st, "fhe_bfv_relinearize_dev (c2)")) != FHE_OK) return rc;
    hipError_t e = fhe::launch_bfv32_forward(a, st);
    if (e != hipSuccess) return fhe_hip_fail(e, "bfv32_forward_kernel");
    a.x = (const uint32_t *)wsv; a.key = (const uint32_t *)d_prep;
    a.addend = (const u64 *)d_c; a.out = (u64 *)d_out; a.batch = batch; a.q = q; a.qmu = ~0ull / q; a.numf = 1.0; a.denf = (double)(pq / q);
    a.rdenf = (pq % q == 0 && pq / q >= (1ull << 12)) ? bfv32_rden(pq / q) : 0.0;      // |R| <= 2^63: quotients below 2^51
    // |R| <= 2^63 (an i64): R / p stays below 2^50 for p >= 2^14
    a.small_f64 = (bfv32_small_f64_on() && q < (1ull << 30) && pq / q >= (1ull << 14)) ? 1u : 0u;
    a.qinvf = 1.0 / (double)q;
    e = fhe::launch_bfv32_relin_inverse(a, st);
    return e == hipSuccess ? FHE_OK : fhe_hip_fail(e, "bfv32_relin_inverse_kernel");
}

// ---- BFV: RLWE::tensor / relinearize_204 / RLWE::mul ------------------------------------------
// d_ab: [a0 | a1 | b0 | b1], each batch x n (mod q).  d_c: [c0 | c1 | c2], each batch x n.
extern "C" int fhe_bfv_tensor_dev(uint64_t q, uint64_t n, uint64_t t, const void *d_ab, void *d_c, size_t batch,
                                  void *hip_stream) {
    int rc = check_pow2_n(n, "fhe_bfv_tensor_dev");
    if (rc != FHE_OK) return rc;
    if (q < 2 || (q >> 63)) return fhe_fail(FHE_E_BAD_Q, "fhe_bfv_tensor_dev: q must be in [2, 2^63)");
    if (batch == 0) return FHE_OK;
    if (!d_ab || !d_c) return fhe_fail(FHE_E_NULL, "fhe_bfv_tensor_dev: NULL buffer");
    REQUIRE_ALIGNED(d_ab); REQUIRE_ALIGNED(d_c);
    if (bfv32_on(q, n, 0)) return bfv32_tensor(q, n, t, d_ab, d_c, batch, (hipStream_t)hip_stream);
    const u64 n2 = 2 * n;
    ZCtx z;
    // c1 = a0*b1 + a1*b0 < 2 * n * q^2
    rc = zctx_init(&z, n2, primes_for_bits(2 * bits_of(q - 1) + ceil_log2(n) + 1, false));
    if (rc != FHE_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    const u64 words = batch * n2;   // per polynomial set
    void *wsv = nullptr;
    rc = fhe_workspace_get(1, (4 + 3 * (size_t)z.K) * words * 8, st, &wsv);
    if (rc != FHE_OK) return rc;
    u64 *AB = (u64 *)wsv, *R = AB + 4 * words;
    for (int k = 0; k < z.K; k++) {
        if ((rc = z_forward_src(z, k, (const u64 *)d_ab, AB, 4 * batch, n, st)) != FHE_OK) return rc;
        u64 *Rk = R + 3 * (u64)k * words;
        { fhe::KernelTimer kt_("zr_tensor", 0, st);
        hipLaunchKernelGGL(fhe::zr_tensor_kernel, dim3(fhe_ew_grid(words)), dim3(256), 0, st, (const u64 *)AB, Rk, words, z.cc.m[k]);
        }
        LAUNCH_OK("zr_tensor_kernel");
        if (z.K == 1) {   // one prime: scale by t/q, round, reduce and fold in the inverse's last pass
            hipError_t e = z_inverse_mdr(z, Rk, (u64 *)d_c, 3 * batch, q, t, q, st);
            if (e == hipSuccess) return FHE_OK;
            if (e != hipErrorNotSupported) return fhe_hip_fail(e, "zr_inv_strided_mdr_kernel");
            (void)hipGetLastError();
        }
        if ((rc = z_inverse(z, k, Rk, Rk, 3 * batch, st)) != FHE_OK) return rc;
    }
    // recombine, scale by t/q, round, reduce, fold — in one kernel (the integers are never stored)
    return z_crt_mdr(z, R, R + 3 * words, R + 6 * words, nullptr, (u64 *)d_c, 3 * batch, n, q, t, q, st);
}

// The relinearisation key in the form the products consume: for each of the K CRT primes, NTT_k(rlk0 mod P_k) and
// NTT_k(rlk1 mod P_k) zero-padded to 2n — K x 2 rows of 2n words.  A key relinearises every product of a
// computation (bfv/src/lib.rs:87-90 passes the same rlk to each RLWE::mul), so it is prepared once.
// Two forms: the key split at bit h into two halves and ONE prime (h > 0: one forward transform of c2 per ciphertext
// instead of two, no Garner step) whenever a half-product fits the prime: bits(q-1) + h + log2 n <= 60 with
// h = ceil(bits(pq-1) / 2); otherwise K primes and the whole key.  Either way 4 or 2K key rows of 2n words.
static unsigned relin_split_bits(uint64_t q, uint64_t n, uint64_t pq) {
    const unsigned h = (bits_of(pq - 1) + 1) / 2;
    return (h >= 1 && bits_of(q - 1) + h + ceil_log2(n) <= 60) ? h : 0;
}
static int bfv_relin_ctx(ZCtx *z, unsigned *h, uint64_t q, uint64_t n, uint64_t pq, const char *who) {
    int rc = check_pow2_n(n, who);
    if (rc != FHE_OK) return rc;
    if (q < 2 || (q >> 63) || pq < q || (pq >> 63)) return fhe_fail(FHE_E_BAD_Q, "%s: need 2 <= q <= pq < 2^63", who);
    *h = relin_split_bits(q, n, pq);
    return zctx_init(z, 2 * n, *h ? 1 : primes_for_bits(bits_of(q - 1) + bits_of(pq - 1) + ceil_log2(n), false));
}

extern "C" size_t fhe_bfv_rlk_prepared_words(uint64_t q, uint64_t n, uint64_t pq) {
    if (n < 2 || (n & (n - 1)) || n > (1ull << 19) || q < 2 || (q >> 63) || pq < q || (pq >> 63)) return 0;
    if (bfv32_on(q, n, pq)) return (size_t)6 * n;                            // 3 primes x 2 rows of 2n u32
    if (relin_split_bits(q, n, pq)) return (size_t)4 * 2 * n;
    const int K = primes_for_bits(bits_of(q - 1) + bits_of(pq - 1) + ceil_log2(n), false);
    return K >= 1 && K <= 3 ? (size_t)K * 2 * 2 * n : 0;
}

// prepared key into `prep` (fhe_bfv_rlk_prepared_words words); `scratch` = 4n words for the split form
static int bfv_rlk_prepare(const ZCtx &z, unsigned h, uint64_t n, const u64 *d_rlk, u64 *prep, u64 *scratch, hipStream_t st) {
    int rc;
    if (h) {
        { fhe::KernelTimer kt_("zr_split_h", 0, st);
        hipLaunchKernelGGL(fhe::zr_split_h_kernel, dim3(fhe_ew_grid(2 * n)), dim3(256), 0, st, d_rlk, scratch, (u64)2, (u32)n, (u32)h);
        }
        LAUNCH_OK("zr_split_h_kernel");
        return z_forward_src(z, 0, scratch, prep, 4, n, st);
    }
    for (int k = 0; k < z.K; k++)
        if ((rc = z_forward_src(z, k, d_rlk, prep + (u64)k * 4 * n, 2, n, st)) != FHE_OK) return rc;
    return FHE_OK;
}

extern "C" int fhe_bfv_rlk_prepare_dev(uint64_t q, uint64_t n, uint64_t pq, const void *d_rlk, void *d_prepared, void *hip_stream) {
    ZCtx z;
    unsigned h = 0;
    int rc = bfv_relin_ctx(&z, &h, q, n, pq, "fhe_bfv_rlk_prepare_dev");
    if (rc != FHE_OK) return rc;
    if (!d_rlk || !d_prepared) return fhe_fail(FHE_E_NULL, "fhe_bfv_rlk_prepare_dev: NULL buffer");
    REQUIRE_ALIGNED(d_rlk); REQUIRE_ALIGNED(d_prepared);
    hipStream_t st = (hipStream_t)hip_stream;
    if (bfv32_on(q, n, pq)) return bfv32_rlk_prepare(n, (const u64 *)d_rlk, d_prepared, st);
    void *scratch = nullptr;
    if (h && (rc = fhe_workspace_get(1, 4 * n * 8, st, &scratch)) != FHE_OK) return rc;
    return bfv_rlk_prepare(z, h, n, (const u64 *)d_rlk, (u64 *)d_prepared, (u64 *)scratch, st);
}

// the products against a prepared key: d_prep as fhe_bfv_rlk_prepare_dev leaves it
static int bfv_relinearize_with(const ZCtx &z, unsigned h, uint64_t q, uint64_t n, uint64_t pq, const u64 *d_prep, const void *d_c,
                                void *d_out, size_t batch, hipStream_t st) {
    const u64 n2 = 2 * n, p = pq / q;
    const u64 words = batch * n2, bn = batch * n;
    const unsigned sets = h ? 4 : 2 * (unsigned)z.K;          // product rows per ciphertext
    void *wsv = nullptr;
    int rc = fhe_workspace_get(1, (1 + (size_t)sets) * words * 8, st, &wsv);
    if (rc != FHE_OK) return rc;
    u64 *X = (u64 *)wsv, *R = X + words;
    const u64 *c2 = (const u64 *)d_c + 2 * bn;
    if (h) {   // one prime, key halves: rows [rlk0_lo | rlk0_hi | rlk1_lo | rlk1_hi] x batch
        if ((rc = z_forward_src(z, 0, c2, X, batch, n, st)) != FHE_OK) return rc;
        { fhe::KernelTimer kt_("zr_mul_bcast", 0, st);
        hipLaunchKernelGGL(fhe::zr_mul_bcast_kernel, dim3(fhe_ew_grid(words)), dim3(256), 0, st, (const u64 *)X, d_prep, R, (u64)batch, (u32)n2, (u32)4, z.cc.m[0]);
        }
        LAUNCH_OK("zr_mul_bcast_kernel");
        if ((rc = z_inverse(z, 0, R, R, 4 * batch, st)) != FHE_OK) return rc;
        { fhe::KernelTimer kt_("zr_split_mdr", 0, st);
        hipLaunchKernelGGL(fhe::zr_split_mdr_kernel, dim3(fhe_ew_grid(2 * bn)), dim3(256), 0, st, (const u64 *)R, (const u64 *)d_c, (u64 *)d_out, (u64)batch, (u32)n, (u32)h, (u64)q, (u64)1, (u64)p);
        }
        LAUNCH_OK("zr_split_mdr_kernel");
        return FHE_OK;
    }
    for (int k = 0; k < z.K; k++) {
        if ((rc = z_forward_src(z, k, c2, X, batch, n, st)) != FHE_OK) return rc;
        u64 *Rk = R + 2 * (u64)k * words;
        { fhe::KernelTimer kt_("zr_mul_bcast", 0, st);
        hipLaunchKernelGGL(fhe::zr_mul_bcast_kernel, dim3(fhe_ew_grid(words)), dim3(256), 0, st, (const u64 *)X, d_prep + (u64)k * 2 * n2, Rk, (u64)batch, (u32)n2, (u32)2, z.cc.m[k]);
        }
        LAUNCH_OK("zr_mul_bcast_kernel");
        if ((rc = z_inverse(z, k, Rk, Rk, 2 * batch, st)) != FHE_OK) return rc;
    }
    // (c0, c1) + mul_div_round(crt(..), 1, p): recombination, scaling, fold and the final add in one kernel
    return z_crt_mdr(z, R, R + 2 * words, R + 4 * words, (const u64 *)d_c, (u64 *)d_out, 2 * batch, n, q, 1, p, st);
}

extern "C" int fhe_bfv_relinearize_prepared_dev(uint64_t q, uint64_t n, uint64_t pq, const void *d_prepared, const void *d_c,
                                                void *d_out, size_t batch, void *hip_stream) {
    ZCtx z;
    unsigned h = 0;
    int rc = bfv_relin_ctx(&z, &h, q, n, pq, "fhe_bfv_relinearize_prepared_dev");
    if (rc != FHE_OK) return rc;
    if (batch == 0) return FHE_OK;
    if (!d_prepared || !d_c || !d_out) return fhe_fail(FHE_E_NULL, "fhe_bfv_relinearize_prepared_dev: NULL buffer");
    REQUIRE_ALIGNED(d_prepared); REQUIRE_ALIGNED(d_c); REQUIRE_ALIGNED(d_out);
    if (bfv32_on(q, n, pq)) return bfv32_relinearize(q, n, pq, d_prepared, d_c, d_out, batch, (hipStream_t)hip_stream);
    return bfv_relinearize_with(z, h, q, n, pq, (const u64 *)d_prepared, d_c, d_out, batch, (hipStream_t)hip_stream);
}

// d_rlk: [rlk0 | rlk1], each n words mod pq (one key for the whole batch).
// d_c: [c0 | c1 | c2] as produced by fhe_bfv_tensor_dev.  d_out: [o0 | o1], each batch x n.
extern "C" int fhe_bfv_relinearize_dev(uint64_t q, uint64_t n, uint64_t pq, const void *d_rlk, const void *d_c,
                                       void *d_out, size_t batch, void *hip_stream) {
    ZCtx z;
    unsigned h = 0;
    int rc = bfv_relin_ctx(&z, &h, q, n, pq, "fhe_bfv_relinearize_dev");
    if (rc != FHE_OK) return rc;
    if (batch == 0) return FHE_OK;
    if (!d_rlk || !d_c || !d_out) return fhe_fail(FHE_E_NULL, "fhe_bfv_relinearize_dev: NULL buffer");
    REQUIRE_ALIGNED(d_rlk); REQUIRE_ALIGNED(d_c); REQUIRE_ALIGNED(d_out);
    // the key prepared on the fly, in workspace slot 0 behind the tensor result that fhe_bfv_mul_dev keeps there
    // (12n words of prepared key at most, then 4n words of scratch for the split)
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t tensor_words = 3 * batch * n;
    void *w0 = nullptr;
    if ((rc = fhe_workspace_get(0, (tensor_words + 16 * n) * 8, st, &w0)) != FHE_OK) return rc;
    u64 *prep = (u64 *)w0 + tensor_words;
    if (bfv32_on(q, n, pq)) {                                  // 6n words of prepared key
        if ((rc = bfv32_rlk_prepare(n, (const u64 *)d_rlk, prep, st)) != FHE_OK) return rc;
        return bfv32_relinearize(q, n, pq, prep, d_c, d_out, batch, st);
    }
    if ((rc = bfv_rlk_prepare(z, h, n, (const u64 *)d_rlk, prep, prep + 12 * n, st)) != FHE_OK) return rc;
    return bfv_relinearize_with(z, h, q, n, pq, prep, d_c, d_out, batch, st);
}

static int bfv_mul_common(uint64_t q, uint64_t n, uint64_t t, uint64_t pq, const void *d_rlk, bool prepared, const void *d_ab,
                          void *d_out, size_t batch, void *hip_stream) {
    if (batch == 0) return FHE_OK;
    // the tensor result lives in workspace slot 0 (both stages use slot 1); a stream-ordered
    // allocation per call cost up to 2 ms at 2048 ciphertexts whenever the pool had trimmed itself.
    // Sized for the on-the-fly key as well, so that fhe_bfv_relinearize_dev's request does not move it.
    const size_t kw = 16 * (size_t)n;
    void *c = nullptr;
    int rc = fhe_workspace_get(0, (3 * batch * n + kw) * 8, (hipStream_t)hip_stream, &c);
    if (rc != FHE_OK) return rc;
    rc = fhe_bfv_tensor_dev(q, n, t, d_ab, c, batch, hip_stream);
    if (rc != FHE_OK) return rc;
    return prepared ? fhe_bfv_relinearize_prepared_dev(q, n, pq, d_rlk, c, d_out, batch, hip_stream)
                    : fhe_bfv_relinearize_dev(q, n, pq, d_rlk, c, d_out, batch, hip_stream);
}

extern "C" int fhe_bfv_mul_dev(uint64_t q, uint64_t n, uint64_t t, uint64_t pq, const void *d_rlk, const void *d_ab,
                               void *d_out, size_t batch, void *hip_stream) {
    return bfv_mul_common(q, n, t, pq, d_rlk, false, d_ab, d_out, batch, hip_stream);
}
extern "C" int fhe_bfv_mul_prepared_dev(uint64_t q, uint64_t n, uint64_t t, uint64_t pq, const void *d_prepared, const void *d_ab,
                                        void *d_out, size_t batch, void *hip_stream) {
    return bfv_mul_common(q, n, t, pq, d_prepared, true, d_ab, d_out, batch, hip_stream);
}

// ---- TFHE: Tn x Tn -----------------------------------------------------------------------------
extern "C" int fhe_tn_mul_dev(uint64_t n, const void *d_a, const void *d_b, void *d_out, size_t batch,
                              void *hip_stream) {
    int rc = check_pow2_n(n, "fhe_tn_mul_dev");
    if (rc != FHE_OK) return rc;
    if (batch == 0) return FHE_OK;
    if (!d_a || !d_b || !d_out) return fhe_fail(FHE_E_NULL, "fhe_tn_mul_dev: NULL buffer");
    REQUIRE_ALIGNED(d_a); REQUIRE_ALIGNED(d_b); REQUIRE_ALIGNED(d_out);
    ZCtx z;
    rc = zctx_init(&z, n, primes_for_bits(128 + ceil_log2(n), true));   // |c_k| < n * 2^128
    if (rc != FHE_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    const u64 words = batch * n;
    void *wsv = nullptr;
    rc = fhe_workspace_get(1, (2 + (size_t)z.K) * words * 8, st, &wsv);
    if (rc != FHE_OK) return rc;
    u64 *A = (u64 *)wsv, *B = A + words, *R = B + words;
    for (int k = 0; k < z.K; k++) {
        if ((rc = z_forward_src(z, k, (const u64 *)d_a, A, batch, n, st)) != FHE_OK) return rc;
        if ((rc = z_forward_src(z, k, (const u64 *)d_b, B, batch, n, st)) != FHE_OK) return rc;
        hipError_t e = fhe::launch_ntt_inverse(z.dp[k], A, B, nullptr, R + (u64)k * words, batch,
                                               fhe_batch_tile_for(z.plan[k]), st);
        if (e != hipSuccess) return fhe_hip_fail(e, "zring inverse(A.*B)");
    }
    return z_crt(z, true, R, R + words, R + 2 * words, (u64 *)d_out, words, st);
}

// ---- the two-small-prime (27-bit) form of the external product (digit32.hip): per (n, device) tables ---------------------------
namespace {
struct Ext32Tables {
    fhe::Tw32 *fwd[3] = {nullptr, nullptr, nullptr}, *inv[3] = {nullptr, nullptr, nullptr};
    uint32_t *lut[3] = {nullptr, nullptr, nullptr};
    fhe::Tw32 ninv[3]{};
};
std::mutex g_e32_lock;
std::map<std::pair<u64, int>, Ext32Tables> g_e32;
inline fhe::Tw32 tw32(u64 w, u64 p) { return fhe::Tw32{(uint32_t)w, (uint32_t)((w << 32) / p)}; }
}  // namespace

void fhe_ext32_free_all() {
    std::lock_guard<std::mutex> lk(g_e32_lock);
    for (auto &kv : g_e32)
        for (int i = 0; i < 3; i++) {
            if (kv.second.fwd[i]) (void)hipFree(kv.second.fwd[i]);
            if (kv.second.inv[i]) (void)hipFree(kv.second.inv[i]);
            if (kv.second.lut[i]) (void)hipFree(kv.second.lut[i]);
        }
    g_e32.clear();
}

// fills the per-prime fields of `a` (tables on the current device, built on first use from the same plans — psi by the
// reference's search, roots[i] = psi^bitrev(i) — as every other transform of the library)
int fhe_ext32_tables(uint64_t n, fhe::Ext32Args *a) {
    int dev = 0;
    int rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    const u64 primes[3] = {fhe::kExt32PrimeA, fhe::kExt32PrimeB, fhe::kExt32PrimeC};
    std::lock_guard<std::mutex> lk(g_e32_lock);
    Ext32Tables &t = g_e32[std::make_pair(n, dev)];
    if (!t.fwd[0]) {
        for (int i = 0; i < 3; i++) {
            const u64 p = primes[i];
            const fhe_ntt_plan *plan = nullptr;
            if ((rc = fhe_ntt_plan_get(p, n, &plan)) != FHE_OK) return rc;
            std::vector<fhe::Tw32> f(n), v(n);
            for (u64 k = 0; k < n; k++) { f[k] = tw32(plan->roots[k], p); v[k] = tw32(plan->roots_inv[k], p); }
            // bit tables: as capi.hip builds them for the 61-bit plans (ntt_rounds.hpp: round0_bits)
            std::vector<uint32_t> lut(136, 0);
            const u64 *r = plan->roots.data();
            auto add = [&](u64 x, u64 y) { u64 z = x + y; return z >= p ? z - p : z; };
            auto sub = [&](u64 x, u64 y) { return x >= y ? x - y : x + p - y; };
            auto mul = [&](u64 x, u64 y) { return (x * y) % p; };
            for (unsigned pt = 0; pt < 16; pt++) {
                const u64 x0 = pt & 1, x1 = (pt >> 1) & 1, x2 = (pt >> 2) & 1, x3 = (pt >> 3) & 1;
                const u64 a0 = add(x0, mul(r[1], x2)), a2 = sub(x0, mul(r[1], x2));
                const u64 a1 = add(x1, mul(r[1], x3)), a3 = sub(x1, mul(r[1], x3));
                u64 y[4];
                y[0] = add(a0, mul(r[2], a1)); y[1] = sub(a0, mul(r[2], a1));
                y[2] = add(a2, mul(r[3], a3)); y[3] = sub(a2, mul(r[3], a3));
                for (int j = 0; j < 4; j++) {
                    lut[4 * pt + j] = (uint32_t)y[j];
                    lut[64 + 4 * pt + j] = (uint32_t)mul(r[4 + j], y[j]);
                }
            }
            for (unsigned pt = 0; pt < 4; pt++) {
                const u64 x = pt & 1, y = (pt >> 1) & 1;
                lut[128 + 2 * pt] = (uint32_t)add(x, mul(r[1], y));
                lut[128 + 2 * pt + 1] = (uint32_t)sub(x, mul(r[1], y));
            }
            fhe::Tw32 *df = nullptr, *di = nullptr;
            uint32_t *dl = nullptr;
            hipError_t e = hipMalloc((void **)&df, n * sizeof(fhe::Tw32));
            if (e == hipSuccess) e = hipMalloc((void **)&di, n * sizeof(fhe::Tw32));
            if (e == hipSuccess) e = hipMalloc((void **)&dl, lut.size() * 4);
            if (e == hipSuccess) e = hipMemcpy(df, f.data(), n * sizeof(fhe::Tw32), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(di, v.data(), n * sizeof(fhe::Tw32), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(dl, lut.data(), lut.size() * 4, hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                if (df) (void)hipFree(df);
                if (di) (void)hipFree(di);
                if (dl) (void)hipFree(dl);
                for (int j = 0; j < i; j++) { (void)hipFree(t.fwd[j]); (void)hipFree(t.inv[j]); (void)hipFree(t.lut[j]); }
                t = Ext32Tables();
                return fhe_hip_fail(e, "uploading the 27-bit tables");
            }
            t.fwd[i] = df; t.inv[i] = di; t.lut[i] = dl;
            t.ninv[i] = tw32(plan->n_inv, p);
        }
    }
    const u64 pA = primes[0], pB = primes[1];
    for (int i = 0; i < 3; i++) {
        a->tw_fwd[i] = t.fwd[i]; a->tw_inv[i] = t.inv[i]; a->lut[i] = t.lut[i];
        a->p[i] = (uint32_t)primes[i];
        a->mu[i] = ~0ull / primes[i];                 // floor(2^64 / p): p does not divide 2^64
        a->bq[i] = (uint32_t)(0xffffffffull / primes[i]);
        a->ninv[i] = t.ninv[i];
    }
    a->crt = tw32(hpow(pA % pB, pB - 2, pB), pB);     // pA^-1 mod pB
    const u64 pC = primes[2];
    a->crt_ac = tw32(hpow(pA % pC, pC - 2, pC), pC);
    a->crt_bc = tw32(hpow(pB % pC, pC - 2, pC), pC);
    a->P = pA * pB;
    a->halfP = (a->P + 1) / 2;
    return FHE_OK;
}
bool fhe_ext32_enabled() {
    static const bool on = [] { const char *e = getenv("FHE_EXT32"); return !(e && e[0] == '0'); }();
    return on;
}
static bool ext32_on(u64 n, unsigned k, unsigned l) { return fhe_ext32_enabled() && fhe::ext32_shape_supported(n, k, l); }

// ---- TFHE: TGGSW x TGLWE external product -------------------------------------------------------
static bool one_prime_form(u64 n, unsigned k, unsigned l) {
    return (u64)(k + 1) * l * n <= (1ull << 26) && n >= 16 && n <= (1ull << fhe::kMaxSinglePassLog);
}

// The TGGSW key as the external product consumes it: every word split into 32-bit halves, laid out
// [t = TGLev*l + level][half][component][n], forward-transformed modulo P1 — (k+1)*l*2*(k+1) rows of n
// words.  A bootstrapping key is used by every product of a blind rotation (tfhe/src/tlwe.rs), so it
// is prepared ONCE; 0 words = this shape takes the two-prime form and has no prepared layout.
extern "C" size_t fhe_tggsw_prepared_words(uint64_t n, unsigned k, unsigned l) {
    if (n < 2 || (n & (n - 1)) || l < 1 || l > 64 || k < 1 || k > 64 || !one_prime_form(n, k, l)) return 0;
    return (size_t)2 * (k + 1) * l * (k + 1) * n;
}

extern "C" int fhe_tggsw_prepare_dev(uint64_t n, unsigned k, unsigned l, const void *d_tggsw, void *d_prepared, void *hip_stream) {
    int rc = check_pow2_n(n, "fhe_tggsw_prepare_dev");
    if (rc != FHE_OK) return rc;
    if (l < 1 || l > 64 || k < 1 || k > 64) return fhe_fail(FHE_E_INVALID, "fhe_tggsw_prepare_dev: need 1 <= l <= 64, 1 <= k <= 64");
    if (!one_prime_form(n, k, l)) return fhe_fail(FHE_E_INVALID, "fhe_tggsw_prepare_dev: no prepared form for n=%llu, k=%u, l=%u (fhe_tggsw_prepared_words is 0)", (unsigned long long)n, k, l);
    if (!d_tggsw || !d_prepared) return fhe_fail(FHE_E_NULL, "fhe_tggsw_prepare_dev: NULL buffer");
    REQUIRE_ALIGNED(d_tggsw); REQUIRE_ALIGNED(d_prepared);
    ZCtx z1;
    if ((rc = zctx_init(&z1, n, 1)) != FHE_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    const u32 k1 = k + 1;
    const u64 T = (u64)k1 * l, grows = T * k1;
    if (ext32_on(n, k, l)) {
        // two 27-bit primes (digit32.hip): the halves of every word transformed per prime into d_prepared as u32
        // [prime][t][half][c][n] — the same number of bytes as the 61-bit form
        fhe::Ext32Args a{};
        if ((rc = fhe_ext32_tables(n, &a)) != FHE_OK) return rc;
        a.key64 = (const u64 *)d_tggsw; a.key32 = (uint32_t *)d_prepared; a.rows = 2 * grows; a.key_k1 = k1;
        hipError_t e = fhe::launch_ext32_key(a, (int)z1.dp[0].log_n, st);
        return e == hipSuccess ? FHE_OK : fhe_hip_fail(e, "ntt32_fwd_key_kernel");
    }
    { fhe::KernelTimer kt_("zr_split32", 0, st);
    hipLaunchKernelGGL(fhe::zr_split32_kernel, dim3(fhe_ew_grid(grows * n)), dim3(256), 0, st, (const u64 *)d_tggsw, (u64 *)d_prepared, T, k1, (u32)n);
    }
    LAUNCH_OK("zr_split32_kernel");
    return z_forward(z1, 0, (const u64 *)d_prepared, (u64 *)d_prepared, 2 * grows, st);          // halves are < 2^32 < P1
}

extern "C" int fhe_tggsw_external_product_prepared_dev(uint64_t n, unsigned k, unsigned l, const void *d_prepared,
                                                       const void *d_tglwe, void *d_out, size_t batch, void *hip_stream) {
    int rc = check_pow2_n(n, "fhe_tggsw_external_product_prepared_dev");
    if (rc != FHE_OK) return rc;
    if (l < 1 || l > 64 || k < 1 || k > 64) return fhe_fail(FHE_E_INVALID, "external product: need 1 <= l <= 64, 1 <= k <= 64");
    if (!one_prime_form(n, k, l)) return fhe_fail(FHE_E_INVALID, "fhe_tggsw_external_product_prepared_dev: no prepared form for this shape");
    if (batch == 0) return FHE_OK;
    if (!d_prepared || !d_tglwe || !d_out) return fhe_fail(FHE_E_NULL, "fhe_tggsw_external_product_prepared_dev: NULL buffer");
    REQUIRE_ALIGNED(d_prepared); REQUIRE_ALIGNED(d_tglwe); REQUIRE_ALIGNED(d_out);
    ZCtx z1;
    if ((rc = zctx_init(&z1, n, 1)) != FHE_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    const u32 k1 = k + 1;
    const u64 T = (u64)k1 * l, orows = batch * k1, drows = batch * T;
    const u64 *G2 = (const u64 *)d_prepared;
    void *wsv = nullptr;
    if (ext32_on(n, k, l)) {
        fhe::Ext32Args a{};
        if ((rc = fhe_ext32_tables(n, &a)) != FHE_OK) return rc;
        // parts: enough workgroups to fill the chip several times over (n <= 1024: four workgroups' worth of LDS per CU),
        // only as many as leave no CU empty above (one or two workgroups per CU: see fhe_glwe_key_switch_dev)
        const u32 W = fhe::ext32_units((int)z1.dp[0].log_n);
        u32 parts32 = 1;
        const u64 slots = n <= 1024 ? 2048 : n == 2048 ? 512 : 256;
        while (parts32 < 8 && batch * parts32 < slots && (T / (parts32 * 2)) >= 2 * W) parts32 *= 2;
        if ((rc = fhe_workspace_get(1, (u64)batch * parts32 * 2 * (2 * k1) * n * 4, st, &wsv)) != FHE_OK) return rc;
        a.k = k;
        a.key32 = (uint32_t *)const_cast<void *>(d_prepared);
        a.src = (const u64 *)d_tglwe; a.ct_stride = (u64)k1 * n; a.part32 = (uint32_t *)wsv; a.out = (u64 *)d_out; a.batch = batch;
        a.l = l; a.T = (u32)T; a.parts = parts32;
        a.tpp = (((u32)T + parts32 - 1) / parts32 + W - 1) / W * W;
        hipError_t e = fhe::launch_ext32_mac(a, (int)z1.dp[0].log_n, fhe::SRC_DIGITS, st);
        if (e == hipSuccess) e = fhe::launch_ext32_tail(a, (int)z1.dp[0].log_n, st);
        return e == hipSuccess ? FHE_OK : fhe_hip_fail(e, "digit32 kernels");
    }
    // Fused form: digit extraction, forward transform and multiply-accumulate in one kernel
    // (digit_mac.hip); the digit transforms never reach memory.  R[b][half][c] = sum_t G2[t][half][c] * NTT(digit_t(b)).
    static const bool fused_on = [] { const char *e = getenv("FHE_DIGIT_MAC_FUSED"); return !(e && e[0] == '0'); }();
    const u32 parts = fhe::digit_mac_parts(batch, (u32)T, z1.dp[0].log_n, 2 * k1);
    if (fused_on) {
        if ((rc = fhe_workspace_get(1, ((u64)parts + 1) * 2 * orows * n * 8, st, &wsv)) != FHE_OK) return rc;
        u64 *R = (u64 *)wsv, *PART = parts > 1 ? R + 2 * orows * n : R;
        hipError_t e = fhe::launch_digit_mac(z1.dp[0], fhe::SRC_DIGITS, (const u64 *)d_tglwe, (u64)k1 * n, k1, l, G2, 2 * k1, PART, parts, batch, st);
        if (e == hipSuccess) {
            // sum of the parts, the 2(k+1) inverse transforms and the recombination of the halves: one kernel
            // where a workgroup holds whole ciphertexts (4096/n >= 2(k+1) rows), three otherwise
            e = fhe::launch_digit_tail_torus(z1.dp[0], PART, parts, k1, z1.cc.half1, (u64 *)d_out, batch, st);
            if (e == hipSuccess) return FHE_OK;
            if (e != hipErrorNotSupported) return fhe_hip_fail(e, "digit_tail_kernel");
            (void)hipGetLastError();
            if (parts > 1 && (e = fhe::launch_sum_parts(PART, R, batch, parts, 2ull * k1 * n, z1.cc.m[0].q, st)) != hipSuccess)
                return fhe_hip_fail(e, "sum_parts_kernel");
            if ((rc = z_inverse(z1, 0, R, R, 2 * orows, st)) != FHE_OK) return rc;
            { fhe::KernelTimer kt_("zr_combine32", 0, st);
            hipLaunchKernelGGL(fhe::zr_combine32_kernel, dim3(fhe_ew_grid(orows * n)), dim3(256), 0, st, (const u64 *)R, (u64 *)d_out, (u64)batch, k1, (u32)n, z1.cc.p1, z1.cc.half1);
            }
            LAUNCH_OK("zr_combine32_kernel");
            return FHE_OK;
        }
        if (e != hipErrorNotSupported) return fhe_hip_fail(e, "digit_mac_kernel");
        (void)hipGetLastError();
    }
    // unfused: every digit transform written to D, then one multiply-accumulate pass over it
    if ((rc = fhe_workspace_get(1, (drows + 2 * orows) * n * 8, st, &wsv)) != FHE_OK) return rc;
    u64 *D = (u64 *)wsv, *R = D + drows * n;
    hipError_t e = fhe::launch_ntt_forward_digits(z1.dp[0], (const u64 *)d_tglwe, D, orows, (u32)l, st);
    if (e != hipSuccess) return fhe_hip_fail(e, "digit forward NTT");
    { fhe::KernelTimer kt_("mac_rows", 0, st);
    hipLaunchKernelGGL((fhe::mac_rows_kernel<>), dim3(fhe_ew_grid(fhe::mac_rows_threads(batch, 2 * k1, n))), dim3(256), 0, st, G2, (const u64 *)D, R, (u64)batch, (u32)n, (u32)T, 2 * k1, (u64)0, z1.cc.m[0]);
    }
    LAUNCH_OK("mac_rows_kernel");
    e = fhe::launch_digit_tail_torus(z1.dp[0], R, 1, k1, z1.cc.half1, (u64 *)d_out, batch, st);     // inverse + recombination
    if (e == hipSuccess) return FHE_OK;
    if (e != hipErrorNotSupported) return fhe_hip_fail(e, "digit_tail_kernel");
    (void)hipGetLastError();
    if ((rc = z_inverse(z1, 0, R, R, 2 * orows, st)) != FHE_OK) return rc;
    { fhe::KernelTimer kt_("zr_combine32", 0, st);
    hipLaunchKernelGGL(fhe::zr_combine32_kernel, dim3(fhe_ew_grid(orows * n)), dim3(256), 0, st, (const u64 *)R, (u64 *)d_out, (u64)batch, k1, (u32)n, z1.cc.p1, z1.cc.half1);
    }
    LAUNCH_OK("zr_combine32_kernel");
    return FHE_OK;
}


// d_tggsw [(k+1)][l][(k+1)][n], one key for the batch; d_tglwe [batch][(k+1)][n]; d_out likewise.
extern "C" int fhe_tggsw_external_product_dev(uint64_t n, unsigned k, unsigned l, const void *d_tggsw,
                                              const void *d_tglwe, void *d_out, size_t batch, void *hip_stream) {
    int rc = check_pow2_n(n, "fhe_tggsw_external_product_dev");
    if (rc != FHE_OK) return rc;
    if (l < 1 || l > 64 || k < 1 || k > 64) return fhe_fail(FHE_E_INVALID, "external product: need 1 <= l <= 64, 1 <= k <= 64");
    if (batch == 0) return FHE_OK;
    if (!d_tggsw || !d_tglwe || !d_out) return fhe_fail(FHE_E_NULL, "fhe_tggsw_external_product_dev: NULL buffer");
    REQUIRE_ALIGNED(d_tggsw); REQUIRE_ALIGNED(d_tglwe); REQUIRE_ALIGNED(d_out);
    const u32 k1 = k + 1;
    hipStream_t st = (hipStream_t)hip_stream;
    const u64 grows = (u64)k1 * l * k1, drows = batch * k1 * l, orows = batch * k1;
    // One-prime form (see zr_split32_kernel): the digit transforms — the dominant cost — are needed
    // for ONE prime instead of two.  Valid while each half-sum stays below P1/2: (k+1)*l*n <= 2^26.
    // Needs the bit-extracting transform (single-pass sizes); larger n takes the two-prime form.
    if (one_prime_form(n, k, l)) {
        // the key prepared on the fly (a caller with a long-lived key prepares it once:
        // fhe_tggsw_prepare_dev + fhe_tggsw_external_product_prepared_dev)
        void *wsv = nullptr;
        if ((rc = fhe_workspace_get(0, 2 * grows * n * 8, st, &wsv)) != FHE_OK) return rc;
        if ((rc = fhe_tggsw_prepare_dev(n, k, l, d_tggsw, wsv, hip_stream)) != FHE_OK) return rc;
        return fhe_tggsw_external_product_prepared_dev(n, k, l, wsv, d_tglwe, d_out, batch, hip_stream);
    }
    ZCtx z;
    // digits are 0/1: |sum| < (k+1) * l * n * 2^64
    rc = zctx_init(&z, n, primes_for_bits(64 + ceil_log2(n) + ceil_log2((u64)k1 * l), true));
    if (rc != FHE_OK) return rc;
    void *wsv = nullptr;
    rc = fhe_workspace_get(1, (grows + 2 * drows + (size_t)z.K * orows) * n * 8, st, &wsv);
    if (rc != FHE_OK) return rc;
    u64 *G = (u64 *)wsv, *Dg = G + grows * n, *D = Dg + drows * n, *R = D + drows * n;
    bool digits_done = false;
    for (int kk = 0; kk < z.K; kk++) {
        if ((rc = z_forward_src(z, kk, (const u64 *)d_tggsw, G, grows, n, st)) != FHE_OK) return rc;
        // D = NTT of the 0/1 digit polynomials (digits are < every prime).  Single-pass sizes
        // extract the bit in the transform's load; larger n materialises the digits once.
        hipError_t e = fhe::launch_ntt_forward_digits(z.dp[kk], (const u64 *)d_tglwe, D, orows, (u32)l, st);
        if (e == hipErrorNotSupported) {
            if (!digits_done) {
                { fhe::KernelTimer kt_("zr_digits", 0, st);
                hipLaunchKernelGGL(fhe::zr_digits_kernel, dim3(fhe_ew_grid(drows * n)), dim3(256), 0, st, (const u64 *)d_tglwe, Dg, (u64)orows, (u32)n, (u32)l);
                }
                LAUNCH_OK("zr_digits_kernel");
                digits_done = true;
            }
            if ((rc = z_forward(z, kk, Dg, D, drows, st)) != FHE_OK) return rc;
        } else if (e != hipSuccess) {
            return fhe_hip_fail(e, "digit forward NTT");
        }
        u64 *Rk = R + (u64)kk * orows * n;
        // out[b][c] = sum_{i<k1,d<l} G[i][d][c] * D[b][i][d]  (tggsw.rs:57-59,145): T = k1*l terms, k1 rows
        { fhe::KernelTimer kt_("mac_rows", 0, st);
        hipLaunchKernelGGL((fhe::mac_rows_kernel<>), dim3(fhe_ew_grid(fhe::mac_rows_threads(batch, k1, n))), dim3(256), 0, st, (const u64 *)G, (const u64 *)D, Rk, (u64)batch, (u32)n, (u32)(k1 * l), (u32)k1, (u64)0, z.cc.m[kk]);
        }
        LAUNCH_OK("mac_rows_kernel");
        if ((rc = z_inverse(z, kk, Rk, Rk, orows, st)) != FHE_OK) return rc;
    }
    return z_crt(z, true, R, R + orows * n, R + 2 * orows * n, (u64 *)d_out, orows * n, st);
}

// ---- TFHE: TGLWE x Tn (plaintext product) and TGLev x Vec<Tn> -------------------------------------
// tfhe/src/tglwe.rs:182-194: every component of the ciphertext times one torus polynomial.
// d_tglwe, d_out [batch][(k+1)][n]; d_p [batch][n].  Full 64 x 64-bit operands: 3 primes.
extern "C" int fhe_tglwe_mul_tn_dev(uint64_t n, unsigned k, const void *d_tglwe, const void *d_p, void *d_out,
                                    size_t batch, void *hip_stream) {
    int rc = check_pow2_n(n, "fhe_tglwe_mul_tn_dev");
    if (rc != FHE_OK) return rc;
    if (k < 1 || k > 64) return fhe_fail(FHE_E_INVALID, "fhe_tglwe_mul_tn_dev: need 1 <= k <= 64");
    if (batch == 0) return FHE_OK;
    if (!d_tglwe || !d_p || !d_out) return fhe_fail(FHE_E_NULL, "fhe_tglwe_mul_tn_dev: NULL buffer");
    REQUIRE_ALIGNED(d_tglwe); REQUIRE_ALIGNED(d_p); REQUIRE_ALIGNED(d_out);
    const u32 k1 = k + 1;
    ZCtx z;
    rc = zctx_init(&z, n, primes_for_bits(128 + ceil_log2(n), true));   // |c_j| < n * 2^128
    if (rc != FHE_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    const u64 rows = batch * k1;
    void *wsv = nullptr;
    rc = fhe_workspace_get(1, (rows + batch + (size_t)z.K * rows) * n * 8, st, &wsv);
    if (rc != FHE_OK) return rc;
    u64 *A = (u64 *)wsv, *P = A + rows * n, *R = P + batch * n;
    for (int kk = 0; kk < z.K; kk++) {
        if ((rc = z_forward_src(z, kk, (const u64 *)d_tglwe, A, rows, n, st)) != FHE_OK) return rc;
        if ((rc = z_forward_src(z, kk, (const u64 *)d_p, P, batch, n, st)) != FHE_OK) return rc;
        u64 *Rk = R + (u64)kk * rows * n;
        // T = 1 term, nc = k+1 rows, "key" = the ciphertext itself (per batch element)
        { fhe::KernelTimer kt_("mac_rows", 0, st);
        hipLaunchKernelGGL((fhe::mac_rows_kernel<>), dim3(fhe_ew_grid(fhe::mac_rows_threads(batch, k1, n))), dim3(256), 0, st, (const u64 *)A, (const u64 *)P, Rk, (u64)batch, (u32)n, (u32)1, k1, (u64)k1 * n, z.cc.m[kk]);
        }
        LAUNCH_OK("mac_rows_kernel");
        if ((rc = z_inverse(z, kk, Rk, Rk, rows, st)) != FHE_OK) return rc;
    }
    return z_crt(z, true, R, R + rows * n, R + 2 * rows * n, (u64 *)d_out, rows * n, st);
}

// tfhe/src/tggsw.rs:139-149: out[b] = sum_{d<l} tglev[d] * v[b][d]  (TGLWE x Tn summed over the levels).
// d_tglev [l][(k+1)][n] (one for the batch); d_v [batch][l][n] (any 64-bit words); d_out [batch][(k+1)][n].
extern "C" int fhe_tglev_mul_dev(uint64_t n, unsigned k, unsigned l, const void *d_tglev, const void *d_v, void *d_out,
                                 size_t batch, void *hip_stream) {
    int rc = check_pow2_n(n, "fhe_tglev_mul_dev");
    if (rc != FHE_OK) return rc;
    if (l < 1 || l > 64 || k < 1 || k > 64) return fhe_fail(FHE_E_INVALID, "fhe_tglev_mul_dev: need 1 <= l <= 64, 1 <= k <= 64");
    if (batch == 0) return FHE_OK;
    if (!d_tglev || !d_v || !d_out) return fhe_fail(FHE_E_NULL, "fhe_tglev_mul_dev: NULL buffer");
    REQUIRE_ALIGNED(d_tglev); REQUIRE_ALIGNED(d_v); REQUIRE_ALIGNED(d_out);
    const u32 k1 = k + 1;
    ZCtx z;
    rc = zctx_init(&z, n, primes_for_bits(128 + ceil_log2(n) + ceil_log2(l), true));   // |sum| < l * n * 2^128
    if (rc != FHE_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    const u64 grows = (u64)l * k1, vrows = batch * l, orows = batch * k1;
    void *wsv = nullptr;
    rc = fhe_workspace_get(1, (grows + vrows + (size_t)z.K * orows) * n * 8, st, &wsv);
    if (rc != FHE_OK) return rc;
    u64 *G = (u64 *)wsv, *V = G + grows * n, *R = V + vrows * n;
    for (int kk = 0; kk < z.K; kk++) {
        if ((rc = z_forward_src(z, kk, (const u64 *)d_tglev, G, grows, n, st)) != FHE_OK) return rc;
        if ((rc = z_forward_src(z, kk, (const u64 *)d_v, V, vrows, n, st)) != FHE_OK) return rc;
        u64 *Rk = R + (u64)kk * orows * n;
        { fhe::KernelTimer kt_("mac_rows", 0, st);
        hipLaunchKernelGGL((fhe::mac_rows_kernel<>), dim3(fhe_ew_grid(fhe::mac_rows_threads(batch, k1, n))), dim3(256), 0, st, (const u64 *)G, (const u64 *)V, Rk, (u64)batch, (u32)n, (u32)l, k1, (u64)0, z.cc.m[kk]);
        }
        LAUNCH_OK("mac_rows_kernel");
        if ((rc = z_inverse(z, kk, Rk, Rk, orows, st)) != FHE_OK) return rc;
    }
    return z_crt(z, true, R, R + orows * n, R + 2 * orows * n, (u64 *)d_out, orows * n, st);
}

// ---- host-buffer wrappers (what a Rust shim binds) ------------------------------------------------
using HostStage = FheHostStage;

extern "C" int fhe_bfv_mul(uint64_t q, uint64_t n, uint64_t t, uint64_t pq, const uint64_t *rlk, const uint64_t *ab,
                           uint64_t *out, size_t batch) {
    if (batch == 0) return FHE_OK;
    if (!rlk || !ab || !out) return fhe_fail(FHE_E_NULL, "fhe_bfv_mul: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    HostStage hs;
    void *drlk, *dab, *dout;
    if ((rc = hs.up(rlk, 2 * n * 8, &drlk)) != FHE_OK) return rc;
    if ((rc = hs.up(ab, 4 * batch * n * 8, &dab)) != FHE_OK) return rc;
    if ((rc = hs.up(nullptr, 2 * batch * n * 8, &dout)) != FHE_OK) return rc;
    rc = fhe_bfv_mul_dev(q, n, t, pq, drlk, dab, dout, batch, hipStreamPerThread);
    if (rc != FHE_OK) return rc;
    return hs.down(out, dout, 2 * batch * n * 8);
}

extern "C" int fhe_bfv_tensor(uint64_t q, uint64_t n, uint64_t t, const uint64_t *ab, uint64_t *c, size_t batch) {
    if (batch == 0) return FHE_OK;
    if (!ab || !c) return fhe_fail(FHE_E_NULL, "fhe_bfv_tensor: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    HostStage hs;
    void *dab, *dc;
    if ((rc = hs.up(ab, 4 * batch * n * 8, &dab)) != FHE_OK) return rc;
    if ((rc = hs.up(nullptr, 3 * batch * n * 8, &dc)) != FHE_OK) return rc;
    rc = fhe_bfv_tensor_dev(q, n, t, dab, dc, batch, hipStreamPerThread);
    if (rc != FHE_OK) return rc;
    return hs.down(c, dc, 3 * batch * n * 8);
}

extern "C" int fhe_r_naive_mul(uint64_t n, const int64_t *a, const int64_t *b, int64_t *out, size_t batch) {
    if (batch == 0) return FHE_OK;
    if (!a || !b || !out) return fhe_fail(FHE_E_NULL, "fhe_r_naive_mul: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    HostStage hs;
    void *da, *db, *dout;
    if ((rc = hs.up(a, batch * n * 8, &da)) != FHE_OK) return rc;
    if ((rc = hs.up(b, batch * n * 8, &db)) != FHE_OK) return rc;
    if ((rc = hs.up(nullptr, 2 * batch * n * 8, &dout)) != FHE_OK) return rc;
    // operands are read as NON-NEGATIVE 64-bit integers (Rq::to_r gives values in [0,q), ring_n.rs:72-79)
    rc = fhe_r_naive_mul_dev(n, da, db, dout, batch, 64, 64, hipStreamPerThread);
    if (rc != FHE_OK) return rc;
    return hs.down(out, dout, 2 * batch * n * 8);
}

extern "C" int fhe_tn_mul(uint64_t n, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t batch) {
    if (batch == 0) return FHE_OK;
    if (!a || !b || !out) return fhe_fail(FHE_E_NULL, "fhe_tn_mul: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    HostStage hs;
    void *da, *db, *dout;
    if ((rc = hs.up(a, batch * n * 8, &da)) != FHE_OK) return rc;
    if ((rc = hs.up(b, batch * n * 8, &db)) != FHE_OK) return rc;
    if ((rc = hs.up(nullptr, batch * n * 8, &dout)) != FHE_OK) return rc;
    rc = fhe_tn_mul_dev(n, da, db, dout, batch, hipStreamPerThread);
    if (rc != FHE_OK) return rc;
    return hs.down(out, dout, batch * n * 8);
}

extern "C" int fhe_tggsw_external_product(uint64_t n, unsigned k, unsigned l, const uint64_t *tggsw,
                                          const uint64_t *tglwe, uint64_t *out, size_t batch) {
    if (batch == 0) return FHE_OK;
    if (!tggsw || !tglwe || !out) return fhe_fail(FHE_E_NULL, "fhe_tggsw_external_product: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    HostStage hs;
    void *dg, *dt, *dout;
    const size_t k1 = k + 1;
    if ((rc = hs.up(tggsw, k1 * l * k1 * n * 8, &dg)) != FHE_OK) return rc;
    if ((rc = hs.up(tglwe, batch * k1 * n * 8, &dt)) != FHE_OK) return rc;
    if ((rc = hs.up(nullptr, batch * k1 * n * 8, &dout)) != FHE_OK) return rc;
    rc = fhe_tggsw_external_product_dev(n, k, l, dg, dt, dout, batch, hipStreamPerThread);
    if (rc != FHE_OK) return rc;
    return hs.down(out, dout, batch * k1 * n * 8);
}

extern "C" int fhe_tglwe_mul_tn(uint64_t n, unsigned k, const uint64_t *tglwe, const uint64_t *p, uint64_t *out, size_t batch) {
    if (batch == 0) return FHE_OK;
    if (!tglwe || !p || !out) return fhe_fail(FHE_E_NULL, "fhe_tglwe_mul_tn: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    HostStage hs;
    void *dc, *dp, *dout;
    const size_t k1 = (size_t)k + 1;
    if ((rc = hs.up(tglwe, batch * k1 * n * 8, &dc)) != FHE_OK) return rc;
    if ((rc = hs.up(p, batch * n * 8, &dp)) != FHE_OK) return rc;
    if ((rc = hs.up(nullptr, batch * k1 * n * 8, &dout)) != FHE_OK) return rc;
    rc = fhe_tglwe_mul_tn_dev(n, k, dc, dp, dout, batch, hipStreamPerThread);
    if (rc != FHE_OK) return rc;
    return hs.down(out, dout, batch * k1 * n * 8);
}

extern "C" int fhe_tglev_mul(uint64_t n, unsigned k, unsigned l, const uint64_t *tglev, const uint64_t *v, uint64_t *out, size_t batch) {
    if (batch == 0) return FHE_OK;
    if (!tglev || !v || !out) return fhe_fail(FHE_E_NULL, "fhe_tglev_mul: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    HostStage hs;
    void *dg, *dv, *dout;
    const size_t k1 = (size_t)k + 1;
    if ((rc = hs.up(tglev, (size_t)l * k1 * n * 8, &dg)) != FHE_OK) return rc;
    if ((rc = hs.up(v, batch * l * n * 8, &dv)) != FHE_OK) return rc;
    if ((rc = hs.up(nullptr, batch * k1 * n * 8, &dout)) != FHE_OK) return rc;
    rc = fhe_tglev_mul_dev(n, k, l, dg, dv, dout, batch, hipStreamPerThread);
    if (rc != FHE_OK) return rc;
    return hs.down(out, dout, batch * k1 * n * 8);
}
