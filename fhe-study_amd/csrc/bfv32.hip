// bfv32.hip — BFV RLWE::tensor and relinearisation (bfv/src/lib.rs:71-77, 204-277) on TWO 27-bit primes with 32-bit
// arithmetic, for the small moduli the reference's i64 / f64 arithmetic confines BFV to.
//
// Why: at q = 65537, N = 8192 the integers of the tensor are below 2 n q^2 < 2^48 — the 61-bit machinery of zring.hip
// (22 VALU instructions per butterfly, two HBM passes per 2n-point transform) computes 48-bit numbers.  Modulo the two
// primes of digit32.hpp (p < 2^32/25, product 2^54.7) a butterfly is 6 instructions, a 2n = 16384-point transform of
// u32 words is 64 KiB and fits ONE workgroup's LDS: every transform is a single HBM pass, and the steps around the
// transforms ride in their loads and epilogues:
//   bfv32_forward_kernel          n 64-bit words, zero-padded to 2n -> the transforms modulo two or three primes, u32
//   bfv32_tensor_inverse_kernel   per output polynomial c0 / c1 / c2: pointwise tensor of the four transforms in the load
//                                 -> inverse transform, both primes -> CRT -> t/q scale, round, Z_q, X^n+1 fold
//   bfv32_relin_inverse_kernel    per output polynomial o0 / o1: product of the relinearisation key with the transform of
//                                 c2 in the load -> inverse, THREE primes -> Garner modulo 2^64 (all the reference's
//                                 `as i64` keeps) -> 1/p scale, round, fold, + c0 / c1
// The relinearisation's integers (c2 * rlk summed over n terms: 17 + 51 + 13 = 81 bits at q = 65537, p = q^2) need a third
// prime (product 2^82.05).  Splitting the key in 17-bit limbs on two primes instead — six inverse transforms per output
// polynomial and prime — was built first and measured: 2.4 ms per 2048 ciphertexts for this kernel alone.
// Arithmetic, rounding and wrap-around semantics are those of zring.hip's epilogues (same device functions); results are
// bit-exact with them and with the oracle.
#include "bfv32.hpp"
#include "ntt32_big.hpp"

#include <type_traits>

namespace fhe {

// x * y * 2^-32 mod p, in [0, 2p), for x * y < p * 2^32 (Montgomery; the factor is folded into the inverse's scaling)
__device__ __forceinline__ u32 mont32(u32 x, u32 y, u32 p, u32 pinv_neg) {
    const u64 t = (u64)x * y;
    const u32 m = (u32)t * pinv_neg;
    return (u32)((t + (u64)m * p) >> 32);
}
// the two residues (canonical) -> the integer in [0, pA * pB)
__device__ __forceinline__ u64 crt2(u32 rA, u32 rB, u32 pA, u32 pB, Tw32 crt) {
    const u32 rAb = csub_u32(rA, pB);                           // rA mod pB (pA - pB < pB)
    const u32 diff = csub_u32(rB - rAb + pB, pB);
    const u32 h = csub_u32(mul_shoup32(diff, crt, pB), pB);
    return (u64)rA + (u64)pA * h;
}

template <int LP>
__device__ __forceinline__ void stage_all(Tw32 *(&ltw)[3], unsigned char *base, const Tw32 *const (&tw)[3], u32 tid) {
    using C = Big32<LP>;
#pragma unroll
    for (int pr = 0; pr < 3; pr++) {
        ltw[pr] = reinterpret_cast<Tw32 *>(base + pr * C::TW_BYTES);
        stage_tw32<C::TH>(ltw[pr], tw[pr], C::LTW_N, tid);
    }
}

// ---- forward: row r of n words, zero-padded to 2n, modulo the first NPR primes ----------------------------------------
// WORD32: the source words are below 2^32 (ciphertext words modulo a q that small): reduced in one word
template <int LP, int NPR, bool WORD32>
__global__ __launch_bounds__((Big32<LP>::TH)) void bfv32_forward_kernel(Bfv32Args a) {
    using C = Big32<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *lds = reinterpret_cast<u32 *>(smem_raw);
    Tw32 *ltw[3];
    const u32 tf = threadIdx.x;
    stage_all<LP>(ltw, smem_raw + C::TILE_BYTES, a.t.tw_fwd, tf);
    const u64 row = blockIdx.x;
    const u32 n = C::M / 2;
    const u64 *__restrict__ src = a.src + row * n;
    // window [LP-4, LP): register k of logical thread t = point k * (M / 16) + t; k >= 8 is the zero padding
    u64 x[C::VT][8];
#pragma unroll
    for (int s = 0; s < C::VT; s++)
#pragma unroll
        for (int k = 0; k < 8; k++) x[s][k] = src[(u32)k * (C::VT * C::TH) + tf + s * C::TH];
    __syncthreads();                                            // the twiddle tiles
    // NPR == 1: prime blockIdx.y of this row (the relinearisation key: two rows, so three workgroups each instead of three
    // transforms in turn)
#pragma unroll
    for (int i = 0; i < NPR; i++) {
        const u32 pr = NPR == 1 ? blockIdx.y : (u32)i;
        const u32 p = a.t.p[pr], p2 = 2u * p;
        const Tw32 *lt = reinterpret_cast<const Tw32 *>(smem_raw + C::TILE_BYTES + pr * C::TW_BYTES);
        u32 v[C::VT][16];
#pragma unroll
        for (int s = 0; s < C::VT; s++)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                v[s][k] = WORD32 ? csub_u32(barrett2p_32((u32)x[s][k], p, a.t.bq[pr]), p) : reduce64_32(x[s][k], p, a.t.mu[pr]);
                v[s][k + 8] = v[s][k];                          // stage 0 against zeros: x + w * 0 and x - w * 0
            }
        fwd_big<LP, 1>(v, lds, lt, a.t.tw_fwd[pr], tf, p, p2, a.t.bq[pr]);
#pragma unroll
        for (int s = 0; s < C::VT; s++) {
            // stored order (internal to this file): quad j of logical thread t at j * (M / 4) + 4 t — a wave's 16-byte
            // accesses are contiguous (with the natural 16 t + 4 j every access would touch a quarter of each 64-byte line)
            u32 *__restrict__ dst = a.fw + (((u64)pr * a.rows + row) << LP) + (tf + s * C::TH) * 4u;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint4 o;
                o.x = barrett2p_32(v[s][4 * j], p, a.t.bq[pr]); o.y = barrett2p_32(v[s][4 * j + 1], p, a.t.bq[pr]);
                o.z = barrett2p_32(v[s][4 * j + 2], p, a.t.bq[pr]); o.w = barrett2p_32(v[s][4 * j + 3], p, a.t.bq[pr]);
                *reinterpret_cast<uint4 *>(dst + j * (C::M / 4)) = o;    // below 2p: a product of two such is below p * 2^32
            }
        }
    }
}

// the 16 transform values of a logical thread: src = row + 4 t, quads M / 4 apart (the forward kernel's stored order)
template <int LP>
__device__ __forceinline__ void load16(u32 (&v)[16], const u32 *__restrict__ src) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint4 x = *reinterpret_cast<const uint4 *>(src + j * (Big32<LP>::M / 4));
        v[4 * j] = x.x; v[4 * j + 1] = x.y; v[4 * j + 2] = x.z; v[4 * j + 3] = x.w;
    }
}

// ---- tensor: products in the load -> inverse (both primes) -> CRT -> scale, round, Z_q, fold ---------------------------
// workgroup = (ciphertext pair b, output polynomial which): c0 = a0 b0, c1 = a0 b1 + a1 b0, c2 = a1 b1 (lib.rs:71-77)
template <int LP>
__global__ __launch_bounds__((Big32<LP>::TH)) void bfv32_tensor_inverse_kernel(Bfv32Args a) {
    using C = Big32<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *lds = reinterpret_cast<u32 *>(smem_raw);
    Tw32 *ltw[3];
    const u32 tf = threadIdx.x;
    stage_all<LP>(ltw, smem_raw + C::TILE_BYTES, a.t.tw_inv, tf);
    // the three workgroups of a pair read the same four transforms: workgroup ids are dealt round-robin over the 8 XCDs,
    // so ids 24 g + 8 which + (b % 8) put them on ONE XCD (one L2), a few dispatches apart
    const u64 g8 = blockIdx.x / 24;
    const u32 r24 = blockIdx.x % 24;
    const u64 b = g8 * 8 + (r24 & 7u);
    const u32 which = r24 >> 3;
    if (b >= a.batch) return;                                   // padding of the last group (before any barrier)
    const u32 n = C::M / 2;
    __syncthreads();
    u32 resA[C::VT][16];
    u64 *__restrict__ po = a.out + ((u64)which * a.batch + b) * n + tf;
#pragma unroll
    for (int pr = 0; pr < 2; pr++) {
        const u32 p = a.t.p[pr], p2 = 2u * p, pn = a.pinv_neg[pr];
        const u32 *__restrict__ fw = a.fw + (((u64)pr * a.rows) << LP);      // rows: [a0 | a1 | b0 | b1] x batch
        u32 v[C::VT][16];
#pragma unroll
        for (int s = 0; s < C::VT; s++) {
            const u32 off = (tf + s * C::TH) * 4u;
            auto rowp = [&](u32 w) { return fw + (((u64)w * a.batch + b) << LP) + off; };
            u32 y[16];
            if (which == 1) {
                u32 t0[16];
                load16<LP>(v[s], rowp(0)); load16<LP>(y, rowp(3));
#pragma unroll
                for (int k = 0; k < 16; k++) t0[k] = mont32(v[s][k], y[k], p, pn);
                load16<LP>(v[s], rowp(1)); load16<LP>(y, rowp(2));
#pragma unroll
                for (int k = 0; k < 16; k++) v[s][k] = csub_u32(t0[k] + mont32(v[s][k], y[k], p, pn), p2);
            } else {
                load16<LP>(v[s], rowp(which == 0 ? 0 : 1)); load16<LP>(y, rowp(which == 0 ? 2 : 3));
#pragma unroll
                for (int k = 0; k < 16; k++) v[s][k] = mont32(v[s][k], y[k], p, pn);
            }
        }
#ifndef FHE_B32_ABLATE_INV      // timing-only builds (tools/abl_build.sh): the kernels without their transforms / their f64 epilogues
        inv_big<LP>(v, lds, ltw[pr], a.t.tw_inv[pr], tf, p, p2);
#endif
        const Tw32 ni = a.ninv_mont[pr];
        if (pr == 0) {
#pragma unroll
            for (int s = 0; s < C::VT; s++)
#pragma unroll
                for (int k = 0; k < 16; k++) resA[s][k] = csub_u32(mul_shoup32(v[s][k], ni, p), p);
        } else {
            // register k of logical thread t = point k * (M / 16) + t: k and k + 8 are coefficients j and j + n of the 2n-word
            // convolution — the pair the X^n+1 fold subtracts (ring_nq.rs:132-141); mul_div_round (ring_n.rs:130-138) +
            // Rq::from_vec_f64 (ring_nq.rs:160-163)
#pragma unroll
            for (int s = 0; s < C::VT; s++)
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const u32 rl = csub_u32(mul_shoup32(v[s][k], ni, p), p), rh = csub_u32(mul_shoup32(v[s][k + 8], ni, p), p);
                    const long long lo = (long long)crt2(resA[s][k], rl, a.t.p[0], a.t.p[1], a.t.crt);
                    const long long hi = (long long)crt2(resA[s][k + 8], rh, a.t.p[0], a.t.p[1], a.t.crt);
#ifndef FHE_B32_ABLATE_EPI
                    const u64 zl = zq_from_f64_mu(a.q, a.qmu, round((a.numf * (double)lo) / a.denf));
                    const u64 zh = zq_from_f64_mu(a.q, a.qmu, round((a.numf * (double)hi) / a.denf));    // slot 2n-1 of a (2n-1)-term convolution is 0
#else
                    const u64 zl = (u64)lo, zh = (u64)hi;
#endif
                    po[(u32)k * (C::VT * C::TH) + s * C::TH] = zl >= zh ? zl - zh : (a.q + zl) - zh;   // Zq::sub, zq.rs:259-276
                }
        }
    }
}

// ---- relinearisation: key-limb products in the load -> inverse (both primes) -> CRT -> limbs recombined mod 2^64
//      -> 1/p scale, round, Z_q, fold, + c0 / c1 (lib.rs:204-277) ---------------------------------------------------------
// workgroup = (ciphertext b, output polynomial o)
template <int LP>
__global__ __launch_bounds__((Big32<LP>::TH)) void bfv32_relin_inverse_kernel(Bfv32Args a) {
    using C = Big32<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *lds = reinterpret_cast<u32 *>(smem_raw);
    Tw32 *ltw[3];
    const u32 tf = threadIdx.x;
    stage_all<LP>(ltw, smem_raw + C::TILE_BYTES, a.t.tw_inv, tf);
    // both output polynomials of a ciphertext read the same transform of c2: ids 16 g + 8 o + (b % 8) share an XCD
    const u64 g8 = blockIdx.x / 16;
    const u32 r16 = blockIdx.x % 16;
    const u64 b = g8 * 8 + (r16 & 7u);
    const u32 o = r16 >> 3;
    if (b >= a.batch) return;
    const u32 n = C::M / 2;
    __syncthreads();
    // Garner's digits: x = v0 + pA v1 + pA pB v2 with v0 = rA, v1 = (rB - v0) pA^-1 mod pB,
    // v2 = ((rC - v0) pA^-1 - v1) pB^-1 mod pC; only x mod 2^64 is kept
    u32 v0[C::VT][16], v1[C::VT][16];
    u64 R[C::VT][16];
    auto prime = [&](auto prc) __attribute__((always_inline)) {     // a lambda per prime: the loop form is not unrolled by the compiler
        constexpr int pr = decltype(prc)::value;
        const u32 p = a.t.p[pr], p2 = 2u * p, pn = a.pinv_neg[pr];
        u32 v[C::VT][16];
#pragma unroll
        for (int s = 0; s < C::VT; s++) {
            const u32 off = (tf + s * C::TH) * 4u;
            u32 y[16];
            load16<LP>(v[s], a.x + (((u64)pr * a.batch + b) << LP) + off);
            load16<LP>(y, a.key + (((u64)pr * 2 + o) << LP) + off);
#pragma unroll
            for (int k = 0; k < 16; k++) v[s][k] = mont32(v[s][k], y[k], p, pn);
        }
#ifndef FHE_B32_ABLATE_INV      // timing-only builds (tools/abl_build.sh): the kernels without their transforms / their f64 epilogues
        inv_big<LP>(v, lds, ltw[pr], a.t.tw_inv[pr], tf, p, p2);
#endif
        const Tw32 ni = a.ninv_mont[pr];
#pragma unroll
        for (int s = 0; s < C::VT; s++)
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const u32 r = csub_u32(mul_shoup32(v[s][k], ni, p), p);
                if constexpr (pr == 0) {
                    v0[s][k] = r;
                } else if constexpr (pr == 1) {
                    const u32 d = csub_u32(r - csub_u32(v0[s][k], p) + p, p);              // pA - pB < pB: one subtraction reduces v0
                    v1[s][k] = csub_u32(mul_shoup32(d, a.t.crt, p), p);
                } else {
                    const u32 a0 = csub_u32(csub_u32(v0[s][k], p), p);                       // pA - pC < 2 pC
                    const u32 d = csub_u32(r - a0 + p, p);
                    const u32 e = csub_u32(mul_shoup32(d, a.t.crt_ac, p), p);
                    const u32 b1 = csub_u32(v1[s][k], p);                                    // pB - pC < pC
                    const u32 v2 = csub_u32(mul_shoup32(csub_u32(e - b1 + p, p), a.t.crt_bc, p), p);
                    R[s][k] = (u64)v0[s][k] + (u64)a.t.p[0] * v1[s][k] + a.t.P * v2;        // mod 2^64; P = pA pB < 2^55
                }
            }
    };
    prime(std::integral_constant<int, 0>{});
    prime(std::integral_constant<int, 1>{});
    prime(std::integral_constant<int, 2>{});
    const u64 off = ((u64)o * a.batch + b) * n + tf;
#pragma unroll
    for (int s = 0; s < C::VT; s++)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const long long lo = (long long)R[s][k], hi = (long long)R[s][k + 8];
#ifndef FHE_B32_ABLATE_EPI
            const u64 zl = zq_from_f64_mu(a.q, a.qmu, round((a.numf * (double)lo) / a.denf));
            const u64 zh = zq_from_f64_mu(a.q, a.qmu, round((a.numf * (double)hi) / a.denf));
#else
            const u64 zl = (u64)lo, zh = (u64)hi;
#endif
            u64 v = zl >= zh ? zl - zh : (a.q + zl) - zh;      // Zq::sub, zq.rs:259-276
            const u64 at = off + (u32)k * (C::VT * C::TH) + s * C::TH;
            v += a.addend[at];
            if (v >= a.q) v -= a.q;                            // Zq::add, zq.rs:219-231
            a.out[at] = v;
        }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
static unsigned bits_of64(uint64_t x) { unsigned b = 0; while (x) { b++; x >>= 1; } return b; }

bool bfv32_shape_supported(uint64_t q, uint64_t n, uint64_t pq) {
    if (n < 1024 || n > 8192 || (n & (n - 1)) || q < 2) return false;
    const unsigned ln = bits_of64(n - 1), bq = bits_of64(q - 1);
    if (2 * bq + ln + 1 > 54) return false;                   // c1 = a0 b1 + a1 b0 < 2 n q^2 must stay below pA pB = 2^54.7
    if (pq && (pq < q || bq + bits_of64(pq - 1) + ln > 82)) return false;      // c2 * rlk over n terms below pA pB pC = 2^82.05
    return true;
}

template <typename K>
static hipError_t launch_big(K kernel, const char *name, int lp, size_t lds, unsigned th, u64 grid, const Bfv32Args &a, hipStream_t st,
                             unsigned gridy = 1) {
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    if (hipError_t e = allow_big_lds((const void *)kernel, lds)) return e;
    KernelTimer kt(name, lp, st);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid, gridy), dim3(th), lds, st, a);
    return hipGetLastError();
}
#define FHE_BIG_SWITCH(KERNEL, NAME, GRID, GRIDY)                                                                                 \
    switch (a.log_n2) {                                                                                                      \
        case 11: return launch_big(KERNEL<11>, NAME, 11, Big32<11>::TILE_BYTES + 3 * Big32<11>::TW_BYTES, Big32<11>::TH, GRID, a, st, GRIDY); \
        case 12: return launch_big(KERNEL<12>, NAME, 12, Big32<12>::TILE_BYTES + 3 * Big32<12>::TW_BYTES, Big32<12>::TH, GRID, a, st, GRIDY); \
        case 13: return launch_big(KERNEL<13>, NAME, 13, Big32<13>::TILE_BYTES + 3 * Big32<13>::TW_BYTES, Big32<13>::TH, GRID, a, st, GRIDY); \
        case 14: return launch_big(KERNEL<14>, NAME, 14, Big32<14>::TILE_BYTES + 3 * Big32<14>::TW_BYTES, Big32<14>::TH, GRID, a, st, GRIDY); \
    }                                                                                                                        \
    return hipErrorNotSupported;

template <int LP> static constexpr auto bfv32_forward2 = bfv32_forward_kernel<LP, 2, true>;
template <int LP> static constexpr auto bfv32_forward3 = bfv32_forward_kernel<LP, 3, true>;
template <int LP> static constexpr auto bfv32_forward1w = bfv32_forward_kernel<LP, 1, false>;
hipError_t launch_bfv32_forward(const Bfv32Args &a, hipStream_t st) {
    if (a.primes == 2 && a.word32) { FHE_BIG_SWITCH(bfv32_forward2, "bfv32_forward", a.rows, 1) }
    if (a.primes == 3 && a.word32) { FHE_BIG_SWITCH(bfv32_forward3, "bfv32_forward3", a.rows, 1) }
    if (a.primes == 3) { FHE_BIG_SWITCH(bfv32_forward1w, "bfv32_forward_key", a.rows, 3) }      // grid (rows, primes)
    return hipErrorNotSupported;
}
hipError_t launch_bfv32_tensor_inverse(const Bfv32Args &a, hipStream_t st) { FHE_BIG_SWITCH(bfv32_tensor_inverse_kernel, "bfv32_tensor_inverse", 24 * ((a.batch + 7) / 8), 1) }
hipError_t launch_bfv32_relin_inverse(const Bfv32Args &a, hipStream_t st) { FHE_BIG_SWITCH(bfv32_relin_inverse_kernel, "bfv32_relin_inverse", 16 * ((a.batch + 7) / 8), 1) }
#undef FHE_BIG_SWITCH

// timing-only builds (tools/abl_build.sh) produce wrong words by design: fhe_ntt_version() says so (capi.hip)
bool bfv32_ablated() {
#if defined(FHE_B32_ABLATE_INV) || defined(FHE_B32_ABLATE_EPI)
    return true;
#else
    return false;
#endif
}

}  // namespace fhe
