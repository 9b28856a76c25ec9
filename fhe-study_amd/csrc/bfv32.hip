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
    const u32 diff = rB - rAb + pB;                              // in (0, 2 pB): the lazy product takes any word
    const u32 h = csub_u32(mul_shoup32(diff, crt, pB), pB);
    return (u64)rA + (u64)pA * h;
}

// ---- the 2n-point transform of a zero-padded row as TWO n-point blocks (round 3) ------------------------------------------
// Stage 0 of the 2n-point forward transform pairs x[j] with the padding — x + w 0 and x - w 0: both halves start as x —
// and stages 1 .. L-1 never mix the halves: the transform IS block 0 and block 1 (s0 = 1: ntt32_big.hpp) of the same n
// words.  Likewise the inverse: stages L-1 .. 1 act inside the halves and only the last one (roots_inv[1]) pairs point j of
// half 0 with point j of half 1 — coefficients j and j + n of the 2n-word product, exactly the pair the epilogues need
// together (scale, round, X^n+1 fold).  So a workgroup is n / 16 threads (512 at n = 8192) around a 34 KiB tile instead
// of 1024 around 68 KiB, and a CU holds two of them at 128 registers — three at 80, which the tensor's inverse kernel
// and (round 5) the two-prime forward kernel run at: one computes while the others wait at a barrier or
// for its loads.  Until round 2 one 1024-thread workgroup per CU left every load phase and barrier exposed (2048
// products: 2.75 ms).
// how the block kernels run their inverse rounds (shape experiments: -DFHE_B32_INV_T=.. / -DFHE_B32_INV_R=..):
// 0 = twiddles preloaded ahead of the exchanges (inv_big), 1 = read as they go (inv_big_direct), 2 = the same, one stage at a time
#ifndef FHE_B32_INV_T
#define FHE_B32_INV_T 2          // tensor
#endif
#ifndef FHE_B32_INV_R
#define FHE_B32_INV_R 1          // relinearisation (measured, 2048 pairs at n = 8192: 937 / 897 / 1019 us for 0 / 1 / 2; tensor 863 / 898 / 853)
#endif
#ifndef FHE_B32_FWD_PRELOAD
#define FHE_B32_FWD_PRELOAD 1    // forward blocks: twiddles of a round held across the exchange (1) or read as they go (0)
#endif
#ifndef FHE_B32_FWD_WAVES
#define FHE_B32_FWD_WAVES 4      // forward kernels: waves per SIMD the register allocation must allow
#endif
// Round 5: the TWO-prime forward kernel (the tensor's four source polynomials: the largest of the step's forward work) at SIX
// waves per SIMD — three workgroups of 512 threads per CU instead of two, as the tensor's inverse kernel already runs — with
// the twiddles read as they go (no registers held across the exchange: no spills at 80 registers): 502 -> 453 us per 2048
// pairs at N = 8192.  The three-prime kernel (c2) spills at six waves and is slower there (181 -> 196 / 210 us): it keeps
// FHE_B32_FWD_WAVES / FHE_B32_FWD_PRELOAD.  profiles/r05_bfv_occupancy_ab.txt.
#ifndef FHE_B32_RELIN_WAVES
#define FHE_B32_RELIN_WAVES 4    // relinearisation kernel: waves per SIMD its registers must allow (6 = three workgroups per CU: spills, measured slower)
#endif
#ifndef FHE_B32_FWD2_WAVES
#define FHE_B32_FWD2_WAVES 6
#endif
#ifndef FHE_B32_FWD2_PRELOAD
#define FHE_B32_FWD2_PRELOAD 0
#endif
// The THREE-prime forward kernel (c2, ahead of the relinearisation) at six waves: its 16 source words are then not kept in
// registers across the primes but read again before each (FHE_B32_FWD3_RELOAD; an L1 / L2 hit).
#ifndef FHE_B32_FWD3_WAVES
#define FHE_B32_FWD3_WAVES 4
#endif
#ifndef FHE_B32_FWD3_RELOAD
#define FHE_B32_FWD3_RELOAD 0
#endif
constexpr int fwd_waves(int npr) { return npr == 2 ? FHE_B32_FWD2_WAVES : npr == 3 ? FHE_B32_FWD3_WAVES : FHE_B32_FWD_WAVES; }
constexpr bool fwd_reload(int npr) { return npr == 3 && FHE_B32_FWD3_RELOAD != 0; }
constexpr bool fwd_preload(int npr) { return npr == 2 ? FHE_B32_FWD2_PRELOAD != 0 : FHE_B32_FWD_PRELOAD != 0; }
#ifndef FHE_B32_PARK_LDS
#define FHE_B32_PARK_LDS 1       // inverse kernels: as many of the parked words as fit the occupancy's LDS share stay in LDS (round 5)
#endif
constexpr int kRelinWps = 4;     // waves per SIMD of the relinearisation kernel (120 registers: -Rpass-analysis=kernel-resource-usage)
#ifndef FHE_B32_INV_LOOSE
#define FHE_B32_INV_LOOSE 1      // inverse rounds without a conditional subtraction per butterfly (ntt32_rounds.hpp: round_inv32_loose); modes 1 / 2 only
#endif
// what a block transform hands the last stage (units of p): the loose rounds stop at 8p (one stage from values below 4p, nothing
// reduced), the others below 2p
template <int MODE> constexpr int last_in_of() { return (FHE_B32_INV_LOOSE && MODE != 0) ? 8 : 2; }
template <int MODE, int LB>
__device__ __forceinline__ void inv_block(u32 (&v)[1][16], u32 *lds, const Tw32 *ltw, const Tw32 *gtw, u32 tf, u32 p, u32 p2, u32 bq, u32 blk) {
    if constexpr (FHE_B32_INV_LOOSE && MODE != 0) inv_big_loose<LB, MODE == 2, last_in_of<MODE>()>(v, lds, ltw, gtw, tf, p, bq, 1u, blk);
    else if constexpr (MODE == 0) inv_big<LB>(v, lds, ltw, gtw, tf, p, p2, 1u, blk);
    else if constexpr (MODE == 1) inv_big_direct<LB, false>(v, lds, ltw, gtw, tf, p, p2, 1u, blk);
    else inv_big_direct<LB, true>(v, lds, ltw, gtw, tf, p, p2, 1u, blk);
}
template <int LB>
struct Blk {
    using C = Big32<LB>;                                        // the n-point block: VT = 1, TH = n / 16 threads
    static constexpr u32 M = C::M, TH = C::TH;
    static constexpr size_t LDS(int tables) { return C::TILE_BYTES + (size_t)tables * C::TW_BYTES; }
    // Round 5: the words the inverse kernels park between primes (the tensor's first-prime residues, Garner's digits) go
    // to HBM and back — 0.4 / 0.8 GB per step of 2048 pairs, they do not stay in the L2 (profiles/
    // r05_config3_pmc_traffic.txt).  In the relinearisation kernel the first park_lds(...) words per thread of plane 0 (the
    // one read twice) live in the LDS the workgroup may use WITHOUT lowering the occupancy, behind the tile and the twiddle
    // tables; the rest keep their place in memory: 722 -> 677 us per 2048 pairs at N = 8192.
    // WPS = waves per SIMD the kernel's registers allow (relinearisation: 120 registers, 4)
    static constexpr int park_lds(int tables, int words, int wps) {
        const long wgs = (4L * wps * 64) / (long)TH;                         // workgroups per CU at that occupancy
        const long budget = 160L * 1024 / (wgs < 1 ? 1 : wgs) - 1024;        // LDS per workgroup, with a margin
        const long k = (budget - (long)LDS(tables)) / (8L * (long)TH);
        return k < 0 ? 0 : (k > words ? words : (int)k);
    }
    static constexpr size_t LDS_PARKED(int tables, int words, int wps) { return LDS(tables) + (size_t)park_lds(tables, words, wps) * 8 * TH; }
    static __device__ __forceinline__ Tw32 *table(unsigned char *smem, int i) { return reinterpret_cast<Tw32 *>(smem + C::TILE_BYTES + i * C::TW_BYTES); }
};

// ---- forward: row r of n words -> both blocks of its 2n-point transform, modulo the first NPR primes --------------------
// WORD32: the source words are below 2^32 (ciphertext words modulo a q that small): reduced in one word.
// Workgroup ids 16 g + 8 blk + (row % 8): the two blocks of a row read the same words through ONE XCD's L2.
// BELOWP: ... and q <= every prime: a canonical word (v < q, the library's input contract) is its own residue — no reduction
template <int LB, int NPR, bool WORD32, bool BELOWP = false>
__global__ __launch_bounds__((Big32<LB>::TH), fwd_waves(NPR)) void bfv32_forward_kernel(Bfv32Args a) {
    static_assert(!BELOWP || WORD32, "words below a 27-bit prime are below 2^32");
    using C = Big32<LB>;
    using K = Blk<LB>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *lds = reinterpret_cast<u32 *>(smem_raw);
    const u32 tf = threadIdx.x;
    const u64 g8 = blockIdx.x / 16;
    const u32 r16 = blockIdx.x % 16;
    const u64 row = g8 * 8 + (r16 & 7u);
    const u32 blk = r16 >> 3;
    if (row >= a.rows) return;                                  // padding of the last group (before any barrier)
    // NPR == 1: prime blockIdx.y of this row (the relinearisation key: two rows, so three workgroups per block instead of
    // three transforms in turn)
#pragma unroll
    for (int i = 0; i < NPR; i++) stage_tw32_block<C::TH>(K::table(smem_raw, i), a.t.tw_fwd[NPR == 1 ? blockIdx.y : i], C::LTW_N, tf, 1u, blk);
    const u64 *__restrict__ src = a.src + row * K::M;
    // window [LB-4, LB): register k = word k * TH + tf; WORD32: only the low words are kept across the primes
    typename std::conditional<WORD32, u32, u64>::type x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = (typename std::conditional<WORD32, u32, u64>::type)src[(u32)k * C::TH + tf];
    __syncthreads();                                            // the twiddle tiles
#pragma unroll
    for (int i = 0; i < NPR; i++) {
        const u32 pr = NPR == 1 ? blockIdx.y : (u32)i;
        const u32 p = a.t.p[pr], p2 = 2u * p;
        if constexpr (fwd_reload(NPR)) {
            if (i > 0) {                                        // the source words again, through a pointer the compiler cannot see through
                u64 sa = reinterpret_cast<u64>(src);
                asm volatile("" : "+s"(sa));
                const gu64 *ps = reinterpret_cast<const gu64 *>(sa);
#pragma unroll
                for (int k = 0; k < 16; k++) x[k] = (typename std::conditional<WORD32, u32, u64>::type)ps[(u32)k * C::TH + tf];
            }
        }
        u32 v[1][16];
#pragma unroll
        for (int k = 0; k < 16; k++)
            v[0][k] = BELOWP ? (u32)x[k] : WORD32 ? csub_u32(barrett2p_32((u32)x[k], p, a.t.bq[pr]), p) : reduce64_32(x[k], p, a.t.mu[pr]);
        fwd_big<LB, 0, 1, fwd_preload(NPR)>(v, lds, K::table(smem_raw, i), a.t.tw_fwd[pr], tf, p, p2, a.t.bq[pr], 1u, blk);
        // stored order (internal to this file): quad j of logical thread t at j * (M / 4) + 4 t of its block — a wave's
        // 16-byte accesses are contiguous (with the natural 16 t + 4 j every access would touch a quarter of each line)
        u32 *__restrict__ dst = a.fw + (((u64)pr * a.rows + row) << (LB + 1)) + blk * K::M + tf * 4u;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint4 o;
            o.x = barrett2p_32(v[0][4 * j], p, a.t.bq[pr]); o.y = barrett2p_32(v[0][4 * j + 1], p, a.t.bq[pr]);
            o.z = barrett2p_32(v[0][4 * j + 2], p, a.t.bq[pr]); o.w = barrett2p_32(v[0][4 * j + 3], p, a.t.bq[pr]);
            *reinterpret_cast<uint4 *>(dst + j * (K::M / 4)) = o;    // below 2p: a product of two such is below p * 2^32
        }
    }
}

// the 16 transform values of a logical thread: src = block + 4 t, quads M / 4 apart (the forward kernel's stored order)
template <int LB>
__device__ __forceinline__ void load16(u32 (&v)[16], const u32 *__restrict__ src) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint4 x = *reinterpret_cast<const uint4 *>(src + j * (Big32<LB>::M / 4));
        v[4 * j] = x.x; v[4 * j + 1] = x.y; v[4 * j + 2] = x.z; v[4 * j + 3] = x.w;
    }
}
// the inverse's LAST stage (roots_inv[1]) on the two blocks' results, with the scaling by (2n)^-1 (and the Montgomery
// factor of the products): register k of logical thread t -> coefficients j = k * TH + t (lo) and j + n (hi), canonical.
// lo' = (lo + hi) c, hi' = (lo - hi) w1 c with c the scaling: ONE product per output (round 4; w1 c is a table constant) —
// the sum and the difference (+ KIN p) of two values below KIN p stay in a word and the lazy product takes any word.
template <int KIN>
__device__ __forceinline__ void last_stage(u32 &lo, u32 &hi, Tw32 ni, Tw32 w1ni, u32 p) {
    static_assert(2 * KIN <= 16, "the sum of two values below KIN p stays in a word (p < 2^32 / 25)");
    const u32 s = lo + hi, d = lo - hi + (u32)KIN * p;
    lo = csub_u32(mul_shoup32(s, ni, p), p);
    hi = csub_u32(mul_shoup32(d, w1ni, p), p);
}

// Zq::from_f64(round(num * v / q)) — ring_n.rs:130-138 (mul_div_round), ring_nq.rs:160-163, zq.rs:32-39 — for an integer
// v >= 0 with N = num * v < 2^52, WITHOUT f64.  Why it is the same number: in f64 the product num * v is exact (< 2^53);
// the quotient fl(N / q) has absolute error below (N / q) 2^-53 < 1 / (2q); the exact quotient k + f / q (f = N mod q) is
// either ON a half-integer (2f = q: fl is exact there and `round` goes away from zero) or at least 1 / (2q) away from
// every half-integer, and fl cannot land on one it is not on (half an ulp below 2^52 is smaller than that distance).  So
// round(fl(N / q)) = a + [2f >= q] with a = N div q, which `as i64` keeps and Zq::from_f64 reduces modulo q.
// mu = floor((2^64 - 1) / q): the quotient estimates are short by at most two.  The f64 form costs ~150 instructions
// per coefficient (IEEE division), this one ~35.  (The relinearisation's R / p is NOT in this regime — |R| reaches 2^63,
// the f64 quotient's error reaches past the nearest half-integer — and keeps the f64 form.)
__device__ __forceinline__ u64 zq_scale_round_int(u64 N, u64 q, u64 mu) {
    u64 a = __umul64hi(N, mu);
    u64 f = N - a * q;
    if (f >= q) { f -= q; a += 1; }
    if (f >= q) { f -= q; a += 1; }
    a += (2 * f >= q);
    u64 r = a - __umul64hi(a, mu) * q;
    r = r >= q ? r - q : r;
    return r >= q ? r - q : r;
}

// Zq::from_f64 (zq.rs:32-39: round, `as i64`, ((e % q) + q) % q) for |ef| < 2^50 and q < 2^30, without the general form's
// saturating f64 -> i64 conversion and 64-bit remainder (zq_from_f64_mu: ~90 instructions of the ~150 an epilogue spent per
// coefficient).  e = round(ef) is an exact integer; k = floor(e * fl(1/q)) is within one of floor(e / q) (the product's error is
// below 2^-2); r = e - k q is exact (one fma: |k q| < 2^52) and lies in (-q, 2q).  Round 4: the two corrections are done on the
// 32-bit integer — r + (q if r < 0), then min(r, r - q) as unsigned words: five half-cost instructions instead of three
// compare-select pairs on doubles (FHE_B32_INT_TAIL=0: the f64 tail, around rint).  The same integer modulo q: bit-exact.
#ifndef FHE_B32_INT_TAIL
#define FHE_B32_INT_TAIL 1
#endif
__device__ __forceinline__ u64 zq_from_f64_small(double ef, double qf, double qinv, u32 q) {
    const double e = round(ef);
#if FHE_B32_INT_TAIL
    const double k = floor(e * qinv);
    int r = (int)fma(-k, qf, e);                 // in (-q, 2q): fits (q < 2^30)
    r += (r >> 31) & (int)q;                     // [0, 2q)
    return (u64)min((u32)r, (u32)r - q);         // [0, q)
#else
    const double k = rint(e * qinv);
    double r = fma(-k, qf, e);
    r = r < 0.0 ? r + qf : r;
    r = r < 0.0 ? r + qf : r;
    r = r >= qf ? r - qf : r;
    return (u64)(u32)r;
#endif
}
// fl(N / den) WITHOUT the division sequence (v_div_scale x 2, v_rcp, four fma, v_div_fmas, v_div_fixup): y = fl(1 / den) comes
// from the host, q0 = fl(N y), r = N - q0 den (one fma: exact up to a rounding of relative size 2^-53 of r), x = fl(q0 + r y).
// q0 + r y differs from N / den by |N / den - q0| 2^-53 <= 2^-104 |N / den|, so x IS fl(N / den) unless N / den lies within
// 2^-104 (relative) of the midpoint m / 2^j of two neighbouring doubles — impossible here: N is an integer (num, v integers;
// a rounded product of integers above 2^53 is still one), den an ODD integer below 2^48 (q or q^2 / ...: the host checks —
// zring.hip bfv32_rden), so N / den = m / 2^j would force den | N, an integer, not a midpoint, and otherwise
// |N 2^j - m den| >= 1 puts the quotient at least 1 / den > 2^-48 ulp-widths from the midpoint, against an error of the
// two-fma form of ~2^-51 ulp-widths: a margin of 2^3 (with den < 2^53 the same argument leaves none; no counter-example
// in 600 k adversarial near-midpoint samples there either, but the gate now matches the proof — ADVICE r04).
// Same double as the reference's `/`, ~8 instructions shorter per coefficient.
__device__ __forceinline__ double exact_quotient(double N, double den, double rden) {
    const double q0 = N * rden;
    const double r = fma(-q0, den, N);
    return fma(r, rden, q0);
}
// the epilogue's Zq::from_f64(round(num * v / den)): SMALL = the f64-only form above (decided on the host)
template <bool SMALL>
__device__ __forceinline__ u64 scale_round(const Bfv32Args &a, long long v) {
    const double N = a.numf * (double)v;
    const double x = a.rdenf != 0.0 ? exact_quotient(N, a.denf, a.rdenf) : N / a.denf;
    if constexpr (SMALL) return zq_from_f64_small(x, (double)a.q, a.qinvf, (u32)a.q);
    else return zq_from_f64_mu(a.q, a.qmu, round(x));
}

// ---- tensor: products in the load -> inverse (both primes) -> CRT -> scale, round, Z_q, fold ---------------------------
// INT: the integer form of the epilogue's scale-and-round (a.int_num * v < 2^52 for every coefficient: decided on the host)
// workgroup = (ciphertext pair b, output polynomial which): c0 = a0 b0, c1 = a0 b1 + a1 b0, c2 = a1 b1 (lib.rs:71-77)
template <int LB, bool INT, bool SMALL>
__global__ __launch_bounds__((Big32<LB>::TH), 4) void bfv32_tensor_inverse_kernel(Bfv32Args a) {
    using C = Big32<LB>;
    using K = Blk<LB>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *lds = reinterpret_cast<u32 *>(smem_raw);
    const u32 tf = threadIdx.x;
    // the three workgroups of a pair read the same four transforms: workgroup ids are dealt round-robin over the 8 XCDs,
    // so ids 24 g + 8 which + (b % 8) put them on ONE XCD (one L2), a few dispatches apart
    const u64 g8 = blockIdx.x / 24;
    const u32 r24 = blockIdx.x % 24;
    const u64 b = g8 * 8 + (r24 & 7u);
    const u32 which = r24 >> 3;
    if (b >= a.batch) return;                                   // padding of the last group (before any barrier)
#pragma unroll
    for (int i = 0; i < 4; i++) stage_tw32_block<C::TH>(K::table(smem_raw, i), a.t.tw_inv[i >> 1], C::LTW_N, tf, 1u, (u32)(i & 1));
    __syncthreads();
    // The first prime's 32 residues per thread are NOT kept in registers across the second prime's transforms (round 4:
    // the kernel spilled 74 dwords per lane at its 128 registers and waited on its own scratch traffic).  They are parked in
    // the words of the output polynomial this thread writes anyway — slot k TH + tf holds {coefficient j, coefficient j + n}
    // — and come back one pair ahead of the epilogue that consumes them (an L2 hit: written by this workgroup microseconds
    // before).  `ps` is the same pointer made opaque, so that the compiler neither forwards the stored values through
    // registers nor moves the reloads up.
    auto prime = [&](auto prc) __attribute__((always_inline)) {     // a lambda per prime: the loop form is not unrolled by the compiler
        constexpr int pr = decltype(prc)::value;
        const u32 p = a.t.p[pr], p2 = 2u * p, pn = a.pinv_neg[pr];
        const u32 *__restrict__ fw = a.fw + (((u64)pr * a.rows) << (LB + 1));      // rows: [a0 | a1 | b0 | b1] x batch
        auto block = [&](u32 blk, u32 (&v)[1][16]) __attribute__((always_inline)) {
            const u32 off = blk * K::M + tf * 4u;
            auto rowp = [&](u32 w) { return fw + (((u64)w * a.batch + b) << (LB + 1)) + off; };
            // a quad at a time (four 16-byte loads in flight per term), so that the operands of the products are never
            // all live together: the kernel keeps the first prime's 32 residues across this
            const u32 *__restrict__ ra = rowp(which == 2 ? 1 : 0), *__restrict__ rb = rowp(which == 0 ? 2 : 3);
            const u32 *__restrict__ rc = rowp(1), *__restrict__ rd = rowp(2);            // which == 1: + a1 b0
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint4 xa = *reinterpret_cast<const uint4 *>(ra + j * (K::M / 4)), xb = *reinterpret_cast<const uint4 *>(rb + j * (K::M / 4));
                u32 t[4] = {mont32(xa.x, xb.x, p, pn), mont32(xa.y, xb.y, p, pn), mont32(xa.z, xb.z, p, pn), mont32(xa.w, xb.w, p, pn)};
                if (which == 1) {
                    const uint4 xc = *reinterpret_cast<const uint4 *>(rc + j * (K::M / 4)), xd = *reinterpret_cast<const uint4 *>(rd + j * (K::M / 4));
                    t[0] = csub_u32(t[0] + mont32(xc.x, xd.x, p, pn), p2); t[1] = csub_u32(t[1] + mont32(xc.y, xd.y, p, pn), p2);
                    t[2] = csub_u32(t[2] + mont32(xc.z, xd.z, p, pn), p2); t[3] = csub_u32(t[3] + mont32(xc.w, xd.w, p, pn), p2);
                }
                v[0][4 * j] = t[0]; v[0][4 * j + 1] = t[1]; v[0][4 * j + 2] = t[2]; v[0][4 * j + 3] = t[3];
            }
#ifndef FHE_B32_ABLATE_INV      // timing-only builds (tools/abl_build.sh): the kernels without their transforms / their f64 epilogues
            inv_block<FHE_B32_INV_T, LB>(v, lds, K::table(smem_raw, 2 * pr + (int)blk), a.t.tw_inv[pr], tf, p, p2, a.t.bq[pr], blk);
#endif
        };
        u32 v0[1][16], v1[1][16];
        block(0u, v0);
        block(1u, v1);
        const Tw32 ni = a.ninv_mont[pr], w1ni = a.w1ninv_mont[pr];
        u64 parked = 0;
        u32 tfe = tf;
        asm volatile("" : "+v"(tfe));                           // (formed here, from an opaque lane id: no address registers held across the transforms)
        u64 *po = a.out + ((u64)which * a.batch + b) * K::M + tfe;
        // (an integer made opaque, then named in the GLOBAL address space: the laundered generic pointer of round 4 read
        // back with flat_load, which waits for vmcnt(0) AND lgkmcnt(0) — every reload also waited for the previous pair's store)
        u64 psa = reinterpret_cast<u64>(po);
        asm volatile("" : "+v"(psa));
        const gu64 *ps = reinterpret_cast<const gu64 *>(psa);
        // (Round 5 tried keeping some of these 16 words per thread in LDS, as the relinearisation kernel now does with its
        // first plane: 2 or 9 words, 50 or 78 KiB per workgroup — this kernel got 6 % SLOWER either way, 665 against 626 us;
        // profiles/r05_bfv_occupancy_ab.txt.  The words stay in memory.)
        if constexpr (pr == 1) parked = ps[0];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            u32 rl = v0[0][k], rh = v1[0][k];
            last_stage<last_in_of<FHE_B32_INV_T>()>(rl, rh, ni, w1ni, p);
            if constexpr (pr == 0) {
                po[(u32)k * C::TH] = ((u64)rh << 32) | rl;
            } else {
                const u32 loA = (u32)parked, hiA = (u32)(parked >> 32);
                if (k + 1 < 16) parked = ps[(u32)(k + 1) * C::TH];
                // coefficients j and j + n of the 2n-word convolution — the pair the X^n+1 fold subtracts (ring_nq.rs:132-141);
                // mul_div_round (ring_n.rs:130-138) + Rq::from_vec_f64 (ring_nq.rs:160-163)
                const long long lo = (long long)crt2(loA, rl, a.t.p[0], a.t.p[1], a.t.crt);
                const long long hi = (long long)crt2(hiA, rh, a.t.p[0], a.t.p[1], a.t.crt);
#ifndef FHE_B32_ABLATE_EPI
                u64 zl, zh;
                if constexpr (INT) {
                    zl = zq_scale_round_int(a.int_num * (u64)lo, a.q, a.qmu);
                    zh = zq_scale_round_int(a.int_num * (u64)hi, a.q, a.qmu);
                } else {
                    zl = scale_round<SMALL>(a, lo);
                    zh = scale_round<SMALL>(a, hi);    // slot 2n-1 of a (2n-1)-term convolution is 0
                }
#else
                const u64 zl = (u64)lo, zh = (u64)hi;
#endif
                po[(u32)k * C::TH] = zl >= zh ? zl - zh : (a.q + zl) - zh;   // Zq::sub, zq.rs:259-276
                __builtin_amdgcn_sched_barrier(0);              // one coefficient pair at a time: the f64 temporaries of 16 interleaved epilogues do not fit
            }
        }
    };
    prime(std::integral_constant<int, 0>{});
    prime(std::integral_constant<int, 1>{});
}

// ---- relinearisation: key products in the load -> inverse (three primes) -> Garner modulo 2^64 -> 1/p scale, round,
//      Z_q, fold, + c0 / c1 (lib.rs:204-277) --------------------------------------------------------------------------------
// workgroup = (ciphertext b, output polynomial o)
template <int LB, bool SMALL>
__global__ __launch_bounds__((Big32<LB>::TH), FHE_B32_RELIN_WAVES) void bfv32_relin_inverse_kernel(Bfv32Args a) {
    using C = Big32<LB>;
    using K = Blk<LB>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *lds = reinterpret_cast<u32 *>(smem_raw);
    const u32 tf = threadIdx.x;
    // both output polynomials of a ciphertext read the same transform of c2: ids 16 g + 8 o + (b % 8) share an XCD
    const u64 g8 = blockIdx.x / 16;
    const u32 r16 = blockIdx.x % 16;
    const u64 b = g8 * 8 + (r16 & 7u);
    const u32 o = r16 >> 3;
    if (b >= a.batch) return;
#pragma unroll
    for (int i = 0; i < 6; i++) stage_tw32_block<C::TH>(K::table(smem_raw, i), a.t.tw_inv[i >> 1], C::LTW_N, tf, 1u, (u32)(i & 1));
    __syncthreads();
    // Garner's digits: x = v0 + pA v1 + pA pB v2 with v0 = rA, v1 = (rB - v0) pA^-1 mod pB,
    // v2 = ((rC - v0) pA^-1 - v1) pB^-1 mod pC; only x mod 2^64 is kept.
    // Round 4: the digits are NOT kept in registers across the next prime's transforms (64 registers of a budget of 128:
    // the kernel spilled 69 dwords per lane and waited on its own scratch traffic).  v0 and v1 of a coefficient pair
    // {j, j + n} are parked as one 64-bit word each in two planes of the workspace (a.park), written where they are made
    // and read back one pair ahead of the arithmetic that consumes them (L2 hits).  ps0 / ps1: the same pointers made
    // opaque, so that the compiler neither forwards the stored values through registers nor moves the reloads up.
    const u64 off = ((u64)o * a.batch + b) * K::M + tf;
    u64 *park0 = a.park + off, *park1 = park0 + 2 * a.batch * K::M;
    auto prime = [&](auto prc) __attribute__((always_inline)) {
        constexpr int pr = decltype(prc)::value;
        const u32 p = a.t.p[pr], p2 = 2u * p, pn = a.pinv_neg[pr];
        auto block = [&](u32 blk, u32 (&v)[1][16]) __attribute__((always_inline)) {
            const u32 bo = blk * K::M + tf * 4u;
            const u32 *__restrict__ rx = a.x + (((u64)pr * a.batch + b) << (LB + 1)) + bo;
            const u32 *__restrict__ rk = a.key + (((u64)pr * 2 + o) << (LB + 1)) + bo;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint4 xa = *reinterpret_cast<const uint4 *>(rx + j * (K::M / 4)), xb = *reinterpret_cast<const uint4 *>(rk + j * (K::M / 4));
                v[0][4 * j] = mont32(xa.x, xb.x, p, pn); v[0][4 * j + 1] = mont32(xa.y, xb.y, p, pn);
                v[0][4 * j + 2] = mont32(xa.z, xb.z, p, pn); v[0][4 * j + 3] = mont32(xa.w, xb.w, p, pn);
            }
#ifndef FHE_B32_ABLATE_INV
            inv_block<FHE_B32_INV_R, LB>(v, lds, K::table(smem_raw, 2 * pr + (int)blk), a.t.tw_inv[pr], tf, p, p2, a.t.bq[pr], blk);
#endif
        };
        u32 v0[1][16], v1[1][16];
        block(0u, v0);
        block(1u, v1);
        const Tw32 ni = a.ninv_mont[pr], w1ni = a.w1ninv_mont[pr];
        u64 pk0 = 0, pk1 = 0;                                   // the parked digits of the pair at hand
        u64 psa0 = reinterpret_cast<u64>(park0), psa1 = reinterpret_cast<u64>(park1);
        asm volatile("" : "+v"(psa0), "+v"(psa1));              // (here, not at the top: no registers held across the transforms)
        const gu64 *ps0 = reinterpret_cast<const gu64 *>(psa0), *ps1 = reinterpret_cast<const gu64 *>(psa1);   // global, not flat: see the tensor kernel
        // (round 5) plane 0 is read twice (primes 1 and 2): the first KL of its 16 words per thread stay in LDS, behind the
        // tile and the six tables; plane 1 keeps its place in memory
        constexpr int KL = FHE_B32_PARK_LDS ? K::park_lds(6, 16, kRelinWps) : 0;
        u64 *lpark = reinterpret_cast<u64 *>(smem_raw + K::LDS(6)) + tf;
        auto park0_load = [&](int k) __attribute__((always_inline)) { return k < KL ? lpark[(u32)k * C::TH] : ps0[(u32)k * C::TH]; };
        if constexpr (pr >= 1) pk0 = park0_load(0);
        if constexpr (pr == 2) pk1 = ps1[0];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            u32 r2[2] = {v0[0][k], v1[0][k]};                    // [0] = coefficient j, [1] = coefficient j + n
            last_stage<last_in_of<FHE_B32_INV_R>()>(r2[0], r2[1], ni, w1ni, p);
            if constexpr (pr == 0) {
                if (k < KL) lpark[(u32)k * C::TH] = ((u64)r2[1] << 32) | r2[0];
                else park0[(u32)k * C::TH] = ((u64)r2[1] << 32) | r2[0];
            } else if constexpr (pr == 1) {
                const u32 g0[2] = {(u32)pk0, (u32)(pk0 >> 32)};
                if (k + 1 < 16) pk0 = park0_load(k + 1);
                u32 g1[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const u32 d = r2[h] - csub_u32(g0[h], p) + p;                            // pA - pB < pB: one subtraction reduces v0; d in (0, 2p)
                    g1[h] = csub_u32(mul_shoup32(d, a.t.crt, p), p);
                }
                park1[(u32)k * C::TH] = ((u64)g1[1] << 32) | g1[0];
            } else {
                const u32 g0[2] = {(u32)pk0, (u32)(pk0 >> 32)}, g1[2] = {(u32)pk1, (u32)(pk1 >> 32)};
                if (k + 1 < 16) { pk0 = park0_load(k + 1); pk1 = ps1[(u32)(k + 1) * C::TH]; }
                long long R2[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const u32 a0 = csub_u32(csub_u32(g0[h], p), p);                          // pA - pC < 2 pC
                    const u32 d = r2[h] - a0 + p;                                            // in (0, 2p): the lazy products take any word
                    const u32 e = csub_u32(mul_shoup32(d, a.t.crt_ac, p), p);
                    const u32 b1 = csub_u32(g1[h], p);                                       // pB - pC < pC
                    const u32 v2 = csub_u32(mul_shoup32(e - b1 + p, a.t.crt_bc, p), p);
                    R2[h] = (long long)((u64)g0[h] + (u64)a.t.p[0] * g1[h] + a.t.P * v2);    // mod 2^64; P = pA pB < 2^55
                }
                const long long lo = R2[0], hi = R2[1];
#ifndef FHE_B32_ABLATE_EPI
                const u64 zl = scale_round<SMALL>(a, lo);
                const u64 zh = scale_round<SMALL>(a, hi);
#else
                const u64 zl = (u64)lo, zh = (u64)hi;
#endif
                u64 v = zl >= zh ? zl - zh : (a.q + zl) - zh;      // Zq::sub, zq.rs:259-276
                const u64 at = off + (u32)k * C::TH;
                v += a.addend[at];
                if (v >= a.q) v -= a.q;                            // Zq::add, zq.rs:219-231
                a.out[at] = v;
                __builtin_amdgcn_sched_barrier(0);              // as in the tensor kernel
            }
        }
    };
    prime(std::integral_constant<int, 0>{});
    prime(std::integral_constant<int, 1>{});
    prime(std::integral_constant<int, 2>{});
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
// a.log_n2 = log2(2n); the kernels are instantiated on the block size n = 2^(log_n2 - 1); TABLES local twiddle tiles
// PARKW: words per thread the kernel would like to park in LDS (0: none) — Blk::LDS_PARKED
#define FHE_BIG_SWITCH(KERNEL, NAME, GRID, GRIDY, TABLES, PARKW, WPS)                                                               \
    switch (a.log_n2) {                                                                                                      \
        case 11: return launch_big(KERNEL<10>, NAME, 11, Blk<10>::LDS_PARKED(TABLES, PARKW, WPS), Big32<10>::TH, GRID, a, st, GRIDY); \
        case 12: return launch_big(KERNEL<11>, NAME, 12, Blk<11>::LDS_PARKED(TABLES, PARKW, WPS), Big32<11>::TH, GRID, a, st, GRIDY); \
        case 13: return launch_big(KERNEL<12>, NAME, 13, Blk<12>::LDS_PARKED(TABLES, PARKW, WPS), Big32<12>::TH, GRID, a, st, GRIDY); \
        case 14: return launch_big(KERNEL<13>, NAME, 14, Blk<13>::LDS_PARKED(TABLES, PARKW, WPS), Big32<13>::TH, GRID, a, st, GRIDY); \
    }                                                                                                                        \
    return hipErrorNotSupported;

template <int LB> static constexpr auto bfv32_forward2 = bfv32_forward_kernel<LB, 2, true>;
template <int LB> static constexpr auto bfv32_forward3 = bfv32_forward_kernel<LB, 3, true>;
template <int LB> static constexpr auto bfv32_forward2b = bfv32_forward_kernel<LB, 2, true, true>;
template <int LB> static constexpr auto bfv32_forward3b = bfv32_forward_kernel<LB, 3, true, true>;
template <int LB> static constexpr auto bfv32_forward1w = bfv32_forward_kernel<LB, 1, false>;
hipError_t launch_bfv32_forward(const Bfv32Args &a, hipStream_t st) {
    const u64 grid = 16 * ((a.rows + 7) / 8);                                                   // (row, block) pairs, 8 rows x 2 blocks per group
    if (a.primes == 2 && a.word32 && a.below_p) { FHE_BIG_SWITCH(bfv32_forward2b, "bfv32_forward", grid, 1, 2, 0, 4) }
    if (a.primes == 3 && a.word32 && a.below_p) { FHE_BIG_SWITCH(bfv32_forward3b, "bfv32_forward3", grid, 1, 3, 0, 4) }
    if (a.primes == 2 && a.word32) { FHE_BIG_SWITCH(bfv32_forward2, "bfv32_forward", grid, 1, 2, 0, 4) }
    if (a.primes == 3 && a.word32) { FHE_BIG_SWITCH(bfv32_forward3, "bfv32_forward3", grid, 1, 3, 0, 4) }
    if (a.primes == 3) { FHE_BIG_SWITCH(bfv32_forward1w, "bfv32_forward_key", grid, 3, 1, 0, 4) }      // grid (row-blocks, primes)
    return hipErrorNotSupported;
}
template <int LB> static constexpr auto bfv32_tensor_inverse_f64 = bfv32_tensor_inverse_kernel<LB, false, false>;
template <int LB> static constexpr auto bfv32_tensor_inverse_f64s = bfv32_tensor_inverse_kernel<LB, false, true>;
template <int LB> static constexpr auto bfv32_tensor_inverse_int = bfv32_tensor_inverse_kernel<LB, true, false>;
hipError_t launch_bfv32_tensor_inverse(const Bfv32Args &a, hipStream_t st) {
    if (a.int_num) { FHE_BIG_SWITCH(bfv32_tensor_inverse_int, "bfv32_tensor_inverse", 24 * ((a.batch + 7) / 8), 1, 4, 0, 4) }
    if (a.small_f64) { FHE_BIG_SWITCH(bfv32_tensor_inverse_f64s, "bfv32_tensor_inverse", 24 * ((a.batch + 7) / 8), 1, 4, 0, 4) }
    FHE_BIG_SWITCH(bfv32_tensor_inverse_f64, "bfv32_tensor_inverse", 24 * ((a.batch + 7) / 8), 1, 4, 0, 4)
}
template <int LB> static constexpr auto bfv32_relin_inverse_gen = bfv32_relin_inverse_kernel<LB, false>;
template <int LB> static constexpr auto bfv32_relin_inverse_small = bfv32_relin_inverse_kernel<LB, true>;
hipError_t launch_bfv32_relin_inverse(const Bfv32Args &a, hipStream_t st) {
    if (a.small_f64) { FHE_BIG_SWITCH(bfv32_relin_inverse_small, "bfv32_relin_inverse", 16 * ((a.batch + 7) / 8), 1, 6, (FHE_B32_PARK_LDS ? 16 : 0), kRelinWps) }
    FHE_BIG_SWITCH(bfv32_relin_inverse_gen, "bfv32_relin_inverse", 16 * ((a.batch + 7) / 8), 1, 6, (FHE_B32_PARK_LDS ? 16 : 0), kRelinWps)
}
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
