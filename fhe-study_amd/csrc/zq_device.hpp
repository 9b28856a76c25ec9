// zq_device.hpp — Z_q arithmetic for gfx950, 64-bit modulus, in registers.
//
// Replaces the reference's scalar ops on the hot path:
//   Zq::mul  arith/src/zq.rs:315-328  ((a as u128 * b as u128) % q)
//   Zq::add  arith/src/zq.rs:219-231
//   Zq::sub  arith/src/zq.rs:259-276
// The reference reduces every product with a 128-bit remainder (`__umodti3`).
// Here a butterfly multiplies by a KNOWN twiddle w, so the quotient is estimated
// with a precomputed companion w' = floor(w * 2^64 / q) (Shoup), and values stay
// in a redundant range between stages (Harvey); only the last stage of a
// transform canonicalises to [0,q), so outputs are bit-identical to the
// reference's canonical `Zq.v`.
//
// Instruction selection follows measurements on MI355X (tools/ubench_valu.hip,
// tools/ubench_bfly.hip, profiles/r01_ubench_*.txt): v_mad_u64_u32, v_mul_*_u32,
// v_lshl_add_u64 and every carry-producing add issue at ~4 cycles per wave64, a
// plain 32-bit VOP2 at ~2.  So the butterfly is written as v_mad_u64_u32 chains
// (multiply AND 64-bit accumulate in one issue slot), 64-bit adds are
// v_lshl_add_u64, subtraction of a value is "+ ~v + 1" with the +1 folded into a
// constant, and subtraction of a constant is the addition of its negation.
// Compiler-only code: 105 cycles/butterfly-wave (at the nominal 2.4 GHz; the chip holds
// ~1.8 GHz under this load); this form: 77 (q < 2^61) / ~86.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fhe {

typedef unsigned long long u64;
typedef unsigned int u32;

struct Tw {  // one twiddle: w and its Shoup companion floor(w*2^64/q)
    u64 w, wp;
};

// Modulus constants handed to kernels by value (live in SGPRs).
struct Tw32 { uint32_t w, wp; };   // the small-modulus kernels (digit32 / bfv32 / smallq): twiddle and floor(w * 2^32 / p)

struct Mod {
    u64 q;      // modulus, 3 <= q < 2^63 (2^62 and above: the strict arithmetic, AR = 3 — the fields below that hold 2q / 4q are then unused)
    u64 q2;     // 2q
    u64 nq;     // -q      mod 2^64
    u64 neg2q;  // -2q     mod 2^64
    u64 neg4q;  // -4q     mod 2^64
    u64 q2p1;   // 2q + 1
    u64 r64;    // 2^64 mod q            (for the variable x variable product)
    u64 r64p;   // floor(r64 * 2^64 / q)
    u64 onep;   // floor(2^64 / q)       (Shoup companion of w = 1)
    // pseudo-Mersenne moduli q = 2^k - delta (pm_k != 0; see "pseudo-Mersenne" below)
    u64 q3p1;       // 3q + 1
    u32 pm_c2;      // 2 delta = 2^(k+1) mod q
    u32 pm_delta;   // delta   = 2^k mod q
    u32 pm_sh;      // k - 31:  T >> (k+1) = (T >> 32) >> (k - 31)
    u32 pm_mask;    // 2^(k-31) - 1
    u32 pm_rsh;     // k - 32:  x >> k = (x >> 32) >> (k - 32)
    u32 pm_rmask;   // 2^(k-32) - 1
    u32 pm_k;       // k, or 0 when q is not of this form
    // q = qh 2^32 + 1 below 2^61 (see "word Montgomery" below): 2^32 - qh, or 0 when q is not of this form
    u32 mg_nqh;
    u64 mg_r96, mg_r128;   // 2^96 mod q, 2^128 mod q: the table words of the constant 2^64 (mul_var_mg)
};

// ---- single-instruction wrappers (register allocation stays with the compiler) ----
__device__ __forceinline__ u64 mad64(u32 a, u32 b, u64 c) {  // a*b + c  (mod 2^64)
    u64 d;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
    return d;
}
__device__ __forceinline__ u64 mad64z(u32 a, u32 b) {  // a*b (inline-constant addend: no zero pair)
    u64 d;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b) : "vcc");
    return d;
}
// opaque 32-bit add: written in C the compiler widens `hi += x` into a 64-bit add of {0,x},
// i.e. two v_mov and a v_lshl_add_u64 instead of one v_add_u32 (tools/ubench_bfly.hip v6: -4 %)
__device__ __forceinline__ u32 add32(u32 a, u32 b) {
    u32 d;
    asm("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ u64 add64(u64 a, u64 b) {  // a + b in one issue slot
    u64 d;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ u64 dbl_add64(u64 a, u64 b) {  // 2a + b
    u64 d;
    asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// x - m if x >= m else x, for x < 2^63 + m, given negm = -m (mod 2^64), m < 2^63:
// one 64-bit add, then the sign of the difference selects.
__device__ __forceinline__ u64 csub_neg(u64 x, u64 negm) {
    const u64 d = add64(x, negm);
    return ((int)(u32)(d >> 32) < 0) ? x : d;   // sign of the high word: a 32-bit compare
}
__device__ __forceinline__ u64 csub(u64 x, u64 m) { return x >= m ? x - m : x; }

// add + y*w - floor(y*w'/2^64)*q  (mod 2^64)  =  add + (y*w mod q, lazily in [0,2q)),
// for ANY 64-bit y and 0 <= w < q.  Low halves go through one mad chain that also
// absorbs `add`; the four cross terms (only their low 32 bits matter) through another.
__device__ __forceinline__ u64 mul_shoup_acc(u64 add, u64 y, u64 w, u64 wp, u64 nq) {
    const u64 qh = __umul64hi(y, wp);
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
    const u32 h0 = (u32)qh, h1 = (u32)(qh >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
    // One statement, the two chains interleaved by hand: between two dependent asm STATEMENTS the
    // compiler inserts an s_nop (it assumes a dst-forwarding hazard for any inline asm), ~4 per
    // butterfly; v_mad_u64_u32 has no such hazard.  Measured -0.5 % on the 2^16 transform.
    u64 acc, H;
    asm("v_mad_u64_u32 %0, vcc, %3, %5, %2\n\t"
        "v_mad_u64_u32 %1, vcc, %3, %6, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %9, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %7, %10, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %9, %1"
        : "=&v"(acc), "=&v"(H)
        : "v"(add), "v"(y0), "v"(y1), "v"(w0), "v"(w1), "v"(h0), "v"(h1), "v"(n0), "v"(n1)
        : "vcc");
    const u32 hi = add32((u32)(acc >> 32), (u32)H);
    return ((u64)hi << 32) | (u32)acc;
}
// y * w mod q  in [0, 2q)
__device__ __forceinline__ u64 mul_shoup_lazy(u64 y, u64 w, u64 wp, const Mod &m) {
    return mul_shoup_acc(0, y, w, wp, m.nq);
}

// Forward (Cooley-Tukey) butterfly, arith/src/ntt.rs:57-62:
//   U = r[j]; V = r[j+t]*S; r[j] = U+V; r[j+t] = U-V
// CSUB = 0: none (caller guarantees headroom), 2: x -= 2q if x >= 2q, 4: x -= 4q if x >= 4q,
// 6: both (x < 8q -> x < 2q).
//   x' = u + t,  y' = u - t + 2q = (2u + 2q + 1) + ~x'
template <int CSUB>
__device__ __forceinline__ void ct_bfly(u64 &x, u64 &y, u64 w, u64 wp, const Mod &m) {
    u64 u = x;
    if (CSUB == 2) u = csub_neg(x, m.neg2q);
    if (CSUB == 4) u = csub_neg(x, m.neg4q);
    if (CSUB == 6) u = csub_neg(csub_neg(x, m.neg4q), m.neg2q);
    const u64 s = mul_shoup_acc(u, y, w, wp, m.nq);
    y = add64(dbl_add64(u, m.q2p1), ~s);
    x = s;
}

// Inverse (Gentleman-Sande) butterfly, arith/src/ntt.rs:91-96:
//   U = r[j]; V = r[j+t]; r[j] = U+V; r[j+t] = (U-V)*S
// Lazy form: x,y in [0,2q) -> x,y in [0,2q).
__device__ __forceinline__ void gs_bfly(u64 &x, u64 &y, u64 w, u64 wp, const Mod &m) {
    const u64 d = add64(add64(x, m.q2p1), ~y);  // x - y + 2q  in (0,4q)
    x = csub_neg(add64(x, y), m.neg2q);
    y = mul_shoup_acc(0, d, w, wp, m.nq);
}

// [0,2q) -> [0,q)
__device__ __forceinline__ u64 canon2(u64 x, const Mod &m) { return csub_neg(x, m.nq); }
// [0,4q) -> [0,q)
__device__ __forceinline__ u64 canon4(u64 x, const Mod &m) { return canon2(csub_neg(x, m.neg2q), m); }
// [0,8q) -> [0,q)   (q < 2^61)
__device__ __forceinline__ u64 canon8(u64 x, const Mod &m) { return canon4(csub_neg(x, m.neg4q), m); }


// ---------------------------------------------------------------------------------------------------------------------
// Pseudo-Mersenne moduli: q = 2^k - delta with 56 <= k <= 61 and delta <= 2^(k-39) — the headline modulus
// 2^61 - 2^21 + 1 is one.  Zq::mul (arith/src/zq.rs:315-328) by a KNOWN w then needs FIVE 32x32 multiplies instead of
// the ten of the Shoup form above, and no quotient at all:
//   y = y1 2^32 + y0,   y w  =  y0 w + y1 W2  (mod q),   W2 = w 2^32 mod q   (the table holds {w, W2} instead of {w, w'})
//   T = y0 w + y1 W2 < 2^33 q              four multiplies: two v_mad_u64_u32 chains, one carry between them
//   2^(k+1) = 2 delta (mod q):  r = (T mod 2^(k+1)) + (T >> (k+1)) * 2 delta       one more; T >> (k+1) < 2^32
// r = y w (mod q) and r < 2^(k+1) + 2^33 delta <= 2q + q/16 for ANY 64-bit y.  Values are brought down, when their
// bound asks for it, by x -> (x mod 2^k) + (x >> k) delta < q + q/16 (three instructions, against four for a
// conditional subtraction that only halves).  Measured in registers (tools/ubench_bfly.hip v17 against v8,
// profiles/r03_ubench_bfly.txt): 59.9 against 87.6 cycles per butterfly-wave.
//
// The 13 instructions of a butterfly (14 until round 4: see ct_bfly_pm) are ONE asm statement: between two dependent asm statements the compiler inserts an
// s_nop (it must assume a dst-forwarding hazard inside any inline asm), five per butterfly when each instruction is its
// own statement.  Sub-registers of a 64-bit asm operand cannot be named, so the temporaries are the physical registers
// v[2:7], listed as clobbers (the compiler keeps nothing there across a butterfly; inputs and outputs are its own).
// SGPR_TW: the twiddle words are wave-uniform (scalar loads): "s" operands, no copies into VGPRs.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kPmMul = 33;    // bounds in SIXTEENTHS of q: a product r < 2q + q/16
constexpr int kPmRed = 17;    // pm_reduce(x) < q + q/16
constexpr int kPmOne = 16;    // canonical
constexpr int kPmCap = 128;   // 8q <= 2^64: every value stays below

// x -> (x mod 2^k) + (x >> k) delta  =  x (mod q),  < 2^k + 2^(64-k) delta.  Plain C, the SHIFT WRITTEN FIRST (round 5):
// the compiler then masks the high word of x's own register pair and accumulates into it — v_lshrrev / v_and /
// v_mad_u64_u32, three instructions.  With the mask written first (rounds 3-4) it built the masked value in a fresh pair:
// a v_mov_b32 per reduction, 48 per thread of the contiguous pass, 231 of the fused product's 4900 instructions at N = 4096.
// An inline-asm form of the same three instructions (`v_mad_u64_u32 %0, vcc, k, delta, %0` tied with "+v" to the masked
// pair) was tried first and is NOT used: in rq_mul_mid_kernel (two operands live, 109 registers) hipcc 7.2 then handed
// stage 3 of a round a stale register pair for v[0] — words 0..31 of every 256-block wrong, bisected to that statement
// (gpurun_out/r5b).  This form leaves register allocation to the compiler and passes the whole GPU suite (259 tests:
// every kernel that reduces, 2^4 <= n <= 2^20, gpurun_out/r5j); FHE_PM_REDUCE_SHIFT_FIRST=0 restores the old order.
#ifndef FHE_PM_REDUCE_SHIFT_FIRST
#define FHE_PM_REDUCE_SHIFT_FIRST 1
#endif
__device__ __forceinline__ u64 pm_reduce(u64 x, const Mod &m) {
#if FHE_PM_REDUCE_SHIFT_FIRST
    const u32 k = (u32)(x >> 32) >> m.pm_rsh;
    const u64 lo = x & (((u64)m.pm_rmask << 32) | 0xffffffffull);
    return (u64)k * (u64)m.pm_delta + lo;
#else
    const u32 x1 = (u32)(x >> 32);
    const u64 lo = ((u64)(x1 & m.pm_rmask) << 32) | (u32)x;
    return (u64)(x1 >> m.pm_rsh) * (u64)m.pm_delta + lo;
#endif
}
// any value below 2^64 -> canonical
__device__ __forceinline__ u64 pm_canon(u64 x, const Mod &m) { return csub_neg(pm_reduce(x, m), m.nq); }

#define FHE_PM_PRODUCT(Y0, Y1, OUT)                                                                    \
    "v_mad_u64_u32 v[2:3], vcc, " Y0 ", %[a0], 0\n\t"                                                  \
    "v_mad_u64_u32 v[2:3], vcc, " Y1 ", %[b0], v[2:3]\n\t"      /* carry -> vcc */                    \
    "v_mov_b32 v6, v3\n\t"                                                                             \
    "v_addc_co_u32 v7, vcc, 0, 0, vcc\n\t"                       /* {T >> 32 so far} = {n1, carry} */  \
    "v_mad_u64_u32 v[6:7], vcc, " Y0 ", %[a1], v[6:7]\n\t"                                             \
    "v_mad_u64_u32 v[6:7], vcc, " Y1 ", %[b1], v[6:7]\n\t"      /* T >> 32 */                          \
    "v_and_b32 v3, %[mask], v6\n\t"                              /* T mod 2^(k+1) = {n0, b0 & mask} */ \
    "v_alignbit_b32 v6, v7, v6, %[sh]\n\t"                       /* T >> (k+1) */                      \
    "v_mad_u64_u32 " OUT ", vcc, v6, %[c2], v[2:3]\n\t"

// y * w mod q, lazily: < 2q + q/16, for any 64-bit y
template <bool SGPR_TW>
__device__ __forceinline__ u64 mul_pm(u64 y, u64 w, u64 w2, const Mod &m) {
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32);
    const u32 a0 = (u32)w, a1 = (u32)(w >> 32), b0 = (u32)w2, b1 = (u32)(w2 >> 32);
    u64 r;
    if constexpr (SGPR_TW)
        asm(FHE_PM_PRODUCT("%[y0]", "%[y1]", "%[r]")
            : [r] "=&v"(r)
            : [y0] "v"(y0), [y1] "v"(y1), [a0] "s"(a0), [a1] "s"(a1), [b0] "s"(b0), [b1] "s"(b1),
              [c2] "s"(m.pm_c2), [mask] "s"(m.pm_mask), [sh] "s"(m.pm_sh)
            : "vcc", "v2", "v3", "v6", "v7");
    else
        asm(FHE_PM_PRODUCT("%[y0]", "%[y1]", "%[r]")
            : [r] "=&v"(r)
            : [y0] "v"(y0), [y1] "v"(y1), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1),
              [c2] "s"(m.pm_c2), [mask] "s"(m.pm_mask), [sh] "s"(m.pm_sh)
            : "vcc", "v2", "v3", "v6", "v7");
    return r;
}


// VARIABLE x VARIABLE (zip_eq(l,r).map(l*r), arith/src/ring_nq.rs:601-604) on the same five multiplies: the second
// table word of the multiplier b is formed on the fly, b 2^32 = (b >> (k-32)) 2^k + (b mod 2^(k-32)) 2^32
// = (b >> (k-32)) delta + ((b mod 2^(k-32)) << 32)  (mod q), three instructions and a multiply for b < 2^k.
// It is below 2^k (1 + 2^-7), so the product's T stays below 2^(k+33) for every multiplicand a < 7.9 q — anything the
// rounds hand over.  13 instructions against ~50 for the Shoup form (mul_mod_var: two 64-bit quotients).
__device__ __forceinline__ u64 pm_shift32(u64 w, const Mod &m) {
    const u32 w0 = (u32)w, w1 = (u32)(w >> 32);
    const u32 wh = __builtin_amdgcn_alignbit(w1, w0, m.pm_rsh);          // w >> (k-32) < 2^32 for w < 2^k
    return (u64)wh * (u64)m.pm_delta + ((u64)(w0 & m.pm_rmask) << 32);
}
// x -> x (mod q), strictly below 2^k (not canonical): the form a variable multiplier needs
__device__ __forceinline__ u64 pm_below_2k(u64 x, const Mod &m) { return pm_reduce(pm_reduce(x, m), m); }
// a * b mod q, lazily (< 2q + q/16), for a < 7.9 q and b < 2^k
__device__ __forceinline__ u64 mul_var_pm(u64 a, u64 b, const Mod &m) { return mul_pm<false>(a, b, pm_shift32(b, m), m); }

// Forward butterfly (ntt.rs:57-62): x' = u + r, y' = u - r + 3q  (r = y w < 2q + q/16), u = x < B q  ->  both below
// (B + 3) q; the caller reduces x first when B + 3 would pass 8.  13 instructions: the subtraction is a borrow chain
// (v_sub_co_u32 / v_subb_co_u32 on the halves: one instruction fewer than "+ ~r + 1", tools/ubench_bfly.hip v18: 55.4
// against 59.8 cycles per butterfly-wave); the halves of a 64-bit asm operand cannot be named, so u + 3q is formed in the
// physical pair v[4:5] and y' leaves as two 32-bit outputs, which the compiler joins into a register pair without moves.
// Outputs written by the LAST instructions only (y', the inverse butterfly's product) are not early-clobber: every compiler-
// allocated input has been read by then, so they may take over y's registers.
template <bool SGPR_TW>
__device__ __forceinline__ void ct_bfly_pm(u64 &x, u64 &y, u64 w, u64 w2, const Mod &m) {
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32);
    const u32 a0 = (u32)w, a1 = (u32)(w >> 32), b0 = (u32)w2, b1 = (u32)(w2 >> 32);
    const u64 k3 = m.q3p1 - 1ull;                                 // 3q
    u32 yl, yh;
#define FHE_PM_CT_BODY                                                                                 \
    "v_lshl_add_u64 v[4:5], %[x], 0, %[k3]\n\t"                   /* u + 3q */                          \
    FHE_PM_PRODUCT("%[y0]", "%[y1]", "v[2:3]")                                                         \
    "v_lshl_add_u64 %[x], %[x], 0, v[2:3]\n\t"                   /* x' = u + r */                      \
    "v_sub_co_u32 %[yl], vcc, v4, v2\n\t"                                                              \
    "v_subb_co_u32 %[yh], vcc, v5, v3, vcc"                      /* y' = u + 3q - r */
    if constexpr (SGPR_TW)
        asm(FHE_PM_CT_BODY
            : [x] "+&v"(x), [yl] "=v"(yl), [yh] "=v"(yh)
            : [y0] "v"(y0), [y1] "v"(y1), [a0] "s"(a0), [a1] "s"(a1), [b0] "s"(b0), [b1] "s"(b1),
              [c2] "s"(m.pm_c2), [mask] "s"(m.pm_mask), [sh] "s"(m.pm_sh), [k3] "s"(k3)
            : "vcc", "v2", "v3", "v4", "v5", "v6", "v7");
    else
        asm(FHE_PM_CT_BODY
            : [x] "+&v"(x), [yl] "=v"(yl), [yh] "=v"(yh)
            : [y0] "v"(y0), [y1] "v"(y1), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1),
              [c2] "s"(m.pm_c2), [mask] "s"(m.pm_mask), [sh] "s"(m.pm_sh), [k3] "s"(k3)
            : "vcc", "v2", "v3", "v4", "v5", "v6", "v7");
#undef FHE_PM_CT_BODY
    y = ((u64)yh << 32) | yl;
}

// Inverse butterfly (ntt.rs:91-96): x' = x + y, y' = (x - y + K q) w.  kq1 = K q + 1 with K q >= the bound of y.
// x - y + K q is the same borrow chain (12 instructions in all); x is written before the product reads its operands,
// hence early-clobber.
template <bool SGPR_TW>
__device__ __forceinline__ void gs_bfly_pm(u64 &x, u64 &y, u64 w, u64 w2, u64 kq1, const Mod &m) {
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32);
    const u32 a0 = (u32)w, a1 = (u32)(w >> 32), b0 = (u32)w2, b1 = (u32)(w2 >> 32);
    const u64 kq = kq1 - 1ull;
    u64 yo;
#define FHE_PM_GS_BODY                                                                                 \
    "v_lshl_add_u64 v[4:5], %[x], 0, %[kq]\n\t"                                                        \
    "v_sub_co_u32 v4, vcc, v4, %[y0]\n\t"                                                              \
    "v_subb_co_u32 v5, vcc, v5, %[y1], vcc\n\t"                  /* d = x + K q - y */                 \
    "v_lshl_add_u64 %[x], %[x], 0, %[y]\n\t"                     /* x' = x + y */                      \
    FHE_PM_PRODUCT("v4", "v5", "%[yo]")
    if constexpr (SGPR_TW)
        asm(FHE_PM_GS_BODY
            : [x] "+&v"(x), [yo] "=v"(yo)
            : [y] "v"(y), [y0] "v"(y0), [y1] "v"(y1), [a0] "s"(a0), [a1] "s"(a1), [b0] "s"(b0), [b1] "s"(b1),
              [c2] "s"(m.pm_c2), [mask] "s"(m.pm_mask), [sh] "s"(m.pm_sh), [kq] "s"(kq)
            : "vcc", "v2", "v3", "v4", "v5", "v6", "v7");
    else
        asm(FHE_PM_GS_BODY
            : [x] "+&v"(x), [yo] "=v"(yo)
            : [y] "v"(y), [y0] "v"(y0), [y1] "v"(y1), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1),
              [c2] "s"(m.pm_c2), [mask] "s"(m.pm_mask), [sh] "s"(m.pm_sh), [kq] "s"(kq)
            : "vcc", "v2", "v3", "v4", "v5", "v6", "v7");
#undef FHE_PM_GS_BODY
    y = yo;
}
#undef FHE_PM_PRODUCT

// ---------------------------------------------------------------------------------------------------------------------
// Word Montgomery for q = qh 2^32 + 1 < 2^61 (round 4; tools/ubench_bfly.hip v14): the same split multiplicand on a table
// {A = w 2^32 mod q, B = w 2^64 mod q}:  T = y0 A + y1 B = y w 2^32 (mod q), T < 2^33 q  — the same four multiplies — and
// ONE Montgomery word step brings back the factor: q^-1 = 1 (mod 2^32), so the step's multiplier is -T0 itself and
//   r = (T >> 32) - T0 qh + q   =  T 2^-32  =  y w (mod q),   0 < r < 3q  (T >> 32 < 2q, T0 qh < q)
// is one more multiply (by 2^32 - qh, then T0 taken off the high word) and an addition: nine instructions, as the
// pseudo-Mersenne product.  What it lacks is that form's cheap reduction: values come down by a conditional subtraction
// of 4q (four instructions, to below 4q) before every stage from the third on, so a forward butterfly averages ~16.5
// instructions against ~14.5 — and against the ~21 of the Shoup form these moduli ran on.  An inverse stage needs 0.8 such
// subtractions per butterfly (x + y of two values below 4q is at the cap: ntt_rounds.hpp follows the bounds register by
// register): ~15.3 instructions against ~20.  A variable x variable product has no table: mul_var_mg below goes through the
// full 128-bit product instead (28 instructions against ~50).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kMgRed = 64;    // sixteenths of q: after the conditional subtraction of 4q (for values below 8q)
constexpr int kMgMul = 48;    // a product: below 3q
#define FHE_MG_PRODUCT(Y0, Y1, OUT)                                                                    \
    "v_mad_u64_u32 v[2:3], vcc, " Y0 ", %[a0], 0\n\t"                                                  \
    "v_mad_u64_u32 v[2:3], vcc, " Y1 ", %[b0], v[2:3]\n\t"      /* carry -> vcc */                    \
    "v_mov_b32 v6, v3\n\t"                                                                             \
    "v_addc_co_u32 v7, vcc, 0, 0, vcc\n\t"                                                             \
    "v_mad_u64_u32 v[6:7], vcc, " Y0 ", %[a1], v[6:7]\n\t"                                             \
    "v_mad_u64_u32 v[6:7], vcc, " Y1 ", %[b1], v[6:7]\n\t"      /* T >> 32;  T0 = v2 */               \
    "v_mad_u64_u32 v[6:7], vcc, v2, %[nqh], v[6:7]\n\t"         /* + T0 (2^32 - qh) */                \
    "v_sub_u32 v7, v7, v2\n\t"                                  /* - T0 2^32 */                       \
    "v_lshl_add_u64 " OUT ", v[6:7], 0, %[qq]\n\t"              /* + q: in (0, 3q) */

// Forward butterfly on the Montgomery table: x' = u + r, y' = u - r + 3q as ct_bfly_pm (r < 3q)
template <bool SGPR_TW>
__device__ __forceinline__ void ct_bfly_mg(u64 &x, u64 &y, u64 wa, u64 wb, const Mod &m) {
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32);
    const u32 a0 = (u32)wa, a1 = (u32)(wa >> 32), b0 = (u32)wb, b1 = (u32)(wb >> 32);
    const u64 k3 = m.q3p1 - 1ull;                                 // 3q
    u32 yl, yh;
#define FHE_MG_CT_BODY                                                                                 \
    "v_lshl_add_u64 v[4:5], %[x], 0, %[k3]\n\t"                   /* u + 3q */                          \
    FHE_MG_PRODUCT("%[y0]", "%[y1]", "v[2:3]")                                                         \
    "v_lshl_add_u64 %[x], %[x], 0, v[2:3]\n\t"                   /* x' = u + r */                      \
    "v_sub_co_u32 %[yl], vcc, v4, v2\n\t"                                                              \
    "v_subb_co_u32 %[yh], vcc, v5, v3, vcc"                      /* y' = u + 3q - r */
    if constexpr (SGPR_TW)
        asm(FHE_MG_CT_BODY
            : [x] "+&v"(x), [yl] "=v"(yl), [yh] "=v"(yh)
            : [y0] "v"(y0), [y1] "v"(y1), [a0] "s"(a0), [a1] "s"(a1), [b0] "s"(b0), [b1] "s"(b1),
              [nqh] "s"(m.mg_nqh), [qq] "s"(m.q), [k3] "s"(k3)
            : "vcc", "v2", "v3", "v4", "v5", "v6", "v7");
    else
        asm(FHE_MG_CT_BODY
            : [x] "+&v"(x), [yl] "=v"(yl), [yh] "=v"(yh)
            : [y0] "v"(y0), [y1] "v"(y1), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1),
              [nqh] "s"(m.mg_nqh), [qq] "s"(m.q), [k3] "s"(k3)
            : "vcc", "v2", "v3", "v4", "v5", "v6", "v7");
#undef FHE_MG_CT_BODY
    y = ((u64)yh << 32) | yl;
}

// y * w mod q on the Montgomery table words {wa = w 2^32, wb = w 2^64 mod q}: in (0, 3q), for any 64-bit y
template <bool SGPR_TW>
__device__ __forceinline__ u64 mul_mg(u64 y, u64 wa, u64 wb, const Mod &m) {
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32);
    const u32 a0 = (u32)wa, a1 = (u32)(wa >> 32), b0 = (u32)wb, b1 = (u32)(wb >> 32);
    u64 r;
    if constexpr (SGPR_TW)
        asm(FHE_MG_PRODUCT("%[y0]", "%[y1]", "%[r]")
            : [r] "=&v"(r)
            : [y0] "v"(y0), [y1] "v"(y1), [a0] "s"(a0), [a1] "s"(a1), [b0] "s"(b0), [b1] "s"(b1), [nqh] "s"(m.mg_nqh), [qq] "s"(m.q)
            : "vcc", "v2", "v3", "v6", "v7");
    else
        asm(FHE_MG_PRODUCT("%[y0]", "%[y1]", "%[r]")
            : [r] "=&v"(r)
            : [y0] "v"(y0), [y1] "v"(y1), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [nqh] "s"(m.mg_nqh), [qq] "s"(m.q)
            : "vcc", "v2", "v3", "v6", "v7");
    return r;
}
// Inverse butterfly on the Montgomery table: x' = x + y, y' = (x - y + K q) w as gs_bfly_pm (the product below 3q)
template <bool SGPR_TW>
__device__ __forceinline__ void gs_bfly_mg(u64 &x, u64 &y, u64 wa, u64 wb, u64 kq1, const Mod &m) {
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32);
    const u32 a0 = (u32)wa, a1 = (u32)(wa >> 32), b0 = (u32)wb, b1 = (u32)(wb >> 32);
    const u64 kq = kq1 - 1ull;
    u64 yo;
#define FHE_MG_GS_BODY                                                                                 \
    "v_lshl_add_u64 v[4:5], %[x], 0, %[kq]\n\t"                                                        \
    "v_sub_co_u32 v4, vcc, v4, %[y0]\n\t"                                                              \
    "v_subb_co_u32 v5, vcc, v5, %[y1], vcc\n\t"                  /* d = x + K q - y */                 \
    "v_lshl_add_u64 %[x], %[x], 0, %[y]\n\t"                     /* x' = x + y */                      \
    FHE_MG_PRODUCT("v4", "v5", "%[yo]")
    if constexpr (SGPR_TW)
        asm(FHE_MG_GS_BODY
            : [x] "+&v"(x), [yo] "=v"(yo)
            : [y] "v"(y), [y0] "v"(y0), [y1] "v"(y1), [a0] "s"(a0), [a1] "s"(a1), [b0] "s"(b0), [b1] "s"(b1),
              [nqh] "s"(m.mg_nqh), [qq] "s"(m.q), [kq] "s"(kq)
            : "vcc", "v2", "v3", "v4", "v5", "v6", "v7");
    else
        asm(FHE_MG_GS_BODY
            : [x] "+&v"(x), [yo] "=v"(yo)
            : [y] "v"(y), [y0] "v"(y0), [y1] "v"(y1), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1),
              [nqh] "s"(m.mg_nqh), [qq] "s"(m.q), [kq] "s"(kq)
            : "vcc", "v2", "v3", "v4", "v5", "v6", "v7");
#undef FHE_MG_GS_BODY
    y = yo;
}
#undef FHE_MG_PRODUCT

// VARIABLE x VARIABLE on such a modulus (zip_eq(l,r).map(l*r), arith/src/ring_nq.rs:601-604): the full 128-bit product, two
// Montgomery word steps — each is "drop the low word W and subtract W qh", exact because q = 1 (mod 2^32) — give
// a b 2^-64 + q in (0, 7.1 q) for a, b < 7q; one table product by the constant 2^64 (words 2^96, 2^128 mod q) brings back the
// factor: a b mod q in (0, 3q).  Plain C: the compiler emits 19 instructions for the first part; 28 in all against ~50 for the
// Shoup form (mul_mod_var: two 64-bit quotients).
__device__ __forceinline__ u64 mul_var_mg(u64 a, u64 b, const Mod &m) {
    typedef unsigned __int128 u128t;
    typedef __int128 i128t;
    const u128t P = (u128t)a * b;
    const u64 qh = (u64)(0u - m.mg_nqh);
    const u32 P0 = (u32)(u64)P;
    const i128t R1 = (i128t)(P >> 32) - (i128t)((u64)P0 * qh);            // (P - P0 q) / 2^32, exactly
    const u32 R10 = (u32)(u64)R1;
    const i128t R2 = (R1 >> 32) - (i128t)((u64)R10 * qh);                 // (R1 - R10 q) / 2^32: above -q, below 6.1 q + q / 2^32
    return mul_mg<true>((u64)R2 + m.q, m.mg_r96, m.mg_r128, m);
}

// Rust `f64 as i64` (saturating, NaN -> 0)
__device__ __forceinline__ long long f64_as_i64(double x) {
    if (x != x) return 0;
    if (x >= 9223372036854775808.0) return 0x7fffffffffffffffll;
    if (x <= -9223372036854775808.0) return (long long)0x8000000000000000ull;
    return (long long)x;
}
// Zq::from_f64, arith/src/zq.rs:32-39
__device__ __forceinline__ u64 zq_from_f64(u64 q, double ef) {
    const long long e = f64_as_i64(round(ef));
    const long long qi = (long long)q;
    if (e < 0 || e >= qi) return (u64)(((e % qi) + qi) % qi);
    return (u64)e;
}

// The same with the 64-bit remainder by multiplication: mu = floor(2^64 / q) (q >= 2).  e % q by the hardware-less
// software division costs ~150 instructions per coefficient, more than the transform that produced it.
__device__ __forceinline__ u64 zq_from_f64_mu(u64 q, u64 mu, double ef) {
    const long long e = f64_as_i64(round(ef));
    const u64 m = e < 0 ? 0ull - (u64)e : (u64)e;              // |e| <= 2^63; no early exit for 0 <= e < q: straight-line code
    u64 r = m - __umul64hi(m, mu) * q;                         // the quotient estimate is short by at most 2
    r = r >= q ? r - q : r;
    r = r >= q ? r - q : r;
    return (e < 0 && r) ? q - r : r;                           // ((e % q) + q) % q
}

// x mod q for any 64-bit x (Shoup with w = 1), canonical
__device__ __forceinline__ u64 reduce_any(u64 x, const Mod &m) {
    const u64 r = x - __umul64hi(x, m.onep) * m.q;  // [0, 2q)
    return canon2(r, m);
}

// (hi*2^64 + lo) mod q, canonical, for ANY 128-bit value:
// hi*2^64 + lo  ==  hi*(2^64 mod q) + lo  (mod q); both terms are reduced with the Shoup
// estimate against the fixed constants r64 and 1.
__device__ __forceinline__ u64 reduce128(u64 hi, u64 lo, const Mod &m) {
    const u64 t2 = lo - __umul64hi(lo, m.onep) * m.q;          // lo mod q, in [0,2q)
    const u64 t = mul_shoup_acc(t2, hi, m.r64, m.r64p, m.nq);  // + hi*r64 mod q: [0,4q)
    return canon4(t, m);
}
// a * b mod q, canonical, for two VARIABLE operands
// (zip_eq(l,r).map(l*r), arith/src/ring_nq.rs:601-604).
__device__ __forceinline__ u64 mul_mod_var(u64 a, u64 b, const Mod &m) {
    return reduce128(__umul64hi(a, b), a * b, m);
}

// The same for 2^62 <= q < 2^63 (AR = 3 and generic63.hip; 4q no longer fits a word): every partial result canonical before the
// next addition.  canon2's sign test needs x < 2^63 + q, which x < 2q gives for every q < 2^63.
__device__ __forceinline__ u64 reduce128_63(u64 hi, u64 lo, const Mod &m) {
    const u64 l = canon2(lo - __umul64hi(lo, m.onep) * m.q, m);
    const u64 h = canon2(mul_shoup_lazy(hi, m.r64, m.r64p, m), m);
    return canon2(l + h, m);
}
__device__ __forceinline__ u64 mul_mod_var63(u64 a, u64 b, const Mod &m) { return reduce128_63(__umul64hi(a, b), a * b, m); }

// ---- strict arithmetic for 2^62 <= q < 2^63 (AR = 3) -------------------------------------------------------------------
// The reference's Zq works for every q below 2^63 (`self.v + rhs.v`, zq.rs:225, is its limit); a lazy range needs 4q (Harvey)
// or at least the sum of two values below 2q to fit a word, and 4q >= 2^64 here.  So every value is canonical between
// butterflies: the Shoup product is exact in [0, 2q) for ANY 64-bit multiplicand and 2q < 2^64, a sum of two canonical values
// is below 2q, and canon2's sign test (x - q as a signed word) holds for x < 2^63 + q.  Three conditional subtractions per
// butterfly; ~33 instructions against 21 for the lazy Shoup butterfly.
__device__ __forceinline__ u64 add63(u64 x, u64 y, const Mod &m) { return canon2(x + y, m); }
__device__ __forceinline__ u64 sub63(u64 x, u64 y, const Mod &m) { return canon2(x + (m.q - y), m); }   // (0, 2q) -> [0, q)
__device__ __forceinline__ u64 mul63(u64 y, u64 w, u64 wp, const Mod &m) { return canon2(mul_shoup_lazy(y, w, wp, m), m); }
// ntt.rs:57-62 / 91-96 on canonical values
__device__ __forceinline__ void ct_bfly63(u64 &x, u64 &y, u64 w, u64 wp, const Mod &m) {
    const u64 u = x, t = mul63(y, w, wp, m);
    x = add63(u, t, m);
    y = sub63(u, t, m);
}
__device__ __forceinline__ void gs_bfly63(u64 &x, u64 &y, u64 w, u64 wp, const Mod &m) {
    const u64 u = x, t = y;
    x = add63(u, t, m);
    y = mul63(sub63(u, t, m), w, wp, m);
}

// Sum of products of canonical operands, reduced once per kMacChunk terms instead of once per
// term: a term is < q^2 < 2^124, so 8 of them plus a canonical carry-over stay below 2^128.
// One term costs 4 multiplies and a 128-bit add here, against ~18 multiplies for mul_mod_var.
typedef unsigned __int128 u128;
constexpr unsigned kMacChunk = 8;
struct MacAcc {
    u128 v = 0;
    __device__ __forceinline__ void mac(u64 a, u64 b) { v += (u128)a * b; }
    __device__ __forceinline__ u64 fold(const Mod &m) {   // -> canonical, and restart from it
        const u64 r = reduce128((u64)(v >> 64), (u64)v, m);
        v = r;
        return r;
    }
    // 2^62 <= q < 2^63: a term is below 2^126, so TWO terms and a canonical carry-over stay below 2^128 (kMacChunk63)
    __device__ __forceinline__ u64 fold63(const Mod &m) {
        const u64 r = reduce128_63((u64)(v >> 64), (u64)v, m);
        v = r;
        return r;
    }
};
constexpr unsigned kMacChunk63 = 2;

__device__ __forceinline__ u64 splitmix64(u64 x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

}  // namespace fhe
