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
// Measured on MI355X (tools/ubench_valu.hip, profiles/r01_ubench_valu.txt):
// v_mad_u64_u32 / v_mul_lo_u32 / v_mul_hi_u32 issue at ~4 cycles per wave64,
// plain 32-bit VOP2 at ~2; a compiled exact-Shoup butterfly costs ~100 cycles.
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
struct Mod {
    u64 q;    // modulus, 3 <= q < 2^62
    u64 q2;   // 2q
    u64 r64;  // 2^64 mod q            (for the variable x variable product)
    u64 r64p; // floor(r64 * 2^64 / q)
    u64 onep; // floor(2^64 / q)       (Shoup companion of w = 1)
};

// y * w mod q  in [0, 2q), for ANY 64-bit y and 0 <= w < q (exact quotient estimate).
__device__ __forceinline__ u64 mul_shoup_lazy(u64 y, u64 w, u64 wp, u64 q) {
    u64 qh = __umul64hi(y, wp);
    return y * w - qh * q;
}

__device__ __forceinline__ u64 csub(u64 x, u64 m) {  // x >= m ? x - m : x
    return x >= m ? x - m : x;
}

// Forward (Cooley-Tukey) butterfly, arith/src/ntt.rs:57-62:
//   U = r[j]; V = r[j+t]*S; r[j] = U+V; r[j+t] = U-V
// Lazy form: x,y in [0,4q) -> x,y in [0,4q).  Needs 4q < 2^64.
__device__ __forceinline__ void ct_bfly(u64 &x, u64 &y, u64 w, u64 wp, u64 q, u64 q2) {
    u64 u = csub(x, q2);
    u64 t = mul_shoup_lazy(y, w, wp, q);
    x = u + t;
    y = u - t + q2;
}

// Inverse (Gentleman-Sande) butterfly, arith/src/ntt.rs:91-96:
//   U = r[j]; V = r[j+t]; r[j] = U+V; r[j+t] = (U-V)*S
// Lazy form: x,y in [0,2q) -> x,y in [0,2q).
__device__ __forceinline__ void gs_bfly(u64 &x, u64 &y, u64 w, u64 wp, u64 q, u64 q2) {
    u64 s = csub(x + y, q2);
    u64 d = x - y + q2;
    x = s;
    y = mul_shoup_lazy(d, w, wp, q);
}

// [0,4q) -> [0,q)
__device__ __forceinline__ u64 canon4(u64 x, u64 q, u64 q2) { return csub(csub(x, q2), q); }
// [0,2q) -> [0,q)
__device__ __forceinline__ u64 canon2(u64 x, u64 q) { return csub(x, q); }

// a * b mod q, canonical, for two VARIABLE canonical operands
// (zip_eq(l,r).map(l*r), arith/src/ring_nq.rs:601-604).
// a*b = hi*2^64 + lo  ==  hi*(2^64 mod q) + lo  (mod q); both terms are reduced
// with the Shoup estimate against the fixed constants r64 and 1.
__device__ __forceinline__ u64 mul_mod_var(u64 a, u64 b, const Mod &m) {
    u64 lo = a * b;
    u64 hi = __umul64hi(a, b);
    u64 t1 = mul_shoup_lazy(hi, m.r64, m.r64p, m.q);          // [0,2q)
    u64 t2 = lo - __umul64hi(lo, m.onep) * m.q;               // lo mod q, in [0,2q)
    return canon4(t1 + t2, m.q, m.q2);
}

__device__ __forceinline__ u64 splitmix64(u64 x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

}  // namespace fhe
