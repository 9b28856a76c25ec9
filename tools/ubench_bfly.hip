// tools/ubench_bfly.hip — candidate Shoup/Harvey butterfly formulations, measured in
// registers only (diagnostic; results quoted in DESIGN.md).  Each variant is checked
// against a u128 reference on the host before it is timed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned __int128 u128;

__device__ __forceinline__ u64 mad64(u32 a, u32 b, u64 c) {
  u64 d; asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc"); return d;
}
__device__ __forceinline__ u64 add64(u64 a, u64 b) {
  u64 d; asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(a), "v"(b)); return d;
}
__device__ __forceinline__ u64 shl1add64(u64 a, u64 b) {  // (a<<1)+b
  u64 d; asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(d) : "v"(a), "v"(b)); return d;
}

struct Q { u64 q, q2, nq, neg4q, neg2q, q2p1, q4, q4p1; };

// variant 0: what the product kernels do today (compiler everything), [0,4q)
__device__ __forceinline__ void bf_v0(u64& x, u64& y, u64 w, u64 wp, const Q& c) {
  u64 u = x >= c.q2 ? x - c.q2 : x;
  u64 qh = __umul64hi(y, wp);
  u64 t = y * w - qh * c.q;
  x = u + t; y = u - t + c.q2;
}
// t-chain: s = u + y*w + qh*nq (mod 2^64) with mad chains; cross terms via a second 64-bit chain
__device__ __forceinline__ u64 chain_s(u64 u, u64 y, u64 w, u64 qh, const Q& c) {
  u32 y0 = (u32)y, y1 = (u32)(y >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
  u32 h0 = (u32)qh, h1 = (u32)(qh >> 32), n0 = (u32)c.nq, n1 = (u32)(c.nq >> 32);
  u64 acc = mad64(y0, w0, u);
  acc = mad64(h0, n0, acc);
  u64 H = mad64(y0, w1, 0);
  H = mad64(y1, w0, H);
  H = mad64(h0, n1, H);
  H = mad64(h1, n0, H);
  u32 hi = (u32)(acc >> 32) + (u32)H;
  return ((u64)hi << 32) | (u32)acc;
}
// variant 1: exact qh (compiler), fused chain, y' = 2u+2q+1 + ~s, csub each stage via compare
__device__ __forceinline__ void bf_v1(u64& x, u64& y, u64 w, u64 wp, const Q& c) {
  u64 u = x >= c.q2 ? x - c.q2 : x;
  u64 qh = __umul64hi(y, wp);
  u64 s = chain_s(u, y, w, qh, c);
  y = add64(shl1add64(u, c.q2p1), ~s);
  x = s;
}
// variant 2: as v1 but csub by sign of (x - 2q) computed with one 64-bit add
__device__ __forceinline__ void bf_v2(u64& x, u64& y, u64 w, u64 wp, const Q& c) {
  u64 d = add64(x, c.neg2q);
  u64 u = ((long long)d < 0) ? x : d;
  u64 qh = __umul64hi(y, wp);
  u64 s = chain_s(u, y, w, qh, c);
  y = add64(shl1add64(u, c.q2p1), ~s);
  x = s;
}
// variant 3: WIDE (q < 2^61): no csub (the kernel applies one csub(4q) every other stage)
__device__ __forceinline__ void bf_v3(u64& x, u64& y, u64 w, u64 wp, const Q& c, bool do_csub) {
  u64 u = x;
  if (do_csub) { u64 d = add64(x, c.neg4q); u = ((long long)d < 0) ? x : d; }
  u64 qh = __umul64hi(y, wp);
  u64 s = chain_s(u, y, w, qh, c);
  y = add64(shl1add64(u, c.q2p1), ~s);
  x = s;
}
// variant 4: v3 but the cross terms with mul_lo + add3 (compiler's choice) instead of the H chain
__device__ __forceinline__ void bf_v4(u64& x, u64& y, u64 w, u64 wp, const Q& c, bool do_csub) {
  u64 u = x;
  if (do_csub) { u64 d = add64(x, c.neg4q); u = ((long long)d < 0) ? x : d; }
  u64 qh = __umul64hi(y, wp);
  u32 y0 = (u32)y, y1 = (u32)(y >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
  u32 h0 = (u32)qh, h1 = (u32)(qh >> 32), n0 = (u32)c.nq, n1 = (u32)(c.nq >> 32);
  u64 acc = mad64(y0, w0, u);
  acc = mad64(h0, n0, acc);
  u32 hi = (u32)(acc >> 32) + y0 * w1 + y1 * w0 + h0 * n1 + h1 * n0;
  u64 s = ((u64)hi << 32) | (u32)acc;
  y = add64(shl1add64(u, c.q2p1), ~s);
  x = s;
}

// ---- variant 5: v3 with the three sources of v_mov removed -------------------------------
__device__ __forceinline__ u64 mad64z(u32 a, u32 b) {  // a*b
  u64 d; asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b) : "vcc"); return d;
}
__device__ __forceinline__ u32 add32(u32 a, u32 b) { u32 d; asm("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ u32 mulhi32(u32 a, u32 b) { u32 d; asm("v_mul_hi_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ u64 pack(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
// exact mulhi64 with carries in 32-bit adds instead of zero-extended 64-bit operands
__device__ __forceinline__ u64 mulhi64_v5(u64 y, u64 p) {
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32), p0 = (u32)p, p1 = (u32)(p >> 32);
  const u32 t = mulhi32(y0, p0);
  const u64 A = mad64(y1, p0, (u64)t);      // y1*p0 + hi(y0*p0): no overflow
  const u64 B = mad64z(y0, p1);
  u32 s, u, c2;
  asm("v_add_co_u32 %0, vcc, %3, %4\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, %5, %6, vcc\n\ts_nop 1\n\tv_addc_co_u32 %2, vcc, 0, 0, vcc"
      : "=&v"(s), "=&v"(u), "=v"(c2)
      : "v"((u32)A), "v"((u32)B), "v"((u32)(A >> 32)), "v"((u32)(B >> 32)) : "vcc");
  (void)s;
  return mad64(y1, p1, pack(u, c2));
}
__device__ __forceinline__ u64 chain_s5(u64 u, u64 y, u64 w, u64 qh, const Q& c) {
  u32 y0 = (u32)y, y1 = (u32)(y >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
  u32 h0 = (u32)qh, h1 = (u32)(qh >> 32), n0 = (u32)c.nq, n1 = (u32)(c.nq >> 32);
  u64 acc = mad64(y0, w0, u);
  acc = mad64(h0, n0, acc);
  u64 H = mad64z(y0, w1);
  H = mad64(y1, w0, H);
  H = mad64(h0, n1, H);
  H = mad64(h1, n0, H);
  return pack((u32)acc, add32((u32)(acc >> 32), (u32)H));
}
__device__ __forceinline__ void bf_v5(u64& x, u64& y, u64 w, u64 wp, const Q& c, bool do_csub) {
  u64 u = x;
  if (do_csub) { u64 d = add64(x, c.neg4q); u = ((int)(u32)(d >> 32) < 0) ? x : d; }
  u64 qh = mulhi64_v5(y, wp);
  u64 s = chain_s5(u, y, w, qh, c);
  y = add64(shl1add64(u, c.q2p1), ~s);
  x = s;
}
// variant 7: exact mulhi64 without any zero-extension move: 64-bit shifts produce the {hi,0}
// pairs, the one possible overflow is taken from the mad's carry-out SGPR
__device__ __forceinline__ u64 shr32(u64 a) { u64 d; asm("v_lshrrev_b64 %0, 32, %1" : "=v"(d) : "v"(a)); return d; }
__device__ __forceinline__ u64 mad64c(u32 a, u32 b, u64 c, u32& cbit) {
  u64 d, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c));
  asm("v_cndmask_b32 %0, 0, 1, %1" : "=v"(cbit) : "s"(carry));
  return d;
}
__device__ __forceinline__ u64 mulhi64_v7(u64 y, u64 p) {
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32), p0 = (u32)p, p1 = (u32)(p >> 32);
  const u64 A = mad64(y1, p0, shr32(mad64z(y0, p0)));
  u32 cbit;
  const u64 B = mad64c(y0, p1, A, cbit);
  const u64 r = mad64(y1, p1, shr32(B));
  return pack((u32)r, add32((u32)(r >> 32), cbit));
}
__device__ __forceinline__ void bf_v7(u64& x, u64& y, u64 w, u64 wp, const Q& c, bool do_csub) {
  u64 u = x;
  if (do_csub) { u64 d = add64(x, c.neg4q); u = ((int)(u32)(d >> 32) < 0) ? x : d; }
  u64 qh = mulhi64_v7(y, wp);
  u64 s = chain_s5(u, y, w, qh, c);
  y = add64(shl1add64(u, c.q2p1), ~s);
  x = s;
}
// variant 6: v5's chain with the compiler's __umul64hi (isolates the mulhi rewrite)
__device__ __forceinline__ void bf_v6(u64& x, u64& y, u64 w, u64 wp, const Q& c, bool do_csub) {
  u64 u = x;
  if (do_csub) { u64 d = add64(x, c.neg4q); u = ((int)(u32)(d >> 32) < 0) ? x : d; }
  u64 qh = __umul64hi(y, wp);
  u64 s = chain_s5(u, y, w, qh, c);
  y = add64(shl1add64(u, c.q2p1), ~s);
  x = s;
}

// ---- variants 8..11: what the kernels run today (v8) and three candidate trims ---------------
// v8: v6 with both mad chains in ONE asm statement (no compiler s_nop between dependent asm)
__device__ __forceinline__ u64 chain_s8(u64 u, u64 y, u64 w, u64 qh, const Q& c) {
  u32 y0 = (u32)y, y1 = (u32)(y >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
  u32 h0 = (u32)qh, h1 = (u32)(qh >> 32), n0 = (u32)c.nq, n1 = (u32)(c.nq >> 32);
  u64 acc, H;
  asm("v_mad_u64_u32 %0, vcc, %3, %5, %2\n\t"
      "v_mad_u64_u32 %1, vcc, %3, %6, 0\n\t"
      "v_mad_u64_u32 %0, vcc, %7, %9, %0\n\t"
      "v_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
      "v_mad_u64_u32 %1, vcc, %7, %10, %1\n\t"
      "v_mad_u64_u32 %1, vcc, %8, %9, %1"
      : "=&v"(acc), "=&v"(H)
      : "v"(u), "v"(y0), "v"(y1), "v"(w0), "v"(w1), "v"(h0), "v"(h1), "v"(n0), "v"(n1) : "vcc");
  return pack((u32)acc, add32((u32)(acc >> 32), (u32)H));
}
__device__ __forceinline__ void bf_v8(u64& x, u64& y, u64 w, u64 wp, const Q& c, bool do_csub) {
  u64 u = x;
  if (do_csub) { u64 d = add64(x, c.neg4q); u = ((int)(u32)(d >> 32) < 0) ? x : d; }
  u64 qh = __umul64hi(y, wp);
  u64 s = chain_s8(u, y, w, qh, c);
  y = add64(shl1add64(u, c.q2p1), ~s);
  x = s;
}
// v9: v8 with the csub's compare forced to 32 bits (the compiler widens `(int)hi < 0` of a
// 64-bit value to v_cmp_gt_i64, a half-rate op): the high word goes through an empty asm
__device__ __forceinline__ u64 csub32(u64 x, u64 negm) {
  const u64 d = add64(x, negm);
  u32 d1 = (u32)(d >> 32);
  asm("" : "+v"(d1));
  return ((int)d1 < 0) ? x : d;
}
__device__ __forceinline__ void bf_v9(u64& x, u64& y, u64 w, u64 wp, const Q& c, bool do_csub) {
  u64 u = x;
  if (do_csub) u = csub32(x, c.neg4q);
  u64 qh = __umul64hi(y, wp);
  u64 s = chain_s8(u, y, w, qh, c);
  y = add64(shl1add64(u, c.q2p1), ~s);
  x = s;
}
// v10: exact mulhi64 whose last step adds the zero-extended word with a multiply by 1
// (v_mad_u64_u32 d, B.hi, 1, C) instead of v_mov + v_lshl_add_u64
__device__ __forceinline__ u64 mad64_1(u32 a, u64 c) {
  u64 d; asm("v_mad_u64_u32 %0, vcc, %1, 1, %2" : "=v"(d) : "v"(a), "v"(c) : "vcc"); return d;
}
__device__ __forceinline__ u64 mulhi64_v10(u64 y, u64 p) {
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32), p0 = (u32)p, p1 = (u32)(p >> 32);
  const u32 t = mulhi32(y0, p0);
  const u64 A = mad64(y1, p0, (u64)t);
  const u64 B = mad64(y0, p1, (u64)(u32)A);
  const u64 C = mad64(y1, p1, (u64)(u32)(A >> 32));
  return mad64_1((u32)(B >> 32), C);
}
__device__ __forceinline__ void bf_v10(u64& x, u64& y, u64 w, u64 wp, const Q& c, bool do_csub) {
  u64 u = x;
  if (do_csub) u = csub32(x, c.neg4q);
  u64 qh = mulhi64_v10(y, wp);
  u64 s = chain_s8(u, y, w, qh, c);
  y = add64(shl1add64(u, c.q2p1), ~s);
  x = s;
}
// v11: APPROXIMATE quotient qh' = y1*p1 + hi32(y1*p0) + hi32(y0*p1) in [qh-2, qh]: the lazy
// product is then in [0,4q); with q < 2^61 every stage does csub(4q) on x and all values stay
// below 8q <= 2^64:  x' = u + t < 8q,  y' = u - t + 4q in (0, 8q).
__device__ __forceinline__ void bf_v11(u64& x, u64& y, u64 w, u64 wp, const Q& c, bool) {
  const u64 u = csub32(x, c.neg4q);
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32), p0 = (u32)wp, p1 = (u32)(wp >> 32);
  const u32 a = mulhi32(y1, p0), b = mulhi32(y0, p1);
  const u64 qh = mad64_1(b, mad64(y1, p1, (u64)a));
  u64 s = chain_s8(u, y, w, qh, c);
  y = add64(shl1add64(u, c.q4p1), ~s);
  x = s;
}

// v12: SPLIT MULTIPLICAND.  y = y1*2^32 + y0, so y*w = y0*w + y1*w2 with w2 = w*2^32 mod q, and each 32-bit
// half gets its own Shoup quotient from a 32-bit companion (p = floor(w*2^32/q), p2 = floor(w2*2^32/q)): two
// v_mul_hi_u32, no 64-bit mulhi, no zero-extension moves.  Each half lands in [0,2q): t in [0,4q), so (q < 2^61)
// every stage brings x below 4q first and all values stay below 8q <= 2^64:  x' = u + t,  y' = u - t + 4q.
struct T12 { u32 w0, w1, v0, v1, p, p2; };   // w, w2 = w*2^32 mod q, the two 32-bit companions
__device__ __forceinline__ void bf_v12(u64& x, u64& y, const T12& t, const Q& c) {
  const u64 u = csub32(x, c.neg4q);
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32), n0 = (u32)c.nq, n1 = (u32)(c.nq >> 32);
  const u32 qa = mulhi32(y0, t.p), qb = mulhi32(y1, t.p2);
  u64 acc, H;
  asm("v_mad_u64_u32 %0, vcc, %3, %5, %2\n\t"      // acc = y0*w0 + u
      "v_mad_u64_u32 %1, vcc, %3, %6, 0\n\t"       // H   = y0*w1
      "v_mad_u64_u32 %0, vcc, %4, %7, %0\n\t"      // acc += y1*v0
      "v_mad_u64_u32 %1, vcc, %4, %8, %1\n\t"      // H   += y1*v1
      "v_mad_u64_u32 %0, vcc, %9, %11, %0\n\t"     // acc += qa*n0
      "v_mad_u64_u32 %1, vcc, %9, %12, %1\n\t"     // H   += qa*n1
      "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"    // acc += qb*n0
      "v_mad_u64_u32 %1, vcc, %10, %12, %1"          // H   += qb*n1
      : "=&v"(acc), "=&v"(H)
      : "v"(u), "v"(y0), "v"(y1), "v"(t.w0), "v"(t.w1), "v"(t.v0), "v"(t.v1), "v"(qa), "v"(qb), "v"(n0), "v"(n1) : "vcc");
  const u64 s = pack((u32)acc, add32((u32)(acc >> 32), (u32)H));
  y = add64(shl1add64(u, c.q4p1), ~s);
  x = s;
}


// ---- round 3: FIVE-multiply butterflies ------------------------------------------------------------------------------
// Split multiplicand WITHOUT a quotient.  y = y1*2^32 + y0, y*w = y0*w + y1*W2 (mod q) with W2 = w*2^32 mod q kept in the
// table instead of the Shoup companion: T = y0*w + y1*W2 < 2^33 q is four 32x32 products.  What reduces T:
//   v15 (pseudo-Mersenne q = 2^k - delta, the headline modulus 2^61 - 2^21 + 1): 2^(k+1) = 2 delta (mod q), so
//       r = (T mod 2^(k+1)) + (T >> (k+1)) * 2 delta  — ONE more multiply (T >> (k+1) < 2^32), r < 2q + q/64;
//   v14 (q = qh*2^32 + 1, the verdict's Montgomery case; table holds w*2^32, w*2^64 mod q): one word round with
//       m = T0 and q^-1 = 1 (mod 2^32): r = (T >> 32) - T0*qh + q  — one more multiply, r in (0, 3q);
//   v16 (any odd q < 2^62, word Montgomery): m = T0 * (-q^-1 mod 2^32), r = (T + m q) >> 32 — three more, r < 3q.
// All three: x' = u + r, y' = u - r + 3q, so a stage takes x < B q to < (B + 3) q and (q < 2^61) x is brought down before
// every other stage: v15 by k = x >> 61, x = (x mod 2^61) + k delta (three instructions, < q + 8 delta), v14 / v16 by
// a conditional subtraction of 4q then 2q... (here: csub 4q every stage for v14 / v16, the cheaper schedule measured).
__device__ __forceinline__ u64 mad64co(u32 a, u32 b, u64 c, u64& carry) {   // carry-out kept (SGPR pair)
  u64 d; asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c)); return d;
}
__device__ __forceinline__ u32 carry01(u64 carry) {   // 0 / 1 per lane
  u32 d; asm("v_addc_co_u32 %0, vcc, 0, 0, %1" : "=v"(d) : "s"(carry) : "vcc"); return d;
}
__device__ __forceinline__ u32 carry_add(u32 a, u64 carry) {   // a + carry
  u32 d; asm("v_addc_co_u32 %0, vcc, 0, %1, %2" : "=v"(d) : "v"(a), "s"(carry) : "vcc"); return d;
}
__device__ __forceinline__ u32 sub32(u32 a, u32 b) { u32 d; asm("v_sub_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
struct TS { u32 a0, a1, b0, b1; };      // the two table words: a multiplies y0, b multiplies y1
struct QS { u64 q, q3p1, neg4q; u32 c2, delta, sh, mask, qinv, nqh, q0, q1; };

// T = y0*a + y1*b as (T mod 2^32, T >> 32): four multiplies, one carry
__device__ __forceinline__ void split_T(u64 y, const TS& t, u32& T0, u64& B) {
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32);
  u64 carry;
  u64 N = mad64z(y0, t.a0);
  N = mad64co(y1, t.b0, N, carry);
  B = mad64(y0, t.a1, pack((u32)(N >> 32), carry01(carry)));
  B = mad64(y1, t.b1, B);
  T0 = (u32)N;
}
__device__ __forceinline__ u64 pm_reduce(u64 x, const QS& c) {   // q = 2^61 - delta: x -> (x mod 2^61) + (x >> 61) delta
  const u32 x1 = (u32)(x >> 32);
  return mad64(x1 >> 29, c.delta, pack((u32)x, x1 & 0x1fffffffu));
}
__device__ __forceinline__ void bf_v15(u64& x, u64& y, const TS& t, const QS& c, bool red) {
  const u64 u = red ? pm_reduce(x, c) : x;
  u32 T0; u64 B;
  split_T(y, t, T0, B);
  const u32 th = __builtin_amdgcn_alignbit((u32)(B >> 32), (u32)B, 30);       // T >> 62
  const u64 r = mad64(th, c.c2, pack(T0, (u32)B & 0x3fffffffu));               // (T mod 2^62) + th * 2 delta
  x = add64(u, r);
  y = add64(add64(u, c.q3p1), ~r);
}
__device__ __forceinline__ void bf_v14(u64& x, u64& y, const TS& t, const QS& c, bool) {
  const u64 u = csub32(x, c.neg4q);
  u32 T0; u64 B;
  split_T(y, t, T0, B);
  u64 R = mad64(T0, c.nqh, B);                                                 // B + T0*(2^32 - qh)
  R = pack((u32)R, sub32((u32)(R >> 32), T0));                                 // B - T0*qh   in (-2^61, 2^62)
  const u64 rr = add64(R, c.q);                                                // (0, 3q)
  x = add64(u, rr);
  y = add64(add64(u, c.q3p1), ~rr);
}
__device__ __forceinline__ void bf_v16(u64& x, u64& y, const TS& t, const QS& c, bool) {
  const u64 u = csub32(x, c.neg4q);
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32);
  u64 c1, c2;
  u64 N = mad64z(y0, t.a0);
  N = mad64co(y1, t.b0, N, c1);
  u32 m; asm("v_mul_lo_u32 %0, %1, %2" : "=v"(m) : "v"((u32)N), "v"(c.qinv));
  N = mad64co(m, c.q0, N, c2);                                                 // low word becomes 0
  u64 B = mad64(y0, t.a1, pack((u32)(N >> 32), carry_add(carry01(c1), c2)));
  B = mad64(y1, t.b1, B);
  B = mad64(m, c.q1, B);                                                       // (T + m q) >> 32  in [0, 3q)
  x = add64(u, B);
  y = add64(add64(u, c.q3p1), ~B);
}

// v17: v15 as ONE asm statement.  Between two dependent asm STATEMENTS the compiler inserts an s_nop (it must assume a
// dst-forwarding hazard in any inline asm): v15 above carries 5 per butterfly.  Sub-registers of a 64-bit asm operand
// cannot be named, so the temporaries (N = v[60:61], B = v[62:63]) are physical registers listed as clobbers.
__device__ __forceinline__ void bf_v17(u64& x, u64& y, const TS& t, const QS& c, bool red) {
  const u64 u = red ? pm_reduce(x, c) : x;
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32);
  u64 xo, yo;
  asm("v_mad_u64_u32 v[60:61], vcc, %[y0], %[a0], 0\n\t"
        "v_mad_u64_u32 v[60:61], vcc, %[y1], %[b0], v[60:61]\n\t"     // carry -> vcc
        "v_mov_b32 v62, v61\n\t"
        "v_addc_co_u32 v63, vcc, 0, 0, vcc\n\t"                        // {n1, carry}
        "v_mad_u64_u32 v[62:63], vcc, %[y0], %[a1], v[62:63]\n\t"
        "v_mad_u64_u32 v[62:63], vcc, %[y1], %[b1], v[62:63]\n\t"     // B = T >> 32
        "v_and_b32 v61, 0x3fffffff, v62\n\t"                           // T mod 2^62 = {n0, b0 & mask}
        "v_alignbit_b32 v62, v63, v62, 30\n\t"                         // T >> 62
        "v_mad_u64_u32 v[60:61], vcc, v62, %[c2], v[60:61]\n\t"       // r
        "v_lshl_add_u64 %[x], %[u], 0, v[60:61]\n\t"
        "v_not_b32 v60, v60\n\t"
        "v_not_b32 v61, v61\n\t"
        "v_lshl_add_u64 %[yo], %[u], 0, %[k3]\n\t"
        "v_lshl_add_u64 %[yo], %[yo], 0, v[60:61]"
        : [x] "=&v"(xo), [yo] "=&v"(yo)
        : [u] "v"(u), [y0] "v"(y0), [y1] "v"(y1), [a0] "v"(t.a0), [a1] "v"(t.a1), [b0] "v"(t.b0), [b1] "v"(t.b1),
          [c2] "s"(c.c2), [k3] "s"(c.q3p1)
        : "vcc", "v60", "v61", "v62", "v63");
  x = xo; y = yo;
}

// v18 (round 4): v17 with y' = (u + 3q) - r as a borrow chain (v_sub_co_u32 / v_subb_co_u32) instead of
// (u + 3q + 1) + ~r (two v_not + v_lshl_add_u64): 13 instructions.  The halves of a 64-bit asm operand cannot be named, so
// u + 3q is formed in the physical pair v[58:59] and y' leaves as two 32-bit outputs the compiler joins into a pair.
__device__ __forceinline__ void bf_v18(u64& x, u64& y, const TS& t, const QS& c, bool red) {
  const u64 u = red ? pm_reduce(x, c) : x;
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32);
  u64 xo; u32 yl, yh;
  asm("v_mad_u64_u32 v[60:61], vcc, %[y0], %[a0], 0\n\t"
        "v_mad_u64_u32 v[60:61], vcc, %[y1], %[b0], v[60:61]\n\t"
        "v_mov_b32 v62, v61\n\t"
        "v_addc_co_u32 v63, vcc, 0, 0, vcc\n\t"
        "v_mad_u64_u32 v[62:63], vcc, %[y0], %[a1], v[62:63]\n\t"
        "v_mad_u64_u32 v[62:63], vcc, %[y1], %[b1], v[62:63]\n\t"
        "v_lshl_add_u64 v[58:59], %[u], 0, %[k3]\n\t"                // u + 3q
        "v_and_b32 v61, 0x3fffffff, v62\n\t"
        "v_alignbit_b32 v62, v63, v62, 30\n\t"
        "v_mad_u64_u32 v[60:61], vcc, v62, %[c2], v[60:61]\n\t"       // r
        "v_lshl_add_u64 %[x], %[u], 0, v[60:61]\n\t"
        "v_sub_co_u32 %[yl], vcc, v58, v60\n\t"
        "v_subb_co_u32 %[yh], vcc, v59, v61, vcc"
        : [x] "=&v"(xo), [yl] "=&v"(yl), [yh] "=&v"(yh)
        : [u] "v"(u), [y0] "v"(y0), [y1] "v"(y1), [a0] "v"(t.a0), [a1] "v"(t.a1), [b0] "v"(t.b0), [b1] "v"(t.b1),
          [c2] "s"(c.c2), [k3] "s"(c.q3p1 - 1)
        : "vcc", "v58", "v59", "v60", "v61", "v62", "v63");
  x = xo; y = pack(yl, yh);
}
// v19 (round 5): v16 — any odd q < 2^61, seven multiplies — as ONE asm statement, the step that turned v14 into a 14 %
// kernel gain.  14 instructions: N = y0 a0 + y1 b0 (carry c1), m = N0 qinv, N += m q0 (low word -> 0, carry c2),
// H = {N1, c1 + c2} + y0 a1 + y1 b1 + m q1 = (T + m q) >> 32 in [0, 3q); x' = u + H, y' = u + 3q - H as a borrow chain.
__device__ __forceinline__ void bf_v19(u64& x, u64& y, const TS& t, const QS& c, bool) {
  const u64 u = csub32(x, c.neg4q);
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32);
  u64 xo; u32 yl, yh;
  asm("v_mad_u64_u32 v[60:61], vcc, %[y0], %[a0], 0\n\t"
        "v_mad_u64_u32 v[60:61], vcc, %[y1], %[b0], v[60:61]\n\t"     // carry c1 -> vcc
        "v_addc_co_u32 v63, vcc, 0, 0, vcc\n\t"                        // c1
        "v_mul_lo_u32 v58, v60, %[qinv]\n\t"                           // m
        "v_mad_u64_u32 v[60:61], vcc, v58, %[q0], v[60:61]\n\t"       // low word becomes 0; carry c2
        "v_addc_co_u32 v63, vcc, 0, v63, vcc\n\t"                      // c1 + c2
        "v_mov_b32 v62, v61\n\t"
        "v_mad_u64_u32 v[62:63], vcc, %[y0], %[a1], v[62:63]\n\t"
        "v_mad_u64_u32 v[62:63], vcc, %[y1], %[b1], v[62:63]\n\t"
        "v_mad_u64_u32 v[62:63], vcc, v58, %[q1], v[62:63]\n\t"       // H
        "v_lshl_add_u64 v[60:61], %[u], 0, %[k3]\n\t"                  // u + 3q
        "v_lshl_add_u64 %[x], %[u], 0, v[62:63]\n\t"
        "v_sub_co_u32 %[yl], vcc, v60, v62\n\t"
        "v_subb_co_u32 %[yh], vcc, v61, v63, vcc"
        : [x] "=&v"(xo), [yl] "=&v"(yl), [yh] "=&v"(yh)
        : [u] "v"(u), [y0] "v"(y0), [y1] "v"(y1), [a0] "v"(t.a0), [a1] "v"(t.a1), [b0] "v"(t.b0), [b1] "v"(t.b1),
          [qinv] "s"(c.qinv), [q0] "s"(c.q0), [q1] "s"(c.q1), [k3] "s"(c.q3p1 - 1)
        : "vcc", "v58", "v60", "v61", "v62", "v63");
  x = xo; y = pack(yl, yh);
}
template <int V>
__global__ void ks(u64* p, QS c, TS t, int iters) {
  u64 x[4], y[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { x[j] = p[threadIdx.x + 64 * j] % c.q; y[j] = p[threadIdx.x + 64 * j + 256] % c.q; }
  for (int i = 0; i < iters; i += 2) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (V == 14) bf_v14(x[j], y[j], t, c, false);
      if (V == 15) bf_v15(x[j], y[j], t, c, false);
      if (V == 16) bf_v16(x[j], y[j], t, c, false);
      if (V == 17) bf_v17(x[j], y[j], t, c, false);
      if (V == 18) bf_v18(x[j], y[j], t, c, false);
      if (V == 19) bf_v19(x[j], y[j], t, c, false);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (V == 14) bf_v14(x[j], y[j], t, c, true);
      if (V == 15) bf_v15(x[j], y[j], t, c, true);
      if (V == 16) bf_v16(x[j], y[j], t, c, true);
      if (V == 17) bf_v17(x[j], y[j], t, c, true);
      if (V == 18) bf_v18(x[j], y[j], t, c, true);
      if (V == 19) bf_v19(x[j], y[j], t, c, true);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) { p[threadIdx.x + 64 * j] = x[j]; p[threadIdx.x + 64 * j + 256] = y[j]; }
}

template <int V>
__global__ void k(u64* p, Q c, u64 w, u64 wp, int iters) {
  u64 x[4], y[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { x[j] = p[threadIdx.x + 64 * j] % c.q; y[j] = p[threadIdx.x + 64 * j + 256] % c.q; }
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (V == 0) bf_v0(x[j], y[j], w, wp, c);
      if (V == 1) bf_v1(x[j], y[j], w, wp, c);
      if (V == 2) bf_v2(x[j], y[j], w, wp, c);
      if (V == 3) { bf_v3(x[j], y[j], w, wp, c, false); }
      if (V == 4) { bf_v4(x[j], y[j], w, wp, c, false); }
      if (V == 5) { bf_v5(x[j], y[j], w, wp, c, false); }
      if (V == 6) { bf_v6(x[j], y[j], w, wp, c, false); }
      if (V == 7) { bf_v7(x[j], y[j], w, wp, c, false); }
      if (V == 8) bf_v8(x[j], y[j], w, wp, c, false);
      if (V == 9) bf_v9(x[j], y[j], w, wp, c, false);
      if (V == 10) bf_v10(x[j], y[j], w, wp, c, false);
      if (V == 11) bf_v11(x[j], y[j], w, wp, c, false);
    }
    if (V >= 3) {  // second stage of the pair carries the csub
      i++;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (V == 3) bf_v3(x[j], y[j], w, wp, c, true);
        if (V == 4) bf_v4(x[j], y[j], w, wp, c, true);
        if (V == 5) bf_v5(x[j], y[j], w, wp, c, true);
        if (V == 6) bf_v6(x[j], y[j], w, wp, c, true);
        if (V == 7) bf_v7(x[j], y[j], w, wp, c, true);
        if (V == 8) bf_v8(x[j], y[j], w, wp, c, true);
        if (V == 9) bf_v9(x[j], y[j], w, wp, c, true);
        if (V == 10) bf_v10(x[j], y[j], w, wp, c, true);
        if (V == 11) bf_v11(x[j], y[j], w, wp, c, true);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) { p[threadIdx.x + 64 * j] = x[j]; p[threadIdx.x + 64 * j + 256] = y[j]; }
}

__global__ void k12(u64* p, Q c, T12 t, int iters) {
  u64 x[4], y[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { x[j] = p[threadIdx.x + 64 * j] % c.q; y[j] = p[threadIdx.x + 64 * j + 256] % c.q; }
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) bf_v12(x[j], y[j], t, c);
  }
#pragma unroll
  for (int j = 0; j < 4; j++) { p[threadIdx.x + 64 * j] = x[j]; p[threadIdx.x + 64 * j + 256] = y[j]; }
}

// v13: the 32-bit butterfly of digit32.hip / bfv32.hip modulo a 27-bit prime p < 2^32 / 25: negated lazy Shoup product,
// x - nv, x + nv + 2p (v_add3_u32) — six instructions, no conditional subtraction; the values grow by 2p per butterfly,
// so every 8 butterflies both are brought below 2p (x - mulhi(x, floor(2^32 / p)) * p: three instructions each), as the
// kernels do between rounds.  Counted per butterfly including that share (6 + 6/8 instructions).
__global__ void k13(u32* p, u32 pr, u32 bq, u32 w, u32 wp, int iters) {
  u32 x[8], y[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { x[j] = p[threadIdx.x + 64 * j] % pr; y[j] = p[threadIdx.x + 64 * j + 512] % pr; }
  const u32 p2 = 2u * pr;
  for (int i = 0; i < iters; i += 8) {
#pragma unroll
    for (int s = 0; s < 8; s++)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const u32 nv = __umulhi(y[j], wp) * pr - y[j] * w;
        const u32 u = x[j];
        x[j] = u - nv;
        y[j] = u + nv + p2;
      }
#pragma unroll
    for (int j = 0; j < 8; j++) { x[j] -= __umulhi(x[j], bq) * pr; y[j] -= __umulhi(y[j], bq) * pr; }
  }
#pragma unroll
  for (int j = 0; j < 8; j++) { p[threadIdx.x + 64 * j] = x[j]; p[threadIdx.x + 64 * j + 512] = y[j]; }
}

static u64 href(u64 x, u64 y, u64 w, u64 q, int iters, u64* yo) {  // canonical reference
  for (int i = 0; i < iters; i++) {
    u64 t = (u64)(((u128)y * w) % q);
    u64 s = (x + t) % q, d = (x + q - t) % q;
    x = s; y = d;
  }
  *yo = y; return x;
}

static void launch(int v, int blocks, int threads, u64* p, Q c, u64 w, u64 wp, int it) {
  switch (v) {
    case 0: k<0><<<blocks, threads>>>(p, c, w, wp, it); break;   case 1: k<1><<<blocks, threads>>>(p, c, w, wp, it); break;
    case 2: k<2><<<blocks, threads>>>(p, c, w, wp, it); break;   case 3: k<3><<<blocks, threads>>>(p, c, w, wp, it); break;
    case 4: k<4><<<blocks, threads>>>(p, c, w, wp, it); break;   case 5: k<5><<<blocks, threads>>>(p, c, w, wp, it); break;
    case 6: k<6><<<blocks, threads>>>(p, c, w, wp, it); break;   case 7: k<7><<<blocks, threads>>>(p, c, w, wp, it); break;
    case 8: k<8><<<blocks, threads>>>(p, c, w, wp, it); break;   case 9: k<9><<<blocks, threads>>>(p, c, w, wp, it); break;
    case 10: k<10><<<blocks, threads>>>(p, c, w, wp, it); break; case 11: k<11><<<blocks, threads>>>(p, c, w, wp, it); break;
  }
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  u64 q = 2305843009211596801ull, w = 1681162619342215248ull;
  u64 wp = (u64)((((u128)w) << 64) / q);
  Q c{q, 2 * q, (u64)0 - q, (u64)0 - 4 * q, (u64)0 - 2 * q, 2 * q + 1, 4 * q, 4 * q + 1};
  u64* p; hipMalloc(&p, 4096 * 8);
  std::vector<u64> h(512), o(512);
  for (int i = 0; i < 512; i++) h[i] = (0x9E3779B97F4A7C15ull * (i + 1)) ^ (0xD1B54A32D192ED03ull * (i + 7));
  const char* names[] = {"v0 compiler", "v1 chain+not", "v2 chain+signcsub", "v3 wide csub/2", "v4 wide mul_lo", "v5 no-mov mulhi+pack", "v6 no-mov pack only", "v7 shift-zext mulhi", "v8 fused chains (prod)", "v9 v8+cmp32", "v10 v9+mad-by-1 mulhi", "v11 approx qh, csub/1"};
  for (int v = 8; v < 12; v++) {
    // correctness at 6 iterations (even, so v3/v4 pairs are whole)
    hipMemcpy(p, h.data(), 512 * 8, hipMemcpyHostToDevice);
    int it = 6;
    launch(v, 1, 64, p, c, w, wp, it);
    hipMemcpy(o.data(), p, 512 * 8, hipMemcpyDeviceToHost);
    int bad = 0; u64 mx = 0;
    for (int t = 0; t < 64; t++) for (int j = 0; j < 4; j++) {
      u64 yo, xo = href(h[t + 64 * j] % q, h[t + 64 * j + 256] % q, w, q, it, &yo);
      u64 gx = o[t + 64 * j], gy = o[t + 64 * j + 256];
      if (gx % q != xo || gy % q != yo) bad++;
      if (gx > mx) mx = gx; if (gy > mx) mx = gy;
    }
    printf("%-20s correctness: %s (max value / q = %.3f)\n", names[v], bad ? "FAIL" : "ok", (double)mx / (double)q);
    for (int wpS : {2, 4, 8}) {
      int blocks = cus * wpS, threads = 256, iters = 2000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float best = 1e30f;
      for (int r = 0; r < 4; r++) {
        hipEventRecord(e0);
        launch(v, blocks, threads, p, c, w, wp, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
      }
      double bf = (double)blocks * threads * 4 * iters;
      printf("   wpS=%d %8.3f ms  %8.1f Gbfly/s  %6.1f cyc/bfly-wave/SIMD(nominal clk)\n", wpS, best, bf / best * 1e-6,
             best * 1e-3 * clk / ((double)iters * 4 * wpS));
    }
  }
  {  // v12
    const u64 w2 = (u64)((((u128)w) << 32) % q);
    T12 t{(u32)w, (u32)(w >> 32), (u32)w2, (u32)(w2 >> 32), (u32)((((u128)w) << 32) / q), (u32)((((u128)w2) << 32) / q)};
    hipMemcpy(p, h.data(), 512 * 8, hipMemcpyHostToDevice);
    int it = 6;
    k12<<<1, 64>>>(p, c, t, it);
    hipMemcpy(o.data(), p, 512 * 8, hipMemcpyDeviceToHost);
    int bad = 0; u64 mx = 0;
    for (int tt = 0; tt < 64; tt++) for (int j = 0; j < 4; j++) {
      u64 yo, xo = href(h[tt + 64 * j] % q, h[tt + 64 * j + 256] % q, w, q, it, &yo);
      u64 gx = o[tt + 64 * j], gy = o[tt + 64 * j + 256];
      if (gx % q != xo || gy % q != yo) bad++;
      if (gx > mx) mx = gx; if (gy > mx) mx = gy;
    }
    printf("%-20s correctness: %s (max value / q = %.3f)\n", "v12 split multiplicand", bad ? "FAIL" : "ok", (double)mx / (double)q);
    for (int wpS : {2, 4, 8}) {
      int blocks = cus * wpS, threads = 256, iters = 2000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float best = 1e30f;
      for (int r = 0; r < 4; r++) {
        hipEventRecord(e0);
        k12<<<blocks, threads>>>(p, c, t, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
      }
      double bf = (double)blocks * threads * 4 * iters;
      printf("   wpS=%d %8.3f ms  %8.1f Gbfly/s  %6.1f cyc/bfly-wave/SIMD(nominal clk)\n", wpS, best, bf / best * 1e-6,
             best * 1e-3 * clk / ((double)iters * 4 * wpS));
    }
  }

  for (int v : {14, 15, 16, 17, 18, 19}) {   // round 3: five-multiply (v14, v15) and seven-multiply (v16) butterflies
    const u64 qq = v == 14 ? 0x1ffffff900000001ull : q;           // v14 needs q = 1 (mod 2^32)
    const u64 ww = w % qq;
    u64 a, b;
    if (v == 15 || v == 17 || v == 18) { a = ww; b = (u64)((((u128)ww) << 32) % qq); }
    else { a = (u64)((((u128)ww) << 32) % qq); b = (u64)((((u128)a) << 32) % qq); }
    TS t{(u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32)};
    u32 qi = 1; for (int i = 0; i < 5; i++) qi *= 2u - (u32)qq * qi;             // q^-1 mod 2^32
    const u64 delta = (1ull << 61) - qq;
    QS cs{qq, 3 * qq + 1, (u64)0 - 4 * qq, (u32)(2 * delta), (u32)delta, 30u, 0x3fffffffu, 0u - qi,
          (u32)(0u - (u32)(qq >> 32)), (u32)qq, (u32)(qq >> 32)};
    hipMemcpy(p, h.data(), 512 * 8, hipMemcpyHostToDevice);
    int it = 6;
    if (v == 14) ks<14><<<1, 64>>>(p, cs, t, it); else if (v == 15) ks<15><<<1, 64>>>(p, cs, t, it); else if (v == 16) ks<16><<<1, 64>>>(p, cs, t, it); else if (v == 17) ks<17><<<1, 64>>>(p, cs, t, it); else if (v == 18) ks<18><<<1, 64>>>(p, cs, t, it); else ks<19><<<1, 64>>>(p, cs, t, it);
    hipMemcpy(o.data(), p, 512 * 8, hipMemcpyDeviceToHost);
    int bad = 0; u64 mx = 0;
    for (int tt = 0; tt < 64; tt++) for (int j = 0; j < 4; j++) {
      u64 yo, xo = href(h[tt + 64 * j] % qq, h[tt + 64 * j + 256] % qq, ww, qq, it, &yo);
      u64 gx = o[tt + 64 * j], gy = o[tt + 64 * j + 256];
      if (gx % qq != xo || gy % qq != yo) bad++;
      if (gx > mx) mx = gx; if (gy > mx) mx = gy;
    }
    const char* nm = v == 14 ? "v14 split+Montgomery q=1 mod 2^32 (5 mul)" : v == 15 ? "v15 split+pseudo-Mersenne q61 (5 mul)" : v == 16 ? "v16 split+word Montgomery any q (7 mul)" : v == 17 ? "v17 v15 in one asm statement" : v == 18 ? "v18 v17 with a borrow-chain subtraction (13 instr)" : "v19 v16 in one asm statement (any odd q, 7 mul, 14 instr)";
    printf("%-20s correctness: %s (max value / q = %.3f)\n", nm, bad ? "FAIL" : "ok", (double)mx / (double)qq);
    for (int wpS : {2, 4, 8}) {
      int blocks = cus * wpS, threads = 256, iters = 2000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float best = 1e30f;
      for (int r = 0; r < 4; r++) {
        hipEventRecord(e0);
        if (v == 14) ks<14><<<blocks, threads>>>(p, cs, t, iters); else if (v == 15) ks<15><<<blocks, threads>>>(p, cs, t, iters); else if (v == 16) ks<16><<<blocks, threads>>>(p, cs, t, iters); else if (v == 17) ks<17><<<blocks, threads>>>(p, cs, t, iters); else if (v == 18) ks<18><<<blocks, threads>>>(p, cs, t, iters); else ks<19><<<blocks, threads>>>(p, cs, t, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
      }
      double bf = (double)blocks * threads * 4 * iters;
      printf("   wpS=%d %8.3f ms  %8.1f Gbfly/s  %6.1f cyc/bfly-wave/SIMD(nominal clk)\n", wpS, best, bf / best * 1e-6,
             best * 1e-3 * clk / ((double)iters * 4 * wpS));
    }
  }
  {  // v13
    const u32 pr = 0x0a3c8001u, w32 = 123456789u % pr, wp32 = (u32)(((u64)w32 << 32) / pr), bq = 0xffffffffu / pr;
    u32* p32 = (u32*)p;
    std::vector<u32> h32(1024), o32(1024);
    for (int i = 0; i < 1024; i++) h32[i] = (u32)((0x9E3779B97F4A7C15ull * (i + 1)) >> 20);
    hipMemcpy(p32, h32.data(), 1024 * 4, hipMemcpyHostToDevice);
    int it = 16;
    k13<<<1, 64>>>(p32, pr, bq, w32, wp32, it);
    hipMemcpy(o32.data(), p32, 1024 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int tt = 0; tt < 64; tt++) for (int j = 0; j < 8; j++) {
      u64 yo, xo = href(h32[tt + 64 * j] % pr, h32[tt + 64 * j + 512] % pr, w32, pr, it, &yo);
      if (o32[tt + 64 * j] % pr != xo || o32[tt + 64 * j + 512] % pr != yo) bad++;
    }
    printf("%-20s correctness: %s\n", "v13 32-bit, 27-bit p", bad ? "FAIL" : "ok");
    for (int wpS : {2, 4, 8}) {
      int blocks = cus * wpS, threads = 256, iters = 4000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float best = 1e30f;
      for (int r = 0; r < 4; r++) {
        hipEventRecord(e0);
        k13<<<blocks, threads>>>(p32, pr, bq, w32, wp32, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
      }
      double bf = (double)blocks * threads * 8 * iters;
      printf("   wpS=%d %8.3f ms  %8.1f Gbfly/s  %6.1f cyc/bfly-wave/SIMD(nominal clk)\n", wpS, best, bf / best * 1e-6,
             best * 1e-3 * clk / ((double)iters * 8 * wpS));
    }
  }
  return 0;
}
