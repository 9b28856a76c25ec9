/*
 * oracle/fhe_next_oracle.c — CPU restatement of the reference's two callers of
 * polynomial products that sit either side of the NTT path:
 *   N1  BFV tensor + relinearise   bfv/src/lib.rs:59-90,251-271
 *       (schoolbook over Z with i64 truncation, f64 scale-and-round, fold mod q)
 *   N2  TFHE Tn x Tn and TGGSW x TGLWE external product
 *       arith/src/ring_torus.rs:266-298, tfhe/src/tggsw.rs:45-62,139-149
 *
 * TEST INFRASTRUCTURE ONLY (same rules as ntt_oracle.c).  These are O(n^2)
 * schoolbook loops exactly like the reference's; the HIP path computes the same
 * words through multi-prime NTTs + CRT.  Pinned by the reference's literal KATs
 * arith/src/ring_n.rs:453-470 (naive_mul + fold, n = 2) and by its property tests
 * restated in tests/ (decrypt-free: word parity against this file).
 *
 * Build note: compiled with -ffp-contract=off; the f64 steps must be plain IEEE
 * double operations, one rounding each, as in the Rust.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef __int128 i128;

/* ---- N1: arith::ring_n ---------------------------------------------------- */

/* arith/src/ring_n.rs:307-320 naive_mul: (2n-1)-term linear convolution, i128
 * accumulate, `*c as i64` truncation (wraps mod 2^64). */
void oracle_r_naive_mul(uint64_t n, const int64_t *a, const int64_t *b, int64_t *out) {
    uint64_t len = 2 * n - 1;
    i128 *res = (i128 *)calloc(len, sizeof(i128));
    for (uint64_t i = 0; i < n; i++)
        for (uint64_t j = 0; j < n; j++) res[i + j] += (i128)a[i] * (i128)b[j];
    for (uint64_t i = 0; i < len; i++) out[i] = (int64_t)(uint64_t)(u128)res[i];
    free(res);
}

/* arith/src/ring_n.rs:142-151 modulus (X^n+1 fold on Vec<i64>, wrapping in --release) */
void oracle_r_modulus(uint64_t n, const int64_t *p, uint64_t len, int64_t *out) {
    for (uint64_t i = 0; i < n; i++) out[i] = i < len ? p[i] : 0;
    for (uint64_t i = n; i < len; i++)
        out[i - n] = (int64_t)((uint64_t)out[i - n] - (uint64_t)p[i]);
}

/* Rust `f64 as i64`: saturating, NaN -> 0 */
static int64_t f64_as_i64(double x) {
    if (x != x) return 0;
    if (x >= 9223372036854775808.0) return INT64_MAX;
    if (x <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)x;
}

/* arith/src/zq.rs:32-39 Zq::from_f64 (+ from_u64 :21-31) */
static uint64_t zq_from_f64(uint64_t q, double e_f) {
    int64_t e = f64_as_i64(round(e_f));
    int64_t qi = (int64_t)q;
    if (e < 0 || e >= qi) {
        uint64_t v = (uint64_t)(((e % qi) + qi) % qi);
        if (v >= q) v = (v % q + q) % q; /* modulus_u64, zq.rs:12-14: never taken here */
        return v;
    }
    return (uint64_t)e;
}
uint64_t oracle_zq_from_f64(uint64_t q, double e) { return zq_from_f64(q, e); }

/* arith/src/ring_n.rs:130-138 mul_div_round(q, n, v, num, den) -> Rq:
 *   r[i] = ((num as f64 * v[i] as f64) / den as f64).round()
 * then Rq::from_vec_f64 (ring_nq.rs:160-163): Zq::from_f64 per element and the
 * X^n+1 fold of ring_nq.rs:132-141 (p[i-n] = p[i-n] - p[i] in Z_q). */
void oracle_mul_div_round(uint64_t q, uint64_t n, const int64_t *v, uint64_t len, uint64_t num,
                          uint64_t den, uint64_t *out) {
    uint64_t *z = (uint64_t *)malloc(len * sizeof(uint64_t));
    for (uint64_t i = 0; i < len; i++) {
        double r = round(((double)num * (double)v[i]) / (double)den);
        z[i] = zq_from_f64(q, r);
    }
    if (len < n) { /* modulus() returns early: the vector keeps its length; callers never do this */
        memcpy(out, z, len * sizeof(uint64_t));
        free(z);
        return;
    }
    for (uint64_t i = 0; i < n; i++) out[i] = z[i];
    for (uint64_t i = n; i < len; i++) {
        uint64_t a = out[i - n], b = z[i];
        out[i - n] = a >= b ? a - b : (q + a) - b; /* Zq::sub, zq.rs:259-276 */
    }
    free(z);
}

static void rq_add(uint64_t q, uint64_t n, const uint64_t *a, const uint64_t *b, uint64_t *c) {
    for (uint64_t i = 0; i < n; i++) { /* Zq::add, zq.rs:219-231 */
        uint64_t v = a[i] + b[i];
        if (v >= q) v -= q;
        c[i] = v;
    }
}

/* bfv/src/lib.rs:59-85 RLWE::tensor(t, a, b) -> (c0, c1, c2); a = (a0,a1), b = (b0,b1)
 * are Rq mod q read as R (to_r: v as i64, ring_n.rs:72-79). */
void oracle_bfv_tensor(uint64_t q, uint64_t n, uint64_t t, const uint64_t *a0, const uint64_t *a1,
                       const uint64_t *b0, const uint64_t *b1, uint64_t *c0, uint64_t *c1,
                       uint64_t *c2) {
    uint64_t len = 2 * n - 1;
    int64_t *x = (int64_t *)malloc(len * sizeof(int64_t));
    int64_t *y = (int64_t *)malloc(len * sizeof(int64_t));
    oracle_r_naive_mul(n, (const int64_t *)a0, (const int64_t *)b0, x);
    oracle_mul_div_round(q, n, x, len, t, q, c0);
    oracle_r_naive_mul(n, (const int64_t *)a0, (const int64_t *)b1, x);
    oracle_r_naive_mul(n, (const int64_t *)a1, (const int64_t *)b0, y);
    for (uint64_t i = 0; i < len; i++) x[i] = (int64_t)((uint64_t)x[i] + (uint64_t)y[i]); /* l + r, :76 */
    oracle_mul_div_round(q, n, x, len, t, q, c1);
    oracle_r_naive_mul(n, (const int64_t *)a1, (const int64_t *)b1, x);
    oracle_mul_div_round(q, n, x, len, t, q, c2);
    free(x);
    free(y);
}

/* bfv/src/lib.rs:251-271 relinearize_204(rlk, c0, c1, c2) -> (c0 + r0, c1 + r1);
 * rlk = (rlk0, rlk1) are Rq mod p*q, p = pq / q. */
void oracle_bfv_relinearize_204(uint64_t q, uint64_t n, uint64_t pq, const uint64_t *rlk0,
                                const uint64_t *rlk1, const uint64_t *c0, const uint64_t *c1,
                                const uint64_t *c2, uint64_t *o0, uint64_t *o1) {
    uint64_t len = 2 * n - 1, p = pq / q;
    int64_t *x = (int64_t *)malloc(len * sizeof(int64_t));
    uint64_t *r = (uint64_t *)malloc(n * sizeof(uint64_t));
    oracle_r_naive_mul(n, (const int64_t *)c2, (const int64_t *)rlk0, x);
    oracle_mul_div_round(q, n, x, len, 1, p, r);
    rq_add(q, n, c0, r, o0);
    oracle_r_naive_mul(n, (const int64_t *)c2, (const int64_t *)rlk1, x);
    oracle_mul_div_round(q, n, x, len, 1, p, r);
    rq_add(q, n, c1, r, o1);
    free(x);
    free(r);
}

/* bfv/src/lib.rs:87-90 RLWE::mul = relinearize_204(tensor(..)) */
void oracle_bfv_mul(uint64_t q, uint64_t n, uint64_t t, uint64_t pq, const uint64_t *rlk0,
                    const uint64_t *rlk1, const uint64_t *a0, const uint64_t *a1, const uint64_t *b0,
                    const uint64_t *b1, uint64_t *o0, uint64_t *o1) {
    uint64_t *c0 = (uint64_t *)malloc(3 * n * sizeof(uint64_t)), *c1 = c0 + n, *c2 = c1 + n;
    oracle_bfv_tensor(q, n, t, a0, a1, b0, b1, c0, c1, c2);
    oracle_bfv_relinearize_204(q, n, pq, rlk0, rlk1, c0, c1, c2, o0, o1);
    free(c0);
}

/* ---- N2: arith::ring_torus / tfhe ------------------------------------------ */

/* arith/src/ring_torus.rs:266-298 naive_poly_mul: u128 accumulate (wrapping in
 * --release), X^n+1 fold with wrapping_sub, low 64 bits kept. */
void oracle_tn_mul(uint64_t n, const uint64_t *a, const uint64_t *b, uint64_t *out) {
    uint64_t len = 2 * n - 1;
    u128 *res = (u128 *)calloc(len, sizeof(u128));
    for (uint64_t i = 0; i < n; i++)
        for (uint64_t j = 0; j < n; j++) res[i + j] += (u128)a[i] * (u128)b[j];
    for (uint64_t i = n; i < len; i++) res[i - n] -= res[i];
    for (uint64_t i = 0; i < n; i++) out[i] = (uint64_t)res[i];
    free(res);
}

/* arith/src/torus.rs:43-52 T64::decompose(beta = 2, l): bits l-1 .. 0, most significant first;
 * arith/src/ring_torus.rs:67-77 Tn::decompose: out[d][j] = d-th digit of coefficient j. */
void oracle_tn_decompose(uint64_t n, uint32_t l, const uint64_t *a, uint64_t *out /* l x n */) {
    for (uint32_t d = 0; d < l; d++)
        for (uint64_t j = 0; j < n; j++) out[(uint64_t)d * n + j] = (a[j] >> (l - 1 - d)) & 1ull;
}

/* tfhe/src/tggsw.rs:45-62 TGGSW * TGLWE (beta = 2, l = 64 hard-coded there; l is a
 * parameter here), :139-149 TGLev * Vec<Tn>, tfhe/src/tglwe.rs:182-194 TGLWE * Tn.
 * Layouts (u64, row-major):
 *   tggsw [(k+1)][l][(k+1)][n]  — TGLev i (i < k: the `a` rows, i = k: the `b` row), level d,
 *                                 TGLWE component c (c < k: mask a_c, c = k: body b)
 *   tglwe [(k+1)][n]            — (a_0..a_{k-1}, b)
 *   out   [(k+1)][n]
 * out[c] = sum_i sum_d tggsw[i][d][c] * decompose(tglwe[i])[d]   (all mod 2^64, X^n+1) */
void oracle_external_product(uint64_t n, uint32_t k, uint32_t l, const uint64_t *tggsw,
                             const uint64_t *tglwe, uint64_t *out) {
    uint32_t k1 = k + 1;
    uint64_t *dec = (uint64_t *)malloc((uint64_t)l * n * sizeof(uint64_t));
    uint64_t *prod = (uint64_t *)malloc(n * sizeof(uint64_t));
    memset(out, 0, (uint64_t)k1 * n * sizeof(uint64_t));
    for (uint32_t i = 0; i < k1; i++) {
        oracle_tn_decompose(n, l, tglwe + (uint64_t)i * n, dec);
        for (uint32_t d = 0; d < l; d++)
            for (uint32_t c = 0; c < k1; c++) {
                const uint64_t *row = tggsw + ((((uint64_t)i * l + d) * k1) + c) * n;
                oracle_tn_mul(n, row, dec + (uint64_t)d * n, prod);
                for (uint64_t j = 0; j < n; j++) out[(uint64_t)c * n + j] += prod[j]; /* T64 add wraps */
            }
    }
    free(dec);
    free(prod);
}
