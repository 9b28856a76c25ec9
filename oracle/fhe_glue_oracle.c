/*
 * oracle/fhe_glue_oracle.c — CPU restatement of the reference's batch surfaces and
 * element-wise glue around Rq products:
 *   N3  TR<Rq>.TR<Rq>, TR<Rq> x Rq / GLWE<Rq> x Rq, GLev<Rq> x Vec<Rq>, GLWE::key_switch
 *       arith/src/tuple_ring.rs:117-155, gfhe/src/glwe.rs:126-137,251-280, gfhe/src/glev.rs:68-80
 *   N4  Rq add/sub/neg, mul_by_u64, mod_switch, mul_div_round, decompose
 *       arith/src/ring_nq.rs:67-113,267-292,406-488,551-561, arith/src/zq.rs:134-207,219-337
 *
 * TEST INFRASTRUCTURE ONLY (same rules as ntt_oracle.c).  Products go through
 * oracle_rq_mul (the NTT restatement), exactly as the reference's `Rq * Rq` does.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* from ntt_oracle.c / fhe_next_oracle.c */
int oracle_roots(uint64_t q, uint64_t n, uint64_t *roots, uint64_t *roots_inv, uint64_t *n_inv,
                 uint64_t *psi_out);
void oracle_rq_mul(uint64_t q, uint64_t n, const uint64_t *roots, const uint64_t *roots_inv,
                   uint64_t n_inv, const uint64_t *a, const uint64_t *b, uint64_t *c, uint64_t *c_evals,
                   uint64_t *a_evals, uint64_t *b_evals);
uint64_t oracle_zq_from_f64(uint64_t q, double e);

/* ---- Zq scalars ------------------------------------------------------------ */
static inline uint64_t zadd(uint64_t q, uint64_t a, uint64_t b) { uint64_t v = a + b; return v >= q ? v - q : v; } /* zq.rs:219-231 */
static inline uint64_t zsub(uint64_t q, uint64_t a, uint64_t b) { return a >= b ? a - b : (q + a) - b; }       /* zq.rs:259-276 */
static inline uint64_t zneg(uint64_t q, uint64_t a) { return a == 0 ? 0 : q - a; }                                /* zq.rs:301-313 */
static inline uint64_t zmul(uint64_t q, uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % q); }       /* zq.rs:315-328 */
static inline uint64_t zfrom_u64(uint64_t q, uint64_t v) { return v >= q ? (v % q + q) % q : v; }               /* zq.rs:21-31  */

/* Rust `f64 as u64`: saturating, NaN -> 0, negatives -> 0 */
static uint64_t f64_as_u64(double x) {
    if (x != x || x <= 0.0) return 0;
    if (x >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)x;
}

/* ---- N4: element-wise on polynomials of n coefficients ------------------------ */
void oracle_rq_add(uint64_t q, uint64_t n, const uint64_t *a, const uint64_t *b, uint64_t *c) { for (uint64_t i = 0; i < n; i++) c[i] = zadd(q, a[i], b[i]); } /* ring_nq.rs:406-431 */
void oracle_rq_sub(uint64_t q, uint64_t n, const uint64_t *a, const uint64_t *b, uint64_t *c) { for (uint64_t i = 0; i < n; i++) c[i] = zsub(q, a[i], b[i]); } /* ring_nq.rs:453-479 */
void oracle_rq_neg(uint64_t q, uint64_t n, const uint64_t *a, uint64_t *c) { for (uint64_t i = 0; i < n; i++) c[i] = zneg(q, a[i]); }                           /* ring_nq.rs:551-561 */
/* Rq::mul_by_u64, ring_nq.rs:274-281 */
void oracle_rq_mul_by_u64(uint64_t q, uint64_t n, const uint64_t *a, uint64_t s, uint64_t *c) {
    uint64_t sq = zfrom_u64(q, s);
    for (uint64_t i = 0; i < n; i++) c[i] = zmul(q, a[i], sq);
}
/* Rq::mod_switch(p) = Zq::mod_switch per coefficient, ring_nq.rs:88-98, zq.rs:134-139 */
void oracle_rq_mod_switch(uint64_t q, uint64_t n, const uint64_t *a, uint64_t p, uint64_t *c) {
    for (uint64_t i = 0; i < n; i++)
        c[i] = zfrom_u64(p, f64_as_u64(round(((double)a[i] * (double)p) / (double)q)));
}
/* Ring::mul_div_round for Rq, ring_nq.rs:100-113 (vector already has length n: no fold) */
void oracle_rq_mul_div_round(uint64_t q, uint64_t n, const uint64_t *a, uint64_t num, uint64_t den,
                             uint64_t *c) {
    for (uint64_t i = 0; i < n; i++)
        c[i] = oracle_zq_from_f64(q, round(((double)num * (double)a[i]) / (double)den));
}
/* Rq::remodule(p), ring_nq.rs:82-88 -> Rq::from_vec_u64 -> Zq::from_u64 */
void oracle_rq_remodule(uint64_t n, const uint64_t *a, uint64_t p, uint64_t *c) {
    for (uint64_t i = 0; i < n; i++) c[i] = zfrom_u64(p, a[i]);
}
/* Rq::mul_by_f64, ring_nq.rs:282-292 */
void oracle_rq_mul_by_f64(uint64_t q, uint64_t n, const uint64_t *a, double s, uint64_t *c) {
    for (uint64_t i = 0; i < n; i++) c[i] = oracle_zq_from_f64(q, (double)a[i] * s);
}
/* Rq::div_round, ring_nq.rs:299-306 (-> from_vec_f64 -> Zq::from_f64; length n: no fold) */
void oracle_rq_div_round(uint64_t q, uint64_t n, const uint64_t *a, uint64_t s, uint64_t *c) {
    for (uint64_t i = 0; i < n; i++) c[i] = oracle_zq_from_f64(q, round((double)a[i] / (double)s));
}
/* Zq::decompose, zq.rs:141-207 (base 2 and base beta), applied per coefficient and transposed:
 * Rq::decompose, ring_nq.rs:67-78.  out[d][j], d < l. */
void oracle_rq_decompose(uint64_t q, uint64_t n, const uint64_t *a, uint32_t beta, uint32_t l,
                         uint64_t *out) {
    for (uint64_t j = 0; j < n; j++) {
        uint64_t v = a[j];
        if (beta == 2) {
            /* zq.rs:176-180 `self.v >= 1 << l as u64`: in --release a shift by >= 64 wraps the amount */
            if (v >= (1ull << (l & 63))) {
                for (uint32_t d = 0; d < l; d++) out[(uint64_t)d * n + j] = 1 % q;
                continue;
            }
            for (uint32_t d = 0; d < l; d++)
                out[(uint64_t)d * n + j] = zfrom_u64(q, (v >> (l - 1 - d)) & 1ull);
        } else {
            uint32_t bl = 1; /* beta.pow(l) as u32 (the reference would overflow-panic in debug) */
            for (uint32_t t = 0; t < l; t++) bl *= beta;
            if (v >= (uint64_t)bl) { /* zq.rs:152-160 */
                for (uint32_t d = 0; d < l; d++) out[(uint64_t)d * n + j] = (uint64_t)beta - 1;
                continue;
            }
            uint64_t rem = v;
            uint32_t bi = 1;
            for (uint32_t d = 0; d < l; d++) {
                bi *= beta;
                uint64_t den = q / (uint64_t)bi;
                uint64_t x = rem / den;
                out[(uint64_t)d * n + j] = zfrom_u64(q, x);
                if (x != 0) rem = rem % den;
            }
        }
    }
}

/* ---- N3: batch surfaces ---------------------------------------------------------- */
typedef struct { uint64_t q, n, n_inv; uint64_t *r, *ri; } tabs;
static int tabs_init(tabs *t, uint64_t q, uint64_t n) {
    t->q = q; t->n = n;
    t->r = (uint64_t *)malloc(n * 8); t->ri = (uint64_t *)malloc(n * 8);
    return oracle_roots(q, n, t->r, t->ri, &t->n_inv, NULL);
}
static void tabs_free(tabs *t) { free(t->r); free(t->ri); }
static void rmul(const tabs *t, const uint64_t *a, const uint64_t *b, uint64_t *c) {
    oracle_rq_mul(t->q, t->n, t->r, t->ri, t->n_inv, a, b, c, NULL, NULL, NULL);
}

/* TR . TR, tuple_ring.rs:117-134: sum_i a[i] * b[i]  (Rq products, then Zq sums in order) */
int oracle_tr_dot(uint64_t q, uint64_t n, uint32_t k, const uint64_t *a, const uint64_t *b, uint64_t *c) {
    tabs t; if (tabs_init(&t, q, n)) { tabs_free(&t); return -1; }
    uint64_t *p = (uint64_t *)malloc(n * 8);
    for (uint32_t i = 0; i < k; i++) {
        rmul(&t, a + (uint64_t)i * n, b + (uint64_t)i * n, p);
        if (i == 0) memcpy(c, p, n * 8); else oracle_rq_add(q, n, c, p, c);
    }
    free(p); tabs_free(&t);
    return 0;
}
/* TR x R (tuple_ring.rs:137-155) and GLWE x R (glwe.rs:263-280): out[i] = a[i] * p, i < rows */
int oracle_tr_mul_r(uint64_t q, uint64_t n, uint32_t rows, const uint64_t *a, const uint64_t *p, uint64_t *out) {
    tabs t; if (tabs_init(&t, q, n)) { tabs_free(&t); return -1; }
    for (uint32_t i = 0; i < rows; i++) rmul(&t, a + (uint64_t)i * n, p, out + (uint64_t)i * n);
    tabs_free(&t);
    return 0;
}
/* GLev x Vec<R> -> GLWE, glev.rs:68-80: out[c] = sum_d glev[d][c] * v[d]; glev [l][k+1][n], v [l][n] */
int oracle_glev_mul(uint64_t q, uint64_t n, uint32_t k, uint32_t l, const uint64_t *glev, const uint64_t *v,
                    uint64_t *out) {
    tabs t; if (tabs_init(&t, q, n)) { tabs_free(&t); return -1; }
    uint32_t k1 = k + 1;
    uint64_t *p = (uint64_t *)malloc(n * 8);
    for (uint32_t d = 0; d < l; d++)
        for (uint32_t c = 0; c < k1; c++) {
            rmul(&t, glev + ((uint64_t)d * k1 + c) * n, v + (uint64_t)d * n, p);
            uint64_t *o = out + (uint64_t)c * n;
            if (d == 0) memcpy(o, p, n * 8); else oracle_rq_add(q, n, o, p, o);
        }
    free(p); tabs_free(&t);
    return 0;
}
/* GLWE::key_switch, glwe.rs:126-137: (0, b) - sum_i ksk[i] * decompose(a[i]);
 * glwe [(k+1)][n] = (a_0..a_{k-1}, b); ksk [k][l][(k+1)][n]; out [(k+1)][n] */
int oracle_key_switch(uint64_t q, uint64_t n, uint32_t k, uint32_t beta, uint32_t l, const uint64_t *glwe,
                      const uint64_t *ksk, uint64_t *out) {
    uint32_t k1 = k + 1;
    uint64_t *dec = (uint64_t *)malloc((uint64_t)l * n * 8);
    uint64_t *part = (uint64_t *)malloc((uint64_t)k1 * n * 8);
    uint64_t *rhs = (uint64_t *)calloc((uint64_t)k1 * n, 8);
    int rc = 0;
    for (uint32_t i = 0; i < k && !rc; i++) {
        oracle_rq_decompose(q, n, glwe + (uint64_t)i * n, beta, l, dec);
        rc = oracle_glev_mul(q, n, k, l, ksk + (uint64_t)i * l * k1 * n, dec, part);
        if (i == 0) memcpy(rhs, part, (uint64_t)k1 * n * 8);
        else for (uint32_t c = 0; c < k1; c++) oracle_rq_add(q, n, rhs + (uint64_t)c * n, part + (uint64_t)c * n, rhs + (uint64_t)c * n);
    }
    for (uint32_t c = 0; c < k1 && !rc; c++) {
        uint64_t *o = out + (uint64_t)c * n;
        if (c < k) { for (uint64_t j = 0; j < n; j++) o[j] = zsub(q, 0, rhs[(uint64_t)c * n + j]); }   /* TR::zero - rhs */
        else oracle_rq_sub(q, n, glwe + (uint64_t)k * n, rhs + (uint64_t)c * n, o);
    }
    free(dec); free(part); free(rhs);
    return rc;
}
