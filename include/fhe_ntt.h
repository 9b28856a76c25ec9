/*
 * fhe_ntt.h — C ABI of libfhe_ntt.so: the MI355X (gfx950) negacyclic-NTT engine
 * for R_q = Z_q[X]/(X^N+1).
 *
 * This is the drop-in boundary for the hot path of arnaucube/fhe-study's
 * `arith` crate.  The reference has no FFI of its own (it is 100 % safe Rust);
 * the seam is the Rust API `arith::NTT::{ntt,intt}` and `ring_nq::{mul,mul_mut}`.
 * Each entry point below names the reference item it replaces (paths relative
 * to the reference root).  INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions
 *   - Polynomials are `batch × n` row-major arrays of uint64_t coefficient
 *     VALUES (the `v` of `Zq{q,v}`, arith/src/zq.rs:6-10), canonical: v < q.
 *   - The caller owns every buffer; `out` may alias `in` (in-place).
 *   - The arithmetic behind an entry point is the library's choice and never
 *     shows in the words: moduli up to 2^62 run on 64-bit Shoup butterflies (a
 *     pseudo-Mersenne one on five-multiply butterflies: fhe_ntt_plan_arithmetic;
 *     2^62 <= q < 2^63 on strict ones, every value canonical, in the same kernels); a
 *     modulus below 2^30 (e.g. the reference's test moduli 65537, 12289)
 *     and the keyed products whose integers are small run in 32-bit words on
 *     the same tables (env FHE_EXT32=0 disables that; results are identical).
 *   - No function unwinds or aborts: every failure is a negative FHE_E_* code
 *     and a message retrievable with fhe_last_error() (thread-local).  The
 *     reference panics in the same situations; the Rust shim turns a non-zero
 *     return into `panic!`.
 *   - All functions are thread-safe and may be called concurrently, on the same
 *     or different plans, streams and threads.  Entry points that need
 *     intermediates use a library-owned workspace keyed by (device, stream,
 *     calling thread): a thread's calls on a stream are ordered by the stream,
 *     and no two threads or streams ever share a buffer — two host threads
 *     enqueueing on the SAME stream (the NULL default stream included) are
 *     fine.  Plans are immutable and owned by the library until
 *     fhe_ntt_shutdown().
 *   - `*_dev` variants take DEVICE pointers of the current HIP device and a
 *     `hipStream_t` passed as `void*` (NULL = the default stream); they only
 *     enqueue work and never synchronise — except the first use of a plan on a
 *     device, which uploads its tables (call fhe_ntt_plan_prepare() beforehand
 *     to keep that out of, e.g., a stream capture; the products of rows N1 - N3
 *     — BFV, TGGSW x TGLWE, key switching — upload the tables of their internal
 *     primes on the first call per (n, device): run one call before capturing),
 *     growth of a library workspace, and the opt-in FHE_NTT_CHECK_CANONICAL mode.  Device buffers must be 16-byte
 *     aligned (FHE_E_INVALID otherwise; anything hipMalloc returns is).  Host-pointer variants copy
 *     host→device→host around the same kernels and return when `out` is valid.
 *   - There is no CPU fallback: without a HIP device every compute entry point
 *     returns FHE_E_NO_DEVICE.
 */
#ifndef FHE_NTT_H
#define FHE_NTT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes -------------------------------------------------------- */
#define FHE_OK 0
/* n is not a power of two (assert, arith/src/ntt.rs:116), n < 2 (degenerate
 * in the reference, ntt.rs:139) or n > 2^20 (engine limit). */
#define FHE_E_BAD_N (-1)
/* (q-1) % 2n != 0 (assert, ntt.rs:117), q < 3, or q >= 2^63 (where the
 * reference's own Zq::add overflows, zq.rs:225). */
#define FHE_E_BAD_Q (-2)
/* the k = 1,2,... search found no primitive 2n-th root (panic, ntt.rs:130). */
#define FHE_E_NO_ROOT (-3)
/* a NULL pointer where a buffer / plan is required. */
#define FHE_E_NULL (-4)
/* a HIP runtime call or kernel launch failed (message has the HIP error). */
#define FHE_E_HIP (-5)
/* no usable HIP device in this process. */
#define FHE_E_NO_DEVICE (-6)
/* operands of a ring multiply have different (q,n)
 * (assert_eq!(lhs.param, rhs.param), arith/src/ring_nq.rs:565,587). */
#define FHE_E_PARAM_MISMATCH (-7)
/* a coefficient >= q was found by fhe_rq_check_canonical(). */
#define FHE_E_NOT_CANONICAL (-8)
/* invalid argument other than the above (e.g. bad flag, count overflow). */
#define FHE_E_INVALID (-9)

typedef struct fhe_ntt_plan fhe_ntt_plan; /* opaque */

/* ---- plan cache: replaces `roots(q,n)` + CACHE, arith/src/ntt.rs:18-38 ---
 * Memoised per (q,n).  Derives psi by the reference's k=1,2,.. search
 * (ntt.rs:115-131), roots[i] = psi^bitrev(i) (ntt.rs:133-147), roots_inv[i] =
 * roots[i]^(q-2) (ntt.rs:149-161, the same Fermat power, so the tables agree
 * with the reference even for a composite q) and n_inv (ntt.rs:27-30).
 * Host-only: needs no GPU. */
int fhe_ntt_plan_get(uint64_t q, uint64_t n, const fhe_ntt_plan **out);
int fhe_ntt_plan_info(const fhe_ntt_plan *plan, uint64_t *q, uint64_t *n, uint64_t *psi,
                      uint64_t *n_inv);
/* copies the n-entry tables (either pointer may be NULL). */
int fhe_ntt_plan_tables(const fhe_ntt_plan *plan, uint64_t *roots, uint64_t *roots_inv);
/* Uploads the plan's twiddle tables to the CURRENT device now (blocking) instead
 * of at the first transform there — the device-side half of CACHE's one-off
 * table build (ntt.rs:20-38). */
int fhe_ntt_plan_prepare(const fhe_ntt_plan *plan);
/* Which arithmetic the transform kernels run for this plan's modulus — Zq::mul (arith/src/zq.rs:315-328) is generic
 * over q; the engine picks the cheapest exact form at plan time.  Host-only, never fails for a valid plan:
 *   FHE_ARITH_SHOUP62   q < 2^62: Shoup product (10 32-bit multiplies), values in [0,4q) between stages
 *   FHE_ARITH_SHOUP61   q < 2^61: the same with compile-time value bounds (a correction every other stage)
 *   FHE_ARITH_PMERSENNE q = 2^k - delta, 56 <= k <= 61, delta <= 2^(k-39) (2^61 - 2^21 + 1 is one): split
 *                       multiplicand, no quotient, 5 multiplies (FHE_PM=0 in the environment selects SHOUP61 instead)
 *   FHE_ARITH_WORD32    q < 2^30 and 2^8 <= n <= 2^18: one 32-bit word per coefficient (FHE_EXT32=0: SHOUP61); without
 *                       any conditional subtraction in the forward transform below 2^32/25, Harvey's form above.
 *                       Moduli between 2^30 and 2^32, and n outside that range, are NOT covered by this form.
 *   FHE_ARITH_STRICT63  2^62 <= q < 2^63 (the top of the reference's range, zq.rs:225): 4q no longer fits a word, so
 *                       every value is canonical between stages (three conditional subtractions per butterfly, ~33
 *                       instructions against 13 - 21) in the same two-pass / fused kernels: about 0.6 x the Shoup rate
 *   FHE_ARITH_MONTGOMERY q = 1 (mod 2^32), q < 2^61, n >= 16: transforms and Rq x Rq run word-Montgomery butterflies on
 *                       {w 2^32, w 2^64 mod q} — q^-1 = 1 (mod 2^32), so a word step needs no multiplication by it: 5
 *                       multiplies; only the inverse transform that multiplies two evaluation operands in its load keeps
 *                       SHOUP61's kernel (FHE_MG=0: everything does)
 * Results are the same words in every case.  Returns a negative FHE_E_* for a NULL plan. */
#define FHE_ARITH_SHOUP62 0
#define FHE_ARITH_SHOUP61 1
#define FHE_ARITH_PMERSENNE 2
#define FHE_ARITH_WORD32 3
#define FHE_ARITH_STRICT63 4
#define FHE_ARITH_MONTGOMERY 5
int fhe_ntt_plan_arithmetic(const fhe_ntt_plan *plan);

/* ---- transforms: host buffers ------------------------------------------- */
/* NTT::ntt(&Rq)->Rq, arith/src/ntt.rs:44-73: natural order in, bit-reversed
 * order out, canonical. */
int fhe_ntt_forward(const fhe_ntt_plan *plan, const uint64_t *in, uint64_t *out, size_t batch);
/* NTT::intt(&Rq)->Rq, arith/src/ntt.rs:78-110 (includes the n_inv scaling). */
int fhe_ntt_inverse(const fhe_ntt_plan *plan, const uint64_t *in, uint64_t *out, size_t batch);

/* Rq x Rq, arith/src/ring_nq.rs:564-607 (`mul`, `mul_mut`, the `Mul` impls
 * :490-503 and Rq::mul :294-296).
 *   a, b           operands, batch × n each.
 *   a_is_evals /   non-zero: that operand pointer already holds NTT-domain
 *   b_is_evals     values (the reference's cached `evals`, ring_nq.rs:590-599).
 *   c              product coefficients (required).
 *   c_evals        nullable: pointwise product in the NTT domain — the `evals`
 *                  the reference attaches to the product (ring_nq.rs:606).
 *   a_evals_out /  nullable: NTT(a), NTT(b) — what `mul_mut` stores back into
 *   b_evals_out    its operands (ring_nq.rs:568-573). */
int fhe_rq_mul(const fhe_ntt_plan *plan, const uint64_t *a, int a_is_evals, const uint64_t *b,
               int b_is_evals, uint64_t *c, uint64_t *c_evals, uint64_t *a_evals_out,
               uint64_t *b_evals_out, size_t batch);
/* same with two plans, as the reference compares `param` of both operands;
 * returns FHE_E_PARAM_MISMATCH unless they are the same (q,n). */
int fhe_rq_mul_checked(const fhe_ntt_plan *plan_a, const fhe_ntt_plan *plan_b, const uint64_t *a,
                       const uint64_t *b, uint64_t *c, uint64_t *c_evals, size_t batch);
/* zip_eq(l,r).map(l*r), arith/src/ring_nq.rs:601-604: c[i] = a[i]*b[i] mod q */
int fhe_rq_pointwise_mul(const fhe_ntt_plan *plan, const uint64_t *a, const uint64_t *b,
                         uint64_t *c, size_t batch);
/* FHE_OK, or FHE_E_NOT_CANONICAL if any of the batch*n values is >= q. */
int fhe_rq_check_canonical(const fhe_ntt_plan *plan, const uint64_t *x, size_t batch);
/* Opt-in validation: with FHE_NTT_CHECK_CANONICAL=1 in the environment, or after
 * fhe_ntt_set_check_canonical(1), fhe_ntt_forward/inverse, fhe_rq_mul and
 * fhe_rq_pointwise_mul (host and _dev forms) scan their coefficient inputs on the
 * device first and return FHE_E_NOT_CANONICAL for a value >= q instead of
 * undefined words (the reference cannot build such a Zq, zq.rs:21-30; a C caller
 * can).  The scan synchronises the stream: a debugging aid, off by default. */
int fhe_ntt_set_check_canonical(int on);

/* ---- transforms: device-resident buffers -------------------------------- */
int fhe_ntt_forward_dev(const fhe_ntt_plan *plan, const void *d_in, void *d_out, size_t batch,
                        void *hip_stream);
int fhe_ntt_inverse_dev(const fhe_ntt_plan *plan, const void *d_in, void *d_out, size_t batch,
                        void *hip_stream);
/* d_work: device scratch of fhe_rq_mul_workspace_bytes(plan,batch) bytes, or
 * NULL to use a library-owned grow-only workspace (not graph-capturable while
 * it grows). */
int fhe_rq_mul_dev(const fhe_ntt_plan *plan, const void *d_a, int a_is_evals, const void *d_b,
                   int b_is_evals, void *d_c, void *d_c_evals, void *d_a_evals_out,
                   void *d_b_evals_out, size_t batch, void *d_work, void *hip_stream);
size_t fhe_rq_mul_workspace_bytes(const fhe_ntt_plan *plan, size_t batch);
int fhe_rq_pointwise_mul_dev(const fhe_ntt_plan *plan, const void *d_a, const void *d_b, void *d_c,
                             size_t batch, void *hip_stream);
/* Synthetic coefficients: x[i] = mulhi64(splitmix64(seed ^
 * (first_index+i)), q), generated on the device. */
int fhe_fill_synthetic_dev(uint64_t q, uint64_t seed, uint64_t first_index, size_t count,
                           void *d_out, void *hip_stream);

/* ---- tuning / measurement ----------------------------------------------- */
/* Polynomials per launch for two-pass sizes (n >= 2^14): the batch can be
 * walked in tiles so that the intermediate of the first pass is still in the
 * 256 MiB Infinity Cache when the second pass reads it.  0 restores the default
 * (one launch per pass over the whole batch — measured fastest). */
int fhe_ntt_set_batch_tile(size_t polys);
/* (The one-launch forward transform of round 4 — persistent workgroups, slower than the two-pass kernels on every
 * setting measured — is NOT part of this boundary: its switches live in include/fhe_ntt_experimental.h.) */
/* When enabled, every kernel launch is bracketed by HIP events on its stream;
 * fhe_ntt_kernel_timing_read() synchronises, and returns per-kernel totals
 * since the last reset.  `names` receives up to `cap` NUL-terminated names of
 * at most 63 chars (64-byte slots), `total_ms`/`launches` the matching sums.
 * Returns the number of distinct kernels recorded (may exceed cap). */
int fhe_ntt_kernel_timing_enable(int on);
int fhe_ntt_kernel_timing_read(char *names, double *total_ms, uint64_t *launches, int cap);
int fhe_ntt_kernel_timing_reset(void);

/* ======================================================================== *
 * Next rows: the reference's SCHOOLBOOK callers either side
 * of the NTT path, rebuilt on the engine.  Products over Z are exact (K <= 3
 * CRT primes of 61 bits, chosen from the operand sizes) and only then reduced
 * mod 2^64, which is what the reference's `as i64` / `as u64` truncation keeps.
 * ======================================================================== */

/* arith::ring_n::naive_mul, arith/src/ring_n.rs:307-320: linear convolution of
 * two length-n coefficient vectors over Z, truncated `as i64`.  Operands are read
 * as NON-NEGATIVE 64-bit integers (Rq::to_r yields [0,q), ring_n.rs:72-79).
 * out: batch x 2n words (the 2n-1 results, then one 0).  a_bits/b_bits: bound on
 * the operands' bit length (0 = 64), used to pick the number of primes. */
int fhe_r_naive_mul(uint64_t n, const int64_t *a, const int64_t *b, int64_t *out, size_t batch);
int fhe_r_naive_mul_dev(uint64_t n, const void *d_a, const void *d_b, void *d_out, size_t batch,
                        unsigned a_bits, unsigned b_bits, void *hip_stream);

/* arith::ring_n::mul_div_round(q, n, v, num, den) -> Rq, arith/src/ring_n.rs:130-138:
 * z[i] = Zq::from_f64(((num as f64 * v[i] as f64) / den as f64).round())  (zq.rs:32-39),
 * then the X^n+1 fold of ring_nq.rs:132-141.  d_v: batch x 2n words read as i64
 * (layout of fhe_r_naive_mul_dev); d_out: batch x n words mod q.  IEEE f64, one
 * rounding per operation, round half away from zero — as the Rust. */
int fhe_mul_div_round_dev(uint64_t q, uint64_t n, const void *d_v, uint64_t num, uint64_t den,
                          void *d_out, size_t batch, void *hip_stream);

/* RLWE::tensor(t, a, b) -> (c0, c1, c2), bfv/src/lib.rs:59-85.
 * ab = [a0 | a1 | b0 | b1], each batch x n words mod q; c = [c0 | c1 | c2]. */
int fhe_bfv_tensor(uint64_t q, uint64_t n, uint64_t t, const uint64_t *ab, uint64_t *c, size_t batch);
int fhe_bfv_tensor_dev(uint64_t q, uint64_t n, uint64_t t, const void *d_ab, void *d_c, size_t batch,
                       void *hip_stream);
/* BFV::relinearize_204(rlk, c0, c1, c2) -> (c0 + r0, c1 + r1), bfv/src/lib.rs:251-271.
 * rlk = [rlk0 | rlk1], n words each mod pq (one key for the whole batch), p = pq / q;
 * c as produced by fhe_bfv_tensor_dev; out = [o0 | o1], each batch x n. */
int fhe_bfv_relinearize_dev(uint64_t q, uint64_t n, uint64_t pq, const void *d_rlk, const void *d_c,
                            void *d_out, size_t batch, void *hip_stream);
/* RLWE::mul(t, rlk, a, b) = relinearize_204(tensor(..)), bfv/src/lib.rs:87-90. */
int fhe_bfv_mul(uint64_t q, uint64_t n, uint64_t t, uint64_t pq, const uint64_t *rlk, const uint64_t *ab,
                uint64_t *out, size_t batch);
int fhe_bfv_mul_dev(uint64_t q, uint64_t n, uint64_t t, uint64_t pq, const void *d_rlk, const void *d_ab,
                    void *d_out, size_t batch, void *hip_stream);
/* Resident relinearisation key (RLWE::mul is called with the same `rlk` for every product of a computation,
 * bfv/src/lib.rs:87-90; relinearize_204 :251-271): rlk in the form the products consume — its transforms modulo three
 * 27-bit primes where the products stay below 2^82 (q = 65537, p = q^2, n <= 8192), else split in two halves and
 * transformed modulo one 61-bit prime where the half-products fit, reduced and transformed per CRT prime otherwise — built
 * once per key (fhe_bfv_rlk_prepared_words words; 0 = invalid parameters; the layout is opaque) and reused by every
 * later call; same words as the plain entry points produce. */
size_t fhe_bfv_rlk_prepared_words(uint64_t q, uint64_t n, uint64_t pq);
int fhe_bfv_rlk_prepare_dev(uint64_t q, uint64_t n, uint64_t pq, const void *d_rlk, void *d_prepared, void *hip_stream);
int fhe_bfv_relinearize_prepared_dev(uint64_t q, uint64_t n, uint64_t pq, const void *d_prepared, const void *d_c,
                                     void *d_out, size_t batch, void *hip_stream);
int fhe_bfv_mul_prepared_dev(uint64_t q, uint64_t n, uint64_t t, uint64_t pq, const void *d_prepared, const void *d_ab,
                             void *d_out, size_t batch, void *hip_stream);

/* Tn x Tn, arith/src/ring_torus.rs:251-298 (naive_poly_mul): negacyclic product of
 * coefficient vectors mod 2^64. */
int fhe_tn_mul(uint64_t n, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t batch);
int fhe_tn_mul_dev(uint64_t n, const void *d_a, const void *d_b, void *d_out, size_t batch,
                   void *hip_stream);

/* TGGSW x TGLWE external product, tfhe/src/tggsw.rs:45-62 with TGLev x Vec<Tn> (:139-149),
 * TGLWE x Tn (tglwe.rs:182-194) and Tn::decompose(beta = 2, l) (ring_torus.rs:67-77,
 * torus.rs:43-52; the reference hard-codes l = 64).
 *   tggsw [(k+1)][l][(k+1)][n]   TGLev i (i < k: the `a` rows; i = k: the `b` row), level d,
 *                                component c (c < k: mask a_c; c = k: body)   — one key
 *   tglwe [batch][(k+1)][n]      (a_0 .. a_{k-1}, b) per ciphertext;  out likewise. */
int fhe_tggsw_external_product(uint64_t n, unsigned k, unsigned l, const uint64_t *tggsw,
                               const uint64_t *tglwe, uint64_t *out, size_t batch);
int fhe_tggsw_external_product_dev(uint64_t n, unsigned k, unsigned l, const void *d_tggsw,
                                   const void *d_tglwe, void *d_out, size_t batch, void *hip_stream);
/* Resident key ("pre-transformed, reusable across the 630 products"): the TGGSW rows in
 * the layout the product consumes (32-bit halves, forward-transformed), built once and used by every
 * later call — the bootstrapping key of a blind rotation serves all its external products.
 *   fhe_tggsw_prepared_words   u64 words d_prepared must hold (0: this shape has no prepared form,
 *                              i.e. (k+1)*l*n > 2^26 or n outside [16, 2^13]: use the plain entry point)
 *   fhe_tggsw_prepare_dev      d_tggsw [(k+1)][l][(k+1)][n]  ->  d_prepared
 *   ..._prepared_dev           the external product against a prepared key; same words as
 *                              fhe_tggsw_external_product_dev on the original key. */
size_t fhe_tggsw_prepared_words(uint64_t n, unsigned k, unsigned l);
int fhe_tggsw_prepare_dev(uint64_t n, unsigned k, unsigned l, const void *d_tggsw, void *d_prepared, void *hip_stream);
int fhe_tggsw_external_product_prepared_dev(uint64_t n, unsigned k, unsigned l, const void *d_prepared,
                                            const void *d_tglwe, void *d_out, size_t batch, void *hip_stream);

/* TGLWE x Tn (plaintext product), tfhe/src/tglwe.rs:182-194: out[b][i] = tglwe[b][i] * p[b] mod
 * (2^64, X^n+1), i <= k.  tglwe, out: [batch][(k+1)][n]; p: [batch][n]. */
int fhe_tglwe_mul_tn(uint64_t n, unsigned k, const uint64_t *tglwe, const uint64_t *p, uint64_t *out, size_t batch);
int fhe_tglwe_mul_tn_dev(uint64_t n, unsigned k, const void *d_tglwe, const void *d_p, void *d_out,
                         size_t batch, void *hip_stream);
/* TGLev x Vec<Tn> -> TGLWE, tfhe/src/tggsw.rs:139-149: out[b] = sum_{d<l} tglev[d] * v[b][d].
 * tglev: [l][(k+1)][n] (one for the batch); v: [batch][l][n] (any 64-bit words, e.g. a decomposition);
 * out: [batch][(k+1)][n]. */
int fhe_tglev_mul(uint64_t n, unsigned k, unsigned l, const uint64_t *tglev, const uint64_t *v, uint64_t *out,
                  size_t batch);
int fhe_tglev_mul_dev(uint64_t n, unsigned k, unsigned l, const void *d_tglev, const void *d_v, void *d_out,
                      size_t batch, void *hip_stream);

/* ---- rows N3 / N4: batch surfaces and element-wise glue, device-resident ----
 * Sums of products are accumulated in the NTT domain and transformed back once; arithmetic
 * mod q is exact, so the words equal the reference's sum of canonical products.
 *
 * `flags` generalise the cached `Rq.evals` (arith/src/ring_nq.rs:24-26,590-599) to the batch
 * surfaces: an operand that is already in the NTT domain (fhe_ntt_forward_dev of it, e.g. a
 * key transformed once) is not transformed again, and a result can be left there for the next
 * product.  Output buffers must not overlap the inputs. */
#define FHE_A_IS_EVALS 1u /* first operand (a / glev / ksk) holds NTT-domain values          */
#define FHE_B_IS_EVALS 2u /* second operand (b / p / v) holds NTT-domain values              */
#define FHE_OUT_EVALS  4u /* leave the result in the NTT domain (no inverse transform)       */

/* TR<Rq> . TR<Rq>, arith/src/tuple_ring.rs:117-134: c[b] = sum_{i<k} a[b][i] * b[b][i].
 * a, b: [batch][k][n]; c: [batch][n]. */
int fhe_tr_dot_dev(const fhe_ntt_plan *plan, const void *d_a, const void *d_b, void *d_c, unsigned k,
                   size_t batch, unsigned flags, void *hip_stream);
/* TR<Rq> x Rq (tuple_ring.rs:137-155) and GLWE<Rq> x Rq (gfhe/src/glwe.rs:263-280):
 * out[b][i] = a[b][i] * p[b], i < rows.  a, out: [batch][rows][n]; p: [batch][n]. */
int fhe_tr_mul_r_dev(const fhe_ntt_plan *plan, const void *d_a, const void *d_p, void *d_out,
                     unsigned rows, size_t batch, unsigned flags, void *hip_stream);
/* GLev<Rq> x Vec<Rq> -> GLWE, gfhe/src/glev.rs:68-80: out[b][c] = sum_{d<l} glev[d][c] * v[b][d].
 * glev: [l][k+1][n] (a key: shared by the batch); v: [batch][l][n]; out: [batch][k+1][n]. */
int fhe_glev_mul_dev(const fhe_ntt_plan *plan, unsigned k, unsigned l, const void *d_glev,
                     const void *d_v, void *d_out, size_t batch, unsigned flags, void *hip_stream);
/* GLWE<Rq>::key_switch, gfhe/src/glwe.rs:126-137: (0, b) - sum_{i<k} ksk[i] * decompose(a_i, beta, l).
 * glwe, out: [batch][k+1][n] as (a_0..a_{k-1}, b); ksk: [k][l][k+1][n] (shared).
 * flags: FHE_A_IS_EVALS only (the key; the ciphertext is decomposed in coefficients). */
int fhe_glwe_key_switch_dev(const fhe_ntt_plan *plan, unsigned k, unsigned beta, unsigned l,
                            const void *d_glwe, const void *d_ksk, void *d_out, size_t batch,
                            unsigned flags, void *hip_stream);
/* Resident key-switching key: the key in the form the product consumes, built once and used by every later call
 * (a key-switching key serves every ciphertext under its secret).  The form is OPAQUE and depends on the shape: with
 * beta = 2, k = 1, 2^8 <= n <= 2^12 and q < 2^61 it holds transforms of the key's 32-bit halves modulo two 27-bit
 * primes (twice the words of the key); otherwise fhe_ntt_forward_dev of the key (what FHE_A_IS_EVALS takes).
 *   fhe_glwe_ksk_prepared_words   u64 words d_prepared must hold (0: invalid arguments)
 *   fhe_glwe_ksk_prepare_dev      d_ksk [k][l][k+1][n] -> d_prepared (distinct buffers)
 *   ..._prepared_dev              the key switch against a prepared key; same words as fhe_glwe_key_switch_dev on
 *                                 the original key.  plan, k, beta, l must be those of the preparation. */
size_t fhe_glwe_ksk_prepared_words(const fhe_ntt_plan *plan, unsigned k, unsigned beta, unsigned l);
int fhe_glwe_ksk_prepare_dev(const fhe_ntt_plan *plan, unsigned k, unsigned beta, unsigned l, const void *d_ksk,
                             void *d_prepared, void *hip_stream);
int fhe_glwe_key_switch_prepared_dev(const fhe_ntt_plan *plan, unsigned k, unsigned beta, unsigned l,
                                     const void *d_glwe, const void *d_prepared, void *d_out, size_t batch,
                                     void *hip_stream);

/* Host-buffer forms of the four surfaces above (same layouts, no flags): what a shim of `gfhe`,
 * whose ciphertexts are host `Vec`s, binds.  They stage through device memory on the calling
 * thread's stream and return when `out` is complete. */
int fhe_tr_dot(const fhe_ntt_plan *plan, const uint64_t *a, const uint64_t *b, uint64_t *c, unsigned k, size_t batch);
int fhe_tr_mul_r(const fhe_ntt_plan *plan, const uint64_t *a, const uint64_t *p, uint64_t *out, unsigned rows, size_t batch);
int fhe_glev_mul(const fhe_ntt_plan *plan, unsigned k, unsigned l, const uint64_t *glev, const uint64_t *v,
                 uint64_t *out, size_t batch);
int fhe_glwe_key_switch(const fhe_ntt_plan *plan, unsigned k, unsigned beta, unsigned l, const uint64_t *glwe,
                        const uint64_t *ksk, uint64_t *out, size_t batch);

/* Rq + Rq, Rq - Rq, -Rq (ring_nq.rs:406-488,551-561), Rq::mul_by_u64 (ring_nq.rs:274-281). */
int fhe_rq_add_dev(const fhe_ntt_plan *plan, const void *d_a, const void *d_b, void *d_c, size_t batch, void *hip_stream);
int fhe_rq_sub_dev(const fhe_ntt_plan *plan, const void *d_a, const void *d_b, void *d_c, size_t batch, void *hip_stream);
int fhe_rq_neg_dev(const fhe_ntt_plan *plan, const void *d_a, void *d_c, size_t batch, void *hip_stream);
int fhe_rq_mul_by_u64_dev(const fhe_ntt_plan *plan, const void *d_a, uint64_t s, void *d_c, size_t batch, void *hip_stream);
/* Rq::mod_switch(p), ring_nq.rs:88-98 / zq.rs:134-139: round(v * p / q) in f64, mod p. */
int fhe_rq_mod_switch_dev(uint64_t q, uint64_t p, const void *d_a, void *d_c, size_t count, void *hip_stream);
/* Ring::mul_div_round for Rq, ring_nq.rs:100-113: Zq::from_f64(round(num * v / den)). */
int fhe_rq_mul_div_round_dev(uint64_t q, uint64_t num, uint64_t den, const void *d_a, void *d_c, size_t count, void *hip_stream);
/* Rq::remodule(p), ring_nq.rs:82-88: every coefficient through Zq::from_u64(p, v) (v mod p).
 * Rq::mul_by_f64(s), ring_nq.rs:282-292: Zq::from_f64(q, v as f64 * s).
 * Rq::div_round(s), ring_nq.rs:299-306: Zq::from_f64(q, round(v as f64 / s as f64)).
 * f64 steps are single IEEE operations, `round` is half away from zero, `as i64` saturates. */
int fhe_rq_remodule_dev(uint64_t p, const void *d_a, void *d_c, size_t count, void *hip_stream);
int fhe_rq_mul_by_f64_dev(uint64_t q, double s, const void *d_a, void *d_c, size_t count, void *hip_stream);
int fhe_rq_div_round_dev(uint64_t q, uint64_t s, const void *d_a, void *d_c, size_t count, void *hip_stream);
/* Rq::decompose(beta, l), ring_nq.rs:67-78 with Zq::decompose zq.rs:141-207 (base 2 and
 * base beta, including their saturation branch).  a: [rows][n] -> out: [rows][l][n].
 * Argument ranges: beta = 2 takes 1 <= l <= 64; at l = 64 the reference's `1 << l` (zq.rs:176-180) overflows — a
 * panic in a debug build, a shift taken modulo 64 in a --release build — and the kernels follow the RELEASE build
 * (the saturation threshold is then 1).  beta > 2 needs beta^l < 2^32 and q / beta^l > 0; anything else is
 * FHE_E_INVALID (the reference panics there in every build).  The same ranges hold for key switching. */
int fhe_rq_decompose_dev(uint64_t q, uint64_t n, unsigned beta, unsigned l, const void *d_a, void *d_out, size_t rows, void *hip_stream);

/* ---- misc ---------------------------------------------------------------- */
/* ---- multi-device --------------------------------------------------------
 * The path shards over independent polynomials / products: one
 * host thread (or process) per device, each calling hipSetDevice() and then the
 * entry points above on its own block of rows; plans are shared, tables and
 * workspaces are per device.  fhe_shard_range gives the contiguous block
 * [*begin, *end) of `total` units owned by `rank` of `world`: ceil(total/world)
 * units per rank, the last ranks short or empty (630 units over 8 ranks:
 * 79 x 7 + 77).  No collective is involved; gathering shards, where a consumer
 * needs them on one device, is the caller's (RCCL all-gather / hipMemcpyPeer). */
int fhe_shard_range(size_t total, unsigned world, unsigned rank, size_t *begin, size_t *end);
/* The one gather a block-partitioned batch may need ("only when a downstream op needs the full batch on one rank"), for a
 * caller WITHOUT torch / RCCL — one host process driving several devices, as a Rust shim behind arith::Rq would: entry r
 * of d_src_shards points at rank r's rows [begin_r, end_r) of fhe_shard_range(total_rows, world, r) on device
 * src_devices[r]; they are copied, in rank order, into d_dst (total_rows x row_words 64-bit words on dst_device) with
 * hipMemcpyPeerAsync (xGMI between devices, a plain copy on the same one) on `hip_stream`, a stream of dst_device.
 * Ordering against the kernels that produced the shards on OTHER devices' streams is the caller's (events or a
 * synchronise), as for any cross-device copy.  Empty shards (ranks past the end) may be NULL. */
int fhe_shard_gather_dev(size_t total_rows, size_t row_words, unsigned world, const int *src_devices,
                         const void *const *d_src_shards, int dst_device, void *d_dst, void *hip_stream);

/* Library workspaces are kept per (device, stream, calling thread) until fhe_ntt_shutdown(): two host threads that
 * enqueue on one stream never share an intermediate.  A thread that exits leaves its buffers to the next thread that
 * needs one on the same (device, stream) — stream order makes that safe — so short-lived threads do not accumulate
 * buffers.  A caller about to DESTROY a stream returns that stream's workspaces — every thread's, whatever entry point
 * took them — with this call, which synchronises the stream first; the caller must make sure that no other thread is
 * inside a library call on that stream at that moment (the same condition under which the stream may be destroyed at
 * all): a buffer another thread has just been handed would be freed under it.  hipStreamPerThread names a different
 * stream in every thread: the call then returns only the calling thread's buffers (a thread that used the host-buffer
 * entry points, which run on hipStreamPerThread, may call it before it exits; otherwise those buffers are released at
 * shutdown). */
int fhe_ntt_release_stream_workspace(void *hip_stream);
/* bytes of library workspace currently held, over all devices, streams and threads (diagnostic) */
size_t fhe_ntt_workspace_bytes(void);

int fhe_ntt_device_count(void);            /* HIP devices visible (0 if none) */
const char *fhe_last_error(void);          /* thread-local, never NULL */
const char *fhe_ntt_version(void);
int fhe_ntt_shutdown(void);                /* frees plans, tables, workspaces */

#ifdef __cplusplus
}
#endif
#endif /* FHE_NTT_H */
