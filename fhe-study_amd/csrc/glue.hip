// glue.hip — rows N3 / N4 of SURVEY.md §8f: the reference's batch surfaces around `Rq * Rq`
// and the element-wise operations between them, on the device, so that a caller can stay
// resident between NTT products.
//
//   N3  TR<Rq> . TR<Rq>                 arith/src/tuple_ring.rs:117-134   fhe_tr_dot_dev
//       TR<Rq> x Rq, GLWE<Rq> x Rq      tuple_ring.rs:137-155, gfhe/src/glwe.rs:263-280   fhe_tr_mul_r_dev
//       GLev<Rq> x Vec<Rq> -> GLWE      gfhe/src/glev.rs:68-80            fhe_glev_mul_dev
//       GLWE<Rq>::key_switch            gfhe/src/glwe.rs:126-137          fhe_glwe_key_switch_dev
//   N4  Rq + - neg, mul_by_u64          arith/src/ring_nq.rs:406-488,551-561,274-281
//       mod_switch, mul_div_round       ring_nq.rs:88-113, arith/src/zq.rs:134-139,32-39
//       decompose(beta, l)              ring_nq.rs:67-78, zq.rs:141-207
//
// The reference forms every product with its own pair of forward NTTs and an inverse and then
// sums canonical polynomials.  Arithmetic mod q is exact, so summing in the NTT domain and
// transforming back once gives the same canonical words: k (or l*(k+1)) products cost their
// forward transforms, one multiply-accumulate pass and ONE inverse per output polynomial.
#include <vector>

#include <cstdlib>

#include "capi_internal.hpp"
#include "digit_mac.hpp"
#include "digit32.hpp"
#include "smallq.hpp"
#include "zq_device.hpp"
#include "mac_kernel.hpp"

using fhe::Mod;
using fhe::u32;
using fhe::u64;

namespace fhe {

enum class Ew { Add, Sub, Neg, MulScalar };

template <Ew OP>
__global__ __launch_bounds__(256) void ew_kernel(const u64 *__restrict__ a, const u64 *__restrict__ b,
                                                 u64 *__restrict__ c, u64 count, Mod m, u64 s) {
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        const u64 x = a[i];
        u64 r;
        if (OP == Ew::Add) { r = x + b[i]; r = r >= m.q ? r - m.q : r; }                 // zq.rs:219-231
        else if (OP == Ew::Sub) { const u64 y = b[i]; r = x >= y ? x - y : (m.q + x) - y; }  // zq.rs:259-276
        else if (OP == Ew::Neg) r = x == 0 ? 0 : m.q - x;                                // zq.rs:301-313
        else r = (m.q >> 62) ? mul_mod_var63(x, s, m) : mul_mod_var(x, s, m);            // zq.rs:315-328 (uniform branch)
        c[i] = r;
    }
}

__device__ __forceinline__ u64 f64_as_u64(double x) {   // Rust `f64 as u64`
    if (x != x || x <= 0.0) return 0;
    if (x >= 18446744073709551616.0) return ~0ull;
    return (u64)x;
}
// Zq::mod_switch, zq.rs:134-139
__global__ __launch_bounds__(256) void mod_switch_kernel(const u64 *__restrict__ a, u64 *__restrict__ c,
                                                         u64 count, u64 q, u64 p) {
    const u64 stride = (u64)gridDim.x * 256;
    const double qf = (double)q, pf = (double)p;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        const u64 v = f64_as_u64(round(((double)a[i] * pf) / qf));
        c[i] = v >= p ? v % p : v;
    }
}

// Ring::mul_div_round for Rq, ring_nq.rs:100-113 (+ Zq::from_f64, zq.rs:32-39)
__global__ __launch_bounds__(256) void rq_mul_div_round_kernel(const u64 *__restrict__ a, u64 *__restrict__ c,
                                                               u64 count, u64 q, u64 num, u64 den) {
    const u64 stride = (u64)gridDim.x * 256;
    const double nf = (double)num, df = (double)den;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        const long long e = f64_as_i64(round(round((nf * (double)a[i]) / df)));
        const long long qi = (long long)q;
        c[i] = (e < 0 || e >= qi) ? (u64)(((e % qi) + qi) % qi) : (u64)e;
    }
}

// Rq::remodule(p) ring_nq.rs:82-88 (Zq::from_u64 per coefficient), Rq::mul_by_f64 :282-292,
// Rq::div_round :299-306 (+ Rq::from_vec_f64 -> Zq::from_f64)
enum class EwF { Remodule, MulF64, DivRound };
template <EwF OP>
__global__ __launch_bounds__(256) void ewf_kernel(const u64 *__restrict__ a, u64 *__restrict__ c, u64 count,
                                                  u64 q, u64 su, double sf) {
    const u64 stride = (u64)gridDim.x * 256;
    const u64 qmu = OP == EwF::Remodule || q == 0 ? 0ull : ~0ull / q;          // Zq::from_f64's remainder by multiplication
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        const u64 v = a[i];
        if (OP == EwF::Remodule) c[i] = v >= q ? v % q : v;                       // q = the new modulus p
        else if (OP == EwF::MulF64) c[i] = zq_from_f64_mu(q, qmu, (double)v * sf);
        else c[i] = zq_from_f64_mu(q, qmu, round((double)v / (double)su));
    }
}

// Rq::decompose(beta, l), ring_nq.rs:67-78 with Zq::decompose zq.rs:141-207.
// input row r = (group r / grp, member r % grp) lives at a + group*gstride + member*n (so the k
// mask rows of each (k+1)-row ciphertext can be picked without a gather); out [rows][l][n]
__global__ __launch_bounds__(256) void decompose_kernel(const u64 *__restrict__ a, u64 *__restrict__ out,
                                                        u64 rows, u32 n, u64 q, u32 beta, u32 l, u32 grp,
                                                        u64 gstride) {
    const u64 total = rows * n, stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const u64 r = i / n;
        const u32 j = (u32)(i - r * n);
        const u64 v = a[(r / grp) * gstride + (r % grp) * n + j];
        u64 *o = out + r * l * n + j;
        if (beta == 2) {
            if (v >= (1ull << (l & 63u))) {                       // zq.rs:176-180 (--release shift wrap)
                for (u32 d = 0; d < l; d++) o[(u64)d * n] = 1 % q;
            } else {
                for (u32 d = 0; d < l; d++) {
                    const u64 bit = (v >> (l - 1 - d)) & 1ull;
                    o[(u64)d * n] = bit >= q ? bit % q : bit;
                }
            }
        } else {
            u32 bl = 1;
            for (u32 t = 0; t < l; t++) bl *= beta;                // beta.pow(l) in u32
            if (v >= (u64)bl) {                                    // zq.rs:152-160
                for (u32 d = 0; d < l; d++) o[(u64)d * n] = (u64)beta - 1;
            } else {
                u64 rem = v;
                u32 bi = 1;
                for (u32 d = 0; d < l; d++) {
                    bi *= beta;
                    const u64 den = q / (u64)bi;
                    const u64 x = rem / den;
                    o[(u64)d * n] = x >= q ? x % q : x;
                    if (x != 0) rem = rem % den;
                }
            }
        }
    }
}

// key_switch tail, glwe.rs:129-136: out[c] = (c < k ? 0 : b) - rhs[c]
__global__ __launch_bounds__(256) void ks_tail_kernel(const u64 *__restrict__ glwe, const u64 *__restrict__ rhs,
                                                      u64 *__restrict__ out, u64 batch, u32 n, u32 k, u64 q) {
    const u32 k1 = k + 1;
    const u64 total = batch * k1 * n, stride = (u64)gridDim.x * 256;
    for (u64 idx = (u64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
        const u32 c = (u32)((idx / n) % k1);
        const u64 x = c < k ? 0ull : glwe[idx], y = rhs[idx];
        out[idx] = x >= y ? x - y : (q + x) - y;
    }
}

}  // namespace fhe


// the transforms of the batch surfaces: the 32-bit kernels where the plan has them (small q: smallq.hip; in == out allowed)
static int fwd(const fhe_ntt_plan *plan, const fhe::DevicePlan &dp, const u64 *in, u64 *out, u64 rows, hipStream_t st) {
    fhe::SmallQArgs sq{};
    if (fhe_smallq_args(plan, dp, &sq)) {
        sq.a = in; sq.out = out; sq.rows = rows;
        if (int rc = fhe_smallq_scratch(dp.log_n, rows, st, &sq)) return rc;
        hipError_t se = fhe::launch_sq_forward(sq, (int)dp.log_n, st);
        return se == hipSuccess ? FHE_OK : fhe_hip_fail(se, "glue forward NTT (32-bit)");
    }
    hipError_t e = fhe::launch_ntt_forward(dp, in, out, rows, fhe_batch_tile_for(plan), st);
    return e == hipSuccess ? FHE_OK : fhe_hip_fail(e, "glue forward NTT");
}
static int inv(const fhe_ntt_plan *plan, const fhe::DevicePlan &dp, const u64 *in, u64 *out, u64 rows, hipStream_t st) {
    fhe::SmallQArgs sq{};
    if (fhe_smallq_args(plan, dp, &sq)) {
        sq.a = in; sq.out = out; sq.rows = rows;
        if (int rc = fhe_smallq_scratch(dp.log_n, rows, st, &sq)) return rc;
        hipError_t se = fhe::launch_sq_inverse(sq, (int)dp.log_n, st);
        return se == hipSuccess ? FHE_OK : fhe_hip_fail(se, "glue inverse NTT (32-bit)");
    }
    hipError_t e = fhe::launch_ntt_inverse(dp, in, nullptr, nullptr, out, rows, fhe_batch_tile_for(plan), st);
    return e == hipSuccess ? FHE_OK : fhe_hip_fail(e, "glue inverse NTT");
}

// ---- N4 ------------------------------------------------------------------------------------
template <fhe::Ew OP>
static int ew_call(const fhe_ntt_plan *plan, const void *a, const void *b, void *c, size_t batch, u64 s, void *stream,
                   const char *who) {
    if (!plan) return fhe_fail(FHE_E_NULL, "%s: plan is NULL", who);
    if (batch == 0) return FHE_OK;
    if (!a || !c || ((OP == fhe::Ew::Add || OP == fhe::Ew::Sub) && !b)) return fhe_fail(FHE_E_NULL, "%s: NULL buffer", who);
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    const u64 count = batch * plan->n;
    { fhe::KernelTimer kt_("ew", 0, (hipStream_t)stream);
    hipLaunchKernelGGL((fhe::ew_kernel<OP>), dim3(fhe_ew_grid(count)), dim3(256), 0, (hipStream_t)stream, (const u64 *)a,
                       (const u64 *)b, (u64 *)c, count, plan->mod, s);
    }
    LAUNCH_OK(who);
    return FHE_OK;
}
extern "C" int fhe_rq_add_dev(const fhe_ntt_plan *plan, const void *d_a, const void *d_b, void *d_c, size_t batch, void *st) {
    return ew_call<fhe::Ew::Add>(plan, d_a, d_b, d_c, batch, 0, st, "fhe_rq_add_dev");
}
extern "C" int fhe_rq_sub_dev(const fhe_ntt_plan *plan, const void *d_a, const void *d_b, void *d_c, size_t batch, void *st) {
    return ew_call<fhe::Ew::Sub>(plan, d_a, d_b, d_c, batch, 0, st, "fhe_rq_sub_dev");
}
extern "C" int fhe_rq_neg_dev(const fhe_ntt_plan *plan, const void *d_a, void *d_c, size_t batch, void *st) {
    return ew_call<fhe::Ew::Neg>(plan, d_a, nullptr, d_c, batch, 0, st, "fhe_rq_neg_dev");
}
extern "C" int fhe_rq_mul_by_u64_dev(const fhe_ntt_plan *plan, const void *d_a, uint64_t s, void *d_c, size_t batch, void *st) {
    if (!plan) return fhe_fail(FHE_E_NULL, "fhe_rq_mul_by_u64_dev: plan is NULL");
    return ew_call<fhe::Ew::MulScalar>(plan, d_a, nullptr, d_c, batch, s % plan->q, st, "fhe_rq_mul_by_u64_dev");
}
extern "C" int fhe_rq_mod_switch_dev(uint64_t q, uint64_t p, const void *d_a, void *d_c, size_t count, void *st) {
    if (q == 0 || p == 0) return fhe_fail(FHE_E_BAD_Q, "fhe_rq_mod_switch_dev: q and p must be non-zero");
    if (count == 0) return FHE_OK;
    if (!d_a || !d_c) return fhe_fail(FHE_E_NULL, "fhe_rq_mod_switch_dev: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    { fhe::KernelTimer kt_("mod_switch", 0, (hipStream_t)st);
    hipLaunchKernelGGL(fhe::mod_switch_kernel, dim3(fhe_ew_grid(count)), dim3(256), 0, (hipStream_t)st, (const u64 *)d_a, (u64 *)d_c, (u64)count, (u64)q, (u64)p);
    }
    LAUNCH_OK("mod_switch_kernel");
    return FHE_OK;
}
extern "C" int fhe_rq_mul_div_round_dev(uint64_t q, uint64_t num, uint64_t den, const void *d_a, void *d_c, size_t count, void *st) {
    if (q == 0 || den == 0 || (q >> 63)) return fhe_fail(FHE_E_BAD_Q, "fhe_rq_mul_div_round_dev: need 0 < q < 2^63, den > 0");
    if (count == 0) return FHE_OK;
    if (!d_a || !d_c) return fhe_fail(FHE_E_NULL, "fhe_rq_mul_div_round_dev: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    { fhe::KernelTimer kt_("rq_mul_div_round", 0, (hipStream_t)st);
    hipLaunchKernelGGL(fhe::rq_mul_div_round_kernel, dim3(fhe_ew_grid(count)), dim3(256), 0, (hipStream_t)st, (const u64 *)d_a, (u64 *)d_c, (u64)count, (u64)q, (u64)num, (u64)den);
    }
    LAUNCH_OK("rq_mul_div_round_kernel");
    return FHE_OK;
}
extern "C" int fhe_rq_remodule_dev(uint64_t p, const void *d_a, void *d_c, size_t count, void *st) {
    if (p == 0) return fhe_fail(FHE_E_BAD_Q, "fhe_rq_remodule_dev: p = 0");
    if (count == 0) return FHE_OK;
    if (!d_a || !d_c) return fhe_fail(FHE_E_NULL, "fhe_rq_remodule_dev: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    { fhe::KernelTimer kt_("ewf", 0, (hipStream_t)st);
    hipLaunchKernelGGL((fhe::ewf_kernel<fhe::EwF::Remodule>), dim3(fhe_ew_grid(count)), dim3(256), 0, (hipStream_t)st, (const u64 *)d_a, (u64 *)d_c, (u64)count, (u64)p, (u64)0, 0.0);
    }
    LAUNCH_OK("ewf_kernel<Remodule>");
    return FHE_OK;
}
extern "C" int fhe_rq_mul_by_f64_dev(uint64_t q, double s, const void *d_a, void *d_c, size_t count, void *st) {
    if (q == 0 || (q >> 63)) return fhe_fail(FHE_E_BAD_Q, "fhe_rq_mul_by_f64_dev: need 0 < q < 2^63");
    if (count == 0) return FHE_OK;
    if (!d_a || !d_c) return fhe_fail(FHE_E_NULL, "fhe_rq_mul_by_f64_dev: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    { fhe::KernelTimer kt_("ewf", 0, (hipStream_t)st);
    hipLaunchKernelGGL((fhe::ewf_kernel<fhe::EwF::MulF64>), dim3(fhe_ew_grid(count)), dim3(256), 0, (hipStream_t)st, (const u64 *)d_a, (u64 *)d_c, (u64)count, (u64)q, (u64)0, s);
    }
    LAUNCH_OK("ewf_kernel<MulF64>");
    return FHE_OK;
}
extern "C" int fhe_rq_div_round_dev(uint64_t q, uint64_t s, const void *d_a, void *d_c, size_t count, void *st) {
    if (q == 0 || (q >> 63)) return fhe_fail(FHE_E_BAD_Q, "fhe_rq_div_round_dev: need 0 < q < 2^63");
    if (s == 0) return fhe_fail(FHE_E_INVALID, "fhe_rq_div_round_dev: s = 0");
    if (count == 0) return FHE_OK;
    if (!d_a || !d_c) return fhe_fail(FHE_E_NULL, "fhe_rq_div_round_dev: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    { fhe::KernelTimer kt_("ewf", 0, (hipStream_t)st);
    hipLaunchKernelGGL((fhe::ewf_kernel<fhe::EwF::DivRound>), dim3(fhe_ew_grid(count)), dim3(256), 0, (hipStream_t)st, (const u64 *)d_a, (u64 *)d_c, (u64)count, (u64)q, (u64)s, 0.0);
    }
    LAUNCH_OK("ewf_kernel<DivRound>");
    return FHE_OK;
}
// Argument ranges in which Zq::decompose (zq.rs:141-190) is defined.  Outside them the reference
// panics (`beta.pow(l)` overflowing u32 and `>> i` with i >= 64 in a debug build, `q / beta^i` = 0
// as a divisor in any build); the ABI answers FHE_E_INVALID and never launches.
static int check_decompose_args(const char *who, u64 q, unsigned beta, unsigned l) {
    if (beta < 2 || l < 1) return fhe_fail(FHE_E_INVALID, "%s: need beta >= 2, l >= 1", who);
    if (beta == 2) {
        if (l > 64) return fhe_fail(FHE_E_INVALID, "%s: beta = 2 needs l <= 64 (got %u)", who, l);
        return FHE_OK;
    }
    u64 bl = 1;
    for (unsigned i = 0; i < l; i++) {
        bl *= beta;
        if (bl >> 32) return fhe_fail(FHE_E_INVALID, "%s: beta^l = %u^%u overflows u32 (beta.pow(l), zq.rs:152)", who, beta, l);
    }
    if (q / bl == 0) return fhe_fail(FHE_E_INVALID, "%s: q / beta^l = 0 for q=%llu, beta=%u, l=%u (divisor at zq.rs:164-165)", who,
                                     (unsigned long long)q, beta, l);
    return FHE_OK;
}

extern "C" int fhe_rq_decompose_dev(uint64_t q, uint64_t n, unsigned beta, unsigned l, const void *d_a, void *d_out, size_t rows, void *st) {
    if (q == 0 || n == 0) return fhe_fail(FHE_E_INVALID, "fhe_rq_decompose_dev: need q, n > 0");
    if (int vrc = check_decompose_args("fhe_rq_decompose_dev", q, beta, l)) return vrc;
    if (rows == 0) return FHE_OK;
    if (!d_a || !d_out) return fhe_fail(FHE_E_NULL, "fhe_rq_decompose_dev: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    { fhe::KernelTimer kt_("decompose", 0, (hipStream_t)st);
    hipLaunchKernelGGL(fhe::decompose_kernel, dim3(fhe_ew_grid(rows * n)), dim3(256), 0, (hipStream_t)st, (const u64 *)d_a, (u64 *)d_out, (u64)rows, (u32)n, (u64)q, (u32)beta, (u32)l, (u32)1, (u64)n);
    }
    LAUNCH_OK("decompose_kernel");
    return FHE_OK;
}

// ---- N3 ------------------------------------------------------------------------------------
// flags (include/fhe_ntt.h): FHE_A_IS_EVALS / FHE_B_IS_EVALS = that operand is already in the NTT
// domain (the generalisation of Rq.evals, ring_nq.rs:24-26: a key transformed once serves every
// later call); FHE_OUT_EVALS = leave the result there.
static int bad_flags(const char *who, unsigned flags, unsigned allowed) {
    if (flags & ~allowed) return fhe_fail(FHE_E_INVALID, "%s: unsupported flags 0x%x (allowed 0x%x)", who, flags, allowed);
    return FHE_OK;
}

// c[b] = sum_{i<k} a[b][i] * b[b][i];  a, b: [batch][k][n];  c: [batch][n]
extern "C" int fhe_tr_dot_dev(const fhe_ntt_plan *plan, const void *d_a, const void *d_b, void *d_c, unsigned k, size_t batch, unsigned flags, void *stream) {
    if (!plan) return fhe_fail(FHE_E_NULL, "fhe_tr_dot_dev: plan is NULL");
    int rc = bad_flags("fhe_tr_dot_dev", flags, FHE_A_IS_EVALS | FHE_B_IS_EVALS | FHE_OUT_EVALS);
    if (rc != FHE_OK) return rc;
    if (batch == 0 || k == 0) return FHE_OK;
    if (!d_a || !d_b || !d_c) return fhe_fail(FHE_E_NULL, "fhe_tr_dot_dev: NULL buffer");
    REQUIRE_ALIGNED(d_a); REQUIRE_ALIGNED(d_b); REQUIRE_ALIGNED(d_c);
    fhe::DevicePlan dp;
    if ((rc = fhe_device_plan(plan, &dp)) != FHE_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const u64 n = plan->n, rows = batch * k;
    const bool a_ev = flags & FHE_A_IS_EVALS, b_ev = flags & FHE_B_IS_EVALS, out_ev = flags & FHE_OUT_EVALS;
    void *w = nullptr;
    if ((rc = fhe_workspace_get(1, (2 * rows + batch) * n * 8, st, &w)) != FHE_OK) return rc;
    u64 *WA = (u64 *)w, *WB = WA + rows * n, *WC = WB + rows * n;
    const u64 *A = (const u64 *)d_a, *B = (const u64 *)d_b;
    if (!a_ev) { if ((rc = fwd(plan, dp, A, WA, rows, st)) != FHE_OK) return rc; A = WA; }
    if (!b_ev) { if ((rc = fwd(plan, dp, B, WB, rows, st)) != FHE_OK) return rc; B = WB; }
    u64 *C = out_ev ? (u64 *)d_c : WC;
    // T = k terms, nc = 1 output row, "G" = A per batch element
    { fhe::KernelTimer kt_("mac_rows", 0, st);
    fhe::launch_mac_rows(dp.arith == fhe::kArStrict63, fhe_ew_grid(fhe::mac_rows_threads(batch, 1, n)), st, A, B, C, (u64)batch, (u32)n, (u32)k, (u32)1, (u64)k * n, plan->mod);
    }
    LAUNCH_OK("mac_rows_kernel");
    return out_ev ? FHE_OK : inv(plan, dp, C, (u64 *)d_c, batch, st);
}

// out[b][i] = a[b][i] * p[b], i < rows;  a, out: [batch][rows][n];  p: [batch][n]
extern "C" int fhe_tr_mul_r_dev(const fhe_ntt_plan *plan, const void *d_a, const void *d_p, void *d_out, unsigned rows, size_t batch, unsigned flags, void *stream) {
    if (!plan) return fhe_fail(FHE_E_NULL, "fhe_tr_mul_r_dev: plan is NULL");
    int rc = bad_flags("fhe_tr_mul_r_dev", flags, FHE_A_IS_EVALS | FHE_B_IS_EVALS | FHE_OUT_EVALS);
    if (rc != FHE_OK) return rc;
    if (batch == 0 || rows == 0) return FHE_OK;
    if (!d_a || !d_p || !d_out) return fhe_fail(FHE_E_NULL, "fhe_tr_mul_r_dev: NULL buffer");
    REQUIRE_ALIGNED(d_a); REQUIRE_ALIGNED(d_p); REQUIRE_ALIGNED(d_out);
    fhe::DevicePlan dp;
    if ((rc = fhe_device_plan(plan, &dp)) != FHE_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const u64 n = plan->n, total = batch * rows;
    const bool a_ev = flags & FHE_A_IS_EVALS, p_ev = flags & FHE_B_IS_EVALS, out_ev = flags & FHE_OUT_EVALS;
    void *w = nullptr;
    if ((rc = fhe_workspace_get(1, (2 * total + batch) * n * 8, st, &w)) != FHE_OK) return rc;
    u64 *WA = (u64 *)w, *WC = WA + total * n, *WP = WC + total * n;
    const u64 *A = (const u64 *)d_a, *P = (const u64 *)d_p;
    if (!a_ev) { if ((rc = fwd(plan, dp, A, WA, total, st)) != FHE_OK) return rc; A = WA; }
    if (!p_ev) { if ((rc = fwd(plan, dp, P, WP, batch, st)) != FHE_OK) return rc; P = WP; }
    u64 *C = out_ev ? (u64 *)d_out : WC;
    // T = 1, nc = rows: out[b][c] = A[b][c] * P[b]
    { fhe::KernelTimer kt_("mac_rows", 0, st);
    fhe::launch_mac_rows(dp.arith == fhe::kArStrict63, fhe_ew_grid(fhe::mac_rows_threads(batch, rows, n)), st, A, P, C, (u64)batch, (u32)n, (u32)1, (u32)rows, (u64)rows * n, plan->mod);
    }
    LAUNCH_OK("mac_rows_kernel");
    return out_ev ? FHE_OK : inv(plan, dp, C, (u64 *)d_out, total, st);
}

// shared worker: out[b][c] = sum_{t<T} key[t][c] * v[b][t]; key [T][nc][n] (one for the batch)
static int keyed_mac(const fhe_ntt_plan *plan, const fhe::DevicePlan &dp, const u64 *d_key, bool key_is_evals, const u64 *d_v, bool v_is_evals,
                     u64 *d_out, bool out_evals, u32 T, u32 nc, u64 batch, u64 *ws, hipStream_t st) {
    const u64 n = plan->n;
    u64 *WK = ws, *WV = WK + (u64)T * nc * n;
    const u64 *K = d_key, *V = d_v;
    int rc;
    if (!key_is_evals) { if ((rc = fwd(plan, dp, d_key, WK, (u64)T * nc, st)) != FHE_OK) return rc; K = WK; }
    if (!v_is_evals) { if ((rc = fwd(plan, dp, d_v, WV, batch * T, st)) != FHE_OK) return rc; V = WV; }
    { fhe::KernelTimer kt_("mac_rows", 0, st);
    fhe::launch_mac_rows(dp.arith == fhe::kArStrict63, fhe_ew_grid(fhe::mac_rows_threads(batch, nc, n)), st, K, V, d_out, batch, (u32)n, T, nc, (u64)0, plan->mod);
    }
    LAUNCH_OK("mac_rows_kernel");
    return out_evals ? FHE_OK : inv(plan, dp, d_out, d_out, batch * nc, st);
}

// GLev x Vec<R> -> GLWE: glev [l][k+1][n] (a key, shared), v [batch][l][n], out [batch][k+1][n]
extern "C" int fhe_glev_mul_dev(const fhe_ntt_plan *plan, unsigned k, unsigned l, const void *d_glev, const void *d_v, void *d_out, size_t batch, unsigned flags, void *stream) {
    if (!plan) return fhe_fail(FHE_E_NULL, "fhe_glev_mul_dev: plan is NULL");
    int rc = bad_flags("fhe_glev_mul_dev", flags, FHE_A_IS_EVALS | FHE_B_IS_EVALS | FHE_OUT_EVALS);
    if (rc != FHE_OK) return rc;
    if (batch == 0) return FHE_OK;
    if (l == 0) return fhe_fail(FHE_E_INVALID, "fhe_glev_mul_dev: l = 0");
    if (!d_glev || !d_v || !d_out) return fhe_fail(FHE_E_NULL, "fhe_glev_mul_dev: NULL buffer");
    REQUIRE_ALIGNED(d_glev); REQUIRE_ALIGNED(d_v); REQUIRE_ALIGNED(d_out);
    fhe::DevicePlan dp;
    if ((rc = fhe_device_plan(plan, &dp)) != FHE_OK) return rc;
    const u64 n = plan->n;
    void *w = nullptr;
    if ((rc = fhe_workspace_get(1, ((u64)l * (k + 1) + batch * l) * n * 8, (hipStream_t)stream, &w)) != FHE_OK) return rc;
    return keyed_mac(plan, dp, (const u64 *)d_glev, flags & FHE_A_IS_EVALS, (const u64 *)d_v, flags & FHE_B_IS_EVALS, (u64 *)d_out,
                     flags & FHE_OUT_EVALS, l, k + 1, batch, (u64 *)w, (hipStream_t)stream);
}

// ---- key switching on two 27-bit primes (digit32.hip) -------------------------------------------------------------
// Base 2, k = 1: the sums  sum_t ksk[t][c] * digit_t  are small integers (|half-sum| < k l n 2^32 with the key words
// split at bit 32), so they are computed modulo two 27-bit primes with 32-bit arithmetic: the key halves transformed
// per prime (per call, or once by fhe_glwe_ksk_prepare_dev), then digit extraction -> both transforms ->
// multiply-accumulate in one kernel, and a tail that lifts the sums back, reduces them modulo q and forms (0, b) - rhs.
static bool ks32_usable(const fhe_ntt_plan *plan, unsigned k, unsigned beta, unsigned l) {
    return beta == 2 && fhe_ext32_enabled() && (plan->q >> 61) == 0 && fhe::ks32_shape_supported(plan->n, k, l);
}
// key32: the prepared key ([prime][t][half][c][n] u32), or nullptr to prepare d_ksk into the workspace first
static int ks32_run(const fhe_ntt_plan *plan, const fhe::DevicePlan &dp, unsigned k, unsigned l, const void *d_glwe,
                    const uint32_t *key32, const void *d_ksk, void *d_out, size_t batch, hipStream_t st) {
    const u64 n = plan->n;
    const u32 k1 = k + 1, T = k * l;
    fhe::Ext32Args a{};
    int rc = fhe_ext32_tables(n, &a);
    if (rc != FHE_OK) return rc;
    const u32 W = fhe::ext32_units((int)dp.log_n);
    // parts: split the digits of a ciphertext over workgroups only while the batch alone leaves CUs empty (one
    // workgroup of 512 threads per CU at n = 4096, two of 256 below); measured at n = 4096, batch 256: 1 part 354 us,
    // 2 parts 376, 4 parts 395 (every part ends in 64 reductions per thread and its own partial sums)
    u32 parts = 1;
    const u64 slots = dp.log_n == 12 ? 256 : 512;
    while (parts < 8 && batch * parts < slots && (T / (parts * 2)) >= 2 * W) parts *= 2;
    // [partial sums: batch*parts*2*2*k1 rows of u32] [key transforms: 2 primes * 2*T*k1 rows of u32, unless prepared]
    const u64 krows = 2ull * T * k1, part_words = (u64)batch * parts * 2 * (2 * k1) * n;
    void *w = nullptr;
    if ((rc = fhe_workspace_get(1, (part_words + (key32 ? 0 : 2 * krows * n)) * 4, st, &w)) != FHE_OK) return rc;
    uint32_t *PART32 = (uint32_t *)w;
    hipError_t e = hipSuccess;
    if (!key32) {
        uint32_t *KEY32 = PART32 + part_words;
        a.key64 = (const u64 *)d_ksk; a.key32 = KEY32; a.rows = krows; a.key_k1 = k1;
        if ((e = fhe::launch_ext32_key(a, (int)dp.log_n, st)) != hipSuccess) return fhe_hip_fail(e, "ntt32_fwd_key_kernel");
        key32 = KEY32;
    }
    a.key32 = const_cast<uint32_t *>(key32);
    a.src = (const u64 *)d_glwe; a.ct_stride = (u64)k1 * n; a.part32 = PART32; a.out = (u64 *)d_out; a.batch = batch;
    a.l = l; a.T = T; a.parts = parts;
    a.tpp = ((T + parts - 1) / parts + W - 1) / W * W;
    a.mod = dp.mod; a.two32 = (1ull << 32) % plan->q; a.glwe = (const u64 *)d_glwe; a.k = k;
    e = fhe::launch_ext32_mac(a, (int)dp.log_n, fhe::SRC_ZQBITS, st);
    if (e == hipSuccess) e = fhe::launch_ext32_tail_ks(a, (int)dp.log_n, st);
    return e == hipSuccess ? FHE_OK : fhe_hip_fail(e, "digit32 kernels");
}

// GLWE::key_switch: glwe [batch][k+1][n]; ksk [k][l][k+1][n] (shared); out [batch][k+1][n]
extern "C" int fhe_glwe_key_switch_dev(const fhe_ntt_plan *plan, unsigned k, unsigned beta, unsigned l, const void *d_glwe, const void *d_ksk,
                                       void *d_out, size_t batch, unsigned flags, void *stream) {
    if (!plan) return fhe_fail(FHE_E_NULL, "fhe_glwe_key_switch_dev: plan is NULL");
    // only the key may be pre-transformed: the ciphertext is decomposed, and the tail subtracts, in coefficients
    int rc = bad_flags("fhe_glwe_key_switch_dev", flags, FHE_A_IS_EVALS);
    if (rc != FHE_OK) return rc;
    if (batch == 0) return FHE_OK;
    if (k == 0 || l == 0 || beta < 2) return fhe_fail(FHE_E_INVALID, "fhe_glwe_key_switch_dev: need k, l >= 1, beta >= 2");
    if ((rc = check_decompose_args("fhe_glwe_key_switch_dev", plan->q, beta, l)) != FHE_OK) return rc;
    if (!d_glwe || !d_ksk || !d_out) return fhe_fail(FHE_E_NULL, "fhe_glwe_key_switch_dev: NULL buffer");
    REQUIRE_ALIGNED(d_glwe); REQUIRE_ALIGNED(d_ksk); REQUIRE_ALIGNED(d_out);
    fhe::DevicePlan dp;
    if ((rc = fhe_device_plan(plan, &dp)) != FHE_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const u64 n = plan->n;
    const u32 k1 = k + 1, T = k * l;
    void *w = nullptr;
    // Base 2 at 2^8 <= n <= 2^12: decomposition, digit transforms and the multiply-accumulate against the
    // key are ONE kernel (digit_mac.hip): RHS[b][c] = sum_t KSK[t][c] (.) NTT(digit_t(b)), nothing else stored.
    // Base 2, k = 1, key in coefficients: two 27-bit primes and 32-bit arithmetic (ks32_run above)
    if (!(flags & FHE_A_IS_EVALS) && ks32_usable(plan, k, beta, l))
        return ks32_run(plan, dp, k, l, d_glwe, nullptr, d_ksk, d_out, batch, st);
    static const bool fused_on = [] { const char *e = getenv("FHE_DIGIT_MAC_FUSED"); return !(e && e[0] == '0'); }();
    if (beta == 2 && fused_on && dp.wide && dp.log_n >= 8 && dp.log_n <= 12 && (k1 == 2 || k1 == 3)) {
        const u32 parts = fhe::digit_mac_parts(batch, T, dp.log_n, k1);
        // [key transforms: T*k1 rows] [rhs: batch*k1 rows] [partial sums: parts*batch*k1 rows]
        if ((rc = fhe_workspace_get(1, ((u64)T * k1 + (u64)(parts + 1) * batch * k1) * n * 8, st, &w)) != FHE_OK) return rc;
        u64 *KEY = (u64 *)w, *RHS = KEY + (u64)T * k1 * n, *PART = parts > 1 ? RHS + batch * k1 * n : RHS;
        const u64 *key = (const u64 *)d_ksk;
        if (!(flags & FHE_A_IS_EVALS)) { if ((rc = fwd(plan, dp, key, KEY, (u64)T * k1, st)) != FHE_OK) return rc; key = KEY; }
        hipError_t e = fhe::launch_digit_mac(dp, fhe::SRC_ZQBITS, (const u64 *)d_glwe, (u64)k1 * n, k, l, key, k1, PART, parts, batch, st);
        if (e == hipSuccess) {
            // sum of the parts, the k+1 inverse transforms and (0, b) - rhs (glwe.rs:129-136) in one kernel
            e = fhe::launch_digit_tail_ks(dp, PART, parts, k, (const u64 *)d_glwe, (u64 *)d_out, batch, st);
            if (e == hipSuccess) return FHE_OK;
            if (e != hipErrorNotSupported) return fhe_hip_fail(e, "digit_tail_kernel");
            (void)hipGetLastError();
            if (parts > 1 && (e = fhe::launch_sum_parts(PART, RHS, batch, parts, (u64)k1 * n, plan->q, st)) != hipSuccess)
                return fhe_hip_fail(e, "sum_parts_kernel");
            if ((rc = inv(plan, dp, RHS, RHS, batch * k1, st)) != FHE_OK) return rc;
            { fhe::KernelTimer kt_("ks_tail", 0, st);
            hipLaunchKernelGGL(fhe::ks_tail_kernel, dim3(fhe_ew_grid(batch * k1 * n)), dim3(256), 0, st, (const u64 *)d_glwe, (const u64 *)RHS, (u64 *)d_out, (u64)batch, (u32)n, (u32)k, (u64)plan->q);
            }
            LAUNCH_OK("ks_tail_kernel");
            return FHE_OK;
        }
        if (e != hipErrorNotSupported) return fhe_hip_fail(e, "digit_mac_kernel");
        (void)hipGetLastError();
    }
    // [decomposition: batch*k*l rows] [rhs: batch*k1 rows] [keyed_mac scratch: T*k1 + batch*T rows]
    if ((rc = fhe_workspace_get(1, (batch * T + batch * k1 + (u64)T * k1 + batch * T) * n * 8, st, &w)) != FHE_OK) return rc;
    u64 *DEC = (u64 *)w, *RHS = DEC + batch * T * n, *WS = RHS + batch * k1 * n;
    // decompose the k mask polynomials of every ciphertext (the body row is skipped): rows (b, i) -> [b][i][d].
    // Base 2 at single-pass sizes: the digit is extracted in the load of its forward transform
    // (SRC_ZQBITS), so DEC receives the transforms and the digit polynomials never exist.
    bool dec_is_evals = false;
    if (beta == 2) {
        hipError_t e = fhe::launch_ntt_forward_zqbits(dp, (const u64 *)d_glwe, DEC, (u64)batch * k, (u32)l, (u32)k, (u64)k1 * n, st);
        if (e == hipSuccess) dec_is_evals = true;
        else if (e != hipErrorNotSupported) return fhe_hip_fail(e, "digit forward NTT");
        else (void)hipGetLastError();
    }
    if (!dec_is_evals) {
        { fhe::KernelTimer kt_("decompose", 0, st);
        hipLaunchKernelGGL(fhe::decompose_kernel, dim3(fhe_ew_grid(batch * k * n)), dim3(256), 0, st, (const u64 *)d_glwe, DEC, (u64)batch * k, (u32)n, (u64)plan->q, (u32)beta, (u32)l, (u32)k, (u64)k1 * n);
        }
        LAUNCH_OK("decompose_kernel");
    }
    // ksk viewed as [T = k*l][k1][n]; DEC as [batch][T][n]
    if (dp.wide && dp.log_n >= 8 && dp.log_n <= 12) {
        // sums left in the NTT domain; inverse transforms and (0, b) - rhs in one kernel (digit_mac.hip)
        if ((rc = keyed_mac(plan, dp, (const u64 *)d_ksk, flags & FHE_A_IS_EVALS, DEC, dec_is_evals, RHS, true, T, k1, batch, WS, st)) != FHE_OK) return rc;
        hipError_t e = fhe::launch_digit_tail_ks(dp, RHS, 1, k, (const u64 *)d_glwe, (u64 *)d_out, batch, st);
        if (e == hipSuccess) return FHE_OK;
        return fhe_hip_fail(e, "digit_tail_kernel");
    }
    if ((rc = keyed_mac(plan, dp, (const u64 *)d_ksk, flags & FHE_A_IS_EVALS, DEC, dec_is_evals, RHS, false, T, k1, batch, WS, st)) != FHE_OK) return rc;
    { fhe::KernelTimer kt_("ks_tail", 0, st);
    hipLaunchKernelGGL(fhe::ks_tail_kernel, dim3(fhe_ew_grid(batch * k1 * n)), dim3(256), 0, st, (const u64 *)d_glwe, (const u64 *)RHS, (u64 *)d_out, (u64)batch, (u32)n, (u32)k, (u64)plan->q);
    }
    LAUNCH_OK("ks_tail_kernel");
    return FHE_OK;
}

// Resident key-switching key: the form the product consumes, built once.  Base 2 with k = 1 at 2^8 <= n <= 2^12: the
// two-small-prime transforms of the key halves; otherwise the forward transforms modulo q (what FHE_A_IS_EVALS takes).
static int ksk_args_ok(const char *fn, const fhe_ntt_plan *plan, unsigned k, unsigned beta, unsigned l) {
    if (!plan) return fhe_fail(FHE_E_NULL, "%s: plan is NULL", fn);
    if (k == 0 || l == 0 || beta < 2) return fhe_fail(FHE_E_INVALID, "%s: need k, l >= 1, beta >= 2", fn);
    return check_decompose_args(fn, plan->q, beta, l);
}
extern "C" size_t fhe_glwe_ksk_prepared_words(const fhe_ntt_plan *plan, unsigned k, unsigned beta, unsigned l) {
    if (!plan || k == 0 || l == 0 || beta < 2) return 0;
    {   // the arguments fhe_glwe_ksk_prepare_dev would reject (check_decompose_args) have no prepared form: 0, and the
        // caller's fhe_last_error() is left as it was
        unsigned bad = beta == 2 ? l > 64 : 0;
        u64 bl = 1;
        for (unsigned i = 0; beta != 2 && i < l && !bad; i++) { bl *= beta; bad = (bl >> 32) != 0; }
        if (!bad && beta != 2 && plan->q / bl == 0) bad = 1;
        if (bad) return 0;
    }
    const size_t rows = (size_t)k * l * (k + 1);
    return (ks32_usable(plan, k, beta, l) ? 2 : 1) * rows * plan->n;
}
extern "C" int fhe_glwe_ksk_prepare_dev(const fhe_ntt_plan *plan, unsigned k, unsigned beta, unsigned l, const void *d_ksk,
                                        void *d_prepared, void *stream) {
    int rc = ksk_args_ok("fhe_glwe_ksk_prepare_dev", plan, k, beta, l);
    if (rc != FHE_OK) return rc;
    if (!d_ksk || !d_prepared) return fhe_fail(FHE_E_NULL, "fhe_glwe_ksk_prepare_dev: NULL buffer");
    REQUIRE_ALIGNED(d_ksk); REQUIRE_ALIGNED(d_prepared);
    if (d_ksk == d_prepared) return fhe_fail(FHE_E_INVALID, "fhe_glwe_ksk_prepare_dev: the prepared key cannot overwrite the key");
    fhe::DevicePlan dp;
    if ((rc = fhe_device_plan(plan, &dp)) != FHE_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const u64 rows = (u64)k * l * (k + 1);
    if (!ks32_usable(plan, k, beta, l)) return fwd(plan, dp, (const u64 *)d_ksk, (u64 *)d_prepared, rows, st);
    fhe::Ext32Args a{};
    if ((rc = fhe_ext32_tables(plan->n, &a)) != FHE_OK) return rc;
    a.key64 = (const u64 *)d_ksk; a.key32 = (uint32_t *)d_prepared; a.rows = 2 * rows; a.key_k1 = k + 1;
    hipError_t e = fhe::launch_ext32_key(a, (int)dp.log_n, st);
    return e == hipSuccess ? FHE_OK : fhe_hip_fail(e, "ntt32_fwd_key_kernel");
}
extern "C" int fhe_glwe_key_switch_prepared_dev(const fhe_ntt_plan *plan, unsigned k, unsigned beta, unsigned l, const void *d_glwe,
                                                const void *d_prepared, void *d_out, size_t batch, void *stream) {
    int rc = ksk_args_ok("fhe_glwe_key_switch_prepared_dev", plan, k, beta, l);
    if (rc != FHE_OK) return rc;
    if (!ks32_usable(plan, k, beta, l))
        return fhe_glwe_key_switch_dev(plan, k, beta, l, d_glwe, d_prepared, d_out, batch, FHE_A_IS_EVALS, stream);
    if (batch == 0) return FHE_OK;
    if (!d_glwe || !d_prepared || !d_out) return fhe_fail(FHE_E_NULL, "fhe_glwe_key_switch_prepared_dev: NULL buffer");
    REQUIRE_ALIGNED(d_glwe); REQUIRE_ALIGNED(d_prepared); REQUIRE_ALIGNED(d_out);
    fhe::DevicePlan dp;
    if ((rc = fhe_device_plan(plan, &dp)) != FHE_OK) return rc;
    return ks32_run(plan, dp, k, l, d_glwe, (const uint32_t *)d_prepared, nullptr, d_out, batch, (hipStream_t)stream);
}

// ---- host-buffer forms of the N3 surfaces (what a shim of gfhe binds: its data are host Vecs) ----
extern "C" int fhe_tr_dot(const fhe_ntt_plan *plan, const uint64_t *a, const uint64_t *b, uint64_t *c, unsigned k, size_t batch) {
    if (!plan) return fhe_fail(FHE_E_NULL, "fhe_tr_dot: plan is NULL");
    if (batch == 0 || k == 0) return FHE_OK;
    if (!a || !b || !c) return fhe_fail(FHE_E_NULL, "fhe_tr_dot: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    const size_t row = plan->n * 8;
    FheHostStage hs;
    void *da, *db, *dc;
    if ((rc = hs.up(a, batch * k * row, &da)) != FHE_OK) return rc;
    if ((rc = hs.up(b, batch * k * row, &db)) != FHE_OK) return rc;
    if ((rc = hs.up(nullptr, batch * row, &dc)) != FHE_OK) return rc;
    if ((rc = fhe_tr_dot_dev(plan, da, db, dc, k, batch, 0, hipStreamPerThread)) != FHE_OK) return rc;
    return hs.down(c, dc, batch * row);
}

extern "C" int fhe_tr_mul_r(const fhe_ntt_plan *plan, const uint64_t *a, const uint64_t *p, uint64_t *out, unsigned rows, size_t batch) {
    if (!plan) return fhe_fail(FHE_E_NULL, "fhe_tr_mul_r: plan is NULL");
    if (batch == 0 || rows == 0) return FHE_OK;
    if (!a || !p || !out) return fhe_fail(FHE_E_NULL, "fhe_tr_mul_r: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    const size_t row = plan->n * 8;
    FheHostStage hs;
    void *da, *dp, *dout;
    if ((rc = hs.up(a, batch * rows * row, &da)) != FHE_OK) return rc;
    if ((rc = hs.up(p, batch * row, &dp)) != FHE_OK) return rc;
    if ((rc = hs.up(nullptr, batch * rows * row, &dout)) != FHE_OK) return rc;
    if ((rc = fhe_tr_mul_r_dev(plan, da, dp, dout, rows, batch, 0, hipStreamPerThread)) != FHE_OK) return rc;
    return hs.down(out, dout, batch * rows * row);
}

extern "C" int fhe_glev_mul(const fhe_ntt_plan *plan, unsigned k, unsigned l, const uint64_t *glev, const uint64_t *v, uint64_t *out, size_t batch) {
    if (!plan) return fhe_fail(FHE_E_NULL, "fhe_glev_mul: plan is NULL");
    if (batch == 0) return FHE_OK;
    if (l == 0) return fhe_fail(FHE_E_INVALID, "fhe_glev_mul: l = 0");
    if (!glev || !v || !out) return fhe_fail(FHE_E_NULL, "fhe_glev_mul: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    const size_t row = plan->n * 8, k1 = (size_t)k + 1;
    FheHostStage hs;
    void *dg, *dv, *dout;
    if ((rc = hs.up(glev, l * k1 * row, &dg)) != FHE_OK) return rc;
    if ((rc = hs.up(v, batch * l * row, &dv)) != FHE_OK) return rc;
    if ((rc = hs.up(nullptr, batch * k1 * row, &dout)) != FHE_OK) return rc;
    if ((rc = fhe_glev_mul_dev(plan, k, l, dg, dv, dout, batch, 0, hipStreamPerThread)) != FHE_OK) return rc;
    return hs.down(out, dout, batch * k1 * row);
}

extern "C" int fhe_glwe_key_switch(const fhe_ntt_plan *plan, unsigned k, unsigned beta, unsigned l, const uint64_t *glwe, const uint64_t *ksk,
                                   uint64_t *out, size_t batch) {
    if (!plan) return fhe_fail(FHE_E_NULL, "fhe_glwe_key_switch: plan is NULL");
    if (batch == 0) return FHE_OK;
    if (k == 0 || l == 0 || beta < 2) return fhe_fail(FHE_E_INVALID, "fhe_glwe_key_switch: need k, l >= 1, beta >= 2");
    if (!glwe || !ksk || !out) return fhe_fail(FHE_E_NULL, "fhe_glwe_key_switch: NULL buffer");
    int dev, rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    const size_t row = plan->n * 8, k1 = (size_t)k + 1;
    FheHostStage hs;
    void *dg, *dk, *dout;
    if ((rc = hs.up(glwe, batch * k1 * row, &dg)) != FHE_OK) return rc;
    if ((rc = hs.up(ksk, (size_t)k * l * k1 * row, &dk)) != FHE_OK) return rc;
    if ((rc = hs.up(nullptr, batch * k1 * row, &dout)) != FHE_OK) return rc;
    if ((rc = fhe_glwe_key_switch_dev(plan, k, beta, l, dg, dk, dout, batch, 0, hipStreamPerThread)) != FHE_OK) return rc;
    return hs.down(out, dout, batch * k1 * row);
}
