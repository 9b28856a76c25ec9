// ntt_kernels.hpp — internal interface between the C ABI (capi.hip) and the
// kernels (ntt_kernels.hip).  Not part of the public boundary (include/fhe_ntt.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zq_device.hpp"

namespace fhe {

// n <= 2^13 runs as ONE contiguous pass (the whole polynomial is one LDS tile);
// larger n is split into a strided and a contiguous pass.
constexpr int kMaxSinglePassLog = 13;
constexpr int kMaxLog = 20;

// Per-device, per-(q,n) constants.  Tables are `n` entries of {w, floor(w*2^64/q)},
// indexed exactly like the reference's `roots_of_unity[m+i]` (ntt.rs:54,88).
struct DevicePlan {
    const Tw *tw_fwd = nullptr;
    const Tw *tw_inv = nullptr;
    const u64 *digit_lut = nullptr;   // tables for transforms of 0/1 polynomials (ntt_rounds.hpp: round0_bits)
    const Tw32 *tw32_fwd = nullptr;   // the same tables in 32-bit words when q < 2^32 / 25 and 2^8 <= n <= 2^17 (smallq.hip)
    const Tw32 *tw32_inv = nullptr;
    Mod mod{};
    Tw ninv{};    // n^-1                     (ntt.rs:27-30)
    Tw s_ninv{};  // roots_inv[1] * n^-1      (last GS stage folded with ntt.rs:100-102)
    uint32_t log_n = 0;
    bool wide = false;  // q < 2^61: forward butterflies correct every other stage
    // Pseudo-Mersenne q = 2^k - delta (zq_device.hpp; the headline modulus 2^61 - 2^21 + 1): the transforms and the
    // products run five-multiply butterflies on a second pair of tables {w, w 2^32 mod q}.  The kernels that
    // transform WHILE loading (digits, reducing loads, zring.hip's epilogues) keep the Shoup tables above.
    const Tw *tw_fwd_pm = nullptr;
    const Tw *tw_inv_pm = nullptr;
    Tw ninv_pm{}, s_ninv_pm{};
    // q = qh 2^32 + 1 below 2^61 (zq_device.hpp, word Montgomery): the plain transforms run on {w 2^32, w 2^64 mod q};
    // nullptr otherwise (or FHE_MG=0).  The plan's arith stays kArWide61 (what the transforming-load kernels run); the launchers of
    // the transforms and of Rq x Rq pick the Montgomery kernels (AR = 4) when these are set.
    const Tw *tw_fwd_mg = nullptr;
    const Tw *tw_inv_mg = nullptr;   // (an inverse transform that multiplies two evaluation operands in its load keeps the Shoup tables)
    Tw ninv_mg{}, s_ninv_mg{};       // n^-1 and roots_inv[1] n^-1 in the same form
    int arith = 0;      // which kernels the plan's transforms run: kArShoup62 ... kArStrict63 below (the template parameter AR of the kernels)
};

// DevicePlan::arith — INTERNAL numbering (the kernels' template parameter AR).  The PUBLIC numbering of
// fhe_ntt_plan_arithmetic() (include/fhe_ntt.h: FHE_ARITH_*) is a different one: never compare the two.
enum : int {
    kArShoup62 = 0,     // q < 2^62: Harvey [0, 4q), Shoup products
    kArWide61 = 1,      // q < 2^61: Shoup products, compile-time bounds up to 8q
    kArPMersenne = 2,   // q = 2^k - delta: five-multiply butterflies on {w, w 2^32 mod q}
    kArStrict63 = 3,    // 2^62 <= q < 2^63: strict butterflies (every value canonical) in the same kernels; n < 16 in generic63.hip
    kArMontgomery = 4,   // never a plan's arith: what launch_ntt_forward / launch_ntt_inverse pass their kernels when tw_*_mg are set (q = 1 mod 2^32)
};

struct PassArgs {
    const u64 *in;
    const u64 *in2;  // second operand of a fused pointwise product, or nullptr
    u64 *out;
    u64 *out2;       // where the fused pointwise product is also stored, or nullptr
    const Tw *tw;
    Mod mod;
    Tw ninv, s_ninv;
    u64 batch;
    uint32_t log_n;
    uint32_t digit_l;    // SRC_DIGITS: output polynomial p is bit (digit_l-1 - p%digit_l) of input row p/digit_l
    uint32_t src_log_n;  // SRC_REDUCE: input rows have 2^src_log_n arbitrary 64-bit words (<= n); the rest is 0
    // SRC_ZQBITS (Zq::decompose, base 2): input row r = (group r / src_grp, member r % src_grp) lives at
    // in + group*src_gstride + member*n, so the k mask rows of each (k+1)-row ciphertext are picked in place
    uint32_t src_grp;
    u64 src_gstride;
    // fused product kernel (rq_mul_fused_kernel) only:
    const Tw *tw_inv;    // inverse table (tw holds the forward one)
    u64 *out3, *out4;    // evals of the two operands, or nullptr
    uint32_t flags;      // bit 0 / 1: operand in / in2 already holds NTT-domain values
    const u64 *lut;      // SRC_DIGITS / SRC_ZQBITS: the plan's digit tables (DevicePlan::digit_lut)
};

// what a forward kernel's load does besides loading
enum : int { SRC_PLAIN = 0, SRC_DIGITS = 1, SRC_REDUCE = 2, SRC_ZQBITS = 3 };

// Brackets one launch with HIP events when fhe_ntt_kernel_timing_enable(1).
struct KernelTimer {
    KernelTimer(const char *name, int tag, hipStream_t st);
    ~KernelTimer();
    void *slot_;
    hipStream_t st_;
};

hipError_t launch_ntt_forward(const DevicePlan &p, const u64 *in, u64 *out, u64 batch,
                              u64 batch_tile, hipStream_t st);
// out[r*l + d] = NTT(bit l-1-d of every word of in[r])  (Tn::decompose, beta = 2, fused into the
// load).  Single-pass sizes and q < 2^61 only: returns hipErrorNotSupported otherwise.
hipError_t launch_ntt_forward_digits(const DevicePlan &p, const u64 *in, u64 *out, u64 rows, uint32_t l,
                                     hipStream_t st);
// out[r*l + d] = NTT(digit d of Zq::decompose(beta = 2, l) of every coefficient of input row r), rows
// gathered as PassArgs::src_grp / src_gstride describe (arith/src/zq.rs:176-190: bit l-1-d, or all ones
// when the value is >= 2^l).  Single-pass sizes and q < 2^61 only: hipErrorNotSupported otherwise.
hipError_t launch_ntt_forward_zqbits(const DevicePlan &p, const u64 *in, u64 *out, u64 rows, uint32_t l,
                                     uint32_t grp, u64 gstride, hipStream_t st);
// out[r] = NTT(in[r] reduced mod q and zero-padded from 2^src_log_n to n words): the operand
// preparation of the exact products over Z (zring.hip) fused into the load.  q < 2^61 and
// n >= 16 only: returns hipErrorNotSupported otherwise.
hipError_t launch_ntt_forward_reduce(const DevicePlan &p, const u64 *in, u64 *out, u64 rows,
                                     uint32_t src_log_n, u64 batch_tile, hipStream_t st);
hipError_t launch_ntt_inverse(const DevicePlan &p, const u64 *in, const u64 *in2, u64 *evals_out,
                              u64 *out, u64 batch, u64 batch_tile, hipStream_t st);
// Two-pass sizes only: the contiguous (first) pass of the inverse transform, without the strided last pass.
hipError_t launch_ntt_inverse_first_pass(const DevicePlan &p, const u64 *in, u64 *out, u64 batch, hipStream_t st);
// c = a * b in Z_q[X]/(X^n+1) for single-pass sizes (16 <= n <= 2^13) in ONE kernel: both forward
// transforms, the pointwise product and the inverse transform of a polynomial stay in registers /
// LDS.  evals_* may be nullptr.  Returns hipErrorNotSupported for other sizes.
hipError_t launch_rq_mul_fused(const DevicePlan &p, const u64 *a, bool a_is_evals, const u64 *b, bool b_is_evals,
                               u64 *c, u64 *c_evals, u64 *a_evals, u64 *b_evals, u64 batch, hipStream_t st);
// The same product at two-pass sizes (2^14 <= n <= 2^20): the two strided forward passes in one launch,
// ONE middle kernel for contiguous-forward(a), contiguous-forward(b), pointwise product and
// contiguous-inverse on each block, then the strided inverse pass — 72n bytes of HBM traffic instead
// of 104n.  wa / wb (batch * n words each) receive the strided pass of a coefficient operand and, when
// keep_*_evals, end up holding that operand's canonical evals; unused for an operand that is evals.
hipError_t launch_rq_mul_two_pass(const DevicePlan &p, const u64 *a, bool a_is_evals, const u64 *b, bool b_is_evals,
                                  u64 *c, u64 *c_evals, u64 *wa, bool keep_a_evals, u64 *wb, bool keep_b_evals,
                                  u64 batch, u64 batch_tile, hipStream_t st);
hipError_t launch_pointwise_mul(const DevicePlan &p, const u64 *x, const u64 *y, u64 *z, u64 count,
                                hipStream_t st);
// 2^62 <= q < 2^63 (DevicePlan::arith == 3) in PLAIN kernels (generic63.hip): strict butterflies, up to four stages per
// launch in global memory.  Since round 5 the transforms and products of such a plan run in the two-pass / fused kernels
// (AR = 3); launch_ntt_forward / launch_ntt_inverse route here only for n < 16 or under FHE_G63_PLAIN=1, launch_pointwise_mul
// always; the transforming-load entry points still return hipErrorNotSupported for such a plan and their callers compose.
hipError_t launch_g63_forward(const DevicePlan &p, const u64 *in, u64 *out, u64 batch, hipStream_t st);
hipError_t launch_g63_inverse(const DevicePlan &p, const u64 *in, const u64 *in2, u64 *evals_out, u64 *out, u64 batch, hipStream_t st);
hipError_t launch_g63_pointwise(const DevicePlan &p, const u64 *x, const u64 *y, u64 *z, u64 count, hipStream_t st);
hipError_t launch_fill_synthetic(u64 *out, u64 count, u64 q, u64 seed, u64 first, hipStream_t st);
hipError_t launch_check_canonical(const u64 *x, u64 count, u64 q, int *d_flag, hipStream_t st);
// true when the translation unit was compiled with one of its timing-only switches (FHE_*_ABLATE_*): its kernels then
// skip work and return wrong words by design; fhe_ntt_version() reports it and the Python binding refuses such a library
bool ntt_kernels_ablated();
bool digit_mac_ablated();
bool digit32_ablated();
bool bfv32_ablated();

}  // namespace fhe
