// smallq.hpp — internal interface of the small-modulus transforms and product (smallq.hip).  Not part of the public boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ntt_kernels.hpp"

namespace fhe {

struct SmallQArgs {
    const Tw32 *tw_fwd, *tw_inv;   // the plan's roots / roots_inv as 32-bit Shoup pairs (DevicePlan::tw32_*)
    uint32_t q, bq;                // modulus, floor(2^32 / q)
    uint32_t qinv_neg;             // -q^-1 mod 2^32
    Tw32 ninv, ninv_mont;          // n^-1 mod q; n^-1 * 2^32 mod q (after a Montgomery product)
    const u64 *a, *b;              // rows of n 64-bit words (b: the product's second operand)
    u64 *out;
    // the product's cached evals (ring_nq.rs:586-607): flags bit 0 / 1 = a / b hold NTT-domain values; optional evals outputs
    uint32_t flags;
    u64 *c_evals, *a_evals, *b_evals;
    u64 mu;                        // floor(2^64 / q)
    uint32_t *mid, *mid_b;         // two-pass sizes (n > 2^14): rows * n u32 words between the passes, per operand (smallq_scratch_bytes)
    u64 rows;
    uint32_t loose;                // 25 q < 2^32: forward butterflies without conditional subtractions (smallq_loose)
};

bool smallq_supported(uint64_t q, unsigned log_n);
bool smallq_loose(uint64_t q);
size_t smallq_scratch_bytes(unsigned log_n, uint64_t rows);
hipError_t launch_sq_forward(const SmallQArgs &a, int log_n, hipStream_t st);
hipError_t launch_sq_inverse(const SmallQArgs &a, int log_n, hipStream_t st);
hipError_t launch_sq_rq_mul(const SmallQArgs &a, int log_n, hipStream_t st);

}  // namespace fhe
