// bfv32.hpp — internal interface of the BFV tensor / relinearisation on two 27-bit primes (bfv32.hip).
// Not part of the public boundary.
#pragma once
#include "digit32.hpp"

namespace fhe {

struct Bfv32Args {
    Ext32Args t;               // per-prime tables of the 2n-point transform: tw_fwd, tw_inv, p, mu, bq, ninv, crt, P
    uint32_t pinv_neg[3];      // -p^-1 mod 2^32 (Montgomery products of two residues)
    Tw32 ninv_mont[3];         // (2n)^-1 * 2^32 mod p: the inverse's scaling with the Montgomery factor folded in
    Tw32 w1ninv_mont[3];       // roots_inv[1] * that: the inverse's last stage and its scaling as ONE product per output (bfv32.hip: last_stage)
    uint32_t log_n2;
    // forward: rows of n 64-bit words (zero-padded to 2n) -> fw[prime][row][2n] u32 for the first `primes` primes
    const u64 *src;
    uint32_t *fw;
    u64 rows;
    uint32_t primes;
    uint32_t word32;           // the source words are below 2^32 (words modulo q; not the relinearisation key)
    uint32_t below_p;          // ... and q <= every prime in use: canonical words (include/fhe_ntt.h: v < q) ARE their residues
    // tensor: fw rows [a0 | a1 | b0 | b1] x batch -> out [c0 | c1 | c2] x batch x n  (scaled by num/den, rounded, folded)
    // relinearisation (three primes): x[prime][batch][2n] (transform of c2), key[prime][2][2n] -> out [o0 | o1] x batch x n
    const uint32_t *x, *key;
    const u64 *addend;         // relinearisation: [c0 | c1] x batch x n
    u64 *out;
    u64 *park;                 // relinearisation: 2 planes x [o0 | o1] x batch x n words, where Garner's digits wait between primes (bfv32.hip)
    u64 batch, q;
    u64 qmu;                   // floor(2^64 / q)
    double numf, denf;
    double rdenf;              // fl(1 / denf) when the epilogue's quotient may be formed as reciprocal + two fma (bfv32.hip: exact_quotient), else 0
    u64 int_num;               // tensor: numf as an integer when numf * v < 2^52 for every coefficient v (integer epilogue), else 0
    uint32_t small_f64;        // |numf * v / denf| < 2^50 for every coefficient: Zq::from_f64 stays in f64 (zq_from_f64_small)
    double qinvf;              // fl(1 / q)
};

// q, n, pq for which the small-prime form applies: the tensor's integers below pA pB, the relinearisation's (pq != 0)
// below pA pB pC (pq = 0: tensor only)
bool bfv32_shape_supported(uint64_t q, uint64_t n, uint64_t pq);
hipError_t launch_bfv32_forward(const Bfv32Args &a, hipStream_t st);
hipError_t launch_bfv32_tensor_inverse(const Bfv32Args &a, hipStream_t st);
hipError_t launch_bfv32_relin_inverse(const Bfv32Args &a, hipStream_t st);

}  // namespace fhe
