// digit32.hpp — internal interface of the 27-bit-prime external product (digit32.hip).  Not part of the public boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ntt_kernels.hpp"

namespace fhe {


// the two primes: the largest below 2^32 / 25 with p = 1 mod 2^15 (27.36 and 27.35 bits; product 2^54.7).  Below
// 2^32 / 25 so that up to twelve lazy butterfly stages need no conditional subtraction (digit32.hip: ct32_loose).
constexpr uint32_t kExt32PrimeA = 0x0a3c8001u, kExt32PrimeB = 0x0a320001u;
// the next one down, for the products that need three (bfv32.hip: relinearisation, integers below 2^82)
constexpr uint32_t kExt32PrimeC = 0x0a318001u;
static_assert((uint64_t)kExt32PrimeA * 25 < (1ull << 32) && kExt32PrimeA > kExt32PrimeB && kExt32PrimeA - kExt32PrimeB < kExt32PrimeB, "prime bounds");

struct Ext32Args {
    // key preparation
    const u64 *key64;          // [T][key_k1][n] 64-bit words (the key as the reference holds it)
    uint32_t *key32;           // [prime][T][half][key_k1][n], NTT domain
    u64 rows;                  // T * 2 * key_k1
    uint32_t key_k1;
    // product
    const u64 *src;            // ciphertexts, source row r of ciphertext b at src + b*ct_stride + r*n
    u64 ct_stride;
    uint32_t *part32;          // [batch][parts][prime][NC][n] canonical partial sums
    u64 *out;                  // [batch][k+1][n]
    u64 batch;
    uint32_t l, T, parts, tpp;
    // per prime
    const Tw32 *tw_fwd[3], *tw_inv[3];
    const uint32_t *lut[3];
    uint32_t p[3];             // kExt32PrimeA, B, C (digit32.hip uses the first two)
    u64 mu[3];                 // floor(2^64 / p)
    uint32_t bq[3];            // floor(2^32 / p)
    Tw32 ninv[3];              // n^-1 mod p
    Tw32 crt;                  // pA^-1 mod pB
    Tw32 crt_ac, crt_bc;       // pA^-1 mod pC, pB^-1 mod pC (Garner's third digit)
    u64 P, halfP;              // pA * pB, ceil(P / 2)
    // key switching tail (digit_tail32_ks_kernel)
    Mod mod;                   // the ring's q
    u64 two32;                 // 2^32 mod q
    const u64 *glwe;           // the ciphertexts again, for the body row
    uint32_t k;
};

bool ext32_shape_supported(u64 n, unsigned k, unsigned l);        // TGGSW x TGLWE
bool ks32_shape_supported(u64 n, unsigned k, unsigned l);         // GLWE::key_switch, base 2
uint32_t ext32_units(int log_n);                                   // digits per step of the fused kernel
hipError_t launch_ext32_key(const Ext32Args &a, int log_n, hipStream_t st);
hipError_t launch_ext32_mac(const Ext32Args &a, int log_n, int src_kind, hipStream_t st);
hipError_t launch_ext32_tail(const Ext32Args &a, int log_n, hipStream_t st);
hipError_t launch_ext32_tail_ks(const Ext32Args &a, int log_n, hipStream_t st);

}  // namespace fhe
