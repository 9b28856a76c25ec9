// digit_mac.hpp — internal interface of the fused decompose -> forward NTT -> multiply-accumulate
// kernel (digit_mac.hip).  Not part of the public boundary (include/fhe_ntt.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ntt_kernels.hpp"

namespace fhe {

struct DigitMacArgs {
    const u64 *src;     // ciphertexts: source row r of ciphertext b at src + b*ct_stride + r*n
    const u64 *key;     // [T][nc][n], NTT domain, canonical (shared by the batch)
    u64 *out;           // [batch][parts][nc][n] canonical partial sums, NTT domain
    const Tw *tw;
    const u64 *lut;     // the plan's digit tables (DevicePlan::digit_lut)
    Mod mod;
    u64 batch, ct_stride;
    uint32_t l;         // digits per source row (beta = 2)
    uint32_t T;         // rows * l digit polynomials per ciphertext
    uint32_t parts;     // workgroups per ciphertext
    uint32_t tpp;       // digits per part (a multiple of the units per workgroup)
};

struct DigitTailArgs {
    const u64 *partial;  // [batch][parts][nc][n] canonical partial sums, NTT domain
    const u64 *src;      // key switch: the input ciphertexts [batch][k+1][n] (their body row enters the tail)
    u64 *out;
    const Tw *tw;        // INVERSE table
    Mod mod;
    Tw ninv, s_ninv;
    u64 batch, half1;    // half1: (P1 + 1) / 2, threshold of the centred lift (torus form)
    uint32_t parts, nc, k;
};

// how many parts to split each ciphertext's T digits into so that `batch` ciphertexts fill the chip
uint32_t digit_mac_parts(u64 batch, uint32_t T, uint32_t log_n, uint32_t nc);

// partial[b][p][c] = sum over part p of KEY[t][c] (.) NTT(digit_t(b)), t = r*l + d over `rows` source rows.
// src_kind: SRC_DIGITS (bit l-1-d of a torus word) or SRC_ZQBITS (Zq::decompose base 2).
// 2^8 <= n <= 2^12, q < 2^61, nc in {2,4} (torus) / {2,3} (Zq), at most 16 accumulators per thread at
// 1024 threads (n = 4096 with nc = 4 is out): hipErrorNotSupported otherwise.
hipError_t launch_digit_mac(const DevicePlan &p, int src_kind, const u64 *src, u64 ct_stride, uint32_t rows, uint32_t l,
                            const u64 *key, uint32_t nc, u64 *partial, uint32_t parts, u64 batch, hipStream_t st);
// sum over the parts + inverse transform + the caller's last step in one kernel (2^8 <= n <= 2^12, q < 2^61):
//   key switch:  out[b][c] = (c < k ? 0 : glwe[b][c]) - intt(sum_p partial[b][p][c]),  c <= k
//   torus:       out[b][c] = lift(intt(S[b][c])) + (lift(intt(S[b][k1 + c])) << 32)  mod 2^64;
//                needs whole ciphertexts per workgroup (4096/n rows): hipErrorNotSupported otherwise
hipError_t launch_digit_tail_ks(const DevicePlan &p, const u64 *partial, uint32_t parts, uint32_t k, const u64 *glwe, u64 *out,
                                u64 batch, hipStream_t st);
hipError_t launch_digit_tail_torus(const DevicePlan &p, const u64 *partial, uint32_t parts, uint32_t k1, u64 half1, u64 *out,
                                   u64 batch, hipStream_t st);
// out[b][.] = sum_p partial[b][p][.]  mod q over row_words = nc * n words per ciphertext
hipError_t launch_sum_parts(const u64 *partial, u64 *out, u64 batch, uint32_t parts, u64 row_words, u64 q, hipStream_t st);

}  // namespace fhe
