// ntt32_big.hpp — a 2^LP-point 32-bit transform held by ONE workgroup (2^11 <= 2^LP <= 2^14 points): the rounds of
// ntt32_rounds.hpp with an LDS tile of the whole transform.  Shared by bfv32.hip (2n-point transforms of the BFV tensor)
// and smallq.hip (n = 8192 / 16384 at small moduli).
#pragma once
#include <type_traits>
#include "ntt32_rounds.hpp"

namespace fhe {

// rounds of a 2^LP-point transform held by ONE workgroup (the index algebra of ContigCfg with W = 1): 2^LP / 16 logical
// threads of 16 coefficients each, VT of them per thread.  VT = 1: a 16384-point transform is a workgroup of 1024
// threads, one per CU (68 KiB of LDS, 128 VGPRs).  VT = 2 (512 threads, two workgroups per CU, so that one computes while
// the other waits at a barrier) needs 32 coefficients + 30 twiddle registers per thread inside the same 128: it spills
// (57-373 registers); 512 threads with 256 registers (one workgroup per CU, more work per thread) ran 2048 BFV
// products in 3.52 ms against 2.86.  VT stays a parameter.
template <int LP>
struct Big32 {
    static constexpr int VT = 1;                               // logical threads (register windows of 16 coefficients) per thread
    static constexpr int M = 1 << LP, TH = M / (16 * VT);
    static constexpr int NR = (LP + 3) / 4, R0 = LP - 4 * (NR - 1), A0 = LP - 4;
    static constexpr int LTW_LOG = 8, LTW_N = 1 << LTW_LOG;
    static constexpr size_t TILE_BYTES = (size_t)(M + M / 16) * 4, TW_BYTES = (size_t)LTW_N * sizeof(Tw32);
    static constexpr int a_of(int j) { return j == 0 ? A0 : LP - R0 - 4 * j; }
    static constexpr int ls0_of(int j) { return j == 0 ? 0 : R0 + 4 * (j - 1); }
    static constexpr bool in_lds(int j) { return ls0_of(j) + (j == 0 ? R0 : 4) <= LTW_LOG; }
    static_assert(LP >= 10 && LP <= 14 && NR >= 3 && NR <= 4, "1024 .. 16384 points");
};

// the register windows of a thread through the tile: barrier (the tile may have been gathered from by the transform or
// exchange before this one), scatter, barrier, gather
template <int LP, int AF, int AT>
__device__ __forceinline__ void exchange_big(u32 (&v)[Big32<LP>::VT][16], u32 *lds, u32 tf) {
    using C = Big32<LP>;
    constexpr u32 TH = C::TH;
    // (slot of register 0) + a constant per register: ntt_rounds.hpp pad16_koff — per register the compiler recomputed the slot
    __syncthreads();
#pragma unroll
    for (int s = 0; s < C::VT; s++) {
        const u32 bf = pad16(field_of<AF>(tf + s * TH, 0));
#pragma unroll
        for (int k = 0; k < 16; k++) lds[bf + pad16_koff<AF>(k)] = v[s][k];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < C::VT; s++) {
        const u32 bt = pad16(field_of<AT>(tf + s * TH, 0));
#pragma unroll
        for (int k = 0; k < 16; k++) v[s][k] = lds[bt + pad16_koff<AT>(k)];
    }
}

// Forward stages with ct32_loose: the bound of the values grows by 2p per stage from B·p and must stay below 25p, so
// the values are brought below 2p (barrett2p_32) before a round that would pass it.  Round 0 starts at stage I0.
// Ends with the values in window [0,4) (16 consecutive points per logical thread), below 25p.
// Block (s0, blk): the 2^LP points are block `blk` of a larger transform whose first s0 stages were done elsewhere
// (smallq.hip's two-pass sizes); `ltw` is then the block's LOCAL table (stage_tw32_block), indexed as if the block were a
// transform of its own, and the global table is indexed at (1 << (s0 + ls)) + (blk << ls) + H.  (0, 0): a whole transform.
template <int LP, int J>
__device__ __forceinline__ u32 big_t0(u32 s0, u32 blk, u32 H) {
    using C = Big32<LP>;
    constexpr int LS = C::ls0_of(J);
    return C::in_lds(J) ? (1u << LS) + H : (1u << (s0 + LS)) + (blk << LS) + H;
}
template <int TH>
__device__ __forceinline__ void stage_tw32_block(Tw32 *ltw, const Tw32 *__restrict__ tw, int count, u32 tid, u32 s0, u32 blk) {
    for (u32 li = tid; li < (u32)count; li += TH) {
        const u32 ls = 31u - (u32)__builtin_clz(li | 1u);       // entry 0 is never used: a copy of entry 1
        const u32 l1 = li | (li == 0);
        ltw[li] = tw[(1u << (s0 + ls)) + (blk << ls) + (l1 - (1u << ls))];
    }
}
// PRELOAD = false: the rounds read their table as they go instead of holding a round's 30 twiddle words across the
// exchange (fewer registers: more workgroups per CU, which then cover the table latency)
// LOOSE = false (moduli up to 2^30, smallq.hip): Harvey's butterflies, values in [0, 4p) throughout, nothing to bring down
template <int LP, int J, int B, bool PRELOAD = true, bool LOOSE = true>
__device__ __forceinline__ void fwd_round_big(u32 (&v)[Big32<LP>::VT][16], u32 *lds, const Tw32 *ltw, const Tw32 *gtw, u32 tf, u32 p, u32 p2, u32 bq,
                                              u32 s0 = 0, u32 blk = 0) {
    using C = Big32<LP>;
    if constexpr (J < C::NR) {
        constexpr int A = C::a_of(J);
        Tw32 t[C::VT][15];                                      // requested before the exchange: see load_tw32
        if constexpr (PRELOAD) {
#pragma unroll
            for (int s = 0; s < C::VT; s++) load_tw32<4>(t[s], C::in_lds(J) ? ltw : gtw, big_t0<LP, J>(s0, blk, (tf + s * C::TH) >> A));
        }
        exchange_big<LP, C::a_of(J - 1), A>(v, lds, tf);
        constexpr bool RED = LOOSE && B + 8 > 25;
#pragma unroll
        for (int s = 0; s < C::VT; s++) {
            if constexpr (RED) {
#pragma unroll
                for (int k = 0; k < 16; k++) v[s][k] = barrett2p_32(v[s][k], p, bq);
            }
            if constexpr (PRELOAD) round_fwd32_tw<4, 0, LOOSE>(v[s], t[s], p, p2);
            else round_fwd32<4, 0, LOOSE>(v[s], C::in_lds(J) ? ltw : gtw, big_t0<LP, J>(s0, blk, (tf + s * C::TH) >> A), p, p2);
        }
        fwd_round_big<LP, J + 1, (RED ? 2 : B) + 8, PRELOAD, LOOSE>(v, lds, ltw, gtw, tf, p, p2, bq, s0, blk);
    }
}
// B0: the bound (in p) of the values on entry
template <int LP, int I0, int B0 = 1, bool PRELOAD = true, bool LOOSE = true>
__device__ __forceinline__ void fwd_big(u32 (&v)[Big32<LP>::VT][16], u32 *lds, const Tw32 *ltw, const Tw32 *gtw, u32 tf, u32 p, u32 p2, u32 bq,
                                        u32 s0 = 0, u32 blk = 0) {
    using C = Big32<LP>;
#pragma unroll
    for (int s = 0; s < C::VT; s++) round_fwd32<C::R0, I0, LOOSE>(v[s], ltw, 1u, p, p2);
    fwd_round_big<LP, 1, B0 + 2 * C::R0, PRELOAD, LOOSE>(v, lds, ltw, gtw, tf, p, p2, bq, s0, blk);
}
// Inverse stages (gs32: values below 2p throughout): window [0,4) -> window [LP-4, LP), not yet scaled.  `t` holds the
// twiddles of round J on entry (requested by the caller / the round before, ahead of the exchange).
template <int LP, int J>
__device__ __forceinline__ void inv_round_big(u32 (&v)[Big32<LP>::VT][16], Tw32 (&t)[Big32<LP>::VT][15], u32 *lds, const Tw32 *ltw, const Tw32 *gtw,
                                              u32 tf, u32 p, u32 p2, u32 s0, u32 blk) {
    using C = Big32<LP>;
    constexpr int A = C::a_of(J);
#pragma unroll
    for (int s = 0; s < C::VT; s++) round_inv32_tw<4>(v[s], t[s], p, p2);
    if constexpr (J > 1) {
        constexpr int AN = C::a_of(J - 1);
#pragma unroll
        for (int s = 0; s < C::VT; s++) load_tw32<4>(t[s], C::in_lds(J - 1) ? ltw : gtw, big_t0<LP, J - 1>(s0, blk, (tf + s * C::TH) >> AN));
    }
    exchange_big<LP, A, C::a_of(J - 1)>(v, lds, tf);
}
// The same without the 30 twiddle registers: every round reads its table (LDS tile or global) as it goes.  For kernels
// that hold other results across the transform (bfv32.hip's block kernels: residues of the primes done so far) and run
// two workgroups per CU, where the other workgroup covers the table latency the preloading was there to hide.
// STAGED: a scheduling barrier after every stage (round_inv32_staged), so that at most 8 table entries are in flight
template <int LP, int J, bool STAGED>
__device__ __forceinline__ void inv_round_big_direct(u32 (&v)[Big32<LP>::VT][16], u32 *lds, const Tw32 *ltw, const Tw32 *gtw, u32 tf, u32 p, u32 p2,
                                                     u32 s0, u32 blk) {
    using C = Big32<LP>;
    constexpr int A = C::a_of(J);
#pragma unroll
    for (int s = 0; s < C::VT; s++) {
        if constexpr (STAGED) round_inv32_staged<4>(v[s], C::in_lds(J) ? ltw : gtw, big_t0<LP, J>(s0, blk, (tf + s * C::TH) >> A), p, p2);
        else round_inv32<4>(v[s], C::in_lds(J) ? ltw : gtw, big_t0<LP, J>(s0, blk, (tf + s * C::TH) >> A), p, p2);
    }
    exchange_big<LP, A, C::a_of(J - 1)>(v, lds, tf);
}
template <int LP, bool STAGED = true>
__device__ __forceinline__ void inv_big_direct(u32 (&v)[Big32<LP>::VT][16], u32 *lds, const Tw32 *ltw, const Tw32 *gtw, u32 tf, u32 p, u32 p2,
                                               u32 s0 = 0, u32 blk = 0) {
    using C = Big32<LP>;
    if constexpr (C::NR > 3) inv_round_big_direct<LP, 3, STAGED>(v, lds, ltw, gtw, tf, p, p2, s0, blk);
    inv_round_big_direct<LP, 2, STAGED>(v, lds, ltw, gtw, tf, p, p2, s0, blk);
    inv_round_big_direct<LP, 1, STAGED>(v, lds, ltw, gtw, tf, p, p2, s0, blk);
#pragma unroll
    for (int s = 0; s < C::VT; s++) round_inv32<C::R0>(v[s], ltw, 1u, p, p2);
}
// inv_big_direct on the loose rounds (ntt32_rounds.hpp: round_inv32_loose; moduli below 2^32 / 25): inputs below 2p, rounds hand on
// values below 4p, the result is below BOUT p (2 as inv_big_direct's; a caller whose next step takes larger words asks for more)
template <int LP, bool STAGED = true, int BOUT = 2>
__device__ __forceinline__ void inv_big_loose(u32 (&v)[Big32<LP>::VT][16], u32 *lds, const Tw32 *ltw, const Tw32 *gtw, u32 tf, u32 p, u32 bq,
                                              u32 s0 = 0, u32 blk = 0) {
    using C = Big32<LP>;
    auto round = [&](auto jc, auto binc) __attribute__((always_inline)) {
        constexpr int J = decltype(jc)::value, BIN = decltype(binc)::value, A = C::a_of(J);
#pragma unroll
        for (int s = 0; s < C::VT; s++)
            round_inv32_loose<4, BIN, 4, STAGED>(v[s], C::in_lds(J) ? ltw : gtw, big_t0<LP, J>(s0, blk, (tf + s * C::TH) >> A), p, bq);
        exchange_big<LP, A, C::a_of(J - 1)>(v, lds, tf);
    };
    if constexpr (C::NR > 3) {
        round(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});
        round(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
    } else {
        round(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
    }
    round(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
#pragma unroll
    for (int s = 0; s < C::VT; s++) round_inv32_loose<C::R0, 4, BOUT, false>(v[s], ltw, 1u, p, bq);
}
template <int LP>
__device__ __forceinline__ void inv_big(u32 (&v)[Big32<LP>::VT][16], u32 *lds, const Tw32 *ltw, const Tw32 *gtw, u32 tf, u32 p, u32 p2,
                                        u32 s0 = 0, u32 blk = 0) {
    using C = Big32<LP>;
    Tw32 t[C::VT][15];
    constexpr int JT = C::NR - 1;
#pragma unroll
    for (int s = 0; s < C::VT; s++) load_tw32<4>(t[s], C::in_lds(JT) ? ltw : gtw, big_t0<LP, JT>(s0, blk, (tf + s * C::TH) >> C::a_of(JT)));
    if constexpr (C::NR > 3) inv_round_big<LP, 3>(v, t, lds, ltw, gtw, tf, p, p2, s0, blk);
    inv_round_big<LP, 2>(v, t, lds, ltw, gtw, tf, p, p2, s0, blk);
    inv_round_big<LP, 1>(v, t, lds, ltw, gtw, tf, p, p2, s0, blk);
#pragma unroll
    for (int s = 0; s < C::VT; s++) round_inv32<C::R0>(v[s], ltw, 1u, p, p2);
}

}  // namespace fhe
