// smallq.hip — NTT::ntt / NTT::intt / Rq x Rq (arith/src/ntt.rs:44-104, ring_nq.rs:586-607) for SMALL moduli in 32-bit words.
//
// The reference's own tests and its BFV / GLWE demos run at q = 65537 (and 12289, 1021, ...): 64-bit Shoup butterflies —
// ten 32-bit multiplies and ~22 instructions each — for 17-bit numbers.  For an NTT-friendly q below 2^30 the same
// transform (same psi, same tables, same bit-reversed layout: the plan's) runs in one word per coefficient with the
// butterflies of ntt32_rounds.hpp: 3 multiplies + 3 additions and, below 2^32 / 25, no conditional subtraction on the forward side
// (between 2^32 / 25 and 2^30 — round 3 — Harvey's form with one per butterfly: 4 q fits the word).  The
// interface stays 64-bit words; a transform is then bound by its 16 n bytes of traffic instead of by multiplier issue.
// Two passes with a u32 intermediate for 2^15 <= n <= 2^18; single-workgroup sizes 2^8 <= n <= 2^14 (256 threads holding W = 4096 / n polynomials up to n = 4096; n = 8192 / 16384 as
// one workgroup of n / 16 threads around a whole-transform LDS tile, ntt32_big.hpp); everything else (and FHE_EXT32=0)
// keeps the 61-bit kernels (2^15 points — 1024 threads x 32 coefficients in 128 registers — were tried: 64-143 registers
// spilled, 3.8 M NTT/s against 4.8 M on the 61-bit two-pass kernels).  Same values, word for word: every result is canonical
// modulo the same q.
// NOT covered, and therefore on the 61-bit kernels (stated here and in include/fhe_ntt.h): moduli between
// 2^30 and 2^32 (4 q no longer fits a word), n < 2^8 and n >= 2^19 (a second strided level), and n = 8192 / 16384 run as ONE 1024-thread workgroup per
// CU (4.7 / 3.9 TB/s where the 256-thread sizes reach 5.5 - 5.8).
//   sq_forward_kernel   n words in (natural order) -> forward transform -> n words out (the reference's bit-reversed order)
//   sq_inverse_kernel   the inverse, n^-1 folded in
//   sq_rq_mul_kernel    both forward transforms in lockstep (one twiddle load for both), pointwise Montgomery product,
//                       inverse transform — the whole product on chip, as rq_mul_fused_kernel does for 61-bit q
#include "smallq.hpp"
#include "ntt32_big.hpp"

namespace fhe {

template <int LP>
struct SqCfg {
    using C = ContigCfg<LP>;
    static constexpr int M = C::M, W = C::W, TPB = C::TPB;
    static constexpr size_t TILE_BYTES = (size_t)(W * M + W * M / 16) * 4, TW_BYTES = (size_t)C::LTW_N * sizeof(Tw32);
    static_assert(LP >= 8 && LP <= 12, "n = 256 .. 4096");
};

// two polynomials through two tiles with one pair of barriers
template <int LP, int AF, int AT, bool FIRST>
__device__ __forceinline__ void sq_exchange2(u32 (&va)[16], u32 (&vb)[16], u32 *la, u32 *lb, u32 w, u32 tf) {
    constexpr int M = 1 << LP;
    const u32 bf = pad16(w * M + field_of<AF>(tf, 0)), bt = pad16(w * M + field_of<AT>(tf, 0));   // + a constant per register (ntt_rounds.hpp: pad16_koff)
    if (!FIRST) __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u32 s = bf + pad16_koff<AF>(k);
        la[s] = va[k];
        lb[s] = vb[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u32 s = bt + pad16_koff<AT>(k);
        va[k] = la[s];
        vb[k] = lb[s];
    }
}

// forward stages of two polynomials in lockstep, ct32_loose (canonical inputs: values end below (1 + 2 LP) q <= 25 q)
// LOOSE (25 q < 2^32): ct32_loose, no conditional subtraction; otherwise (4 q < 2^32: moduli up to 2^30) Harvey's
// butterflies with one conditional subtraction each, values in [0, 4q) throughout
template <int LP, bool LOOSE>
__device__ __forceinline__ void sq_fwd2(u32 (&va)[16], u32 (&vb)[16], u32 *la, u32 *lb, const Tw32 *ltw, const Tw32 *gtw, u32 w, u32 tf,
                                        u32 q, u32 q2) {
    using C = ContigCfg<LP>;
    Tw32 t[15];
    load_tw32<C::R0>(t, gtw, 1u);
    round_fwd32_tw<C::R0, 0, LOOSE>(va, t, q, q2);
    round_fwd32_tw<C::R0, 0, LOOSE>(vb, t, q, q2);
    {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        load_tw32<4>(t, C::in_lds(1) ? ltw : gtw, (1u << LS) + (tf >> A));
        sq_exchange2<LP, C::A0, A, true>(va, vb, la, lb, w, tf);
        round_fwd32_tw<4, 0, LOOSE>(va, t, q, q2);
        round_fwd32_tw<4, 0, LOOSE>(vb, t, q, q2);
    }
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        load_tw32<4>(t, C::in_lds(2) ? ltw : gtw, (1u << LS) + (tf >> A));
        sq_exchange2<LP, C::a_of(1), A, false>(va, vb, la, lb, w, tf);
        round_fwd32_tw<4, 0, LOOSE>(va, t, q, q2);
        round_fwd32_tw<4, 0, LOOSE>(vb, t, q, q2);
    }
    static_assert(C::NR >= 2 && C::NR <= 3, "256 .. 4096 points");
}

// forward stages of one polynomial (window [LP-4, LP) -> [0,4)); FRESH: nobody has touched the tile
template <int LP, bool LOOSE>
__device__ __forceinline__ void sq_fwd1(u32 (&v)[16], u32 *lds, const Tw32 *ltw, const Tw32 *gtw, u32 w, u32 tf, u32 q, u32 q2) {
    using C = ContigCfg<LP>;
    Tw32 t[15];
    load_tw32<C::R0>(t, gtw, 1u);
    round_fwd32_tw<C::R0, 0, LOOSE>(v, t, q, q2);
    {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        load_tw32<4>(t, C::in_lds(1) ? ltw : gtw, (1u << LS) + (tf >> A));
        exchange32<LP, C::A0, A, true>(v, lds, w, tf);
        round_fwd32_tw<4, 0, LOOSE>(v, t, q, q2);
    }
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        load_tw32<4>(t, C::in_lds(2) ? ltw : gtw, (1u << LS) + (tf >> A));
        exchange32<LP, C::a_of(1), A, false>(v, lds, w, tf);
        round_fwd32_tw<4, 0, LOOSE>(v, t, q, q2);
    }
}
// inverse stages (gs32, values below 2q throughout): window [0,4) -> [LP-4, LP); FRESH: the tile is untouched
template <int LP, bool FRESH>
__device__ __forceinline__ void sq_inv1(u32 (&v)[16], u32 *lds, const Tw32 *ltw, const Tw32 *gtw, u32 w, u32 tf, u32 q, u32 q2) {
    using C = ContigCfg<LP>;
    Tw32 t[15];
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        load_tw32<4>(t, C::in_lds(2) ? ltw : gtw, (1u << LS) + (tf >> A));
        round_inv32_tw<4>(v, t, q, q2);
        constexpr int A1 = C::a_of(1), LS1 = C::ls0_of(1);
        load_tw32<4>(t, C::in_lds(1) ? ltw : gtw, (1u << LS1) + (tf >> A1));      // ahead of the exchange
        exchange32<LP, A, A1, FRESH>(v, lds, w, tf);
    } else {
        constexpr int A1 = C::a_of(1), LS1 = C::ls0_of(1);
        load_tw32<4>(t, C::in_lds(1) ? ltw : gtw, (1u << LS1) + (tf >> A1));
    }
    {
        constexpr int A = C::a_of(1);
        round_inv32_tw<4>(v, t, q, q2);
        load_tw32<C::R0>(t, gtw, 1u);
        exchange32<LP, A, C::A0, FRESH && (C::NR <= 2)>(v, lds, w, tf);
    }
    round_inv32_tw<C::R0>(v, t, q, q2);
}

// a row of n 64-bit words at the positions of window [LP-4, LP): register k = word k * TPB + tf (512 bytes per wave)
template <int LP>
__device__ __forceinline__ void sq_load_natural(u32 (&v)[16], const u64 *__restrict__ src, u32 tf, u32 q, u32 bq) {
    using C = ContigCfg<LP>;
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = csub_u32(barrett2p_32((u32)src[(u32)k * C::TPB + tf], q, bq), q);   // words are below q; any 32-bit word is reduced
}

// x * y * 2^-32 mod q, in [0, 2q), for x * y < q * 2^32 (Montgomery; q is odd: 2n divides q - 1)
__device__ __forceinline__ u32 sq_mont(u32 x, u32 y, u32 q, u32 qinv_neg) {
    const u64 t = (u64)x * y;
    const u32 m = (u32)t * qinv_neg;
    return (u32)((t + (u64)m * q) >> 32);
}

struct SqLds {
    u32 *tile_a, *tile_b;
    Tw32 *ltw;
};
template <int LP, bool TWO>
__device__ __forceinline__ SqLds sq_lds(unsigned char *smem, const Tw32 *gtw, u32 tid) {
    using K = SqCfg<LP>;
    SqLds l;
    l.tile_a = reinterpret_cast<u32 *>(smem);
    l.tile_b = reinterpret_cast<u32 *>(smem + K::TILE_BYTES);
    l.ltw = reinterpret_cast<Tw32 *>(smem + (TWO ? 2 : 1) * K::TILE_BYTES);
    stage_tw32(l.ltw, gtw, ContigCfg<LP>::LTW_N, tid);
    __syncthreads();
    return l;
}

// ---- forward: natural order in, the reference's bit-reversed order out --------------------------------------------------
template <int LP, bool LOOSE>
__global__ __launch_bounds__(256) void sq_forward_kernel(SmallQArgs a) {
    using C = ContigCfg<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const u32 tid = threadIdx.x, w = tid / C::TPB, tf = tid % C::TPB;
    const SqLds l = sq_lds<LP, false>(smem_raw, a.tw_fwd, tid);
    const u64 R0 = (u64)blockIdx.x * C::W;
    const u64 row = min(R0 + w, a.rows - 1);                    // idle units redo the last row and store nothing
    u32 v[16];
    sq_load_natural<LP>(v, a.a + row * C::M, tf, a.q, a.bq);
    sq_fwd1<LP, LOOSE>(v, l.tile_a, l.ltw, a.tw_fwd, w, tf, a.q, 2u * a.q);
    // window [0,4): register k = output word 16 tf + k.  Through the tile so that a wave stores 512 contiguous bytes
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) l.tile_a[pad16(w * C::M + tf * 16u + k)] = csub_u32(barrett2p_32(v[k], a.q, a.bq), a.q);
    __syncthreads();
    if (R0 + w < a.rows) {
        u64 *__restrict__ dst = a.out + row * C::M;
#pragma unroll
        for (int k = 0; k < 16; k++) dst[(u32)k * C::TPB + tf] = l.tile_a[pad16(w * C::M + (u32)k * C::TPB + tf)];
    }
}

// ---- inverse: bit-reversed order in, natural order out, n^-1 folded in --------------------------------------------------
template <int LP>
__global__ __launch_bounds__(256) void sq_inverse_kernel(SmallQArgs a) {
    using C = ContigCfg<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const u32 tid = threadIdx.x, w = tid / C::TPB, tf = tid % C::TPB;
    const SqLds l = sq_lds<LP, false>(smem_raw, a.tw_inv, tid);
    const u64 R0 = (u64)blockIdx.x * C::W;
    const u64 row = min(R0 + w, a.rows - 1);
    {   // coalesced loads, through the tile into window [0,4)
        const u64 *__restrict__ src = a.a + row * C::M;
#pragma unroll
        for (int k = 0; k < 16; k++)
            l.tile_a[pad16(w * C::M + (u32)k * C::TPB + tf)] = csub_u32(barrett2p_32((u32)src[(u32)k * C::TPB + tf], a.q, a.bq), a.q);
    }
    __syncthreads();
    u32 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = l.tile_a[pad16(w * C::M + tf * 16u + k)];
    sq_inv1<LP, false>(v, l.tile_a, l.ltw, a.tw_inv, w, tf, a.q, 2u * a.q);
    if (R0 + w < a.rows) {
        u64 *__restrict__ dst = a.out + row * C::M;
#pragma unroll
        for (int k = 0; k < 16; k++) dst[(u32)k * C::TPB + tf] = csub_u32(mul_shoup32(v[k], a.ninv, a.q), a.q);
    }
}

// ---- Rq x Rq: forward(a), forward(b), pointwise product, inverse — one kernel -------------------------------------------
// ring_nq.rs:586-607 with its cached evals: an operand flagged as evals (flags bit 0 / 1) is read as such (no forward
// transform); c_evals / a_evals / b_evals, when given, receive the canonical transforms of the product and of the operands
// (ring_nq.rs:568-573,606).  NTT-domain rows move through the tile so that a wave touches 512 contiguous bytes.
template <int LP, bool LOOSE>
__global__ __launch_bounds__(256) void sq_rq_mul_kernel(SmallQArgs a) {
    using C = ContigCfg<LP>;
    using K = SqCfg<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const u32 tid = threadIdx.x, w = tid / C::TPB, tf = tid % C::TPB;
    u32 *tile_a = reinterpret_cast<u32 *>(smem_raw), *tile_b = reinterpret_cast<u32 *>(smem_raw + K::TILE_BYTES);
    Tw32 *ltw = reinterpret_cast<Tw32 *>(smem_raw + 2 * K::TILE_BYTES), *ltw_inv = ltw + C::LTW_N;
    stage_tw32(ltw, a.tw_fwd, C::LTW_N, tid);
    stage_tw32(ltw_inv, a.tw_inv, C::LTW_N, tid);
    __syncthreads();
    const u64 R0 = (u64)blockIdx.x * C::W;
    const bool active = R0 + w < a.rows;
    const u64 row = min(R0 + w, a.rows - 1);                    // idle units redo the last row and store nothing
    const u32 q = a.q, q2 = 2u * q;
    // evals in: coalesced words into the (free) tile, then the thread's 16 consecutive values — window [0,4)
    auto load_evals = [&](u32 (&v)[16], const u64 *__restrict__ src, u32 *tile) {
#pragma unroll
        for (int k = 0; k < 16; k++) tile[pad16(w * C::M + (u32)k * C::TPB + tf)] = csub_u32(barrett2p_32((u32)src[(u32)k * C::TPB + tf], q, a.bq), q);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = tile[pad16(w * C::M + tf * 16u + k)];
    };
    // evals out (canonical window-[0,4) values): barrier first — the tile may still be gathered from
    auto store_evals = [&](u64 *__restrict__ dst, const u32 (&v)[16], u32 *tile) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) tile[pad16(w * C::M + tf * 16u + k)] = v[k];
        __syncthreads();
        if (active) {
#pragma unroll
            for (int k = 0; k < 16; k++) dst[row * C::M + (u32)k * C::TPB + tf] = tile[pad16(w * C::M + (u32)k * C::TPB + tf)];
        }
    };
    u32 va[16], vb[16];
    if (a.flags == 0u) {
        sq_load_natural<LP>(va, a.a + row * C::M, tf, q, a.bq);
        sq_load_natural<LP>(vb, a.b + row * C::M, tf, q, a.bq);
        sq_fwd2<LP, LOOSE>(va, vb, tile_a, tile_b, ltw, a.tw_fwd, w, tf, q, q2);
    } else {
        if (a.flags & 1u) load_evals(va, a.a + row * C::M, tile_a);
        else { sq_load_natural<LP>(va, a.a + row * C::M, tf, q, a.bq); sq_fwd1<LP, LOOSE>(va, tile_a, ltw, a.tw_fwd, w, tf, q, q2); }
        if (a.flags & 2u) load_evals(vb, a.b + row * C::M, tile_b);
        else { sq_load_natural<LP>(vb, a.b + row * C::M, tf, q, a.bq); sq_fwd1<LP, LOOSE>(vb, tile_b, ltw, a.tw_fwd, w, tf, q, q2); }
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {                              // canonical: what the evals outputs hold
        va[k] = csub_u32(barrett2p_32(va[k], q, a.bq), q);
        vb[k] = csub_u32(barrett2p_32(vb[k], q, a.bq), q);
    }
    if (a.a_evals) store_evals(a.a_evals, va, tile_a);
    if (a.b_evals) store_evals(a.b_evals, vb, tile_b);
    Tw32 scale = a.ninv_mont;
    if (a.c_evals) {                                            // the product itself, canonical (zip_eq(l,r).map(l*r), ring_nq.rs:601-604)
#pragma unroll
        for (int k = 0; k < 16; k++) va[k] = reduce64_32((u64)va[k] * vb[k], q, a.mu);
        store_evals(a.c_evals, va, tile_b);
        scale = a.ninv;
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) va[k] = sq_mont(va[k], vb[k], q, a.qinv_neg);       // * 2^-32: leaves with n^-1 below
    }
    sq_inv1<LP, false>(va, tile_a, ltw_inv, a.tw_inv, w, tf, q, q2);
    if (active) {
        u64 *__restrict__ dst = a.out + row * C::M;
#pragma unroll
        for (int k = 0; k < 16; k++) dst[(u32)k * C::TPB + tf] = csub_u32(mul_shoup32(va[k], scale, q), q);
    }
}

// ---- n = 8192 / 16384: one workgroup of n / 16 threads per polynomial (ntt32_big.hpp) ------------------------------------
template <int LP>
__device__ __forceinline__ const Tw32 *sq_big_stage(unsigned char *smem, const Tw32 *gtw, u32 tid, int slot) {
    using C = Big32<LP>;
    Tw32 *ltw = reinterpret_cast<Tw32 *>(smem + C::TILE_BYTES + slot * C::TW_BYTES);
    stage_tw32<C::TH>(ltw, gtw, C::LTW_N, tid);
    return ltw;
}
template <int LP>
__device__ __forceinline__ void sq_big_load(u32 (&v)[1][16], const u64 *__restrict__ src, u32 tf, u32 q, u32 bq) {
#pragma unroll
    for (int k = 0; k < 16; k++) v[0][k] = csub_u32(barrett2p_32((u32)src[(u32)k * Big32<LP>::TH + tf], q, bq), q);
}
template <int LP, bool LOOSE>
__global__ __launch_bounds__((Big32<LP>::TH)) void sq_big_forward_kernel(SmallQArgs a) {
    using C = Big32<LP>;
    static_assert(C::VT == 1, "one register window per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *lds = reinterpret_cast<u32 *>(smem_raw);
    const u32 tf = threadIdx.x;
    const Tw32 *ltw = sq_big_stage<LP>(smem_raw, a.tw_fwd, tf, 0);
    __syncthreads();
    const u64 row = blockIdx.x;
    u32 v[1][16];
    sq_big_load<LP>(v, a.a + row * C::M, tf, a.q, a.bq);
    fwd_big<LP, 0, 1, true, LOOSE>(v, lds, ltw, a.tw_fwd, tf, a.q, 2u * a.q, a.bq);
    __syncthreads();                                            // window [0,4) out through the tile: 512 contiguous bytes per wave
#pragma unroll
    for (int k = 0; k < 16; k++) lds[pad16(tf * 16u + k)] = csub_u32(barrett2p_32(v[0][k], a.q, a.bq), a.q);
    __syncthreads();
    u64 *__restrict__ dst = a.out + row * C::M;
#pragma unroll
    for (int k = 0; k < 16; k++) dst[(u32)k * C::TH + tf] = lds[pad16((u32)k * C::TH + tf)];
}
template <int LP>
__global__ __launch_bounds__((Big32<LP>::TH)) void sq_big_inverse_kernel(SmallQArgs a) {
    using C = Big32<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *lds = reinterpret_cast<u32 *>(smem_raw);
    const u32 tf = threadIdx.x;
    const Tw32 *ltw = sq_big_stage<LP>(smem_raw, a.tw_inv, tf, 0);
    const u64 row = blockIdx.x;
    const u64 *__restrict__ src = a.a + row * C::M;
#pragma unroll
    for (int k = 0; k < 16; k++) lds[pad16((u32)k * C::TH + tf)] = csub_u32(barrett2p_32((u32)src[(u32)k * C::TH + tf], a.q, a.bq), a.q);
    __syncthreads();
    u32 v[1][16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[0][k] = lds[pad16(tf * 16u + k)];
    inv_big<LP>(v, lds, ltw, a.tw_inv, tf, a.q, 2u * a.q);      // its exchanges start with a barrier
    u64 *__restrict__ dst = a.out + row * C::M;
#pragma unroll
    for (int k = 0; k < 16; k++) dst[(u32)k * C::TH + tf] = csub_u32(mul_shoup32(v[0][k], a.ninv, a.q), a.q);
}
template <int LP, bool LOOSE>
__global__ __launch_bounds__((Big32<LP>::TH)) void sq_big_rq_mul_kernel(SmallQArgs a) {
    using C = Big32<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *lds = reinterpret_cast<u32 *>(smem_raw);
    const u32 tf = threadIdx.x;
    const Tw32 *ltw = sq_big_stage<LP>(smem_raw, a.tw_fwd, tf, 0);
    const Tw32 *ltw_inv = sq_big_stage<LP>(smem_raw, a.tw_inv, tf, 1);
    __syncthreads();
    const u64 row = blockIdx.x;
    const u32 q = a.q, q2 = 2u * q;
    // cached evals (see sq_rq_mul_kernel): NTT-domain rows through the tile, barrier first (it may still be gathered from)
    auto load_evals = [&](u32 (&v)[1][16], const u64 *__restrict__ src) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) lds[pad16((u32)k * C::TH + tf)] = csub_u32(barrett2p_32((u32)src[(u32)k * C::TH + tf], q, a.bq), q);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) v[0][k] = lds[pad16(tf * 16u + k)];
    };
    auto store_evals = [&](u64 *__restrict__ dst, const u32 (&v)[1][16]) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) lds[pad16(tf * 16u + k)] = v[0][k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) dst[(u32)k * C::TH + tf] = lds[pad16((u32)k * C::TH + tf)];
    };
    u32 va[1][16], vb[1][16];
    if (a.flags & 1u) load_evals(va, a.a + row * C::M);
    else { sq_big_load<LP>(va, a.a + row * C::M, tf, q, a.bq); fwd_big<LP, 0, 1, true, LOOSE>(va, lds, ltw, a.tw_fwd, tf, q, q2, a.bq); }
    if (a.flags & 2u) load_evals(vb, a.b + row * C::M);
    else { sq_big_load<LP>(vb, a.b + row * C::M, tf, q, a.bq); fwd_big<LP, 0, 1, true, LOOSE>(vb, lds, ltw, a.tw_fwd, tf, q, q2, a.bq); }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        va[0][k] = csub_u32(barrett2p_32(va[0][k], q, a.bq), q);
        vb[0][k] = csub_u32(barrett2p_32(vb[0][k], q, a.bq), q);
    }
    if (a.a_evals) store_evals(a.a_evals + row * C::M, va);
    if (a.b_evals) store_evals(a.b_evals + row * C::M, vb);
    Tw32 scale = a.ninv_mont;
    if (a.c_evals) {
#pragma unroll
        for (int k = 0; k < 16; k++) va[0][k] = reduce64_32((u64)va[0][k] * vb[0][k], q, a.mu);
        store_evals(a.c_evals + row * C::M, va);
        scale = a.ninv;
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) va[0][k] = sq_mont(va[0][k], vb[0][k], q, a.qinv_neg);
    }
    inv_big<LP>(va, lds, ltw_inv, a.tw_inv, tf, q, q2);
    u64 *__restrict__ dst = a.out + row * C::M;
#pragma unroll
    for (int k = 0; k < 16; k++) dst[(u32)k * C::TH + tf] = csub_u32(mul_shoup32(va[0][k], scale, q), q);
}

// ---- two-pass sizes 2^15 <= n <= 2^18 ----------------------------------------------------------------------------------
// The first LA = log2(n) - 12 stages pair rows of the 2^LA x 4096 view (uniform twiddles per row pair: an element-wise
// pass of 2^LA registers per thread, lanes along the columns — bound by its traffic); the remaining 12 stages are 2^LA
// independent 4096-point blocks (256 threads each; block `blk` after s0 = LA stages reads the table at
// (1 << (s0 + ls)) + (blk << ls) + H, as the contiguous pass of the 61-bit transforms does).  The intermediate between
// the passes is u32: 24 n bytes per transform instead of 32 n.  (2^14-point blocks — 1024 threads, one workgroup per CU —
// were built first: N = 2^16 forward 1.42 ms per 4096 polynomials (their block pass bound by butterflies: 0.84 ms) against
// 1.21 ms with 4096-point blocks, where both passes run at the memory system's rate.)
constexpr int kSqBlockLog = 12;
template <int LA, bool LOOSE>
__global__ __launch_bounds__(256) void sq2_strided_fwd_kernel(SmallQArgs a) {
    constexpr u32 M = 1u << kSqBlockLog, R = 1u << LA;
    const u64 row = blockIdx.x / (M / 256);
    const u32 c = (blockIdx.x % (M / 256)) * 256 + threadIdx.x;
    const u64 *__restrict__ src = a.a + (row << (kSqBlockLog + LA)) + c;
    const u32 q = a.q, q2 = 2u * q;
    u32 v[R];
#pragma unroll
    for (u32 r = 0; r < R; r++) v[r] = csub_u32(barrett2p_32((u32)src[(u64)r * M], q, a.bq), q);
#pragma unroll
    for (int s = 0; s < LA; s++) {
        const int span = (int)R >> (s + 1);
#pragma unroll
        for (int g = 0; g < (1 << s); g++) {
            const Tw32 t = a.tw_fwd[(1u << s) + g];
#pragma unroll
            for (int l = 0; l < span; l++) {
                if constexpr (LOOSE) ct32_loose(v[g * 2 * span + l], v[g * 2 * span + l + span], t, q, q2);
                else ct32(v[g * 2 * span + l], v[g * 2 * span + l + span], t, q, q2);
            }
        }
    }
    u32 *__restrict__ dst = a.mid + (row << (kSqBlockLog + LA)) + c;
#pragma unroll
    for (u32 r = 0; r < R; r++) dst[(u64)r * M] = csub_u32(barrett2p_32(v[r], q, a.bq), q);      // below (1 + 2 LA) q before
}
template <int LA>
__global__ __launch_bounds__(256) void sq2_strided_inv_kernel(SmallQArgs a) {
    constexpr u32 M = 1u << kSqBlockLog, R = 1u << LA;
    const u64 row = blockIdx.x / (M / 256);
    const u32 c = (blockIdx.x % (M / 256)) * 256 + threadIdx.x;
    const u32 *__restrict__ src = a.mid + (row << (kSqBlockLog + LA)) + c;
    const u32 q = a.q, q2 = 2u * q;
    u32 v[R];
#pragma unroll
    for (u32 r = 0; r < R; r++) v[r] = src[(u64)r * M];          // canonical
#pragma unroll
    for (int s = LA - 1; s >= 0; s--) {
        const int span = (int)R >> (s + 1);
#pragma unroll
        for (int g = 0; g < (1 << s); g++) {
            const Tw32 t = a.tw_inv[(1u << s) + g];
#pragma unroll
            for (int l = 0; l < span; l++) gs32(v[g * 2 * span + l], v[g * 2 * span + l + span], t, q, q2);
        }
    }
    u64 *__restrict__ dst = a.out + (row << (kSqBlockLog + LA)) + c;
#pragma unroll
    for (u32 r = 0; r < R; r++) dst[(u64)r * M] = csub_u32(mul_shoup32(v[r], a.ninv, q), q);
}

// the table index of round J of block (s0, blk): local table (staged per block) or the global one
template <int LP, int J>
__device__ __forceinline__ u32 sq_block_t0(u32 s0, u32 blk, u32 H) {
    using C = ContigCfg<LP>;
    constexpr int LS = C::ls0_of(J);
    return C::in_lds(J) ? (1u << LS) + H : (1u << (s0 + LS)) + (blk << LS) + H;
}
__device__ __forceinline__ void sq_stage_block(Tw32 *ltw, const Tw32 *__restrict__ tw, int count, u32 tid, u32 s0, u32 blk) {
    for (u32 li = tid; li < (u32)count; li += 256) {
        const u32 ls = 31u - (u32)__builtin_clz(li | 1u);       // entry 0 is never used: a copy of entry 1
        const u32 l1 = li | (li == 0);
        ltw[li] = tw[(1u << (s0 + ls)) + (blk << ls) + (l1 - (1u << ls))];
    }
}
// block `blk` of row `row`: u32 intermediate (natural order) -> 12 stages -> the block's 4096 output words
template <bool LOOSE>
__global__ __launch_bounds__(256) void sq2_block_fwd_kernel(SmallQArgs a, u32 la) {
    constexpr int LP = kSqBlockLog;
    using C = ContigCfg<LP>;
    using K = SqCfg<LP>;
    static_assert(C::W == 1 && C::NR == 3, "one 4096-point block per workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *lds = reinterpret_cast<u32 *>(smem_raw);
    Tw32 *ltw = reinterpret_cast<Tw32 *>(smem_raw + K::TILE_BYTES);
    const u32 tf = threadIdx.x;
    const u32 blk = blockIdx.x & ((1u << la) - 1u);
    const u64 base = ((u64)(blockIdx.x >> la) << (LP + la)) + ((u64)blk << LP);
    sq_stage_block(ltw, a.tw_fwd, C::LTW_N, tf, la, blk);
    __syncthreads();
    const u32 q = a.q, q2 = 2u * q;
    u32 v[16];
    const u32 *__restrict__ src = a.mid + base;
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = src[(u32)k * C::TPB + tf];
    Tw32 t[15];
    load_tw32<C::R0>(t, a.tw_fwd, (1u << la) + blk);
    round_fwd32_tw<C::R0, 0, LOOSE>(v, t, q, q2);
    load_tw32<4>(t, C::in_lds(1) ? ltw : a.tw_fwd, sq_block_t0<LP, 1>(la, blk, tf >> C::a_of(1)));
    exchange32<LP, C::A0, C::a_of(1), true>(v, lds, 0u, tf);
    round_fwd32_tw<4, 0, LOOSE>(v, t, q, q2);
    load_tw32<4>(t, C::in_lds(2) ? ltw : a.tw_fwd, sq_block_t0<LP, 2>(la, blk, tf >> C::a_of(2)));
    exchange32<LP, C::a_of(1), C::a_of(2), false>(v, lds, 0u, tf);
    round_fwd32_tw<4, 0, LOOSE>(v, t, q, q2);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) lds[pad16(tf * 16u + k)] = csub_u32(barrett2p_32(v[k], q, a.bq), q);
    __syncthreads();
    u64 *__restrict__ dst = a.out + base;
#pragma unroll
    for (int k = 0; k < 16; k++) dst[(u32)k * C::TPB + tf] = lds[pad16((u32)k * C::TPB + tf)];
}
__global__ __launch_bounds__(256) void sq2_block_inv_kernel(SmallQArgs a, u32 la) {
    constexpr int LP = kSqBlockLog;
    using C = ContigCfg<LP>;
    using K = SqCfg<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *lds = reinterpret_cast<u32 *>(smem_raw);
    Tw32 *ltw = reinterpret_cast<Tw32 *>(smem_raw + K::TILE_BYTES);
    const u32 tf = threadIdx.x;
    const u32 blk = blockIdx.x & ((1u << la) - 1u);
    const u64 base = ((u64)(blockIdx.x >> la) << (LP + la)) + ((u64)blk << LP);
    sq_stage_block(ltw, a.tw_inv, C::LTW_N, tf, la, blk);
    const u32 q = a.q, q2 = 2u * q;
    const u64 *__restrict__ src = a.a + base;
#pragma unroll
    for (int k = 0; k < 16; k++) lds[pad16((u32)k * C::TPB + tf)] = csub_u32(barrett2p_32((u32)src[(u32)k * C::TPB + tf], q, a.bq), q);
    __syncthreads();
    u32 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = lds[pad16(tf * 16u + k)];
    Tw32 t[15];
    load_tw32<4>(t, C::in_lds(2) ? ltw : a.tw_inv, sq_block_t0<LP, 2>(la, blk, tf >> C::a_of(2)));
    round_inv32_tw<4>(v, t, q, q2);
    load_tw32<4>(t, C::in_lds(1) ? ltw : a.tw_inv, sq_block_t0<LP, 1>(la, blk, tf >> C::a_of(1)));
    exchange32<LP, C::a_of(2), C::a_of(1), false>(v, lds, 0u, tf);
    round_inv32_tw<4>(v, t, q, q2);
    load_tw32<C::R0>(t, a.tw_inv, (1u << la) + blk);
    exchange32<LP, C::a_of(1), C::A0, false>(v, lds, 0u, tf);
    round_inv32_tw<C::R0>(v, t, q, q2);
    u32 *__restrict__ dst = a.mid + base;
#pragma unroll
    for (int k = 0; k < 16; k++) dst[(u32)k * C::TPB + tf] = csub_u32(v[k], q);       // below 2q after the Gentleman-Sande rounds
}

// the MIDDLE of Rq x Rq at the two-pass sizes: block-forward(a), block-forward(b) in lockstep, pointwise Montgomery
// product, block-inverse — on one 4096-point block; reads the two u32 intermediates of the strided forward passes, leaves
// the u32 intermediate the strided inverse pass consumes (in place of a's)
template <bool LOOSE>
__global__ __launch_bounds__(256) void sq2_block_mul_kernel(SmallQArgs a, u32 la) {
    constexpr int LP = kSqBlockLog;
    using C = ContigCfg<LP>;
    using K = SqCfg<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *la_tile = reinterpret_cast<u32 *>(smem_raw), *lb_tile = reinterpret_cast<u32 *>(smem_raw + K::TILE_BYTES);
    Tw32 *ltw = reinterpret_cast<Tw32 *>(smem_raw + 2 * K::TILE_BYTES), *ltw_inv = ltw + C::LTW_N;
    const u32 tf = threadIdx.x;
    const u32 blk = blockIdx.x & ((1u << la) - 1u);
    const u64 base = ((u64)(blockIdx.x >> la) << (LP + la)) + ((u64)blk << LP);
    sq_stage_block(ltw, a.tw_fwd, C::LTW_N, tf, la, blk);
    sq_stage_block(ltw_inv, a.tw_inv, C::LTW_N, tf, la, blk);
    __syncthreads();
    const u32 q = a.q, q2 = 2u * q;
    // cached evals (ring_nq.rs:586-607): a block of an operand's / the product's evals IS the block's 4096 NTT-domain words
    auto load_evals = [&](u32 (&v)[16], const u64 *__restrict__ src, u32 *tile) {
#pragma unroll
        for (int k = 0; k < 16; k++) tile[pad16((u32)k * C::TPB + tf)] = csub_u32(barrett2p_32((u32)src[base + (u32)k * C::TPB + tf], q, a.bq), q);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = tile[pad16(tf * 16u + k)];
    };
    auto store_evals = [&](u64 *__restrict__ dst, const u32 (&v)[16], u32 *tile) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) tile[pad16(tf * 16u + k)] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) dst[base + (u32)k * C::TPB + tf] = tile[pad16((u32)k * C::TPB + tf)];
    };
    // one operand's block: u32 intermediate of its strided pass -> the 12 block stages, alone on its tile
    auto forward1 = [&](u32 (&v)[16], const u32 *__restrict__ mid, u32 *tile) {
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = mid[base + (u32)k * C::TPB + tf];
        Tw32 t[15];
        load_tw32<C::R0>(t, a.tw_fwd, (1u << la) + blk);
        round_fwd32_tw<C::R0, 0, LOOSE>(v, t, q, q2);
        load_tw32<4>(t, C::in_lds(1) ? ltw : a.tw_fwd, sq_block_t0<LP, 1>(la, blk, tf >> C::a_of(1)));
        exchange32<LP, C::A0, C::a_of(1), true>(v, tile, 0u, tf);
        round_fwd32_tw<4, 0, LOOSE>(v, t, q, q2);
        load_tw32<4>(t, C::in_lds(2) ? ltw : a.tw_fwd, sq_block_t0<LP, 2>(la, blk, tf >> C::a_of(2)));
        exchange32<LP, C::a_of(1), C::a_of(2), false>(v, tile, 0u, tf);
        round_fwd32_tw<4, 0, LOOSE>(v, t, q, q2);
    };
    u32 va[16], vb[16];
    Tw32 t[15];
    if (a.flags == 0u) {                                        // both from their strided passes, in lockstep
#pragma unroll
        for (int k = 0; k < 16; k++) {
            va[k] = a.mid[base + (u32)k * C::TPB + tf];
            vb[k] = a.mid_b[base + (u32)k * C::TPB + tf];
        }
        load_tw32<C::R0>(t, a.tw_fwd, (1u << la) + blk);
        round_fwd32_tw<C::R0, 0, LOOSE>(va, t, q, q2);
        round_fwd32_tw<C::R0, 0, LOOSE>(vb, t, q, q2);
        load_tw32<4>(t, C::in_lds(1) ? ltw : a.tw_fwd, sq_block_t0<LP, 1>(la, blk, tf >> C::a_of(1)));
        sq_exchange2<LP, C::A0, C::a_of(1), true>(va, vb, la_tile, lb_tile, 0u, tf);
        round_fwd32_tw<4, 0, LOOSE>(va, t, q, q2);
        round_fwd32_tw<4, 0, LOOSE>(vb, t, q, q2);
        load_tw32<4>(t, C::in_lds(2) ? ltw : a.tw_fwd, sq_block_t0<LP, 2>(la, blk, tf >> C::a_of(2)));
        sq_exchange2<LP, C::a_of(1), C::a_of(2), false>(va, vb, la_tile, lb_tile, 0u, tf);
        round_fwd32_tw<4, 0, LOOSE>(va, t, q, q2);
        round_fwd32_tw<4, 0, LOOSE>(vb, t, q, q2);
    } else {
        if (a.flags & 1u) load_evals(va, a.a, la_tile); else forward1(va, a.mid, la_tile);
        if (a.flags & 2u) load_evals(vb, a.b, lb_tile); else forward1(vb, a.mid_b, lb_tile);
    }
    const bool keep = a.c_evals || a.a_evals || a.b_evals;
    if (keep) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            va[k] = csub_u32(barrett2p_32(va[k], q, a.bq), q);
            vb[k] = csub_u32(barrett2p_32(vb[k], q, a.bq), q);
        }
        if (a.a_evals) store_evals(a.a_evals, va, la_tile);
        if (a.b_evals) store_evals(a.b_evals, vb, lb_tile);
    }
    if (a.c_evals) {                                            // canonical product: the strided inverse then scales by n^-1 alone
#pragma unroll
        for (int k = 0; k < 16; k++) va[k] = reduce64_32((u64)va[k] * vb[k], q, a.mu);
        store_evals(a.c_evals, va, lb_tile);
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) va[k] = sq_mont(barrett2p_32(va[k], q, a.bq), barrett2p_32(vb[k], q, a.bq), q, a.qinv_neg);
    }
    load_tw32<4>(t, C::in_lds(2) ? ltw_inv : a.tw_inv, sq_block_t0<LP, 2>(la, blk, tf >> C::a_of(2)));
    round_inv32_tw<4>(va, t, q, q2);
    load_tw32<4>(t, C::in_lds(1) ? ltw_inv : a.tw_inv, sq_block_t0<LP, 1>(la, blk, tf >> C::a_of(1)));
    exchange32<LP, C::a_of(2), C::a_of(1), false>(va, la_tile, 0u, tf);
    round_inv32_tw<4>(va, t, q, q2);
    load_tw32<C::R0>(t, a.tw_inv, (1u << la) + blk);
    exchange32<LP, C::a_of(1), C::A0, false>(va, la_tile, 0u, tf);
    round_inv32_tw<C::R0>(va, t, q, q2);
#pragma unroll
    for (int k = 0; k < 16; k++) a.mid[base + (u32)k * C::TPB + tf] = csub_u32(va[k], q);
}

// ---- host side ----------------------------------------------------------------------------------------------------------
// q < 2^30: 4 q fits a word (Harvey's butterflies); q < 2^32 / 25: twelve stages without any conditional subtraction
bool smallq_supported(uint64_t q, unsigned log_n) {
    return q >= 3 && (q & 1) && q < (1ull << 30) && log_n >= 8 && log_n <= 18;
}
bool smallq_loose(uint64_t q) { return q * 25 < (1ull << 32); }
size_t smallq_scratch_bytes(unsigned log_n, uint64_t rows) { return log_n > 14 ? (rows << log_n) * 4 : 0; }      // per operand

template <typename K>
static hipError_t sq_launch(K kernel, const char *name, int lp, size_t lds, unsigned units, const SmallQArgs &a, hipStream_t st) {
    if (a.rows == 0) return hipSuccess;
    const u64 grid = (a.rows + units - 1) / units;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    if (hipError_t e = allow_big_lds((const void *)kernel, lds)) return e;
    KernelTimer kt(name, lp, st);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(256), lds, st, a);
    return hipGetLastError();
}
// KERNEL<LP>: an alias template with the butterfly form already chosen (the _l / _h variables below)
#define FHE_SQ_SWITCH(KERNEL, NAME, TILES, TWS)                                                                                         \
    switch (log_n) {                                                                                                                    \
        case 8: return sq_launch(KERNEL<8>, NAME, 8, TILES * SqCfg<8>::TILE_BYTES + TWS * SqCfg<8>::TW_BYTES, ContigCfg<8>::W, a, st);      \
        case 9: return sq_launch(KERNEL<9>, NAME, 9, TILES * SqCfg<9>::TILE_BYTES + TWS * SqCfg<9>::TW_BYTES, ContigCfg<9>::W, a, st);      \
        case 10: return sq_launch(KERNEL<10>, NAME, 10, TILES * SqCfg<10>::TILE_BYTES + TWS * SqCfg<10>::TW_BYTES, ContigCfg<10>::W, a, st); \
        case 11: return sq_launch(KERNEL<11>, NAME, 11, TILES * SqCfg<11>::TILE_BYTES + TWS * SqCfg<11>::TW_BYTES, ContigCfg<11>::W, a, st); \
        case 12: return sq_launch(KERNEL<12>, NAME, 12, TILES * SqCfg<12>::TILE_BYTES + TWS * SqCfg<12>::TW_BYTES, ContigCfg<12>::W, a, st); \
    }                                                                                                                                   \
    return hipErrorNotSupported;

template <typename K>
static hipError_t sq_big_launch(K kernel, const char *name, int lp, size_t lds, unsigned th, const SmallQArgs &a, hipStream_t st) {
    if (a.rows == 0) return hipSuccess;
    if (a.rows > 0x7fffffffull) return hipErrorInvalidValue;
    if (hipError_t e = allow_big_lds((const void *)kernel, lds)) return e;
    KernelTimer kt(name, lp, st);
    hipLaunchKernelGGL(kernel, dim3((unsigned)a.rows), dim3(th), lds, st, a);
    return hipGetLastError();
}
#define FHE_SQ_BIG(KERNEL, NAME, TWS)                                                                                             \
    if (log_n == 13) return sq_big_launch(KERNEL<13>, NAME, 13, Big32<13>::TILE_BYTES + TWS * Big32<13>::TW_BYTES, Big32<13>::TH, a, st); \
    if (log_n == 14) return sq_big_launch(KERNEL<14>, NAME, 14, Big32<14>::TILE_BYTES + TWS * Big32<14>::TW_BYTES, Big32<14>::TH, a, st);

template <typename KS, typename KB>
static hipError_t sq2_launch(KS strided, KB block, bool forward, int log_n, const SmallQArgs &a, hipStream_t st) {
    const unsigned la = (unsigned)(log_n - kSqBlockLog);
    if (a.rows == 0) return hipSuccess;
    if (!a.mid) return hipErrorInvalidValue;
    const u64 gs = a.rows * ((1u << kSqBlockLog) / 256), gb = a.rows << la;
    if (gs > 0x7fffffffull || gb > 0x7fffffffull) return hipErrorInvalidValue;
    constexpr size_t lds = SqCfg<kSqBlockLog>::TILE_BYTES + SqCfg<kSqBlockLog>::TW_BYTES;
    auto run_strided = [&]() {
        KernelTimer kt(forward ? "sq2_strided_fwd" : "sq2_strided_inv", log_n, st);
        hipLaunchKernelGGL(strided, dim3((unsigned)gs), dim3(256), 0, st, a);
        return hipGetLastError();
    };
    auto run_block = [&]() {
        KernelTimer kt(forward ? "sq2_block_fwd" : "sq2_block_inv", log_n, st);
        hipLaunchKernelGGL(block, dim3((unsigned)gb), dim3(256), lds, st, a, la);
        return hipGetLastError();
    };
    if (forward) { if (hipError_t e = run_strided()) return e; return run_block(); }
    if (hipError_t e = run_block()) return e;
    return run_strided();
}
template <int LP> static constexpr auto sq_forward_l = sq_forward_kernel<LP, true>;
template <int LP> static constexpr auto sq_forward_h = sq_forward_kernel<LP, false>;
template <int LP> static constexpr auto sq_big_forward_l = sq_big_forward_kernel<LP, true>;
template <int LP> static constexpr auto sq_big_forward_h = sq_big_forward_kernel<LP, false>;
template <int LP> static constexpr auto sq_rq_mul_l = sq_rq_mul_kernel<LP, true>;
template <int LP> static constexpr auto sq_rq_mul_h = sq_rq_mul_kernel<LP, false>;
template <int LP> static constexpr auto sq_big_rq_mul_l = sq_big_rq_mul_kernel<LP, true>;
template <int LP> static constexpr auto sq_big_rq_mul_h = sq_big_rq_mul_kernel<LP, false>;
template <bool LOOSE>
static hipError_t launch_sq_forward_t(const SmallQArgs &a, int log_n, hipStream_t st) {
    if (log_n == 15) return sq2_launch(sq2_strided_fwd_kernel<3, LOOSE>, sq2_block_fwd_kernel<LOOSE>, true, log_n, a, st);
    if (log_n == 16) return sq2_launch(sq2_strided_fwd_kernel<4, LOOSE>, sq2_block_fwd_kernel<LOOSE>, true, log_n, a, st);
    if (log_n == 17) return sq2_launch(sq2_strided_fwd_kernel<5, LOOSE>, sq2_block_fwd_kernel<LOOSE>, true, log_n, a, st);
    if (log_n == 18) return sq2_launch(sq2_strided_fwd_kernel<6, LOOSE>, sq2_block_fwd_kernel<LOOSE>, true, log_n, a, st);
    if constexpr (LOOSE) {
        FHE_SQ_BIG(sq_big_forward_l, "sq_forward", 1)
        FHE_SQ_SWITCH(sq_forward_l, "sq_forward", 1, 1)
    } else {
        FHE_SQ_BIG(sq_big_forward_h, "sq_forward", 1)
        FHE_SQ_SWITCH(sq_forward_h, "sq_forward", 1, 1)
    }
}
hipError_t launch_sq_forward(const SmallQArgs &a, int log_n, hipStream_t st) {
    return a.loose ? launch_sq_forward_t<true>(a, log_n, st) : launch_sq_forward_t<false>(a, log_n, st);
}
hipError_t launch_sq_inverse(const SmallQArgs &a, int log_n, hipStream_t st) {
    if (log_n == 15) return sq2_launch(sq2_strided_inv_kernel<3>, sq2_block_inv_kernel, false, log_n, a, st);
    if (log_n == 16) return sq2_launch(sq2_strided_inv_kernel<4>, sq2_block_inv_kernel, false, log_n, a, st);
    if (log_n == 17) return sq2_launch(sq2_strided_inv_kernel<5>, sq2_block_inv_kernel, false, log_n, a, st);
    if (log_n == 18) return sq2_launch(sq2_strided_inv_kernel<6>, sq2_block_inv_kernel, false, log_n, a, st);
    FHE_SQ_BIG(sq_big_inverse_kernel, "sq_inverse", 1)
    FHE_SQ_SWITCH(sq_inverse_kernel, "sq_inverse", 1, 1)
}
template <typename KF, typename KI, typename KM>
static hipError_t sq2_mul_launch(KF sfwd, KI sinv, KM smid, int log_n, const SmallQArgs &a, hipStream_t st) {
    const unsigned la = (unsigned)(log_n - kSqBlockLog);
    if (a.rows == 0) return hipSuccess;
    if (!a.mid || !a.mid_b) return hipErrorInvalidValue;
    const u64 gs = a.rows * ((1u << kSqBlockLog) / 256), gb = a.rows << la;
    if (gs > 0x7fffffffull || gb > 0x7fffffffull) return hipErrorInvalidValue;
    constexpr size_t lds = 2 * SqCfg<kSqBlockLog>::TILE_BYTES + 2 * SqCfg<kSqBlockLog>::TW_BYTES;
    SmallQArgs fa = a, fb = a, inv = a;
    fb.a = a.b; fb.mid = a.mid_b;                               // strided forward of b into its own intermediate
    if (!a.c_evals) inv.ninv = a.ninv_mont;                     // the Montgomery product's 2^-32 leaves with n^-1
    {
        KernelTimer kt("sq2_strided_fwd", log_n, st);           // operands given as evals have no strided pass
        if (!(a.flags & 1u)) hipLaunchKernelGGL(sfwd, dim3((unsigned)gs), dim3(256), 0, st, fa);
        if (!(a.flags & 2u)) hipLaunchKernelGGL(sfwd, dim3((unsigned)gs), dim3(256), 0, st, fb);
    }
    if (hipError_t e = hipGetLastError()) return e;
    {
        KernelTimer kt("sq2_block_mul", log_n, st);
        hipLaunchKernelGGL(smid, dim3((unsigned)gb), dim3(256), lds, st, a, la);
    }
    if (hipError_t e = hipGetLastError()) return e;
    KernelTimer kt("sq2_strided_inv", log_n, st);
    hipLaunchKernelGGL(sinv, dim3((unsigned)gs), dim3(256), 0, st, inv);
    return hipGetLastError();
}
template <bool LOOSE>
static hipError_t launch_sq_rq_mul_t(const SmallQArgs &a, int log_n, hipStream_t st) {
    if (log_n == 15) return sq2_mul_launch(sq2_strided_fwd_kernel<3, LOOSE>, sq2_strided_inv_kernel<3>, sq2_block_mul_kernel<LOOSE>, log_n, a, st);
    if (log_n == 16) return sq2_mul_launch(sq2_strided_fwd_kernel<4, LOOSE>, sq2_strided_inv_kernel<4>, sq2_block_mul_kernel<LOOSE>, log_n, a, st);
    if (log_n == 17) return sq2_mul_launch(sq2_strided_fwd_kernel<5, LOOSE>, sq2_strided_inv_kernel<5>, sq2_block_mul_kernel<LOOSE>, log_n, a, st);
    if (log_n == 18) return sq2_mul_launch(sq2_strided_fwd_kernel<6, LOOSE>, sq2_strided_inv_kernel<6>, sq2_block_mul_kernel<LOOSE>, log_n, a, st);
    if constexpr (LOOSE) {
        FHE_SQ_BIG(sq_big_rq_mul_l, "sq_rq_mul", 2)
        FHE_SQ_SWITCH(sq_rq_mul_l, "sq_rq_mul", 2, 2)
    } else {
        FHE_SQ_BIG(sq_big_rq_mul_h, "sq_rq_mul", 2)
        FHE_SQ_SWITCH(sq_rq_mul_h, "sq_rq_mul", 2, 2)
    }
}
hipError_t launch_sq_rq_mul(const SmallQArgs &a, int log_n, hipStream_t st) {
    return a.loose ? launch_sq_rq_mul_t<true>(a, log_n, st) : launch_sq_rq_mul_t<false>(a, log_n, st);
}
#undef FHE_SQ_SWITCH
#undef FHE_SQ_BIG

}  // namespace fhe
