// digit32.hip — the TGGSW x TGLWE external product on TWO 27-bit primes with 32-bit arithmetic.
//
// Why a second arithmetic: every kernel of the 61-bit engine is bound by the issue rate of 32-bit multiplies (ten per
// Shoup butterfly; ~15 issue slots per 128-bit multiply-accumulate).  The external product's integers are
// small — a half-sum  S = sum_t key_half[t] * digit_t  is below (k+1) l n 2^32 <= 2^53 in magnitude — so they are also
// determined by their residues modulo two primes just below 2^32 / 25 (product 2^54.7).  There a butterfly is 3
// multiplies (v_mul_hi_u32 + 2 v_mul_lo_u32) and, because 25 p fits a word, 3 additions with NO conditional subtraction
// through all the stages of a transform; a multiply-accumulate is ONE v_mad_u64_u32 into a 64-bit accumulator that needs
// no reduction for 255 terms.  Twice the transforms, a third of the multiplies each; the accumulate phase shrinks
// from ~60 to ~6 cycles per term.
//
// What is computed is the reference's  TGGSW * TGLWE  (tfhe/src/tggsw.rs:45-62,139-149; Tn::decompose beta = 2,
// ring_torus.rs:67-77, torus.rs:43-52) with the key words split in 32-bit halves exactly as in zring.hip's one-prime
// form:  out[c] = lift(S_lo[c]) + (lift(S_hi[c]) << 32)  mod 2^64,  lift = the centred CRT lift from the two residues.
//
// Kernels (single-pass sizes 2^8 <= n <= 2^12 with k = 1 — BASELINE.json configs[3] is n = 2^10; anything else keeps
// the 61-bit path of zring.hip / digit_mac.hip):
//   ntt32_fwd_key_kernel   key preparation: halves reduced mod p, forward transform, stored as u32
//   digit_mac32_kernel     digit extraction -> forward transform modulo BOTH primes (round 0 by table look-up, as in
//                          ntt_rounds.hpp round0_bits) -> multiply-accumulate against the key rows, per (ciphertext, part)
//   digit_tail32_kernel    sum of the parts -> inverse transforms modulo both primes -> CRT lift -> lo + (hi << 32), or
//                          modulo q and (0, b) - rhs (key switching)
// Index algebra, register windows and LDS exchanges are those of ntt_rounds.hpp (ContigCfg, field_of, pad16).
#include "digit32.hpp"
#include "ntt32_rounds.hpp"

namespace fhe {

// the LP forward stages: window [LP-4, LP) -> window [0,4); BITS: round 0 by look-up; values end below 4p
template <int LP, bool BITS, bool FRESH, bool LOOSE = false>
__device__ __forceinline__ void fwd_rounds32(u32 (&v)[16], u32 *lds, const Tw32 *ltw, const Tw32 *gtw, const u32 *lut, u32 w, u32 tf,
                                             u32 p, u32 p2) {
    using C = ContigCfg<LP>;
    auto TW = [&](bool in_lds) -> const Tw32 * { return in_lds ? ltw : gtw; };
    if constexpr (BITS) round0_bits32<C::R0, LOOSE>(v, lut, gtw, p, p2);
    else round_fwd32<C::R0, 0, LOOSE>(v, gtw, 1u, p, p2);
    if constexpr (C::NR > 1) {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        exchange32<LP, C::A0, A, FRESH>(v, lds, w, tf);
        round_fwd32<4, 0, LOOSE>(v, TW(C::in_lds(1)), (1u << LS) + (tf >> A), p, p2);
    }
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        if constexpr (!C::in_lds(2)) {
            // twiddles from the global table (unique per thread): requested BEFORE the exchange, so that their latency
            // runs under its barriers (ntt32_rounds.hpp: load_tw32) — with one workgroup per CU at n = 4096 nothing else
            // hides it: 256 key switches 351 -> 316 us
            Tw32 t[15];
            load_tw32<4>(t, gtw, (1u << LS) + (tf >> A));
            exchange32<LP, C::a_of(1), A, false>(v, lds, w, tf);
            round_fwd32_tw<4, 0, LOOSE>(v, t, p, p2);
        } else {
            exchange32<LP, C::a_of(1), A, false>(v, lds, w, tf);
            round_fwd32<4, 0, LOOSE>(v, ltw, (1u << LS) + (tf >> A), p, p2);
        }
    }
    static_assert(C::NR <= 3, "n <= 4096");
}
// The same for BOTH primes in lockstep (va modulo p[0] through tile / table 0, vb modulo p[1] through 1): one pair of
// barriers per exchange instead of two and twice the independent work between barriers; butterflies WITHOUT conditional
// subtractions (ct32_loose) — the values end below (1 + 2 LP) p <= 25 p < 2^32.
template <int LP, int AF, int AT, bool FIRST>
__device__ __forceinline__ void exchange32x2(u32 (&va)[16], u32 (&vb)[16], u32 *la, u32 *lb, u32 w, u32 tf) {
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
template <int LP, bool FRESH>
__device__ __forceinline__ void fwd_rounds32x2_bits(u32 (&va)[16], u32 (&vb)[16], u32 *const (&tile)[2], const Tw32 *const (&ltw)[2],
                                                    const Tw32 *const (&gtw)[2], const u32 *const (&lut)[2], u32 w, u32 tf,
                                                    const u32 (&p)[2]) {
    using C = ContigCfg<LP>;
    const u32 pa = p[0], pa2 = 2u * pa, pb = p[1], pb2 = 2u * pb;
    round0_bits32<C::R0, true>(va, lut[0], gtw[0], pa, pa2);
    round0_bits32<C::R0, true>(vb, lut[1], gtw[1], pb, pb2);
    if constexpr (C::NR > 1) {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        constexpr bool L = C::in_lds(1);
        exchange32x2<LP, C::A0, A, FRESH>(va, vb, tile[0], tile[1], w, tf);
        round_fwd32<4, 0, true>(va, L ? ltw[0] : gtw[0], (1u << LS) + (tf >> A), pa, pa2);
        round_fwd32<4, 0, true>(vb, L ? ltw[1] : gtw[1], (1u << LS) + (tf >> A), pb, pb2);
    }
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        constexpr bool L = C::in_lds(2);
        // (requesting the global twiddles of both primes ahead of this exchange, as fwd_rounds32 does, costs 35 registers
        // here and was measured slower: 329 vs 320 us per 630 products — two workgroups per CU hide that latency already)
        exchange32x2<LP, C::a_of(1), A, false>(va, vb, tile[0], tile[1], w, tf);
        round_fwd32<4, 0, true>(va, L ? ltw[0] : gtw[0], (1u << LS) + (tf >> A), pa, pa2);
        round_fwd32<4, 0, true>(vb, L ? ltw[1] : gtw[1], (1u << LS) + (tf >> A), pb, pb2);
    }
}

// the LP inverse stages: window [0,4) (canonical inputs) -> window [LP-4, LP), values below 2p, NOT yet scaled by n^-1
template <int LP, bool FRESH>
__device__ __forceinline__ void inv_rounds32(u32 (&v)[16], u32 *lds, const Tw32 *ltw, const Tw32 *gtw, u32 w, u32 tf, u32 p, u32 p2) {
    using C = ContigCfg<LP>;
    auto TW = [&](bool in_lds) -> const Tw32 * { return in_lds ? ltw : gtw; };
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        round_inv32<4>(v, TW(C::in_lds(2)), (1u << LS) + (tf >> A), p, p2);
        exchange32<LP, A, C::a_of(1), FRESH>(v, lds, w, tf);
    }
    if constexpr (C::NR > 1) {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        round_inv32<4>(v, TW(C::in_lds(1)), (1u << LS) + (tf >> A), p, p2);
        exchange32<LP, A, C::A0, FRESH && (C::NR <= 2)>(v, lds, w, tf);
    }
    round_inv32<C::R0>(v, TW(C::in_lds(0)), 1u, p, p2);
}

// Shapes of the transform kernels (key preparation, tails): ContigCfg's — 256 threads, W = 4096 / n units side by side.
template <int LP>
struct Cfg32 {
    using C = ContigCfg<LP>;
    static constexpr int M = C::M, TPB = C::TPB, W = C::W, TH = 256, PPT = M / TH;
    static constexpr size_t TILE_BYTES = (size_t)(W * M + W * M / 16) * 4;          // one padded tile of u32
    static constexpr size_t TW_BYTES = (size_t)C::LTW_N * sizeof(Tw32);
    static constexpr size_t LUT_BYTES = (size_t)kLut32Words * 4;
    static_assert(LP >= 8 && LP <= 12, "n = 256 .. 4096");
};
// Shape of the fused digit kernel: TH threads transform W = 16 TH / n digits side by side, then multiply at
// PPT = n / TH positions per thread into 2 (primes) * NC * PPT 64-bit accumulators.  With NC = 4 that is 32 accumulators
// up to n = 1024 — room for both primes' coefficients, so their transforms run in lockstep (LOCK) — and 64 beyond, where
// the primes take turns on 16 coefficients; n = 4096 takes 512 threads (two digits side by side, one workgroup per CU).
// Measured at n = 4096 (256 key switches, l = 61): 512 threads 354 us; 256 threads with ONE prime per workgroup (twice
// the workgroups, 64 accumulators each, two per CU) 384 us; a software pipeline that multiplies step s-1 while step s
// transforms (double-buffered tiles, key loads issued a half-round ahead) spills and is 2-3x slower at every size.
// At n = 1024 (630 external products, lockstep 320 us): the primes taking turns in 168 registers, three workgroups per
// CU, 340 us; the multiply phase made unconditional so that a step's key loads issue together, 322 us (351 -> 371 at
// n = 4096).
// Shape experiments at n = 1024 (tools: -DFHE_D32_TH10=512 -DFHE_D32_WAVES10=4 -DFHE_D32_MAC_UNROLL=1 ...): defaults = production
#ifndef FHE_D32_TH10
#define FHE_D32_TH10 256          // threads per workgroup at n = 1024
#endif
#ifndef FHE_D32_WAVES10
#define FHE_D32_WAVES10 2         // waves per SIMD the register allocation must allow at n = 1024
#endif
#ifndef FHE_D32_MAC_UNROLL
#define FHE_D32_MAC_UNROLL 0      // digits of a step whose key loads the multiply phase has in flight at once (0 = all W)
#endif
#ifndef FHE_D32_ONEP10
#define FHE_D32_ONEP10 0          // 1: at n = 1024 a workgroup serves ONE prime (gridDim.y = 2): half the accumulators and tiles
#endif
#define FHE_D32_PRAGMA_(x) _Pragma(#x)
#define FHE_D32_PRAGMA(x) FHE_D32_PRAGMA_(x)
template <int LP>
struct Mac32Cfg {
    using C = ContigCfg<LP>;
    static constexpr int M = C::M, TPB = C::TPB;
    static constexpr int TH = LP == 12 ? 512 : LP == 10 ? FHE_D32_TH10 : 256;
    static constexpr int WAVES = LP == 10 ? FHE_D32_WAVES10 : 512 / TH;     // __launch_bounds__: minimum waves per SIMD
    static constexpr int W = TH / TPB, PPT = M / TH;
    static constexpr bool ONEP = LP == 10 && FHE_D32_ONEP10 != 0;
    static constexpr int NP = ONEP ? 1 : 2;                                  // primes per workgroup
    static constexpr bool LOCK = PPT <= 4 && !ONEP;
    static constexpr size_t TILE_BYTES = (size_t)(W * M + W * M / 16) * 4;
    static constexpr size_t TW_BYTES = Cfg32<LP>::TW_BYTES, LUT_BYTES = Cfg32<LP>::LUT_BYTES;
    static constexpr size_t LDS_BYTES = NP * (TILE_BYTES + TW_BYTES + LUT_BYTES);   // per prime: tile, twiddle tile, look-up table
};

// ---- key preparation: [T][k1][n] u64 words  ->  [prime][T][half][k1][n] u32, NTT domain (rows = T * 2 * k1) -------------------------
template <int LP>
__global__ __launch_bounds__(256) void ntt32_fwd_key_kernel(Ext32Args a) {
    using C = ContigCfg<LP>;
    using K = Cfg32<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *lds = reinterpret_cast<u32 *>(smem_raw);
    Tw32 *ltw = reinterpret_cast<Tw32 *>(smem_raw + K::TILE_BYTES);
    const u32 tid = threadIdx.x, w = tid / C::TPB, tf = tid % C::TPB;
    const u32 prime = blockIdx.y;
    const u32 p = a.p[prime], p2 = 2u * p;
    const Tw32 *gtw = a.tw_fwd[prime];
    stage_tw32(ltw, gtw, C::LTW_N, tid);
    const u64 row = (u64)blockIdx.x * C::W + w;
    const bool live = row < a.rows;
    // output row (t, half, c) is the 32-bit half `half` of source row (t, c) of the key as the reference holds it
    const u64 orow = live ? row : 0, t = orow / (2 * a.key_k1), rem = orow - t * (2 * a.key_k1);
    const u32 half = (u32)(rem / a.key_k1), c = (u32)(rem - (u64)half * a.key_k1);
    const u64 *__restrict__ src = a.key64 + (t * a.key_k1 + c) * C::M;
    u32 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u64 x = src[field_of<C::A0>(tf, k)];
        v[k] = csub_u32(barrett2p_32((u32)(x >> (32u * half)), p, a.bq[prime]), p);
    }
    fwd_rounds32<LP, false, true>(v, lds, ltw, gtw, nullptr, w, tf, p, p2);
    if (live) {
        u32 *__restrict__ dst = a.key32 + ((u64)prime * a.rows + row) * C::M + tf * 16u;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint4 o;
            o.x = canon4_32(v[4 * j], p, p2); o.y = canon4_32(v[4 * j + 1], p, p2);
            o.z = canon4_32(v[4 * j + 2], p, p2); o.w = canon4_32(v[4 * j + 3], p, p2);
            *reinterpret_cast<uint4 *>(dst + 4 * j) = o;
        }
    }
}

// ---- digits -> transforms modulo both primes -> multiply-accumulate ----------------------------------------------
// digit d (0 = most significant) of a source word: SRC_DIGITS  Tn::decompose beta = 2 (torus.rs:43-52);
// SRC_ZQBITS  Zq::decompose_base2 (zq.rs:176-190): every digit is 1 when the value is >= 2^l, with the reference's
// `1 << l` taken modulo 64 as a --release build does (the same rule as digit_mac.hip digit_of).
template <int SRC>
__device__ __forceinline__ u32 digit32_of(u64 x, u32 l, u32 d) {
    const u32 bit = (u32)(x >> (l - 1u - d)) & 1u;
    if (SRC == SRC_DIGITS) return bit;
    return x >= (1ull << (l & 63u)) ? 1u : bit;
}

// key32 layout: [prime][t][c][n], t = row*l + digit, c < NC (NC = 2 * output rows: half-major, then component).
// out: partial sums [b][part][prime][c][n] u32 canonical.
template <int LP, int NC, int SRC>
__global__ __launch_bounds__((Mac32Cfg<LP>::TH), (Mac32Cfg<LP>::WAVES)) void digit_mac32_kernel(Ext32Args a) {
    using C = ContigCfg<LP>;
    using K = Mac32Cfg<LP>;
    constexpr int PPT = K::PPT, W = K::W, TH = K::TH, NP = K::NP;
    const u32 pr0 = K::ONEP ? blockIdx.y : 0u;                      // the first prime this workgroup serves
    static_assert(NP * NC * PPT <= 64, "accumulators");
    static_assert(1 + 2 * LP <= 25, "ct32_loose: the bound of the values after LP stages");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *tile[NP];
    Tw32 *ltw_w[NP];
    u32 *llut_w[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        tile[i] = reinterpret_cast<u32 *>(smem_raw + i * K::TILE_BYTES);
        ltw_w[i] = reinterpret_cast<Tw32 *>(smem_raw + NP * K::TILE_BYTES + i * K::TW_BYTES);
        llut_w[i] = reinterpret_cast<u32 *>(smem_raw + NP * (K::TILE_BYTES + K::TW_BYTES) + i * K::LUT_BYTES);
    }
    const u32 tid = threadIdx.x, w = tid / C::TPB, tf = tid % C::TPB;
    const u64 b = blockIdx.x / a.parts;
    const u32 part = blockIdx.x % a.parts;
    const u32 t_begin = part * a.tpp, t_end = min(a.T, t_begin + a.tpp);
    const u32 n = 1u << LP;
    u32 p[NP], bq[NP];
    const Tw32 *gtw[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        p[i] = a.p[pr0 + i]; bq[i] = a.bq[pr0 + i]; gtw[i] = a.tw_fwd[pr0 + i];
        stage_tw32<TH>(ltw_w[i], gtw[i], C::LTW_N, tid);
        for (u32 e = tid; e < (u32)kLut32Words; e += TH) llut_w[i][e] = a.lut[pr0 + i][e];
    }
    __syncthreads();
    const u64 *__restrict__ ct = a.src + b * a.ct_stride;

    u64 acc[NP][NC][PPT];
#pragma unroll
    for (int pr = 0; pr < NP; pr++)
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int i = 0; i < PPT; i++) acc[pr][c][i] = 0;
    u32 pending = 0;
    const u32 j0 = tid * PPT;
    for (u32 t0 = t_begin; t0 < t_end; t0 += W) {
        const u32 t = t0 + w;
        const u32 tt = t < t_end ? t : t_begin;               // idle units redo a valid digit, never multiplied
        const u32 r = tt / a.l, d = tt - r * a.l;
        const u64 *__restrict__ row = ct + (u64)r * n;
        // the tiles were read by the previous step's multiply phase: barrier first (FRESH = false); every thread then
        // rewrites exactly the slots it gathered in the last exchange, reduced below 2p (a product is below 2 p^2 < 2^55.8)
        if constexpr (K::LOCK) {
            u32 va[16], vb[16];
#pragma unroll
            for (int k = 0; k < 16; k++) va[k] = vb[k] = digit32_of<SRC>(row[field_of<C::A0>(tf, k)], a.l, d);
            const Tw32 *const lt[2] = {ltw_w[0], ltw_w[1]};
            const u32 *const ll[2] = {llut_w[0], llut_w[1]};
            u32 *const tl[2] = {tile[0], tile[1]};
#ifndef FHE_D32_ABLATE_NTT      // timing-only builds (tools/abl_build.sh): the kernel without its transforms / its multiply phase
            const Tw32 *const gt[2] = {gtw[0], gtw[1]};
            const u32 pp[2] = {p[0], p[1]};
            fwd_rounds32x2_bits<LP, false>(va, vb, tl, lt, gt, ll, w, tf, pp);
#else
            __syncthreads();
#endif
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const u32 sl = pad16(w * C::M + field_of<0>(tf, k));
                tile[0][sl] = barrett2p_32(va[k], p[0], bq[0]);
                tile[NP - 1][sl] = barrett2p_32(vb[k], p[NP - 1], bq[NP - 1]);
            }
        } else {
            u32 bits = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) bits |= digit32_of<SRC>(row[field_of<C::A0>(tf, k)], a.l, d) << k;
#pragma unroll
            for (int pr = 0; pr < NP; pr++) {
                u32 v[16];
#pragma unroll
                for (int k = 0; k < 16; k++) v[k] = (bits >> k) & 1u;
#ifndef FHE_D32_ABLATE_NTT
                fwd_rounds32<LP, true, false, true>(v, tile[pr], ltw_w[pr], gtw[pr], llut_w[pr], w, tf, p[pr], 2u * p[pr]);
#else
                __syncthreads();
#endif
#pragma unroll
                for (int k = 0; k < 16; k++) tile[pr][pad16(w * C::M + field_of<0>(tf, k))] = barrett2p_32(v[k], p[pr], bq[pr]);
            }
        }
        __syncthreads();
        const u32 nu = min((u32)W, t_end - t0);
#ifdef FHE_D32_ABLATE_MAC
        if (a.T == 0xffffffffu)
#endif
#if FHE_D32_MAC_UNROLL == 0
#pragma unroll
#else
        FHE_D32_PRAGMA(unroll FHE_D32_MAC_UNROLL)
#endif
        for (int u = 0; u < W; u++) {
            if ((u32)u < nu) {
#pragma unroll
                for (int pr = 0; pr < NP; pr++) {
                    u32 x[PPT];
#pragma unroll
                    for (int i = 0; i < PPT; i++) x[i] = tile[pr][pad16(u * C::M + j0 + i)];
                    const u32 *__restrict__ g = a.key32 + (((u64)(pr0 + pr) * a.T + (t0 + u)) * NC) * n + j0;
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        u32 gv[PPT];
                        if constexpr (PPT >= 4) {
#pragma unroll
                            for (int i = 0; i < PPT; i += 4) {
                                const uint4 q4 = *reinterpret_cast<const uint4 *>(g + (u64)c * n + i);
                                gv[i] = q4.x; gv[i + 1] = q4.y; gv[i + 2] = q4.z; gv[i + 3] = q4.w;
                            }
                        } else if constexpr (PPT == 2) {
                            const uint2 q2 = *reinterpret_cast<const uint2 *>(g + (u64)c * n);
                            gv[0] = q2.x; gv[1] = q2.y;
                        } else {
                            gv[0] = g[(u64)c * n];
                        }
#pragma unroll
                        for (int i = 0; i < PPT; i++) acc[pr][c][i] += (u64)gv[i] * x[i];       // < 2^55.8 per term
                    }
                }
            }
        }
        pending += nu;
        if (pending + W > 255u) {                                 // 2^8.2 terms of 2 p^2 reach 2^64: reduce first (T > 255 only)
#pragma unroll
            for (int pr = 0; pr < NP; pr++)
#pragma unroll
                for (int c = 0; c < NC; c++)
#pragma unroll
                    for (int i = 0; i < PPT; i++) acc[pr][c][i] = reduce64_32(acc[pr][c][i], p[pr], a.mu[pr0 + pr]);
            pending = 1;
        }
    }
#pragma unroll
    for (int pr = 0; pr < NP; pr++) {
        u32 *__restrict__ o = a.part32 + ((((b * a.parts + part) * 2 + pr0 + pr) * NC) * (u64)n) + j0;
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int i = 0; i < PPT; i++) o[(u64)c * n + i] = reduce64_32(acc[pr][c][i], p[pr], a.mu[pr0 + pr]);
    }
}

// ---- the tail: sum of parts -> inverse transforms -> CRT lift -> recombination of the halves ----------------------------
// Unit w = output row (ciphertext b, component c < k+1).  Its two half-sums (key words split at bit 32) are each two
// inverse transforms; a thread ends every one of them on the SAME 16 positions, so the lifts meet in registers:
//   EPI32_TORUS  out[b][c] = lift(S_lo) + (lift(S_hi) << 32)  mod 2^64          (TGGSW x TGLWE)
//   EPI32_KS     rhs = (lift(S_lo) + lift(S_hi) * 2^32) mod q,   out[b][c] = (c < k ? 0 : glwe[b][c]) - rhs   (glwe.rs:129-136)
// lift = the centred representative modulo pA pB.  (512 threads with the two primes side by side — two transforms in
// sequence instead of four — was measured: 46 -> 41 us for a single product, but 31 -> 36 us per 630 and 37 -> 43 us per
// 256 key switches; the sequential form stays.)
enum : int { EPI32_TORUS = 0, EPI32_KS = 1 };
template <int LP, int EPI>
__global__ __launch_bounds__(256) void digit_tail32_kernel(Ext32Args a) {
    using C = ContigCfg<LP>;
    using K = Cfg32<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *tile = reinterpret_cast<u32 *>(smem_raw);
    Tw32 *ltw[2] = {reinterpret_cast<Tw32 *>(smem_raw + K::TILE_BYTES), reinterpret_cast<Tw32 *>(smem_raw + K::TILE_BYTES + K::TW_BYTES)};
    const u32 tid = threadIdx.x, w = tid / C::TPB, tf = tid % C::TPB;
    const u32 n = 1u << LP;
    const u32 k1 = a.k + 1u, NC = 2u * k1;
    const u64 rows = a.batch * k1;
    const u64 R0 = (u64)blockIdx.x * C::W;
    const u32 live = (u32)min((u64)C::W, rows - R0);
    const bool active = w < live;
    const u64 R = R0 + (active ? w : 0u);                       // idle units redo the first row and store nothing
    const u64 b = R / k1;
    const u32 c = (u32)(R - b * k1);
#pragma unroll
    for (int pr = 0; pr < 2; pr++) stage_tw32(ltw[pr], a.tw_inv[pr], C::LTW_N, tid);
    __syncthreads();
    const u32 pA = a.p[0], pB = a.p[1];
    u64 S[2][16];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        u32 res[2][16];
#pragma unroll
        for (int pr = 0; pr < 2; pr++) {
            const u32 p = a.p[pr], p2 = 2u * p;
            u32 v[16];
            const u32 *__restrict__ src = a.part32 + (((b * a.parts) * 2 + pr) * NC + h * k1 + c) * (u64)n + tf * 16u;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint4 x = *reinterpret_cast<const uint4 *>(src + 4 * j);
                v[4 * j] = x.x; v[4 * j + 1] = x.y; v[4 * j + 2] = x.z; v[4 * j + 3] = x.w;
            }
            for (u32 q = 1; q < a.parts; q++) {
                const u32 *__restrict__ sp = src + (u64)q * 2 * NC * n;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint4 x = *reinterpret_cast<const uint4 *>(sp + 4 * j);
                    v[4 * j] = csub_u32(v[4 * j] + x.x, p); v[4 * j + 1] = csub_u32(v[4 * j + 1] + x.y, p);
                    v[4 * j + 2] = csub_u32(v[4 * j + 2] + x.z, p); v[4 * j + 3] = csub_u32(v[4 * j + 3] + x.w, p);
                }
            }
            // one tile for the four transforms: all but the first find it read by the one before (FRESH = false)
            if (h == 0 && pr == 0) inv_rounds32<LP, true>(v, tile, ltw[pr], a.tw_inv[pr], w, tf, p, p2);
            else inv_rounds32<LP, false>(v, tile, ltw[pr], a.tw_inv[pr], w, tf, p, p2);
            const Tw32 ni = a.ninv[pr];
#pragma unroll
            for (int k = 0; k < 16; k++) res[pr][k] = csub_u32(mul_shoup32(v[k], ni, p), p);      // * n^-1, canonical
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const u32 rA = res[0][k], rB = res[1][k];
            const u32 rAb = csub_u32(rA, pB);
            const u32 diff = csub_u32(rB - rAb + pB, pB);
            const u32 hh = csub_u32(mul_shoup32(diff, a.crt, pB), pB);
            S[h][k] = (u64)rA + (u64)pA * hh;                       // in [0, pA * pB); centred below
        }
    }
    if (!active) return;
    const u64 base = (b * k1 + c) * (u64)n;
    if constexpr (EPI == EPI32_TORUS) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            u64 lo = S[0][k], hi = S[1][k];
            if (lo >= a.halfP) lo -= a.P;                       // two's complement of the centred value
            if (hi >= a.halfP) hi -= a.P;
            a.out[base + field_of<C::A0>(tf, k)] = lo + (hi << 32);
        }
    } else {
        const Mod &m = a.mod;
        const bool body = c >= a.k;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const u32 pos = field_of<C::A0>(tf, k);
            u64 r[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const bool neg = S[h][k] >= a.halfP;              // the centred value is S - P
                const u64 mag = reduce_any(neg ? a.P - S[h][k] : S[h][k], m);
                r[h] = (neg && mag) ? m.q - mag : mag;
            }
            u64 y = r[0] + mul_mod_var(r[1], a.two32, m);         // < 2q
            y = canon2(y, m);
            const u64 x = body ? a.glwe[base + pos] : 0ull;
            a.out[base + pos] = x >= y ? x - y : x + m.q - y;
        }
    }
}

// ---- launchers ----------------------------------------------------------------------------------------------------
template <int LP>
static hipError_t launch_key32_lp(const Ext32Args &a, hipStream_t st) {
    using C = ContigCfg<LP>;
    using K = Cfg32<LP>;
    constexpr size_t lds = K::TILE_BYTES + K::TW_BYTES;
    const u64 grid = (a.rows + C::W - 1) / C::W;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    if (hipError_t e = allow_big_lds((const void *)ntt32_fwd_key_kernel<LP>, lds)) return e;
    KernelTimer kt("ntt32_fwd_key", LP, st);
    hipLaunchKernelGGL((ntt32_fwd_key_kernel<LP>), dim3((unsigned)grid, 2), dim3(256), lds, st, a);
    return hipGetLastError();
}
template <int LP, int SRC>
static hipError_t launch_mac32_lp(const Ext32Args &a, hipStream_t st) {
    using K = Mac32Cfg<LP>;
    const u64 grid = a.batch * a.parts;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    if (hipError_t e = allow_big_lds((const void *)digit_mac32_kernel<LP, 4, SRC>, K::LDS_BYTES)) return e;
    KernelTimer kt("digit_mac32", LP, st);
    hipLaunchKernelGGL((digit_mac32_kernel<LP, 4, SRC>), dim3((unsigned)grid, K::ONEP ? 2u : 1u), dim3(K::TH), K::LDS_BYTES, st, a);
    return hipGetLastError();
}
template <int LP, int EPI>
static hipError_t launch_tail32_lp(const Ext32Args &a, hipStream_t st) {
    using C = ContigCfg<LP>;
    using K = Cfg32<LP>;
    constexpr size_t lds = K::TILE_BYTES + 2 * K::TW_BYTES;
    const u64 grid = (a.batch * (a.k + 1) + C::W - 1) / C::W;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    if (hipError_t e = allow_big_lds((const void *)digit_tail32_kernel<LP, EPI>, lds)) return e;
    KernelTimer kt(EPI == EPI32_KS ? "digit_tail32_ks" : "digit_tail32", LP, st);
    hipLaunchKernelGGL((digit_tail32_kernel<LP, EPI>), dim3((unsigned)grid), dim3(256), lds, st, a);
    return hipGetLastError();
}

// TGGSW x TGLWE: k = 1, 2^8 <= n <= 2^12
bool ext32_shape_supported(u64 n, unsigned k, unsigned l) {
    if (k != 1 || l < 1 || l > 64) return false;
    if (n < 256 || n > 4096 || (n & (n - 1))) return false;
    return (u64)(k + 1) * l * n <= (1ull << 21);          // |half-sum| < T n 2^32 <= 2^53 < P / 2
}
// GLWE::key_switch, base 2: k = 1, 2^8 <= n <= 2^12
bool ks32_shape_supported(u64 n, unsigned k, unsigned l) {
    if (k != 1 || l < 1 || l > 64) return false;
    if (n < 256 || n > 4096 || (n & (n - 1))) return false;
    return (u64)k * l * n <= (1ull << 21);
}
uint32_t ext32_units(int log_n) { return log_n >= 8 && log_n <= 12 ? (log_n == 12 ? 2u : 4096u >> log_n) : 0u; }

#define FHE_LP_SWITCH(FN, ...)                                        \
    switch (log_n) {                                                  \
        case 8: return FN<8 __VA_ARGS__>(a, st);                      \
        case 9: return FN<9 __VA_ARGS__>(a, st);                      \
        case 10: return FN<10 __VA_ARGS__>(a, st);                    \
        case 11: return FN<11 __VA_ARGS__>(a, st);                    \
        case 12: return FN<12 __VA_ARGS__>(a, st);                    \
    }                                                                 \
    return hipErrorNotSupported;
#define FHE_COMMA ,
hipError_t launch_ext32_key(const Ext32Args &a, int log_n, hipStream_t st) { FHE_LP_SWITCH(launch_key32_lp) }
hipError_t launch_ext32_mac(const Ext32Args &a, int log_n, int src_kind, hipStream_t st) {
    if (src_kind == SRC_DIGITS) { FHE_LP_SWITCH(launch_mac32_lp, FHE_COMMA SRC_DIGITS) }
    if (src_kind == SRC_ZQBITS) { FHE_LP_SWITCH(launch_mac32_lp, FHE_COMMA SRC_ZQBITS) }
    return hipErrorNotSupported;
}
hipError_t launch_ext32_tail(const Ext32Args &a, int log_n, hipStream_t st) { FHE_LP_SWITCH(launch_tail32_lp, FHE_COMMA EPI32_TORUS) }
hipError_t launch_ext32_tail_ks(const Ext32Args &a, int log_n, hipStream_t st) { FHE_LP_SWITCH(launch_tail32_lp, FHE_COMMA EPI32_KS) }
#undef FHE_LP_SWITCH
#undef FHE_COMMA

// timing-only builds (tools/abl_build.sh) produce wrong words by design: fhe_ntt_version() says so (capi.hip)
bool digit32_ablated() {
#if defined(FHE_D32_ABLATE_NTT) || defined(FHE_D32_ABLATE_MAC)
    return true;
#else
    return false;
#endif
}

}  // namespace fhe
