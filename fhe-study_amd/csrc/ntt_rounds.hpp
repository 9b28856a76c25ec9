// ntt_rounds.hpp — the device-side building blocks shared by every transform kernel: rounds of up to
// four stages on 16 register-resident coefficients (forward Cooley-Tukey / inverse Gentleman-Sande,
// with compile-time tracking of the lazy value bounds), the LDS exchange between rounds, the
// pass-local twiddle tile and the global access helpers.  Used by ntt_kernels.hip (transforms,
// products) and digit_mac.hip (decompose -> transform -> multiply-accumulate).
// Index algebra: see the header of ntt_kernels.hip.
#pragma once
#include <type_traits>

#include "ntt_kernels.hpp"
#include "zq_device.hpp"

namespace fhe {

// ---------------------------------------------------------------------------
// one round: R stages on the 16 register-resident coefficients
// ---------------------------------------------------------------------------
// Lazy ranges of the forward rounds:
//   WIDE (q < 2^61, 8q < 2^64): values are tracked as multiples of q at compile time.  A stage
//   takes x < B*q to x' = u + t, y' = u - t + 2q < (B+2)*q, so it needs B <= 6; when B > 6 the
//   stage first subtracts 4q from x >= 4q (B <= 8 -> 4).  BIN is the bound of the round's inputs
//   (2 for the first round of a transform — canonical inputs, with slack —, 6 after a round),
//   fwd_bound_out the bound of its outputs: one conditional subtraction per TWO butterflies in
//   steady state, one per FOUR in a first round.  TIGHT_LAST: the round's last stage brings x
//   below 2q first, so the outputs are below 4q (cheaper to canonicalise than < 6q or < 8q).
//   END6: the round's last stage also corrects when its inputs exceed 4q, so the round (the last
//   of a strided pass) ends below 6q whatever its length — the bound the next pass starts from.
//   otherwise (q < 2^62): Harvey's [0,4q) with a 2q correction in every butterfly.
//   WIDE == kStrict (2^62 <= q < 2^63, AR = 3): no lazy range at all — ct_bfly63 / gs_bfly63 take canonical values to
//   canonical values (zq_device.hpp); the bound parameters are carried along and ignored.
constexpr int kStrict = 3;
constexpr int fwd_stage_needs_csub(int bound_in) { return bound_in > 6; }
constexpr int fwd_bound_out(int R, int bin) {
    int b = bin;
    for (int i = 0; i < R; i++) b = (fwd_stage_needs_csub(b) ? 4 : b) + 2;
    return b;
}
constexpr int kPassBound = 6;   // bound (in q) of what a forward strided pass hands to the contiguous pass
// I0: the round's first I0 stages have been done by other means (the table look-ups of round0_bits):
// run stages I0 .. R-1, BIN being the bound of what enters stage I0.
template <int R, int WIDE, int BIN = 6, bool TIGHT_LAST = false, bool END6 = false, int I0 = 0>
__device__ __forceinline__ void round_fwd(u64 (&v)[16], const Tw *__restrict__ tw, u32 T0,
                                          const Mod &m) {
    static_assert(BIN >= 1 && BIN <= 8, "input bound out of range");
#pragma unroll
    for (int i = I0; i < R; i++) {
        const int span = 8 >> i;
        constexpr int kNone = 0;
        const int bin_i = fwd_bound_out(i - I0, BIN);            // bound of this stage's inputs
        const bool corr = fwd_stage_needs_csub(bin_i) || (END6 && i == R - 1 && bin_i > 4);
        const bool tight = TIGHT_LAST && i == R - 1;             // bring x below 2q: x' and y' < 4q
#ifdef FHE_ABLATE_NO_BUTTERFLIES   // timing-only build: memory pattern without the arithmetic
        if (i >= 0) continue;
#endif
#pragma unroll
        for (int g = 0; g < (1 << i); g++) {
            const Tw t = tw[(T0 << i) + g];
#pragma unroll
            for (int l = 0; l < span; l++) {
                const int k = g * 2 * span + l;
                if (WIDE == kStrict) ct_bfly63(v[k], v[k + span], t.w, t.wp, m);
                else if (!WIDE) ct_bfly<2>(v[k], v[k + span], t.w, t.wp, m);
                else if (tight && bin_i > 4) ct_bfly<6>(v[k], v[k + span], t.w, t.wp, m);
                else if (tight && bin_i > 2) ct_bfly<2>(v[k], v[k + span], t.w, t.wp, m);
                else if (tight) ct_bfly<kNone>(v[k], v[k + span], t.w, t.wp, m);
                else if (corr) ct_bfly<4>(v[k], v[k + span], t.w, t.wp, m);
                else ct_bfly<kNone>(v[k], v[k + span], t.w, t.wp, m);
            }
        }
    }
}

// FOLD: this round contains the transform's last GS stage (m = 1, ntt.rs:85 loop
// exit) and the n^-1 scaling of ntt.rs:100-102 is folded into it:
//   r[j] = (U+V)*n_inv,  r[j+t] = (U-V)*(roots_inv[1]*n_inv).
template <int R, bool FOLD>
__device__ __forceinline__ void round_inv(u64 (&v)[16], const Tw *__restrict__ tw, u32 T0,
                                          const Mod &m, const Tw ninv, const Tw s_ninv) {
#pragma unroll
    for (int i = R - 1; i >= 0; i--) {
        const int span = 8 >> i;
#pragma unroll
        for (int g = 0; g < (1 << i); g++) {
            if (FOLD && i == 0) {
#pragma unroll
                for (int l = 0; l < span; l++) {
                    const int k = g * 2 * span + l;
                    const u64 s = add64(v[k], v[k + span]);                    // < 4q, fine for Shoup
                    const u64 d = add64(add64(v[k], m.q2p1), ~v[k + span]);    // x - y + 2q
                    v[k] = mul_shoup_lazy(s, ninv.w, ninv.wp, m);
                    v[k + span] = mul_shoup_lazy(d, s_ninv.w, s_ninv.wp, m);
                }
            } else {
                const Tw t = tw[(T0 << i) + g];
#pragma unroll
                for (int l = 0; l < span; l++) {
                    const int k = g * 2 * span + l;
                    gs_bfly(v[k], v[k + span], t.w, t.wp, m);
                }
            }
        }
    }
}

// ---- inverse rounds for q < 2^61 (8q < 2^64): per-register value bounds -----------------------
// A Gentleman-Sande butterfly only needs x + y < 2^64 and x - y + K*q > 0: with bounds (in q) bx, by
// of its inputs, bx + by <= 8 is enough, the sum leaves with bound bx + by and the product with 2.
// So instead of one conditional subtraction per butterfly the round follows the bounds of its 16
// registers at compile time and subtracts 4q only where a pair would exceed 8: 12 instead of 32
// per round from canonical inputs, 20 instead of 32 in steady state.  A round starts from a uniform
// bound BIN (after the LDS transpose a register may come from any register of another thread) and
// ends by bringing every register below BOUT*q.
struct InvSched {
    unsigned char cx[4][8];   // stage (in execution order), butterfly: x -= 4q if x >= 4q first
    unsigned char cy[4][8];   // same for y
    unsigned char ky[4][8];   // bound of y entering the subtraction: d = x - y + ky*q
    unsigned char fin[16];    // register: final conditional subtraction of 4q
};
constexpr InvSched inv_sched(int R, int bin, bool fold, int bout) {
    InvSched s{};
    int B[16] = {};
    for (int k = 0; k < 16; k++) B[k] = bin;
    int st = 0;
    for (int i = R - 1; i >= 0; i--, st++) {
        const int span = 8 >> i;
        int j = 0;
        for (int g = 0; g < (1 << i); g++)
            for (int l = 0; l < span; l++, j++) {
                const int k = g * 2 * span + l, k2 = k + span;
                int bx = B[k], by = B[k2];
                bool cx = false, cy = false;
                if (bx + by > 8) {
                    if (bx >= by) { cx = true; bx = bx > 4 ? 4 : bx; }
                    else { cy = true; by = by > 4 ? 4 : by; }
                }
                if (bx + by > 8) {
                    if (!cx) { cx = true; bx = bx > 4 ? 4 : bx; }
                    else { cy = true; by = by > 4 ? 4 : by; }
                }
                s.cx[st][j] = cx;
                s.cy[st][j] = cy;
                s.ky[st][j] = (unsigned char)by;
                B[k] = (fold && i == 0) ? 2 : bx + by;
                B[k2] = 2;
            }
    }
    for (int k = 0; k < 16; k++) s.fin[k] = B[k] > bout;
    return s;
}

template <int R, bool FOLD, int BIN, int BOUT>
__device__ __forceinline__ void round_inv_w(u64 (&v)[16], const Tw *__restrict__ tw, u32 T0, const Mod &m,
                                            const Tw ninv, const Tw s_ninv) {
    static_assert(BIN == 2 || BIN == 4, "rounds start from canonical (2) or normalised (4) inputs");
    static_assert(BOUT == 4, "rounds end below 4q");
    constexpr InvSched S = inv_sched(R, BIN, FOLD, BOUT);
    const u64 q4 = 0ull - m.neg4q;
    const u64 K[5] = {0ull, m.q2p1, q4 + 1ull, q4 + m.q2p1, 2ull * q4 + 1ull};   // (2j)*q + 1
    int st = 0;
#pragma unroll
    for (int i = R - 1; i >= 0; i--, st++) {
        const int span = 8 >> i;
#pragma unroll
        for (int g = 0; g < (1 << i); g++) {
            Tw t{};
            if (!(FOLD && i == 0)) t = tw[(T0 << i) + g];
#pragma unroll
            for (int l = 0; l < span; l++) {
                const int k = g * 2 * span + l, j = g * span + l;
                u64 x = v[k], y = v[k + span];
                if (S.cx[st][j]) x = csub_neg(x, m.neg4q);
                if (S.cy[st][j]) y = csub_neg(y, m.neg4q);
                const u64 d = add64(add64(x, K[S.ky[st][j] / 2]), ~y);        // x - y + ky*q  in (0, 8q)
                const u64 s = add64(x, y);                                      // < 8q
                if (FOLD && i == 0) {
                    v[k] = mul_shoup_lazy(s, ninv.w, ninv.wp, m);
                    v[k + span] = mul_shoup_lazy(d, s_ninv.w, s_ninv.wp, m);
                } else {
                    v[k] = s;
                    v[k + span] = mul_shoup_acc(0, d, t.w, t.wp, m.nq);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (S.fin[k]) v[k] = csub_neg(v[k], m.neg4q);
}

// the inverse round on canonical values (WIDE == kStrict); FOLD as in round_inv
template <int R, bool FOLD>
__device__ __forceinline__ void round_inv63(u64 (&v)[16], const Tw *__restrict__ tw, u32 T0, const Mod &m, const Tw ninv,
                                            const Tw s_ninv) {
#pragma unroll
    for (int i = R - 1; i >= 0; i--) {
        const int span = 8 >> i;
#pragma unroll
        for (int g = 0; g < (1 << i); g++) {
            Tw t{};
            if (!(FOLD && i == 0)) t = tw[(T0 << i) + g];
#pragma unroll
            for (int l = 0; l < span; l++) {
                const int k = g * 2 * span + l;
                if (FOLD && i == 0) {   // the sum and the difference go straight into the scaling products: any 64-bit multiplicand
                    const u64 s = v[k] + v[k + span], d = v[k] + (m.q - v[k + span]);
                    v[k] = mul63(s, ninv.w, ninv.wp, m);
                    v[k + span] = mul63(d, s_ninv.w, s_ninv.wp, m);
                } else {
                    gs_bfly63(v[k], v[k + span], t.w, t.wp, m);
                }
            }
        }
    }
}

// WIDE: the bound-tracking rounds above (values below 4q between rounds); otherwise [0,2q) throughout
template <int R, bool FOLD, int WIDE, int BIN>
__device__ __forceinline__ void round_inv_sel(u64 (&v)[16], const Tw *__restrict__ tw, u32 T0, const Mod &m,
                                              const Tw ninv, const Tw s_ninv) {
    if constexpr (WIDE == kStrict) round_inv63<R, FOLD>(v, tw, T0, m, ninv, s_ninv);
    else if constexpr (WIDE == 1) round_inv_w<R, FOLD, BIN, 4>(v, tw, T0, m, ninv, s_ninv);
    else round_inv<R, FOLD>(v, tw, T0, m, ninv, s_ninv);
}


// ---- rounds for pseudo-Mersenne moduli (zq_device.hpp: ct_bfly_pm / gs_bfly_pm) ----------------------------------
// Bounds are tracked in SIXTEENTHS of q (kPmOne = 16).  A forward stage takes x < B to x' = u + r < B + 33 and
// y' = u - r + 3q < B + 48, whatever y was; x is reduced first (x -> (x mod 2^k) + (x >> k) delta, below 17) when
// B + 48 would pass 8q: every other stage.  A pass hands the next one values below kPmPassBound.
constexpr bool pm_fwd_needs_red(int b) { return b + 3 * kPmOne > kPmCap; }
// AK = 2: the pseudo-Mersenne butterflies; AK = 4: the word-Montgomery ones (zq_device.hpp: ct_bfly_mg) — the same growth
// per stage (x' = u + r < B + 48, y' = u + 3q - r < B + 48), but a reduction is a conditional subtraction of 4q and only
// brings a value below 4q: from canonical inputs the bounds run 16, 64, 112, then 112 after every further stage.
constexpr int pm_fwd_bound_out(int R, int bin, int ak = 2) {
    int b = bin;
    for (int i = 0; i < R; i++) b = (pm_fwd_needs_red(b) ? (ak == 4 ? kMgRed : kPmRed) : b) + 3 * kPmOne;
    return b;
}
constexpr int kPmPassBound = 113;   // what 8 stages from canonical inputs (and from 113 again) end with (AK = 4: 112)

template <int R, int BIN, bool SGPR_TW, int AK = 2>
__device__ __forceinline__ void round_fwd_pm(u64 (&v)[16], const Tw *__restrict__ tw, u32 T0, const Mod &m) {
    static_assert(AK == 2 || AK == 4, "pseudo-Mersenne or word-Montgomery tables");
    static_assert(BIN >= kPmOne && BIN <= kPmCap, "input bound out of range");
    static_assert(pm_fwd_bound_out(R, BIN, AK) <= kPmCap, "a stage would overflow");
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int span = 8 >> i;
        const bool red = pm_fwd_needs_red(pm_fwd_bound_out(i, BIN, AK));
#ifdef FHE_ABLATE_NO_BUTTERFLIES
        if (i >= 0) continue;
#endif
#pragma unroll
        for (int g = 0; g < (1 << i); g++) {
            const Tw t = tw[(T0 << i) + g];
#pragma unroll
            for (int l = 0; l < span; l++) {
                const int k = g * 2 * span + l;
                if constexpr (AK == 4) {
                    if (red) v[k] = csub_neg(v[k], m.neg4q);
                    ct_bfly_mg<SGPR_TW>(v[k], v[k + span], t.w, t.wp, m);
                } else {
                    if (red) v[k] = pm_reduce(v[k], m);
                    ct_bfly_pm<SGPR_TW>(v[k], v[k + span], t.w, t.wp, m);
                }
            }
        }
    }
}

// Inverse rounds: a Gentleman-Sande butterfly needs x + y < 8q and x - y + K q < 8q with K q >= the bound of y
// (K = ceil(by / 16)); the sum leaves with bx + by, the product with 33.  The bounds of the 16 registers are followed
// at compile time (as inv_sched does for the Shoup rounds) and a register is reduced only where a pair would not fit;
// a round ends by bringing every register below BOUT (after the LDS transpose a register may come from any register).
struct PmInvSched {
    unsigned char rx[4][8];   // stage (execution order), butterfly: reduce x first
    unsigned char ry[4][8];   // reduce y first
    unsigned char ky[4][8];   // K of d = x - y + K q
    unsigned char fin[16];    // final reduction of the register
};
constexpr bool pm_gs_fits(int bx, int by) { return bx + by <= kPmCap && bx + kPmOne * ((by + kPmOne - 1) / kPmOne) <= kPmCap; }
constexpr PmInvSched pm_inv_sched(int R, int bin, bool fold, int bout, int ak = 2) {
    const int kRed = ak == 4 ? kMgRed : kPmRed, kMul = ak == 4 ? kMgMul : kPmMul;      // AK = 4: the word-Montgomery butterflies
    PmInvSched s{};
    int B[16] = {};
    for (int k = 0; k < 16; k++) B[k] = bin;
    int st = 0;
    for (int i = R - 1; i >= 0; i--, st++) {
        const int span = 8 >> i;
        int j = 0;
        for (int g = 0; g < (1 << i); g++)
            for (int l = 0; l < span; l++, j++) {
                const int k = g * 2 * span + l, k2 = k + span;
                int bx = B[k], by = B[k2];
                bool rx = false, ry = false;
                if (!pm_gs_fits(bx, by)) {
                    if (bx >= by) { rx = true; bx = kRed; } else { ry = true; by = kRed; }
                }
                if (!pm_gs_fits(bx, by)) {
                    if (!rx) { rx = true; bx = kRed; } else { ry = true; by = kRed; }
                }
                s.rx[st][j] = rx;
                s.ry[st][j] = ry;
                s.ky[st][j] = (unsigned char)((by + kPmOne - 1) / kPmOne);
                B[k] = (fold && i == 0) ? kMul : bx + by;
                B[k2] = kMul;
            }
    }
    for (int k = 0; k < 16; k++) s.fin[k] = B[k] > bout;
    return s;
}
constexpr int kPmInvBound = 33;   // what an inverse round (and pass) hands on: products are below it as they are
constexpr int kMgInvBound = 64;   // ... on the Montgomery tables: what a conditional subtraction of 4q leaves (products: 48)
constexpr int ar_inv_bound(int ak) { return ak == 4 ? kMgInvBound : kPmInvBound; }

template <int R, bool FOLD, int BIN, bool SGPR_TW, int AK = 2>
__device__ __forceinline__ void round_inv_pm(u64 (&v)[16], const Tw *__restrict__ tw, u32 T0, const Mod &m,
                                             const Tw ninv, const Tw s_ninv) {
    static_assert(AK == 2 || AK == 4, "pseudo-Mersenne or word-Montgomery tables");
    constexpr PmInvSched S = pm_inv_sched(R, BIN, FOLD, ar_inv_bound(AK), AK);
    int st = 0;
#pragma unroll
    for (int i = R - 1; i >= 0; i--, st++) {
        const int span = 8 >> i;
#pragma unroll
        for (int g = 0; g < (1 << i); g++) {
            Tw t{};
            if (!(FOLD && i == 0)) t = tw[(T0 << i) + g];
#pragma unroll
            for (int l = 0; l < span; l++) {
                const int k = g * 2 * span + l, j = g * span + l;
                u64 x = v[k], y = v[k + span];
                if (S.rx[st][j]) x = AK == 4 ? csub_neg(x, m.neg4q) : pm_reduce(x, m);
                if (S.ry[st][j]) y = AK == 4 ? csub_neg(y, m.neg4q) : pm_reduce(y, m);
                const u64 kq1 = (u64)S.ky[st][j] * m.q + 1ull;
                if (FOLD && i == 0) {
                    const u64 s = x + y, d = x + kq1 + ~y;
                    if constexpr (AK == 4) {
                        x = mul_mg<true>(s, ninv.w, ninv.wp, m);    // kernel arguments: wave-uniform
                        y = mul_mg<true>(d, s_ninv.w, s_ninv.wp, m);
                    } else {
                        x = mul_pm<true>(s, ninv.w, ninv.wp, m);
                        y = mul_pm<true>(d, s_ninv.w, s_ninv.wp, m);
                    }
                } else if constexpr (AK == 4) {
                    gs_bfly_mg<SGPR_TW>(x, y, t.w, t.wp, kq1, m);
                } else {
                    gs_bfly_pm<SGPR_TW>(x, y, t.w, t.wp, kq1, m);
                }
                v[k] = x;
                v[k + span] = y;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (S.fin[k]) v[k] = AK == 4 ? csub_neg(v[k], m.neg4q) : pm_reduce(v[k], m);
}

// ---- round 0 of a transform whose inputs are BITS (gadget digits, base 2) -------------------------
// The first stages of the transform of a 0/1 polynomial need no multiplication at all: stages 0 and 1
// act on the four registers {c, c+4, c+8, c+12} of a thread, whose 4 input bits select one of 16
// possible outcomes — a table look-up (Y); stage 2 then adds / subtracts w * Y[..], which is again one
// of 16 values per twiddle (WY).  Per thread that replaces 24 of the round's Shoup butterflies by 16
// (R0 = 2) or 16 + 16 modular additions' worth of LDS reads.  Tables per plan, built on the host
// (capi.hip: build_digit_lut), staged into LDS by the digit kernels:
//   [0,64)    Y[p][j]    j-th output of stages 0,1 on the bit pattern p = x_c + 2 x_{c+4} + 4 x_{c+8} + 8 x_{c+12}
//   [64,128)  WY[p][j]   roots[4+j] * Y[p][j]  (stage 2's twiddle for registers with k >> 2 = j)
//   [128,136) Y1[p][h]   stage 0 alone on a pair: p = x + 2y -> (x + W0 y, x - W0 y)   (passes with R0 = 1)
// All entries canonical, so the stages that follow start from bound 1.
constexpr int kDigitLutWords = 136;

template <int R0, int WIDE, bool TIGHT>
__device__ __forceinline__ void round0_bits(u64 (&v)[16], const u64 *lut, const Tw *__restrict__ gtw, const Mod &m) {
    if constexpr (R0 == 1) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u32 p = (u32)v[k] + 2u * (u32)v[k + 8];
            const ulonglong2 y = *reinterpret_cast<const ulonglong2 *>(lut + 128 + 2 * p);
            v[k] = y.x;
            v[k + 8] = y.y;
        }
    } else {
        u32 p[4];
#pragma unroll
        for (int c = 0; c < 4; c++) p[c] = (u32)v[c] + 2u * (u32)v[c + 4] + 4u * (u32)v[c + 8] + 8u * (u32)v[c + 12];
        if constexpr (R0 == 2) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const ulonglong2 lo = *reinterpret_cast<const ulonglong2 *>(lut + 4 * p[c]);
                const ulonglong2 hi = *reinterpret_cast<const ulonglong2 *>(lut + 4 * p[c] + 2);
                v[c] = lo.x; v[c + 4] = lo.y; v[c + 8] = hi.x; v[c + 12] = hi.y;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                u64 a[4], b[4];
                const ulonglong2 a0 = *reinterpret_cast<const ulonglong2 *>(lut + 4 * p[c]);
                const ulonglong2 a1 = *reinterpret_cast<const ulonglong2 *>(lut + 4 * p[c] + 2);
                const ulonglong2 b0 = *reinterpret_cast<const ulonglong2 *>(lut + 64 + 4 * p[c + 2]);
                const ulonglong2 b1 = *reinterpret_cast<const ulonglong2 *>(lut + 64 + 4 * p[c + 2] + 2);
                a[0] = a0.x; a[1] = a0.y; a[2] = a1.x; a[3] = a1.y;
                b[0] = b0.x; b[1] = b0.y; b[2] = b1.x; b[3] = b1.y;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const u64 s = a[j] + b[j];                       // both < q < 2^63
                    v[c + 4 * j] = s >= m.q ? s - m.q : s;
                    v[c + 2 + 4 * j] = a[j] >= b[j] ? a[j] - b[j] : a[j] + m.q - b[j];
                }
            }
            if constexpr (R0 == 4) round_fwd<4, WIDE, 1, TIGHT, false, 3>(v, gtw, 1u, m);
        }
    }
}

// field value of register k for a thread whose non-register field bits are tf,
// register window = field bits [A, A+4)
template <int A>
__device__ __forceinline__ u32 field_of(u32 tf, int k) {
    const u32 lo = tf & ((1u << A) - 1u);
    const u32 hi = tf >> A;
    return (hi << (A + 4)) | ((u32)k << A) | lo;
}

// LDS slot of tile element e in the contiguous kernels: one 8-byte pad every 16
// elements so that the a=0 window (lane stride 16 elements) is conflict-free.
__device__ __forceinline__ u32 pad16(u32 e) { return e + (e >> 4); }

// stage a pass-local twiddle table into LDS: local index li in [1, M): ls = floor(log2 li),
// global index (1 << (s0+ls)) + (blk << ls) + (li - 2^ls); the rounds then index it with
// T0 = (1 << ls0) + H, i.e. as if the pass were a transform of its own.
template <int M, int TH>
__device__ __forceinline__ void stage_twiddles(Tw *ltw, const Tw *__restrict__ tw, u32 s0, u32 blk,
                                               u32 tid) {
    for (u32 li = tid; li < (u32)M; li += TH) {
        const u32 ls = 31u - (u32)__builtin_clz(li | 1u);   // entry 0 is never used: copy tw[.] of li = 1
        const u32 l1 = li | (li == 0);
        ltw[li] = tw[(1u << (s0 + ls)) + (blk << ls) + (l1 - (1u << ls))];
    }
}

// Global access as (wave-uniform 64-bit base) + (32-bit per-lane BYTE offset): the form the
// saddr/voffset addressing mode takes, so an access costs one v_add_u32, not 64-bit arithmetic.
// Coefficient data is touched once per pass, so accesses in which a wave instruction covers whole
// cache lines carry the non-temporal hint (plain copy: 5.41 -> 5.71 TB/s with it,
// tools/ubench_mem.hip; forward 2^16 transform 7.40 -> 7.25 ms).  The inverse contiguous pass reads
// and writes a line in pieces spread over several instructions and needs the cache to merge them:
// with the hint it ran 3.69 -> 5.5 ms, so it uses the plain forms (ld_c / st_c).
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ T ld_at(const u64 *ubase, u32 byte_off);
template <>
__device__ __forceinline__ u64 ld_at<u64>(const u64 *ubase, u32 byte_off) {
    return __builtin_nontemporal_load(
        reinterpret_cast<const u64 *>(reinterpret_cast<const unsigned char *>(ubase) + byte_off));
}
__device__ __forceinline__ void st_at(u64 *ubase, u32 byte_off, u64 x) {
    __builtin_nontemporal_store(x, reinterpret_cast<u64 *>(reinterpret_cast<unsigned char *>(ubase) + byte_off));
}
template <typename T>
__device__ __forceinline__ T ld_c(const u64 *ubase, u32 byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const unsigned char *>(ubase) + byte_off);
}
template <typename T>
__device__ __forceinline__ void st_c(u64 *ubase, u32 byte_off, T x) {
    *reinterpret_cast<T *>(reinterpret_cast<unsigned char *>(ubase) + byte_off) = x;
}

// Sixteen accesses whose element indices are (k << sh) + (a per-lane term), k = 0..15: the k-term goes onto the
// wave-uniform 64-bit base — scalar adds, or the instruction's immediate offset where it fits — and ONE per-lane byte
// offset serves all sixteen.  Written as `base + off + k-term` in 32-bit lane arithmetic (rounds 1-4) every access
// cost one to three vector instructions of its own (v_or / v_lshlrev / v_add_lshl: the compiler may not fold a 32-bit
// add into the 64-bit address): 96 of the strided kernel's 1032 vector instructions per wave (round 5).
// NT: the non-temporal forms (whole lines per wave instruction, touched once).
// A wave-uniform address pinned to a scalar register pair and made opaque: left to itself the compiler reassociates
// (base + k-term) + lane offset into (base + lane offset) + k-term, a 64-bit vector add per access.  The opaque value is
// an integer, so the access names the global address space itself (a laundered generic pointer becomes a flat_ access).
typedef __attribute__((address_space(1))) u64 gu64;
__device__ __forceinline__ u64 scalar_addr(const u64 *p) {
    u64 a = reinterpret_cast<u64>(p);
    asm("" : "+s"(a));
    return a;
}
template <bool NT>
__device__ __forceinline__ u64 ld_s(u64 saddr, u32 lane_boff) {   // global load at (scalar address) + (32-bit lane offset)
    const gu64 *p = reinterpret_cast<const gu64 *>(saddr + lane_boff);
    return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT>
__device__ __forceinline__ void st_s(u64 saddr, u32 lane_boff, u64 x) {
    gu64 *p = reinterpret_cast<gu64 *>(saddr + lane_boff);
    if (NT) __builtin_nontemporal_store(x, p); else *p = x;
}
// LOOP: the accesses sit in a loop whose lane offset is loop-invariant: the offset is made opaque once per call, or the
// compiler hoists its zero-extension out of the loop and every access becomes a 64-bit vector add + a vaddr access
template <bool NT, bool LOOP = false>
__device__ __forceinline__ void ld16(u64 (&v)[16], const u64 *ubase, u32 sh, u32 lane_boff) {
    if (LOOP) asm volatile("" : "+v"(lane_boff));
#pragma unroll
    for (int k = 0; k < 16; k++) {
        v[k] = ld_s<NT>(scalar_addr(ubase + ((u64)k << sh)), lane_boff);
    }
}
template <bool NT, bool LOOP = false>
__device__ __forceinline__ void st16(u64 *ubase, u32 sh, u32 lane_boff, const u64 (&v)[16]) {
    if (LOOP) asm volatile("" : "+v"(lane_boff));
#pragma unroll
    for (int k = 0; k < 16; k++) {
        st_s<NT>(scalar_addr(ubase + ((u64)k << sh)), lane_boff, v[k]);
    }
}

// ---------------------------------------------------------------------------
// CONTIGUOUS pass: blocks of M = 2^LP consecutive coefficients.
// Workgroup = W units (unit = one M-block of one polynomial, all W units share
// `blk`, hence the twiddles), TPB = M/16 threads per unit.
// ---------------------------------------------------------------------------
#ifndef FHE_CONTIG_TH8
#define FHE_CONTIG_TH8 256        // threads of a workgroup around 256-point blocks (shape experiments: 128 / 512)
#endif
template <int LP>
struct ContigCfg {
    static constexpr int M = 1 << LP;
    static constexpr int TPB = M / 16;
    static constexpr int TH = (LP == 8) ? FHE_CONTIG_TH8 : (LP <= 12) ? 256 : 512;
    static constexpr int W = TH / TPB;
    static constexpr int TILE = W * M;  // = 16 * TH
    static constexpr int NR = (LP + 3) / 4;
    static constexpr int R0 = LP - 4 * (NR - 1);
    static constexpr int A0 = LP - 4;  // register window of round 0 = top 4 field bits
    // The first 2^LTW_LOG entries of the pass-local twiddle table (all W units share `blk`) are
    // staged into LDS once per workgroup: a round whose stages all lie below local stage
    // LTW_LOG reads them with ds_read_b128 instead of 15 global loads through L1.  Later rounds
    // (LP > 8: per-thread-unique twiddles, up to 64 KiB per block) stay on the global table.
    static constexpr int LTW_LOG = LP < 8 ? LP : 8;
    static constexpr int LTW_N = 1 << LTW_LOG;
    static constexpr size_t DATA_BYTES = (size_t)(TILE + TILE / 16) * 8;
    static constexpr size_t LDS_BYTES = DATA_BYTES + (size_t)LTW_N * sizeof(Tw);
    static constexpr size_t LDS_BYTES_BITS = LDS_BYTES + (size_t)kDigitLutWords * 8;   // + the digit tables (BITS kernels)
    // window base of round j >= 1
    static constexpr int a_of(int j) { return j == 0 ? A0 : LP - R0 - 4 * j; }
    static constexpr int ls0_of(int j) { return j == 0 ? 0 : R0 + 4 * (j - 1); }
    static constexpr bool in_lds(int j) { return ls0_of(j) + (j == 0 ? R0 : 4) <= LTW_LOG; }
};

// The workgroup's tile taken in SLAB order: element e = i TH + tid, i = 0..15 — a wave instruction covers 512 contiguous
// bytes, whole lines.  e splits into polynomial wu = e >> LP and position f = e & (M - 1); i TH is a multiple of min(TH, M),
// so the i-term of both is wave-uniform and goes onto the scalar base, the lane keeps one byte offset for all sixteen
// accesses and one LDS slot (pad16 is additive over multiples of 16).  Round 5: written per element, every store of the
// forward kernel cost seven vector instructions, a branch and its own LDS wait (profiles/r05_isa_*).
template <int LP>
struct SlabIo {
    using C = ContigCfg<LP>;
    static constexpr bool kLaneWu = C::TH > C::M;                       // several polynomials per slab: the lane picks one
    static constexpr u32 wu_u(int i) { return (u32)(i * C::TH) >> LP; }
    static constexpr u32 f_u(int i) { return (u32)(i * C::TH) & (u32)(C::M - 1); }
    static constexpr u32 lds_step = C::TH + C::TH / 16;                  // pad16(i TH + tid) = i lds_step + pad16(tid)
    u32 wu_l, boff, slot;
    __device__ __forceinline__ SlabIo(u32 tid, u32 log_n) {
        wu_l = kLaneWu ? tid >> LP : 0u;
        const u32 f_l = kLaneWu ? tid & (u32)(C::M - 1) : tid;
        boff = ((wu_l << log_n) + f_l) * 8u;
        slot = tid + (tid >> 4);
    }
    // scalar address of slab i of the tile at p (for ld_s / st_s with `boff`)
    __device__ __forceinline__ u64 base(const u64 *p, int i, u32 log_n) const { return scalar_addr(p + ((u64)wu_u(i) << log_n) + f_u(i)); }
    __device__ __forceinline__ u32 wu(int i) const { return wu_u(i) + wu_l; }
};

// scatter registers (window AF) -> barrier -> gather registers (window AT).
// FIRST = false: the tile was read by an earlier exchange, so a barrier precedes the scatter.
// No trailing barrier: whoever writes the tile next either is this function (FIRST = false)
// or writes exactly the slots it has just gathered (the store transpose of the forward pass).
// pad16 slot of register k of window A, as (slot of register 0) + (a constant): the bit fields of field_of<A>(tf, k) are
// disjoint, so both e and e >> 4 split into the thread's part and the k-part — (k << A) + ((k << A) >> 4).  Written as
// pad16(w M + field_of<A>(tf, k)) per register, the compiler recomputed the slot for every k at LP >= 9 (v_or + v_lshr +
// v_add3 per access: ~100 vector instructions per exchange of the N = 4096 kernels, round 5).
template <int A>
constexpr u32 pad16_koff(int k) { return ((u32)k << A) + (((u32)k << A) >> 4); }
template <int LP, int AF, int AT, bool FIRST>
__device__ __forceinline__ void exchange_contig(u64 (&v)[16], u64 *lds, u32 w, u32 tf) {
    constexpr int M = 1 << LP;
    const u32 bf = pad16(w * M + field_of<AF>(tf, 0)), bt = pad16(w * M + field_of<AT>(tf, 0));
    if (!FIRST) __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) lds[bf + pad16_koff<AF>(k)] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = lds[bt + pad16_koff<AT>(k)];
}

// The LP stages of a contiguous pass on the 16 registers of each thread: round 0 in the window the
// load filled (field bits [LP-4, LP)), an LDS exchange before every later round, ending in window
// [0,4) (16 consecutive coefficients per thread).  `ltw` is the pass-local twiddle tile staged by
// stage_twiddles (published by the first exchange's barrier), `gtw` the global table for the rounds
// that do not fit it; s0 / blk place the block in the transform (0, 0: the pass is the transform).
// BIN: bound (in q) of the inputs; TIGHT: this pass holds the transform's last stage, which then
// leaves x', y' < 4q.  FRESH: nobody has read the LDS tile since the last barrier.
// BITS: the inputs are 0/1 and the pass is the whole transform (s0 = blk = 0): round 0 by table
// look-up (round0_bits; `lut` = the plan's digit tables in LDS, already published).
template <int LP, int WIDE, bool TIGHT, int BIN, bool FRESH, bool BITS = false>
__device__ __forceinline__ void fwd_rounds_contig(u64 (&v)[16], u64 *lds, const Tw *ltw, const Tw *gtw, u32 s0, u32 blk,
                                                  u32 w, u32 tf, const Mod &m, const u64 *lut = nullptr) {
    using C = ContigCfg<LP>;
    auto TW = [&](bool lds_round) -> const Tw * { return lds_round ? ltw : gtw; };
    auto T0 = [&](bool lds_round, int ls, u32 H) -> u32 {
        return lds_round ? (1u << ls) + H : (1u << (s0 + ls)) + (blk << ls) + H;
    };
    constexpr int B0 = BIN, B1 = BITS ? (C::R0 == 4 ? fwd_bound_out(1, 1) : 1) : fwd_bound_out(C::R0, B0),
                  B2 = fwd_bound_out(4, B1), B3 = fwd_bound_out(4, B2);
    // Round 0's twiddles are the same for the whole workgroup (H = 0): read from the global
    // table at a wave-uniform address (scalar loads, SGPR operands).
    if constexpr (BITS) round0_bits<C::R0, WIDE, TIGHT && C::NR == 1>(v, lut, gtw, m);
    else round_fwd<C::R0, WIDE, B0, TIGHT && C::NR == 1>(v, gtw, (1u << s0) + blk, m);
    if constexpr (C::NR > 1) {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        constexpr bool L = C::in_lds(1);
        exchange_contig<LP, C::A0, A, FRESH>(v, lds, w, tf);
        round_fwd<4, WIDE, B1, TIGHT && C::NR == 2>(v, TW(L), T0(L, LS, tf >> A), m);
    }
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        constexpr bool L = C::in_lds(2);
        exchange_contig<LP, C::a_of(1), A, false>(v, lds, w, tf);
        round_fwd<4, WIDE, B2, TIGHT && C::NR == 3>(v, TW(L), T0(L, LS, tf >> A), m);
    }
    if constexpr (C::NR > 3) {
        constexpr int A = C::a_of(3), LS = C::ls0_of(3);
        constexpr bool L = C::in_lds(3);
        exchange_contig<LP, C::a_of(2), A, false>(v, lds, w, tf);
        round_fwd<4, WIDE, B3, TIGHT && C::NR == 4>(v, TW(L), T0(L, LS, tf >> A), m);
    }
}

// The mirror image for an inverse contiguous pass: from window [0,4) (canonical inputs: the first
// round that runs starts from bound 2, later ones from the normalised 4) up to window [LP-4, LP).
// FOLD: the pass holds the transform's last GS stage (s0 = 0) with the n^-1 scaling folded in.
// FRESH as above (an exchange precedes every round but the first).
template <int LP, int WIDE, bool FOLD, bool FRESH>
__device__ __forceinline__ void inv_rounds_contig(u64 (&v)[16], u64 *lds, const Tw *ltw, const Tw *gtw, u32 s0, u32 blk,
                                                  u32 w, u32 tf, const Mod &m, const Tw ninv, const Tw s_ninv) {
    using C = ContigCfg<LP>;
    auto TW = [&](bool lds_round) -> const Tw * { return lds_round ? ltw : gtw; };
    auto T0 = [&](bool lds_round, int ls, u32 H) -> u32 {
        return lds_round ? (1u << ls) + H : (1u << (s0 + ls)) + (blk << ls) + H;
    };
    constexpr int BF = 2, BN = 4;
    if constexpr (C::NR > 3) {
        constexpr int A = C::a_of(3), LS = C::ls0_of(3);
        constexpr bool L = C::in_lds(3);
        round_inv_sel<4, false, WIDE, BF>(v, TW(L), T0(L, LS, tf >> A), m, ninv, s_ninv);
        exchange_contig<LP, A, C::a_of(2), FRESH>(v, lds, w, tf);
    }
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        constexpr bool L = C::in_lds(2);
        round_inv_sel<4, false, WIDE, (C::NR == 3 ? BF : BN)>(v, TW(L), T0(L, LS, tf >> A), m, ninv, s_ninv);
        exchange_contig<LP, A, C::a_of(1), FRESH && (C::NR <= 3)>(v, lds, w, tf);
    }
    if constexpr (C::NR > 1) {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        constexpr bool L = C::in_lds(1);
        round_inv_sel<4, false, WIDE, (C::NR == 2 ? BF : BN)>(v, TW(L), T0(L, LS, tf >> A), m, ninv, s_ninv);
        exchange_contig<LP, A, C::A0, FRESH && (C::NR <= 2)>(v, lds, w, tf);
    }
    // FOLD implies s0 == 0 (this pass holds the m = 1 stage)
    round_inv_sel<C::R0, FOLD, WIDE, (C::NR == 1 ? BF : BN)>(v, TW(C::in_lds(0)), T0(C::in_lds(0), 0, 0), m, ninv, s_ninv);
}


// The same two drivers for pseudo-Mersenne moduli (round_fwd_pm / round_inv_pm): `ltw` / `gtw` then hold {w, w 2^32 mod q}.
// Round 0 of a forward pass and the last round of an inverse pass have workgroup-uniform twiddles (H = 0): global
// table, scalar loads, SGPR operands.  The forward driver leaves values below pm_fwd_bound_out over its rounds
// (<= kPmPassBound from BIN = 16 or kPmPassBound); the inverse driver below kPmInvBound.
template <int LP, int BIN, bool FRESH, int AK = 2>
__device__ __forceinline__ void fwd_rounds_contig_pm(u64 (&v)[16], u64 *lds, const Tw *ltw, const Tw *gtw, u32 s0, u32 blk,
                                                     u32 w, u32 tf, const Mod &m) {
    using C = ContigCfg<LP>;
    auto TW = [&](bool lds_round) -> const Tw * { return lds_round ? ltw : gtw; };
    auto T0 = [&](bool lds_round, int ls, u32 H) -> u32 {
        return lds_round ? (1u << ls) + H : (1u << (s0 + ls)) + (blk << ls) + H;
    };
    constexpr int B0 = BIN, B1 = pm_fwd_bound_out(C::R0, B0, AK), B2 = pm_fwd_bound_out(4, B1, AK), B3 = pm_fwd_bound_out(4, B2, AK);
    round_fwd_pm<C::R0, B0, true, AK>(v, gtw, (1u << s0) + blk, m);
    if constexpr (C::NR > 1) {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        constexpr bool L = C::in_lds(1);
        exchange_contig<LP, C::A0, A, FRESH>(v, lds, w, tf);
        round_fwd_pm<4, B1, false, AK>(v, TW(L), T0(L, LS, tf >> A), m);
    }
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        constexpr bool L = C::in_lds(2);
        exchange_contig<LP, C::a_of(1), A, false>(v, lds, w, tf);
        round_fwd_pm<4, B2, false, AK>(v, TW(L), T0(L, LS, tf >> A), m);
    }
    if constexpr (C::NR > 3) {
        constexpr int A = C::a_of(3), LS = C::ls0_of(3);
        constexpr bool L = C::in_lds(3);
        exchange_contig<LP, C::a_of(2), A, false>(v, lds, w, tf);
        round_fwd_pm<4, B3, false, AK>(v, TW(L), T0(L, LS, tf >> A), m);
    }
}

// BFIRST: bound (sixteenths of q) of what the first round that runs receives: canonical evals from memory (kPmOne), or
// the lazy pointwise product formed in registers (kPmMul)
template <int LP, bool FOLD, bool FRESH, int BFIRST = kPmOne, int AK = 2>
__device__ __forceinline__ void inv_rounds_contig_pm(u64 (&v)[16], u64 *lds, const Tw *ltw, const Tw *gtw, u32 s0, u32 blk,
                                                     u32 w, u32 tf, const Mod &m, const Tw ninv, const Tw s_ninv) {
    using C = ContigCfg<LP>;
    auto TW = [&](bool lds_round) -> const Tw * { return lds_round ? ltw : gtw; };
    auto T0 = [&](bool lds_round, int ls, u32 H) -> u32 {
        return lds_round ? (1u << ls) + H : (1u << (s0 + ls)) + (blk << ls) + H;
    };
    constexpr int BF = BFIRST, BN = ar_inv_bound(AK);
    if constexpr (C::NR > 3) {
        constexpr int A = C::a_of(3), LS = C::ls0_of(3);
        constexpr bool L = C::in_lds(3);
        round_inv_pm<4, false, BF, false, AK>(v, TW(L), T0(L, LS, tf >> A), m, ninv, s_ninv);
        exchange_contig<LP, A, C::a_of(2), FRESH>(v, lds, w, tf);
    }
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        constexpr bool L = C::in_lds(2);
        round_inv_pm<4, false, (C::NR == 3 ? BF : BN), false, AK>(v, TW(L), T0(L, LS, tf >> A), m, ninv, s_ninv);
        exchange_contig<LP, A, C::a_of(1), FRESH && (C::NR <= 3)>(v, lds, w, tf);
    }
    if constexpr (C::NR > 1) {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        constexpr bool L = C::in_lds(1);
        round_inv_pm<4, false, (C::NR == 2 ? BF : BN), false, AK>(v, TW(L), T0(L, LS, tf >> A), m, ninv, s_ninv);
        exchange_contig<LP, A, C::A0, FRESH && (C::NR <= 2)>(v, lds, w, tf);
    }
    round_inv_pm<C::R0, FOLD, (C::NR == 1 ? BF : BN), true, AK>(v, gtw, (1u << s0) + blk, m, ninv, s_ninv);
}

// ---------------------------------------------------------------------------
// STRIDED pass: the first LA stages of a forward transform (last LA of an
// inverse), on the (2^LA rows) x (2^LB columns) view of one polynomial.
// Workgroup tile = all 2^LA rows x CW adjacent columns; lanes run along columns,
// so every global access is a CW*8-byte contiguous run and LDS needs no padding.
// ---------------------------------------------------------------------------
template <int LA, int CW>
struct StridedCfg {
    static constexpr int F = 1 << LA;
    static constexpr int TPF = F / 16;
    static constexpr int TH = TPF * CW;
    static constexpr int NR = (LA + 3) / 4;
    static constexpr int R0 = LA - 4 * (NR - 1);
    static constexpr int A0 = LA - 4;
    static constexpr size_t DATA_BYTES = (size_t)F * CW * 8;
    static constexpr size_t LDS_BYTES = DATA_BYTES + (size_t)F * sizeof(Tw);  // + the 2^LA twiddles
    static constexpr int a_of(int j) { return j == 0 ? A0 : LA - R0 - 4 * j; }
    static constexpr int ls0_of(int j) { return j == 0 ? 0 : R0 + 4 * (j - 1); }
};

template <int CW, int AF, int AT, bool FIRST>
__device__ __forceinline__ void exchange_strided(u64 (&v)[16], u64 *lds, u32 c, u32 tf) {
    if (!FIRST) __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) lds[field_of<AF>(tf, k) * CW + c] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = lds[field_of<AT>(tf, k) * CW + c];
}

// pass split for n >= 2^14: LB = max(8, L-8) contiguous stages, LA = L-LB in 6..8
static inline int contig_bits(int L) { return L <= kMaxSinglePassLog ? L : (L - 8 > 8 ? L - 8 : 8); }

// dynamic LDS above 64 KiB must be opted into per kernel
static inline hipError_t allow_big_lds(const void *fn, size_t bytes) {
    if (bytes <= 65536) return hipSuccess;
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace fhe
