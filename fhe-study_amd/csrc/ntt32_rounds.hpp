// ntt32_rounds.hpp — device-side pieces of the 32-bit transforms modulo the two small primes of digit32.hpp: Z_p
// arithmetic in one word, the rounds of <= 4 stages on a thread's 16 coefficients, the LDS exchange between rounds.
// Shared by digit32.hip (gadget products, n <= 4096) and bfv32.hip (BFV tensor / relinearisation, 2n <= 16384).
// Index algebra (register windows, field_of, pad16) is that of ntt_rounds.hpp.
#pragma once
#include "digit32.hpp"
#include "ntt_rounds.hpp"

namespace fhe {

// ---- Z_p arithmetic in one 32-bit word, 4p < 2^32 (25p < 2^32 for the loose forms) ---------------------------------------------------------------

__device__ __forceinline__ u32 csub_u32(u32 x, u32 m) { return min(x, x - m); }     // x - m if x >= m (x < 2m), else x
// y * w mod p, lazily in [0, 2p), for ANY 32-bit y and w < p
__device__ __forceinline__ u32 mul_shoup32(u32 y, Tw32 t, u32 p) { return y * t.w - __umulhi(y, t.wp) * p; }
// Cooley-Tukey butterfly (ntt.rs:57-62), Harvey's lazy form: x, y in [0,4p) -> [0,4p)
__device__ __forceinline__ void ct32(u32 &x, u32 &y, Tw32 t, u32 p, u32 p2) {
    const u32 u = csub_u32(x, p2);
    const u32 v = mul_shoup32(y, t, p);
    x = u + v;
    y = u - v + p2;
}
// Gentleman-Sande butterfly (ntt.rs:91-96): x, y in [0,2p) -> [0,2p)
__device__ __forceinline__ void gs32(u32 &x, u32 &y, Tw32 t, u32 p, u32 p2) {
    const u32 d = x - y + p2;
    x = csub_u32(x + y, p2);
    y = mul_shoup32(d, t, p);
}
__device__ __forceinline__ u32 canon4_32(u32 x, u32 p, u32 p2) { return csub_u32(csub_u32(x, p2), p); }
// The same butterfly with NO conditional subtraction: the bound of the values grows by 2p per stage, and with p below
// 2^32 / 25 (digit32.hpp) twelve stages from canonical inputs stay in one word: 1 + 2*12 = 25.
// Six instructions: the NEGATED lazy product nv = q p - y w (one subtraction), x - nv, and x + nv + 2p as one v_add3_u32.
__device__ __forceinline__ void ct32_loose(u32 &x, u32 &y, Tw32 t, u32 p, u32 p2) {
    const u32 nv = __umulhi(y, t.wp) * p - y * t.w;
    const u32 u = x;
    x = u - nv;
    y = u + nv + p2;
}
// any 32-bit x -> [0, 2p), bq = floor(2^32 / p): the quotient estimate is short by at most one
__device__ __forceinline__ u32 barrett2p_32(u32 x, u32 p, u32 bq) { return x - __umulhi(x, bq) * p; }

template <int R, int I0 = 0, bool LOOSE = false>
__device__ __forceinline__ void round_fwd32(u32 (&v)[16], const Tw32 *__restrict__ tw, u32 T0, u32 p, u32 p2) {
#pragma unroll
    for (int i = I0; i < R; i++) {
        const int span = 8 >> i;
#pragma unroll
        for (int g = 0; g < (1 << i); g++) {
            const Tw32 t = tw[(T0 << i) + g];
#pragma unroll
            for (int l = 0; l < span; l++) {
                if constexpr (LOOSE) ct32_loose(v[g * 2 * span + l], v[g * 2 * span + l + span], t, p, p2);
                else ct32(v[g * 2 * span + l], v[g * 2 * span + l + span], t, p, p2);
            }
        }
    }
}
template <int R>
__device__ __forceinline__ void round_inv32(u32 (&v)[16], const Tw32 *__restrict__ tw, u32 T0, u32 p, u32 p2) {
#pragma unroll
    for (int i = R - 1; i >= 0; i--) {
        const int span = 8 >> i;
#pragma unroll
        for (int g = 0; g < (1 << i); g++) {
            const Tw32 t = tw[(T0 << i) + g];
#pragma unroll
            for (int l = 0; l < span; l++) gs32(v[g * 2 * span + l], v[g * 2 * span + l + span], t, p, p2);
        }
    }
}

// ---- inverse rounds WITHOUT a conditional subtraction per butterfly (round 4; moduli below 2^32 / 25 only) ----------------
// A Gentleman-Sande stage doubles the bound of its sums and resets its products to 2p.  With p < 2^32 / 25 a word holds 16p:
// the bounds of the 16 registers are followed at compile time (units of p, as pm_inv_sched does for the 64-bit rounds) and a
// value is reduced — barrett2p_32: any word -> [0, 2p), three instructions — only where a pair would pass 16p, plus what the
// round must hand on below BOUT.  From bound 4 to bound 4 that is 8 reductions per 32 butterflies where gs32 spends 32
// conditional subtractions.  x - y + K p needs K p >= y: K = the bound of y.
struct Inv32Sched {
    bool rx[4][8], ry[4][8];
    unsigned char ky[4][8];
    bool fin[16];
};
constexpr int kInv32Cap = 16;
constexpr Inv32Sched inv32_sched(int R, int bin, int bout) {
    Inv32Sched s{};
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
                if (bx + by > kInv32Cap) {
                    if (bx >= by) { rx = true; bx = 2; } else { ry = true; by = 2; }
                }
                if (bx + by > kInv32Cap) {
                    if (!rx) { rx = true; bx = 2; } else { ry = true; by = 2; }
                }
                s.rx[st][j] = rx;
                s.ry[st][j] = ry;
                s.ky[st][j] = (unsigned char)by;
                B[k] = bx + by;
                B[k2] = 2;
            }
    }
    for (int k = 0; k < 16; k++) s.fin[k] = B[k] > bout;
    return s;
}
// values below BIN p in, below BOUT p out (BIN, BOUT <= 16); STAGED: a scheduling barrier after every stage (round_inv32_staged)
template <int R, int BIN, int BOUT, bool STAGED>
__device__ __forceinline__ void round_inv32_loose(u32 (&v)[16], const Tw32 *__restrict__ tw, u32 T0, u32 p, u32 bq) {
    static_assert(BIN >= 2 && BIN <= kInv32Cap && BOUT >= 2 && BOUT <= kInv32Cap, "bounds in units of p");
    constexpr Inv32Sched S = inv32_sched(R, BIN, BOUT);
    int st = 0;
#pragma unroll
    for (int i = R - 1; i >= 0; i--, st++) {
        const int span = 8 >> i;
#pragma unroll
        for (int g = 0; g < (1 << i); g++) {
            const Tw32 t = tw[(T0 << i) + g];
#pragma unroll
            for (int l = 0; l < span; l++) {
                const int k = g * 2 * span + l, j = g * span + l;
                u32 x = v[k], y = v[k + span];
                if (S.rx[st][j]) x = barrett2p_32(x, p, bq);
                if (S.ry[st][j]) y = barrett2p_32(y, p, bq);
                const u32 d = x - y + (u32)S.ky[st][j] * p;
                v[k] = x + y;
                v[k + span] = mul_shoup32(d, t, p);
            }
        }
        if constexpr (STAGED) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (S.fin[k]) v[k] = barrett2p_32(v[k], p, bq);
}

// The inverse round with its table reads kept INSIDE their stage (a scheduling barrier between stages): the compiler
// otherwise hoists the 15 reads of a round to its top — 30 registers that kernels holding other results cannot spare.
template <int R>
__device__ __forceinline__ void round_inv32_staged(u32 (&v)[16], const Tw32 *__restrict__ tw, u32 T0, u32 p, u32 p2) {
#pragma unroll
    for (int i = R - 1; i >= 0; i--) {
        const int span = 8 >> i;
#pragma unroll
        for (int g = 0; g < (1 << i); g++) {
            const Tw32 t = tw[(T0 << i) + g];
#pragma unroll
            for (int l = 0; l < span; l++) gs32(v[g * 2 * span + l], v[g * 2 * span + l + span], t, p, p2);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same rounds with their twiddles in registers: load_tw32 BEFORE the LDS exchange that precedes the round, so that
// the latency of the table (L2 for the late rounds, whose twiddles are unique per thread) runs under the exchange's
// barriers instead of after them.  t[(1 << i) - 1 + g] = stage i, group g.
template <int R, int I0 = 0>
__device__ __forceinline__ void load_tw32(Tw32 (&t)[15], const Tw32 *__restrict__ tw, u32 T0) {
#pragma unroll
    for (int i = I0; i < R; i++)
#pragma unroll
        for (int g = 0; g < (1 << i); g++) t[(1 << i) - 1 + g] = tw[(T0 << i) + g];
}
template <int R, int I0 = 0, bool LOOSE = false>
__device__ __forceinline__ void round_fwd32_tw(u32 (&v)[16], const Tw32 (&t)[15], u32 p, u32 p2) {
#pragma unroll
    for (int i = I0; i < R; i++) {
        const int span = 8 >> i;
#pragma unroll
        for (int g = 0; g < (1 << i); g++) {
#pragma unroll
            for (int l = 0; l < span; l++) {
                if constexpr (LOOSE) ct32_loose(v[g * 2 * span + l], v[g * 2 * span + l + span], t[(1 << i) - 1 + g], p, p2);
                else ct32(v[g * 2 * span + l], v[g * 2 * span + l + span], t[(1 << i) - 1 + g], p, p2);
            }
        }
    }
}
template <int R>
__device__ __forceinline__ void round_inv32_tw(u32 (&v)[16], const Tw32 (&t)[15], u32 p, u32 p2) {
#pragma unroll
    for (int i = R - 1; i >= 0; i--) {
        const int span = 8 >> i;
#pragma unroll
        for (int g = 0; g < (1 << i); g++) {
#pragma unroll
            for (int l = 0; l < span; l++) gs32(v[g * 2 * span + l], v[g * 2 * span + l + span], t[(1 << i) - 1 + g], p, p2);
        }
    }
}

template <int LP, int AF, int AT, bool FIRST>
__device__ __forceinline__ void exchange32(u32 (&v)[16], u32 *lds, u32 w, u32 tf) {
    constexpr int M = 1 << LP;
    const u32 bf = pad16(w * M + field_of<AF>(tf, 0)), bt = pad16(w * M + field_of<AT>(tf, 0));   // + a constant per register (ntt_rounds.hpp: pad16_koff)
    if (!FIRST) __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) lds[bf + pad16_koff<AF>(k)] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = lds[bt + pad16_koff<AT>(k)];
}

// round 0 on BITS by table look-up (see ntt_rounds.hpp round0_bits; tables per prime, built on the host)
constexpr int kLut32Words = 136;
template <int R0, bool LOOSE = false>
__device__ __forceinline__ void round0_bits32(u32 (&v)[16], const u32 *lut, const Tw32 *__restrict__ gtw, u32 p, u32 p2) {
    if constexpr (R0 == 1) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u32 pt = v[k] + 2u * v[k + 8];
            v[k] = lut[128 + 2 * pt];
            v[k + 8] = lut[128 + 2 * pt + 1];
        }
    } else {
        u32 pt[4];
#pragma unroll
        for (int c = 0; c < 4; c++) pt[c] = v[c] + 2u * v[c + 4] + 4u * v[c + 8] + 8u * v[c + 12];
        if constexpr (R0 == 2) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint4 y = *reinterpret_cast<const uint4 *>(lut + 4 * pt[c]);
                v[c] = y.x; v[c + 4] = y.y; v[c + 8] = y.z; v[c + 12] = y.w;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const uint4 a = *reinterpret_cast<const uint4 *>(lut + 4 * pt[c]);
                const uint4 b = *reinterpret_cast<const uint4 *>(lut + 64 + 4 * pt[c + 2]);
                const u32 av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    v[c + 4 * j] = csub_u32(av[j] + bv[j], p);
                    v[c + 2 + 4 * j] = csub_u32(av[j] - bv[j] + p, p);
                }
            }
            if constexpr (R0 == 4) round_fwd32<4, 3, LOOSE>(v, gtw, 1u, p, p2);
        }
    }
}

// x mod p for x < 2^64, p < 2^30 (Barrett with mu = floor(2^64 / p)): canonical
__device__ __forceinline__ u32 reduce64_32(u64 x, u32 p, u64 mu) {
    const u64 qh = __umul64hi(x, mu);
    u32 r = (u32)(x - qh * p);          // in [0, 2p)
    return csub_u32(r, p);
}

template <int TH = 256>
__device__ __forceinline__ void stage_tw32(Tw32 *ltw, const Tw32 *__restrict__ tw, int count, u32 tid) {
    for (u32 i = tid; i < (u32)count; i += TH) ltw[i] = tw[i];   // s0 = blk = 0: the local table is the head of the global one
}

}  // namespace fhe
