// digit_mac.hip — gadget decomposition -> forward NTT -> multiply-accumulate in ONE kernel.
//
// What is computed (per ciphertext b, output row c):
//     S[b][c] = sum_{t < T} KEY[t][c] (.) NTT(digit_t(b))           (NTT domain, canonical)
// with t = r*l + d: digit d of source row r of the ciphertext.  This is the inner loop of
//   TGGSW x TGLWE   tfhe/src/tggsw.rs:45-62,139-149   Tn::decompose ring_torus.rs:67-77, torus.rs:43-52
//                   (T = (k+1)*l digits, KEY = the TGGSW rows split in 32-bit halves: 2(k+1) output rows)
//   GLWE::key_switch gfhe/src/glwe.rs:126-137          Rq::decompose ring_nq.rs:67-78, zq.rs:176-190
//                   (T = k*l digits of the mask rows, KEY = the key-switching key: k+1 output rows)
// The reference forms every one of the T*(k+1) products by schoolbook (torus) or with three
// transforms each (Rq) and sums coefficient vectors.  Round 1 of this engine transformed all digit
// polynomials into a buffer (1 MiB per external product at n = 1024) and read it back in a separate
// multiply-accumulate kernel.  Here the digit transforms never leave the chip:
//
//   a workgroup owns (ciphertext b, part p of its T digits); per step its W units each
//     - read their source row and extract one digit (a 0/1 polynomial) in registers,
//     - run the log2(n) forward stages in registers / LDS (the rounds of ntt_rounds.hpp),
//     - leave the canonical transform in the LDS tile;
//   then every thread multiplies the tile's W transforms, at the M/256 coefficient positions it owns,
//   with the key rows (L2-resident: the key is shared by the whole batch) into 128-bit
//   accumulators that live in registers across all steps and are reduced once per 32 terms
//   (q < 2^61: 32 products of canonical operands stay below 2^127).
//
// Since round 2 the shapes with k = 1 (TGGSW x TGLWE up to n = 1024, key switching up to n = 4096) run the same scheme on
// two 27-bit primes with 32-bit arithmetic (digit32.hip); this 61-bit form serves the others and FHE_EXT32=0.
//
// Output: out[b][p][c][n] — canonical partial sums, one per part; parts exist so that a small batch
// still fills the chip (630 ciphertexts x 4 parts = 2520 workgroups); the caller adds the parts
// (sum_parts_kernel below) before the inverse transforms.  HBM traffic per external product at
// n = 1024: 16 KiB in + 128 KiB of partial sums, against 2 MiB for the materialised transforms.
#include "digit_mac.hpp"
#include "ntt_rounds.hpp"

#include <cstdlib>

namespace fhe {

// Workgroup shape: 256 threads = W = 4096/n units (digit polynomials transformed side by side per step); in the
// multiply phase every thread owns PPT = n/256 coefficient positions of all NC output rows: NC * PPT accumulators of
// 128 bits (4 VGPRs each) live across the whole digit loop, next to the 16 coefficients and the temporaries of the
// transform rounds.  Measured (630 external products, n = 1024, NC = 4: 16 accumulators):
//   256 threads, 192 VGPRs, 2 waves per SIMD                                   548 us   <- this form
//   the same forced into 168 VGPRs (3 waves per SIMD, 21 registers spilled)     584 us
//   512 threads, 8 accumulators, 128 VGPRs (4 waves per SIMD, 25 spilled)       626 us
//   transform kernel + separate multiply-accumulate kernel (round 1's form)     349 + 356 us
// Ablation of this form (tools/abl_digit_mac.py): 553 us = 317 us of transforms + 225 us of multiply-accumulate (no overlap:
// both phases are VALU-bound; the 128-bit multiply-accumulate costs ~15 issue slots, ~60 cycles).  Requesting the key rows
// one unit ahead (two register buffers, 228 VGPRs) changed nothing (565 us): the phase is not bound by L2 latency.
// and with 32 accumulators (n = 4096, NC = 2: key switching) the fused kernel is SLOWER than the two-kernel form
// (0.52-0.56 vs 0.46 ms per 256 ciphertexts), so shapes beyond 16 accumulators keep the two kernels.
template <int LP, int NC>
struct DigitMacCfg {
    using C = ContigCfg<LP>;
    static constexpr int M = C::M, TPB = C::TPB;
    static constexpr int TH = 256;
    static constexpr int W = TH / TPB;                  // units (digit polynomials) per step
    static constexpr int PPT = M / TH;                  // coefficient positions a thread owns in the multiply phase
    static constexpr int ACC = NC * PPT;
    static constexpr int MAX_ACC = 16;
    static constexpr int CHUNK = 32;                    // terms between reductions of an accumulator
    static constexpr size_t DATA_BYTES = (size_t)(W * M + W * M / 16) * 8;      // the padded tile of pad16()
    static constexpr size_t LDS_BYTES = DATA_BYTES + (size_t)C::LTW_N * sizeof(Tw) + (size_t)kDigitLutWords * 8;
    static_assert(LP >= 8 && LP <= 12, "256 .. 4096 coefficients");
    static_assert(TH % TPB == 0 && M % TH == 0, "whole units, whole positions");
};

// digit d (0 = most significant) of a source word
template <int SRC>
__device__ __forceinline__ u64 digit_of(u64 x, u32 l, u32 d) {
    if (SRC == SRC_DIGITS) return (x >> (l - 1u - d)) & 1ull;                       // torus.rs:43-52, beta = 2
    // Zq::decompose_base2, zq.rs:176-190: every digit is 1 when the value is >= 2^l (with the
    // reference's `1 << l` taken modulo 64, as a --release build does)
    const u64 sat = 1ull << (l & 63u);
    return x >= sat ? 1ull : (x >> (l - 1u - d)) & 1ull;
}

template <int LP, int SRC, int NC>
__global__ __launch_bounds__((DigitMacCfg<LP, NC>::TH)) void digit_mac_kernel(DigitMacArgs a) {
    using C = ContigCfg<LP>;
    using K = DigitMacCfg<LP, NC>;
    constexpr int PPT = K::PPT, W = K::W, TH = K::TH;
    static_assert(K::ACC <= K::MAX_ACC, "accumulators must fit the register file");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + K::DATA_BYTES);
    const u32 tid = threadIdx.x, w = tid / C::TPB, tf = tid % C::TPB;
    const u64 b = blockIdx.x / a.parts;
    const u32 part = blockIdx.x % a.parts;
    const u32 t_begin = part * a.tpp, t_end = min(a.T, t_begin + a.tpp);
    const Mod &m = a.mod;
    const u32 n = 1u << LP;
    stage_twiddles<C::LTW_N, TH>(ltw, a.tw, 0u, 0u, tid);
    u64 *llut = reinterpret_cast<u64 *>(ltw + C::LTW_N);          // digit tables: round 0 is look-ups (round0_bits)
    for (u32 i = tid; i < (u32)kDigitLutWords; i += TH) llut[i] = a.lut[i];
    __syncthreads();
    const u64 *__restrict__ ct = a.src + b * a.ct_stride;

    MacAcc acc[NC][PPT];
    u32 pending = 0;                                               // terms since the last reduction
    const u32 j0 = tid * PPT;                                      // positions [j0, j0 + PPT) of every row
    for (u32 t0 = t_begin; t0 < t_end; t0 += W) {
        // ---- digit -> registers -> LP forward stages -> canonical transform in the LDS tile ----
        const u32 t = t0 + w;
        const bool live = t < t_end;
        const u32 tt = live ? t : t_begin;                          // idle units redo a valid digit, never multiplied
        const u32 r = tt / a.l, d = tt - r * a.l;
        const u64 *__restrict__ row = ct + (u64)r * n;
        u64 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = digit_of<SRC>(row[field_of<C::A0>(tf, k)], a.l, d);
        // FRESH = false: the tile was read by the previous step's multiply phase
#ifndef FHE_DM_ABLATE_NTT     // timing-only builds (tools/abl_digit_mac.py): the kernel without its transform / its multiply phase
        fwd_rounds_contig<LP, true, true, 2, false, true>(v, lds, ltw, a.tw, 0u, 0u, w, tf, m, llut);
#endif
        // every thread rewrites exactly the slots it gathered in the last exchange: no barrier before
#pragma unroll
        for (int k = 0; k < 16; k++) lds[pad16(w * C::M + field_of<0>(tf, k))] = canon4(v[k], m);
        __syncthreads();
        // ---- multiply-accumulate: W transforms x NC key rows at this thread's PPT positions ----
        const u32 nu = min((u32)W, t_end - t0);
#ifndef FHE_DM_ABLATE_MAC
#pragma unroll
        for (int u = 0; u < W; u++) {
            if ((u32)u < nu) {
                u64 x[PPT];
#pragma unroll
                for (int i = 0; i < PPT; i++) x[i] = lds[pad16(u * C::M + j0 + i)];
                const u64 *__restrict__ g = a.key + ((u64)(t0 + u) * NC) * n + j0;
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    u64 gv[PPT];
                    if constexpr (PPT >= 2) {
#pragma unroll
                        for (int i = 0; i < PPT; i += 2) {
                            const ulonglong2 q2 = *reinterpret_cast<const ulonglong2 *>(g + (u64)c * n + i);
                            gv[i] = q2.x;
                            gv[i + 1] = q2.y;
                        }
                    } else {
                        gv[0] = g[(u64)c * n];
                    }
#pragma unroll
                    for (int i = 0; i < PPT; i++) acc[c][i].mac(gv[i], x[i]);
                }
            }
        }
#else
        acc[0][0].mac(lds[pad16(j0)], 3);
#endif
        pending += nu;
        if (pending + W > (u32)K::CHUNK) {                           // the next step could overflow 2^128
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int i = 0; i < PPT; i++) acc[c][i].fold(m);
            pending = 1;                                            // the canonical carry-over counts as a term
        }
    }
    u64 *__restrict__ o = a.out + ((b * a.parts + part) * NC) * (u64)n + j0;
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int i = 0; i < PPT; i++) o[(u64)c * n + i] = acc[c][i].fold(m);
}

// out[b][c][j] = sum_p part[b][p][c][j]  mod q   (canonical in, canonical out)
__global__ __launch_bounds__(256) void sum_parts_kernel(const u64 *__restrict__ part, u64 *__restrict__ out, u64 batch,
                                                        u32 parts, u64 row_words, u64 q) {
    const u64 total = batch * row_words, stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const u64 b = i / row_words, cj = i - b * row_words;
        const u64 *__restrict__ p = part + b * parts * row_words + cj;
        u64 s = p[0];
        for (u32 k = 1; k < parts; k++) {
            s += p[(u64)k * row_words];
            s = s >= q ? s - q : s;
        }
        out[i] = s;
    }
}

// ---- the tail: add the parts, inverse transform, finish — one kernel ---------------------------------
// Row R = (ciphertext b, output row c) of the accumulated sums: S[b][c] = sum_p partial[b][p][c] (canonical
// adds in the load), the LP inverse stages with the n^-1 scaling folded in, then
//   EPI_KS     out[b][c] = (c < k ? 0 : glwe[b][c]) - S[b][c]  mod q      GLWE::key_switch's tail, glwe.rs:129-136
//   EPI_TORUS  out[b][c] = lift(S[b][c]) + (lift(S[b][k1 + c]) << 32)      the two 32-bit key halves recombined
//              mod 2^64, lift = the centred representative mod P1 (zr_combine32_kernel); the two rows meet
//              through the LDS tile, so a workgroup must hold whole ciphertexts: W % nc == 0.
// Replaces sum_parts_kernel + ntt_inv_contig_kernel + ks_tail_kernel / zr_combine32_kernel.
enum : int { EPI_KS = 0, EPI_TORUS = 1 };

template <int LP, int EPI>
__global__ __launch_bounds__(256) void digit_tail_kernel(DigitTailArgs a) {
    using C = ContigCfg<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + C::DATA_BYTES);
    const u32 tid = threadIdx.x, w = tid / C::TPB, tf = tid % C::TPB;
    const u64 rows = a.batch * a.nc;
    const u64 R0 = (u64)blockIdx.x * C::W;
    const u32 live = (u32)min((u64)C::W, rows - R0);
    const bool active = w < live;
    const u64 R = R0 + (active ? w : 0u);                       // idle units redo the first row and store nothing
    const u64 b = R / a.nc;
    const u32 c = (u32)(R - b * a.nc);
    const u32 n = 1u << LP;
    const Mod &m = a.mod;
    stage_twiddles<C::LTW_N, C::TH>(ltw, a.tw, 0u, 0u, tid);

    // window [0,4): 16 consecutive NTT-domain values per thread, summed over the parts
    u64 v[16];
    {
        const u64 *__restrict__ src = a.partial + ((b * a.parts) * a.nc + c) * (u64)n + tf * 16u;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(src + 2 * j);
            v[2 * j] = x.x;
            v[2 * j + 1] = x.y;
        }
        for (u32 p = 1; p < a.parts; p++) {
            const u64 *__restrict__ sp = src + (u64)p * a.nc * n;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(sp + 2 * j);
                u64 s0 = v[2 * j] + x.x, s1 = v[2 * j + 1] + x.y;
                v[2 * j] = s0 >= m.q ? s0 - m.q : s0;
                v[2 * j + 1] = s1 >= m.q ? s1 - m.q : s1;
            }
        }
    }
    __syncthreads();                                             // the twiddle tile
    inv_rounds_contig<LP, true, true, true>(v, lds, ltw, a.tw, 0u, 0u, w, tf, m, a.ninv, a.s_ninv);
    if constexpr (EPI == EPI_KS) {
        if (active) {
            const u64 base = (b * a.nc + c) * (u64)n;
            const bool body = c >= a.k;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const u32 pos = field_of<C::A0>(tf, k);
                const u64 y = canon2(v[k], m);
                const u64 x = body ? a.src[base + pos] : 0ull;
                a.out[base + pos] = x >= y ? x - y : x + m.q - y;
            }
        }
    } else {
        // natural positions into the tile (the last exchange was gathered from it by every thread: barrier first)
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) lds[pad16(w * C::M + field_of<C::A0>(tf, k))] = canon2(v[k], m);
        __syncthreads();
        const u32 k1 = a.nc / 2;
        const u32 cts = live / a.nc;                              // whole ciphertexts in this workgroup
        const u64 b0 = R0 / a.nc;
        for (u32 e = tid; e < cts * k1 * n; e += 256) {
            const u32 j = e & (n - 1), cc = (e >> LP) % k1, bb = (e >> LP) / k1;
            u64 lo = lds[pad16((bb * a.nc + cc) * C::M + j)], hi = lds[pad16((bb * a.nc + k1 + cc) * C::M + j)];
            if (lo >= a.half1) lo -= m.q;                          // centred lift, two's complement
            if (hi >= a.half1) hi -= m.q;
            a.out[((b0 + bb) * k1 + cc) * (u64)n + j] = lo + (hi << 32);
        }
    }
}

static inline unsigned dm_ew_grid(u64 count) {
    u64 g = (count + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    return (unsigned)(g ? g : 1);
}

template <int LP, int SRC, int NC>
static hipError_t launch_dm(const DigitMacArgs &a, hipStream_t st) {
    using K = DigitMacCfg<LP, NC>;
    if constexpr (K::ACC > K::MAX_ACC) {
        return hipErrorNotSupported;
    } else {
        const u64 grid = a.batch * a.parts;
        if (grid > 0x7fffffffull) return hipErrorInvalidValue;
        if (hipError_t e = allow_big_lds((const void *)digit_mac_kernel<LP, SRC, NC>, K::LDS_BYTES)) return e;
        KernelTimer kt(SRC == SRC_DIGITS ? "digit_mac_torus" : "digit_mac_zq", LP, st);
        hipLaunchKernelGGL((digit_mac_kernel<LP, SRC, NC>), dim3((unsigned)grid), dim3(K::TH), K::LDS_BYTES, st, a);
        return hipGetLastError();
    }
}

template <int SRC, int NC>
static hipError_t launch_dm_lp(int lp, const DigitMacArgs &a, hipStream_t st) {
    switch (lp) {
        case 8: return launch_dm<8, SRC, NC>(a, st);
        case 9: return launch_dm<9, SRC, NC>(a, st);
        case 10: return launch_dm<10, SRC, NC>(a, st);
        case 11: return launch_dm<11, SRC, NC>(a, st);
        case 12: return launch_dm<12, SRC, NC>(a, st);
    }
    return hipErrorNotSupported;
}

// units per step of the kernel that (log_n, nc) selects; 0: no fused kernel for this shape
static uint32_t dm_units(uint32_t log_n, uint32_t nc) {
    if (log_n < 8 || log_n > 12 || nc < 2 || nc > 4) return 0;
    const uint32_t M = 1u << log_n;
    if (nc * (M / 256) > 16) return 0;
    return 4096 / M;
}

uint32_t digit_mac_parts(u64 batch, uint32_t T, uint32_t log_n, uint32_t nc) {
    // enough workgroups to fill 256 CUs several times over, but at least two steps per workgroup
    const uint32_t W = dm_units(log_n, nc);
    if (!W) return 1;
    uint32_t parts = 1;
    while (parts < 8 && batch * parts < 2048 && (T / (parts * 2)) >= 2 * W) parts *= 2;
    return parts;
}

hipError_t launch_digit_mac(const DevicePlan &p, int src_kind, const u64 *src, u64 ct_stride, uint32_t rows, uint32_t l,
                            const u64 *key, uint32_t nc, u64 *partial, uint32_t parts, u64 batch, hipStream_t st) {
    const int L = p.log_n;
    if (L < 8 || L > 12 || !p.wide || l == 0 || l > 64 || rows == 0 || parts == 0 || !p.digit_lut) return hipErrorNotSupported;
    if (src_kind == SRC_ZQBITS && p.mod.q < 3) return hipErrorNotSupported;
    const uint32_t W = dm_units((uint32_t)L, nc);
    if (!W) return hipErrorNotSupported;
    if (batch == 0) return hipSuccess;
    DigitMacArgs a{};
    a.src = src; a.key = key; a.out = partial; a.tw = p.tw_fwd; a.lut = p.digit_lut; a.mod = p.mod;
    a.batch = batch; a.ct_stride = ct_stride; a.l = l; a.T = rows * l; a.parts = parts;
    a.tpp = ((a.T + parts - 1) / parts + W - 1) / W * W;           // whole steps per part
    if (src_kind == SRC_DIGITS) {
        if (nc == 2) return launch_dm_lp<SRC_DIGITS, 2>(L, a, st);
        if (nc == 4) return launch_dm_lp<SRC_DIGITS, 4>(L, a, st);
    } else if (src_kind == SRC_ZQBITS) {
        if (nc == 2) return launch_dm_lp<SRC_ZQBITS, 2>(L, a, st);
        if (nc == 3) return launch_dm_lp<SRC_ZQBITS, 3>(L, a, st);
    }
    return hipErrorNotSupported;
}

template <int EPI>
static hipError_t launch_tail_lp(int lp, const DigitTailArgs &a, hipStream_t st) {
    const u64 rows = a.batch * a.nc;
    switch (lp) {
#define X(LP_)                                                                                              \
    case LP_: {                                                                                             \
        using C = ContigCfg<LP_>;                                                                           \
        if (EPI == EPI_TORUS && (C::W % a.nc) != 0) return hipErrorNotSupported;                            \
        const u64 grid = (rows + C::W - 1) / C::W;                                                          \
        if (grid > 0x7fffffffull) return hipErrorInvalidValue;                                              \
        if (hipError_t e = allow_big_lds((const void *)digit_tail_kernel<LP_, EPI>, C::LDS_BYTES)) return e; \
        KernelTimer kt(EPI == EPI_KS ? "digit_tail_ks" : "digit_tail_torus", LP_, st);                       \
        hipLaunchKernelGGL((digit_tail_kernel<LP_, EPI>), dim3((unsigned)grid), dim3(256), C::LDS_BYTES, st, a); \
        return hipGetLastError();                                                                           \
    }
        X(8) X(9) X(10) X(11) X(12)
#undef X
    }
    return hipErrorNotSupported;
}

hipError_t launch_digit_tail_ks(const DevicePlan &p, const u64 *partial, uint32_t parts, uint32_t k, const u64 *glwe, u64 *out,
                                u64 batch, hipStream_t st) {
    if (p.log_n < 8 || p.log_n > 12 || !p.wide) return hipErrorNotSupported;
    if (batch == 0) return hipSuccess;
    DigitTailArgs a{};
    a.partial = partial; a.src = glwe; a.out = out; a.tw = p.tw_inv; a.mod = p.mod; a.ninv = p.ninv; a.s_ninv = p.s_ninv;
    a.batch = batch; a.parts = parts; a.nc = k + 1; a.k = k;
    return launch_tail_lp<EPI_KS>(p.log_n, a, st);
}

hipError_t launch_digit_tail_torus(const DevicePlan &p, const u64 *partial, uint32_t parts, uint32_t k1, u64 half1, u64 *out,
                                   u64 batch, hipStream_t st) {
    if (p.log_n < 8 || p.log_n > 12 || !p.wide) return hipErrorNotSupported;
    if (batch == 0) return hipSuccess;
    DigitTailArgs a{};
    a.partial = partial; a.out = out; a.tw = p.tw_inv; a.mod = p.mod; a.ninv = p.ninv; a.s_ninv = p.s_ninv;
    a.batch = batch; a.parts = parts; a.nc = 2 * k1; a.k = k1; a.half1 = half1;
    return launch_tail_lp<EPI_TORUS>(p.log_n, a, st);
}

hipError_t launch_sum_parts(const u64 *partial, u64 *out, u64 batch, uint32_t parts, u64 row_words, u64 q, hipStream_t st) {
    if (batch == 0 || row_words == 0) return hipSuccess;
    KernelTimer kt("sum_parts", 0, st);
    hipLaunchKernelGGL(sum_parts_kernel, dim3(dm_ew_grid(batch * row_words)), dim3(256), 0, st, partial, out, batch, parts, row_words, q);
    return hipGetLastError();
}

// timing-only builds (tools/abl_digit_mac.py) produce wrong words by design: fhe_ntt_version() says so (capi.hip)
bool digit_mac_ablated() {
#if defined(FHE_DM_ABLATE_NTT) || defined(FHE_DM_ABLATE_MAC) || defined(FHE_ABLATE_NO_BUTTERFLIES)
    return true;
#else
    return false;
#endif
}

}  // namespace fhe
