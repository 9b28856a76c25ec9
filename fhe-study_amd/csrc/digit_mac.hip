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
// Output: out[b][p][c][n] — canonical partial sums, one per part; parts exist so that a small batch
// still fills the chip (630 ciphertexts x 4 parts = 2520 workgroups); the caller adds the parts
// (sum_parts_kernel below) before the inverse transforms.  HBM traffic per external product at
// n = 1024: 16 KiB in + 128 KiB of partial sums, against 2 MiB for the materialised transforms.
#include "digit_mac.hpp"
#include "ntt_rounds.hpp"

#include <cstdlib>

namespace fhe {

template <int LP>
struct DigitMacCfg {
    using C = ContigCfg<LP>;
    static constexpr int TH = 256;
    static constexpr int PPT = C::M / TH;               // coefficient positions a thread owns in the multiply phase
    static constexpr int CHUNK = 32;                    // terms between reductions of an accumulator
    static_assert(LP >= 8 && LP <= 12, "tile of 4096 coefficients, 256 threads");
    static_assert(C::TH == TH, "one workgroup shape");
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

// MW: waves per SIMD the register allocation must admit (2: up to 256 VGPRs, 3: 168).  Registers: 4 per
// accumulator + the 16 coefficients + the round's temporaries.
template <int LP, int SRC, int NC, int MW>
__global__ __launch_bounds__(256, MW) void digit_mac_kernel(DigitMacArgs a) {
    using C = ContigCfg<LP>;
    using K = DigitMacCfg<LP>;
    constexpr int PPT = K::PPT, W = C::W;
    static_assert(NC * PPT <= 32, "accumulators must fit the register file");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + C::DATA_BYTES);
    const u32 tid = threadIdx.x, w = tid / C::TPB, tf = tid % C::TPB;
    const u64 b = blockIdx.x / a.parts;
    const u32 part = blockIdx.x % a.parts;
    const u32 t_begin = part * a.tpp, t_end = min(a.T, t_begin + a.tpp);
    const Mod &m = a.mod;
    const u32 n = 1u << LP;
    stage_twiddles<C::LTW_N, C::TH>(ltw, a.tw, 0u, 0u, tid);
    u64 *llut = reinterpret_cast<u64 *>(ltw + C::LTW_N);          // digit tables: round 0 is look-ups (round0_bits)
    for (u32 i = tid; i < (u32)kDigitLutWords; i += C::TH) llut[i] = a.lut[i];
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
        fwd_rounds_contig<LP, true, true, 2, false, true>(v, lds, ltw, a.tw, 0u, 0u, w, tf, m, llut);
        // every thread rewrites exactly the slots it gathered in the last exchange: no barrier before
#pragma unroll
        for (int k = 0; k < 16; k++) lds[pad16(w * C::M + field_of<0>(tf, k))] = canon4(v[k], m);
        __syncthreads();
        // ---- multiply-accumulate: W transforms x NC key rows at this thread's PPT positions ----
        const u32 nu = min((u32)W, t_end - t0);
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

static inline unsigned dm_ew_grid(u64 count) {
    u64 g = (count + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    return (unsigned)(g ? g : 1);
}

template <int LP, int SRC, int NC>
static hipError_t launch_dm(const DigitMacArgs &a, hipStream_t st) {
    using C = ContigCfg<LP>;
    if constexpr (NC * DigitMacCfg<LP>::PPT > 32) {
        return hipErrorNotSupported;
    } else {
        const u64 grid = a.batch * a.parts;
        if (grid > 0x7fffffffull) return hipErrorInvalidValue;
        // 3 waves per SIMD up to 16 accumulators per thread (measured); FHE_DIGIT_MAC_WAVES=2|3 for A/B runs
        static const int waves_env = [] { const char *e = getenv("FHE_DIGIT_MAC_WAVES"); return e ? atoi(e) : 0; }();
        const int mw = waves_env == 2 || waves_env == 3 ? waves_env : (NC * DigitMacCfg<LP>::PPT <= 16 ? 3 : 2);
        KernelTimer kt(SRC == SRC_DIGITS ? "digit_mac_torus" : "digit_mac_zq", LP, st);
        if (mw == 3) {
            if (hipError_t e = allow_big_lds((const void *)digit_mac_kernel<LP, SRC, NC, 3>, C::LDS_BYTES_BITS)) return e;
            hipLaunchKernelGGL((digit_mac_kernel<LP, SRC, NC, 3>), dim3((unsigned)grid), dim3(256), C::LDS_BYTES_BITS, st, a);
        } else {
            if (hipError_t e = allow_big_lds((const void *)digit_mac_kernel<LP, SRC, NC, 2>, C::LDS_BYTES_BITS)) return e;
            hipLaunchKernelGGL((digit_mac_kernel<LP, SRC, NC, 2>), dim3((unsigned)grid), dim3(256), C::LDS_BYTES_BITS, st, a);
        }
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

uint32_t digit_mac_parts(u64 batch, uint32_t T, uint32_t log_n) {
    // enough workgroups to fill 256 CUs several times over, but at least two steps per workgroup
    const uint32_t W = log_n >= 12 ? 1u : 1u << (12 - log_n);
    uint32_t parts = 1;
    while (parts < 8 && batch * parts < 2048 && (T / (parts * 2)) >= 2 * W) parts *= 2;
    return parts;
}

hipError_t launch_digit_mac(const DevicePlan &p, int src_kind, const u64 *src, u64 ct_stride, uint32_t rows, uint32_t l,
                            const u64 *key, uint32_t nc, u64 *partial, uint32_t parts, u64 batch, hipStream_t st) {
    const int L = p.log_n;
    if (L < 8 || L > 12 || !p.wide || l == 0 || l > 64 || rows == 0 || parts == 0 || !p.digit_lut) return hipErrorNotSupported;
    if (src_kind == SRC_ZQBITS && p.mod.q < 3) return hipErrorNotSupported;
    // Beyond 16 accumulators per thread (n = 4096 with 2 output rows, n = 2048 with 4) the kernel holds 2 waves
    // per SIMD and measured slower than transform-then-accumulate (key switch, n = 4096: 0.60 vs 0.49 ms per 256):
    // those shapes keep the two-kernel form.  FHE_DIGIT_MAC_MAXACC overrides the limit for A/B runs.
    static const uint32_t max_acc = [] { const char *e = getenv("FHE_DIGIT_MAC_MAXACC"); return e ? (uint32_t)atoi(e) : 16u; }();
    if (nc * ((1u << L) / 256u) > max_acc) return hipErrorNotSupported;
    if (batch == 0) return hipSuccess;
    DigitMacArgs a{};
    a.src = src; a.key = key; a.out = partial; a.tw = p.tw_fwd; a.lut = p.digit_lut; a.mod = p.mod;
    a.batch = batch; a.ct_stride = ct_stride; a.l = l; a.T = rows * l; a.parts = parts;
    const uint32_t W = L >= 12 ? 1u : 1u << (12 - L);
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

hipError_t launch_sum_parts(const u64 *partial, u64 *out, u64 batch, uint32_t parts, u64 row_words, u64 q, hipStream_t st) {
    if (batch == 0 || row_words == 0) return hipSuccess;
    KernelTimer kt("sum_parts", 0, st);
    hipLaunchKernelGGL(sum_parts_kernel, dim3(dm_ew_grid(batch * row_words)), dim3(256), 0, st, partial, out, batch, parts, row_words, q);
    return hipGetLastError();
}

}  // namespace fhe
