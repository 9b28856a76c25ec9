// ntt_kernels.hip — batched negacyclic NTT / iNTT / pointwise kernels for gfx950.
//
// What is computed (bit-exact with the reference):
//   forward  NTT::ntt   arith/src/ntt.rs:44-73   CT, natural in -> bit-reversed out
//   inverse  NTT::intt  arith/src/ntt.rs:78-110  GS, bit-reversed in -> natural out, * n^-1
//   pointwise           arith/src/ring_nq.rs:601-604
// The reference walks one stage at a time over the whole polynomial, one u128 %
// per butterfly.  Here the log2(n) stages are grouped into ROUNDS of up to four
// stages that run entirely in registers (16 coefficients per thread); between
// rounds the workgroup transposes through LDS; a polynomial that does not fit
// one workgroup's LDS tile (n >= 2^14) is split into a STRIDED pass (the stages
// with t >= 2^LB, columns of the n/2^LB x 2^LB view) and a CONTIGUOUS pass (the
// last LB stages on 2^LB-coefficient blocks).  Every global access is a
// coalesced slab: >=128 B runs per 16 lanes.
//
// Index algebra shared by all kernels (L = log2 n, stage s = 0..L-1 acts on bit
// L-1-s of the coefficient index j, twiddle = roots[2^s + (j >> (L-s))], exactly
// ntt.rs:54 `roots_of_unity[m + i]` with m = 2^s, i = j / 2t):
//   a pass covers LP consecutive stages, i.e. an LP-bit FIELD f of j; the bits of
//   j above the field are `blk`, with s0 of them; local stage ls = s - s0;
//   twiddle index = (1 << (s0+ls)) + (blk << ls) + (f >> (LP-ls)).
//   A round holds field bits [a, a+4) in registers (k = those 4 bits) and runs
//   the stages for bits a+3, a+2, ... ; with H = f >> (a+4) the index of stage i
//   of the round is (T0 << i) + (k >> (4-i)),  T0 = (1<<(s0+ls0)) + (blk<<ls0) + H.
#include "ntt_rounds.hpp"

#include <cstdlib>

// A/B switches of round 4 (defaults = what was measured faster on one box; tools/abl_build.sh builds the others)
#ifndef FHE_MID_ONE_TILE
#define FHE_MID_ONE_TILE 1        // rq_mul_mid_kernel: one twiddle tile that changes hands (4 workgroups per CU) / two tiles (3)
#endif
#ifndef FHE_MID_EARLY_FETCH
#define FHE_MID_EARLY_FETCH 1     // rq_mul_mid_kernel: both operands' loads issued up front
#endif
#ifndef FHE_INV_TLOAD
#define FHE_INV_TLOAD 1           // ntt_inv_contig_kernel (first pass of a two-pass inverse): whole-line loads transposed through LDS
#endif

namespace fhe {
// SRC_DIGITS (single-pass sizes only): the input is `batch / digit_l` rows of 64-bit words and
// output polynomial p is the transform of bit digit_l-1-(p % digit_l) of row p / digit_l — the
// gadget decomposition of ring_torus.rs:67-77 / torus.rs:43-52 done in the load, so the 0/1
// polynomials never exist in memory.
// SRC_REDUCE (single-pass sizes only; the two-pass sizes do it in the strided pass): the input
// rows are 2^src_log_n arbitrary 64-bit words, reduced mod q and zero-padded to n in the load.
// AR: 0 = q < 2^62 (Harvey [0,4q)), 1 = q < 2^61 (Shoup, compile-time bounds), 2 = pseudo-Mersenne q (zq_device.hpp:
// five-multiply butterflies; `a.tw` then holds {w, w 2^32 mod q}; SRC_PLAIN only), 4 = q = 1 (mod 2^32) below 2^61 (word
// Montgomery, `a.tw` = {w 2^32, w 2^64 mod q}; forward and inverse transforms and both Rq products; SRC_PLAIN only).
template <int LP, bool FINAL, int AR, int SRC = SRC_PLAIN>
__global__ __launch_bounds__(ContigCfg<LP>::TH) void ntt_fwd_contig_kernel(PassArgs a) {
    constexpr int WIDE = AR == 1 ? 1 : AR == 3 ? kStrict : 0;
    static_assert((AR != 2 && AR != 3 && AR != 4) || SRC == SRC_PLAIN, "transforming loads run on the Shoup tables, below 2^62");
    using C = ContigCfg<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const u32 tid = threadIdx.x, w = tid / C::TPB, tf = tid % C::TPB;
    const u32 s0 = a.log_n - LP;
    const u32 blk = blockIdx.x & ((1u << s0) - 1u);
    const u64 pg = (u64)(blockIdx.x >> s0);
    const u64 n = 1ull << a.log_n;
    // polynomials of this group that exist (the last group of a batch may be ragged); lanes of
    // a missing polynomial transform a copy of the group's first one and store nothing
    const u32 live = (u32)min((u64)C::W, a.batch - pg * C::W);
    // wave-uniform 64-bit base + 32-bit per-lane byte offsets (W*n*8 < 2^32)
    const u64 ubase = pg * C::W * n + (u64)blk * C::M;
    const u64 *__restrict__ pin = a.in + ubase;
    u64 *__restrict__ pout = a.out + ubase;
    const u32 off = ((w < live ? w : 0u) << a.log_n) * 8u;
    const Mod &m = a.mod;
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + C::DATA_BYTES);

    u64 v[16];
    if constexpr (SRC == SRC_DIGITS) {
        static_assert(FINAL, "digit loads exist for the single-pass kernels only");
        const u64 p = pg * C::W + (w < live ? w : 0u);
        const u64 row = p / a.digit_l;
        const u32 sh = a.digit_l - 1u - (u32)(p - row * a.digit_l);
        const u64 *__restrict__ src = a.in + row * n;
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = (src[field_of<C::A0>(tf, k)] >> sh) & 1ull;
    } else if constexpr (SRC == SRC_ZQBITS) {
        // Rq::decompose(2, l) in the load (ring_nq.rs:67-78, zq.rs:176-190): digit d of a coefficient is
        // bit l-1-d, or 1 for every d when the value is >= 2^l (with the reference's `1 << l` taken
        // modulo 64 as a --release build does)
        static_assert(FINAL, "digit loads exist for the single-pass kernels only");
        const u64 p = pg * C::W + (w < live ? w : 0u);
        const u64 row = p / a.digit_l;
        const u32 d = (u32)(p - row * a.digit_l);
        const u64 grp_i = row / a.src_grp;
        const u64 *__restrict__ src = a.in + grp_i * a.src_gstride + (row - grp_i * a.src_grp) * n;
        const u64 sat = 1ull << (a.digit_l & 63u);
        const u32 sh = a.digit_l - 1u - d;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const u64 x = src[field_of<C::A0>(tf, k)];
            v[k] = x >= sat ? 1ull : (x >> sh) & 1ull;        // q >= 3: both are canonical
        }
    } else if constexpr (SRC == SRC_REDUCE) {
        static_assert(FINAL, "reducing loads exist for the single-pass kernels only");
        const u64 p = pg * C::W + (w < live ? w : 0u);
        const u64 *__restrict__ src = a.in + (p << a.src_log_n);
        const u32 nsrc = 1u << a.src_log_n;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const u32 f = field_of<C::A0>(tf, k);
            v[k] = f < nsrc ? reduce_any(src[f], a.mod) : 0ull;
        }
    } else {
        ld16<true>(v, pin, C::A0, off + tf * 8u);   // field_of<A0>(tf, k) = (k << A0) + tf: the k-term on the scalar base
    }
    // Round 0's twiddles are the same for the whole workgroup (H = 0): read from the global
    // table at a wave-uniform address (scalar loads, SGPR operands).  The LDS copy is only
    // needed from round 1 on, so the barrier of the first exchange also publishes it.
    stage_twiddles<C::LTW_N, C::TH>(ltw, a.tw, s0, blk, tid);

    if constexpr (SRC == SRC_DIGITS || SRC == SRC_ZQBITS) {
        // inputs are bits: round 0 is table look-ups (round0_bits); the tables go to LDS first
        u64 *llut = reinterpret_cast<u64 *>(ltw + C::LTW_N);
        for (u32 i = tid; i < (u32)kDigitLutWords; i += C::TH) llut[i] = a.lut[i];
        __syncthreads();
        fwd_rounds_contig<LP, WIDE, FINAL, 2, true, true>(v, lds, ltw, a.tw, s0, blk, w, tf, m, llut);
    } else if constexpr (AR == 2 || AR == 4) {
        fwd_rounds_contig_pm<LP, kPmPassBound, true, AR>(v, lds, ltw, a.tw, s0, blk, w, tf, m);
    } else {
        fwd_rounds_contig<LP, WIDE, FINAL, kPassBound, true>(v, lds, ltw, a.tw, s0, blk, w, tf, m);
    }
    // transpose through LDS so the store is one contiguous slab per wave.  (Storing the 128
    // contiguous bytes a thread owns after the last round as 8 x 16 B straight from registers
    // was measured slower for the forward kernel: 4.91 ms vs 4.40 ms per 16384 polynomials;
    // the mirrored direct 16-byte LOADS of the inverse kernel are faster: 4.69 vs 5.79 ms.)
    constexpr int ALAST = C::a_of(C::NR - 1);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u64 x = !FINAL || AR == 3 ? v[k] : AR == 2 ? pm_canon(v[k], m) : AR == 4 ? canon8(v[k], m) : canon4(v[k], m);   // FINAL: < 4q in both Shoup modes, < 7q from the Montgomery rounds; AR = 3: canonical throughout
        lds[pad16(w * C::M + field_of<ALAST>(tf, k))] = x;
    }
    __syncthreads();
    const SlabIo<LP> io(tid, a.log_n);
    if (live == (u32)C::W) {   // wave-uniform: every polynomial of the group exists — sixteen LDS reads in flight, no per-store test
        u64 t[16];
#pragma unroll
        for (int i = 0; i < 16; i++) t[i] = lds[io.slot + i * io.lds_step];
#pragma unroll
        for (int i = 0; i < 16; i++) st_s<true>(io.base(pout, i, a.log_n), io.boff, t[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (io.wu(i) < live) st_s<true>(io.base(pout, i, a.log_n), io.boff, lds[io.slot + i * io.lds_step]);
    }
}

// MUL_IN: the input is the pointwise product in .* in2 (fused
// zip_eq(l,r).map(l*r), ring_nq.rs:601-604); if a.out2 != nullptr the product
// (the `evals` of the result, ring_nq.rs:606) is also written there.
template <int LP, bool FINAL, bool MUL_IN, int AR>
__global__ __launch_bounds__(ContigCfg<LP>::TH) void ntt_inv_contig_kernel(PassArgs a) {
    constexpr int WIDE = AR == 1 ? 1 : AR == 3 ? kStrict : 0;
    static_assert(!(AR == 4 && MUL_IN), "a variable x variable product has no Montgomery table: the fused product runs on the Shoup tables");
    using C = ContigCfg<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const u32 tid = threadIdx.x, w = tid / C::TPB, tf = tid % C::TPB;
    const u32 s0 = a.log_n - LP;
    const u32 blk = blockIdx.x & ((1u << s0) - 1u);
    const u64 pg = (u64)(blockIdx.x >> s0);
    const u64 n = 1ull << a.log_n;
    const u32 live = (u32)min((u64)C::W, a.batch - pg * C::W);   // see the forward kernel
    const bool active = w < live;
    const u64 ubase = pg * C::W * n + (u64)blk * C::M;
    const u32 off = ((active ? w : 0u) << a.log_n) * 8u;
    u64 *__restrict__ pout = a.out + ubase;
    const Mod &m = a.mod;
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + C::DATA_BYTES);
    stage_twiddles<C::LTW_N, C::TH>(ltw, a.tw, s0, blk, tid);  // published by the barrier below

    // first window = field bits [0,4): a thread's 16 coefficients are 128 contiguous bytes.
    constexpr int ALAST = C::a_of(C::NR - 1);
    static_assert(ALAST == 0, "first inverse window is the low 4 bits");
    // TLOAD (round 4; the first pass of a two-pass inverse on the pseudo-Mersenne tables, which runs at the memory system's
    // rate): the tile is loaded the way the forward kernel stores it — element e = i TH + tid, a wave instruction = 512
    // contiguous bytes, whole lines, so the non-temporal hint applies — and transposed through LDS into the register window;
    // the pointwise product of a fused multiply is formed (and its evals stored) in that layout, position by position.
    // Otherwise: 8 x 16 B per lane straight into registers (a line arrives in pieces; the cache merges them).
    constexpr bool TLOAD = FHE_INV_TLOAD && !FINAL && (AR == 2 || AR == 4);
    u64 v[16];
    if constexpr (TLOAD) {
        const u64 *__restrict__ pin = a.in + ubase;
        const SlabIo<LP> io(tid, a.log_n);
        const bool full = live == (u32)C::W;                     // wave-uniform: every polynomial of the group exists
        // one element of the slab: loaded (multiplied, its product stored) and put into its LDS slot
        auto element = [&](int i, u64 pi, u64 pi2, u64 po2, u32 o, bool exists) {   // scalar addresses + one lane offset
            u64 x = ld_s<true>(pi, o);
            if constexpr (MUL_IN && AR == 2) {
                const u64 y = ld_s<true>(pi2, o);
                x = mul_var_pm(x, y, a.mod);                     // both canonical: five multiplies
                if (a.out2) {
                    x = pm_canon(x, a.mod);                      // canonical only if the product is an output
                    if (exists) st_s<true>(po2, o, x);
                }
            }
            lds[io.slot + i * io.lds_step] = x;
        };
        if (full) {                                              // the i-term on the scalar base, one lane offset
#pragma unroll
            for (int i = 0; i < 16; i++)
                element(i, io.base(pin, i, a.log_n), io.base(a.in2 + ubase, i, a.log_n), io.base(a.out2 + ubase, i, a.log_n), io.boff, true);
        } else {                                                 // ragged last group: a missing polynomial's lanes read the group's first one
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const u32 e = i * C::TH + tid, wu = e >> LP, f = e & (C::M - 1);
                const u32 o = (((wu < live ? wu : 0u) << a.log_n) + f) * 8u;
                element(i, scalar_addr(pin), scalar_addr(a.in2 + ubase), scalar_addr(a.out2 + ubase), o, wu < live);
            }
        }
        __syncthreads();                                         // (also publishes the twiddle tile)
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = lds[pad16(w * C::M + field_of<ALAST>(tf, k))];
    } else {
        const u32 g0 = off + tf * 128u;
        const u64 *__restrict__ pin = a.in + ubase;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const ulonglong2 x = ld_c<ulonglong2>(pin, g0 + j * 16u);
            v[2 * j] = x.x;
            v[2 * j + 1] = x.y;
        }
        if constexpr (MUL_IN) {
            const u64 *__restrict__ pin2 = a.in2 + ubase;
            u64 *__restrict__ pout2 = a.out2 + ubase;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const ulonglong2 y = ld_c<ulonglong2>(pin2, g0 + j * 16u);
                ulonglong2 p;
                if constexpr (AR == 2) {   // both canonical: five multiplies; canonical only if the product is an output
                    p.x = mul_var_pm(v[2 * j], y.x, a.mod);
                    p.y = mul_var_pm(v[2 * j + 1], y.y, a.mod);
                    if (a.out2) { p.x = pm_canon(p.x, a.mod); p.y = pm_canon(p.y, a.mod); }
                } else if constexpr (AR == 3) {
                    p.x = mul_mod_var63(v[2 * j], y.x, a.mod);
                    p.y = mul_mod_var63(v[2 * j + 1], y.y, a.mod);
                } else {
                    p.x = mul_mod_var(v[2 * j], y.x, a.mod);
                    p.y = mul_mod_var(v[2 * j + 1], y.y, a.mod);
                }
                v[2 * j] = p.x;
                v[2 * j + 1] = p.y;
                if (a.out2 && active) st_c<ulonglong2>(pout2, g0 + j * 16u, p);
            }
        }
        __syncthreads();
    }

    // inputs are canonical (evals, or their product); a non-FINAL pass hands values below 4q (WIDE) / 2q on
    if constexpr (AR == 2 || AR == 4) inv_rounds_contig_pm<LP, FINAL, !TLOAD, (MUL_IN ? kPmMul : kPmOne), AR>(v, lds, ltw, a.tw, s0, blk, w, tf, m, a.ninv, a.s_ninv);
    else inv_rounds_contig<LP, WIDE, FINAL, true>(v, lds, ltw, a.tw, s0, blk, w, tf, m, a.ninv, a.s_ninv);

    if (active) {
        if constexpr (FINAL) {
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = AR == 3 ? v[k] : AR == 2 ? pm_canon(v[k], m) : AR == 4 ? canon4(v[k], m) : canon2(v[k], m);   // AR == 4: products, below 3q
        }
        st16<false>(pout, C::A0, off + tf * 8u, v);   // field_of<A0>(tf, k) = (k << A0) + tf
    }
}

// ---------------------------------------------------------------------------
// FUSED PRODUCT for single-pass sizes: c = intt(ntt(a) .* ntt(b)) (ring_nq.rs:586-607) with the
// polynomial resident in registers / LDS from the first load to the last store — one launch
// instead of three and 3 (+ evals) instead of 7 passes over memory.  The forward rounds end in the
// register window (field bits [0,4)) the inverse rounds start from, so nothing is rearranged
// between the transforms.  Operands flagged as evals skip their forward transform.
// ---------------------------------------------------------------------------
template <int LP, int WIDE, bool TILE_FRESH>
__device__ __forceinline__ void fwd_rounds_single(u64 (&v)[16], u64 *lds, const Tw *ltw, const Tw *gtw, u32 w,
                                                  u32 tf, const Mod &m) {
    using C = ContigCfg<LP>;
    auto TW = [&](bool lds_round) -> const Tw * { return lds_round ? ltw : gtw; };
    constexpr int B0 = 2, B1 = fwd_bound_out(C::R0, B0), B2 = fwd_bound_out(4, B1), B3 = fwd_bound_out(4, B2);   // canonical inputs
    round_fwd<C::R0, WIDE, B0, C::NR == 1>(v, gtw, 1u, m);
    if constexpr (C::NR > 1) {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        exchange_contig<LP, C::A0, A, TILE_FRESH>(v, lds, w, tf);
        round_fwd<4, WIDE, B1, C::NR == 2>(v, TW(C::in_lds(1)), (1u << LS) + (tf >> A), m);
    }
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        exchange_contig<LP, C::a_of(1), A, false>(v, lds, w, tf);
        round_fwd<4, WIDE, B2, C::NR == 3>(v, TW(C::in_lds(2)), (1u << LS) + (tf >> A), m);
    }
    if constexpr (C::NR > 3) {
        constexpr int A = C::a_of(3), LS = C::ls0_of(3);
        exchange_contig<LP, C::a_of(2), A, false>(v, lds, w, tf);
        round_fwd<4, WIDE, B3, C::NR == 4>(v, TW(C::in_lds(3)), (1u << LS) + (tf >> A), m);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = WIDE == kStrict ? v[k] : canon4(v[k], m);   // the last stage left x', y' < 4q (strict: canonical already)
}

template <int LP, int WIDE>
__device__ __forceinline__ void inv_rounds_single(u64 (&v)[16], u64 *lds, const Tw *ltw, const Tw *gtw, u32 w,
                                                  u32 tf, const Mod &m, const Tw ninv, const Tw s_ninv) {
    using C = ContigCfg<LP>;
    auto TW = [&](bool lds_round) -> const Tw * { return lds_round ? ltw : gtw; };
    constexpr int BF = 2, BN = 4;   // canonical inputs for the first round that runs, 4 after a round
    // the tile was last gathered by a forward exchange: every scatter here is preceded by a barrier
    if constexpr (C::NR > 3) {
        constexpr int A = C::a_of(3), LS = C::ls0_of(3);
        round_inv_sel<4, false, WIDE, BF>(v, TW(C::in_lds(3)), (1u << LS) + (tf >> A), m, ninv, s_ninv);
        exchange_contig<LP, A, C::a_of(2), false>(v, lds, w, tf);
    }
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        round_inv_sel<4, false, WIDE, (C::NR == 3 ? BF : BN)>(v, TW(C::in_lds(2)), (1u << LS) + (tf >> A), m, ninv, s_ninv);
        exchange_contig<LP, A, C::a_of(1), false>(v, lds, w, tf);
    }
    if constexpr (C::NR > 1) {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        round_inv_sel<4, false, WIDE, (C::NR == 2 ? BF : BN)>(v, TW(C::in_lds(1)), (1u << LS) + (tf >> A), m, ninv, s_ninv);
        exchange_contig<LP, A, C::A0, false>(v, lds, w, tf);
    }
    round_inv_sel<C::R0, true, WIDE, (C::NR == 1 ? BF : BN)>(v, TW(C::in_lds(0)), 1u, m, ninv, s_ninv);
}

template <int LP, int AR>
__global__ __launch_bounds__(ContigCfg<LP>::TH) void rq_mul_fused_kernel(PassArgs a) {
    constexpr int WIDE = AR == 1 ? 1 : AR == 3 ? kStrict : 0;
    using C = ContigCfg<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    Tw *ltw_f = reinterpret_cast<Tw *>(smem_raw + C::DATA_BYTES);
    Tw *ltw_i = ltw_f + C::LTW_N;
    const u32 tid = threadIdx.x, w = tid / C::TPB, tf = tid % C::TPB;
    const u64 pg = (u64)blockIdx.x;
    const u32 live = (u32)min((u64)C::W, a.batch - pg * C::W);
    const bool active = w < live;
    const u64 ubase = pg * C::W * (u64)C::M;                 // single pass: n = M
    const u32 off = ((active ? w : 0u) << LP) * 8u;
    const Mod &m = a.mod;
    stage_twiddles<C::LTW_N, C::TH>(ltw_f, a.tw, 0u, 0u, tid);       // both published by the barrier(s)
    stage_twiddles<C::LTW_N, C::TH>(ltw_i, a.tw_inv, 0u, 0u, tid);   // that precede their first LDS use
    if constexpr (C::NR == 1) __syncthreads();   // LP = 4: no exchange would publish them

    // an operand: coefficients -> forward transform (natural-order load, window A0), or evals ->
    // the 16 consecutive values of this thread; canonical in the [0,4) register window either way
    auto operand = [&](const u64 *__restrict__ src, bool is_evals, bool keep, u64 (&v)[16], auto fresh) {
        const u64 *__restrict__ p = src + ubase;
        if (is_evals) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const ulonglong2 x = ld_c<ulonglong2>(p, off + tf * 128u + j * 16u);
                v[2 * j] = x.x;
                v[2 * j + 1] = x.y;
            }
        } else {
            ld16<false>(v, p, C::A0, off + tf * 8u);
            if constexpr (AR == 2 || AR == 4) {   // canonical only where the evals are an output (see the product below)
                fwd_rounds_contig_pm<LP, kPmOne, decltype(fresh)::value, AR>(v, lds, ltw_f, a.tw, 0u, 0u, w, tf, m);
                if (keep) {
#pragma unroll
                    for (int k = 0; k < 16; k++) v[k] = AR == 2 ? pm_canon(v[k], m) : canon8(v[k], m);
                }
            } else {
                fwd_rounds_single<LP, WIDE, decltype(fresh)::value>(v, lds, ltw_f, a.tw, w, tf, m);
            }
        }
    };
    auto store_evals = [&](u64 *dst, const u64 (&v)[16]) {
        if (!dst || !active) return;
        u64 *__restrict__ p = dst + ubase;
#pragma unroll
        for (int j = 0; j < 8; j++) st_c<ulonglong2>(p, off + tf * 128u + j * 16u, ulonglong2{v[2 * j], v[2 * j + 1]});
    };

    u64 va[16], vb[16];
    operand(a.in, a.flags & 1u, a.out3 != nullptr, va, std::true_type{});
    store_evals(a.out3, va);
    operand(a.in2, a.flags & 2u, a.out4 != nullptr, vb, std::false_type{});   // the tile may have been used by the first operand
    store_evals(a.out4, vb);
    // zip_eq(l,r).map(l*r), ring_nq.rs:601-604
    if constexpr (AR == 2) {   // lazy operands are fine: a < 7.1 q as it is, b brought below 2^k; the product is canonical only as an output
#pragma unroll
        for (int k = 0; k < 16; k++) va[k] = mul_var_pm(va[k], pm_below_2k(vb[k], m), m);
        if (a.out2) {
#pragma unroll
            for (int k = 0; k < 16; k++) va[k] = pm_canon(va[k], m);
        }
    } else if constexpr (AR == 4) {   // q = 1 (mod 2^32): lazy operands below 7q, the product below 3q (zq_device.hpp: mul_var_mg)
#pragma unroll
        for (int k = 0; k < 16; k++) va[k] = mul_var_mg(va[k], vb[k], m);
        if (a.out2) {
#pragma unroll
            for (int k = 0; k < 16; k++) va[k] = canon4(va[k], m);
        }
    } else if constexpr (AR == 3) {
#pragma unroll
        for (int k = 0; k < 16; k++) va[k] = mul_mod_var63(va[k], vb[k], m);
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) va[k] = mul_mod_var(va[k], vb[k], m);
    }
    store_evals(a.out2, va);
    if constexpr (C::NR == 1) { /* twiddles published above */ } else if (a.flags == 3u) __syncthreads();   // no forward exchange ran
    if constexpr (AR == 2) inv_rounds_contig_pm<LP, true, false, kPmMul>(va, lds, ltw_i, a.tw_inv, 0u, 0u, w, tf, m, a.ninv, a.s_ninv);
    else if constexpr (AR == 4) inv_rounds_contig_pm<LP, true, false, kMgMul, 4>(va, lds, ltw_i, a.tw_inv, 0u, 0u, w, tf, m, a.ninv, a.s_ninv);
    else inv_rounds_single<LP, WIDE>(va, lds, ltw_i, a.tw_inv, w, tf, m, a.ninv, a.s_ninv);
    if (active) {
        u64 *__restrict__ pout = a.out + ubase;
#pragma unroll
        for (int k = 0; k < 16; k++) va[k] = AR == 3 ? va[k] : AR == 2 ? pm_canon(va[k], m) : AR == 4 ? canon4(va[k], m) : canon2(va[k], m);
        st16<false>(pout, C::A0, off + tf * 8u, va);
    }
}

// ---------------------------------------------------------------------------
// MIDDLE of the product at two-pass sizes (n >= 2^14).  ring_nq.rs:586-607 as passes over HBM is
//   strided(a) -> contiguous(a) | strided(b) -> contiguous(b) | [A .* B -> contiguous^-1] -> strided^-1.
// The forward's contiguous pass and the inverse's first pass act on the SAME 2^LP-coefficient blocks
// (block `blk` of a polynomial), so the three middle passes are one kernel here: a block of each
// operand comes in from its strided pass (lazy, < 6q), runs its LP forward stages in registers / LDS,
// the pointwise product is formed in registers (zip_eq(l,r).map(l*r), ring_nq.rs:601-604) and runs the
// LP inverse stages; what leaves is what ntt_inv_contig_kernel would have written (lazy, < 4q / 2q),
// for the strided inverse pass to finish.  HBM traffic of a product: 104n -> 72n bytes (+8n per
// evals output kept).  An operand flagged as evals (flags bit 0 / 1) is read as such: 16 consecutive
// canonical values per thread, no forward stages.  out2 / out3 / out4: the evals of the product and
// of the two operands (ring_nq.rs:568-573,606), optional.
// ---------------------------------------------------------------------------
template <int LP, int AR>
__global__ __launch_bounds__(ContigCfg<LP>::TH) void rq_mul_mid_kernel(PassArgs a) {
    constexpr int WIDE = AR == 1 ? 1 : AR == 3 ? kStrict : 0;
    using C = ContigCfg<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    // ONE twiddle tile (round 4): the forward table's while the operands are transformed, the inverse table's afterwards —
    // each lane holds its entry of the second in registers from the start and swaps it in between two barriers around the
    // pointwise product.  38 KiB instead of 42 KiB of LDS: FOUR workgroups per CU instead of three.
    Tw *ltw_f = reinterpret_cast<Tw *>(smem_raw + C::DATA_BYTES);
    Tw *ltw_i = FHE_MID_ONE_TILE ? ltw_f : ltw_f + C::LTW_N;
    static_assert(C::LTW_N == C::TH, "one tile entry per lane");
    const u32 tid = threadIdx.x, w = tid / C::TPB, tf = tid % C::TPB;
    const u32 s0 = a.log_n - LP;
    const u32 blk = blockIdx.x & ((1u << s0) - 1u);
    const u64 pg = (u64)(blockIdx.x >> s0);
    const u64 n = 1ull << a.log_n;
    const u32 live = (u32)min((u64)C::W, a.batch - pg * C::W);
    const bool active = w < live;
    const u64 ubase = pg * C::W * n + (u64)blk * C::M;
    const u32 off = ((active ? w : 0u) << a.log_n) * 8u;
    const Mod &m = a.mod;
    stage_twiddles<C::LTW_N, C::TH>(ltw_f, a.tw, s0, blk, tid);       // published by the barrier that precedes its first LDS use
    // this lane's entry of the inverse tile (stage_twiddles' formula), in registers until the tile changes hands
    const u32 tw_i_ls = 31u - (u32)__builtin_clz(tid | 1u), tw_i_l1 = tid | (tid == 0);
    const Tw tw_i_entry = a.tw_inv[(1u << (s0 + tw_i_ls)) + (blk << tw_i_ls) + (tw_i_l1 - (1u << tw_i_ls))];
    static_assert(C::NR >= 2, "two-pass sizes have LP >= 8");

    // keep: the operand's evals are an output, so they must be canonical; otherwise they stay as the last
    // stage left them (< 4q < 2^63): the variable x variable product reduces any 128-bit value
    // both operands' loads are issued before anything is computed: the second operand's latency hides behind the first's stages
    auto fetch = [&](const u64 *__restrict__ src, bool is_evals, u64 (&v)[16]) {
        const u64 *__restrict__ p = src + ubase;
        if (is_evals) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const ulonglong2 x = ld_c<ulonglong2>(p, off + tf * 128u + j * 16u);
                v[2 * j] = x.x;
                v[2 * j + 1] = x.y;
            }
        } else {
            ld16<true>(v, p, C::A0, off + tf * 8u);
        }
    };
    auto operand = [&](bool is_evals, bool keep, u64 (&v)[16], auto fresh) {
        if (!is_evals) {
            if constexpr (AR == 2 || AR == 4) fwd_rounds_contig_pm<LP, kPmPassBound, decltype(fresh)::value, AR>(v, lds, ltw_f, a.tw, s0, blk, w, tf, m);
            else fwd_rounds_contig<LP, WIDE, true, kPassBound, decltype(fresh)::value>(v, lds, ltw_f, a.tw, s0, blk, w, tf, m);
            if (keep) {
#pragma unroll
                for (int k = 0; k < 16; k++) v[k] = AR == 3 ? v[k] : AR == 2 ? pm_canon(v[k], m) : AR == 4 ? canon8(v[k], m) : canon4(v[k], m);
            }
        }
    };
    auto store_evals = [&](u64 *dst, const u64 (&v)[16]) {
        if (!dst || !active) return;
        u64 *__restrict__ p = dst + ubase;
#pragma unroll
        for (int j = 0; j < 8; j++) st_c<ulonglong2>(p, off + tf * 128u + j * 16u, ulonglong2{v[2 * j], v[2 * j + 1]});
    };

    u64 va[16], vb[16];
    fetch(a.in, a.flags & 1u, va);
    if (FHE_MID_EARLY_FETCH) fetch(a.in2, a.flags & 2u, vb);
    operand(a.flags & 1u, a.out3 != nullptr, va, std::true_type{});
    store_evals(a.out3, va);
    if (!FHE_MID_EARLY_FETCH) fetch(a.in2, a.flags & 2u, vb);
    operand(a.flags & 2u, a.out4 != nullptr, vb, std::false_type{});   // the tile may have been used by the first operand
    store_evals(a.out4, vb);
    if (FHE_MID_ONE_TILE) __syncthreads();   // every wave is past the forward rounds: the twiddle tile changes hands
    ltw_i[tid] = tw_i_entry;
    if constexpr (AR == 2) {   // as in rq_mul_fused_kernel
#pragma unroll
        for (int k = 0; k < 16; k++) va[k] = mul_var_pm(va[k], pm_below_2k(vb[k], m), m);
        if (a.out2) {
#pragma unroll
            for (int k = 0; k < 16; k++) va[k] = pm_canon(va[k], m);
        }
    } else if constexpr (AR == 4) {   // as in rq_mul_fused_kernel
#pragma unroll
        for (int k = 0; k < 16; k++) va[k] = mul_var_mg(va[k], vb[k], m);
        if (a.out2) {
#pragma unroll
            for (int k = 0; k < 16; k++) va[k] = canon4(va[k], m);
        }
    } else if constexpr (AR == 3) {
#pragma unroll
        for (int k = 0; k < 16; k++) va[k] = mul_mod_var63(va[k], vb[k], m);
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) va[k] = mul_mod_var(va[k], vb[k], m);
    }
    store_evals(a.out2, va);
    __syncthreads();                     // the inverse tile is in place (the first inverse round reads it before any exchange)
    if constexpr (AR == 2) inv_rounds_contig_pm<LP, false, false, kPmMul>(va, lds, ltw_i, a.tw_inv, s0, blk, w, tf, m, a.ninv, a.s_ninv);
    else if constexpr (AR == 4) inv_rounds_contig_pm<LP, false, false, kMgMul, 4>(va, lds, ltw_i, a.tw_inv, s0, blk, w, tf, m, a.ninv, a.s_ninv);
    else inv_rounds_contig<LP, WIDE, false, false>(va, lds, ltw_i, a.tw_inv, s0, blk, w, tf, m, a.ninv, a.s_ninv);
    if (active) {
        u64 *__restrict__ pout = a.out + ubase;
        st16<false>(pout, C::A0, off + tf * 8u, va);   // lazy
    }
}

// RSRC: the input rows are 2^src_log_n arbitrary words, reduced mod q and zero-padded in the load
// (see SRC_REDUCE above).
template <int LA, int CW, int AR, bool RSRC = false>
__global__ __launch_bounds__((StridedCfg<LA, CW>::TH)) void ntt_fwd_strided_kernel(PassArgs a) {
    constexpr int WIDE = AR == 1 ? 1 : AR == 3 ? kStrict : 0;
    static_assert(AR != 2 || !RSRC, "reducing loads run on the Shoup tables");
    using C = StridedCfg<LA, CW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const u32 tid = threadIdx.x, c = tid % CW, tf = tid / CW;
    const u32 lb = a.log_n - LA;               // log2 of the row length
    const u32 lcg = lb - __builtin_ctz(CW);    // log2(column groups per polynomial)
    const u32 cg = blockIdx.x & ((1u << lcg) - 1u);
    const u64 poly = (u64)(blockIdx.x >> lcg);
    // wave-uniform 64-bit base (SGPRs) + 32-bit per-lane element offsets: one v_add_u32 per
    // access instead of 64-bit address arithmetic (a polynomial spans < 2^32 bytes)
    const u64 ubase = (poly << a.log_n) + (u64)cg * CW;
    // gridDim.y = 2: the two operands of a product in one launch (row 1 = in2 -> out2)
    const u64 *__restrict__ pin = (blockIdx.y ? a.in2 : a.in) + ubase;
    u64 *__restrict__ pout = (blockIdx.y ? a.out2 : a.out) + ubase;
    const Mod &m = a.mod;
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + C::DATA_BYTES);
    const Tw *tw = ltw;

    u64 v[16];
    if constexpr (RSRC) {
        const u64 *__restrict__ src = a.in + (poly << a.src_log_n) + (u64)cg * CW;
        const u32 nsrc = 1u << a.src_log_n, col = cg * CW + c;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const u32 e = field_of<C::A0>(tf, k) << lb;          // row start; element e + col
            v[k] = e + col < nsrc ? reduce_any(ld_at<u64>(src, (e + c) * 8u), m) : 0ull;
        }
    } else {
        ld16<true>(v, pin, C::A0 + lb, ((field_of<C::A0>(tf, 0) << lb) + c) * 8u);   // row k-term on the scalar base
    }
    for (u32 li = tid; li < (u32)C::F; li += C::TH) ltw[li] = a.tw[li];   // first pass: s0 = 0, blk = 0

    // inputs are canonical (bound 2 leaves slack); the pass ends below kPassBound*q (END6)
    constexpr int B0 = 2, B1 = fwd_bound_out(C::R0, B0), B2 = fwd_bound_out(4, B1);
    static_assert(kPassBound == 6, "END6 ends a pass below 6q");
    constexpr int AK = AR == 4 ? 4 : 2;
    constexpr int P0 = kPmOne, P1 = pm_fwd_bound_out(C::R0, P0, AK), P2 = pm_fwd_bound_out(4, P1, AK), P3 = pm_fwd_bound_out(4, P2, AK);
    static_assert((C::NR == 1 ? P1 : C::NR == 2 ? P2 : P3) <= kPmPassBound, "AR == 2 / 4: a strided pass ends below kPmPassBound");
    if constexpr (AR == 2 || AR == 4) round_fwd_pm<C::R0, P0, true, AK>(v, a.tw, 1u, m);
    else round_fwd<C::R0, WIDE, B0, false, C::NR == 1>(v, a.tw, 1u, m);   // uniform twiddles: scalar loads from the global table
    if constexpr (C::NR > 1) {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        exchange_strided<CW, C::A0, A, true>(v, lds, c, tf);   // its barrier also publishes ltw
        if constexpr (AR == 2 || AR == 4) round_fwd_pm<4, P1, false, AK>(v, tw, (1u << LS) + (tf >> A), m);
        else round_fwd<4, WIDE, B1, false, C::NR == 2>(v, tw, (1u << LS) + (tf >> A), m);
    }
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        exchange_strided<CW, C::a_of(1), A, false>(v, lds, c, tf);
        if constexpr (AR == 2 || AR == 4) round_fwd_pm<4, P2, false, AK>(v, tw, (1u << LS) + (tf >> A), m);
        else round_fwd<4, WIDE, B2, false, C::NR == 3>(v, tw, (1u << LS) + (tf >> A), m);
    }
    constexpr int ALAST = C::a_of(C::NR - 1);
    st16<true>(pout, ALAST + lb, ((field_of<ALAST>(tf, 0) << lb) + c) * 8u, v);  // lazy: < 4q, or < 6q (WIDE)
}

template <int LA, int CW, int AR>
__global__ __launch_bounds__((StridedCfg<LA, CW>::TH)) void ntt_inv_strided_kernel(PassArgs a) {
    constexpr int WIDE = AR == 1 ? 1 : AR == 3 ? kStrict : 0;
    using C = StridedCfg<LA, CW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const u32 tid = threadIdx.x, c = tid % CW, tf = tid / CW;
    const u32 lb = a.log_n - LA;
    const u32 lcg = lb - __builtin_ctz(CW);
    const u32 cg = blockIdx.x & ((1u << lcg) - 1u);
    const u64 poly = (u64)(blockIdx.x >> lcg);
    const u64 ubase = (poly << a.log_n) + (u64)cg * CW;
    const u64 *__restrict__ pin = a.in + ubase;
    u64 *__restrict__ pout = a.out + ubase;
    const Mod &m = a.mod;
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + C::DATA_BYTES);
    const Tw *tw = ltw;

    constexpr int ALAST = C::a_of(C::NR - 1);
    u64 v[16];
    ld16<true>(v, pin, ALAST + lb, ((field_of<ALAST>(tf, 0) << lb) + c) * 8u);  // < 2q
    for (u32 li = tid; li < (u32)C::F; li += C::TH) ltw[li] = a.tw[li];
    __syncthreads();

    // the contiguous pass before this one hands values below 4q (WIDE) / 2q: every round starts from 4
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        if constexpr (AR == 2 || AR == 4) round_inv_pm<4, false, ar_inv_bound(AR), false, AR>(v, tw, (1u << LS) + (tf >> A), m, a.ninv, a.s_ninv);
        else round_inv_sel<4, false, WIDE, 4>(v, tw, (1u << LS) + (tf >> A), m, a.ninv, a.s_ninv);
        exchange_strided<CW, A, C::a_of(1), true>(v, lds, c, tf);
    }
    if constexpr (C::NR > 1) {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        if constexpr (AR == 2 || AR == 4) round_inv_pm<4, false, ar_inv_bound(AR), false, AR>(v, tw, (1u << LS) + (tf >> A), m, a.ninv, a.s_ninv);
        else round_inv_sel<4, false, WIDE, 4>(v, tw, (1u << LS) + (tf >> A), m, a.ninv, a.s_ninv);
        exchange_strided<CW, A, C::A0, (C::NR <= 2)>(v, lds, c, tf);
    }
    if constexpr (AR == 2 || AR == 4) round_inv_pm<C::R0, true, ar_inv_bound(AR), true, AR>(v, a.tw, 1u, m, a.ninv, a.s_ninv);   // uniform: scalar loads
    else round_inv_sel<C::R0, true, WIDE, 4>(v, tw, 1u, m, a.ninv, a.s_ninv);
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = AR == 3 ? v[k] : AR == 2 ? pm_canon(v[k], m) : AR == 4 ? canon4(v[k], m) : canon2(v[k], m);
    st16<true>(pout, C::A0 + lb, ((field_of<C::A0>(tf, 0) << lb) + c) * 8u, v);
}

// ---------------------------------------------------------------------------
// n in {2,4,8}: one thread per polynomial, stage loops as in the reference.
// ---------------------------------------------------------------------------
template <bool INV>
__global__ __launch_bounds__(256) void ntt_tiny_kernel(PassArgs a) {
    const u64 poly = (u64)blockIdx.x * 256 + threadIdx.x;
    if (poly >= a.batch) return;
    const u32 n = 1u << a.log_n;
    const Mod &m = a.mod;
    u64 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (u32)i < n ? a.in[poly * n + i] : 0ull;
    if (!INV) {
        for (u32 s = 0; s < a.log_n; s++) {
            const u32 t = n >> (s + 1);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if ((u32)j < n && !((u32)j & t)) {
                    const Tw w = a.tw[(1u << s) + ((u32)j >> (a.log_n - s))];
                    // static register indexing: j + t is one of j+1, j+2, j+4
                    u64 x = v[j], y = (t == 1) ? v[(j + 1) & 7] : (t == 2) ? v[(j + 2) & 7] : v[(j + 4) & 7];
                    ct_bfly<2>(x, y, w.w, w.wp, m);
                    v[j] = x;
                    if (t == 1) v[(j + 1) & 7] = y; else if (t == 2) v[(j + 2) & 7] = y; else v[(j + 4) & 7] = y;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
            if ((u32)i < n) a.out[poly * n + i] = canon4(v[i], m);
    } else {
        for (int s = (int)a.log_n - 1; s >= 0; s--) {
            const u32 t = n >> (s + 1);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if ((u32)j < n && !((u32)j & t)) {
                    const Tw w = a.tw[(1u << s) + ((u32)j >> (a.log_n - s))];
                    u64 x = v[j], y = (t == 1) ? v[(j + 1) & 7] : (t == 2) ? v[(j + 2) & 7] : v[(j + 4) & 7];
                    gs_bfly(x, y, w.w, w.wp, m);
                    v[j] = x;
                    if (t == 1) v[(j + 1) & 7] = y; else if (t == 2) v[(j + 2) & 7] = y; else v[(j + 4) & 7] = y;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
            if ((u32)i < n)
                a.out[poly * n + i] = canon2(mul_shoup_lazy(v[i], a.ninv.w, a.ninv.wp, m), m);
    }
}

#ifndef FHE_NTT_TU_Q62   // (non-template kernels: the main translation unit only)
// ---------------------------------------------------------------------------
// element-wise kernels
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pointwise_mul_kernel(const u64 *__restrict__ x,
                                                            const u64 *__restrict__ y,
                                                            u64 *__restrict__ z, u64 count, Mod m) {
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += stride)
        z[i] = mul_mod_var(x[i], y[i], m);
}

__global__ __launch_bounds__(256) void fill_synthetic_kernel(u64 *__restrict__ out, u64 count,
                                                             u64 q, u64 seed, u64 first) {
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += stride)
        out[i] = __umul64hi(splitmix64(seed ^ (first + i)), q);
}

// any value >= q sets *flag (fhe_rq_check_canonical)
__global__ __launch_bounds__(256) void check_canonical_kernel(const u64 *__restrict__ x, u64 count,
                                                              u64 q, int *flag) {
    const u64 stride = (u64)gridDim.x * 256;
    int bad = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) bad |= (x[i] >= q);
    if (bad) atomicOr(flag, 1);
}

#endif

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static inline hipError_t post_launch() { return hipGetLastError(); }


template <int LP, bool FINAL, int AR, int SRC = SRC_PLAIN>
static hipError_t launch_fwd_contig(const PassArgs &a, hipStream_t st) {
    using C = ContigCfg<LP>;
    const u64 nb = 1ull << (a.log_n - LP);
    const u64 groups = (a.batch + C::W - 1) / C::W;
    const u64 grid = nb * groups;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    constexpr size_t lds_bytes = (SRC == SRC_DIGITS || SRC == SRC_ZQBITS) ? C::LDS_BYTES_BITS : C::LDS_BYTES;
    if (hipError_t e = allow_big_lds((const void *)ntt_fwd_contig_kernel<LP, FINAL, AR, SRC>, lds_bytes)) return e;
    KernelTimer kt(SRC == SRC_DIGITS ? "ntt_fwd_digits" : SRC == SRC_ZQBITS ? "ntt_fwd_zqbits" : SRC == SRC_REDUCE ? "ntt_fwd_reduce" : (FINAL ? "ntt_fwd_contig_final" : "ntt_fwd_contig"), LP, st);
    hipLaunchKernelGGL((ntt_fwd_contig_kernel<LP, FINAL, AR, SRC>), dim3((unsigned)grid), dim3(C::TH),
                       lds_bytes, st, a);
    return post_launch();
}

template <int LP, bool FINAL, bool MUL_IN, int AR>
static hipError_t launch_inv_contig(const PassArgs &a, hipStream_t st) {
    using C = ContigCfg<LP>;
    const u64 nb = 1ull << (a.log_n - LP);
    const u64 groups = (a.batch + C::W - 1) / C::W;
    const u64 grid = nb * groups;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    if (hipError_t e = allow_big_lds((const void *)ntt_inv_contig_kernel<LP, FINAL, MUL_IN, AR>, C::LDS_BYTES)) return e;
    KernelTimer kt(MUL_IN ? "ntt_inv_contig_mul" : (FINAL ? "ntt_inv_contig_final" : "ntt_inv_contig"), LP, st);
    hipLaunchKernelGGL((ntt_inv_contig_kernel<LP, FINAL, MUL_IN, AR>), dim3((unsigned)grid),
                       dim3(C::TH), C::LDS_BYTES, st, a);
    return post_launch();
}

template <int LA, int CW, bool INV, int AR, bool RSRC = false>
static hipError_t launch_strided(const PassArgs &a, hipStream_t st, unsigned operands = 1) {
    using C = StridedCfg<LA, CW>;
    const u64 ncg = (1ull << (a.log_n - LA)) / CW;
    const u64 grid = ncg * a.batch;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    if (hipError_t e = allow_big_lds(INV ? (const void *)ntt_inv_strided_kernel<LA, CW, AR>
                                         : (const void *)ntt_fwd_strided_kernel<LA, CW, AR, RSRC>, C::LDS_BYTES)) return e;
    KernelTimer kt(INV ? "ntt_inv_strided" : (RSRC ? "ntt_fwd_strided_reduce" : "ntt_fwd_strided"), LA, st);
    if (INV)
        hipLaunchKernelGGL((ntt_inv_strided_kernel<LA, CW, AR>), dim3((unsigned)grid), dim3(C::TH),
                           C::LDS_BYTES, st, a);
    else
        hipLaunchKernelGGL((ntt_fwd_strided_kernel<LA, CW, AR, RSRC>), dim3((unsigned)grid, operands), dim3(C::TH),
                           C::LDS_BYTES, st, a);
    return post_launch();
}

#ifndef FHE_STRIDED_CW8
#define FHE_STRIDED_CW8 32        // columns of a strided-pass workgroup at 8 strided stages (shape experiments: 16 / 64)
#endif
#ifndef CONTIG_CASES             // (tools/isa_audit.sh narrows the list for a quick listing of the headline kernels)
#define CONTIG_CASES(X) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)
#endif

template <int LP, int AR>
static hipError_t launch_rq_mul_fused_lp(const PassArgs &a, hipStream_t st) {
    using C = ContigCfg<LP>;
    const u64 grid = (a.batch + C::W - 1) / C::W;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    constexpr size_t lds_bytes = C::LDS_BYTES + (size_t)C::LTW_N * sizeof(Tw);   // a second twiddle tile
    if (hipError_t e = allow_big_lds((const void *)rq_mul_fused_kernel<LP, AR>, lds_bytes)) return e;
    KernelTimer kt("rq_mul_fused", LP, st);
    hipLaunchKernelGGL((rq_mul_fused_kernel<LP, AR>), dim3((unsigned)grid), dim3(C::TH), lds_bytes, st, a);
    return post_launch();
}

template <int LP, int AR>
static hipError_t launch_rq_mul_mid_lp(const PassArgs &a, hipStream_t st) {
    using C = ContigCfg<LP>;
    const u64 nb = 1ull << (a.log_n - LP);
    const u64 grid = nb * ((a.batch + C::W - 1) / C::W);
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    constexpr size_t lds_bytes = C::LDS_BYTES + (FHE_MID_ONE_TILE ? 0 : (size_t)C::LTW_N * sizeof(Tw));   // ONE twiddle tile: four workgroups per CU
    if (hipError_t e = allow_big_lds((const void *)rq_mul_mid_kernel<LP, AR>, lds_bytes)) return e;
    KernelTimer kt("rq_mul_mid", LP, st);
    hipLaunchKernelGGL((rq_mul_mid_kernel<LP, AR>), dim3((unsigned)grid), dim3(C::TH), lds_bytes, st, a);
    return post_launch();
}

// ---- two translation units ------------------------------------------------------------------------------------------------
// Every kernel above is instantiated per arithmetic AR, per size and per variant; one translation unit for all of them was
// the library's longest compile by far.  The arithmetics with the narrow headroom — AR = 0 (2^61 <= q < 2^62, Harvey) and
// AR = 3 (2^62 <= q < 2^63, strict) — are instantiated in ntt_kernels_q62.hip, which includes this file with
// FHE_NTT_TU_Q62 defined and provides the five entry points below; this unit keeps AR = 1, 2, 4 and the public launchers.
hipError_t q62_fwd_contig(int ar, int lp, bool final, const PassArgs &a, hipStream_t st);
hipError_t q62_inv_contig(int ar, int lp, bool final, bool mul_in, const PassArgs &a, hipStream_t st);
hipError_t q62_strided(int ar, bool inv, int la, const PassArgs &a, hipStream_t st, unsigned operands);
hipError_t q62_rq_mul_fused(int ar, int lp, const PassArgs &a, hipStream_t st);
hipError_t q62_rq_mul_mid(int ar, int lp, const PassArgs &a, hipStream_t st);

#ifdef FHE_NTT_TU_Q62
template <int AR>
static hipError_t q62_fwd_contig_ar(int lp, bool final, const PassArgs &a, hipStream_t st) {
    switch (lp) {
#define X(LP_) case LP_: return final ? launch_fwd_contig<LP_, true, AR>(a, st) : launch_fwd_contig<LP_, false, AR>(a, st);
        CONTIG_CASES(X)
#undef X
    }
    return hipErrorInvalidValue;
}
template <int AR>
static hipError_t q62_inv_contig_ar(int lp, bool final, bool mul_in, const PassArgs &a, hipStream_t st) {
    switch (lp) {
#define X(LP_)                                                                                   \
    case LP_:                                                                                    \
        if (final) return mul_in ? launch_inv_contig<LP_, true, true, AR>(a, st)                 \
                                 : launch_inv_contig<LP_, true, false, AR>(a, st);               \
        return mul_in ? launch_inv_contig<LP_, false, true, AR>(a, st)                           \
                      : launch_inv_contig<LP_, false, false, AR>(a, st);
        CONTIG_CASES(X)
#undef X
    }
    return hipErrorInvalidValue;
}
template <bool INV, int AR>
static hipError_t q62_strided_ar(int la, const PassArgs &a, hipStream_t st, unsigned operands) {
    switch (la) {
        case 6: return launch_strided<6, 128, INV, AR>(a, st, operands);
        case 7: return launch_strided<7, 64, INV, AR>(a, st, operands);
        case 8: return launch_strided<8, FHE_STRIDED_CW8, INV, AR>(a, st, operands);
    }
    return hipErrorInvalidValue;
}
hipError_t q62_fwd_contig(int ar, int lp, bool final, const PassArgs &a, hipStream_t st) {
    return ar == 3 ? q62_fwd_contig_ar<3>(lp, final, a, st) : q62_fwd_contig_ar<0>(lp, final, a, st);
}
hipError_t q62_inv_contig(int ar, int lp, bool final, bool mul_in, const PassArgs &a, hipStream_t st) {
    return ar == 3 ? q62_inv_contig_ar<3>(lp, final, mul_in, a, st) : q62_inv_contig_ar<0>(lp, final, mul_in, a, st);
}
hipError_t q62_strided(int ar, bool inv, int la, const PassArgs &a, hipStream_t st, unsigned operands) {
    if (ar == 3) return inv ? q62_strided_ar<true, 3>(la, a, st, operands) : q62_strided_ar<false, 3>(la, a, st, operands);
    return inv ? q62_strided_ar<true, 0>(la, a, st, operands) : q62_strided_ar<false, 0>(la, a, st, operands);
}
hipError_t q62_rq_mul_fused(int ar, int lp, const PassArgs &a, hipStream_t st) {
    switch (lp) {
#define X(LP_) case LP_: return ar == 3 ? launch_rq_mul_fused_lp<LP_, 3>(a, st) : launch_rq_mul_fused_lp<LP_, 0>(a, st);
        CONTIG_CASES(X)
#undef X
    }
    return hipErrorInvalidValue;
}
hipError_t q62_rq_mul_mid(int ar, int lp, const PassArgs &a, hipStream_t st) {
    switch (lp) {
#define X(LP_) case LP_: return ar == 3 ? launch_rq_mul_mid_lp<LP_, 3>(a, st) : launch_rq_mul_mid_lp<LP_, 0>(a, st);
        X(8) X(9) X(10) X(11) X(12)
#undef X
    }
    return hipErrorInvalidValue;
}
#else   // ---- the main translation unit: AR = 1, 2, 4, the dispatchers and the public launchers -------------------------------

// ar: the kernels' AR (0 / 1 = Shoup tables, 2 = pseudo-Mersenne tables: DevicePlan::arith)
static hipError_t fwd_contig_dispatch(int lp, bool final, int ar, const PassArgs &a, hipStream_t st) {
    if (ar == 0 || ar == 3) return q62_fwd_contig(ar, lp, final, a, st);
    switch (lp) {
#define X(LP_)                                                                                   \
    case LP_:                                                                                    \
        if (ar == 4) return final ? launch_fwd_contig<LP_, true, 4>(a, st) : launch_fwd_contig<LP_, false, 4>(a, st); \
        if (ar == 2) return final ? launch_fwd_contig<LP_, true, 2>(a, st) : launch_fwd_contig<LP_, false, 2>(a, st); \
        return final ? launch_fwd_contig<LP_, true, 1>(a, st) : launch_fwd_contig<LP_, false, 1>(a, st);
        CONTIG_CASES(X)
#undef X
    }
    return hipErrorInvalidValue;
}

template <int AR>
static hipError_t inv_contig_dispatch_ar(int lp, bool final, bool mul_in, const PassArgs &a,
                                         hipStream_t st) {
    switch (lp) {
#define X(LP_)                                                                                   \
    case LP_:                                                                                    \
        if (final) return mul_in ? launch_inv_contig<LP_, true, true, AR>(a, st)                 \
                                 : launch_inv_contig<LP_, true, false, AR>(a, st);               \
        return mul_in ? launch_inv_contig<LP_, false, true, AR>(a, st)                           \
                      : launch_inv_contig<LP_, false, false, AR>(a, st);
        CONTIG_CASES(X)
#undef X
    }
    return hipErrorInvalidValue;
}
// the Montgomery tables (AR = 4): plain inverse transforms only (no product in the load)
static hipError_t inv_contig_dispatch_mg(int lp, bool final, const PassArgs &a, hipStream_t st) {
    switch (lp) {
#define X(LP_) case LP_: return final ? launch_inv_contig<LP_, true, false, 4>(a, st) : launch_inv_contig<LP_, false, false, 4>(a, st);
        CONTIG_CASES(X)
#undef X
    }
    return hipErrorInvalidValue;
}
static hipError_t inv_contig_dispatch(int ar, int lp, bool final, bool mul_in, const PassArgs &a, hipStream_t st) {
    if (ar == kArMontgomery) return mul_in ? hipErrorInvalidValue : inv_contig_dispatch_mg(lp, final, a, st);
    if (ar == 0 || ar == 3) return q62_inv_contig(ar, lp, final, mul_in, a, st);
    return ar == 2 ? inv_contig_dispatch_ar<2>(lp, final, mul_in, a, st) : inv_contig_dispatch_ar<1>(lp, final, mul_in, a, st);
}

template <bool INV, int AR>
static hipError_t strided_dispatch_ar(int la, const PassArgs &a, hipStream_t st, unsigned operands) {
    switch (la) {
        case 6: return launch_strided<6, 128, INV, AR>(a, st, operands);
        case 7: return launch_strided<7, 64, INV, AR>(a, st, operands);
        case 8: return launch_strided<8, FHE_STRIDED_CW8, INV, AR>(a, st, operands);   // 16 / 64 columns measured equal / slower
    }
    return hipErrorInvalidValue;
}
template <bool INV>
static hipError_t strided_dispatch(int ar, int la, const PassArgs &a, hipStream_t st, unsigned operands = 1) {
    if (ar == kArMontgomery) return strided_dispatch_ar<INV, 4>(la, a, st, operands);
    if (ar == 0 || ar == 3) return q62_strided(ar, INV, la, a, st, operands);
    return ar == 2 ? strided_dispatch_ar<INV, 2>(la, a, st, operands) : strided_dispatch_ar<INV, 1>(la, a, st, operands);
}
// the tables and n^-1 constants a pass runs on: {w, w 2^32 mod q} for pseudo-Mersenne plans, {w, floor(w 2^64 / q)} otherwise
static inline bool plan_runs_montgomery(const DevicePlan &p) {
    return p.log_n >= 4 && p.tw_fwd_mg != nullptr && p.tw_inv_mg != nullptr && p.arith == kArWide61;
}
// FHE_G63_PLAIN=1: 2^62 <= q < 2^63 on the plain strict kernels of generic63.hip at every size (what shipped until round 5:
// ceil(log2 n / 4) launches); default: the two-pass / fused kernels above with AR = 3
static bool g63_plain() {
    static const bool v = [] { const char *e = getenv("FHE_G63_PLAIN"); return e && e[0] == '1'; }();
    return v;
}
static inline void set_tables(PassArgs &a, const DevicePlan &p, bool inverse) {
    const bool pm = p.arith == kArPMersenne;
    a.tw = inverse ? (pm ? p.tw_inv_pm : p.tw_inv) : (pm ? p.tw_fwd_pm : p.tw_fwd);
    a.mod = p.mod;
    a.ninv = pm ? p.ninv_pm : p.ninv;
    a.s_ninv = pm ? p.s_ninv_pm : p.s_ninv;
    a.log_n = p.log_n;
}


hipError_t launch_ntt_forward(const DevicePlan &p, const u64 *in, u64 *out, u64 batch,
                              u64 batch_tile, hipStream_t st) {
    if (p.arith == kArStrict63 && (p.log_n < 4 || g63_plain())) return launch_g63_forward(p, in, out, batch, st);   // 2^62 <= q < 2^63 at n < 16 (or FHE_G63_PLAIN=1): generic63.hip
    PassArgs a{};
    const int L = p.log_n;
    // n < 16: one thread per polynomial on the Shoup tables
    // q = 1 (mod 2^32): the forward kernels on the word-Montgomery table (AR = 4); every other entry point keeps p.arith
    const bool mg = L >= 4 && p.tw_fwd_mg != nullptr && p.arith == kArWide61;
    const int ar = L < 4 ? (p.wide ? 1 : 0) : mg ? (int)kArMontgomery : p.arith;
    DevicePlan pt = p;
    pt.arith = mg ? (int)kArWide61 : ar;
    set_tables(a, pt, false);
    if (mg) a.tw = p.tw_fwd_mg;
    if (batch == 0) return hipSuccess;
    if (L < 4) {
        a.in = in; a.out = out; a.batch = batch;
        KernelTimer kt("ntt_tiny_fwd", L, st);
        hipLaunchKernelGGL(ntt_tiny_kernel<false>, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, a);
        return post_launch();
    }
    if (L <= kMaxSinglePassLog) {
        a.in = in; a.out = out; a.batch = batch;
        return fwd_contig_dispatch(L, true, ar, a, st);
    }
    const int LB = contig_bits(L), LA = L - LB;
    const u64 n = 1ull << L;
    if (batch_tile == 0) batch_tile = batch;
    for (u64 b0 = 0; b0 < batch; b0 += batch_tile) {
        const u64 nb = batch - b0 < batch_tile ? batch - b0 : batch_tile;
        a.in = in + b0 * n; a.out = out + b0 * n; a.batch = nb;
        hipError_t e = strided_dispatch<false>(ar, LA, a, st);
        if (e != hipSuccess) return e;
        a.in = out + b0 * n;
        e = fwd_contig_dispatch(LB, true, ar, a, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_rq_mul_fused(const DevicePlan &p, const u64 *a_, bool a_is_evals, const u64 *b_, bool b_is_evals,
                               u64 *c, u64 *c_evals, u64 *a_evals, u64 *b_evals, u64 batch, hipStream_t st) {
    const int L = p.log_n;
    if (L < 4 || L > kMaxSinglePassLog || (p.arith == kArStrict63 && g63_plain())) return hipErrorNotSupported;
    if (batch == 0) return hipSuccess;
    PassArgs a{};
    set_tables(a, p, false);
    a.tw_inv = p.arith == kArPMersenne ? p.tw_inv_pm : p.tw_inv;
    const bool mg = plan_runs_montgomery(p);      // q = 1 (mod 2^32): both transforms and the product on the Montgomery tables
    if (mg) { a.tw = p.tw_fwd_mg; a.tw_inv = p.tw_inv_mg; a.ninv = p.ninv_mg; a.s_ninv = p.s_ninv_mg; }
    a.in = a_; a.in2 = b_; a.out = c; a.out2 = c_evals; a.out3 = a_evals; a.out4 = b_evals;
    a.flags = (a_is_evals ? 1u : 0u) | (b_is_evals ? 2u : 0u);
    a.batch = batch;
    if (!mg && (p.arith == 0 || p.arith == 3)) return q62_rq_mul_fused(p.arith, L, a, st);
    switch (L) {
#define X(LP_) case LP_: return mg ? launch_rq_mul_fused_lp<LP_, 4>(a, st) : p.arith == 2 ? launch_rq_mul_fused_lp<LP_, 2>(a, st) : launch_rq_mul_fused_lp<LP_, 1>(a, st);
        CONTIG_CASES(X)
#undef X
    }
    return hipErrorInvalidValue;
}

// c = a * b at two-pass sizes: [strided(a) | strided(b)] (one launch) -> middle kernel -> strided^-1.
// wa / wb: where the strided pass of each coefficient operand goes (scratch, or the operand's evals
// output, which the middle kernel then overwrites in place with the canonical evals).
hipError_t launch_rq_mul_two_pass(const DevicePlan &p, const u64 *a_, bool a_is_evals, const u64 *b_, bool b_is_evals,
                                  u64 *c, u64 *c_evals, u64 *wa, bool keep_a_evals, u64 *wb, bool keep_b_evals,
                                  u64 batch, u64 batch_tile, hipStream_t st) {
    const int L = p.log_n;
    if (L <= kMaxSinglePassLog || L > kMaxLog || (p.arith == kArStrict63 && g63_plain())) return hipErrorNotSupported;
    if (batch == 0) return hipSuccess;
    const int LB = contig_bits(L), LA = L - LB;
    const u64 n = 1ull << L;
    if (batch_tile == 0) batch_tile = batch;
    // (Round 5 measured the tiles of a batch alternating between the caller's stream and a helper stream, so that one
    // tile's middle kernel — bound by its instruction count — runs beside another's memory-bound strided passes: 0.909 ->
    // 0.903 .. 0.933 M products/s at 2^16 for tiles of 128 .. 1024 polynomials, i.e. nothing; gpurun_out/r5f.  Not kept.)
    for (u64 b0 = 0; b0 < batch; b0 += batch_tile) {
        const u64 nb = batch - b0 < batch_tile ? batch - b0 : batch_tile, o = b0 * n;
        const bool mg = plan_runs_montgomery(p);
        const int ar = mg ? (int)kArMontgomery : p.arith;
        PassArgs f{};
        set_tables(f, p, false); f.batch = nb;
        if (mg) f.tw = p.tw_fwd_mg;
        hipError_t e = hipSuccess;
        if (!a_is_evals && !b_is_evals) {
            f.in = a_ + o; f.out = wa + o; f.in2 = b_ + o; f.out2 = wb + o;
            e = strided_dispatch<false>(ar, LA, f, st, 2);
        } else if (!a_is_evals) {
            f.in = a_ + o; f.out = wa + o;
            e = strided_dispatch<false>(ar, LA, f, st);
        } else if (!b_is_evals) {
            f.in = b_ + o; f.out = wb + o;
            e = strided_dispatch<false>(ar, LA, f, st);
        }
        if (e != hipSuccess) return e;
        PassArgs m{};
        set_tables(m, p, false); m.tw_inv = ar == 2 ? p.tw_inv_pm : p.tw_inv; m.batch = nb;
        if (mg) { m.tw = p.tw_fwd_mg; m.tw_inv = p.tw_inv_mg; m.ninv = p.ninv_mg; m.s_ninv = p.s_ninv_mg; }
        m.in = (a_is_evals ? a_ : wa) + o; m.in2 = (b_is_evals ? b_ : wb) + o;
        m.out = c + o; m.out2 = c_evals ? c_evals + o : nullptr;
        m.out3 = (keep_a_evals && !a_is_evals) ? wa + o : nullptr;
        m.out4 = (keep_b_evals && !b_is_evals) ? wb + o : nullptr;
        m.flags = (a_is_evals ? 1u : 0u) | (b_is_evals ? 2u : 0u);
        if (ar == 0 || ar == 3) e = q62_rq_mul_mid(ar, LB, m, st);
        else switch (LB) {
#define X(LP_) case LP_: e = ar == 4 ? launch_rq_mul_mid_lp<LP_, 4>(m, st) : ar == 2 ? launch_rq_mul_mid_lp<LP_, 2>(m, st) : launch_rq_mul_mid_lp<LP_, 1>(m, st); break;
            X(8) X(9) X(10) X(11) X(12)
#undef X
            default: return hipErrorInvalidValue;
        }
        if (e != hipSuccess) return e;
        PassArgs i{};
        set_tables(i, p, true); i.batch = nb;
        if (mg) { i.tw = p.tw_inv_mg; i.ninv = p.ninv_mg; i.s_ninv = p.s_ninv_mg; }
        i.in = c + o; i.out = c + o;
        e = strided_dispatch<true>(ar, LA, i, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_ntt_forward_digits(const DevicePlan &p, const u64 *in, u64 *out, u64 rows, uint32_t l,
                                     hipStream_t st) {
    const int L = p.log_n;
    if (L < 4 || L > kMaxSinglePassLog || !p.wide || l == 0 || l > 64 || !p.digit_lut) return hipErrorNotSupported;
    if (rows == 0) return hipSuccess;
    PassArgs a{};
    a.tw = p.tw_fwd;
    a.mod = p.mod;
    a.log_n = p.log_n;
    a.in = in; a.out = out; a.batch = rows * l; a.digit_l = l; a.lut = p.digit_lut;
    switch (L) {
#define X(LP_) case LP_: return launch_fwd_contig<LP_, true, true, SRC_DIGITS>(a, st);
        CONTIG_CASES(X)
#undef X
    }
    return hipErrorInvalidValue;
}

hipError_t launch_ntt_forward_zqbits(const DevicePlan &p, const u64 *in, u64 *out, u64 rows, uint32_t l,
                                     uint32_t grp, u64 gstride, hipStream_t st) {
    const int L = p.log_n;
    if (L < 4 || L > kMaxSinglePassLog || !p.wide || l == 0 || l > 64 || grp == 0 || p.mod.q < 3 || !p.digit_lut) return hipErrorNotSupported;
    if (rows == 0) return hipSuccess;
    PassArgs a{};
    a.tw = p.tw_fwd;
    a.mod = p.mod;
    a.log_n = p.log_n;
    a.in = in; a.out = out; a.batch = rows * l; a.digit_l = l; a.src_grp = grp; a.src_gstride = gstride; a.lut = p.digit_lut;
    switch (L) {
#define X(LP_) case LP_: return launch_fwd_contig<LP_, true, true, SRC_ZQBITS>(a, st);
        CONTIG_CASES(X)
#undef X
    }
    return hipErrorInvalidValue;
}

hipError_t launch_ntt_forward_reduce(const DevicePlan &p, const u64 *in, u64 *out, u64 rows,
                                     uint32_t src_log_n, u64 batch_tile, hipStream_t st) {
    const int L = p.log_n;
    if (L < 4 || !p.wide || src_log_n > (uint32_t)L) return hipErrorNotSupported;
    if (rows == 0) return hipSuccess;
    PassArgs a{};
    a.tw = p.tw_fwd;
    a.mod = p.mod;
    a.log_n = p.log_n;
    a.src_log_n = src_log_n;
    if (L <= kMaxSinglePassLog) {
        a.in = in; a.out = out; a.batch = rows;
        switch (L) {
#define X(LP_) case LP_: return launch_fwd_contig<LP_, true, true, SRC_REDUCE>(a, st);
            CONTIG_CASES(X)
#undef X
        }
        return hipErrorInvalidValue;
    }
    const int LB = contig_bits(L), LA = L - LB;
    const u64 n = 1ull << L, nsrc = 1ull << src_log_n;
    if (batch_tile == 0) batch_tile = rows;
    for (u64 b0 = 0; b0 < rows; b0 += batch_tile) {
        const u64 nb = rows - b0 < batch_tile ? rows - b0 : batch_tile;
        a.in = in + b0 * nsrc; a.out = out + b0 * n; a.batch = nb;
        hipError_t e = hipErrorInvalidValue;
        switch (LA) {
            case 6: e = launch_strided<6, 128, false, true, true>(a, st); break;
            case 7: e = launch_strided<7, 64, false, true, true>(a, st); break;
            case 8: e = launch_strided<8, 32, false, true, true>(a, st); break;
        }
        if (e != hipSuccess) return e;
        a.in = out + b0 * n;
        if ((e = fwd_contig_dispatch(LB, true, 1, a, st)) != hipSuccess) return e;
    }
    return hipSuccess;
}

// in2 != nullptr: transform the pointwise product in .* in2 (and write it to
// evals_out when that is non-null).
hipError_t launch_ntt_inverse(const DevicePlan &p, const u64 *in, const u64 *in2, u64 *evals_out,
                              u64 *out, u64 batch, u64 batch_tile, hipStream_t st) {
    if (p.arith == kArStrict63 && (p.log_n < 4 || g63_plain())) return launch_g63_inverse(p, in, in2, evals_out, out, batch, st);
    PassArgs a{};
    const int L = p.log_n;
    // q = 1 (mod 2^32): a plain inverse transform (no product in its load) runs on the word-Montgomery table (AR = 4)
    const bool mg = L >= 4 && p.tw_inv_mg != nullptr && p.arith == kArWide61 && in2 == nullptr;
    const int ar = L < 4 ? (p.wide ? 1 : 0) : mg ? (int)kArMontgomery : p.arith;
    DevicePlan pt = p;
    pt.arith = mg ? (int)kArWide61 : ar;
    set_tables(a, pt, true);
    if (mg) { a.tw = p.tw_inv_mg; a.ninv = p.ninv_mg; a.s_ninv = p.s_ninv_mg; }
    if (batch == 0) return hipSuccess;
    if (L < 4) {
        const u64 *src = in;
        if (in2) {  // tiny sizes: unfused pointwise into evals_out (or out) first
            u64 *dst = evals_out ? evals_out : out;
            hipError_t e = launch_pointwise_mul(p, in, in2, dst, batch << L, st);
            if (e != hipSuccess) return e;
            src = dst;
        }
        a.in = src; a.out = out; a.batch = batch;
        KernelTimer kt("ntt_tiny_inv", L, st);
        hipLaunchKernelGGL(ntt_tiny_kernel<true>, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, a);
        return post_launch();
    }
    if (L <= kMaxSinglePassLog) {
        a.in = in; a.in2 = in2; a.out2 = evals_out; a.out = out; a.batch = batch;
        return inv_contig_dispatch(ar, L, true, in2 != nullptr, a, st);
    }
    const int LB = contig_bits(L), LA = L - LB;
    const u64 n = 1ull << L;
    if (batch_tile == 0) batch_tile = batch;
    for (u64 b0 = 0; b0 < batch; b0 += batch_tile) {
        const u64 nb = batch - b0 < batch_tile ? batch - b0 : batch_tile;
        a.in = in + b0 * n; a.in2 = in2 ? in2 + b0 * n : nullptr;
        a.out2 = evals_out ? evals_out + b0 * n : nullptr;
        a.out = out + b0 * n; a.batch = nb;
        hipError_t e = inv_contig_dispatch(ar, LB, false, in2 != nullptr, a, st);
        if (e != hipSuccess) return e;
        a.in = out + b0 * n; a.in2 = nullptr; a.out2 = nullptr;
        e = strided_dispatch<true>(ar, LA, a, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// the CONTIGUOUS pass of a two-pass inverse transform alone (n >= 2^14): leaves the lazy intermediate (< 4q for
// q < 2^61, < 2q otherwise) for a strided last pass the caller runs itself (zring.hip fuses its epilogue there)
hipError_t launch_ntt_inverse_first_pass(const DevicePlan &p, const u64 *in, u64 *out, u64 batch, hipStream_t st) {
    const int L = p.log_n;
    if (L <= kMaxSinglePassLog || L > kMaxLog || p.arith == kArStrict63) return hipErrorNotSupported;
    if (batch == 0) return hipSuccess;
    PassArgs a{};
    a.tw = p.tw_inv; a.mod = p.mod; a.ninv = p.ninv; a.s_ninv = p.s_ninv; a.log_n = p.log_n;
    a.in = in; a.out = out; a.batch = batch;
    // always on the Shoup tables: the caller's own last pass (zring.hip) continues from them
    return inv_contig_dispatch(p.wide ? 1 : 0, contig_bits(L), false, false, a, st);
}

static inline unsigned ew_grid(u64 count) {
    u64 g = (count + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;  // grid-stride beyond 16 blocks per CU
    return (unsigned)(g ? g : 1);
}

hipError_t launch_pointwise_mul(const DevicePlan &p, const u64 *x, const u64 *y, u64 *z, u64 count,
                                hipStream_t st) {
    if (p.arith == kArStrict63) return launch_g63_pointwise(p, x, y, z, count, st);
    if (count == 0) return hipSuccess;
    KernelTimer kt("pointwise_mul", 0, st);
    hipLaunchKernelGGL(pointwise_mul_kernel, dim3(ew_grid(count)), dim3(256), 0, st, x, y, z, count, p.mod);
    return post_launch();
}

hipError_t launch_fill_synthetic(u64 *out, u64 count, u64 q, u64 seed, u64 first, hipStream_t st) {
    if (count == 0) return hipSuccess;
    KernelTimer kt("fill_synthetic", 0, st);
    hipLaunchKernelGGL(fill_synthetic_kernel, dim3(ew_grid(count)), dim3(256), 0, st, out, count, q, seed, first);
    return post_launch();
}

hipError_t launch_check_canonical(const u64 *x, u64 count, u64 q, int *d_flag, hipStream_t st) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(check_canonical_kernel, dim3(ew_grid(count)), dim3(256), 0, st, x, count, q, d_flag);
    return post_launch();
}

// timing-only build (memory pattern without the arithmetic): wrong words by design, fhe_ntt_version() says so (capi.hip)
bool ntt_kernels_ablated() {
#if defined(FHE_ABLATE_NO_BUTTERFLIES)
    return true;
#else
    return false;
#endif
}

#endif   // FHE_NTT_TU_Q62

}  // namespace fhe
