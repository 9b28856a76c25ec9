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
#include <type_traits>

#include "ntt_kernels.hpp"
#include "zq_device.hpp"

#include <cstdlib>

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
constexpr int fwd_stage_needs_csub(int bound_in) { return bound_in > 6; }
constexpr int fwd_bound_out(int R, int bin) {
    int b = bin;
    for (int i = 0; i < R; i++) b = (fwd_stage_needs_csub(b) ? 4 : b) + 2;
    return b;
}
constexpr int kPassBound = 6;   // bound (in q) of what a forward strided pass hands to the contiguous pass
template <int R, bool WIDE, int BIN = 6, bool TIGHT_LAST = false, bool END6 = false>
__device__ __forceinline__ void round_fwd(u64 (&v)[16], const Tw *__restrict__ tw, u32 T0,
                                          const Mod &m) {
    static_assert(BIN >= 1 && BIN <= 8, "input bound out of range");
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int span = 8 >> i;
        constexpr int kNone = 0;
        const int bin_i = fwd_bound_out(i, BIN);                 // bound of this stage's inputs
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
                if (!WIDE) ct_bfly<2>(v[k], v[k + span], t.w, t.wp, m);
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

// WIDE: the bound-tracking rounds above (values below 4q between rounds); otherwise [0,2q) throughout
template <int R, bool FOLD, bool WIDE, int BIN>
__device__ __forceinline__ void round_inv_sel(u64 (&v)[16], const Tw *__restrict__ tw, u32 T0, const Mod &m,
                                              const Tw ninv, const Tw s_ninv) {
    if constexpr (WIDE) round_inv_w<R, FOLD, BIN, 4>(v, tw, T0, m, ninv, s_ninv);
    else round_inv<R, FOLD>(v, tw, T0, m, ninv, s_ninv);
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

// ---------------------------------------------------------------------------
// CONTIGUOUS pass: blocks of M = 2^LP consecutive coefficients.
// Workgroup = W units (unit = one M-block of one polynomial, all W units share
// `blk`, hence the twiddles), TPB = M/16 threads per unit.
// ---------------------------------------------------------------------------
template <int LP>
struct ContigCfg {
    static constexpr int M = 1 << LP;
    static constexpr int TPB = M / 16;
    static constexpr int TH = (LP <= 12) ? 256 : 512;
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
    // window base of round j >= 1
    static constexpr int a_of(int j) { return j == 0 ? A0 : LP - R0 - 4 * j; }
    static constexpr int ls0_of(int j) { return j == 0 ? 0 : R0 + 4 * (j - 1); }
    static constexpr bool in_lds(int j) { return ls0_of(j) + (j == 0 ? R0 : 4) <= LTW_LOG; }
};

// scatter registers (window AF) -> barrier -> gather registers (window AT).
// FIRST = false: the tile was read by an earlier exchange, so a barrier precedes the scatter.
// No trailing barrier: whoever writes the tile next either is this function (FIRST = false)
// or writes exactly the slots it has just gathered (the store transpose of the forward pass).
template <int LP, int AF, int AT, bool FIRST>
__device__ __forceinline__ void exchange_contig(u64 (&v)[16], u64 *lds, u32 w, u32 tf) {
    constexpr int M = 1 << LP;
    if (!FIRST) __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) lds[pad16(w * M + field_of<AF>(tf, k))] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = lds[pad16(w * M + field_of<AT>(tf, k))];
}

// SRC_DIGITS (single-pass sizes only): the input is `batch / digit_l` rows of 64-bit words and
// output polynomial p is the transform of bit digit_l-1-(p % digit_l) of row p / digit_l — the
// gadget decomposition of ring_torus.rs:67-77 / torus.rs:43-52 done in the load, so the 0/1
// polynomials never exist in memory.
// SRC_REDUCE (single-pass sizes only; the two-pass sizes do it in the strided pass): the input
// rows are 2^src_log_n arbitrary 64-bit words, reduced mod q and zero-padded to n in the load.
template <int LP, bool FINAL, bool WIDE, int SRC = SRC_PLAIN>
__global__ __launch_bounds__(ContigCfg<LP>::TH) void ntt_fwd_contig_kernel(PassArgs a) {
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
    // twiddle source and index of round j (local stage ls0, high field bits H)
    auto TW = [&](bool lds_round) -> const Tw * { return lds_round ? ltw : a.tw; };
    auto T0 = [&](bool lds_round, int ls, u32 H) -> u32 {
        return lds_round ? (1u << ls) + H : (1u << (s0 + ls)) + (blk << ls) + H;
    };

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
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = ld_at<u64>(pin, off + field_of<C::A0>(tf, k) * 8u);
    }
    // Round 0's twiddles are the same for the whole workgroup (H = 0): read from the global
    // table at a wave-uniform address (scalar loads, SGPR operands).  The LDS copy is only
    // needed from round 1 on, so the barrier of the first exchange also publishes it.
    stage_twiddles<C::LTW_N, C::TH>(ltw, a.tw, s0, blk, tid);

    // bounds (in q) entering each round; inputs are canonical or come from a strided pass (< 6q).
    // The transform's very last stage (FINAL) brings x below 2q: outputs < 4q.
    constexpr int B0 = kPassBound, B1 = fwd_bound_out(C::R0, B0), B2 = fwd_bound_out(4, B1), B3 = fwd_bound_out(4, B2);
    round_fwd<C::R0, WIDE, B0, FINAL && C::NR == 1>(v, a.tw, (1u << s0) + blk, m);
    if constexpr (C::NR > 1) {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        constexpr bool L = C::in_lds(1);
        exchange_contig<LP, C::A0, A, true>(v, lds, w, tf);
        round_fwd<4, WIDE, B1, FINAL && C::NR == 2>(v, TW(L), T0(L, LS, tf >> A), m);
    }
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        constexpr bool L = C::in_lds(2);
        exchange_contig<LP, C::a_of(1), A, false>(v, lds, w, tf);
        round_fwd<4, WIDE, B2, FINAL && C::NR == 3>(v, TW(L), T0(L, LS, tf >> A), m);
    }
    if constexpr (C::NR > 3) {
        constexpr int A = C::a_of(3), LS = C::ls0_of(3);
        constexpr bool L = C::in_lds(3);
        exchange_contig<LP, C::a_of(2), A, false>(v, lds, w, tf);
        round_fwd<4, WIDE, B3, FINAL && C::NR == 4>(v, TW(L), T0(L, LS, tf >> A), m);
    }
    // transpose through LDS so the store is one contiguous slab per wave.  (Storing the 128
    // contiguous bytes a thread owns after the last round as 8 x 16 B straight from registers
    // was measured slower for the forward kernel: 4.91 ms vs 4.40 ms per 16384 polynomials;
    // the mirrored direct 16-byte LOADS of the inverse kernel are faster: 4.69 vs 5.79 ms.)
    constexpr int ALAST = C::a_of(C::NR - 1);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u64 x = FINAL ? canon4(v[k], m) : v[k];   // FINAL: < 4q in both modes
        lds[pad16(w * C::M + field_of<ALAST>(tf, k))] = x;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const u32 e = i * C::TH + tid;
        const u32 wu = e >> LP, f = e & (C::M - 1);
        if (wu < live) st_at(pout, ((wu << a.log_n) + f) * 8u, lds[pad16(e)]);
    }
}

// MUL_IN: the input is the pointwise product in .* in2 (fused
// zip_eq(l,r).map(l*r), ring_nq.rs:601-604); if a.out2 != nullptr the product
// (the `evals` of the result, ring_nq.rs:606) is also written there.
template <int LP, bool FINAL, bool MUL_IN, bool WIDE>
__global__ __launch_bounds__(ContigCfg<LP>::TH) void ntt_inv_contig_kernel(PassArgs a) {
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
    auto TW = [&](bool lds_round) -> const Tw * { return lds_round ? ltw : a.tw; };
    auto T0 = [&](bool lds_round, int ls, u32 H) -> u32 {
        return lds_round ? (1u << ls) + H : (1u << (s0 + ls)) + (blk << ls) + H;
    };
    stage_twiddles<C::LTW_N, C::TH>(ltw, a.tw, s0, blk, tid);  // published by the barrier below

    // first window = field bits [0,4): a thread's 16 coefficients are 128 contiguous bytes,
    // fetched as 8 x 16 B straight into registers (no staging through LDS)
    constexpr int ALAST = C::a_of(C::NR - 1);
    static_assert(ALAST == 0, "first inverse window is the low 4 bits");
    u64 v[16];
    {
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
                p.x = mul_mod_var(v[2 * j], y.x, a.mod);
                p.y = mul_mod_var(v[2 * j + 1], y.y, a.mod);
                v[2 * j] = p.x;
                v[2 * j + 1] = p.y;
                if (a.out2 && active) st_c<ulonglong2>(pout2, g0 + j * 16u, p);
            }
        }
    }

    __syncthreads();
    // inputs are canonical (evals, or their product): the first round that runs starts from bound 2,
    // later ones from the normalised 4 (WIDE); a non-FINAL pass hands values below 4q (WIDE) / 2q on
    constexpr int BF = 2, BN = 4;
    if constexpr (C::NR > 3) {
        constexpr int A = C::a_of(3), LS = C::ls0_of(3);
        constexpr bool L = C::in_lds(3);
        round_inv_sel<4, false, WIDE, BF>(v, TW(L), T0(L, LS, tf >> A), m, a.ninv, a.s_ninv);
        exchange_contig<LP, A, C::a_of(2), true>(v, lds, w, tf);
    }
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        constexpr bool L = C::in_lds(2);
        round_inv_sel<4, false, WIDE, (C::NR == 3 ? BF : BN)>(v, TW(L), T0(L, LS, tf >> A), m, a.ninv, a.s_ninv);
        exchange_contig<LP, A, C::a_of(1), (C::NR <= 3)>(v, lds, w, tf);
    }
    if constexpr (C::NR > 1) {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        constexpr bool L = C::in_lds(1);
        round_inv_sel<4, false, WIDE, (C::NR == 2 ? BF : BN)>(v, TW(L), T0(L, LS, tf >> A), m, a.ninv, a.s_ninv);
        exchange_contig<LP, A, C::A0, (C::NR <= 2)>(v, lds, w, tf);
    }
    // FINAL implies s0 == 0 (this pass holds the m = 1 stage)
    round_inv_sel<C::R0, FINAL, WIDE, (C::NR == 1 ? BF : BN)>(v, TW(C::in_lds(0)), T0(C::in_lds(0), 0, 0), m, a.ninv, a.s_ninv);

    if (active) {
#pragma unroll
        for (int k = 0; k < 16; k++)
            st_c<u64>(pout, off + field_of<C::A0>(tf, k) * 8u, FINAL ? canon2(v[k], m) : v[k]);
    }
}

// ---------------------------------------------------------------------------
// FUSED PRODUCT for single-pass sizes: c = intt(ntt(a) .* ntt(b)) (ring_nq.rs:586-607) with the
// polynomial resident in registers / LDS from the first load to the last store — one launch
// instead of three and 3 (+ evals) instead of 7 passes over memory.  The forward rounds end in the
// register window (field bits [0,4)) the inverse rounds start from, so nothing is rearranged
// between the transforms.  Operands flagged as evals skip their forward transform.
// ---------------------------------------------------------------------------
template <int LP, bool WIDE, bool TILE_FRESH>
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
    for (int k = 0; k < 16; k++) v[k] = canon4(v[k], m);   // the last stage left x', y' < 4q
}

template <int LP, bool WIDE>
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

template <int LP, bool WIDE>
__global__ __launch_bounds__(ContigCfg<LP>::TH) void rq_mul_fused_kernel(PassArgs a) {
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
    auto operand = [&](const u64 *__restrict__ src, bool is_evals, u64 (&v)[16], auto fresh) {
        const u64 *__restrict__ p = src + ubase;
        if (is_evals) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const ulonglong2 x = ld_c<ulonglong2>(p, off + tf * 128u + j * 16u);
                v[2 * j] = x.x;
                v[2 * j + 1] = x.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = ld_c<u64>(p, off + field_of<C::A0>(tf, k) * 8u);
            fwd_rounds_single<LP, WIDE, decltype(fresh)::value>(v, lds, ltw_f, a.tw, w, tf, m);
        }
    };
    auto store_evals = [&](u64 *dst, const u64 (&v)[16]) {
        if (!dst || !active) return;
        u64 *__restrict__ p = dst + ubase;
#pragma unroll
        for (int j = 0; j < 8; j++) st_c<ulonglong2>(p, off + tf * 128u + j * 16u, ulonglong2{v[2 * j], v[2 * j + 1]});
    };

    u64 va[16], vb[16];
    operand(a.in, a.flags & 1u, va, std::true_type{});
    store_evals(a.out3, va);
    operand(a.in2, a.flags & 2u, vb, std::false_type{});   // the tile may have been used by the first operand
    store_evals(a.out4, vb);
#pragma unroll
    for (int k = 0; k < 16; k++) va[k] = mul_mod_var(va[k], vb[k], m);   // zip_eq(l,r).map(l*r), ring_nq.rs:601-604
    store_evals(a.out2, va);
    if constexpr (C::NR == 1) { /* twiddles published above */ } else if (a.flags == 3u) __syncthreads();   // no forward exchange ran
    inv_rounds_single<LP, WIDE>(va, lds, ltw_i, a.tw_inv, w, tf, m, a.ninv, a.s_ninv);
    if (active) {
        u64 *__restrict__ pout = a.out + ubase;
#pragma unroll
        for (int k = 0; k < 16; k++) st_c<u64>(pout, off + field_of<C::A0>(tf, k) * 8u, canon2(va[k], m));
    }
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

// RSRC: the input rows are 2^src_log_n arbitrary words, reduced mod q and zero-padded in the load
// (see SRC_REDUCE above).
template <int LA, int CW, bool WIDE, bool RSRC = false>
__global__ __launch_bounds__((StridedCfg<LA, CW>::TH)) void ntt_fwd_strided_kernel(PassArgs a) {
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
    const u64 *__restrict__ pin = a.in + ubase;
    u64 *__restrict__ pout = a.out + ubase;
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
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = ld_at<u64>(pin, ((field_of<C::A0>(tf, k) << lb) + c) * 8u);
    }
    for (u32 li = tid; li < (u32)C::F; li += C::TH) ltw[li] = a.tw[li];   // first pass: s0 = 0, blk = 0

    // inputs are canonical (bound 2 leaves slack); the pass ends below kPassBound*q (END6)
    constexpr int B0 = 2, B1 = fwd_bound_out(C::R0, B0), B2 = fwd_bound_out(4, B1);
    static_assert(kPassBound == 6, "END6 ends a pass below 6q");
    round_fwd<C::R0, WIDE, B0, false, C::NR == 1>(v, a.tw, 1u, m);   // uniform twiddles: scalar loads from the global table
    if constexpr (C::NR > 1) {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        exchange_strided<CW, C::A0, A, true>(v, lds, c, tf);   // its barrier also publishes ltw
        round_fwd<4, WIDE, B1, false, C::NR == 2>(v, tw, (1u << LS) + (tf >> A), m);
    }
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        exchange_strided<CW, C::a_of(1), A, false>(v, lds, c, tf);
        round_fwd<4, WIDE, B2, false, C::NR == 3>(v, tw, (1u << LS) + (tf >> A), m);
    }
    constexpr int ALAST = C::a_of(C::NR - 1);
#pragma unroll
    for (int k = 0; k < 16; k++) st_at(pout, ((field_of<ALAST>(tf, k) << lb) + c) * 8u, v[k]);  // lazy: < 4q, or < 6q (WIDE)
}

template <int LA, int CW, bool WIDE>
__global__ __launch_bounds__((StridedCfg<LA, CW>::TH)) void ntt_inv_strided_kernel(PassArgs a) {
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
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = ld_at<u64>(pin, ((field_of<ALAST>(tf, k) << lb) + c) * 8u);  // < 2q
    for (u32 li = tid; li < (u32)C::F; li += C::TH) ltw[li] = a.tw[li];
    __syncthreads();

    // the contiguous pass before this one hands values below 4q (WIDE) / 2q: every round starts from 4
    if constexpr (C::NR > 2) {
        constexpr int A = C::a_of(2), LS = C::ls0_of(2);
        round_inv_sel<4, false, WIDE, 4>(v, tw, (1u << LS) + (tf >> A), m, a.ninv, a.s_ninv);
        exchange_strided<CW, A, C::a_of(1), true>(v, lds, c, tf);
    }
    if constexpr (C::NR > 1) {
        constexpr int A = C::a_of(1), LS = C::ls0_of(1);
        round_inv_sel<4, false, WIDE, 4>(v, tw, (1u << LS) + (tf >> A), m, a.ninv, a.s_ninv);
        exchange_strided<CW, A, C::A0, (C::NR <= 2)>(v, lds, c, tf);
    }
    round_inv_sel<C::R0, true, WIDE, 4>(v, tw, 1u, m, a.ninv, a.s_ninv);
#pragma unroll
    for (int k = 0; k < 16; k++) st_at(pout, ((field_of<C::A0>(tf, k) << lb) + c) * 8u, canon2(v[k], m));
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

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static inline hipError_t post_launch() { return hipGetLastError(); }

// dynamic LDS above 64 KiB must be opted into per kernel
static inline hipError_t allow_big_lds(const void *fn, size_t bytes) {
    if (bytes <= 65536) return hipSuccess;
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int LP, bool FINAL, bool WIDE, int SRC = SRC_PLAIN>
static hipError_t launch_fwd_contig(const PassArgs &a, hipStream_t st) {
    using C = ContigCfg<LP>;
    const u64 nb = 1ull << (a.log_n - LP);
    const u64 groups = (a.batch + C::W - 1) / C::W;
    const u64 grid = nb * groups;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    if (hipError_t e = allow_big_lds((const void *)ntt_fwd_contig_kernel<LP, FINAL, WIDE, SRC>, C::LDS_BYTES)) return e;
    KernelTimer kt(SRC == SRC_DIGITS ? "ntt_fwd_digits" : SRC == SRC_ZQBITS ? "ntt_fwd_zqbits" : SRC == SRC_REDUCE ? "ntt_fwd_reduce" : (FINAL ? "ntt_fwd_contig_final" : "ntt_fwd_contig"), LP, st);
    hipLaunchKernelGGL((ntt_fwd_contig_kernel<LP, FINAL, WIDE, SRC>), dim3((unsigned)grid), dim3(C::TH),
                       C::LDS_BYTES, st, a);
    return post_launch();
}

template <int LP, bool FINAL, bool MUL_IN, bool WIDE>
static hipError_t launch_inv_contig(const PassArgs &a, hipStream_t st) {
    using C = ContigCfg<LP>;
    const u64 nb = 1ull << (a.log_n - LP);
    const u64 groups = (a.batch + C::W - 1) / C::W;
    const u64 grid = nb * groups;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    if (hipError_t e = allow_big_lds((const void *)ntt_inv_contig_kernel<LP, FINAL, MUL_IN, WIDE>, C::LDS_BYTES)) return e;
    KernelTimer kt(MUL_IN ? "ntt_inv_contig_mul" : (FINAL ? "ntt_inv_contig_final" : "ntt_inv_contig"), LP, st);
    hipLaunchKernelGGL((ntt_inv_contig_kernel<LP, FINAL, MUL_IN, WIDE>), dim3((unsigned)grid),
                       dim3(C::TH), C::LDS_BYTES, st, a);
    return post_launch();
}

template <int LA, int CW, bool INV, bool WIDE, bool RSRC = false>
static hipError_t launch_strided(const PassArgs &a, hipStream_t st) {
    using C = StridedCfg<LA, CW>;
    const u64 ncg = (1ull << (a.log_n - LA)) / CW;
    const u64 grid = ncg * a.batch;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    if (hipError_t e = allow_big_lds(INV ? (const void *)ntt_inv_strided_kernel<LA, CW, WIDE>
                                         : (const void *)ntt_fwd_strided_kernel<LA, CW, WIDE, RSRC>, C::LDS_BYTES)) return e;
    KernelTimer kt(INV ? "ntt_inv_strided" : (RSRC ? "ntt_fwd_strided_reduce" : "ntt_fwd_strided"), LA, st);
    if (INV)
        hipLaunchKernelGGL((ntt_inv_strided_kernel<LA, CW, WIDE>), dim3((unsigned)grid), dim3(C::TH),
                           C::LDS_BYTES, st, a);
    else
        hipLaunchKernelGGL((ntt_fwd_strided_kernel<LA, CW, WIDE, RSRC>), dim3((unsigned)grid), dim3(C::TH),
                           C::LDS_BYTES, st, a);
    return post_launch();
}

#define CONTIG_CASES(X) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)

static hipError_t fwd_contig_dispatch(int lp, bool final, bool wide, const PassArgs &a, hipStream_t st) {
    switch (lp) {
#define X(LP_)                                                                                   \
    case LP_:                                                                                    \
        if (wide) return final ? launch_fwd_contig<LP_, true, true>(a, st) : launch_fwd_contig<LP_, false, true>(a, st); \
        return final ? launch_fwd_contig<LP_, true, false>(a, st) : launch_fwd_contig<LP_, false, false>(a, st);
        CONTIG_CASES(X)
#undef X
    }
    return hipErrorInvalidValue;
}

template <bool WIDE>
static hipError_t inv_contig_dispatch(int lp, bool final, bool mul_in, const PassArgs &a,
                                      hipStream_t st) {
    switch (lp) {
#define X(LP_)                                                                                   \
    case LP_:                                                                                    \
        if (final) return mul_in ? launch_inv_contig<LP_, true, true, WIDE>(a, st)               \
                                 : launch_inv_contig<LP_, true, false, WIDE>(a, st);             \
        return mul_in ? launch_inv_contig<LP_, false, true, WIDE>(a, st)                         \
                      : launch_inv_contig<LP_, false, false, WIDE>(a, st);
        CONTIG_CASES(X)
#undef X
    }
    return hipErrorInvalidValue;
}

template <bool INV, bool WIDE>
static hipError_t strided_dispatch(int la, const PassArgs &a, hipStream_t st) {
    switch (la) {
        case 6: return launch_strided<6, 128, INV, WIDE>(a, st);
        case 7: return launch_strided<7, 64, INV, WIDE>(a, st);
        case 8: return launch_strided<8, 32, INV, WIDE>(a, st);   // 16 / 64 columns measured equal / slower
    }
    return hipErrorInvalidValue;
}

// pass split for n >= 2^14: LB = max(8, L-8) contiguous stages, LA = L-LB in 6..8
static inline int contig_bits(int L) { return L <= kMaxSinglePassLog ? L : (L - 8 > 8 ? L - 8 : 8); }

hipError_t launch_ntt_forward(const DevicePlan &p, const u64 *in, u64 *out, u64 batch,
                              u64 batch_tile, hipStream_t st) {
    PassArgs a{};
    a.tw = p.tw_fwd;
    a.mod = p.mod;
    a.ninv = p.ninv;
    a.s_ninv = p.s_ninv;
    a.log_n = p.log_n;
    const int L = p.log_n;
    if (batch == 0) return hipSuccess;
    if (L < 4) {
        a.in = in; a.out = out; a.batch = batch;
        KernelTimer kt("ntt_tiny_fwd", L, st);
        hipLaunchKernelGGL(ntt_tiny_kernel<false>, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, a);
        return post_launch();
    }
    if (L <= kMaxSinglePassLog) {
        a.in = in; a.out = out; a.batch = batch;
        return fwd_contig_dispatch(L, true, p.wide, a, st);
    }
    const int LB = contig_bits(L), LA = L - LB;
    const u64 n = 1ull << L;
    if (batch_tile == 0) batch_tile = batch;
    for (u64 b0 = 0; b0 < batch; b0 += batch_tile) {
        const u64 nb = batch - b0 < batch_tile ? batch - b0 : batch_tile;
        a.in = in + b0 * n; a.out = out + b0 * n; a.batch = nb;
        hipError_t e = p.wide ? strided_dispatch<false, true>(LA, a, st) : strided_dispatch<false, false>(LA, a, st);
        if (e != hipSuccess) return e;
        a.in = out + b0 * n;
        e = fwd_contig_dispatch(LB, true, p.wide, a, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <int LP, bool WIDE>
static hipError_t launch_rq_mul_fused_lp(const PassArgs &a, hipStream_t st) {
    using C = ContigCfg<LP>;
    const u64 grid = (a.batch + C::W - 1) / C::W;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    constexpr size_t lds_bytes = C::LDS_BYTES + (size_t)C::LTW_N * sizeof(Tw);   // a second twiddle tile
    if (hipError_t e = allow_big_lds((const void *)rq_mul_fused_kernel<LP, WIDE>, lds_bytes)) return e;
    KernelTimer kt("rq_mul_fused", LP, st);
    hipLaunchKernelGGL((rq_mul_fused_kernel<LP, WIDE>), dim3((unsigned)grid), dim3(C::TH), lds_bytes, st, a);
    return post_launch();
}

hipError_t launch_rq_mul_fused(const DevicePlan &p, const u64 *a_, bool a_is_evals, const u64 *b_, bool b_is_evals,
                               u64 *c, u64 *c_evals, u64 *a_evals, u64 *b_evals, u64 batch, hipStream_t st) {
    const int L = p.log_n;
    if (L < 4 || L > kMaxSinglePassLog) return hipErrorNotSupported;
    if (batch == 0) return hipSuccess;
    PassArgs a{};
    a.tw = p.tw_fwd;
    a.tw_inv = p.tw_inv;
    a.mod = p.mod;
    a.ninv = p.ninv;
    a.s_ninv = p.s_ninv;
    a.log_n = p.log_n;
    a.in = a_; a.in2 = b_; a.out = c; a.out2 = c_evals; a.out3 = a_evals; a.out4 = b_evals;
    a.flags = (a_is_evals ? 1u : 0u) | (b_is_evals ? 2u : 0u);
    a.batch = batch;
    switch (L) {
#define X(LP_) case LP_: return p.wide ? launch_rq_mul_fused_lp<LP_, true>(a, st) : launch_rq_mul_fused_lp<LP_, false>(a, st);
        CONTIG_CASES(X)
#undef X
    }
    return hipErrorInvalidValue;
}

hipError_t launch_ntt_forward_digits(const DevicePlan &p, const u64 *in, u64 *out, u64 rows, uint32_t l,
                                     hipStream_t st) {
    const int L = p.log_n;
    if (L < 4 || L > kMaxSinglePassLog || !p.wide || l == 0 || l > 64) return hipErrorNotSupported;
    if (rows == 0) return hipSuccess;
    PassArgs a{};
    a.tw = p.tw_fwd;
    a.mod = p.mod;
    a.log_n = p.log_n;
    a.in = in; a.out = out; a.batch = rows * l; a.digit_l = l;
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
    if (L < 4 || L > kMaxSinglePassLog || !p.wide || l == 0 || l > 64 || grp == 0 || p.mod.q < 3) return hipErrorNotSupported;
    if (rows == 0) return hipSuccess;
    PassArgs a{};
    a.tw = p.tw_fwd;
    a.mod = p.mod;
    a.log_n = p.log_n;
    a.in = in; a.out = out; a.batch = rows * l; a.digit_l = l; a.src_grp = grp; a.src_gstride = gstride;
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
        if ((e = fwd_contig_dispatch(LB, true, true, a, st)) != hipSuccess) return e;
    }
    return hipSuccess;
}

// in2 != nullptr: transform the pointwise product in .* in2 (and write it to
// evals_out when that is non-null).
hipError_t launch_ntt_inverse(const DevicePlan &p, const u64 *in, const u64 *in2, u64 *evals_out,
                              u64 *out, u64 batch, u64 batch_tile, hipStream_t st) {
    PassArgs a{};
    a.tw = p.tw_inv;
    a.mod = p.mod;
    a.ninv = p.ninv;
    a.s_ninv = p.s_ninv;
    a.log_n = p.log_n;
    const int L = p.log_n;
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
        return p.wide ? inv_contig_dispatch<true>(L, true, in2 != nullptr, a, st)
                      : inv_contig_dispatch<false>(L, true, in2 != nullptr, a, st);
    }
    const int LB = contig_bits(L), LA = L - LB;
    const u64 n = 1ull << L;
    if (batch_tile == 0) batch_tile = batch;
    for (u64 b0 = 0; b0 < batch; b0 += batch_tile) {
        const u64 nb = batch - b0 < batch_tile ? batch - b0 : batch_tile;
        a.in = in + b0 * n; a.in2 = in2 ? in2 + b0 * n : nullptr;
        a.out2 = evals_out ? evals_out + b0 * n : nullptr;
        a.out = out + b0 * n; a.batch = nb;
        hipError_t e = p.wide ? inv_contig_dispatch<true>(LB, false, in2 != nullptr, a, st)
                              : inv_contig_dispatch<false>(LB, false, in2 != nullptr, a, st);
        if (e != hipSuccess) return e;
        a.in = out + b0 * n; a.in2 = nullptr; a.out2 = nullptr;
        e = p.wide ? strided_dispatch<true, true>(LA, a, st) : strided_dispatch<true, false>(LA, a, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

static inline unsigned ew_grid(u64 count) {
    u64 g = (count + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;  // grid-stride beyond 16 blocks per CU
    return (unsigned)(g ? g : 1);
}

hipError_t launch_pointwise_mul(const DevicePlan &p, const u64 *x, const u64 *y, u64 *z, u64 count,
                                hipStream_t st) {
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

}  // namespace fhe
