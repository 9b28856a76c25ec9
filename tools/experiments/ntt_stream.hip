// ---------------------------------------------------------------------------------------------------------------------
// EXPERIMENT RECORD (round 5) — NOT built into libfhe_ntt.so, NOT part of the product.  Kept because the measurements
// below are quoted in DESIGN.md section 5 and the next attempt should start from what this one learned.
//   * what it is: both passes of the n = 2^16 forward transform as persistent ("streaming") kernels that always have the
//     next tile's sixteen loads per lane in flight.  Bit-exact (eight batch shapes against the plain kernels, every word).
//   * ceiling: the plain passes WITHOUT their butterflies run 346 + 350 us per 2048 polynomials, with them 359 + 366:
//     at most 4 % to win (gpurun_out/r5d/kbench_nobfly.txt).
//   * first version: 447 + 493 us (gpurun_out/r5e).  vmcnt counts vector-memory operations in ISSUE order: (i) round 0's
//     uniform twiddles, read inside the loop after stores the compiler cannot tell apart from the table, became vector
//     loads queued BEHIND the prefetch; (ii) a conditional prefetch makes the compiler wait for the minimum over all
//     paths, i.e. for the loads just issued; (iii) one scratch reload in the loop drains the queue the same way.
//   * this version fixes (i) and (ii) — the steady loop waits with vmcnt(39..32), exactly the loads it needs — at the
//     price of 60 scalar registers of resident twiddles: 98 / 36 spilled SGPRs (88 v_readlane per tile in the strided
//     kernel, +9 % vector instructions) and, in the contiguous kernel, 128 VGPRs with three spilled (each reload is a
//     vmcnt(0) again).  The remaining fixes (stage 3 of round 0 on LDS-broadcast twiddles, the store slab in the freed
//     register set) cost about what the 4 % ceiling offers; not pursued.
// To try it again: add the file to fhe-study_amd/binding.py SOURCES, declare launch_ntt_forward_stream in ntt_kernels.hpp
// and call it at the top of the tile loop of launch_ntt_forward (ntt_kernels.hip).
// ---------------------------------------------------------------------------------------------------------------------
// ntt_stream.hip — the two passes of a large forward transform (NTT::ntt, arith/src/ntt.rs:44-73, n >= 2^16) as
// STREAMING kernels: as many workgroups as the chip holds, each walking its share of the launch's tiles with the NEXT
// tile's sixteen loads per lane in flight while the current tile runs its stages (round 5).
//
// Why: ntt_kernels.hip launches one workgroup per tile; a workgroup has loads in flight only for the first fifth of its
// life, so what keeps HBM busy is four workgroups per CU taking turns.  The passes without their butterflies
// (-DFHE_ABLATE_NO_BUTTERFLIES, gpurun_out/r5d/kbench_nobfly.txt) run 346 / 350 us per 2048 polynomials, the real ones
// 359 / 366: the gap is load latency the butterflies do not cover.  Here every workgroup always has a tile's worth of
// loads outstanding, stages its twiddle tile ONCE (the strided pass's 2^LA twiddles do not depend on the tile; a
// contiguous workgroup keeps its block index and walks the polynomial groups), keeps round 0's scalar twiddles in
// SGPRs across tiles, and the launch has no per-tile ramp or tail.
// The two register sets take turns (the loop body is written twice) so that no tile is ever copied between them.
// Same arithmetic, same rounds (ntt_rounds.hpp), same words as ntt_fwd_strided_kernel / ntt_fwd_contig_kernel.
#include "../../fhe-study_amd/csrc/ntt_rounds.hpp"

#include <cstdlib>

namespace fhe {

// ---- strided pass: tiles = (polynomial, column group), walked round-robin over the grid -----------------------------
template <int LA, int CW, int AR>
__global__ __launch_bounds__((StridedCfg<LA, CW>::TH), 4) void ntt_fwd_strided_stream_kernel(PassArgs a) {
    static_assert(AR == 2 || AR == 4, "pseudo-Mersenne or word-Montgomery tables");
    using C = StridedCfg<LA, CW>;
    static_assert(C::NR == 2, "6 <= LA <= 8: two rounds, one exchange");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const u32 tid = threadIdx.x, c = tid % CW, tf = tid / CW;
    const u32 lb = a.log_n - LA;               // log2 of the row length
    const u32 lcg = lb - __builtin_ctz(CW);    // log2(column groups per polynomial)
    const u64 ntiles = a.batch << lcg;
    const Mod &m = a.mod;
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + C::DATA_BYTES);
    for (u32 li = tid; li < (u32)C::F; li += C::TH) ltw[li] = a.tw[li];   // first pass: s0 = 0, blk = 0 — once per workgroup

    constexpr int AK = AR == 4 ? 4 : 2;
    constexpr int P0 = kPmOne, P1 = pm_fwd_bound_out(C::R0, P0, AK), P2 = pm_fwd_bound_out(4, P1, AK);
    static_assert(P2 <= kPmPassBound, "a strided pass ends below kPmPassBound");
    constexpr int A1 = C::a_of(1), LS1 = C::ls0_of(1);
    const u32 lane_in = ((field_of<C::A0>(tf, 0) << lb) + c) * 8u, lane_out = ((field_of<A1>(tf, 0) << lb) + c) * 8u;
    const u32 cgmask = (1u << lcg) - 1u;
    auto tile_base = [&](u64 t) -> u64 { return ((t >> lcg) << a.log_n) + (u64)((u32)t & cgmask) * CW; };
    // Round 0's twiddles (entries 1 .. 2^R0 - 1, the same for every tile) are read HERE, before the kernel has stored
    // anything: scalar loads.  Read inside the loop they come after stores the compiler cannot tell apart from the table,
    // become vector loads, and — vmcnt counts in issue order — every wave would wait for its prefetch before its first
    // butterfly (measured: 447 instead of 367 us per 2048 polynomials).
    Tw r0[1 << C::R0];
#pragma unroll
    for (int i = 1; i < (1 << C::R0); i++) r0[i] = a.tw[i];
    auto fetch = [&](u64 (&v)[16], u64 t) { ld16<true, true>(v, a.in + tile_base(t), C::A0 + lb, lane_in); };
    auto run = [&](u64 (&v)[16], u64 t) {
        round_fwd_pm<C::R0, P0, true, AK>(v, r0, 1u, m);            // uniform twiddles, in scalar registers across tiles
        exchange_strided<CW, C::A0, A1, false>(v, lds, c, tf);       // leading barrier: the previous tile's gather is over (first tile: publishes ltw)
        round_fwd_pm<4, P1, false, AK>(v, ltw, (1u << LS1) + (tf >> A1), m);
        st16<true, true>(a.out + tile_base(t), A1 + lb, lane_out, v);   // lazy: below kPmPassBound
    };

    // Tiles of this workgroup: t_j = blockIdx.x + j G, j < cnt.  The loop is shaped so that the queue of vector-memory
    // operations looks the same on every path into a wait — sixteen stores and sixteen prefetch loads are always younger
    // than the loads a tile waits for (vmcnt counts in issue order, and the compiler takes the minimum over all paths: a
    // conditional prefetch made every tile wait for the loads just issued).  So the prefetch is unconditional — past the
    // last tile it re-reads tile 0, an L2 hit that is never used — and the first tile is peeled.
    const u64 G = gridDim.x, t0 = blockIdx.x;
    if (t0 >= ntiles) return;
    const u64 cnt = (ntiles - t0 + G - 1) / G;
    auto tile_of = [&](u64 j) -> u64 { return j < cnt ? t0 + j * G : 0; };
    u64 va[16], vb[16];
    fetch(va, tile_of(0));
    fetch(vb, tile_of(1));
    run(va, tile_of(0));
    for (u64 j = 1; j < cnt; j += 2) {                               // vb holds tile j, va is free
        fetch(va, tile_of(j + 1));
        run(vb, tile_of(j));
        if (j + 1 >= cnt) break;
        fetch(vb, tile_of(j + 2));
        run(va, tile_of(j + 1));
    }
}

// ---- contiguous pass: a workgroup keeps its block `blk` (hence its twiddle tile) and walks the polynomial groups ------
template <int LP, int AR>
__global__ __launch_bounds__(ContigCfg<LP>::TH, 4) void ntt_fwd_contig_stream_kernel(PassArgs a, u32 gq) {
    static_assert(AR == 2 || AR == 4, "pseudo-Mersenne or word-Montgomery tables");
    using C = ContigCfg<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const u32 tid = threadIdx.x, w = tid / C::TPB, tf = tid % C::TPB;
    const u32 s0 = a.log_n - LP;
    const u32 blk = blockIdx.x & ((1u << s0) - 1u);
    const u64 n = 1ull << a.log_n;
    const u64 groups = (a.batch + C::W - 1) / C::W;
    const Mod &m = a.mod;
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + C::DATA_BYTES);
    static_assert(LP == 8, "rounds after the first read the LDS tile only at LP = 8 (a global twiddle load inside the loop would queue behind the prefetch)");
    // round 0's twiddles, workgroup-uniform and the same for every group this workgroup walks: scalar loads, once (see the
    // strided kernel); indexed like round_fwd_pm does with T0 = 2^s0 + blk, and read before stage_twiddles' divergent
    // branch (values the two share would otherwise be merged there and count as divergent: vector loads, spilled)
    Tw r0[1 << C::R0];
    {
        const u32 T0 = (u32)__builtin_amdgcn_readfirstlane((int)((1u << s0) + blk));
#pragma unroll
        for (int i = 0; i < C::R0; i++)
#pragma unroll
            for (int g = 0; g < (1 << i); g++) r0[(1 << i) + g] = a.tw[(T0 << i) + g];
    }
    stage_twiddles<C::LTW_N, C::TH>(ltw, a.tw, s0, blk, tid);       // once: published by the first exchange's barriers
    constexpr int ALAST = C::a_of(C::NR - 1);
    const SlabIo<LP> io(tid, a.log_n);

    auto live_of = [&](u64 pg) -> u32 { return (u32)min((u64)C::W, a.batch - pg * C::W); };
    auto fetch = [&](u64 (&v)[16], u64 pg) {
        // lanes of a missing polynomial (ragged last group) transform a copy of the group's first one and store nothing
        const u32 off = ((w < live_of(pg) ? w : 0u) << a.log_n) * 8u;
        ld16<true, true>(v, a.in + pg * C::W * n + (u64)blk * C::M, C::A0, off + tf * 8u);
    };
    auto run = [&](u64 (&v)[16], u64 pg) {
        const u32 live = live_of(pg);
        u64 *pout = a.out + pg * C::W * n + (u64)blk * C::M;
        // the pass as fwd_rounds_contig_pm runs it at LP = 8, round 0 on the preloaded twiddles (indexed as a transform of its own: T0 = 1)
        constexpr int B0 = kPmPassBound, B1 = pm_fwd_bound_out(C::R0, B0, AR);
        round_fwd_pm<C::R0, B0, true, AR>(v, r0, 1u, m);
        exchange_contig<LP, C::A0, C::a_of(1), false>(v, lds, w, tf);   // leading barrier: the previous group's slab reads are over
        round_fwd_pm<4, B1, false, AR>(v, ltw, (1u << C::ls0_of(1)) + (tf >> C::a_of(1)), m);
        // (no barrier: the store transpose writes exactly the slots this thread has just gathered)
#pragma unroll
        for (int k = 0; k < 16; k++) lds[pad16(w * C::M + field_of<ALAST>(tf, k))] = AR == 2 ? pm_canon(v[k], m) : canon8(v[k], m);
        __syncthreads();
        if (live == (u32)C::W) {
            u32 boff = io.boff;
            asm volatile("" : "+v"(boff));                          // (loop-invariant lane offset: see ld16)
#pragma unroll
            for (int h = 0; h < 2; h++) {                            // two halves: eight LDS reads in flight, sixteen registers
                u64 t[8];
#pragma unroll
                for (int i = 0; i < 8; i++) t[i] = lds[io.slot + (8 * h + i) * io.lds_step];
#pragma unroll
                for (int i = 0; i < 8; i++) st_s<true>(io.base(pout, 8 * h + i, a.log_n), boff, t[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++)
                if (io.wu(i) < live) st_s<true>(io.base(pout, i, a.log_n), io.boff, lds[io.slot + i * io.lds_step]);
        }
    };

    // groups of this workgroup: pg_j = pg0 + j gq, j < cnt; the loop is shaped like the strided kernel's (see there)
    const u64 pg0 = blockIdx.x >> s0;
    if (pg0 >= groups) return;
    const u64 cnt = (groups - pg0 + gq - 1) / gq;
    auto group_of = [&](u64 j) -> u64 { return j < cnt ? pg0 + j * gq : 0; };
    u64 va[16], vb[16];
    fetch(va, group_of(0));
    fetch(vb, group_of(1));
    run(va, group_of(0));
    for (u64 j = 1; j < cnt; j += 2) {
        fetch(va, group_of(j + 1));
        run(vb, group_of(j));
        if (j + 1 >= cnt) break;
        fetch(vb, group_of(j + 2));
        run(va, group_of(j + 1));
    }
}

// ---- launcher --------------------------------------------------------------------------------------------------------
static int stream_mode() {          // FHE_NTT_STREAM=0: the one-workgroup-per-tile kernels of ntt_kernels.hip
    static const int on = [] {
        const char *e = getenv("FHE_NTT_STREAM");
        return e ? atoi(e) : 1;
    }();
    return on;
}

// workgroups of `fn` the device holds at once (occupancy x CUs), cached per device
template <typename F>
static int resident_blocks(F fn, int threads, size_t lds_bytes, int *cache) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cache[dev] == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds_bytes) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cache[dev] = per_cu * cus > 0 ? per_cu * cus : -1;
    }
    return cache[dev] > 0 ? cache[dev] : 0;
}

template <int LA, int CW, int AR>
static hipError_t strided_stream(const PassArgs &a, hipStream_t st, bool launch) {
    using C = StridedCfg<LA, CW>;
    static int cache[64];
    const auto fn = ntt_fwd_strided_stream_kernel<LA, CW, AR>;
    if (hipError_t e = allow_big_lds((const void *)fn, C::LDS_BYTES)) return e;
    const int slots = resident_blocks(fn, C::TH, C::LDS_BYTES, cache);
    if (slots <= 0) return hipErrorNotSupported;
    if (!launch) return hipSuccess;
    const u64 ntiles = a.batch * ((1ull << (a.log_n - LA)) / CW);
    const u64 grid = ntiles < (u64)slots ? ntiles : (u64)slots;
    KernelTimer kt("ntt_fwd_strided_stream", LA, st);
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(C::TH), C::LDS_BYTES, st, a);
    return hipGetLastError();
}

template <int LP, int AR>
static hipError_t contig_stream(const PassArgs &a, hipStream_t st, bool launch) {
    using C = ContigCfg<LP>;
    static int cache[64];
    const auto fn = ntt_fwd_contig_stream_kernel<LP, AR>;
    if (hipError_t e = allow_big_lds((const void *)fn, C::LDS_BYTES)) return e;
    const int slots = resident_blocks(fn, C::TH, C::LDS_BYTES, cache);
    if (slots <= 0) return hipErrorNotSupported;
    if (!launch) return hipSuccess;
    const u64 nb = 1ull << (a.log_n - LP);
    const u64 groups = (a.batch + C::W - 1) / C::W;
    u64 gq = (u64)slots / nb;                 // polynomial groups in flight: every block index `blk` gets gq workgroups
    if (gq < 1) gq = 1;
    if (gq > groups) gq = groups;
    KernelTimer kt("ntt_fwd_contig_stream", LP, st);
    hipLaunchKernelGGL(fn, dim3((unsigned)(nb * gq)), dim3(C::TH), C::LDS_BYTES, st, a, (u32)gq);
    return hipGetLastError();
}

static hipError_t contig_stream_lb(int LB, int ar, const PassArgs &a, hipStream_t st, bool launch) {
    switch (LB) {
#define X(LP_) case LP_: return ar == 4 ? contig_stream<LP_, 4>(a, st, launch) : contig_stream<LP_, 2>(a, st, launch);
        X(8)
#undef X
    }
    return hipErrorNotSupported;   // LB > 8: later rounds read per-lane twiddles from the global table (see the kernel)
}

// The forward transform of `batch` polynomials (args.in -> args.out, which may be the same buffer), or
// hipErrorNotSupported — BEFORE anything has been launched — when these kernels do not apply (the caller then runs the
// one-workgroup-per-tile kernels): pseudo-Mersenne / word-Montgomery tables, 2^16 <= n <= 2^20, and enough tiles that
// every resident workgroup gets several.
hipError_t launch_ntt_forward_stream(const PassArgs &args, int ar, u64 batch, hipStream_t st) {
    if (!stream_mode() || (ar != 2 && ar != 4)) return hipErrorNotSupported;
    const int L = (int)args.log_n;
    if (L != 16) return hipErrorNotSupported;
    const int LB = contig_bits(L), LA = L - LB;
    if (LA != 8 || LB != 8) return hipErrorNotSupported;
    if (batch * ((1ull << LB) / 32) < 2048) return hipErrorNotSupported;   // fewer than ~4 tiles per workgroup: the plain kernels
    PassArgs a = args;
    a.batch = batch;
    // both kernels must be launchable before the first one overwrites anything (in == out is allowed)
    hipError_t e = ar == 4 ? strided_stream<8, 32, 4>(a, st, false) : strided_stream<8, 32, 2>(a, st, false);
    if (e != hipSuccess) return e;
    if ((e = contig_stream_lb(LB, ar, a, st, false)) != hipSuccess) return e;
    e = ar == 4 ? strided_stream<8, 32, 4>(a, st, true) : strided_stream<8, 32, 2>(a, st, true);
    if (e != hipSuccess) return e;
    a.in = a.out;
    return contig_stream_lb(LB, ar, a, st, true);
}

}  // namespace fhe
