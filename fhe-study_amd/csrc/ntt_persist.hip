// ntt_persist.hip — NTT::ntt (arith/src/ntt.rs:44-73) at n = 2^16 as ONE launch of persistent workgroups.
//
// The two-pass kernels (ntt_kernels.hip) move every coefficient through HBM twice: 32 n bytes per transform where the
// algorithm needs 16 n.  Here the intermediate between the strided stages (0..7) and the contiguous stages (8..15) of a
// polynomial is handed from workgroup to workgroup THROUGH THE L2 OF ONE XCD (a small ring of polynomial-sized slots per
// XCD, rewritten in place) or, for large tiles, through the output buffer while it is still in the Infinity Cache.
// Work order, dependencies and the control block: persist_sched.hpp.  What keeps it safe:
//   * a workgroup reads its XCD from the hardware (HW_REG_XCC_ID) and draws tickets from THAT queue only, so a producer
//     and its consumers always share an L2: plain stores, then s_waitcnt vmcnt(0) + barrier + one agent-scope counter add;
//     the consumer polls the counter (agent scope), barrier, then loads with sc1 (they bypass the CU's own L1, which is
//     never refreshed by another CU's stores);
//   * every wait is on an EARLIER ticket of the same queue and every spin is bounded: a wait that runs out sets the
//     error word and the workgroup leaves — a logic error is a failed call, never a hung GPU;
//   * no co-residency is assumed: the grid is whatever the chip holds, but any number of resident workgroups >= 1 works.
// Arithmetic: the pseudo-Mersenne butterflies of zq_device.hpp (AR = 2); other moduli stay on the two-pass kernels.
#include "ntt_persist.hpp"
#include "ntt_rounds.hpp"

namespace fhe {

namespace {

constexpr int kTH = 512;                       // threads of a persistent workgroup
constexpr int kUnits = 32;                     // 256-coefficient units of a C item (16 threads each)
constexpr u32 kSpinCap = 1u << 21;             // polls before a wait gives up (seconds)
constexpr size_t kTileBytes = (size_t)(kUnits * 256 + kUnits * 16) * 8;   // padded C tile (>= the 64 KiB S tile)
constexpr size_t kTw0Bytes = (size_t)kUnits * 15 * sizeof(Tw);             // stages 8..11 of a C item: 15 twiddles per block
constexpr size_t kLdsBytes = kTileBytes + 256 * sizeof(Tw) + kTw0Bytes + 64;   // + the strided stages' twiddles + control words

__device__ __forceinline__ u32 xcc_id() {
    u32 v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & (kPersistQueues - 1u);
}
__device__ __forceinline__ u32 ctl_add(u32 *p, u32 x) { return __hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 ctl_load(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ctl_store(u32 *p, u32 x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one lane: poll until *p >= target (or != 0 when target == 0 means "any"); false when the spin ran out
__device__ __forceinline__ bool wait_ge(const u32 *p, u32 target, u32 *got) {
    for (u32 it = 0; it < kSpinCap; it++) {
        const u32 v = ctl_load(p);
        if (v >= target) { *got = v; return true; }
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}
// coefficient loads of the intermediate: sc1 = served by the L2, never by this CU's L1
__device__ __forceinline__ u64 ld_mid(const u64 *base, u32 byte_off) {
    return __hip_atomic_load(reinterpret_cast<const u64 *>(reinterpret_cast<const unsigned char *>(base) + byte_off),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// R stages on the 16 registers with the twiddle of (stage i, group g) supplied by `twf`
template <int R, int BIN, typename F>
__device__ __forceinline__ void round_fwd_pm_f(u64 (&v)[16], F twf, const Mod &m) {
    static_assert(pm_fwd_bound_out(R, BIN) <= kPmCap, "a stage would overflow");
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int span = 8 >> i;
        const bool red = pm_fwd_needs_red(pm_fwd_bound_out(i, BIN));
#pragma unroll
        for (int g = 0; g < (1 << i); g++) {
            const Tw t = twf(i, g);
#pragma unroll
            for (int l = 0; l < span; l++) {
                const int k = g * 2 * span + l;
                if (red) v[k] = pm_reduce(v[k], m);
                ct_bfly_pm<false>(v[k], v[k + span], t.w, t.wp, m);
            }
        }
    }
}

// what lane 0 hands the workgroup for one ticket: five words in LDS
enum : int { kCtlPhase = 0, kCtlOrd = 1, kCtlR = 2, kCtlBind = 3, kCtlStatus = 4 };

}  // namespace

// MIDRING: the intermediate lives in a.ring (per-XCD slots, rewritten in place); otherwise in a.out
template <bool MIDRING>
__global__ __launch_bounds__(kTH, 4) void ntt_fwd_persist_kernel(PersistArgs a) {
    using S = StridedCfg<8, 32>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + kTileBytes);
    Tw *ltw0 = ltw + 256;                                                   // a C item's stage 8..11 twiddles [block][15]
    u32 *ctrl = reinterpret_cast<u32 *>(smem_raw + kTileBytes + 256 * sizeof(Tw) + kTw0Bytes);
    const u32 tid0 = threadIdx.x;
    const u32 xq = xcc_id();
    const Mod &m = a.mod;
    const u32 log_t = a.log_t, T = 1u << log_t, I = 8u << log_t;
    const u32 maxord = a.maxord;
    u32 *const ctl = a.ctl;
    u32 *const head = ctl + persist_ctl_head(xq);

    for (u32 li = tid0; li < 256u; li += kTH) ltw[li] = a.tw[li];   // the strided stages' twiddles, once per workgroup

    for (;;) {
        // opaque per iteration: the lane's index arithmetic (LDS slots, byte offsets) is then redone per item, as in the
        // two-pass kernels, instead of being hoisted out of the loop into ~30 registers that spill
        u32 tid = tid0;
        asm volatile("" : "+v"(tid));
        if (tid == 0) {
            const u64 k = ctl_add(head, 1u);
            const PersistItem it = persist_decode(k, log_t, a.lag);
            u32 status = 0, b = kPersistInvalid;
            // Ordinals at or past maxord cannot be bound to a tile (a queue binds at most ntiles of them, as a prefix): they are
            // the tickets workgroups draw on their way out, and touch no control word.
            if (it.ord < maxord) {
                u32 *bp = ctl + persist_ctl_bind(xq, it.ord, maxord);
                if (it.phase == kPersistS && it.r == 0) {
                    // bind this ordinal to the next global tile — after the previous ordinal of this queue has been bound, so
                    // that the bound ordinals of a queue are a PREFIX (a workgroup leaves at the first C ticket without a tile)
                    u32 prev = 1u;
                    if (it.ord > 0 && !wait_ge(bp - 1, 1u, &prev)) status = kPersistErrBind;
                    if (!status) {
                        if (prev != kPersistInvalid) {
                            const u32 g = ctl_add(ctl + persist_ctl_gtile(), 1u);
                            b = (u64)g < a.ntiles ? g + 1u : kPersistInvalid;
                        }
                        ctl_store(bp, b);
                    }
                } else if (!wait_ge(bp, 1u, &b)) {
                    status = kPersistErrBind;
                }
                if (!status && b != kPersistInvalid) {
                    u32 got;
                    if (it.phase == kPersistC) {
                        if (!wait_ge(ctl + persist_ctl_sdone(xq, it.ord, maxord), I, &got)) status = kPersistErrSdone;
                    } else if (MIDRING && it.ord >= a.ringslots) {
                        if (!wait_ge(ctl + persist_ctl_cdone(xq, it.ord - a.ringslots, maxord), I, &got)) status = kPersistErrCdone;
                    }
                }
            }
            if (status) {
                atomicOr(ctl + persist_ctl_err(), status);
                __hip_atomic_fetch_or(a.host_err, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // pinned host word
            }
            ctrl[kCtlPhase] = it.phase; ctrl[kCtlOrd] = it.ord; ctrl[kCtlR] = it.r; ctrl[kCtlBind] = b; ctrl[kCtlStatus] = status;
        }
        __syncthreads();
        const u32 phase = __builtin_amdgcn_readfirstlane(ctrl[kCtlPhase]), ord = __builtin_amdgcn_readfirstlane(ctrl[kCtlOrd]),
                  r = __builtin_amdgcn_readfirstlane(ctrl[kCtlR]), bind = __builtin_amdgcn_readfirstlane(ctrl[kCtlBind]),
                  status = __builtin_amdgcn_readfirstlane(ctrl[kCtlStatus]);
        if (status) return;
        if (bind == kPersistInvalid) {
            if (phase == kPersistC) return;     // the queue has run dry
            __syncthreads();                    // everybody has read ctrl before lane 0 rewrites it
            continue;
        }
        const u64 tile0 = (u64)(bind - 1u) << log_t;                         // first polynomial of the tile
        u64 *const ring_slot = MIDRING ? a.ring + ((((u64)xq * a.ringslots + ord % a.ringslots) << log_t) << 16) : nullptr;

        if (phase == kPersistS) {
            // ---- strided stages 0..7 of 32 columns of one polynomial (ntt_fwd_strided_kernel<8, 32>) ----
            const u32 pl = r >> 3, cg = r & 7u;
            const u64 poly = tile0 + pl;
            if (poly < a.batch) {
                const u32 c = tid % 32u, tf = tid / 32u;
                const u64 *__restrict__ pin = a.in + (poly << 16) + cg * 32u;
                u64 *__restrict__ pout = (MIDRING ? ring_slot + ((u64)pl << 16) : a.out + (poly << 16)) + cg * 32u;
                u64 v[16];
#pragma unroll
                for (int k = 0; k < 16; k++) v[k] = ld_at<u64>(pin, ((field_of<S::A0>(tf, k) << 8) + c) * 8u);
                constexpr int P0 = kPmOne, P1 = pm_fwd_bound_out(S::R0, P0);
                round_fwd_pm<S::R0, P0, true>(v, a.tw, 1u, m);
                exchange_strided<32, S::A0, S::a_of(1), true>(v, lds, c, tf);
                round_fwd_pm<4, P1, false>(v, ltw, (1u << S::ls0_of(1)) + (tf >> S::a_of(1)), m);
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const u32 off = ((field_of<S::a_of(1)>(tf, k) << 8) + c) * 8u;
                    if (MIDRING) st_c<u64>(pout, off, v[k]);     // stays in this XCD's L2 for its consumers
                    else st_at(pout, off, v[k]);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();                                      // every wave's stores have been acknowledged
            if (tid == 0) ctl_add(ctl + persist_ctl_sdone(xq, ord, maxord), 1u);
        } else {
            // ---- contiguous stages 8..15 of 32 units (256 coefficients each) of the tile ----
            const u32 log_pb = log_t < 5u ? log_t : 5u, pbm = (1u << log_pb) - 1u;     // polynomials per item
            const u32 pgrp = r & ((T >> log_pb) - 1u), bgrp = r >> (log_t - log_pb);
            const u32 u = tid >> 4, tf = tid & 15u;
            auto unit_poly = [&](u32 uu) -> u32 { return (pgrp << log_pb) + (uu & pbm); };
            auto unit_blk = [&](u32 uu) -> u32 { return bgrp * (kUnits >> log_pb) + (uu >> log_pb); };
            const bool any = tile0 + (pgrp << log_pb) < a.batch;     // the item's first polynomial exists
            if (any) {
                const u32 blk = unit_blk(u);
                u32 pl = unit_poly(u);
                if (tile0 + pl >= a.batch) pl = pgrp << log_pb;       // a missing polynomial's lanes redo the first one
                const u64 *__restrict__ src = (MIDRING ? ring_slot + ((u64)pl << 16) : a.out + ((tile0 + pl) << 16)) + blk * 256u;
                u64 v[16];
#pragma unroll
                for (int k = 0; k < 16; k++) v[k] = ld_mid(src, field_of<4>(tf, k) * 8u);
                // stages 8..11: twiddles roots[((256 + blk) << i) + g], the same for the 16 lanes of a unit: staged once per
                // item ([block of the item][2^i - 1 + g]) and read back as LDS broadcasts
                const u32 bpi = kUnits >> log_pb;
                if (tid < bpi * 15u) {
                    const u32 bl = tid / 15u, j = tid - bl * 15u;
                    const u32 i = 31u - (u32)__builtin_clz(j + 1u), g = j + 1u - (1u << i);
                    ltw0[tid] = a.tw[((256u + bgrp * bpi + bl) << i) + g];
                }
                __syncthreads();
                const Tw *tw0 = ltw0 + (u >> log_pb) * 15u;
                round_fwd_pm_f<4, kPmPassBound>(v, [&](int i, int g) { return tw0[(1 << i) - 1 + g]; }, m);
#pragma unroll
                for (int k = 0; k < 16; k++) lds[pad16(u * 256u + field_of<4>(tf, k))] = v[k];
            }
            __syncthreads();
            // every lane's loads of the intermediate have landed: the ring slot may be rewritten
            if (MIDRING && tid == 0) ctl_add(ctl + persist_ctl_cdone(xq, ord, maxord), 1u);
            if (any) {
                const u32 blk = unit_blk(u);
                u64 v[16];
#pragma unroll
                for (int k = 0; k < 16; k++) v[k] = lds[pad16(u * 256u + field_of<0>(tf, k))];
                // stages 12..15: roots[((4096 + 16 blk + tf) << i) + g], laid out [blk][2^i - 1 + g][tf] (twc)
                const Tw *__restrict__ tc = a.twc + (size_t)blk * 240u + tf;
                constexpr int B1 = pm_fwd_bound_out(4, kPmPassBound);
                round_fwd_pm_f<4, B1>(v, [&](int i, int g) { return tc[((1 << i) - 1 + g) * 16]; }, m);
                // a thread rewrites exactly the slots it has just gathered: no barrier before the scatter
#pragma unroll
                for (int k = 0; k < 16; k++) lds[pad16(u * 256u + field_of<0>(tf, k))] = pm_canon(v[k], m);
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const u32 e = i * kTH + tid, wu = e >> 8, f = e & 255u;
                    const u64 poly = tile0 + unit_poly(wu);
                    if (poly < a.batch) st_at(a.out + (poly << 16) + unit_blk(wu) * 256u, f * 8u, lds[pad16(e)]);
                }
            }
        }
    }
}

// twc[blk][2^i - 1 + g][tf] = tw[((2^(s0+4) + 16 blk + tf) << i) + g]  (the last four stages' twiddles of 256-blocks, in
// the order the lanes of a unit read them: 16 lanes = 256 contiguous bytes)
__global__ __launch_bounds__(256) void persist_twc_kernel(const Tw *__restrict__ tw, Tw *__restrict__ twc, u32 s0) {
    const u32 e = blockIdx.x * 256u + threadIdx.x;
    const u32 nblk = 1u << s0;
    if (e >= nblk * 240u) return;
    const u32 blk = e / 240u, rem = e % 240u, j = rem / 16u, tf = rem % 16u;
    const u32 i = 31u - (u32)__builtin_clz(j + 1u), g = j + 1u - (1u << i);
    twc[e] = tw[(((1u << (s0 + 4)) + 16u * blk + tf) << i) + g];
}

size_t persist_twc_entries(unsigned log_n) { return ((size_t)1 << (log_n - 8)) * 240u; }

hipError_t launch_persist_twc(const Tw *tw, Tw *twc, unsigned log_n, hipStream_t st) {
    const u32 s0 = log_n - 8;
    const u32 total = (1u << s0) * 240u;
    hipLaunchKernelGGL(persist_twc_kernel, dim3((total + 255u) / 256u), dim3(256), 0, st, tw, twc, s0);
    return hipGetLastError();
}

bool persist_supported(const DevicePlan &p) { return p.log_n == 16 && p.arith == 2; }

size_t persist_ctl_bytes(const PersistTune &t, u64 batch) {
    const u64 ntiles = (batch + ((1ull << t.log_t) - 1)) >> t.log_t;
    return persist_ctl_words(persist_maxord(ntiles, t.lag)) * sizeof(u32);
}
size_t persist_ring_bytes(const PersistTune &t) {
    return t.ringslots ? ((size_t)kPersistQueues * t.ringslots << t.log_t) << 19 : 0;   // 512 KiB per polynomial
}

hipError_t launch_ntt_forward_persist(const DevicePlan &p, const Tw *twc, const u64 *in, u64 *out, u64 batch,
                                      const PersistTune &t, u32 *ctl, u64 *ring, u32 *host_err, unsigned grid, hipStream_t st) {
    if (!persist_supported(p)) return hipErrorNotSupported;
    if (batch == 0) return hipSuccess;
    if (t.log_t > 10 || (t.ringslots && t.ringslots < t.lag + 1)) return hipErrorInvalidValue;
    PersistArgs a{};
    a.in = in; a.out = out; a.ring = ring;
    a.tw = p.tw_fwd_pm; a.twc = twc; a.mod = p.mod;
    a.batch = batch;
    a.ntiles = (batch + ((1ull << t.log_t) - 1)) >> t.log_t;
    a.log_t = t.log_t; a.lag = t.lag; a.ringslots = t.ringslots;
    a.maxord = persist_maxord(a.ntiles, t.lag);
    a.ctl = ctl; a.host_err = host_err;
    hipError_t e = hipMemsetAsync(ctl, 0, persist_ctl_words(a.maxord) * sizeof(u32), st);
    if (e != hipSuccess) return e;
    const void *fn = t.ringslots ? (const void *)ntt_fwd_persist_kernel<true> : (const void *)ntt_fwd_persist_kernel<false>;
    if ((e = allow_big_lds(fn, kLdsBytes)) != hipSuccess) return e;
    KernelTimer kt("ntt_fwd_persist", (int)t.log_t, st);
    if (t.ringslots) hipLaunchKernelGGL(ntt_fwd_persist_kernel<true>, dim3(grid), dim3(kTH), kLdsBytes, st, a);
    else hipLaunchKernelGGL(ntt_fwd_persist_kernel<false>, dim3(grid), dim3(kTH), kLdsBytes, st, a);
    return hipGetLastError();
}

// workgroups the chip holds at once: 2 per CU (LDS: 2 x 74 KiB of 160)
hipError_t persist_grid(unsigned *grid) {
    int dev = 0, cus = 0, per = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess) e = allow_big_lds((const void *)ntt_fwd_persist_kernel<true>, kLdsBytes);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ntt_fwd_persist_kernel<true>, kTH, kLdsBytes);
    if (e != hipSuccess) return e;
    if (per < 1) per = 1;
    *grid = (unsigned)(cus * per);
    return hipSuccess;
}

}  // namespace fhe
