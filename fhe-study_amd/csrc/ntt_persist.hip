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

constexpr int kTH = 256;                       // threads of a persistent workgroup: four per CU, as independent instruction streams
constexpr int kUnits = 16;                     // 256-coefficient units of a C item (16 threads each)
constexpr int kCW = 16;                        // columns of an S item
constexpr int kSRow = 17;                      // its row stride in the LDS tile: odd, so that the two row groups a half-wave gathers from hit different banks
constexpr u32 kSpinCap = 1u << 21;             // polls before a wait gives up (seconds)
constexpr size_t kTileBytes = (size_t)(kUnits * 256 + kUnits * 16) * 8;   // padded C tile (>= the 32 KiB S tile)
constexpr size_t kProfWords = 27;              // profile accumulators of lane 0 (u64): 2 phases x 12 + items x 2 + last
// tile + ONE 256-entry twiddle tile (whatever the current item needs) + control words + profile
constexpr size_t kLdsBytes = kTileBytes + 256 * sizeof(Tw) + 64 + kProfWords * 8;

__device__ __forceinline__ u32 xcc_id() {
    u32 v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & (kPersistQueues - 1u);
}
__device__ __forceinline__ u32 ctl_add(u32 *p, u32 x) { return __hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 ctl_load(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ctl_store(u32 *p, u32 x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one lane: poll until *p >= target; false when the spin ran out
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

// compiler-only fence: memory operations are not moved across it (bounds how far loads are hoisted = registers held)
#define FHE_SCHED_FENCE() asm volatile("" ::: "memory")

// R stages on the 16 registers with the twiddle of (stage i, group g) supplied by `twf`.  FENCE > 0: a compiler fence
// every FENCE twiddles, so that twiddles READ FROM LDS are not all hoisted to the front of the round (60 registers).
template <int R, int BIN, int FENCE, typename F>
__device__ __forceinline__ void round_fwd_pm_f(u64 (&v)[16], F twf, const Mod &m) {
    static_assert(pm_fwd_bound_out(R, BIN) <= kPmCap, "a stage would overflow");
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int span = 8 >> i;
        const bool red = pm_fwd_needs_red(pm_fwd_bound_out(i, BIN));
#pragma unroll
        for (int g = 0; g < (1 << i); g++) {
            if (FENCE > 0 && (g % (FENCE > 0 ? FENCE : 1)) == 0) FHE_SCHED_FENCE();
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

// what lane 0 hands the workgroup for a ticket: one record of eight words in LDS (two records: the current ticket when it
// has to be resolved at the top of an iteration, and the NEXT ticket, resolved while the current item computes)
enum : int { kCtlPhase = 0, kCtlOrd = 1, kCtlR = 2, kCtlBind = 3, kCtlStatus = 4, kCtlRes = 5, kCtlWords = 8 };

struct Desc {          // a ticket as every lane knows it (wave-uniform)
    u32 phase, ord, r;
    u32 bind;          // global tile + 1, or kPersistInvalid
};

}  // namespace

// MIDRING: the intermediate lives in a.ring (per-XCD slots, rewritten in place); otherwise in a.out.
//
// Work items (persist_sched.hpp) are 4096 coefficients: an S item = stages 0..7 of 16 columns of a polynomial, a C item =
// stages 8..15 of 16 units of 256 coefficients — 16 polynomials x one block when the tile has 16 polynomials or more (WIDE:
// all 16 units share their twiddles, as in the two-pass contiguous kernel), fewer polynomials x more blocks otherwise.
// The twiddles an item needs are ONE tile of <= 256 entries in LDS, one entry per lane: the 255 of the strided stages; a
// block's 255 (wide C item); or 15 per block for stages 8..11, with stages 12..15 read from the lane-ordered global table.
//
// One iteration = one work item, with the control two tickets deep:
//   top      lane 0 draws the ticket after next (nobody waits for it) and starts the loads of the NEXT ticket's control
//            words (its tile binding, the counter it depends on)
//   half 1   first four stages of the item; lane 0 then looks at the control words that have arrived: next ticket
//            resolved or not
//   barrier  (the LDS exchange's) — behind it the previous item's completion is signalled (its stores have been waited
//            for) and every lane learns the next ticket
//   half 2   last four stages, stores; then, as the registers free up, the next item's coefficient loads and its twiddle
//            entry are issued, to land across the hand-over
// so that neither the ticket nor the dependency check is waited for — unless the next ticket's dependencies are not met
// yet: then it is resolved at the top of its own iteration by polling (bounded), which is the only place a workgroup ever
// waits for another.  Holding tickets ahead is safe: a workgroup runs its tickets in order and only ever blocks on the
// one it is running.
template <bool MIDRING>
__global__ __launch_bounds__(kTH, 4) void ntt_fwd_persist_kernel(PersistArgs a) {
    using S = StridedCfg<8, kCW>;
    static_assert(S::TH == kTH && S::NR == 2 && S::R0 == 4, "S items: 16 columns x 256 rows on 256 threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + kTileBytes);                // the current item's twiddle tile
    u32 *ctrl = reinterpret_cast<u32 *>(smem_raw + kTileBytes + 256 * sizeof(Tw));
    u64 *prof = reinterpret_cast<u64 *>(smem_raw + kTileBytes + 256 * sizeof(Tw) + 64);
    const u32 tid0 = threadIdx.x;
    const u32 xq = xcc_id();
    const Mod &m = a.mod;
    const u32 log_t = a.log_t, T = 1u << log_t, I = 16u << log_t;
    const u32 maxord = a.maxord;
    const u32 log_pb = log_t < 4u ? log_t : 4u, pbm = (1u << log_pb) - 1u, bpi = kUnits >> log_pb;   // C items: polynomials x blocks
    const bool wide = log_pb == 4u;
    u32 *const ctl = a.ctl;
    u32 *const head = ctl + persist_ctl_head(xq);

    // optional profile (a.prof != nullptr): lane 0 accumulates shader-clock ticks per part of an iteration, by item kind
    const bool profiling = a.prof != nullptr;
    if (profiling && tid0 == 0) {
        for (u32 i = 0; i < kProfWords; i++) prof[i] = 0;
        prof[kProfWords - 1] = (u64)clock64();
    }
    auto tick = [&](u32 tid, u32 phase, u32 part) {           // time since the last tick goes to (phase, part)
        if (profiling && tid == 0) {
            const u64 now = (u64)clock64();
            prof[phase * 12u + part] += now - prof[kProfWords - 1];
            prof[kProfWords - 1] = now;
        }
    };
    auto prof_flush = [&](u32 tid) {
        if (profiling && tid == 0)
            for (u32 i = 0; i < kProfWords - 1; i++) atomicAdd((unsigned long long *)a.prof + i, (unsigned long long)prof[i]);
    };

    // ---- lane 0's side of the protocol ----
    auto rd_desc = [&](int base, Desc &d, u32 &res, u32 &status) {
        d.phase = __builtin_amdgcn_readfirstlane(ctrl[base + kCtlPhase]);
        d.ord = __builtin_amdgcn_readfirstlane(ctrl[base + kCtlOrd]);
        d.r = __builtin_amdgcn_readfirstlane(ctrl[base + kCtlR]);
        d.bind = __builtin_amdgcn_readfirstlane(ctrl[base + kCtlBind]);
        res = __builtin_amdgcn_readfirstlane(ctrl[base + kCtlRes]);
        status = __builtin_amdgcn_readfirstlane(ctrl[base + kCtlStatus]);
    };
    auto wr_desc = [&](int base, const PersistItem &it, u32 bind, u32 res, u32 status) {
        ctrl[base + kCtlPhase] = it.phase; ctrl[base + kCtlOrd] = it.ord; ctrl[base + kCtlR] = it.r;
        ctrl[base + kCtlBind] = bind; ctrl[base + kCtlRes] = res; ctrl[base + kCtlStatus] = status;
    };
    auto dep_word = [&](const PersistItem &it) -> u32 * {     // the counter a bound ticket waits for, or nullptr
        if (it.phase == kPersistC) return ctl + persist_ctl_sdone(xq, it.ord, maxord);
        if (MIDRING && it.ord >= a.ringslots) return ctl + persist_ctl_cdone(xq, it.ord - a.ringslots, maxord);
        return nullptr;
    };
    // bind an ordinal to the next global tile (the holder of S(j, 0), once ordinal j - 1 is bound: prev != 0)
    auto claim = [&](const PersistItem &it, u32 prev) -> u32 {
        u32 b = kPersistInvalid;
        if (prev != kPersistInvalid) {
            const u32 g = ctl_add(ctl + persist_ctl_gtile(), 1u);
            if ((u64)g < a.ntiles) b = g + 1u;
        }
        ctl_store(ctl + persist_ctl_bind(xq, it.ord, maxord), b);
        return b;
    };
    // resolve a ticket by polling: the only place a workgroup waits for others.  -> bind, status
    auto resolve_blocking = [&](const PersistItem &it, u32 &bind) -> u32 {
        bind = kPersistInvalid;
        if (it.ord >= maxord) return 0u;      // a ticket drawn on the way out: no tile can be bound that far
        u32 *bp = ctl + persist_ctl_bind(xq, it.ord, maxord);
        if (it.phase == kPersistS && it.r == 0) {
            bind = ctl_load(bp);                  // non-zero: bound while looking ahead, only its dependency was missing
            if (bind == 0) {
                u32 prev = 1u;
                if (it.ord > 0 && !wait_ge(bp - 1, 1u, &prev)) return (u32)kPersistErrBind;
                bind = claim(it, prev);
            }
        } else if (!wait_ge(bp, 1u, &bind)) {
            return (u32)kPersistErrBind;
        }
        if (bind != kPersistInvalid) {
            u32 got;
            if (u32 *dw = dep_word(it))
                if (!wait_ge(dw, I, &got)) return it.phase == kPersistC ? (u32)kPersistErrSdone : (u32)kPersistErrCdone;
        }
        return 0u;
    };
    auto fail = [&](u32 status) {
        atomicOr(ctl + persist_ctl_err(), status);
        __hip_atomic_fetch_or(a.host_err, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // pinned host word
    };

    // ---- every lane's side ----
    auto ring_slot = [&](u32 ord) -> u64 * { return a.ring + ((((u64)xq * a.ringslots + ord % a.ringslots) << log_t) << 16); };
    auto unit_poly = [&](const Desc &d, u32 uu) -> u32 { return ((d.r & ((T >> log_pb) - 1u)) << log_pb) + (uu & pbm); };
    auto unit_blk = [&](const Desc &d, u32 uu) -> u32 { return (d.r >> (log_t - log_pb)) * bpi + (uu >> log_pb); };
    auto has_work = [&](const Desc &d) -> bool {               // false: a ticket whose polynomials lie past the batch
        const u64 tile0 = (u64)(d.bind - 1u) << log_t;
        return d.phase == kPersistS ? tile0 + (d.r >> 4) < a.batch : tile0 + unit_poly(d, 0u) < a.batch;
    };
    // which twiddle tile an item needs (two items with the same key share it)
    auto tw_key = [&](const Desc &d) -> u32 { return d.phase == kPersistS ? 0u : 1u + (d.r >> (log_t - log_pb)); };
    // this lane's entry of that tile
    auto load_twe = [&](const Desc &d, u32 tid) -> Tw {
        u32 idx;
        if (d.phase == kPersistS) {
            idx = tid;                                         // roots[1 .. 255]: stages 0..7 (entry 0 unused)
        } else if (wide) {
            // one block for all 16 units: local index li -> roots[(1 << (8 + ls)) + (blk << ls) + (li - 2^ls)], ls = floor(log2 li)
            const u32 blk = unit_blk(d, 0u), l1 = tid | (tid == 0), ls = 31u - (u32)__builtin_clz(l1);
            idx = (1u << (8u + ls)) + (blk << ls) + (l1 - (1u << ls));
        } else {
            // stages 8..11 only, [block of the item][2^i - 1 + g] = roots[((256 + blk) << i) + g]; lanes past the tile load its last entry
            const u32 e = tid < bpi * 15u ? tid : bpi * 15u - 1u;
            const u32 bl = e / 15u, j = e - bl * 15u;
            const u32 i = 31u - (u32)__builtin_clz(j + 1u), g = j + 1u - (1u << i);
            idx = ((256u + unit_blk(d, 0u) + bl) << i) + g;
        }
        return a.tw[idx];
    };
    // coefficient loads of a resolved ticket
    auto issue_loads = [&](const Desc &d, u32 tid, u64 (&x)[16]) {
        const u64 tile0 = (u64)(d.bind - 1u) << log_t;
        if (d.phase == kPersistS) {
            const u32 pl = d.r >> 4, cg = d.r & 15u, c = tid % kCW, tf = tid / kCW;
            const u64 *__restrict__ pin = a.in + ((tile0 + pl) << 16) + cg * kCW;
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = ld_at<u64>(pin, ((field_of<S::A0>(tf, k) << 8) + c) * 8u);
        } else {
            const u32 u = tid >> 4, tf = tid & 15u, blk = unit_blk(d, u);
            u32 pl = unit_poly(d, u);
            if (tile0 + pl >= a.batch) pl = unit_poly(d, 0u);      // a missing polynomial's lanes redo the item's first one
            const u64 *__restrict__ src = (MIDRING ? ring_slot(d.ord) + ((u64)pl << 16) : a.out + ((tile0 + pl) << 16)) + blk * 256u;
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = ld_mid(src, field_of<4>(tf, k) * 8u);
        }
    };

    // lane 0: the tickets it holds beyond the current one, the control words in flight for the next one, and the
    // completion it still owes for the previous item
    u32 k_nxt = 0, k_nn = 0, pf_bind = 0, pf_dep = 0, owed = 0;   // owed: 1 + index of the sdone word
    bool owes = false;                                            // uniform: the previous item was an S item with stores
    Desc cur{};
    u32 cur_res = 0, cur_loaded = 0;
    u32 staged = 0xffffffffu;                                     // key of the twiddle tile in LDS
    if (tid0 == 0) {
        const u32 k0 = ctl_add(head, 1u);
        k_nxt = ctl_add(head, 1u);
        wr_desc(0, persist_decode(k0, log_t, a.lag), 0u, 0u, 0u);
    }
    __syncthreads();
    { u32 res, st; rd_desc(0, cur, res, st); }

    u64 v[16];
    // completion of the previous item: its stores have been acknowledged in every wave, then ONE counter add
    auto settle = [&](u32 tid) {
        if (owes) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) ctl_add(ctl + (owed - 1u), 1u);
            owes = false;
        }
    };

    for (;;) {
        // opaque per iteration: the lane's index arithmetic (LDS slots, byte offsets) is then redone per item, as in the
        // two-pass kernels, instead of being hoisted out of the loop into ~30 registers that spill
        u32 tid = tid0;
        asm volatile("" : "+v"(tid));
        tick(tid, cur.phase, 6u);                          // hand-over of the previous iteration (charged to the new item's kind)
        __syncthreads();                                   // the LDS tile is free, a prefetched item's twiddle tile is in place
        tick(tid, cur.phase, 0u);                          // wait at the top barrier
        PersistItem nit{};
        if (tid == 0) {
            k_nn = ctl_add(head, 1u);                      // the ticket after next: needed one iteration from now
            nit = persist_decode(k_nxt, log_t, a.lag);
            pf_bind = 0; pf_dep = I;                       // control words of the next ticket: loads in flight until half 1 is done
            if (nit.ord < maxord) {
                const u32 *bp = ctl + persist_ctl_bind(xq, nit.ord, maxord);
                if (nit.phase == kPersistS && nit.r == 0) pf_bind = nit.ord ? ctl_load(bp - 1) : 1u;
                else pf_bind = ctl_load(bp);
                if (const u32 *dw = dep_word(nit)) pf_dep = ctl_load(dw);
            }
        }
        // lane 0, once the control words have arrived: is the next ticket ready to run?  (never waits for another workgroup)
        auto look_ahead = [&]() {
            if (tid == 0) {
                u32 bind = kPersistInvalid, res = 1u;
                if (nit.ord < maxord) {
                    if (nit.phase == kPersistS && nit.r == 0) {
                        if (pf_bind != 0) bind = claim(nit, pf_bind); else res = 0u;
                    } else {
                        bind = pf_bind;
                        if (bind == 0) res = 0u;
                    }
                    if (res && bind != kPersistInvalid && pf_dep < I) res = 0u;
                }
                wr_desc(kCtlWords, nit, bind, res, 0u);
            }
        };

        if (!cur_res) {                                    // the ticket could not be resolved ahead: poll for it now
            settle(tid);                                   // nothing this workgroup owes may be what the ticket waits for
            if (tid == 0) {
                const PersistItem it{cur.phase, cur.ord, cur.r};
                u32 bind;
                const u32 status = resolve_blocking(it, bind);
                if (status) fail(status);
                wr_desc(0, it, bind, 1u, status);
            }
            __syncthreads();
            u32 res, status;
            rd_desc(0, cur, res, status);
            if (status) return;
            cur_res = 1u;
        }
        const bool valid = cur.bind != kPersistInvalid;
        if (!valid && cur.phase == kPersistC) {            // the queue has run dry: everything after this ticket is void too
            settle(tid);
            prof_flush(tid);
            // ... including the two tickets this workgroup holds and will not run.  If one of them is the S(j, 0) that
            // binds ordinal j, the holders of the other S(j, .) tickets are waiting for that word: bind it (to nothing).
            if (tid == 0) {
                const u32 held[2] = {k_nxt, k_nn};
                for (int h = 0; h < 2; h++) {
                    const PersistItem it = persist_decode(held[h], log_t, a.lag);
                    if (it.phase == kPersistS && it.r == 0 && it.ord < maxord)
                        ctl_store(ctl + persist_ctl_bind(xq, it.ord, maxord), kPersistInvalid);
                }
            }
            return;
        }
        const bool work = valid && has_work(cur);
        if (work && !cur_loaded) {                         // nothing could be issued ahead for this item
            const u32 key = tw_key(cur);
            const Tw te = load_twe(cur, tid);
            issue_loads(cur, tid, v);
            if (key != staged) {
                __syncthreads();                           // (the tile may still be read: a previous S item's second half)
                ltw[tid] = te;
                staged = key;
                __syncthreads();
            }
        }
        tick(tid, cur.phase, 1u);                          // resolving by polling / loads that could not be issued ahead
        if (profiling && tid == 0 && work) prof[24 + cur.phase] += 1;

        // ---- first half: stages 0..3 of the pass in registers, scatter into the LDS tile ----
        const u32 c = tid % kCW, tfs = tid / kCW;          // S item: column, row group
        const u32 u = tid >> 4, tfc = tid & 15u;           // C item: unit, lane of the unit
        if (work) {
            if (cur.phase == kPersistS) {
                // (in a persistent loop the compiler cannot prove the global table unclobbered, so it would not use scalar
                // loads for these workgroup-uniform twiddles: they come from the LDS tile, as broadcast reads)
                round_fwd_pm_f<S::R0, kPmOne, 4>(v, [&](int i, int g) { return ltw[(1 << i) + g]; }, m);
#pragma unroll
                for (int k = 0; k < 16; k++) lds[field_of<S::A0>(tfs, k) * kSRow + c] = v[k];
            } else {
                // stages 8..11: the same twiddles for the 16 lanes of a unit (wide: for the whole workgroup)
                const Tw *tw0 = wide ? ltw + 1 : ltw + (u >> log_pb) * 15u;
                round_fwd_pm_f<4, kPmPassBound, 4>(v, [&](int i, int g) { return tw0[(1 << i) - 1 + g]; }, m);
#pragma unroll
                for (int k = 0; k < 16; k++) lds[pad16(u * 256u + field_of<4>(tfc, k))] = v[k];
            }
        }
        tick(tid, cur.phase, 2u);                          // first half (waits for the coefficients)
        look_ahead();
        if (owes) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the previous item's stores (nothing younger is in flight)
        tick(tid, cur.phase, 3u);                          // looking ahead + the previous item's store acknowledgements
        __syncthreads();
        tick(tid, cur.phase, 4u);                          // wait at the exchange barrier
        if (tid == 0) {
            if (owes) ctl_add(ctl + (owed - 1u), 1u);
            // every lane's loads of the intermediate have landed (they were scattered): the ring slot may be rewritten
            if (MIDRING && valid && cur.phase == kPersistC) ctl_add(ctl + persist_ctl_cdone(xq, cur.ord, maxord), 1u);
        }
        owes = false;
        Desc nxt;
        u32 nxt_res, nst;
        rd_desc(kCtlWords, nxt, nxt_res, nst);
        // The next item's coefficient loads are issued as soon as this item's registers are free (after its stores / its
        // last scatter) and land across the hand-over.  (Issued here, into a second set of 32 registers, they would land
        // while the second half computes — but that does not fit 128 registers: the compiler spills the whole set.)
        const bool pre = nxt_res && nxt.bind != kPersistInvalid && has_work(nxt);
        const u32 nkey = pre ? tw_key(nxt) : staged;
        const bool restage = nkey != staged;
        Tw ten = ltw[0];                                   // (placeholder; the next item's twiddle entry when restage)
        if (restage) ten = load_twe(nxt, tid);             // issued first: it is waited for first (hand-over)

        // ---- second half: gather, stages 4..7 of the pass, stores ----
        if (work) {
            const u64 tile0 = (u64)(cur.bind - 1u) << log_t;
            if (cur.phase == kPersistS) {
#pragma unroll
                for (int k = 0; k < 16; k++) v[k] = lds[field_of<S::a_of(1)>(tfs, k) * kSRow + c];
                tick(tid, 0u, 7u);                         // S: descriptor read + gather
                const u32 T1 = (1u << S::ls0_of(1)) + (tfs >> S::a_of(1));
                constexpr int P1 = pm_fwd_bound_out(S::R0, kPmOne);
                round_fwd_pm_f<4, P1, 4>(v, [&](int i, int g) { return ltw[(T1 << i) + g]; }, m);
                tick(tid, 0u, 8u);                         // S: stages 4..7
                const u32 pl = cur.r >> 4, cg = cur.r & 15u;
                u64 *__restrict__ pout = (MIDRING ? ring_slot(cur.ord) + ((u64)pl << 16) : a.out + ((tile0 + pl) << 16)) + cg * kCW;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const u32 off = ((field_of<S::a_of(1)>(tfs, k) << 8) + c) * 8u;
                    if (MIDRING) st_c<u64>(pout, off, v[k]);     // stays in this XCD's L2 for its consumers
                    else st_at(pout, off, v[k]);
                    if ((k & 3) == 3) FHE_SCHED_FENCE();
                }
                tick(tid, 0u, 9u);                         // S: stores
                if (pre) issue_loads(nxt, tid, v);
                tick(tid, 0u, 10u);                        // S: issuing the next item's loads
            } else {
                const u32 blk = unit_blk(cur, u);
#pragma unroll
                for (int k = 0; k < 16; k++) v[k] = lds[pad16(u * 256u + field_of<0>(tfc, k))];
                tick(tid, 1u, 7u);                         // C: descriptor read + gather
                constexpr int B1 = pm_fwd_bound_out(4, kPmPassBound);
                if (wide) {
                    // stages 12..15 from the block's tile: local index ((16 + tf) << i) + g
                    const u32 T1 = 16u + tfc;
                    round_fwd_pm_f<4, B1, 4>(v, [&](int i, int g) { return ltw[(T1 << i) + g]; }, m);
                } else {
                    // stages 12..15: roots[((4096 + 16 blk + tf) << i) + g], laid out [blk][2^i - 1 + g][tf] (twc): global
                    // loads, not fenced — the compiler issues all 15 up front and their latency is paid once
                    const Tw *__restrict__ tc = a.twc + (size_t)blk * 240u + tfc;
                    round_fwd_pm_f<4, B1, 0>(v, [&](int i, int g) { return tc[((1 << i) - 1 + g) * 16]; }, m);
                }
                tick(tid, 1u, 8u);                         // C: stages 12..15
                // a thread rewrites exactly the slots it has just gathered: no barrier before the scatter
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    lds[pad16(u * 256u + field_of<0>(tfc, k))] = pm_canon(v[k], m);
                    if ((k & 3) == 3) FHE_SCHED_FENCE();
                }
                tick(tid, 1u, 9u);                         // C: canonical + scatter
                if (pre) issue_loads(nxt, tid, v);
                tick(tid, 1u, 10u);                        // C: issuing the next item's loads
                __syncthreads();
                tick(tid, 1u, 11u);                        // C: wait at the store barrier
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const u32 e = i * kTH + tid, wu = e >> 8, f = e & 255u;
                    const u64 poly = tile0 + unit_poly(cur, wu);
                    if (poly < a.batch) st_at(a.out + (poly << 16) + unit_blk(cur, wu) * 256u, f * 8u, lds[pad16(e)]);
                    if ((i & 3) == 3) FHE_SCHED_FENCE();
                }
            }
        }
        tick(tid, cur.phase, 5u);                          // second half
        // an S ticket owes its completion even when its polynomial lies past the batch (its consumers count to 16T)
        if (valid && cur.phase == kPersistS) {
            owes = true;
            if (tid == 0) owed = 1u + (u32)persist_ctl_sdone(xq, cur.ord, maxord);
        }
        // ---- hand-over ----
        if (pre && !work) issue_loads(nxt, tid, v);        // (an item without work did not reach the places above)
        if (restage) {
            // the tile is read in second halves: behind a C item's store barrier everybody is past that; an S item has none
            if (!(work && cur.phase == kPersistC)) __syncthreads();
            ltw[tid] = ten;                                // read after the barrier at the top
            staged = nkey;
        }
        cur = nxt;
        cur_res = nxt_res;
        cur_loaded = pre ? 1u : 0u;
        if (tid == 0) k_nxt = k_nn;
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
                                      const PersistTune &t, u32 *ctl, u64 *ring, u32 *host_err, u64 *prof, unsigned grid, hipStream_t st) {
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
    a.ctl = ctl; a.host_err = host_err; a.prof = prof;
    hipError_t e = hipMemsetAsync(ctl, 0, persist_ctl_words(a.maxord) * sizeof(u32), st);
    if (e != hipSuccess) return e;
    const void *fn = t.ringslots ? (const void *)ntt_fwd_persist_kernel<true> : (const void *)ntt_fwd_persist_kernel<false>;
    if ((e = allow_big_lds(fn, kLdsBytes)) != hipSuccess) return e;
    KernelTimer kt("ntt_fwd_persist", (int)t.log_t, st);
    if (t.ringslots) hipLaunchKernelGGL(ntt_fwd_persist_kernel<true>, dim3(grid), dim3(kTH), kLdsBytes, st, a);
    else hipLaunchKernelGGL(ntt_fwd_persist_kernel<false>, dim3(grid), dim3(kTH), kLdsBytes, st, a);
    return hipGetLastError();
}

// workgroups the chip holds at once: 4 per CU (LDS: 4 x 38.3 KiB of 160)
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
