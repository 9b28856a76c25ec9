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
enum : int { kCtlPhase = 0, kCtlOrd = 1, kCtlR = 2, kCtlBind = 3, kCtlStatus = 4, kCtlRes = 5, kCtlGuard = 6, kCtlWords = 8 };

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

// =====================================================================================================================
// TEAMS (variant B): the unit of exchange is ONE polynomial and the exchange stays inside the L2.
//
// A ticket of a queue is (j, r): part r of 16 of the queue's j-th polynomial — polynomial 8 j + x of the batch for queue
// x: the queues share the batch out statically, so nothing is bound and nobody waits for a binding.  Its holder runs the
// strided stages of 16 columns of that polynomial, S(j, r), stores them into slot j mod R of the XCD's ring, adds to
// sdone[j] — and then WAITS for sdone[j] == 16: the other fifteen parts were drawn in the same breath by workgroups of
// the same XCD that are running them right now, so the wait is the skew between sixteen workgroups that started within
// microseconds of each other.  Then it runs the contiguous stages of 16 rows of the SAME polynomial, C(j, r), reading the
// ring slot a few microseconds after it was written — from the L2 — and writes the result.  The slot's lines die in the
// cache (or are written back once, unread, when the cache needs room): the transform reads 8 bytes and writes 8 to 16
// bytes per coefficient of HBM instead of 16 + 16.  While a workgroup waits for its team, the three other workgroups of
// its CU — members of other teams, in other phases — have the CU.
//
// No co-residency is assumed.  A workgroup holds NO undone ticket while it waits (the next one is drawn after the team has
// met), and a wait that lasts (kHelpPolls) turns into work: the waiter draws the next ticket itself — if that is another
// part of its own team (it was never drawn: the queue has fewer than sixteen workgroups at the moment) it runs that S
// part too, and later the C part of every part it owns; if it is a later ticket, every part of the team HAS a holder that
// is running it, and the waiter keeps the ticket for later and goes on waiting.  So a queue served by a single workgroup
// still finishes (sixteen S parts, then sixteen C parts, per polynomial), and every wait is bounded on top of that.
// A queue served by NO workgroup leaves its eighth of the batch untouched: the last workgroup out checks that every
// queue's tickets were all drawn and raises the error word otherwise.
// =====================================================================================================================
constexpr u32 kHelpPolls = 1024;                // polls of the team counter before a waiter starts drawing tickets itself
enum : u32 { kTeamDone = 0, kTeamHelp = 1, kTeamFail = 2 };

// Twiddles: ONE 256-entry tile in LDS that holds the strided stages' roots[1..255] while an S part runs and the 240
// entries of stages 8..11 of the part's rows while a C part runs (re-staged at the two hand-overs, one entry per lane);
// stages 12..15 of a C part are 15 words per lane of the lane-ordered table twc8 — only w, 8 bytes: the second table word
// w 2^32 mod q is recomputed (pm_shift32, three instructions) — fetched into 30 registers when the S part has stored, so
// that they arrive while the team meets.  Four workgroups per CU.
// DEEP: the form for TWO workgroups per CU (launched with half the grid): 256 registers per lane and 43 KiB of LDS, so
// both twiddle tiles stay in LDS and the next S part's coefficients are loaded into a second set of registers as soon as the
// C part's exchange barrier is passed — they land while its last four stages compute.  Fewer workgroups = fewer polynomials
// between "written" and "read" per XCD: at two per CU the ring's lines are still in the L2 when they are read back, at one
// per CU they are not even written back (profiles/r04_persist_B_l2_probe.txt).
constexpr size_t kTeamDeepLdsBytes = kTileBytes + 2 * 256 * sizeof(Tw) + 64 + kProfWords * 8;
template <bool DEEP>
__global__ __launch_bounds__(kTH, (DEEP ? 2 : 4)) void ntt_fwd_team_kernel(PersistArgs a) {
    using S = StridedCfg<8, kCW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + kTileBytes);               // the twiddle tile (see above); DEEP: the C parts' only
    Tw *stw = DEEP ? ltw + 256 : ltw;                                       // the strided stages' tile
    constexpr size_t kTwBytes = (DEEP ? 2 : 1) * 256 * sizeof(Tw);
    u32 *ctrl = reinterpret_cast<u32 *>(smem_raw + kTileBytes + kTwBytes);
    u64 *prof = reinterpret_cast<u64 *>(smem_raw + kTileBytes + kTwBytes + 64);
    const u32 tid0 = threadIdx.x;
    const u32 xq = xcc_id();
    const Mod &m = a.mod;
    const u32 maxord = a.maxord, R = a.ringslots, G = a.groups, nq = kPersistQueues * G;
    constexpr u32 kParts = 16u;
    u32 *const ctl = a.ctl;

    const bool profiling = a.prof != nullptr;
    if (profiling && tid0 == 0) {
        for (u32 i = 0; i < kProfWords; i++) prof[i] = 0;
        prof[kProfWords - 1] = (u64)clock64();
    }
    auto tick = [&](u32 tid, u32 phase, u32 part) {
        if (profiling && tid == 0) {
            const u64 now = (u64)clock64();
            prof[phase * 12u + part] += now - prof[kProfWords - 1];
            prof[kProfWords - 1] = now;
        }
    };
    auto fail = [&](u32 status) {
        atomicOr(ctl + team_ctl_err(nq), status);
        __hip_atomic_fetch_or(a.host_err, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    // Stable GROUPS: the workgroups of an XCD sort themselves into groups of sixteen as they arrive (member number / 16), and
    // a group has its own queue, its own share of the batch and its own ring slots.  Without them every team is a fresh
    // mixture of the XCD's workgroups, each wait couples its sixteen members to the slowest, and within a few polynomials
    // the whole XCD runs in lockstep — everybody loading, everybody waiting — which is the opposite of what the four
    // workgroups of a CU are for.  Groups drift apart (and start staggered), so a CU's workgroups are in different phases.
    // Nothing depends on a group being complete: waiters help (below), and a workgroup whose group's queue has run dry
    // moves on to the next group's of its XCD.
    u32 grp = 0;
    if (tid0 == 0) {
        const u32 mnum = ctl_add(ctl + team_ctl_members(nq, xq), 1u);
        ctrl[kCtlBind] = (mnum / kParts) % G;
    }
    __syncthreads();
    grp = __builtin_amdgcn_readfirstlane(ctrl[kCtlBind]);
    if (tid0 == 0)
        for (u32 i = 0; i < grp * a.lag; i++) __builtin_amdgcn_s_sleep(127);   // stagger: a.lag x ~8k cycles per group
    u32 qx = xq * G + grp;                                 // this workgroup's queue
    u32 dry = 0;                                           // queues of this XCD found empty so far
    auto poly_of = [&](u32 ord) -> u64 { return team_poly(ord, xq, grp, G); };
    auto slot_of = [&](u32 ord) -> u64 * { return a.ring + (((u64)qx * R + ord % R) << 16); };
    auto sdone_of = [&](u32 ord) -> u32 * { return ctl + team_ctl_sdone(nq, qx, ord, maxord); };
    auto cdone_of = [&](u32 ord) -> u32 * { return ctl + team_ctl_cdone(nq, qx, ord, maxord); };
    auto head_of = [&]() -> u32 * { return ctl + team_ctl_head(qx); };
    auto issue_s_loads = [&](u32 ord, u32 r, u32 tid, u64 (&x)[16]) {
        const u32 c = tid % kCW, tf = tid / kCW;
        const u64 *__restrict__ pin = a.in + (poly_of(ord) << 16) + r * kCW;
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = ld_at<u64>(pin, ((field_of<S::A0>(tf, k) << 8) + c) * 8u);
    };
    // this lane's entry of the C part's twiddle tile: roots[((256 + 16 r + row) << i) + g] at [row][2^i - 1 + g]
    auto load_te_c = [&](u32 r, u32 tid) -> Tw {
        const u32 e = tid < 240u ? tid : 239u;
        const u32 bl = e / 15u, j = e - bl * 15u;
        const u32 i = 31u - (u32)__builtin_clz(j + 1u), g = j + 1u - (1u << i);
        return a.tw[((256u + r * kUnits + bl) << i) + g];
    };

    stw[tid0] = a.tw[tid0];
    u32 cur_ord = 0, cur_r = 0, cur_loaded = 0;
    u32 pf_dep = 0;                             // lane 0: cdone of the current ticket's ring slot's previous tenant, as loaded ahead
    if (tid0 == 0) {
        const u32 k0 = ctl_add(head_of(), 1u);
        ctrl[kCtlOrd] = k0 >> 4; ctrl[kCtlR] = k0 & 15u;
    }
    __syncthreads();
    cur_ord = __builtin_amdgcn_readfirstlane(ctrl[kCtlOrd]);
    cur_r = __builtin_amdgcn_readfirstlane(ctrl[kCtlR]);

    u64 v[16], vn[16];                                     // vn (DEEP): the next S part's coefficients, in flight during a C part
    u32 own = 0;                                           // parts of the current polynomial this workgroup has run the S part of
    u32 held = 0, have_held = 0, k_h = 0;                  // the ticket drawn for afterwards (k_h: lane 0, in flight)
    u32 guard_ahead = 1;                                   // lane 0's pf_dep belongs to the current ticket
    u32 nxt_ord = 0, nxt_r = 0;
    bool pre = false;

    // C(team, r): stages 8..15 of rows [16 r, 16 r + 16), twiddles of the last four stages in tcx; `last`: the workgroup's last
    // part of this polynomial — the ticket for afterwards is published and its coefficient loads are issued on the way
    auto c_part = [&](u32 tid, u32 team, u32 r, const u64 (&tcx)[15], bool last) __attribute__((always_inline)) {
        u64 *const slot = slot_of(team);
        const u32 u = tid >> 4, tfc = tid & 15u;
        const u32 blk = r * kUnits + u;                    // the unit's row of the 256 x 256 view
        tick(tid, 1u, 0u);
        {
            const u64 *__restrict__ src = slot + blk * 256u;
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = ld_mid(src, field_of<4>(tfc, k) * 8u);
        }
        if (profiling && tid == 0) prof[25] += 1;
        {
            const Tw *tw0 = ltw + u * 15u;
            round_fwd_pm_f<4, kPmPassBound, 4>(v, [&](int i, int g) { return tw0[(1 << i) - 1 + g]; }, m);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) lds[pad16(u * 256u + field_of<4>(tfc, k))] = v[k];
        tick(tid, 1u, 2u);
        if (last && tid == 0) {                            // the ticket for afterwards, and its ring slot's state
            const u32 t = have_held ? held : k_h;
            const u32 nord = t >> 4;
            pf_dep = kParts;
            if (poly_of(nord) < a.batch && nord >= R) pf_dep = ctl_load(cdone_of(nord - R));
            ctrl[kCtlWords + kCtlOrd] = nord; ctrl[kCtlWords + kCtlR] = t & 15u;
        }
        tick(tid, 1u, 3u);
        __syncthreads();
        tick(tid, 1u, 4u);
        if (tid == 0) ctl_add(cdone_of(team), 1u);   // this part has read the slot
        if (last) {
            nxt_ord = __builtin_amdgcn_readfirstlane(ctrl[kCtlWords + kCtlOrd]);
            nxt_r = __builtin_amdgcn_readfirstlane(ctrl[kCtlWords + kCtlR]);
            pre = poly_of(nxt_ord) < a.batch;
            if (DEEP && pre) issue_s_loads(nxt_ord, nxt_r, tid, vn);   // land while stages 12..15 compute
        }
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = lds[pad16(u * 256u + field_of<0>(tfc, k))];
        tick(tid, 1u, 7u);
        {
            constexpr int B1 = pm_fwd_bound_out(4, kPmPassBound);
            round_fwd_pm_f<4, B1, 0>(v, [&](int i, int g) { const u64 w = tcx[(1 << i) - 1 + g]; return Tw{w, pm_shift32(w, m)}; }, m);
        }
        tick(tid, 1u, 8u);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            lds[pad16(u * 256u + field_of<0>(tfc, k))] = pm_canon(v[k], m);
            if ((k & 3) == 3) FHE_SCHED_FENCE();
        }
        tick(tid, 1u, 9u);
        if (!DEEP && last && pre) issue_s_loads(nxt_ord, nxt_r, tid, v);   // the next item's coefficients: land across the hand-over
        tick(tid, 1u, 10u);
        __syncthreads();
        tick(tid, 1u, 11u);
        if (!DEEP && last) ltw[tid] = a.tw[tid];           // the strided stages' tile again (read after the barrier at the top)
        {
            u64 *__restrict__ pout = a.out + (poly_of(team) << 16) + (u64)r * kUnits * 256u;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const u32 e = i * kTH + tid;
                st_at(pout, e * 8u, lds[pad16(e)]);        // rows 16 r .. 16 r + 15 are contiguous: 32 KiB
                if ((i & 3) == 3) FHE_SCHED_FENCE();
            }
        }
        tick(tid, 1u, 5u);
    };

    for (;;) {                                             // one S part; then, once the team has met, this workgroup's C part(s)
        u32 tid = tid0;
        asm volatile("" : "+v"(tid));
        tick(tid, 0u, 6u);
        __syncthreads();                                   // the LDS tile is free
        tick(tid, 0u, 0u);
        if (poly_of(cur_ord) >= a.batch) {                 // past this queue's last polynomial: so is every later ticket of it
            if (++dry < G) {                               // on to the next group's queue of this XCD (same L2), as one more member
                grp = grp + 1u == G ? 0u : grp + 1u;
                qx = xq * G + grp;
                if (tid == 0) {
                    const u32 t = ctl_add(head_of(), 1u);
                    ctrl[kCtlOrd] = t >> 4; ctrl[kCtlR] = t & 15u;
                }
                __syncthreads();
                cur_ord = __builtin_amdgcn_readfirstlane(ctrl[kCtlOrd]);
                cur_r = __builtin_amdgcn_readfirstlane(ctrl[kCtlR]);
                cur_loaded = 0;
                guard_ahead = 0;
                continue;
            }
            if (profiling && tid == 0)
                for (u32 i = 0; i < kProfWords - 1; i++) atomicAdd((unsigned long long *)a.prof + i, (unsigned long long)prof[i]);
            if (tid == 0 && ctl_add(ctl + team_ctl_left(nq), 1u) + 1u == gridDim.x) {
                // the last workgroup out: were all tickets of all queues drawn?  (an XCD nobody ran on leaves its polynomials
                // untouched: that must be an error, never a silent hole in the output)
                u32 bad = 0;
                for (u32 q = 0; q < nq; q++)
                    if ((u64)ctl_load(ctl + team_ctl_head(q)) < team_queue_polys(a.batch, q / G, q % G, G) * kParts) bad = 1;
                if (bad) fail(kPersistErrQueue);
            }
            return;
        }
        const u32 team = cur_ord;
        // ================= S(team, cur_r) =================
        {
            const u32 c = tid % kCW, tfs = tid / kCW;
            if (!cur_loaded) issue_s_loads(team, cur_r, tid, v);
            tick(tid, 0u, 1u);
            if (profiling && tid == 0) prof[24] += 1;
            round_fwd_pm_f<S::R0, kPmOne, 4>(v, [&](int i, int g) { return stw[(1 << i) + g]; }, m);
#pragma unroll
            for (int k = 0; k < 16; k++) lds[field_of<S::A0>(tfs, k) * kSRow + c] = v[k];
            tick(tid, 0u, 2u);
            if (tid == 0) {                                // the ring slot's previous tenant must have been read (normally: long ago)
                u32 status = 0u, got;
                if (team >= R && !(guard_ahead && pf_dep >= kParts) &&
                    !wait_ge(cdone_of(team - R), kParts, &got)) { status = kPersistErrCdone; fail(status); }
                ctrl[kCtlStatus] = status;
            }
            __syncthreads();
            tick(tid, 0u, 4u);
            if (__builtin_amdgcn_readfirstlane(ctrl[kCtlStatus])) return;
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = lds[field_of<S::a_of(1)>(tfs, k) * kSRow + c];
            {
                constexpr int P1 = pm_fwd_bound_out(S::R0, kPmOne);
                const u32 T1 = 16u + tfs;
                round_fwd_pm_f<4, P1, 4>(v, [&](int i, int g) { return stw[(T1 << i) + g]; }, m);
            }
            tick(tid, 0u, 8u);
            u64 *__restrict__ pout = slot_of(team) + cur_r * kCW;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                st_c<u64>(pout, ((field_of<S::a_of(1)>(tfs, k) << 8) + c) * 8u, v[k]);     // plain: stays in this XCD's L2
                if ((k & 3) == 3) FHE_SCHED_FENCE();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have been acknowledged by the L2
            tick(tid, 0u, 9u);
        }
        __syncthreads();                                   // ... every wave's
        own |= 1u << cur_r;
        const bool single = own == (1u << cur_r);
        // the C part's twiddles start their way now: this lane's entry of the stage 8..11 tile, and stages 12..15 into registers
        // (only for a workgroup's first part of a polynomial — the normal case; a helper fetches them per part, below)
        u64 tc[15];
        if (single) {
            const Tw te_c = load_te_c(cur_r, tid);
            const u64 *__restrict__ tcp = a.twc8 + (size_t)(cur_r * kUnits + (tid >> 4)) * 240u + (tid & 15u);
#pragma unroll
            for (int j = 0; j < 15; j++) tc[j] = tcp[j * 16];
            ltw[tid] = te_c;                               // (every wave is past its S part: the barrier above; entries 240..255 unused)
        }
        // ================= the team meets =================
        if (tid == 0) {
            u32 *const sd = sdone_of(team);
            ctl_add(sd, 1u);
            u32 res = kTeamFail;
            const u32 cap = have_held ? kSpinCap : kHelpPolls;   // holding a later ticket: every part has a holder, just wait
            for (u32 it = 0; it < cap; it++) {
                if (ctl_load(sd) >= kParts) { res = kTeamDone; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (res != kTeamDone && !have_held) res = kTeamHelp;
            if (res == kTeamFail) fail(kPersistErrSdone);
            if (res == kTeamDone && !have_held) k_h = ctl_add(head_of(), 1u);   // the next ticket: a whole C part ahead of its use
            ctrl[kCtlStatus] = res;
        }
        tick(tid, 0u, 3u);                                 // waiting for the team (lane 0)
        __syncthreads();
        tick(tid, 0u, 5u);
        u32 res = __builtin_amdgcn_readfirstlane(ctrl[kCtlStatus]);
        if (res == kTeamFail) return;
        if (res == kTeamHelp) {
            // ---- a long wait: draw the next ticket here and now ----
            if (tid == 0) {
                const u32 t = ctl_add(head_of(), 1u);
                ctrl[kCtlOrd] = t >> 4; ctrl[kCtlR] = t & 15u;
            }
            __syncthreads();
            const u32 t_ord = __builtin_amdgcn_readfirstlane(ctrl[kCtlOrd]), t_r = __builtin_amdgcn_readfirstlane(ctrl[kCtlR]);
            if (t_ord == team) {                           // a part of this very polynomial that nobody had drawn: run its S part too
                cur_r = t_r;
                cur_loaded = 0;
                guard_ahead = 0;
                if (!DEEP) ltw[tid] = a.tw[tid];           // the strided stages' tile again (nobody reads the tile now)
                continue;
            }
            held = (t_ord << 4) | t_r;                     // a later ticket: every part of this polynomial has a holder
            have_held = 1;
            if (tid == 0) {
                u32 got, status = kTeamDone;
                if (!wait_ge(sdone_of(team), kParts, &got)) { status = kTeamFail; fail(kPersistErrSdone); }
                ctrl[kCtlStatus] = status;
            }
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(ctrl[kCtlStatus]) == kTeamFail) return;
        }
        // ================= C part(s) =================
        if (single) {
            c_part(tid, team, cur_r, tc, true);
        } else {
            for (u32 todo = own; todo;) {                  // (rare) every part this workgroup ran the S part of
                const u32 r = (u32)__builtin_ctz(todo);
                todo &= todo - 1u;
                u32 tid2 = tid0;                           // opaque per iteration (see the top of the outer loop)
                asm volatile("" : "+v"(tid2));
                u64 tc2[15];
                const Tw te = load_te_c(r, tid2);
                const u64 *__restrict__ tcp = a.twc8 + (size_t)(r * kUnits + (tid2 >> 4)) * 240u + (tid2 & 15u);
#pragma unroll
                for (int j = 0; j < 15; j++) tc2[j] = tcp[j * 16];
                __syncthreads();                           // the previous part's store loop / stage 8..11 reads are over
                ltw[tid2] = te;
                __syncthreads();
                c_part(tid2, team, r, tc2, todo == 0);
            }
        }
        own = 0; have_held = 0; guard_ahead = 1;
        if (DEEP && pre) {
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = vn[i];
        }
        cur_ord = nxt_ord; cur_r = nxt_r; cur_loaded = pre ? 1u : 0u;
    }
}

// =====================================================================================================================
// FLOW (variant E): the teams again, two workgroups per CU, but nobody waits in the normal course of things.
//
// A workgroup keeps a FIFO of the parts whose strided half it has run and whose contiguous half is still to come, and runs
// one half per iteration: the contiguous half of its OLDEST pending part when that polynomial's sixteen strided halves are
// all in (and it is at least one strided half ahead, or has no strided work left), a strided half of the next ticket
// otherwise.  In steady state that is S(j+1), C(j), S(j+2), C(j+1), ...: the team of polynomial j "meets" while its members
// run their strided half of j+1, so the meeting costs nothing, and since the next half is known one half ahead its
// coefficients — from HBM for an S half, from the ring in the L2 for a C half — its 15 register twiddles and its twiddle tile
// are loaded while the current half computes.  The same rule is the helping rule: a workgroup that finds its oldest part's
// polynomial incomplete simply runs another strided half (all sixteen of a polynomial, if it is alone on its queue), bounded by
// the ring (a slot is rewritten only when all sixteen parts of its previous tenant have been read).  It waits — bounded — only
// when it has no strided work left and its oldest part is not ready; before it waits it signals what it owes.
// Per XCD: 64 workgroups = 4 groups, about 1.5 ring slots per group between "written" and "read" (2 - 3 MiB of a 4 MiB L2).
// =====================================================================================================================
constexpr u32 kFlowFifo = 64;                   // pending parts a workgroup can hold; >= 16 or a lone workgroup could never complete a polynomial (host/test_persist_sched.cpp)
static_assert(kFlowFifo >= 16, "a lone workgroup runs all sixteen strided halves of a polynomial before the first contiguous one");
constexpr size_t kFlowLdsBytes = kTileBytes + 2 * 256 * sizeof(Tw) + 64 + kFlowFifo * 4 + kProfWords * 8;

__global__ __launch_bounds__(kTH, 2) void ntt_fwd_flow_kernel(PersistArgs a) {
    using S = StridedCfg<8, kCW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + kTileBytes);               // a C half's stage 8..11 tile [row][15]
    Tw *stw = ltw + 256;                                                    // the strided stages' tile
    u32 *ctrl = reinterpret_cast<u32 *>(smem_raw + kTileBytes + 2 * 256 * sizeof(Tw));
    u32 *fifo = ctrl + 16;                                                  // pending parts: (queue << 24) | (ordinal << 4) | part
    u64 *prof = reinterpret_cast<u64 *>(smem_raw + kTileBytes + 2 * 256 * sizeof(Tw) + 64 + kFlowFifo * 4);
    const u32 tid0 = threadIdx.x;
    const u32 xq = xcc_id();
    const Mod &m = a.mod;
    const u32 maxord = a.maxord, R = a.ringslots, G = a.groups, nq = kPersistQueues * G;
    constexpr u32 kParts = 16u;
    u32 *const ctl = a.ctl;

    const bool profiling = a.prof != nullptr;
    if (profiling && tid0 == 0) {
        for (u32 i = 0; i < kProfWords; i++) prof[i] = 0;
        prof[kProfWords - 1] = (u64)clock64();
    }
    auto tick = [&](u32 tid, u32 phase, u32 part) {
        if (profiling && tid == 0) {
            const u64 now = (u64)clock64();
            prof[phase * 12u + part] += now - prof[kProfWords - 1];
            prof[kProfWords - 1] = now;
        }
    };
    auto fail = [&](u32 status) {
        atomicOr(ctl + team_ctl_err(nq), status);
        __hip_atomic_fetch_or(a.host_err, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    // a part = (queue, ordinal, part of 16); everything about it follows from that
    auto q_of = [](u32 e) -> u32 { return e >> 24; };
    auto ord_of = [](u32 e) -> u32 { return (e >> 4) & 0xfffffu; };
    auto r_of = [](u32 e) -> u32 { return e & 15u; };
    auto poly_of = [&](u32 q, u32 ord) -> u64 { return team_poly(ord, q / G, q % G, G); };
    auto slot_of = [&](u32 q, u32 ord) -> u64 * { return a.ring + (((u64)q * R + ord % R) << 16); };
    auto issue_s_loads = [&](u32 e, u32 tid, u64 (&x)[16]) {
        const u32 c = tid % kCW, tf = tid / kCW;
        const u64 *__restrict__ pin = a.in + (poly_of(q_of(e), ord_of(e)) << 16) + r_of(e) * kCW;
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = ld_at<u64>(pin, ((field_of<S::A0>(tf, k) << 8) + c) * 8u);
    };
    auto issue_c_loads = [&](u32 e, u32 tid, u64 (&x)[16], u64 (&tcx)[15], u64 &te_w, u64 &te_p) {
        const u32 u = tid >> 4, tfc = tid & 15u, r = r_of(e), blk = r * kUnits + u;
        {   // this lane's entry of the stage 8..11 tile: roots[((256 + 16 r + row) << i) + g] at [row][2^i - 1 + g]
            const u32 ee = tid < 240u ? tid : 239u;
            const u32 bl = ee / 15u, j = ee - bl * 15u;
            const u32 i = 31u - (u32)__builtin_clz(j + 1u), g = j + 1u - (1u << i);
            const Tw *tp = a.tw + (((256u + r * kUnits + bl) << i) + g);
            te_w = tp->w;
            te_p = tp->wp;
        }
        const u64 *__restrict__ src = slot_of(q_of(e), ord_of(e)) + blk * 256u;
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = ld_mid(src, field_of<4>(tfc, k) * 8u);
        const u64 *__restrict__ tcp = a.twc8 + (size_t)blk * 240u + tfc;
#pragma unroll
        for (int j = 0; j < 15; j++) tcx[j] = tcp[j * 16];
    };

    // ---- join a group of this XCD, start staggered ----
    u32 grp = 0;
    if (tid0 == 0) {
        const u32 mnum = ctl_add(ctl + team_ctl_members(nq, xq), 1u);
        ctrl[kCtlBind] = (mnum / kParts) % G;
    }
    stw[tid0] = a.tw[tid0];
    __syncthreads();
    grp = __builtin_amdgcn_readfirstlane(ctrl[kCtlBind]);
    if (tid0 == 0)
        for (u32 i = 0; i < grp * a.lag; i++) __builtin_amdgcn_s_sleep(127);
    u32 qx = xq * G + grp, dry = 0;

    // the next S ticket: drawn by lane 0 (synchronously here; one S half ahead in the loop), or the next queue's when this
    // one has run dry; s_valid = 0: no strided work left on this XCD
    u32 s_e = 0, s_valid = 0;
    auto draw_sync = [&](u32 tid) {           // uniform result through LDS; moves on to the next group's queue when one is dry
        for (;;) {
            if (tid == 0) ctrl[kCtlOrd] = ctl_add(ctl + team_ctl_head(qx), 1u);
            __syncthreads();
            const u32 t = __builtin_amdgcn_readfirstlane(ctrl[kCtlOrd]);
            __syncthreads();
            if (poly_of(qx, t >> 4) < a.batch) { s_e = (qx << 24) | ((t >> 4) << 4) | (t & 15u); s_valid = 1; return; }
            if (++dry >= G) { s_valid = 0; return; }
            grp = grp + 1u == G ? 0u : grp + 1u;
            qx = xq * G + grp;
        }
    };
    draw_sync(tid0);

    u32 pend_head = 0, pend_cnt = 0;                       // FIFO of parts whose C half is to come (uniform)
    u32 s_loaded = 0, c_loaded = 0;                        // vn / vc hold the next S half's / the oldest pending part's coefficients
    bool owes = false;                                     // the previous S half's completion is still to be signalled
    u32 owed_e = 0;
    u64 v[16], vn[16], vc[16], tc[15];
    u64 te_w = 0, te_p = 0;
    u32 k_next = 0;                                        // lane 0: the S ticket after the one at hand, in flight

    auto settle = [&](u32 tid) {
        if (owes) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) ctl_add(ctl + team_ctl_sdone(nq, q_of(owed_e), ord_of(owed_e), maxord), 1u);
            owes = false;
        }
    };

    // lane 0: has the ring slot the next strided half (s_e) writes been read by all sixteen parts of its previous tenant?
    // (issued BEFORE the coefficient loads of that half and consumed at the end of the current one: the counter that orders
    // a wave's loads is in order, so waiting for this word right away would wait for everything issued after it as well)
    auto guard_of_next = [&]() -> u32 {
        if (!s_valid || ord_of(s_e) < R) return kParts;
        return ctl_load(ctl + team_ctl_cdone(nq, q_of(s_e), ord_of(s_e) - R, maxord));
    };
    if (tid0 == 0) ctrl[kCtlGuard] = 1u;
    bool tile_owed = false;                                // vc's twiddle tile entry (te_w, te_p) is still to be written to ltw

    for (;;) {
        u32 tid = tid0;
        asm volatile("" : "+v"(tid));
        tick(tid, 0u, 6u);
        __syncthreads();                                   // the LDS tile is free; a prefetched C half's twiddle tile is in place
        tick(tid, 0u, 0u);
        const u32 oldest = pend_cnt ? __builtin_amdgcn_readfirstlane(fifo[pend_head]) : 0u;
        const bool guard_ok = __builtin_amdgcn_readfirstlane(ctrl[kCtlGuard]) != 0u;   // the ring slot of s_e is free (seen one half ago)
        const bool can_s = s_valid && pend_cnt < kFlowFifo;
        // ---------------- which half now? ----------------
        u32 act;                                           // 0 = S(s_e), 1 = C(oldest), 2 = leave
        if (c_loaded && (pend_cnt >= 2 || !(can_s && guard_ok))) act = 1;    // stay one strided half ahead when that is possible
        else if (can_s && s_loaded && guard_ok) act = 0;
        else if (c_loaded) act = 1;
        else {
            // nothing was decided ahead, or the ring slot the next strided half writes is still being read.
            // Look at both things this workgroup could do, and wait — bounded — only when it can do neither.
            settle(tid);                                   // what this workgroup owes may be what others (or it) wait for
            if (tid == 0) {
                const u32 *sd = pend_cnt ? ctl + team_ctl_sdone(nq, q_of(oldest), ord_of(oldest), maxord) : nullptr;
                const u32 *gd = can_s && ord_of(s_e) >= R ? ctl + team_ctl_cdone(nq, q_of(s_e), ord_of(s_e) - R, maxord) : nullptr;
                u32 res = 0, st = 0;
                if (sd || can_s) {
                    u32 it = 0;
                    for (; it < kSpinCap; it++) {
                        if (sd && ctl_load(sd) >= kParts) res |= 1u;
                        if (can_s && (!gd || ctl_load(gd) >= kParts)) res |= 2u;
                        if (res) break;
                        __builtin_amdgcn_s_sleep(8);
                    }
                    if (!res) { st = sd ? kPersistErrSdone : kPersistErrCdone; fail(st); }
                }
                ctrl[kCtlRes] = res; ctrl[kCtlStatus] = st;
            }
            __syncthreads();
            const u32 res = __builtin_amdgcn_readfirstlane(ctrl[kCtlRes]);
            if (__builtin_amdgcn_readfirstlane(ctrl[kCtlStatus])) return;
            __syncthreads();
            if (profiling && tid == 0) {                   // (diagnostics, reported in columns the halves do not use)
                prof[11] += 1;                             // S row, last column: how often nothing was decided ahead
                prof[10] += s_loaded && !guard_ok;         // S row, "next-loads": ... because the ring slot was still being read
                prof[13] += !s_loaded;                     // C row, "poll": ... because no strided half was fetched
                prof[18] += (res & 1u) && (pend_cnt >= 2 || !(res & 2u));   // C row, "hand-over": ... and a C half followed
                prof[22] += res == 0u;                     // C row, "next-loads": ... and there was nothing left to do
            }
            if ((res & 1u) && (pend_cnt >= 2 || !(res & 2u))) act = 1;
            else if (res & 2u) act = 0;
            else act = 2;                                  // nothing pending, no strided work left on this XCD
        }
        if (act == 2) {                                    // no strided work left on this XCD, nothing pending
            settle(tid);
            if (profiling && tid == 0)
                for (u32 i = 0; i < kProfWords - 1; i++) atomicAdd((unsigned long long *)a.prof + i, (unsigned long long)prof[i]);
            if (tid == 0 && ctl_add(ctl + team_ctl_left(nq), 1u) + 1u == gridDim.x) {
                u32 bad = 0;                               // the last workgroup out: every queue's tickets must have been drawn
                for (u32 q = 0; q < nq; q++)
                    if ((u64)ctl_load(ctl + team_ctl_head(q)) < team_queue_polys(a.batch, q / G, q % G, G) * kParts) bad = 1;
                if (bad) fail(kPersistErrQueue);
            }
            return;
        }
        tick(tid, 0u, 1u);

        if (act == 0) {
            // ================= S half of s_e =================
            const u32 e = s_e, eq = q_of(e), eo = ord_of(e), er = r_of(e);
            const u32 c = tid % kCW, tfs = tid / kCW;
            if (s_loaded) {
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = vn[i];
            } else {
                issue_s_loads(e, tid, v);
            }
            s_loaded = 0;
            if (profiling && tid == 0) prof[24] += 1;
            u32 pf_meet = 0;
            if (tid == 0) {                                // control words, in flight while stages 0..3 compute
                k_next = ctl_add(ctl + team_ctl_head(qx), 1u);
                if (pend_cnt && !c_loaded) pf_meet = ctl_load(ctl + team_ctl_sdone(nq, q_of(oldest), ord_of(oldest), maxord));
            }
            round_fwd_pm_f<S::R0, kPmOne, 4>(v, [&](int i, int g) { return stw[(1 << i) + g]; }, m);
#pragma unroll
            for (int k = 0; k < 16; k++) lds[field_of<S::A0>(tfs, k) * kSRow + c] = v[k];
            tick(tid, 0u, 2u);
            if (tid == 0) {
                ctrl[kCtlOrd] = k_next;                    // the next S ticket of this queue (validity: every lane computes it)
                ctrl[kCtlRes] = pend_cnt && !c_loaded && pf_meet >= kParts;
                fifo[(pend_head + pend_cnt) % kFlowFifo] = e;
            }
            if (owes) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tick(tid, 0u, 3u);
            __syncthreads();
            tick(tid, 0u, 4u);
            if (owes && tid == 0) ctl_add(ctl + team_ctl_sdone(nq, q_of(owed_e), ord_of(owed_e), maxord), 1u);
            owes = true; owed_e = e;
            const u32 t_next = __builtin_amdgcn_readfirstlane(ctrl[kCtlOrd]);
            const u32 meet = __builtin_amdgcn_readfirstlane(ctrl[kCtlRes]);
            pend_cnt++;
            // the next S ticket: this queue's, or — once, when this one has run dry — the next group's (drawn synchronously)
            if (poly_of(qx, t_next >> 4) < a.batch) { s_e = (qx << 24) | ((t_next >> 4) << 4) | (t_next & 15u); s_valid = 1; }
            else if (++dry >= G) s_valid = 0;
            else { grp = grp + 1u == G ? 0u : grp + 1u; qx = xq * G + grp; __syncthreads(); draw_sync(tid); }
            // everything the next two halves need starts now: the ring guard of the next S half, the oldest pending part's
            // coefficients and twiddles if its polynomial is complete, the next S half's coefficients
            u32 pf_guard = kParts;
            if (tid == 0) pf_guard = guard_of_next();
            if (meet) { issue_c_loads(oldest, tid, vc, tc, te_w, te_p); c_loaded = 1; tile_owed = true; }
            if (s_valid && pend_cnt < kFlowFifo) { issue_s_loads(s_e, tid, vn); s_loaded = 1; }
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = lds[field_of<S::a_of(1)>(tfs, k) * kSRow + c];
            tick(tid, 0u, 7u);
            {
                constexpr int P1 = pm_fwd_bound_out(S::R0, kPmOne);
                const u32 T1 = 16u + tfs;
                round_fwd_pm_f<4, P1, 4>(v, [&](int i, int g) { return stw[(T1 << i) + g]; }, m);
            }
            tick(tid, 0u, 8u);
            {
                u64 *__restrict__ pout = slot_of(eq, eo) + er * kCW;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    st_c<u64>(pout, ((field_of<S::a_of(1)>(tfs, k) << 8) + c) * 8u, v[k]);     // plain: stays in this XCD's L2
                    if ((k & 3) == 3) FHE_SCHED_FENCE();
                }
            }
            tick(tid, 0u, 9u);
            if (tile_owed) { ltw[tid] = Tw{te_w, te_p}; tile_owed = false; }   // read after the barrier at the top (no S half reads this tile)
            if (tid == 0) ctrl[kCtlGuard] = pf_guard >= kParts;
            tick(tid, 0u, 5u);
        } else {
            // ================= C half of the oldest pending part =================
            const u32 e = oldest, eq = q_of(e), eo = ord_of(e), er = r_of(e);
            const u32 u = tid >> 4, tfc = tid & 15u;
            tick(tid, 1u, 0u);
            if (!c_loaded) {                               // decided at the top: nothing was fetched ahead
                issue_c_loads(e, tid, vc, tc, te_w, te_p);
                ltw[tid] = Tw{te_w, te_p};
                tile_owed = false;
                __syncthreads();
            }
            c_loaded = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = vc[i];
            u64 tcc[15];
#pragma unroll
            for (int i = 0; i < 15; i++) tcc[i] = tc[i];
            if (profiling && tid == 0) prof[25] += 1;
            // control words, in flight while stages 8..11 compute: the ring guard of the next S half (it may have failed
            // when it was last looked at), and whether the NEXT pending part's polynomial is complete
            const u32 second = pend_cnt >= 2 ? __builtin_amdgcn_readfirstlane(fifo[(pend_head + 1u) % kFlowFifo]) : 0u;
            u32 pf_guard = kParts, pf_meet = 0;
            if (tid == 0) {
                pf_guard = guard_of_next();
                if (pend_cnt >= 2) pf_meet = ctl_load(ctl + team_ctl_sdone(nq, q_of(second), ord_of(second), maxord));
            }
            {
                const Tw *tw0 = ltw + u * 15u;
                round_fwd_pm_f<4, kPmPassBound, 4>(v, [&](int i, int g) { return tw0[(1 << i) - 1 + g]; }, m);
            }
#pragma unroll
            for (int k = 0; k < 16; k++) lds[pad16(u * 256u + field_of<4>(tfc, k))] = v[k];
            tick(tid, 1u, 2u);
            if (tid == 0) ctrl[kCtlRes] = pend_cnt >= 2 && pf_meet >= kParts;
            if (owes) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tick(tid, 1u, 3u);
            __syncthreads();
            tick(tid, 1u, 4u);
            if (tid == 0) {
                if (owes) ctl_add(ctl + team_ctl_sdone(nq, q_of(owed_e), ord_of(owed_e), maxord), 1u);
                ctl_add(ctl + team_ctl_cdone(nq, eq, eo, maxord), 1u);    // this part has read the slot
            }
            owes = false;
            pend_head = (pend_head + 1u) % kFlowFifo;
            pend_cnt--;
            if (__builtin_amdgcn_readfirstlane(ctrl[kCtlRes])) { issue_c_loads(second, tid, vc, tc, te_w, te_p); c_loaded = 1; tile_owed = true; }
            if (!s_loaded && s_valid && pend_cnt < kFlowFifo) { issue_s_loads(s_e, tid, vn); s_loaded = 1; }
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = lds[pad16(u * 256u + field_of<0>(tfc, k))];
            tick(tid, 1u, 7u);
            {
                constexpr int B1 = pm_fwd_bound_out(4, kPmPassBound);
                round_fwd_pm_f<4, B1, 0>(v, [&](int i, int g) { const u64 w = tcc[(1 << i) - 1 + g]; return Tw{w, pm_shift32(w, m)}; }, m);
            }
            tick(tid, 1u, 8u);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                lds[pad16(u * 256u + field_of<0>(tfc, k))] = pm_canon(v[k], m);
                if ((k & 3) == 3) FHE_SCHED_FENCE();
            }
            tick(tid, 1u, 9u);
            __syncthreads();
            tick(tid, 1u, 11u);
            {
                u64 *__restrict__ pout = a.out + (poly_of(eq, eo) << 16) + (u64)er * kUnits * 256u;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const u32 ee = i * kTH + tid;
                    st_at(pout, ee * 8u, lds[pad16(ee)]);  // rows 16 r .. 16 r + 15 are contiguous: 32 KiB
                    if ((i & 3) == 3) FHE_SCHED_FENCE();
                }
            }
            if (tile_owed) { ltw[tid] = Tw{te_w, te_p}; tile_owed = false; }
            if (tid == 0) ctrl[kCtlGuard] = pf_guard >= kParts;
            tick(tid, 1u, 5u);
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

// twc8: the same entries, first table word only (the team kernel recomputes the second)
__global__ __launch_bounds__(256) void persist_twc8_kernel(const Tw *__restrict__ twc, u64 *__restrict__ twc8, u32 total) {
    const u32 e = blockIdx.x * 256u + threadIdx.x;
    if (e < total) twc8[e] = twc[e].w;
}

hipError_t launch_persist_twc(const Tw *tw, Tw *twc, u64 *twc8, unsigned log_n, hipStream_t st) {
    const u32 s0 = log_n - 8;
    const u32 total = (1u << s0) * 240u;
    hipLaunchKernelGGL(persist_twc_kernel, dim3((total + 255u) / 256u), dim3(256), 0, st, tw, twc, s0);
    hipLaunchKernelGGL(persist_twc8_kernel, dim3((total + 255u) / 256u), dim3(256), 0, st, twc, twc8, total);
    return hipGetLastError();
}

bool persist_supported(const DevicePlan &p) { return p.log_n == 16 && p.arith == kArPMersenne; }

// teams: stable groups of sixteen workgroups per XCD
static u32 groups_for(unsigned grid) { const u32 g = grid / (kPersistQueues * 16u); return g ? g : 1u; }
// ordinals a queue can touch: lagged tiles — every tile could land on one queue; teams — its static share of the batch
static u32 maxord_for(const PersistTune &t, u64 batch, unsigned grid) {
    if (t.teams) return (u32)(batch / (kPersistQueues * groups_for(grid)) + 4);
    return persist_maxord((batch + ((1ull << t.log_t) - 1)) >> t.log_t, t.lag);
}
size_t persist_ctl_bytes(const PersistTune &t, u64 batch, unsigned grid) {
    if (t.teams) return team_ctl_words(kPersistQueues * groups_for(grid), maxord_for(t, batch, grid)) * sizeof(u32);
    return persist_ctl_words(maxord_for(t, batch, grid)) * sizeof(u32);
}
size_t persist_ring_bytes(const PersistTune &t, unsigned grid) {
    if (t.teams) return ((size_t)kPersistQueues * groups_for(grid) * t.ringslots) << 19;   // 512 KiB per polynomial
    return t.ringslots ? ((size_t)kPersistQueues * t.ringslots << t.log_t) << 19 : 0;
}

hipError_t launch_ntt_forward_persist(const DevicePlan &p, const Tw *twc, const u64 *twc8, const u64 *in, u64 *out, u64 batch,
                                      const PersistTune &t, u32 *ctl, u64 *ring, u32 *host_err, u64 *prof, unsigned grid, hipStream_t st) {
    if (!persist_supported(p)) return hipErrorNotSupported;
    if (batch == 0) return hipSuccess;
    if (t.log_t > 10 || (!t.teams && t.ringslots && t.ringslots < t.lag + 1)) return hipErrorInvalidValue;
    PersistArgs a{};
    a.in = in; a.out = out; a.ring = ring;
    a.tw = p.tw_fwd_pm; a.twc = twc; a.twc8 = twc8; a.mod = p.mod;
    a.batch = batch;
    a.ntiles = (batch + ((1ull << t.log_t) - 1)) >> t.log_t;
    a.log_t = t.log_t; a.lag = t.lag; a.ringslots = t.ringslots;
    a.maxord = maxord_for(t, batch, grid);
    a.groups = groups_for(grid);
    a.ctl = ctl; a.host_err = host_err; a.prof = prof;
    hipError_t e = hipMemsetAsync(ctl, 0, persist_ctl_bytes(t, batch, grid), st);
    if (e != hipSuccess) return e;
    if (t.teams && t.flow) {
        if (t.log_t != 0 || t.ringslots < 2) return hipErrorInvalidValue;
        KernelTimer kt("ntt_fwd_flow", (int)t.ringslots, st);
        hipLaunchKernelGGL(ntt_fwd_flow_kernel, dim3(grid), dim3(kTH), kFlowLdsBytes, st, a);
        return hipGetLastError();
    }
    if (t.teams) {
        if (t.log_t != 0 || t.ringslots == 0) return hipErrorInvalidValue;
        KernelTimer kt("ntt_fwd_team", (int)t.ringslots, st);
        if (t.deep) hipLaunchKernelGGL(ntt_fwd_team_kernel<true>, dim3(grid), dim3(kTH), kTeamDeepLdsBytes, st, a);
        else hipLaunchKernelGGL(ntt_fwd_team_kernel<false>, dim3(grid), dim3(kTH), kLdsBytes, st, a);
        return hipGetLastError();
    }
    KernelTimer kt("ntt_fwd_persist", (int)t.log_t, st);
    if (t.ringslots) hipLaunchKernelGGL(ntt_fwd_persist_kernel<true>, dim3(grid), dim3(kTH), kLdsBytes, st, a);
    else hipLaunchKernelGGL(ntt_fwd_persist_kernel<false>, dim3(grid), dim3(kTH), kLdsBytes, st, a);
    return hipGetLastError();
}

// workgroups a launch uses: as many as the chip holds at once (4 per CU, 38.3 KiB of LDS each); the teams' deep form: 2 per CU
hipError_t persist_grid(const PersistTune &t, unsigned *grid) {
    int dev = 0, cus = 0, per = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess) {
        if (t.teams && (t.deep || t.flow)) per = 2;
        else if (t.teams) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ntt_fwd_team_kernel<false>, kTH, kLdsBytes);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ntt_fwd_persist_kernel<true>, kTH, kLdsBytes);
    }
    if (e != hipSuccess) return e;
    if (per < 1) per = 1;
    *grid = (unsigned)(cus * per);
    return hipSuccess;
}

}  // namespace fhe
