// ntt_persist.hpp — internal interface of the one-launch forward transform (ntt_persist.hip).  Not part of the public
// boundary (include/fhe_ntt.h); capi.hip routes fhe_ntt_forward_dev here when FHE_NTT_PERSIST asks for it.
#pragma once
#include "ntt_kernels.hpp"
#include "persist_sched.hpp"

namespace fhe {

struct PersistTune {
    uint32_t log_t = 0;       // tile = 2^log_t polynomials
    uint32_t lag = 1;         // chunks of S work a queue runs ahead of its C work (teams: start-up stagger between groups, x ~8k cycles)
    uint32_t ringslots = 4;   // tile-sized slots of the per-XCD ring holding the intermediate; 0: it lives in `out`
    bool flow = false;        // teams, two workgroups per CU, a FIFO of pending parts per workgroup: one half per iteration, nobody waits (variant E)
    bool deep = false;        // teams: two workgroups per CU, 256 registers, the next part's coefficients prefetched into a second register set
    bool teams = false;       // variant B: one ticket = both halves of 1/16 of ONE polynomial, the sixteen holders meet in between (log_t = 0, lag unused)
};

struct PersistArgs {
    const u64 *in;
    u64 *out;
    u64 *ring;
    const Tw *tw;     // the plan's forward table {w, w 2^32 mod q}
    const Tw *twc;    // the last four stages' twiddles in lane order (launch_persist_twc)
    const u64 *twc8;  // the same, first word only (teams)
    Mod mod;
    u64 batch, ntiles;
    uint32_t log_t, lag, ringslots, maxord;
    uint32_t groups;  // teams: stable groups of sixteen workgroups per XCD, each with its own queue and ring slots
    uint32_t *ctl;    // control block, zeroed before the launch (persist_sched.hpp)
    uint32_t *host_err;   // pinned host word (device pointer): error bits are OR-ed in here too, where the host can see them
    u64 *prof;            // nullptr, or 26 words: lane 0's shader-clock ticks per part of an iteration (fhe_ntt_persist_profile)
};

bool persist_supported(const DevicePlan &p);
size_t persist_twc_entries(unsigned log_n);
hipError_t launch_persist_twc(const Tw *tw, Tw *twc, u64 *twc8, unsigned log_n, hipStream_t st);
size_t persist_ctl_bytes(const PersistTune &t, u64 batch, unsigned grid);
size_t persist_ring_bytes(const PersistTune &t, unsigned grid);
hipError_t persist_grid(const PersistTune &t, unsigned *grid);
// ctl: persist_ctl_bytes(t, batch) bytes; ring: persist_ring_bytes(t) bytes (nullptr when t.ringslots == 0).
// The error word is ctl[persist_ctl_err()], mirrored into *host_err: non-zero once the launch has finished = a bounded wait ran out.
hipError_t launch_ntt_forward_persist(const DevicePlan &p, const Tw *twc, const u64 *twc8, const u64 *in, u64 *out, u64 batch,
                                      const PersistTune &t, uint32_t *ctl, u64 *ring, uint32_t *host_err, u64 *prof, unsigned grid, hipStream_t st);

}  // namespace fhe
