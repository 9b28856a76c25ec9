// persist_sched.hpp — the work order of the one-launch forward transform (ntt_persist.hip), as plain arithmetic
// that compiles for the host too (fhe-study_amd/host/test_persist_sched.cpp simulates the whole protocol on the CPU).
//
// NTT::ntt (arith/src/ntt.rs:44-73) at n = 2^16 is 16 stages.  The two-pass kernels run stages 0..7 on the columns of
// the 256 x 256 view of a polynomial (STRIDED pass, all polynomials), then stages 8..15 on its rows (CONTIGUOUS pass),
// as two launches: the intermediate makes a round trip through HBM.  Here ONE launch runs both: persistent workgroups
// draw TICKETS from a queue — one queue per XCD, chosen by the hardware XCC id, so that everything a queue exchanges
// stays behind ONE L2 — and a ticket names a work item:
//     S(j, r)   strided stages of 16 columns of one polynomial of tile j        (r in [0, 16T): polynomial r / 16, columns r % 16)
//     C(j, r)   contiguous stages of 16 rows ("units" of 256 coefficients) of tile j
// A TILE is T = 2^log_t polynomials; tile ordinals j = 0, 1, ... are per queue and are BOUND to global tiles on first
// touch (the ticket S(j, 0) takes the next global tile, once ordinal j - 1 has been bound: the bound ordinals of a queue
// are a prefix), so queues that run faster simply take more tiles.
// Order of the chunks of I = 16T tickets in a queue, for a lag L:   S(0) .. S(L),  C(0), S(L+1), C(1), S(L+2), ...
// Dependencies, all on EARLIER tickets of the same queue (so the holder of the smallest unfinished ticket can always
// finish: no co-residency is assumed, and no workgroup ever waits for a later ticket):
//     every ticket of ordinal j   waits for  bind[j]        (written by the holder of S(j, 0) before it does anything else)
//     C(j, .)                     waits for  sdone[j] == I  (all S(j, .) stored)
//     S(j, .), ring of R slots    waits for  cdone[j-R] == I  (the slot's previous tenant has been read): needs R >= L + 1
// A workgroup leaves when it draws a C ticket whose tile does not exist (everything after it does not exist either).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FHE_HD __host__ __device__
#else
#define FHE_HD
#endif

namespace fhe {

enum : uint32_t { kPersistS = 0, kPersistC = 1 };
constexpr uint32_t kPersistQueues = 8;          // one per XCD (XCC id & 7)
constexpr uint32_t kPersistInvalid = 0xffffffffu;   // bind[] value of an ordinal past the last tile
constexpr uint32_t kPersistLineWords = 32;      // control words that are hammered sit 128 bytes apart

struct PersistItem {
    uint32_t phase;   // kPersistS / kPersistC
    uint32_t ord;     // tile ordinal within the queue
    uint32_t r;       // item within the tile, [0, 16T)
};

// ticket k of a queue -> work item
FHE_HD inline PersistItem persist_decode(uint64_t k, uint32_t log_t, uint32_t lag) {
    const uint32_t log_i = log_t + 4;                       // I = 16T tickets per chunk
    const uint64_t chunk = k >> log_i;
    PersistItem it;
    it.r = (uint32_t)(k & ((1ull << log_i) - 1ull));
    if (chunk <= lag) {
        it.phase = kPersistS;
        it.ord = (uint32_t)chunk;
    } else {
        const uint64_t d = chunk - (lag + 1);
        it.phase = (d & 1ull) ? kPersistS : kPersistC;
        it.ord = (uint32_t)((d & 1ull) ? lag + 1 + (d >> 1) : (d >> 1));
    }
    return it;
}

// control block (uint32_t words): [queue heads: 8 lines][global tile counter: 1 line][error word: 1 line]
//                                 [bind: 8 x maxord][sdone: 8 x maxord][cdone: 8 x maxord]
FHE_HD inline uint64_t persist_ctl_head(uint32_t q) { return (uint64_t)q * kPersistLineWords; }
FHE_HD inline uint64_t persist_ctl_gtile() { return (uint64_t)kPersistQueues * kPersistLineWords; }
FHE_HD inline uint64_t persist_ctl_err() { return (uint64_t)(kPersistQueues + 1) * kPersistLineWords; }
FHE_HD inline uint64_t persist_ctl_bind(uint32_t q, uint32_t ord, uint32_t maxord) {
    return (uint64_t)(kPersistQueues + 2) * kPersistLineWords + (uint64_t)q * maxord + ord;
}
FHE_HD inline uint64_t persist_ctl_sdone(uint32_t q, uint32_t ord, uint32_t maxord) {
    return persist_ctl_bind(0, 0, maxord) + (uint64_t)(kPersistQueues + q) * maxord + ord;
}
FHE_HD inline uint64_t persist_ctl_cdone(uint32_t q, uint32_t ord, uint32_t maxord) {
    return persist_ctl_bind(0, 0, maxord) + (uint64_t)(2 * kPersistQueues + q) * maxord + ord;
}
FHE_HD inline uint64_t persist_ctl_words(uint32_t maxord) { return persist_ctl_bind(0, 0, maxord) + 3ull * kPersistQueues * maxord; }
// ordinals a queue can touch: every tile could land on one queue, plus the lag + 1 ordinals past the end it looks at
FHE_HD inline uint32_t persist_maxord(uint64_t ntiles, uint32_t lag) { return (uint32_t)(ntiles + lag + 2); }

// ---- teams (variant B): queues are (XCD, group) pairs, q = xcd * groups + group, nq = 8 * groups of them ----
// control block: [queue heads: nq lines][workgroups that have left: 1 line][error word: 1 line][members per XCD: 8 lines]
//                [sdone: nq x maxord][cdone: nq x maxord]
FHE_HD inline uint64_t team_ctl_head(uint32_t q) { return (uint64_t)q * kPersistLineWords; }
FHE_HD inline uint64_t team_ctl_left(uint32_t nq) { return (uint64_t)nq * kPersistLineWords; }
FHE_HD inline uint64_t team_ctl_err(uint32_t nq) { return (uint64_t)(nq + 1) * kPersistLineWords; }
FHE_HD inline uint64_t team_ctl_members(uint32_t nq, uint32_t xcd) { return (uint64_t)(nq + 2 + xcd) * kPersistLineWords; }
FHE_HD inline uint64_t team_ctl_sdone(uint32_t nq, uint32_t q, uint32_t ord, uint32_t maxord) {
    return (uint64_t)(nq + 2 + kPersistQueues) * kPersistLineWords + (uint64_t)q * maxord + ord;
}
FHE_HD inline uint64_t team_ctl_cdone(uint32_t nq, uint32_t q, uint32_t ord, uint32_t maxord) {
    return team_ctl_sdone(nq, 0, 0, maxord) + (uint64_t)(nq + q) * maxord + ord;
}
FHE_HD inline uint64_t team_ctl_words(uint32_t nq, uint32_t maxord) { return team_ctl_sdone(nq, 0, 0, maxord) + 2ull * nq * maxord; }
// polynomial of ordinal `ord` of queue (xcd, group): the batch is dealt out statically, 8 * groups polynomials per round
FHE_HD inline uint64_t team_poly(uint32_t ord, uint32_t xcd, uint32_t group, uint32_t groups) {
    return ((uint64_t)ord * groups + group) * kPersistQueues + xcd;
}
// polynomials of the batch that queue (xcd, group) owns
FHE_HD inline uint64_t team_queue_polys(uint64_t batch, uint32_t xcd, uint32_t group, uint32_t groups) {
    const uint64_t first = (uint64_t)group * kPersistQueues + xcd, step = (uint64_t)groups * kPersistQueues;
    return batch > first ? (batch - first + step - 1) / step : 0;
}

// error word bits
enum : uint32_t { kPersistErrBind = 1u, kPersistErrSdone = 2u, kPersistErrCdone = 4u, kPersistErrQueue = 8u };

}  // namespace fhe
