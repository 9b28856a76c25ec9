// test_persist_sched.cpp — the protocols of the one-launch forward transform (csrc/ntt_persist.hip), simulated on the CPU.
//
// No GPU and no HIP: the ticket arithmetic is csrc/persist_sched.hpp itself (it compiles for the host), and the two
// workgroup state machines below follow the kernels step for step — what a workgroup may do with the control words, in
// which order, and when it waits — while a random scheduler decides who runs next.  Checked for many shapes (workgroups
// per queue from ONE upwards, ragged batches, every lag / ring size the launcher accepts):
//   * termination: some workgroup can always make a step until all have left (no deadlock without co-residency);
//   * coverage: every part of every polynomial / tile is run exactly once;
//   * the ring: a slot is never rewritten before all parts of its previous tenant have read it, and a C part always
//     finds its own polynomial in the slot.
// Variant E (the teams without the meeting) is section 4.
// Usage: test_persist_sched [seeds]     (prints "all persist schedule tests passed")
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../csrc/persist_sched.hpp"

using namespace fhe;
typedef uint32_t u32;
typedef uint64_t u64;

static u64 rng_state = 1;
static u32 rnd() {
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (u32)(rng_state >> 33);
}
#define CHECK(cond, ...)                                              \
    do {                                                              \
        if (!(cond)) {                                                \
            std::fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
            std::fprintf(stderr, __VA_ARGS__);                        \
            std::fprintf(stderr, "\n");                               \
            std::exit(1);                                             \
        }                                                             \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// 1. the ticket order of the lagged-tile kernels (variant A)
// ---------------------------------------------------------------------------------------------------------------------
static void test_decode() {
    for (u32 log_t = 0; log_t <= 3; log_t++)
        for (u32 lag = 0; lag <= 4; lag++) {
            const u32 I = 16u << log_t, ords = 12;
            std::vector<int> seenS(ords * I, 0), seenC(ords * I, 0);
            std::vector<u64> firstS(ords, ~0ull), lastS(ords, 0), firstC(ords, ~0ull), lastC(ords, 0);
            for (u64 k = 0; k < (u64)I * (2 * ords + lag + 2); k++) {
                const PersistItem it = persist_decode(k, log_t, lag);
                CHECK(it.r < I, "r out of range");
                if (it.ord >= ords) continue;
                auto &seen = it.phase == kPersistS ? seenS : seenC;
                seen[it.ord * I + it.r]++;
                auto &f = it.phase == kPersistS ? firstS : firstC;
                auto &l = it.phase == kPersistS ? lastS : lastC;
                if (k < f[it.ord]) f[it.ord] = k;
                if (k > l[it.ord]) l[it.ord] = k;
            }
            for (u32 j = 0; j + lag + 2 < ords; j++) {
                for (u32 r = 0; r < I; r++) CHECK(seenS[j * I + r] == 1 && seenC[j * I + r] == 1, "ticket (%u,%u) drawn %d/%d times", j, r, seenS[j * I + r], seenC[j * I + r]);
                CHECK(lastS[j] < firstC[j], "C(%u) before the end of S(%u)", j, j);                      // a C ticket waits for earlier tickets only
                for (u32 R = lag + 1; R <= lag + 3 && j >= R; R++)
                    CHECK(lastC[j - R] < firstS[j], "S(%u) before the end of C(%u) with a ring of %u (lag %u)", j, j - R, R, lag);
                if (j > 0) CHECK(firstS[j - 1] < firstS[j] && firstC[j - 1] < firstC[j], "ordinals out of order");
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2. variant A: lagged tiles.  One queue, W workgroups, tiles bound on first touch, two tickets held ahead.
// ---------------------------------------------------------------------------------------------------------------------
struct QueueA {
    u32 log_t, lag, R, I;
    u64 ntiles;
    u32 head = 0, gtile = 0;
    std::vector<u32> bind, sdone, cdone;       // per ordinal
    std::vector<int> ranS, ranC;               // per (tile, r)
    std::vector<int> tenant;                   // ring slot -> ordinal (-1 free)
};
struct WgA {
    u32 k_cur, k_nxt, k_nn = 0;
    bool cur_res = false, left = false;
    u32 cur_bind = 0;
    int owed = -1;                             // ordinal whose sdone add is still owed
    // look-ahead result for the next ticket
    bool nxt_res = false;
    u32 nxt_bind = 0;
};

static void run_variant_a(u32 W, u32 log_t, u32 lag, u32 R, u64 ntiles) {
    QueueA q;
    q.log_t = log_t; q.lag = lag; q.R = R; q.I = 16u << log_t; q.ntiles = ntiles;
    const u32 maxord = persist_maxord(ntiles, lag);
    q.bind.assign(maxord, 0); q.sdone.assign(maxord, 0); q.cdone.assign(maxord, 0);
    q.ranS.assign(ntiles * q.I, 0); q.ranC.assign(ntiles * q.I, 0);
    q.tenant.assign(R ? R : 1, -1);
    std::vector<WgA> wg(W);
    for (auto &w : wg) { w.k_cur = q.head++; w.k_nxt = q.head++; }

    auto claim = [&](u32 ord, u32 prev) -> u32 {
        u32 b = kPersistInvalid;
        if (prev != kPersistInvalid) { const u32 g = q.gtile++; if (g < ntiles) b = g + 1; }
        q.bind[ord] = b;
        return b;
    };
    // non-blocking resolution (look-ahead); returns resolved?
    auto try_resolve = [&](const PersistItem &it, u32 &bind) -> bool {
        bind = kPersistInvalid;
        if (it.ord >= maxord) return true;
        if (it.phase == kPersistS && it.r == 0) {
            if (q.bind[it.ord]) bind = q.bind[it.ord];
            else {
                const u32 prev = it.ord ? q.bind[it.ord - 1] : 1u;
                if (!prev) return false;
                bind = claim(it.ord, prev);
            }
        } else {
            bind = q.bind[it.ord];
            if (!bind) return false;
        }
        if (bind != kPersistInvalid) {
            if (it.phase == kPersistC) { if (q.sdone[it.ord] < q.I) return false; }
            else if (R && it.ord >= R) { if (q.cdone[it.ord - R] < q.I) return false; }
        }
        return true;
    };
    u32 alive = W, idle = 0;
    u64 steps = 0;
    while (alive) {
        CHECK(idle < 400000, "variant A: no progress (W=%u T=%u lag=%u R=%u tiles=%llu)", W, 1u << log_t, lag, R, (unsigned long long)ntiles);
        WgA &w = wg[rnd() % W];
        if (w.left) { idle++; continue; }
        steps++;
        const PersistItem it = persist_decode(w.k_cur, log_t, lag);
        if (!w.cur_res) {
            // the slow path: what is owed is settled first, then the ticket is polled for
            if (w.owed >= 0) { q.sdone[w.owed]++; w.owed = -1; idle = 0; continue; }
            u32 b;
            if (!try_resolve(it, b)) { idle++; continue; }       // (one poll)
            w.cur_res = true; w.cur_bind = b;
        }
        idle = 0;
        if (w.k_nn == 0) w.k_nn = q.head++;                       // the ticket after next
        const bool valid = w.cur_bind != kPersistInvalid;
        if (!valid && it.phase == kPersistC) {                    // leaves; what it owes and the tickets it holds are taken care of
            if (w.owed >= 0) { q.sdone[w.owed]++; w.owed = -1; }
            for (u32 t : {w.k_nxt, w.k_nn}) {
                const PersistItem h = persist_decode(t, log_t, lag);
                if (h.phase == kPersistS && h.r == 0 && h.ord < maxord) q.bind[h.ord] = kPersistInvalid;
            }
            w.left = true; alive--;
            continue;
        }
        // look ahead (never waits)
        const PersistItem nit = persist_decode(w.k_nxt, log_t, lag);
        w.nxt_res = try_resolve(nit, w.nxt_bind);
        // behind the exchange barrier: the previous item's completion
        if (w.owed >= 0) { q.sdone[w.owed]++; w.owed = -1; }
        if (valid) {
            const u64 tile = w.cur_bind - 1;
            if (it.phase == kPersistS) {
                q.ranS[tile * q.I + it.r]++;
                if (R) {
                    const u32 s = it.ord % R;
                    CHECK(q.tenant[s] < 0 || q.tenant[s] == (int)it.ord || q.cdone[q.tenant[s]] == q.I,
                          "ring slot %u rewritten by ordinal %u before ordinal %d was read", s, it.ord, q.tenant[s]);
                    q.tenant[s] = (int)it.ord;
                }
                w.owed = (int)it.ord;
            } else {
                CHECK(q.sdone[it.ord] == q.I, "C(%u) ran before S(%u) was complete", it.ord, it.ord);
                if (R) CHECK(q.tenant[it.ord % R] == (int)it.ord, "C(%u) found ordinal %d in its ring slot", it.ord, q.tenant[it.ord % R]);
                q.ranC[tile * q.I + it.r]++;
                q.cdone[it.ord]++;
            }
        }
        w.k_cur = w.k_nxt; w.cur_res = w.nxt_res; w.cur_bind = w.nxt_bind;
        w.k_nxt = w.k_nn; w.k_nn = 0;
    }
    for (u64 i = 0; i < ntiles * q.I; i++) CHECK(q.ranS[i] == 1 && q.ranC[i] == 1, "variant A: item %llu ran %d / %d times", (unsigned long long)i, q.ranS[i], q.ranC[i]);
    (void)steps;
}

// ---------------------------------------------------------------------------------------------------------------------
// 3. variant B: teams.  One (XCD, group) queue, W workgroups, polynomials dealt out statically, waiters help.
// ---------------------------------------------------------------------------------------------------------------------
static void run_variant_b(u32 W, u32 polys, u32 R, u32 help_polls) {
    const u32 kParts = 16;
    u32 head = 0;
    std::vector<u32> sdone(polys + 4, 0), cdone(polys + 4, 0);
    std::vector<int> ranS(polys * kParts, 0), ranC(polys * kParts, 0), tenant(R, -1);
    struct Wg { u32 ord, r, own = 0, held = 0, k_h = 0, polls = 0; bool have_held = false, have_kh = false, left = false; int st = 0; };
    std::vector<Wg> wg(W);
    for (auto &w : wg) { const u32 t = head++; w.ord = t >> 4; w.r = t & 15; }
    u32 alive = W, idle = 0;
    while (alive) {
        CHECK(idle < 400000, "variant B: no progress (W=%u polys=%u R=%u)", W, polys, R);
        Wg &w = wg[rnd() % W];
        if (w.left) { idle++; continue; }
        switch (w.st) {
            case 0:   // top of an iteration: an S part
                if (w.ord >= polys) { w.left = true; alive--; idle = 0; break; }
                if (w.ord >= R && cdone[w.ord - R] < kParts) { idle++; break; }              // the ring slot's previous tenant (polled)
                {
                    const u32 s = w.ord % R;
                    CHECK(tenant[s] < 0 || tenant[s] == (int)w.ord || cdone[tenant[s]] == kParts, "slot %u rewritten before ordinal %d was read", s, tenant[s]);
                    tenant[s] = (int)w.ord;
                }
                ranS[w.ord * kParts + w.r]++;
                w.own |= 1u << w.r;
                sdone[w.ord]++;
                w.polls = 0; w.st = 1; idle = 0;
                break;
            case 1:   // the team meets
                if (sdone[w.ord] >= kParts) {
                    if (!w.have_held) { w.k_h = head++; w.have_kh = true; }
                    w.st = 2; idle = 0;
                    break;
                }
                w.polls++;
                if (!w.have_held && w.polls > help_polls) {                                   // a long wait turns into work
                    const u32 t = head++;
                    if ((t >> 4) == w.ord) { w.r = t & 15; w.st = 0; }                       // a part of its own polynomial nobody had drawn
                    else { w.held = t; w.have_held = true; }
                    idle = 0;
                } else {
                    idle++;
                }
                break;
            case 2:   // its C part(s)
                for (u32 r = 0; r < kParts; r++)
                    if (w.own >> r & 1u) {
                        CHECK(sdone[w.ord] == kParts, "C before the team has met");
                        CHECK(tenant[w.ord % R] == (int)w.ord, "C(%u) found ordinal %d in its slot", w.ord, tenant[w.ord % R]);
                        ranC[w.ord * kParts + r]++;
                        cdone[w.ord]++;
                    }
                {
                    const u32 t = w.have_held ? w.held : w.k_h;
                    w.ord = t >> 4; w.r = t & 15; w.own = 0; w.have_held = false; w.have_kh = false; w.st = 0;
                }
                idle = 0;
                break;
        }
    }
    for (u32 i = 0; i < polys * kParts; i++) CHECK(ranS[i] == 1 && ranC[i] == 1, "variant B: part %u ran %d / %d times (W=%u)", i, ranS[i], ranC[i], W);
}

// ---------------------------------------------------------------------------------------------------------------------
// 4. variant E: the teams without the meeting.  G queues (the groups of ONE XCD), W workgroups, each with a FIFO of the parts
//    whose contiguous half is still to come; one half per iteration, each half in two steps (before / at its exchange
//    barrier) so that the deferred strided-done signal, the look-ahead samples and the other workgroups interleave as on the
//    chip.  Follows ntt_fwd_flow_kernel: the decision at the top, the general path's one poll of two counters, the ring
//    guard seen one half ago, the migration to the next group's queue when one runs dry.
// ---------------------------------------------------------------------------------------------------------------------
static void run_variant_e(u32 W, u32 G, const std::vector<u32> &polys, u32 R, u32 fifo_cap) {
    const u32 kParts = 16;
    std::vector<u32> head(G, 0);
    std::vector<std::vector<u32>> sdone(G), cdone(G);
    std::vector<std::vector<int>> ranS(G), ranC(G), tenant(G);
    for (u32 g = 0; g < G; g++) {
        sdone[g].assign(polys[g] + 8 + W, 0); cdone[g].assign(polys[g] + 8 + W, 0);
        ranS[g].assign((size_t)polys[g] * kParts, 0); ranC[g].assign((size_t)polys[g] * kParts, 0);
        tenant[g].assign(R, -1);
    }
    struct Part { u32 q, ord, r; };
    struct Wg {
        u32 qx = 0, dry = 0;
        Part s_e{}; bool s_valid = false, s_loaded = false, c_loaded = false, guard_ok = true, owes = false, left = false;
        Part owed{};
        std::vector<Part> fifo;
        int st = 0;                    // 0 = top, 1 = S before its barrier, 2 = C before its barrier
        bool meet = false; u32 k_next = 0;
        Part cur{};
    };
    std::vector<Wg> wg(W);
    auto valid = [&](u32 q, u32 t) { return (t >> 4) < polys[q]; };
    auto draw_sync = [&](Wg &w) {
        for (;;) {
            const u32 t = head[w.qx]++;
            if (valid(w.qx, t)) { w.s_e = Part{w.qx, t >> 4, t & 15}; w.s_valid = true; return; }
            if (++w.dry >= G) { w.s_valid = false; return; }
            w.qx = (w.qx + 1) % G;
        }
    };
    auto guard = [&](const Wg &w) { return !w.s_valid || w.s_e.ord < R || cdone[w.s_e.q][w.s_e.ord - R] >= kParts; };
    auto settle = [&](Wg &w) { if (w.owes) { sdone[w.owed.q][w.owed.ord]++; w.owes = false; } };
    for (u32 i = 0; i < W; i++) { wg[i].qx = (i / kParts) % G; draw_sync(wg[i]); }
    u32 alive = W, idle = 0;
    while (alive) {
        CHECK(idle < 600000, "variant E: no progress (W=%u G=%u R=%u)", W, G, R);
        Wg &w = wg[rnd() % W];
        if (w.left) { idle++; continue; }
        if (w.st == 0) {
            const bool can_s = w.s_valid && w.fifo.size() < fifo_cap;
            int act = -1;                                                       // 0 = S, 1 = C, 2 = leave, -1 = poll again
            if (w.c_loaded && (w.fifo.size() >= 2 || !(can_s && w.guard_ok))) act = 1;
            else if (can_s && w.s_loaded && w.guard_ok) act = 0;
            else if (w.c_loaded) act = 1;
            else {
                settle(w);
                u32 res = 0;
                if (!w.fifo.empty() && sdone[w.fifo[0].q][w.fifo[0].ord] >= kParts) res |= 1u;
                if (can_s && guard(w)) res |= 2u;
                if (w.fifo.empty() && !can_s) act = 2;
                else if (!res) { idle++; continue; }                            // one round of the bounded poll
                else if ((res & 1u) && (w.fifo.size() >= 2 || !(res & 2u))) act = 1;
                else act = 0;
            }
            idle = 0;
            if (act == 2) {
                CHECK(!w.owes && w.fifo.empty(), "left with work pending");
                w.left = true; alive--;
                continue;
            }
            if (act == 0) {
                w.cur = w.s_e; w.s_loaded = false;
                CHECK(w.cur.ord < R || cdone[w.cur.q][w.cur.ord - R] >= kParts, "S(%u) before its ring slot was read", w.cur.ord);
                w.k_next = head[w.qx]++;                                        // the next ticket, drawn one half ahead
                w.meet = !w.c_loaded && !w.fifo.empty() && sdone[w.fifo[0].q][w.fifo[0].ord] >= kParts;
                w.st = 1;
            } else {
                CHECK(!w.fifo.empty(), "C with nothing pending");
                w.cur = w.fifo[0]; w.c_loaded = false;
                CHECK(sdone[w.cur.q][w.cur.ord] == kParts, "C before its polynomial was complete");
                CHECK(tenant[w.cur.q][w.cur.ord % R] == (int)w.cur.ord, "C(%u) found ordinal %d in its slot", w.cur.ord, tenant[w.cur.q][w.cur.ord % R]);
                w.guard_ok = guard(w);                                          // (sampled now, used at the next top)
                w.meet = w.fifo.size() >= 2 && sdone[w.fifo[1].q][w.fifo[1].ord] >= kParts;
                w.st = 2;
            }
        } else if (w.st == 1) {                                                 // S at its exchange barrier
            idle = 0;
            settle(w);
            w.fifo.push_back(w.cur);
            w.owes = true; w.owed = w.cur;
            if (valid(w.qx, w.k_next)) w.s_e = Part{w.qx, w.k_next >> 4, w.k_next & 15};
            else if (++w.dry >= G) w.s_valid = false;
            else { w.qx = (w.qx + 1) % G; draw_sync(w); }
            w.guard_ok = guard(w);
            if (w.meet) w.c_loaded = true;
            if (w.s_valid && w.fifo.size() < fifo_cap) w.s_loaded = true;
            {   // the stores of this half: the ring slot changes tenant
                int &t = tenant[w.cur.q][w.cur.ord % R];
                CHECK(t < 0 || t == (int)w.cur.ord || cdone[w.cur.q][(u32)t] == kParts, "slot rewritten before ordinal %d was read", t);
                t = (int)w.cur.ord;
            }
            ranS[w.cur.q][(size_t)w.cur.ord * kParts + w.cur.r]++;
            w.st = 0;
        } else {                                                                // C at its exchange barrier
            idle = 0;
            settle(w);
            cdone[w.cur.q][w.cur.ord]++;
            ranC[w.cur.q][(size_t)w.cur.ord * kParts + w.cur.r]++;
            w.fifo.erase(w.fifo.begin());
            if (w.meet) w.c_loaded = true;
            if (!w.s_loaded && w.s_valid && w.fifo.size() < fifo_cap) w.s_loaded = true;
            w.st = 0;
        }
    }
    for (u32 g = 0; g < G; g++) {
        CHECK(head[g] >= polys[g] * kParts, "queue %u: tickets left undrawn", g);
        for (size_t i = 0; i < ranS[g].size(); i++)
            CHECK(ranS[g][i] == 1 && ranC[g][i] == 1, "variant E: queue %u part %zu ran %d / %d times (W=%u G=%u R=%u)", g, i, ranS[g][i], ranC[g][i], W, G, R);
    }
}

int main(int argc, char **argv) {
    const int seeds = argc > 1 ? std::atoi(argv[1]) : 6;
    test_decode();
    // the static shares of the teams' queues cover a batch exactly once
    for (u32 G : {1u, 3u, 8u})
        for (u64 batch : {0ull, 1ull, 7ull, 8ull, 9ull, 63ull, 64ull, 65ull, 1030ull}) {
            std::vector<int> seen(batch, 0);
            u64 total = 0;
            for (u32 x = 0; x < kPersistQueues; x++)
                for (u32 g = 0; g < G; g++) {
                    const u64 n = team_queue_polys(batch, x, g, G);
                    total += n;
                    for (u32 ord = 0; ord < n; ord++) { const u64 p = team_poly(ord, x, g, G); CHECK(p < batch, "share past the batch"); seen[p]++; }
                    CHECK(team_poly((u32)n, x, g, G) >= batch, "share too short");
                }
            CHECK(total == batch, "shares sum to %llu of %llu", (unsigned long long)total, (unsigned long long)batch);
            for (u64 p = 0; p < batch; p++) CHECK(seen[p] == 1, "polynomial %llu dealt %d times", (unsigned long long)p, seen[p]);
        }
    for (int s = 0; s < seeds; s++) {
        rng_state = 0x9E3779B97F4A7C15ull * (u64)(s + 1);
        for (u32 W : {1u, 2u, 3u, 7u, 16u, 33u, 130u})
            for (u32 log_t : {0u, 1u, 2u})
                for (u32 lag : {0u, 1u, 3u})
                    for (u32 R : {0u, lag + 1, lag + 3})
                        for (u64 ntiles : {1ull, 2ull, 5ull, 11ull}) run_variant_a(W, log_t, lag, R, ntiles);
        for (u32 W : {1u, 2u, 5u, 15u, 16u, 17u, 40u, 128u})
            for (u32 polys : {1u, 2u, 3u, 9u, 20u})
                for (u32 R : {1u, 2u, 4u})
                    for (u32 help : {0u, 3u, 50u}) run_variant_b(W, polys, R, help);
        // E: from ONE workgroup for all the groups of an XCD upwards, ragged queues (an empty one included), every ring size
        // the launcher accepts, and FIFOs shorter than the kernel's 64 so that the "no room" branch runs too.  (Sixteen is the
        // least a FIFO may hold: a lone workgroup must be able to run all sixteen strided halves of a polynomial before the
        // first contiguous one — with fewer the simulation deadlocks, as it should.)
        for (u32 W : {1u, 2u, 5u, 16u, 17u, 31u, 64u, 70u})
            for (u32 G : {1u, 2u, 4u})
                for (u32 R : {2u, 3u, 6u})
                    for (u32 cap : {16u, 17u, 64u})
                        for (u32 shape = 0; shape < 3; shape++) {
                            std::vector<u32> polys(G);
                            for (u32 g = 0; g < G; g++) polys[g] = shape == 0 ? 1u + g : shape == 1 ? (g == 0 ? 0u : 7u) : 3u + (rnd() % 9u);
                            run_variant_e(W, G, polys, R, cap);
                        }
    }
    std::printf("all persist schedule tests passed\n");
    return 0;
}
