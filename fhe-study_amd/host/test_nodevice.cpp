// test_nodevice.cpp — every path of the C ABI (include/fhe_ntt.h) that runs WITHOUT a device: plan construction and its
// cache under contention (arith/src/ntt.rs:18-38,115-185), argument validation and the error convention, the shard
// arithmetic, the library's switches.  It exists to be run under AddressSanitizer / UndefinedBehaviorSanitizer against a
// host-only build of the library (make -C fhe-study_amd/host san): the plan cache, the workspace map and
// the validation code are host C++ and that is where a lifetime or overflow bug would sit.  With a device present the
// compute calls at the end run too (tiny sizes); without one they must return FHE_E_NO_DEVICE, never crash.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/fhe_ntt.h"
#include "../../include/fhe_ntt_experimental.h"   // the persistent kernels' switches (argument checks only)

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::fprintf(stderr, "FAILED %s:%d: %s  (last error: %s)\n", __FILE__, __LINE__, #cond, fhe_last_error()); \
            std::exit(1);                                                            \
        }                                                                            \
    } while (0)

static const uint64_t Q61 = 2305843009211596801ull, Q16 = 65537ull;

int main() {
    // ---- plans: the reference's checks in the reference's order (ntt.rs:115-131) ----
    const fhe_ntt_plan *p = nullptr;
    CHECK(fhe_ntt_plan_get(Q16, 3, &p) == FHE_E_BAD_N && p == nullptr);
    CHECK(fhe_ntt_plan_get(Q16, 1, &p) == FHE_E_BAD_N);
    CHECK(fhe_ntt_plan_get(Q16, 0, &p) == FHE_E_BAD_N);
    CHECK(fhe_ntt_plan_get(Q16, 1ull << 21, &p) == FHE_E_BAD_N);
    CHECK(fhe_ntt_plan_get(2, 4, &p) == FHE_E_BAD_Q);
    CHECK(fhe_ntt_plan_get(1ull << 63, 4, &p) == FHE_E_BAD_Q);
    CHECK(fhe_ntt_plan_get(Q16, 1ull << 16, &p) == FHE_E_BAD_Q);          // (q - 1) % 2n != 0
    CHECK(fhe_ntt_plan_get(Q16, 4, nullptr) == FHE_E_NULL);
    CHECK(std::strlen(fhe_last_error()) > 0);
    CHECK(fhe_ntt_plan_get(Q16, 4, &p) == FHE_OK && p != nullptr);
    uint64_t q = 0, n = 0, psi = 0, ninv = 0;
    CHECK(fhe_ntt_plan_info(p, &q, &n, &psi, &ninv) == FHE_OK && q == Q16 && n == 4 && psi == 4096 && ninv == 49153);
    uint64_t roots[4], inv[4];
    CHECK(fhe_ntt_plan_tables(p, roots, inv) == FHE_OK);
    CHECK(roots[0] == 1 && roots[1] == 65281 && roots[2] == 4096 && roots[3] == 16);
    CHECK(inv[0] == 1 && inv[1] == 256 && inv[2] == 65521 && inv[3] == 61441);
    CHECK(fhe_ntt_plan_info(nullptr, &q, &n, &psi, &ninv) == FHE_E_NULL);
    // the cache hands out one plan per (q, n), from any number of threads (ntt.rs:18-25: a global Mutex)
    {
        std::vector<const fhe_ntt_plan *> got(16, nullptr);
        std::vector<std::thread> th;
        for (int t = 0; t < 16; t++)
            th.emplace_back([&, t] {
                for (int r = 0; r < 50; r++) {
                    const fhe_ntt_plan *x = nullptr;
                    const uint64_t nn = 1ull << (1 + (t + r) % 12);
                    if (fhe_ntt_plan_get(Q61, nn, &x) != FHE_OK || !x) std::exit(2);
                    if (nn == 2048) got[t] = x;
                }
                const fhe_ntt_plan *x = nullptr;
                if (fhe_ntt_plan_get(Q61, 2048, &x) != FHE_OK) std::exit(2);
                got[t] = x;
            });
        for (auto &t : th) t.join();
        for (int t = 1; t < 16; t++) CHECK(got[t] == got[0]);
    }
    // a thread asking for an EXISTING plan does not wait behind another thread's table build (round 5: plans are built
    // outside the cache lock): 2^20 entries of a modulus not seen yet take ~0.5 s to build; meanwhile every look-up of the
    // small plan must come back at once
    {
        const fhe_ntt_plan *small = nullptr;
        CHECK(fhe_ntt_plan_get(Q61, 64, &small) == FHE_OK);
        std::atomic<bool> building{true};
        std::thread big([&] {
            const fhe_ntt_plan *x = nullptr;
            if (fhe_ntt_plan_get(0x1ffffffffc000001ull, 1ull << 20, &x) != FHE_OK) std::exit(2);
            building = false;
        });
        double worst = 0;
        int looks = 0;
        while (building) {
            const auto t0 = std::chrono::steady_clock::now();
            const fhe_ntt_plan *x = nullptr;
            if (fhe_ntt_plan_get(Q61, 64, &x) != FHE_OK || x != small) std::exit(2);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms > worst) worst = ms;
            looks++;
        }
        big.join();
        CHECK(looks > 10 && worst < 50.0);
    }
    // a composite modulus is not detected (ring_nq.rs:17), tables are still built
    CHECK(fhe_ntt_plan_get(65537ull * 3ull - 2ull * 65537ull + 0ull, 4, &p) == FHE_OK);
    // the largest plan: 2^20 entries per table
    CHECK(fhe_ntt_plan_get(Q61, 1ull << 20, &p) == FHE_OK);
    CHECK(fhe_ntt_plan_arithmetic(p) == FHE_ARITH_PMERSENNE);
    CHECK(fhe_ntt_plan_get(0x1ffffffffc000001ull, 1ull << 12, &p) == FHE_OK && fhe_ntt_plan_arithmetic(p) != FHE_ARITH_PMERSENNE);

    // ---- shard arithmetic ----
    size_t b = 0, e = 0;
    CHECK(fhe_shard_range(630, 8, 0, &b, &e) == FHE_OK && b == 0 && e == 79);
    CHECK(fhe_shard_range(630, 8, 7, &b, &e) == FHE_OK && b == 553 && e == 630);
    CHECK(fhe_shard_range(3, 8, 5, &b, &e) == FHE_OK && b == 3 && e == 3);
    CHECK(fhe_shard_range(0, 1, 0, &b, &e) == FHE_OK && b == 0 && e == 0);
    CHECK(fhe_shard_range(10, 0, 0, &b, &e) == FHE_E_INVALID);
    CHECK(fhe_shard_range(10, 2, 2, &b, &e) == FHE_E_INVALID);
    {   // the gather of shards: argument checks come before any device is needed; without a device it refuses to copy
        int devs[2] = {0, 0};
        uint64_t dummy[4] = {0, 0, 0, 0};
        const void *shards[2] = {dummy, dummy};
        CHECK(fhe_shard_gather_dev(4, 1, 0, devs, shards, 0, dummy, nullptr) == FHE_E_INVALID);
        CHECK(fhe_shard_gather_dev(4, 1, 2, nullptr, shards, 0, dummy, nullptr) == FHE_E_NULL);
        CHECK(fhe_shard_gather_dev(4, 1, 2, devs, nullptr, 0, dummy, nullptr) == FHE_E_NULL);
        CHECK(fhe_shard_gather_dev(0, 1, 2, devs, shards, 0, nullptr, nullptr) == FHE_OK);          // nothing to move
        CHECK(fhe_shard_gather_dev(4, 1, 2, devs, shards, 0, nullptr, nullptr) == FHE_E_NULL);
        if (fhe_ntt_device_count() == 0) CHECK(fhe_shard_gather_dev(4, 1, 2, devs, shards, 0, dummy, nullptr) == FHE_E_NO_DEVICE);
    }
    CHECK(fhe_shard_range(10, 2, 0, nullptr, &e) == FHE_E_NULL);
    {
        size_t total = ~(size_t)0 / 2, covered = 0;                      // no overflow at the top of size_t
        for (unsigned r = 0; r < 7; r++) { CHECK(fhe_shard_range(total, 7, r, &b, &e) == FHE_OK && b <= e); covered += e - b; }
        CHECK(covered == total);
    }

    // ---- switches ----
    CHECK(fhe_ntt_set_batch_tile(128) == FHE_OK && fhe_ntt_set_batch_tile(0) == FHE_OK);
    CHECK(fhe_ntt_set_check_canonical(1) == FHE_OK && fhe_ntt_set_check_canonical(0) == FHE_OK);
    CHECK(fhe_ntt_set_persist(1, 16, 1, 0) == FHE_OK && fhe_ntt_set_persist(2, 1, 1, 2) == FHE_OK);
    CHECK(fhe_ntt_set_persist(1, 3, 1, 0) == FHE_E_INVALID && fhe_ntt_set_persist(1, 16, 2, 2) == FHE_E_INVALID);
    CHECK(fhe_ntt_set_persist(2, 1, 0, 0) == FHE_E_INVALID && fhe_ntt_set_persist(9, 1, 1, 1) == FHE_E_INVALID);
    CHECK(fhe_ntt_set_persist(3, 1, 1, 2) == FHE_OK && fhe_ntt_set_persist(3, 1, 1, 0) == FHE_E_INVALID);
    CHECK(fhe_ntt_set_persist(0, 0, 0, 0) == FHE_OK);
    CHECK(fhe_ntt_set_persist_grid(5) == FHE_OK && fhe_ntt_set_persist_grid(0) == FHE_OK);
    CHECK(fhe_ntt_persist_status() == FHE_OK);
    CHECK(fhe_ntt_kernel_timing_enable(1) == FHE_OK && fhe_ntt_kernel_timing_reset() == FHE_OK && fhe_ntt_kernel_timing_enable(0) == FHE_OK);
    CHECK(fhe_ntt_workspace_bytes() == 0 || fhe_ntt_device_count() > 0);
    CHECK(std::strlen(fhe_ntt_version()) > 0);
    CHECK(fhe_rq_mul_workspace_bytes(p, 3) == 2u * 3u * 4096u * 8u);

    // ---- compute entry points: NULLs first, then the device (or its absence) ----
    CHECK(fhe_ntt_plan_get(Q61, 1024, &p) == FHE_OK);
    std::vector<uint64_t> a(2 * 1024, 1), c(2 * 1024);
    CHECK(fhe_ntt_forward(nullptr, a.data(), c.data(), 2) == FHE_E_NULL);
    CHECK(fhe_ntt_forward(p, nullptr, c.data(), 2) == FHE_E_NULL);
    CHECK(fhe_ntt_forward(p, a.data(), c.data(), 0) == FHE_OK);            // an empty batch is not an error
    CHECK(fhe_rq_mul(p, a.data(), 0, nullptr, 0, c.data(), nullptr, nullptr, nullptr, 1) == FHE_E_NULL);
    const int devs = fhe_ntt_device_count();
    int rc = fhe_ntt_forward(p, a.data(), c.data(), 2);
    if (devs <= 0) {
        CHECK(rc == FHE_E_NO_DEVICE);
        CHECK(fhe_ntt_inverse(p, a.data(), c.data(), 2) == FHE_E_NO_DEVICE);
        CHECK(fhe_rq_mul(p, a.data(), 0, a.data(), 0, c.data(), nullptr, nullptr, nullptr, 2) == FHE_E_NO_DEVICE);
        CHECK(fhe_ntt_plan_prepare(p) == FHE_E_NO_DEVICE);
        CHECK(std::strstr(fhe_last_error(), "no CPU fallback") != nullptr);
    } else {
        CHECK(rc == FHE_OK);
        std::vector<uint64_t> back(2 * 1024);
        CHECK(fhe_ntt_inverse(p, c.data(), back.data(), 2) == FHE_OK && back == a);
    }
    CHECK(fhe_ntt_shutdown() == FHE_OK);
    // plans can be built again after a shutdown
    CHECK(fhe_ntt_plan_get(Q16, 8, &p) == FHE_OK && fhe_ntt_shutdown() == FHE_OK);
    std::printf("all no-device tests passed (%d device(s))\n", devs);
    return 0;
}
