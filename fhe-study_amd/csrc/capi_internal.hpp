// capi_internal.hpp — shared between capi.hip (plans, transforms) and zring.hip (exact
// integer products built on them).  Not part of the public boundary.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "../../include/fhe_ntt.h"
#include "ntt_kernels.hpp"

constexpr int kMaxDevices = 16;

struct DeviceTables {
    fhe::Tw *tw_fwd = nullptr;
    fhe::Tw *tw_inv = nullptr;
    bool ready = false;
};

struct fhe_ntt_plan {
    fhe::u64 q = 0, n = 0, psi = 0, n_inv = 0;
    unsigned log_n = 0;
    std::vector<fhe::u64> roots, roots_inv;  // as the reference's CACHE value (ntt.rs:18)
    fhe::Mod mod{};
    fhe::Tw ninv{}, s_ninv{};
    mutable std::mutex dev_lock;
    mutable DeviceTables dev[kMaxDevices];
};

int fhe_fail(int code, const char *fmt, ...);
int fhe_hip_fail(hipError_t e, const char *what);
#define HIP_TRY(expr)                                         \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) return fhe_hip_fail(e_, #expr); \
    } while (0)

int fhe_current_device(int *dev);
int fhe_device_plan(const fhe_ntt_plan *plan, fhe::DevicePlan *dp);
fhe::u64 fhe_batch_tile_for(const fhe_ntt_plan *plan);
// grow-only per-device scratch; slot 0 = fhe_rq_mul_dev, slot 1 = zring
int fhe_workspace_get(int slot, size_t bytes, void **out);
// Serialises HOST-buffer entry points that use the shared workspace (each runs on its own
// per-thread stream, so two host threads would otherwise overlap in it).  Held from the first
// launch until the results are back on the host.
std::mutex &fhe_host_workspace_lock();
void fhe_workspace_free_all();

static inline bool fhe_misaligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }
#define REQUIRE_ALIGNED(p)                                                                          \
    do {                                                                                            \
        if ((p) && fhe_misaligned(p))                                                               \
            return fhe_fail(FHE_E_INVALID, #p " must be 16-byte aligned (got %p)", (const void *)(p)); \
    } while (0)

// grid of a grid-stride element-wise kernel: at most 16 blocks of 256 per CU
static inline unsigned fhe_ew_grid(fhe::u64 count) {
    fhe::u64 g = (count + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    return (unsigned)(g ? g : 1);
}
#define LAUNCH_OK(what)                                      \
    do {                                                     \
        hipError_t e_ = hipGetLastError();                   \
        if (e_ != hipSuccess) return fhe_hip_fail(e_, what); \
    } while (0)
