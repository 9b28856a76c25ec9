// capi_internal.hpp — shared between capi.hip (plans, transforms) and zring.hip (exact
// integer products built on them).  Not part of the public boundary.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <utility>
#include <vector>

#include "../../include/fhe_ntt.h"
#include "../../include/fhe_ntt_experimental.h"   // the persistent kernels' switches: exported, outside the boundary
#include "ntt_kernels.hpp"

constexpr int kMaxDevices = 16;

struct DeviceTables {
    fhe::Tw *tw_fwd = nullptr;
    fhe::Tw *tw_inv = nullptr;
    fhe::u64 *digit_lut = nullptr;   // 136 words, n >= 8 (ntt_rounds.hpp: round0_bits)
    fhe::Tw32 *tw32_fwd = nullptr, *tw32_inv = nullptr;   // small moduli (smallq.hip)
    fhe::Tw *tw_fwd_pm = nullptr, *tw_inv_pm = nullptr;   // pseudo-Mersenne moduli: {w, w 2^32 mod q} (zq_device.hpp)
    fhe::Tw *tw_fwd_mg = nullptr, *tw_inv_mg = nullptr;   // q = 1 (mod 2^32) below 2^61: {w 2^32, w 2^64 mod q}
    fhe::Tw *twc_pm = nullptr;   // the one-launch transform's lane-ordered table of the last four stages (ntt_persist.hip), built on first use
    bool ready = false;
};

struct fhe_ntt_plan {
    fhe::u64 q = 0, n = 0, psi = 0, n_inv = 0;
    unsigned log_n = 0;
    std::vector<fhe::u64> roots, roots_inv;  // as the reference's CACHE value (ntt.rs:18)
    fhe::Mod mod{};
    fhe::Tw ninv{}, s_ninv{};
    fhe::Tw ninv_pm{}, s_ninv_pm{};   // the same two constants as {w, w 2^32 mod q} when mod.pm_k != 0
    fhe::Tw ninv_mg{}, s_ninv_mg{};   // ... as {w 2^32, w 2^64 mod q} when mod.mg_nqh != 0
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
// FHE_NTT_CHECK_CANONICAL / fhe_ntt_set_check_canonical(1): FHE_E_NOT_CANONICAL if any of `count` device words is >= q (synchronises `st`); FHE_OK when the check is off
int fhe_check_canonical_words(uint64_t q, const void *d_x, size_t count, hipStream_t st, const char *who);
int fhe_device_plan(const fhe_ntt_plan *plan, fhe::DevicePlan *dp);
fhe::u64 fhe_batch_tile_for(const fhe_ntt_plan *plan);
// grow-only scratch per (slot, device, stream); slot 0 = fhe_rq_mul_dev / bfv tensor, slot 1 = zring, glue
int fhe_workspace_get(int slot, size_t bytes, hipStream_t st, void **out);
void fhe_workspace_free_all();
void fhe_ext32_free_all();   // zring.hip: tables of the two-small-prime (27-bit) products (digit32.hip)
namespace fhe { struct Ext32Args; }
int fhe_ext32_tables(uint64_t n, fhe::Ext32Args *a);   // fills the per-prime fields for the current device
namespace fhe { struct SmallQArgs; }
// fills the modulus-dependent fields when the plan has a 32-bit form on this device (smallq.hip) and FHE_EXT32 is on
bool fhe_smallq_args(const fhe_ntt_plan *plan, const fhe::DevicePlan &dp, fhe::SmallQArgs *a);
int fhe_smallq_scratch(unsigned log_n, uint64_t rows, hipStream_t st, fhe::SmallQArgs *a);   // a->mid for n > 2^14
bool fhe_mg_enabled();                                 // FHE_MG=0 keeps q = 1 (mod 2^32) on the Shoup forward kernels
bool fhe_pm_enabled();                                 // FHE_PM=0 keeps pseudo-Mersenne moduli on the Shoup kernels
bool fhe_ext32_enabled();                              // FHE_EXT32=0 keeps every product on the 61-bit kernels
// pooled device staging for the host-buffer entry points (capi.hip); release only idle buffers
int fhe_stage_acquire(size_t bytes, void **out, size_t *got, int *dev);
void fhe_stage_release(void *ptr, size_t bytes, int dev);

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

// Staging of HOST buffers around a *_dev entry point: uploads on the calling thread's stream
// (hipStreamPerThread, for which the library workspace is per thread: no lock is needed); the
// destructor drains the stream on error paths before the buffers go back to the pool.
struct FheHostStage {
    struct Buf { void *p; size_t cap; int dev; };
    std::vector<Buf> bufs;
    bool clean = false;   // the stream has been synchronised after the last use of the buffers
    ~FheHostStage() {
        if (!clean) (void)hipStreamSynchronize(hipStreamPerThread);   // error path: drain before reuse
        for (auto &b : bufs) fhe_stage_release(b.p, b.cap, b.dev);
    }
    int up(const void *h, size_t bytes, void **d) {
        *d = nullptr;
        size_t got = 0;
        int dev = 0;
        int rc = fhe_stage_acquire(bytes, d, &got, &dev);
        if (rc != FHE_OK) return rc;
        bufs.push_back(Buf{*d, got, dev});
        if (h && bytes) {
            hipError_t e = hipMemcpyAsync(*d, h, bytes, hipMemcpyHostToDevice, hipStreamPerThread);
            if (e != hipSuccess) return fhe_hip_fail(e, "hipMemcpyAsync H2D");
        }
        return FHE_OK;
    }
    int down(void *h, const void *d, size_t bytes) {
        HIP_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, hipStreamPerThread));
        HIP_TRY(hipStreamSynchronize(hipStreamPerThread));
        clean = true;
        return FHE_OK;
    }
};
