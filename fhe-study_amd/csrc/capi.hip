// capi.hip — the C ABI of libfhe_ntt.so (include/fhe_ntt.h): plan cache, table
// generation, buffer plumbing.  No compute happens on the host: every transform
// goes through the HIP kernels in ntt_kernels.hip, and without a HIP device the
// compute entry points fail with FHE_E_NO_DEVICE (there is no CPU fallback).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "capi_internal.hpp"
#include "ntt_persist.hpp"
#include "smallq.hpp"

using fhe::u64;
typedef unsigned __int128 u128;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int fhe_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int fhe_hip_fail(hipError_t e, const char *what) {
    (void)hipGetLastError();   // the error is reported through the return code: leave no sticky state behind
    return fhe_fail(FHE_E_HIP, "%s: %s", what, hipGetErrorString(e));
}
#define fail fhe_fail
#define hip_fail fhe_hip_fail

// ---------------------------------------------------------------------------
// host number theory (one-off, per plan) — follows arith/src/ntt.rs:115-185
// ---------------------------------------------------------------------------
static inline u64 mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }

// ntt.rs:164-179 const_exp_mod
static u64 exp_mod(u64 q, u64 x, u64 k) {
    u64 r = 1;
    x %= q;
    while (k > 0) {
        if (k & 1) r = mulmod(r, x, q);
        x = mulmod(x, x, q);
        k >>= 1;
    }
    return r;
}
// ntt.rs:182-185 const_inv_mod (Fermat; q assumed prime as in the reference)
static u64 inv_mod(u64 q, u64 x) { return exp_mod(q, x, q - 2); }

static inline u64 shoup(u64 w, u64 q) { return (u64)((((u128)w) << 64) / q); }

static inline u64 bitrev(u64 i, unsigned log_n) {
    u64 r = 0;
    for (unsigned b = 0; b < log_n; b++) r |= ((i >> b) & 1ull) << (log_n - 1 - b);
    return r;
}

// ---------------------------------------------------------------------------
// plans
// ---------------------------------------------------------------------------
static std::mutex g_plans_lock;
static std::map<std::pair<u64, u64>, std::unique_ptr<fhe_ntt_plan>> g_plans;

static int build_plan(u64 q, u64 n, fhe_ntt_plan *p) {
    // order of checks mirrors primitive_root_of_unity(q, 2n), ntt.rs:115-131
    if (n < 2 || (n & (n - 1)) != 0)
        return fail(FHE_E_BAD_N, "n=%llu: must be a power of two >= 2 (ntt.rs:116,139)",
                    (unsigned long long)n);
    unsigned log_n = 0;
    while ((1ull << log_n) < n) log_n++;
    if ((int)log_n > fhe::kMaxLog)
        return fail(FHE_E_BAD_N, "n=%llu: engine supports n <= 2^%d", (unsigned long long)n, fhe::kMaxLog);
    if (q < 3) return fail(FHE_E_BAD_Q, "q=%llu: modulus too small", (unsigned long long)q);
    if (q >> 63)   // the reference's own limit: Zq::add computes self.v + rhs.v in a u64 (zq.rs:225)
        return fail(FHE_E_BAD_Q, "q=%llu: needs q < 2^63 (as the reference's Zq::add does, zq.rs:225)",
                    (unsigned long long)q);
    if ((q - 1) % (2 * n) != 0)
        return fail(FHE_E_BAD_Q, "q=%llu, n=%llu: (q-1) %% 2n != 0 (ntt.rs:117)",
                    (unsigned long long)q, (unsigned long long)n);
    // first k = 1,2,.. with w = k^((q-1)/2n), w^n != 1   (ntt.rs:120-129)
    u64 psi = 0;
    for (u64 k = 1; k < q; k++) {
        u64 w = exp_mod(q, k, (q - 1) / (2 * n));
        if (exp_mod(q, w, n) != 1) { psi = w; break; }
    }
    if (psi == 0) return fail(FHE_E_NO_ROOT, "No primitive root of unity (ntt.rs:130)");

    p->q = q; p->n = n; p->log_n = log_n; p->psi = psi;
    p->n_inv = inv_mod(q, n);  // ntt.rs:27-30
    p->roots.resize(n);
    p->roots_inv.resize(n);
    // roots[i] = psi^bitrev(i) (ntt.rs:133-147): running powers, then bit-reverse
    // placement — same values as the reference's n modexps.
    {
        u64 pw = 1;
        for (u64 j = 0; j < n; j++) {
            p->roots[bitrev(j, log_n)] = pw;
            pw = mulmod(pw, psi, q);
        }
    }
    // roots_inv[i] = roots[i]^(q-2) (ntt.rs:149-161) — kept as the reference's
    // Fermat power so that the tables agree even if q is not prime.
    for (u64 i = 0; i < n; i++) p->roots_inv[i] = inv_mod(q, p->roots[i]);

    p->mod.q = q;
    p->mod.q2 = 2 * q;
    p->mod.nq = (u64)0 - q;
    p->mod.neg2q = (u64)0 - 2 * q;
    p->mod.neg4q = (u64)0 - 4 * q;  // only used when q < 2^61
    p->mod.q2p1 = 2 * q + 1;
    p->mod.r64 = (u64)((((u128)1) << 64) % q);
    p->mod.r64p = shoup(p->mod.r64, q);
    p->mod.onep = (u64)((((u128)1) << 64) / q);
    p->ninv.w = p->n_inv;
    p->ninv.wp = shoup(p->n_inv, q);
    p->s_ninv.w = mulmod(p->roots_inv[1], p->n_inv, q);
    p->s_ninv.wp = shoup(p->s_ninv.w, q);
    // Pseudo-Mersenne form q = 2^k - delta, 56 <= k <= 61, delta <= 2^(k-39) (zq_device.hpp): the transforms then run the
    // five-multiply butterflies on {w, w 2^32 mod q}.  2^61 - 2^21 + 1 (SURVEY.md section 8's modulus) qualifies.
    p->mod.q3p1 = 3 * q + 1;
    {
        unsigned k = 0;
        while ((q >> k) != 0) k++;                       // 2^(k-1) <= q < 2^k
        const u64 delta = (k < 64 ? (1ull << k) : 0ull) - q;
        if (k >= 56 && k <= 61 && delta <= (1ull << (k - 39))) {
            p->mod.pm_k = k;
            p->mod.pm_delta = (uint32_t)delta;
            p->mod.pm_c2 = (uint32_t)(2 * delta);
            p->mod.pm_sh = k - 31;
            p->mod.pm_mask = (1u << (k - 31)) - 1u;
            p->mod.pm_rsh = k - 32;
            p->mod.pm_rmask = (1u << (k - 32)) - 1u;
            p->ninv_pm.w = p->ninv.w;
            p->ninv_pm.wp = mulmod(p->ninv.w, 1ull << 32, q);
            p->s_ninv_pm.w = p->s_ninv.w;
            p->s_ninv_pm.wp = mulmod(p->s_ninv.w, 1ull << 32, q);
        }
    }
    // q = qh 2^32 + 1 below 2^61 (zq_device.hpp, word Montgomery): the forward transforms run on {w 2^32, w 2^64 mod q}
    if ((q & 0xffffffffull) == 1ull && (q >> 32) != 0 && (q >> 61) == 0 && p->mod.pm_k == 0) {
        p->mod.mg_nqh = (uint32_t)(0u - (uint32_t)(q >> 32));
        p->mod.mg_r96 = mulmod(p->mod.r64, 1ull << 32, q);
        p->mod.mg_r128 = mulmod(p->mod.r64, p->mod.r64, q);
        p->ninv_mg.w = mulmod(p->ninv.w, 1ull << 32, q);
        p->ninv_mg.wp = mulmod(p->ninv_mg.w, 1ull << 32, q);
        p->s_ninv_mg.w = mulmod(p->s_ninv.w, 1ull << 32, q);
        p->s_ninv_mg.wp = mulmod(p->s_ninv_mg.w, 1ull << 32, q);
    }
    return FHE_OK;
}

extern "C" int fhe_ntt_plan_get(uint64_t q, uint64_t n, const fhe_ntt_plan **out) {
    if (!out) return fail(FHE_E_NULL, "fhe_ntt_plan_get: out is NULL");
    *out = nullptr;
    auto key = std::make_pair((u64)q, (u64)n);
    {
        std::lock_guard<std::mutex> lk(g_plans_lock);
        auto it = g_plans.find(key);
        if (it != g_plans.end()) {
            *out = it->second.get();
            return FHE_OK;
        }
    }
    // Build OUTSIDE the lock (0.5 s at n = 2^20): a thread asking for an EXISTING plan never waits behind another
    // thread's table build — the reference holds its CACHE mutex across the build (ntt.rs:20-22); the header promises
    // re-entrancy.  Two threads that miss on the same (q, n) both build; the first to insert wins, the loser's
    // tables are discarded (same values: the construction is deterministic).
    std::unique_ptr<fhe_ntt_plan> p(new fhe_ntt_plan());
    int rc = build_plan(q, n, p.get());
    if (rc != FHE_OK) return rc;
    std::lock_guard<std::mutex> lk(g_plans_lock);
    auto it = g_plans.find(key);
    if (it == g_plans.end()) it = g_plans.emplace(key, std::move(p)).first;
    *out = it->second.get();
    return FHE_OK;
}

extern "C" int fhe_ntt_plan_info(const fhe_ntt_plan *plan, uint64_t *q, uint64_t *n, uint64_t *psi,
                                 uint64_t *n_inv) {
    if (!plan) return fail(FHE_E_NULL, "plan is NULL");
    if (q) *q = plan->q;
    if (n) *n = plan->n;
    if (psi) *psi = plan->psi;
    if (n_inv) *n_inv = plan->n_inv;
    return FHE_OK;
}

extern "C" int fhe_ntt_plan_tables(const fhe_ntt_plan *plan, uint64_t *roots, uint64_t *roots_inv) {
    if (!plan) return fail(FHE_E_NULL, "plan is NULL");
    if (roots) memcpy(roots, plan->roots.data(), plan->n * sizeof(u64));
    if (roots_inv) memcpy(roots_inv, plan->roots_inv.data(), plan->n * sizeof(u64));
    return FHE_OK;
}

// ---------------------------------------------------------------------------
// device state
// ---------------------------------------------------------------------------
int fhe_current_device(int *dev) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(FHE_E_NO_DEVICE, "no HIP device available (%s); libfhe_ntt has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    HIP_TRY(hipGetDevice(dev));
    if (*dev < 0 || *dev >= kMaxDevices) return fail(FHE_E_HIP, "device ordinal %d out of range", *dev);
    return FHE_OK;
}

int fhe_device_plan(const fhe_ntt_plan *plan, fhe::DevicePlan *dp) {
    int dev = 0;
    int rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    std::lock_guard<std::mutex> lk(plan->dev_lock);
    DeviceTables &t = plan->dev[dev];
    if (!t.ready) {
        // First use on this device: a blocking allocation + upload (fhe_ntt_plan_prepare does it
        // ahead of time, e.g. before a stream capture).  Nothing is kept on a failure.
        const u64 n = plan->n, q = plan->q;
        std::vector<fhe::Tw> f(n), i(n);
        for (u64 k = 0; k < n; k++) {
            f[k].w = plan->roots[k];
            f[k].wp = shoup(plan->roots[k], q);
            i[k].w = plan->roots_inv[k];
            i[k].wp = shoup(plan->roots_inv[k], q);
        }
        // tables for transforms of 0/1 polynomials (gadget digits): outcomes of the first stages per
        // input bit pattern (ntt_rounds.hpp: round0_bits).  Needs roots[1..7]: n >= 8.
        std::vector<u64> lut;
        if (n >= 8) {
            lut.assign(136, 0);
            const u64 *r = plan->roots.data();
            auto add = [&](u64 x, u64 y) { u64 s = x + y; return s >= q ? s - q : s; };
            auto sub = [&](u64 x, u64 y) { return x >= y ? x - y : x + q - y; };
            for (unsigned p = 0; p < 16; p++) {
                const u64 x0 = p & 1, x1 = (p >> 1) & 1, x2 = (p >> 2) & 1, x3 = (p >> 3) & 1;   // registers c, c+4, c+8, c+12
                const u64 a0 = add(x0, mulmod(r[1], x2, q)), a2 = sub(x0, mulmod(r[1], x2, q));   // stage 0: pairs (c, c+8), (c+4, c+12)
                const u64 a1 = add(x1, mulmod(r[1], x3, q)), a3 = sub(x1, mulmod(r[1], x3, q));
                u64 y[4];
                y[0] = add(a0, mulmod(r[2], a1, q)); y[1] = sub(a0, mulmod(r[2], a1, q));         // stage 1: (c, c+4) with roots[2]
                y[2] = add(a2, mulmod(r[3], a3, q)); y[3] = sub(a2, mulmod(r[3], a3, q));         //          (c+8, c+12) with roots[3]
                for (int j = 0; j < 4; j++) {
                    lut[4 * p + j] = y[j];
                    lut[64 + 4 * p + j] = mulmod(r[4 + j], y[j], q);                              // stage 2's products
                }
            }
            for (unsigned p = 0; p < 4; p++) {
                const u64 x = p & 1, y = (p >> 1) & 1;
                lut[128 + 2 * p] = add(x, mulmod(r[1], y, q));
                lut[128 + 2 * p + 1] = sub(x, mulmod(r[1], y, q));
            }
        }
        fhe::Tw *df = nullptr, *di = nullptr;
        u64 *dl = nullptr;
        hipError_t e = hipMalloc((void **)&df, n * sizeof(fhe::Tw));
        if (e == hipSuccess) e = hipMalloc((void **)&di, n * sizeof(fhe::Tw));
        if (e == hipSuccess && !lut.empty()) e = hipMalloc((void **)&dl, lut.size() * sizeof(u64));
        if (e == hipSuccess) e = hipMemcpy(df, f.data(), n * sizeof(fhe::Tw), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(di, i.data(), n * sizeof(fhe::Tw), hipMemcpyHostToDevice);
        if (e == hipSuccess && dl) e = hipMemcpy(dl, lut.data(), lut.size() * sizeof(u64), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (df) (void)hipFree(df);
            if (di) (void)hipFree(di);
            if (dl) (void)hipFree(dl);
            return hip_fail(e, "uploading the twiddle tables");
        }
        // small moduli: the same two tables as 32-bit Shoup pairs (smallq.hip)
        fhe::Tw32 *sf = nullptr, *si = nullptr;
        if (fhe::smallq_supported(q, plan->log_n)) {
            std::vector<fhe::Tw32> f32(n), i32(n);
            for (u64 k = 0; k < n; k++) {
                f32[k] = fhe::Tw32{(uint32_t)plan->roots[k], (uint32_t)((plan->roots[k] << 32) / q)};
                i32[k] = fhe::Tw32{(uint32_t)plan->roots_inv[k], (uint32_t)((plan->roots_inv[k] << 32) / q)};
            }
            e = hipMalloc((void **)&sf, n * sizeof(fhe::Tw32));
            if (e == hipSuccess) e = hipMalloc((void **)&si, n * sizeof(fhe::Tw32));
            if (e == hipSuccess) e = hipMemcpy(sf, f32.data(), n * sizeof(fhe::Tw32), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(si, i32.data(), n * sizeof(fhe::Tw32), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                if (sf) (void)hipFree(sf);
                if (si) (void)hipFree(si);
                (void)hipFree(df); (void)hipFree(di);
                if (dl) (void)hipFree(dl);
                return hip_fail(e, "uploading the 32-bit twiddle tables");
            }
        }
        // pseudo-Mersenne moduli: the second pair of tables, {w, w 2^32 mod q}
        fhe::Tw *pf = nullptr, *pi = nullptr;
        if (plan->mod.pm_k != 0) {
            for (u64 k = 0; k < n; k++) {
                f[k].wp = mulmod(plan->roots[k], 1ull << 32, q);
                i[k].wp = mulmod(plan->roots_inv[k], 1ull << 32, q);
            }
            e = hipMalloc((void **)&pf, n * sizeof(fhe::Tw));
            if (e == hipSuccess) e = hipMalloc((void **)&pi, n * sizeof(fhe::Tw));
            if (e == hipSuccess) e = hipMemcpy(pf, f.data(), n * sizeof(fhe::Tw), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(pi, i.data(), n * sizeof(fhe::Tw), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                if (pf) (void)hipFree(pf);
                if (pi) (void)hipFree(pi);
                if (sf) (void)hipFree(sf);
                if (si) (void)hipFree(si);
                (void)hipFree(df); (void)hipFree(di);
                if (dl) (void)hipFree(dl);
                return hip_fail(e, "uploading the pseudo-Mersenne twiddle tables");
            }
        }
        // q = 1 (mod 2^32): both tables in word-Montgomery form {w 2^32, w 2^64 mod q}
        fhe::Tw *mf = nullptr, *mi = nullptr;
        if (plan->mod.mg_nqh != 0) {
            for (u64 k = 0; k < n; k++) {
                f[k].w = mulmod(plan->roots[k], 1ull << 32, q);
                f[k].wp = mulmod(f[k].w, 1ull << 32, q);
                i[k].w = mulmod(plan->roots_inv[k], 1ull << 32, q);
                i[k].wp = mulmod(i[k].w, 1ull << 32, q);
            }
            e = hipMalloc((void **)&mf, n * sizeof(fhe::Tw));
            if (e == hipSuccess) e = hipMalloc((void **)&mi, n * sizeof(fhe::Tw));
            if (e == hipSuccess) e = hipMemcpy(mf, f.data(), n * sizeof(fhe::Tw), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(mi, i.data(), n * sizeof(fhe::Tw), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                if (mf) (void)hipFree(mf);
                if (mi) (void)hipFree(mi);
                if (sf) (void)hipFree(sf);
                if (si) (void)hipFree(si);
                (void)hipFree(df); (void)hipFree(di);
                if (dl) (void)hipFree(dl);
                return hip_fail(e, "uploading the Montgomery twiddle table");
            }
        }
        t.tw_fwd_mg = mf;
        t.tw_inv_mg = mi;
        t.tw_fwd_pm = pf;
        t.tw_inv_pm = pi;
        t.tw_fwd = df;
        t.tw_inv = di;
        t.digit_lut = dl;
        t.tw32_fwd = sf;
        t.tw32_inv = si;
        t.ready = true;
    }
    dp->tw_fwd = t.tw_fwd;
    dp->tw_inv = t.tw_inv;
    dp->digit_lut = t.digit_lut;
    dp->tw32_fwd = t.tw32_fwd;
    dp->tw32_inv = t.tw32_inv;
    dp->mod = plan->mod;
    dp->ninv = plan->ninv;
    dp->s_ninv = plan->s_ninv;
    dp->log_n = plan->log_n;
    dp->wide = (plan->q >> 61) == 0;
    dp->tw_fwd_pm = t.tw_fwd_pm;
    dp->tw_inv_pm = t.tw_inv_pm;
    dp->ninv_pm = plan->ninv_pm;
    dp->s_ninv_pm = plan->s_ninv_pm;
    dp->arith = (plan->q >> 62) ? fhe::kArStrict63 : (t.tw_fwd_pm && fhe_pm_enabled()) ? fhe::kArPMersenne : dp->wide ? fhe::kArWide61 : fhe::kArShoup62;
    dp->tw_fwd_mg = fhe_mg_enabled() ? t.tw_fwd_mg : nullptr;
    dp->tw_inv_mg = fhe_mg_enabled() ? t.tw_inv_mg : nullptr;
    dp->ninv_mg = plan->ninv_mg;
    dp->s_ninv_mg = plan->s_ninv_mg;
    return FHE_OK;
}

// FHE_MG=0 (read once): q = 1 (mod 2^32) keeps the Shoup forward kernels — the A/B of profiles/r04_montgomery_ab.txt
bool fhe_mg_enabled() {
    static const bool on = [] { const char *e = getenv("FHE_MG"); return !(e && e[0] == '0'); }();
    return on;
}

// FHE_PM=0 (read once): pseudo-Mersenne moduli stay on the Shoup kernels — how the A/B numbers of DESIGN.md were taken
bool fhe_pm_enabled() {
    static const bool on = [] { const char *e = getenv("FHE_PM"); return !(e && e[0] == '0'); }();
    return on;
}

extern "C" int fhe_ntt_plan_arithmetic(const fhe_ntt_plan *plan) {
    if (!plan) return fail(FHE_E_NULL, "plan is NULL");
    if (fhe::smallq_supported(plan->q, plan->log_n) && fhe_ext32_enabled()) return FHE_ARITH_WORD32;
    if (plan->mod.pm_k != 0 && fhe_pm_enabled()) return FHE_ARITH_PMERSENNE;
    if (plan->mod.mg_nqh != 0 && plan->log_n >= 4 && fhe_mg_enabled()) return FHE_ARITH_MONTGOMERY;
    if (plan->q >> 62) return FHE_ARITH_STRICT63;
    return (plan->q >> 61) == 0 ? FHE_ARITH_SHOUP61 : FHE_ARITH_SHOUP62;
}

extern "C" int fhe_ntt_plan_prepare(const fhe_ntt_plan *plan) {
    if (!plan) return fail(FHE_E_NULL, "plan is NULL");
    fhe::DevicePlan dp;
    return fhe_device_plan(plan, &dp);
}

// contiguous block partition of `total` independent units over `world` devices:
// rank r owns [r*ceil(total/world), min(total, (r+1)*ceil(total/world)))
extern "C" int fhe_shard_range(size_t total, unsigned world, unsigned rank, size_t *begin, size_t *end) {
    if (!begin || !end) return fail(FHE_E_NULL, "fhe_shard_range: NULL output");
    if (world == 0 || rank >= world) return fail(FHE_E_INVALID, "fhe_shard_range: need rank < world (got %u, %u)", rank, world);
    const size_t per = total / world + (total % world != 0);
    const size_t b0 = (size_t)rank * per < total ? (size_t)rank * per : total;
    *begin = b0;
    *end = total - b0 < per ? total : b0 + per;
    return FHE_OK;
}

// shards of a block-partitioned batch -> one device, without torch / RCCL (include/fhe_ntt.h)
extern "C" int fhe_shard_gather_dev(size_t total_rows, size_t row_words, unsigned world, const int *src_devices,
                                    const void *const *d_src_shards, int dst_device, void *d_dst, void *hip_stream) {
    if (world == 0) return fail(FHE_E_INVALID, "fhe_shard_gather_dev: world is 0");
    if (!src_devices || !d_src_shards) return fail(FHE_E_NULL, "fhe_shard_gather_dev: NULL shard table");
    if (total_rows == 0 || row_words == 0) return FHE_OK;
    if (!d_dst) return fail(FHE_E_NULL, "fhe_shard_gather_dev: NULL destination");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(FHE_E_NO_DEVICE, "fhe_shard_gather_dev: no HIP device available; libfhe_ntt has no CPU fallback");
    }
    if (dst_device < 0 || dst_device >= ndev) return fail(FHE_E_INVALID, "fhe_shard_gather_dev: dst_device %d of %d", dst_device, ndev);
    for (unsigned r = 0; r < world; r++) {
        size_t b = 0, e = 0;
        int rc = fhe_shard_range(total_rows, world, r, &b, &e);
        if (rc != FHE_OK) return rc;
        if (e == b) continue;
        if (!d_src_shards[r]) return fail(FHE_E_NULL, "fhe_shard_gather_dev: shard %u (rows %zu..%zu) is NULL", r, b, e);
        if (src_devices[r] < 0 || src_devices[r] >= ndev) return fail(FHE_E_INVALID, "fhe_shard_gather_dev: src_devices[%u] = %d of %d", r, src_devices[r], ndev);
        HIP_TRY(hipMemcpyPeerAsync((u64 *)d_dst + b * row_words, dst_device, d_src_shards[r], src_devices[r],
                                   (e - b) * row_words * sizeof(u64), (hipStream_t)hip_stream));
    }
    return FHE_OK;
}

// ---- opt-in input validation ---------------------------------------------------------------
// The reference cannot construct a Zq with v >= q (Zq::from_u64 reduces, zq.rs:21-30); a C caller
// can hand one over, and the lazy butterflies then return words that are simply wrong.  With
// FHE_NTT_CHECK_CANONICAL=1 in the environment (or fhe_ntt_set_check_canonical(1)) every transform /
// product entry point first scans its coefficient inputs on the device and returns
// FHE_E_NOT_CANONICAL instead.  It synchronises the stream, so it is a debugging aid, off by default.
static std::mutex g_check_lock;
static int g_check_canonical = -1;   // -1: not read from the environment yet
static bool check_canonical_on() {
    std::lock_guard<std::mutex> lk(g_check_lock);
    if (g_check_canonical < 0) {
        const char *e = getenv("FHE_NTT_CHECK_CANONICAL");
        g_check_canonical = (e && e[0] == '1') ? 1 : 0;
    }
    return g_check_canonical == 1;
}
extern "C" int fhe_ntt_set_check_canonical(int on) {
    std::lock_guard<std::mutex> lk(g_check_lock);
    g_check_canonical = on ? 1 : 0;
    return FHE_OK;
}
static int check_canonical_dev(const fhe_ntt_plan *plan, const void *d_x, size_t count, hipStream_t st, const char *who) {
    return fhe_check_canonical_words(plan->q, d_x, count, st, who);
}
// the same check for entry points that have a modulus but no plan (zring.hip: the BFV products read canonical words as
// their own residues when q is below every prime in use — FHE_NTT_CHECK_CANONICAL=1 makes them verify that contract)
int fhe_check_canonical_words(uint64_t q, const void *d_x, size_t count, hipStream_t st, const char *who) {
    if (!check_canonical_on() || !d_x || count == 0) return FHE_OK;
    void *df = nullptr;
    size_t cap = 0;
    int dev = 0;
    int rc = fhe_stage_acquire(sizeof(int), &df, &cap, &dev);
    if (rc != FHE_OK) return rc;
    int flag = 0;
    hipError_t e = hipMemsetAsync(df, 0, sizeof(int), st);
    if (e == hipSuccess) e = fhe::launch_check_canonical((const u64 *)d_x, count, q, (int *)df, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&flag, df, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    else (void)hipStreamSynchronize(st);
    fhe_stage_release(df, cap, dev);
    if (e != hipSuccess) return hip_fail(e, "canonical check");
    if (flag) return fail(FHE_E_NOT_CANONICAL, "%s: an input coefficient >= q=%llu (FHE_NTT_CHECK_CANONICAL)", who,
                          (unsigned long long)q);
    return FHE_OK;
}

// polynomials per launch for two-pass sizes
static std::mutex g_cfg_lock;
static size_t g_batch_tile_override = 0;
// Default for the two-pass sizes: 1 GiB of coefficients per launch pair (2048 polynomials at n = 2^16).  Rounds 1-2 ran the
// whole batch in one launch per pass (VALU-bound kernels: 128-polynomial, Infinity-Cache-sized tiles 9.9 ms per 16384, 2048
// 8.5 ms, untiled 8.3 ms).  With the passes memory-bound (round 3) the contiguous pass runs 8-11 % faster on a 32 k-workgroup
// grid that follows its strided pass than on a 1 M-workgroup one: 65536 polynomials at n = 2^16, three repetitions on one box,
// ms per step: untiled 24.74-25.04, 1024-polynomial tiles 24.26-24.39, 1536 24.08-24.18, 2048 24.01-24.07, 3072 24.00-24.28;
// tiles the size of the Infinity Cache still lose (256: 26.3, 512: 25.0: launch ramp and tail).
static constexpr u64 kMaxTilePolys = 1ull << 22;  // keeps every grid below 2^31 workgroups up to n = 2^20
static constexpr unsigned kTileCoeffLog = 27;      // 2^27 coefficients = 1 GiB per launch pair

u64 fhe_batch_tile_for(const fhe_ntt_plan *plan) {
    size_t ov;
    {
        std::lock_guard<std::mutex> lk(g_cfg_lock);
        ov = g_batch_tile_override;
    }
    if (ov == 0) {
        static const size_t env = [] {
            const char *s = getenv("FHE_NTT_BATCH_TILE");
            return s ? (size_t)strtoull(s, nullptr, 10) : (size_t)0;
        }();
        ov = env;
    }
    if (ov) return ov;
    if (plan && (int)plan->log_n > fhe::kMaxSinglePassLog && plan->log_n < kTileCoeffLog) return 1ull << (kTileCoeffLog - plan->log_n);
    return plan && plan->log_n >= kTileCoeffLog ? 1 : kMaxTilePolys;
}

extern "C" int fhe_ntt_set_batch_tile(size_t polys) {
    std::lock_guard<std::mutex> lk(g_cfg_lock);
    g_batch_tile_override = polys;
    return FHE_OK;
}

// ---- the one-launch forward transform (ntt_persist.hip) ---------------------------------------------------------
// FHE_NTT_PERSIST (or fhe_ntt_set_persist): n = 2^16 transforms on a pseudo-Mersenne modulus run as ONE launch of
// persistent workgroups.  "A:T,L,R" — tiles of T polynomials, the strided stages running L tiles ahead of the contiguous
// ones, the intermediate in a ring of R tile slots per XCD (R = 0: in the output buffer).  "B:R" — teams: sixteen
// workgroups of one XCD take one polynomial through both halves, the intermediate in a ring of R polynomial slots per
// XCD, read back out of the L2.  Unset: the two-pass kernels.
static bool g_persist_set = false;
static fhe::PersistTune g_persist{};
static bool g_persist_on = false;
static int g_persist_grid = -1;                      // > 0: workgroups to launch instead of what the chip holds; -1: not read yet
static uint32_t *g_persist_host_err = nullptr;      // pinned, device-visible: a bounded wait that ran out lands here
static bool persist_tune(fhe::PersistTune *t) {
    std::lock_guard<std::mutex> lk(g_cfg_lock);
    if (!g_persist_set) {
        g_persist_set = true;
        const char *e = getenv("FHE_NTT_PERSIST");       // "A:T,L,R" (tiles, lagged) or "B:R" (teams)
        unsigned T = 0, L = 1, R = 4;
        unsigned stagger = 0;
        if (e && strchr("BbDdEe", e[0]) && e[1] == ':' && sscanf(e + 2, "%u,%u", &R, &stagger) >= 1 && R >= ((e[0] == 'E' || e[0] == 'e') ? 2u : 1u)) {
            g_persist = fhe::PersistTune{};
            g_persist.log_t = 0; g_persist.lag = stagger; g_persist.ringslots = R; g_persist.teams = true;
            g_persist.deep = e[0] == 'D' || e[0] == 'd';      // "D:R,s": the teams at two workgroups per CU
            g_persist.flow = e[0] == 'E' || e[0] == 'e';      // "E:R,s": ... that never wait (a FIFO of pending parts per workgroup)
            g_persist_on = true;
        } else if (e && (e[0] == 'A' || e[0] == 'a') && e[1] == ':' && sscanf(e + 2, "%u,%u,%u", &T, &L, &R) >= 1 && T > 0 &&
                   (T & (T - 1)) == 0 && T <= 1024 && (R == 0 || R >= L + 1)) {
            g_persist = fhe::PersistTune{};
            while ((1u << g_persist.log_t) < T) g_persist.log_t++;
            g_persist.lag = L; g_persist.ringslots = R;
            g_persist_on = true;
        }
    }
    *t = g_persist;
    return g_persist_on;
}
extern "C" int fhe_ntt_set_persist(unsigned mode, unsigned tile_polys, unsigned lag, unsigned ringslots) {
    if (mode > 4) return fail(FHE_E_INVALID, "fhe_ntt_set_persist: mode %u (0 off, 1 = A: lagged tiles, 2 = B: teams, 3 = D: teams at two workgroups per CU, 4 = E: teams that never wait)", mode);
    if (mode == 4 && ringslots < 2) return fail(FHE_E_INVALID, "fhe_ntt_set_persist: mode 4 needs at least two ring slots per group");
    if (mode == 1 && (tile_polys == 0 || (tile_polys & (tile_polys - 1)) != 0 || tile_polys > 1024))
        return fail(FHE_E_INVALID, "fhe_ntt_set_persist: tile of %u polynomials (need a power of two <= 1024)", tile_polys);
    if (mode == 1 && ringslots && ringslots < lag + 1)
        return fail(FHE_E_INVALID, "fhe_ntt_set_persist: a ring of %u slots cannot hold a lag of %u tiles (need >= lag + 1)", ringslots, lag);
    if (mode >= 2 && ringslots == 0) return fail(FHE_E_INVALID, "fhe_ntt_set_persist: teams need a ring (ringslots >= 1)");
    std::lock_guard<std::mutex> lk(g_cfg_lock);
    g_persist_set = true;
    g_persist_on = mode != 0;
    g_persist = fhe::PersistTune{};
    if (mode == 1) {
        while ((1u << g_persist.log_t) < tile_polys) g_persist.log_t++;
        g_persist.lag = lag; g_persist.ringslots = ringslots;
    } else if (mode >= 2) {
        g_persist.log_t = 0; g_persist.lag = lag; g_persist.ringslots = ringslots; g_persist.teams = true;   // lag: start-up stagger
        g_persist.deep = mode == 3;
        g_persist.flow = mode == 4;
    }
    return FHE_OK;
}
// Diagnostic: with a device buffer of 26 u64 words registered here, lane 0 of every persistent workgroup adds the shader-
// clock ticks it spent in each part of its iterations (by item kind: 12 + 12 words), and the items it ran (2 words).
static void *g_persist_prof = nullptr;
extern "C" int fhe_ntt_persist_profile(void *d_words26) {
    std::lock_guard<std::mutex> lk(g_cfg_lock);
    g_persist_prof = d_words26;
    return FHE_OK;
}
extern "C" int fhe_ntt_set_persist_grid(int workgroups) {
    std::lock_guard<std::mutex> lk(g_cfg_lock);
    g_persist_grid = workgroups > 0 ? workgroups : 0;
    return FHE_OK;
}
// Everything forward_persist needs from the process-wide settings, taken in ONE critical section (the grid
// override, the profile buffer and the error word were read outside g_cfg_lock): the pinned error word (allocated on first
// use, released in fhe_ntt_shutdown), whether an earlier launch left it set, the workgroup count for this (device,
// variant) — an occupancy query, cached per DEVICE — or its override, and the profile buffer.
struct PersistSnap {
    uint32_t *d_err = nullptr;
    uint32_t pending = 0;
    unsigned grid = 0;
    void *prof = nullptr;
};
static std::map<std::pair<int, int>, unsigned> g_persist_grids;      // (device, variant) -> workgroups the chip holds
static int persist_snapshot(const fhe::PersistTune &tune, PersistSnap *s) {
    int dev = 0;
    int rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    std::lock_guard<std::mutex> lk(g_cfg_lock);
    if (!g_persist_host_err) {
        void *h = nullptr;
        HIP_TRY(hipHostMalloc(&h, 64, hipHostMallocMapped));
        memset(h, 0, 64);
        g_persist_host_err = (uint32_t *)h;
    }
    void *d = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&d, g_persist_host_err, 0));
    s->d_err = (uint32_t *)d;
    s->pending = *(volatile uint32_t *)g_persist_host_err;
    if (g_persist_grid < 0) { const char *e = getenv("FHE_NTT_PERSIST_GRID"); g_persist_grid = e ? atoi(e) : 0; }
    if (g_persist_grid > 0) {
        s->grid = (unsigned)g_persist_grid;      // tests: fewer workgroups than the chip holds
    } else {
        unsigned &grid = g_persist_grids[{dev, tune.teams ? ((tune.deep || tune.flow) ? 2 : 1) : 0}];
        if (!grid) {
            unsigned g = 0;
            HIP_TRY(fhe::persist_grid(tune, &g));
            grid = g;
        }
        s->grid = grid;
    }
    s->prof = g_persist_prof;
    return FHE_OK;
}
static void persist_free_all() {
    std::lock_guard<std::mutex> lk(g_cfg_lock);
    if (g_persist_host_err) (void)hipHostFree(g_persist_host_err);
    g_persist_host_err = nullptr;
    g_persist_grids.clear();
}
// FHE_OK, or FHE_E_HIP when a persistent launch that has FINISHED gave up a bounded wait (its outputs are then not valid);
// reading clears the word.  Call after synchronising the stream.
extern "C" int fhe_ntt_persist_status(void) {
    uint32_t v = 0;
    {
        std::lock_guard<std::mutex> lk(g_cfg_lock);
        if (g_persist_host_err) { v = *(volatile uint32_t *)g_persist_host_err; *(volatile uint32_t *)g_persist_host_err = 0; }
    }
    if (v) return fail(FHE_E_HIP, "persistent transform: a bounded wait ran out (bits 0x%x: 1 bind, 2 strided-done, 4 ring slot; 8: an XCD's queue was never served)", v);
    return FHE_OK;
}
// the lane-ordered table of the last four stages, built once per (plan, device)
static int persist_tables(const fhe_ntt_plan *plan, const fhe::DevicePlan &dp, const fhe::Tw **twc, const u64 **twc8, hipStream_t st) {
    int dev = 0;
    int rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    std::lock_guard<std::mutex> lk(plan->dev_lock);
    DeviceTables &t = plan->dev[dev];
    if (!t.twc_pm) {
        const size_t n = fhe::persist_twc_entries(plan->log_n);
        fhe::Tw *d = nullptr;
        HIP_TRY(hipMalloc((void **)&d, n * (sizeof(fhe::Tw) + sizeof(u64))));      // [Tw x n][u64 x n]
        u64 *d8 = (u64 *)(d + n);
        hipError_t e = fhe::launch_persist_twc(dp.tw_fwd_pm, d, d8, plan->log_n, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { (void)hipFree(d); return hip_fail(e, "persist_twc_kernel"); }
        t.twc_pm = d;
    }
    *twc = t.twc_pm;
    *twc8 = (const u64 *)(t.twc_pm + fhe::persist_twc_entries(plan->log_n));
    return FHE_OK;
}
static int forward_persist(const fhe_ntt_plan *plan, const fhe::DevicePlan &dp, const fhe::PersistTune &tune, const void *d_in,
                           void *d_out, size_t batch, hipStream_t st) {
    PersistSnap cfg;
    int rc = persist_snapshot(tune, &cfg);
    if (rc != FHE_OK) return rc;
    if (cfg.pending) return fail(FHE_E_HIP, "an earlier persistent transform failed (fhe_ntt_persist_status())");
    const fhe::Tw *twc = nullptr;
    const u64 *twc8 = nullptr;
    if ((rc = persist_tables(plan, dp, &twc, &twc8, st)) != FHE_OK) return rc;
    const size_t cb = (fhe::persist_ctl_bytes(tune, batch, cfg.grid) + 255) & ~(size_t)255, rb = fhe::persist_ring_bytes(tune, cfg.grid);
    void *w = nullptr;
    if ((rc = fhe_workspace_get(4, cb + rb, st, &w)) != FHE_OK) return rc;
    hipError_t e = fhe::launch_ntt_forward_persist(dp, twc, twc8, (const u64 *)d_in, (u64 *)d_out, batch, tune, (uint32_t *)w,
                                                   rb ? (u64 *)((char *)w + cb) : nullptr, cfg.d_err, (u64 *)cfg.prof, cfg.grid, st);
    if (e != hipSuccess) return hip_fail(e, "launch_ntt_forward_persist");
    return FHE_OK;
}

// Grow-only library workspaces, one per (slot, device, STREAM, calling THREAD): slot 0 = fhe_rq_mul_dev(d_work = NULL)
// and the tensor result of fhe_bfv_mul_dev, slot 1 = zring / glue intermediates, slot 3 = small-modulus scratch,
// slot 4 = the persistent transform's control block and ring.  The contents belong to one call; a thread's calls on a
// stream are ordered by the stream, and no two threads or streams share a buffer, so the *_dev entry points may be issued
// concurrently from any number of streams and threads — on the same stream too
// (tests/test_parity_gpu.py::test_two_streams_share_no_workspace, test_two_threads_on_the_default_stream).  Growing never frees a buffer that enqueued work may
// still be using: the old buffer is retired and released at fhe_ntt_shutdown().
// Round 5: a thread that EXITS hands its buffers to an orphan pool of their (slot, device, stream) — a
// thread_local object's destructor does it — and the next thread that needs a workspace of that (slot, device, stream)
// adopts the best-fitting orphan instead of allocating: work the dead thread enqueued is ahead of the adopter's on the
// SAME stream, so reuse needs no further ordering, and a pool of short-lived threads on one stream holds as many
// buffers as ran concurrently, not as many as ever lived.
struct Workspace {
    void *ptr = nullptr;
    size_t bytes = 0;
};
struct WsKey {
    int slot, dev;
    hipStream_t st;
    size_t tid;
    bool operator<(const WsKey &o) const {
        if (slot != o.slot) return slot < o.slot;
        if (dev != o.dev) return dev < o.dev;
        if (st != o.st) return st < o.st;
        return tid < o.tid;
    }
};
static std::mutex g_ws_lock;
static std::map<WsKey, Workspace> g_ws;
static std::vector<void *> g_ws_retired;
static bool g_ws_alive = true;               // false once the process is tearing the statics down
constexpr size_t kWsOrphanTid = ~(size_t)0;  // the tid under which a dead thread's buffers wait (a multimap would do; slots are few)
static std::multimap<WsKey, Workspace> g_ws_orphans;

static size_t ws_thread_id() { return std::hash<std::thread::id>()(std::this_thread::get_id()); }

namespace {
struct WsThreadReaper {      // one per thread that ever took a workspace; runs at thread exit
    bool armed = false;
    ~WsThreadReaper() {
        if (!armed || !g_ws_alive) return;
        const size_t tid = ws_thread_id();
        std::lock_guard<std::mutex> lk(g_ws_lock);
        for (auto it = g_ws.begin(); it != g_ws.end();) {
            if (it->first.tid == tid) {
                // hipStreamPerThread named THIS thread's own stream, which dies with it: nobody can adopt in stream order
                if (it->second.ptr) {
                    if (it->first.st == hipStreamPerThread) g_ws_retired.push_back(it->second.ptr);
                    else g_ws_orphans.emplace(WsKey{it->first.slot, it->first.dev, it->first.st, kWsOrphanTid}, it->second);
                }
                it = g_ws.erase(it);
            } else {
                ++it;
            }
        }
    }
};
struct WsStaticsGuard { ~WsStaticsGuard() { g_ws_alive = false; } } g_ws_statics_guard;   // destroyed before the maps above (reverse order)
thread_local WsThreadReaper g_ws_reaper;
}  // namespace

int fhe_workspace_get(int slot, size_t bytes, hipStream_t st, void **out) {
    int dev = 0;
    int rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    // keyed by the calling THREAD for every stream (round 4): two host threads that enqueue on the same stream — the NULL
    // default stream is the common case for a shim whose Rq operations are freely shareable — get different buffers, so
    // their kernel sequences may interleave on the stream without sharing an intermediate
    const size_t tid = ws_thread_id();
    g_ws_reaper.armed = true;
    std::lock_guard<std::mutex> lk(g_ws_lock);
    Workspace &w = g_ws[WsKey{slot, dev, st, tid}];
    if (w.bytes < bytes && st != hipStreamPerThread) {
        // adopt a dead thread's buffer of this (slot, device, stream) if one is large enough (the smallest such)
        auto range = g_ws_orphans.equal_range(WsKey{slot, dev, st, kWsOrphanTid});
        auto best = range.second;
        for (auto it = range.first; it != range.second; ++it)
            if (it->second.bytes >= bytes && (best == range.second || it->second.bytes < best->second.bytes)) best = it;
        if (best != range.second) {
            if (w.ptr) g_ws_retired.push_back(w.ptr);
            w = best->second;
            g_ws_orphans.erase(best);
        }
    }
    if (w.bytes < bytes) {
        // grow geometrically so that the buffers retired on the way sum to less than the live one
        size_t want = w.bytes + w.bytes / 2;
        if (want < bytes) want = bytes;
        void *p = nullptr;
        if (want != bytes && hipMalloc(&p, want) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            want = bytes;
        }
        if (!p) HIP_TRY(hipMalloc(&p, want));
        if (w.ptr) g_ws_retired.push_back(w.ptr);
        w.ptr = p;
        w.bytes = want;
    }
    *out = w.ptr;
    return FHE_OK;
}
// ---- staging pool for the host-buffer entry points -------------------------------------------
// A shim that swaps the bodies of NTT::ntt / Rq mul calls in once per polynomial; hipMalloc +
// hipFree (which drains the device) per call cost more than the transform.  Idle staging buffers
// are kept per device — at most kStageSlots of them, kStageCapBytes in total, none above
// kStageMaxOne — and handed out best-fit.  A buffer is released only after the stream that used
// it has been synchronised, so reuse needs no further ordering.
namespace {
struct StageBuf { void *ptr; size_t bytes; int dev; };
constexpr size_t kStageSlots = 16, kStageCapBytes = 1ull << 30, kStageMaxOne = 256ull << 20;
std::mutex g_stage_lock;
std::vector<StageBuf> g_stage_idle;
size_t g_stage_idle_bytes = 0;
}  // namespace

int fhe_stage_acquire(size_t bytes, void **out, size_t *got, int *dev_out) {
    int dev = 0;
    int rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    *dev_out = dev;
    if (bytes == 0) bytes = 16;
    {
        std::lock_guard<std::mutex> lk(g_stage_lock);
        size_t best = g_stage_idle.size();
        for (size_t i = 0; i < g_stage_idle.size(); i++) {
            const StageBuf &b = g_stage_idle[i];
            if (b.dev == dev && b.bytes >= bytes && b.bytes <= 4 * bytes + 4096 &&
                (best == g_stage_idle.size() || b.bytes < g_stage_idle[best].bytes))
                best = i;
        }
        if (best != g_stage_idle.size()) {
            *out = g_stage_idle[best].ptr;
            *got = g_stage_idle[best].bytes;
            g_stage_idle_bytes -= g_stage_idle[best].bytes;
            g_stage_idle.erase(g_stage_idle.begin() + best);
            return FHE_OK;
        }
    }
    // round small requests up so that nearby sizes share buffers
    size_t want = bytes < 4096 ? 4096 : bytes;
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, want));
    *out = p;
    *got = want;
    return FHE_OK;
}

void fhe_stage_release(void *ptr, size_t bytes, int dev) {
    if (!ptr) return;
    if (bytes <= kStageMaxOne) {
        std::lock_guard<std::mutex> lk(g_stage_lock);
        if (g_stage_idle.size() < kStageSlots && g_stage_idle_bytes + bytes <= kStageCapBytes) {
            g_stage_idle.push_back(StageBuf{ptr, bytes, dev});
            g_stage_idle_bytes += bytes;
            return;
        }
    }
    (void)hipFree(ptr);
}

static void stage_free_all() {
    std::lock_guard<std::mutex> lk(g_stage_lock);
    for (auto &b : g_stage_idle) (void)hipFree(b.ptr);
    g_stage_idle.clear();
    g_stage_idle_bytes = 0;
}

// A caller that creates and destroys streams hands their workspaces back here (after synchronising the stream and
// before destroying it); without the call they live until fhe_ntt_shutdown().
extern "C" int fhe_ntt_release_stream_workspace(void *hip_stream) {
    int dev = 0;
    int rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    HIP_TRY(hipStreamSynchronize(st));
    // hipStreamPerThread names a different stream in every thread: only the caller's buffers; an explicit stream is
    // about to be destroyed: the buffers of EVERY thread that used it
    const bool per_thread = st == hipStreamPerThread;
    const size_t tid = std::hash<std::thread::id>()(std::this_thread::get_id());
    std::lock_guard<std::mutex> lk(g_ws_lock);
    for (auto it = g_ws.begin(); it != g_ws.end();) {        // every slot of this (device, stream[, thread])
        if (it->first.dev == dev && it->first.st == st && (!per_thread || it->first.tid == tid)) {
            if (it->second.ptr) (void)hipFree(it->second.ptr);
            it = g_ws.erase(it);
        } else {
            ++it;
        }
    }
    for (auto it = g_ws_orphans.begin(); it != g_ws_orphans.end();) {   // and what dead threads left on it
        if (it->first.dev == dev && it->first.st == st) {
            if (it->second.ptr) (void)hipFree(it->second.ptr);
            it = g_ws_orphans.erase(it);
        } else {
            ++it;
        }
    }
    return FHE_OK;
}

// bytes of library workspace currently held on every device (live buffers; buffers retired by a growth are released at
// fhe_ntt_shutdown() and not counted)
extern "C" size_t fhe_ntt_workspace_bytes(void) {
    std::lock_guard<std::mutex> lk(g_ws_lock);
    size_t total = 0;
    for (auto &kv : g_ws) total += kv.second.bytes;
    for (auto &kv : g_ws_orphans) total += kv.second.bytes;      // left by threads that exited, waiting for adoption
    return total;
}

void fhe_workspace_free_all() {
    stage_free_all();
    std::lock_guard<std::mutex> lk(g_ws_lock);
    for (auto &kv : g_ws)
        if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    g_ws.clear();
    for (auto &kv : g_ws_orphans)
        if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    g_ws_orphans.clear();
    for (void *p : g_ws_retired) (void)hipFree(p);
    g_ws_retired.clear();
}

// ---------------------------------------------------------------------------
// kernel timing (bench.py's roofline leg)
// ---------------------------------------------------------------------------
struct TimingSlot {
    std::string name;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double total_ms = 0;
    uint64_t launches = 0;
};
static std::mutex g_timing_lock;
static bool g_timing_on = false;
static std::map<std::string, TimingSlot> g_timing;

fhe::KernelTimer::KernelTimer(const char *name, int tag, hipStream_t st) : slot_(nullptr), st_(st) {
    if (!g_timing_on) return;
    std::lock_guard<std::mutex> lk(g_timing_lock);
    char key[64];
    snprintf(key, sizeof(key), "%s_%d", name, tag);
    TimingSlot &s = g_timing[key];
    s.name = key;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    (void)hipEventRecord(a, st);
    s.events.emplace_back(a, b);
    slot_ = &s;
}
fhe::KernelTimer::~KernelTimer() {
    if (!slot_) return;
    std::lock_guard<std::mutex> lk(g_timing_lock);
    TimingSlot *s = static_cast<TimingSlot *>(slot_);
    (void)hipEventRecord(s->events.back().second, st_);
}

static void timing_drain_locked() {
    for (auto &kv : g_timing) {
        TimingSlot &s = kv.second;
        for (auto &ev : s.events) {
            float ms = 0;
            if (hipEventSynchronize(ev.second) == hipSuccess &&
                hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
                s.total_ms += ms;
                s.launches++;
            }
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        s.events.clear();
    }
}

extern "C" int fhe_ntt_kernel_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(g_timing_lock);
    g_timing_on = on != 0;
    return FHE_OK;
}
extern "C" int fhe_ntt_kernel_timing_reset(void) {
    std::lock_guard<std::mutex> lk(g_timing_lock);
    timing_drain_locked();
    g_timing.clear();
    return FHE_OK;
}
extern "C" int fhe_ntt_kernel_timing_read(char *names, double *total_ms, uint64_t *launches, int cap) {
    std::lock_guard<std::mutex> lk(g_timing_lock);
    timing_drain_locked();
    int i = 0;
    for (auto &kv : g_timing) {
        if (i < cap) {
            if (names) {
                strncpy(names + (size_t)i * 64, kv.second.name.c_str(), 63);
                names[(size_t)i * 64 + 63] = 0;
            }
            if (total_ms) total_ms[i] = kv.second.total_ms;
            if (launches) launches[i] = kv.second.launches;
        }
        i++;
    }
    return i;
}

// small moduli (q < 2^30, 2^8 <= n <= 2^18): one 32-bit word per coefficient (smallq.hip); FHE_EXT32=0 keeps the 61-bit kernels
bool fhe_smallq_args(const fhe_ntt_plan *plan, const fhe::DevicePlan &dp, fhe::SmallQArgs *a) {
    if (!dp.tw32_fwd || !fhe_ext32_enabled()) return false;
    const u64 q = plan->q;
    a->tw_fwd = dp.tw32_fwd; a->tw_inv = dp.tw32_inv;
    a->q = (uint32_t)q; a->bq = (uint32_t)(0xffffffffull / q);
    uint32_t inv = (uint32_t)q;                                 // Newton: q^-1 mod 2^32 (q odd)
    for (int it = 0; it < 5; it++) inv *= 2u - (uint32_t)q * inv;
    a->qinv_neg = 0u - inv;
    a->ninv = fhe::Tw32{(uint32_t)plan->n_inv, (uint32_t)((plan->n_inv << 32) / q)};
    const u64 nm = (plan->n_inv << 32) % q;
    a->ninv_mont = fhe::Tw32{(uint32_t)nm, (uint32_t)((nm << 32) / q)};
    a->mu = ~0ull / q;
    a->loose = fhe::smallq_loose(q) ? 1u : 0u;
    return true;
}
// the u32 buffer between the passes of the two-pass sizes: its own workspace slot (the callers' slots 0 / 1 stay theirs)
int fhe_smallq_scratch(unsigned log_n, uint64_t rows, hipStream_t st, fhe::SmallQArgs *a) {
    a->mid = nullptr;
    const size_t bytes = fhe::smallq_scratch_bytes(log_n, rows);
    if (!bytes) return FHE_OK;
    void *w = nullptr;
    int rc = fhe_workspace_get(3, bytes, st, &w);
    if (rc == FHE_OK) a->mid = (uint32_t *)w;
    return rc;
}

// ---------------------------------------------------------------------------
// device-resident entry points
// ---------------------------------------------------------------------------
extern "C" int fhe_ntt_forward_dev(const fhe_ntt_plan *plan, const void *d_in, void *d_out,
                                   size_t batch, void *hip_stream) {
    if (!plan) return fail(FHE_E_NULL, "plan is NULL");
    if (batch == 0) return FHE_OK;
    if (!d_in || !d_out) return fail(FHE_E_NULL, "fhe_ntt_forward_dev: NULL buffer");
    REQUIRE_ALIGNED(d_in);
    REQUIRE_ALIGNED(d_out);
    fhe::DevicePlan dp;
    int rc = fhe_device_plan(plan, &dp);
    if (rc != FHE_OK) return rc;
    if ((rc = check_canonical_dev(plan, d_in, batch * plan->n, (hipStream_t)hip_stream, "fhe_ntt_forward_dev")) != FHE_OK) return rc;
    fhe::SmallQArgs sq{};
    if (fhe_smallq_args(plan, dp, &sq)) {
        sq.a = (const u64 *)d_in; sq.out = (u64 *)d_out; sq.rows = batch;
        if ((rc = fhe_smallq_scratch(dp.log_n, batch, (hipStream_t)hip_stream, &sq)) != FHE_OK) return rc;
        hipError_t se = fhe::launch_sq_forward(sq, (int)dp.log_n, (hipStream_t)hip_stream);
        return se == hipSuccess ? FHE_OK : hip_fail(se, "sq_forward_kernel");
    }
    fhe::PersistTune tune;
    if (persist_tune(&tune) && fhe::persist_supported(dp))
        return forward_persist(plan, dp, tune, d_in, d_out, batch, (hipStream_t)hip_stream);
    hipError_t e = fhe::launch_ntt_forward(dp, (const u64 *)d_in, (u64 *)d_out, batch,
                                           fhe_batch_tile_for(plan), (hipStream_t)hip_stream);
    if (e != hipSuccess) return hip_fail(e, "launch_ntt_forward");
    return FHE_OK;
}

extern "C" int fhe_ntt_inverse_dev(const fhe_ntt_plan *plan, const void *d_in, void *d_out,
                                   size_t batch, void *hip_stream) {
    if (!plan) return fail(FHE_E_NULL, "plan is NULL");
    if (batch == 0) return FHE_OK;
    if (!d_in || !d_out) return fail(FHE_E_NULL, "fhe_ntt_inverse_dev: NULL buffer");
    REQUIRE_ALIGNED(d_in);
    REQUIRE_ALIGNED(d_out);
    fhe::DevicePlan dp;
    int rc = fhe_device_plan(plan, &dp);
    if (rc != FHE_OK) return rc;
    if ((rc = check_canonical_dev(plan, d_in, batch * plan->n, (hipStream_t)hip_stream, "fhe_ntt_inverse_dev")) != FHE_OK) return rc;
    fhe::SmallQArgs sq{};
    if (fhe_smallq_args(plan, dp, &sq)) {
        sq.a = (const u64 *)d_in; sq.out = (u64 *)d_out; sq.rows = batch;
        if ((rc = fhe_smallq_scratch(dp.log_n, batch, (hipStream_t)hip_stream, &sq)) != FHE_OK) return rc;
        hipError_t se = fhe::launch_sq_inverse(sq, (int)dp.log_n, (hipStream_t)hip_stream);
        return se == hipSuccess ? FHE_OK : hip_fail(se, "sq_inverse_kernel");
    }
    hipError_t e = fhe::launch_ntt_inverse(dp, (const u64 *)d_in, nullptr, nullptr, (u64 *)d_out,
                                           batch, fhe_batch_tile_for(plan), (hipStream_t)hip_stream);
    if (e != hipSuccess) return hip_fail(e, "launch_ntt_inverse");
    return FHE_OK;
}

extern "C" size_t fhe_rq_mul_workspace_bytes(const fhe_ntt_plan *plan, size_t batch) {
    if (!plan) return 0;
    return 2 * batch * plan->n * sizeof(u64);
}

extern "C" int fhe_rq_mul_dev(const fhe_ntt_plan *plan, const void *d_a, int a_is_evals,
                              const void *d_b, int b_is_evals, void *d_c, void *d_c_evals,
                              void *d_a_evals_out, void *d_b_evals_out, size_t batch, void *d_work,
                              void *hip_stream) {
    if (!plan) return fail(FHE_E_NULL, "plan is NULL");
    if (batch == 0) return FHE_OK;
    if (!d_a || !d_b || !d_c) return fail(FHE_E_NULL, "fhe_rq_mul_dev: NULL operand");
    REQUIRE_ALIGNED(d_a);
    REQUIRE_ALIGNED(d_b);
    REQUIRE_ALIGNED(d_c);
    REQUIRE_ALIGNED(d_c_evals);
    REQUIRE_ALIGNED(d_a_evals_out);
    REQUIRE_ALIGNED(d_b_evals_out);
    REQUIRE_ALIGNED(d_work);
    fhe::DevicePlan dp;
    int rc = fhe_device_plan(plan, &dp);
    if (rc != FHE_OK) return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t elems = batch * plan->n, bytes = elems * sizeof(u64);
    const u64 tile = fhe_batch_tile_for(plan);
    if ((rc = check_canonical_dev(plan, d_a, elems, st, "fhe_rq_mul_dev (a)")) != FHE_OK) return rc;
    if ((rc = check_canonical_dev(plan, d_b, elems, st, "fhe_rq_mul_dev (b)")) != FHE_OK) return rc;

    // small modulus, plain product (no cached evals in or out): the whole product in 32-bit words (smallq.hip)
    fhe::SmallQArgs sq{};
    if (fhe_smallq_args(plan, dp, &sq)) {
        sq.a = (const u64 *)d_a; sq.b = (const u64 *)d_b; sq.out = (u64 *)d_c; sq.rows = batch;
        sq.flags = (a_is_evals ? 1u : 0u) | (b_is_evals ? 2u : 0u);
        sq.c_evals = (u64 *)d_c_evals; sq.a_evals = (u64 *)d_a_evals_out; sq.b_evals = (u64 *)d_b_evals_out;
        if (d_work && dp.log_n > 14) sq.mid = (uint32_t *)d_work;                                // caller-owned scratch (e.g. inside a capture): 2 * batch * n * 8 bytes hold both
        else if ((rc = fhe_smallq_scratch(dp.log_n, 2 * batch, st, &sq)) != FHE_OK) return rc;  // two operands' u32 intermediates
        if (sq.mid) sq.mid_b = sq.mid + ((u64)batch << dp.log_n);
        hipError_t se = fhe::launch_sq_rq_mul(sq, (int)dp.log_n, st);
        return se == hipSuccess ? FHE_OK : hip_fail(se, "sq_rq_mul_kernel");
    }
    // single-pass sizes: the whole product in one kernel (both forward transforms, the pointwise
    // product and the inverse transform stay on chip); FHE_RQ_MUL_FUSED=0 selects the three-kernel path
    static const bool fused_on = [] {
        const char *e = getenv("FHE_RQ_MUL_FUSED");
        return !(e && e[0] == '0');
    }();
    if (fused_on) {
        hipError_t fe = fhe::launch_rq_mul_fused(dp, (const u64 *)d_a, a_is_evals != 0, (const u64 *)d_b, b_is_evals != 0,
                                                 (u64 *)d_c, (u64 *)d_c_evals, (u64 *)d_a_evals_out,
                                                 (u64 *)d_b_evals_out, batch, st);
        if (fe == hipSuccess) return FHE_OK;
        if (fe != hipErrorNotSupported) return hip_fail(fe, "launch_rq_mul_fused");
        (void)hipGetLastError();
    }

    const bool need_wa = !a_is_evals && !d_a_evals_out;
    const bool need_wb = !b_is_evals && !d_b_evals_out;
    u64 *work = (u64 *)d_work;
    if ((need_wa || need_wb) && !work) {
        void *w = nullptr;
        rc = fhe_workspace_get(0, 2 * bytes, st, &w);
        if (rc != FHE_OK) return rc;
        work = (u64 *)w;
    }
    // an operand that already is evals keeps them as they are (ring_nq.rs:590-599); a copy if the caller
    // wants them in a second place
    if (a_is_evals && d_a_evals_out && d_a_evals_out != d_a)
        HIP_TRY(hipMemcpyAsync(d_a_evals_out, d_a, bytes, hipMemcpyDeviceToDevice, st));
    if (b_is_evals && d_b_evals_out && d_b_evals_out != d_b)
        HIP_TRY(hipMemcpyAsync(d_b_evals_out, d_b, bytes, hipMemcpyDeviceToDevice, st));

    // two-pass sizes: strided(a) | strided(b) in one launch, one middle kernel (both contiguous forward
    // passes, the pointwise product, the contiguous inverse pass), strided inverse.  The product must not
    // overwrite an operand's evals the middle kernel still has to read: c aliasing an evals INPUT is the
    // one case that takes the unfused chain below.  FHE_RQ_MUL_FUSED=0 selects that chain too.
    u64 *wa = d_a_evals_out ? (u64 *)d_a_evals_out : work;
    u64 *wb = d_b_evals_out ? (u64 *)d_b_evals_out : (work ? work + elems : nullptr);
    const bool c_aliases_evals_in = (a_is_evals && d_c == d_a) || (b_is_evals && d_c == d_b);
    if (fused_on && !c_aliases_evals_in) {
        hipError_t fe = fhe::launch_rq_mul_two_pass(dp, (const u64 *)d_a, a_is_evals != 0, (const u64 *)d_b, b_is_evals != 0,
                                                    (u64 *)d_c, (u64 *)d_c_evals, wa, d_a_evals_out != nullptr, wb,
                                                    d_b_evals_out != nullptr, batch, tile, st);
        if (fe == hipSuccess) return FHE_OK;
        if (fe != hipErrorNotSupported) return hip_fail(fe, "launch_rq_mul_two_pass");
        (void)hipGetLastError();
    }
    // A = NTT(a): the operand's cached evals if it has them
    const u64 *A = (const u64 *)d_a;
    if (!a_is_evals) {
        hipError_t e = fhe::launch_ntt_forward(dp, (const u64 *)d_a, wa, batch, tile, st);
        if (e != hipSuccess) return hip_fail(e, "forward(a)");
        A = wa;
    }
    const u64 *B = (const u64 *)d_b;
    if (!b_is_evals) {
        hipError_t e = fhe::launch_ntt_forward(dp, (const u64 *)d_b, wb, batch, tile, st);
        if (e != hipSuccess) return hip_fail(e, "forward(b)");
        B = wb;
    }
    // c = intt(A .* B), C = A .* B optionally kept (ring_nq.rs:601-606)
    hipError_t e = fhe::launch_ntt_inverse(dp, A, B, (u64 *)d_c_evals, (u64 *)d_c, batch, tile, st);
    if (e != hipSuccess) return hip_fail(e, "inverse(A.*B)");
    return FHE_OK;
}

extern "C" int fhe_rq_pointwise_mul_dev(const fhe_ntt_plan *plan, const void *d_a, const void *d_b,
                                        void *d_c, size_t batch, void *hip_stream) {
    if (!plan) return fail(FHE_E_NULL, "plan is NULL");
    if (batch == 0) return FHE_OK;
    if (!d_a || !d_b || !d_c) return fail(FHE_E_NULL, "fhe_rq_pointwise_mul_dev: NULL buffer");
    fhe::DevicePlan dp;
    int rc = fhe_device_plan(plan, &dp);
    if (rc != FHE_OK) return rc;
    if ((rc = check_canonical_dev(plan, d_a, batch * plan->n, (hipStream_t)hip_stream, "fhe_rq_pointwise_mul_dev (a)")) != FHE_OK) return rc;
    if ((rc = check_canonical_dev(plan, d_b, batch * plan->n, (hipStream_t)hip_stream, "fhe_rq_pointwise_mul_dev (b)")) != FHE_OK) return rc;
    hipError_t e = fhe::launch_pointwise_mul(dp, (const u64 *)d_a, (const u64 *)d_b, (u64 *)d_c,
                                             batch * plan->n, (hipStream_t)hip_stream);
    if (e != hipSuccess) return hip_fail(e, "launch_pointwise_mul");
    return FHE_OK;
}

extern "C" int fhe_fill_synthetic_dev(uint64_t q, uint64_t seed, uint64_t first_index, size_t count,
                                      void *d_out, void *hip_stream) {
    if (count == 0) return FHE_OK;
    if (!d_out) return fail(FHE_E_NULL, "fhe_fill_synthetic_dev: NULL buffer");
    if (q == 0) return fail(FHE_E_BAD_Q, "q is 0");
    int dev;
    int rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    hipError_t e = fhe::launch_fill_synthetic((u64 *)d_out, count, q, seed, first_index,
                                              (hipStream_t)hip_stream);
    if (e != hipSuccess) return hip_fail(e, "launch_fill_synthetic");
    return FHE_OK;
}

// ---------------------------------------------------------------------------
// host-buffer entry points: stage through device memory around the same kernels
// ---------------------------------------------------------------------------
// Every host entry point synchronises its stream before it returns; on an error path the
// buffer may still be in use, so it is drained before going back to the pool.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int dev = 0;          // the device the buffer lives on (the caller's current device at alloc)
    bool clean = false;   // set once the stream has been synchronised after the last use
    ~DevBuf() {
        if (!p) return;
        if (!clean) (void)hipStreamSynchronize(hipStreamPerThread);
        fhe_stage_release(p, cap, dev);
    }
    int alloc(size_t bytes) { return fhe_stage_acquire(bytes, &p, &cap, &dev); }
};

static int host_transform(const fhe_ntt_plan *plan, const uint64_t *in, uint64_t *out, size_t batch,
                          bool inverse) {
    if (!plan) return fail(FHE_E_NULL, "plan is NULL");
    if (batch == 0) return FHE_OK;
    if (!in || !out) return fail(FHE_E_NULL, "NULL buffer");
    int dev;
    int rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    const size_t bytes = batch * plan->n * sizeof(u64);
    DevBuf d;
    if ((rc = d.alloc(bytes)) != FHE_OK) return rc;
    hipStream_t st = hipStreamPerThread;
    HIP_TRY(hipMemcpyAsync(d.p, in, bytes, hipMemcpyHostToDevice, st));
    rc = inverse ? fhe_ntt_inverse_dev(plan, d.p, d.p, batch, st)
                 : fhe_ntt_forward_dev(plan, d.p, d.p, batch, st);
    if (rc != FHE_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out, d.p, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    d.clean = true;
    // a persistent launch (FHE_NTT_PERSIST) that gave up a bounded wait left words that are not the transform
    if (!inverse && (rc = fhe_ntt_persist_status()) != FHE_OK) return rc;
    return FHE_OK;
}

extern "C" int fhe_ntt_forward(const fhe_ntt_plan *plan, const uint64_t *in, uint64_t *out,
                               size_t batch) {
    return host_transform(plan, in, out, batch, false);
}
extern "C" int fhe_ntt_inverse(const fhe_ntt_plan *plan, const uint64_t *in, uint64_t *out,
                               size_t batch) {
    return host_transform(plan, in, out, batch, true);
}

extern "C" int fhe_rq_mul(const fhe_ntt_plan *plan, const uint64_t *a, int a_is_evals,
                          const uint64_t *b, int b_is_evals, uint64_t *c, uint64_t *c_evals,
                          uint64_t *a_evals_out, uint64_t *b_evals_out, size_t batch) {
    if (!plan) return fail(FHE_E_NULL, "plan is NULL");
    if (batch == 0) return FHE_OK;
    if (!a || !b || !c) return fail(FHE_E_NULL, "fhe_rq_mul: NULL operand");
    int dev;
    int rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    const size_t bytes = batch * plan->n * sizeof(u64);
    DevBuf da, db, dc, dce;
    if ((rc = da.alloc(bytes)) != FHE_OK) return rc;
    if ((rc = db.alloc(bytes)) != FHE_OK) return rc;
    if ((rc = dc.alloc(bytes)) != FHE_OK) return rc;
    if (c_evals && (rc = dce.alloc(bytes)) != FHE_OK) return rc;
    hipStream_t st = hipStreamPerThread;
    HIP_TRY(hipMemcpyAsync(da.p, a, bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(db.p, b, bytes, hipMemcpyHostToDevice, st));
    // forward transforms run in place on the staged copies, which then ARE the evals
    rc = fhe_rq_mul_dev(plan, da.p, a_is_evals, db.p, b_is_evals, dc.p, dce.p,
                        a_is_evals ? nullptr : da.p, b_is_evals ? nullptr : db.p, batch, nullptr, st);
    if (rc != FHE_OK) return rc;
    HIP_TRY(hipMemcpyAsync(c, dc.p, bytes, hipMemcpyDeviceToHost, st));
    if (c_evals) HIP_TRY(hipMemcpyAsync(c_evals, dce.p, bytes, hipMemcpyDeviceToHost, st));
    if (a_evals_out) HIP_TRY(hipMemcpyAsync(a_evals_out, da.p, bytes, hipMemcpyDeviceToHost, st));
    if (b_evals_out) HIP_TRY(hipMemcpyAsync(b_evals_out, db.p, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    da.clean = db.clean = dc.clean = dce.clean = true;
    return FHE_OK;
}

extern "C" int fhe_rq_mul_checked(const fhe_ntt_plan *plan_a, const fhe_ntt_plan *plan_b,
                                  const uint64_t *a, const uint64_t *b, uint64_t *c,
                                  uint64_t *c_evals, size_t batch) {
    if (!plan_a || !plan_b) return fail(FHE_E_NULL, "plan is NULL");
    if (plan_a->q != plan_b->q || plan_a->n != plan_b->n)
        return fail(FHE_E_PARAM_MISMATCH,
                    "operands have different RingParam: (q=%llu,n=%llu) vs (q=%llu,n=%llu) "
                    "(assert_eq!(lhs.param, rhs.param), ring_nq.rs:565,587)",
                    (unsigned long long)plan_a->q, (unsigned long long)plan_a->n,
                    (unsigned long long)plan_b->q, (unsigned long long)plan_b->n);
    return fhe_rq_mul(plan_a, a, 0, b, 0, c, c_evals, nullptr, nullptr, batch);
}

extern "C" int fhe_rq_pointwise_mul(const fhe_ntt_plan *plan, const uint64_t *a, const uint64_t *b,
                                    uint64_t *c, size_t batch) {
    if (!plan) return fail(FHE_E_NULL, "plan is NULL");
    if (batch == 0) return FHE_OK;
    if (!a || !b || !c) return fail(FHE_E_NULL, "fhe_rq_pointwise_mul: NULL buffer");
    int dev;
    int rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    const size_t bytes = batch * plan->n * sizeof(u64);
    DevBuf da, db;
    if ((rc = da.alloc(bytes)) != FHE_OK) return rc;
    if ((rc = db.alloc(bytes)) != FHE_OK) return rc;
    hipStream_t st = hipStreamPerThread;
    HIP_TRY(hipMemcpyAsync(da.p, a, bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(db.p, b, bytes, hipMemcpyHostToDevice, st));
    rc = fhe_rq_pointwise_mul_dev(plan, da.p, db.p, da.p, batch, st);
    if (rc != FHE_OK) return rc;
    HIP_TRY(hipMemcpyAsync(c, da.p, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    da.clean = db.clean = true;
    return FHE_OK;
}

extern "C" int fhe_rq_check_canonical(const fhe_ntt_plan *plan, const uint64_t *x, size_t batch) {
    if (!plan) return fail(FHE_E_NULL, "plan is NULL");
    if (batch == 0) return FHE_OK;
    if (!x) return fail(FHE_E_NULL, "fhe_rq_check_canonical: NULL buffer");
    int dev;
    int rc = fhe_current_device(&dev);
    if (rc != FHE_OK) return rc;
    const size_t count = batch * plan->n, bytes = count * sizeof(u64);
    DevBuf dx, df;
    if ((rc = dx.alloc(bytes)) != FHE_OK) return rc;
    if ((rc = df.alloc(sizeof(int))) != FHE_OK) return rc;
    hipStream_t st = hipStreamPerThread;
    HIP_TRY(hipMemcpyAsync(dx.p, x, bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(df.p, 0, sizeof(int), st));
    hipError_t e = fhe::launch_check_canonical((const u64 *)dx.p, count, plan->q, (int *)df.p, st);
    if (e != hipSuccess) return hip_fail(e, "launch_check_canonical");
    int flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, df.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    dx.clean = df.clean = true;
    if (flag) return fail(FHE_E_NOT_CANONICAL, "a coefficient >= q=%llu was found", (unsigned long long)plan->q);
    return FHE_OK;
}

// ---------------------------------------------------------------------------
// misc
// ---------------------------------------------------------------------------
extern "C" int fhe_ntt_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count;
}

extern "C" const char *fhe_last_error(void) { return g_err; }
extern "C" const char *fhe_ntt_version(void) {
    // a library linked from a timing-only object (tools/abl_build.sh) computes wrong words by design: say so
    const bool ablated = fhe::ntt_kernels_ablated() || fhe::digit_mac_ablated() || fhe::digit32_ablated() || fhe::bfv32_ablated();
    return ablated ? "fhe_ntt 0.3 (gfx950) ABLATED" : "fhe_ntt 0.3 (gfx950)";
}

extern "C" int fhe_ntt_shutdown(void) {
    {
        std::lock_guard<std::mutex> lk(g_timing_lock);
        timing_drain_locked();
        g_timing.clear();
    }
    fhe_workspace_free_all();
    fhe_ext32_free_all();
    persist_free_all();
    std::lock_guard<std::mutex> lk(g_plans_lock);
    for (auto &kv : g_plans) {
        fhe_ntt_plan *p = kv.second.get();
        std::lock_guard<std::mutex> lk2(p->dev_lock);
        for (auto &t : p->dev) {
            if (t.tw_fwd) (void)hipFree(t.tw_fwd);
            if (t.tw_inv) (void)hipFree(t.tw_inv);
            if (t.digit_lut) (void)hipFree(t.digit_lut);
            if (t.tw32_fwd) (void)hipFree(t.tw32_fwd);
            if (t.tw32_inv) (void)hipFree(t.tw32_inv);
            if (t.tw_fwd_mg) (void)hipFree(t.tw_fwd_mg);
            if (t.tw_inv_mg) (void)hipFree(t.tw_inv_mg);
            if (t.tw_fwd_pm) (void)hipFree(t.tw_fwd_pm);
            if (t.tw_inv_pm) (void)hipFree(t.tw_inv_pm);
            if (t.twc_pm) (void)hipFree(t.twc_pm);
            t = DeviceTables();
        }
    }
    g_plans.clear();
    return FHE_OK;
}
