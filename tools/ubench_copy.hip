// tools/ubench_copy.hip — how fast can ANY kernel copy on this box?  (diagnostic for DESIGN.md section 5, round 3: the two pass
// kernels of the N = 2^16 transform run at the rate of tools/ubench_mem.hip's one-shot copy, 5.4-5.7 TB/s of read+write
// traffic; the architecture guide quotes 6.29 TB/s for a float4 copy.)  Variants: one element per thread, grid-stride
// persistent loops with 1 / 4 / 8 16-byte loads in flight, non-temporal, 1024-thread blocks, read-only, write-only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef uint4 V;

__global__ __launch_bounds__(256) void k_one(const V* __restrict__ a, V* __restrict__ b, u64 n) {
  u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = a[i];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_stride(const V* __restrict__ a, V* __restrict__ b, u64 n) {
  const u64 stride = (u64)gridDim.x * 256 * U;
  for (u64 i = (u64)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
    V v[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
      if (NT) { const u64* p = (const u64*)(a + i + k * 256); u64 x = __builtin_nontemporal_load(p), y = __builtin_nontemporal_load(p + 1); v[k] = V{(unsigned)x, (unsigned)(x >> 32), (unsigned)y, (unsigned)(y >> 32)}; }
      else v[k] = a[i + k * 256];
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
      if (NT) { u64* p = (u64*)(b + i + k * 256); __builtin_nontemporal_store(((u64)v[k].y << 32) | v[k].x, p); __builtin_nontemporal_store(((u64)v[k].w << 32) | v[k].z, p + 1); }
      else b[i + k * 256] = v[k];
    }
  }
}
__global__ __launch_bounds__(1024) void k_big(const V* __restrict__ a, V* __restrict__ b, u64 n) {
  const u64 stride = (u64)gridDim.x * 1024 * 4;
  for (u64 i = (u64)blockIdx.x * 4096 + threadIdx.x; i < n; i += stride) {
    V v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = a[i + k * 1024];
#pragma unroll
    for (int k = 0; k < 4; k++) b[i + k * 1024] = v[k];
  }
}
// one-shot: a block of TH threads copies U * TH consecutive 16-byte elements; all U loads first, then U stores
// (what the transform passes do: 16 loads, work, 16 stores).  LDS bytes per block = 0 or `lds` (limits blocks per CU).
template <int U, int TH>
__global__ __launch_bounds__(TH) void k_shot(const V* __restrict__ a, V* __restrict__ b, u64 n) {
  extern __shared__ char smem[];
  const u64 base = (u64)blockIdx.x * (U * TH) + threadIdx.x;
  V v[U];
#pragma unroll
  for (int k = 0; k < U; k++) v[k] = a[base + k * TH];
#pragma unroll
  for (int k = 0; k < U; k++) b[base + k * TH] = v[k];
}
// the same with 8-byte elements (the passes' access width)
template <int U, int TH>
__global__ __launch_bounds__(TH) void k_shot8(const u64* __restrict__ a, u64* __restrict__ b, u64 n) {
  extern __shared__ char smem[];
  const u64 base = (u64)blockIdx.x * (U * TH) + threadIdx.x;
  u64 v[U];
#pragma unroll
  for (int k = 0; k < U; k++) v[k] = a[base + k * TH];
#pragma unroll
  for (int k = 0; k < U; k++) b[base + k * TH] = v[k];
}
// U elements per thread in chunks of C: (C loads, C stores) x U/C — the same footprint per thread, smaller bursts
template <int U, int C, int TH>
__global__ __launch_bounds__(TH) void k_roll8(const u64* __restrict__ a, u64* __restrict__ b, u64 n) {
  const u64 base = (u64)blockIdx.x * (U * TH) + threadIdx.x;
#pragma unroll
  for (int c0 = 0; c0 < U; c0 += C) {
    u64 v[C];
#pragma unroll
    for (int k = 0; k < C; k++) v[k] = a[base + (c0 + k) * TH];
#pragma unroll
    for (int k = 0; k < C; k++) b[base + (c0 + k) * TH] = v[k];
  }
}
// 16 loads, a delay of `spin` dependent VALU instructions per element (standing in for the butterflies), 16 stores
template <int TH>
__global__ __launch_bounds__(TH) void k_work8(const u64* __restrict__ a, u64* __restrict__ b, u64 n, int spin) {
  const u64 base = (u64)blockIdx.x * (16 * TH) + threadIdx.x;
  u64 v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = a[base + k * TH];
  for (int i = 0; i < spin; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = v[k] * 0x9E3779B97F4A7C15ull + (u64)i;
  }
#pragma unroll
  for (int k = 0; k < 16; k++) b[base + k * TH] = v[k];
}
template <typename F> static float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 5; r++) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms; }
  return best;
}
int main() {
  const u64 bytes = 8ull << 30, n = bytes / 16;
  V *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  auto rep = [&](const char* nm, float ms) { printf("%-44s %8.3f ms  %6.2f TB/s (read+write)\n", nm, ms, 2.0 * bytes / ms * 1e-9); };
  rep("one 16 B element per thread", timeit([&] { k_one<<<(unsigned)(n / 256), 256>>>(a, b, n); }));
  for (int g : {1024, 2048, 4096, 8192}) {
    char nm[96];
    snprintf(nm, 96, "grid-stride x1, %d blocks", g); rep(nm, timeit([&] { k_stride<1, false><<<g, 256>>>(a, b, n); }));
    snprintf(nm, 96, "grid-stride x4, %d blocks", g); rep(nm, timeit([&] { k_stride<4, false><<<g, 256>>>(a, b, n); }));
    snprintf(nm, 96, "grid-stride x8, %d blocks", g); rep(nm, timeit([&] { k_stride<8, false><<<g, 256>>>(a, b, n); }));
    snprintf(nm, 96, "grid-stride x8 non-temporal, %d blocks", g); rep(nm, timeit([&] { k_stride<8, true><<<g, 256>>>(a, b, n); }));
  }
#define SHOT(U, TH, LDS) { char nm[96]; snprintf(nm, 96, "one-shot 16 B x %d per thread, %d threads, %d KiB LDS", U, TH, LDS / 1024); \
    hipFuncSetAttribute((const void*)k_shot<U, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 * 2); \
    rep(nm, timeit([&] { k_shot<U, TH><<<(unsigned)(n / (U * TH)), TH, LDS>>>(a, b, n); })); }
#define SHOT8(U, TH, LDS) { char nm[96]; snprintf(nm, 96, "one-shot  8 B x %d per thread, %d threads, %d KiB LDS", U, TH, LDS / 1024); \
    hipFuncSetAttribute((const void*)k_shot8<U, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 * 2); \
    rep(nm, timeit([&] { k_shot8<U, TH><<<(unsigned)(2 * n / (U * TH)), TH, LDS>>>((const u64*)a, (u64*)b, 2 * n); })); }
  SHOT(1, 256, 0) SHOT(2, 256, 0) SHOT(4, 256, 0) SHOT(8, 256, 0) SHOT(16, 256, 0)
  SHOT(1, 1024, 0) SHOT(4, 1024, 0) SHOT(1, 64, 0)
  SHOT8(1, 256, 0) SHOT8(2, 256, 0) SHOT8(4, 256, 0) SHOT8(16, 256, 0) SHOT8(16, 512, 0)
  SHOT8(16, 256, 40960) SHOT8(16, 512, 69632) SHOT8(16, 256, 20480) SHOT8(8, 256, 20480) SHOT8(4, 256, 10240)
  rep("8 B x 16 per thread in chunks of 4 (256 threads)", timeit([&] { k_roll8<16, 4, 256><<<(unsigned)(2 * n / (16 * 256)), 256>>>((const u64*)a, (u64*)b, 2 * n); }));
  rep("8 B x 16 per thread in chunks of 2 (256 threads)", timeit([&] { k_roll8<16, 2, 256><<<(unsigned)(2 * n / (16 * 256)), 256>>>((const u64*)a, (u64*)b, 2 * n); }));
  rep("8 B x 16 per thread in chunks of 1 (256 threads)", timeit([&] { k_roll8<16, 1, 256><<<(unsigned)(2 * n / (16 * 256)), 256>>>((const u64*)a, (u64*)b, 2 * n); }));
  for (int spin : {0, 4, 16, 32}) { char nm[96]; snprintf(nm, 96, "8 B x 16, %d x 16 multiply-adds between (256 thr)", spin);
    rep(nm, timeit([&] { k_work8<256><<<(unsigned)(2 * n / (16 * 256)), 256>>>((const u64*)a, (u64*)b, 2 * n, spin); })); }
  rep("1024-thread blocks x4, 512 blocks", timeit([&] { k_big<<<512, 1024>>>(a, b, n); }));
  rep("hipMemcpyAsync D2D", timeit([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }));
  return 0;
}
