// tools/ubench_mem.hip — HBM streaming rate vs access width/pattern on MI355X
// (diagnostic; numbers quoted in DESIGN.md).  All kernels copy `bytes` from a to b.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;

// (a) 8 B per lane, contiguous, 16 loads in flight per thread, then 16 stores
__global__ __launch_bounds__(256) void copy_b64(const u64* __restrict__ a, u64* __restrict__ b, u64 n) {
  u64 base = (u64)blockIdx.x * 4096 + threadIdx.x;
  u64 v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = a[base + k * 256];
#pragma unroll
  for (int k = 0; k < 16; k++) b[base + k * 256] = v[k];
}
// (b) 16 B per lane
__global__ __launch_bounds__(256) void copy_b128(const ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, u64 n) {
  u64 base = (u64)blockIdx.x * 2048 + threadIdx.x;
  ulonglong2 v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = a[base + k * 256];
#pragma unroll
  for (int k = 0; k < 8; k++) b[base + k * 256] = v[k];
}
// (c) the strided-pass pattern: tile = 256 rows x 32 columns of a 256x256 u64 matrix,
// 512 threads, each 16 x 8-byte loads at row stride 2 KiB
__global__ __launch_bounds__(512) void copy_strided_b64(const u64* __restrict__ a, u64* __restrict__ b, u64 n) {
  u32 c = threadIdx.x % 32, tf = threadIdx.x / 32;
  u64 poly = blockIdx.x >> 3, cg = blockIdx.x & 7;
  u64 base = (poly << 16) + cg * 32 + c;
  u64 v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = a[base + ((u64)(k * 16 + tf) << 8)];
#pragma unroll
  for (int k = 0; k < 16; k++) b[base + ((u64)(tf * 16 + k) << 8)] = v[k];
}
// (d) same tile, 16 B per lane: 256 rows x 32 cols; lane handles 2 adjacent columns
__global__ __launch_bounds__(256) void copy_strided_b128(const ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, u64 n) {
  u32 c = threadIdx.x % 16, tf = threadIdx.x / 16;   // 16 lanes x 16 B = 256 B row segment
  u64 poly = blockIdx.x >> 3, cg = blockIdx.x & 7;
  u64 base = (poly << 15) + cg * 16 + c;             // in 16-byte units: row = 128 units
  ulonglong2 v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = a[base + ((u64)(k * 16 + tf) << 7)];
#pragma unroll
  for (int k = 0; k < 16; k++) b[base + ((u64)(tf * 16 + k) << 7)] = v[k];
}
// (e) LDS-DMA in (global_load_lds_dwordx4), ds_read_b128 + global_store_dwordx4 out; 32 KiB tile
__global__ __launch_bounds__(256) void copy_ldsdma(const char* __restrict__ a, char* __restrict__ b, u64 n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const u32 tid = threadIdx.x, wave = tid >> 6;
  const char* src = a + (u64)blockIdx.x * 32768;
#pragma unroll
  for (int i = 0; i < 8; i++) {   // each wave-instruction moves 1 KiB: wave w handles chunks w*8 + i
    const u32 chunk = wave * 8 + i;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + chunk * 1024 + (tid & 63) * 16),
                                     (void __attribute__((address_space(3)))*)(smem + chunk * 1024), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  char* dst = b + (u64)blockIdx.x * 32768;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u32 off = (i * 256 + tid) * 16;
    *(ulonglong2*)(dst + off) = *(const ulonglong2*)(smem + off);
  }
}

// (f) 16 B per lane with non-temporal loads and stores (streaming hint: no reuse expected)
__global__ __launch_bounds__(256) void copy_b128_nt(const ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, u64 n) {
  u64 base = (u64)blockIdx.x * 2048 + threadIdx.x;
  u64 x[8], y[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const u64* p = (const u64*)(a + base + k * 256);
    x[k] = __builtin_nontemporal_load(p); y[k] = __builtin_nontemporal_load(p + 1);
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    u64* p = (u64*)(b + base + k * 256);
    __builtin_nontemporal_store(x[k], p); __builtin_nontemporal_store(y[k], p + 1);
  }
}
// (g) non-temporal stores only
__global__ __launch_bounds__(256) void copy_b64_nts(const u64* __restrict__ a, u64* __restrict__ b, u64 n) {
  u64 base = (u64)blockIdx.x * 4096 + threadIdx.x;
  u64 v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = a[base + k * 256];
#pragma unroll
  for (int k = 0; k < 16; k++) __builtin_nontemporal_store(v[k], b + base + k * 256);
}
// (h) read only (xor-reduce, one store per workgroup), (i) write only
__global__ __launch_bounds__(256) void read_only(const ulonglong2* __restrict__ a, u64* __restrict__ b, u64 n) {
  u64 base = (u64)blockIdx.x * 2048 + threadIdx.x;
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { ulonglong2 v = a[base + k * 256]; acc ^= v.x ^ v.y; }
  if (acc == 0x123456789abcdefull) b[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void write_only(ulonglong2* __restrict__ b, u64 n) {
  u64 base = (u64)blockIdx.x * 2048 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; k++) b[base + k * 256] = ulonglong2{base, (u64)k};
}

template <typename F> static float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; r++) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best;
}
int main() {
  const u64 bytes = 8ull << 30, n = bytes / 8;
  char *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  auto rep = [&](const char* nm, float ms) { printf("%-28s %8.3f ms  %6.2f TB/s (read+write)\n", nm, ms, 2.0 * bytes / ms * 1e-9); };
  rep("contiguous 8B/lane", timeit([&] { copy_b64<<<n / 4096, 256>>>((const u64*)a, (u64*)b, n); }));
  rep("contiguous 16B/lane", timeit([&] { copy_b128<<<n / 4096, 256>>>((const ulonglong2*)a, (ulonglong2*)b, n); }));
  rep("strided tile 8B/lane", timeit([&] { copy_strided_b64<<<n / 8192, 512>>>((const u64*)a, (u64*)b, n); }));
  rep("strided tile 16B/lane", timeit([&] { copy_strided_b128<<<n / 8192, 256>>>((const ulonglong2*)a, (ulonglong2*)b, n); }));
  rep("LDS-DMA in, b128 out", timeit([&] { copy_ldsdma<<<bytes / 32768, 256, 32768>>>(a, b, n); }));
  rep("contiguous 16B/lane, nt ld+st", timeit([&] { copy_b128_nt<<<n / 4096, 256>>>((const ulonglong2*)a, (ulonglong2*)b, n); }));
  rep("contiguous 8B/lane, nt st", timeit([&] { copy_b64_nts<<<n / 4096, 256>>>((const u64*)a, (u64*)b, n); }));
  { float ms = timeit([&] { read_only<<<n / 4096, 256>>>((const ulonglong2*)a, (u64*)b, n); });
    printf("%-28s %8.3f ms  %6.2f TB/s (read only)\n", "read only 16B/lane", ms, 1.0 * bytes / ms * 1e-9); }
  { float ms = timeit([&] { write_only<<<n / 4096, 256>>>((ulonglong2*)b, n); });
    printf("%-28s %8.3f ms  %6.2f TB/s (write only)\n", "write only 16B/lane", ms, 1.0 * bytes / ms * 1e-9); }
  { float ms = timeit([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    rep("hipMemcpyAsync D2D", ms); }
  return 0;
}
