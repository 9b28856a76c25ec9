// tools/ubench_mall.hip — does the 256 MiB Infinity Cache serve a re-read / overwrite of a
// recently written tile?  (diagnostic)  Pattern = the two-pass NTT: kernel A reads X tile,
// writes Y tile; kernel B reads Y tile, writes Y tile in place.  Timed per tile size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
__global__ __launch_bounds__(256) void copy_b128(const ulonglong2* __restrict__ a, ulonglong2* __restrict__ b) {
  u64 base = (u64)blockIdx.x * 2048 + threadIdx.x;
  ulonglong2 v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = a[base + k * 256];
#pragma unroll
  for (int k = 0; k < 8; k++) { v[k].x += 1; b[base + k * 256] = v[k]; }
}
__global__ __launch_bounds__(256) void read_b128(const ulonglong2* __restrict__ a, ulonglong2* __restrict__ sink) {
  u64 base = (u64)blockIdx.x * 2048 + threadIdx.x;
  ulonglong2 v[8]; u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = a[base + k * 256];
#pragma unroll
  for (int k = 0; k < 8; k++) acc += v[k].x ^ v[k].y;
  if (acc == 0x1234567) sink[0].x = acc;
}
int main() {
  const u64 total = 8ull << 30;
  char *x, *y; hipMalloc(&x, total); hipMalloc(&y, total); hipMemset(x, 1, total); hipMemset(y, 2, total);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("two-pass pattern over 8 GiB, per tile: A: X->Y, B: Y->Y (in place)\n");
  for (u64 tile : {16ull << 20, 32ull << 20, 64ull << 20, 128ull << 20, 256ull << 20, 512ull << 20, 2048ull << 20, 8192ull << 20}) {
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
      hipEventRecord(e0);
      for (u64 off = 0; off < total; off += tile) {
        copy_b128<<<tile / 32768, 256>>>((const ulonglong2*)(x + off), (ulonglong2*)(y + off));
        copy_b128<<<tile / 32768, 256>>>((const ulonglong2*)(y + off), (ulonglong2*)(y + off));
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("tile %5llu MiB: %8.3f ms   %6.2f TB/s of kernel-level traffic (4 x 8 GiB)\n", tile >> 20, best, 4.0 * total / best * 1e-9);
  }
  printf("re-read of one resident buffer (read-only kernel, repeated 20x):\n");
  for (u64 sz : {32ull << 20, 64ull << 20, 128ull << 20, 192ull << 20, 256ull << 20, 512ull << 20, 2048ull << 20}) {
    read_b128<<<sz / 32768, 256>>>((const ulonglong2*)x, (ulonglong2*)y); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 20; r++) read_b128<<<sz / 32768, 256>>>((const ulonglong2*)x, (ulonglong2*)y);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("size %5llu MiB: %6.2f TB/s read\n", sz >> 20, 20.0 * sz / ms * 1e-9);
  }
  return 0;
}
