// Do the 8 XCDs start a grid at the same time?  Every wave stamps the 100 MHz wall clock and its XCC id at entry; the
// kernel's duration comes from events attached to the dispatch.  An (almost) empty kernel and one that spins for ~5 us.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include <chrono>
__global__ __launch_bounds__(256) void copy_k(const int4* src, int4* dst, int n16) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void copy_nt_k(const int4* src, int4* dst, int n16) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) { const int4 v = src[i]; int* d = (int*)(dst + i); __builtin_nontemporal_store(v.x, d); __builtin_nontemporal_store(v.y, d + 1); __builtin_nontemporal_store(v.z, d + 2); __builtin_nontemporal_store(v.w, d + 3); }
}
__global__ void tiny_k(int* p) { if (p[0] == 123456789) p[1] = 1; }
struct Big { unsigned long long* stamps; int* xcc; int spin_ticks; long pad[72]; };
template <int LDS>
__global__ __launch_bounds__(256) void kbig(Big b) {
  __shared__ int4 lds[LDS ? 1344 : 1];
  unsigned long long* stamps = b.stamps; int* xcc = b.xcc; const int spin_ticks = b.spin_ticks;
  const unsigned long long t0 = wall_clock64();
  if (LDS && spin_ticks == 12345) lds[threadIdx.x] = make_int4(1, 2, 3, (int)b.pad[threadIdx.x % 72]);
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = t0;
    xcc[w] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11)) & 0xf;
  }
  while ((long long)(wall_clock64() - t0) < spin_ticks) {}
  if ((threadIdx.x & 63) == 0) stamps[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = wall_clock64() + (LDS && spin_ticks == 12345 ? lds[0].x : 0);
}
__global__ __launch_bounds__(256) void k(unsigned long long* stamps, int* xcc, int spin_ticks) {
  const unsigned long long t0 = wall_clock64();
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = t0;
    xcc[w] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11)) & 0xf;  // HW_REG_XCC_ID, bits 3:0
  }
  while ((long long)(wall_clock64() - t0) < spin_ticks) {}
  if ((threadIdx.x & 63) == 0) stamps[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = wall_clock64();
}
int main() {
  const int blocks = 962, waves = blocks * 4;
  unsigned long long* st; int* xc;
  hipHostMalloc(&st, waves * 16, hipHostMallocMapped); hipHostMalloc(&xc, waves * 4, hipHostMallocMapped);
  unsigned long long* dst; int* dxc;
  hipHostGetDevicePointer((void**)&dst, st, 0); hipHostGetDevicePointer((void**)&dxc, xc, 0);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e0, e1, ec; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreateWithFlags(&ec, hipEventDisableTiming);
  hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  void* dbuf; hipMalloc(&dbuf, 200000); void* hbuf; hipHostMalloc(&hbuf, 200000, 0);
  void* hdev; hipHostGetDevicePointer(&hdev, hbuf, 0);
  for (int mode = 0; mode < 9; mode++) {
    const int spin = 500;
    printf("mode %d: %s\n", mode, mode == 0 ? "small arguments" : mode == 1 ? "600-byte arguments + 21 KB LDS" : mode == 2 ? "+ a 190 KB H2D copy in front of every launch" : mode == 3 ? "small arguments, H2D copy in front" : mode == 4 ? "600-byte arguments, copy KERNEL (reads pinned host memory) in front" : mode == 5 ? "600-byte arguments, hipMemcpyAsync on a second stream + event wait" : mode == 6 ? "copy kernel with non-temporal stores in front" : mode == 7 ? "a one-block kernel that writes nothing in front" : "copy kernel with 8 blocks in front");
    std::vector<float> dur, wall;
    for (int it = 0; it < 50; it++) {
      const auto w0 = std::chrono::steady_clock::now();
      Big big{dst, dxc, spin, {}};
      if (mode == 2 || mode == 3) hipMemcpyAsync(dbuf, hbuf, 190000, hipMemcpyHostToDevice, s);
      if (mode == 4) hipLaunchKernelGGL(copy_k, dim3(48), dim3(256), 0, s, (const int4*)hdev, (int4*)dbuf, 190000 / 16);
      if (mode == 6) hipLaunchKernelGGL(copy_nt_k, dim3(48), dim3(256), 0, s, (const int4*)hdev, (int4*)dbuf, 190000 / 16);
      if (mode == 7) hipLaunchKernelGGL(tiny_k, dim3(1), dim3(64), 0, s, (int*)dbuf);
      if (mode == 8) hipLaunchKernelGGL(copy_k, dim3(8), dim3(256), 0, s, (const int4*)hdev, (int4*)dbuf, 190000 / 16);
      if (mode == 5) { hipMemcpyAsync(dbuf, hbuf, 190000, hipMemcpyHostToDevice, s2); hipEventRecord(ec, s2); hipStreamWaitEvent(s, ec, 0); }
      if (mode == 0 || mode == 3) hipExtLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, s, e0, e1, 0, dst, dxc, spin);
      else hipExtLaunchKernelGGL(kbig<1>, dim3(blocks), dim3(256), 0, s, e0, e1, 0, big);
      hipStreamSynchronize(s);
      wall.push_back(std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - w0).count());
      float ms; hipEventElapsedTime(&ms, e0, e1); dur.push_back(ms * 1000.f);
    }
    std::sort(dur.begin(), dur.end());
    unsigned long long t0 = ~0ull, tend = 0;
    for (int w = 0; w < waves; w++) { t0 = std::min(t0, st[2 * w]); tend = std::max(tend, st[2 * w + 1]); }
    std::sort(wall.begin(), wall.end());
    printf("host wall enqueue..sync median %.1f us; ", wall[wall.size() / 2]);
    printf("spin %d ticks: kernel %.2f us by events; first entry -> last exit %.2f us by the wall clock\n", spin, dur[dur.size() / 2], (tend - t0) / 100.0);
    for (int x = 0; x < 8; x++) {
      unsigned long long lo = ~0ull, hi = 0; int n = 0;
      for (int w = 0; w < waves; w++) if (xc[w] == x) { lo = std::min(lo, st[2 * w]); hi = std::max(hi, st[2 * w]); n++; }
      if (n) printf("  XCC %d: %4d waves, entries %.2f .. %.2f us after the first\n", x, n, (lo - t0) / 100.0, (hi - t0) / 100.0);
    }
  }
  return 0;
}
