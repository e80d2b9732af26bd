// tools/bar_write_probe.hip -- can the host put the per-call occurrence tables (95 KB) straight into device memory
// (CPU stores through the PCIe BAR into a fine-grained / uncached device allocation) instead of a pinned staging slot
// + copy kernel? Measures (1) host time of the 95 KB write, (2) whether a kernel launched right after sees the new
// bytes every time (same addresses rewritten per call), (3) what random 8-byte lookups into such memory cost a
// kernel compared with ordinary device memory.
//   hipcc --offload-arch=gfx950 -O2 -o build_ab/bar_write_probe tools/bar_write_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <immintrin.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void check_kernel(const unsigned* t, int n, unsigned want, unsigned* bad) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (t[i] != want + (unsigned)i) atomicAdd(bad, 1u);
}
__global__ void lookup_kernel(const unsigned long long* t, int n, const unsigned* idx, int m, unsigned long long* out) {
  unsigned long long acc = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) acc += t[idx[i] % n];
  if (acc == 0x1234567) out[0] = acc;
}

int main() {
  const size_t bytes = 96 * 1024;
  const int n = bytes / 4;
  struct Kind { const char* name; unsigned flag; } kinds[] = {{"finegrained", hipDeviceMallocFinegrained}, {"uncached", hipDeviceMallocUncached}};
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned* d_bad; CK(hipMalloc(&d_bad, 4));
  std::vector<unsigned> src(n);
  // lookups: 1.5 M random indices like the scoring kernel's occurrence reads (sorted-ish: runs of equal values)
  const int m = 1500000; std::vector<unsigned> idx(m); for (int i = 0; i < m; i++) idx[i] = (unsigned)(i / 190);
  unsigned* d_idx; CK(hipMalloc(&d_idx, m * 4)); CK(hipMemcpy(d_idx, idx.data(), m * 4, hipMemcpyHostToDevice));
  unsigned long long* d_out; CK(hipMalloc(&d_out, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  {  // baseline: ordinary device memory
    void* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes));
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(lookup_kernel, dim3(768), dim3(256), 0, st, (const unsigned long long*)p, (int)(bytes / 8), d_idx, m, d_out);
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep == 2) printf("lookups in hipMalloc memory: %.1f us\n", ms * 1e3);
    }
  }
  for (auto& k : kinds) {
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, k.flag);
    if (e != hipSuccess) { printf("%s: hipExtMallocWithFlags failed: %s\n", k.name, hipGetErrorString(e)); continue; }
    hipPointerAttribute_t at; memset(&at, 0, sizeof(at));
    (void)hipPointerGetAttributes(&at, p);
    printf("%s: device ptr %p host ptr %p type %d\n", k.name, at.devicePointer, at.hostPointer, (int)at.type);
    // does a CPU store fault? try under a guard: write one word
    volatile unsigned* hp = (volatile unsigned*)p;
    CK(hipMemset(p, 0, bytes)); CK(hipDeviceSynchronize());
    unsigned total_bad = 0; double t_write = 0; int reps = 200;
    for (int rep = 0; rep < reps; rep++) {
      const unsigned want = 1000u * (rep + 1);
      for (int i = 0; i < n; i++) src[i] = want + i;
      double t0 = now_us();
      memcpy((void*)hp, src.data(), bytes);
      _mm_sfence();
      t_write += now_us() - t0;
      CK(hipMemsetAsync(d_bad, 0, 4, st));
      hipLaunchKernelGGL(check_kernel, dim3(64), dim3(256), 0, st, (const unsigned*)p, n, want, d_bad);
      unsigned bad = 0; CK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
      total_bad += bad;
    }
    printf("%s: host write of %zu KB: %.2f us avg; stale words seen by the kernel over %d rewrites: %u\n", k.name, bytes / 1024, t_write / reps, reps, total_bad);
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(lookup_kernel, dim3(768), dim3(256), 0, st, (const unsigned long long*)p, (int)(bytes / 8), d_idx, m, d_out);
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep == 2) printf("%s: lookups: %.1f us\n", k.name, ms * 1e3);
    }
  }
  return 0;
}
