// Launch/sync latency probe for the blocking CalcProb path (numbers quoted in DESIGN.md).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] = 1; }
__global__ void work_kernel(const int* src, volatile int* flag, int n, int val) {
  long s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s += src[i];
  if (s == 123456789) flag[1] = 1;
  if (blockIdx.x == 0 && threadIdx.x == 0) flag[0] = val;  // not a completion flag, just a host-visible store
}
int main() {
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const size_t bytes = 190 * 1024;
  void *h, *d; hipHostMalloc(&h, bytes, hipHostMallocMapped); hipMalloc(&d, bytes); memset(h, 1, bytes);
  int* hflag; hipHostMalloc((void**)&hflag, 64, hipHostMallocMapped); int* dflag; hipHostGetDevicePointer((void**)&dflag, hflag, 0);
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  std::vector<double> a, b, c, e;
  for (int it = 0; it < 300; it++) {
    double t0 = now_us();
    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, (int*)nullptr); hipStreamSynchronize(st);
    a.push_back(now_us() - t0);
    t0 = now_us();
    hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(work_kernel, dim3(1800), dim3(256), 0, st, (const int*)d, dflag, (int)(bytes / 4), it);
    hipStreamSynchronize(st);
    b.push_back(now_us() - t0);
    // same, but the host spins on the host-visible word instead of hipStreamSynchronize
    hflag[0] = -1;
    t0 = now_us();
    hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(work_kernel, dim3(1800), dim3(256), 0, st, (const int*)d, dflag, (int)(bytes / 4), it);
    while (((volatile int*)hflag)[0] != it) {}
    c.push_back(now_us() - t0);
    hipStreamSynchronize(st);
    // kernel reads the table straight from pinned host memory (no H2D copy)
    t0 = now_us();
    hipLaunchKernelGGL(work_kernel, dim3(1800), dim3(256), 0, st, (const int*)h, dflag, (int)(bytes / 4), it);
    hipStreamSynchronize(st);
    e.push_back(now_us() - t0);
  }
  printf("empty kernel + sync            : %.1f us\n", med(a));
  printf("H2D 190 KB + kernel + sync     : %.1f us\n", med(b));
  printf("H2D 190 KB + kernel + host spin: %.1f us\n", med(c));
  printf("kernel reading pinned host mem : %.1f us\n", med(e));
  return 0;
}
