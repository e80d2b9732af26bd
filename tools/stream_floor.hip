// What does a launch of this size cost on this GPU, whatever it computes?  HIP-event durations of
//  (a) an empty kernel, 960 blocks x 256 threads,
//  (b) a pure stream of the scoring kernel's volume at cfg3: read 2 x 6.0 MB of 8-byte records + 0.75 MB of
//      1-byte codes, write 6.0 MB of f64 -- 8 bytes per lane and load, like the compact class,
//  (c) the same bytes with 16-byte loads / stores,
// each repeated back to back on data that stays cache / Infinity-Cache resident (as in bench.py).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__global__ void empty_kernel() {}
__global__ __launch_bounds__(256) void stream8(const unsigned long long* a, const unsigned long long* b, const unsigned char* c, double* out, int n) {
  double acc = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const unsigned long long x = a[i], y = b[i];
    const unsigned char z = c[i];
    const double v = (double)(int)(x + y + z);
    out[i] = v; acc += v;
  }
  if (acc == 1.2345e300) out[0] = acc;
}
__global__ __launch_bounds__(256) void stream16(const ulonglong2* a, const ulonglong2* b, const unsigned short* c, double2* out, int n2) {
  double acc = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n2; i += gridDim.x * 256) {
    const ulonglong2 x = a[i], y = b[i];
    const unsigned short z = c[i];
    const double2 v = make_double2((double)(int)(x.x + y.x + (z & 255)), (double)(int)(x.y + y.y + (z >> 8)));
    out[i] = v; acc += v.x + v.y;
  }
  if (acc == 1.2345e300) out[0].x = acc;
}
int main() {
  const int n = 753295;
  void *a, *b, *c, *o;
  hipMalloc(&a, (size_t)n * 8 + 64); hipMalloc(&b, (size_t)n * 8 + 64); hipMalloc(&c, n + 64); hipMalloc(&o, (size_t)n * 8 + 64);
  hipMemset(a, 1, (size_t)n * 8); hipMemset(b, 2, (size_t)n * 8); hipMemset(c, 3, n);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](int which, int blocks) {
    std::vector<float> t;
    for (int it = 0; it < 200; it++) {
      hipEventRecord(e0, st);
      if (which == 0) hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(256), 0, st);
      if (which == 1) hipLaunchKernelGGL(stream8, dim3(blocks), dim3(256), 0, st, (const unsigned long long*)a, (const unsigned long long*)b, (const unsigned char*)c, (double*)o, n);
      if (which == 2) hipLaunchKernelGGL(stream16, dim3(blocks), dim3(256), 0, st, (const ulonglong2*)a, (const ulonglong2*)b, (const unsigned short*)c, (double2*)o, n / 2);
      hipEventRecord(e1, st);
      hipStreamSynchronize(st);
      float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1000.f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
  };
  printf("empty kernel, 960 blocks                       : %.2f us\n", run(0, 960));
  for (int blocks : {384, 768, 1536, 2944}) printf("stream 8 B/lane  (18.8 MB), %4d blocks          : %.2f us\n", blocks, run(1, blocks));
  for (int blocks : {384, 768, 1472}) printf("stream 16 B/lane (18.8 MB), %4d blocks          : %.2f us\n", blocks, run(2, blocks));
  return 0;
}
