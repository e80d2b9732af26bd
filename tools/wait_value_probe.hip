// tools/wait_value_probe.hip -- can the scoring launch be ENQUEUED before the host has finished the per-call tables,
// held back by a stream memory operation instead of an in-kernel gate (tools/bar_gate_probe.hip: stale reads / slow)?
//   A: host prepares for P us (spin), writes the table through the BAR, launches; waits for the kernel's sentinel.
//   B: hipStreamWaitValue64(stream, flag, seq, EQ) + launch go out FIRST; the host then prepares for P us, writes the
//      table, fences and writes the flag (pinned host memory); waits for the sentinel.
// Printed: host time of one iteration and "table written -> sentinel seen". The kernel checks every table word, so a
// launch that started before the table was complete shows up as wrong words.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/wait_value_probe tools/wait_value_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <immintrin.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ __launch_bounds__(256) void check_kernel(const unsigned* table, int n, unsigned want, unsigned* wrong, unsigned* ticket,
                                                    volatile unsigned long long* done, unsigned long long seq, int spin_clocks) {
  unsigned bad = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) bad += table[i] != want + (unsigned)i;
  if (bad) atomicAdd(wrong, bad);
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin_clocks) {}  // stands for the scoring work (100 MHz clock)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) { *ticket = 0; __threadfence_system(); *done = seq; }
  }
}

int main(int argc, char** argv) {
  const double prep_us = argc > 1 ? atof(argv[1]) : 10.0;
  const int n = 24 * 1024, blocks = 933, iters = 400;  // 96 KB table
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned* table = nullptr;
  CK(hipExtMallocWithFlags((void**)&table, n * sizeof(unsigned), hipDeviceMallocFinegrained));
  unsigned *wrong, *ticket;
  CK(hipMalloc(&wrong, 4)); CK(hipMalloc(&ticket, 4));
  CK(hipMemset(wrong, 0, 4)); CK(hipMemset(ticket, 0, 4));
  unsigned long long* done; unsigned long long* flag;
  CK(hipHostMalloc((void**)&done, 64, hipHostMallocMapped)); CK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped));
  *done = 0; *flag = 0;
  unsigned long long *d_done, *d_flag;
  CK(hipHostGetDevicePointer((void**)&d_done, done, 0)); CK(hipHostGetDevicePointer((void**)&d_flag, flag, 0));
  std::vector<unsigned> img(n);
  auto write_table = [&](unsigned want) {
    for (int i = 0; i < n; i++) img[i] = want + (unsigned)i;
    memcpy(table, img.data(), n * sizeof(unsigned));  // plain stores through the BAR
    _mm_sfence();
  };
  auto spin = [&](double us) { const double t = now_us(); while (now_us() - t < us) {} };
  for (int mode = 0; mode < 4; mode++) {
    const bool early = mode & 1;
    std::vector<double> total, tail;
    unsigned long long seq = (unsigned long long)mode << 32;
    for (int it = 0; it < iters; it++) {
      seq++;
      const unsigned want = (unsigned)(seq * 2654435761u);
      const double t0 = now_us();
      if (early) {
        CK(hipStreamWaitValue64(st, d_flag, seq, hipStreamWaitValueEq, ~0ull));
        hipLaunchKernelGGL(check_kernel, dim3(blocks), dim3(256), 0, st, table, n, want, wrong, ticket, d_done, seq, 500);
      }
      spin(prep_us);
      write_table(want);
      const double t1 = now_us();
      if (early) { *(volatile unsigned long long*)flag = seq; _mm_sfence(); }
      else hipLaunchKernelGGL(check_kernel, dim3(blocks), dim3(256), 0, st, table, n, want, wrong, ticket, d_done, seq, 500);
      while (*(volatile unsigned long long*)done != seq) {}
      const double t2 = now_us();
      if (it >= 20) { total.push_back(t2 - t0); tail.push_back(t2 - t1); }
    }
    CK(hipStreamSynchronize(st));
    unsigned w = 0;
    CK(hipMemcpy(&w, wrong, 4, hipMemcpyDeviceToHost));
    std::sort(total.begin(), total.end()); std::sort(tail.begin(), tail.end());
    printf("%s: iteration median %.1f us (p90 %.1f); table written -> result seen median %.1f us (p90 %.1f); wrong words %u\n",
           early ? "wait-value + launch first" : "launch after the table    ", total[total.size() / 2], total[total.size() * 9 / 10],
           tail[tail.size() / 2], tail[tail.size() * 9 / 10], w);
    CK(hipMemset(wrong, 0, 4));
  }
  return 0;
}
