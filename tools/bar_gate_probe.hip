// tools/bar_gate_probe.hip -- can a scoring launch go out BEFORE the host has finished the per-call tables?
// The kernel is launched first; its waves read the tables once (so that stale copies sit in whatever caches there are),
// then spin on a gate word; the host meanwhile rewrites the tables through the PCIe BAR (fine-grained device memory),
// fences, and writes the gate. After the gate the kernel reads the tables again and counts words that are not the new
// ones. Also measured: host time from "gate written" to "kernel's result visible" against a plain launch after the
// tables were written.
//   hipcc --offload-arch=gfx950 -O2 -o build_ab/bar_gate_probe tools/bar_gate_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <immintrin.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// mode 0: plain loads after the gate; 1: the gate is polled with an acquire load at system scope; 2: one wave polls the host word
template <int MODE>
__global__ __launch_bounds__(256) void gated_kernel(const unsigned* table, int n, const unsigned long long* gate, unsigned long long seq,
                                                    unsigned want_old, unsigned want_new, unsigned* bad_old, unsigned* bad_new,
                                                    volatile unsigned long long* done, unsigned long long* spins, unsigned* ticket, unsigned long long* go) {
  // 1. read the tables as they are now (what an earlier launch would have left in the caches)
  unsigned stale = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) stale += table[i] != want_old + (unsigned)i;
  if (stale) atomicAdd(bad_old, stale);
  // 2. wait for the host
  unsigned long long polls = 0;
  const long long t0 = wall_clock64();
  if (MODE == 2) {
    // ONE wave of the grid polls the host's word; everybody else waits for a word in ordinary device memory that this
    // wave sets (thousands of waves polling one fine-grained address serialise on it)
    if ((threadIdx.x & 63) == 0) {
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        while (__hip_atomic_load(gate, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) { polls++; if (wall_clock64() - t0 > 100000000ll) break; }
        __hip_atomic_store(go, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq) { polls++; if (wall_clock64() - t0 > 100000000ll) break; __builtin_amdgcn_s_sleep(8); }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  } else if ((threadIdx.x & 63) == 0) {
    while (true) {
      const unsigned long long v = MODE == 1 ? __hip_atomic_load(gate, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)
                                             : *(volatile const unsigned long long*)gate;
      polls++;
      if (v == seq) break;
      if (wall_clock64() - t0 > 100000000ll) break;  // 1 s at 100 MHz: give up
      __builtin_amdgcn_s_sleep(2);
    }
  }
  if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __builtin_amdgcn_wave_barrier();
  // 3. the tables again: every word must be the new one
  unsigned wrong = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) wrong += table[i] != want_new + (unsigned)i;
  if (wrong) atomicAdd(bad_new, wrong);
  if ((threadIdx.x & 63) == 0) atomicAdd(spins, polls);
  __syncthreads();
  if (threadIdx.x == 0) {  // the last block to finish tells the host
    __threadfence();
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) { *ticket = 0; __threadfence_system(); *done = gridDim.x; }
  }
}

__global__ __launch_bounds__(256) void plain_kernel(const unsigned* table, int n, unsigned want_new, unsigned* bad_new, volatile unsigned long long* done, unsigned* ticket) {
  unsigned wrong = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) wrong += table[i] != want_new + (unsigned)i;
  if (wrong) atomicAdd(bad_new, wrong);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) { *ticket = 0; __threadfence_system(); *done = gridDim.x; }
  }
}

int main() {
  const size_t bytes = 192 * 1024;
  const int n = bytes / 4, grid = 1024;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  void* p = nullptr; CK(hipExtMallocWithFlags(&p, bytes + 256, hipDeviceMallocFinegrained));
  unsigned* table = (unsigned*)p;
  unsigned long long* gate = (unsigned long long*)((char*)p + bytes);
  unsigned *d_bad_old, *d_bad_new; CK(hipMalloc(&d_bad_old, 4)); CK(hipMalloc(&d_bad_new, 4));
  unsigned long long* d_spins; CK(hipMalloc(&d_spins, 8));
  unsigned* d_ticket; CK(hipMalloc(&d_ticket, 4)); CK(hipMemset(d_ticket, 0, 4));
  unsigned long long* d_go; CK(hipMalloc(&d_go, 8)); CK(hipMemset(d_go, 0, 8));
  unsigned long long* h_done; CK(hipHostMalloc(&h_done, 64, hipHostMallocMapped | hipHostMallocCoherent));
  unsigned long long* d_done; CK(hipHostGetDevicePointer((void**)&d_done, h_done, 0));
  std::vector<unsigned> src(n);
  auto write_tables = [&](unsigned want, int words) { for (int i = 0; i < words; i++) src[i] = want + i; memcpy(table, src.data(), (size_t)words * 4); _mm_sfence(); };
  for (int mode = 0; mode < 3; mode++) {
    for (int words : {n, 24}) {  // the whole tables / a handful of entries (an annealing move)
      CK(hipMemset(d_bad_old, 0, 4)); CK(hipMemset(d_bad_new, 0, 4)); CK(hipMemset(d_spins, 0, 8));
      write_tables(7, n); *gate = 0; _mm_sfence(); CK(hipDeviceSynchronize());
      const int reps = 300;
      double t_gate = 0, t_total = 0;
      for (int rep = 0; rep < reps; rep++) {
        const unsigned want_old = 7u + 1000u * rep, want_new = 7u + 1000u * (rep + 1);
        const unsigned long long seq = rep + 1;
        *h_done = 0;
        const double t0 = now_us();
        if (mode == 0) hipLaunchKernelGGL(gated_kernel<0>, dim3(grid), dim3(256), 0, st, table, n, gate, seq, want_old, want_new, d_bad_old, d_bad_new, d_done, d_spins, d_ticket, d_go);
        else if (mode == 1) hipLaunchKernelGGL(gated_kernel<1>, dim3(grid), dim3(256), 0, st, table, n, gate, seq, want_old, want_new, d_bad_old, d_bad_new, d_done, d_spins, d_ticket, d_go);
        else hipLaunchKernelGGL(gated_kernel<2>, dim3(grid), dim3(256), 0, st, table, n, gate, seq, want_old, want_new, d_bad_old, d_bad_new, d_done, d_spins, d_ticket, d_go);
        // the host's turn: "planning" for ~6 us, then the tables, then the gate
        const double tw = now_us();
        while (now_us() - tw < 6.0) _mm_pause();
        if (words == n) write_tables(want_new, n);
        else { for (int i = 0; i < n; i++) src[i] = want_new + i; memcpy(table, src.data(), bytes); _mm_sfence(); }  // (every word changes value: the check needs it)
        *(volatile unsigned long long*)gate = seq; _mm_sfence();
        const double t1 = now_us();
        while (*(volatile unsigned long long*)h_done < (unsigned long long)grid) _mm_pause();
        const double t2 = now_us();
        t_gate += t2 - t1; t_total += t2 - t0;
      }
      CK(hipStreamSynchronize(st));
      unsigned bo = 0, bn = 0; unsigned long long sp = 0;
      CK(hipMemcpy(&bo, d_bad_old, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&bn, d_bad_new, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&sp, d_spins, 8, hipMemcpyDeviceToHost));
      printf("mode %d (%s): %d launches; words wrong before the gate %u, STALE words after the gate %u; gate -> all blocks done %.1f us, launch -> done %.1f us; %.1f polls per wave\n",
             mode, mode == 0 ? "volatile poll" : mode == 1 ? "acquire, system scope" : "one poller, agent-scope hand-over", reps, bo, bn, t_gate / reps, t_total / reps, (double)sp / reps / (grid * 4));
    }
  }
  {  // the plain order: tables, then launch, then wait
    CK(hipMemset(d_bad_new, 0, 4));
    const int reps = 300; double t_total = 0, t_launch = 0;
    for (int rep = 0; rep < reps; rep++) {
      const unsigned want_new = 99u + 1000u * (rep + 1);
      *h_done = 0;
      const double t0 = now_us();
      const double tw = now_us();
      while (now_us() - tw < 6.0) _mm_pause();
      write_tables(want_new, n);
      const double t1 = now_us();
      hipLaunchKernelGGL(plain_kernel, dim3(grid), dim3(256), 0, st, table, n, want_new, d_bad_new, d_done, d_ticket);
      while (*(volatile unsigned long long*)h_done < (unsigned long long)grid) _mm_pause();
      const double t2 = now_us();
      t_total += t2 - t0; t_launch += t2 - t1;
    }
    unsigned bn = 0; CK(hipMemcpy(&bn, d_bad_new, 4, hipMemcpyDeviceToHost));
    printf("plain order: stale words %u; tables written -> done %.1f us, start -> done %.1f us\n", bn, t_launch / reps, t_total / reps);
  }
  return 0;
}
