// multi.hip -- everything of libgaml_hip.so that involves more than one device shard:
//
//   * CommState: the RCCL communicator of a sharded context (one process per GPU: gaml_hip_comm_init_rank, or the
//     N shards of one process: ncclCommInitAll). With it gaml_hip_calc_prob runs the whole sharded evaluation in the
//     library -- cold path: all-reduce(max) of the new windows' largest record positions; penalty_constant > 0 only:
//     all-gather of the coverage maps / PacBio interval events; every evaluation: ONE all-reduce(sum) of 4 f64 per
//     read set, enqueued on the library's stream right behind the scoring kernels (the north star's
//     "single RCCL all-reduce of the per-readset log-likelihood over xGMI").
//   * MultiState: N device shards behind ONE context (gaml_hip_create_multi). The reference's caller is one process
//     holding one ProbCalculator (gaml.cc:1010, prob_calculator.h:37-124); this is how that caller reaches GPUs 1..N-1.
//     One worker thread per shard (its device stays current there), reads split by contiguous id range, graph /
//     paths / window registration replicated; the exchange is RCCL (above, one communicator per shard) or, as the
//     measured alternative, a sum of the shards' pinned-host partials in rank order on the calling thread.
//
// This file uses only the C ABI of the shards plus the few accessors of internal.h; RCCL is loaded on first use
// (dlopen), so single-GPU users never map it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "internal.h"

namespace gaml {

// =============================================================================================================
// RCCL, loaded on first use
// =============================================================================================================
namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;  // (optional)
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // the soname first: a process that already holds RCCL (torch) hands back that copy
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.lib) break; }
    if (!r.lib) { r.err = std::string("librccl.so.1 could not be loaded: ") + dlerror(); return; }
    auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p && r.err.empty()) r.err = std::string("librccl: missing symbol ") + n; return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))dlsym(r.lib, "ncclCommAbort");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  });
  return &r;
}

struct DevMem {  // grow-only device buffer (the caller synchronises the stream that may still read the old one)
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes, hipStream_t st) {
    if (bytes <= cap) return hipSuccess;
    if (p) { hipError_t e = hipStreamSynchronize(st); if (e != hipSuccess) return e; (void)hipFree(p); p = nullptr; cap = 0; }
    const size_t want = std::max<size_t>(256, bytes + bytes / 4);
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct CommState {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  int mode = 0;       // hot-path exchange: 0 all-gather of the ranks' partials, summed in RANK ORDER by every rank; 1 all-reduce(sum)
  bool dead = false;  // aborted after a time-out: every later evaluation fails at once
  double timeout_s = 60.0;
  DevMem part, small, own, all, pack, gath;  // partials (f64) + 2 status words | maxima, sizes | this rank's map / intervals | the gathered ones | the gathered intervals, contiguous | every rank's partials
};

// one block writes two doubles behind a rank's partials: its status for the exchange (no host-to-device copy command)
__global__ void comm_status_kernel(double* at, double a, double b) { if (threadIdx.x == 0 && blockIdx.x == 0) { at[0] = a; at[1] = b; } }

static double comm_timeout_s() {
  const char* e = getenv("GAML_HIP_COMM_TIMEOUT_S");
  const double v = e ? atof(e) : 0.0;
  return v > 0.0 ? v : 60.0;
}

#define COMM_HIP(c, expr)                                                                            \
  do {                                                                                               \
    hipError_t e__ = (expr);                                                                         \
    if (e__ != hipSuccess) return ctx_fail(c, GAML_HIP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
  } while (0)
#define COMM_NCCL(c, expr)                                                                           \
  do {                                                                                               \
    ncclResult_t r__ = (expr);                                                                       \
    if (r__ != ncclSuccess) return ctx_fail(c, GAML_HIP_EHIP, std::string(#expr) + ": " + rccl()->GetErrorString(r__)); \
  } while (0)

void comm_destroy(CommState* s) {
  if (!s) return;
  if (s->comm && !s->dead && rccl()->CommDestroy) (void)rccl()->CommDestroy(s->comm);  // (an aborted communicator is gone already)
  s->part.release(); s->small.release(); s->own.release(); s->all.release(); s->pack.release(); s->gath.release();
  delete s;
}

namespace {

// Every wait behind a collective is bounded (GAML_HIP_COMM_TIMEOUT_S): `n_doubles` doubles at d_src come to the host through
// the context's fetch slot (pinned memory behind a sequence word); when they have not arrived in time -- a peer failed
// before the collective, or entered another one: RCCL leaves mismatched collectives undefined -- this rank aborts its
// communicator and reports instead of hanging in its stream.
int comm_fetch_bounded(gaml_hip_ctx* c, CommState* s, const void* d_src, double* host, int32_t n_doubles, const char* what) {
  if (int e = gaml_hip_fetch_async(c, d_src, n_doubles, ctx_stream(c))) return e;
  const int w = ctx_fetch_wait_bounded(c, host, n_doubles, s->timeout_s);
  if (w < 0) return w;
  if (w > 0) {
    s->dead = true;
    if (rccl()->CommAbort) (void)rccl()->CommAbort(s->comm);
    return ctx_fail(c, GAML_HIP_EHIP, std::string("sharded evaluation: ") + what + " did not complete within " + std::to_string((int)s->timeout_s) +
                                      " s (a peer failed or left the protocol); communicator aborted");
  }
  return 0;
}

// phase 1 of a sharded evaluation: registration / alignment, then the largest record position of every newly
// aligned window as a maximum over ALL ranks' reads (the reference's position filter, graph.cc:577)
int comm_begin(gaml_hip_ctx* c, CommState* s, const int32_t* paths, const int64_t* offs, int32_t n_paths, int32_t* tl, int64_t* aligned) {
  int64_t pending = 0;
  if (int e = gaml_hip_eval_begin(c, paths, offs, n_paths, &pending, tl)) return e;
  if (aligned) *aligned = pending;
  if (pending > 0) {
    hipStream_t st = ctx_stream(c);
    std::vector<int32_t> mx((size_t)pending);
    if (gaml_hip_eval_pending_maxpos(c, mx.data(), pending) != pending) return ctx_fail(c, GAML_HIP_ESTATE, "pending maxima changed under the exchange");
    const size_t nd = ((size_t)pending * sizeof(int32_t) + 7) / 8;  // (fetched as whole doubles)
    COMM_HIP(c, s->small.reserve(nd * sizeof(double), st));
    COMM_HIP(c, hipMemcpyAsync(s->small.p, mx.data(), (size_t)pending * sizeof(int32_t), hipMemcpyHostToDevice, st));
    COMM_NCCL(c, rccl()->AllReduce(s->small.p, s->small.p, (size_t)pending, ncclInt32, ncclMax, s->comm, st));
    std::vector<double> back(nd);
    if (int e = comm_fetch_bounded(c, s, s->small.p, back.data(), (int32_t)nd, "the exchange of the new windows' largest positions")) return e;
    memcpy(mx.data(), back.data(), (size_t)pending * sizeof(int32_t));
    if (int e = gaml_hip_eval_apply_maxpos(c, mx.data(), pending)) return e;
  }
  return 0;
}

// phase 2: scoring kernels -> 4 f64 per read set at d_part (device); with a coverage penalty the maps / interval
// events of all ranks are merged first (SURVEY 8e "the one non-separable piece")
int comm_score(gaml_hip_ctx* c, CommState* s, double* d_part) {
  hipStream_t st = ctx_stream(c);
  const int32_t n_maps = gaml_hip_eval_score_async(c, d_part, st);
  if (n_maps < 0) return n_maps;
  for (int32_t i = 0; i < n_maps; i++) {
    int64_t bytes = 0;
    if (int e = gaml_hip_eval_coverage_export_async(c, i, nullptr, 0, &bytes, st)) return e;
    COMM_HIP(c, s->own.reserve((size_t)std::max<int64_t>(16, bytes), st));
    COMM_HIP(c, s->all.reserve((size_t)std::max<int64_t>(16, bytes) * s->world, st));
    if (int e = gaml_hip_eval_coverage_export_async(c, i, s->own.p, (int64_t)s->own.cap, &bytes, st)) return e;
    if (bytes > 0) COMM_NCCL(c, rccl()->AllGather(s->own.p, s->all.p, (size_t)bytes, ncclChar, s->comm, st));
    if (int e = gaml_hip_eval_coverage_finish_async(c, i, s->all.p, bytes > 0 ? s->world : 0, s->rank == 0, st)) return e;
  }
  const int32_t n_pb = gaml_hip_eval_pacbio_pending(c);
  for (int32_t i = 0; i < n_pb; i++) {  // PacBio sets with a penalty: the alignment intervals of all ranks' reads (device lists)
    const int64_t n_own = gaml_hip_eval_pacbio_intervals(c, i);
    if (n_own < 0) return (int)n_own;
    std::vector<long long> sizes((size_t)s->world, 0);
    long long mine = n_own;
    COMM_HIP(c, s->small.reserve(sizeof(long long) * (size_t)(s->world + 1), st));
    long long* d_sz = (long long*)s->small.p;
    COMM_HIP(c, hipMemcpyAsync(d_sz, &mine, sizeof(long long), hipMemcpyHostToDevice, st));
    COMM_NCCL(c, rccl()->AllGather(d_sz, d_sz + 1, 1, ncclInt64, s->comm, st));
    {
      std::vector<double> back((size_t)s->world);
      if (int e = comm_fetch_bounded(c, s, d_sz + 1, back.data(), s->world, "the exchange of the interval counts")) return e;
      memcpy(sizes.data(), back.data(), sizeof(long long) * (size_t)s->world);
    }
    long long total = 0;
    for (long long v : sizes) total += v;
    const long long width = std::max<long long>(1, *std::max_element(sizes.begin(), sizes.end()));  // intervals per rank in the gather (16 bytes each)
    COMM_HIP(c, s->own.reserve((size_t)width * 16, st));
    COMM_HIP(c, s->all.reserve((size_t)width * 16 * (size_t)s->world, st));
    COMM_HIP(c, s->pack.reserve((size_t)std::max<long long>(1, total) * 16, st));
    if (int e = gaml_hip_eval_pacbio_export_async(c, i, s->own.p, width, st)) return e;
    COMM_NCCL(c, rccl()->AllGather(s->own.p, s->all.p, (size_t)width * 16, ncclChar, s->comm, st));
    long long at = 0;
    for (int r = 0; r < s->world; r++) {  // rank order, the padding dropped
      if (sizes[r]) COMM_HIP(c, hipMemcpyAsync((char*)s->pack.p + at * 16, (const char*)s->all.p + (size_t)r * (size_t)width * 16, (size_t)sizes[r] * 16, hipMemcpyDeviceToDevice, st));
      at += sizes[r];
    }
    if (int e = gaml_hip_eval_pacbio_finish_async(c, i, s->pack.p, total, s->rank == 0, st)) return e;
  }
  return 0;
}

}  // namespace

namespace {

// The hot path's ONE exchange. Every rank contributes its partials (n doubles, in s->part) and two status words {its
// return code so far, the windows it aligned in this evaluation}:
//   mode 0 (default): all-gather; every rank adds the blocks up in RANK ORDER on the host -- the same doubles in the
//     same order everywhere and every run, whatever algorithm RCCL picks for the collective (the annealing
//     loop compares likelihoods with strict >, ties must be stable), and every rank sees every rank's status: a rank
//     that failed before the exchange still joins it, and ALL ranks return its error instead of waiting for it;
//   mode 1: all-reduce(sum) of the partials (the status words: sum of codes, sum of counts), the measured alternative.
// The wait is bounded (GAML_HIP_COMM_TIMEOUT_S, default 60): a peer that died inside an earlier collective never
// arrives -- then this rank aborts its communicator (ncclCommAbort) and reports, instead of hanging in the stream.
int comm_exchange(gaml_hip_ctx* c, CommState* s, size_t n, int local_rc, int64_t aligned, double* out) {
  hipStream_t st = ctx_stream(c);
  const size_t blk = n + 2;
  double* mine = (double*)s->part.p;
  if (!(local_rc == 0 && ctx_status_done(c))) {  // (a scoring launch that finished its own partials wrote the status words with them)
    hipLaunchKernelGGL(comm_status_kernel, dim3(1), dim3(64), 0, st, mine + n, (double)local_rc, (double)aligned);
    COMM_HIP(c, hipGetLastError());
  }
  ctx_set_status(c, nullptr, 0.0, 0.0);
  std::vector<double> host(blk * (size_t)(s->mode == 0 ? s->world : 1));
  if (s->mode == 0) {
    COMM_HIP(c, s->gath.reserve(blk * (size_t)s->world * sizeof(double), st));
    COMM_NCCL(c, rccl()->AllGather(mine, s->gath.p, blk, ncclDouble, s->comm, st));
    if (int e = gaml_hip_fetch_async(c, s->gath.p, (int32_t)host.size(), st)) return e;
  } else {
    COMM_NCCL(c, rccl()->AllReduce(mine, mine, blk, ncclDouble, ncclSum, s->comm, st));
    if (int e = gaml_hip_fetch_async(c, mine, (int32_t)host.size(), st)) return e;
  }
  const int w = ctx_fetch_wait_bounded(c, host.data(), (int32_t)host.size(), s->timeout_s);
  if (w < 0) return w;
  if (w > 0) {
    s->dead = true;
    if (rccl()->CommAbort) (void)rccl()->CommAbort(s->comm);
    return ctx_fail(c, GAML_HIP_EHIP, "sharded evaluation: the exchange did not complete within " + std::to_string((int)s->timeout_s) +
                                      " s (a peer failed or left the protocol); communicator aborted");
  }
  if (s->mode == 0) {
    for (size_t j = 0; j < n; j++) { double v = 0; for (int r = 0; r < s->world; r++) v += host[(size_t)r * blk + j]; out[j] = v; }
    for (int r = 0; r < s->world; r++)
      if (host[(size_t)r * blk + n] != 0.0)
        return r == s->rank ? local_rc : ctx_fail(c, (int)host[(size_t)r * blk + n], "sharded evaluation: rank " + std::to_string(r) + " failed (its own error text says why)");
    for (int r = 0; r < s->world; r++)
      if (host[(size_t)r * blk + n + 1] != (double)aligned)
        return ctx_fail(c, GAML_HIP_ESTATE, "sharded evaluation: ranks aligned different numbers of windows for the same paths (rank " + std::to_string(r) + ")");
  } else {
    for (size_t j = 0; j < n; j++) out[j] = host[j];
    if (host[n] != 0.0) return local_rc ? local_rc : ctx_fail(c, GAML_HIP_EHIP, "sharded evaluation: a rank failed");
    if (host[n + 1] != (double)aligned * s->world) return ctx_fail(c, GAML_HIP_ESTATE, "sharded evaluation: ranks aligned different numbers of windows for the same paths");
  }
  return local_rc;
}

}  // namespace

// the part of a sharded evaluation behind gaml_hip_eval_begin (+ the maxima exchange): kernels, penalty merges, exchange.
// begin_rc: what this rank's first phase returned -- a rank that failed there skips the kernels and joins the exchange
// with its code, so that no peer waits for it.
static int comm_finish_reduced(gaml_hip_ctx* c, CommState* s, int begin_rc, int64_t aligned, double* partials_out) {
  hipStream_t st = ctx_stream(c);
  const size_t nd = 4 * (size_t)std::max(1, gaml_hip_num_readsets(c));
  COMM_HIP(c, s->part.reserve((nd + 2) * sizeof(double), st));
  int rc = begin_rc;
  ctx_set_status(c, (double*)s->part.p + nd, 0.0, (double)aligned);
  if (!rc) rc = comm_score(c, s, (double*)s->part.p);
  else COMM_HIP(c, hipMemsetAsync(s->part.p, 0, nd * sizeof(double), st));
  if (rc) ctx_eval_abandon(c);
  const std::string why = rc ? gaml_hip_last_error(c) : "";
  const int e = comm_exchange(c, s, nd, rc, aligned, partials_out);
  if (e) { if (rc && e == rc) ctx_fail(c, rc, why); return e; }  // (this rank's own failure keeps its own text)
  ctx_note_reduced(c, partials_out);
  return 0;
}

int comm_eval_reduced(gaml_hip_ctx* c, const int32_t* paths, const int64_t* offs, int32_t n_paths, double* partials_out,
                      int32_t* total_len_out) {
  CommState* s = ctx_comm(c);
  if (!s || !s->comm) return ctx_fail(c, GAML_HIP_ESTATE, "no communicator on this context");
  if (s->dead) return ctx_fail(c, GAML_HIP_ESTATE, "the communicator of this context was aborted after a time-out");
  COMM_HIP(c, hipSetDevice(ctx_device(c)));
  int32_t tl = 0;
  int64_t aligned = 0;
  const int rc = comm_begin(c, s, paths, offs, n_paths, &tl, &aligned);
  if (total_len_out) *total_len_out = tl;
  return comm_finish_reduced(c, s, rc, aligned, partials_out);
}

int comm_eval_reduced_batch(gaml_hip_ctx* c, int32_t n_sets, const int32_t* paths, const int64_t* offs, const int32_t* set_offs,
                            double* partials_out, int32_t* total_lens_out) {
  CommState* s = ctx_comm(c);
  if (!s || !s->comm) return ctx_fail(c, GAML_HIP_ESTATE, "no communicator on this context");
  if (s->dead) return ctx_fail(c, GAML_HIP_ESTATE, "the communicator of this context was aborted after a time-out");
  if (n_sets <= 0) return 0;
  COMM_HIP(c, hipSetDevice(ctx_device(c)));
  hipStream_t st = ctx_stream(c);
  const size_t nd = 4 * (size_t)std::max(1, gaml_hip_num_readsets(c));
  COMM_HIP(c, s->part.reserve((nd * (size_t)n_sets + 2) * sizeof(double), st));
  int rc = 0;
  int64_t aligned_all = 0;
  std::string why;
  for (int32_t i = 0; i < n_sets && !rc; i++) {
    const int32_t p0 = set_offs[i], p1 = set_offs[i + 1];
    if (p1 < p0) { rc = ctx_fail(c, GAML_HIP_EINVAL, "set offsets must not decrease"); break; }
    int32_t tl = 0;
    int64_t aligned = 0;
    rc = comm_begin(c, s, paths, offs + p0, p1 - p0, &tl, &aligned);
    aligned_all += aligned;
    if (total_lens_out) total_lens_out[i] = tl;
    if (!rc) rc = comm_score(c, s, (double*)s->part.p + nd * (size_t)i);
  }
  if (rc) { why = gaml_hip_last_error(c); ctx_eval_abandon(c); }
  const int e = comm_exchange(c, s, nd * (size_t)n_sets, rc, aligned_all, partials_out);  // ONE for the batch
  if (e) { if (rc && e == rc) ctx_fail(c, rc, why); return e; }
  ctx_note_reduced(c, partials_out + nd * (size_t)(n_sets - 1));
  return 0;
}

// =============================================================================================================
// N device shards in one process
// =============================================================================================================
namespace {

// One thread per shard: its device stays current there, the shard's host preparation (window placement, occurrence
// images) runs in parallel with the other shards'. Jobs are handed over through two counters; the worker spins for a
// while after a job (a blocking CalcProb every ~50 us keeps it hot) and sleeps on a condition variable otherwise.
struct Worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<int()> job;
  std::atomic<unsigned long long> posted{0}, done{0};
  std::atomic<bool> sleeping{false};
  bool quit = false;
  int rc = 0;
  int device = -1;

  void loop() {
    if (device >= 0) (void)hipSetDevice(device);
    unsigned long long seen = 0;
    for (;;) {
      unsigned spins = 0;
      while (posted.load(std::memory_order_acquire) == seen) {
        if (++spins < 200000u) { __builtin_ia32_pause(); continue; }  // ~ a few hundred us
        std::unique_lock<std::mutex> lk(m);
        sleeping.store(true, std::memory_order_seq_cst);
        cv.wait(lk, [&] { return posted.load(std::memory_order_acquire) != seen; });
        sleeping.store(false, std::memory_order_seq_cst);
        break;
      }
      seen = posted.load(std::memory_order_acquire);
      if (quit) { done.store(seen, std::memory_order_release); return; }
      rc = job();
      done.store(seen, std::memory_order_release);
    }
  }
  void post(std::function<int()> f) {
    job = std::move(f);
    posted.fetch_add(1, std::memory_order_seq_cst);
    if (sleeping.load(std::memory_order_seq_cst)) { std::lock_guard<std::mutex> lk(m); cv.notify_one(); }
  }
  int wait() {
    const unsigned long long want = posted.load(std::memory_order_acquire);
    unsigned spins = 0;
    while (done.load(std::memory_order_acquire) != want) { if (++spins > 4096u) std::this_thread::yield(); else __builtin_ia32_pause(); }
    return rc;
  }
};

}  // namespace

struct MultiState {
  gaml_hip_ctx* parent = nullptr;
  std::vector<gaml_hip_ctx*> kids;
  std::vector<int> devices;
  std::vector<std::unique_ptr<Worker>> workers;
  bool have_comm = false;   // every shard holds a communicator (distinct devices, RCCL loaded)
  int exchange = 0;         // 0: pinned-host partials summed in rank order by the calling thread; 1: RCCL all-reduce
  // host-exchange scratch, per shard
  struct KidBufs { DevMem part, own, all; };
  std::vector<KidBufs> bufs;
  std::string err;

  int n() const { return (int)kids.size(); }
  // run f(k, shard k) on every shard's thread; first failure wins, its text becomes the context's error
  template <class F>
  int run_all(F f) {
    for (int k = 0; k < n(); k++) workers[k]->post([this, k, f] { return f(k, kids[k]); });
    int rc = 0, bad = -1;
    for (int k = 0; k < n(); k++) { const int r = workers[k]->wait(); if (r < 0 && rc == 0) { rc = r; bad = k; } }
    if (rc) ctx_fail(parent, rc, std::string("shard ") + std::to_string(bad) + " (device " + std::to_string(devices[bad]) + "): " + gaml_hip_last_error(kids[bad]));
    return rc;
  }
};

void multi_destroy(MultiState* m) {
  if (!m) return;
  if (!m->kids.empty()) {
    m->run_all([m](int k, gaml_hip_ctx* kid) {
      auto& b = m->bufs[k];
      if (ctx_device(kid) >= 0) { (void)hipStreamSynchronize(ctx_stream(kid)); b.part.release(); b.own.release(); b.all.release(); }
      gaml_hip_destroy(kid);  // incl. its communicator
      return 0;
    });
  }
  for (auto& w : m->workers) {
    w->quit = true;
    w->post([] { return 0; });
    if (w->th.joinable()) w->th.join();
  }
  delete m;
}

gaml_hip_ctx* multi_shard(const MultiState* m, int i) { return (i >= 0 && i < m->n()) ? m->kids[i] : nullptr; }
int multi_num_shards(const MultiState* m) { return m->n(); }
const char* multi_last_error(const MultiState* m) { return m->err.c_str(); }

int multi_set_graph(MultiState* m, int32_t n_nodes, const char* bases, const int64_t* offs) {
  return m->run_all([=](int, gaml_hip_ctx* kid) { return gaml_hip_set_graph(kid, n_nodes, bases, offs); });
}
int multi_load_graph(MultiState* m, const char* file) {
  return m->run_all([=](int, gaml_hip_ctx* kid) { return gaml_hip_load_graph(kid, file); });
}

namespace {
// every shard must hand out the same read-set handle
template <class F>
int add_everywhere(MultiState* m, F f) {
  std::vector<int> h((size_t)m->n(), -1);
  int rc = m->run_all([&](int k, gaml_hip_ctx* kid) { h[k] = f(kid); return h[k] < 0 ? h[k] : 0; });
  if (rc) return rc;
  for (int k = 1; k < m->n(); k++) if (h[k] != h[0]) return ctx_fail(m->parent, GAML_HIP_ESTATE, "shards disagree on the read-set handle");
  return h[0];
}
}  // namespace

int multi_add_single(MultiState* m, const gaml_single_cfg* cfg, int32_t n, const char* bases, const int64_t* offs) {
  return add_everywhere(m, [=](gaml_hip_ctx* kid) { return gaml_hip_add_single(kid, cfg, n, bases, offs); });
}
int multi_add_paired(MultiState* m, const gaml_paired_cfg* cfg, int32_t n, const char* b1, const int64_t* o1, const char* b2,
                     const int64_t* o2) {
  return add_everywhere(m, [=](gaml_hip_ctx* kid) { return gaml_hip_add_paired(kid, cfg, n, b1, o1, b2, o2); });
}
int multi_add_pacbio(MultiState* m, const gaml_single_cfg* cfg, int32_t n, const int32_t* lens) {
  return add_everywhere(m, [=](gaml_hip_ctx* kid) { return gaml_hip_add_pacbio(kid, cfg, n, lens); });
}
int multi_add_pacbio_reads(MultiState* m, const gaml_single_cfg* cfg, int32_t n, const char* bases, const int64_t* offs,
                           const char* names) {
  return add_everywhere(m, [=](gaml_hip_ctx* kid) { return gaml_hip_add_pacbio_reads(kid, cfg, n, bases, offs, names); });
}
int multi_put_window_records(MultiState* m, int rs, int mate, const int32_t* sub, int32_t len, const gaml_aligment* recs, int64_t n) {
  return m->run_all([=](int, gaml_hip_ctx* kid) { return gaml_hip_put_window_records(kid, rs, mate, sub, len, recs, n); });  // a shard keeps its own reads' records
}
int multi_put_pacbio_records(MultiState* m, int rs, const int32_t* sub, int32_t len, const gaml_pacbio_aligment* recs, int64_t n) {
  return m->run_all([=](int, gaml_hip_ctx* kid) { return gaml_hip_put_pacbio_records(kid, rs, sub, len, recs, n); });
}
int32_t multi_pacbio_missing(MultiState* m, int rs, const int32_t* path, int32_t n, int32_t* ranges, int32_t cap) {
  // which sub-walks are cached is the same on every shard (entries are created for all of them, with or without records)
  int32_t r = gaml_hip_pacbio_missing(m->kids[0], rs, path, n, ranges, cap);
  if (r < 0) ctx_fail(m->parent, r, gaml_hip_last_error(m->kids[0]));
  return r;
}
int multi_pacbio_ingest_sam(MultiState* m, int rs, const int32_t* path, int32_t n, const char* sam, int64_t sam_len, int64_t* filed) {
  std::vector<int64_t> f((size_t)m->n(), 0);
  int rc = m->run_all([&](int k, gaml_hip_ctx* kid) { return gaml_hip_pacbio_ingest_sam(kid, rs, path, n, sam, sam_len, &f[k]); });
  if (filed) { *filed = 0; for (int64_t v : f) *filed += v; }
  return rc;
}
int64_t multi_pacbio_records(MultiState* m, int rs, const int32_t* sub, int32_t len, gaml_pacbio_aligment* out, int64_t cap) {
  int64_t total = 0;
  bool cached = false;
  for (gaml_hip_ctx* kid : m->kids) {  // host-only: shard by shard (ascending read ids)
    const int64_t r = gaml_hip_pacbio_records(kid, rs, sub, len, out ? out + std::min(total, cap) : nullptr, std::max<int64_t>(0, cap - total));
    if (r < -1) return r;
    if (r >= 0) { cached = true; total += r; }
  }
  return cached ? total : -1;
}

// -------------------------------------------------------------------------------------------------------------
// one evaluation over all shards
// -------------------------------------------------------------------------------------------------------------
namespace {

int multi_maxima_exchange(MultiState* m, const std::vector<int64_t>& pending) {
  for (int k = 1; k < m->n(); k++)
    if (pending[k] != pending[0]) return ctx_fail(m->parent, GAML_HIP_ESTATE, "shards registered different windows for the same paths");
  const int64_t np = pending[0];
  if (np == 0) return 0;
  std::vector<int32_t> mx((size_t)np, INT_MIN), one((size_t)np);
  for (gaml_hip_ctx* kid : m->kids) {  // host values: window maxima of the shard's reads
    if (gaml_hip_eval_pending_maxpos(kid, one.data(), np) != np) return ctx_fail(m->parent, GAML_HIP_ESTATE, "pending maxima changed under the exchange");
    for (int64_t i = 0; i < np; i++) mx[i] = std::max(mx[i], one[i]);
  }
  for (gaml_hip_ctx* kid : m->kids)
    if (int e = gaml_hip_eval_apply_maxpos(kid, mx.data(), np)) return ctx_fail(m->parent, e, gaml_hip_last_error(kid));
  return 0;
}

// host exchange with a coverage penalty somewhere: scoring kernels per shard, maps / events merged through host memory
int multi_finish_with_penalty(MultiState* m, std::vector<std::vector<double>>& part) {
  const int n = m->n();
  const size_t nd = part[0].size();
  std::vector<int32_t> n_maps((size_t)n, 0);
  int rc = m->run_all([&](int k, gaml_hip_ctx* kid) {
    hipStream_t st = ctx_stream(kid);
    if (hipSuccess != m->bufs[k].part.reserve(nd * sizeof(double), st)) return ctx_fail(kid, GAML_HIP_EHIP, "hipMalloc of the partials failed");
    n_maps[k] = gaml_hip_eval_score_async(kid, m->bufs[k].part.p, st);
    return n_maps[k] < 0 ? n_maps[k] : 0;
  });
  if (rc) return rc;
  for (int k = 1; k < n; k++) if (n_maps[k] != n_maps[0]) return ctx_fail(m->parent, GAML_HIP_ESTATE, "shards disagree on the number of coverage maps");
  for (int32_t i = 0; i < n_maps[0]; i++) {
    std::vector<int64_t> bytes((size_t)n, 0);
    for (int k = 0; k < n; k++) if (int e = gaml_hip_eval_coverage_export_async(m->kids[k], i, nullptr, 0, &bytes[k], nullptr)) return ctx_fail(m->parent, e, gaml_hip_last_error(m->kids[k]));
    for (int k = 1; k < n; k++) if (bytes[k] != bytes[0]) return ctx_fail(m->parent, GAML_HIP_ESTATE, "shards disagree on the coverage map size");
    const size_t b = (size_t)bytes[0];
    std::vector<unsigned char> all(std::max<size_t>(1, b * (size_t)n));
    rc = m->run_all([&](int k, gaml_hip_ctx* kid) {
      hipStream_t st = ctx_stream(kid);
      auto& kb = m->bufs[k];
      if (hipSuccess != kb.own.reserve(std::max<size_t>(16, b), st) || hipSuccess != kb.all.reserve(std::max<size_t>(16, b * (size_t)n), st))
        return ctx_fail(kid, GAML_HIP_EHIP, "hipMalloc of the coverage maps failed");
      int64_t got = 0;
      if (int e = gaml_hip_eval_coverage_export_async(kid, i, kb.own.p, (int64_t)kb.own.cap, &got, st)) return e;
      if (b && hipSuccess != hipMemcpyAsync(all.data() + b * (size_t)k, kb.own.p, b, hipMemcpyDeviceToHost, st)) return ctx_fail(kid, GAML_HIP_EHIP, "D2H of the coverage map failed");
      return hipSuccess == hipStreamSynchronize(st) ? 0 : ctx_fail(kid, GAML_HIP_EHIP, "stream synchronise failed");
    });
    if (rc) return rc;
    rc = m->run_all([&](int k, gaml_hip_ctx* kid) {
      hipStream_t st = ctx_stream(kid);
      if (b && hipSuccess != hipMemcpyAsync(m->bufs[k].all.p, all.data(), b * (size_t)n, hipMemcpyHostToDevice, st)) return ctx_fail(kid, GAML_HIP_EHIP, "H2D of the coverage maps failed");
      int e = gaml_hip_eval_coverage_finish_async(kid, i, m->bufs[k].all.p, b ? n : 0, k == 0, st);
      if (e) return e;
      return hipSuccess == hipStreamSynchronize(st) ? 0 : ctx_fail(kid, GAML_HIP_EHIP, "stream synchronise failed");  // `all` (host) is read by the copy
    });
    if (rc) return rc;
  }
  const int32_t n_pb = gaml_hip_eval_pacbio_pending(m->kids[0]);
  for (int32_t i = 0; i < n_pb; i++) {
    // every shard's alignment intervals to every shard, device to device (rank order)
    std::vector<int64_t> cnt((size_t)n, 0);
    int64_t total = 0;
    for (int k = 0; k < n; k++) {
      cnt[k] = gaml_hip_eval_pacbio_intervals(m->kids[k], i);
      if (cnt[k] < 0) return ctx_fail(m->parent, (int)cnt[k], gaml_hip_last_error(m->kids[k]));
      total += cnt[k];
    }
    rc = m->run_all([&](int k, gaml_hip_ctx* kid) {
      hipStream_t st = ctx_stream(kid);
      auto& kb = m->bufs[k];
      if (hipSuccess != kb.own.reserve((size_t)std::max<int64_t>(1, cnt[k]) * 16, st) || hipSuccess != kb.all.reserve((size_t)std::max<int64_t>(1, total) * 16, st))
        return ctx_fail(kid, GAML_HIP_EHIP, "hipMalloc of the interval lists failed");
      if (int e = gaml_hip_eval_pacbio_export_async(kid, i, kb.own.p, cnt[k], st)) return e;
      return hipSuccess == hipStreamSynchronize(st) ? 0 : ctx_fail(kid, GAML_HIP_EHIP, "stream synchronise failed");
    });
    if (rc) return rc;
    rc = m->run_all([&](int k, gaml_hip_ctx* kid) {
      hipStream_t st = ctx_stream(kid);
      int64_t at = 0;
      for (int r = 0; r < n; r++) {
        if (cnt[r]) {
          const int dev_k = ctx_device(kid), dev_r = ctx_device(m->kids[r]);
          char* dst = (char*)m->bufs[k].all.p + at * 16;
          const hipError_t e = dev_k == dev_r ? hipMemcpyAsync(dst, m->bufs[r].own.p, (size_t)cnt[r] * 16, hipMemcpyDeviceToDevice, st)
                                              : hipMemcpyPeerAsync(dst, dev_k, m->bufs[r].own.p, dev_r, (size_t)cnt[r] * 16, st);
          if (e != hipSuccess) return ctx_fail(kid, GAML_HIP_EHIP, std::string("copy of a shard's interval list failed: ") + hipGetErrorString(e));
        }
        at += cnt[r];
      }
      if (int e = gaml_hip_eval_pacbio_finish_async(kid, i, m->bufs[k].all.p, total, k == 0, st)) return e;
      return hipSuccess == hipStreamSynchronize(st) ? 0 : ctx_fail(kid, GAML_HIP_EHIP, "stream synchronise failed");  // (the peers' `own` lists are read by the copies)
    });
    if (rc) return rc;
  }
  return m->run_all([&](int k, gaml_hip_ctx* kid) {
    if (int e = gaml_hip_fetch_async(kid, m->bufs[k].part.p, (int32_t)nd, ctx_stream(kid))) return e;
    return gaml_hip_fetch_wait(kid, part[k].data(), (int32_t)nd);
  });
}

}  // namespace

int multi_calc_partials(MultiState* m, const int32_t* paths, const int64_t* offs, int32_t n_paths, double* partials_out,
                        int32_t* total_len_out) {
  const int n = m->n();
  const size_t nd = 4 * (size_t)std::max(1, gaml_hip_num_readsets(m->kids[0]));
  std::vector<std::vector<double>> part((size_t)n, std::vector<double>(nd, 0.0));
  std::vector<int32_t> tl((size_t)n, 0);
  if (m->exchange == 1) {
    // Two hand-overs per step. First every shard registers / aligns on its own thread; back here the shards' return
    // codes and the windows they aligned are compared BEFORE any collective is posted -- a shard that failed must not
    // leave the others waiting in one -- and the new windows' maxima are merged on the host (the shards share this
    // process). Then kernels and the ONE RCCL exchange per shard, on its stream; all shards end up with the same values.
    std::vector<int64_t> pending((size_t)n, 0);
    int rc = m->run_all([&](int k, gaml_hip_ctx* kid) { return gaml_hip_eval_begin(kid, paths, offs, n_paths, &pending[k], &tl[k]); });
    if (!rc) rc = multi_maxima_exchange(m, pending);
    if (rc) { for (gaml_hip_ctx* kid : m->kids) ctx_eval_abandon(kid); return rc; }
    rc = m->run_all([&](int k, gaml_hip_ctx* kid) {
      CommState* s = ctx_comm(kid);
      if (!s || !s->comm) return ctx_fail(kid, GAML_HIP_ESTATE, "no communicator on this shard");
      if (s->dead) return ctx_fail(kid, GAML_HIP_ESTATE, "the communicator of this shard was aborted after a time-out");
      if (hipSetDevice(ctx_device(kid)) != hipSuccess) return ctx_fail(kid, GAML_HIP_EHIP, "hipSetDevice failed");
      return comm_finish_reduced(kid, s, 0, pending[k], part[k].data());
    });
    if (rc) return rc;
    memcpy(partials_out, part[0].data(), nd * sizeof(double));
    if (total_len_out) *total_len_out = tl[0];
    return 0;
  }
  // host exchange: blocking evaluation per shard (per-block partials land in pinned host memory), summed here in rank order
  const bool penalty = ctx_has_penalty(m->kids[0]);
  std::vector<int64_t> pending((size_t)n, 0);
  std::vector<char> finished((size_t)n, 0);
  int rc = m->run_all([&](int k, gaml_hip_ctx* kid) {
    if (int e = gaml_hip_eval_begin(kid, paths, offs, n_paths, &pending[k], &tl[k])) return e;
    if (pending[k] == 0 && !penalty) { finished[k] = 1; return gaml_hip_eval_finish(kid, part[k].data()); }  // warm path: one hand-over per step
    return 0;
  });
  if (rc) return rc;
  if (int e = multi_maxima_exchange(m, pending)) return e;
  if (penalty) {
    if (int e = multi_finish_with_penalty(m, part)) return e;
  } else if (!finished[0]) {
    rc = m->run_all([&](int k, gaml_hip_ctx* kid) { return gaml_hip_eval_finish(kid, part[k].data()); });
    if (rc) return rc;
  }
  for (size_t j = 0; j < nd; j++) { double v = 0; for (int k = 0; k < n; k++) v += part[(size_t)k][j]; partials_out[j] = v; }
  for (gaml_hip_ctx* kid : m->kids) ctx_note_reduced(kid, partials_out);
  if (total_len_out) *total_len_out = tl[0];
  return 0;
}

int multi_combine(MultiState* m, const double* partials, int32_t total_len, double* prob_out, int32_t* zeros_out) {
  return gaml_hip_combine_partials(m->kids[0], partials, total_len, prob_out, zeros_out);
}

int multi_calc_prob_batch(MultiState* m, int32_t n_sets, const int32_t* paths, const int64_t* offs, const int32_t* set_offs,
                          double* probs_out, int32_t* zeros_out, int32_t* total_lens_out) {
  const int ns = std::max(1, gaml_hip_num_readsets(m->kids[0]));
  const size_t nd = 4 * (size_t)ns;
  if (m->exchange == 1) {  // ONE all-reduce for the whole batch
    const int n = m->n();
    std::vector<std::vector<double>> part((size_t)n, std::vector<double>(nd * (size_t)n_sets, 0.0));
    std::vector<std::vector<int32_t>> tls((size_t)n, std::vector<int32_t>((size_t)n_sets, 0));
    int rc = m->run_all([&](int k, gaml_hip_ctx* kid) { return comm_eval_reduced_batch(kid, n_sets, paths, offs, set_offs, part[k].data(), tls[k].data()); });
    if (rc) return rc;
    for (int32_t i = 0; i < n_sets; i++) {
      if (int e = multi_combine(m, part[0].data() + nd * (size_t)i, tls[0][i], &probs_out[i], zeros_out ? zeros_out + (size_t)i * 2 * ns : nullptr)) return e;
      if (total_lens_out) total_lens_out[i] = tls[0][i];
    }
    return 0;
  }
  std::vector<double> part(nd);
  for (int32_t i = 0; i < n_sets; i++) {
    const int32_t p0 = set_offs[i], p1 = set_offs[i + 1];
    if (p1 < p0) return ctx_fail(m->parent, GAML_HIP_EINVAL, "set offsets must not decrease");
    int32_t tl = 0;
    if (int e = multi_calc_partials(m, paths, offs + p0, p1 - p0, part.data(), &tl)) return e;
    if (int e = multi_combine(m, part.data(), tl, &probs_out[i], zeros_out ? zeros_out + (size_t)i * 2 * ns : nullptr)) return e;
    if (total_lens_out) total_lens_out[i] = tl;
  }
  return 0;
}

// -------------------------------------------------------------------------------------------------------------
// introspection
// -------------------------------------------------------------------------------------------------------------
int multi_num_readsets(const MultiState* m) { return gaml_hip_num_readsets(m->kids[0]); }
int multi_readset_kind(const MultiState* m, int rs) { return gaml_hip_readset_kind(m->kids[0], rs); }
int64_t multi_readset_reads(const MultiState* m, int rs) { return gaml_hip_readset_reads(m->kids[0], rs); }
int32_t multi_num_nodes(const MultiState* m) { return gaml_hip_num_nodes(m->kids[0]); }
int32_t multi_node_len(const MultiState* m, int32_t node) { return gaml_hip_node_len(m->kids[0], node); }

int multi_read_probs(MultiState* m, int rs, double* out, int64_t n) {
  // shards are contiguous id ranges in rank order: concatenation = read order
  const int64_t total = gaml_hip_readset_reads(m->kids[0], rs);
  if (total < 0 || !out) return ctx_fail(m->parent, GAML_HIP_EINVAL, "bad arguments");
  if (n < total) return ctx_fail(m->parent, GAML_HIP_EINVAL, "output too small");
  std::vector<int64_t> lo((size_t)m->n() + 1, 0);
  for (int k = 0; k <= m->n(); k++) lo[k] = total * k / m->n();
  int rc = m->run_all([&](int k, gaml_hip_ctx* kid) { const int r = gaml_hip_read_probs(kid, rs, out + lo[k], lo[k + 1] - lo[k]); return r < 0 ? r : 0; });
  return rc ? rc : (int)std::min<int64_t>(total, 0x7fffffff);
}
int multi_bad_bases(MultiState* m, int rs, int64_t* out) { return gaml_hip_bad_bases(m->kids[0], rs, out); }
int64_t multi_window_count(const MultiState* m, int rs, int mate) { return gaml_hip_window_count(m->kids[0], rs, mate); }
int64_t multi_window_records(MultiState* m, int rs, int mate, const int32_t* sub, int32_t len, gaml_aligment* out, int64_t cap) {
  std::vector<gaml_aligment> all;
  bool cached = false;
  for (gaml_hip_ctx* kid : m->kids) {
    const int64_t cnt = gaml_hip_window_records(kid, rs, mate, sub, len, nullptr, 0);
    if (cnt < -1) return cnt;
    if (cnt < 0) continue;
    cached = true;
    const size_t at = all.size();
    all.resize(at + (size_t)cnt);
    if (cnt) gaml_hip_window_records(kid, rs, mate, sub, len, all.data() + at, cnt);
  }
  if (!cached) return -1;
  std::stable_sort(all.begin(), all.end(), [](const gaml_aligment& a, const gaml_aligment& b) {  // (position, read) like one window vector
    return a.position != b.position ? a.position < b.position : a.read_id < b.read_id;
  });
  for (int64_t i = 0; i < (int64_t)all.size() && i < cap; i++) out[i] = all[(size_t)i];
  return (int64_t)all.size();
}
int64_t multi_align_window(MultiState* m, int rs, int mate, const int32_t* sub, int32_t len) {
  std::vector<int64_t> cnt((size_t)m->n(), 0);
  int rc = m->run_all([&](int k, gaml_hip_ctx* kid) { cnt[k] = gaml_hip_align_window(kid, rs, mate, sub, len); return cnt[k] < 0 ? (int)cnt[k] : 0; });
  if (rc) return rc;
  int64_t total = 0;
  for (int64_t v : cnt) total += v;
  return total;
}
int multi_compact_tables(MultiState* m) { for (gaml_hip_ctx* kid : m->kids) gaml_hip_compact_tables(kid); return GAML_HIP_OK; }
int multi_sync(MultiState* m) { return m->run_all([](int, gaml_hip_ctx* kid) { return gaml_hip_sync(kid); }); }
int multi_set_event_timing(MultiState* m, int on) { return m->run_all([on](int, gaml_hip_ctx* kid) { return gaml_hip_set_event_timing(kid, on); }); }
int multi_kernel_stats(MultiState* m, int reset, int64_t* launches, double* device_us, double* algo_bytes) {
  std::vector<int64_t> l((size_t)m->n(), 0);
  std::vector<double> d((size_t)m->n(), 0), b((size_t)m->n(), 0);
  int rc = m->run_all([&](int k, gaml_hip_ctx* kid) { return gaml_hip_kernel_stats(kid, reset, &l[k], &d[k], &b[k]); });
  if (launches) { *launches = 0; for (int64_t v : l) *launches += v; }
  if (device_us) { *device_us = 0; for (double v : d) *device_us += v; }
  if (algo_bytes) { *algo_bytes = 0; for (double v : b) *algo_bytes += v; }
  return rc;
}
int multi_last_timing(const MultiState* m, double* out3) { return gaml_hip_last_timing(m->kids[0], out3); }

}  // namespace gaml

// =============================================================================================================
// C ABI of this file
// =============================================================================================================
using namespace gaml;

extern "C" {

int gaml_hip_create_multi(gaml_hip_ctx** out, const int32_t* devices, int32_t n_devices) {
  if (!out || !devices || n_devices < 1 || n_devices > 64) return GAML_HIP_EINVAL;
  *out = nullptr;
  std::unique_ptr<MultiState> m(new MultiState());
  gaml_hip_ctx* parent = ctx_new_parent();
  m->parent = parent;
  bool distinct = true, all_gpu = true;
  for (int32_t i = 0; i < n_devices; i++) {
    all_gpu = all_gpu && devices[i] >= 0;
    for (int32_t j = 0; j < i; j++) distinct = distinct && devices[i] != devices[j];
  }
  int rc = GAML_HIP_OK;
  int caller_device = -1;  // gaml_hip_create makes its device current: the caller's choice is put back below
  if (all_gpu && hipGetDevice(&caller_device) != hipSuccess) caller_device = -1;
  for (int32_t i = 0; i < n_devices && rc == GAML_HIP_OK; i++) {
    gaml_hip_ctx* kid = nullptr;
    rc = gaml_hip_create(&kid, devices[i]);
    if (rc == GAML_HIP_OK) rc = gaml_hip_set_shard(kid, i, n_devices);
    if (kid) { m->kids.push_back(kid); m->devices.push_back(devices[i]); }
  }
  if (rc != GAML_HIP_OK) {
    for (gaml_hip_ctx* kid : m->kids) gaml_hip_destroy(kid);
    gaml_hip_destroy(parent);
    return rc;
  }
  m->bufs.resize((size_t)n_devices);
  for (int32_t i = 0; i < n_devices; i++) {
    m->workers.emplace_back(new Worker());
    Worker* w = m->workers.back().get();
    w->device = devices[i];
    w->th = std::thread([w] { w->loop(); });
  }
  // one RCCL communicator per shard when every shard has its own GPU (RCCL refuses two ranks on one device)
  const char* want = getenv("GAML_HIP_EXCHANGE");  // "rccl" (default when possible) | "host"
  if (distinct && all_gpu && n_devices > 0 && !(want && !strcmp(want, "host"))) {
    Rccl* r = rccl();
    if (r->err.empty()) {
      std::vector<ncclComm_t> comms((size_t)n_devices, nullptr);
      std::vector<int> devs(devices, devices + n_devices);
      const ncclResult_t nr = r->CommInitAll(comms.data(), n_devices, devs.data());
      if (nr == ncclSuccess) {
        for (int32_t i = 0; i < n_devices; i++) {
          CommState* s = new CommState();
          s->comm = comms[(size_t)i]; s->rank = i; s->world = n_devices; s->timeout_s = comm_timeout_s();
          ctx_set_comm(m->kids[(size_t)i], s);
        }
        m->have_comm = true;
        m->exchange = 1;
      } else {
        m->err = std::string("ncclCommInitAll: ") + r->GetErrorString(nr) + " -- falling back to the host exchange";
      }
    } else {
      m->err = r->err + " -- falling back to the host exchange";
    }
    if (!m->have_comm && want && !strcmp(want, "rccl")) {  // asked for explicitly: fail loudly
      fprintf(stderr, "gaml_hip_create_multi: %s\n", m->err.c_str());
      MultiState* raw = m.release();
      ctx_set_multi(parent, raw);
      gaml_hip_destroy(parent);
      return GAML_HIP_EHIP;
    }
  }
  ctx_set_multi(parent, m.release());
  if (caller_device >= 0) (void)hipSetDevice(caller_device);
  *out = parent;
  return GAML_HIP_OK;
}

int gaml_hip_create_from_env(gaml_hip_ctx** out) {
  if (!out) return GAML_HIP_EINVAL;
  const char* e = getenv("GAML_HIP_DEVICES");
  if (!e || !*e) return gaml_hip_create(out, 0);
  std::vector<int32_t> devs;
  if (!strcmp(e, "all")) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) { *out = nullptr; fprintf(stderr, "GAML_HIP_DEVICES=all: no HIP device\n"); return GAML_HIP_ENODEVICE; }
    for (int i = 0; i < count; i++) devs.push_back(i);
  } else {
    for (const char* p = e; *p;) {
      char* end = nullptr;
      const long v = strtol(p, &end, 10);
      if (end == p || v < -1 || v > 1023) { *out = nullptr; fprintf(stderr, "GAML_HIP_DEVICES: expected \"all\" or a comma-separated list of device ordinals, got \"%s\"\n", e); return GAML_HIP_EINVAL; }
      devs.push_back((int32_t)v);
      p = *end == ',' ? end + 1 : end;
      if (*end && *end != ',') { *out = nullptr; fprintf(stderr, "GAML_HIP_DEVICES: unexpected character in \"%s\"\n", e); return GAML_HIP_EINVAL; }
    }
  }
  if (devs.size() == 1) return gaml_hip_create(out, devs[0]);
  return gaml_hip_create_multi(out, devs.data(), (int32_t)devs.size());
}

int gaml_hip_num_shards(const gaml_hip_ctx* ctx) {
  if (!ctx) return 0;
  return ctx_multi(ctx) ? multi_num_shards(ctx_multi(ctx)) : 1;
}

int gaml_hip_set_exchange(gaml_hip_ctx* ctx, int32_t mode) {
  if (!ctx || (mode != GAML_HIP_EXCHANGE_HOST && mode != GAML_HIP_EXCHANGE_RCCL && mode != GAML_HIP_EXCHANGE_RCCL_ALLREDUCE)) return ctx_fail(ctx, GAML_HIP_EINVAL, "bad arguments");
  const bool rccl_mode = mode != GAML_HIP_EXCHANGE_HOST;
  MultiState* m = ctx_multi(ctx);
  if (!m) {  // one shard per process: only the form of the RCCL exchange can be chosen
    CommState* s = ctx_comm(ctx);
    if (!s || !rccl_mode) return ctx_fail(ctx, GAML_HIP_ESTATE, "not a multi-device context (and no communicator whose exchange could be chosen)");
    s->mode = mode == GAML_HIP_EXCHANGE_RCCL_ALLREDUCE ? 1 : 0;
    return GAML_HIP_OK;
  }
  if (rccl_mode && !m->have_comm)
    return ctx_fail(ctx, GAML_HIP_ESTATE, "no RCCL communicator on this context (" + (m->err.empty() ? std::string("shards share a device") : m->err) + ")");
  m->exchange = rccl_mode ? 1 : 0;
  if (rccl_mode) for (gaml_hip_ctx* kid : m->kids) if (CommState* s = ctx_comm(kid)) s->mode = mode == GAML_HIP_EXCHANGE_RCCL_ALLREDUCE ? 1 : 0;
  return GAML_HIP_OK;
}

int gaml_hip_get_exchange(const gaml_hip_ctx* ctx) {
  if (!ctx) return GAML_HIP_EINVAL;
  if (MultiState* m = ctx_multi(ctx)) {
    if (!m->exchange) return GAML_HIP_EXCHANGE_HOST;
    CommState* s = ctx_comm(m->kids[0]);
    return s && s->mode == 1 ? GAML_HIP_EXCHANGE_RCCL_ALLREDUCE : GAML_HIP_EXCHANGE_RCCL;
  }
  CommState* s = ctx_comm(ctx);
  return !s ? GAML_HIP_EXCHANGE_HOST : (s->mode == 1 ? GAML_HIP_EXCHANGE_RCCL_ALLREDUCE : GAML_HIP_EXCHANGE_RCCL);
}

int gaml_hip_comm_unique_id(void* id_out) {
  if (!id_out) return GAML_HIP_EINVAL;
  Rccl* r = rccl();
  if (!r->err.empty()) { fprintf(stderr, "gaml_hip_comm_unique_id: %s\n", r->err.c_str()); return GAML_HIP_EHIP; }
  ncclUniqueId id;
  if (r->GetUniqueId(&id) != ncclSuccess) return GAML_HIP_EHIP;
  static_assert(sizeof(id) == GAML_HIP_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id_out, &id, sizeof(id));
  return GAML_HIP_OK;
}

int gaml_hip_comm_init_rank(gaml_hip_ctx* c, const void* id_bytes, int32_t rank, int32_t world) {
  if (!c || !id_bytes || world < 1 || rank < 0 || rank >= world) return ctx_fail(c, GAML_HIP_EINVAL, "bad arguments");
  if (ctx_multi(c)) return ctx_fail(c, GAML_HIP_ESTATE, "a multi-device context owns its communicators");
  if (ctx_device(c) < 0) return ctx_fail(c, GAML_HIP_ENODEVICE, "a communicator needs a HIP device: this context is host-only");
  if (ctx_comm(c)) return ctx_fail(c, GAML_HIP_ESTATE, "this context already has a communicator");
  if (ctx_peers(c) != world) return ctx_fail(c, GAML_HIP_ESTATE, "world size does not match gaml_hip_set_shard / gaml_hip_set_presharded");
  if (ctx_world(c) > 1 && ctx_rank(c) != rank) return ctx_fail(c, GAML_HIP_ESTATE, "rank does not match gaml_hip_set_shard");
  Rccl* r = rccl();
  if (!r->err.empty()) return ctx_fail(c, GAML_HIP_EHIP, r->err);
  COMM_HIP(c, hipSetDevice(ctx_device(c)));
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof(id));
  std::unique_ptr<CommState> s(new CommState());
  s->rank = rank; s->world = world; s->timeout_s = comm_timeout_s();
  COMM_NCCL(c, r->CommInitRank(&s->comm, world, id, rank));
  ctx_set_comm(c, s.release());
  return GAML_HIP_OK;
}

}  // extern "C"
