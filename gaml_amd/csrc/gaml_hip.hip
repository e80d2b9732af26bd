// gaml_hip.hip -- context, device memory, launches and the C ABI of libgaml_hip.so.
// The kernels are in kernels.hip.h, the host data model in host_model.{h,cc}.
#include "ctx.hip.h"

using namespace gaml;
using namespace gaml::detail;

namespace {

int fail(gaml_hip_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}
#define HIP_TRY(c, expr)                                                                      \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess)                                                                    \
      return fail(c, GAML_HIP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__));      \
  } while (0)

// reference order of zeros / summation: single sets, paired sets, pacbio sets (prob_calculator.h:70-107)
std::vector<SetRef> scoring_order(const gaml_hip_ctx* c) {
  std::vector<SetRef> o;
  for (int k = 0; k < 3; k++)
    for (auto& h : c->handles) if (h.kind == k) o.push_back(h);
  return o;
}

void shard_range(const gaml_hip_ctx* c, int64_t n, int64_t* lo, int64_t* hi) {
  *lo = n * c->rank / c->world;
  *hi = n * (c->rank + 1) / c->world;
}

int grid_for(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  return (int)std::max<int64_t>(1, std::min<int64_t>(b, kMaxBlocks));
}

// take the next staging slot; waits only if the device is kRing evaluations behind
// GAML_HIP_BACKTRACE=1: a backtrace on stderr when the process aborts or faults inside the library (debugging aid)
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <exception>
namespace {
void gaml_hip_crash_handler(int sig) {
  void* frames[64];
  const int n = backtrace(frames, 64);
  const char msg[] = "libgaml_hip: fatal signal, backtrace:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
struct CrashHook {
  CrashHook() {
    if (!getenv("GAML_HIP_BACKTRACE")) return;
    signal(SIGABRT, gaml_hip_crash_handler);
    signal(SIGSEGV, gaml_hip_crash_handler);
    std::set_terminate([] {
      if (auto e = std::current_exception()) {
        try { std::rethrow_exception(e); } catch (const std::exception& x) { fprintf(stderr, "libgaml_hip: uncaught exception: %s\n", x.what()); } catch (...) { fprintf(stderr, "libgaml_hip: uncaught exception\n"); }
      } else fprintf(stderr, "libgaml_hip: std::terminate without an exception (a joinable std::thread destroyed or assigned?)\n");
      abort();
    });
  }
} crash_hook;
}  // namespace

int stage_acquire(gaml_hip_ctx* c, Staging& s, size_t bytes, void** host) {
  int k = s.next;
  s.next = (s.next + 1) % kRing;
  if (s.armed[k]) { HIP_TRY(c, hipEventSynchronize(s.done[k])); s.armed[k] = false; }
  if (!s.done[k]) HIP_TRY(c, hipEventCreateWithFlags(&s.done[k], hipEventDisableTiming));
  HIP_TRY(c, s.host[k].reserve(bytes));
  *host = s.host[k].p;
  return k;
}
int stage_release(gaml_hip_ctx* c, Staging& s, int k, hipStream_t st) {
  // a blocking call returns after the device is done with the slot: no event (a marker packet and ~1.5 us of host time)
  if (c->host_results) return 0;
  HIP_TRY(c, hipEventRecord(s.done[k], st));
  s.armed[k] = true;
  return 0;
}

// Staged host data -> device memory by a small kernel that reads the pinned slot directly. Per blocking step 3 us
// faster end to end than hipMemcpyAsync on the same stream (tools/xcd_start_probe.hip: 31.3 vs 34.2 us for copy +
// kernel + sync).
__global__ __launch_bounds__(kBlock) void stage_copy_kernel(const int4* __restrict__ src, int4* __restrict__ dst, int n16) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n16; i += gridDim.x * kBlock) dst[i] = src[i];
}
int stage_upload(gaml_hip_ctx* c, Staging& s, int k, void* dst, size_t bytes, hipStream_t st) {
  if (bytes == 0) return 0;
  if (c->knobs[8] == 1 || (bytes & 15) || bytes > ((size_t)1 << 30)) {
    HIP_TRY(c, hipMemcpyAsync(dst, s.host[k].p, bytes, hipMemcpyHostToDevice, st));
    return 0;
  }
  const int n16 = (int)(bytes / 16);
  hipLaunchKernelGGL(stage_copy_kernel, dim3((unsigned)std::min(64, (n16 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                     (const int4*)s.host[k].dev, (int4*)dst, n16);
  HIP_TRY(c, hipGetLastError());
  return 0;
}

size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// sum the elapsed time of every event pair recorded since the last collection (the stream
// they were recorded on must have been synchronised by the caller)
int collect_events(gaml_hip_ctx* c) {
  // t_kernel_us (gaml_hip_last_timing [2]) describes the LAST call: the scoring kernels of the newest evaluation that
  // carried events -- not the sum over everything collected lazily (gaml_hip_kernel_stats drains hundreds of launches)
  double sum = 0, last = 0;
  for (size_t i = 0; i < c->ev_used; i++) {
    float ms = 0;
    HIP_TRY(c, hipEventSynchronize(c->ev_pool[i].second));
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_pool[i].first, c->ev_pool[i].second));
    if (c->ev_kind[i] == 1) { c->stat_general_us += ms * 1000.0; c->stat_general_launches++; continue; }
    sum += ms * 1000.0;
    last = c->ev_call[i] == c->ev_call[c->ev_used - 1] ? last + ms * 1000.0 : 0.0;
  }
  if (c->ev_used) c->t_kernel_us = last;
  c->stat_device_us += sum;
  c->ev_used = 0;
  return 0;
}

int take_events(gaml_hip_ctx* c, std::pair<hipEvent_t, hipEvent_t>** out, int kind = 0) {
  if (c->ev_used == c->ev_pool.size() && c->ev_pool.size() >= 2048) { if (int e = collect_events(c)) return e; }
  if (c->ev_used == c->ev_pool.size()) {
    hipEvent_t a, b;
    HIP_TRY(c, hipEventCreate(&a));
    HIP_TRY(c, hipEventCreate(&b));
    c->ev_pool.emplace_back(a, b);
  }
  if (c->ev_call.size() < c->ev_pool.size()) c->ev_call.resize(c->ev_pool.size(), 0);
  if (c->ev_kind.size() < c->ev_pool.size()) c->ev_kind.resize(c->ev_pool.size(), 0);
  c->ev_kind[c->ev_used] = (uint8_t)kind;
  c->ev_call[c->ev_used] = c->eval_serial;
  *out = &c->ev_pool[c->ev_used++];
  return 0;
}

// layout of one OccTable inside an arena
struct OccLayout { size_t direct, multi_off, multi, end; };
// n_entries > the image's window count: the table is padded with "does not occur" entries (a batch: windows that later
// path sets of the same batch add must read as absent in the earlier sets' tables)
OccLayout layout_image(const OccImage& t, size_t at, size_t n_entries = 0) {  // `direct` holds the 12-byte entries here
  OccLayout l;
  l.direct = at;
  l.multi_off = (l.direct + std::max<size_t>(1, std::max(n_entries, t.occ12.size())) * sizeof(Occ12) + 15) & ~(size_t)15;
  l.multi = (l.multi_off + t.multi_off.size() * sizeof(int32_t) + 15) & ~(size_t)15;
  l.end = (l.multi + std::max<size_t>(1, t.multi.size()) * sizeof(OccQuad) + 15) & ~(size_t)15;
  return l;
}
void pack_image(const OccImage& t, const OccLayout& l, char* base) {
  if (!t.occ12.empty()) memcpy(base + l.direct, t.occ12.data(), t.occ12.size() * sizeof(Occ12));
  const size_t used = l.direct + t.occ12.size() * sizeof(Occ12);
  if (l.multi_off > used) memset(base + used, 0xff, l.multi_off - used);  // padding entries: all ones = the window does not occur
  memcpy(base + l.multi_off, t.multi_off.data(), t.multi_off.size() * sizeof(int32_t));
  if (!t.multi.empty()) memcpy(base + l.multi, t.multi.data(), t.multi.size() * sizeof(OccQuad));
}
OccLayout layout_occ(const OccTable& t, size_t at) {
  OccLayout l;
  l.direct = at;
  l.multi_off = align16(l.direct + std::max<size_t>(1, t.direct.size()) * sizeof(OccQuad));
  l.multi = align16(l.multi_off + t.multi_off.size() * sizeof(int32_t));
  l.end = align16(l.multi + std::max<size_t>(1, t.multi.size()) * sizeof(OccQuad));
  return l;
}
void pack_occ(const OccTable& t, const OccLayout& l, char* base) {
  if (!t.direct.empty()) memcpy(base + l.direct, t.direct.data(), t.direct.size() * sizeof(OccQuad));
  memcpy(base + l.multi_off, t.multi_off.data(), t.multi_off.size() * sizeof(int32_t));
  if (!t.multi.empty()) memcpy(base + l.multi, t.multi.data(), t.multi.size() * sizeof(OccQuad));
}

int upload_mate(gaml_hip_ctx* c, const ShortMate& m, ReadMajor& rm, MateDev& d, hipStream_t st,
                const std::vector<int32_t>* slot_of_read = nullptr, bool force = false) {
  if (d.pow_n == 0) {
    d.pow_n = m.match_pow.size();
    HIP_TRY(c, d.pows.reserve(2 * d.pow_n * sizeof(double)));
    HIP_TRY(c, hipMemcpy(d.pows.p, m.mismatch_pow.data(), d.pow_n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(d.pows.as<double>() + d.pow_n, m.match_pow.data(), d.pow_n * sizeof(double), hipMemcpyHostToDevice));
  }
  if (d.uploaded_generation == m.active_generation && !force) return 0;
  // the window cache changed: rebuild the read-major table (cold path) and upload it
  build_read_major(m, slot_of_read, rm);
  HIP_TRY(c, hipStreamSynchronize(st));  // earlier evaluations may still read the old table
  HIP_TRY(c, d.first.reserve(std::max<size_t>(1, rm.first.size()) * sizeof(RecQuad)));
  HIP_TRY(c, d.extra.reserve(std::max<size_t>(1, rm.extra.size()) * sizeof(RecQuad)));
  if (!rm.first.empty()) HIP_TRY(c, hipMemcpy(d.first.p, rm.first.data(), rm.first.size() * sizeof(RecQuad), hipMemcpyHostToDevice));
  if (!rm.extra.empty()) HIP_TRY(c, hipMemcpy(d.extra.p, rm.extra.data(), rm.extra.size() * sizeof(RecQuad), hipMemcpyHostToDevice));
  d.uploaded_generation = m.active_generation;
  return 0;
}

MateView view_of(const MateDev& d, const char* arena, const OccLayout& l) {
  MateView v;
  v.first = d.first.as<int4>();
  v.extra = d.extra.as<int4>();
  v.occ = (const int4*)(arena + l.direct);
  v.occ12 = nullptr;
  v.multi_off = (const int*)(arena + l.multi_off);
  v.multi = (const int4*)(arena + l.multi);
  v.mism_pow = d.pows.as<double>();
  v.match_pow = d.pows.as<double>() + d.pow_n;
  return v;
}

std::vector<Walk> unflatten(const int32_t* flat, const int64_t* offs, int32_t n) {
  std::vector<Walk> r(n);
  for (int32_t i = 0; i < n; i++) r[i].assign(flat + offs[i], flat + offs[i + 1]);
  return r;
}

}  // namespace

#include "paired_launch.hip.h"

namespace {

// ---------------------------------------------------------------------------------------
// single-end read set (CalcScoreForPaths graph.cc:1650-1743)
// ---------------------------------------------------------------------------------------
// host: per path (offset by 1,000,000 each, graph.cc:1685), per contig: register + occurrences
int32_t prepare_single_host(gaml_hip_ctx* c, SingleSet& s, const std::vector<Walk>& paths, std::vector<Occ>& occs) {
  int32_t rank = 0, tl = 0, stv = 0;
  std::vector<std::pair<int32_t, int32_t>> ranges;
  std::vector<int32_t> gaps;
  for (const Walk& path : paths) {
    split_contigs(path, ranges, gaps);
    for (size_t ci = 0; ci < ranges.size(); ci++) {
      if (ci > 0) tl += gaps[ci - 1];
      const int32_t* ctg = path.data() + ranges[ci].first;
      const int32_t n = ranges[ci].second - ranges[ci].first;
      register_for_contig(c->g, s.mate, ctg, n);
      occurrences_single_contig(c->g, s.mate, ctg, n, stv + tl, &rank, occs);
      for (int32_t k = 0; k < n; k++) tl += c->g.len(ctg[k]);
    }
    stv += 1000000;
  }
  s.last_occ = occs;
  return tl;
}

int launch_single(gaml_hip_ctx* c, SingleSet& s, const std::vector<Walk>& paths, int32_t total_len, hipStream_t st, double* out4) {
  if (!s.tabs_uploaded) {
    const int lmax = s.mate.max_len;
    s.floor_tab.resize(lmax + 1); s.logfloor_tab.resize(lmax + 1);
    for (int v = 0; v <= lmax; v++) {
      s.floor_tab[v] = std::exp(s.cfg.min_prob_start + s.cfg.min_prob_per_base * v);  // graph.cc:1528
      s.logfloor_tab[v] = std::log(s.floor_tab[v]);
    }
    HIP_TRY(c, s.tabs.reserve(2 * (size_t)(lmax + 1) * sizeof(double)));
    HIP_TRY(c, hipMemcpy(s.tabs.p, s.floor_tab.data(), (lmax + 1) * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(s.tabs.as<double>() + lmax + 1, s.logfloor_tab.data(), (lmax + 1) * sizeof(double), hipMemcpyHostToDevice));
    const int64_t n = s.mate.n_local();
    HIP_TRY(c, s.lens.reserve(std::max<size_t>(1, n) * sizeof(int32_t)));
    if (n) HIP_TRY(c, hipMemcpy(s.lens.p, s.mate.lens.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(c, s.probs.reserve(std::max<size_t>(1, n) * sizeof(double)));
    HIP_TRY(c, s.red.init());
    s.tabs_uploaded = true;
  }
  std::vector<Occ> occs;
  int32_t tl = prepare_single_host(c, s, paths, occs);
  (void)total_len;
  OccTable occ;
  build_occ_table(s.mate.wins.size(), occs, occ);
  const double t_after_host = now_us();
  if (int e = upload_mate(c, s.mate, s.rm, s.dev, st)) return e;
  OccLayout l0 = layout_occ(occ, 0);
  void* host = nullptr;
  int slot = stage_acquire(c, s.stage, l0.end, &host);
  if (slot < 0) return slot;
  pack_occ(occ, l0, (char*)host);
  if (l0.end > s.occ_arena.cap) { HIP_TRY(c, hipStreamSynchronize(st)); HIP_TRY(c, s.occ_arena.reserve(l0.end)); }
  if (int e = stage_upload(c, s.stage, slot, s.occ_arena.p, l0.end, st)) return e;
  if (int e = stage_release(c, s.stage, slot, st)) return e;
  const int64_t n = s.mate.n_local();
  SingleArgs a;
  a.m = view_of(s.dev, (const char*)s.occ_arena.p, l0);
  a.lens = s.lens.as<int>();
  a.floor_tab = s.tabs.as<double>();
  a.logfloor_tab = s.tabs.as<double>() + s.mate.max_len + 1;
  int t2 = tl == 0 ? 1 : tl;
  a.two_T = (double)(2 * t2);
  a.n = (int)n;
  a.probs = s.probs.as<double>();
  a.part_sum = s.red.part_sum.as<double>(); a.part_zero = s.red.part_zero.as<int>();
  a.ticket = s.red.ticket.as<unsigned>(); a.out = out4;
  a.n_reads = (double)n;
  if (n > 0) {
    hipLaunchKernelGGL(single_score_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, a);
    HIP_TRY(c, hipGetLastError());
  } else {
    HIP_TRY(c, hipMemsetAsync(out4, 0, 4 * sizeof(double), st));
  }
  c->stat_algo_bytes += 16.0 * (double)s.rm.total_records + 12.0 * (double)n;  // 16k + 4 + 8 per read
  c->stat_launches++;
  c->t_host_us += t_after_host;
  return 0;
}

// ---------------------------------------------------------------------------------------
// coverage sweep of a PacBio set on the device (graph.cc:3198-3250; pacbio_sweep.hip.h)
// ---------------------------------------------------------------------------------------
// the walk-major interval arrays follow the record cache
int pacbio_sweep_sync(gaml_hip_ctx* c, PacbioSet& s, hipStream_t st) {
  PbSweepDev& d = s.sweep;
  if (d.generation == s.generation) return 0;
  std::vector<int32_t>& off = d.iv_off_host;
  off.assign(s.recs.size() + 1, 0);
  std::vector<int32_t> iv;
  for (size_t w = 0; w < s.recs.size(); w++) {
    for (const auto& r : s.recs[w]) {
      // only records that clear GetMinReadProb (graph.h:478-481) count (graph.cc:3216)
      const double min_lp = s.log_mismatch * (s.lens[r.read_id] * 0.25) + s.log_match * (s.lens[r.read_id] * 0.75);
      if (r.logprob < min_lp) continue;
      iv.push_back(r.position); iv.push_back(r.position_end);
    }
    off[w + 1] = (int32_t)(iv.size() / 2);
  }
  HIP_TRY(c, hipStreamSynchronize(st));  // an earlier evaluation may still read the old arrays
  HIP_TRY(c, d.iv_off.reserve(off.size() * sizeof(int32_t)));
  HIP_TRY(c, d.iv.reserve(std::max<size_t>(1, iv.size()) * sizeof(int32_t)));
  HIP_TRY(c, hipMemcpy(d.iv_off.p, off.data(), off.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  if (!iv.empty()) HIP_TRY(c, hipMemcpy(d.iv.p, iv.data(), iv.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  d.generation = s.generation;
  return 0;
}

// this evaluation's intervals into PbSweepDev::all: the node intervals (`node`, computed on the host: every rank has
// them), then the alignment intervals of this rank's records, expanded on the device from the occurrence list
int pacbio_sweep_prepare(gaml_hip_ctx* c, PacbioSet& s, const std::vector<int32_t>& tl, const std::vector<int32_t>& node /* 4 per interval */,
                         std::vector<PbOcc>& occ, hipStream_t st, int64_t* n_own_out) {
  PbSweepDev& d = s.sweep;
  if (int e = pacbio_sweep_sync(c, s, st)) return e;
  int64_t n_own = 0;
  for (PbOcc& o : occ) { o.out = (int32_t)n_own; n_own += d.iv_off_host[o.walk + 1] - d.iv_off_host[o.walk]; }
  const int64_t n_node = (int64_t)node.size() / 4;
  if (n_node + n_own > (int64_t)1 << 28) return fail(c, GAML_HIP_EINVAL, "too many alignment intervals in one evaluation");
  for (PbOcc& o : occ) o.out += (int32_t)n_node;
  const size_t tl_bytes = align16(tl.size() * sizeof(int32_t)), node_bytes = node.size() * sizeof(int32_t), occ_bytes = occ.size() * sizeof(PbOcc);
  const size_t bytes = std::max<size_t>(16, tl_bytes + node_bytes + occ_bytes);
  void* host = nullptr;
  int slot = stage_acquire(c, d.stage, bytes, &host);
  if (slot < 0) return slot;
  memcpy(host, tl.data(), tl.size() * sizeof(int32_t));
  if (node_bytes) memcpy((char*)host + tl_bytes, node.data(), node_bytes);
  if (occ_bytes) memcpy((char*)host + tl_bytes + node_bytes, occ.data(), occ_bytes);
  const size_t all_bytes = (size_t)std::max<int64_t>(1, n_node + n_own) * sizeof(int4);
  if (bytes > d.in.cap || all_bytes > d.all.cap) {
    HIP_TRY(c, hipStreamSynchronize(st));
    HIP_TRY(c, d.in.reserve(bytes));
    HIP_TRY(c, d.all.reserve(all_bytes));
  }
  if (int e = stage_upload(c, d.stage, slot, d.in.p, bytes, st)) return e;
  if (int e = stage_release(c, d.stage, slot, st)) return e;
  if (node_bytes) HIP_TRY(c, hipMemcpyAsync(d.all.p, (const char*)d.in.p + tl_bytes, node_bytes, hipMemcpyDeviceToDevice, st));
  if (!occ.empty() && n_own > 0) {
    const unsigned grid = (unsigned)std::min<size_t>((occ.size() + 3) / 4, 1024);
    hipLaunchKernelGGL(pacbio_intervals_kernel, dim3(grid), dim3(256), 0, st, (const PbOcc*)((const char*)d.in.p + tl_bytes + node_bytes), (int)occ.size(),
                       d.iv_off.as<int>(), d.iv.as<int2>(), d.all.as<int4>());
    HIP_TRY(c, hipGetLastError());
  }
  *n_own_out = n_own;
  return 0;
}

// sort + running maximum + sweep over the n intervals in PbSweepDev::all; bad_bases (times `scale`) into out4[2]
int pacbio_sweep_run(gaml_hip_ctx* c, PacbioSet& s, int64_t n, int32_t n_paths, hipStream_t st, double* out4, double scale) {
  PbSweepDev& d = s.sweep;
  typedef unsigned long long u64;
  const size_t n1 = (size_t)std::max<int64_t>(1, n);
  int path_bits = 1;
  while ((1 << path_bits) < n_paths && path_bits < 30) path_bits++;
  const int end_bit = 32 + path_bits;
  size_t tmp1 = 0, tmp2 = 0, tmp3 = 0;
  HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp1, (const u64*)nullptr, (u64*)nullptr, (const u64*)nullptr, (u64*)nullptr, (int)n1, 0, end_bit, st));
  HIP_TRY(c, hipcub::DeviceRadixSort::SortKeys(nullptr, tmp2, (const u64*)nullptr, (u64*)nullptr, (int)(2 * n1), 0, end_bit, st));
  HIP_TRY(c, hipcub::DeviceScan::InclusiveScan(nullptr, tmp3, (const u64*)nullptr, (u64*)nullptr, hipcub::Max(), (int)n1, st));
  const size_t tmp_bytes = std::max(std::max(tmp1, tmp2), std::max<size_t>(tmp3, 16));
  if (n1 * sizeof(u64) > d.key_begin.cap || tmp_bytes > d.tmp.cap || !d.bad.p) {
    HIP_TRY(c, hipStreamSynchronize(st));
    for (DevBuf* b : {&d.key_begin, &d.key_end, &d.key_begin_s, &d.key_end_s, &d.end_max}) HIP_TRY(c, b->reserve(n1 * sizeof(u64)));
    for (DevBuf* b : {&d.pos, &d.pos_s}) HIP_TRY(c, b->reserve(2 * n1 * sizeof(u64)));
    HIP_TRY(c, d.tmp.reserve(tmp_bytes));
    HIP_TRY(c, d.bad.reserve(sizeof(u64)));
  }
  HIP_TRY(c, hipMemsetAsync(d.bad.p, 0, sizeof(u64), st));
  if (n > 0) {
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(pacbio_sweep_keys_kernel, dim3(grid), dim3(256), 0, st, d.all.as<int4>(), (int)n, d.key_begin.as<u64>(), d.key_end.as<u64>(), d.pos.as<u64>());
    HIP_TRY(c, hipGetLastError());
    size_t t = d.tmp.cap;
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(d.tmp.p, t, d.key_begin.as<u64>(), d.key_begin_s.as<u64>(), d.key_end.as<u64>(), d.key_end_s.as<u64>(), (int)n, 0, end_bit, st));
    t = d.tmp.cap;
    HIP_TRY(c, hipcub::DeviceRadixSort::SortKeys(d.tmp.p, t, d.pos.as<u64>(), d.pos_s.as<u64>(), (int)(2 * n), 0, end_bit, st));
    t = d.tmp.cap;
    HIP_TRY(c, hipcub::DeviceScan::InclusiveScan(d.tmp.p, t, d.key_end_s.as<u64>(), d.end_max.as<u64>(), hipcub::Max(), (int)n, st));
    const unsigned grid2 = (unsigned)std::min<int64_t>((2 * n + 255) / 256, 1024);
    hipLaunchKernelGGL(pacbio_sweep_kernel, dim3(grid2), dim3(256), 0, st, d.pos_s.as<u64>(), (int)(2 * n), d.key_begin_s.as<u64>(), d.end_max.as<u64>(), (int)n,
                       d.in.as<int>(), s.cfg.step, d.bad.as<u64>());
    HIP_TRY(c, hipGetLastError());
  }
  hipLaunchKernelGGL(store_bad_bases_kernel, dim3(1), dim3(64), 0, st, d.bad.as<u64>(), out4, scale);
  HIP_TRY(c, hipGetLastError());
  return 0;
}

// PacBio read set (CalcScoreForPacbio graph.cc:3171-3261)
// ---------------------------------------------------------------------------------------
int launch_pacbio(gaml_hip_ctx* c, PacbioSet& s, const std::vector<Walk>& paths_in, hipStream_t st, double* out4) {
  const int64_t n = s.hi - s.lo;
  if (!s.red.part_sum.p) {
    HIP_TRY(c, s.red.init());
    HIP_TRY(c, s.d_lens.reserve(std::max<size_t>(1, n) * sizeof(int32_t)));
    if (n) HIP_TRY(c, hipMemcpy(s.d_lens.p, s.lens.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(c, s.logprobs.reserve(std::max<size_t>(1, n) * sizeof(double)));
  }
  // sub-walk occurrence counts + coverage events, per path (graph.cc:3183-3251)
  std::vector<int32_t> count(s.recs.size(), 0);
  const bool cov = s.cfg.penalty_constant > 0;
  const bool defer = cov && c->defer_cov;  // sharded: the sweep waits for the other ranks' intervals
  std::vector<int32_t> sweep_tl, sweep_node;  // contig lengths; node intervals {contig, begin, end, 0}
  std::vector<PbOcc> sweep_occ;
  int32_t path_no = -1;
  for (Walk path : paths_in) {
    path_no++;
    for (auto& x : path) if (x >= 0) x = c->g.norm[x];  // NormalizePath graph.h:268-273
    const int32_t m = (int32_t)path.size();
    std::vector<int32_t> begins(m), ends(m);
    int32_t len = 0;
    for (int32_t i = 0; i < m; i++) {  // graph.cc:2412-2431 (a leading gap contributes its length)
      begins[i] = len;
      len += path[i] < 0 ? -path[i] : c->g.len(path[i]);
      ends[i] = len;
    }
    if (cov) {
      sweep_tl.push_back(len);
      auto interval = [&](int32_t b, int32_t e) { sweep_node.push_back(path_no); sweep_node.push_back(b); sweep_node.push_back(e); sweep_node.push_back(0); };
      interval(-1000, 2000);  // the events (-1000, 1), (2000, -3000) of graph.cc:3198-3199
      int32_t pp = 0;
      for (int32_t e : path) {
        if (e >= 0) { const int32_t cl = c->g.len(e); if (cl > 0) interval(pp, pp + cl); pp += cl; }  // graph.cc:3202-3210
        else pp += -e;
      }
    }
    Walk sub;
    for (int32_t i = 0; i < m; i++) {  // graph.cc:2438-2454
      sub.clear();
      for (int32_t j = i; j < m; j++) {
        sub.push_back(path[j]);
        auto it = s.walk_id.find(sub);
        if (it == s.walk_id.end()) s.misses++;  // the reference would run BLASR here (out of scope)
        else {
          count[it->second]++;
          if (cov) sweep_occ.push_back(PbOcc{it->second, begins[i], path_no, 0});  // its records' intervals (graph.cc:3214-3222)
        }
        if ((ends[j] - begins[i]) - (ends[i] - begins[i]) > s.max_len) break;
      }
    }
  }
  const double t_after_host = now_us();
  // read-major CSR of the cached records (rebuilt when the cache changed)
  if (s.uploaded_generation != s.generation) {
    std::vector<int32_t> off(n + 1, 0);
    for (auto& v : s.recs) for (auto& r : v) off[r.read_id + 1]++;
    for (int64_t i = 0; i < n; i++) off[i + 1] += off[i];
    std::vector<int32_t> walk(off[n]), fill(off.begin(), off.end() - 1);
    std::vector<double> lp(off[n]);
    for (size_t w = 0; w < s.recs.size(); w++)
      for (auto& r : s.recs[w]) { int32_t at = fill[r.read_id]++; walk[at] = (int32_t)w; lp[at] = r.logprob; }
    HIP_TRY(c, hipStreamSynchronize(st));
    HIP_TRY(c, s.rec_off.reserve((n + 1) * sizeof(int32_t)));
    HIP_TRY(c, s.rec_walk.reserve(std::max<size_t>(1, walk.size()) * sizeof(int32_t)));
    HIP_TRY(c, s.rec_logp.reserve(std::max<size_t>(1, lp.size()) * sizeof(double)));
    HIP_TRY(c, hipMemcpy(s.rec_off.p, off.data(), (n + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!walk.empty()) {
      HIP_TRY(c, hipMemcpy(s.rec_walk.p, walk.data(), walk.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      HIP_TRY(c, hipMemcpy(s.rec_logp.p, lp.data(), lp.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    s.uploaded_generation = s.generation;
  }
  size_t bytes = align16(std::max<size_t>(1, count.size()) * sizeof(int32_t));  // whole 16-byte units: the copy kernel moves int4s
  void* host = nullptr;
  int slot = stage_acquire(c, s.stage, bytes, &host);
  if (slot < 0) return slot;
  if (!count.empty()) memcpy(host, count.data(), count.size() * sizeof(int32_t));
  if (bytes > s.walk_count.cap) { HIP_TRY(c, hipStreamSynchronize(st)); HIP_TRY(c, s.walk_count.reserve(bytes)); }
  if (int e = stage_upload(c, s.stage, slot, s.walk_count.p, bytes, st)) return e;
  if (int e = stage_release(c, s.stage, slot, st)) return e;
  PacbioArgs a;
  a.rec_off = s.rec_off.as<int>(); a.rec_walk = s.rec_walk.as<int>(); a.rec_logp = s.rec_logp.as<double>();
  a.walk_count = s.walk_count.as<int>(); a.lens = s.d_lens.as<int>();
  a.floor_a = std::log(std::exp(s.cfg.min_prob_start));     // logdouble(exp(c)) graph.cc:3075
  a.floor_b = std::log(std::exp(s.cfg.min_prob_per_base));  // logdouble(exp(k))
  a.n = (int)n;
  a.logprobs = s.logprobs.as<double>();
  a.part_sum = s.red.part_sum.as<double>(); a.part_zero = s.red.part_zero.as<int>();
  a.ticket = s.red.ticket.as<unsigned>(); a.out = out4;
  a.n_reads = (double)n; a.bad_bases = 0.0;  // (with a penalty: the sweep below stores it)
  if (n > 0) {
    int64_t threads = n * 64;  // one wave per read
    hipLaunchKernelGGL(pacbio_score_kernel, dim3(grid_for(threads)), dim3(kBlock), 0, st, a);
    HIP_TRY(c, hipGetLastError());
  } else {
    HIP_TRY(c, hipMemsetAsync(out4, 0, 4 * sizeof(double), st));
  }
  if (cov) {
    int64_t n_own = 0;
    if (int e = pacbio_sweep_prepare(c, s, sweep_tl, sweep_node, sweep_occ, st, &n_own)) return e;
    const int64_t n_node = (int64_t)sweep_node.size() / 4;
    if (defer) {
      gaml_hip_ctx::PendingPacbio pend;
      pend.out4 = out4; pend.n_paths = (int32_t)sweep_tl.size(); pend.n_node = n_node; pend.n_own = n_own; pend.pacbio_idx = -1;
      for (size_t i = 0; i < c->pacbios.size(); i++) if (c->pacbios[i].get() == &s) pend.pacbio_idx = (int)i;
      c->pending_pb.push_back(pend);
    } else {
      if (int e = pacbio_sweep_run(c, s, n_node + n_own, (int32_t)sweep_tl.size(), st, out4, 1.0)) return e;
    }
  }
  int64_t nrec = 0;
  for (auto& v : s.recs) nrec += (int64_t)v.size();
  c->stat_algo_bytes += 24.0 * (double)nrec + 12.0 * (double)n;
  c->stat_launches++;
  c->t_host_us += t_after_host;
  return 0;
}

// ---------------------------------------------------------------------------------------
// GPU window alignment of every pending window of one mate (cold path). Falls back to the host
// aligner for inputs the kernels do not cover (reads shorter than 16 or longer than 254 bases,
// mixed read lengths are fine). Records are identical to the host aligner's.
// ---------------------------------------------------------------------------------------
// Small batches -- what an annealing move brings: a handful of new junction windows, a few thousand seed candidates.
// Everything on the library's stream, ONE wait: the window strings and descriptors are written by the host straight
// into device memory (large BAR) or copied asynchronously from pinned memory; the three kernels run back to back (the
// extension kernel strides over a candidate count it reads on the device); a last small kernel leaves counters and
// hits in mapped pinned memory and publishes a sequence word the host polls. Returns 1 when the batch does not fit
// the fixed capacities (the caller takes the general route), 0 with `hits` filled, < 0 on error.
constexpr unsigned kFastSpans = 1u << 16, kFastCands = 1u << 17;

// device copies the aligner kernels need: the reads (1 byte per base) and the max-hash index, once per mate
int aln_upload_index(gaml_hip_ctx* c, const ShortMate& m, AlignDev& d) {
  if (d.uploaded) return 0;
  auto up = [&](DevBuf& b, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = b.reserve(std::max<size_t>(16, bytes));
    return (e != hipSuccess || bytes == 0) ? e : hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
  };
  HIP_TRY(c, up(d.reads, m.bases.data(), m.bases.size()));
  HIP_TRY(c, up(d.read_off, m.roff.data(), m.roff.size() * sizeof(int64_t)));
  HIP_TRY(c, up(d.bucket_hash, m.bucket_hash.data(), m.bucket_hash.size() * sizeof(uint64_t)));
  {
    std::vector<int32_t> top(65537, (int32_t)m.bucket_hash.size());
    size_t k = 0;
    for (uint32_t h = 0; h < 65536; h++) {
      while (k < m.bucket_hash.size() && (m.bucket_hash[k] >> 16) < h) k++;
      top[h] = (int32_t)k;
    }
    HIP_TRY(c, up(d.bucket_top, top.data(), top.size() * sizeof(int32_t)));
  }
  HIP_TRY(c, up(d.bucket_off, m.bucket_off.data(), m.bucket_off.size() * sizeof(int32_t)));
  HIP_TRY(c, up(d.bucket_reads, m.bucket_reads.data(), m.bucket_reads.size() * sizeof(int32_t)));
  d.uploaded = true;
  return 0;
}

// the small-batch pipeline's fixed-capacity buffers, allocated together with the index upload (the cold first
// evaluation), not in the annealing call that first brings a small batch
int aln_small_reserve(gaml_hip_ctx* c, AlignSmall& S) {
  if (!S.counters.p) { HIP_TRY(c, S.counters.reserve(256)); HIP_TRY(c, hipMemset(S.counters.p, 0, 256)); }  // (publish_hits_kernel leaves them at zero)
  HIP_TRY(c, S.spans.reserve((size_t)kFastSpans * sizeof(AlnSpan)));
  HIP_TRY(c, S.cands.reserve((size_t)kFastCands * sizeof(AlnCand)));
  HIP_TRY(c, S.hits.reserve((size_t)kFastCands * sizeof(AlnHit)));
  if (!S.out_host.p) { HIP_TRY(c, S.out_host.reserve(64 + 64 + (size_t)kFastCands * sizeof(AlnHit))); memset(S.out_host.p, 0, 128); }
  if (!S.in_dev) {
    const bool direct = c->direct_write && c->knobs[8] == 0;
    const size_t want = (size_t)1 << 18;
    if (direct) HIP_TRY(c, hipExtMallocWithFlags(&S.in_dev, want, hipDeviceMallocFinegrained));
    else { HIP_TRY(c, hipMalloc(&S.in_dev, want)); HIP_TRY(c, S.in_host.reserve(want)); }
    S.in_cap = want; S.in_direct = direct;
  }
  return 0;
}

bool aln_gpu_capable(const gaml_hip_ctx* c, const ShortMate& m) {
  return !(c->device < 0 || c->knobs[5] == 1 || m.index_read_len < 16 || m.max_len > kAlnMaxRead || m.n_local() == 0 || m.bucket_hash.empty());
}

// window strings (graph.cc:846-857) of the pending windows, concatenated
void aln_prepare(const gaml_hip_ctx* c, const ShortMate& m, AlnJob& job) {
  const int nw = (int)m.pending.size();
  job.wstr.clear();
  job.wins.resize((size_t)nw);
  job.blk.assign((size_t)nw + 1, 0);
  for (int k = 0; k < nw; k++) {
    int32_t off = 0;
    std::string ws = m.window_string(c->g, *m.win_walk[m.pending[k]], &off);
    job.wins[k] = AlnWindow{(int32_t)job.wstr.size(), (int32_t)ws.size(), off};
    job.blk[(size_t)k + 1] = job.blk[(size_t)k] + 2 * aln_span_chunks((int)ws.size(), m.index_read_len);
    job.wstr += ws;
  }
  job.prepared = true;
  job.enqueued = false;
}

// enqueue the whole small-batch pipeline on the library's stream; 1: the batch does not fit the fixed capacities
int aln_small_enqueue(gaml_hip_ctx* c, const ShortMate& m, AlignDev& d, AlignSmall& S, AlnJob& job, hipStream_t st) {
  job.stream = st;
  const int nw = (int)job.wins.size();
  const size_t in_bytes = align16(nw * sizeof(AlnWindow)) + align16((nw + 1) * sizeof(int32_t)) + align16(job.wstr.size() + 16);
  if (nw == 0 || nw > 4096 || in_bytes > ((size_t)1 << 20) || job.blk[(size_t)nw] == 0) return 1;
  // input block: [windows][code-buffer offsets][window strings]
  const bool direct = c->direct_write && c->knobs[8] == 0;
  if (in_bytes > S.in_cap || S.in_direct != direct) {
    HIP_TRY(c, hipStreamSynchronize(st));
    if (S.in_dev) { HIP_TRY(c, hipFree(S.in_dev)); S.in_dev = nullptr; }
    const size_t want = std::max<size_t>(in_bytes * 2, (size_t)1 << 16);
    if (direct) HIP_TRY(c, hipExtMallocWithFlags(&S.in_dev, want, hipDeviceMallocFinegrained));
    else HIP_TRY(c, hipMalloc(&S.in_dev, want));
    S.in_cap = want; S.in_direct = direct;
  }
  char* wp = (char*)S.in_dev;
  if (!direct) { HIP_TRY(c, S.in_host.reserve(in_bytes)); wp = (char*)S.in_host.p; }
  const size_t off_blk = align16(nw * sizeof(AlnWindow)), off_str = off_blk + align16((nw + 1) * sizeof(int32_t));
  memcpy(wp, job.wins.data(), nw * sizeof(AlnWindow));
  memcpy(wp + off_blk, job.blk.data(), (nw + 1) * sizeof(int32_t));
  memcpy(wp + off_str, job.wstr.data(), job.wstr.size());
  if (direct) _mm_sfence();
  else HIP_TRY(c, hipMemcpyAsync(S.in_dev, S.in_host.p, in_bytes, hipMemcpyHostToDevice, st));
  const char* dbase = (const char*)S.in_dev;
  if (int e = aln_small_reserve(c, S)) return e;
  const AlnWindow* d_wins = (const AlnWindow*)dbase;
  const int* d_blk = (const int*)(dbase + off_blk);
  const char* d_wstr = dbase + off_str;
  hipLaunchKernelGGL(span_maxima_kernel<false>, dim3((unsigned)job.blk[(size_t)nw]), dim3(kAlnBlock), 0, st, d_wstr, d_wins, nw, m.index_read_len, d_blk,
                     S.spans.as<AlnSpan>(), S.counters.as<unsigned>(), kFastSpans, INT_MAX, 0, AlnMates{}, (AlnCand*)nullptr, (unsigned*)nullptr, 0u);
  hipLaunchKernelGGL(candidates_kernel, dim3(64), dim3(kAlnBlock), 0, st, S.spans.as<AlnSpan>(), S.counters.as<unsigned>(), kFastSpans,
                     d.bucket_hash.as<uint64_t>(), d.bucket_top.as<int32_t>(), d.bucket_off.as<int32_t>(), d.bucket_reads.as<int32_t>(), (int)m.bucket_hash.size(), S.cands.as<AlnCand>(),
                     S.counters.as<unsigned>() + 1, kFastCands);
  hipLaunchKernelGGL(extend_kernel, dim3(1024), dim3(64 * kAlnWaves), 0, st, S.cands.as<AlnCand>(), S.counters.as<unsigned>() + 1, kFastCands, d_wstr,
                     d_wins, d.reads.as<char>(), d.read_off.as<int64_t>(), S.hits.as<AlnHit>());
  job.seq = ++S.out_seq;
  char* oh = (char*)S.out_host.dev;
  hipLaunchKernelGGL(publish_hits_kernel, dim3(1), dim3(256), 0, st, S.counters.as<unsigned>(), S.hits.as<AlnHit>(), kFastCands, (unsigned*)(oh + 64),
                     (AlnHit*)(oh + 128), kFastCands, (volatile unsigned long long*)oh, job.seq);
  HIP_TRY(c, hipGetLastError());
  job.enqueued = true;
  return 0;
}

// the one wait of a small batch: poll the sequence word (bounded), then the runtime's wait. 1: capacities exceeded
int aln_small_collect(gaml_hip_ctx* c, AlignSmall& S, AlnJob& job, std::vector<AlnHit>& hits, unsigned* n_cands_out) {
  volatile unsigned long long* word = (volatile unsigned long long*)S.out_host.p;
  const double t0 = now_us();
  bool seen = false;
  while (!seen && now_us() - t0 < 5000.0) { for (int k = 0; k < 256 && !seen; k++) { seen = *word == job.seq; __builtin_ia32_pause(); } }
  if (!seen) { HIP_TRY(c, hipStreamSynchronize((hipStream_t)job.stream)); if (*word != job.seq) return fail(c, GAML_HIP_ESTATE, "aligner: the publish kernel finished without its sequence word"); }
  std::atomic_thread_fence(std::memory_order_acquire);
  job.enqueued = false;
  const unsigned* counts = (const unsigned*)((const char*)S.out_host.p + 64);
  if (counts[0] > kFastSpans || counts[1] > kFastCands) return 1;  // did not fit: the general route redoes the batch
  const unsigned nc = counts[1];
  hits.resize(nc);
  if (nc) memcpy(hits.data(), (const char*)S.out_host.p + 128, (size_t)nc * sizeof(AlnHit));
  *n_cands_out = nc;
  return 0;
}

// The hits of a batch over the pending windows of `m` (window numbers = positions in m.pending) become the windows'
// records: per window sorted by (position, read), the first alignment found for a key survives
// (graph.cc:841, 891, 895-897); per read its candidates are visited forward-strand spans first.
void aln_file_hits(ShortMate& m, int nw, std::vector<AlnHit>& hits, bool device_sorted) {
  // Hits are bucketed by window first (counting sort), then every window is sorted on its own --
  // large batches on a few host threads.
  std::vector<int64_t> wstart(nw + 1, 0);
  for (const AlnHit& h : hits) if (h.edit >= 0) wstart[h.win + 1]++;
  for (int k = 0; k < nw; k++) wstart[k + 1] += wstart[k];
  std::vector<AlnHit> ok;
  if (device_sorted) {
    ok.swap(hits);  // already (window, position, read, strand, order), successful extensions only
  } else {
    ok.resize((size_t)wstart[nw]);
    {
      std::vector<int64_t> fill(wstart.begin(), wstart.end() - 1);
      for (const AlnHit& h : hits) if (h.edit >= 0) ok[(size_t)fill[h.win]++] = h;
    }
    auto sort_range = [&](int k0, int k1) {
      for (int k = k0; k < k1; k++)
        std::sort(ok.begin() + wstart[k], ok.begin() + wstart[k + 1], [](const AlnHit& a, const AlnHit& b) {
          if (a.pos != b.pos) return a.pos < b.pos;
          if (a.read != b.read) return a.read < b.read;
          if (a.strand != b.strand) return a.strand < b.strand;
          return a.order < b.order;
        });
    };
    const int n_threads = ok.size() > (size_t)200000 ? (int)std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency())) : 1;
    if (n_threads > 1) {
      // contiguous window ranges of about equal hit counts
      std::vector<std::thread> pool;
      int k0 = 0;
      for (int t = 0; t < n_threads; t++) {
        const int64_t target = wstart[nw] * (t + 1) / n_threads;
        int k1 = k0;
        while (k1 < nw && wstart[k1 + 1] <= target) k1++;
        if (t == n_threads - 1) k1 = nw;
        pool.emplace_back(sort_range, k0, k1);
        k0 = k1;
      }
      for (auto& th : pool) th.join();
    } else {
      sort_range(0, nw);
    }
  }
  // One growth step for the whole batch. Growing copies the pool (45 MB at cfg3: ~5 ms per mate), so the cold batch
  // leaves room for twice its size -- untouched pages cost nothing -- and later growth is geometric.
  if (m.pool.capacity() < m.pool.size() + ok.size())
    m.pool.reserve(std::max(2 * (m.pool.size() + ok.size()), m.pool.capacity() + m.pool.capacity() / 2));
  if (ok.size() >= (size_t)200000) {
    // large batch (the cold first evaluation files ~2.8 M records): count the surviving records per window, then fill the
    // pool segment and the window headers on a few threads (windows are independent; same result as the loop below)
    const int n_threads = (int)std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<int> cut(n_threads + 1, nw);
    cut[0] = 0;
    for (int t = 1, k = 0; t < n_threads; t++) {  // contiguous window ranges of about equal hit counts
      const int64_t target = wstart[nw] * t / n_threads;
      while (k < nw && wstart[k + 1] <= target) k++;
      cut[t] = k;
    }
    std::vector<int64_t> ustart(nw + 1, 0);
    auto survives = [&](int k, int64_t at) { return at == wstart[k] || ok[at].pos != ok[at - 1].pos || ok[at].read != ok[at - 1].read; };
    auto run = [&](auto fn) {
      std::vector<std::thread> pool;
      for (int t = 0; t < n_threads; t++) pool.emplace_back(fn, cut[t], cut[t + 1]);
      for (auto& th : pool) th.join();
    };
    run([&](int k0, int k1) {
      for (int k = k0; k < k1; k++) { int64_t u = 0; for (int64_t at = wstart[k]; at < wstart[k + 1]; at++) u += survives(k, at); ustart[k + 1] = u; }
    });
    for (int k = 0; k < nw; k++) ustart[k + 1] += ustart[k];
    const size_t base = m.pool.size();
    m.pool.resize(base + (size_t)ustart[nw]);
    run([&](int k0, int k1) {
      for (int k = k0; k < k1; k++) {
        Window& win = m.wins[m.pending[k]];  // as ShortMate::finalize_window
        gaml_aligment* dst = m.pool.data() + base + ustart[k];
        int32_t max_pos = INT_MIN;
        for (int64_t at = wstart[k]; at < wstart[k + 1]; at++)
          if (survives(k, at)) { *dst++ = gaml_aligment{ok[at].pos, ok[at].edit, ok[at].read, ok[at].strand}; max_pos = std::max(max_pos, ok[at].pos); }
        win.first = (int64_t)(base + ustart[k]);
        win.count = (int32_t)(ustart[k + 1] - ustart[k]);
        win.max_pos = max_pos;
        win.global_max_pos = max_pos;
        win.pending = false;
      }
    });
  } else {
    std::vector<gaml_aligment> recs;
    for (int k = 0; k < nw; k++) {
      recs.clear();
      for (int64_t at = wstart[k]; at < wstart[k + 1]; at++)
        if (at == wstart[k] || ok[at].pos != ok[at - 1].pos || ok[at].read != ok[at - 1].read)
          recs.push_back(gaml_aligment{ok[at].pos, ok[at].edit, ok[at].read, ok[at].strand});
      m.finalize_window(m.pending[k], recs);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Small batches of BOTH mates of a paired set in one pipeline: the junction windows an annealing move brings are
// the same strings for the two mates, looked up in each mate's index. One input block, one chain of launches, one
// wait (two pipelines side by side cost two of each on the host). 1: not this way (capacities, one mate not on the
// device): the per-mate route takes over.
// ---------------------------------------------------------------------------------------------------------
int aln_pair_small(gaml_hip_ctx* c, PairedSet& ps) {
  ShortMate* mm[2] = {&ps.mate[0], &ps.mate[1]};
  if (c->knobs[5] == 3 || c->knobs[5] == 4) return 1;  // knob 5 = 3: general route, 4: one small pipeline per mate (A/B, tests)
  for (int mt = 0; mt < 2; mt++) if (mm[mt]->pending.empty() || !aln_gpu_capable(c, *mm[mt])) return 1;
  const double t0 = now_us();
  HIP_TRY(c, hipSetDevice(c->device));
  AlignSmall& S = c->aln_small[0];
  for (int mt = 0; mt < 2; mt++) {
    if (!ps.dev[mt].aln.uploaded) { if (int e = aln_small_reserve(c, c->aln_small[mt])) return e; }
    if (int e = aln_upload_index(c, *mm[mt], ps.dev[mt].aln)) return e;
  }
  // windows: mate 1's, then mate 2's; a mate-2 window with the walk of the mate-1 window at the same place shares its string
  const int n0 = (int)mm[0]->pending.size(), n1 = (int)mm[1]->pending.size(), nw = n0 + n1;
  AlnJob job;
  aln_prepare(c, *mm[0], job);
  job.wins.resize((size_t)nw);
  job.blk.resize((size_t)nw + 1);
  for (int k = 0; k < n1; k++) {
    const Walk& wk = *mm[1]->win_walk[mm[1]->pending[k]];
    if (k < n0 && wk == *mm[0]->win_walk[mm[0]->pending[k]] && mm[0]->index_read_len == mm[1]->index_read_len) {
      job.wins[(size_t)(n0 + k)] = job.wins[(size_t)k];
    } else {
      int32_t off = 0;
      std::string ws = mm[1]->window_string(c->g, wk, &off);
      job.wins[(size_t)(n0 + k)] = AlnWindow{(int32_t)job.wstr.size(), (int32_t)ws.size(), off};
      job.wstr += ws;
    }
    job.blk[(size_t)(n0 + k) + 1] = job.blk[(size_t)(n0 + k)] + 2 * aln_span_chunks(job.wins[(size_t)(n0 + k)].len, mm[1]->index_read_len);
  }
  hipStream_t st = c->stream;
  job.stream = st;
  const size_t in_bytes = align16(nw * sizeof(AlnWindow)) + align16((nw + 1) * sizeof(int32_t)) + align16(job.wstr.size() + 16);
  if (nw > 4096 || in_bytes > ((size_t)1 << 18) || job.blk[(size_t)nw] == 0) return 1;
  if (int e = aln_small_reserve(c, S)) return e;
  const bool direct = c->direct_write && c->knobs[8] == 0;
  if (in_bytes > S.in_cap || S.in_direct != direct) return 1;  // (the per-mate route sizes its own input block)
  char* wp = (char*)S.in_dev;
  if (!direct) { HIP_TRY(c, S.in_host.reserve(in_bytes)); wp = (char*)S.in_host.p; }
  const size_t off_blk = align16(nw * sizeof(AlnWindow)), off_str = off_blk + align16((nw + 1) * sizeof(int32_t));
  memcpy(wp, job.wins.data(), nw * sizeof(AlnWindow));
  memcpy(wp + off_blk, job.blk.data(), (nw + 1) * sizeof(int32_t));
  memcpy(wp + off_str, job.wstr.data(), job.wstr.size());
  if (direct) _mm_sfence();
  else HIP_TRY(c, hipMemcpyAsync(S.in_dev, S.in_host.p, in_bytes, hipMemcpyHostToDevice, st));
  const char* dbase = (const char*)S.in_dev;
  const AlnWindow* d_wins = (const AlnWindow*)dbase;
  const int* d_blk = (const int*)(dbase + off_blk);
  const char* d_wstr = dbase + off_str;
  AlnMates ix;
  for (int mt = 0; mt < 2; mt++) {
    const AlignDev& d = ps.dev[mt].aln;
    ix.bucket_hash[mt] = d.bucket_hash.as<uint64_t>(); ix.bucket_top[mt] = d.bucket_top.as<int32_t>(); ix.bucket_off[mt] = d.bucket_off.as<int32_t>(); ix.bucket_reads[mt] = d.bucket_reads.as<int32_t>();
    ix.n_buckets[mt] = (int)mm[mt]->bucket_hash.size();
    ix.reads[mt] = d.reads.as<char>(); ix.read_off[mt] = d.read_off.as<int64_t>();
  }
  ix.split = n0;
  hipLaunchKernelGGL(span_maxima_kernel<true>, dim3((unsigned)job.blk[(size_t)nw]), dim3(kAlnBlock), 0, st, d_wstr, d_wins, nw, mm[0]->index_read_len, d_blk,
                     S.spans.as<AlnSpan>(), S.counters.as<unsigned>(), kFastSpans, n0, mm[1]->index_read_len, ix, S.cands.as<AlnCand>(),
                     S.counters.as<unsigned>() + 1, kFastCands);
  hipLaunchKernelGGL(extend_pair_kernel, dim3(1024), dim3(64 * kAlnWaves), 0, st, S.cands.as<AlnCand>(), S.counters.as<unsigned>() + 1, kFastCands, d_wstr,
                     d_wins, ix, S.hits.as<AlnHit>());
  job.seq = ++S.out_seq;
  char* oh = (char*)S.out_host.dev;
  hipLaunchKernelGGL(publish_hits_kernel, dim3(1), dim3(256), 0, st, S.counters.as<unsigned>(), S.hits.as<AlnHit>(), kFastCands, (unsigned*)(oh + 64),
                     (AlnHit*)(oh + 128), kFastCands, (volatile unsigned long long*)oh, job.seq);
  HIP_TRY(c, hipGetLastError());
  job.enqueued = true;
  const double t1 = now_us();
  std::vector<AlnHit> hits;
  unsigned nc = 0;
  const int rc = aln_small_collect(c, S, job, hits, &nc);
  if (rc != 0) return rc;  // 1: the candidates did not fit: the per-mate route redoes the batch
  const double t2 = now_us();
  // split by mate (the window numbers of mate 2 start at n0)
  std::vector<AlnHit> h1;
  h1.reserve(hits.size());
  size_t keep = 0;
  for (const AlnHit& h : hits) {
    if (h.win >= n0) { AlnHit g = h; g.win -= n0; h1.push_back(g); }
    else hits[keep++] = h;
  }
  hits.resize(keep);
  aln_file_hits(*mm[0], n0, hits, false);
  aln_file_hits(*mm[1], n1, h1, false);
  for (int mt = 0; mt < 2; mt++) mm[mt]->pending.clear();
  const double t3 = now_us();
  c->aln_windows += nw;
  c->aln_candidates += nc;
  c->aln_us += t3 - t0;
  c->aln_stage_us[0] += t1 - t0; c->aln_stage_us[1] += t2 - t1; c->aln_stage_us[4] += t3 - t2;
  c->aln_batches++;
  return 0;
}

// `small`: this mate's small-batch buffers (null: general route only); `job`: strings already built and possibly the
// small-batch pipeline already in flight (eval_begin starts both mates' pipelines before it waits for either)
int gpu_align_pending(gaml_hip_ctx* c, ShortMate& m, AlignDev& d, AlignSmall* small = nullptr, AlnJob* job_in = nullptr) {
  if (m.pending.empty()) return 0;
  if (!aln_gpu_capable(c, m)) {
    m.flush_pending_cpu(c->g);
    return 0;
  }
  const double t0 = now_us();
  HIP_TRY(c, hipSetDevice(c->device));
  if (!d.uploaded && small) { if (int e = aln_small_reserve(c, *small)) return e; }
  if (int e = aln_upload_index(c, m, d)) return e;
  AlignScratch& S = c->aln_scratch;
  AlnJob local;
  AlnJob& job = job_in ? *job_in : local;
  if (!job.prepared) aln_prepare(c, m, job);
  const int nw = (int)m.pending.size();
  const std::string& wstr = job.wstr;
  const std::vector<AlnWindow>& wins = job.wins;
  const std::vector<int32_t>& blk = job.blk;
  bool small_done = false, device_sorted = false;
  unsigned nc = 0;
  std::vector<AlnHit> hits;
  double t1 = now_us(), t2 = t1, t3 = t1;
  if (small && c->knobs[5] != 3) {  // knob 5 = 3: always the general route (tests compare the two)
    int rc = job.enqueued ? 0 : aln_small_enqueue(c, m, d, *small, job, c->stream);
    if (rc == 0) rc = aln_small_collect(c, *small, job, hits, &nc);
    if (rc < 0) return rc;
    small_done = rc == 0;
    t2 = t3 = now_us();
  }
  if (!small_done) {
  HIP_TRY(c, S.wstr.reserve(std::max<size_t>(16, wstr.size())));
  HIP_TRY(c, S.wins.reserve(nw * sizeof(AlnWindow)));
  HIP_TRY(c, S.blk.reserve((nw + 1) * sizeof(int32_t)));
  HIP_TRY(c, S.counters.reserve(256));
  if (!wstr.empty()) HIP_TRY(c, hipMemcpy(S.wstr.p, wstr.data(), wstr.size(), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(S.wins.p, wins.data(), nw * sizeof(AlnWindow), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(S.blk.p, blk.data(), (nw + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
  t1 = now_us();
  size_t cap_spans = std::max<size_t>(1 << 16, wstr.size());        // a span per window base and strand at most ~2x
  size_t cap_cands = std::max<size_t>(1 << 18, 8 * wstr.size());
  unsigned counts[2] = {0, 0};
  for (int attempt = 0; attempt < 6; attempt++) {
    HIP_TRY(c, S.spans.reserve(cap_spans * sizeof(AlnSpan)));
    HIP_TRY(c, S.cands.reserve(cap_cands * sizeof(AlnCand)));
    HIP_TRY(c, hipMemset(S.counters.p, 0, 16));
    if (blk[(size_t)nw] > 0) {
      hipLaunchKernelGGL(span_maxima_kernel<false>, dim3((unsigned)blk[(size_t)nw]), dim3(kAlnBlock), 0, 0, S.wstr.as<char>(), S.wins.as<AlnWindow>(), nw,
                         m.index_read_len, S.blk.as<int>(), S.spans.as<AlnSpan>(), S.counters.as<unsigned>(), (unsigned)cap_spans, INT_MAX, 0, AlnMates{},
                         (AlnCand*)nullptr, (unsigned*)nullptr, 0u);
      HIP_TRY(c, hipGetLastError());
    }
    hipLaunchKernelGGL(candidates_kernel, dim3(256), dim3(kAlnBlock), 0, 0, S.spans.as<AlnSpan>(), S.counters.as<unsigned>(),
                       (unsigned)cap_spans, d.bucket_hash.as<uint64_t>(), d.bucket_top.as<int32_t>(), d.bucket_off.as<int32_t>(), d.bucket_reads.as<int32_t>(),
                       (int)m.bucket_hash.size(), S.cands.as<AlnCand>(), S.counters.as<unsigned>() + 1, (unsigned)cap_cands);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpy(counts, S.counters.p, sizeof(counts), hipMemcpyDeviceToHost));
    if (counts[0] <= cap_spans && counts[1] <= cap_cands) break;
    cap_spans = std::max<size_t>(cap_spans, (size_t)counts[0] + 16);
    cap_cands = std::max<size_t>(cap_cands, (size_t)counts[1] + 16);
    if (attempt == 5) { m.flush_pending_cpu(c->g); return 0; }
  }
  t2 = now_us();
  t3 = t2;
  nc = counts[1];
  hits.assign(nc, AlnHit{0, 0, -1, 0, 0, 0});
  if (nc) {
    HIP_TRY(c, S.hits.reserve((size_t)nc * sizeof(AlnHit)));
    hipLaunchKernelGGL(extend_kernel, dim3((nc + kAlnWaves - 1) / kAlnWaves), dim3(64 * kAlnWaves), 0, 0, S.cands.as<AlnCand>(), S.counters.as<unsigned>() + 1,
                       (unsigned)cap_cands, S.wstr.as<char>(), S.wins.as<AlnWindow>(), d.reads.as<char>(), d.read_off.as<int64_t>(),
                       S.hits.as<AlnHit>());
    HIP_TRY(c, hipGetLastError());
    if (c->knobs[9]) {
      HIP_TRY(c, hipDeviceSynchronize()); t3 = now_us();
    }
    // Large batches: order the hits on the device (window, position, read, strand, order; failed extensions
    // last) and fetch only the successful ones; the host then only walks them. Keys: read < 2^31, order < 2^24.
    int32_t longest = 0;  // (the device sort packs a span's order -- an index into its window -- into 24 bits)
    for (const AlnWindow& w : wins) longest = std::max(longest, w.len);
    if (nc >= 100000 && c->knobs[5] != 2 && longest < (1 << 24)) {
      const size_t n = nc;
      HIP_TRY(c, S.sort_keys.reserve(4 * n * sizeof(unsigned long long)));   // minor | major | two alternates
      HIP_TRY(c, S.sort_idx.reserve(2 * n * sizeof(unsigned)));
      HIP_TRY(c, S.hits_sorted.reserve(n * sizeof(AlnHit)));
      unsigned long long* k_minor = S.sort_keys.as<unsigned long long>();
      unsigned long long* k_major = k_minor + n;
      unsigned long long* k_alt = k_major + n;
      unsigned long long* k_alt2 = k_alt + n;
      unsigned* idx = S.sort_idx.as<unsigned>();
      unsigned* idx_alt = idx + n;
      unsigned* n_ok = S.counters.as<unsigned>() + 2;
      HIP_TRY(c, hipMemset(n_ok, 0, sizeof(unsigned)));
      const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
      hipLaunchKernelGGL(hit_keys_kernel, dim3(grid), dim3(256), 0, 0, S.hits.as<AlnHit>(), (unsigned)n, k_minor, k_major, idx, n_ok);
      HIP_TRY(c, hipGetLastError());
      size_t tmp_bytes = 0;
      HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k_minor, k_alt, idx, idx_alt, (int)n, 0, 56, (hipStream_t)0));
      HIP_TRY(c, S.sort_tmp.reserve(tmp_bytes));
      HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(S.sort_tmp.p, tmp_bytes, k_minor, k_alt, idx, idx_alt, (int)n, 0, 56, (hipStream_t)0));
      // second, stable pass by (window, position): the major keys in the order of the first pass
      hipLaunchKernelGGL(gather_u64_kernel, dim3(grid), dim3(256), 0, 0, k_major, idx_alt, (unsigned)n, k_alt);
      HIP_TRY(c, hipGetLastError());
      size_t tmp2 = 0;
      HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp2, k_alt, k_alt2, idx_alt, idx, (int)n, 0, 64, (hipStream_t)0));
      HIP_TRY(c, S.sort_tmp.reserve(tmp2));
      HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(S.sort_tmp.p, tmp2, k_alt, k_alt2, idx_alt, idx, (int)n, 0, 64, (hipStream_t)0));
      hipLaunchKernelGGL(gather_hits_kernel, dim3(grid), dim3(256), 0, 0, S.hits.as<AlnHit>(), idx, (unsigned)n, S.hits_sorted.as<AlnHit>());
      HIP_TRY(c, hipGetLastError());
      unsigned ok_count = 0;
      HIP_TRY(c, hipMemcpy(&ok_count, n_ok, sizeof(unsigned), hipMemcpyDeviceToHost));
      if (c->knobs[9]) t3 = now_us();
      hits.resize(ok_count);
      if (ok_count) HIP_TRY(c, hipMemcpy(hits.data(), S.hits_sorted.p, (size_t)ok_count * sizeof(AlnHit), hipMemcpyDeviceToHost));
      device_sorted = true;
    } else {
      HIP_TRY(c, hipMemcpy(hits.data(), S.hits.p, (size_t)nc * sizeof(AlnHit), hipMemcpyDeviceToHost));
    }
  }
  }  // general route
  const double t4 = now_us();
  if (!c->knobs[9]) t3 = t4;
  aln_file_hits(m, nw, hits, device_sorted);
  m.pending.clear();
  c->aln_windows += nw;
  c->aln_candidates += nc;
  const double t5 = now_us();
  c->aln_us += t5 - t0;
  c->aln_stage_us[0] += t1 - t0; c->aln_stage_us[1] += t2 - t1; c->aln_stage_us[2] += t3 - t2; c->aln_stage_us[3] += t4 - t3; c->aln_stage_us[4] += t5 - t4;
  c->aln_batches++;
  return 0;
}

int align_pending_pair(gaml_hip_ctx* c, PairedSet& ps) {
  if (ps.mate[0].pending.empty() && ps.mate[1].pending.empty()) return 0;
  if (c->device >= 0) {
    const int rc = aln_pair_small(c, ps);
    if (rc <= 0) return rc;
  }
  AlnJob job[2];
  if (c->device >= 0 && c->knobs[5] != 3) {
    for (int mt = 0; mt < 2; mt++) {
      ShortMate& m = ps.mate[mt];
      if (m.pending.empty() || !aln_gpu_capable(c, m)) continue;
      HIP_TRY(c, hipSetDevice(c->device));
      if (!ps.dev[mt].aln.uploaded) { if (int e = aln_small_reserve(c, c->aln_small[mt])) return e; }
      if (int e = aln_upload_index(c, m, ps.dev[mt].aln)) return e;
      aln_prepare(c, m, job[mt]);
      const int rc = aln_small_enqueue(c, m, ps.dev[mt].aln, c->aln_small[mt], job[mt], mt == 0 ? c->stream : c->aux_stream);  // side by side
      if (rc < 0) return rc;
    }
  }
  for (int mt = 0; mt < 2; mt++)
    if (int e = gpu_align_pending(c, ps.mate[mt], ps.dev[mt].aln, c->device >= 0 ? &c->aln_small[mt] : nullptr, &job[mt])) return e;
  return 0;
}

std::vector<ShortMate*> filter_mates(gaml_hip_ctx* c) {  // mates whose windows feed a position filter, in handle order
  std::vector<ShortMate*> v;
  for (auto& h : c->handles)
    if (h.kind == 1) { v.push_back(&c->paireds[h.idx]->mate[0]); v.push_back(&c->paireds[h.idx]->mate[1]); }
  return v;
}

// phase 1: path checks, window registration/alignment, window placements. Returns (via *pending)
// how many windows were added whose largest positions other shards have not seen yet.
int eval_begin(gaml_hip_ctx* c, const int32_t* flat, const int64_t* offs, int32_t n_paths, int64_t* pending) {
  if (!c->have_graph) return fail(c, GAML_HIP_ESTATE, "no graph set");
  if (n_paths < 0 || (n_paths > 0 && (!flat || !offs))) return fail(c, GAML_HIP_EINVAL, "bad path arguments");
  const double t0 = now_us();
  c->eval_serial++;
  for (int32_t i = 0; i < n_paths; i++) if (offs[i + 1] < offs[i]) return fail(c, GAML_HIP_EINVAL, "path offsets must not decrease");
  // Paired sets take the paths in the ABI's flat form (their planner diffs them against the previous call's); the
  // other kinds, or a context without a paired set, get them as vectors (reused: a set of ~900 short paths would
  // otherwise cost ~900 allocations per call).
  const bool want_vectors = c->paireds.empty() || !c->singles.empty() || !c->pacbios.empty();
  c->pending_paths_valid = want_vectors;
  if (want_vectors) {
    c->pending_paths.resize((size_t)n_paths);
    for (int32_t i = 0; i < n_paths; i++) c->pending_paths[i].assign(flat + offs[i], flat + offs[i + 1]);
    for (auto& p : c->pending_paths)
      for (int32_t x : p)
        if (x >= c->g.n()) return fail(c, GAML_HIP_EINVAL, "path refers to a node outside the graph");
    c->pending_total_len = 0;
    for (auto& p : c->pending_paths) c->pending_total_len += walk_length(c->g, p);  // GetTotalLen graph.cc:1775-1781
  }
  if (c->pending_prep.size() != c->paireds.size()) { c->pending_prep.clear(); c->pending_prep.resize(c->paireds.size()); }
  for (size_t i = 0; i < c->paireds.size(); i++) {
    if (!c->pending_prep[i]) c->pending_prep[i].reset(new PairedPrep());  // (reused from call to call: its vectors keep their capacity)
    if (int e = prepare_paired_structure(c, *c->paireds[i], flat, offs, n_paths)) return e;
    if (!want_vectors) c->pending_total_len = c->paireds[i]->planner.total_len();
    // windows registered by pass 1 get their records now, all at once (GPU aligner when there is a device): small
    // batches of the two mates go out together on the stream before either is waited for
    const double ta = now_us();
    if (int e = align_pending_pair(c, *c->paireds[i])) return e;
    c->prof[2] = (i ? c->prof[2] : 0.0) + now_us() - ta;  // alignment of newly registered windows (inside pass 1)
  }
  int64_t n = 0;
  for (ShortMate* m : filter_mates(c)) {
    if (c->peers == 1) m->unsynced.clear();  // nothing to exchange: global == local
    n += (int64_t)m->unsynced.size();
  }
  for (auto& s : c->singles) s->mate.unsynced.clear();  // single-end scoring has no position filter
  if (pending) *pending = n;
  c->pending_open = true;
  c->pending_host_us = now_us() - t0;
  c->prof[0] = c->pending_host_us;  // pass 1
  return 0;
}

// phase 2: thresholds + device tables + launches; leaves 4 doubles per read set at d_partials
int eval_finish(gaml_hip_ctx* c, void* d_partials, hipStream_t st) {
  if (c->device < 0) return fail(c, GAML_HIP_ENODEVICE, "scoring needs a HIP device: this context is host-only");
  if (!c->pending_open) return fail(c, GAML_HIP_ESTATE, "no evaluation in progress");
  for (ShortMate* m : filter_mates(c))
    if (!m->unsynced.empty())
      return fail(c, GAML_HIP_ESTATE, "sharded context: new windows were aligned; exchange their largest positions "
                                      "(gaml_hip_eval_pending_maxpos -> all-reduce(max) -> gaml_hip_eval_apply_maxpos) before finishing");
  if (c->peers > 1) {
    // the coverage penalty depends on the union of all ranks' well-aligned pairs per path position
    // (graph.cc:1893-1919) / of all reads' intervals (graph.cc:3226-3250): a per-rank value would be wrong
    for (auto& ps : c->paireds)
      if (ps->cfg.penalty_constant > 0 && !c->defer_cov)
        return fail(c, GAML_HIP_ESTATE, "penalty_constant > 0 on a sharded context: the coverage maps of all ranks must be merged "
                                        "(gaml_hip_eval_score_async -> gaml_hip_eval_coverage_export_async -> all-gather -> "
                                        "gaml_hip_eval_coverage_finish_async)");
    // (single-end sets: bad_bases is identically 0 in the reference, graph.cc:1701-1733 -- nothing to exchange)
    for (auto& ps : c->pacbios)
      if (ps->cfg.penalty_constant > 0 && !c->defer_cov)
        return fail(c, GAML_HIP_ESTATE, "penalty_constant > 0 on a sharded PacBio set: the alignment intervals of all ranks must be merged "
                                        "(gaml_hip_eval_score_async -> gaml_hip_eval_pacbio_export_async -> all-gather -> gaml_hip_eval_pacbio_finish_async)");
  }
  const std::vector<Walk>& paths = c->pending_paths;
  const int32_t total_len = c->pending_total_len;
  auto order = scoring_order(c);
  const double t0 = now_us();
  c->t_host_us = 0;
  int k = 0;
  for (auto& h : order) {
    const double tk = now_us();
    double* out4 = (double*)d_partials + 4 * k;  // every scorer's last block writes its 4 partials here
    int e = 0;
    if (h.kind == 0) e = launch_single(c, *c->singles[h.idx], paths, total_len, st, out4);
    else if (h.kind == 1) e = launch_paired(c, *c->paireds[h.idx], *c->pending_prep[h.idx], total_len, st, out4);
    else e = launch_pacbio(c, *c->pacbios[h.idx], paths, st, out4);
    if (e) return e;
    c->t_host_us -= tk;  // launch_* added its "host part finished" stamp
    k++;
  }
  c->t_dev_wall_us = now_us() - t0 - c->t_host_us;
  c->t_host_us += c->pending_host_us;
  c->pending_open = false;
  return 0;
}

int evaluate(gaml_hip_ctx* c, const int32_t* flat, const int64_t* offs, int32_t n_paths, void* d_partials,
             hipStream_t st, int32_t* total_len_out) {
  if (c->device < 0) return fail(c, GAML_HIP_ENODEVICE, "scoring needs a HIP device: this context is host-only");
  int64_t pending = 0;
  if (int e = eval_begin(c, flat, offs, n_paths, &pending)) return e;
  if (total_len_out) *total_len_out = c->pending_total_len;
  return eval_finish(c, d_partials, st);
}

int combine(gaml_hip_ctx* c, const double* partials, double* prob_out, int32_t* zeros_out, int32_t total_len) {
  // prob = sum over read sets of weight * (mean log-probability - bad_bases * penalty)
  // (prob_calculator.h:70-107; graph.cc:1515,1988 / 1536,1742 / 3087,3260)
  auto order = scoring_order(c);
  double prob = 0;
  int k = 0;
  for (auto& h : order) {
    const double sum = partials[4 * k], zeros = partials[4 * k + 1], bad = partials[4 * k + 2], n = partials[4 * k + 3];
    double v;
    if (h.kind == 0) {
      const gaml_single_cfg& cfg = c->singles[h.idx]->cfg;
      v = (sum / (double)(int64_t)n - bad * cfg.penalty_constant) * cfg.weight;
    } else if (h.kind == 1) {
      const gaml_paired_cfg& cfg = c->paireds[h.idx]->cfg;
      // no sum of logs and floors is NaN; a block of the scoring launch poisons its partial when the two mates'
      // occurrence tables disagree about a window both hold (kernels.hip.h paired_static4_body): report, do not score
      if (std::isnan(sum)) return fail(c, GAML_HIP_ESTATE, "paired read set: the mates' occurrence tables disagree about a shared window (static memo index)");
      v = (sum / (double)(int64_t)n - bad * cfg.penalty_constant) * cfg.weight;
    } else {
      const gaml_single_cfg& cfg = c->pacbios[h.idx]->cfg;
      int tl = total_len == 0 ? 1 : total_len;
      v = (sum / (double)(int64_t)n - std::log((double)(2 * tl)) - bad * cfg.penalty_constant) * cfg.weight;
    }
    prob += v;
    if (zeros_out) { zeros_out[2 * k] = (int32_t)zeros; zeros_out[2 * k + 1] = (int32_t)n; }
    k++;
  }
  *prob_out = prob;
  return 0;
}

}  // namespace

// =========================================================================================
// what multi.hip needs from a context (internal.h)
// =========================================================================================
namespace gaml {
int ctx_fail(gaml_hip_ctx* c, int code, const std::string& msg) { return fail(c, code, msg); }
hipStream_t ctx_stream(const gaml_hip_ctx* c) { return c->stream; }
int ctx_device(const gaml_hip_ctx* c) { return c->device; }
int ctx_rank(const gaml_hip_ctx* c) { return c->rank; }
int ctx_world(const gaml_hip_ctx* c) { return c->world; }
int ctx_peers(const gaml_hip_ctx* c) { return c->peers; }
MultiState* ctx_multi(const gaml_hip_ctx* c) { return c->multi; }
void ctx_set_multi(gaml_hip_ctx* c, MultiState* m) { c->multi = m; }
CommState* ctx_comm(const gaml_hip_ctx* c) { return c->comm; }
void ctx_set_comm(gaml_hip_ctx* c, CommState* s) { c->comm = s; }
gaml_hip_ctx* ctx_new_parent() { return new gaml_hip_ctx(); }
void ctx_note_reduced(gaml_hip_ctx* c, const double* partials) {
  auto order = scoring_order(c);
  for (size_t k = 0; k < order.size(); k++) {
    if (order[k].kind == 1) c->paireds[order[k].idx]->last_bad_bases = (int64_t)partials[4 * k + 2];
    else if (order[k].kind == 2) c->pacbios[order[k].idx]->last_bad_bases = (int64_t)partials[4 * k + 2];
  }
}
int ctx_fetch_wait_bounded(gaml_hip_ctx* c, double* out, int32_t n_doubles, double timeout_s) {
  if (!c || !out || n_doubles <= 0 || !c->fetch_host.p) return fail(c, GAML_HIP_EINVAL, "bad arguments / no fetch in flight");
  volatile unsigned long long* word = (volatile unsigned long long*)c->fetch_host.p;
  const double t0 = now_us(), limit = timeout_s * 1e6;
  unsigned spins = 0;
  while (*word != c->fetch_seq) {
    if ((++spins & 1023u) == 0) {
      const double waited = now_us() - t0;
      if (waited > limit) return 1;
      if (waited > 5000.0) std::this_thread::sleep_for(std::chrono::microseconds(50));  // (a collective that takes milliseconds: stop burning the core)
    }
    __builtin_ia32_pause();
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  memcpy(out, (const char*)c->fetch_host.p + 512, (size_t)n_doubles * sizeof(double));
  return 0;
}
void ctx_eval_abandon(gaml_hip_ctx* c) { if (c) { c->pending_open = false; c->pending_cov.clear(); c->pending_pb.clear(); } }
bool ctx_has_penalty(const gaml_hip_ctx* c) {
  for (auto& ps : c->paireds) if (ps->cfg.penalty_constant > 0) return true;
  for (auto& ps : c->pacbios) if (ps->cfg.penalty_constant > 0) return true;
  return false;
}
}  // namespace gaml

// a multi-device context forwards every call to its shards (multi.hip)
#define MULTI_FWD(c, call) do { if ((c) && (c)->multi) return gaml::call; } while (0)
// introspection / tuning entry points without a multi-device meaning act on shard 0
#define MULTI_SHARD0(c) do { if ((c) && (c)->multi) (c) = gaml::multi_shard((c)->multi, 0); } while (0)
#define MULTI_REFUSE(c, what) do { if ((c) && (c)->multi) return fail(c, GAML_HIP_ESTATE, "multi-device context: " what); } while (0)

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {


int gaml_hip_create(gaml_hip_ctx** out, int device) {
  if (!out) return GAML_HIP_EINVAL;
  *out = nullptr;
  std::unique_ptr<gaml_hip_ctx> c(new gaml_hip_ctx());
  c->device = device;
  if (device >= 0) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || device >= count) {
      fprintf(stderr, "gaml_hip_create: HIP device %d not available (%s)\n", device, e == hipSuccess ? "ordinal out of range" : hipGetErrorString(e));
      return GAML_HIP_ENODEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) return GAML_HIP_EHIP;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return GAML_HIP_EHIP;
    if (hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess) return GAML_HIP_EHIP;
    // large BAR (every MI300-class part): per-call tables are written by the host straight into device memory
    // (paired_launch.hip.h: Arena). GAML_HIP_DIRECT_WRITE=0 forces the staged path.
    int large_bar = 0;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) != hipSuccess) large_bar = 0;
    const char* dw = getenv("GAML_HIP_DIRECT_WRITE");
    c->direct_write = large_bar != 0 && !(dw && dw[0] == '0');
  }
  *out = c.release();
  return GAML_HIP_OK;
}

void gaml_hip_destroy(gaml_hip_ctx* c) {
  if (c && c->shm_base) { munmap(c->shm_base, c->shm_bytes); c->shm_base = nullptr; }
  if (!c) return;
  if (c->multi) { gaml::multi_destroy(c->multi); c->multi = nullptr; }
  if (c->device >= 0) {
    (void)hipSetDevice(c->device);
    // everything that may still use this context's memory: the library's streams, a caller's stream the last fetch /
    // stream-ordered evaluation went to (all streams of the device: the caller's is not ours to name), then the communicator
    (void)hipDeviceSynchronize();
    if (c->comm) { gaml::comm_destroy(c->comm); c->comm = nullptr; }
    auto drop_stage = [](Staging& s) { for (int k = 0; k < kRing; k++) { s.host[k].release(); if (s.done[k]) (void)hipEventDestroy(s.done[k]); } };
    for (auto& s : c->singles) { s->dev.first.release(); s->dev.extra.release(); s->dev.pows.release(); s->lens.release(); s->probs.release(); s->tabs.release(); s->occ_arena.release(); s->red.release(); drop_stage(s->stage); }
    for (auto& s : c->paireds) {
      if (s->rebuild.th.joinable()) s->rebuild.th.join();
      if (s->rebuild.stream) (void)hipStreamDestroy(s->rebuild.stream);
      s->tab.release(); s->rebuild.tab.release();
      for (int m = 0; m < 2; m++) { s->dev[m].first.release(); s->dev[m].extra.release(); s->dev[m].pows.release(); s->dev[m].aln.release(); }
      s->delta_dev.release(); s->dl_slot.release(); s->dl_spill.release(); s->dl_rec[0].release(); s->dl_rec[1].release(); s->dl_patch.release(); s->rebuild.sh_dl_slot.release(); s->rebuild.sh_dl_spill.release(); s->rebuild.sh_dl_rec[0].release(); s->rebuild.sh_dl_rec[1].release(); drop_stage(s->stage_delta); s->h_part_sum.release(); s->h_part_zero.release(); s->h_timeline.release();
      s->probs.release(); s->tabs.release(); s->arena.release(); s->persist.release(); s->cov_bits.release(); s->bad.release(); if (s->ev_tables) (void)hipEventDestroy(s->ev_tables); if (s->ev_ovf) (void)hipEventDestroy(s->ev_ovf);
      s->red.release(); s->gen_bits.release();
    }
    for (auto& s : c->pacbios) { s->d_lens.release(); s->rec_off.release(); s->rec_walk.release(); s->rec_logp.release(); s->walk_count.release(); s->logprobs.release(); s->red.release(); drop_stage(s->stage);
      s->d_bases.release(); s->dp.release(); s->sweep.release(); }
    c->packed.release(); c->packed_host.release(); c->batch_dev.release(); c->batch_host.release(); c->aln_scratch.release();
    c->aln_small[0].release(); c->aln_small[1].release(); c->fetch_host.release();
    for (auto& e : c->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
  }
  delete c;
}

const char* gaml_hip_last_error(const gaml_hip_ctx* c) {
  if (!c) return "null context";
  if (c->multi && c->err.empty()) return gaml::multi_last_error(c->multi);  // e.g. why the RCCL exchange is not available
  return c->err.c_str();
}

int gaml_hip_set_graph(gaml_hip_ctx* c, int32_t n_nodes, const char* bases, const int64_t* offs) {
  if (!c || n_nodes < 0 || !offs || (n_nodes > 0 && !bases)) return fail(c, GAML_HIP_EINVAL, "bad graph arguments");
  if (n_nodes & 1) return fail(c, GAML_HIP_EINVAL, "node count must be even (twin of i is i^1)");
  MULTI_FWD(c, multi_set_graph(c->multi, n_nodes, bases, offs));
  c->g.bases.assign(bases + offs[0], bases + offs[n_nodes]);
  c->g.off.resize(n_nodes + 1);
  for (int32_t i = 0; i <= n_nodes; i++) c->g.off[i] = offs[i] - offs[0];
  c->g.finish();
  c->have_graph = true;
  return GAML_HIP_OK;
}

int gaml_hip_load_graph(gaml_hip_ctx* c, const char* file) {
  if (!c || !file) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  MULTI_FWD(c, multi_load_graph(c->multi, file));
  std::string err;
  if (!c->g.load_lastgraph(file, &err)) return fail(c, GAML_HIP_EINVAL, err);
  c->have_graph = true;
  return GAML_HIP_OK;
}

int gaml_hip_set_shard(gaml_hip_ctx* c, int32_t rank, int32_t world) {
  if (!c || world < 1 || rank < 0 || rank >= world) return fail(c, GAML_HIP_EINVAL, "bad shard");
  MULTI_REFUSE(c, "its shards are the devices given to gaml_hip_create_multi");
  if (!c->handles.empty()) return fail(c, GAML_HIP_ESTATE, "set the shard before adding read sets");
  c->rank = rank; c->world = world; c->peers = world;
  return GAML_HIP_OK;
}

int gaml_hip_set_presharded(gaml_hip_ctx* c, int32_t world) {
  if (!c || world < 1) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  MULTI_REFUSE(c, "its shards are the devices given to gaml_hip_create_multi");
  if (c->world != 1) return fail(c, GAML_HIP_ESTATE, "gaml_hip_set_shard already partitions this context's reads");
  c->peers = world;
  return GAML_HIP_OK;
}

static void init_mate(gaml_hip_ctx* c, ShortMate& m, double mismatch, int64_t n, const char* b, const int64_t* offs) {
  m.mismatch = mismatch;
  m.match = 1.0 - 4 * mismatch;  // gaml.cc:813,854
  int64_t lo, hi;
  shard_range(c, n, &lo, &hi);
  m.set_reads(n, lo, hi, b, offs);
}

int gaml_hip_add_single(gaml_hip_ctx* c, const gaml_single_cfg* cfg, int32_t n, const char* bases, const int64_t* offs) {
  if (!c || !cfg || n < 0 || !offs) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  MULTI_FWD(c, multi_add_single(c->multi, cfg, n, bases, offs));
  std::unique_ptr<SingleSet> s(new SingleSet());
  s->cfg = *cfg;
  init_mate(c, s->mate, cfg->mismatch_prob, n, bases, offs);
  c->singles.push_back(std::move(s));
  c->handles.push_back(SetRef{0, (int)c->singles.size() - 1});
  return (int)c->handles.size() - 1;
}

int gaml_hip_add_paired(gaml_hip_ctx* c, const gaml_paired_cfg* cfg, int32_t n, const char* b1, const int64_t* o1,
                        const char* b2, const int64_t* o2) {
  if (!c || !cfg || n < 0 || !o1 || !o2) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  MULTI_FWD(c, multi_add_paired(c->multi, cfg, n, b1, o1, b2, o2));
  std::unique_ptr<PairedSet> s(new PairedSet());
  s->cfg = *cfg;
  {  // the two mates' read indexes are independent: build them side by side
    std::thread second([&] { init_mate(c, s->mate[1], cfg->mismatch_prob, n, b2, o2); });
    init_mate(c, s->mate[0], cfg->mismatch_prob, n, b1, o1);
    second.join();
  }
  s->mate[0].defer_alignment = s->mate[1].defer_alignment = true;  // batched: see gpu_align_pending
  if (s->mate[0].max_len > 65535 || s->mate[1].max_len > 65535) return fail(c, GAML_HIP_EINVAL, "paired reads longer than 65535 bases");
  c->paireds.push_back(std::move(s));
  c->handles.push_back(SetRef{1, (int)c->paireds.size() - 1});
  return (int)c->handles.size() - 1;
}

int gaml_hip_add_pacbio(gaml_hip_ctx* c, const gaml_single_cfg* cfg, int32_t n, const int32_t* lens) {
  if (!c || !cfg || n < 0 || (n > 0 && !lens)) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  MULTI_FWD(c, multi_add_pacbio(c->multi, cfg, n, lens));
  std::unique_ptr<PacbioSet> s(new PacbioSet());
  s->cfg = *cfg;
  s->n_global = n;
  shard_range(c, n, &s->lo, &s->hi);
  s->lens.assign(lens + s->lo, lens + s->hi);
  s->max_len = 0;
  for (int32_t i = 0; i < n; i++) s->max_len = std::max(s->max_len, lens[i]);  // CalcMaxReadLen graph.cc:1456-1461 (all reads)
  s->log_mismatch = std::log(cfg->mismatch_prob);          // logdouble(mismatch_prob) graph.h:446-449
  s->log_match = std::log(1.0 - 4 * cfg->mismatch_prob);
  c->pacbios.push_back(std::move(s));
  c->handles.push_back(SetRef{2, (int)c->pacbios.size() - 1});
  return (int)c->handles.size() - 1;
}

int gaml_hip_add_pacbio_reads(gaml_hip_ctx* c, const gaml_single_cfg* cfg, int32_t n, const char* bases, const int64_t* offs,
                              const char* names) {
  if (!c || !cfg || n < 0 || !offs || (n > 0 && (!bases || !names))) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  MULTI_FWD(c, multi_add_pacbio_reads(c->multi, cfg, n, bases, offs, names));
  for (int64_t i = offs[0]; i < offs[n]; i++)
    if (bases[i] == '\n' || bases[i] == '-') return fail(c, GAML_HIP_EINVAL, "a PacBio read holds a separator or gap character");
  std::vector<int32_t> lens(n);
  for (int32_t i = 0; i < n; i++) {
    if (offs[i + 1] < offs[i] || offs[i + 1] - offs[i] > INT32_MAX) return fail(c, GAML_HIP_EINVAL, "bad read offsets");
    lens[i] = (int32_t)(offs[i + 1] - offs[i]);
  }
  int h = gaml_hip_add_pacbio(c, cfg, n, lens.data());
  if (h < 0) return h;
  PacbioSet& s = *c->pacbios[c->handles[h].idx];
  s.have_reads = true;
  s.base_off.assign(1, 0);
  for (int64_t i = s.lo; i < s.hi; i++) {
    s.bases.append(bases + offs[i], bases + offs[i + 1]);
    s.base_off.push_back((int64_t)s.bases.size());
  }
  // read ids in order of first appearance (GetReadId graph.h:410-420); names are '\n'-separated
  const char* p = names;
  for (int32_t i = 0; i < n; i++) {
    const char* e = strchr(p, '\n');
    std::string name = e ? std::string(p, e) : std::string(p);
    p = e ? e + 1 : p + name.size();
    s.name_id[name] = i;
  }
  return h;
}

int gaml_hip_add_single_fastq(gaml_hip_ctx* c, const gaml_single_cfg* cfg, const char* fastq) {
  if (!c || !cfg || !fastq) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  std::string bases, err; std::vector<int64_t> offs;
  if (!read_fastq(fastq, bases, offs, &err)) return fail(c, GAML_HIP_EINVAL, err);
  return gaml_hip_add_single(c, cfg, (int32_t)offs.size() - 1, bases.data(), offs.data());
}
int gaml_hip_add_paired_fastq(gaml_hip_ctx* c, const gaml_paired_cfg* cfg, const char* f1, const char* f2) {
  if (!c || !cfg || !f1 || !f2) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  std::string b1, b2, err; std::vector<int64_t> o1, o2;
  if (!read_fastq(f1, b1, o1, &err) || !read_fastq(f2, b2, o2, &err)) return fail(c, GAML_HIP_EINVAL, err);
  if (o1.size() != o2.size()) return fail(c, GAML_HIP_EINVAL, "mate files hold different numbers of reads");  // graph.cc:1962
  return gaml_hip_add_paired(c, cfg, (int32_t)o1.size() - 1, b1.data(), o1.data(), b2.data(), o2.data());
}
int gaml_hip_add_pacbio_fastq(gaml_hip_ctx* c, const gaml_single_cfg* cfg, const char* fastq) {
  if (!c || !cfg || !fastq) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  std::string bases, err, joined; std::vector<int64_t> offs; std::vector<std::string> names;
  if (!read_fastq(fastq, bases, offs, &err, &names)) return fail(c, GAML_HIP_EINVAL, err);
  for (auto& nm : names) { joined += nm; joined += '\n'; }
  return gaml_hip_add_pacbio_reads(c, cfg, (int32_t)offs.size() - 1, bases.data(), offs.data(), joined.c_str());
}

static ShortMate* mate_of(gaml_hip_ctx* c, int readset, int mate) {
  if (!c || readset < 0 || readset >= (int)c->handles.size()) return nullptr;
  SetRef h = c->handles[readset];
  if (h.kind == 0) return mate == 0 ? &c->singles[h.idx]->mate : nullptr;
  if (h.kind == 1) return (mate == 0 || mate == 1) ? &c->paireds[h.idx]->mate[mate] : nullptr;
  return nullptr;
}

int gaml_hip_put_window_records(gaml_hip_ctx* c, int readset, int mate, const int32_t* subpath, int32_t len,
                                const gaml_aligment* recs, int64_t n) {
  MULTI_FWD(c, multi_put_window_records(c->multi, readset, mate, subpath, len, recs, n));
  ShortMate* m = mate_of(c, readset, mate);
  if (!m || !subpath || len <= 0 || n < 0 || (n > 0 && !recs)) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  if (!c->have_graph) return fail(c, GAML_HIP_ESTATE, "no graph set");
  Walk w(subpath, subpath + len);
  int64_t walk_len = 0;
  for (int32_t x : w) {
    if (x < 0 || x >= c->g.n()) return fail(c, GAML_HIP_EINVAL, "window refers to a node outside the graph");
    walk_len += c->g.len(x);
  }
  if (m->find(w) >= 0) return fail(c, GAML_HIP_ESTATE, "window already cached");
  std::vector<gaml_aligment> v;
  for (int64_t i = 0; i < n; i++) {
    // Records from outside are indexed with on the device (pow tables by edit count, coverage bitmap by position):
    // anything the library's own aligner could not have produced is refused, not clamped.
    if (recs[i].read_id < 0 || recs[i].read_id >= m->n_global) return fail(c, GAML_HIP_EINVAL, "record names a read outside the read set");
    if (recs[i].orientation != 0 && recs[i].orientation != 1) return fail(c, GAML_HIP_EINVAL, "record orientation must be 0 or 1");
    if (recs[i].position < 0 || recs[i].position > walk_len) return fail(c, GAML_HIP_EINVAL, "record position outside the window");
    if (recs[i].edit_dist < 0 || recs[i].edit_dist > 255) return fail(c, GAML_HIP_EINVAL, "record edit distance outside 0..255");
    if (recs[i].read_id < m->lo || recs[i].read_id >= m->hi) continue;  // other shard
    if (recs[i].edit_dist > m->lens[recs[i].read_id - m->lo]) return fail(c, GAML_HIP_EINVAL, "record edit distance exceeds the read length");
    gaml_aligment r = recs[i];
    r.read_id -= (int32_t)m->lo;
    v.push_back(r);
  }
  std::stable_sort(v.begin(), v.end(), [](const gaml_aligment& a, const gaml_aligment& b) {  // graph.cc:1024-1026
    return a.position == b.position ? a.read_id < b.read_id : a.position < b.position;
  });
  m->add_window(c->g, w, v);
  return GAML_HIP_OK;
}

int gaml_hip_put_pacbio_records(gaml_hip_ctx* c, int readset, const int32_t* subpath, int32_t len,
                                const gaml_pacbio_aligment* recs, int64_t n) {
  MULTI_FWD(c, multi_put_pacbio_records(c->multi, readset, subpath, len, recs, n));
  if (!c || readset < 0 || readset >= (int)c->handles.size() || c->handles[readset].kind != 2 || !subpath || len <= 0 || n < 0)
    return fail(c, GAML_HIP_EINVAL, "bad arguments");
  PacbioSet& s = *c->pacbios[c->handles[readset].idx];
  for (int64_t i = 0; i < n; i++) {
    if (recs[i].read_id < 0 || recs[i].read_id >= s.n_global) return fail(c, GAML_HIP_EINVAL, "record names a read outside the read set");
    // an alignment interval ends behind its begin: the reference's coverage sweep closes an interval that is not open
    // otherwise (inters.erase(end()), graph.cc:3229-3231)
    if (recs[i].position_end < recs[i].position) return fail(c, GAML_HIP_EINVAL, "record with position_end < position");
  }
  Walk w(subpath, subpath + len);
  auto it = s.walk_id.find(w);
  int32_t id;
  if (it == s.walk_id.end()) { id = (int32_t)s.recs.size(); s.walk_id.emplace(w, id); s.recs.emplace_back(); }
  else id = it->second;
  for (int64_t i = 0; i < n; i++) {
    if (recs[i].read_id < s.lo || recs[i].read_id >= s.hi) continue;
    gaml_pacbio_aligment r = recs[i];
    r.read_id -= (int32_t)s.lo;
    s.recs[id].push_back(r);
  }
  s.generation++;
  return GAML_HIP_OK;
}

namespace {
PacbioSet* pacbio_of(gaml_hip_ctx* c, int readset) {
  if (!c || readset < 0 || readset >= (int)c->handles.size() || c->handles[readset].kind != 2) return nullptr;
  return c->pacbios[c->handles[readset].idx].get();
}
// path string + node boundaries of a (normalised) path (graph.cc:2412-2431, 2662-2688)
void pacbio_path_string(const gaml_hip_ctx* c, const Walk& path, std::string* seq, std::vector<int32_t>& begins, std::vector<int32_t>& ends) {
  int64_t len = 0;
  begins.clear(); ends.clear();
  for (int32_t x : path) {
    begins.push_back((int32_t)len);
    if (x < 0) { if (seq) seq->append((size_t)-x, 'N'); len += -x; }
    else { if (seq) seq->append(c->g.seq(x), c->g.seq(x) + c->g.len(x)); len += c->g.len(x); }
    ends.push_back((int32_t)len);
  }
}
// upload one batch of DP jobs, run the banded DP kernel, fetch the log probabilities
int run_pacbio_dp(gaml_hip_ctx* c, DpDev& d, const std::string& both, const unsigned char* d_reads, const std::vector<DpJob>& jobs,
                  const std::vector<std::pair<const uint32_t*, size_t>>& ops, int64_t scratch, double log_match, double log_mismatch, double* logp,
                  float* kernel_ms, int64_t* cells_out, int32_t* dbg_lo = nullptr, int32_t* dbg_hi = nullptr, int32_t dbg_rows = 0) {
  hipStream_t st = c->stream;
  const size_t nj = jobs.size();
  HIP_TRY(c, d.path.reserve(both.size()));
  HIP_TRY(c, d.jobs.reserve(nj * sizeof(DpJob)));
  size_t n_ops = 0;
  for (const auto& part : ops) n_ops += part.second;
  HIP_TRY(c, d.ops.reserve(std::max<size_t>(1, n_ops) * sizeof(uint32_t)));
  HIP_TRY(c, d.scratch.reserve(std::max<size_t>(1, (size_t)scratch) * sizeof(double)));
  HIP_TRY(c, d.out.reserve(nj * (sizeof(double) + sizeof(long long))));
  HIP_TRY(c, hipMemcpy(d.path.p, both.data(), both.size(), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(d.jobs.p, jobs.data(), nj * sizeof(DpJob), hipMemcpyHostToDevice));
  {
    size_t at = 0;
    for (const auto& part : ops) {
      if (part.second) HIP_TRY(c, hipMemcpy(d.ops.as<uint32_t>() + at, part.first, part.second * sizeof(uint32_t), hipMemcpyHostToDevice));
      at += part.second;
    }
  }
  if (dbg_rows > 0) HIP_TRY(c, d.dbg.reserve(2 * (size_t)dbg_rows * sizeof(int32_t)));
  DpArgs a;
  a.path = d.path.as<unsigned char>(); a.path_len = (int32_t)both.size();
  a.reads = d_reads;
  a.jobs = d.jobs.as<DpJob>(); a.ops = d.ops.as<uint32_t>();
  a.scratch = d.scratch.as<double>(); a.out = d.out.as<double>(); a.cells = (long long*)(d.out.as<double>() + nj);
  a.dbg_lo = dbg_rows > 0 ? d.dbg.as<int32_t>() : nullptr;
  a.dbg_hi = dbg_rows > 0 ? d.dbg.as<int32_t>() + dbg_rows : nullptr;
  a.n_jobs = (int32_t)nj;
  a.log_match = log_match; a.log_mismatch = log_mismatch;
  hipEvent_t ev0, ev1;
  HIP_TRY(c, hipEventCreate(&ev0));
  HIP_TRY(c, hipEventCreate(&ev1));
  HIP_TRY(c, hipEventRecord(ev0, st));
  constexpr int kLanes = 16;  // lanes per alignment (one DPP row): 15 columns per chunk cover a typical row in one step
  const unsigned grid = (unsigned)(((int64_t)nj * kLanes + 255) / 256);
  hipLaunchKernelGGL(pacbio_dp_kernel<kLanes>, dim3(grid), dim3(256), 0, st, a);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipEventRecord(ev1, st));
  std::vector<long long> cells(nj);
  HIP_TRY(c, hipMemcpyAsync(logp, d.out.p, nj * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipMemcpyAsync(cells.data(), a.cells, nj * sizeof(long long), hipMemcpyDeviceToHost, st));
  if (dbg_rows > 0) {
    HIP_TRY(c, hipMemcpyAsync(dbg_lo, a.dbg_lo, dbg_rows * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(dbg_hi, a.dbg_hi, dbg_rows * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(c, hipStreamSynchronize(st));
  HIP_TRY(c, hipEventElapsedTime(kernel_ms, ev0, ev1));
  (void)hipEventDestroy(ev0);
  (void)hipEventDestroy(ev1);
  if (cells_out) { *cells_out = 0; for (long long v : cells) *cells_out += v; }
  return GAML_HIP_OK;
}
}  // namespace

int32_t gaml_hip_pacbio_missing(gaml_hip_ctx* c, int readset, const int32_t* path_in, int32_t n, int32_t* ranges, int32_t cap) {
  MULTI_FWD(c, multi_pacbio_missing(c->multi, readset, path_in, n, ranges, cap));
  PacbioSet* sp = pacbio_of(c, readset);
  if (!sp || !path_in || n <= 0 || cap < 0 || (cap > 0 && !ranges)) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  if (!c->have_graph) return fail(c, GAML_HIP_ESTATE, "no graph");
  Walk path(path_in, path_in + n);
  for (auto& x : path) {
    if (x >= c->g.n()) return fail(c, GAML_HIP_EINVAL, "node id out of range");
    if (x >= 0) x = c->g.norm[x];
  }
  std::vector<int32_t> begins, ends;
  pacbio_path_string(c, path, nullptr, begins, ends);
  std::vector<std::pair<int32_t, int32_t>> missing;  // graph.cc:2438-2454
  Walk sub;
  for (int32_t i = 0; i < n; i++) {
    sub.clear();
    for (int32_t j = i; j < n; j++) {
      sub.push_back(path[j]);
      if (!sp->walk_id.count(sub)) missing.emplace_back(i, j);
      if ((ends[j] - begins[i]) - (ends[i] - begins[i]) > sp->max_len) break;
    }
  }
  std::sort(missing.begin(), missing.end());
  int32_t out = 0, mb = -1, me = -1;  // merge overlapping index ranges (graph.cc:2455-2478)
  auto emit = [&]() { if (out < cap) { ranges[2 * out] = mb; ranges[2 * out + 1] = me; } out++; };
  for (auto& m : missing) {
    if (mb < 0) { mb = m.first; me = m.second; continue; }
    if (m.first > me) { emit(); mb = m.first; me = m.second; }
    me = std::max(me, m.second);
  }
  if (mb >= 0) emit();
  return out;
}

int gaml_hip_pacbio_ingest_sam(gaml_hip_ctx* c, int readset, const int32_t* path_in, int32_t n, const char* sam, int64_t sam_len,
                               int64_t* filed_out) {
  MULTI_FWD(c, multi_pacbio_ingest_sam(c->multi, readset, path_in, n, sam, sam_len, filed_out));
  PacbioSet* sp = pacbio_of(c, readset);
  if (!sp || !path_in || n <= 0 || sam_len < 0 || (sam_len > 0 && !sam)) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  if (!c->have_graph) return fail(c, GAML_HIP_ESTATE, "no graph");
  PacbioSet& s = *sp;
  if (!s.have_reads) return fail(c, GAML_HIP_ESTATE, "read set was added without bases (use gaml_hip_add_pacbio_reads)");
  if (c->device < 0) return fail(c, GAML_HIP_ENODEVICE, "the alignment DP needs a HIP device: this context is host-only");
  HIP_TRY(c, hipSetDevice(c->device));
  const double t0 = now_us();
  Walk path(path_in, path_in + n);
  for (auto& x : path) {
    if (x >= c->g.n()) return fail(c, GAML_HIP_EINVAL, "node id out of range");
    if (x >= 0) x = c->g.norm[x];  // the scorer normalises before it looks up sub-walks (graph.cc:3180)
  }
  std::string seq;
  std::vector<int32_t> begins, ends;
  pacbio_path_string(c, path, &seq, begins, ends);
  if (2 * (int64_t)seq.size() + 1 > INT32_MAX) return fail(c, GAML_HIP_EINVAL, "path too long");
  const int32_t seq_len = (int32_t)seq.size();
  std::string both;  // path + separator + reverse complement (graph.cc:2687-2688)
  both.resize(2 * (size_t)seq_len + 1);
  {
    char comp[256];  // ReverseBase graph.h:58-64
    for (int k = 0; k < 256; k++) comp[k] = (char)k;
    comp[(unsigned char)'A'] = 'T'; comp[(unsigned char)'C'] = 'G'; comp[(unsigned char)'G'] = 'C'; comp[(unsigned char)'T'] = 'A';
    char* out = &both[0];
    memcpy(out, seq.data(), (size_t)seq_len);
    out[seq_len] = '\n';
    char* rc = out + seq_len + 1;
    for (int32_t i = 0; i < seq_len; i++) rc[i] = comp[(unsigned char)seq[seq_len - 1 - i]];
  }
  // sub-walks this call may file under (graph.cc:2724-2743): new ones get an (empty) cache entry,
  // ones cached before are left alone
  struct SubWalk { int32_t start; int32_t fresh_id; };  // last index it starts at in this path; cache id when this call created the entry, else -1
  std::unordered_map<Walk, SubWalk, WalkHasher> subs;
  {
    Walk sub;
    for (int32_t i = 0; i < n; i++) {
      sub.clear();
      for (int32_t j = i; j < n; j++) {
        sub.push_back(path[j]);
        auto mine = subs.try_emplace(sub, SubWalk{i, -1});
        if (mine.second) {  // first time in this call: new to the cache?
          auto ins = s.walk_id.try_emplace(sub, (int32_t)s.recs.size());
          if (ins.second) { s.recs.emplace_back(); mine.first->second.fresh_id = ins.first->second; }
        } else {
          mine.first->second.start = i;
        }
        if ((ends[j] - begins[i]) - (ends[i] - begins[i]) > s.max_len) break;
      }
    }
    s.generation++;
  }
  // SAM lines -> DP jobs for the records that will be filed (graph.cc:2746-2786). Lines are independent:
  // large inputs are cut at line boundaries into one chunk per host thread; chunk results are joined in
  // order, so records are filed in SAM order as in the reference.
  struct Filed { int32_t walk, pos, pos_end, read_local; };
  struct Chunk {
    std::vector<Filed> filed;
    std::vector<DpJob> jobs;
    std::vector<uint32_t> ops;
    int64_t records = 0, rows = 0;
    int err = 0;
    std::string msg;
    double us[4] = {0, 0, 0, 0};  // trace: record fields + CIGAR, name lookup + filing rule, DP operations, whole chunk
  };
  const int32_t both_len = (int32_t)both.size();
  static const bool trace_host = getenv("GAML_HIP_TRACE_HOST") != nullptr;
  auto parse_chunk = [&](const char* cb, const char* ce, Chunk& out) {
    SamRecord rec;
    const double c_begin = now_us();
    for (const char* p = cb; p < ce;) {
      const char* e = (const char*)memchr(p, '\n', (size_t)(ce - p));
      const char* le = e ? e : ce;
      if (le > p && *p != '@') {
        const double q0 = trace_host ? now_us() : 0;
        if (!parse_sam_record(p, le, both_len, rec)) { out.err = GAML_HIP_EINVAL; out.msg = "SAM line with fewer than 10 columns"; return; }
        if (trace_host) out.us[0] += now_us() - q0;
        out.records++;
        auto id = s.name_id.find(rec.name);
        if (id == s.name_id.end()) { out.err = GAML_HIP_EINVAL; out.msg = "SAM record names a read that is not in the read set: " + rec.name; return; }  // assert graph.cc:2751
        const int32_t ib = (int32_t)(std::lower_bound(ends.begin(), ends.end(), std::max(0, rec.tstart - 5)) - ends.begin());
        const int32_t ie = (int32_t)(std::lower_bound(ends.begin(), ends.end(), std::min(rec.tstart + rec.len + 5, seq_len)) - ends.begin());
        if (ib < n && ie < n && ie >= ib && id->second >= s.lo && id->second < s.hi) {
          thread_local Walk sub;
          sub.assign(path.begin() + ib, path.begin() + ie + 1);
          auto sw = subs.find(sub);
          if (sw != subs.end() && sw->second.start == ib && sw->second.fresh_id >= 0) {
            const int32_t local = (int32_t)(id->second - s.lo);
            const int32_t pos_begin = ib > 0 ? ends[ib - 1] : 0;
            out.filed.push_back(Filed{sw->second.fresh_id, rec.tstart - pos_begin, rec.tend - pos_begin, local});
            DpShape shape;
            DpJob j;
            j.ops_off = (int64_t)out.ops.size();  // chunk-relative until the chunks are joined
            const double q1 = trace_host ? now_us() : 0;
            pacbio_dp_ops(rec.cigar, out.ops, shape);
            if (trace_host) out.us[2] += now_us() - q1;
            j.read_off = s.base_off[local];
            j.read_len = (int32_t)(s.base_off[local + 1] - s.base_off[local]);
            j.scratch_off = 0;
            j.posstart = rec.posstart;
            j.n_ops = shape.n_ops; j.row_f = shape.row_f; j.col_f = shape.col_f; j.bl = shape.bl; j.el = shape.el;
            j.max_width = shape.max_width;
            out.rows += shape.row_f + std::max(shape.el, 1) + 4 + shape.bl;
            out.jobs.push_back(j);
          }
        }
      }
      if (!e) break;
      p = e + 1;
    }
    out.us[3] = now_us() - c_begin;
  };
  const double t_parse0 = now_us();
  const int n_chunks = sam_len < (1 << 20) ? 1 : (int)std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
  std::vector<Chunk> chunks(n_chunks);
  {
    std::vector<const char*> cut(n_chunks + 1, sam + sam_len);
    cut[0] = sam;
    for (int k = 1; k < n_chunks; k++) {  // the next line start at or after the k-th share of the text
      const char* at = sam + sam_len * k / n_chunks;
      const char* nl = at < sam + sam_len ? (const char*)memchr(at, '\n', (size_t)(sam + sam_len - at)) : nullptr;
      cut[k] = nl ? nl + 1 : sam + sam_len;
      if (cut[k] < cut[k - 1]) cut[k] = cut[k - 1];
    }
    std::vector<std::thread> pool;
    for (int k = 1; k < n_chunks; k++) pool.emplace_back(parse_chunk, cut[k], cut[k + 1], std::ref(chunks[k]));
    parse_chunk(cut[0], cut[1], chunks[0]);
    for (auto& th : pool) th.join();
  }
  const double t_join0 = now_us();
  std::vector<Filed> filed;
  std::vector<DpJob> jobs;
  std::vector<std::pair<const uint32_t*, size_t>> ops;  // the chunks' operation lists go to the device one after the other, unjoined
  int64_t scratch = 0, records = 0, cells = 0, rows = 0, ops_total = 0;
  for (Chunk& ch : chunks) {
    if (ch.err) return fail(c, ch.err, ch.msg);
    records += ch.records; rows += ch.rows;
    const int64_t ops_base = ops_total;
    ops.emplace_back(ch.ops.data(), ch.ops.size());
    ops_total += (int64_t)ch.ops.size();
    filed.insert(filed.end(), ch.filed.begin(), ch.filed.end());
    for (DpJob j : ch.jobs) {
      j.ops_off += ops_base;
      j.scratch_off = scratch;
      scratch += 2 * dp_row_stride(j.max_width);
      jobs.push_back(j);
    }
  }
  const double t1 = now_us();
  if (trace_host) {
    double f = 0, o = 0, w = 0;
    for (Chunk& ch : chunks) { f += ch.us[0]; o += ch.us[2]; w = std::max(w, ch.us[3]); }
    fprintf(stderr, "pacbio ingest: %d chunks; path string + sub-walks %.1f ms, parse (slowest chunk) %.1f ms [all chunks: record fields + CIGAR %.1f, DP operations %.1f], join %.1f ms\n",
            n_chunks, (t_parse0 - t0) * 1e-3, w * 1e-3, f * 1e-3, o * 1e-3, (t1 - t_join0) * 1e-3);
  }
  float kernel_ms = 0;
  std::vector<double> logp(jobs.size());
  if (!jobs.empty()) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!s.bases_uploaded) {
      HIP_TRY(c, s.d_bases.reserve(std::max<size_t>(1, s.bases.size())));
      HIP_TRY(c, hipMemcpy(s.d_bases.p, s.bases.data(), s.bases.size(), hipMemcpyHostToDevice));
      s.bases_uploaded = true;
    }
    if (int e = run_pacbio_dp(c, s.dp, both, s.d_bases.as<unsigned char>(), jobs, ops, scratch, s.log_match, s.log_mismatch, logp.data(), &kernel_ms, &cells))
      return e;
  }
  for (size_t i = 0; i < filed.size(); i++) {
    gaml_pacbio_aligment r;
    r.position = filed[i].pos; r.position_end = filed[i].pos_end; r.read_id = filed[i].read_local; r.logprob = logp[i];
    s.recs[filed[i].walk].push_back(r);
  }
  s.generation++;
  if (filed_out) *filed_out = (int64_t)filed.size();
  s.dp_stats[0] = (double)records; s.dp_stats[1] = (double)jobs.size(); s.dp_stats[2] = (double)rows;
  s.dp_stats[3] = (double)cells; s.dp_stats[4] = kernel_ms; s.dp_stats[5] = (t1 - t0) * 1e-3; s.dp_stats[6] = (now_us() - t1) * 1e-3;
  s.dp_stats[7] = (double)scratch * 8;
  return GAML_HIP_OK;
}

int32_t gaml_hip_debug_sam_band(const char* sam_line, int64_t len, int32_t total_len, int32_t* fields10, int32_t* row0, int32_t* lo,
                                int32_t* hi, int32_t cap) {
  if (!sam_line || !fields10 || !row0) return GAML_HIP_EINVAL;
  SamRecord a;
  if (!parse_sam_record(sam_line, sam_line + len, total_len, a)) return GAML_HIP_EINVAL;
  const int32_t f[10] = {a.flags, a.len, a.posstart, a.posend, a.sstart, a.send, a.slen, a.tstart, a.tend, a.edit_dist};
  memcpy(fields10, f, sizeof(f));
  DpBand b;
  pacbio_dp_band(a.cigar, b);
  *row0 = b.row0;
  const int32_t n = (int32_t)b.lo.size();
  if (n <= cap && lo && hi) { memcpy(lo, b.lo.data(), n * sizeof(int32_t)); memcpy(hi, b.hi.data(), n * sizeof(int32_t)); }
  return n;
}

int gaml_hip_debug_sam_logprob(gaml_hip_ctx* c, const char* target, int32_t target_len, const char* read, int32_t read_len,
                               const char* sam_line, int64_t sam_len, double mismatch_prob, double* logprob_out, int32_t* band_lo,
                               int32_t* band_hi, int32_t band_cap) {
  if (!c || !target || target_len <= 0 || !read || read_len < 0 || !sam_line || !logprob_out || band_cap < 0)
    return fail(c, GAML_HIP_EINVAL, "bad arguments");
  if (c->device < 0) return fail(c, GAML_HIP_ENODEVICE, "the alignment DP needs a HIP device: this context is host-only");
  SamRecord rec;
  if (!parse_sam_record(sam_line, sam_line + sam_len, target_len, rec)) return fail(c, GAML_HIP_EINVAL, "SAM line with fewer than 10 columns");
  std::vector<uint32_t> ops;
  DpShape shape;
  pacbio_dp_ops(rec.cigar, ops, shape);
  DpJob j;
  j.read_off = 0; j.ops_off = 0; j.scratch_off = 0; j.read_len = read_len; j.posstart = rec.posstart;
  j.n_ops = shape.n_ops; j.row_f = shape.row_f; j.col_f = shape.col_f; j.bl = shape.bl; j.el = shape.el; j.max_width = shape.max_width;
  const int32_t r_first = shape.bl > 0 ? -shape.bl : 0;
  const int32_t r_last = std::max(std::max(shape.row_f, shape.row_f + shape.el - 1), shape.bl > 0 ? 2 : 0);
  const int32_t n_rows = r_last - r_first + 5;
  const bool want_band = band_lo && band_hi && band_cap >= n_rows;
  DpDev dev;
  DevBuf d_read;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, d_read.reserve(std::max(1, read_len)));
  HIP_TRY(c, hipMemcpy(d_read.p, read, read_len, hipMemcpyHostToDevice));
  float ms = 0;
  int e = run_pacbio_dp(c, dev, std::string(target, target + target_len), d_read.as<unsigned char>(), std::vector<DpJob>(1, j),
                        std::vector<std::pair<const uint32_t*, size_t>>(1, std::make_pair((const uint32_t*)ops.data(), ops.size())),
                        2 * dp_row_stride(shape.max_width), std::log(1.0 - 4 * mismatch_prob), std::log(mismatch_prob), logprob_out, &ms, nullptr,
                        band_lo, band_hi, want_band ? n_rows : 0);
  dev.release();
  d_read.release();
  return e ? e : n_rows;
}

int gaml_hip_debug_sam_shape(const char* sam_line, int64_t len, int32_t total_len, int32_t* out6, uint32_t* ops, int32_t cap) {
  if (!sam_line || !out6) return GAML_HIP_EINVAL;
  SamRecord a;
  if (!parse_sam_record(sam_line, sam_line + len, total_len, a)) return GAML_HIP_EINVAL;
  std::vector<uint32_t> v;
  DpShape sh;
  pacbio_dp_ops(a.cigar, v, sh);
  const int32_t f[6] = {sh.n_ops, sh.row_f, sh.col_f, sh.bl, sh.el, sh.max_width};
  memcpy(out6, f, sizeof(f));
  if (ops && cap >= sh.n_ops) memcpy(ops, v.data(), v.size() * sizeof(uint32_t));
  return sh.n_ops;
}

int gaml_hip_pacbio_dp_stats(gaml_hip_ctx* c, int readset, double* out8) {
  MULTI_SHARD0(c);
  PacbioSet* sp = pacbio_of(c, readset);
  if (!sp || !out8) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  memcpy(out8, sp->dp_stats, sizeof(sp->dp_stats));
  return GAML_HIP_OK;
}

int64_t gaml_hip_pacbio_records(gaml_hip_ctx* c, int readset, const int32_t* subpath, int32_t len, gaml_pacbio_aligment* out, int64_t cap) {
  MULTI_FWD(c, multi_pacbio_records(c->multi, readset, subpath, len, out, cap));
  PacbioSet* sp = pacbio_of(c, readset);
  if (!sp || !subpath || len <= 0 || cap < 0 || (cap > 0 && !out)) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  auto it = sp->walk_id.find(Walk(subpath, subpath + len));
  if (it == sp->walk_id.end()) return -1;
  const auto& v = sp->recs[it->second];
  for (int64_t i = 0; i < (int64_t)v.size() && i < cap; i++) { out[i] = v[i]; out[i].read_id += (int32_t)sp->lo; }
  return (int64_t)v.size();
}

int gaml_hip_eval_begin(gaml_hip_ctx* c, const int32_t* paths, const int64_t* offs, int32_t n_paths, int64_t* pending_out,
                        int32_t* total_len_out) {
  if (!c) return GAML_HIP_EINVAL;
  MULTI_REFUSE(c, "the gaml_hip_eval_* protocol is for one shard per process; gaml_hip_calc_prob runs it over all shards");
  int e = eval_begin(c, paths, offs, n_paths, pending_out);
  if (!e && total_len_out) *total_len_out = c->pending_total_len;
  return e;
}

int64_t gaml_hip_eval_pending_maxpos(gaml_hip_ctx* c, int32_t* out, int64_t cap) {
  if (!c || c->multi) return -1;
  int64_t n = 0;
  for (ShortMate* m : filter_mates(c))
    for (int32_t wid : m->unsynced) { if (out && n < cap) out[n] = m->wins[wid].max_pos; n++; }
  return n;
}

int gaml_hip_eval_apply_maxpos(gaml_hip_ctx* c, const int32_t* reduced, int64_t n) {
  if (!c || (n > 0 && !reduced)) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  MULTI_REFUSE(c, "the gaml_hip_eval_* protocol is for one shard per process");
  int64_t have = 0;
  for (ShortMate* m : filter_mates(c)) have += (int64_t)m->unsynced.size();
  if (have != n) return fail(c, GAML_HIP_EINVAL, "count does not match gaml_hip_eval_pending_maxpos");
  int64_t k = 0;
  for (ShortMate* m : filter_mates(c)) {
    for (int32_t wid : m->unsynced) {
      if (reduced[k] < m->wins[wid].max_pos) return fail(c, GAML_HIP_EINVAL, "reduced maximum below the local one: not an all-reduce(max)?");
      m->wins[wid].global_max_pos = reduced[k++];
    }
    m->unsynced.clear();
  }
  for (auto& ps : c->paireds) ps->planner.invalidate_thresholds();
  return GAML_HIP_OK;
}

// The stream a caller names. NULL is the legacy default stream -- what `torch.cuda.current_stream().cuda_stream` is on
// torch's default stream -- and is used as such: the library's work is then ordered against everything else the caller
// has on that stream, like any other handle. (Until round 3 NULL selected the context's PRIVATE stream: three contexts
// of one process handed NULL worked on three unrelated streams, unordered against the caller's copies -- a sharded
// coverage penalty came out as 2771 bad bases instead of 694, silently.) gaml_hip_sync covers it.
static hipStream_t caller_stream(gaml_hip_ctx* c, void* stream) {
  if (!stream) c->used_default_stream = true;
  return (hipStream_t)stream;
}

int gaml_hip_eval_finish_async(gaml_hip_ctx* c, void* d_partials, void* stream) {
  if (!c || !d_partials) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  MULTI_REFUSE(c, "the gaml_hip_eval_* protocol is for one shard per process");
  if (c->device >= 0) HIP_TRY(c, hipSetDevice(c->device));
  return eval_finish(c, d_partials, caller_stream(c, stream));
}

int32_t gaml_hip_eval_score_async(gaml_hip_ctx* c, void* d_partials, void* stream) {
  if (!c || !d_partials) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  MULTI_REFUSE(c, "the gaml_hip_eval_* protocol is for one shard per process");
  if (c->device >= 0) HIP_TRY(c, hipSetDevice(c->device));
  c->pending_cov.clear();
  c->pending_pb.clear();
  c->defer_cov = c->peers > 1;
  const int e = eval_finish(c, d_partials, caller_stream(c, stream));
  c->defer_cov = false;
  if (e) { c->pending_cov.clear(); c->pending_pb.clear(); return e; }
  return (int32_t)c->pending_cov.size();
}

int32_t gaml_hip_eval_pacbio_pending(gaml_hip_ctx* c) { return c ? (int32_t)c->pending_pb.size() : GAML_HIP_EINVAL; }

int64_t gaml_hip_eval_pacbio_intervals(gaml_hip_ctx* c, int32_t i) {
  if (!c || i < 0 || i >= (int32_t)c->pending_pb.size()) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  return c->pending_pb[i].n_own;
}

int gaml_hip_eval_pacbio_export_async(gaml_hip_ctx* c, int32_t i, void* dst, int64_t cap, void* stream) {
  if (!c || i < 0 || i >= (int32_t)c->pending_pb.size() || cap < 0 || (cap > 0 && !dst)) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  const gaml_hip_ctx::PendingPacbio& pb = c->pending_pb[i];
  if (cap < pb.n_own) return fail(c, GAML_HIP_EINVAL, "the intervals do not fit the destination");
  HIP_TRY(c, hipSetDevice(c->device));
  PacbioSet& s = *c->pacbios[pb.pacbio_idx];
  if (pb.n_own > 0)
    HIP_TRY(c, hipMemcpyAsync(dst, s.sweep.all.as<int4>() + pb.n_node, (size_t)pb.n_own * sizeof(int4), hipMemcpyDeviceToDevice, caller_stream(c, stream)));
  return GAML_HIP_OK;
}

int gaml_hip_eval_pacbio_finish_async(gaml_hip_ctx* c, int32_t i, const void* intervals, int64_t n_intervals, int32_t contribute, void* stream) {
  if (!c || i < 0 || i >= (int32_t)c->pending_pb.size() || n_intervals < 0 || (n_intervals > 0 && !intervals) || n_intervals > ((int64_t)1 << 28))
    return fail(c, GAML_HIP_EINVAL, "bad arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t st = caller_stream(c, stream);
  const gaml_hip_ctx::PendingPacbio& pb = c->pending_pb[i];
  PacbioSet& s = *c->pacbios[pb.pacbio_idx];
  PbSweepDev& d = s.sweep;
  const size_t want = (size_t)std::max<int64_t>(1, pb.n_node + n_intervals) * sizeof(int4);
  if (want > d.all.cap) {  // the node intervals move along
    DevBuf bigger;
    HIP_TRY(c, hipStreamSynchronize(st));
    HIP_TRY(c, bigger.reserve(want));
    if (pb.n_node) HIP_TRY(c, hipMemcpy(bigger.p, d.all.p, (size_t)pb.n_node * sizeof(int4), hipMemcpyDeviceToDevice));
    d.all.release();
    d.all = bigger;
  }
  // all ranks' alignment intervals behind the node intervals (this rank's own are among them)
  if (n_intervals > 0) HIP_TRY(c, hipMemcpyAsync(d.all.as<int4>() + pb.n_node, intervals, (size_t)n_intervals * sizeof(int4), hipMemcpyDeviceToDevice, st));
  // every rank computes the same bad_bases; one of them contributes it to the all-reduce(sum) of the partials
  return pacbio_sweep_run(c, s, pb.n_node + n_intervals, pb.n_paths, st, pb.out4, contribute ? 1.0 : 0.0);
}

int gaml_hip_eval_coverage_export_async(gaml_hip_ctx* c, int32_t i, void* dst, int64_t cap, int64_t* bytes_out, void* stream) {
  if (!c || i < 0 || i >= (int32_t)c->pending_cov.size() || !bytes_out) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  const int64_t bytes = (int64_t)c->pending_cov[i].args.total_words * 4;
  *bytes_out = bytes;
  if (!dst) return GAML_HIP_OK;  // size query
  if (cap < bytes) return fail(c, GAML_HIP_EINVAL, "coverage map does not fit the destination");
  HIP_TRY(c, hipSetDevice(c->device));
  if (bytes > 0) HIP_TRY(c, hipMemcpyAsync(dst, c->pending_cov[i].args.bits, (size_t)bytes, hipMemcpyDeviceToDevice, caller_stream(c, stream)));
  return GAML_HIP_OK;
}

int gaml_hip_eval_coverage_finish_async(gaml_hip_ctx* c, int32_t i, const void* maps, int32_t n_maps, int32_t contribute, void* stream) {
  if (!c || i < 0 || i >= (int32_t)c->pending_cov.size() || n_maps < 0 || (n_maps > 0 && !maps)) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t st = caller_stream(c, stream);
  const gaml_hip_ctx::PendingCov& pc = c->pending_cov[i];
  PairedSet& s = *c->paireds[pc.paired_idx];
  if (pc.args.total_words > 0) {
    if (n_maps > 0) {
      hipLaunchKernelGGL(or_maps_kernel, dim3(grid_for(pc.args.total_words)), dim3(kBlock), 0, st, s.cov_bits.as<uint32_t>(),
                         (const uint32_t*)maps, n_maps, pc.args.total_words);
      HIP_TRY(c, hipGetLastError());
    }
    hipLaunchKernelGGL(coverage_sweep_kernel, dim3(grid_for(pc.args.total_words)), dim3(kBlock), 0, st, pc.args);
    HIP_TRY(c, hipGetLastError());
  }
  // every rank computes the same bad_bases; one of them contributes it to the all-reduce(sum) of the partials
  hipLaunchKernelGGL(store_bad_bases_kernel, dim3(1), dim3(64), 0, st, s.bad.as<unsigned long long>(), pc.out4, contribute ? 1.0 : 0.0);
  HIP_TRY(c, hipGetLastError());
  return GAML_HIP_OK;
}

// ---------------------------------------------------------------------------------------------
// Host-side exchange for one node (the hot path's one collective is a sum of 4 doubles per read set).
// A blocking evaluation leaves a rank's partials in host memory; adding them up across the ranks of a node through
// shared memory costs ~1 us, against ~30 us for device partials -> finisher kernel -> RCCL all-reduce -> fetch.
// Block layout: [parity 0 | parity 1] x [rank] x {sequence word, 7 words pad, cap doubles}; a rank publishes its
// values and then the step number; everybody adds the slots in rank order (same bits on every rank). A rank can be
// at most one step ahead of the slowest (it needs everyone's step-k values to finish step k): two parities suffice.
// ---------------------------------------------------------------------------------------------
static size_t shm_slot_bytes(int cap) { return 64 + (((size_t)cap * sizeof(double) + 63) & ~(size_t)63); }

int gaml_hip_shm_exchange_open(gaml_hip_ctx* c, const char* name, int32_t rank, int32_t world, int32_t cap_doubles) {
  if (!c || !name || world < 1 || rank < 0 || rank >= world || cap_doubles < 1) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  MULTI_REFUSE(c, "its shards are summed in-process");
  if (c->shm_base) return fail(c, GAML_HIP_ESTATE, "exchange already open");
  const size_t bytes = 2 * (size_t)world * shm_slot_bytes(cap_doubles);
  if (rank == 0) shm_unlink(name);  // a block left behind by a crashed run would carry old step numbers: rank 0 starts a fresh one (it opens FIRST)
  int fd = shm_open(name, rank == 0 ? (O_RDWR | O_CREAT | O_EXCL) : O_RDWR, 0600);
  if (fd < 0) return fail(c, GAML_HIP_ESTATE, std::string("shm_open ") + name + " failed");
  if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); return fail(c, GAML_HIP_ESTATE, "ftruncate on the shared block failed"); }  // new pages read as zero: sequence 0 = nothing published
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return fail(c, GAML_HIP_ESTATE, "mmap of the shared block failed");
  c->shm_base = (char*)p; c->shm_bytes = bytes; c->shm_rank = rank; c->shm_world = world; c->shm_cap = cap_doubles; c->shm_step = 0; c->shm_name = name;
  return GAML_HIP_OK;
}

int gaml_hip_shm_exchange_close(gaml_hip_ctx* c, int32_t unlink_name) {
  if (!c) return GAML_HIP_EINVAL;
  if (c->shm_base) munmap(c->shm_base, c->shm_bytes);
  if (unlink_name && !c->shm_name.empty()) shm_unlink(c->shm_name.c_str());
  c->shm_base = nullptr; c->shm_bytes = 0; c->shm_name.clear();
  return GAML_HIP_OK;
}

int gaml_hip_shm_allreduce_sum(gaml_hip_ctx* c, double* inout, int32_t n) {
  if (!c || !inout || n < 1) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  if (!c->shm_base) return fail(c, GAML_HIP_ESTATE, "exchange not open");
  if (n > c->shm_cap) return fail(c, GAML_HIP_EINVAL, "more values than the exchange was opened for");
  const unsigned long long step = ++c->shm_step;
  const size_t slot = shm_slot_bytes(c->shm_cap);
  char* half = c->shm_base + (step & 1) * (size_t)c->shm_world * slot;
  char* mine = half + (size_t)c->shm_rank * slot;
  memcpy(mine + 64, inout, (size_t)n * sizeof(double));
  __atomic_store_n((unsigned long long*)mine, step, __ATOMIC_RELEASE);
  const double t0 = now_us();
  for (int r = 0; r < c->shm_world; r++) {
    const unsigned long long* seq = (const unsigned long long*)(half + (size_t)r * slot);
    unsigned spins = 0;
    while (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != step) {
      if ((++spins & 1023) == 0 && now_us() - t0 > 30e6) return fail(c, GAML_HIP_ESTATE, "shared-memory exchange: a rank did not arrive within 30 s");
      __builtin_ia32_pause();
    }
  }
  for (int k = 0; k < n; k++) inout[k] = 0.0;
  for (int r = 0; r < c->shm_world; r++) {  // rank order on every rank: identical sums
    const double* v = (const double*)(half + (size_t)r * slot + 64);
    for (int k = 0; k < n; k++) inout[k] += v[k];
  }
  return GAML_HIP_OK;
}

// Device results -> host without a D2H copy command and without the runtime's completion wake-up: a one-block kernel
// writes the values and then a sequence word into mapped pinned memory; the host polls the word.
__global__ void fetch_kernel(const double* src, int n, double* dst, unsigned long long* seq_word, unsigned long long seq) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence_system(); *(volatile unsigned long long*)seq_word = seq; }
}

int gaml_hip_fetch_async(gaml_hip_ctx* c, const void* d_src, int32_t n_doubles, void* stream) {
  if (!c || !d_src || n_doubles <= 0) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  MULTI_REFUSE(c, "no single device to fetch from");
  if (c->device < 0) return fail(c, GAML_HIP_ENODEVICE, "no device");
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t bytes = 512 + (size_t)n_doubles * sizeof(double);
  if (bytes > c->fetch_host.cap) {
    HIP_TRY(c, c->fetch_host.reserve(bytes));
    memset(c->fetch_host.p, 0, c->fetch_host.cap);
  }
  c->fetch_stream = caller_stream(c, stream);
  c->fetch_seq++;
  char* dev = (char*)c->fetch_host.dev;
  hipLaunchKernelGGL(fetch_kernel, dim3(1), dim3(64), 0, c->fetch_stream, (const double*)d_src, (int)n_doubles, (double*)(dev + 512),
                     (unsigned long long*)dev, c->fetch_seq);
  HIP_TRY(c, hipGetLastError());
  return GAML_HIP_OK;
}

int gaml_hip_fetch_wait(gaml_hip_ctx* c, double* out, int32_t n_doubles) {
  if (!c || !out || n_doubles <= 0 || !c->fetch_host.p) return fail(c, GAML_HIP_EINVAL, "bad arguments / no fetch in flight");
  volatile unsigned long long* word = (volatile unsigned long long*)c->fetch_host.p;
  const double t0 = now_us();
  bool seen = false;
  while (now_us() - t0 < 2000.0) {  // bounded spin, then the runtime's wait
    for (int k = 0; k < 64 && !seen; k++) seen = *word == c->fetch_seq;
    if (seen) break;
  }
  if (!seen) {
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->fetch_stream));
    if (*word != c->fetch_seq) return fail(c, GAML_HIP_ESTATE, "fetch kernel finished without publishing its sequence word");
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  memcpy(out, (const char*)c->fetch_host.p + 512, (size_t)n_doubles * sizeof(double));
  return GAML_HIP_OK;
}

int gaml_hip_sync(gaml_hip_ctx* c) {
  MULTI_FWD(c, multi_sync(c->multi));
  if (!c || c->device < 0) return fail(c, GAML_HIP_ENODEVICE, "no device");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->used_default_stream) { HIP_TRY(c, hipStreamSynchronize(nullptr)); c->used_default_stream = false; }  // (an *_async call was handed NULL)
  return GAML_HIP_OK;
}

static int fetch_partials(gaml_hip_ctx* c, double* partials_out);

int gaml_hip_eval_finish(gaml_hip_ctx* c, double* partials_out) {
  if (!c || !partials_out) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  MULTI_REFUSE(c, "the gaml_hip_eval_* protocol is for one shard per process");
  if (c->device < 0) return fail(c, GAML_HIP_ENODEVICE, "scoring needs a HIP device: this context is host-only");
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t bytes = std::max<size_t>(1, c->handles.size()) * 4 * sizeof(double);
  HIP_TRY(c, c->packed_host.reserve(bytes));
  void* dres = nullptr;
  HIP_TRY(c, hipHostGetDevicePointer(&dres, c->packed_host.p, 0));
  c->host_results = true;
  int e = eval_finish(c, dres, c->stream);
  c->host_results = false;
  if (e) return e;
  return fetch_partials(c, partials_out);
}

int gaml_hip_calc_partials_async(gaml_hip_ctx* c, const int32_t* paths, const int64_t* offs, int32_t n_paths,
                                 void* d_partials, void* stream, int32_t* total_len_out) {
  if (!c || !d_partials) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  MULTI_REFUSE(c, "partials of several devices have no single device address; use gaml_hip_calc_partials");
  if (c->device >= 0) HIP_TRY(c, hipSetDevice(c->device));
  return evaluate(c, paths, offs, n_paths, d_partials, caller_stream(c, stream), total_len_out);
}

int gaml_hip_calc_partials(gaml_hip_ctx* c, const int32_t* paths, const int64_t* offs, int32_t n_paths,
                           double* partials_out, int32_t* total_len_out) {
  if (!c || !partials_out) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  MULTI_FWD(c, multi_calc_partials(c->multi, paths, offs, n_paths, partials_out, total_len_out));  // reduced over the shards
  if (c->device < 0) return fail(c, GAML_HIP_ENODEVICE, "scoring needs a HIP device: this context is host-only");
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t bytes = std::max<size_t>(1, c->handles.size()) * 4 * sizeof(double);
  HIP_TRY(c, c->packed_host.reserve(bytes));
  void* dres = nullptr;
  HIP_TRY(c, hipHostGetDevicePointer(&dres, c->packed_host.p, 0));
  c->host_results = true;
  int e = evaluate(c, paths, offs, n_paths, dres, c->stream, total_len_out);
  c->host_results = false;
  if (e) return e;
  return fetch_partials(c, partials_out);
}

// the one-block finisher kernel's summation order, on the host: lane t adds partials t, t+256, ...;
// each 64-lane wave folds by halving strides (what __shfl_down does); wave results are added in order
static void finisher_order_sum(const double* ps, const int* pz, int n, double* sum_out, double* zeros_out) {
  double v[kBlock];
  long long z = 0;
  for (int t = 0; t < kBlock; t++) { double a = 0; for (int b = t; b < n; b += kBlock) a += ps[b]; v[t] = a; }
  for (int b = 0; b < n; b++) z += pz[b];
  double total = 0;
  for (int w = 0; w < kBlock / 64; w++) {
    double* l = v + 64 * w;
    for (int off = 32; off > 0; off >>= 1) for (int i = 0; i < off; i++) l[i] += l[i + off];
    total += l[0];
  }
  *sum_out = total;
  *zeros_out = (double)z;
}

// Blocking calls: the scoring kernels wrote their per-block partials into pinned host memory. When paired scorers
// without coverage penalty are all there is, those partials ARE the result: spin on their sentinels instead of
// waiting for the stream's completion signal (saves the runtime's ~6 us wake-up, tools/latency_probe.hip). Bounded;
// any doubt -> a real stream sync. Returns whether the spin sufficed.
static int wait_host_partials(gaml_hip_ctx* c, bool* spun) {
  const double t0 = now_us();
  bool spin = c->knobs[7] == 0 && !c->handles.empty();
  for (auto& h : c->handles) if (h.kind != 1) spin = false;
  for (auto& ps : c->paireds) if (ps->cfg.penalty_constant > 0 || !ps->last_host_partials || ps->last_total_blocks == 0) spin = false;
  if (spin) {
    const double deadline = t0 + 2e6;
    for (auto& ps : c->paireds) {
      for (int k = 0; k < ps->last_sets && spin; k++) {
        const volatile double* hs = (const volatile double*)ps->h_part_sum.p + (size_t)k * ps->host_part_stride;
        const volatile int* hz = (const volatile int*)ps->h_part_zero.p + (size_t)k * ps->host_part_stride;
        int done = 0;
        while (done < ps->last_blocks[k]) {
          if (hs[done] == hs[done] && hz[done] != INT_MIN) { done++; continue; }  // NaN != NaN
          __builtin_ia32_pause();
          if ((done & 63) == 0 && now_us() > deadline) { spin = false; break; }
        }
      }
      if (!spin) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  if (!spin) HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->t_dev_wall_us += now_us() - t0;
  c->prof[7] = now_us() - t0;  // wait for the device
  *spun = spin;
  return 0;
}

static int fetch_partials(gaml_hip_ctx* c, double* partials_out) {
  bool spun = false;
  if (int e = wait_host_partials(c, &spun)) return e;
  {
    double* res = (double*)c->packed_host.p;
    auto ord = scoring_order(c);
    for (size_t k = 0; k < ord.size(); k++) {
      if (ord[k].kind != 1) continue;
      PairedSet& s = *c->paireds[ord[k].idx];
      if (!s.last_host_partials) continue;
      if (s.last_blocks[0] > 0)
        finisher_order_sum((const double*)s.h_part_sum.p, (const int*)s.h_part_zero.p, s.last_blocks[0], &res[4 * k], &res[4 * k + 1]);
      else res[4 * k] = res[4 * k + 1] = 0;
      if (!(s.cfg.penalty_constant > 0) || s.last_total_blocks == 0) res[4 * k + 2] = 0;  // else store_bad_bases_kernel wrote it
      res[4 * k + 3] = (double)s.mate[0].n_local();
    }
  }
  memcpy(partials_out, c->packed_host.p, c->handles.size() * 4 * sizeof(double));
  c->t_kernel_us = 0;
  if (!spun) { if (int e2 = collect_events(c)) return e2; }  // after a spin the events are collected lazily (gaml_hip_kernel_stats)
  // bookkeeping for gaml_hip_bad_bases
  auto order = scoring_order(c);
  for (size_t k = 0; k < order.size(); k++) {
    if (order[k].kind == 1) c->paireds[order[k].idx]->last_bad_bases = (int64_t)partials_out[4 * k + 2];
    else if (order[k].kind == 2) c->pacbios[order[k].idx]->last_bad_bases = (int64_t)partials_out[4 * k + 2];
  }
  return GAML_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// gaml_hip_calc_prob_batch, fast path: up to kMaxSets path sets in ONE pass over the records of every paired set
// (paired_score_multi_kernel). The host plans the sets one after the other straight into consecutive regions of one
// arena slot; then one launch per read set, one wait. Contexts with other kinds of read sets, a coverage penalty or
// without a memo take the sequential path below (same results).
// ---------------------------------------------------------------------------------------------------------
static bool batch_fast_capable(const gaml_hip_ctx* c) {
  if (c->handles.empty() || c->knobs[11] == 1) return false;  // knob 11 = 1: force the sequential path (A/B, tools/)
  for (auto& h : c->handles) if (h.kind != 1) return false;
  for (auto& ps : c->paireds) if (!paired_multi_capable(c, *ps)) return false;
  return true;
}

// The same with the sets' tables built on the device (batch_tables_kernel): on a large-BAR device the resident copy
// of the tables mirrors the previous call's path set, and candidates differ from it -- and from each other -- in a
// few dozen entries. Returns 1 when this chunk cannot go that way (tables rebuilt as a whole, list changes, growth
// past the resident capacities): the caller takes the full-tables route, which plans the chunk again.
static int batch_chunk_patched(gaml_hip_ctx* c, int n, const int32_t* paths, const int64_t* offs, const int32_t* set_offs,
                               double* partials_out, int32_t* tls) {
  hipStream_t st = c->stream;
  const size_t nps = c->paireds.size();
  if (!c->direct_write || c->knobs[8] != 0 || c->knobs[13] != 0 || c->knobs[11] == 2) return 1;  // knob 11 = 2: full tables per set (A/B)
  constexpr size_t kPatchCap = 8192;  // entries per read set and batch
  struct PerSet {
    int slot = 0; char* wp = nullptr; size_t stride = 0;
    PairedLayout L; std::vector<PairedLayout> Ls; std::vector<PairedPrep> prep;
    std::vector<int> patch_off; size_t n_patches = 0;
    size_t tail_fixed = 0, chg_bytes[2] = {0, 0};
    int launches = 0;
    std::vector<int32_t> touched[2];  // union of the changed entries: the resident copy follows after the batch
  };
  std::vector<PerSet> per(nps);
  c->host_results = true;
  struct Reset { gaml_hip_ctx* c; ~Reset() { c->host_results = false; c->pending_open = false; } } reset{c};
  for (size_t i = 0; i < nps; i++) {
    PairedSet& ps = *c->paireds[i];
    if (int e = prepare_paired_tables(c, ps)) return e;
    PairedSet::Persist& P = ps.persist;
    // bring the copy up to the images (made here if no blocking call has yet; entries changed by a call that did not go through it)
    if (!P.valid || ps.image[0].changed_all || ps.image[1].changed_all || !ps.image[0].changed.empty() || !ps.image[1].changed.empty() ||
        ps.image[0].lists_changed || ps.image[1].lists_changed) {
      if (int e = paired_persist_update(c, ps, 2.0, st)) return e;
    }
    PerSet& r = per[i];
    r.stride = align16(P.bytes);
    r.L.tfloor_off = P.off_tfloor;
    r.L.l0 = OccLayout{P.off_occ[0], P.off_lo[0], P.off_m[0], P.off_lo[1] /* unused */};
    r.L.l1 = OccLayout{P.off_occ[1], P.off_lo[1], P.off_m[1], P.bytes};
    r.L.pb_off = r.L.so_off = r.L.st_off = 0; r.L.total = P.bytes;
    r.Ls.assign((size_t)n, r.L);
    r.prep.resize((size_t)n);
    r.patch_off.assign(2 * (size_t)n + 1, 0);
    // behind the regions: the patches, their offsets, and per launch and mate one byte per table entry (MultiSets::chg)
    r.chg_bytes[0] = align16(P.cap_w[0]); r.chg_bytes[1] = align16(P.cap_w[1]);
    r.tail_fixed = align16(kPatchCap * sizeof(BatchPatch)) + align16((2 * (size_t)kMaxSets + 1) * sizeof(int));
    const size_t bytes = r.stride * (size_t)n + r.tail_fixed + 2 * (r.chg_bytes[0] + r.chg_bytes[1]);
    if (int e = arena_acquire(c, ps.arena, bytes, st, &r.slot, &r.wp)) return e;
  }
  auto give_up = [&](bool in_flight) -> int {  // the resident copies no longer mirror the images: rewritten as a whole next time
    for (size_t i = 0; i < nps; i++) c->paireds[i]->persist.valid = false;
    if (in_flight) { bool spun = false; (void)wait_host_partials(c, &spun); if (!spun) (void)collect_events(c); }
    return 1;
  };
  const int half = n > 4 ? (n + 1) / 2 : n;  // (where the batch is cut makes no measurable difference: 13.1-14.8 us per set for 1+7 .. 6+2)
  int launched = 0;
  auto launch_upto = [&](int upto) -> int {
    for (size_t i = 0; i < nps; i++) {
      PairedSet& ps = *c->paireds[i];
      PerSet& r = per[i];
      const PairedSet::Persist& P = ps.persist;
      if (int e = paired_sync_tables(c, ps, st)) return e;
      for (int k = launched; k < upto; k++) paired_pack_thresholds(ps, r.L, (double)(2 * (tls[k] == 0 ? 1 : tls[k])), r.wp + (size_t)k * r.stride);
      char* tail = r.wp + r.stride * (size_t)n;
      int* d_off = (int*)(tail + align16(kPatchCap * sizeof(BatchPatch)));
      memcpy(d_off, r.patch_off.data(), (2 * (size_t)upto + 1) * sizeof(int));
      if (int e = arena_commit(c, ps.arena, r.slot, 0, st)) return e;  // (direct route: drains the write-combining buffers)
      BatchTabArgs ta;
      ta.base = (const char*)P.dev;
      ta.regions = (char*)ps.arena.dev[r.slot];
      ta.stride = r.stride;
      for (int mt = 0; mt < 2; mt++) {
        ta.off_occ[mt] = P.off_occ[mt]; ta.bytes_occ[mt] = P.cap_w[mt] * sizeof(Occ12);
        ta.off_lo[mt] = P.off_lo[mt]; ta.bytes_lo[mt] = ps.image[mt].multi_off.size() * sizeof(int32_t);
        ta.off_m[mt] = P.off_m[mt]; ta.bytes_m[mt] = ps.image[mt].multi.size() * sizeof(OccQuad);
      }
      ta.patches = (const BatchPatch*)((const char*)ps.arena.dev[r.slot] + r.stride * (size_t)n);
      ta.patch_off = (const int*)((const char*)ta.patches + align16(kPatchCap * sizeof(BatchPatch)));
      ta.first = launched;
      ta.n_sets = upto - launched;
      char* chg0 = (char*)ps.arena.dev[r.slot] + r.stride * (size_t)n + r.tail_fixed + (size_t)(r.launches & 1) * (r.chg_bytes[0] + r.chg_bytes[1]);
      ta.chg[0] = (unsigned char*)chg0; ta.chg[1] = (unsigned char*)chg0 + r.chg_bytes[0];
      ta.chg_bytes[0] = r.chg_bytes[0]; ta.chg_bytes[1] = r.chg_bytes[1];
      r.launches++;
      hipLaunchKernelGGL(batch_tables_kernel, dim3((unsigned)(upto - launched) + 1, 2), dim3(1024), 0, st, ta);
      HIP_TRY(c, hipGetLastError());
      const unsigned char* chg[2] = {ta.chg[0], ta.chg[1]};
      if (int e = launch_paired_multi(c, ps, launched, upto - launched, r.Ls.data(), r.prep.data(), tls, (const char*)ps.arena.dev[r.slot], r.stride, st,
                                      c->knobs[11] == 3 ? nullptr : chg)) return e;  // knob 11 = 3: every set resolves every pair (A/B)
    }
    launched = upto;
    return 0;
  };
  for (int k = 0; k < n; k++) {
    int64_t pending = 0;
    if (int e = eval_begin(c, paths, offs + set_offs[k], set_offs[k + 1] - set_offs[k], &pending)) return e;
    tls[k] = c->pending_total_len;
    for (size_t i = 0; i < nps; i++) {
      PairedSet& ps = *c->paireds[i];
      PerSet& r = per[i];
      const PairedSet::Persist& P = ps.persist;
      prepare_paired_tables_host(c, ps, r.prep[(size_t)k]);
      OccImage* im = ps.image;
      bool ok = !im[0].changed_all && !im[1].changed_all && !im[0].lists_changed && !im[1].lists_changed;
      for (int mt = 0; mt < 2 && ok; mt++) ok = im[mt].occ12.size() <= P.cap_w[mt] && r.n_patches + im[mt].changed.size() <= kPatchCap;
      if (!ok) {
        if (getenv("GAML_HIP_TRACE_HOST"))
          fprintf(stderr, "batch set %d: not a patch (all %d %d, lists %d %d, windows %zu/%zu %zu/%zu, patches %zu + %zu + %zu)\n", k, (int)im[0].changed_all, (int)im[1].changed_all,
                  (int)im[0].lists_changed, (int)im[1].lists_changed, im[0].occ12.size(), P.cap_w[0], im[1].occ12.size(), P.cap_w[1], r.n_patches, im[0].changed.size(), im[1].changed.size());
        c->pending_open = false;
        return give_up(launched > 0);
      }
      BatchPatch* dp = (BatchPatch*)(r.wp + r.stride * (size_t)n);
      for (int mt = 0; mt < 2; mt++) {
        for (int32_t w : im[mt].changed) {
          const Occ12& o = im[mt].occ12[w];
          dp[r.n_patches++] = BatchPatch{w, o.lo, o.hi, o.rank};
          r.touched[mt].push_back(w);
        }
        r.patch_off[2 * (size_t)k + mt + 1] = (int)r.n_patches;
        im[mt].take_changed();
      }
    }
    c->pending_open = false;
    if (k + 1 == half && half < n) { if (int e = launch_upto(half)) return e; }
  }
  if (int e = launch_upto(n)) return e;
  if (getenv("GAML_HIP_TRACE_HOST")) {
    fprintf(stderr, "batch of %d sets, patch entries per set (mate 1 + mate 2):", n);
    for (int k = 0; k < n; k++) fprintf(stderr, " %d+%d", per[0].patch_off[2 * k + 1] - per[0].patch_off[2 * k], per[0].patch_off[2 * k + 2] - per[0].patch_off[2 * k + 1]);
    fprintf(stderr, "\n");
  }
  bool spun = false;
  if (int e = wait_host_partials(c, &spun)) return e;
  if (!spun) { if (int e2 = collect_events(c)) return e2; }
  // the device is done with the resident copies: they follow the images (now the last set's)
  for (size_t i = 0; i < nps; i++) {
    PairedSet& ps = *c->paireds[i];
    char* occ[2] = {(char*)ps.persist.dev + ps.persist.off_occ[0], (char*)ps.persist.dev + ps.persist.off_occ[1]};
    for (int mt = 0; mt < 2; mt++)
      for (int32_t w : per[i].touched[mt]) memcpy(occ[mt] + (size_t)w * sizeof(Occ12), &ps.image[mt].occ12[w], sizeof(Occ12));
  }
  _mm_sfence();
  for (size_t i = 0; i < nps; i++) c->paireds[i]->batches_patched++;
  for (int k = 0; k < n; k++)
    for (size_t i = 0; i < nps; i++) {
      PairedSet& ps = *c->paireds[i];
      double* out = partials_out + ((size_t)k * nps + i) * 4;
      out[0] = out[1] = out[2] = 0;
      if (ps.last_blocks[k] > 0)
        finisher_order_sum((const double*)ps.h_part_sum.p + (size_t)k * ps.host_part_stride, (const int*)ps.h_part_zero.p + (size_t)k * ps.host_part_stride,
                           ps.last_blocks[k], &out[0], &out[1]);
      out[3] = (double)ps.mate[0].n_local();
      ps.last_bad_bases = 0;
    }
  return 0;
}

// returns 1 when a set's tables did not fit the region reserved for it (the caller falls back for this chunk)
static int batch_chunk_fast(gaml_hip_ctx* c, int n, const int32_t* paths, const int64_t* offs, const int32_t* set_offs,
                            double* partials_out, int32_t* tls) {
  hipStream_t st = c->stream;
  const size_t nps = c->paireds.size();
  struct PerSet { int slot = 0; char* wp = nullptr; size_t stride = 0, cap_w[2] = {0, 0}; std::vector<PairedLayout> L; std::vector<PairedPrep> prep; };
  std::vector<PerSet> per(nps);
  c->host_results = true;
  struct Reset { gaml_hip_ctx* c; ~Reset() { c->host_results = false; c->pending_open = false; } } reset{c};
  for (size_t i = 0; i < nps; i++) {
    PairedSet& ps = *c->paireds[i];
    if (int e = prepare_paired_tables(c, ps)) return e;
    // a region per path set: the occurrence images of the current window count plus room for windows and lists that
    // this very batch adds
    // windows the batch itself may add: every set's tables are padded to this many entries per mate
    per[i].cap_w[0] = ps.mate[0].wins.size() + 256 + ps.batch_slack / 24;
    per[i].cap_w[1] = ps.mate[1].wins.size() + 256 + ps.batch_slack / 24;
    const size_t lists = 2 * (sizeof(int32_t) * (ps.image[0].multi_off.size() + ps.image[1].multi_off.size()) + sizeof(OccQuad) * (ps.image[0].multi.size() + ps.image[1].multi.size()));
    const size_t est = 4096 + 12 * (per[i].cap_w[0] + per[i].cap_w[1]) + 16384 + lists + ps.batch_slack;
    per[i].stride = align16(est);
    if (int e = arena_acquire(c, ps.arena, per[i].stride * (size_t)n, st, &per[i].slot, &per[i].wp)) return e;
    per[i].L.resize((size_t)n);
    per[i].prep.resize((size_t)n);
  }
  // the batch goes out in two launches: the host plans the second half while the device scores the first
  const int half = n > 4 ? (n + 1) / 2 : n;  // (where the batch is cut makes no measurable difference: 13.1-14.8 us per set for 1+7 .. 6+2)
  int launched = 0;
  auto launch_upto = [&](int upto) -> int {
    for (size_t i = 0; i < nps; i++) {
      PairedSet& ps = *c->paireds[i];
      if (int e = paired_sync_tables(c, ps, st)) return e;
      for (int k = launched; k < upto; k++) paired_pack_thresholds(ps, per[i].L[(size_t)k], (double)(2 * (tls[k] == 0 ? 1 : tls[k])), per[i].wp + (size_t)k * per[i].stride);
      // (the staged route copies the regions written so far; the direct route only drains the write-combining buffers)
      if (int e = arena_commit(c, ps.arena, per[i].slot, per[i].stride * (size_t)upto, st)) return e;
      if (int e = launch_paired_multi(c, ps, launched, upto - launched, per[i].L.data(), per[i].prep.data(), tls, (const char*)ps.arena.dev[per[i].slot], per[i].stride, st)) return e;
    }
    launched = upto;
    return 0;
  };
  for (int k = 0; k < n; k++) {
    int64_t pending = 0;
    if (int e = eval_begin(c, paths, offs + set_offs[k], set_offs[k + 1] - set_offs[k], &pending)) return e;
    tls[k] = c->pending_total_len;
    for (size_t i = 0; i < nps; i++) {
      PairedSet& ps = *c->paireds[i];
      PairedPrep& p = per[i].prep[(size_t)k];
      prepare_paired_tables_host(c, ps, p);
      bool fits = ps.mate[0].wins.size() <= per[i].cap_w[0] && ps.mate[1].wins.size() <= per[i].cap_w[1];
      if (fits) { per[i].L[(size_t)k] = paired_layout(ps, p, per[i].cap_w); fits = per[i].L[(size_t)k].total <= per[i].stride; }
      if (!fits) {  // the tables outgrew the region reserved per set: the sequential path takes this chunk (after what is in flight)
        ps.batch_slack += 24 * 16384 + 2 * per[i].stride;
        if (launched > 0) { bool spun = false; (void)wait_host_partials(c, &spun); if (!spun) (void)collect_events(c); }
        return 1;
      }
      paired_pack(ps, p, per[i].L[(size_t)k], per[i].wp + (size_t)k * per[i].stride);
    }
    c->pending_open = false;
    if (k + 1 == half && half < n) { if (int e = launch_upto(half)) return e; }
  }
  if (int e = launch_upto(n)) return e;
  bool spun = false;
  if (int e = wait_host_partials(c, &spun)) return e;
  if (!spun) { if (int e2 = collect_events(c)) return e2; }
  for (size_t i = 0; i < nps; i++) c->paireds[i]->batches_full++;
  for (int k = 0; k < n; k++)
    for (size_t i = 0; i < nps; i++) {
      PairedSet& ps = *c->paireds[i];
      double* out = partials_out + ((size_t)k * nps + i) * 4;
      out[0] = out[1] = out[2] = 0;
      if (ps.last_blocks[k] > 0)
        finisher_order_sum((const double*)ps.h_part_sum.p + (size_t)k * ps.host_part_stride, (const int*)ps.h_part_zero.p + (size_t)k * ps.host_part_stride,
                           ps.last_blocks[k], &out[0], &out[1]);
      out[3] = (double)ps.mate[0].n_local();
      ps.last_bad_bases = 0;
    }
  return 0;
}

int gaml_hip_combine_partials(gaml_hip_ctx* c, const double* partials, int32_t total_len, double* prob_out, int32_t* zeros_out) {
  if (!c || !partials || !prob_out) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  MULTI_FWD(c, multi_combine(c->multi, partials, total_len, prob_out, zeros_out));
  return combine(c, partials, prob_out, zeros_out, total_len);
}

int gaml_hip_calc_prob(gaml_hip_ctx* c, const int32_t* paths, const int64_t* offs, int32_t n_paths,
                       double* prob_out, int32_t* zeros_out, int32_t* total_len_out) {
  if (!c || !prob_out) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  const size_t ns = c->multi ? (size_t)gaml::multi_num_readsets(c->multi) : c->handles.size();
  std::vector<double> partials(std::max<size_t>(1, ns) * 4);
  int32_t tl = 0;
  int e;
  if (!c->multi && c->comm) {
    // one shard per process with a communicator (gaml_hip_comm_init_rank): registration, maxima, kernels and the
    // one all-reduce(sum) of the partials over RCCL, all in here -- every rank calls with the same paths
    e = gaml::comm_eval_reduced(c, paths, offs, n_paths, partials.data(), &tl);
  } else {
    if (c->peers != 1)
      return fail(c, GAML_HIP_ESTATE, "sharded context without a communicator: gaml_hip_comm_init_rank first, or use gaml_hip_calc_partials + "
                                      "your own all-reduce + gaml_hip_combine_partials");
    e = gaml_hip_calc_partials(c, paths, offs, n_paths, partials.data(), &tl);
  }
  if (e) return e;
  if (total_len_out) *total_len_out = tl;
  return gaml_hip_combine_partials(c, partials.data(), tl, prob_out, zeros_out);
}

int gaml_hip_calc_prob_batch(gaml_hip_ctx* c, int32_t n_sets, const int32_t* paths, const int64_t* offs, const int32_t* set_offs,
                             double* probs_out, int32_t* zeros_out, int32_t* total_lens_out) {
  if (!c || n_sets < 0 || !set_offs || !probs_out || (n_sets > 0 && !offs)) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  MULTI_FWD(c, multi_calc_prob_batch(c->multi, n_sets, paths, offs, set_offs, probs_out, zeros_out, total_lens_out));
  if (c->comm && n_sets > 0) {  // one shard per process with a communicator: ONE all-reduce over the whole batch
    const size_t nsr = std::max<size_t>(1, c->handles.size());
    std::vector<double> part((size_t)n_sets * 4 * nsr);
    std::vector<int32_t> tls((size_t)n_sets, 0);
    if (int e = gaml::comm_eval_reduced_batch(c, n_sets, paths, offs, set_offs, part.data(), tls.data())) return e;
    for (int32_t i = 0; i < n_sets; i++) {
      if (int e = combine(c, part.data() + (size_t)i * 4 * nsr, &probs_out[i], zeros_out ? zeros_out + (size_t)i * 2 * nsr : nullptr, tls[i])) return e;
      if (total_lens_out) total_lens_out[i] = tls[i];
    }
    return GAML_HIP_OK;
  }
  if (c->world != 1 || c->peers != 1)
    return fail(c, GAML_HIP_ESTATE, "sharded context without a communicator: gaml_hip_comm_init_rank first, or run the gaml_hip_eval_* protocol per path "
                                    "set and all-reduce the partials of the batch at once");
  if (c->device < 0) return fail(c, GAML_HIP_ENODEVICE, "scoring needs a HIP device: this context is host-only");
  if (n_sets == 0) return GAML_HIP_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t ns = std::max<size_t>(1, c->handles.size());
  for (int32_t i = 0; i < n_sets; i++) if (set_offs[i + 1] < set_offs[i]) return fail(c, GAML_HIP_EINVAL, "set offsets must not decrease");
  int32_t done_sets = 0;
  if (batch_fast_capable(c)) {
    std::vector<double> part((size_t)kMaxSets * 4 * ns);
    int32_t tls[kMaxSets];
    while (done_sets < n_sets) {
      const int n = std::min<int32_t>(kMaxSets, n_sets - done_sets);
      int rc = n > 1 ? batch_chunk_patched(c, n, paths, offs, set_offs + done_sets, part.data(), tls) : 1;
      if (rc > 0 && n > 1) rc = batch_chunk_fast(c, n, paths, offs, set_offs + done_sets, part.data(), tls);
      if (rc < 0) return rc;
      if (rc > 0) break;  // a single leftover set, or tables that outgrew their region: the sequential path takes the rest
      for (int k = 0; k < n; k++) {
        const int32_t i = done_sets + k;
        if (int e = combine(c, part.data() + (size_t)k * 4 * ns, &probs_out[i], zeros_out ? zeros_out + (size_t)i * 2 * ns : nullptr, tls[k])) return e;
        if (total_lens_out) total_lens_out[i] = tls[k];
      }
      done_sets += n;
    }
    if (done_sets == n_sets) return GAML_HIP_OK;
    // (the sequential path below handles sets [done_sets, n_sets))
    set_offs += done_sets; probs_out += done_sets;  // (paths / offs stay: set_offs indexes into them)
    if (zeros_out) zeros_out += (size_t)done_sets * 2 * ns;
    if (total_lens_out) total_lens_out += done_sets;
    n_sets -= done_sets;
  }
  const size_t doubles = (size_t)n_sets * 4 * ns;
  if (doubles * sizeof(double) > c->batch_dev.cap) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, c->batch_dev.reserve(doubles * sizeof(double))); }
  HIP_TRY(c, c->batch_host.reserve(doubles * sizeof(double)));
  std::vector<int32_t> tls(n_sets, 0);
  // every evaluation is enqueued behind the previous one on the library's stream; the host prepares
  // set i+1 (window placement, occurrence tables, staging) while the device scores set i
  for (int32_t i = 0; i < n_sets; i++) {
    const int32_t p0 = set_offs[i], p1 = set_offs[i + 1];
    if (p1 < p0) return fail(c, GAML_HIP_EINVAL, "set offsets must not decrease");
    // path offsets of the set stay absolute: eval_begin reads paths[offs[k] .. offs[k+1])
    if (int e = evaluate(c, paths, offs + p0, p1 - p0, c->batch_dev.as<double>() + (size_t)i * 4 * ns, c->stream, &tls[i])) {
      (void)hipStreamSynchronize(c->stream);
      return e;
    }
  }
  HIP_TRY(c, hipMemcpyAsync(c->batch_host.p, c->batch_dev.p, doubles * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const double* res = (const double*)c->batch_host.p;
  for (int32_t i = 0; i < n_sets; i++) {
    if (int e = combine(c, res + (size_t)i * 4 * ns, &probs_out[i], zeros_out ? zeros_out + (size_t)i * 2 * ns : nullptr, tls[i])) return e;
    if (total_lens_out) total_lens_out[i] = tls[i];
  }
  return GAML_HIP_OK;
}

int gaml_hip_num_readsets(const gaml_hip_ctx* c) { MULTI_FWD(c, multi_num_readsets(c->multi)); return c ? (int)c->handles.size() : 0; }
int gaml_hip_readset_kind(const gaml_hip_ctx* c, int rs) {
  MULTI_FWD(c, multi_readset_kind(c->multi, rs));
  return (c && rs >= 0 && rs < (int)c->handles.size()) ? c->handles[rs].kind : -1;
}
int64_t gaml_hip_readset_reads(const gaml_hip_ctx* c, int rs) {
  MULTI_FWD(c, multi_readset_reads(c->multi, rs));
  if (!c || rs < 0 || rs >= (int)c->handles.size()) return -1;
  SetRef h = c->handles[rs];
  if (h.kind == 0) return c->singles[h.idx]->mate.n_global;
  if (h.kind == 1) return c->paireds[h.idx]->mate[0].n_global;
  return c->pacbios[h.idx]->n_global;
}
int32_t gaml_hip_num_nodes(const gaml_hip_ctx* c) { MULTI_FWD(c, multi_num_nodes(c->multi)); return c && c->have_graph ? c->g.n() : 0; }
int32_t gaml_hip_node_len(const gaml_hip_ctx* c, int32_t node) { MULTI_FWD(c, multi_node_len(c->multi, node)); return (c && c->have_graph && node >= 0 && node < c->g.n()) ? c->g.len(node) : -1; }

int gaml_hip_read_probs(gaml_hip_ctx* c, int rs, double* out, int64_t n) {
  MULTI_FWD(c, multi_read_probs(c->multi, rs, out, n));
  if (!c || rs < 0 || rs >= (int)c->handles.size() || !out) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  if (c->device < 0) return fail(c, GAML_HIP_ENODEVICE, "host-only context");
  SetRef h = c->handles[rs];
  const void* src; int64_t have;
  if (h.kind == 0) { src = c->singles[h.idx]->probs.p; have = c->singles[h.idx]->mate.n_local(); }
  else if (h.kind == 1) { src = c->paireds[h.idx]->probs.p; have = c->paireds[h.idx]->mate[0].n_local(); }
  else { src = c->pacbios[h.idx]->logprobs.p; have = c->pacbios[h.idx]->hi - c->pacbios[h.idx]->lo; }
  if (!src) return fail(c, GAML_HIP_ESTATE, "read set not scored yet");
  if (n < have) return fail(c, GAML_HIP_EINVAL, "output too small");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipDeviceSynchronize());
  if (have) HIP_TRY(c, hipMemcpy(out, src, have * sizeof(double), hipMemcpyDeviceToHost));
  if (h.kind == 1 && have) {  // device order -> read order
    const auto& ros = c->paireds[h.idx]->pt.read_of_slot;
    std::vector<double> tmp(out, out + have);
    for (int64_t j = 0; j < have; j++) out[ros[j]] = tmp[j];
  }
  return (int)std::min<int64_t>(have, 0x7fffffff);
}

int gaml_hip_bad_bases(gaml_hip_ctx* c, int rs, int64_t* out) {
  MULTI_FWD(c, multi_bad_bases(c->multi, rs, out));
  if (!c || rs < 0 || rs >= (int)c->handles.size() || !out) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  SetRef h = c->handles[rs];
  if (h.kind == 0) *out = 0;  // graph.cc:1701-1733 can never count a base (last_event_type is only -1 or 1)
  else if (h.kind == 1) *out = c->paireds[h.idx]->last_bad_bases;
  else *out = c->pacbios[h.idx]->last_bad_bases;
  return GAML_HIP_OK;
}

int64_t gaml_hip_window_count(const gaml_hip_ctx* c, int rs, int mate) {
  MULTI_FWD(c, multi_window_count(c->multi, rs, mate));
  ShortMate* m = mate_of(const_cast<gaml_hip_ctx*>(c), rs, mate);
  return m ? (int64_t)m->wins.size() : -1;
}

int64_t gaml_hip_window_records(gaml_hip_ctx* c, int rs, int mate, const int32_t* subpath, int32_t len, gaml_aligment* out, int64_t cap) {
  MULTI_FWD(c, multi_window_records(c->multi, rs, mate, subpath, len, out, cap));
  ShortMate* m = mate_of(c, rs, mate);
  if (!m || !subpath || len <= 0) return -2;
  int32_t id = m->find(Walk(subpath, subpath + len));
  if (id < 0) return -1;
  if (m->wins[id].pending) m->flush_pending_cpu(c->g);
  const Window& w = m->wins[id];
  for (int64_t i = 0; i < w.count && i < cap; i++) {
    out[i] = m->pool[w.first + i];
    out[i].read_id += (int32_t)m->lo;
  }
  return w.count;
}

int64_t gaml_hip_align_window(gaml_hip_ctx* c, int rs, int mate, const int32_t* subpath, int32_t len) {
  MULTI_FWD(c, multi_align_window(c->multi, rs, mate, subpath, len));
  ShortMate* m = mate_of(c, rs, mate);
  if (!m || !subpath || len <= 0 || !c->have_graph) return -2;
  int32_t id = m->align(c->g, Walk(subpath, subpath + len));
  if (m->wins[id].pending) {
    SetRef h = c->handles[rs];
    if (h.kind == 1) { if (gpu_align_pending(c, *m, c->paireds[h.idx]->dev[mate].aln)) return -3; }
    else m->flush_pending_cpu(c->g);
  }
  return m->wins[id].count;
}

int gaml_hip_compact_tables(gaml_hip_ctx* c) {
  if (!c) return GAML_HIP_EINVAL;
  MULTI_FWD(c, multi_compact_tables(c->multi));
  for (auto& ps : c->paireds) ps->compact_requested = true;
  return GAML_HIP_OK;
}

int gaml_hip_debug_prepare(gaml_hip_ctx* c, const int32_t* flat, const int64_t* offs, int32_t n_paths) {
  if (!c || n_paths < 0 || (n_paths > 0 && (!flat || !offs))) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  if (c->multi) {  // every shard registers / aligns / places on its own reads (host-only shards included)
    for (int k = 0; k < gaml::multi_num_shards(c->multi); k++)
      if (int e = gaml_hip_debug_prepare(gaml::multi_shard(c->multi, k), flat, offs, n_paths)) return fail(c, e, gaml_hip_last_error(gaml::multi_shard(c->multi, k)));
    return GAML_HIP_OK;
  }
  if (!c->have_graph) return fail(c, GAML_HIP_ESTATE, "no graph set");
  std::vector<Walk> paths = unflatten(flat, offs, n_paths);
  for (auto& h : scoring_order(c)) {
    if (h.kind == 0) { std::vector<Occ> occs; prepare_single_host(c, *c->singles[h.idx], paths, occs); }
    else if (h.kind == 1) {
      PairedPrep p;
      PairedSet& ps = *c->paireds[h.idx];
      if (int e = prepare_paired_structure(c, ps, flat, offs, n_paths)) return e;
      if (int e = align_pending_pair(c, ps)) return e;
      prepare_paired_tables_host(c, ps, p);
    }
  }
  if (c->peers == 1) {
    for (ShortMate* m : filter_mates(c)) m->unsynced.clear();
  }
  return GAML_HIP_OK;
}

int64_t gaml_hip_debug_occurrences(gaml_hip_ctx* c, int rs, int mate, int32_t* out5, int64_t cap) {
  MULTI_SHARD0(c);
  if (!c || rs < 0 || rs >= (int)c->handles.size()) return -1;
  SetRef h = c->handles[rs];
  const std::vector<Occ>* v = nullptr;
  if (h.kind == 0) v = &c->singles[h.idx]->last_occ;
  else if (h.kind == 1 && (mate == 0 || mate == 1)) { PairedSet& ps = *c->paireds[h.idx]; ps.planner.flat_occurrences(mate, ps.scratch_occ[mate]); v = &ps.scratch_occ[mate]; }
  if (!v) return -1;
  for (int64_t i = 0; i < (int64_t)v->size() && i < cap; i++) {
    const Occ& o = (*v)[i];
    out5[5 * i] = o.wid; out5[5 * i + 1] = o.shift; out5[5 * i + 2] = o.min_pos; out5[5 * i + 3] = o.path; out5[5 * i + 4] = o.rank;
  }
  return (int64_t)v->size();
}

int64_t gaml_hip_debug_table_occurrences(gaml_hip_ctx* c, int rs, int mate, int32_t* out5, int64_t cap, int64_t* info3) {
  MULTI_SHARD0(c);
  if (!c || rs < 0 || rs >= (int)c->handles.size() || c->handles[rs].kind != 1 || (mate != 0 && mate != 1)) return -1;
  PairedSet& ps = *c->paireds[c->handles[rs].idx];
  if (info3) { info3[0] = ps.planner.last_was_incremental(); info3[1] = (int64_t)ps.planner.incremental_calls; info3[2] = (int64_t)ps.planner.full_calls; }
  std::vector<Occ> v;
  ps.image[mate].dump(v);
  const std::vector<int32_t>& slots = ps.planner.slots();
  std::unordered_map<int32_t, int32_t> pos;
  for (size_t k = 0; k < slots.size(); k++) pos[slots[k]] = (int32_t)k;
  for (Occ& o : v) { auto it = pos.find(o.path); o.path = it == pos.end() ? -1 : it->second; }
  std::sort(v.begin(), v.end(), [](const Occ& a, const Occ& b) { return a.path != b.path ? a.path < b.path : (a.rank != b.rank ? a.rank < b.rank : a.wid < b.wid); });
  for (int64_t i = 0; i < (int64_t)v.size() && i < cap; i++) {
    const Occ& o = v[(size_t)i];
    out5[5 * i] = o.wid; out5[5 * i + 1] = o.shift; out5[5 * i + 2] = o.min_pos; out5[5 * i + 3] = o.path; out5[5 * i + 4] = o.rank;
  }
  return (int64_t)v.size();
}

int32_t gaml_hip_debug_window_walk(gaml_hip_ctx* c, int rs, int mate, int32_t wid, int32_t* out, int32_t cap) {
  MULTI_SHARD0(c);
  ShortMate* m = mate_of(c, rs, mate);
  if (!m || wid < 0 || wid >= (int32_t)m->win_walk.size()) return -1;
  const Walk& w = *m->win_walk[wid];
  for (int32_t i = 0; i < (int32_t)w.size() && i < cap; i++) out[i] = w[i];
  return (int32_t)w.size();
}

int gaml_hip_debug_table_stats(gaml_hip_ctx* c, int rs, int64_t* out10) {
  int64_t* out8 = out10;
  MULTI_SHARD0(c);
  if (!c || rs < 0 || rs >= (int)c->handles.size() || c->handles[rs].kind != 1 || !out10) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  PairedSet& s = *c->paireds[c->handles[rs].idx];
  out8[0] = s.full_rebuilds; out8[1] = s.delta_updates; out8[2] = (int64_t)s.dirty.size(); out8[3] = s.async_rebuilds;
  out8[4] = s.batches_patched; out8[5] = s.batches_full;
  out8[6] = s.pt.dropped_records[0]; out8[7] = s.pt.dropped_records[1];
  out10[8] = s.delta_left_out; out10[9] = s.pt.n0a;
  return GAML_HIP_OK;
}

int gaml_hip_aligner_stats(gaml_hip_ctx* c, int64_t* windows, int64_t* candidates, double* microseconds) {
  if (!c) return GAML_HIP_EINVAL;
  MULTI_SHARD0(c);
  if (windows) *windows = c->aln_windows;
  if (candidates) *candidates = c->aln_candidates;
  if (microseconds) *microseconds = c->aln_us;
  if (getenv("GAML_HIP_TRACE_ALIGNER"))
    fprintf(stderr, "aligner: %lld batches; us: strings+upload %.0f, spans+candidates %.0f, extension %.0f, hits D2H %.0f, sort+finalize %.0f\n",
            (long long)c->aln_batches, c->aln_stage_us[0], c->aln_stage_us[1], c->aln_stage_us[2], c->aln_stage_us[3], c->aln_stage_us[4]);
  return GAML_HIP_OK;
}

int gaml_hip_debug_profile(gaml_hip_ctx* c, double* out8) {
  if (!c || !out8) return GAML_HIP_EINVAL;
  MULTI_SHARD0(c);
  for (int i = 0; i < 8; i++) out8[i] = c->prof[i];
  return GAML_HIP_OK;
}

int gaml_hip_debug_timeline(gaml_hip_ctx* c, int rs, unsigned long long* out, int64_t cap_waves) {
  MULTI_SHARD0(c);
  if (!c || rs < 0 || rs >= (int)c->handles.size() || c->handles[rs].kind != 1 || !out) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  PairedSet& s = *c->paireds[c->handles[rs].idx];
  if (!s.h_timeline.p) return 0;
  const int64_t n = std::min<int64_t>(cap_waves, s.timeline_waves);
  memcpy(out, s.h_timeline.p, (size_t)n * 8 * sizeof(unsigned long long));
  return (int)n;
}

int gaml_hip_debug_set_knob(gaml_hip_ctx* c, int knob, int value) {
  if (!c || knob < 0 || knob >= 24) return GAML_HIP_EINVAL;
  if (c->multi) { for (int k = 0; k < gaml::multi_num_shards(c->multi); k++) gaml::multi_shard(c->multi, k)->knobs[knob] = value; return GAML_HIP_OK; }
  c->knobs[knob] = value;
  return GAML_HIP_OK;
}

int gaml_hip_debug_class_counts(gaml_hip_ctx* c, int rs, int64_t* out4) {
  MULTI_SHARD0(c);
  if (!c || rs < 0 || rs >= (int)c->handles.size() || c->handles[rs].kind != 1 || !out4) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  for (int k = 0; k < 4; k++) out4[k] = c->paireds[c->handles[rs].idx]->pt.class_count[k];
  return GAML_HIP_OK;
}

// Host-only check of the record tables' rule "a record that is always overwritten stays out" (host_model.cc
// dominated_records) on the windows that are active now: builds the tables with and without the rule (no device) and
// verifies, record by record, that every pair's records with the rule are the records without it minus records of a
// junction window J for which the first node's own window -- active -- holds a record of the same read at the same
// position. out6 = {records left out mate 1, mate 2, pairs of the compact class with / without the rule, records
// checked, violations}. Returns GAML_HIP_ESTATE when a violation was found.
int gaml_hip_debug_fold_check(gaml_hip_ctx* c, int rs, int64_t* out6) {
  MULTI_SHARD0(c);
  if (!c || rs < 0 || rs >= (int)c->handles.size() || c->handles[rs].kind != 1 || !out6) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  PairedSet& s = *c->paireds[c->handles[rs].idx];
  PairTables with, without;
  build_pair_tables(s.mate[0], s.mate[1], with, true);
  build_pair_tables(s.mate[0], s.mate[1], without, false);
  int64_t checked = 0, bad = 0;
  const int64_t n = s.mate[0].n_local();
  PairedSet::RecList a, b;
  for (int mt = 0; mt < 2; mt++) {
    const ShortMate& m = s.mate[mt];
    // per (window, read, position): is it a record of an active single-node window?
    for (int64_t read = 0; read < n; read++) {
      a = PairedSet::RecList(); b = PairedSet::RecList();
      paired_base_records(with, with.slot_of_read[read], mt, a);
      paired_base_records(without, without.slot_of_read[read], mt, b);
      size_t ia = 0;
      for (size_t ib = 0; ib < b.size(); ib++) {
        checked++;
        const RecQuad& r = b[ib];
        if (ia < a.size() && a[ia].wid == r.wid && a[ia].pos == r.pos && a[ia].flags == r.flags) { ia++; continue; }
        // left out: must be a junction window whose first node's own window holds (read, position)
        const Window& j = m.wins[r.wid];
        bool ok = false;
        if (j.head >= 0) {
          auto it = m.solo_of_node.find(j.head);
          if (it != m.solo_of_node.end() && m.wins[it->second].active)
            for (size_t q = 0; q < b.size(); q++) ok = ok || (b[q].wid == it->second && b[q].pos == r.pos);
        }
        bad += !ok;
      }
      bad += ia != a.size();  // (a record with the rule that the tables without it do not hold)
    }
  }
  out6[0] = with.dropped_records[0]; out6[1] = with.dropped_records[1];
  out6[2] = with.class_count[0]; out6[3] = without.class_count[0];
  out6[4] = checked; out6[5] = bad;
  return bad ? fail(c, GAML_HIP_ESTATE, "record tables: a record was left out that is not always overwritten") : GAML_HIP_OK;
}

// Host-only check of the static memo indices (PairTables::static_idx): tables of the windows that are active now,
// every compact-class pair looked at again from the window cache -- the two records' windows compared by their node
// walks, orientation rule and insert distance recomputed (graph.cc:1864-1876). out8 = {pairs with a static index, other
// compact-class pairs, violations (an index that differs, or a pair that qualifies and has none), then why the other
// pairs have none: a mate without record, records in different windows, orientation rule, distance outside the
// insert-size table, edit count / length code outside the memo}.
int gaml_hip_debug_static_check(gaml_hip_ctx* c, int rs, int64_t* out8) {
  MULTI_SHARD0(c);
  if (!c || rs < 0 || rs >= (int)c->handles.size() || c->handles[rs].kind != 1 || !out8) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  PairedSet& s = *c->paireds[c->handles[rs].idx];
  if (int e = paired_host_tabs(c, s)) return e;
  const int ins_n = (int)s.ins_tab.size();
  link_mate_windows(s.mate[0], s.mate[1]);
  PairTables pt;
  build_pair_tables(s.mate[0], s.mate[1], pt, true, ins_n);
  for (int k = 0; k < 8; k++) out8[k] = 0;
  const int64_t n0 = pt.class_count[0];
  out8[0] = pt.n0a; out8[1] = n0 - pt.n0a;
  const int codes = (int)std::min<size_t>(pt.len_combo.size(), kMemoCodes);
  for (int64_t slot = 0; slot < n0; slot++) {
    const uint64_t r1 = pt.rec8[0][slot], r2 = pt.rec8[1][slot];
    const int32_t read = pt.read_of_slot[slot];
    int why = 0;  // 0: qualifies
    int32_t idx = -1;
    if (r1 == kNoRec8 || r2 == kNoRec8) { idx = kStaticZero; }  // never scores: static, too
    else {
      const int32_t w1 = (int32_t)(r1 & 0xffffff), w2 = (int32_t)(r2 & 0xffffff);
      if (*s.mate[0].win_walk[w1] != *s.mate[1].win_walk[w2]) why = 4;
      else {
        const int32_t p1 = (int32_t)((r1 >> 24) & 0xfffffff), p2 = (int32_t)((r2 >> 24) & 0xfffffff);
        const int32_t e1 = (int32_t)((r1 >> 52) & 63), e2 = (int32_t)((r2 >> 52) & 63), o1 = (int32_t)((r1 >> 58) & 1), o2 = (int32_t)((r2 >> 58) & 1);
        const int32_t L1 = s.mate[0].lens[read], L2 = s.mate[1].lens[read];
        int32_t dist = -1;
        if (o1 != o2) {  // graph.cc:1864-1876 on window positions (both alignments get the window's shift)
          if (p1 < p2) { if (o1 == 0 && o2 == 1) dist = p2 - p1 + L2; }
          else if (o1 == 1 && o2 == 0) dist = p1 - p2 + L1;
        }
        const int lc = pt.len_code[slot];
        if (dist < 0 && !(o1 != o2 && ((p1 < p2 && o1 == 0) || (p1 >= p2 && o1 == 1)))) why = 5;
        else if (dist < 0 || dist >= ins_n) why = 6;
        else if (e1 >= 7 || e2 >= 7 || lc >= codes) why = 7;
        else idx = ((lc * 7 + e1) * 7 + e2) * ins_n + dist;
      }
    }
    if (slot < pt.n0a) out8[2] += (why != 0 || idx != pt.static_idx[slot]);
    else { out8[2] += why == 0; if (why) out8[why]++; }
  }
  return out8[2] ? fail(c, GAML_HIP_ESTATE, "record tables: a static memo index is wrong or missing") : GAML_HIP_OK;
}

// per-block partial sums of the last blocking evaluation of paired read set rs (path set `set` of a batch launch; 0 for a
// single call), in block order [lane-per-pair classes | wave-per-pair blocks | paired_general_kernel blocks]: which
// block's sum differs when two routes that should agree bit for bit do not. Returns the number of blocks.
int32_t gaml_hip_debug_block_partials(gaml_hip_ctx* c, int rs, int32_t set, double* sums, int32_t* zeros, int32_t cap, int32_t* layout8) {
  MULTI_SHARD0(c);
  if (!c || rs < 0 || rs >= (int)c->handles.size() || c->handles[rs].kind != 1) return GAML_HIP_EINVAL;
  PairedSet& s = *c->paireds[c->handles[rs].idx];
  if (!s.last_host_partials || set < 0 || set >= kMaxSets) return 0;
  const int n = s.last_blocks[set];
  const double* hs = (const double*)s.h_part_sum.p + (size_t)set * s.host_part_stride;
  const int* hz = (const int*)s.h_part_zero.p + (size_t)set * s.host_part_stride;
  for (int b = 0; b < n && b < cap; b++) { if (sums) sums[b] = hs[b]; if (zeros) zeros[b] = hz[b]; }
  if (layout8) {
    PairedArgs a; GridPlan gp;
    paired_base_args(c, s, a, gp);
    layout8[0] = gp.blocks0a; layout8[1] = gp.blocks0; layout8[2] = a.blocks01; layout8[3] = a.blocks012; layout8[4] = a.main_blocks; layout8[5] = a.total_blocks;
    layout8[6] = gp.gen_blocks; layout8[7] = n;
  }
  return n;
}

// device time of paired_general_kernel (the second launch of a path set in which some window occurs several times),
// from events attached to its dispatches while event timing is on; reset with gaml_hip_kernel_stats(reset)
int gaml_hip_debug_general_stats(gaml_hip_ctx* c, int64_t* launches, double* device_us) {
  MULTI_SHARD0(c);
  if (!c) return GAML_HIP_EINVAL;
  if (c->device >= 0 && c->ev_used) { if (int e = collect_events(c)) return e; }
  if (launches) *launches = c->stat_general_launches;
  if (device_us) *device_us = c->stat_general_us;
  return GAML_HIP_OK;
}

int gaml_hip_last_timing(const gaml_hip_ctx* c, double* out3) {
  if (!c || !out3) return GAML_HIP_EINVAL;
  MULTI_FWD(c, multi_last_timing(c->multi, out3));
  out3[0] = c->t_host_us; out3[1] = c->t_dev_wall_us; out3[2] = c->t_kernel_us;
  return GAML_HIP_OK;
}
int gaml_hip_set_event_timing(gaml_hip_ctx* c, int on) {
  if (!c) return GAML_HIP_EINVAL;
  MULTI_FWD(c, multi_set_event_timing(c->multi, on));
  c->event_timing = on != 0;
  c->event_every = on > 1 ? on : 1;
  c->event_tick = 0;
  if (c->event_timing && c->device >= 0) {
    // event pairs are collected lazily (gaml_hip_kernel_stats); create a pool up front so that no
    // hipEventCreate lands inside a caller's timed region
    HIP_TRY(c, hipSetDevice(c->device));
    while (c->ev_pool.size() < 2048) {
      hipEvent_t a, b;
      HIP_TRY(c, hipEventCreate(&a));
      HIP_TRY(c, hipEventCreate(&b));
      c->ev_pool.emplace_back(a, b);
    }
  }
  return GAML_HIP_OK;
}
int gaml_hip_kernel_stats(gaml_hip_ctx* c, int reset, int64_t* launches, double* device_us, double* algo_bytes) {
  if (!c) return GAML_HIP_EINVAL;
  MULTI_FWD(c, multi_kernel_stats(c->multi, reset, launches, device_us, algo_bytes));
  if (c->device >= 0 && c->ev_used) { if (int e = collect_events(c)) return e; }  // async calls leave pairs pending
  if (launches) *launches = c->stat_launches;
  if (device_us) *device_us = c->stat_device_us;
  if (algo_bytes) *algo_bytes = c->stat_algo_bytes;
  if (reset) { c->stat_launches = 0; c->stat_device_us = 0; c->stat_algo_bytes = 0; c->stat_general_us = 0; c->stat_general_launches = 0; }
  return GAML_HIP_OK;
}

}  // extern "C"
