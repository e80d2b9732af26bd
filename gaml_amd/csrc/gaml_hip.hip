// gaml_hip.hip -- context, device memory, launches and the C ABI of libgaml_hip.so.
// The kernels are in kernels.hip.h, the host data model in host_model.{h,cc}.
#include "ctx.hip.h"

using namespace gaml;
using namespace gaml::detail;

namespace {

// The scoring launch takes ~600 bytes of kernel arguments; with the argument segment in HOST memory (the HIP runtime's
// default) every wave's first scalar loads cross PCIe: 12.9 instead of 9.05 us per launch. HIP_FORCE_DEV_KERNARG=1 puts the
// segment into device memory; the runtime reads it when it initialises -- at the process's first HIP call, which cannot
// have happened when this library is loaded at program start (a gaml binary linked against it). An explicit setting of
// the caller's is left alone (gaml_hip_create then says so once).
namespace {
struct KernargEnv {
  bool was_set;
  KernargEnv() { was_set = getenv("HIP_FORCE_DEV_KERNARG") != nullptr; setenv("HIP_FORCE_DEV_KERNARG", "1", 0); }
} kernarg_env;
}  // namespace

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
#ifdef GAML_HIP_DEV
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
#endif  // GAML_HIP_DEV

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
  if (KNOB(c, 8) == 1 || (bytes & 15) || bytes > ((size_t)1 << 30)) {
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
    sum += ms * 1000.0;
    last = c->ev_call[i] == c->ev_call[c->ev_used - 1] ? last + ms * 1000.0 : 0.0;
  }
  if (c->ev_used) c->t_kernel_us = last;
  c->stat_device_us += sum;
  c->ev_used = 0;
  return 0;
}

int take_events(gaml_hip_ctx* c, std::pair<hipEvent_t, hipEvent_t>** out) {
  if (c->ev_used == c->ev_pool.size() && c->ev_pool.size() >= 2048) { if (int e = collect_events(c)) return e; }
  if (c->ev_used == c->ev_pool.size()) {
    hipEvent_t a, b;
    HIP_TRY(c, hipEventCreate(&a));
    HIP_TRY(c, hipEventCreate(&b));
    c->ev_pool.emplace_back(a, b);
  }
  if (c->ev_call.size() < c->ev_pool.size()) c->ev_call.resize(c->ev_pool.size(), 0);
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

#include "single_launch.hip.h"
#include "pacbio_launch.hip.h"
#include "aligner_launch.hip.h"
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
    gpu_probe(c->stream, c->warm_buf.p, "before alignment");
    if (int e = align_pending_pair(c, *c->paireds[i])) return e;
    gpu_probe(c->stream, c->warm_buf.p, "after alignment");
    c->prof[2] = (i ? c->prof[2] : 0.0) + now_us() - ta;  // alignment of newly registered windows (inside pass 1)
  }
  int64_t n = 0;
  for (ShortMate* m : filter_mates(c)) {
    if (c->peers == 1) m->unsynced.clear();  // nothing to exchange: global == local
    n += (int64_t)m->unsynced.size();
  }
  for (auto& s : c->singles) s->mate.unsynced.clear();  // single-end scoring has no position filter
  if (pending) *pending = n;
  gpu_probe(c->stream, c->warm_buf.p, "end of eval_begin");
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
  static const bool trace = getenv("GAML_HIP_TRACE_HOST") != nullptr;
  const double t0 = trace ? now_us() : 0.0;
  if (int e = eval_begin(c, flat, offs, n_paths, &pending)) return e;
  if (total_len_out) *total_len_out = c->pending_total_len;
  const double t1 = trace ? now_us() : 0.0;
  const int e = eval_finish(c, d_partials, st);
  if (trace && now_us() - t0 > 5000.0) fprintf(stderr, "evaluation: begin (paths, windows, alignment) %.2f ms, finish (tables, launches) %.2f ms\n", (t1 - t0) * 1e-3, (now_us() - t1) * 1e-3);
  return e;
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
void ctx_set_status(gaml_hip_ctx* c, double* dst, double a, double b) { c->status_dst = dst; c->status_a = a; c->status_b = b; c->status_done = false; }
bool ctx_status_done(const gaml_hip_ctx* c) { return c->status_done; }
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
    {
      const char* ka = getenv("HIP_FORCE_DEV_KERNARG");
      static bool warned = false;
      if ((!ka || strcmp(ka, "1") != 0) && !warned) {
        warned = true;
        c->err = "HIP_FORCE_DEV_KERNARG is not 1: kernel arguments live in host memory, every scoring launch pays ~4 us for it";
        fprintf(stderr, "libgaml_hip: warning: %s\n", c->err.c_str());
      }
    }
    {  // the evaluations' stream: highest priority (a table build runs beside it on a stream of the lowest, paired_tables.hip.h)
      int lo = 0, hi = 0;
      if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi) != hipSuccess) return GAML_HIP_EHIP;
    }
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
#ifdef GAML_GEN_STAMPS
  if (c && c->device >= 0) {
    unsigned long long z[32];
    if (hipMemcpyFromSymbol(z, HIP_SYMBOL(gaml::g_gen_stamp), sizeof(z)) == hipSuccess && z[15]) {
      fprintf(stderr, "general_pair_staged: longest stage us: entries %.1f, bounds %.1f, lists %.1f, liveness %.1f, terms %.1f; waves above 10 us in a stage: %llu %llu %llu %llu %llu\n",
              z[16] * 0.01, z[17] * 0.01, z[18] * 0.01, z[19] * 0.01, z[20] * 0.01, z[24], z[25], z[26], z[27], z[28]);
    }
    if (z[6]) fprintf(stderr, "general_pair_call as the caller sees it: %llu calls, mean %.2f us; from the class body's entry to the call %.2f us\n", z[6], z[5] * 0.01 / z[6], z[21] * 0.01 / z[6]);
    if (z[15])
      fprintf(stderr, "general_pair_staged: %llu waves; mean us per wave: entries %.2f, bounds %.2f, lists %.2f, liveness %.2f, terms %.2f; candidates per lane-0 pair %.1f / %.1f, live %.1f, lanes %.1f; pairs that did not fit %llu; longest %.2f us, most candidates %llu / %llu, most live %llu / %llu\n",
              z[15], z[0] * 0.01 / z[15], z[1] * 0.01 / z[15], z[2] * 0.01 / z[15], z[3] * 0.01 / z[15], z[4] * 0.01 / z[15], (double)z[8] / z[15], (double)z[9] / z[15], (double)z[10] / z[15], (double)z[11] / z[15], z[12], z[13] * 0.01, z[14] / 1000, z[14] % 1000, z[7] / 1000, z[7] % 1000);
  }
#endif
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
      if (s->rebuild.stream) (void)hipStreamDestroy(s->rebuild.stream);
      if (s->rebuild.done) (void)hipEventDestroy(s->rebuild.done);
      if (s->rebuild.mark) (void)hipEventDestroy(s->rebuild.mark);
      s->tab.release(); s->rebuild.tab.release(); s->scratch.release();
      for (int m = 0; m < 2; m++) { s->dev[m].first.release(); s->dev[m].extra.release(); s->dev[m].pows.release(); s->dev[m].aln.release(); s->dev[m].pool.release(); s->dev[m].lens.release();
                                    s->dl_rec[m].release(); s->sp_rng[m].release(); s->sp_rec[m].release(); }
      s->dl_slot.release(); s->dl_spill.release(); s->sp_slot.release(); s->dstate.release(); s->dl_bins.release(); s->dl_bin_count.release(); s->dl_blk_tot.release(); s->dl_wlist.release(); s->h_dstate.release(); s->lcode.release(); s->len_combo_dev.release(); s->combo_tabs.release(); s->memo.release();
      drop_stage(s->stage_pool); s->h_part_sum.release(); s->h_part_zero.release(); s->h_timeline.release();
      s->probs.release(); s->tabs.release(); s->arena.release(); s->persist.release(); s->cov_bits.release(); s->bad.release(); if (s->ev_tables) (void)hipEventDestroy(s->ev_tables); if (s->ev_ovf) (void)hipEventDestroy(s->ev_ovf);
      s->red.release();
    }
    for (auto& s : c->pacbios) { s->d_lens.release(); s->rec_off.release(); s->rec_walk.release(); s->rec_logp.release(); s->walk_count.release(); s->logprobs.release(); s->red.release(); drop_stage(s->stage);
      s->d_bases.release(); s->dp.release(); s->sweep.release(); }
    c->packed.release(); c->packed_host.release(); c->batch_dev.release(); c->batch_host.release(); c->aln_scratch.release();
    c->aln_small[0].release(); c->aln_small[1].release(); c->fetch_host.release(); c->warm_buf.release();
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

// The GEN instantiations of the paired scoring kernels (path sets with windows that occur several times) call functions and keep
// candidates in private memory: the first launch of such a kernel makes the runtime load its code and allocate the queue's
// scratch memory -- 7 ms, once. Paid here, when a paired read set is added, by a launch of one block that finds nothing to do,
// not inside the annealing call whose move first duplicates a node.
static int warm_general_kernels(gaml_hip_ctx* c) {
  if (c->general_warm || c->device < 0) return 0;
  HIP_TRY(c, hipSetDevice(c->device));  // (a shard of a multi-device context: the calling thread may have another device current)
  HIP_TRY(c, c->warm_buf.reserve(256));
  HIP_TRY(c, hipMemsetAsync(c->warm_buf.p, 0, 256, c->stream));
  PairedArgs a;
  memset((void*)&a, 0, sizeof(a));
  a.total_blocks = 1;  // main_blocks = 0: the block takes the wave-per-pair branch, whose item count (n - n_main + dstate[kDsSpill]) is 0
  a.dstate = c->warm_buf.as<int>();
  a.part_sum = c->warm_buf.as<double>() + 8; a.part_zero = c->warm_buf.as<int>() + 32;
  hipLaunchKernelGGL((paired_score_kernel<false, true>), dim3(1), dim3(kBlock), 0, c->stream, a);
  MultiSets ms;
  memset((void*)&ms, 0, sizeof(ms));
  ms.pad_ = 16;  // (the wave-per-pair blocks are "left out": with no path set in `ms` the block returns at once)
  hipLaunchKernelGGL((paired_score_multi_kernel<true>), dim3(1), dim3(kBlock), 0, c->stream, a, ms);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->general_warm = true;
  return 0;
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
  if (int e = warm_general_kernels(c)) return e;
  // the reads and their max-hash index go to the device with the read set (the reference reads its reads when it starts, too),
  // not inside the first evaluation: 250 MB of bases at cfg3
  if (c->device >= 0) {
    PairedSet& ps = *c->paireds.back();
    HIP_TRY(c, hipSetDevice(c->device));
    for (int mt = 0; mt < 2; mt++) {
      if (!aln_gpu_capable(c, ps.mate[mt])) continue;
      if (int e = aln_small_reserve(c, c->aln_small[mt])) return e;
      if (int e = aln_upload_index(c, ps.mate[mt], ps.dev[mt].aln)) return e;
    }
    if (int e = prepare_paired_tables(c, ps)) return e;  // the set's fixed tables (insert sizes, floors), its per-read probabilities' buffer
    if (ps.mate[0].n_local() > 0) { if (int e = paired_upload_statics(c, ps, c->stream)) return e; }  // length codes, per-combination tables, the memo of pair terms
    if (int e = paired_prereserve(c, ps, c->stream)) return e;  // the record tables' buffers, the builds' scratch, the delta store
  }
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

#include "pacbio_api.hip.h"
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
  for (auto& ps : c->paireds) paired_refresh_counts(*ps);
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
  bool spin = KNOB(c, 7) == 0 && !c->handles.empty();
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
  // the device is through with this evaluation: the delta lists' exact counts are in pinned memory (written by the kernels
  // that maintain them, in stream order before the scoring launch)
  for (auto& ps : c->paireds) {
    paired_refresh_counts(*ps);
    if (paired_delta_overflowed(*ps)) return fail(c, GAML_HIP_ESTATE, "delta store overflow (spill area); call gaml_hip_compact_tables and evaluate again");
  }
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

#include "paired_batch.hip.h"
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
    TableDev& T = c->paireds[h.idx]->tab;
    if (!T.ros_valid) {
      T.read_of_slot_host.resize((size_t)have);
      HIP_TRY(c, hipMemcpy(T.read_of_slot_host.data(), T.read_of_slot.p, (size_t)have * sizeof(int32_t), hipMemcpyDeviceToHost));
      T.ros_valid = true;
    }
    const auto& ros = T.read_of_slot_host;
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
  if (w.first < 0 && w.count > 0 && cap > 0) {  // filed by the aligner's kernels: the records exist in the device pool only
    SetRef h = c->handles[rs];
    if (h.kind != 1 || w.dfirst < 0 || c->device < 0) return -2;
    const int64_t k = std::min<int64_t>(w.count, cap);
    std::vector<int4> tmp((size_t)k);
    if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(tmp.data(), c->paireds[h.idx]->dev[mate].pool.as<int4>() + w.dfirst, (size_t)k * sizeof(int4), hipMemcpyDeviceToHost) != hipSuccess) return -2;
    for (int64_t i = 0; i < k; i++) out[i] = gaml_aligment{tmp[i].y, tmp[i].z & 0xff, tmp[i].w + (int32_t)m->lo, (tmp[i].z >> 8) & 1};
    return w.count;
  }
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

// ---- introspection a caller may want in production (what bench.py prints beside its numbers)
int gaml_hip_table_stats(gaml_hip_ctx* c, int rs, int64_t* out10) {
  int64_t* out8 = out10;
  MULTI_SHARD0(c);
  if (!c || rs < 0 || rs >= (int)c->handles.size() || c->handles[rs].kind != 1 || !out10) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  PairedSet& s = *c->paireds[c->handles[rs].idx];
  paired_refresh_counts(s);
  out8[0] = s.full_rebuilds; out8[1] = s.delta_updates; out8[2] = s.nd_est; out8[3] = s.async_rebuilds;
  out8[4] = s.batches_patched; out8[5] = s.batches_full;
  out8[6] = s.pt.dropped_records[0]; out8[7] = s.pt.dropped_records[1];
  out10[8] = s.delta_left_out; out10[9] = s.pt.n0a;
  return GAML_HIP_OK;
}

int gaml_hip_last_phases(gaml_hip_ctx* c, double* out8) {
  if (!c || !out8) return GAML_HIP_EINVAL;
  MULTI_SHARD0(c);
  for (int i = 0; i < 8; i++) out8[i] = c->prof[i];
  return GAML_HIP_OK;
}

int gaml_hip_pair_classes(gaml_hip_ctx* c, int rs, int64_t* out4) {
  MULTI_SHARD0(c);
  if (!c || rs < 0 || rs >= (int)c->handles.size() || c->handles[rs].kind != 1 || !out4) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  for (int k = 0; k < 4; k++) out4[k] = c->paireds[c->handles[rs].idx]->pt.class_count[k];
  return GAML_HIP_OK;
}


#ifdef GAML_HIP_DEV
#include "debug_api.hip.h"
#endif

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

int gaml_hip_aligner_stages(gaml_hip_ctx* c, double* out6) {
  if (!c || !out6) return GAML_HIP_EINVAL;
  MULTI_SHARD0(c);
  for (int k = 0; k < 5; k++) out6[k] = c->aln_stage_us[k];
  out6[5] = (double)c->aln_batches;
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
  if (reset) { c->stat_launches = 0; c->stat_device_us = 0; c->stat_algo_bytes = 0; }
  return GAML_HIP_OK;
}

}  // extern "C"
