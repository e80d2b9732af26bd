// paired_launch.hip.h -- a paired read set on the device: tables, per-call arena, kernel arguments, launches.
// (CalcScoreForPathsNew graph.cc:1952-1989 evaluated from scratch; included by gaml_hip.hip only.)
//
//   prepare_paired_tables      once per set: insert-size / floor tables (host libm) on the device
//   paired_sync_tables         cold path: the record tables follow the window cache (delta lists or a full rebuild)
//   paired_layout / _pack      per path set: where its tables sit in an arena slot, and the bytes
//   arena_acquire / _commit    the slot itself: device memory the HOST writes directly (PCIe BAR), or pinned staging
//                              + copy when the device has no large BAR
//   paired_base_args / paired_set_view   kernel arguments: what is common to all path sets / what one set changes
//   launch_paired              one path set: ONE dispatch (paired_score_kernel) on the warm path
//   launch_paired_multi        up to 8 path sets in one pass over the records (paired_score_multi_kernel)
#pragma once

namespace {

// the host half of prepare_paired_tables: insert-size / floor tables (host libm); needs no device
int paired_host_tabs(gaml_hip_ctx* c, PairedSet& s) {
  if (!s.ins_tab.empty() || !s.floor_tab.empty()) return 0;
  const double c0 = s.cfg.min_prob_start, k0 = s.cfg.min_prob_per_base;
  // The reference tabulates GetInsertProbability for d < mean + 5 sd (graph.cc:1801-1804) and calls
  // the same function directly beyond (:1877-1882). Same formula, so one host table serves both;
  // it is extended until the f64 value is exactly 0.0 (exp underflows at z ~ 38.6; it is
  // monotone beyond the mean, so everything further is 0.0 too). No exp() on the device.
  auto ins = [&](int d) {
    double z = ((double)d - s.cfg.insert_mean) / s.cfg.insert_std;  // graph.cc:1593-1598
    return std::exp(-z * z / 2.0) / (std::sqrt(2 * M_PI) * s.cfg.insert_std);
  };
  const int kInsCap = 1 << 22;
  s.ins_tab.clear();
  for (int d = 0; d < kInsCap; d++) {
    double v = ins(d);
    if (v == 0.0 && (double)d > s.cfg.insert_mean) break;
    s.ins_tab.push_back(v);
  }
  if ((int)s.ins_tab.size() >= kInsCap)
    return fail(c, GAML_HIP_EINVAL, "insert_std too large: the insert-size table would exceed 4M entries");
  int smax = s.mate[0].max_len + s.mate[1].max_len;
  s.floor_tab.resize(smax + 1);
  s.logfloor_tab.resize(smax + 1);
  for (int v = 0; v <= smax; v++) {
    s.floor_tab[v] = std::exp(c0 + k0 * v);          // graph.cc:1506-1507
    s.logfloor_tab[v] = std::log(s.floor_tab[v]);    // graph.cc:1510-1512 on a floored read
    if (!(s.floor_tab[v] > 0.0)) s.floor_positive = false;  // exp underflow: the reference then takes log(0) for a read without alignment
  }
  s.covthr_tab.resize(s.mate[1].max_len + 1);
  for (int v = 0; v <= s.mate[1].max_len; v++) s.covthr_tab[v] = std::exp(c0 + k0 * (v + v));  // graph.cc:1855-1857
  return 0;
}

int prepare_paired_tables(gaml_hip_ctx* c, PairedSet& s) {
  if (s.tabs_uploaded) return 0;
  if (int e = paired_host_tabs(c, s)) return e;
  size_t total = s.ins_tab.size() + s.floor_tab.size() + s.logfloor_tab.size() + s.covthr_tab.size();
  HIP_TRY(c, s.tabs.reserve(std::max<size_t>(1, total) * sizeof(double)));
  double* d = s.tabs.as<double>();
  size_t at = 0;
  auto up = [&](const std::vector<double>& v) -> hipError_t {
    hipError_t e = v.empty() ? hipSuccess : hipMemcpy(d + at, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice);
    at += v.size();
    return e;
  };
  HIP_TRY(c, up(s.ins_tab));
  HIP_TRY(c, up(s.floor_tab));
  HIP_TRY(c, up(s.logfloor_tab));
  HIP_TRY(c, up(s.covthr_tab));
  const int64_t n = s.mate[0].n_local();
  HIP_TRY(c, s.probs.reserve(std::max<size_t>(1, n) * sizeof(double)));
  HIP_TRY(c, s.red.init());
  HIP_TRY(c, s.bad.reserve(sizeof(unsigned long long)));
  s.tabs_uploaded = true;
  return 0;
}

// pass 1: window registration / alignment of missing windows and the placement of cached windows
// (memoised per distinct path: PairedPlanner)
int prepare_paired_structure(gaml_hip_ctx* c, PairedSet& s, const int32_t* flat, const int64_t* offs, int32_t n_paths) {
  // a coverage penalty needs per-path bitmap layout by position (and the sweep's contig starts): whole-set planning;
  // knob 12 = 1 forces it for A/B runs and tests
  const bool incremental = !(s.cfg.penalty_constant > 0) && KNOB(c, 12) == 0;
  std::string err;
  if (!s.planner.begin(c->g, s.mate, flat, offs, n_paths, incremental, &err)) return fail(c, GAML_HIP_EINVAL, err);
  return 0;
}

// pass 2: position-filter thresholds (need the windows' global largest positions) + occurrence images
void prepare_paired_tables_host(gaml_hip_ctx* c, PairedSet& s, PairedPrep& p) {
  (void)c;
  static const bool trace = getenv("GAML_HIP_TRACE_HOST") != nullptr;
  const double q0 = now_us();
  s.planner.finish(s.mate);
  const double q1 = now_us();
  const bool cov = s.cfg.penalty_constant > 0;
  // coverage bitmap layout + contig starts (events of type 1, graph.cc:1826,1833-1835)
  p.path_base.assign(1, 0);
  p.start_off.assign(1, 0);
  p.starts.clear();
  if (cov) {  // only the coverage sweep reads these (whole-set planning: slots are positions)
    const PlanView& v = s.planner.view();
    p.path_base.reserve(v.paths.size() + 1);
    p.start_off.reserve(v.paths.size() + 1);
    for (const PathMemo* pm : v.paths) {
      p.starts.insert(p.starts.end(), pm->starts.begin(), pm->starts.end());
      p.start_off.push_back((int32_t)p.starts.size());
      int32_t bits = ((pm->length + 64 + 31) / 32) * 32;  // one bit per path position, padded to words (+ slack)
      p.path_base.push_back(p.path_base.back() + bits);
    }
  }
  p.total_bits = p.path_base.back();
  p.n_paths = s.planner.n_paths();
  const double q2 = now_us();
  s.planner.apply(s.mate, s.image);  // the occurrence tables: whole-set rebuild, or the changed paths' entries in / out
  p.assembled_records = s.planner.assembled(0) + s.planner.assembled(1);
  p.general = !s.image[0].general_wids.empty() || !s.image[1].general_wids.empty();
  if (trace) fprintf(stderr, "pass2: finish %.1f us, starts %.1f us, images %.1f us\n", q1 - q0, q2 - q1, now_us() - q2);
}

}  // namespace
#include "paired_tables.hip.h"
namespace {

// ---------------------------------------------------------------------------------------------------------
// per path set: its tables inside an arena slot
// ---------------------------------------------------------------------------------------------------------
// GetTotalProb's "p = t / 2T; p < floor" as a threshold on t: the smallest double whose quotient by 2T reaches the
// floor, found with the same IEEE division (monotone in t), so `t < tfloor` IS the reference's decision
double tfloor_for(double floor, double two_T) {
  if (!(floor > 0.0)) return 0.0;
  double x = floor * two_T;
  for (int k = 0; k < 64 && x > 0.0 && (std::nextafter(x, 0.0) / two_T) >= floor; k++) x = std::nextafter(x, 0.0);
  for (int k = 0; k < 64 && (x / two_T) < floor; k++) x = std::nextafter(x, std::numeric_limits<double>::infinity());
  return x;
}

struct PairedLayout { size_t tfloor_off; OccLayout l0, l1; size_t pb_off, so_off, st_off, total; };

// pad_to: table entries per mate (a batch pads every set's tables to the window count the batch may reach)
PairedLayout paired_layout(const PairedSet& s, const PairedPrep& p, const size_t* pad_to = nullptr) {
  PairedLayout L;
  L.tfloor_off = 0;
  const size_t hdr = align16(256 * sizeof(double));  // thresholds per length code (the table of codes may grow with a rebuild)
  L.l0 = layout_image(s.image[0], hdr, pad_to ? pad_to[0] : 0);
  L.l1 = layout_image(s.image[1], L.l0.end, pad_to ? pad_to[1] : 0);
  L.pb_off = L.l1.end; L.so_off = L.st_off = 0; L.total = L.l1.end;
  if (s.cfg.penalty_constant > 0) {
    L.so_off = align16(L.pb_off + p.path_base.size() * sizeof(int32_t));
    L.st_off = align16(L.so_off + p.start_off.size() * sizeof(int32_t));
    L.total = align16(L.st_off + p.starts.size() * sizeof(int32_t));
  }
  return L;
}

// the thresholds on t per length code for this set's 2T -- AFTER paired_sync_tables (a rebuild renumbers the codes)
void paired_pack_thresholds(const PairedSet& s, const PairedLayout& L, double two_T, char* dst) {
  double tf[256];
  const size_t nc = std::min<size_t>(256, s.pt.len_combo.size());
  for (size_t ci = 0; ci < nc; ci++) {
    const int L0 = (int)(s.pt.len_combo[ci] & 0xffff), L1 = (int)(s.pt.len_combo[ci] >> 16);
    tf[ci] = tfloor_for(s.floor_tab[L0 + L1], two_T);
  }
  if (nc) memcpy(dst + L.tfloor_off, tf, nc * sizeof(double));
}

// write-only (dst may be device memory behind the PCIe BAR: never read it back)
void paired_pack(const PairedSet& s, const PairedPrep& p, const PairedLayout& L, char* dst) {
  pack_image(s.image[0], L.l0, dst);
  pack_image(s.image[1], L.l1, dst);
  if (s.cfg.penalty_constant > 0) {
    memcpy(dst + L.pb_off, p.path_base.data(), p.path_base.size() * sizeof(int32_t));
    memcpy(dst + L.so_off, p.start_off.data(), p.start_off.size() * sizeof(int32_t));
    memcpy(dst + L.st_off, p.starts.data(), p.starts.size() * sizeof(int32_t));
  }
}

// ---------------------------------------------------------------------------------------------------------
// the arena: a ring of slots holding per-call tables. On a large-BAR device (MI355X) a slot is fine-grained device
// memory that the host fills with plain stores through the PCIe BAR (write-combined: 96 KB in ~2 us,
// tools/bar_write_probe.hip) -- no staging buffer, no copy command, no copy kernel in front of the scoring launch.
// Otherwise (or knob 8 != 0): pinned staging slot + hipMemcpyAsync (knob 8 = 1) / copy kernel (knob 8 = 2).
// ---------------------------------------------------------------------------------------------------------
int arena_acquire(gaml_hip_ctx* c, Arena& A, size_t bytes, hipStream_t st, int* slot_out, char** write_ptr) {
  const int k = A.next;
  A.next = (A.next + 1) % kRing;
  if (A.armed[k]) { HIP_TRY(c, hipEventSynchronize(A.done[k])); A.armed[k] = false; }
  if (!A.done[k]) HIP_TRY(c, hipEventCreateWithFlags(&A.done[k], hipEventDisableTiming));
  const bool direct = c->direct_write && KNOB(c, 8) == 0;
  if (bytes > A.cap[k] || A.direct[k] != direct) {
    HIP_TRY(c, hipStreamSynchronize(st));
    if (A.dev[k]) { HIP_TRY(c, hipFree(A.dev[k])); A.dev[k] = nullptr; A.cap[k] = 0; }
    const size_t want = std::max<size_t>(bytes + bytes / 2, (size_t)1 << 20);
    if (direct) HIP_TRY(c, hipExtMallocWithFlags(&A.dev[k], want, hipDeviceMallocFinegrained));
    else HIP_TRY(c, hipMalloc(&A.dev[k], want));
    A.cap[k] = want; A.direct[k] = direct;
  }
  if (direct) *write_ptr = (char*)A.dev[k];
  else { HIP_TRY(c, A.host[k].reserve(bytes)); *write_ptr = (char*)A.host[k].p; }
  *slot_out = k;
  return 0;
}

// after packing: make the bytes visible to the kernels launched next on `st`
int arena_commit(gaml_hip_ctx* c, Arena& A, int k, size_t bytes, hipStream_t st) {
  if (A.direct[k]) { _mm_sfence(); return 0; }  // drain the write-combining buffers; the doorbell write of the launch orders behind them
  if (bytes == 0) return 0;
  if (KNOB(c, 8) == 1 || (bytes & 15)) { HIP_TRY(c, hipMemcpyAsync(A.dev[k], A.host[k].p, bytes, hipMemcpyHostToDevice, st)); return 0; }
  const int n16 = (int)(bytes / 16);
  hipLaunchKernelGGL(stage_copy_kernel, dim3((unsigned)std::min(64, (n16 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                     (const int4*)A.host[k].dev, (int4*)A.dev[k], n16);
  HIP_TRY(c, hipGetLastError());
  return 0;
}

int arena_release(gaml_hip_ctx* c, Arena& A, int k, hipStream_t st) {
  if (c->host_results) return 0;  // a blocking call returns after the device is done with the slot: no event
  HIP_TRY(c, hipEventRecord(A.done[k], st));
  A.armed[k] = true;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// the resident copy of the tables (blocking calls on a large-BAR device): patched in place through the BAR
// ---------------------------------------------------------------------------------------------------------
int paired_persist_update(gaml_hip_ctx* c, PairedSet& s, double two_T, hipStream_t st) {
  PairedSet::Persist& P = s.persist;
  OccImage* im = s.image;
  size_t need_w[2], need_lo[2], need_m[2];
  bool relayout = P.dev == nullptr;
  for (int mt = 0; mt < 2; mt++) {
    need_w[mt] = std::max<size_t>(1, im[mt].occ12.size());
    need_lo[mt] = im[mt].multi_off.size();
    need_m[mt] = std::max<size_t>(1, im[mt].multi.size());
    if (need_w[mt] > P.cap_w[mt] || need_lo[mt] > P.cap_lo[mt] || need_m[mt] > P.cap_m[mt]) relayout = true;
  }
  bool full = !P.valid || im[0].changed_all || im[1].changed_all;
  if (relayout) {
    size_t at = align16(256 * sizeof(double));  // thresholds per length code
    P.off_tfloor = 0;
    for (int mt = 0; mt < 2; mt++) {
      // (room for many times the windows there are: growing means a new fine-grained allocation -- milliseconds, and an annealing run
      //  adds two windows per move; 12 bytes a window)
      P.cap_w[mt] = std::max<size_t>(4 * need_w[mt], (size_t)1 << 18);
      P.cap_lo[mt] = 4 * need_lo[mt] + 16384;
      P.cap_m[mt] = 4 * need_m[mt] + 65536;
      P.off_occ[mt] = at; at = align16(at + P.cap_w[mt] * sizeof(Occ12));
      P.off_lo[mt] = at; at = align16(at + P.cap_lo[mt] * sizeof(int32_t));
      P.off_m[mt] = at; at = align16(at + P.cap_m[mt] * sizeof(OccQuad));
    }
    if (at > P.bytes) {
      HIP_TRY(c, hipStreamSynchronize(st));
      if (P.dev) { HIP_TRY(c, hipFree(P.dev)); P.dev = nullptr; }
      HIP_TRY(c, hipExtMallocWithFlags(&P.dev, at, hipDeviceMallocFinegrained));
      P.bytes = at;
    }
    // windows to come read as absent: every table entry all ones, once per layout (the image only ever grows)
    for (int mt = 0; mt < 2; mt++) memset((char*)P.dev + P.off_occ[mt], 0xff, P.cap_w[mt] * sizeof(Occ12));
    full = true;
  }
  char* wp = (char*)P.dev;  // write-only: device memory behind the PCIe BAR
  for (int mt = 0; mt < 2; mt++) {
    OccImage& t = im[mt];
    char* occ = wp + P.off_occ[mt];
    if (full) {
      if (!t.occ12.empty()) memcpy(occ, t.occ12.data(), t.occ12.size() * sizeof(Occ12));
    } else {
      for (int32_t w : t.changed) memcpy(occ + (size_t)w * sizeof(Occ12), &t.occ12[w], sizeof(Occ12));
    }
    if (full || t.lists_changed) {
      memcpy(wp + P.off_lo[mt], t.multi_off.data(), t.multi_off.size() * sizeof(int32_t));
      if (!t.multi.empty()) memcpy(wp + P.off_m[mt], t.multi.data(), t.multi.size() * sizeof(OccQuad));
    }
    t.take_changed();
  }
  PairedLayout L;  // only the threshold offset is used here
  L.tfloor_off = P.off_tfloor;
  paired_pack_thresholds(s, L, two_T, wp);
  P.valid = true;
  _mm_sfence();  // drain the write-combining buffers; the launch's doorbell write orders behind them
  return 0;
}

// tfloor_c[0] for this 2T, by value in the kernel arguments (what paired_pack_thresholds writes for length combination 0)
double paired_tfloor0(const PairedSet& s, double two_T) {
  if (s.pt.len_combo.empty()) return 0.0;
  return tfloor_for(s.floor_tab[(s.pt.len_combo[0] & 0xffff) + (s.pt.len_combo[0] >> 16)], two_T);
}

void paired_persist_view(const PairedSet& s, int32_t total_len, SetDev& sd) {
  const PairedSet::Persist& P = s.persist;
  const char* base = (const char*)P.dev;
  for (int mt = 0; mt < 2; mt++) {
    sd.occ12[mt] = (const Occ12*)(base + P.off_occ[mt]);
    sd.multi_off[mt] = (const int*)(base + P.off_lo[mt]);
    sd.multi[mt] = (const int4*)(base + P.off_m[mt]);
  }
  sd.tfloor_c = (const double*)(base + P.off_tfloor);
  const int tl = total_len == 0 ? 1 : total_len;
  sd.two_T = (double)(2 * tl);
  sd.tfloor0 = paired_tfloor0(s, sd.two_T);
  sd.log_two_T = std::log(sd.two_T);
  sd.part_sum = nullptr; sd.part_zero = nullptr;
}

// ---------------------------------------------------------------------------------------------------------
// kernel arguments
// ---------------------------------------------------------------------------------------------------------
struct GridPlan { int blocks0a, blocks0, blocks1, blocks2, blocks_d, main_blocks, ovf_blocks, total_blocks; };

// what every path set of a launch shares: record tables, length tables, memo, classes, grid
void paired_base_args(gaml_hip_ctx* c, PairedSet& s, PairedArgs& a, GridPlan& gp) {
  const int64_t n = s.mate[0].n_local();
  memset(&a, 0, sizeof(a));
  for (int mt = 0; mt < 2; mt++) {
    a.m[mt].first = s.tab.first[mt].as<int4>();
    a.m[mt].extra = s.tab.extra[mt].as<int4>();
    a.m[mt].mism_pow = s.dev[mt].pows.as<double>();
    a.m[mt].match_pow = s.dev[mt].pows.as<double>() + s.dev[mt].pow_n;
    a.rec8[mt] = s.tab.rec8[mt].as<unsigned long long>();
    a.inl[mt] = s.tab.inl[mt].as<int4>();
    a.dirty_recs[mt] = s.dl_rec[mt].as<int4>();
    a.spill_rng[mt] = s.sp_rng[mt].as<int2>();
    a.spill_recs[mt] = s.sp_rec[mt].as<int4>();
  }
  a.len12 = s.tab.len12.as<uint32_t>();
  const double* tabs = s.tabs.as<double>();
  a.ins_tab = tabs; a.ins_n = (int)s.ins_tab.size();
  a.floor_tab = tabs + s.ins_tab.size();
  a.logfloor_tab = a.floor_tab + s.floor_tab.size();
  a.covthr_tab = a.logfloor_tab + s.logfloor_tab.size();
  a.n = (int)n;
  a.probs = s.probs.as<double>();
  a.n_reads = (double)n;
  a.ticket = s.red.ticket.as<unsigned>();
  const int64_t n0 = s.pt.class_count[0], n01 = n0 + s.pt.class_count[1], n_main = n01 + s.pt.class_count[2];
  a.n0 = (int)n0; a.n01 = (int)n01; a.n_main = (int)n_main;
  const int64_t n0a = s.pt.n0a, n0b = n0 - n0a;
  a.n0a = (int)n0a;
  a.static_idx = s.tab.static_idx.as<int>();
  a.static_val = s.tab.static_val.as<double2>();
  const size_t nc = std::max<size_t>(1, s.pt.len_combo.size());
  const double* ct = s.combo_tabs.as<double>();
  a.pe[0] = ct; a.pe[1] = ct + nc * 64; a.floor_c = ct + 2 * nc * 64; a.logfloor_c = a.floor_c + nc; a.covthr_c = a.logfloor_c + nc;
  a.len_code = s.tab.len_code.as<unsigned char>();
  a.len_combo = s.len_combo_dev.as<uint32_t>();
  a.n_codes = (int)std::min<size_t>(256, s.pt.len_combo.size());
  a.len_combo0 = s.pt.len_combo.empty() ? 0xffffffffu : s.pt.len_combo[0];
  a.logfloor0 = s.pt.len_combo.empty() ? 0.0 : s.logfloor_tab[(s.pt.len_combo[0] & 0xffff) + (s.pt.len_combo[0] >> 16)];  // = logfloor_c[0], by value
  a.memo = s.memo_codes > 0 ? s.memo.as<double2>() : nullptr;
  a.lt_codes = s.memo_codes;
  // the delta lists are maintained on the device: the kernels read their exact counts there (PairedArgs::dstate); the host
  // sizes the grid from what it knows -- an upper bound of the delta pairs (exact after a blocking call), the long lists as
  // last read back (+ a block when activations may have added some: the wave-per-pair items are strided over the blocks,
  // any number of blocks >= 1 scores them all). Both are functions of the call sequence alone.
  const size_t nd = (size_t)s.nd_est;
  a.spill_slot = s.sp_slot.as<int>();
  a.dstate = s.dstate.as<int>();
  a.dirty_slots = s.dl_slot.as<int>();
  a.dirty_spill = s.dl_spill.as<int>();
  const int64_t ovf_total = n - n_main + s.ns_est + (s.spill_may_grow ? 4 : 0);  // wave-per-pair items: table pairs with long lists, then delta pairs with long lists
  // 3 blocks per CU and one round of four pairs per lane at cfg3 (tools/kbench.py sweep); larger sets get more blocks, up to 8 per CU
  // A second round for a few lanes doubles the launch (every block is resident: the launch lasts as long as its longest
  // lane): up to 5 blocks per CU the compact class gets exactly the blocks one round needs.
  // The two parts of class 0 get blocks of their own (PairedArgs::blocks0a): the first by the rule above; the second --
  // a few per cent of the pairs, one more round trip per pair -- a lane per pair up to a third of that.
  const int64_t one_round = (n0a + 4 * kBlock - 1) / (4 * kBlock);
  const int cap0 = KNOB(c, 0) > 0 ? KNOB(c, 0)
                                   : (int)std::min<int64_t>(kMaxBlocks, one_round > 768 && one_round <= 1280 ? one_round : std::max<int64_t>(768, n0 / 2900));
  gp.blocks0a = n0a > 0 ? (int)std::max<int64_t>(1, std::min<int64_t>((n0a + 2 * kBlock - 1) / (2 * kBlock), cap0)) : 0;
  const int cap0b = KNOB(c, 20) > 0 ? KNOB(c, 20) : std::max(1, n0a > 0 ? cap0 / 3 : cap0);
  const int64_t blocks0b = n0b > 0 ? std::max<int64_t>(1, std::min<int64_t>(n0a > 0 ? (n0b + kBlock - 1) / kBlock : (n0b + 2 * kBlock - 1) / (2 * kBlock), cap0b)) : 0;
  gp.blocks0 = (int)std::max<int64_t>(1, gp.blocks0a + blocks0b);
  // the 2-record class: a third of the compact class's blocks (one block per CU at cfg3), lanes take 1-2 pairs; more
  // blocks only crowd the compact class out (tools/blocks_sweep.py at cfg3, pairs ordered by window in every class:
  // 128 blocks 11.4 us, 192: 10.1, 224-256: 9.8-9.9, 320: 10.2)
  const int cap1 = KNOB(c, 10) > 0 ? KNOB(c, 10) : cap0 / 3;
  gp.blocks1 = (int)std::max<int64_t>(1, std::min<int64_t>((n01 - n0 + kBlock - 1) / kBlock, cap1));
  gp.blocks2 = (int)std::max<int64_t>(1, std::min<int64_t>((n_main - n01 + kBlock - 1) / kBlock, kMaxBlocks / 4));
  // delta pairs: one lane per pair behind the table classes
  gp.blocks_d = nd ? (int)std::min<int64_t>(((int64_t)nd + kBlock - 1) / kBlock, 1024) : 0;
  gp.main_blocks = gp.blocks0 + gp.blocks1 + gp.blocks2 + gp.blocks_d;
  gp.ovf_blocks = ovf_total > 0 ? (int)std::min<int64_t>((ovf_total + 3) / 4, kOvfMaxBlocks) : 0;
  gp.total_blocks = gp.main_blocks + gp.ovf_blocks;
  a.blocks0a = gp.blocks0a;
  a.blocks0 = gp.blocks0;
  a.blocks01 = gp.blocks0 + gp.blocks1;
  a.blocks012 = gp.blocks0 + gp.blocks1 + gp.blocks2;
  a.main_blocks = gp.main_blocks;
  a.total_blocks = gp.total_blocks;
}

// what one path set changes: its occurrence tables inside `arena`, 2T and the thresholds that follow from it
void paired_set_view(const PairedSet& s, const PairedLayout& L, const char* arena, int32_t total_len, SetDev& sd) {
  const OccLayout* l[2] = {&L.l0, &L.l1};
  for (int mt = 0; mt < 2; mt++) {
    sd.occ12[mt] = (const Occ12*)(arena + l[mt]->direct);
    sd.multi_off[mt] = (const int*)(arena + l[mt]->multi_off);
    sd.multi[mt] = (const int4*)(arena + l[mt]->multi);
  }
  sd.tfloor_c = (const double*)(arena + L.tfloor_off);
  const int tl = total_len == 0 ? 1 : total_len;  // graph.cc:1500-1502
  sd.two_T = (double)(2 * tl);
  sd.tfloor0 = paired_tfloor0(s, sd.two_T);
  sd.log_two_T = std::log(sd.two_T);
  sd.part_sum = nullptr; sd.part_zero = nullptr;
}

void paired_apply_set(PairedArgs& a, const SetDev& sd) {
  for (int mt = 0; mt < 2; mt++) { a.m[mt].occ12 = sd.occ12[mt]; a.m[mt].occ = nullptr; a.occ12[mt] = sd.occ12[mt]; a.m[mt].multi_off = sd.multi_off[mt]; a.m[mt].multi = sd.multi[mt]; }
  a.tfloor_c = sd.tfloor_c; a.tfloor0 = sd.tfloor0; a.two_T = sd.two_T; a.log_two_T = sd.log_two_T;
  a.part_sum = sd.part_sum; a.part_zero = sd.part_zero;
}

CovArgs paired_cov_args(const PairedSet& s, const PairedPrep& p, const PairedLayout& L, const char* arena) {
  CovArgs ca;
  ca.bits = s.cov_bits.as<uint32_t>();
  ca.path_base = (const int*)(arena + L.pb_off);
  ca.start_off = (const int*)(arena + L.so_off);
  ca.starts = (const int*)(arena + L.st_off);
  ca.n_paths = p.n_paths;
  ca.total_words = p.total_bits / 32;
  ca.cov_move = s.cfg.step;
  ca.far = s.cfg.insert_mean + 5 * s.cfg.insert_std;
  ca.bad = s.bad.as<unsigned long long>();
  return ca;
}

// per-block partials in pinned host memory (blocking calls): every block stores its partial straight there, the
// host adds them up in the finisher kernel's order (no finisher launch, no D2H copy). Sentinels let the host see
// when every partial has landed without waiting for the runtime's completion signal (fetch_partials).
int paired_host_partials(gaml_hip_ctx* c, PairedSet& s, int first_set, int n_sets, int n_partials, double** d_sum, int** d_zero) {
  const size_t per = (size_t)(4 * kMaxBlocks + kOvfMaxBlocks);
  HIP_TRY(c, s.h_part_sum.reserve(per * sizeof(double) * kMaxSets));
  HIP_TRY(c, s.h_part_zero.reserve(per * sizeof(int) * kMaxSets));
  *d_sum = (double*)s.h_part_sum.dev;
  *d_zero = (int*)s.h_part_zero.dev;
  double* hs = (double*)s.h_part_sum.p;
  int* hz = (int*)s.h_part_zero.p;
  for (int k = first_set; k < first_set + n_sets; k++) {
    for (int b2 = 0; b2 < n_partials; b2++) { hs[(size_t)k * per + b2] = std::numeric_limits<double>::quiet_NaN(); hz[(size_t)k * per + b2] = INT_MIN; }
    s.last_blocks[k] = n_partials;
  }
  s.host_part_stride = per;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// ONE path set
// ---------------------------------------------------------------------------------------------------------
int launch_paired(gaml_hip_ctx* c, PairedSet& s, PairedPrep& p, int32_t total_len, hipStream_t st, double* out4) {
  const double tpp = now_us();
  if (int e = prepare_paired_tables(c, s)) return e;
  const double tp0 = now_us();
  prepare_paired_tables_host(c, s, p);  // pass 2 (pass 1 ran in eval_begin)
  const double t_after_host = now_us();
  c->prof[1] = t_after_host - tp0;  // thresholds + occurrence tables
  gpu_probe(st, c->warm_buf.p, "before table sync");
  if (int e = paired_sync_tables(c, s, st)) return e;  // cold path: nothing to do on a warm cache
  const double tp1 = now_us();
  c->prof[4] = tp1 - t_after_host;
  if (tp1 - t_after_host > 1000.0 && getenv("GAML_HIP_TRACE_HOST")) fprintf(stderr, "table sync took %.0f us (evaluation %lld of the set, rebuild active %d slice %d)\n", tp1 - t_after_host, (long long)s.eval_count, (int)s.rebuild.active, s.rebuild.next_slice);

  const bool cov = s.cfg.penalty_constant > 0;
  const int tl = total_len == 0 ? 1 : total_len;
  // blocking call on a large-BAR device: the resident copy of the tables is patched in place (a few entries when the
  // path set shares most paths with the previous call's); otherwise a ring slot receives the whole tables
  const bool resident = c->host_results && c->direct_write && KNOB(c, 8) == 0 && KNOB(c, 13) == 0 && !cov;
  PairedLayout L;
  memset(&L, 0, sizeof(L));
  int slot = -1;
  const char* arena = nullptr;
  SetDev sd;
  if (resident) {
    c->prof[6] = (double)((s.image[0].changed_all || s.image[1].changed_all || !s.persist.valid) ? (s.image[0].occ12.size() + s.image[1].occ12.size()) * sizeof(Occ12)
                                                                                            : (s.image[0].changed.size() + s.image[1].changed.size()) * sizeof(Occ12));
    if (int e = paired_persist_update(c, s, (double)(2 * tl), st)) return e;
    paired_persist_view(s, total_len, sd);
  } else {
    L = paired_layout(s, p);
    char* wp = nullptr;
    if (int e = arena_acquire(c, s.arena, L.total, st, &slot, &wp)) return e;
    paired_pack(s, p, L, wp);
    paired_pack_thresholds(s, L, (double)(2 * tl), wp);
    if (int e = arena_commit(c, s.arena, slot, L.total, st)) return e;
    arena = (const char*)s.arena.dev[slot];
    paired_set_view(s, L, arena, total_len, sd);
    c->prof[6] = (double)L.total;
  }
  const double tp2 = now_us();
  c->prof[3] = tp2 - tp1;  // per-call tables written (directly into device memory, or staged)

  PairedArgs a;
  GridPlan gp;
  paired_base_args(c, s, a, gp);
  const int64_t n = s.mate[0].n_local();
  if (cov) {
    size_t words = (size_t)p.total_bits / 32;
    if (words * 4 > s.cov_bits.cap) { HIP_TRY(c, hipStreamSynchronize(st)); HIP_TRY(c, s.cov_bits.reserve(std::max<size_t>(4, words * 4))); }
    HIP_TRY(c, hipMemsetAsync(s.cov_bits.p, 0, std::max<size_t>(4, words * 4), st));
    HIP_TRY(c, hipMemsetAsync(s.bad.p, 0, sizeof(unsigned long long), st));
    a.cov_bits = s.cov_bits.as<uint32_t>();
    a.path_base = (const int*)(arena + L.pb_off);
  }
  // some window occurs several times in this path set (or needs the long occurrence form): the GEN instantiation, whose
  // lanes score such a pair where they meet it
  const bool gen_set = a.n_main > 0 && p.general;
  const int n_partials = gp.total_blocks;
  sd.part_sum = s.red.part_sum.as<double>();
  sd.part_zero = s.red.part_zero.as<int>();
  if (c->host_results && n > 0) { if (int e = paired_host_partials(c, s, 0, 1, n_partials, &sd.part_sum, &sd.part_zero)) return e; }
  s.last_blocks[0] = n > 0 ? n_partials : 0;
  s.last_host_partials = c->host_results;
  s.last_sets = 1;
  paired_apply_set(a, sd);
  a.out = out4;
  a.timeline = nullptr;
  const bool timeline = KNOB(c, 3) == 8;
  if (timeline) {  // in-kernel timeline (tools/kernel_timeline.py): stamps land in mapped host memory
    HIP_TRY(c, s.h_timeline.reserve((size_t)(4 * kMaxBlocks + kOvfMaxBlocks + 256) * (kBlock / 64) * 8 * sizeof(unsigned long long)));
    memset(s.h_timeline.p, 0, s.h_timeline.cap);
    a.timeline = (unsigned long long*)s.h_timeline.dev;
    s.timeline_waves = a.total_blocks * (kBlock / 64);
  }

  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  if (n > 0) {
    // HIP events bracket the dominant kernel only (bench.py's roofline; rocprofv3 must agree)
    if (c->event_timing && (c->event_tick++ % c->event_every) == 0) { if (int e = take_events(c, &ev)) return e; }
    // 0 (knob 2 = 1): the block that draws the last ticket adds the partials up (two-level tickets, grid_finish: the ~1,000 blocks
    // end together and their atomics on 17 words still take 10 us -- a finisher dispatch costs less); 1: finisher kernel
    // (stream-ordered calls; always when a second launch shares the partials); 2: the host adds them (blocking calls)
    int fin_mode = c->host_results ? 2 : (KNOB(c, 2) ? KNOB(c, 2) - 1 : 1);
    // a sharded evaluation's status words are written by whatever finishes the partials on the device
    double* status_out = nullptr;
    if (fin_mode != 2 && c->status_dst && !c->status_done) { status_out = c->status_dst; c->status_done = true; }
    if (fin_mode == 0) { a.status_out = status_out; a.status_a = c->status_a; a.status_b = c->status_b; }
    s.last_total_blocks = n_partials;
    const dim3 grid(a.total_blocks), block(kBlock);
    // Timed launches attach the two events to the dispatch itself (hipExtLaunchKernelGGL: the events carry
    // the kernel's own begin / end stamps, what rocprofv3's kernel trace reports). Separate hipEventRecord
    // markers around the launch would add the marker packets' processing to the interval: an EMPTY kernel
    // of this grid reads 6 us that way (tools/stream_floor.hip).
    hipEvent_t e0 = ev ? ev->first : nullptr, e1 = ev ? ev->second : nullptr;
#define GAML_LAUNCH_SCORE(...) hipExtLaunchKernelGGL((paired_score_kernel<__VA_ARGS__>), grid, block, KNOB(c, 1), st, e0, e1, 0, a)
#ifdef GAML_HIP_DEV
    if (timeline) GAML_LAUNCH_SCORE(false, false, true);  // (the instantiation with in-kernel stamps: development builds only)
    else
#endif
    if (gen_set) { if (fin_mode) GAML_LAUNCH_SCORE(false, true); else GAML_LAUNCH_SCORE(true, true); }
    else if (fin_mode) GAML_LAUNCH_SCORE(false, false);
    else GAML_LAUNCH_SCORE(true, false);
#undef GAML_LAUNCH_SCORE
    HIP_TRY(c, hipGetLastError());
    if (fin_mode == 1) {
      hipLaunchKernelGGL(finish_partials_kernel, dim3(1), dim3(kBlock), 0, st, a.part_sum, a.part_zero, n_partials, out4, cov ? -1.0 : 0.0, (double)n, status_out, c->status_a, c->status_b);
      HIP_TRY(c, hipGetLastError());
    }
  } else {
    s.last_total_blocks = 0;
    if (!c->host_results) HIP_TRY(c, hipMemsetAsync(out4, 0, 4 * sizeof(double), st));
  }
  if (cov && c->defer_cov) {
    // the sweep needs the union of all ranks' coverage marks: gaml_hip_eval_coverage_finish_async runs it
    int idx = 0;
    for (size_t i = 0; i < c->paireds.size(); i++) if (c->paireds[i].get() == &s) idx = (int)i;
    c->pending_cov.push_back(gaml_hip_ctx::PendingCov{idx, paired_cov_args(s, p, L, arena), out4});
  } else if (cov && n > 0 && p.total_bits > 0) {
    const CovArgs ca = paired_cov_args(s, p, L, arena);
    hipLaunchKernelGGL(coverage_sweep_kernel, dim3(grid_for(ca.total_words)), dim3(kBlock), 0, st, ca);
    HIP_TRY(c, hipGetLastError());
  }
  if (cov && n > 0 && !c->defer_cov) {
    hipLaunchKernelGGL(store_bad_bases_kernel, dim3(1), dim3(64), 0, st, s.bad.as<unsigned long long>(), out4, 1.0);
    HIP_TRY(c, hipGetLastError());
  }
  if (slot >= 0) { if (int e = arena_release(c, s.arena, slot, st)) return e; }
  c->prof[5] = now_us() - tp2;  // kernel launches
  if (now_us() - tpp > 5000.0 && getenv("GAML_HIP_TRACE_HOST"))
    fprintf(stderr, "launch_paired: fixed tables %.2f ms, pass 2 %.2f, table sync %.2f, per-call tables %.2f, launches %.2f\n", (tp0 - tpp) * 1e-3, (t_after_host - tp0) * 1e-3,
            (tp1 - t_after_host) * 1e-3, (tp2 - tp1) * 1e-3, (now_us() - tp2) * 1e-3);
  // SURVEY.md 8d accounting: 16 B per record, 8 B read lengths, 8 B probability written, per pair
  if (!c->event_timing || ev) {  // with timing on, the statistics describe the timed launches
    c->stat_algo_bytes += 16.0 * (double)p.assembled_records + 16.0 * (double)n;
    c->stat_launches++;
  }
  c->t_host_us += t_after_host;  // caller subtracts the start stamp
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// several path sets, one pass over the records. The caller has planned every set (pass 1 + pass 2 + pack) into
// consecutive regions of ONE arena slot: `arena` + k * stride, layouts L[k]. Blocking only (partials in pinned memory).
// ---------------------------------------------------------------------------------------------------------
bool paired_multi_capable(const gaml_hip_ctx* c, const PairedSet& s) {
  return !(s.cfg.penalty_constant > 0) && KNOB(c, 3) == 0 && KNOB(c, 4) == 0 && s.floor_positive;
}

// Sets [first, first + n_sets) of a batch (L / preps / total_lens / arena regions / partial regions are indexed by the
// set's number in the batch): a batch may go out in two launches so that the host plans the second half while the
// device scores the first.
int launch_paired_multi(gaml_hip_ctx* c, PairedSet& s, int first, int n_sets, const PairedLayout* L, const PairedPrep* preps, const int32_t* total_lens,
                        const char* arena, size_t stride, hipStream_t st, const unsigned char* const* chg = nullptr) {
  PairedArgs a;
  GridPlan gp;
  paired_base_args(c, s, a, gp);
  if (!a.memo) return fail(c, GAML_HIP_ESTATE, "multi-set launch without a memo (caller must check paired_multi_capable)");
  const int64_t n = s.mate[0].n_local();
  bool any_set = false;
  for (int k = first; k < first + n_sets; k++) any_set = any_set || (a.n_main > 0 && preps[k].general);
  const int n_partials = gp.total_blocks;
  double* d_sum = nullptr;
  int* d_zero = nullptr;
  s.last_total_blocks = n > 0 ? n_partials : 0;
  s.last_host_partials = true;
  s.last_sets = first + n_sets;
  if (n == 0) { for (int k = first; k < first + n_sets; k++) s.last_blocks[k] = 0; return 0; }
  if (int e = paired_host_partials(c, s, first, n_sets, n_partials, &d_sum, &d_zero)) return e;
  MultiSets ms;
  memset(&ms, 0, sizeof(ms));
  ms.n = n_sets;
  if (KNOB(c, 11) >= 32) ms.pad_ = (KNOB(c, 11) - 32) & 31;  // timing experiments: leave out classes of blocks (bits: compact, <=2, <=4, delta, wave-per-pair)
  if (chg && KNOB(c, 11) < 64) { ms.chg[0] = chg[0]; ms.chg[1] = chg[1]; }  // (>= 64: and every set resolves every pair)  // (which table entries differ between the sets: only a batch built from patches knows)
  for (int k = 0; k < n_sets; k++) {
    const int g = first + k;  // the set's number in the batch
    paired_set_view(s, L[g], arena + (size_t)g * stride, total_lens[g], ms.set[k]);
    ms.set[k].part_sum = d_sum + (size_t)g * s.host_part_stride;
    ms.set[k].part_zero = d_zero + (size_t)g * s.host_part_stride;
  }
  paired_apply_set(a, ms.set[0]);  // (fields every set overrides; harmless defaults)
  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  if (c->event_timing && (c->event_tick++ % c->event_every) == 0) { if (int e = take_events(c, &ev)) return e; }
  hipEvent_t e0 = ev ? ev->first : nullptr, e1 = ev ? ev->second : nullptr;
  const dim3 grid(a.total_blocks), block(kBlock);
  // (any set with a window that occurs several times: the GEN instantiation for all of them -- a set without such a window
  // never takes its extra branches, its sums are those of the other instantiation)
  if (any_set) hipExtLaunchKernelGGL((paired_score_multi_kernel<true>), grid, block, 0, st, e0, e1, 0, a, ms);
  else hipExtLaunchKernelGGL((paired_score_multi_kernel<false>), grid, block, 0, st, e0, e1, 0, a, ms);
  HIP_TRY(c, hipGetLastError());
  if (!c->event_timing || ev) {
    double rec = 0;
    for (int k = first; k < first + n_sets; k++) rec += (double)preps[k].assembled_records;
    c->stat_algo_bytes += 16.0 * rec + 16.0 * (double)n * n_sets;
    c->stat_launches++;
  }
  return 0;
}

}  // namespace
