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

// ---------------------------------------------------------------------------------------------------------
// cold path: the device record tables follow the alignment-window cache
// ---------------------------------------------------------------------------------------------------------
// the pair's records as the device tables hold them (activated before the last full build)
void paired_base_records(const PairTables& pt, int32_t slot, int mt, PairedSet::RecList& out) {
  const int64_t n0s = pt.class_count[0];
  if (slot < n0s) {
    const uint64_t r = pt.rec8[mt][slot];
    if (r != kNoRec8) out.push_back(RecQuad{(int32_t)(r & 0xffffff), (int32_t)((r >> 24) & 0xfffffff), (int32_t)((r >> 52) & 63) | ((int32_t)((r >> 58) & 1) << 8), 0});
  } else {
    const RecQuad& f = pt.rm[mt].first[slot - n0s];
    if (f.wid >= 0) {
      const int cnt1 = 1 + (int)((uint32_t)f.flags >> 9);
      for (int q = 0; q < cnt1; q++) { RecQuad r = q == 0 ? f : pt.rm[mt].extra[f.link + q - 1]; r.flags &= 0x1ff; r.link = 0; out.push_back(r); }
    }
  }
}

// windows activated since the tables were built: their pairs move to the delta list (host side)
// the records of window `w` of mate `mt` onto delta lists that are relative to the tables `pt` (the live lists and
// tables, or the lists being prepared for tables a worker has built). keep: null, or per record 0 = left out.
// touched: the lists' indices that changed (null: not tracked).
// The patch of a call written while its lists are made: a delta pair's entry (short form) is rewritten whenever the pair is
// touched -- its lists are in the cache right then; a second pass over the touched pairs re-read 200 bytes per pair, 24 ns
// an entry. `broken`: some touched pair holds more than two records on a mate (the long form: the second pass does it).
struct PatchSink { DeltaPatch2* buf; int32_t* of; int cap, n, n_new; int64_t mark_from; bool broken; std::vector<int32_t>* longer; };  // longer: pairs that need the long form
static void delta_add_window(const PairedSet& s, const PairTables& pt, std::vector<PairedSet::DirtyPair>& dirty, std::vector<int32_t>& of_slot,
                             std::vector<int32_t>* touched, int mt, int32_t w, const uint8_t* keep, PatchSink* sink = nullptr) {
  const ShortMate& m = s.mate[mt];
  const Window& win = m.wins[w];
  const int64_t n0s = pt.class_count[0];
  // A record touches four places chosen by its read id (its pair's slot, the slot's delta index, the pair's records
  // in the tables of both mates): ~100 ns of cache misses each when taken one after the other. Ask for them ahead.
  constexpr int64_t kAhead = 24, kAhead2 = 12;
  const int64_t end = win.first + win.count;
  const int32_t* const lens0 = s.mate[0].lens.data();  // (a new delta pair notes its read lengths here, where the read id is at hand:
  const int32_t* const lens1 = s.mate[1].lens.data();  //  the patch would otherwise go slot -> read -> lengths, three misses in a row)
  for (int64_t k = win.first; k < end; k++) {
    if (k + kAhead < end) { const int32_t rd = m.pool[k + kAhead].read_id; __builtin_prefetch(&pt.slot_of_read[rd]); __builtin_prefetch(&lens0[rd]); __builtin_prefetch(&lens1[rd]); }
    if (k + kAhead2 < end) {
      const int32_t sl = pt.slot_of_read[m.pool[k + kAhead2].read_id];
      __builtin_prefetch(&of_slot[sl]);
      if (sl < n0s) { __builtin_prefetch(&pt.rec8[0][sl]); __builtin_prefetch(&pt.rec8[1][sl]); }
      else { __builtin_prefetch(&pt.rm[0].first[sl - n0s]); __builtin_prefetch(&pt.rm[1].first[sl - n0s]); }
    }
    if (keep && !keep[(size_t)(k - win.first)]) continue;
    const gaml_aligment& r = m.pool[k];
    const int32_t slot = pt.slot_of_read[r.read_id];
    int32_t dj = of_slot[slot];
    if (dj < 0) {
      dj = of_slot[slot] = (int32_t)dirty.size();
      dirty.emplace_back();
      dirty.back().slot = slot;
      dirty.back().len12 = (uint32_t)lens0[r.read_id] | ((uint32_t)lens1[r.read_id] << 16);
      paired_base_records(pt, slot, 0, dirty.back().recs[0]);
      paired_base_records(pt, slot, 1, dirty.back().recs[1]);
    }
    if (touched) touched->push_back(dj);
    auto& lst = dirty[dj].recs[mt];
    RecQuad q{w, r.position, (r.edit_dist & 0xff) | ((r.orientation & 1) << 8), 0};
    // keep the device-table order: (window id, position)
    const RecQuad* b = lst.data();
    const RecQuad* pos = std::upper_bound(b, b + lst.size(), q, [](const RecQuad& x, const RecQuad& y) { return x.wid != y.wid ? x.wid < y.wid : x.pos < y.pos; });
    lst.insert((size_t)(pos - b), q);
    if (sink && !sink->broken) {
      const auto& d = dirty[dj];
      const size_t c0 = d.recs[0].size(), c1 = d.recs[1].size();
      int pi = sink->of[dj];
      if (c0 > 2 || c1 > 2) sink->longer->push_back(dj);  // (an entry it may have here already is overwritten by the long one: the long patch is applied second)
      else if (pi < 0 && sink->n >= sink->cap) sink->broken = true;
      else {
        if (pi < 0) { pi = sink->n++; sink->of[dj] = pi; sink->n_new += dj >= sink->mark_from; }
        DeltaPatch2& pe = sink->buf[pi];
        pe.dj = dj; pe.slot = d.slot; pe.spill = -1; pe.pad = 0;
        for (int m2 = 0; m2 < 2; m2++)
          for (int k2 = 0; k2 < 2; k2++) {
            const RecQuad none{-1, 0, 0, 0};
            const RecQuad& rr = k2 < (int)d.recs[m2].size() ? d.recs[m2][k2] : none;
            pe.rec[m2][k2] = make_int4(rr.wid, rr.pos, rr.flags, rr.link);
          }
        pe.rec[0][0].w = (int)d.len12;
        pe.rec[1][0].w = (int)(c0 | (c1 << 8));
      }
    }
  }
}

// windows activated since the tables were built: their pairs move to the delta lists (host side). While a worker
// builds new tables (log_after) the windows are also noted for the lists that will go with those tables.
void paired_extend_delta(PairedSet& s, bool fold, bool log_after, PatchSink* sink = nullptr) {
  if (s.dirty_of_slot.size() != (size_t)s.mate[0].n_local()) s.dirty_of_slot.assign((size_t)s.mate[0].n_local(), -1);
  TableRebuild& rb = s.rebuild;
  std::vector<uint8_t> keep;
  for (int mt = 0; mt < 2; mt++) {
    const ShortMate& m = s.mate[mt];
    for (int32_t w : m.activated_log) {
      const Window& win = m.wins[w];
      // a junction window's records that its first node's window always overwrites change nothing (host_model.cc
      // dominated_records): their pairs stay where they are
      const bool some_left_out = fold && undominated_records(m, w, keep) < win.count;
      if (some_left_out) for (uint8_t kp : keep) s.delta_left_out += !kp;
      if (log_after) {
        rb.after.push_back(TableRebuild::After{mt, w, some_left_out ? (int64_t)rb.after_keep.size() : -1});
        if (some_left_out) rb.after_keep.insert(rb.after_keep.end(), keep.begin(), keep.end());
        rb.sh_records += win.count;
      }
      delta_add_window(s, s.pt, s.dirty, s.dirty_of_slot, &s.dirty_touched, mt, w, some_left_out ? keep.data() : nullptr, sink);
    }
  }
  for (int mt = 0; mt < 2; mt++) { s.mate[mt].activated_log.clear(); s.dev[mt].uploaded_generation = s.mate[mt].active_generation; }
  s.delta_updates++;
}

// The lists that go with the tables a worker has built: up to `budget` records' worth of the windows noted since the
// snapshot (whole windows, in the order they were activated -- the lists do not depend on how the work was sliced).
void paired_shadow_advance(PairedSet& s, int64_t budget) {
  TableRebuild& rb = s.rebuild;
  if (!rb.sh_open) {
    rb.sh_dirty.clear();  // (normally emptied by the worker already)
    rb.sh_dirty.reserve(s.dirty.capacity());
    rb.sh_of_slot.assign((size_t)s.mate[0].n_local(), -1);
    rb.sh_touched.clear(); rb.sh_spill_of.clear(); rb.sh_spill_pairs.clear();
    rb.sh_next = 0;
    rb.sh_open = true;
  }
  while (rb.sh_next < rb.after.size() && budget > 0) {
    const TableRebuild::After& a = rb.after[rb.sh_next++];
    const int64_t cnt = s.mate[a.mate].wins[a.wid].count;
    delta_add_window(s, rb.pt, rb.sh_dirty, rb.sh_of_slot, &rb.sh_touched, a.mate, a.wid, a.keep_at >= 0 ? rb.after_keep.data() + a.keep_at : nullptr);
    budget -= cnt;
    rb.sh_records -= cnt;
  }
}

// One build of the record tables onto the device: uploads `pt` into `T`, derives the per-length-combination tables
// and tabulates the memo of pair terms (nothing in it depends on a path set: PairedArgs::memo) on `st`. Runs on the
// caller's thread -- the evaluation's, or the rebuild worker's (then `err` receives the text: the context's error
// string belongs to the evaluating thread).
int paired_upload_tables(gaml_hip_ctx* c, PairedSet& s, const PairTables& pt, TableDev& T, hipStream_t st, std::string* err) {
  auto bad = [&](const char* what, hipError_t e) { const std::string m = std::string(what) + ": " + hipGetErrorString(e); if (err) { *err = m; return GAML_HIP_EHIP; } return fail(c, GAML_HIP_EHIP, m); };
  auto up = [&](DevBuf& d, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = d.reserve(std::max<size_t>(16, bytes));
    if (e != hipSuccess || bytes == 0) return e;
    return hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice);
  };
#define UP_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return bad(#expr, e__); } while (0)
  for (int mt = 0; mt < 2; mt++) {
    UP_TRY(up(T.rec8[mt], pt.rec8[mt].data(), pt.rec8[mt].size() * sizeof(uint64_t)));
    UP_TRY(up(T.first[mt], pt.rm[mt].first.data(), pt.rm[mt].first.size() * sizeof(RecQuad)));
    UP_TRY(up(T.extra[mt], pt.rm[mt].extra.data(), pt.rm[mt].extra.size() * sizeof(RecQuad)));
    UP_TRY(up(T.inl[mt], pt.inl[mt].data(), pt.inl[mt].size() * sizeof(RecQuad)));
  }
  UP_TRY(up(T.static_idx, pt.static_idx.data(), pt.static_idx.size() * sizeof(int32_t)));
  UP_TRY(up(T.len_code, pt.len_code.data(), pt.len_code.size()));
  UP_TRY(up(T.len_combo, pt.len_combo.data(), pt.len_combo.size() * sizeof(uint32_t)));
  UP_TRY(up(T.len12, pt.len12.data(), pt.len12.size() * sizeof(uint32_t)));
  // per length-combination tables of the compact path: [pe mate 0 | pe mate 1 | floor | logfloor | covthr]
  const size_t nc = std::max<size_t>(1, pt.len_combo.size());
  std::vector<double> t(nc * 64 * 2 + nc * 3, 0.0);
  for (size_t ci = 0; ci < pt.len_combo.size(); ci++) {
    const int L[2] = {(int)(pt.len_combo[ci] & 0xffff), (int)(pt.len_combo[ci] >> 16)};
    for (int mt = 0; mt < 2; mt++)
      for (int e = 0; e < 64 && e <= L[mt]; e++)
        t[(size_t)mt * nc * 64 + ci * 64 + e] = s.mate[mt].mismatch_pow[e] * s.mate[mt].match_pow[L[mt] - e];  // graph.cc:1859-1863
    t[2 * nc * 64 + ci] = s.floor_tab[L[0] + L[1]];
    t[2 * nc * 64 + nc + ci] = s.logfloor_tab[L[0] + L[1]];
    t[2 * nc * 64 + 2 * nc + ci] = s.covthr_tab[L[1]];
  }
  UP_TRY(up(T.combo_tabs, t.data(), t.size() * sizeof(double)));
  // memo of the pair terms a single-term pair can take (first 4 length combinations, edits < 7, every tabulated distance)
  T.memo_codes = 0;
  if (KNOB(c, 4) == 0 && s.floor_positive && !pt.len_combo.empty() && !s.ins_tab.empty()) {
    const int codes = (int)std::min<size_t>(pt.len_combo.size(), kMemoCodes);
    const size_t entries = (size_t)codes * 49 * s.ins_tab.size();
    if (entries <= kMemoMaxEntries) {  // (the same bound build_pair_tables applies to its static indices)
      UP_TRY(T.memo.reserve(entries * sizeof(double2)));
      const double* ct = T.combo_tabs.as<double>();
      hipLaunchKernelGGL(logterm_kernel, dim3((unsigned)std::min<size_t>((entries + kBlock - 1) / kBlock, 1024)), dim3(kBlock), 0, st,
                         ct, ct + nc * 64, s.tabs.as<double>(), (int)s.ins_tab.size(), codes, T.memo.as<double2>());
      UP_TRY(hipGetLastError());
      T.memo_codes = codes;
      if (pt.n0a > 0) {  // the static pairs' entries, streamed with their records from here on
        UP_TRY(T.static_val.reserve((size_t)pt.n0a * sizeof(double2)));
        hipLaunchKernelGGL(static_values_kernel, dim3((unsigned)std::min<int64_t>((pt.n0a + kBlock - 1) / kBlock, 2048)), dim3(kBlock), 0, st,
                           T.static_idx.as<int>(), (int)pt.n0a, T.memo.as<double2>(), T.static_val.as<double2>());
        UP_TRY(hipGetLastError());
      }
    }
  }
  if (pt.n0a > 0 && T.memo_codes == 0) return bad("record tables carry static memo indices but the memo is off", hipErrorInvalidValue);
#undef UP_TRY
  return 0;
}

void paired_reset_delta(PairedSet& s) {
  for (const auto& d : s.dirty) if ((size_t)d.slot < s.dirty_of_slot.size()) s.dirty_of_slot[d.slot] = -1;
  s.dirty.clear();
  s.dirty_marked = 0;
  s.dirty_touched.clear(); s.spill_of.clear(); s.spill_pairs.clear(); s.spill_changed = false;
}

int paired_upload_pows(gaml_hip_ctx* c, PairedSet& s) {
  for (int mt = 0; mt < 2; mt++) {
    MateDev& d = s.dev[mt];
    const ShortMate& m = s.mate[mt];
    if (d.pow_n) continue;
    d.pow_n = m.match_pow.size();
    HIP_TRY(c, d.pows.reserve(2 * d.pow_n * sizeof(double)));
    HIP_TRY(c, hipMemcpy(d.pows.p, m.mismatch_pow.data(), d.pow_n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(d.pows.as<double>() + d.pow_n, m.match_pow.data(), d.pow_n * sizeof(double), hipMemcpyHostToDevice));
  }
  return 0;
}

int paired_reserve_delta(gaml_hip_ctx* c, PairedSet& s);
int paired_shadow_upload(gaml_hip_ctx* c, PairedSet& s, hipStream_t st);

// full rebuild on the calling thread: new device order of the pairs, record tables built on the host and uploaded
// a rebuild is also when windows that no scored path set has used since the previous rebuild leave the device tables
// (knob 15 = 1: never). The current path set's windows stay, whatever their marks.
void paired_retire_windows(gaml_hip_ctx* c, PairedSet& s) {
  if (KNOB(c, 15) == 1) return;
  s.planner.mark_used(s.mate, s.image);
  int64_t n = 0;
  for (int mt = 0; mt < 2; mt++) n += s.mate[mt].retire_unused();
  s.retired_windows += n;
  s.planner.note_rebuild(n > 0);
}

// the table length the static memo indices of class 0 are built over (PairTables::static_idx), 0: none -- no memo
// (knob 4, or a floor of 0: the reference then takes log(0)), or knob 19 = 1 (A/B: every class-0 pair resolved per call)
int paired_static_ins_n(const gaml_hip_ctx* c, const PairedSet& s) {
  return (KNOB(c, 4) == 0 && KNOB(c, 19) == 0 && s.floor_positive) ? (int)s.ins_tab.size() : 0;
}

int paired_rebuild_tables(gaml_hip_ctx* c, PairedSet& s, hipStream_t st) {
  paired_reset_delta(s);
  s.full_rebuilds++;
  paired_retire_windows(c, s);
  for (int mt = 0; mt < 2; mt++) s.mate[mt].activated_log.clear();
  const double tb0 = now_us();
  link_mate_windows(s.mate[0], s.mate[1]);
  build_pair_tables(s.mate[0], s.mate[1], s.pt, KNOB(c, 16) != 1, paired_static_ins_n(c, s));
  s.built_keep_dominated = KNOB(c, 16) == 1;
  const double tb1 = now_us();
  HIP_TRY(c, hipStreamSynchronize(st));  // earlier evaluations may still read the old tables
  if (int e = paired_upload_pows(c, s)) return e;
  if (int e = paired_upload_tables(c, s, s.pt, s.tab, st, nullptr)) return e;
  for (int mt = 0; mt < 2; mt++) s.dev[mt].uploaded_generation = s.mate[mt].active_generation;
  if (int e = paired_reserve_delta(c, s)) return e;
  // room for the private copy a later rebuild off this thread takes (paired_snapshot_mate): allocated and touched here,
  // inside a call that takes tens of milliseconds anyway, so that the snapshot itself is a plain copy
  if (KNOB(c, 14) != 1 && s.rebuild.state.load(std::memory_order_acquire) == 0 && !s.rebuild.th.joinable()) {
    if (!s.rebuild.stream) HIP_TRY(c, hipStreamCreateWithFlags(&s.rebuild.stream, hipStreamNonBlocking));  // (a new queue: ~ms)
    for (int mt = 0; mt < 2; mt++) {
      ShortMate& sn = s.rebuild.snap[mt];
      if (sn.lens.size() != s.mate[mt].lens.size()) sn.lens = s.mate[mt].lens;
      const size_t want = (size_t)s.mate[mt].active_records + (size_t)s.mate[mt].active_records / 2 + 65536;
      if (sn.pool.capacity() < want) { sn.pool.clear(); sn.pool.resize(want); sn.pool.clear(); }  // resize touches the pages, clear keeps them
    }
  }
  if (getenv("GAML_HIP_TRACE_HOST")) fprintf(stderr, "rebuild: tables on the host %.1f ms, uploads %.1f ms\n", (tb1 - tb0) * 1e-3, (now_us() - tb1) * 1e-3);
  return 0;
}

// ---- the same off the caller's thread -----------------------------------------------------------------
// what build_pair_tables reads of a mate, copied: lengths, the window headers, and the records of the ACTIVE windows
// (compacted: `first` re-pointed into the copy). The live mate keeps growing meanwhile. The headers (with the active
// flags) are copied when the rebuild is decided; the records -- immutable once a window is aligned -- follow in
// slices of at most `budget` records per evaluation, so that no single call pays for the whole copy (24 MB at cfg3).
void paired_snapshot_begin(const ShortMate& m, ShortMate& out) {
  out.n_global = m.n_global; out.lo = m.lo; out.hi = m.hi;
  if (out.lens.size() != m.lens.size()) out.lens = m.lens;  // read lengths never change
  out.wins = m.wins;
  out.pool.clear();
  if (out.pool.capacity() < (size_t)m.active_records) out.pool.reserve((size_t)m.active_records + (size_t)m.active_records / 4);
  out.active_records = m.active_records;
  out.active_generation = m.active_generation;
}
// returns true when the copy is complete; *next_w = first window not copied yet
bool paired_snapshot_slice(const ShortMate& m, ShortMate& out, size_t* next_w, int64_t budget) {
  size_t w = *next_w;
  for (; w < out.wins.size() && budget > 0; w++) {
    Window& win = out.wins[w];
    if (!win.active) continue;
    const int64_t first = (int64_t)out.pool.size();
    out.pool.insert(out.pool.end(), m.pool.begin() + win.first, m.pool.begin() + win.first + win.count);
    win.first = first;
    budget -= win.count;
  }
  *next_w = w;
  return w >= out.wins.size();
}

void paired_launch_worker(gaml_hip_ctx* c, PairedSet& s) {
  TableRebuild& rb = s.rebuild;
  rb.state.store(1, std::memory_order_release);
  const int device = c->device;
  rb.th = std::thread([c, &s, &rb, device] {
    const double b0 = now_us();
    int rc = hipSetDevice(device) == hipSuccess ? 0 : GAML_HIP_EHIP;
    if (rc) rb.err = "hipSetDevice failed in the rebuild worker";
    rb.sh_dirty.clear();  // the lists of the tables before last (swapped out at the previous take-over): emptied here, off the caller's thread
    if (!rb.sh_touched_pages && rb.sh_dirty.capacity()) {  // ... and their reserved storage's pages touched once, here (paired_reserve_delta does it for the live lists)
      volatile char* base = (volatile char*)rb.sh_dirty.data();
      for (size_t o = 0; o < rb.sh_dirty.capacity() * sizeof(PairedSet::DirtyPair); o += 4096) base[o] = 0;
      rb.sh_touched_pages = true;
    }
    if (!rc) {
      rb.keep_dominated = KNOB(c, 16) == 1;
      build_pair_tables(rb.snap[0], rb.snap[1], rb.pt, !rb.keep_dominated, rb.static_ins_n);
      rc = paired_upload_tables(c, s, rb.pt, rb.tab, rb.stream, &rb.err);
      if (!rc && hipStreamSynchronize(rb.stream) != hipSuccess) { rc = GAML_HIP_EHIP; rb.err = "stream synchronise failed in the rebuild worker"; }
    }
    rb.build_ms = (now_us() - b0) * 1e-3;
    rb.state.store(rc ? 3 : 2, std::memory_order_release);
  });
}

// state 4: the private copy is being taken, a slice per evaluation; then the worker starts (state 1)
int paired_continue_snapshot(gaml_hip_ctx* c, PairedSet& s, bool all_at_once) {
  TableRebuild& rb = s.rebuild;
  const double t0 = now_us();
  const int64_t budget = all_at_once ? INT64_MAX / 4 : 24 * 1024;  // ~0.4 MB of records per mate and evaluation (~40 us; 96 K records were the 250-380 us calls at the tail of an annealing run)
  bool done = true;
  for (int mt = 0; mt < 2; mt++) done = paired_snapshot_slice(s.mate[mt], rb.snap[mt], &rb.next_w[mt], budget) && done;
  rb.snapshot_us += now_us() - t0;
  if (done) paired_launch_worker(c, s);
  return 0;
}

int paired_start_async_rebuild(gaml_hip_ctx* c, PairedSet& s) {
  TableRebuild& rb = s.rebuild;
  const double t0 = now_us();
  if (rb.th.joinable()) rb.th.join();
  paired_retire_windows(c, s);
  link_mate_windows(s.mate[0], s.mate[1]);  // (the snapshot's window headers carry the links)
  rb.static_ins_n = paired_static_ins_n(c, s);
  const double t1 = now_us();
  for (int mt = 0; mt < 2; mt++) {
    paired_snapshot_begin(s.mate[mt], rb.snap[mt]);
    rb.gen_snap[mt] = s.mate[mt].active_generation;
    rb.next_w[mt] = 0;
  }
  rb.after.clear(); rb.after_keep.clear(); rb.sh_next = 0; rb.sh_open = false; rb.sh_records = 0;
  const double t2 = now_us();
  if (!rb.stream) HIP_TRY(c, hipStreamCreateWithFlags(&rb.stream, hipStreamNonBlocking));
  rb.err.clear();
  rb.start_eval = s.eval_count;
  rb.state.store(4, std::memory_order_release);
  rb.snapshot_us = now_us() - t0;
  const int rc = paired_continue_snapshot(c, s, false);
  if (getenv("GAML_HIP_TRACE_HOST"))
    fprintf(stderr, "rebuild start: retire %.0f us, headers %.0f, first slice of the records %.0f\n", t1 - t0, t2 - t1, now_us() - t2);
  return rc;
}

// the worker is done (or: wait for it): the new tables take over; pairs touched by windows activated since the
// snapshot go (back) onto the delta lists, now relative to the new tables
int paired_finish_async_rebuild(gaml_hip_ctx* c, PairedSet& s, hipStream_t st) {
  TableRebuild& rb = s.rebuild;
  if (rb.state.load(std::memory_order_acquire) == 4) { if (int e = paired_continue_snapshot(c, s, true)) return e; }  // (the copy had not finished: take the rest now)
  if (rb.th.joinable()) rb.th.join();
  const int state = rb.state.load(std::memory_order_acquire);
  rb.state.store(0, std::memory_order_release);
  if (state == 3) return fail(c, GAML_HIP_EHIP, "table rebuild worker: " + rb.err);
  if (state != 2) return 0;
  // what this call's planning just activated joins the windows noted since the snapshot (on the old lists as well: the
  // bookkeeping is the same as in any other call), then the lists for the new tables are completed -- usually all
  // but the last few windows are in them already (paired_shadow_advance, a slice per evaluation)
  if (!s.mate[0].activated_log.empty() || !s.mate[1].activated_log.empty()) paired_extend_delta(s, KNOB(c, 16) != 1, true);
  paired_shadow_advance(s, INT64_MAX / 4);
  HIP_TRY(c, hipStreamSynchronize(st));  // launches in flight may still read the old tables
  std::swap(s.tab, rb.tab);
  std::swap(s.pt, rb.pt);
  s.built_keep_dominated = rb.keep_dominated;
  s.dirty.swap(rb.sh_dirty);            // the old lists are emptied off this thread (the next worker does it)
  s.dirty_of_slot.swap(rb.sh_of_slot);
  s.spill_of.swap(rb.sh_spill_of); s.spill_pairs.swap(rb.sh_spill_pairs);
  s.dirty_touched.swap(rb.sh_touched);  // what the slices have not sent to the second store yet: this call's upload
  rb.sh_touched.clear();
  if (rb.sh_dl_slot.p) {
    std::swap(s.dl_slot, rb.sh_dl_slot); std::swap(s.dl_spill, rb.sh_dl_spill);
    for (int mt = 0; mt < 2; mt++) std::swap(s.dl_rec[mt], rb.sh_dl_rec[mt]);
  } else {  // no second store (it is reserved with the first table build unless rebuilds never leave the calling thread): everything from here
    s.dirty_touched.resize(s.dirty.size());
    for (size_t k = 0; k < s.dirty_touched.size(); k++) s.dirty_touched[k] = (int32_t)k;
    s.spill_of.clear(); s.spill_pairs.clear();
  }
  rb.sh_open = false; rb.after.clear(); rb.after_keep.clear(); rb.sh_next = 0; rb.sh_records = 0;
  s.dirty_marked = 0;
  s.spill_changed = true;  // the long lists' CSR goes with the lists
  s.full_rebuilds++;
  s.async_rebuilds++;
  if (getenv("GAML_HIP_TRACE_HOST")) fprintf(stderr, "rebuild (worker): snapshot %.1f ms on the calling thread, build + upload %.1f ms beside it\n", rb.snapshot_us * 1e-3, rb.build_ms);
  return 0;
}

// The delta store and its staging, allocated ONCE (with the first table build, inside a call that takes tens of
// milliseconds anyway): device / pinned allocations cost 0.1-3 ms each, and an annealing run must not meet them in the
// call that happens to activate a large window.
int paired_reserve_delta(gaml_hip_ctx* c, PairedSet& s) {
  if (s.delta_cap) return 0;
  const int64_t np_all = s.mate[0].n_local();
  // four times the rebuild threshold. Right after a take-over the lists hold what was activated while the worker built
  // (often more than the threshold already: the next rebuild starts at once), and as much again arrives before that one
  // takes over -- an annealing run that keeps activating windows at cfg3's rate peaks near 2 x 130 k pairs
  s.delta_cap = (size_t)std::max<int64_t>(4096, np_all / 2) + 8192;
  s.rebuild.sh_dirty.reserve(s.delta_cap);
  HIP_TRY(c, s.dl_slot.reserve(s.delta_cap * sizeof(int32_t)));
  HIP_TRY(c, s.dl_spill.reserve(s.delta_cap * sizeof(int32_t)));
  for (int mt = 0; mt < 2; mt++) HIP_TRY(c, s.dl_rec[mt].reserve(s.delta_cap * 4 * sizeof(RecQuad)));
  if (KNOB(c, 14) != 1) {  // the second store, for the lists that go with a worker's tables
    HIP_TRY(c, s.rebuild.sh_dl_slot.reserve(s.delta_cap * sizeof(int32_t)));
    HIP_TRY(c, s.rebuild.sh_dl_spill.reserve(s.delta_cap * sizeof(int32_t)));
    for (int mt = 0; mt < 2; mt++) HIP_TRY(c, s.rebuild.sh_dl_rec[mt].reserve(s.delta_cap * 4 * sizeof(RecQuad)));
    s.rebuild.sh_spill_of.reserve(s.delta_cap);
  }
  // patches: a call rarely touches more than a few thousand pairs; the largest activations (a long node's twin) ~30 k
  const size_t patch = std::min<size_t>(s.delta_cap, 32768) * sizeof(DeltaPatch);
  HIP_TRY(c, s.dl_patch.reserve(patch));
  for (int k = 0; k < kRing; k++) {
    HIP_TRY(c, s.stage_delta.host[k].reserve(patch));
    if (!s.stage_delta.done[k]) HIP_TRY(c, hipEventCreateWithFlags(&s.stage_delta.done[k], hipEventDisableTiming));  // (not in the call that first needs the slot)
  }
  s.patch_of.assign(s.delta_cap + 4096, -1);  // (1.7 MB written here: the first patch of an annealing run found it unallocated, 0.4 ms)
  HIP_TRY(c, s.delta_dev.reserve((size_t)1 << 20));
  // host lists: never moved while they fill (a DirtyPair is ~200 bytes; growing a vector of 60 k of them costs milliseconds)
  s.dirty.reserve(s.delta_cap);
  // ... and their pages are touched here (85 MB at cfg3, ~20 ms inside a call of seconds): a vector's reserved storage is
  // address space until it is written, and a delta pair made on a fresh page paid the page fault -- 116 of the 212 cycles a
  // record cost when a window was activated, 7 ms of an annealing run's first 1000 calls
  {
    volatile char* base = (volatile char*)s.dirty.data();
    const size_t bytes = s.delta_cap * sizeof(PairedSet::DirtyPair);
    for (size_t o = 0; o < bytes; o += 4096) base[o] = 0;
  }
  s.spill_of.reserve(s.delta_cap);
  s.dirty_touched.reserve(65536);
  for (size_t o = 0; o < s.delta_cap * sizeof(int32_t); o += 4096) ((volatile char*)s.spill_of.data())[o] = 0;     // (pages touched: as above)
  for (size_t o = 0; o < 65536 * sizeof(int32_t); o += 4096) ((volatile char*)s.dirty_touched.data())[o] = 0;
  s.dirty_of_slot.assign((size_t)np_all, -1);  // (3.3 MB at cfg3: touched here, not in the call that first activates a window)
  return 0;
}

// a patch for the lists `touched` names: their fixed-stride device copy (slot, spill index, 4 records per mate). The
// lists are the live ones or those being prepared for a worker's tables (`pt` = the tables the slots refer to).
struct DeltaStore { DevBuf* slot; DevBuf* spill; DevBuf* rec0; DevBuf* rec1; };
// mark_from >= 0: delta pairs [mark_from, dirty.size()) are new on the LIVE tables and need their marks there; when
// the patch names every one of them (the usual case: a new pair is a touched pair) the patch kernel sets them and
// *marked_out becomes true
static int delta_upload_patch(gaml_hip_ctx* c, PairedSet& s, hipStream_t st, const PairTables& pt, const std::vector<PairedSet::DirtyPair>& dirty,
                              std::vector<int32_t>& touched, std::vector<int32_t>& spill_of, std::vector<int32_t>& spill_pairs, bool* spill_changed,
                              const DeltaStore& dev, int64_t mark_from = -1, bool* marked_out = nullptr) {
  const size_t nd = dirty.size();
  {  // ascending and distinct as a rule (new delta pairs are numbered as they are touched): one pass instead of a sort
    bool ordered = true;
    for (size_t k = 1; k < touched.size() && ordered; k++) ordered = touched[k - 1] < touched[k];
    if (!ordered) {
      std::sort(touched.begin(), touched.end());
      touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
    }
  }
  const bool fuse_marks = mark_from >= 0 && (int64_t)nd > mark_from &&
                          (int64_t)(touched.end() - std::lower_bound(touched.begin(), touched.end(), (int32_t)mark_from)) == (int64_t)nd - mark_from;
  if (marked_out) *marked_out = fuse_marks;
  spill_of.resize(nd, -1);
  const size_t np_patch = touched.size();
  // the short form when no touched pair holds more than two records on a mate (nearly always)
  bool short_form = true;
  for (size_t t = 0; t < np_patch && short_form; t++) { const auto& d = dirty[touched[t]]; short_form = d.recs[0].size() <= 2 && d.recs[1].size() <= 2; }
  void* ph = nullptr;
  int pslot = stage_acquire(c, s.stage_delta, np_patch * (short_form ? sizeof(DeltaPatch2) : sizeof(DeltaPatch)), &ph);
  if (pslot < 0) return pslot;
  auto fill = [&](auto* patch, const int K) {
    for (size_t t = 0; t < np_patch; t++) {
      // an entry touches its delta pair (~200 bytes; the pair carries its read lengths): ask for it ahead
      if (t + 16 < np_patch) { const char* q = (const char*)&dirty[touched[t + 16]]; __builtin_prefetch(q); __builtin_prefetch(q + 64); __builtin_prefetch(q + 128); }
      const int32_t dj = touched[t];
      const auto& d = dirty[dj];
      auto& pe = patch[t];
      pe.dj = dj; pe.slot = d.slot; pe.pad = 0;
      const bool lng = d.recs[0].size() > 4 || d.recs[1].size() > 4;
      if (lng) {
        if (spill_of[dj] < 0) { spill_of[dj] = (int32_t)spill_pairs.size(); spill_pairs.push_back(dj); }
        *spill_changed = true;
      }
      pe.spill = spill_of[dj];
      for (int mt = 0; mt < 2; mt++)
        for (int k = 0; k < K; k++) {
          const RecQuad none{-1, 0, 0, 0};
          const RecQuad& r = (!lng && k < (int)d.recs[mt].size()) ? d.recs[mt][k] : none;
          pe.rec[mt][k] = make_int4(r.wid, r.pos, r.flags, r.link);
        }
      // the spare words of the two first records: the pair's read lengths and the lengths of its lists (paired_delta_body)
      pe.rec[0][0].w = (int)d.len12;
      pe.rec[1][0].w = lng ? 0 : (int)(d.recs[0].size() | (d.recs[1].size() << 8));
    }
  };
  if (short_form) fill((DeltaPatch2*)ph, 2); else fill((DeltaPatch*)ph, 4);
  // the kernel reads the patch where the host wrote it (mapped pinned memory; the slot is held until the kernel is through)
  const int n0 = (int)s.pt.class_count[0], n01 = n0 + (int)s.pt.class_count[1], n_main = n01 + (int)s.pt.class_count[2];
  const dim3 pgrid((unsigned)std::min<size_t>((np_patch + kBlock - 1) / kBlock, 256));
  if (short_form)
    hipLaunchKernelGGL((apply_delta_patch_kernel<DeltaPatch2, 2>), pgrid, dim3(kBlock), 0, st,
                       (const DeltaPatch2*)s.stage_delta.host[pslot].dev, (int)np_patch, dev.slot->as<int>(), dev.spill->as<int>(), dev.rec0->as<int4>(), dev.rec1->as<int4>(),
                       fuse_marks ? (int)mark_from : -1, s.tab.rec8[0].as<unsigned long long>(), n0, s.tab.inl[0].as<int4>(), n01, n_main, s.tab.first[0].as<int4>());
  else
    hipLaunchKernelGGL((apply_delta_patch_kernel<DeltaPatch, 4>), pgrid, dim3(kBlock), 0, st,
                       (const DeltaPatch*)s.stage_delta.host[pslot].dev, (int)np_patch, dev.slot->as<int>(), dev.spill->as<int>(), dev.rec0->as<int4>(), dev.rec1->as<int4>(),
                       fuse_marks ? (int)mark_from : -1, s.tab.rec8[0].as<unsigned long long>(), n0, s.tab.inl[0].as<int4>(), n01, n_main, s.tab.first[0].as<int4>());
  HIP_TRY(c, hipGetLastError());
  if (int e = stage_release(c, s.stage_delta, pslot, st)) return e;
  touched.clear();
  return 0;
}

// delta pairs: a patch for the pairs whose lists changed since the last upload (new windows were activated)
int paired_upload_delta(gaml_hip_ctx* c, PairedSet& s, hipStream_t st) {
  if (s.dirty_touched.empty() && !s.spill_changed) return 0;
  if (int e = paired_reserve_delta(c, s)) return e;
  if (s.dirty.size() > s.delta_cap) return fail(c, GAML_HIP_ESTATE, "delta store overflow (rebuild policy violated)");
  if (!s.dirty_touched.empty() && s.patch_ready) {
    // the patch was written with the lists (paired_sync_tables): one dispatch applies it and, when it names every new pair, sets their marks
    const size_t nd = s.dirty.size();
    const bool fuse_marks = s.patch_long.empty() && nd > s.dirty_marked && (size_t)s.patch_new == nd - s.dirty_marked;
    s.spill_of.resize(nd, -1);
    if (s.patch_n > 0) {
      const int n0 = (int)s.pt.class_count[0], n01 = n0 + (int)s.pt.class_count[1], n_main = n01 + (int)s.pt.class_count[2];
      hipLaunchKernelGGL((apply_delta_patch_kernel<DeltaPatch2, 2>), dim3((unsigned)std::min<size_t>(((size_t)s.patch_n + kBlock - 1) / kBlock, 256)), dim3(kBlock), 0, st,
                         (const DeltaPatch2*)s.stage_delta.host[s.patch_slot].dev, s.patch_n, s.dl_slot.as<int>(), s.dl_spill.as<int>(), s.dl_rec[0].as<int4>(), s.dl_rec[1].as<int4>(),
                         fuse_marks ? (int)s.dirty_marked : -1, s.tab.rec8[0].as<unsigned long long>(), n0, s.tab.inl[0].as<int4>(), n01, n_main, s.tab.first[0].as<int4>());
      HIP_TRY(c, hipGetLastError());
    }
    if (int e = stage_release(c, s.stage_delta, s.patch_slot, st)) return e;
    if (fuse_marks) s.dirty_marked = nd;
    if (!s.patch_long.empty()) {  // the few pairs with three or more records on a mate: long form, applied behind the short one (same stream)
      if (int e = delta_upload_patch(c, s, st, s.pt, s.dirty, s.patch_long, s.spill_of, s.spill_pairs, &s.spill_changed,
                                     DeltaStore{&s.dl_slot, &s.dl_spill, &s.dl_rec[0], &s.dl_rec[1]})) return e;
    }
    s.dirty_touched.clear();
    s.patch_ready = false;
  } else if (!s.dirty_touched.empty()) {
    s.patch_ready = false;
    bool marked = false;
    if (int e = delta_upload_patch(c, s, st, s.pt, s.dirty, s.dirty_touched, s.spill_of, s.spill_pairs, &s.spill_changed,
                                   DeltaStore{&s.dl_slot, &s.dl_spill, &s.dl_rec[0], &s.dl_rec[1]}, (int64_t)s.dirty_marked, &marked)) return e;
    if (marked) s.dirty_marked = s.dirty.size();  // (paired_sync_tables marks what a patch did not cover: after a take-over)
  }
  if (s.spill_changed) {  // the few long lists: CSR rebuilt as a whole
    const size_t ns = s.spill_pairs.size();
    size_t dn[2] = {0, 0};
    for (int32_t dj : s.spill_pairs) { dn[0] += s.dirty[dj].recs[0].size(); dn[1] += s.dirty[dj].recs[1].size(); }
    size_t dt = 0;
    for (int mt = 0; mt < 2; mt++) {
      s.delta_off[2 * mt] = dt; dt = align16(dt + (ns + 1) * sizeof(int32_t));
      s.delta_off[2 * mt + 1] = dt; dt = align16(dt + std::max<size_t>(1, dn[mt]) * sizeof(RecQuad));
    }
    s.delta_off[4] = dt; dt = align16(dt + std::max<size_t>(1, ns) * sizeof(int32_t));  // the pairs' slots
    void* dh = nullptr;
    int dslot = stage_acquire(c, s.stage_delta, dt, &dh);
    if (dslot < 0) return dslot;
    for (int mt = 0; mt < 2; mt++) {
      int32_t* of = (int32_t*)((char*)dh + s.delta_off[2 * mt]);
      RecQuad* rc = (RecQuad*)((char*)dh + s.delta_off[2 * mt + 1]);
      int32_t at = 0;
      for (size_t k = 0; k < ns; k++) {
        of[k] = at;
        const auto& l = s.dirty[s.spill_pairs[k]].recs[mt];
        if (l.size()) memcpy(rc + at, l.data(), l.size() * sizeof(RecQuad));
        at += (int32_t)l.size();
      }
      of[ns] = at;
    }
    {
      int32_t* sl = (int32_t*)((char*)dh + s.delta_off[4]);
      for (size_t k = 0; k < ns; k++) sl[k] = s.dirty[s.spill_pairs[k]].slot;
    }
    if (dt > s.delta_dev.cap) { HIP_TRY(c, hipStreamSynchronize(st)); HIP_TRY(c, s.delta_dev.reserve(dt + dt / 2)); }
    if (int e = stage_upload(c, s.stage_delta, dslot, s.delta_dev.p, dt, st)) return e;  // stream order: after the kernels that read the old lists
    if (int e = stage_release(c, s.stage_delta, dslot, st)) return e;
    s.spill_changed = false;
  }
  return 0;
}

// the lists prepared for a worker's tables: what a slice changed goes to the second device store
int paired_shadow_upload(gaml_hip_ctx* c, PairedSet& s, hipStream_t st) {
  TableRebuild& rb = s.rebuild;
  if (rb.sh_touched.empty() || !rb.sh_dl_slot.p || rb.sh_dirty.size() > s.delta_cap) return 0;  // (too many: the take-over reports it)
  bool long_lists = false;  // their CSR is built at the take-over
  return delta_upload_patch(c, s, st, rb.pt, rb.sh_dirty, rb.sh_touched, rb.sh_spill_of, rb.sh_spill_pairs, &long_lists,
                            DeltaStore{&rb.sh_dl_slot, &rb.sh_dl_spill, &rb.sh_dl_rec[0], &rb.sh_dl_rec[1]});
}

// Everything the record tables need before a scoring launch; enqueued on `st`. One call per evaluation (or batch).
int paired_sync_tables(gaml_hip_ctx* c, PairedSet& s, hipStream_t st) {
  static const bool trace = getenv("GAML_HIP_TRACE_HOST") != nullptr;
  const double ts0 = now_us();
  double ts1 = ts0, ts2 = ts0;
  size_t tr_new = 0, tr_touched = 0;
  TableRebuild& rb = s.rebuild;
  const int64_t np = s.mate[0].n_local();
  int rstate = rb.state.load(std::memory_order_acquire);
  s.eval_count++;
  // The new tables take over a FIXED number of evaluations after the worker was started -- not whenever the worker
  // happens to be done: a rebuild changes the order of the final sum (last bits), and equal inputs must give equal
  // outputs run to run (the annealing loop compares likelihoods with strict >). The worker needs ~30 ms at
  // 833 k pairs, 768 evaluations take at least that long; if it is not done by then, this call waits for it.
  const int64_t swap_after = KNOB(c, 14) > 1 ? KNOB(c, 14) : 1152;
  if (rstate != 0 && s.eval_count - rb.start_eval >= swap_after) { if (int e = paired_finish_async_rebuild(c, s, st)) return e; rstate = 0; }
  else if (rstate == 4) { if (int e = paired_continue_snapshot(c, s, false)) return e; rstate = 1; }  // the next slice of the private copy
  else if (rstate != 0) {  // (ready or not: not yet)
    if (rstate == 2) {
      // the worker is done: until the new tables take over, every evaluation re-bases a slice of what was activated since
      // the snapshot onto them, sized to be through a few calls before the take-over
      const int64_t calls_left = swap_after - (s.eval_count - rb.start_eval);
      const int64_t budget = std::max<int64_t>(384, rb.sh_records / std::max<int64_t>(1, calls_left - 8) * 3 / 2);
      if (rb.sh_records > 0 || !rb.sh_open) {
        paired_shadow_advance(s, std::min<int64_t>(budget, 8192));
        if (int e = paired_shadow_upload(c, s, st)) return e;
      }
    }
    rstate = 1;
  }
  const bool first_build = s.dev[0].pow_n == 0;
  bool activated_now = s.dev[0].uploaded_generation != s.mate[0].active_generation || s.dev[1].uploaded_generation != s.mate[1].active_generation;
  s.quiet_calls = activated_now ? 0 : s.quiet_calls + 1;
  if (first_build) {
    if (int e = paired_rebuild_tables(c, s, st)) return e;
    activated_now = false;
  } else {
    // Windows activated since the tables were built put their pairs on the delta lists (one lane per pair, a few ns each per
    // evaluation). The tables are rebuilt -- ~30 ms of host work at 833 k pairs -- when the delta passes pairs / 8, when the
    // cache has been quiet for 64 evaluations with pairs still on the delta path (steady re-scoring: an annealing run adds
    // windows every few calls), or on request. On request the calling thread does it (gaml_hip_compact_tables: "at the next
    // evaluation"); otherwise a worker does, and the evaluations go on over the old tables + delta lists meanwhile.
    const size_t limit = KNOB(c, 6) == 1 ? 0 : (size_t)std::max<int64_t>(4096, np / (KNOB(c, 18) > 0 ? KNOB(c, 18) : 8));
    size_t new_records = 0;
    if (activated_now) for (int mt = 0; mt < 2; mt++) for (int32_t w : s.mate[mt].activated_log) new_records += s.mate[mt].wins[w].count;
    const bool over = activated_now && s.dirty.size() + new_records > limit;
    const bool quiet = !activated_now && !s.dirty.empty() && s.quiet_calls >= 64 && KNOB(c, 6) != 2;
    const bool refold = (KNOB(c, 16) == 1) != s.built_keep_dominated;  // A/B of the table contents: a request rebuilds even without delta pairs
    const bool asked = s.compact_requested && (!s.dirty.empty() || activated_now || refold);
    s.compact_requested = false;
    const bool use_worker = KNOB(c, 14) != 1 && KNOB(c, 6) != 1;
    // the delta store must hold what accumulates while a worker builds; when it cannot, wait for the worker
    const size_t hard = s.delta_cap ? s.delta_cap - 2048 : (size_t)std::max<int64_t>(4096, np / 2);
    const bool overflow = s.dirty.size() + new_records > hard;
    if (asked || ((over || quiet) && !use_worker) || overflow) {
      if (rstate == 1) { if (int e = paired_finish_async_rebuild(c, s, st)) return e; rstate = 0; activated_now = !s.mate[0].activated_log.empty() || !s.mate[1].activated_log.empty(); }
      const bool still = asked || overflow || !use_worker;
      if (still && (!s.dirty.empty() || activated_now || refold)) { if (int e = paired_rebuild_tables(c, s, st)) return e; activated_now = false; }
    } else if ((over || quiet) && rstate == 0) {
      if (int e = paired_start_async_rebuild(c, s)) return e;
      rstate = 1;
      s.quiet_calls = 0;
    }
    ts1 = now_us();
    if (activated_now) {
      // the call's patch is written alongside the lists when nothing is waiting from an earlier call (the usual case)
      PatchSink sink{};
      PatchSink* use = nullptr;
      s.patch_ready = false;
      if (s.delta_cap && s.dirty_touched.empty() && new_records > 0 && new_records <= 32768 && c->device >= 0 && KNOB(c, 17) != 1) {
        void* ph = nullptr;
        const int slot = stage_acquire(c, s.stage_delta, new_records * sizeof(DeltaPatch2), &ph);
        if (slot < 0) return slot;
        if (s.patch_of.size() < s.delta_cap + 4096) s.patch_of.assign(s.delta_cap + 4096, -1);
        s.patch_long.clear();
        sink = PatchSink{(DeltaPatch2*)ph, s.patch_of.data(), (int)new_records, 0, 0, (int64_t)s.dirty_marked, false, &s.patch_long};
        s.patch_slot = slot;
        use = &sink;
      }
      paired_extend_delta(s, KNOB(c, 16) != 1, rstate == 1, use);
      if (use) {
        for (int k = 0; k < sink.n; k++) s.patch_of[sink.buf[k].dj] = -1;  // (the map is all -1 again: entries are found by their pair numbers)
        if (s.dirty.size() > s.patch_of.size()) sink.broken = true;          // (cannot happen: the store's capacity bounds the pairs)
        s.patch_ready = !sink.broken && (sink.n > 0 || !s.patch_long.empty());
        s.patch_n = sink.n; s.patch_new = sink.n_new;
      }
    }
    tr_new = new_records; tr_touched = s.dirty_touched.size();
    ts2 = now_us();
  }
  if (int e = paired_upload_delta(c, s, st)) return e;
  static const double trace_above = getenv("GAML_HIP_TRACE_SYNC_US") ? atof(getenv("GAML_HIP_TRACE_SYNC_US")) : 300.0;
  if (trace && now_us() - ts0 > trace_above)
    fprintf(stderr, "table sync %.0f us: policy %.0f, delta lists %.0f (%zu new records, %zu list entries touched, %zu delta pairs, %zu spill), upload %.0f; worker state %d\n",
            now_us() - ts0, ts1 - ts0, ts2 - ts1, tr_new, tr_touched, s.dirty.size(), s.spill_pairs.size(), now_us() - ts2, rstate);
  const size_t nd = s.dirty.size();
  if (nd > s.dirty_marked) {  // marks stay on the device until the next full build: only new delta pairs need one
    const size_t fresh = nd - s.dirty_marked;
    const int n0 = (int)s.pt.class_count[0], n01 = n0 + (int)s.pt.class_count[1], n_main = n01 + (int)s.pt.class_count[2];
    hipLaunchKernelGGL(mark_dirty_kernel, dim3((unsigned)std::min<size_t>((fresh + kBlock - 1) / kBlock, 256)), dim3(kBlock), 0, st,
                       s.dl_slot.as<int>() + s.dirty_marked, (int)fresh, s.tab.rec8[0].as<unsigned long long>(), n0, s.tab.inl[0].as<int4>(), n01,
                       n_main, s.tab.first[0].as<int4>());
    HIP_TRY(c, hipGetLastError());
    s.dirty_marked = nd;
  }
  return 0;
}

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
      P.cap_w[mt] = need_w[mt] + need_w[mt] / 4 + 2048;
      P.cap_lo[mt] = 2 * need_lo[mt] + 256;
      P.cap_m[mt] = 2 * need_m[mt] + 1024;
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
  sd.gen_bits = nullptr; sd.part_sum = nullptr; sd.part_zero = nullptr;
}

// ---------------------------------------------------------------------------------------------------------
// kernel arguments
// ---------------------------------------------------------------------------------------------------------
struct GridPlan { int blocks0a, blocks0, blocks1, blocks2, blocks_d, main_blocks, ovf_blocks, total_blocks, gen_blocks; int64_t gen_words[4]; };  // gen_words: notes of class 0 / 1 / 2 / the delta pairs

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
    a.spill_off[mt] = (const int*)((const char*)s.delta_dev.p + s.delta_off[2 * mt]);
    a.spill_recs[mt] = (const int4*)((const char*)s.delta_dev.p + s.delta_off[2 * mt + 1]);
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
  const double* ct = s.tab.combo_tabs.as<double>();
  a.pe[0] = ct; a.pe[1] = ct + nc * 64; a.floor_c = ct + 2 * nc * 64; a.logfloor_c = a.floor_c + nc; a.covthr_c = a.logfloor_c + nc;
  a.len_code = s.tab.len_code.as<unsigned char>();
  a.len_combo = s.tab.len_combo.as<uint32_t>();
  a.n_codes = (int)std::min<size_t>(256, s.pt.len_combo.size());
  a.len_combo0 = s.pt.len_combo.empty() ? 0xffffffffu : s.pt.len_combo[0];
  a.logfloor0 = s.pt.len_combo.empty() ? 0.0 : s.logfloor_tab[(s.pt.len_combo[0] & 0xffff) + (s.pt.len_combo[0] >> 16)];  // = logfloor_c[0], by value
  a.memo = s.tab.memo_codes > 0 ? s.tab.memo.as<double2>() : nullptr;
  a.lt_codes = s.tab.memo_codes;
  const size_t nd = s.dirty.size();
  a.spill_slot = (const int*)((const char*)s.delta_dev.p + s.delta_off[4]);
  a.n_spill = (int)s.spill_pairs.size();
  a.n_dirty = (int)nd;
  a.dirty_slots = s.dl_slot.as<int>();
  a.dirty_spill = s.dl_spill.as<int>();
  const int64_t ovf_total = n - n_main + (int64_t)s.spill_pairs.size();  // wave-per-pair items: table pairs with long lists, then delta pairs with long lists
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
  gp.gen_words[0] = (n0a + 63) / 64 + (n0b + 63) / 64; gp.gen_words[1] = (n01 - n0 + 63) / 64; gp.gen_words[2] = (n_main - n01 + 63) / 64;
  gp.gen_words[3] = ((int64_t)nd + 63) / 64;
  gp.gen_blocks = n_main > 0 ? (int)std::min<int64_t>((n_main + kBlock - 1) / kBlock, KNOB(c, 21) > 0 ? KNOB(c, 21) : kMaxBlocks) : 0;
  a.blocks0a = gp.blocks0a;
  a.gen_w0b = (int)((n0a + 63) / 64);
  a.blocks0 = gp.blocks0;
  a.blocks01 = gp.blocks0 + gp.blocks1;
  a.blocks012 = gp.blocks0 + gp.blocks1 + gp.blocks2;
  a.main_blocks = gp.main_blocks;
  a.total_blocks = gp.total_blocks;
  a.gen_w1 = (int)gp.gen_words[0]; a.gen_w2 = (int)(gp.gen_words[0] + gp.gen_words[1]);
  a.gen_wd = (int)(gp.gen_words[0] + gp.gen_words[1] + gp.gen_words[2]);
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
  sd.gen_bits = nullptr; sd.part_sum = nullptr; sd.part_zero = nullptr;
}

void paired_apply_set(PairedArgs& a, const SetDev& sd) {
  for (int mt = 0; mt < 2; mt++) { a.m[mt].occ12 = sd.occ12[mt]; a.m[mt].occ = nullptr; a.occ12[mt] = sd.occ12[mt]; a.m[mt].multi_off = sd.multi_off[mt]; a.m[mt].multi = sd.multi[mt]; }
  a.tfloor_c = sd.tfloor_c; a.tfloor0 = sd.tfloor0; a.two_T = sd.two_T; a.log_two_T = sd.log_two_T;
  a.gen_bits = sd.gen_bits; a.part_sum = sd.part_sum; a.part_zero = sd.part_zero;
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
  if (int e = prepare_paired_tables(c, s)) return e;
  const double tp0 = now_us();
  prepare_paired_tables_host(c, s, p);  // pass 2 (pass 1 ran in eval_begin)
  const double t_after_host = now_us();
  c->prof[1] = t_after_host - tp0;  // thresholds + occurrence tables
  if (int e = paired_sync_tables(c, s, st)) return e;  // cold path: nothing to do on a warm cache
  const double tp1 = now_us();
  c->prof[4] = tp1 - t_after_host;

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
  // some window occurs several times in this path set (or needs the long occurrence form): second launch over
  // the pairs the main kernel notes
  const bool gen_pass = a.n_main > 0 && p.general;
  int gen_blocks = 0;
  if (gen_pass) {
    const size_t bytes = (size_t)(gp.gen_words[0] + gp.gen_words[1] + gp.gen_words[2] + gp.gen_words[3]) * sizeof(unsigned long long);
    if (bytes > s.gen_bits.cap) { HIP_TRY(c, hipStreamSynchronize(st)); HIP_TRY(c, s.gen_bits.reserve(bytes + bytes / 4)); }
    sd.gen_bits = s.gen_bits.as<unsigned long long>();
    gen_blocks = gp.gen_blocks;
  }
  const int n_partials = gp.total_blocks + gen_blocks;
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
    int fin_mode = c->host_results ? 2 : (KNOB(c, 2) ? KNOB(c, 2) - 1 : 1);  // 0: ticket in the kernel (2048 same-address atomics: ~20 us), 1: finisher kernel, 2: host adds the partials
    if (fin_mode == 0 && gen_pass) fin_mode = 1;
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
    if (gen_pass) GAML_LAUNCH_SCORE(false, true);
    else if (fin_mode) GAML_LAUNCH_SCORE(false, false);
    else GAML_LAUNCH_SCORE(true, false);
#undef GAML_LAUNCH_SCORE
    HIP_TRY(c, hipGetLastError());
    if (gen_pass) {
      std::pair<hipEvent_t, hipEvent_t>* gev = nullptr;
      if (ev) { if (int e = take_events(c, &gev, 1)) return e; }  // (timed like the scoring launch: gaml_hip_debug_general_stats)
      hipExtLaunchKernelGGL(paired_general_kernel, dim3(gen_blocks), dim3(kBlock), 0, st, gev ? gev->first : nullptr, gev ? gev->second : nullptr, 0, a, a.total_blocks);
      HIP_TRY(c, hipGetLastError());
    }
    if (fin_mode == 1) {
      hipLaunchKernelGGL(finish_partials_kernel, dim3(1), dim3(kBlock), 0, st, a.part_sum, a.part_zero, n_partials, out4, cov ? -1.0 : 0.0, (double)n);
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
  bool any_general = false;
  for (int k = first; k < first + n_sets; k++) any_general = any_general || (a.n_main > 0 && preps[k].general);
  const size_t gen_bytes = (size_t)(gp.gen_words[0] + gp.gen_words[1] + gp.gen_words[2] + gp.gen_words[3]) * sizeof(unsigned long long);
  if (any_general && gen_bytes * kMaxSets > s.gen_bits.cap) { HIP_TRY(c, hipStreamSynchronize(st)); HIP_TRY(c, s.gen_bits.reserve(gen_bytes * kMaxSets + 64)); }
  const int n_partials = gp.total_blocks + (any_general ? gp.gen_blocks : 0);
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
    if (any_general) ms.set[k].gen_bits = (unsigned long long*)((char*)s.gen_bits.p + (size_t)g * gen_bytes);
  }
  paired_apply_set(a, ms.set[0]);  // (fields every set overrides; harmless defaults)
  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  if (c->event_timing && (c->event_tick++ % c->event_every) == 0) { if (int e = take_events(c, &ev)) return e; }
  hipEvent_t e0 = ev ? ev->first : nullptr, e1 = ev ? ev->second : nullptr;
  const dim3 grid(a.total_blocks), block(kBlock);
  if (any_general) hipExtLaunchKernelGGL((paired_score_multi_kernel<true>), grid, block, 0, st, e0, e1, 0, a, ms);
  else hipExtLaunchKernelGGL((paired_score_multi_kernel<false>), grid, block, 0, st, e0, e1, 0, a, ms);
  HIP_TRY(c, hipGetLastError());
  if (any_general) {
    for (int k = 0; k < n_sets; k++) {  // the notes of every set were written; only sets with such windows have any bit set
      PairedArgs b = a;
      paired_apply_set(b, ms.set[k]);
      hipLaunchKernelGGL(paired_general_kernel, dim3(gp.gen_blocks), dim3(kBlock), 0, st, b, a.total_blocks);
      HIP_TRY(c, hipGetLastError());
    }
  }
  if (!c->event_timing || ev) {
    double rec = 0;
    for (int k = first; k < first + n_sets; k++) rec += (double)preps[k].assembled_records;
    c->stat_algo_bytes += 16.0 * rec + 16.0 * (double)n * n_sets;
    c->stat_launches++;
  }
  return 0;
}

}  // namespace
