// paired_tables.hip.h -- cold path of a paired set: the device record tables follow the alignment-window cache.
// (included by paired_launch.hip.h; one translation unit with gaml_hip.hip)
//
// The reference keeps a window's alignments in aligment_cache_ (graph.cc:911-922, graph.h:427) and re-derives every read's
// list per call through hash maps (GetPositionsOnlyPath graph.cc:535-598). Here the records live in a window-major pool ON
// THE DEVICE and never come back to the host; the host keeps a window's header (count, largest position, where it sits):
//
//   pool_mirror            windows the HOST filed (host aligner, caller-supplied records) are copied into the device pool
//   paired_build_enqueue   the read-major tables as a chain of kernels over the pool (table_build.hip.h)
//   paired_delta_apply     windows activated since the tables were built: their pairs move to the delta lists (delta_dev.hip.h)
//   paired_sync_tables     the policy: delta lists, or a rebuild -- on the calling stream when asked for / when the lists
//                          would overflow, otherwise beside the evaluations on a stream of its own, taking over a FIXED
//                          number of evaluations later (what was activated meanwhile is applied to the new tables then)
#pragma once

namespace {

constexpr int64_t kTakeOverAfter = 96;    // evaluations between the start of a rebuild beside the evaluations and its take-over
constexpr int64_t kDeltaMaxPerCall = 131072;  // more new records than this in one call: rebuild instead of lists

int bits_for(uint64_t v) { int b = 1; while (b < 64 && (v >> b)) b++; return b; }

// ---- device pool -------------------------------------------------------------------------------------------------
int pool_reserve(gaml_hip_ctx* c, PairedSet& s, int mt, int64_t want) {
  MateDev& d = s.dev[mt];
  if ((size_t)want * sizeof(int4) <= d.pool.cap) return 0;
  // growing moves the pool: everything that reads it must be through (cold: the pool leaves room for twice what it holds)
  HIP_TRY(c, hipDeviceSynchronize());
  const size_t bytes = std::max<size_t>((size_t)want * 2, 1 << 16) * sizeof(int4);
  void* np = nullptr;
  HIP_TRY(c, hipMalloc(&np, bytes));
  if (d.pool.p && d.pool_n) HIP_TRY(c, hipMemcpy(np, d.pool.p, (size_t)d.pool_n * sizeof(int4), hipMemcpyDeviceToDevice));
  if (d.pool.p) HIP_TRY(c, hipFree(d.pool.p));
  d.pool.p = np; d.pool.cap = bytes;
  return 0;
}

// windows filed on the host since the last call: into the device pool, on `st`
int pool_mirror(gaml_hip_ctx* c, PairedSet& s, hipStream_t st) {
  for (int mt = 0; mt < 2; mt++) {
    ShortMate& m = s.mate[mt];
    MateDev& d = s.dev[mt];
    if (d.filed_done == m.filed.size()) continue;
    int64_t add = 0;
    for (size_t k = d.filed_done; k < m.filed.size(); k++) add += m.wins[m.filed[k]].count;
    if (int e = pool_reserve(c, s, mt, d.pool_n + add)) return e;
    if (add > 0) {
      const size_t bytes = (size_t)add * sizeof(int4);
      const bool small = bytes <= ((size_t)1 << 20);
      std::vector<int4> big;
      void* hp = nullptr;
      int slot = -1;
      if (small) { slot = stage_acquire(c, s.stage_pool, bytes, &hp); if (slot < 0) return slot; }
      else { big.resize((size_t)add); hp = big.data(); }
      int4* out = (int4*)hp;
      int64_t at = 0;
      for (size_t k = d.filed_done; k < m.filed.size(); k++) {
        Window& w = m.wins[m.filed[k]];
        w.dfirst = d.pool_n + at;
        for (int64_t q = w.first; q < w.first + w.count; q++) {
          const gaml_aligment& r = m.pool[(size_t)q];
          out[at++] = make_int4(m.filed[k], r.position, (r.edit_dist & 0xff) | ((r.orientation & 1) << 8), r.read_id);
        }
      }
      int4* dst = d.pool.as<int4>() + d.pool_n;
      if (small) {
        if (int e = stage_upload(c, s.stage_pool, slot, dst, bytes, st)) return e;
        if (int e = stage_release(c, s.stage_pool, slot, st)) return e;
      } else {
        HIP_TRY(c, hipStreamSynchronize(st));
        HIP_TRY(c, hipMemcpy(dst, hp, bytes, hipMemcpyHostToDevice));
      }
      d.pool_n += add;
    } else {
      for (size_t k = d.filed_done; k < m.filed.size(); k++) m.wins[m.filed[k]].dfirst = d.pool_n;
    }
    d.filed_done = m.filed.size();
  }
  return 0;
}

// ---- per set, once: read lengths, length-combination codes, per-combination tables, memo of pair terms -------------------
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

int paired_upload_statics(gaml_hip_ctx* c, PairedSet& s, hipStream_t st) {
  if (s.statics_uploaded) return 0;
  if (int e = paired_upload_pows(c, s)) return e;
  const int64_t n = s.mate[0].n_local();
  const ShortMate& a = s.mate[0];
  const ShortMate& b = s.mate[1];
  // the distinct (L1, L2) of the set in order of first appearance, at most 256 (a pair beyond them is not compact)
  std::vector<int16_t> lc((size_t)n, -1);
  std::unordered_map<uint32_t, int32_t> combo_id;
  s.pt.len_combo.clear();
  uint32_t last_c = 0; int32_t last_id = -2;
  for (int64_t i = 0; i < n; i++) {
    const uint32_t cb = (uint32_t)a.lens[i] | ((uint32_t)b.lens[i] << 16);
    if (last_id != -2 && cb == last_c) { lc[i] = (int16_t)last_id; continue; }
    auto it = combo_id.find(cb);
    if (it == combo_id.end()) {
      if (s.pt.len_combo.size() >= 256) continue;
      it = combo_id.emplace(cb, (int32_t)s.pt.len_combo.size()).first;
      s.pt.len_combo.push_back(cb);
    }
    lc[i] = (int16_t)it->second;
    last_c = cb; last_id = it->second;
  }
  auto up = [&](DevBuf& d, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = d.reserve(std::max<size_t>(16, bytes));
    if (e != hipSuccess || bytes == 0) return e;
    return hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice);
  };
  HIP_TRY(c, up(s.lcode, lc.data(), lc.size() * sizeof(int16_t)));
  for (int mt = 0; mt < 2; mt++) HIP_TRY(c, up(s.dev[mt].lens, s.mate[mt].lens.data(), s.mate[mt].lens.size() * sizeof(int32_t)));
  HIP_TRY(c, up(s.len_combo_dev, s.pt.len_combo.data(), s.pt.len_combo.size() * sizeof(uint32_t)));
  // per length-combination tables of the compact path: [pe mate 0 | pe mate 1 | floor | logfloor | covthr]
  const size_t nc = std::max<size_t>(1, s.pt.len_combo.size());
  std::vector<double> t(nc * 64 * 2 + nc * 3, 0.0);
  for (size_t ci = 0; ci < s.pt.len_combo.size(); ci++) {
    const int L[2] = {(int)(s.pt.len_combo[ci] & 0xffff), (int)(s.pt.len_combo[ci] >> 16)};
    for (int mt = 0; mt < 2; mt++)
      for (int e = 0; e < 64 && e <= L[mt]; e++)
        t[(size_t)mt * nc * 64 + ci * 64 + e] = s.mate[mt].mismatch_pow[e] * s.mate[mt].match_pow[L[mt] - e];  // graph.cc:1859-1863
    t[2 * nc * 64 + ci] = s.floor_tab[L[0] + L[1]];
    t[2 * nc * 64 + nc + ci] = s.logfloor_tab[L[0] + L[1]];
    t[2 * nc * 64 + 2 * nc + ci] = s.covthr_tab[L[1]];
  }
  HIP_TRY(c, up(s.combo_tabs, t.data(), t.size() * sizeof(double)));
  // memo of the pair terms a single-term pair can take (first 4 length combinations, edits < 7, every tabulated distance):
  // nothing in it depends on a path set or on the records
  s.memo_codes = 0;
  if (KNOB(c, 4) == 0 && s.floor_positive && !s.pt.len_combo.empty() && !s.ins_tab.empty()) {
    const int codes = (int)std::min<size_t>(s.pt.len_combo.size(), kMemoCodes);
    const size_t entries = (size_t)codes * 49 * s.ins_tab.size();
    if (entries <= kMemoMaxEntries) {
      HIP_TRY(c, s.memo.reserve(entries * sizeof(double2)));
      const double* ct = s.combo_tabs.as<double>();
      hipLaunchKernelGGL(logterm_kernel, dim3((unsigned)std::min<size_t>((entries + kBlock - 1) / kBlock, 1024)), dim3(kBlock), 0, st,
                         ct, ct + nc * 64, s.tabs.as<double>(), (int)s.ins_tab.size(), codes, s.memo.as<double2>());
      HIP_TRY(c, hipGetLastError());
      s.memo_codes = codes;
    }
  }
  s.statics_uploaded = true;
  return 0;
}

// the table length the static memo indices of class 0 are built over, 0: none -- no memo (knob 4, or a floor of 0: the
// reference then takes log(0)), or knob 19 = 1 (A/B: every class-0 pair resolved per call)
int paired_static_ins_n(const gaml_hip_ctx* c, const PairedSet& s) {
  return (KNOB(c, 4) == 0 && KNOB(c, 19) == 0 && s.floor_positive && s.memo_codes > 0) ? (int)s.ins_tab.size() : 0;
}

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

// the window whose records always overwrite window w's, as the device kernels want it ({first, count} in the pool; count 0:
// none): w is a junction whose first node is long, and that node's own window is active (host_model.cc dominated_records)
void dominating_window(const ShortMate& m, const Window& w, bool fold, int* dom_first, int* dom_count) {
  *dom_first = 0; *dom_count = 0;
  if (!fold || w.head < 0) return;
  auto it = m.solo_of_node.find(w.head);
  if (it == m.solo_of_node.end()) return;
  const Window& sw = m.wins[it->second];
  if (!sw.active || sw.count == 0 || sw.dfirst < 0) return;
  *dom_first = (int)sw.dfirst; *dom_count = sw.count;
}

// room for the records of a mate's active windows in the tables and the build's scratch: every record aligned so far may be
// active one day, and as many again (an annealing run adds a few hundred records per move; growing costs milliseconds of
// device allocations inside whichever call starts the build that no longer fits: geometric, so it happens a handful of times)
int64_t paired_records_cap(const PairedSet& s, int mt) {
  const int64_t have = s.dev[mt].pool_n;
  return std::max<int64_t>(2 * have, s.mate[0].n_local()) + 65536;
}

// one set of table buffers for n pairs and A[mate] active records (grow-only)
int paired_reserve_tabledev(gaml_hip_ctx* c, TableDev& T, int64_t n, const int64_t* A) {
  HIP_TRY(c, T.cnt.reserve(kTbInts * sizeof(int)));
  for (int mt = 0; mt < 2; mt++) {
    HIP_TRY(c, T.rec8[mt].reserve((size_t)n * sizeof(unsigned long long)));
    HIP_TRY(c, T.first[mt].reserve((size_t)n * sizeof(int4)));
    HIP_TRY(c, T.extra[mt].reserve(std::max<size_t>(1, (size_t)A[mt]) * sizeof(int4)));
    HIP_TRY(c, T.inl[mt].reserve(std::max<size_t>(1, (size_t)std::min<int64_t>(4 * n, 3 * (A[0] + A[1]) + 4)) * sizeof(int4)));
  }
  HIP_TRY(c, T.len_code.reserve(std::max<size_t>(1, (size_t)n))); HIP_TRY(c, T.len12.reserve((size_t)n * sizeof(unsigned)));
  HIP_TRY(c, T.static_idx.reserve((size_t)n * sizeof(int))); HIP_TRY(c, T.static_val.reserve((size_t)n * sizeof(double2)));
  HIP_TRY(c, T.slot_of_read.reserve((size_t)n * sizeof(int))); HIP_TRY(c, T.read_of_slot.reserve((size_t)n * sizeof(unsigned)));
  HIP_TRY(c, T.dirty_of_slot.reserve((size_t)n * sizeof(int)));
  return 0;
}

// ---- table build ---------------------------------------------------------------------------------------------------
// Enqueues one build of the record tables into T on `st`: the ACTIVE windows of both mates as they are now. The host's part
// is the list of those windows (a few thousand headers) and the link between the two mates' windows.
// The chain's launches are numbered as they come (unit 0: the host's part): a build beside the evaluations enqueues a few
// units per evaluation -- the ~65 launches of a build are 250-700 us of host time (a launch into the build's own stream
// costs ~10 us), too much for one annealing call.
constexpr int kBuildAll = 1 << 20;
int paired_build_enqueue(gaml_hip_ctx* c, PairedSet& s, TableDev& T, hipStream_t st, int slice_from = 0, int slice_to = kBuildAll) {
  BuildScratch& B = s.scratch;
  BuildPlan& P = s.build_plan;
  const int64_t n = s.mate[0].n_local();
  RsGate gate;
  gate.from = slice_from; gate.to = slice_to;
  auto in = [&]() { return gate(); };
  if (in()) {
    const bool fold = KNOB(c, 16) != 1;
    T.keep_dominated = !fold;
    T.built = false; T.ros_valid = false;
    HIP_TRY(c, T.cnt.reserve(kTbInts * sizeof(int)));
    HIP_TRY(c, B.h_cnt.reserve(kTbInts * sizeof(int)));
    HIP_TRY(c, hipMemsetAsync(T.cnt.p, 0, kTbInts * sizeof(int), st));
    P.empty = n == 0;
    P.units = 1;
    if (n == 0) { memset(B.h_cnt.p, 0, kTbInts * sizeof(int)); return 0; }
    link_mate_windows(s.mate[0], s.mate[1]);
    for (int mt = 0; mt < 2; mt++) {
      const ShortMate& m = s.mate[mt];
      HIP_TRY(c, B.h_wins[mt].reserve((2 * m.wins.size() + 16384) * sizeof(TbWin)));  // (room to grow: a later build must not meet a pinned allocation)
      TbWin* hw = (TbWin*)B.h_wins[mt].p;
      int k = 0;
      int64_t at = 0;
      for (size_t wid = 0; wid < m.wins.size(); wid++) {
        const Window& w = m.wins[wid];
        if (!w.active || w.count == 0) continue;
        if (w.dfirst < 0) return fail(c, GAML_HIP_ESTATE, "table build: an active window's records are not in the device pool");
        TbWin t;
        t.first = (int)w.dfirst; t.count = w.count; t.wid = (int)wid; t.astart = (int)at;
        dominating_window(m, w, fold, &t.dom_first, &t.dom_count);
        hw[k++] = t;
        at += w.count;
      }
      if (at >= ((int64_t)1 << 31)) return fail(c, GAML_HIP_EINVAL, "table build: more than 2^31 active records per mate");
      P.A[mt] = at; P.n_act[mt] = k;
      HIP_TRY(c, B.wins[mt].reserve((2 * m.wins.size() + 16384) * sizeof(TbWin)));
      if (k) HIP_TRY(c, hipMemcpyAsync(B.wins[mt].p, hw, (size_t)k * sizeof(TbWin), hipMemcpyHostToDevice, st));
    }
    {
      const size_t nw0 = s.mate[0].wins.size();
      HIP_TRY(c, B.h_peer.reserve((2 * nw0 + 16384) * sizeof(int32_t)));
      int32_t* hp = (int32_t*)B.h_peer.p;
      for (size_t w = 0; w < nw0; w++) hp[w] = s.mate[0].wins[w].peer;
      HIP_TRY(c, B.peer0.reserve((2 * nw0 + 16384) * sizeof(int32_t)));
      if (nw0) HIP_TRY(c, hipMemcpyAsync(B.peer0.p, hp, nw0 * sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
    const int64_t* A = P.A;
    HIP_TRY(c, B.cl.reserve((size_t)n)); HIP_TRY(c, B.sidx.reserve((size_t)n * sizeof(int)));
    P.tiles = (unsigned)((n + kTbScanTile - 1) / kTbScanTile);
    HIP_TRY(c, B.tiles.reserve((size_t)P.tiles * sizeof(int)));
    for (int mt = 0; mt < 2; mt++) {
      HIP_TRY(c, B.rstart[mt].reserve((size_t)n * sizeof(int))); HIP_TRY(c, B.rend[mt].reserve((size_t)n * sizeof(int)));
      HIP_TRY(c, B.one[mt].reserve((size_t)n * sizeof(unsigned long long)));
      HIP_TRY(c, B.more[mt].reserve((size_t)n * sizeof(int))); HIP_TRY(c, B.start[mt].reserve((size_t)n * sizeof(int)));
    }
    {  // what exists is kept when it is large enough; what has to grow grows to the pool's size and a quarter (paired_records_cap)
      const size_t inl_need = (size_t)std::min<int64_t>(4 * n, 3 * (A[0] + A[1]) + 4) * sizeof(int4);
      bool fits = T.rec8[0].p != nullptr && T.inl[0].cap >= inl_need && T.inl[1].cap >= inl_need;
      for (int mt = 0; mt < 2; mt++) fits = fits && T.extra[mt].cap >= (size_t)A[mt] * sizeof(int4) && B.v_sorted[mt].cap >= (size_t)A[mt] * sizeof(unsigned);
      const size_t maxA0 = (size_t)std::max<int64_t>(std::max(A[0], A[1]), n);
      fits = fits && B.k_in.cap >= maxA0 * sizeof(rs_u64) && B.hist.cap >= rs_hist_bytes(maxA0);
      const int64_t As[2] = {fits ? A[0] : std::max(A[0], paired_records_cap(s, 0)), fits ? A[1] : std::max(A[1], paired_records_cap(s, 1))};
      if (int e = paired_reserve_tabledev(c, T, n, As)) return e;
      for (int mt = 0; mt < 2; mt++) HIP_TRY(c, B.v_sorted[mt].reserve((size_t)As[mt] * sizeof(unsigned)));
      const size_t maxAs = (size_t)std::max<int64_t>(std::max(As[0], As[1]), n);
      HIP_TRY(c, B.k_in.reserve(maxAs * sizeof(rs_u64))); HIP_TRY(c, B.k_out.reserve(maxAs * sizeof(rs_u64))); HIP_TRY(c, B.k_tmp.reserve(maxAs * sizeof(rs_u64)));
      HIP_TRY(c, B.v_in.reserve(maxAs * sizeof(unsigned))); HIP_TRY(c, B.v_tmp.reserve(maxAs * sizeof(unsigned)));
      HIP_TRY(c, B.hist.reserve(rs_hist_bytes(maxAs)));
    }
    P.ins_n = paired_static_ins_n(c, s);
    P.none1 = (unsigned)s.mate[0].wins.size() + 1; P.none2 = (unsigned)s.mate[1].wins.size() + 1;
  }
  if (P.empty) return 0;
  const int64_t* A = P.A;
  int* cnt = T.cnt.as<int>();
  auto grid = [](int64_t items) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((items + 255) / 256, 4096))); };
  const int read_bits = bits_for((uint64_t)n), read_passes = rs_passes(0, read_bits);
  (void)read_passes;
  for (int mt = 0; mt < 2; mt++) {  // a mate's records ordered by read: keys, then a slice per pass of the sort, then the reads' runs
    if (in()) {
      HIP_TRY(c, hipMemsetAsync(B.rstart[mt].p, 0, (size_t)n * sizeof(int), st));
      HIP_TRY(c, hipMemsetAsync(B.rend[mt].p, 0, (size_t)n * sizeof(int), st));
      if (A[mt] > 0) {
        hipLaunchKernelGGL(tb_keys_kernel, grid(A[mt]), dim3(256), 0, st, s.dev[mt].pool.as<int4>(), B.wins[mt].as<TbWin>(), P.n_act[mt], (int)A[mt], (int)n,
                           B.k_in.as<rs_u64>(), B.v_in.as<unsigned>(), cnt + kTbDropped0 + mt);
        HIP_TRY(c, hipGetLastError());
      }
    }
    if (A[mt] > 0)
      HIP_TRY(c, rs_sort<unsigned>(B.k_in.as<rs_u64>(), B.k_out.as<rs_u64>(), B.k_tmp.as<rs_u64>(), B.v_in.as<unsigned>(), B.v_sorted[mt].as<unsigned>(), B.v_tmp.as<unsigned>(),
                                   (size_t)A[mt], 0, read_bits, B.hist.as<unsigned>(), st, &gate));
    else for (int q = 0; q < 4 * read_passes; q++) (void)in();  // (the numbering does not depend on what a mate holds)
    if (in() && A[mt] > 0) {
      hipLaunchKernelGGL(tb_segments_kernel, grid(A[mt]), dim3(256), 0, st, B.k_out.as<rs_u64>(), (int)A[mt], (int)n, B.rstart[mt].as<int>(), B.rend[mt].as<int>());
      HIP_TRY(c, hipGetLastError());
    }
  }
  const int ins_n = P.ins_n;
  const int memo_codes = (int)std::min<size_t>(s.pt.len_combo.size(), kMemoCodes);
  const int bits1 = bits_for(P.none1), bits2 = bits_for(P.none2);
  if (in()) {  // classes and the pairs' sort keys
    TbClassArgs ca;
    for (int mt = 0; mt < 2; mt++) {
      ca.pool[mt] = s.dev[mt].pool.as<int4>(); ca.vals[mt] = B.v_sorted[mt].as<unsigned>(); ca.rstart[mt] = B.rstart[mt].as<int>(); ca.rend[mt] = B.rend[mt].as<int>();
      ca.lens[mt] = s.dev[mt].lens.as<int>(); ca.one[mt] = B.one[mt].as<unsigned long long>();
    }
    ca.lcode = s.lcode.as<short>(); ca.peer0 = B.peer0.as<int>();
    ca.n = (int)n; ca.ins_n = ins_n; ca.memo_codes = memo_codes;
    ca.memo_fits = ins_n > 0 && (size_t)memo_codes * 49 * (size_t)ins_n <= kMemoMaxEntries;
    ca.cl = B.cl.as<unsigned char>(); ca.sidx = B.sidx.as<int>(); ca.cnt = cnt;
    hipLaunchKernelGGL(tb_class_kernel, grid(n), dim3(256), 0, st, ca);
    HIP_TRY(c, hipGetLastError());
  }
  if (in()) {
    hipLaunchKernelGGL(tb_pairkey_kernel, grid(n), dim3(256), 0, st, B.one[0].as<unsigned long long>(), B.one[1].as<unsigned long long>(), B.cl.as<unsigned char>(), (int)n,
                       (int)kFoldClass2Below, P.none1, P.none2, bits1, bits2, B.k_in.as<rs_u64>(), B.v_in.as<unsigned>(), cnt);
    HIP_TRY(c, hipGetLastError());
  }
  // the device order of the pairs
  HIP_TRY(c, rs_sort<unsigned>(B.k_in.as<rs_u64>(), B.k_out.as<rs_u64>(), B.k_tmp.as<rs_u64>(), B.v_in.as<unsigned>(), T.read_of_slot.as<unsigned>(), B.v_tmp.as<unsigned>(), (size_t)n, 0,
                               3 + bits1 + bits2, B.hist.as<unsigned>(), st, &gate));
  if (in()) {
    TbCompactArgs ka;
    ka.order = T.read_of_slot.as<unsigned>();
    for (int mt = 0; mt < 2; mt++) {
      ka.one[mt] = B.one[mt].as<unsigned long long>(); ka.lens[mt] = s.dev[mt].lens.as<int>(); ka.rstart[mt] = B.rstart[mt].as<int>(); ka.rend[mt] = B.rend[mt].as<int>();
      ka.rec8[mt] = T.rec8[mt].as<unsigned long long>(); ka.more[mt] = B.more[mt].as<int>();
    }
    ka.lcode = s.lcode.as<short>(); ka.sidx = B.sidx.as<int>(); ka.cnt = cnt; ka.n = (int)n;
    ka.slot_of_read = T.slot_of_read.as<int>(); ka.dirty_of_slot = T.dirty_of_slot.as<int>();
    ka.len_code = T.len_code.as<unsigned char>(); ka.static_idx = T.static_idx.as<int>(); ka.len12 = T.len12.as<unsigned>();
    hipLaunchKernelGGL(tb_compact_kernel, grid(n), dim3(256), 0, st, ka);
    HIP_TRY(c, hipGetLastError());
  }
  const unsigned tiles = P.tiles;
  for (int mt = 0; mt < 2; mt++) {
    if (in()) {
      hipLaunchKernelGGL(tb_scan_tiles_kernel, dim3(tiles), dim3(256), 0, st, B.more[mt].as<int>(), cnt, (int)n, B.tiles.as<int>());
      hipLaunchKernelGGL(tb_scan_top_kernel, dim3(1), dim3(1024), 0, st, B.tiles.as<int>(), (int)tiles, cnt + kTbExtras0 + mt);
    }
    if (!in()) continue;
    hipLaunchKernelGGL(tb_scan_apply_kernel, dim3(tiles), dim3(256), 0, st, B.more[mt].as<int>(), cnt, (int)n, B.tiles.as<int>(), B.start[mt].as<int>());
    TbFillArgs fa;
    fa.pool = s.dev[mt].pool.as<int4>(); fa.vals = B.v_sorted[mt].as<unsigned>(); fa.rstart = B.rstart[mt].as<int>(); fa.rend = B.rend[mt].as<int>();
    fa.order = T.read_of_slot.as<unsigned>(); fa.start = B.start[mt].as<int>(); fa.cnt = cnt; fa.n = (int)n;
    fa.first = T.first[mt].as<int4>(); fa.extra = T.extra[mt].as<int4>(); fa.inl = T.inl[mt].as<int4>();
    hipLaunchKernelGGL(tb_fill16_kernel, grid(n), dim3(256), 0, st, fa);
    HIP_TRY(c, hipGetLastError());
  }
  if (in()) {
    if (ins_n > 0) {
      hipLaunchKernelGGL(tb_static_values_kernel, grid(n), dim3(256), 0, st, T.static_idx.as<int>(), cnt, s.memo.as<double2>(), T.static_val.as<double2>());
      HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipMemcpyAsync(B.h_cnt.p, T.cnt.p, kTbInts * sizeof(int), hipMemcpyDeviceToHost, st));
  }
  P.units = gate.next;
  return 0;
}

// the build's stream has completed: what the host keeps of the new tables
int paired_build_collect(gaml_hip_ctx* c, PairedSet& s, TableDev& T) {
  const int* h = (const int*)s.scratch.h_cnt.p;
  for (int k = 0; k < 4; k++) T.class_count[k] = h[kTbClass0 + k];
  T.n0a = h[kTbN0a];
  T.extras[0] = h[kTbExtras0]; T.extras[1] = h[kTbExtras1];
  T.dropped[0] = h[kTbDropped0]; T.dropped[1] = h[kTbDropped1];
  const int64_t n = s.mate[0].n_local();
  if (T.class_count[0] + T.class_count[1] + T.class_count[2] + T.class_count[3] != n)
    return fail(c, GAML_HIP_ESTATE, "table build: the classes do not add up to the pairs");
  if (T.n0a > 0 && s.memo_codes == 0) return fail(c, GAML_HIP_ESTATE, "record tables carry static memo indices but the memo is off");
  T.built = true;
  return 0;
}

void paired_adopt_tables(PairedSet& s) {  // s.tab is the live set of buffers from here on
  for (int k = 0; k < 4; k++) s.pt.class_count[k] = s.tab.class_count[k];
  s.pt.n0a = s.tab.n0a;
  s.pt.dropped_records[0] = s.tab.dropped[0]; s.pt.dropped_records[1] = s.tab.dropped[1];
}

// ---- delta store -----------------------------------------------------------------------------------------------------
int paired_reserve_delta(gaml_hip_ctx* c, PairedSet& s) {
  if (s.delta_cap) return 0;
  const int64_t np_all = s.mate[0].n_local();
  // four times the rebuild threshold: the lists also hold what is activated while a rebuild runs beside the evaluations
  s.delta_cap = (size_t)std::max<int64_t>(4096, np_all / 2) + 8192;
  s.cap_spill = (size_t)(65536 + np_all / 16);
  s.cap_sprec = (size_t)((2 << 20) + np_all);
  HIP_TRY(c, s.dl_slot.reserve(s.delta_cap * sizeof(int32_t)));
  HIP_TRY(c, s.dl_spill.reserve(s.delta_cap * sizeof(int32_t)));
  for (int mt = 0; mt < 2; mt++) {
    HIP_TRY(c, s.dl_rec[mt].reserve(s.delta_cap * 4 * sizeof(int4)));
    HIP_TRY(c, s.sp_rng[mt].reserve(s.cap_spill * sizeof(int2)));
    HIP_TRY(c, s.sp_rec[mt].reserve(s.cap_sprec * sizeof(int4)));
  }
  HIP_TRY(c, s.sp_slot.reserve(s.cap_spill * sizeof(int32_t)));
  HIP_TRY(c, s.dstate.reserve(kDsInts * sizeof(int)));
  HIP_TRY(c, hipMemset(s.dstate.p, 0, kDsInts * sizeof(int)));
  HIP_TRY(c, s.dl_bins.reserve((size_t)kDlBins * kDlBinCap * sizeof(unsigned long long)));
  HIP_TRY(c, s.dl_bin_count.reserve(kDlBins * sizeof(int)));
  HIP_TRY(c, hipMemset(s.dl_bin_count.p, 0, kDlBins * sizeof(int)));
  HIP_TRY(c, s.dl_blk_tot.reserve((4 * kDlBins + 8) * sizeof(int)));
  HIP_TRY(c, s.dl_wlist.reserve(16384 * sizeof(DlWin)));
  HIP_TRY(c, hipMemset(s.dl_blk_tot.p, 0, (4 * kDlBins + 8) * sizeof(int)));
  HIP_TRY(c, s.h_dstate.reserve(64));
  memset(s.h_dstate.p, 0, 64);
  return 0;
}

// the exact counts of the lists, when the device has written them since the last maintenance launch (a blocking call has
// returned, or the caller synchronised): otherwise the host goes on with its upper bounds -- which are a function of the
// call sequence alone, like everything that decides the grid
void paired_refresh_counts(PairedSet& s) {
  if (!s.h_dstate.p) return;
  const volatile int* h = (const volatile int*)s.h_dstate.p;
  if (h[kDsSeq] != s.dl_seq) return;
  std::atomic_thread_fence(std::memory_order_acquire);
  s.nd_est = h[kDsDirty];
  s.ns_est = h[kDsSpill];
  s.delta_left_out = s.delta_left_out_base + h[6];
  s.spill_may_grow = false;
}
bool paired_delta_overflowed(const PairedSet& s) {
  if (!s.h_dstate.p) return false;
  const volatile int* h = (const volatile int*)s.h_dstate.p;
  return h[kDsSeq] == s.dl_seq && h[kDsOverflow] != 0;
}

// the records of `wins` ((mate, window), active, in the device pool) onto the delta lists that go with the tables T
int paired_delta_apply(gaml_hip_ctx* c, PairedSet& s, TableDev& T, std::vector<std::pair<int32_t, int32_t>>& wins, hipStream_t st) {
  if (wins.empty()) return 0;
  if (int e = paired_reserve_delta(c, s)) return e;
  const bool fold = KNOB(c, 16) != 1;
  std::sort(wins.begin(), wins.end());  // mate 0's windows first, each mate's by window id: a pair's new records arrive in table order
  DlArgs a;
  memset(&a, 0, sizeof(a));
  for (int mt = 0; mt < 2; mt++) {
    a.pool[mt] = s.dev[mt].pool.as<int4>();
    a.rec8[mt] = T.rec8[mt].as<unsigned long long>(); a.first[mt] = T.first[mt].as<int4>(); a.extra[mt] = T.extra[mt].as<int4>();
    a.dl_rec[mt] = s.dl_rec[mt].as<int4>(); a.sp_rng[mt] = s.sp_rng[mt].as<int2>(); a.sp_rec[mt] = s.sp_rec[mt].as<int4>();
  }
  a.inl0 = T.inl[0].as<int4>();
  a.len_code = T.len_code.as<unsigned char>(); a.len_combo = s.len_combo_dev.as<unsigned>(); a.len12 = T.len12.as<unsigned>();
  a.slot_of_read = T.slot_of_read.as<int>(); a.dirty_of_slot = T.dirty_of_slot.as<int>();
  a.n0 = (int)T.class_count[0]; a.n01 = a.n0 + (int)T.class_count[1]; a.n_main = a.n01 + (int)T.class_count[2];
  a.dl_slot = s.dl_slot.as<int>(); a.dl_spill = s.dl_spill.as<int>(); a.sp_slot = s.sp_slot.as<int>();
  a.state = s.dstate.as<int>(); a.host_state = (int*)s.h_dstate.dev;
  a.cap_pairs = (int)s.delta_cap; a.cap_spill = (int)s.cap_spill; a.cap_sprec = (int)s.cap_sprec;
  a.slot_bits = bits_for((uint64_t)s.mate[0].n_local());
  a.bins = s.dl_bins.as<unsigned long long>(); a.bin_count = s.dl_bin_count.as<int>(); a.blk_tot = s.dl_blk_tot.as<int>();
  int64_t all_records = 0;
  for (const auto& mw : wins) all_records += s.mate[mw.first].wins[mw.second].count;
  const bool multi_block = (all_records > 3000 || wins.size() > (size_t)kDlMaxWins) && KNOB(c, 22) != 1;  // knob 22 = 1: one-block launches only (A/B, tests)
  const int max_recs = multi_block ? kDlMbMaxRecs : kDlMaxRecs;
  std::vector<DlWin> big;  // a multi-block launch's window list when the argument block cannot hold it
#ifdef GAML_HIP_DEV
  static const bool stamp = getenv("GAML_DL_STAMPS") != nullptr;
  if (stamp) { HIP_TRY(c, s.dl_stamps.reserve(64)); a.stamps = s.dl_stamps.as<unsigned long long>(); }
#endif
  auto flush = [&]() -> int {
    if (a.n_wins == 0) return 0;
    a.seq = ++s.dl_seq;
    if (multi_block) {
      a.wlist = nullptr;
      if (a.n_wins > kDlMaxWins) {  // the list goes through a staging slot into device memory
        void* hp = nullptr;
        const size_t bytes = big.size() * sizeof(DlWin);
        const int slot = stage_acquire(c, s.stage_pool, align16(bytes), &hp);
        if (slot < 0) return slot;
        memcpy(hp, big.data(), bytes);
        if (align16(bytes) > s.dl_wlist.cap) { HIP_TRY(c, hipStreamSynchronize(st)); HIP_TRY(c, s.dl_wlist.reserve(2 * align16(bytes))); }
        if (int e = stage_upload(c, s.stage_pool, slot, s.dl_wlist.p, align16(bytes), st)) return e;
        if (int e = stage_release(c, s.stage_pool, slot, st)) return e;
        a.wlist = s.dl_wlist.as<DlWin>();
      }
      hipLaunchKernelGGL(delta_mb_begin_kernel, dim3(1), dim3(64), 0, st, a.blk_tot);
      hipLaunchKernelGGL(delta_mb_keys_kernel, dim3((unsigned)std::min(256, (a.n_total + 255) / 256)), dim3(256), 0, st, a);
      hipLaunchKernelGGL((delta_apply_kernel<8, 1>), dim3(kDlBins), dim3(kDlThreads), 0, st, a);
      hipLaunchKernelGGL((delta_apply_kernel<8, 2>), dim3(kDlBins), dim3(kDlThreads), 0, st, a);
      HIP_TRY(c, hipGetLastError());
#ifdef GAML_HIP_DEV
      if (stamp) { const double t0 = now_us(); HIP_TRY(c, hipStreamSynchronize(st)); fprintf(stderr, "delta launch (multi-block) of %d records / %d windows: %.1f us until the stream is through\n", a.n_total, a.n_wins, now_us() - t0); }
#endif
      s.nd_est += a.n_total;
      s.spill_may_grow = true;
      a.n_wins = 0; a.n_total = 0;
      big.clear();
      return 0;
    }
    if (a.n_total <= kDlThreads) { int thr = 64; while (thr < a.n_total) thr <<= 1; hipLaunchKernelGGL(delta_apply_kernel<1>, dim3(1), dim3(thr), 0, st, a); }
    else if (a.n_total <= 2 * kDlThreads) hipLaunchKernelGGL(delta_apply_kernel<2>, dim3(1), dim3(kDlThreads), 0, st, a);
    else if (a.n_total <= 4 * kDlThreads) hipLaunchKernelGGL(delta_apply_kernel<4>, dim3(1), dim3(kDlThreads), 0, st, a);
    else hipLaunchKernelGGL(delta_apply_kernel<8>, dim3(1), dim3(kDlThreads), 0, st, a);
    HIP_TRY(c, hipGetLastError());
#ifdef GAML_HIP_DEV
    if (stamp) {
      unsigned long long z[8];
      HIP_TRY(c, hipStreamSynchronize(st));
      HIP_TRY(c, hipMemcpy(z, s.dl_stamps.p, sizeof(z), hipMemcpyDeviceToHost));
      fprintf(stderr, "delta launch of %d records / %d windows: keys %.1f, sort %.1f, heads %.1f, scan %.1f, merge %.1f, counters %.1f us\n", a.n_total, a.n_wins, (z[1] - z[0]) * 0.01, (z[2] - z[1]) * 0.01,
              (z[3] - z[2]) * 0.01, (z[4] - z[3]) * 0.01, (z[5] - z[4]) * 0.01, (z[6] - z[5]) * 0.01);
    }
#endif
    s.nd_est += a.n_total;
    s.spill_may_grow = true;
    a.n_wins = 0; a.n_total = 0;
    return 0;
  };
  const int max_wins = multi_block ? 16384 : kDlMaxWins;
  for (const auto& mw : wins) {
    const ShortMate& m = s.mate[mw.first];
    const Window& w = m.wins[mw.second];
    if (w.count == 0) continue;
    if (w.dfirst < 0) return fail(c, GAML_HIP_ESTATE, "delta lists: an activated window's records are not in the device pool");
    int dom_first, dom_count;
    dominating_window(m, w, fold, &dom_first, &dom_count);
    int done = 0;
    while (done < w.count) {  // (a window larger than a launch holds is cut: the lists compose)
      if (a.n_wins == max_wins || a.n_total == max_recs) { if (int e = flush()) return e; }
      const int take = std::min(w.count - done, max_recs - a.n_total);
      const DlWin dw{mw.first, mw.second, (int)w.dfirst + done, take, dom_first, dom_count, a.n_total};
      if (a.n_wins < kDlMaxWins) a.w[a.n_wins] = dw;
      if (multi_block) big.push_back(dw);
      a.n_wins++;
      a.n_total += take;
      done += take;
    }
  }
  if (int e = flush()) return e;
  s.delta_updates++;
  return 0;
}

// every maintenance kernel once, on no records (the counters keep their values): the first launch of a kernel costs the
// runtime 50-150 us -- paid here, inside the first table build, not in the annealing call that first needs the variant
int paired_warm_delta_kernels(gaml_hip_ctx* c, PairedSet& s, TableDev& T, hipStream_t st) {
  DlArgs a;
  memset(&a, 0, sizeof(a));
  for (int mt = 0; mt < 2; mt++) {
    a.pool[mt] = s.dev[mt].pool.as<int4>();
    a.rec8[mt] = T.rec8[mt].as<unsigned long long>(); a.first[mt] = T.first[mt].as<int4>(); a.extra[mt] = T.extra[mt].as<int4>();
    a.dl_rec[mt] = s.dl_rec[mt].as<int4>(); a.sp_rng[mt] = s.sp_rng[mt].as<int2>(); a.sp_rec[mt] = s.sp_rec[mt].as<int4>();
  }
  a.inl0 = T.inl[0].as<int4>();
  a.len_code = T.len_code.as<unsigned char>(); a.len_combo = s.len_combo_dev.as<unsigned>(); a.len12 = T.len12.as<unsigned>();
  a.slot_of_read = T.slot_of_read.as<int>(); a.dirty_of_slot = T.dirty_of_slot.as<int>();
  a.n0 = (int)T.class_count[0]; a.n01 = a.n0 + (int)T.class_count[1]; a.n_main = a.n01 + (int)T.class_count[2];
  a.dl_slot = s.dl_slot.as<int>(); a.dl_spill = s.dl_spill.as<int>(); a.sp_slot = s.sp_slot.as<int>();
  a.state = s.dstate.as<int>(); a.host_state = (int*)s.h_dstate.dev;
  a.cap_pairs = (int)s.delta_cap; a.cap_spill = (int)s.cap_spill; a.cap_sprec = (int)s.cap_sprec;
  a.bins = s.dl_bins.as<unsigned long long>(); a.bin_count = s.dl_bin_count.as<int>(); a.blk_tot = s.dl_blk_tot.as<int>();
  a.seq = s.dl_seq;
  hipLaunchKernelGGL(delta_apply_kernel<1>, dim3(1), dim3(64), 0, st, a);
  hipLaunchKernelGGL(delta_apply_kernel<2>, dim3(1), dim3(kDlThreads), 0, st, a);
  hipLaunchKernelGGL(delta_apply_kernel<4>, dim3(1), dim3(kDlThreads), 0, st, a);
  hipLaunchKernelGGL(delta_apply_kernel<8>, dim3(1), dim3(kDlThreads), 0, st, a);
  hipLaunchKernelGGL(delta_mb_begin_kernel, dim3(1), dim3(64), 0, st, a.blk_tot);
  hipLaunchKernelGGL(delta_mb_keys_kernel, dim3(1), dim3(256), 0, st, a);
  hipLaunchKernelGGL((delta_apply_kernel<8, 1>), dim3(kDlBins), dim3(kDlThreads), 0, st, a);
  hipLaunchKernelGGL((delta_apply_kernel<8, 2>), dim3(kDlBins), dim3(kDlThreads), 0, st, a);
  HIP_TRY(c, hipGetLastError());
  return 0;
}

int paired_delta_reset(gaml_hip_ctx* c, PairedSet& s, hipStream_t st) {
  if (int e = paired_reserve_delta(c, s)) return e;
  const int seq = ++s.dl_seq;
  hipLaunchKernelGGL(delta_reset_kernel, dim3(1), dim3(64), 0, st, s.dstate.as<int>(), (int*)s.h_dstate.dev, seq);
  HIP_TRY(c, hipGetLastError());
  s.nd_est = 0; s.ns_est = 0; s.spill_may_grow = false;
  s.delta_left_out_base = s.delta_left_out;  // (the device counter starts again at zero)
  return 0;
}

// ---- rebuilds ----------------------------------------------------------------------------------------------------------
int paired_build_continue(gaml_hip_ctx* c, PairedSet& s, bool rest);
// the build's stream: lowest priority -- its workgroups yield to the evaluations' (whose stream has the highest)
hipError_t paired_build_stream(hipStream_t* out) {
  int lo = 0, hi = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);  // (numerically: lo is the LEAST urgent)
  if (e != hipSuccess) return e;
  return hipStreamCreateWithPriority(out, hipStreamNonBlocking, lo);
}

// Every buffer the record tables, their builds and their delta lists need, for a pool of up to two records per read and mate
// -- allocated AND used once (a 4-byte fill): a device allocation becomes usable by a queue when it is first touched, and the
// ~45 buffers of a first build cost the first evaluation 15-20 ms of an idle device at cfg3 (rocprofv3 --hip-trace: the
// build's first dispatch started 20 ms after it was enqueued). Called when a paired read set is added (the reference, too,
// sets its read sets up before it scores anything) and again after every build on the calling stream (no-ops then, unless
// the pool has outgrown the assumption).
int paired_prereserve(gaml_hip_ctx* c, PairedSet& s, hipStream_t st) {
  const int64_t n = s.mate[0].n_local();
  if (n == 0 || c->device < 0) return 0;
  TableRebuild& rb = s.rebuild;
  BuildScratch& B = s.scratch;
  const bool first = !s.prereserved;
  const int64_t A2[2] = {std::max<int64_t>(paired_records_cap(s, 0), 2 * n + 65536), std::max<int64_t>(paired_records_cap(s, 1), 2 * n + 65536)};
  if (first) { for (int mt = 0; mt < 2; mt++) { if (int e = pool_reserve(c, s, mt, n + n / 8)) return e; } }
  if (first || !s.tab.rec8[0].p) { if (int e = paired_reserve_tabledev(c, s.tab, n, A2)) return e; }
  if ((first || !rb.tab.rec8[0].p) && KNOB(c, 14) != 1) { if (int e = paired_reserve_tabledev(c, rb.tab, n, A2)) return e; }
  if (first || !B.k_in.p) {
    const size_t maxA = (size_t)std::max<int64_t>(std::max(A2[0], A2[1]), n);
    HIP_TRY(c, B.k_in.reserve(maxA * sizeof(rs_u64))); HIP_TRY(c, B.k_out.reserve(maxA * sizeof(rs_u64))); HIP_TRY(c, B.k_tmp.reserve(maxA * sizeof(rs_u64)));
    HIP_TRY(c, B.v_in.reserve(maxA * sizeof(unsigned))); HIP_TRY(c, B.v_tmp.reserve(maxA * sizeof(unsigned))); HIP_TRY(c, B.hist.reserve(rs_hist_bytes(maxA)));
    for (int mt = 0; mt < 2; mt++) HIP_TRY(c, B.v_sorted[mt].reserve((size_t)A2[mt] * sizeof(unsigned)));
  }
  if (first) {
    HIP_TRY(c, B.cl.reserve((size_t)n)); HIP_TRY(c, B.sidx.reserve((size_t)n * sizeof(int)));
    HIP_TRY(c, B.tiles.reserve((size_t)((n + kTbScanTile - 1) / kTbScanTile) * sizeof(int)));
    for (int mt = 0; mt < 2; mt++) {
      HIP_TRY(c, B.rstart[mt].reserve((size_t)n * sizeof(int))); HIP_TRY(c, B.rend[mt].reserve((size_t)n * sizeof(int)));
      HIP_TRY(c, B.one[mt].reserve((size_t)n * sizeof(unsigned long long)));
      HIP_TRY(c, B.more[mt].reserve((size_t)n * sizeof(int))); HIP_TRY(c, B.start[mt].reserve((size_t)n * sizeof(int)));
      HIP_TRY(c, B.wins[mt].reserve((size_t)131072 * sizeof(TbWin))); HIP_TRY(c, B.h_wins[mt].reserve((size_t)131072 * sizeof(TbWin)));
    }
    HIP_TRY(c, B.peer0.reserve((size_t)131072 * sizeof(int32_t))); HIP_TRY(c, B.h_peer.reserve((size_t)131072 * sizeof(int32_t)));
    HIP_TRY(c, B.h_cnt.reserve(kTbInts * sizeof(int)));
    if (int e = paired_reserve_delta(c, s)) return e;
  }
  for (int k = 0; k < kRing; k++) {  // (pinned allocations take milliseconds: not in the annealing call that first stages a window list)
    HIP_TRY(c, s.stage_pool.host[k].reserve((size_t)1 << 20));
    if (!s.stage_pool.done[k]) HIP_TRY(c, hipEventCreateWithFlags(&s.stage_pool.done[k], hipEventDisableTiming));
  }
  if (KNOB(c, 14) != 1) {
    if (!rb.stream) HIP_TRY(c, paired_build_stream(&rb.stream));
    if (!rb.done) HIP_TRY(c, hipEventCreateWithFlags(&rb.done, hipEventDisableTiming));
    if (!rb.mark) HIP_TRY(c, hipEventCreateWithFlags(&rb.mark, hipEventDisableTiming));
  }
  if (first) {
    DevBuf* all[] = {&s.dev[0].pool, &s.dev[1].pool, &B.k_in, &B.k_out, &B.k_tmp, &B.v_in, &B.v_tmp, &B.hist, &B.v_sorted[0], &B.v_sorted[1], &B.rstart[0], &B.rstart[1],
                     &B.rend[0], &B.rend[1], &B.one[0], &B.one[1], &B.cl, &B.sidx, &B.more[0], &B.more[1], &B.start[0], &B.start[1], &B.tiles, &B.wins[0], &B.wins[1], &B.peer0,
                     &s.dl_slot, &s.dl_spill, &s.dl_rec[0], &s.dl_rec[1], &s.sp_rng[0], &s.sp_rng[1], &s.sp_rec[0], &s.sp_rec[1], &s.sp_slot, &s.dl_bins, &s.dl_wlist};
    for (DevBuf* b : all) if (b->p) HIP_TRY(c, hipMemsetAsync(b->p, 0, 4, st));
    for (TableDev* T : {&s.tab, &rb.tab}) {
      DevBuf* tb[] = {&T->cnt, &T->rec8[0], &T->rec8[1], &T->first[0], &T->first[1], &T->extra[0], &T->extra[1], &T->inl[0], &T->inl[1], &T->len_code, &T->len12,
                      &T->static_idx, &T->static_val, &T->slot_of_read, &T->read_of_slot, &T->dirty_of_slot};
      for (DevBuf* b : tb) if (b->p) HIP_TRY(c, hipMemsetAsync(b->p, 0, 4, st));
    }
    HIP_TRY(c, hipStreamSynchronize(st));
    s.prereserved = true;
  }
  return 0;
}

// on the calling stream: the spare buffers are built from the windows that are active now and take over at once
int paired_rebuild_tables(gaml_hip_ctx* c, PairedSet& s, hipStream_t st) {
  TableRebuild& rb = s.rebuild;
  if (rb.active) {  // a build beside the evaluations is under way: let it finish (its buffers are the spare ones), then discard it
    if (int e = paired_build_continue(c, s, true)) return e;
    HIP_TRY(c, hipEventSynchronize(rb.done));
    rb.active = false; rb.after.clear();
  }
  s.full_rebuilds++;
  rb.retired = false;
  gpu_probe(st, c->warm_buf.p, "before retire_windows");
  paired_retire_windows(c, s);
  gpu_probe(st, c->warm_buf.p, "after retire_windows");
  for (int mt = 0; mt < 2; mt++) s.mate[mt].activated_log.clear();
  const double tb0 = now_us();
  if (int e = paired_upload_statics(c, s, st)) return e;
  gpu_probe(st, c->warm_buf.p, "before build enqueue");
  const double tb1 = now_us();
  if (int e = paired_build_enqueue(c, s, rb.tab, st)) return e;
  const double tb2 = now_us();
  HIP_TRY(c, hipStreamSynchronize(st));
  const double tb3 = now_us();
  if (int e = paired_build_collect(c, s, rb.tab)) return e;
  std::swap(s.tab, rb.tab);
  paired_adopt_tables(s);
  if (int e = paired_delta_reset(c, s, st)) return e;
  const double tb4 = now_us();
  for (int mt = 0; mt < 2; mt++) s.dev[mt].uploaded_generation = s.mate[mt].active_generation;
  // Everything a later call would otherwise allocate (device allocations cost 0.1-3 ms each; an annealing run must not meet
  // them in the call that happens to activate a window or to start a rebuild): paired_prereserve -- normally done when the
  // read set was added; here for what a larger pool than it assumed makes necessary
  if (int e = paired_prereserve(c, s, st)) return e;
  if (!s.kernels_warm && s.mate[0].n_local() > 0) { if (int e = paired_warm_delta_kernels(c, s, s.tab, st)) return e; s.kernels_warm = true; }
  if (getenv("GAML_HIP_TRACE_HOST"))
    fprintf(stderr, "rebuild on the calling stream: %.2f ms (statics %.2f, reserve + enqueue %.2f, device %.2f, collect + reset %.2f, spare buffers / warm-up %.2f)\n", (now_us() - tb0) * 1e-3,
            (tb1 - tb0) * 1e-3, (tb2 - tb1) * 1e-3, (tb3 - tb2) * 1e-3, (tb4 - tb3) * 1e-3, (now_us() - tb4) * 1e-3);
  return 0;
}

// beside the evaluations: the build runs on a stream of its own; the evaluations go on over the old tables + delta lists
int paired_start_async_rebuild(gaml_hip_ctx* c, PairedSet& s, hipStream_t st) {
  TableRebuild& rb = s.rebuild;
  const double t0 = now_us();
  if (!rb.retired) {  // first of two evaluations: unused windows leave (O(windows) on the host); the window lists come with the next
    paired_retire_windows(c, s);
    rb.retired = true;
    return 0;
  }
  rb.retired = false;
  const double t1 = now_us();
  if (!rb.stream) HIP_TRY(c, paired_build_stream(&rb.stream));
  if (!rb.done) HIP_TRY(c, hipEventCreateWithFlags(&rb.done, hipEventDisableTiming));
  if (!rb.mark) HIP_TRY(c, hipEventCreateWithFlags(&rb.mark, hipEventDisableTiming));
  // the build reads the pool as the calling stream has filled it, and overwrites buffers earlier launches may still read
  HIP_TRY(c, hipEventRecord(rb.mark, st));
  HIP_TRY(c, hipStreamWaitEvent(rb.stream, rb.mark, 0));
  if (int e = paired_build_enqueue(c, s, rb.tab, rb.stream, 0, 0)) return e;  // (the other slices: one per evaluation, paired_sync_tables)
  rb.next_slice = 1;
  if (s.build_plan.units <= 1) HIP_TRY(c, hipEventRecord(rb.done, rb.stream));  // (an empty read set: nothing follows)
  rb.after.clear();
  rb.start_eval = s.eval_count;
  rb.active = true;
  if (getenv("GAML_HIP_TRACE_HOST")) fprintf(stderr, "rebuild beside the evaluations started: retire %.0f us, window lists + launches %.0f us\n", t1 - t0, now_us() - t1);
  return 0;
}

// the next slice of a build beside the evaluations (all that are left when `rest`)
int paired_build_continue(gaml_hip_ctx* c, PairedSet& s, bool rest) {
  TableRebuild& rb = s.rebuild;
  if (!rb.active || rb.next_slice >= s.build_plan.units) return 0;
  const int to = rest ? kBuildAll : rb.next_slice + 1;  // two launches per evaluation
  if (int e = paired_build_enqueue(c, s, rb.tab, rb.stream, rb.next_slice, to)) return e;
  rb.next_slice = rest ? s.build_plan.units : to + 1;
  if (rb.next_slice >= s.build_plan.units) HIP_TRY(c, hipEventRecord(rb.done, rb.stream));
  return 0;
}

// the new tables take over; what was activated since the build started goes onto their (empty) delta lists
int paired_finish_async_rebuild(gaml_hip_ctx* c, PairedSet& s, hipStream_t st) {
  TableRebuild& rb = s.rebuild;
  const double t0 = now_us();
  if (int e = paired_build_continue(c, s, true)) return e;
  HIP_TRY(c, hipEventSynchronize(rb.done));  // (long done as a rule: the take-over is a fixed number of evaluations after the start)
  const double t1 = now_us();
  rb.active = false;
  if (int e = paired_build_collect(c, s, rb.tab)) return e;
  std::swap(s.tab, rb.tab);
  paired_adopt_tables(s);
  if (int e = paired_delta_reset(c, s, st)) return e;
  // this call's own activations join the windows noted since the build started
  for (int mt = 0; mt < 2; mt++) { for (int32_t w : s.mate[mt].activated_log) rb.after.emplace_back(mt, w); s.mate[mt].activated_log.clear(); }
  int64_t replayed = 0;
  for (const auto& mw : rb.after) replayed += s.mate[mw.first].wins[mw.second].count;
  if (int e = paired_delta_apply(c, s, s.tab, rb.after, st)) return e;
  if (getenv("GAML_HIP_TRACE_HOST")) fprintf(stderr, "take-over: waited %.0f us for the build, %zu windows / %lld records onto the new lists, %.0f us in all\n", t1 - t0, rb.after.size(), (long long)replayed, now_us() - t0);
  rb.after.clear();
  for (int mt = 0; mt < 2; mt++) s.dev[mt].uploaded_generation = s.mate[mt].active_generation;
  s.full_rebuilds++;
  s.async_rebuilds++;
  return 0;
}

// Everything the record tables need before a scoring launch; enqueued on `st`. One call per evaluation (or batch).
int paired_sync_tables(gaml_hip_ctx* c, PairedSet& s, hipStream_t st) {
  TableRebuild& rb = s.rebuild;
  const int64_t np = s.mate[0].n_local();
  s.eval_count++;
  if (int e = pool_mirror(c, s, st)) return e;
  // The new tables take over a FIXED number of evaluations after their build was started -- not whenever the build
  // happens to be done: a rebuild changes the order of the final sum (last bits), and equal inputs must give equal outputs
  // run to run (the annealing loop compares likelihoods with strict >).
  const int64_t swap_after = KNOB(c, 14) > 1 ? KNOB(c, 14) : kTakeOverAfter;
  if (rb.active && s.eval_count - rb.start_eval >= swap_after) { if (int e = paired_finish_async_rebuild(c, s, st)) return e; }
  else if (rb.active) { if (int e = paired_build_continue(c, s, false)) return e; }
  bool activated_now = !s.mate[0].activated_log.empty() || !s.mate[1].activated_log.empty();
  s.quiet_calls = activated_now ? 0 : s.quiet_calls + 1;
  if (!s.tab.built) return paired_rebuild_tables(c, s, st);
  // Windows activated since the tables were built put their pairs on the delta lists. The tables are rebuilt when the
  // lists pass pairs / 8, when the cache has been quiet for 64 evaluations with pairs still on the lists, or on request
  // (gaml_hip_compact_tables: at the next evaluation, on the calling stream).
  const int64_t limit = KNOB(c, 6) == 1 ? 0 : (KNOB(c, 18) > 0 ? std::max<int64_t>(256, np / KNOB(c, 18)) : std::max<int64_t>(4096, np / 8));  // (knob 18: tests and soaks want rebuilds at small sizes)
  int64_t new_records = 0;
  if (activated_now) for (int mt = 0; mt < 2; mt++) for (int32_t w : s.mate[mt].activated_log) new_records += s.mate[mt].wins[w].count;
  const bool over = activated_now && s.nd_est + new_records > limit;
  const bool quiet = !activated_now && s.nd_est > 0 && s.quiet_calls >= 64 && KNOB(c, 6) != 2;
  const bool refold = (KNOB(c, 16) == 1) != s.tab.keep_dominated;  // A/B of the table contents: a request rebuilds even without delta pairs
  const bool asked = s.compact_requested && (s.nd_est > 0 || activated_now || refold);
  s.compact_requested = false;
  const bool beside = KNOB(c, 14) != 1 && KNOB(c, 6) != 1;
  const int64_t hard = s.delta_cap ? (int64_t)s.delta_cap - 2048 : std::max<int64_t>(4096, np / 2);
  const bool too_many = s.nd_est + new_records > hard || new_records > kDeltaMaxPerCall;
  if (asked || ((over || quiet) && !beside) || too_many) {
    if (int e = paired_rebuild_tables(c, s, st)) return e;
    activated_now = false;
  } else if ((over || quiet || rb.retired) && !rb.active) {
    if (int e = paired_start_async_rebuild(c, s, st)) return e;  // (this call's activations are in its window list)
    s.quiet_calls = 0;
    if (activated_now && rb.active) {
      std::vector<std::pair<int32_t, int32_t>> wins;
      for (int mt = 0; mt < 2; mt++) { for (int32_t w : s.mate[mt].activated_log) wins.emplace_back(mt, w); s.mate[mt].activated_log.clear(); }
      if (int e = paired_delta_apply(c, s, s.tab, wins, st)) return e;
      activated_now = false;
    }
  }
  if (activated_now) {
    std::vector<std::pair<int32_t, int32_t>> wins;
    for (int mt = 0; mt < 2; mt++) { for (int32_t w : s.mate[mt].activated_log) wins.emplace_back(mt, w); s.mate[mt].activated_log.clear(); }
    if (rb.active) rb.after.insert(rb.after.end(), wins.begin(), wins.end());
    if (int e = paired_delta_apply(c, s, s.tab, wins, st)) return e;
  }
  for (int mt = 0; mt < 2; mt++) s.dev[mt].uploaded_generation = s.mate[mt].active_generation;
  return 0;
}

}  // namespace
