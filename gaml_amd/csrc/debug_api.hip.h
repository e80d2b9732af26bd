// debug_api.hip.h -- the gaml_hip_debug_* entry points (include/gaml_hip_debug.h): development builds only (-DGAML_HIP_DEV)
// (one translation unit with gaml_hip.hip, which includes this file at the place its contents used to stand)
#pragma once

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


// Host-only check of the record tables' rule "a record that is always overwritten stays out" (host_model.cc
// dominated_records) on the windows that are active now: builds the tables with and without the rule (no device) and
// verifies, record by record, that every pair's records with the rule are the records without it minus records of a
// junction window J for which the first node's own window -- active -- holds a record of the same read at the same
// position. out6 = {records left out mate 1, mate 2, pairs of the compact class with / without the rule, records
// checked, violations}. Returns GAML_HIP_ESTATE when a violation was found.
// a pair's records as the host restatement of the tables holds them (build_pair_tables)
static void paired_base_records(const PairTables& pt, int32_t slot, int mt, std::vector<RecQuad>& out) {
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

int gaml_hip_debug_fold_check(gaml_hip_ctx* c, int rs, int64_t* out6) {
  MULTI_SHARD0(c);
  if (!c || rs < 0 || rs >= (int)c->handles.size() || c->handles[rs].kind != 1 || !out6) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  PairedSet& s = *c->paireds[c->handles[rs].idx];
  for (int mt = 0; mt < 2; mt++) for (const Window& w : s.mate[mt].wins) if (w.first < 0 && w.count > 0) return fail(c, GAML_HIP_ESTATE, "host-only check: some windows' records exist in the device pool only (use gaml_hip_debug_tables_check)");
  PairTables with, without;
  build_pair_tables(s.mate[0], s.mate[1], with, true);
  build_pair_tables(s.mate[0], s.mate[1], without, false);
  int64_t checked = 0, bad = 0;
  const int64_t n = s.mate[0].n_local();
  std::vector<RecQuad> a, b;
  for (int mt = 0; mt < 2; mt++) {
    const ShortMate& m = s.mate[mt];
    // per (window, read, position): is it a record of an active single-node window?
    for (int64_t read = 0; read < n; read++) {
      a.clear(); b.clear();
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

// The library's own radix sort and running maximum (radix_sort.hip.h) on caller data, for the tests: keys[n] (+ payload
// vals[n], null: keys only) sorted stably on bits [begin_bit, end_bit) in place; run_max (null: skip) receives the
// inclusive running maximum of the SORTED payload (or of the sorted keys when there is no payload).
int gaml_hip_debug_radix_sort(gaml_hip_ctx* c, uint64_t* keys, uint64_t* vals, int64_t n, int begin_bit, int end_bit, uint64_t* run_max) {
  MULTI_SHARD0(c);
  if (!c || c->device < 0 || !keys || n < 0 || begin_bit < 0 || end_bit > 64) return fail(c, c ? GAML_HIP_EINVAL : GAML_HIP_EINVAL, "bad arguments");
  if (n == 0) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  typedef rs_u64 u64;
  DevBuf k_in, k_out, k_tmp, v_in, v_out, v_tmp, hist, rm, mx;
  const size_t bytes = (size_t)n * sizeof(u64);
  for (DevBuf* b : {&k_in, &k_out, &k_tmp, &mx}) HIP_TRY(c, b->reserve(bytes));
  if (vals) for (DevBuf* b : {&v_in, &v_out, &v_tmp}) HIP_TRY(c, b->reserve(bytes));
  HIP_TRY(c, hist.reserve(rs_hist_bytes((size_t)n)));
  HIP_TRY(c, rm.reserve(rm_scratch_bytes((size_t)n)));
  hipStream_t st = c->stream;
  HIP_TRY(c, hipMemcpyAsync(k_in.p, keys, bytes, hipMemcpyHostToDevice, st));
  if (vals) HIP_TRY(c, hipMemcpyAsync(v_in.p, vals, bytes, hipMemcpyHostToDevice, st));
  HIP_TRY(c, rs_sort<u64>(k_in.as<u64>(), k_out.as<u64>(), k_tmp.as<u64>(), vals ? v_in.as<u64>() : (const u64*)nullptr, vals ? v_out.as<u64>() : (u64*)nullptr,
                          vals ? v_tmp.as<u64>() : (u64*)nullptr, (size_t)n, begin_bit, end_bit, hist.as<unsigned>(), st));
  if (run_max) HIP_TRY(c, rm_inclusive_max(vals ? v_out.as<u64>() : k_out.as<u64>(), mx.as<u64>(), (size_t)n, rm.as<u64>(), st));
  HIP_TRY(c, hipMemcpyAsync(keys, k_out.p, bytes, hipMemcpyDeviceToHost, st));
  if (vals) HIP_TRY(c, hipMemcpyAsync(vals, v_out.p, bytes, hipMemcpyDeviceToHost, st));
  if (run_max) HIP_TRY(c, hipMemcpyAsync(run_max, mx.p, bytes, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipStreamSynchronize(st));
  for (DevBuf* b : {&k_in, &k_out, &k_tmp, &v_in, &v_out, &v_tmp, &hist, &rm, &mx}) b->release();
  return 0;
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
  for (int mt = 0; mt < 2; mt++) for (const Window& w : s.mate[mt].wins) if (w.first < 0 && w.count > 0) return fail(c, GAML_HIP_ESTATE, "host-only check: some windows' records exist in the device pool only (use gaml_hip_debug_tables_check)");
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
// single call), in block order [lane-per-pair classes | wave-per-pair blocks]: which
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
    layout8[6] = 0; layout8[7] = n;
  }
  return n;
}



// The device table build (table_build.hip.h) against the host restatement (host_model.cc build_pair_tables) on the windows
// that are active now: a fresh build into a scratch set of buffers, every array fetched and compared entry by entry. The
// host side works on a copy of the device pool (the records of windows the aligner's kernels filed exist nowhere else).
// out8 = {pairs, compact class, static part, <= 2 records, <= 4, more, entries compared, mismatches}.
int gaml_hip_debug_tables_check(gaml_hip_ctx* c, int rs, int64_t* out8) {
  MULTI_SHARD0(c);
  if (!c || rs < 0 || rs >= (int)c->handles.size() || c->handles[rs].kind != 1 || !out8) return fail(c, GAML_HIP_EINVAL, "bad arguments");
  if (c->device < 0) return fail(c, GAML_HIP_ENODEVICE, "host-only context");
  PairedSet& s = *c->paireds[c->handles[rs].idx];
  HIP_TRY(c, hipSetDevice(c->device));
  if (s.rebuild.active) { if (int e = paired_build_continue(c, s, true)) return e; HIP_TRY(c, hipEventSynchronize(s.rebuild.done)); s.rebuild.active = false; s.rebuild.after.clear(); }
  if (int e = prepare_paired_tables(c, s)) return e;
  if (int e = pool_mirror(c, s, c->stream)) return e;
  if (int e = paired_upload_statics(c, s, c->stream)) return e;
  TableDev T;
  if (int e = paired_build_enqueue(c, s, T, c->stream)) return e;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (int e = paired_build_collect(c, s, T)) { T.release(); return e; }
  // host side
  ShortMate hm[2];
  for (int mt = 0; mt < 2; mt++) {
    const ShortMate& m = s.mate[mt];
    hm[mt].n_global = m.n_global; hm[mt].lo = m.lo; hm[mt].hi = m.hi; hm[mt].lens = m.lens;
    hm[mt].wins = m.wins;
    std::vector<int4> dp((size_t)s.dev[mt].pool_n);
    if (!dp.empty()) HIP_TRY(c, hipMemcpy(dp.data(), s.dev[mt].pool.p, dp.size() * sizeof(int4), hipMemcpyDeviceToHost));
    hm[mt].pool.resize(dp.size());
    for (size_t k = 0; k < dp.size(); k++) hm[mt].pool[k] = gaml_aligment{dp[k].y, dp[k].z & 0xff, dp[k].w, (dp[k].z >> 8) & 1};
    for (Window& w : hm[mt].wins) w.first = w.dfirst < 0 ? 0 : w.dfirst;
    for (Window& w : hm[mt].wins) if (w.dfirst < 0) { w.count = 0; }
  }
  PairTables pt;
  build_pair_tables(hm[0], hm[1], pt, KNOB(c, 16) != 1, paired_static_ins_n(c, s));
  const int64_t n = s.mate[0].n_local();
  int64_t compared = 0, bad = 0;
  auto fetch = [&](const DevBuf& d, size_t bytes, std::vector<char>& out) -> int {
    out.resize(bytes);
    if (bytes) HIP_TRY(c, hipMemcpy(out.data(), d.p, bytes, hipMemcpyDeviceToHost));
    return 0;
  };
  auto cmp = [&](const DevBuf& d, const void* host, size_t bytes, size_t elem, const char* what) -> int {
    std::vector<char> v;
    if (int e = fetch(d, bytes, v)) return e;
    int64_t b0 = bad;
    for (size_t k = 0; k < bytes / elem; k++) { compared++; if (memcmp(v.data() + k * elem, (const char*)host + k * elem, elem) != 0) bad++; }
    if (bad != b0 && getenv("GAML_HIP_TRACE_HOST")) fprintf(stderr, "tables_check: %s differs in %lld of %zu entries\n", what, (long long)(bad - b0), bytes / elem);
    return 0;
  };
  for (int k = 0; k < 4; k++) { compared++; bad += T.class_count[k] != pt.class_count[k]; }
  compared++; bad += T.n0a != pt.n0a;
  if (bad == 0) {
    const int64_t n0 = pt.class_count[0], n16 = n - n0;
    if (int e = cmp(T.slot_of_read, pt.slot_of_read.data(), (size_t)n * 4, 4, "slot_of_read")) return e;
    if (int e = cmp(T.read_of_slot, pt.read_of_slot.data(), (size_t)n * 4, 4, "read_of_slot")) return e;
    for (int mt = 0; mt < 2; mt++) {
      if (int e = cmp(T.rec8[mt], pt.rec8[mt].data(), (size_t)n0 * 8, 8, "rec8")) return e;
      if (int e = cmp(T.first[mt], pt.rm[mt].first.data(), (size_t)n16 * 16, 16, "first")) return e;
      if (int e = cmp(T.extra[mt], pt.rm[mt].extra.data(), pt.rm[mt].extra.size() * 16, 16, "extra")) return e;
      if (int e = cmp(T.inl[mt], pt.inl[mt].data(), pt.inl[mt].size() * 16, 16, "inl")) return e;
      compared++; bad += T.extras[mt] != (int64_t)pt.rm[mt].extra.size();
      compared++; bad += T.dropped[mt] != pt.dropped_records[mt];
    }
    if (int e = cmp(T.len_code, pt.len_code.data(), (size_t)n0, 1, "len_code")) return e;
    if (int e = cmp(T.static_idx, pt.static_idx.data(), (size_t)pt.n0a * 4, 4, "static_idx")) return e;
    if (int e = cmp(T.len12, pt.len12.data(), (size_t)n16 * 4, 4, "len12")) return e;
    compared++; bad += pt.len_combo != s.pt.len_combo;
  }
  out8[0] = n; out8[1] = T.class_count[0]; out8[2] = T.n0a; out8[3] = T.class_count[1]; out8[4] = T.class_count[2]; out8[5] = T.class_count[3];
  out8[6] = compared; out8[7] = bad;
  T.release();
  return bad ? fail(c, GAML_HIP_ESTATE, "record tables: the device build differs from the host restatement") : GAML_HIP_OK;
}
