// paired_batch.hip.h -- gaml_hip_calc_prob_batch over paired sets: per-set tables from patches / whole, one pass over the records
// (one translation unit with gaml_hip.hip, which includes this file at the place its contents used to stand)
#pragma once

// ---------------------------------------------------------------------------------------------------------
// gaml_hip_calc_prob_batch, fast path: up to kMaxSets path sets in ONE pass over the records of every paired set
// (paired_score_multi_kernel). The host plans the sets one after the other straight into consecutive regions of one
// arena slot; then one launch per read set, one wait. Contexts with other kinds of read sets, a coverage penalty or
// without a memo take the sequential path below (same results).
// ---------------------------------------------------------------------------------------------------------
static bool batch_fast_capable(const gaml_hip_ctx* c) {
  if (c->handles.empty() || KNOB(c, 11) == 1) return false;  // knob 11 = 1: force the sequential path (A/B, tools/)
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
  if (!c->direct_write || KNOB(c, 8) != 0 || KNOB(c, 13) != 0 || KNOB(c, 11) == 2) return 1;  // knob 11 = 2: full tables per set (A/B)
  constexpr size_t kPatchCap = 8192;  // entries per read set and batch
  struct PerSet {
    int slot = 0; char* wp = nullptr; size_t stride = 0;
    PairedLayout L; std::vector<PairedLayout> Ls; std::vector<PairedPrep> prep;
    std::vector<int> patch_off; size_t n_patches = 0;
    size_t tail_fixed = 0, chg_bytes[2] = {0, 0};
    size_t bw[2] = {0, 0};  // table entries per mate the batch's regions carry: the windows there are + room for those the batch itself adds (the resident copy's capacity is far larger)
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
    for (int mt = 0; mt < 2; mt++) r.bw[mt] = std::min<size_t>(P.cap_w[mt], ps.image[mt].occ12.size() + 8192);
    r.chg_bytes[0] = align16(r.bw[0]); r.chg_bytes[1] = align16(r.bw[1]);
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
        ta.off_occ[mt] = P.off_occ[mt]; ta.bytes_occ[mt] = r.bw[mt] * sizeof(Occ12);
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
                                      KNOB(c, 11) == 3 ? nullptr : chg)) return e;  // knob 11 = 3: every set resolves every pair (A/B)
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
      for (int mt = 0; mt < 2 && ok; mt++) ok = im[mt].occ12.size() <= r.bw[mt] && r.n_patches + im[mt].changed.size() <= kPatchCap;
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
  for (auto& ps : c->paireds) paired_refresh_counts(*ps);
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
  for (auto& ps : c->paireds) paired_refresh_counts(*ps);
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

