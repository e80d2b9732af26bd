// single_launch.hip.h -- single-end read sets: host preparation and launch (CalcScoreForPaths graph.cc:1650-1743)
// (one translation unit with gaml_hip.hip, which includes this file at the place its contents used to stand)
#pragma once

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

