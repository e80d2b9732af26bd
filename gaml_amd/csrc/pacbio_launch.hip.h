// pacbio_launch.hip.h -- PacBio read sets: coverage sweep on the device, scorer launch (CalcScoreForPacbio graph.cc:3171-3261)
// (one translation unit with gaml_hip.hip, which includes this file at the place its contents used to stand)
#pragma once

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
  // scratch of the sorts and of the running maximum (radix_sort.hip.h): [keys, 2 n][payload, n][digit counts][tile maxima]
  const size_t off_vals = 2 * n1 * sizeof(u64), off_hist = off_vals + n1 * sizeof(u64), off_rm = off_hist + align16(rs_hist_bytes(2 * n1));
  const size_t tmp_bytes = off_rm + align16(rm_scratch_bytes(n1));
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
    u64* const tmp_keys = (u64*)d.tmp.p;
    u64* const tmp_vals = (u64*)((char*)d.tmp.p + off_vals);
    unsigned* const hist = (unsigned*)((char*)d.tmp.p + off_hist);
    // intervals by (contig, begin), their ends travelling with them; positions by (contig, position); running maximum of the ends
    HIP_TRY(c, rs_sort<u64>(d.key_begin.as<u64>(), d.key_begin_s.as<u64>(), tmp_keys, d.key_end.as<u64>(), d.key_end_s.as<u64>(), tmp_vals, (size_t)n, 0, end_bit, hist, st));
    HIP_TRY(c, rs_sort<u64>(d.pos.as<u64>(), d.pos_s.as<u64>(), tmp_keys, (const u64*)nullptr, (u64*)nullptr, (u64*)nullptr, (size_t)(2 * n), 0, end_bit, hist, st));
    HIP_TRY(c, rm_inclusive_max(d.key_end_s.as<u64>(), d.end_max.as<u64>(), (size_t)n, (u64*)((char*)d.tmp.p + off_rm), st));
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

