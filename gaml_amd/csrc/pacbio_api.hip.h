// pacbio_api.hip.h -- C ABI of the PacBio cache-miss side: missing sub-walks, SAM ingestion + banded DP (graph.cc:2650-2795, 2175-2297)
// (one translation unit with gaml_hip.hip, which includes this file at the place its contents used to stand)
#pragma once

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
#ifdef GAML_HIP_DEV

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
#endif  // GAML_HIP_DEV


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

