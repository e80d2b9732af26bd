// aligner_launch.hip.h -- GPU window alignment of pending windows: index upload, small-batch pipeline, general route (AlignSubpathInternal graph.cc:839-899)
// (one translation unit with gaml_hip.hip, which includes this file at the place its contents used to stand)
#pragma once

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
  {
    // key -> bucket as an open-addressing table (span_cands_kernel: a lookup is one or two loads): load <= 1/2, linear
    // probing; hbits = 0 (no table: the small batches of both mates take the per-mate route) if a key does not fit 32
    // bits or a probe sequence would exceed what the kernel walks
    d.hbits = 0;
    int bits = 4;
    while (((size_t)1 << bits) < 2 * m.bucket_hash.size()) bits++;
    bool fits = bits <= 28;
    for (uint64_t k : m.bucket_hash) fits = fits && k <= 0xffffffffull;
    if (fits) {
      std::vector<AlnHashSlot> tab((size_t)1 << bits, AlnHashSlot{0, 0, 0, 0});
      const uint32_t mask = (1u << bits) - 1u;
      int longest = 0;
      for (size_t b = 0; b < m.bucket_hash.size(); b++) {
        uint32_t h = aln_hash_home((uint32_t)m.bucket_hash[b], bits);
        int probes = 1;
        while (tab[h].used) { h = (h + 1) & mask; probes++; }
        tab[h] = AlnHashSlot{(uint32_t)m.bucket_hash[b], m.bucket_off[b], m.bucket_off[b + 1] - m.bucket_off[b], 1};
        longest = std::max(longest, probes);
      }
      if (longest <= 60) {
        HIP_TRY(c, up(d.htab, tab.data(), tab.size() * sizeof(AlnHashSlot)));
        d.hbits = bits;
      }
    }
  }
  d.uploaded = true;
  return 0;
}

// the small-batch pipeline's fixed-capacity buffers, allocated together with the index upload (the cold first
// evaluation), not in the annealing call that first brings a small batch
int aln_small_reserve(gaml_hip_ctx* c, AlignSmall& S) {
  if (!S.counters.p) { HIP_TRY(c, S.counters.reserve(256)); HIP_TRY(c, hipMemset(S.counters.p, 0, 256)); }  // (publish_hits_kernel leaves them at zero)
  HIP_TRY(c, S.spans.reserve((size_t)kFastSpans * sizeof(AlnSpan)));
  HIP_TRY(c, S.cands.reserve((size_t)kFastCands * sizeof(AlnCandX)));  // (AlnCand, or AlnCandX when both mates share a pipeline)
  HIP_TRY(c, S.hits.reserve((size_t)kFastCands * sizeof(AlnHit)));
  HIP_TRY(c, S.wcopy.reserve(((size_t)1 << 18) + 64));  // a batch's window strings in ordinary device memory (span_cands_kernel writes, extend_pair2_kernel reads)
  if (!S.out_host.p) { HIP_TRY(c, S.out_host.reserve(64 + 64 + (size_t)kFastCands * sizeof(AlnHit))); memset(S.out_host.p, 0, 128); }
  if (!S.in_dev) {
    const bool direct = c->direct_write && KNOB(c, 8) == 0;
    const size_t want = (size_t)1 << 18;
    if (direct) HIP_TRY(c, hipExtMallocWithFlags(&S.in_dev, want, hipDeviceMallocFinegrained));
    else { HIP_TRY(c, hipMalloc(&S.in_dev, want)); HIP_TRY(c, S.in_host.reserve(want)); }
    S.in_cap = want; S.in_direct = direct;
  }
  return 0;
}

bool aln_gpu_capable(const gaml_hip_ctx* c, const ShortMate& m) {
  return !(c->device < 0 || KNOB(c, 5) == 1 || m.index_read_len < 16 || m.max_len > kAlnMaxRead || m.n_local() == 0 || m.bucket_hash.empty());
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
  const bool direct = c->direct_write && KNOB(c, 8) == 0;
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
  hipLaunchKernelGGL(span_maxima_kernel, dim3((unsigned)job.blk[(size_t)nw]), dim3(kAlnBlock), 0, st, d_wstr, d_wins, nw, m.index_read_len, d_blk,
                     S.spans.as<AlnSpan>(), S.counters.as<unsigned>(), kFastSpans);
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

// the one wait of a small batch whose hits are filed on the device: the sequence word behind the windows' headers
int aln_small_wait(gaml_hip_ctx* c, AlignSmall& S, AlnJob& job) {
  volatile unsigned long long* word = (volatile unsigned long long*)S.out_host.p;
  const double t0 = now_us();
  bool seen = false;
  while (!seen && now_us() - t0 < 5000.0) { for (int k = 0; k < 256 && !seen; k++) { seen = *word == job.seq; __builtin_ia32_pause(); } }
  if (!seen) { HIP_TRY(c, hipStreamSynchronize((hipStream_t)job.stream)); if (*word != job.seq) return fail(c, GAML_HIP_ESTATE, "aligner: the filing kernel finished without its sequence word"); }
  std::atomic_thread_fence(std::memory_order_acquire);
  S.seen_us = now_us();
  job.enqueued = false;
  return 0;
}

// the one wait of a small batch: poll the sequence word (bounded), then the runtime's wait. 1: capacities exceeded
int aln_small_collect(gaml_hip_ctx* c, AlignSmall& S, AlnJob& job, std::vector<AlnHit>& hits, unsigned* n_cands_out) {
  volatile unsigned long long* word = (volatile unsigned long long*)S.out_host.p;
  const double t0 = now_us();
  bool seen = false;
#ifdef GAML_HIP_DEV
  static const int wait_mode = getenv("GAML_ALN_WAIT") ? atoi(getenv("GAML_ALN_WAIT")) : 0;  // A/B: 1 = the runtime's wait first
  if (wait_mode == 1) { HIP_TRY(c, hipStreamSynchronize((hipStream_t)job.stream)); c->aln_stage_us[2] += now_us() - t0; }
#endif
#ifdef GAML_HIP_DEV
  static const int keepalive = getenv("GAML_ALN_KEEPALIVE") ? atoi(getenv("GAML_ALN_KEEPALIVE")) : 0;  // A/B: PCIe traffic from the host while it waits: 1 reads through the BAR, 2 writes
  if (keepalive && S.in_direct && S.in_dev) {
    volatile unsigned* bar = (volatile unsigned*)((char*)S.in_dev + S.in_cap - 64);  // (the input block's last line: no batch reaches it)
    unsigned sink = 0, tick = 0;
    while (!seen && now_us() - t0 < 5000.0) {
      for (int k = 0; k < 64 && !seen; k++) { if (keepalive == 1) sink += *bar; else { *bar = ++tick; _mm_sfence(); } seen = *word == job.seq; }
    }
    if (sink == 0x12345678u) fprintf(stderr, "\n");
  }
  if (wait_mode == 5) {  // A/B: spin on the (cached) word without PAUSE -- a virtual CPU that executes PAUSE in a loop may be descheduled by its hypervisor
    while (!seen && now_us() - t0 < 5000.0) { for (int k = 0; k < 4096 && !seen; k++) seen = *word == job.seq; }
  }
  if ((wait_mode == 3 || wait_mode == 4) && S.in_direct && S.in_dev) {  // A/B: a read through the BAR per poll keeps the PCIe link out of its idle states
    volatile unsigned* bar = (volatile unsigned*)S.in_dev;
    unsigned sink = 0;
    while (!seen && now_us() - t0 < 5000.0) { for (int k = 0; k < 64 && !seen; k++) { sink += *bar; seen = *word == job.seq; } }
    if (sink == 0x12345678u) fprintf(stderr, "\n");
  }
#endif
  while (!seen && now_us() - t0 < 5000.0) { for (int k = 0; k < 256 && !seen; k++) { seen = *word == job.seq; __builtin_ia32_pause(); } }
  if (!seen) { HIP_TRY(c, hipStreamSynchronize((hipStream_t)job.stream)); if (*word != job.seq) return fail(c, GAML_HIP_ESTATE, "aligner: the publish kernel finished without its sequence word"); }
  std::atomic_thread_fence(std::memory_order_acquire);
  S.seen_us = now_us();
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
    // (the four-part order as two integers per hit: a comparison is two compares of registers instead of four of loaded fields)
    auto key1 = [](const AlnHit& h) { return ((uint64_t)((uint32_t)h.pos + 0x80000000u) << 32) | (uint32_t)((uint32_t)h.read + 0x80000000u); };
    auto key2 = [](const AlnHit& h) { return ((uint64_t)((uint32_t)h.strand + 0x80000000u) << 32) | (uint32_t)((uint32_t)h.order + 0x80000000u); };
    auto sort_range = [&](int k0, int k1) {
      struct Key { uint64_t a, b; uint32_t at; };
      std::vector<Key> keys;
      std::vector<AlnHit> tmp;
      for (int k = k0; k < k1; k++) {
        const int64_t lo = wstart[k], n = wstart[k + 1] - lo;
        if (n < 2) continue;
        keys.resize((size_t)n);
        for (int64_t i = 0; i < n; i++) keys[(size_t)i] = Key{key1(ok[lo + i]), key2(ok[lo + i]), (uint32_t)i};
        std::sort(keys.begin(), keys.end(), [](const Key& x, const Key& y) { return x.a != y.a ? x.a < y.a : x.b < y.b; });
        tmp.assign(ok.begin() + lo, ok.begin() + lo + n);
        for (int64_t i = 0; i < n; i++) ok[lo + i] = tmp[keys[(size_t)i].at];
      }
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
  if (m.pool.capacity() < m.pool.size() + ok.size()) m.pool.reserve(std::max(2 * (m.pool.size() + ok.size()), m.pool.capacity() + m.pool.capacity() / 2));  // one growth step for the whole batch
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
    for (int k = 0; k < nw; k++) m.filed.push_back(m.pending[k]);
  } else {
    thread_local std::vector<gaml_aligment> recs;
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
  if (KNOB(c, 5) == 3 || KNOB(c, 5) == 4) return 1;  // knob 5 = 3: general route, 4: one small pipeline per mate (A/B, tests)
  for (int mt = 0; mt < 2; mt++) if (mm[mt]->pending.empty() || !aln_gpu_capable(c, *mm[mt])) return 1;
  for (int mt = 0; mt < 2; mt++) if (ps.dev[mt].aln.uploaded && ps.dev[mt].aln.hbits == 0) return 1;  // no key table (aln_upload_index)
  const double t0 = now_us();
  HIP_TRY(c, hipSetDevice(c->device));
  AlignSmall& S = c->aln_small[0];
  for (int mt = 0; mt < 2; mt++) {
    if (!ps.dev[mt].aln.uploaded) { if (int e = aln_small_reserve(c, c->aln_small[mt])) return e; }
    if (int e = aln_upload_index(c, *mm[mt], ps.dev[mt].aln)) return e;
    if (ps.dev[mt].aln.hbits == 0) return 1;
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
  const bool direct = c->direct_write && KNOB(c, 8) == 0;
  if (in_bytes > S.in_cap || S.in_direct != direct) return 1;  // (the per-mate route sizes its own input block)
  // a handful of windows and their strings travel with the launches themselves (AlnWinArgs, AlnStrArgs); otherwise
  // through the input block: [windows][code-buffer offsets][window strings]
  const bool in_args = nw <= kAlnArgWins && job.wstr.size() <= (size_t)kAlnArgStr && KNOB(c, 5) != 5;  // knob 5 = 5: always the input block (tests)
  const size_t off_blk = align16(nw * sizeof(AlnWindow)), off_str = off_blk + align16((nw + 1) * sizeof(int32_t));
  if (!in_args) {
    char* wp = (char*)S.in_dev;
    if (!direct) { HIP_TRY(c, S.in_host.reserve(in_bytes)); wp = (char*)S.in_host.p; }
    memcpy(wp, job.wins.data(), nw * sizeof(AlnWindow));
    memcpy(wp + off_blk, job.blk.data(), (nw + 1) * sizeof(int32_t));
    memcpy(wp + off_str, job.wstr.data(), job.wstr.size());
    if (direct) _mm_sfence();
    else HIP_TRY(c, hipMemcpyAsync(S.in_dev, S.in_host.p, in_bytes, hipMemcpyHostToDevice, st));
  }
  const char* dbase = (const char*)S.in_dev;
  const AlnWindow* d_wins = (const AlnWindow*)dbase;
  const int* d_blk = (const int*)(dbase + off_blk);
  const char* d_wstr = dbase + off_str;
  AlnMates ix;
  for (int mt = 0; mt < 2; mt++) {
    const AlignDev& d = ps.dev[mt].aln;
    ix.htab[mt] = d.htab.as<AlnHashSlot>(); ix.hbits[mt] = d.hbits;
    ix.bucket_hash[mt] = d.bucket_hash.as<uint64_t>(); ix.bucket_top[mt] = d.bucket_top.as<int32_t>(); ix.bucket_off[mt] = d.bucket_off.as<int32_t>(); ix.bucket_reads[mt] = d.bucket_reads.as<int32_t>();
    ix.n_buckets[mt] = (int)mm[mt]->bucket_hash.size();
    ix.reads[mt] = d.reads.as<char>(); ix.read_off[mt] = d.read_off.as<int64_t>();
  }
  ix.split = n0;
  // two dispatches: spans + candidates, then the extension, whose last block publishes (aligner_small.hip.h)
#ifdef GAML_ALN_STAMPS
  {
    unsigned long long z[32];
    for (int k = 0; k < 32; k++) z[k] = (k == 0 || k == 8) ? ~0ull : 0ull;
    HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_aln_stamp), z, sizeof(z)));
  }
#endif
  AlnWinArgs wa{};
  AlnStrArgs& sa = S.str_args;  // (only its first job.wstr.size() bytes are read)
  if (in_args) {
    memcpy(sa.s, job.wstr.data(), job.wstr.size());
    wa.n = nw;
    for (int k = 0; k < nw; k++) { wa.blk[k] = job.blk[(size_t)k]; wa.w[k] = job.wins[(size_t)k]; }
    wa.blk[nw] = job.blk[(size_t)nw];
  }
#ifdef GAML_ALN_STAMPS
  if (getenv("GAML_ALN_TWICE")) {  // the same batch once before, unstamped results thrown away: does a warm instruction cache change the stamps?
    hipLaunchKernelGGL(span_cands_kernel, dim3((unsigned)job.blk[(size_t)nw]), dim3(kAlnBlock), 0, st, sa, in_args ? 1 : 0, d_wstr, d_wins, nw, mm[0]->index_read_len, d_blk, n0,
                       mm[1]->index_read_len, ix, wa, S.cands.as<AlnCandX>(), S.counters.as<unsigned>() + 1, kFastCands, S.wcopy.as<char>());
    const unsigned long long sq = ++S.out_seq;
    char* oh0 = (char*)S.out_host.dev;
    hipLaunchKernelGGL(extend_pair2_kernel, dim3(512), dim3(128 * kAlnPairs), 0, st, sa, in_args ? 1 : 0, S.cands.as<AlnCandX>(), S.counters.as<unsigned>(), kFastCands, S.wcopy.as<char>(), ix,
                       (AlnHit*)(oh0 + 128), (unsigned*)(oh0 + 64), (volatile unsigned long long*)oh0, sq, S.hits.as<AlnHit>());
    unsigned long long z[32];
    for (int k = 0; k < 32; k++) z[k] = (k == 0 || k == 8) ? ~0ull : 0ull;
    HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_aln_stamp), z, sizeof(z)));
  }
#endif
#ifdef GAML_HIP_DEV
  // A/B (GAML_ALN_WAIT=2): events attached to the two dispatches -- the device's own begin / end stamps next to the host's wait
  static const int ev_mode = getenv("GAML_ALN_WAIT") ? atoi(getenv("GAML_ALN_WAIT")) : 0;
  hipEvent_t* const aev = S.probe_ev;  // (per context: events belong to the device and thread that made them)
  if ((ev_mode == 2 || ev_mode == 4 || ev_mode == 6) && !aev[0]) for (int k = 0; k < 4; k++) HIP_TRY(c, hipEventCreate(&aev[k]));
  const bool timed = ev_mode == 2 || ev_mode == 4 || ev_mode == 6;
#else
  constexpr bool timed = false;
  constexpr int ev_mode = 0;
  hipEvent_t aev[4] = {nullptr, nullptr, nullptr, nullptr};
#endif
  const double ta = now_us();
  hipExtLaunchKernelGGL(span_cands_kernel, dim3((unsigned)job.blk[(size_t)nw]), dim3(kAlnBlock), 0, st, timed ? aev[0] : nullptr, timed ? aev[1] : nullptr, 0, sa, in_args ? 1 : 0, d_wstr, d_wins, nw, mm[0]->index_read_len, d_blk, n0,
                        mm[1]->index_read_len, ix, wa, S.cands.as<AlnCandX>(), S.counters.as<unsigned>() + 1, kFastCands, S.wcopy.as<char>(),
                        timed ? (unsigned long long*)((char*)S.out_host.dev + 32) : nullptr, S.out_seq + 1);
  double tb = now_us();
  double h_start_seen = 0;
  if (timed && ev_mode == 6) {  // wait for the grid-started word BEFORE the second launch: host and device clocks side by side
    volatile unsigned long long* sw = (volatile unsigned long long*)((char*)S.out_host.p + 32);
    while (*sw != S.out_seq + 1 && now_us() - tb < 2000.0) { }
    h_start_seen = now_us();
    tb = h_start_seen;
  }
  job.seq = ++S.out_seq;
  char* oh = (char*)S.out_host.dev;
  // The hits are filed on the device (aligner_file.hip.h): ordered, de-duplicated and appended to the mates' record pools by
  // one more small dispatch; the host gets the windows' headers back. Not when the batch has more windows than the filing
  // kernel's argument block names (knob 5 = 6: never -- the host files, A/B and tests).
  const bool file_dev = nw <= kFileMaxWins && KNOB(c, 5) != 6 && !timed;
  if (file_dev) {
    for (int mt = 0; mt < 2; mt++) if (int e = pool_reserve(c, ps, mt, ps.dev[mt].pool_n + kFileMaxHits)) return e;
    if (int e = pool_mirror(c, ps, st)) return e;  // (windows the host filed earlier come first in the pools)
  }
  hipExtLaunchKernelGGL(extend_pair2_kernel, dim3(512), dim3(128 * kAlnPairs), 0, st, timed ? aev[2] : nullptr, timed ? aev[3] : nullptr, 0, sa, in_args ? 1 : 0, S.cands.as<AlnCandX>(), S.counters.as<unsigned>(), kFastCands, S.wcopy.as<char>(), ix,
                        (AlnHit*)(oh + 128), (unsigned*)(oh + 64), (volatile unsigned long long*)oh, job.seq, S.hits.as<AlnHit>(), file_dev ? 1 : 0);
  if (file_dev) {
    AlnFileArgs fa;
    fa.n_win = nw; fa.split = n0;
    for (int mt = 0; mt < 2; mt++) { fa.pool_base[mt] = (int)ps.dev[mt].pool_n; fa.pool_cap[mt] = (int)std::min<size_t>(ps.dev[mt].pool.cap / sizeof(int4), 0x7fffffff); fa.pool[mt] = ps.dev[mt].pool.as<int4>(); }
    for (int k = 0; k < nw; k++) fa.wid[k] = k < n0 ? mm[0]->pending[(size_t)k] : mm[1]->pending[(size_t)(k - n0)];
    hipLaunchKernelGGL(aln_file_small_kernel, dim3(1), dim3(256), 0, st, fa, S.hits.as<AlnHit>(), S.counters.as<unsigned>(), kFastCands, (AlnFileOut*)(oh + 128),
                       (volatile unsigned long long*)oh, job.seq);
  }
  HIP_TRY(c, hipGetLastError());
  job.enqueued = true;
  const double t1 = now_us();
  double t_started = t1, t_ev_ready = 0, t_word = 0;
  if (timed) {
    volatile unsigned long long* sw = (volatile unsigned long long*)((char*)S.out_host.p + 32);
    while (*sw != job.seq && now_us() - t1 < 2000.0) __builtin_ia32_pause();
    t_started = now_us();
    // which comes first, and when: the extension kernel's completion as the runtime sees it, or the sequence word?
    volatile unsigned long long* word = (volatile unsigned long long*)S.out_host.p;
    while ((!t_ev_ready || !t_word) && now_us() - t1 < 2000.0) {
      if (!t_word && *word == job.seq) t_word = now_us();
      if (!t_ev_ready && hipEventQuery(aev[3]) == hipSuccess) t_ev_ready = now_us();
    }
  }
  std::vector<AlnHit> hits;
  unsigned nc = 0;
  if (file_dev) {
    if (int e = aln_small_wait(c, S, job)) return e;
    const AlnFileOut* fo = (const AlnFileOut*)((const char*)S.out_host.p + 128);
    if (fo->status != 0) return 1;  // not filed (too many hits for one block, more candidates than the buffers hold): the per-mate route redoes the batch
    nc = fo->n_cands;
    int64_t at[2] = {ps.dev[0].pool_n, ps.dev[1].pool_n};
    for (int k = 0; k < nw; k++) {
      const int mt = k < n0 ? 0 : 1;
      Window& w = mm[mt]->wins[(size_t)(mt == 0 ? mm[0]->pending[(size_t)k] : mm[1]->pending[(size_t)(k - n0)])];
      w.first = -1;  // its records exist in the device pool only
      w.dfirst = at[mt];
      w.count = fo->hdr[k][0];
      w.max_pos = w.count ? fo->hdr[k][1] : INT_MIN;
      w.global_max_pos = w.max_pos;
      w.pending = false;
      at[mt] += w.count;
    }
    for (int mt = 0; mt < 2; mt++) {
      if (at[mt] - ps.dev[mt].pool_n != (int64_t)fo->added[mt]) return fail(c, GAML_HIP_ESTATE, "aligner: the filed records do not add up to the windows' counts");
      ps.dev[mt].pool_n = at[mt];
      mm[mt]->pending.clear();
    }
    const double t3 = now_us();
    c->aln_windows += nw;
    c->aln_candidates += nc;
    c->aln_us += t3 - t0;
    c->aln_stage_us[0] += t1 - t0; c->aln_stage_us[1] += t3 - t1;
    c->aln_batches++;
    return 0;
  }
  const int rc = aln_small_collect(c, S, job, hits, &nc);
  if (rc != 0) return rc;  // 1: the candidates did not fit: the per-mate route redoes the batch
  const double t2 = now_us();
  if (timed) {
    HIP_TRY(c, hipEventSynchronize(aev[3]));
    const double t2b = now_us();
    float k1 = 0, k2 = 0, all = 0, gap = 0;
    (void)hipEventElapsedTime(&k1, aev[0], aev[1]); (void)hipEventElapsedTime(&k2, aev[2], aev[3]); (void)hipEventElapsedTime(&all, aev[0], aev[3]); (void)hipEventElapsedTime(&gap, aev[1], aev[2]);
    if (ev_mode == 6) {
      const unsigned long long g1 = *(volatile unsigned long long*)((char*)S.out_host.p + 40);
      const unsigned* hc = (const unsigned*)((const char*)S.out_host.p + 64);
      const unsigned long long g2 = (unsigned long long)hc[2] | ((unsigned long long)hc[3] << 32);
      const unsigned long long g3 = *(const volatile unsigned long long*)(hc + 4);  // (after the publisher's release: builds with -DGAML_ALN_STAMP_AFTER_FENCE)
      fprintf(stderr, "aln clocks: grid started -> published: device %.1f us (after the release %.1f), host (start word seen -> sequence word seen) %.1f us\n", (double)(long long)(g2 - g1) * 0.01,
              g3 ? (double)(long long)(g3 - g1) * 0.01 : -1.0, S.seen_us - h_start_seen);
    }
    fprintf(stderr, "aln timed: word seen %.1f / event ready %.1f after enqueue | launch1 %.1f launch2 %.1f | enqueued -> grid started (seen by the host) %.1f | enqueued -> sequence word seen %.1f, hits copied %.1f, + event sync %.1f | device: span %.1f gap %.1f extend %.1f, first begin -> last end %.1f us (cands %u)\n",
            t_word - t1, t_ev_ready - t1, tb - ta, t1 - tb, t_started - t1, S.seen_us - t1, t2 - t1, t2b - t2, k1 * 1e3, gap * 1e3, k2 * 1e3, all * 1e3, nc);
  }
#ifdef GAML_ALN_STAMPS
  {
    unsigned long long z[32];
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpyFromSymbol(z, HIP_SYMBOL(g_aln_stamp), sizeof(z)));
    auto us = [&](int k, int ref) { return z[k] ? (double)(long long)(z[k] - z[ref]) * 0.01 : -1.0; };
    fprintf(stderr, "aln stamps (us after the first wave of each kernel; last wave through): windows %d cands %u | span: entry %.2f strings %.2f codes %.2f front %.2f hash %.2f scan+atomic %.2f expanded %.2f | "
            "gap %.2f | extend: entry %.2f cand %.2f reads ptr %.2f read bytes %.2f all bytes %.2f (shader clock %.0f MHz) bytes->lds %.2f seed %.2f fwd %.2f bwd %.2f block done %.2f published %.2f | host enqueue->seen %.1f\n",
            nw, nc, us(1, 0), us(22, 0), us(23, 0), us(2, 0), us(3, 0), us(4, 0), us(5, 0), us(8, 0), us(9, 8), us(10, 8), us(18, 8), us(19, 8), us(17, 8), z[21] ? (double)z[20] / ((double)z[21] * 0.01) : 0.0, us(11, 8), us(12, 8), us(13, 8), us(14, 8), us(15, 8), us(16, 8), t2 - t1);
  }
#endif
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
int gpu_align_pending(gaml_hip_ctx* c, ShortMate& m, AlignDev& d, AlignSmall* small = nullptr, AlnJob* job_in = nullptr, PairedSet* ps = nullptr, int mt = 0) {
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
  const hipStream_t st = c->stream;  // (the library's own stream, like every other launch of an evaluation: a second queue coming
                                     // to life in the middle of the cold call stalled the first table build by 20 ms)
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
  if (small && KNOB(c, 5) != 3) {  // knob 5 = 3: always the general route (tests compare the two)
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
  // pinned staging: [window strings | window headers | block index | filing: window ids, counters in | counters out]
  const size_t o_wins = align16(wstr.size()), o_blk = o_wins + align16((size_t)nw * sizeof(AlnWindow)), o_init = o_blk + align16((size_t)(nw + 1) * sizeof(int32_t));
  const size_t o_back = o_init + align16(((size_t)3 * nw + 16) * sizeof(int)), stage_total = o_back + align16(((size_t)2 * nw + 2) * sizeof(int));
  HIP_TRY(c, S.stage.reserve(stage_total));
  char* const hp = (char*)S.stage.p;
  memcpy(hp, wstr.data(), wstr.size()); memcpy(hp + o_wins, wins.data(), (size_t)nw * sizeof(AlnWindow)); memcpy(hp + o_blk, blk.data(), (size_t)(nw + 1) * sizeof(int32_t));
  if (!wstr.empty()) HIP_TRY(c, hipMemcpyAsync(S.wstr.p, hp, wstr.size(), hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemcpyAsync(S.wins.p, hp + o_wins, nw * sizeof(AlnWindow), hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemcpyAsync(S.blk.p, hp + o_blk, (nw + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
  t1 = now_us();
  size_t cap_spans = std::max<size_t>(1 << 16, wstr.size());        // a span per window base and strand at most ~2x
  size_t cap_cands = std::max<size_t>(1 << 18, 8 * wstr.size());
  unsigned counts[2] = {0, 0};
  for (int attempt = 0; attempt < 6; attempt++) {
    HIP_TRY(c, S.spans.reserve(cap_spans * sizeof(AlnSpan)));
    HIP_TRY(c, S.cands.reserve(cap_cands * sizeof(AlnCand)));
    HIP_TRY(c, hipMemsetAsync(S.counters.p, 0, 16, st));
    if (blk[(size_t)nw] > 0) {
      hipLaunchKernelGGL(span_maxima_kernel, dim3((unsigned)blk[(size_t)nw]), dim3(kAlnBlock), 0, st, S.wstr.as<char>(), S.wins.as<AlnWindow>(), nw,
                         m.index_read_len, S.blk.as<int>(), S.spans.as<AlnSpan>(), S.counters.as<unsigned>(), (unsigned)cap_spans);
      HIP_TRY(c, hipGetLastError());
    }
    hipLaunchKernelGGL(candidates_kernel, dim3(256), dim3(kAlnBlock), 0, st, S.spans.as<AlnSpan>(), S.counters.as<unsigned>(),
                       (unsigned)cap_spans, d.bucket_hash.as<uint64_t>(), d.bucket_top.as<int32_t>(), d.bucket_off.as<int32_t>(), d.bucket_reads.as<int32_t>(),
                       (int)m.bucket_hash.size(), S.cands.as<AlnCand>(), S.counters.as<unsigned>() + 1, (unsigned)cap_cands);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(counts, S.counters.p, sizeof(counts), hipMemcpyDeviceToHost, st)); HIP_TRY(c, hipStreamSynchronize(st));
    if (counts[0] <= cap_spans && counts[1] <= cap_cands) break;
    cap_spans = std::max<size_t>(cap_spans, (size_t)counts[0] + 16);
    cap_cands = std::max<size_t>(cap_cands, (size_t)counts[1] + 16);
    if (attempt == 5) { m.flush_pending_cpu(c->g); return 0; }
  }
  gpu_probe(st, c->warm_buf.p, "  spans + candidates done");
  t2 = now_us();
  t3 = t2;
  nc = counts[1];
  if (nc) {
    HIP_TRY(c, S.hits.reserve((size_t)nc * sizeof(AlnHit)));
    hipLaunchKernelGGL(extend_kernel, dim3((nc + kAlnWaves - 1) / kAlnWaves), dim3(64 * kAlnWaves), 0, st, S.cands.as<AlnCand>(), S.counters.as<unsigned>() + 1,
                       (unsigned)cap_cands, S.wstr.as<char>(), S.wins.as<AlnWindow>(), d.reads.as<char>(), d.read_off.as<int64_t>(),
                       S.hits.as<AlnHit>());
    HIP_TRY(c, hipGetLastError());
    if (KNOB(c, 9)) {
      HIP_TRY(c, hipStreamSynchronize(st)); t3 = now_us();
    }
    // Large batches: order the hits on the device (window, position, read, strand, order; failed extensions
    // last) and fetch only the successful ones; the host then only walks them. Keys: read < 2^31, order < 2^24.
    int32_t longest = 0;  // (the device sort packs a span's order -- an index into its window -- into 24 bits)
    for (const AlnWindow& w : wins) longest = std::max(longest, w.len);
    if (nc >= 100000 && KNOB(c, 5) != 2 && longest < (1 << 24)) {
      const size_t n = nc;
      HIP_TRY(c, S.sort_keys.reserve(4 * n * sizeof(unsigned long long)));   // minor | major | two alternates
      HIP_TRY(c, S.sort_idx.reserve(3 * n * sizeof(unsigned)));
      HIP_TRY(c, S.hits_sorted.reserve(n * sizeof(AlnHit)));
      HIP_TRY(c, S.sort_tmp.reserve(rs_hist_bytes(n)));
      unsigned long long* k_minor = S.sort_keys.as<unsigned long long>();
      unsigned long long* k_major = k_minor + n;
      unsigned long long* k_alt = k_major + n;
      unsigned long long* k_alt2 = k_alt + n;
      unsigned* idx = S.sort_idx.as<unsigned>();
      unsigned* idx_alt = idx + n;
      unsigned* idx_tmp = idx_alt + n;
      unsigned* n_ok = S.counters.as<unsigned>() + 2;
      HIP_TRY(c, hipMemsetAsync(n_ok, 0, sizeof(unsigned), st));
      const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
      hipLaunchKernelGGL(hit_keys_kernel, dim3(grid), dim3(256), 0, st, S.hits.as<AlnHit>(), (unsigned)n, k_minor, k_major, idx, n_ok);
      HIP_TRY(c, hipGetLastError());
      // two stable radix sorts (radix_sort.hip.h): by (read, strand, order), then by (window, position) in that order
      HIP_TRY(c, rs_sort<unsigned>(k_minor, k_alt, k_alt2, idx, idx_alt, idx_tmp, n, 0, 56, S.sort_tmp.as<unsigned>(), st));
      hipLaunchKernelGGL(gather_u64_kernel, dim3(grid), dim3(256), 0, st, k_major, idx_alt, (unsigned)n, k_alt);
      HIP_TRY(c, hipGetLastError());
      HIP_TRY(c, rs_sort<unsigned>(k_alt, k_alt2, k_minor, idx_alt, idx, idx_tmp, n, 0, 64, S.sort_tmp.as<unsigned>(), st));
      hipLaunchKernelGGL(gather_hits_kernel, dim3(grid), dim3(256), 0, st, S.hits.as<AlnHit>(), idx, (unsigned)n, S.hits_sorted.as<AlnHit>());
      HIP_TRY(c, hipGetLastError());
      unsigned ok_count = 0;
      HIP_TRY(c, hipMemcpyAsync(&ok_count, n_ok, sizeof(unsigned), hipMemcpyDeviceToHost, st)); HIP_TRY(c, hipStreamSynchronize(st));
      gpu_probe(st, c->warm_buf.p, "  extension + sorts done");
      if (KNOB(c, 9)) t3 = now_us();
      if (ps && KNOB(c, 5) != 6 && ok_count > 0 && n < ((size_t)1 << 30)) {
        // filed on the device: survivors flagged, placed by a prefix sum, written into this mate's pool; the windows' headers come back
        MateDev& md = ps->dev[mt];
        if (int e = pool_mirror(c, *ps, st)) return e;  // (windows the host filed earlier come first in the pool)
        if (int e = pool_reserve(c, *ps, mt, md.pool_n + (int64_t)ok_count)) return e;
        const unsigned tiles = (unsigned)((n + kTbScanTile - 1) / kTbScanTile);
        // scratch: flags | places | tile sums | window ids | per-window counts | maxima | a zero word + the total
        HIP_TRY(c, S.file_tmp.reserve((2 * n + tiles + 3 * (size_t)nw + 16) * sizeof(int)));
        int* flags = S.file_tmp.as<int>();
        int* place = flags + n;
        int* tile_sum = place + n;
        int* wid_of = tile_sum + tiles;
        int* win_cnt = wid_of + nw;
        int* win_max = win_cnt + nw;
        int* zero = win_max + nw;  // [0] zero (the scan's "slots before"), [1] the number of survivors
        int* const init = (int*)(hp + o_init);
        memset(init, 0, ((size_t)3 * nw + 16) * sizeof(int));
        for (int k = 0; k < nw; k++) { init[(size_t)k] = m.pending[(size_t)k]; init[(size_t)2 * nw + k] = INT_MIN; }
        HIP_TRY(c, hipMemcpyAsync(wid_of, init, ((size_t)3 * nw + 16) * sizeof(int), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(aln_file_flags_kernel, dim3(grid), dim3(256), 0, st, S.hits_sorted.as<AlnHit>(), n_ok, (unsigned)n, flags);
        // (tb_scan_*: exclusive prefix of n ints, "n - cnt[kTbClass0]" entries: `zero` stands in for the counters)
        hipLaunchKernelGGL(tb_scan_tiles_kernel, dim3(tiles), dim3(256), 0, st, flags, zero - kTbClass0, (int)n, tile_sum);
        hipLaunchKernelGGL(tb_scan_top_kernel, dim3(1), dim3(1024), 0, st, tile_sum, (int)tiles, zero + 1);
        hipLaunchKernelGGL(tb_scan_apply_kernel, dim3(tiles), dim3(256), 0, st, flags, zero - kTbClass0, (int)n, tile_sum, place);
        hipLaunchKernelGGL(aln_file_write_kernel, dim3(grid), dim3(256), 0, st, S.hits_sorted.as<AlnHit>(), flags, place, (unsigned)n, md.pool.as<int4>(), (int)md.pool_n, wid_of, win_cnt, win_max);
        HIP_TRY(c, hipGetLastError());
        const int* const back = (const int*)(hp + o_back);
        HIP_TRY(c, hipMemcpyAsync(hp + o_back, win_cnt, ((size_t)2 * nw + 2) * sizeof(int), hipMemcpyDeviceToHost, st)); HIP_TRY(c, hipStreamSynchronize(st));
        gpu_probe(st, c->warm_buf.p, "  filed");
        int64_t at = md.pool_n;
        for (int k = 0; k < nw; k++) {
          Window& w = m.wins[(size_t)m.pending[(size_t)k]];
          w.first = -1;  // its records exist in the device pool only
          w.dfirst = at;
          w.count = back[(size_t)k];
          w.max_pos = w.count ? back[(size_t)nw + k] : INT_MIN;
          w.global_max_pos = w.max_pos;
          w.pending = false;
          at += w.count;
        }
        if (at - md.pool_n != (int64_t)back[(size_t)2 * nw + 1]) return fail(c, GAML_HIP_ESTATE, "aligner: the filed records do not add up to the windows' counts");
        md.pool_n = at;
        m.pending.clear();
        c->aln_windows += nw;
        c->aln_candidates += nc;
        const double t5 = now_us();
        c->aln_us += t5 - t0;
        c->aln_stage_us[0] += t1 - t0; c->aln_stage_us[1] += t2 - t1; c->aln_stage_us[2] += t5 - t2;
        c->aln_batches++;
        gpu_probe(st, c->warm_buf.p, "  mate done");
        return 0;
      }
      // (the host's copy of the hits only on the routes that file on the host: 67 MB of pages to touch at cfg3's cold call)
      hits.assign(ok_count, AlnHit{0, 0, -1, 0, 0, 0});
      if (ok_count) HIP_TRY(c, hipMemcpyAsync(hits.data(), S.hits_sorted.p, (size_t)ok_count * sizeof(AlnHit), hipMemcpyDeviceToHost, st)); HIP_TRY(c, hipStreamSynchronize(st));
      device_sorted = true;
    } else {
      hits.assign(nc, AlnHit{0, 0, -1, 0, 0, 0});
      HIP_TRY(c, hipMemcpyAsync(hits.data(), S.hits.p, (size_t)nc * sizeof(AlnHit), hipMemcpyDeviceToHost, st)); HIP_TRY(c, hipStreamSynchronize(st));
    }
  }
  }  // general route
  const double t4 = now_us();
  if (!KNOB(c, 9)) t3 = t4;
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
  // (the jobs' buffers -- 5 MB of window strings at cfg3's cold batch -- live in the context: giving them back to the system at
  // the end of this call stalled the device's queues for 12-25 ms, see AlignScratch::stage)
  AlnJob* const job = c->aln_job;
  for (int q = 0; q < 2; q++) { job[q].prepared = false; job[q].enqueued = false; job[q].wstr.clear(); job[q].wins.clear(); job[q].blk.clear(); }
  if (c->device >= 0 && KNOB(c, 5) != 3) {
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
    if (int e = gpu_align_pending(c, ps.mate[mt], ps.dev[mt].aln, c->device >= 0 ? &c->aln_small[mt] : nullptr, &job[mt], c->device >= 0 ? &ps : nullptr, mt)) return e;
  return 0;
}

