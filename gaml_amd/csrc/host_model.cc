// host_model.cc -- see host_model.h. Reference citations are file:line under the reference tree.
#include "host_model.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>

namespace gaml {

// ---------------------------------------------------------------------------------------
// graph
// ---------------------------------------------------------------------------------------
void GraphStore::finish() {
  // nodes of length <= 3 with equal strings collapse onto the first one (graph.h:249-266);
  // only the PacBio scorer normalises walks with it (graph.cc:3184).
  norm.resize(n());
  std::unordered_map<std::string, int32_t> tiny;
  for (int32_t i = 0; i < n(); i++) {
    norm[i] = i;
    if (len(i) > 3) continue;
    std::string s(seq(i), len(i));
    auto it = tiny.find(s);
    if (it == tiny.end()) tiny.emplace(s, i);
    else norm[i] = it->second;
  }
}

bool GraphStore::load_lastgraph(const std::string& file, std::string* err) {
  // Velvet LastGraph as LoadGraph reads it (graph.cc:52-106): first field of the header = node
  // count; per node one ignored line, the forward string, the twin string; ARC lines follow.
  std::ifstream f(file.c_str());
  if (!f.is_open()) { if (err) *err = "cannot open graph file " + file; return false; }
  std::string line;
  if (!std::getline(f, line)) { if (err) *err = "empty graph file"; return false; }
  long nn = strtol(line.c_str(), nullptr, 10);
  if (nn < 0) { if (err) *err = "bad node count"; return false; }
  bases.clear();
  off.assign(1, 0);
  std::string fwd, twin;
  for (long i = 0; i < nn; i++) {
    std::getline(f, line);
    std::getline(f, fwd);
    std::getline(f, twin);
    bases += fwd; off.push_back((int64_t)bases.size());
    bases += twin; off.push_back((int64_t)bases.size());
  }
  finish();
  return true;
}

int32_t walk_length(const GraphStore& g, const Walk& w) {  // GetPathLen graph.cc:1766-1773
  int32_t t = 0;
  for (int32_t e : w) t += e < 0 ? -e : g.len(e);
  return t;
}

void split_contigs(const Walk& path, std::vector<std::pair<int32_t, int32_t>>& ranges, std::vector<int32_t>& gaps) {
  // graph.cc:1808-1824: a negative entry -k is a gap of k bases and ends a contig
  ranges.clear(); gaps.clear();
  int32_t last = 0;
  for (int32_t i = 0; i < (int32_t)path.size(); i++)
    if (path[i] < 0) { gaps.push_back(-path[i]); ranges.emplace_back(last, i); last = i + 1; }
  ranges.emplace_back(last, (int32_t)path.size());
}

// ---------------------------------------------------------------------------------------
// seed code / max-hash
// ---------------------------------------------------------------------------------------
// graph.h:326-331 (G0 A1 T2 C3; anything else 0) as a table: a switch on random bases mispredicts every other time
struct BaseCodeTable {
  uint8_t v[256];
  constexpr BaseCodeTable() : v() { v[(unsigned char)'A'] = 1; v[(unsigned char)'T'] = 2; v[(unsigned char)'C'] = 3; }
};
static constexpr BaseCodeTable kBaseCode{};
static inline uint32_t base_code(char c) { return kBaseCode.v[(unsigned char)c]; }
static inline uint64_t scramble(uint64_t code) { return code ^ 0x2204abcdull; }  // graph.cc:1245
static const uint64_t kSeedMask = (1ull << (2 * kSeed)) - 1;

uint64_t read_max_hash(const char* s, int32_t n) {
  uint64_t code = 0, best = 0;
  for (int32_t i = 0; i < n; i++) {
    code = ((code << 2) & kSeedMask) + base_code(s[i]);
    if (i >= kSeed - 1) best = std::max(best, scramble(code));
  }
  return best;
}

void span_maxima(const char* s, int32_t n, int32_t read_len, std::vector<std::pair<uint64_t, int32_t>>& out) {
  // For every i >= read_len-1: M(i) = max scrambled seed code over seeds whose last base lies in
  // [i-read_len+kSeed, i], P(i) = the earliest such seed attaining it. Emit (M(i), P(i)) at the
  // first full span and whenever M(i) != M(i-1)  (graph.cc:1303-1321: `last_mh` always equals the
  // previous span's maximum). Monotone queue on two flat arrays.
  if (n < kSeed) return;
  std::vector<uint64_t> qh(n);
  std::vector<int32_t> qp(n);
  int32_t head = 0, tail = 0;
  uint64_t code = 0, prev = 0;
  for (int32_t i = 0; i < n; i++) {
    code = ((code << 2) & kSeedMask) + base_code(s[i]);
    if (i < kSeed - 1) continue;
    uint64_t h = scramble(code);
    if (i > kSeed - 1) {
      while (head < tail && qp[head] < i - read_len + kSeed) head++;
      while (head < tail && qh[tail - 1] < h) tail--;
    }
    qh[tail] = h; qp[tail] = i; tail++;
    if (i >= kSeed && i >= read_len - 1) {
      uint64_t top = qh[head];
      if (i == read_len - 1 || top != prev) { out.emplace_back(top, qp[head]); prev = top; }
    }
  }
}

// ---------------------------------------------------------------------------------------
// seed extension: same accept set, error counts and end points as the reference's 0-1 BFS
// (graph.cc:730-837), computed as a FIFO of "chain heads":
//   a zero-cost diagonal step is pushed to the FRONT of the reference's deque and therefore
//   popped next, so a popped state simply slides down its diagonal while bases match and the
//   next cell is unvisited; at the first mismatch it appends its (<=3) successors at cost+1 to
//   the back. Heads leave the queue in non-decreasing cost; the first head that slides off the
//   read end wins; a head with cost > 3 means failure.
// Visited cells are kept as one bitset per diagonal (|shift| <= 4 because cost <= 4).
// ---------------------------------------------------------------------------------------
namespace {
struct Visited {
  int32_t words;
  std::vector<uint64_t> bits;
  void reset(int32_t rlen) {
    words = (rlen + 2 + 63) / 64;
    bits.assign((size_t)9 * words, 0);
  }
  // returns true if (diag k, read index r) was unvisited, and marks it
  bool mark(int32_t k, int32_t r) {
    int32_t b = r + 1;
    uint64_t& w = bits[(size_t)(k + 4) * words + (b >> 6)];
    uint64_t m = 1ull << (b & 63);
    if (w & m) return false;
    w |= m;
    return true;
  }
};
struct Head { int32_t d, g, r; };
}  // namespace

bool extend_seed(int32_t win_pos, int32_t read_pos, const char* read, int32_t R, const char* win, int32_t W,
                 Extension* out) {
  static thread_local Visited vis;
  static thread_local std::vector<Head> q;
  auto diag = [&](int32_t g, int32_t r) { return (g - win_pos) - (r - read_pos); };

  // ---- forward (graph.cc:761-793)
  int32_t fwd = -1, end_pos = -1;
  vis.reset(R);
  q.clear();
  q.push_back(Head{0, win_pos + kSeed, read_pos + kSeed});
  for (size_t qi = 0; qi < q.size() && fwd < 0; qi++) {
    Head h = q[qi];
    if (h.d > 3) return false;
    int32_t g = h.g, r = h.r;
    while (true) {
      if (r == R) { fwd = h.d; end_pos = g - 1; break; }
      char wc = g < W ? win[g] : '\0';  // the reference reads the string terminator at g == W
      if (wc == read[r]) {
        if (g + 1 < W || r + 1 == R) {
          if (!vis.mark(diag(g + 1, r + 1), r + 1)) break;
          g++; r++;
          continue;
        }
        break;
      }
      if (g + 1 < W) {
        if (vis.mark(diag(g + 1, r + 1), r + 1)) q.push_back(Head{h.d + 1, g + 1, r + 1});
        if (vis.mark(diag(g + 1, r), r)) q.push_back(Head{h.d + 1, g + 1, r});
      }
      if (vis.mark(diag(g, r + 1), r + 1)) q.push_back(Head{h.d + 1, g, r + 1});
      break;
    }
  }
  if (fwd < 0) return false;

  // ---- backward (graph.cc:794-835)
  int32_t bwd = -1, begin_pos = -1;
  if (win_pos == 0) {
    if (read_pos < 6) bwd = read_pos;
  } else {
    vis.reset(R);
    q.clear();
    q.push_back(Head{0, win_pos - 1, read_pos - 1});
    for (size_t qi = 0; qi < q.size() && bwd < 0; qi++) {
      Head h = q[qi];
      if (h.d > 3) return false;
      int32_t g = h.g, r = h.r;
      while (true) {
        if (r == -1) { bwd = h.d; begin_pos = g + 1; break; }
        if (win[g] == read[r]) {
          if (g - 1 >= 0 || r - 1 == -1) {
            if (!vis.mark(diag(g - 1, r - 1), r - 1)) break;
            g--; r--;
            continue;
          }
          break;
        }
        if (g - 1 >= 0) {
          if (vis.mark(diag(g - 1, r - 1), r - 1)) q.push_back(Head{h.d + 1, g - 1, r - 1});
          if (vis.mark(diag(g - 1, r), r)) q.push_back(Head{h.d + 1, g - 1, r});
        }
        if (vis.mark(diag(g, r - 1), r - 1)) q.push_back(Head{h.d + 1, g, r - 1});
        break;
      }
    }
  }
  if (bwd < 0) return false;
  out->errs = fwd + bwd; out->begin = begin_pos; out->end = end_pos;
  return true;
}

// ---------------------------------------------------------------------------------------
// ShortMate
// ---------------------------------------------------------------------------------------
namespace {
// run fn(lo, hi) over [0, n) on a few host threads (ranges are disjoint; small inputs stay on the caller)
template <class F>
void parallel_ranges(int64_t n, F fn) {
  const int nt = n < (1 << 16) ? 1 : (int)std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
  if (nt == 1) { fn((int64_t)0, n); return; }
  std::vector<std::thread> pool;
  for (int t = 0; t < nt; t++) pool.emplace_back(fn, n * t / nt, n * (t + 1) / nt);
  for (auto& th : pool) th.join();
}
// std::sort of a large vector on a few threads: sorted chunks, then rounds of pairwise merges (same result as
// std::sort for a strict weak order over distinct elements; equal elements may end up in another order)
template <class T>
void parallel_sort(std::vector<T>& v) {
  const size_t n = v.size();
  unsigned nt = n < (1u << 17) ? 1u : std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency()));
  while (nt & (nt - 1)) nt &= nt - 1;  // a power of two
  if (nt <= 1) { std::sort(v.begin(), v.end()); return; }
  std::vector<size_t> cut(nt + 1);
  for (unsigned t = 0; t <= nt; t++) cut[t] = n * t / nt;
  {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; t++) pool.emplace_back([&v, &cut, t] { std::sort(v.begin() + cut[t], v.begin() + cut[t + 1]); });
    for (auto& th : pool) th.join();
  }
  std::vector<T> buf(n);
  std::vector<T>* src = &v;
  std::vector<T>* dst = &buf;
  for (unsigned width = 1; width < nt; width *= 2) {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; t += 2 * width)
      pool.emplace_back([src, dst, &cut, t, width] {
        std::merge(src->begin() + cut[t], src->begin() + cut[t + width], src->begin() + cut[t + width], src->begin() + cut[t + 2 * width],
                   dst->begin() + cut[t]);
      });
    for (auto& th : pool) th.join();
    std::swap(src, dst);
  }
  if (src != &v) v.swap(*src);
}
template <class F>
void parallel_mates(bool parallel, F fn) {  // fn(0) and fn(1) touch different arrays
  if (!parallel) { fn(0); fn(1); return; }
  std::thread other(fn, 1);
  fn(0);
  other.join();
}
}  // namespace

void ShortMate::set_reads(int64_t n_global_, int64_t lo_, int64_t hi_, const char* b, const int64_t* offs) {
  n_global = n_global_; lo = lo_; hi = hi_;
  int64_t n = hi - lo;
  roff.assign(n + 1, 0);
  lens.assign(n, 0);
  max_len = 0;
  bases.assign(b + offs[lo], b + offs[hi]);
  for (int64_t i = 0; i < n; i++) {
    roff[i] = offs[lo + i] - offs[lo];
    lens[i] = (int32_t)(offs[lo + i + 1] - offs[lo + i]);
    max_len = std::max(max_len, lens[i]);
  }
  roff[n] = offs[hi] - offs[lo];
  // pow tables sized max_read_len + 7 (graph.cc:1448-1453). NOTE: with sharding max_len is the
  // shard's maximum; tables are indexed by edit count and by L - edit only, both <= L.
  match_pow.resize(max_len + 7);
  mismatch_pow.resize(max_len + 7);
  for (size_t i = 0; i < match_pow.size(); i++) {
    match_pow[i] = std::pow(match, (double)i);
    mismatch_pow[i] = std::pow(mismatch, (double)i);
  }
  static const bool trace = getenv("GAML_HIP_TRACE_HOST") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  build_index();
  if (trace) fprintf(stderr, "set_reads: index over %lld reads built in %.0f ms\n", (long long)n,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}

static bool only_acgt(const char* s, int32_t n) {  // CheckRead graph.cc:1271-1278
  unsigned bad = 0;
  for (int32_t i = 0; i < n; i++) bad |= (unsigned)(kBaseCode.v[(unsigned char)s[i]] == 0 && s[i] != 'G');
  return bad == 0;
}

void ShortMate::build_index() {
  // bucket = reads whose maximum scrambled 15-mer code equals the key (graph.cc:1280-1287);
  // flat sorted arrays instead of a hash map of vectors.
  int64_t n = n_local();
  const auto t0h = std::chrono::steady_clock::now();
  std::vector<uint64_t> hash(n);
  std::vector<uint8_t> ok(n);
  parallel_ranges(n, [&](int64_t lo_i, int64_t hi_i) {  // the per-read work (136 rolling codes per 150-base read)
    for (int64_t i = lo_i; i < hi_i; i++) {
      ok[i] = lens[i] >= kSeed && only_acgt(read(i), lens[i]);
      if (ok[i]) hash[i] = read_max_hash(read(i), lens[i]);
    }
  });
  static const bool trace = getenv("GAML_HIP_TRACE_HOST") != nullptr;
  const auto t1 = std::chrono::steady_clock::now();
  std::vector<std::pair<uint64_t, int32_t>> keyed;
  keyed.reserve(n);
  index_read_len = 0;
  for (int64_t i = 0; i < n; i++) {
    if (!ok[i]) continue;
    keyed.emplace_back(hash[i], (int32_t)i);
    index_read_len = lens[i];
  }
  const auto t2 = std::chrono::steady_clock::now();
  parallel_sort(keyed);
  if (trace) fprintf(stderr, "build_index: hash %.0f ms, gather %.0f ms, sort %.0f ms\n", std::chrono::duration<double, std::milli>(t1 - t0h).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count());
  bucket_hash.clear(); bucket_off.clear(); bucket_reads.clear();
  bucket_reads.reserve(keyed.size());
  for (size_t i = 0; i < keyed.size(); i++) {
    if (i == 0 || keyed[i].first != keyed[i - 1].first) { bucket_hash.push_back(keyed[i].first); bucket_off.push_back((int32_t)i); }
    bucket_reads.push_back(keyed[i].second);
  }
  bucket_off.push_back((int32_t)keyed.size());
}

std::string ShortMate::window_string(const GraphStore& g, const Walk& w, int32_t* offset) const {
  // first node of a multi-node window keeps only its last kTail bases, the last node only its
  // first kTail (graph.cc:846-857)
  std::string s;
  *offset = 0;
  for (size_t i = 0; i < w.size(); i++) {
    const char* ns = g.seq(w[i]);
    int32_t nl = g.len(w[i]);
    if (i == 0 && w.size() > 1 && nl > kTail) { *offset = nl - kTail; s.append(ns + *offset, kTail); }
    else if (i > 0 && nl > kTail && i + 1 == w.size()) s.append(ns, kTail);
    else s.append(ns, nl);
  }
  return s;
}

static inline char comp(char c) {
  switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return c; }
}
static void revcomp_into(const char* s, int32_t n, std::string& out) {
  out.resize(n);
  for (int32_t i = 0; i < n; i++) out[i] = comp(s[n - 1 - i]);
}

void ShortMate::tag_window(const GraphStore& g, int32_t wid, const Walk& w) {
  Window& win = wins[wid];
  if (w.size() == 1) { win.solo = w[0]; solo_of_node[w[0]] = wid; }
  else if (w.size() > 1 && w[0] >= 0 && g.len(w[0]) > kTail) win.head = w[0];  // the condition of placements_paired_contig
}

int32_t ShortMate::add_window(const GraphStore& g, const Walk& w, std::vector<gaml_aligment>& recs) {
  auto it = win_id.find(w);
  if (it != win_id.end()) return it->second;  // re-alignment would give the same records
  int32_t id = (int32_t)wins.size();
  wins.push_back(Window());
  tag_window(g, id, w);
  auto ins = win_id.emplace(w, id);
  win_walk.push_back(&ins.first->first);
  unsynced.push_back(id);
  added_log.push_back(id);
  generation++;
  finalize_window(id, recs);
  return id;
}

void ShortMate::finalize_window(int32_t wid, std::vector<gaml_aligment>& recs) {
  Window& win = wins[wid];
  win.first = (int64_t)pool.size();
  win.count = (int32_t)recs.size();
  win.max_pos = INT_MIN;
  for (auto& r : recs) win.max_pos = std::max(win.max_pos, r.position);
  win.global_max_pos = win.max_pos;
  win.pending = false;
  pool.insert(pool.end(), recs.begin(), recs.end());
  filed.push_back(wid);
}

void ShortMate::cpu_align_records(const GraphStore& g, const Walk& w, std::vector<gaml_aligment>& recs) const {
  int32_t offset = 0;
  std::string ws = window_string(g, w, &offset), rc;
  revcomp_into(ws.data(), (int32_t)ws.size(), rc);
  const int32_t W = (int32_t)ws.size();

  struct Hit { int32_t pos, edit, read, orient, order; };
  std::vector<Hit> hits;
  std::vector<std::pair<uint64_t, int32_t>> spans;
  std::string rbuf;
  int32_t order = 0;
  for (int strand = 0; strand < 2; strand++) {
    spans.clear();
    span_maxima(strand == 0 ? ws.data() : rc.data(), W, index_read_len, spans);
    for (auto& sp : spans) {
      auto b = std::lower_bound(bucket_hash.begin(), bucket_hash.end(), sp.first);
      if (b == bucket_hash.end() || *b != sp.first) continue;
      size_t bi = b - bucket_hash.begin();
      // seed start in the forward window string (graph.cc:866-872)
      int32_t win_pos = strand == 0 ? sp.second - kSeed + 1 : W - (sp.second + 1);
      for (int32_t k = bucket_off[bi]; k < bucket_off[bi + 1]; k++) {
        int32_t rid = bucket_reads[k];
        const char* rs = read(rid);
        int32_t R = lens[rid];
        if (strand == 1) { revcomp_into(rs, R, rbuf); rs = rbuf.data(); }
        // first position of the read carrying the window's seed (graph.cc:873-879)
        int32_t read_pos = -1;
        for (int32_t i = 0; i + kSeed <= R; i++)
          if (memcmp(rs + i, ws.data() + win_pos, kSeed) == 0) { read_pos = i; break; }
        if (read_pos < 0) continue;  // cannot happen for ACGT windows: the bucket key is a seed of this read
        Extension e;
        if (extend_seed(win_pos, read_pos, rs, R, ws.data(), W, &e))
          hits.push_back(Hit{e.begin + 1 + offset, e.errs, rid, strand, order++});
      }
    }
  }
  // the reference collects into a set ordered by (position, read): the first alignment found for a
  // key survives (graph.cc:841, 891, 895-897). Per read, its candidates are visited forward-strand
  // spans first, then reverse -- the order used above.
  std::sort(hits.begin(), hits.end(), [](const Hit& a, const Hit& b) {
    if (a.pos != b.pos) return a.pos < b.pos;
    if (a.read != b.read) return a.read < b.read;
    return a.order < b.order;
  });
  recs.clear();
  recs.reserve(hits.size());
  for (size_t i = 0; i < hits.size(); i++) {
    if (i > 0 && hits[i].pos == hits[i - 1].pos && hits[i].read == hits[i - 1].read) continue;
    recs.push_back(gaml_aligment{hits[i].pos, hits[i].edit, hits[i].read, hits[i].orient});
  }
}

int32_t ShortMate::align(const GraphStore& g, const Walk& w) {
  int32_t existing = find(w);
  if (existing >= 0) return existing;
  windows_aligned++;
  // register first: what the registration rules and the placements look at is cache membership
  int32_t id = (int32_t)wins.size();
  wins.push_back(Window());
  wins[id].pending = true;
  tag_window(g, id, w);
  auto ins = win_id.emplace(w, id);
  win_walk.push_back(&ins.first->first);
  unsynced.push_back(id);
  added_log.push_back(id);
  generation++;
  if (defer_alignment) { pending.push_back(id); return id; }
  std::vector<gaml_aligment> recs;
  cpu_align_records(g, w, recs);
  finalize_window(id, recs);
  return id;
}

int64_t ShortMate::retire_unused() {
  int64_t n = 0;
  used.resize(wins.size(), 0);
  for (size_t w = 0; w < wins.size(); w++) {
    Window& win = wins[w];
    if (win.active && !used[w] && !win.pending) {
      win.active = false;
      active_records -= win.count;
      n++;
    }
    used[w] = 0;
  }
  if (n) active_generation++;
  return n;
}

void ShortMate::flush_pending_cpu(const GraphStore& g) {
  std::vector<gaml_aligment> recs;
  for (int32_t id : pending) {
    cpu_align_records(g, *win_walk[id], recs);
    finalize_window(id, recs);
  }
  pending.clear();
}

// ---------------------------------------------------------------------------------------
// window registration
// ---------------------------------------------------------------------------------------
// junction window starting at index i of `p` (stops at a gap when stop_at_gap): node i plus
// following nodes until more than kTail bases were added (graph.cc:453-469, 552-561)
static int32_t junction(const GraphStore& g, const int32_t* p, int32_t n, int32_t i, bool stop_at_gap, Walk& w) {
  w.clear();
  w.push_back(p[i]);
  int32_t tail = 0, end = i;
  for (int32_t j = i + 1; j < n; j++) {
    if (stop_at_gap && p[j] < 0) break;
    tail += g.len(p[j]);
    w.push_back(p[j]);
    end = j;
    if (tail > kTail) break;
  }
  return end;
}

static Walk inverted(const Walk& w) {  // InvertPath utility.h:28-38
  Walk r(w.rbegin(), w.rend());
  for (auto& x : r) if (x >= 0) x ^= 1;
  return r;
}

// one position of PrecomputeAlignmentForPaths' rule (graph.cc:471-482)
static void register_position(const GraphStore& g, ShortMate& m, const Walk& p, int32_t i, int32_t last_end,
                              int32_t* end_out, bool* skipped_for_last_end) {
  Walk w;
  const int32_t n = (int32_t)p.size();
  int32_t end = junction(g, p.data(), n, i, true, w);
  const bool by_size = (w.size() == 1 && g.len(w[0]) > 150);
  if (skipped_for_last_end) *skipped_for_last_end = false;
  if (m.find(w) < 0) {
    if (last_end != end || by_size) {
      m.align(g, w);
      m.align(g, inverted(w));
    } else if (skipped_for_last_end) {
      *skipped_for_last_end = true;  // a different predecessor path would have registered it
    }
  }
  if (g.len(p[i]) > kTail) {
    Walk one(1, p[i]);
    if (m.find(one) < 0) { m.align(g, one); m.align(g, Walk(1, p[i] ^ 1)); }
  }
  *end_out = end;
}

// PrecomputeAlignmentForPaths for ONE path given the last_end left by its predecessor; returns the
// last_end it leaves. *first_skipped reports whether the first node position was not registered
// only because of the incoming last_end.
static int32_t register_one_path(const GraphStore& g, ShortMate& m, const Walk& p, int32_t last_end, bool* first_skipped) {
  bool first = true;
  if (first_skipped) *first_skipped = false;
  for (int32_t i = 0; i < (int32_t)p.size(); i++) {
    if (p[i] < 0) continue;
    int32_t end;
    bool skipped = false;
    register_position(g, m, p, i, last_end, &end, first ? &skipped : nullptr);
    if (first && first_skipped) *first_skipped = skipped;
    first = false;
    last_end = end;
  }
  return last_end;
}

void register_for_paths(const GraphStore& g, ShortMate& m, const std::vector<Walk>& paths) {
  // PrecomputeAlignmentForPaths (graph.cc:447-493). `last_end` deliberately survives from one
  // path to the next, as in the reference (:449).
  int32_t last_end = -1;
  for (const Walk& p : paths) last_end = register_one_path(g, m, p, last_end, nullptr);
}

void register_for_contig(const GraphStore& g, ShortMate& m, const int32_t* ctg, int32_t n) {
  // GetSubpathsFromPath + the precompute that follows it (graph.cc:495-533, 538-542)
  int32_t last_end = -1;
  Walk w;
  for (int32_t i = 0; i < n; i++) {
    if (ctg[i] < 0) continue;
    int32_t end = junction(g, ctg, n, i, true, w);
    if (end != last_end && m.find(w) < 0) m.align(g, w);
    last_end = end;
  }
}

// ---------------------------------------------------------------------------------------
// occurrences
// ---------------------------------------------------------------------------------------
void placements_paired_contig(const GraphStore& g, const ShortMate& m, const int32_t* ctg, int32_t n, int32_t st,
                              int32_t path, int32_t contig_serial, std::vector<Placement>& out) {
  // GetPositionsOnlyPath (graph.cc:544-573): node i looks up its junction window and, when the node
  // is longer than kTail, its single-node window. A window that is not cached at this moment
  // contributes nothing (graph.cc:571-573). When the junction window IS the single node, the
  // reference scans the same records twice; the second pass rewrites identical values, so one
  // placement is recorded.
  int32_t cur_pos = st;
  Walk w;
  for (int32_t i = 0; i < n; i++) {
    junction(g, ctg, n, i, false, w);
    int32_t ids[2] = {m.find(w), -1};
    if (g.len(ctg[i]) > kTail && w.size() > 1) ids[1] = m.find(Walk(1, ctg[i]));
    for (int32_t id : ids)
      if (id >= 0) out.push_back(Placement{id, cur_pos, path, contig_serial, i});
    cur_pos += g.len(ctg[i]);
  }
}

void occurrences_from_placements(ShortMate& m, const std::vector<Placement>& pl, std::vector<Occ>& out) {
  // The reference drops a record when its path position is < max_pos - 5 (graph.cc:577), max_pos
  // being the largest position kept at EARLIER nodes of the contig (never below 0, reset per
  // contig). Because the largest record of a node is kept whenever any record of the node is,
  // max_pos before node i equals max(0, max over earlier nodes of (node offset + largest record
  // position over the node's cached windows)): a prefix maximum that needs no per-record state --
  // and, in a sharded run, the largest position over ALL shards (Window::global_max_pos).
  int32_t rank = 0, max_pos = 0, node_max = INT_MIN, node_shift = 0;
  int32_t cur_contig = -1, cur_node = -1;
  for (const Placement& p : pl) {
    if (p.contig != cur_contig || p.node != cur_node) {
      if (node_max != INT_MIN) max_pos = std::max(max_pos, node_shift + node_max);  // commit the finished node
      if (p.contig != cur_contig) max_pos = 0;
      cur_contig = p.contig; cur_node = p.node; node_max = INT_MIN; node_shift = p.shift;
    }
    const Window& win = m.wins[p.wid];
    if (win.count > 0) {
      m.activate(p.wid);
      out.push_back(Occ{p.wid, p.shift, (max_pos - 5) - p.shift, p.path, rank++});
    }
    if (win.global_max_pos != INT_MIN) node_max = std::max(node_max, win.global_max_pos);
  }
}

void occurrences_single_contig(const GraphStore& g, ShortMate& m, const int32_t* ctg, int32_t n, int32_t st,
                               int32_t* rank, std::vector<Occ>& out) {
  // AddPositions (graph.cc:611-647): junction windows only, no position filter; a window missing
  // from the cache contributes nothing.
  int32_t cur_pos = st;
  Walk w;
  for (int32_t i = 0; i < n; i++) {
    junction(g, ctg, n, i, false, w);
    int32_t id = m.find(w);
    if (id >= 0 && m.wins[id].count > 0) { m.activate(id); out.push_back(Occ{id, cur_pos, INT_MIN / 2, 0, (*rank)++}); }
    cur_pos += g.len(ctg[i]);
  }
}

// ---------------------------------------------------------------------------------------
// PairedPlanner
// ---------------------------------------------------------------------------------------
int32_t PairedPlanner::lookup_or_create(const GraphStore& g, const int32_t* pp, int32_t len, std::string* err) {
  Walk p(pp, pp + len);
  auto it = by_path_.find(p);
  if (it != by_path_.end()) { hits++; return it->second; }
  for (int32_t x : p) if (x >= g.n()) { if (err) *err = "path refers to a node outside the graph"; return -1; }
  misses++;
  int32_t id;
  if (memos_.size() < kMaxMemos) {
    id = (int32_t)memos_.size();
    memos_.emplace_back(new PathMemo());
  } else {
    // evict the least recently used memo that is not part of the current path set
    id = -1;
    uint64_t best = ~0ull;
    for (int32_t k = 0; k < (int32_t)memos_.size(); k++)
      if (memos_[k]->use_count == 0 && memos_[k]->last_used < best) { best = memos_[k]->last_used; id = k; }
    if (id < 0) { id = (int32_t)memos_.size(); memos_.emplace_back(new PathMemo()); }
    else { by_path_.erase(memos_[id]->path); uint32_t ser = memos_[id]->serial + 1; memos_[id].reset(new PathMemo()); memos_[id]->serial = ser; }
  }
  PathMemo& pm = *memos_[id];
  pm.path = p;
  pm.touched_serial = retire_serial_;  // its lists are yet to be computed (and activate what they name)
  by_path_.emplace(p, id);
  // path-only facts: contig starts, length, first node position and its window end, final last_end
  std::vector<std::pair<int32_t, int32_t>> ranges;
  std::vector<int32_t> gaps;
  split_contigs(p, ranges, gaps);
  int32_t cur = 0;
  pm.starts.push_back(0);
  for (size_t ci = 0; ci < ranges.size(); ci++) {
    if (ci > 0) { cur += gaps[ci - 1]; pm.starts.push_back(cur); }
    for (int32_t k = ranges[ci].first; k < ranges[ci].second; k++) cur += g.len(p[k]);
  }
  pm.length = walk_length(g, p);  // GetTotalLen counts every gap, also a leading / trailing one (graph.cc:1775-1781)
  Walk w;
  for (int32_t i = 0; i < (int32_t)p.size(); i++) {
    if (p[i] < 0) continue;
    int32_t end = junction(g, p.data(), (int32_t)p.size(), i, true, w);
    if (pm.first_idx < 0) { pm.first_idx = i; pm.first_end = end; }
    pm.final_last_end = end;
  }
  return id;
}

void PairedPlanner::drain(ShortMate mate[2]) {
  for (int mt = 0; mt < 2; mt++) {
    ShortMate& m = mate[mt];
    if (m.added_log.empty()) continue;
    if (!missed_[mt].empty()) {
      for (int32_t wid : m.added_log) {
        auto it = missed_[mt].find(*m.win_walk[wid]);
        if (it == missed_[mt].end()) continue;
        for (auto& ref : it->second)
          if (ref.first < (int32_t)memos_.size() && memos_[ref.first]->serial == ref.second) {
            PathMemo& pm = *memos_[ref.first];
            if (pm.valid[mt] && pm.use_count > 0) stale_.push_back(ref.first);  // in the current set: its table entries need a refresh
            pm.valid[mt] = false;
          }
        missed_[mt].erase(it);
      }
    }
    m.added_log.clear();
  }
}

void PairedPlanner::build_placements(const GraphStore& g, ShortMate& m, int mt, PathMemo& pm, int32_t id) {
  // per contig: register its windows, then note which cached windows sit where (graph.cc:1830-1844)
  std::vector<std::pair<int32_t, int32_t>> ranges;
  std::vector<int32_t> gaps;
  split_contigs(pm.path, ranges, gaps);
  pm.pl[mt].clear();
  int32_t cur_len = 0;
  Walk w;
  for (size_t ci = 0; ci < ranges.size(); ci++) {
    if (ci > 0) cur_len += gaps[ci - 1];
    const int32_t* ctg = pm.path.data() + ranges[ci].first;
    const int32_t n = ranges[ci].second - ranges[ci].first;
    register_for_contig(g, m, ctg, n);
    placements_paired_contig(g, m, ctg, n, cur_len, 0, (int32_t)ci, pm.pl[mt]);
    // remember the keys this contig looked up and missed
    for (int32_t i = 0; i < n; i++) {
      junction(g, ctg, n, i, false, w);
      if (m.find(w) < 0) missed_[mt][w].emplace_back(id, pm.serial);
      if (g.len(ctg[i]) > kTail && w.size() > 1) {
        Walk one(1, ctg[i]);
        if (m.find(one) < 0) missed_[mt][one].emplace_back(id, pm.serial);
      }
    }
    for (int32_t k = 0; k < n; k++) cur_len += g.len(ctg[k]);
  }
  pm.valid[mt] = true;
  pm.occ_valid[mt] = false;
}

// phase 1 (PrecomputeAlignmentForPaths graph.cc:447-493, per mate) for the paths [from, to) of the current set. The
// rule of a path's first node depends on the last_end its predecessor leaves (:449, 471-472): taken from the nearest
// earlier path that has a node.
void PairedPlanner::registration_chain(const GraphStore& g, ShortMate mate[2], int32_t from, int32_t to) {
  for (int mt = 0; mt < 2; mt++) {
    int32_t last_end = -1;
    for (int32_t k = from - 1; k >= 0; k--)
      if (memos_[cur_ids_[k]]->final_last_end != -2) { last_end = memos_[cur_ids_[k]]->final_last_end; break; }
    for (int32_t k = from; k < to; k++) {
      PathMemo& pm = *memos_[cur_ids_[k]];
      if (!pm.registered[mt]) {
        bool skipped = false;
        register_one_path(g, mate[mt], pm.path, last_end, &skipped);
        pm.registered[mt] = true;
        pm.first_pending[mt] = skipped;
      } else if (pm.first_pending[mt] && pm.first_idx >= 0 && last_end != pm.first_end) {
        int32_t end;
        bool skipped = false;
        register_position(g, mate[mt], pm.path, pm.first_idx, last_end, &end, &skipped);
        pm.first_pending[mt] = skipped;
      }
      if (pm.final_last_end != -2) last_end = pm.final_last_end;
    }
  }
}

bool PairedPlanner::begin(const GraphStore& g, ShortMate mate[2], const int32_t* flat, const int64_t* offs, int32_t n_paths,
                          bool allow_incremental, std::string* err) {
  clock_++;
  view_valid_ = false;
  removed_.clear();
  work_.clear();
  drain(mate);  // windows aligned since the last call invalidate the memos that had looked them up in vain
  const int64_t base = n_paths > 0 ? offs[0] : 0;
  auto path_ptr = [&](int32_t k) { return flat + offs[k]; };
  auto path_len = [&](int32_t k) { return (int32_t)(offs[k + 1] - offs[k]); };
  // ---- what changed against the previous call: common prefix / suffix of whole paths
  int32_t n_prev = have_prev_ ? (int32_t)prev_offs_.size() - 1 : 0;
  int32_t P = 0, S = 0;
  incremental_ = allow_incremental && have_prev_ && n_prev > 0 && n_paths > 0;
  if (incremental_) {
    auto same = [&](int32_t a, int32_t b) {  // path a of the previous set == path b of this one
      const int64_t la = prev_offs_[a + 1] - prev_offs_[a];
      return la == path_len(b) && (la == 0 || memcmp(prev_flat_.data() + prev_offs_[a], path_ptr(b), (size_t)la * sizeof(int32_t)) == 0);
    };
    const int32_t lim = std::min(n_prev, n_paths);
    {
      // common prefix: the first differing int of the two flat arrays bounds it (one pass, no call per path)
      const int64_t max_ints = std::min<int64_t>(prev_offs_[n_prev], offs[n_paths] - base);
      const int32_t* a = prev_flat_.data();
      const int32_t* b = flat + base;
      int64_t d = 0;
      while (d + 8 <= max_ints && memcmp(a + d, b + d, 8 * sizeof(int32_t)) == 0) d += 8;
      while (d < max_ints && a[d] == b[d]) d++;
      // paths that end at or before d with equal boundaries are equal
      while (P < lim && prev_offs_[P + 1] == offs[P + 1] - base && prev_offs_[P + 1] <= d) P++;
    }
    while (S < lim - P && same(n_prev - 1 - S, n_paths - 1 - S)) S++;
    // Between the common prefix and suffix: an edit script of (previous range -> new range) pairs with the common
    // runs between them left alone. Two candidates of one assembly differ from each other in BOTH their edits, and
    // what lies between the two is most of the set: greedy -- at a mismatch take the closest pair of equal paths
    // within a few positions as the next point where the two lists run together again.
    edits_.clear();
    int32_t i = P, j = P;
    const int32_t pe = n_prev - S, ce = n_paths - S;
    int64_t changed = 0;
    while (i < pe || j < ce) {
      constexpr int32_t kReach = 4;
      int32_t best_a = -1, best_b = -1;
      for (int32_t sum = 1; sum <= 2 * kReach && best_a < 0; sum++)
        for (int32_t a = std::max(0, sum - kReach); a <= std::min(sum, kReach); a++) {
          const int32_t b = sum - a;
          if (i + a < pe && j + b < ce && same(i + a, j + b)) { best_a = a; best_b = b; break; }
        }
      if (best_a < 0) { edits_.push_back(Edit{i, pe, j, ce}); changed += (pe - i) + (ce - j); break; }
      edits_.push_back(Edit{i, i + best_a, j, j + best_b});
      changed += best_a + best_b;
      i += best_a; j += best_b;
      // the common run: first differing int of the two flat arrays from here, then whole paths inside it
      const int64_t a0 = prev_offs_[i], b0 = offs[j] - base;
      const int64_t max_ints = std::min<int64_t>(prev_offs_[pe] - a0, (offs[ce] - base) - b0);
      const int32_t* pa = prev_flat_.data() + a0;
      const int32_t* pb = flat + base + b0;
      int64_t d = 0;
      while (d + 8 <= max_ints && memcmp(pa + d, pb + d, 8 * sizeof(int32_t)) == 0) d += 8;
      while (d < max_ints && pa[d] == pb[d]) d++;
      while (i < pe && j < ce && prev_offs_[i + 1] - a0 == (offs[j + 1] - base) - b0 && prev_offs_[i + 1] - a0 <= d) { i++; j++; }
    }
    // more than half of the set changed: the whole-set rebuild is cheaper than path-by-path bookkeeping
    if (changed > (int64_t)(n_prev + n_paths) / 2) incremental_ = false;
  }
  if (!incremental_) {
    // ---- from scratch: every path looked up, slots = positions
    full_calls++;
    for (int32_t id : cur_ids_) memos_[id]->use_count--;
    cur_ids_.clear(); cur_slots_.clear(); free_slots_.clear(); stale_.clear();
    total_len_ = 0;
    for (int32_t k = 0; k < n_paths; k++) {
      const int32_t id = lookup_or_create(g, path_ptr(k), path_len(k), err);
      if (id < 0) { for (auto& pm : memos_) pm->use_count = 0; cur_ids_.clear(); cur_slots_.clear(); have_prev_ = false; return false; }
      memos_[id]->last_used = clock_;
      memos_[id]->use_count++;
      cur_ids_.push_back(id);
      cur_slots_.push_back(k);
      total_len_ += memos_[id]->length;
    }
    next_slot_ = n_paths;
    registration_chain(g, mate, 0, n_paths);
    // phase 2: per path in order: (re)build placements where the memo is stale
    for (int32_t id : cur_ids_) {
      drain(mate);
      PathMemo& pm = *memos_[id];
      for (int mt = 0; mt < 2; mt++)
        if (!pm.valid[mt]) build_placements(g, mate[mt], mt, pm, id);
    }
    drain(mate);
    stale_.clear();  // (every memo of the set is looked at again by the next from-scratch call; an incremental one re-collects)
    for (int32_t id : cur_ids_) { const PathMemo& pm = *memos_[id]; if (!pm.valid[0] || !pm.valid[1]) stale_.push_back(id); }
  } else {
    // ---- incremental: per edit, paths [p0, p1) of the previous set leave, paths [c0, c1) of this one enter
    incremental_calls++;
    std::vector<int32_t> freed_now;  // handed out again from the next call on
    for (const Edit& e : edits_)
      for (int32_t k = e.p0; k < e.p1; k++) {
        PathMemo& pm = *memos_[cur_ids_[k]];
        removed_.push_back(Removed{cur_slots_[k], {pm.assembled[0], pm.assembled[1]}, {pm.occ[0], pm.occ[1]}});  // lists as they stand: the memo may be evicted or rebuilt before apply()
        pm.use_count--;
        pm.last_used = clock_;  // in use up to the previous call (window retirement asks which paths were scored since the last rebuild)
        total_len_ -= pm.length;
        freed_now.push_back(cur_slots_[k]);
      }
    std::vector<int32_t> in_ids, in_slots;  // of all edits, in path order
    for (const Edit& e : edits_)
      for (int32_t k = e.c0; k < e.c1; k++) {
        const int32_t id = lookup_or_create(g, path_ptr(k), path_len(k), err);
        if (id < 0) {  // leave a consistent "nothing known" state: the next call starts from scratch
          for (auto& pm : memos_) pm->use_count = 0;
          cur_ids_.clear(); cur_slots_.clear(); stale_.clear(); removed_.clear(); work_.clear(); have_prev_ = false; incremental_ = false;
          return false;
        }
        memos_[id]->last_used = clock_;
        memos_[id]->use_count++;
        total_len_ += memos_[id]->length;
        in_ids.push_back(id);
        int32_t slot;
        if (!free_slots_.empty()) { slot = free_slots_.back(); free_slots_.pop_back(); }
        else slot = next_slot_++;
        in_slots.push_back(slot);
      }
    free_slots_.insert(free_slots_.end(), freed_now.begin(), freed_now.end());
    {  // splice back to front: positions of earlier edits stay valid
      size_t taken = in_ids.size();
      for (size_t q = edits_.size(); q-- > 0;) {
        const Edit& e = edits_[q];
        const size_t cnt = (size_t)(e.c1 - e.c0);
        taken -= cnt;
        cur_ids_.erase(cur_ids_.begin() + e.p0, cur_ids_.begin() + e.p1);
        cur_ids_.insert(cur_ids_.begin() + e.p0, in_ids.begin() + taken, in_ids.begin() + taken + cnt);
        cur_slots_.erase(cur_slots_.begin() + e.p0, cur_slots_.begin() + e.p1);
        cur_slots_.insert(cur_slots_.begin() + e.p0, in_slots.begin() + taken, in_slots.begin() + taken + cnt);
      }
    }
    is_new_.assign((size_t)n_paths, 0);
    for (const Edit& e : edits_) for (int32_t k = e.c0; k < e.c1; k++) is_new_[k] = 1;
    // registration: the new paths, and behind them everything up to the first path that has a node (its predecessor,
    // i.e. the last_end it sees, changed; paths without a node pass the value through)
    for (const Edit& e : edits_) {
      int32_t chain_end = e.c1;
      while (chain_end < n_paths) { const bool has_node = memos_[cur_ids_[chain_end]]->final_last_end != -2; chain_end++; if (has_node) break; }
      registration_chain(g, mate, e.c0, chain_end);
    }
    // memos to (re)place: the new paths, and memos of the set that were invalidated since they were placed. A memo
    // is rebuilt once for all its instances; every instance's table entries are taken out first (with the lists as
    // they stand) and put back after pass 2.
    std::vector<char>& redo = redo_;
    redo.assign(memos_.size(), 0);
    for (int32_t id : in_ids) redo[id] = 1;
    for (int32_t id : stale_) if (id < (int32_t)memos_.size() && memos_[id]->use_count > 0 && (!memos_[id]->valid[0] || !memos_[id]->valid[1] || !memos_[id]->occ_valid[0] || !memos_[id]->occ_valid[1])) redo[id] = 2;
    stale_.clear();
    // phase 2 in path order, as the whole-set pass does it: a placement built at position k may register windows
    // (per-contig rule) that invalidate other memos of the set -- one with an instance BEHIND k is redone in this call
    // too (the whole-set pass would reach it with valid = false), the others keep their tables until the next call
    drain(mate);
    size_t seen = stale_.size();
    for (int32_t k = 0; k < n_paths; k++) {
      const int32_t id = cur_ids_[k];
      if (!redo[id]) continue;
      PathMemo& pm = *memos_[id];
      bool built = false;
      for (int mt = 0; mt < 2; mt++)
        if (!pm.valid[mt]) { build_placements(g, mate[mt], mt, pm, id); built = true; }
      if (!built) continue;
      drain(mate);
      for (; seen < stale_.size(); seen++) {
        const int32_t sid = stale_[seen];
        if (sid >= (int32_t)redo.size() || redo[sid]) continue;
        for (int32_t q = k + 1; q < n_paths; q++) if (cur_ids_[q] == sid) { redo[sid] = 2; break; }
      }
    }
    // every instance of a memo that is being redone: instances that were in the tables before this call lose their
    // entries (lists as they stand: pass 2 has not touched them yet), all instances get (back) in after pass 2
    for (int32_t k = 0; k < n_paths; k++) {
      const int32_t id = cur_ids_[k];
      if (!redo[id]) continue;
      if (!is_new_[k]) {
        const PathMemo& pm = *memos_[id];
        removed_.push_back(Removed{cur_slots_[k], {pm.assembled[0], pm.assembled[1]}, {pm.occ[0], pm.occ[1]}});
      }
      work_.push_back(k);
    }
    // keep for the next call what this one could not settle: memos invalidated behind their last instance
    std::vector<int32_t> keep;
    for (int32_t sid : stale_) if (sid < (int32_t)memos_.size() && memos_[sid]->use_count > 0 && (!memos_[sid]->valid[0] || !memos_[sid]->valid[1])) keep.push_back(sid);
    stale_.swap(keep);
  }
  // remember this call's paths for the next diff
  const int64_t total = n_paths > 0 ? offs[n_paths] - base : 0;
  prev_flat_.assign(flat + base, flat + base + total);
  prev_offs_.resize((size_t)n_paths + 1);
  for (int32_t k = 0; k <= n_paths; k++) prev_offs_[k] = (n_paths > 0 ? offs[k] : 0) - base;
  have_prev_ = true;
  return true;
}

// the entries of a path's occurrences as OccImage::build writes them, minus the path slot (packed as occ8_pack packs them)
static void occ_prepack(const std::vector<Occ>& occ, std::vector<OccPre>& pre) {
  pre.resize(occ.size());
  for (size_t k = 0; k < occ.size(); k++) {
    const Occ& o = occ[k];
    const int32_t mp = o.min_pos < -32768 ? -32768 : o.min_pos;
    pre[k] = OccPre{o.min_pos <= 32767 ? o.wid : ~o.wid, (uint32_t)o.shift, (uint32_t)(uint16_t)(int16_t)mp, o.rank};
  }
}

void PairedPlanner::invalidate_thresholds() {
  for (auto& pm : memos_) pm->occ_valid[0] = pm->occ_valid[1] = false;
  have_prev_ = false;  // every occurrence list changes: the next call rebuilds the tables from scratch
}

void PairedPlanner::finish(ShortMate mate[2]) {
  auto redo_occ = [&](PathMemo& pm) {
    for (int mt = 0; mt < 2; mt++) {
      if (pm.occ_valid[mt]) continue;  // windows stay activated once used; nothing to redo
      pm.occ[mt].clear();
      occurrences_from_placements(mate[mt], pm.pl[mt], pm.occ[mt]);
      pm.assembled[mt] = 0;
      for (const Occ& o : pm.occ[mt]) pm.assembled[mt] += mate[mt].wins[o.wid].count;
      occ_prepack(pm.occ[mt], pm.pre[mt]);
      pm.occ_valid[mt] = true;
    }
  };
  if (!incremental_) {
    for (int32_t id : cur_ids_) redo_occ(*memos_[id]);
    assembled_[0] = assembled_[1] = 0;
    for (int32_t id : cur_ids_) { assembled_[0] += memos_[id]->assembled[0]; assembled_[1] += memos_[id]->assembled[1]; }
  } else {
    for (int32_t k : work_) redo_occ(*memos_[cur_ids_[k]]);
  }
}

const PlanView& PairedPlanner::view() {
  if (!view_valid_) {
    view_.paths.clear();
    for (int32_t id : cur_ids_) view_.paths.push_back(memos_[id].get());
    view_valid_ = true;
  }
  return view_;
}

void PairedPlanner::apply(ShortMate mate[2], OccImage image[2]) {
  if (!incremental_) {
    const PlanView& v = view();
    for (int32_t id : cur_ids_) {
      PathMemo& pm = *memos_[id];
      if (pm.touched_serial == retire_serial_) continue;
      for (int mt = 0; mt < 2; mt++) for (const Occ& o : pm.occ[mt]) mate[mt].touch(o.wid);  // a memoised list may name windows a rebuild retired
      pm.touched_serial = retire_serial_;
    }
    for (int mt = 0; mt < 2; mt++) image[mt].build(mate[mt].wins.size(), v, mt);
    return;
  }
  for (int32_t k : work_) {
    PathMemo& pm = *memos_[cur_ids_[k]];
    if (pm.touched_serial == retire_serial_) continue;
    for (int mt = 0; mt < 2; mt++) for (const Occ& o : pm.occ[mt]) mate[mt].touch(o.wid);
    pm.touched_serial = retire_serial_;
  }
  if ((int32_t)pos_of_slot_.size() < next_slot_) pos_of_slot_.resize((size_t)next_slot_ + 64, 0);
  for (int32_t k = 0; k < (int32_t)cur_slots_.size(); k++) pos_of_slot_[cur_slots_[k]] = k;
  for (int mt = 0; mt < 2; mt++) {
    OccImage& im = image[mt];
    for (const Removed& r : removed_) { im.remove_path(r.occ[mt], r.slot); assembled_[mt] -= r.assembled[mt]; }
    for (int32_t k : work_) {
      const PathMemo& pm = *memos_[cur_ids_[k]];
      im.add_path(mate[mt].wins.size(), pm.occ[mt], cur_slots_[k]);
      assembled_[mt] += pm.assembled[mt];
    }
    im.finalize(pos_of_slot_);
  }
}

void PairedPlanner::mark_used(ShortMate mate[2], const OccImage image[2]) {
  for (int mt = 0; mt < 2; mt++) image[mt].for_each_present([&](int32_t w) { mate[mt].touch(w); });
  for (auto& pm : memos_)
    if (pm->last_used > rebuild_clock_ || pm->use_count > 0)
      for (int mt = 0; mt < 2; mt++) for (const Occ& o : pm->occ[mt]) if (o.wid < (int32_t)mate[mt].wins.size()) mate[mt].touch(o.wid);
}

void PairedPlanner::flat_occurrences(int mate, std::vector<Occ>& out) {
  out.clear();
  int32_t rank0 = 0;
  const PlanView& v = view();
  for (size_t slot = 0; slot < v.paths.size(); slot++) {
    const PathMemo& pm = *v.paths[slot];
    for (const Occ& o : pm.occ[mate]) out.push_back(Occ{o.wid, o.shift, o.min_pos, (int32_t)slot, rank0 + o.rank});
    rank0 += (int32_t)pm.occ[mate].size();
  }
}

// ---------------------------------------------------------------------------------------
// device tables
// ---------------------------------------------------------------------------------------
void build_read_major(const ShortMate& m, const std::vector<int32_t>* slot_of_read, ReadMajor& out) {
  const int64_t n = m.n_local();
  auto slot = [&](int32_t read) { return slot_of_read ? (*slot_of_read)[read] : read; };
  std::vector<int32_t> cnt(n + 1, 0);
  for (const Window& w : m.wins) if (w.active) for (int64_t k = w.first; k < w.first + w.count; k++) cnt[slot(m.pool[k].read_id) + 1]++;
  std::vector<int64_t> start(n + 1, 0);
  int64_t extras = 0;
  for (int64_t i = 0; i < n; i++) { start[i] = extras; extras += cnt[i + 1] > 1 ? cnt[i + 1] - 1 : 0; }
  out.first.assign(n, RecQuad{-1, 0, 0, 0});
  out.extra.assign(extras, RecQuad{-1, 0, 0, 0});
  std::vector<int32_t> seen(n, 0);
  for (size_t wid = 0; wid < m.wins.size(); wid++) {
    const Window& win = m.wins[wid];
    if (!win.active) continue;
    for (int64_t k = win.first; k < win.first + win.count; k++) {
      const gaml_aligment& r = m.pool[k];
      const int32_t at = slot(r.read_id);
      RecQuad q{(int32_t)wid, r.position, (r.edit_dist & 0xff) | ((r.orientation & 1) << 8), 0};
      int32_t s = seen[at]++;
      if (s == 0) {
        int32_t more = cnt[at + 1] - 1;
        q.flags |= more << 9;  // extra_count in bits 9..31
        q.link = (int32_t)start[at];
        out.first[at] = q;
      } else {
        out.extra[start[at] + s - 1] = q;
      }
    }
  }
  out.total_records = m.active_records;
  out.built_generation = m.active_generation;
}

#ifdef GAML_HIP_DEV  // the host restatement of the device table build (table_build.hip.h): what the development build checks the device tables against
// Records that can never survive the overwrite rule (graph.cc:583-592), whatever the path set. A window J of several
// nodes whose first node i is longer than kTail is looked up together with the node's own window {i}, J first, both at
// the node's path position and under the same position filter (GetPositionsOnlyPath graph.cc:563-577;
// placements_paired_contig above). A record of J that {i} also holds for the same read at the same position -- a read
// in the last kTail bases of a long node, seen through the node and through the junction that follows it -- is
// therefore overwritten by {i}'s record wherever J occurs, and absent wherever J does not. Leaving it out of the
// device tables changes no value; it turns most two-record pairs (a tenth of the pairs at BASELINE config 3) into
// one-record pairs of the compact class. {i} must be on the device itself (active) for its record to stand in; a
// rebuild, which is also where windows retire, decides anew.
// Both windows' records are ordered by (position, read) (graph.h:229-232): one merge per junction window.
static void dominated_records(const ShortMate& m, std::vector<uint8_t>& drop) {
  drop.assign(m.pool.size(), 0);
  std::unordered_map<int32_t, int32_t> solo;  // node -> its active single-node window
  bool any_head = false;
  for (size_t wid = 0; wid < m.wins.size(); wid++) {
    const Window& w = m.wins[wid];
    if (!w.active || w.count == 0) continue;
    if (w.solo >= 0) solo.emplace(w.solo, (int32_t)wid);
    any_head |= w.head >= 0;
  }
  if (!any_head || solo.empty()) return;
  auto before = [](const gaml_aligment& x, const gaml_aligment& y) { return x.position != y.position ? x.position < y.position : x.read_id < y.read_id; };
  for (const Window& j : m.wins) {
    if (!j.active || j.count == 0 || j.head < 0) continue;
    auto it = solo.find(j.head);
    if (it == solo.end()) continue;
    const Window& sw = m.wins[it->second];
    const gaml_aligment* a = m.pool.data() + j.first, * const ae = a + j.count;
    const gaml_aligment* const sb = m.pool.data() + sw.first, * const se = sb + sw.count;
    const gaml_aligment* b = std::lower_bound(sb, se, *a, before);
    while (a < ae && b < se) {
      if (before(*a, *b)) a++;
      else if (before(*b, *a)) b++;
      else { drop[(size_t)(a - m.pool.data())] = 1; a++; }
    }
  }
}

static inline bool rec_before(const gaml_aligment& x, const gaml_aligment& y) { return x.position != y.position ? x.position < y.position : x.read_id < y.read_id; }

int64_t undominated_records(const ShortMate& m, int32_t wid, std::vector<uint8_t>& keep) {
  const Window& j = m.wins[wid];
  keep.assign((size_t)j.count, 1);
  if (j.head < 0 || j.count == 0) return j.count;
  auto it = m.solo_of_node.find(j.head);
  if (it == m.solo_of_node.end()) return j.count;
  const Window& sw = m.wins[it->second];
  if (!sw.active || sw.count == 0) return j.count;
  const gaml_aligment* const a0 = m.pool.data() + j.first, * a = a0, * const ae = a0 + j.count;
  const gaml_aligment* const sb = m.pool.data() + sw.first, * const se = sb + sw.count;
  const gaml_aligment* b = std::lower_bound(sb, se, *a, rec_before);
  int64_t kept = j.count;
  while (a < ae && b < se) {
    if (rec_before(*a, *b)) a++;
    else if (rec_before(*b, *a)) b++;
    else { keep[(size_t)(a - a0)] = 0; kept--; a++; }
  }
  return kept;
}

#endif  // GAML_HIP_DEV

void link_mate_windows(ShortMate& a, ShortMate& b) {
  for (size_t w = 0; w < a.wins.size(); w++) {
    if (a.wins[w].peer >= 0) continue;
    const int32_t p = b.find(*a.win_walk[w]);
    a.wins[w].peer = p;
    if (p >= 0) b.wins[p].peer = (int32_t)w;
  }
  for (Window& w : b.wins) if (w.peer == -2) w.peer = -1;  // (not in `a`, or the loop above would have linked it)
}

#ifdef GAML_HIP_DEV
void build_pair_tables(const ShortMate& a, const ShortMate& b, PairTables& out, bool fold, int ins_n) {
  static const bool trace_bpt = getenv("GAML_HIP_TRACE_HOST") != nullptr;
  auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_stage[8]; int n_stage = 0;
  t_stage[n_stage++] = now_ms();
  const int64_t n = a.n_local();
  const ShortMate* mates[2] = {&a, &b};
  // per read: record count over ACTIVE windows, and the single record when there is exactly one
  std::vector<int32_t> k[2];
  std::vector<uint64_t> one[2];
  std::vector<uint8_t> drop[2];  // records left out: dominated_records
  const bool big = n >= (1 << 16);
  parallel_mates(big, [&](int mt) {
    k[mt].assign(n, 0);
    one[mt].assign(n, kNoRec8);
    const ShortMate& m = *mates[mt];
    if (fold) dominated_records(m, drop[mt]); else drop[mt].assign(m.pool.size(), 0);
    out.dropped_records[mt] = 0;
    for (uint8_t d : drop[mt]) out.dropped_records[mt] += d;
    for (size_t wid = 0; wid < m.wins.size(); wid++) {
      const Window& w = m.wins[wid];
      if (!w.active) continue;
      for (int64_t q = w.first; q < w.first + w.count; q++) {
        if (drop[mt][q]) continue;
        const gaml_aligment& r = m.pool[q];
        if (k[mt][r.read_id]++ == 0)
          one[mt][r.read_id] = rec8_fits((int32_t)wid, r.position, r.edit_dist) ? rec8_pack((int32_t)wid, r.position, r.edit_dist, r.orientation)
                                                                                 : kNoRec8 - 1;  // does not fit: not class 0
      }
    }
  });
  t_stage[n_stage++] = now_ms();
  // length combos
  std::unordered_map<uint32_t, int32_t> combo_id;
  out.len_combo.clear();
  auto combo_of = [&](int64_t i) { return (uint32_t)a.lens[i] | ((uint32_t)b.lens[i] << 16); };
  std::vector<int32_t> lc(n, -1);
  uint32_t last_c = 0; int32_t last_id = -2;
  for (int64_t i = 0; i < n; i++) {
    uint32_t c = combo_of(i);
    if (last_id != -2 && c == last_c) { lc[i] = last_id; continue; }  // reads of one library mostly share their lengths
    auto it = combo_id.find(c);
    if (it == combo_id.end()) {
      if (out.len_combo.size() >= 256) continue;  // lc stays -1: not class 0
      it = combo_id.emplace(c, (int32_t)out.len_combo.size()).first;
      out.len_combo.push_back(c);
    }
    lc[i] = it->second;
    last_c = c; last_id = it->second;
  }
  // the memo index of a class-0 pair whose two records sit in the same window, or -1 (PairTables::static_idx): same
  // orientation rule and insert distance as the scorers apply per call (graph.cc:1864-1876), on window positions -- the
  // two alignments get the same shift wherever the window occurs
  const int memo_codes = (int)std::min<size_t>(out.len_combo.size(), kMemoCodes);
  const bool memo_fits = ins_n > 0 && (size_t)memo_codes * 49 * (size_t)ins_n <= kMemoMaxEntries;
  auto static_idx = [&](int64_t i) -> int32_t {  // (only asked for pairs that fit the compact class)
    if (!memo_fits) return -1;
    const uint64_t r1 = one[0][i], r2 = one[1][i];
    if (r1 == kNoRec8 || r2 == kNoRec8) return kStaticZero;  // a mate without alignment: the pair scores nothing, in every path set
    if (lc[i] >= memo_codes) return -1;
    const int32_t w1 = (int32_t)(r1 & 0xffffff), w2 = (int32_t)(r2 & 0xffffff);
    if (a.wins[w1].peer != w2) return -1;
    const int32_t p1 = (int32_t)((r1 >> 24) & 0xfffffff), p2 = (int32_t)((r2 >> 24) & 0xfffffff);
    const int32_t e1 = (int32_t)((r1 >> 52) & 63), e2 = (int32_t)((r2 >> 52) & 63);
    const int32_t or1 = (int32_t)((r1 >> 58) & 1), or2 = (int32_t)((r2 >> 58) & 1);
    if (or1 == or2 || e1 >= 7 || e2 >= 7) return -1;
    const bool fwd = p1 < p2;
    if (or1 != (fwd ? 0 : 1)) return -1;
    const int32_t dist = fwd ? p2 - p1 + b.lens[i] : p1 - p2 + a.lens[i];
    if (dist < 0 || dist >= ins_n) return -1;
    return ((lc[i] * 7 + e1) * 7 + e2) * ins_n + dist;
  };
  // internal classes: 0 = class 0 with a static memo index, 1 = the rest of class 0, 2..4 = the reference classes 1..3
  auto cls = [&](int64_t i) {
    int m = std::max(k[0][i], k[1][i]);
    if (m <= 1 && lc[i] >= 0 && one[0][i] != kNoRec8 - 1 && one[1][i] != kNoRec8 - 1) return static_idx(i) != -1 ? 0 : 1;
    return m <= 2 ? 2 : m <= 4 ? 3 : 4;
  };
  // device order: by (class, window of mate 1, window of mate 2, read id) -- for a read with several records the window
  // of its first one. Lanes of a wave then look up the same few occurrence entries (a broadcast instead of a gather: the
  // two-record class, a tenth of the pairs, made half as many cache transactions again as the compact class while it was
  // ordered by read id). Two stable counting passes (least significant key first), O(pairs + windows).
  std::vector<uint8_t> cl(n);
  parallel_ranges(n, [&](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; i++) cl[i] = (uint8_t)cls(i); });
  {
    // A handful of pairs with 3-4 records per mate do not get a block range of their own: one lane resolving such a pair
    // (16 + 16 liveness tests, up to 16 terms, all in registers) takes ~6.5 us -- as long as the rest of the launch put
    // together at BASELINE config 3, where the class holds ONE pair. The wave-per-pair path spreads the same work over 64
    // lanes; it takes them while they are few.
    int64_t n3 = 0;
    for (int64_t i = 0; i < n; i++) n3 += cl[i] == 3;
    if (n3 > 0 && n3 <= kFoldClass2Below) for (int64_t i = 0; i < n; i++) if (cl[i] == 3) cl[i] = 4;
  }
  const uint32_t nw1 = (uint32_t)a.wins.size() + 2, nw2 = (uint32_t)b.wins.size() + 2;
  // (`one` holds a read's first record, or a marker when it has none / the record does not fit the 8-byte form)
  auto win_of = [&](int mt, int32_t i, uint32_t none) -> uint32_t { return one[mt][i] >= kNoRec8 - 1 ? none : (uint32_t)(one[mt][i] & 0xffffff); };
  auto key2 = [&](int32_t i) -> uint32_t { return win_of(1, i, nw2 - 1); };  // window of mate 2; "no record" sorts last
  auto key1 = [&](int32_t i) -> uint32_t { return (uint32_t)cl[i] * nw1 + win_of(0, i, nw1 - 1); };  // (class, window of mate 1)
  t_stage[n_stage++] = now_ms();
  std::vector<int32_t> order(n), tmp(n);
  // one stable counting pass on a few threads: thread t counts and later scatters the t-th contiguous part of the input
  // (offsets are prefix sums over (key, thread), so equal keys keep their input order)
  auto counting_pass = [&](uint32_t n_keys, auto key_of_pos, auto item_of_pos, std::vector<int32_t>& dst) {
    const int nt = n < (1 << 16) ? 1 : (int)std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<std::vector<int64_t>> cnt(nt, std::vector<int64_t>(n_keys, 0));
    auto part = [&](int t) { return std::pair<int64_t, int64_t>(n * t / nt, n * (t + 1) / nt); };
    auto run = [&](auto fn) {
      if (nt == 1) { fn(0); return; }
      std::vector<std::thread> pool;
      for (int t = 0; t < nt; t++) pool.emplace_back(fn, t);
      for (auto& th : pool) th.join();
    };
    run([&](int t) { auto [lo, hi] = part(t); for (int64_t i = lo; i < hi; i++) cnt[t][key_of_pos(i)]++; });
    int64_t at = 0;
    for (uint32_t key = 0; key < n_keys; key++)
      for (int t = 0; t < nt; t++) { const int64_t c = cnt[t][key]; cnt[t][key] = at; at += c; }
    run([&](int t) { auto [lo, hi] = part(t); for (int64_t i = lo; i < hi; i++) dst[cnt[t][key_of_pos(i)]++] = item_of_pos(i); });
  };
  counting_pass(nw2 + 1, [&](int64_t i) { return key2((int32_t)i); }, [&](int64_t i) { return (int32_t)i; }, tmp);
  counting_pass(5 * nw1, [&](int64_t i) { return key1(tmp[i]); }, [&](int64_t i) { return tmp[i]; }, order);
  t_stage[n_stage++] = now_ms();
  for (int c = 0; c < 4; c++) out.class_count[c] = 0;
  out.slot_of_read.assign(n, 0);
  out.read_of_slot.assign(n, 0);
  out.n0a = 0;
  for (int64_t i = 0; i < n; i++) { out.class_count[cl[i] == 0 ? 0 : cl[i] - 1]++; out.n0a += cl[i] == 0; }
  parallel_ranges(n, [&](int64_t lo, int64_t hi) {
    for (int64_t s = lo; s < hi; s++) { out.read_of_slot[s] = order[s]; out.slot_of_read[order[s]] = (int32_t)s; }
  });
  const int64_t n0 = out.class_count[0];
  for (int mt = 0; mt < 2; mt++) out.rec8[mt].resize(n0);
  out.len_code.resize(n0);
  out.static_idx.resize(out.n0a);
  parallel_ranges(n0, [&](int64_t lo, int64_t hi) {
    for (int64_t s = lo; s < hi; s++) {
      const int32_t r = order[s];
      out.rec8[0][s] = one[0][r];
      out.rec8[1][s] = one[1][r];
      out.len_code[s] = (uint8_t)lc[r];
      if (s < out.n0a) out.static_idx[s] = static_idx(r);
    }
  });
  out.len12.resize(n - n0);
  for (int64_t s = n0; s < n; s++) out.len12[s - n0] = combo_of(order[s]);
  t_stage[n_stage++] = now_ms();
  // 16-byte tables of the remaining slots, indexed slot - n0
  std::vector<int32_t> slot16(n, -1);
  for (int64_t s = n0; s < n; s++) slot16[order[s]] = (int32_t)(s - n0);
  parallel_mates(big, [&](int mt) {
    const ShortMate& m = *mates[mt];
    ReadMajor& rm = out.rm[mt];
    const int64_t n16 = n - n0;
    std::vector<int32_t> cnt(n16 + 1, 0);
    for (const Window& w : m.wins) if (w.active) for (int64_t q = w.first; q < w.first + w.count; q++) { if (drop[mt][q]) continue; int32_t t = slot16[m.pool[q].read_id]; if (t >= 0) cnt[t + 1]++; }
    std::vector<int64_t> start(n16 + 1, 0);
    int64_t extras = 0;
    for (int64_t i = 0; i < n16; i++) { start[i] = extras; extras += cnt[i + 1] > 1 ? cnt[i + 1] - 1 : 0; }
    rm.first.assign(n16, RecQuad{-1, 0, 0, 0});
    rm.extra.assign(extras, RecQuad{-1, 0, 0, 0});
    std::vector<int32_t> seen(n16, 0);
    for (size_t wid = 0; wid < m.wins.size(); wid++) {
      const Window& win = m.wins[wid];
      if (!win.active) continue;
      for (int64_t q = win.first; q < win.first + win.count; q++) {
        const gaml_aligment& r = m.pool[q];
        const int32_t at = slot16[r.read_id];
        if (at < 0 || drop[mt][q]) continue;
        RecQuad rq{(int32_t)wid, r.position, (r.edit_dist & 0xff) | ((r.orientation & 1) << 8), 0};
        int32_t sn = seen[at]++;
        if (sn == 0) { rq.flags |= (cnt[at + 1] - 1) << 9; rq.link = (int32_t)start[at]; rm.first[at] = rq; }
        else rm.extra[start[at] + sn - 1] = rq;
      }
    }
    rm.total_records = m.active_records;
    rm.built_generation = m.active_generation;
    // inline copies for the register paths
    const int64_t n1 = out.class_count[1], n2 = out.class_count[2];
    out.inl[mt].assign((size_t)(2 * n1 + 4 * n2), RecQuad{-1, 0, 0, 0});
    for (int64_t t = 0; t < n1 + n2; t++) {
      const RecQuad& f = rm.first[t];
      if (f.wid < 0) continue;
      const int cnt1 = 1 + (int)((uint32_t)f.flags >> 9);
      RecQuad* dst = t < n1 ? &out.inl[mt][2 * t] : &out.inl[mt][2 * n1 + 4 * (t - n1)];
      for (int q = 0; q < cnt1; q++) {
        RecQuad r = q == 0 ? f : rm.extra[f.link + q - 1];
        r.flags &= 0x1ff;
        dst[q] = r;
      }
    }
  });
  t_stage[n_stage++] = now_ms();
  if (trace_bpt)
    fprintf(stderr, "build_pair_tables: per-read counts %.1f ms, length codes + classes %.1f, two counting sorts %.1f, compact tables %.1f, 16-byte tables %.1f\n",
            t_stage[1] - t_stage[0], t_stage[2] - t_stage[1], t_stage[3] - t_stage[2], t_stage[4] - t_stage[3], t_stage[5] - t_stage[4]);
}

#endif  // GAML_HIP_DEV

static inline uint64_t occ8_pack(const OccQuad& q, bool general) {
  int32_t mp = q.min_pos < -32768 ? -32768 : q.min_pos;
  return (uint64_t)(uint32_t)q.shift | ((uint64_t)(uint16_t)(int16_t)mp << 32) | ((uint64_t)(q.path & 0x7fff) << 48) |
         ((uint64_t)(general ? 1 : 0) << 63);
}

// ---- OccImage: entries -------------------------------------------------------------------------------------
// cnt_[w] = occurrences of window w in the current path set. One occurrence that the 8-byte form can hold lives in
// the table entry itself; everything else (several occurrences, slot or threshold out of range) lives in gen_[w]
// and the entry only carries the "general" flag + the list number.
static inline bool occ_fits_direct(const OccQuad& q) { return q.path < 32767 && q.min_pos <= 32767; }
static inline OccQuad occ_from_direct(const Occ12& f) {
  return OccQuad{(int32_t)f.lo, (int32_t)(int16_t)(f.hi & 0xffff), (int32_t)((f.hi >> 16) & 0x7fff), f.rank};
}

void OccImage::grow(size_t n_windows) {
  if (occ12.size() >= n_windows) return;
  occ12.resize(n_windows, Occ12{~0u, ~0u, 0});
  cnt_.resize(n_windows, 0);
  list_of_.resize(n_windows, -1);
  stamp_.resize(n_windows, 0);
  mark_.resize(n_windows, 0);
}

void OccImage::set_direct(int32_t w, const OccQuad& q) {
  const uint64_t e = occ8_pack(q, false);
  occ12[w] = Occ12{(uint32_t)e, (uint32_t)(e >> 32), q.rank};
  mark(w);
}

void OccImage::add_path(size_t n_windows, const std::vector<Occ>& occ, int32_t slot) {
  grow(n_windows);
  for (const Occ& o : occ) {
    const OccQuad q{o.shift, o.min_pos, slot, o.rank};
    const int32_t w = o.wid;
    if (cnt_[w] == 0) {
      cnt_[w] = 1;
      if (stamp_[w] != serial_) { stamp_[w] = serial_; touched_.push_back(w); }
      if (occ_fits_direct(q)) { set_direct(w, q); continue; }
      gen_[w].push_back(q);
    } else {
      auto it = gen_.find(w);
      if (it == gen_.end()) it = gen_.emplace(w, std::vector<OccQuad>(1, occ_from_direct(occ12[w]))).first;  // the direct occupant moves to the list
      it->second.push_back(q);
      cnt_[w]++;
    }
    lists_dirty_ = true;
  }
}

void OccImage::remove_path(const std::vector<Occ>& occ, int32_t slot) {
  for (const Occ& o : occ) {
    const int32_t w = o.wid;
    if (w >= (int32_t)cnt_.size() || cnt_[w] == 0) continue;  // (cannot happen: every removal mirrors an add)
    auto it = gen_.find(w);
    if (it == gen_.end()) {  // the one direct occupant
      cnt_[w] = 0;
      occ12[w] = Occ12{~0u, ~0u, 0};
      mark(w);
      continue;
    }
    auto& lst = it->second;
    for (size_t k = 0; k < lst.size(); k++)
      if (lst[k].path == slot && lst[k].rank == o.rank && lst[k].shift == o.shift) { lst.erase(lst.begin() + k); break; }
    cnt_[w] = (int32_t)lst.size();
    lists_dirty_ = true;
    if (lst.empty()) { gen_.erase(it); occ12[w] = Occ12{~0u, ~0u, 0}; mark(w); }
    else if (lst.size() == 1 && occ_fits_direct(lst[0])) { const OccQuad q = lst[0]; gen_.erase(it); set_direct(w, q); }
  }
}

void OccImage::finalize(const std::vector<int32_t>& pos_of_slot) {
  if (!lists_dirty_) return;
  lists_dirty_ = false;
  lists_changed = true;
  multi_off.assign(1, 0);
  multi.clear();
  general_wids.clear();
  if (gen_.empty()) return;
  for (auto& e : gen_) general_wids.push_back(e.first);
  std::sort(general_wids.begin(), general_wids.end());  // list numbering: by window id (any fixed order does)
  for (int32_t w : general_wids) {
    auto& lst = gen_[w];
    // entries in visiting order: position of the path in the set, then the path-local rank
    std::sort(lst.begin(), lst.end(), [&](const OccQuad& x, const OccQuad& y) {
      const int32_t px = pos_of_slot[x.path], py = pos_of_slot[y.path];
      return px != py ? px < py : x.rank < y.rank;
    });
    const int32_t list = (int32_t)multi_off.size() - 1;
    multi.insert(multi.end(), lst.begin(), lst.end());
    multi_off.push_back((int32_t)multi.size());
    const uint64_t e = occ8_pack(OccQuad{0, 0, 0, 0}, true);  // only the flag is read (the occurrences are in the list); never all ones
    const Occ12 ent{(uint32_t)e, (uint32_t)(e >> 32), -(list + 1)};
    if (occ12[w].lo != ent.lo || occ12[w].hi != ent.hi || occ12[w].rank != ent.rank) { occ12[w] = ent; mark(w); }
  }
}

void OccImage::dump(std::vector<Occ>& out) const {
  out.clear();
  for (size_t w = 0; w < cnt_.size(); w++) {
    if (cnt_[w] == 0) continue;
    auto it = gen_.find((int32_t)w);
    if (it == gen_.end()) { const OccQuad q = occ_from_direct(occ12[w]); out.push_back(Occ{(int32_t)w, q.shift, q.min_pos, q.path, q.rank}); }
    else for (const OccQuad& q : it->second) out.push_back(Occ{(int32_t)w, q.shift, q.min_pos, q.path, q.rank});
  }
}

void OccImage::build(size_t n_windows, const PlanView& view, int mate) {
  grow(n_windows);
  // The previous set's entries are cleared AFTER the new ones are written, and only where the new set has none: path
  // sets of one assembly mostly name the same windows, whose entries would otherwise be written twice.
  stale_.swap(touched_);
  touched_.clear();
  gen_.clear();
  if (++serial_ == 0) { std::fill(stamp_.begin(), stamp_.end(), 0); serial_ = 1; }
  // ONE pass over the occurrences: a window's first occurrence is written as a direct entry straight away (prepacked
  // with the path's occurrence list: only the path slot is added here); a second occurrence moves the first one to
  // the window's list (the entry holds everything a list entry needs: a filter threshold clamped at -32768 filters like
  // the exact one, positions are >= 0).
  for (size_t slot = 0; slot < view.paths.size(); slot++) {
    const PathMemo& pm = *view.paths[slot];
    const std::vector<OccPre>& pre = pm.pre[mate];
    const bool slot_fits = slot < 32767;
    const uint32_t slot_bits = (uint32_t)(slot & 0x7fff) << 16;
    for (size_t k = 0; k < pre.size(); k++) {
      const OccPre& e = pre[k];
      const bool direct = slot_fits && e.wid >= 0;
      const int32_t w = e.wid >= 0 ? e.wid : ~e.wid;
      if (stamp_[w] != serial_) {  // first occurrence of this window in this path set
        stamp_[w] = serial_; cnt_[w] = 1; touched_.push_back(w);
        if (direct) occ12[w] = Occ12{e.lo, e.hi | slot_bits, e.rank};
        else { const Occ& o = pm.occ[mate][k]; gen_[w].push_back(OccQuad{o.shift, o.min_pos, (int32_t)slot, o.rank}); }
        continue;
      }
      const Occ& o = pm.occ[mate][k];
      auto it = gen_.find(w);
      if (it == gen_.end()) it = gen_.emplace(w, std::vector<OccQuad>(1, occ_from_direct(occ12[w]))).first;
      it->second.push_back(OccQuad{o.shift, o.min_pos, (int32_t)slot, o.rank});
      cnt_[w]++;
    }
  }
  for (int32_t w : stale_) if (stamp_[w] != serial_ && cnt_[w]) { occ12[w] = Occ12{~0u, ~0u, 0}; cnt_[w] = 0; }
  changed_all = true;
  lists_dirty_ = true;
  std::vector<int32_t> ident(view.paths.size());
  for (size_t k = 0; k < ident.size(); k++) ident[k] = (int32_t)k;  // slots are positions here
  finalize(ident);
}

void build_occ_table(size_t n_windows, const std::vector<Occ>& occs, OccTable& out) {
  out.direct.assign(n_windows, OccQuad{0, 0, -1, 0});
  out.multi_off.assign(1, 0);
  out.multi.clear();
  std::vector<int32_t> cnt(n_windows, 0);
  for (const Occ& o : occs) cnt[o.wid]++;
  // windows with several occurrences get a list (in rank order, which is the order of `occs`)
  std::vector<int32_t> slot(n_windows, -1);
  // an entry the compact 8-byte form cannot hold (path id or filter threshold out of range) is
  // stored as a one-element list, so that "rank < 0" means "general path" for every consumer
  for (const Occ& o : occs) if (cnt[o.wid] == 1 && (o.path >= 32767 || o.min_pos > 32767)) cnt[o.wid] = 1 << 30;
  for (const Occ& o : occs) {
    if (cnt[o.wid] == 1) { out.direct[o.wid] = OccQuad{o.shift, o.min_pos, o.path, o.rank}; continue; }
    if (cnt[o.wid] == (1 << 30)) cnt[o.wid] = 1;  // single occurrence kept as a list
    if (slot[o.wid] < 0) {
      slot[o.wid] = (int32_t)out.multi_off.size() - 1;
      out.multi_off.push_back(out.multi_off.back() + cnt[o.wid]);
      out.direct[o.wid] = OccQuad{0, 0, o.path, -(slot[o.wid] + 1)};
      cnt[o.wid] = -cnt[o.wid] - 2;  // < -1: list already opened
    }
  }
  out.multi.resize(out.multi_off.back());
  std::vector<int32_t> fill(out.multi_off.size(), 0);
  for (const Occ& o : occs) {
    if (slot[o.wid] < 0) continue;
    int32_t s = slot[o.wid];
    out.multi[out.multi_off[s] + fill[s]++] = OccQuad{o.shift, o.min_pos, o.path, o.rank};
  }
}

// ---------------------------------------------------------------------------------------
// FASTQ (graph.cc:1366-1415): 4 lines per record, name = first blank-separated token after
// the first character; a repeated name overwrites the earlier read of that id.
// ---------------------------------------------------------------------------------------
bool read_fastq(const std::string& file, std::string& bases, std::vector<int64_t>& offs, std::string* err,
                std::vector<std::string>* names) {
  std::ifstream f(file.c_str());
  if (!f.is_open()) { if (err) *err = "cannot open reads file " + file; return false; }
  std::unordered_map<std::string, int64_t> ids;
  std::vector<std::string> reads;
  std::string l, s;
  while (std::getline(f, l)) {
    std::string name = l.size() > 0 ? l.substr(1) : std::string();
    size_t cut = name.find_first_of(" \t");
    if (cut != std::string::npos) name.resize(cut);
    std::getline(f, s);
    auto it = ids.find(name);
    if (it == ids.end()) { ids.emplace(name, (int64_t)reads.size()); reads.push_back(s); if (names) names->push_back(name); }
    else reads[it->second] = s;
    std::getline(f, l);
    std::getline(f, l);
  }
  bases.clear();
  offs.assign(1, 0);
  for (auto& r : reads) { bases += r; offs.push_back((int64_t)bases.size()); }
  return true;
}

// ---------------------------------------------------------------------------------------
// SAM record (ParseAligment graph.cc:2945-3021, ParseCigar :3023-3038)
// ---------------------------------------------------------------------------------------
namespace {
// atoi over a character range: leading white space, optional sign, digits; 0 when there are none
inline int32_t atoi_range(const char* p, const char* e) {
  while (p < e && (*p == ' ' || (*p >= '\t' && *p <= '\r'))) p++;
  bool neg = false;
  if (p < e && (*p == '-' || *p == '+')) { neg = *p == '-'; p++; }
  int64_t v = 0;
  while (p < e && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); if (v > INT32_MAX) v = INT32_MAX; p++; }
  return (int32_t)(neg ? -v : v);
}
}  // namespace

bool parse_sam_record(const char* b, const char* e, int32_t total_len, SamRecord& a) {
  thread_local std::vector<std::pair<const char*, const char*>> col;
  col.clear();
  for (const char* p = b;;) {
    const char* t = (const char*)memchr(p, '\t', (size_t)(e - p));
    if (!t) { col.emplace_back(p, e); break; }
    col.emplace_back(p, t);
    p = t + 1;
  }
  if (col.size() < 10) return false;
  auto num = [&](size_t i, size_t skip = 0) {  // atoi of a column (leading digits, 0 when none)
    return atoi_range(col[i].first + std::min<size_t>(skip, (size_t)(col[i].second - col[i].first)), col[i].second);
  };
  a.name.clear(); a.cigar.clear();  // (the caller's record is reused: no reallocation per line)
  {  // name = QNAME up to its LAST '/' (empty when there is none; :2951-2957)
    const char* cut = col[0].first;
    for (const char* p = col[0].first; p < col[0].second; p++) if (*p == '/') cut = p;
    a.name.assign(col[0].first, cut);
  }
  a.flags = num(1);
  int32_t posstart = num(3);
  a.len = num(8);
  int32_t posend = posstart + a.len;
  a.slen = a.send = (int32_t)(col[9].second - col[9].first);
  a.sstart = 0;
  a.edit_dist = 100000;
  for (size_t i = 11; i < col.size(); i++) {  // "XS:i:<v>" etc.; the value starts at character 5
    if (col[i].second - col[i].first < 2) continue;
    const char t0 = col[i].first[0], t1 = col[i].first[1];
    const int32_t v = (col[i].second - col[i].first > 5) ? num(i, 5) : 0;
    if (t0 == 'X' && t1 == 'S') a.sstart = v - 1;
    if (t0 == 'X' && t1 == 'E') a.send = v - 1;
    if (t0 == 'X' && t1 == 'Q') a.slen = v;
    if (t0 == 'N' && t1 == 'M') a.edit_dist = v;
  }
  a.tstart = posstart;
  a.tend = posend;
  {  // CIGAR: a count is whatever atoi reads from the end of the previous M/I/D operation
    const char* from = col[5].first;
    for (const char* p = col[5].first; p < col[5].second; p++) {
      if (*p == 'M' || *p == 'I' || *p == 'D') {
        a.cigar.emplace_back(atoi_range(from, p), *p);
        from = p + 1;
      }
    }
  }
  if (a.flags & 16) {
    const int32_t span = posend - posstart;
    posstart = total_len - posend;
    posend = posstart + span;
    std::reverse(a.cigar.begin(), a.cigar.end());
  }
  if (a.send != a.slen) a.cigar.emplace_back(a.slen - a.send, 'I');
  if (a.sstart != 0) {
    const int32_t head = std::min(a.sstart, posstart), rest = a.sstart - head;
    a.cigar.insert(a.cigar.begin(), std::make_pair(head, 'I'));
    if (rest) a.cigar.insert(a.cigar.begin(), std::make_pair(rest, 'I'));
  }
  a.posstart = posstart;
  a.posend = posend;
  return true;
}

// ---------------------------------------------------------------------------------------
// DP band (graph.cc:2183-2235). The reference materialises the cell list and closes it twice
// (Uniquify :2153-2173); the set is a union of row intervals, so the same set is computed here
// from per-row first/last columns of the CIGAR path.
// ---------------------------------------------------------------------------------------
void pacbio_dp_band(const std::vector<std::pair<int32_t, char>>& cigar, DpBand& out) {
  // the CIGAR path: row r is entered at column enter[r] and left at column leave[r]
  std::vector<int32_t> enter(1, 0), leave;
  int64_t total = 0, lead = 0, trail = 0;
  bool seen_other = false;
  int32_t col = 0;
  for (const auto& op : cigar) {
    const int32_t n = std::max(0, op.first);
    if (n == 0) continue;
    total += n;
    if (op.second == 'I') {
      col += n;
      if (!seen_other) lead += n;
      trail += n;
    } else {
      seen_other = true;
      trail = 0;
      for (int32_t k = 0; k < n; k++) {
        leave.push_back(col);
        if (op.second == 'M') col++;
        enter.push_back(col);
      }
    }
  }
  leave.push_back(col);
  const int32_t row_f = (int32_t)enter.size() - 1, col_f = col;
  // clip boxes (GetCigarEnds :2138-2151, capped at 200 :2181-2182); none when the CIGAR is all 'I'
  const int32_t bl = seen_other ? (int32_t)std::min<int64_t>(lead, 200) : 0;
  const int32_t el = seen_other ? (int32_t)std::min<int64_t>(trail + 1, 200) : 0;
  const int32_t r_first = bl > 0 ? -bl : 0;
  const int32_t r_last = std::max(std::max(row_f, row_f + el - 1), bl > 0 ? 2 : 0);
  const int32_t n1 = r_last - r_first + 1;
  std::vector<int32_t> lo1(n1, INT32_MAX), hi1(n1, INT32_MIN);
  auto add = [&](int32_t r, int32_t a, int32_t b) {
    lo1[r - r_first] = std::min(lo1[r - r_first], a);
    hi1[r - r_first] = std::max(hi1[r - r_first], b);
  };
  add(0, 0, 0);
  if (bl > 0) for (int32_t r = -bl; r < 3; r++) add(r, 0, bl - 1);
  for (int32_t r = 0; r <= row_f; r++) add(r, enter[r], leave[r]);
  for (int32_t r = row_f; r < row_f + el; r++) add(r, col_f - el, col_f);
  // widen by 2 in both directions
  out.row0 = r_first - 2;
  const int32_t n2 = n1 + 4;
  out.lo.assign(n2, 0);
  out.hi.assign(n2, 0);
  out.max_width = 0;
  for (int32_t i = 0; i < n2; i++) {
    int32_t a = INT32_MAX, b = INT32_MIN;
    for (int32_t k = i - 4; k <= i; k++) {  // first-pass rows (i-2)-2 .. (i-2)+2 in lo1 indexing
      if (k < 0 || k >= n1) continue;
      a = std::min(a, lo1[k]);
      b = std::max(b, hi1[k]);
    }
    out.lo[i] = a - 2;
    out.hi[i] = b + 2;
    out.max_width = std::max(out.max_width, out.hi[i] - out.lo[i] + 1);
  }
}

void pacbio_dp_ops(const std::vector<std::pair<int32_t, char>>& cigar, std::vector<uint32_t>& ops, DpShape& out) {
  const size_t first = ops.size();
  int64_t rows = 0, cols = 0, lead = 0, trail = 0;
  bool seen_other = false;
  int64_t top[5] = {0, 0, 0, 0, 0};  // the five longest insertion runs, descending
  auto close_run = [&](int64_t run) {
    for (int i = 0; i < 5; i++) if (run > top[i]) { for (int k = 4; k > i; k--) top[k] = top[k - 1]; top[i] = run; break; }
  };
  int64_t run = 0;
  for (const auto& op : cigar) {
    const int64_t n = std::max(0, op.first);
    if (n == 0) continue;
    if (op.second == 'I') {
      if (ops.size() > first && (ops.back() & 3u) == 1u) ops.back() += (uint32_t)(n << 2);
      else ops.push_back((uint32_t)(n << 2) | 1u);
      cols += n; run += n; trail += n;
      if (!seen_other) lead += n;
    } else {
      if (run) { close_run(run); run = 0; }
      seen_other = true;
      trail = 0;
      ops.push_back((uint32_t)(n << 2) | (op.second == 'M' ? 0u : 2u));
      rows += n;
      if (op.second == 'M') cols += n;
    }
  }
  if (run) close_run(run);
  out.n_ops = (int32_t)(ops.size() - first);
  out.row_f = (int32_t)rows;
  out.col_f = (int32_t)cols;
  out.bl = seen_other ? (int32_t)std::min<int64_t>(lead, 200) : 0;      // GetCigarEnds graph.cc:2138-2151, cap :2181-2182
  out.el = seen_other ? (int32_t)std::min<int64_t>(trail + 1, 200) : 0;
  // a widened row spans at most five consecutive path rows (4 steps + their insertion runs) or a clip box
  const int64_t five = 5 + top[0] + top[1] + top[2] + top[3] + top[4];
  out.max_width = (int32_t)std::min<int64_t>(INT32_MAX, std::max<int64_t>(std::max<int64_t>(out.bl, out.el + 1), five) + 6);
}

}  // namespace gaml
