// gaml_oracle.cc -- TEST INFRASTRUCTURE ONLY. See gaml_oracle.hpp for scope and the
// parity-pinning statement. Reference citations are file:line under /root/reference.
#include "gaml_oracle.hpp"

#include <cstdlib>
#include <cstring>

namespace orc {

// ---------------------------------------------------------------------------
// Graph
// ---------------------------------------------------------------------------
void Graph::calc_normalize_map() {  // graph.h:249-266
  std::unordered_map<std::string, int> tiny;
  normalize_map.resize(seq.size());
  for (size_t i = 0; i < seq.size(); i++) normalize_map[i] = (int)i;
  for (size_t i = 0; i < seq.size(); i++) {
    if (seq[i].size() > 3) continue;
    auto it = tiny.find(seq[i]);
    if (it != tiny.end()) normalize_map[i] = it->second;
    else tiny[seq[i]] = (int)i;
  }
}

static int velvet_id_to_node(int x) {  // ConvertNodeId graph.h:48-53
  return x > 0 ? 2 * (x - 1) : 2 * (-x - 1) + 1;
}

bool load_lastgraph(const std::string& file, Graph& g, int* n_arcs) {  // graph.cc:52-106
  std::ifstream f(file.c_str());
  if (!f.is_open()) return false;
  std::string line;
  std::getline(f, line);
  int n = atoi(line.substr(0, line.find('\t')).c_str());  // first tab-separated field (:63-65)
  g.seq.assign(2 * (size_t)n, std::string());
  for (int i = 0; i < n; i++) {
    std::getline(f, line);              // NODE header line, skipped (:70)
    std::getline(f, g.seq[2 * i]);      // forward sequence (:77)
    std::getline(f, g.seq[2 * i + 1]);  // twin sequence (:78)
  }
  int arcs = 0;
  while (std::getline(f, line)) {
    if (line.compare(0, 3, "ARC") == 0) {  // :85-94, endpoints validated only
      size_t a = line.find('\t'), b = line.find('\t', a + 1);
      int s = velvet_id_to_node(atoi(line.substr(a + 1, b - a - 1).c_str()));
      int d = velvet_id_to_node(atoi(line.substr(b + 1).c_str()));
      if (s < 0 || d < 0 || s >= 2 * n || d >= 2 * n) return false;
      arcs++;
    }
  }
  if (n_arcs) *n_arcs = arcs;
  g.calc_normalize_map();
  return true;
}

// ---------------------------------------------------------------------------
// Max-hash index
// ---------------------------------------------------------------------------
bool MaxHashIndex::acgt_only(const std::string& s) {
  for (char c : s) if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return false;
  return true;
}

uint64_t MaxHashIndex::max_hash(const std::string& s) {  // graph.cc:1254-1269
  uint64_t cur = 0, best = 0;
  for (int i = 0; i < kSeedLen; i++) { cur <<= 2; cur += code(s[i]); }
  best = std::max(best, mix(cur));
  for (int i = kSeedLen; i < (int)s.size(); i++) {
    cur <<= 2;
    cur &= (1ll << (2 * kSeedLen)) - 1;
    cur += code(s[i]);
    best = std::max(best, mix(cur));
  }
  return best;
}

void MaxHashIndex::add_read(const std::string& s, int id) {  // graph.cc:1280-1287
  if (!acgt_only(s)) return;
  buckets[max_hash(s)].push_back(id);
  read_len = (int)s.size();
}

// Sliding maximum of the mixed 15-mer code over every read_len-wide span of s
// (graph.cc:1289-1323). Emits (hash, index of the 15-mer's LAST base) whenever the
// span maximum differs from the previously emitted one.
void MaxHashIndex::window_hashes(const std::string& s, std::vector<std::pair<uint64_t, int>>& out) const {
  std::deque<std::pair<uint64_t, int>> dq;
  if ((int)s.size() < kSeedLen) return;
  uint64_t cur = 0;
  for (int i = 0; i < kSeedLen; i++) { cur <<= 2; cur += code(s[i]); }
  dq.push_back(std::make_pair(mix(cur), kSeedLen - 1));
  uint64_t last = 0;
  for (int i = kSeedLen; i < (int)s.size(); i++) {
    while (!dq.empty() && dq.front().second < i - read_len + kSeedLen) dq.pop_front();
    cur <<= 2;
    cur &= (1ll << (2 * kSeedLen)) - 1;
    cur += code(s[i]);
    uint64_t h = mix(cur);
    while (!dq.empty() && dq.back().first < h) dq.pop_back();
    dq.push_back(std::make_pair(h, i));
    if (i >= read_len - 1) {
      uint64_t top = dq.front().first;
      if (i == read_len - 1 || top != last) {
        out.push_back(std::make_pair(top, dq.front().second));
        last = top;
      }
    }
  }
}

void MaxHashIndex::candidates(const std::string& s, std::unordered_map<int, std::vector<int>>& out) const {
  std::vector<std::pair<uint64_t, int>> fwd, rev;  // graph.cc:1325-1348
  window_hashes(s, fwd);
  for (auto& e : fwd) {
    auto it = buckets.find(e.first);
    if (it == buckets.end()) continue;
    for (int id : it->second) out[id].push_back(e.second);
  }
  std::string rc = revcomp(s);
  window_hashes(rc, rev);
  for (auto& e : rev) {
    auto it = buckets.find(e.first);
    if (it == buckets.end()) continue;
    for (int id : it->second) out[id].push_back(-e.second);
  }
}

// ---------------------------------------------------------------------------
// Seed extension (ProcessHit graph.cc:753-837 with push helpers 730-750)
// ---------------------------------------------------------------------------
namespace {
struct Bfs {
  // state = (errors, window index, read index); `seen` plays the role of the reference's
  // function-static `visited` table keyed by (read index + 1, window index - seed diagonal + 20).
  // The reference sizes that table from the first read it ever sees (:757) and never
  // clears it (it stamps an iteration counter); a fresh set per call is equivalent as long
  // as indices stay in range, which holds for uniform read lengths (SURVEY 8c hazards).
  std::deque<std::pair<int, std::pair<int, int>>> q;
  std::set<std::pair<int, int>> seen;
  int win_pos, read_pos;
  bool mark(int w, int r) {
    int diag = w - win_pos + read_pos + 20;
    return seen.insert(std::make_pair(r + 1, diag)).second;
  }
  void back(int d, int w, int r) { if (mark(w, r)) q.push_back(std::make_pair(d, std::make_pair(w, r))); }
  void front(int d, int w, int r) { if (mark(w, r)) q.push_front(std::make_pair(d, std::make_pair(w, r))); }
};
}  // namespace

HitResult extend_hit(int win_pos, int read_pos, const std::string& read, const std::string& win) {
  const int limit = 3;  // error_limit :759
  const int R = (int)read.size(), W = (int)win.size();
  Bfs b; b.win_pos = win_pos; b.read_pos = read_pos;
  // forward from the base after the seed (:761-793). NOTE the first state is pushed
  // without being marked visited (:762), exactly as here.
  int fwd = -1, end_pos = -1;
  b.q.push_back(std::make_pair(0, std::make_pair(win_pos + kSeedLen, read_pos + kSeedLen)));
  while (!b.q.empty()) {
    auto x = b.q.front(); b.q.pop_front();
    int d = x.first, w = x.second.first, r = x.second.second;
    if (d > limit) { b.q.clear(); break; }
    if (r == R) { fwd = d; end_pos = w - 1; b.q.clear(); break; }
    // win[W] reads the string terminator in the reference (std::string operator[] at size()).
    char wc = w < W ? win[w] : '\0';
    if (wc == read[r]) {
      if (w + 1 < W || r + 1 == R) b.front(d, w + 1, r + 1);
    } else {
      if (w + 1 < W) { b.back(d + 1, w + 1, r + 1); b.back(d + 1, w + 1, r); }
      b.back(d + 1, w, r + 1);
    }
  }
  if (fwd == -1) return HitResult{-1, -1, -1};
  // backward from the base before the seed (:794-835)
  int bwd = -1, begin_pos = -1;
  if (win_pos == 0) {
    if (read_pos < 6) bwd = read_pos;  // :797-798 (begin_pos stays -1)
  } else {
    b.q.push_back(std::make_pair(0, std::make_pair(win_pos - 1, read_pos - 1)));
    while (!b.q.empty()) {
      auto x = b.q.front(); b.q.pop_front();
      int d = x.first, w = x.second.first, r = x.second.second;
      if (d > limit) { b.q.clear(); break; }
      if (r == -1) { bwd = d; begin_pos = w + 1; b.q.clear(); break; }
      assert(w >= 0 && r >= 0 && w < W && r < R);
      if (win[w] == read[r]) {
        if (w - 1 >= 0 || r - 1 == -1) b.front(d, w - 1, r - 1);
      } else {
        if (w - 1 >= 0) { b.back(d + 1, w - 1, r - 1); b.back(d + 1, w - 1, r); }
        b.back(d + 1, w, r - 1);
      }
    }
  }
  if (bwd == -1) return HitResult{-1, -1, -1};
  return HitResult{bwd + fwd, begin_pos, end_pos};
}

// ---------------------------------------------------------------------------
// ShortReadSet
// ---------------------------------------------------------------------------
bool ShortReadSet::load_fastq(const std::string& file) {  // graph.cc:1366-1415
  std::ifstream f(file.c_str());
  if (!f.is_open()) return false;
  std::unordered_map<std::string, int> ids;  // read_map_: id by first appearance of the name
  std::string l, s;
  while (std::getline(f, l)) {
    std::string name = l.substr(1);
    size_t cut = name.find_first_of(" \t");
    if (cut != std::string::npos) name = name.substr(0, cut);
    std::getline(f, s);
    auto it = ids.find(name);
    int id;
    if (it == ids.end()) { id = (int)reads.size(); ids[name] = id; reads.push_back(std::string()); }
    else id = it->second;
    reads[id] = s;
    std::getline(f, l);
    std::getline(f, l);
  }
  finalize();
  return true;
}

void ShortReadSet::set_reads(const std::vector<std::string>& r) { reads = r; finalize(); }

void ShortReadSet::finalize() {
  lens.resize(reads.size());
  max_len = 0;
  for (size_t i = 0; i < reads.size(); i++) { lens[i] = (int)reads[i].size(); max_len = std::max(max_len, lens[i]); }
  match_pow.resize(max_len + 7);     // CalcMaxReadLen :1443-1454
  mismatch_pow.resize(max_len + 7);
  for (size_t i = 0; i < match_pow.size(); i++) {
    match_pow[i] = std::pow(match_p, (double)i);
    mismatch_pow[i] = std::pow(mismatch_p, (double)i);
  }
  index = MaxHashIndex();
  for (size_t i = 0; i < reads.size(); i++) index.add_read(reads[i], (int)i);
}

std::string ShortReadSet::window_string(const Graph& g, const std::vector<int>& w, int* offset) const {
  std::string s;  // graph.cc:846-857
  *offset = 0;
  for (size_t i = 0; i < w.size(); i++) {
    const std::string& ns = g.seq[w[i]];
    if (i == 0 && w.size() > 1 && (int)ns.size() > kWindowTail) {
      *offset = (int)ns.size() - kWindowTail;
      s += ns.substr(*offset);
    } else if (i > 0 && (int)ns.size() > kWindowTail && i + 1 == w.size()) {
      s += ns.substr(0, kWindowTail);
    } else {
      s += ns;
    }
  }
  return s;
}

void ShortReadSet::align_window(const Graph& g, const std::vector<int>& w) {  // graph.cc:839-899
  std::set<Rec> found;
  int offset = 0;
  std::string ws = window_string(g, w, &offset);
  std::unordered_map<int, std::vector<int>> cands;
  index.candidates(ws, cands);
  for (auto& c : cands) {
    for (int hit : c.second) {
      int win_pos;
      std::string rs;
      if (hit > 0) { win_pos = hit - kSeedLen + 1; rs = reads[c.first]; }
      else { win_pos = (int)ws.size() - (-hit + 1); rs = revcomp(reads[c.first]); }
      int read_pos = -1;
      for (int i = 0; i + kSeedLen - 1 < (int)rs.size(); i++) {
        if (rs.compare(i, kSeedLen, ws, win_pos, kSeedLen) == 0) { read_pos = i; break; }
      }
      assert(read_pos != -1);
      HitResult h = extend_hit(win_pos, read_pos, rs, ws);
      if (h.errs != -1) found.insert(Rec{h.begin + 1 + offset, h.errs, c.first, hit > 0 ? 0 : 1});
    }
  }
  std::vector<Rec>& dst = cache[w];
  for (auto& r : found) dst.push_back(r);
  windows_aligned++;
}

void ShortReadSet::align_windows(const Graph& g, const std::vector<std::vector<int>>& ws) {
  if (ws.empty()) return;                      // graph.cc:911-922
  for (auto& w : ws) cache[w] = std::vector<Rec>();  // reset, even if already cached (:914-916)
  for (auto& w : ws) align_window(g, w);
}

void ShortReadSet::precompute_for_paths(const Graph& g, const std::vector<std::vector<int>>& paths) {
  std::unordered_set<std::vector<int>, WalkHash> todo;  // graph.cc:447-493
  int last_end = -1;                                    // NOT reset per path (:449)
  for (auto& p : paths) {
    for (int i = 0; i < (int)p.size(); i++) {
      if (p[i] < 0) continue;
      int tail = 0, cur_end = i;
      std::vector<int> w(1, p[i]);
      for (int j = i + 1; j < (int)p.size(); j++) {
        if (p[j] < 0) break;
        tail += g.len(p[j]);
        w.push_back(p[j]);
        cur_end = j;
        if (tail > kWindowTail) break;
      }
      if (cache.count(w) == 0 && (last_end != cur_end || (w.size() == 1 && g.len(w[0]) > 150))) {
        todo.insert(w);
        todo.insert(invert_walk(w));
      }
      if (g.len(p[i]) > kWindowTail) {
        std::vector<int> one(1, p[i]);
        if (cache.count(one) == 0) { todo.insert(one); todo.insert(std::vector<int>(1, p[i] ^ 1)); }
      }
      last_end = cur_end;
    }
  }
  if (!todo.empty()) align_windows(g, std::vector<std::vector<int>>(todo.begin(), todo.end()));
}

void ShortReadSet::missing_windows_of_contig(const Graph& g, const std::vector<int>& ctg,
                                             std::unordered_set<std::vector<int>, WalkHash>& out) const {
  int last_end = -1;  // graph.cc:495-533
  for (int i = 0; i < (int)ctg.size(); i++) {
    if (ctg[i] < 0) continue;
    int tail = 0, cur_end = i;
    std::vector<int> w(1, ctg[i]);
    for (int j = i + 1; j < (int)ctg.size(); j++) {
      if (ctg[j] < 0) break;
      tail += g.len(ctg[j]);
      w.push_back(ctg[j]);
      cur_end = j;
      if (tail > kWindowTail) break;
    }
    if (cur_end != last_end && cache.count(w) == 0) out.insert(w);
    last_end = cur_end;
  }
}

void ShortReadSet::positions_only_path(const Graph& g, const std::vector<int>& ctg, int st,
                                       std::unordered_map<int, std::vector<Rec>>& acc) {
  std::unordered_set<std::vector<int>, WalkHash> todo;  // graph.cc:535-598
  missing_windows_of_contig(g, ctg, todo);
  if (!todo.empty()) align_windows(g, std::vector<std::vector<int>>(todo.begin(), todo.end()));
  int cur_pos = st, max_pos = 0;
  for (int i = 0; i < (int)ctg.size(); i++) {
    int node_max = 0, tail = 0;
    std::vector<int> w(1, ctg[i]);
    for (int j = i + 1; j < (int)ctg.size(); j++) {
      tail += g.len(ctg[j]);
      w.push_back(ctg[j]);
      if (tail > kWindowTail) break;
    }
    std::vector<std::vector<int>> look(1, w);
    if (g.len(w[0]) > kWindowTail) look.push_back(std::vector<int>(1, w[0]));
    for (auto& key : look) {
      auto it = cache.find(key);
      if (it == cache.end()) continue;  // miss -> nothing (:571-573)
      for (Rec al : it->second) {
        al.pos += cur_pos;
        if (al.pos < max_pos - 5) continue;
        node_max = std::max(al.pos, node_max);
        std::vector<Rec>& mine = acc[al.read];
        bool replaced = false;
        for (auto& old : mine) if (old.pos == al.pos) { old = al; replaced = true; break; }
        if (!replaced) mine.push_back(al);
      }
    }
    cur_pos += g.len(ctg[i]);
    max_pos = std::max(max_pos, node_max);
  }
}

void ShortReadSet::clear_positions() {  // graph.cc:316-321
  positions.resize(reads.size());
  for (auto& v : positions) v.clear();
  several.assign(reads.size(), 0);
}

void ShortReadSet::add_positions(const Graph& g, const std::vector<int>& ctg, int& total_len, int st) {
  std::unordered_set<std::vector<int>, WalkHash> todo;  // graph.cc:600-649
  missing_windows_of_contig(g, ctg, todo);
  if (!todo.empty()) align_windows(g, std::vector<std::vector<int>>(todo.begin(), todo.end()));
  int cur_pos = st;
  for (int i = 0; i < (int)ctg.size(); i++) {
    total_len += g.len(ctg[i]);
    int tail = 0;
    std::vector<int> w(1, ctg[i]);
    for (int j = i + 1; j < (int)ctg.size(); j++) {
      tail += g.len(ctg[j]);
      w.push_back(ctg[j]);
      if (tail > kWindowTail) break;
    }
    auto it = cache.find(w);
    if (it != cache.end()) {  // miss: the reference returns a dangling temporary (:1473-1480);
      for (const Rec& al : it->second) {  // specified behaviour = no alignments
        auto& mine = positions[al.read];
        bool replaced = false;
        for (auto& old : mine)
          if (old.first == al.pos + cur_pos) {
            if (old.second != std::make_pair(al.edit, al.orient)) several[al.read] = 1;  // (bookkeeping for the cross-check below, not in the reference)
            old.second = std::make_pair(al.edit, al.orient); replaced = true; break;
          }
        if (!replaced) { if (!mine.empty()) several[al.read] = 1; mine.push_back(std::make_pair(al.pos + cur_pos, std::make_pair(al.edit, al.orient))); }
      }
    }
    cur_pos += g.len(ctg[i]);
  }
}

// ---------------------------------------------------------------------------
// Paired-end scoring
// ---------------------------------------------------------------------------
double insert_prob(double len, double mean, double sd) {  // graph.cc:1593-1598
  double z = (len - mean) / sd;
  double e = std::exp(-z * z / 2.0);
  double c = std::sqrt(2 * M_PI) * sd;
  return e / c;
}

void path_changes(const std::vector<std::vector<int>>& now, const std::vector<std::vector<int>>& old,
                  std::vector<std::vector<int>>& erased, std::vector<std::vector<int>>& added) {
  std::unordered_multiset<std::vector<int>, WalkHash> pool(old.begin(), old.end());  // :1745-1764
  for (auto& p : now) {
    auto it = pool.find(p);
    if (it == pool.end()) added.push_back(p);
    else pool.erase(it);
  }
  erased.insert(erased.end(), pool.begin(), pool.end());
}

int walk_len(const Graph& g, const std::vector<int>& w) {
  int t = 0;
  for (int e : w) t += e < 0 ? -e : g.len(e);
  return t;
}
int total_walk_len(const Graph& g, const std::vector<std::vector<int>>& p) {
  int t = 0;
  for (auto& w : p) t += walk_len(g, w);
  return t;
}

// split a walk at gap entries into contigs + gap lengths (:1808-1824)
static void split_at_gaps(const std::vector<int>& path, std::vector<std::vector<int>>& ctgs, std::vector<int>& gaps) {
  int last = 0;
  for (int i = 0; i < (int)path.size(); i++) {
    if (path[i] < 0) {
      gaps.push_back(-path[i]);
      ctgs.push_back(std::vector<int>(path.begin() + last, path.begin() + i));
      last = i + 1;
    }
  }
  ctgs.push_back(std::vector<int>(path.begin() + last, path.end()));
}

void score_path_paired(const Graph& g, const std::vector<int>& path, ShortReadSet& r1, ShortReadSet& r2,
                       double ins_mean, double ins_sd, double cov_move, bool all_to_cov,
                       double floor_per_base, double floor_start, PathScore& out) {
  std::vector<double> ins_tab((int)(ins_mean + 5 * ins_sd));  // :1801-1804
  for (size_t i = 0; i < ins_tab.size(); i++) ins_tab[i] = insert_prob((double)i, ins_mean, ins_sd);
  std::vector<std::pair<int, int>> events;
  std::vector<std::vector<int>> ctgs;
  std::vector<int> gaps;
  split_at_gaps(path, ctgs, gaps);
  events.push_back(std::make_pair(0, 1));
  std::unordered_map<int, std::vector<Rec>> pos1, pos2;
  int cur_len = 0;
  for (size_t i = 0; i < ctgs.size(); i++) {  // :1830-1844
    if (i > 0) { cur_len += gaps[i - 1]; events.push_back(std::make_pair(cur_len, 1)); }
    r1.positions_only_path(g, ctgs[i], cur_len, pos1);
    r2.positions_only_path(g, ctgs[i], cur_len, pos2);
    cur_len += walk_len(g, ctgs[i]);
  }
  for (auto& e : pos1) {  // :1853-1892
    auto other = pos2.find(e.first);
    if (other == pos2.end()) continue;
    int id = e.first;
    // quirk kept: the coverage threshold uses read set 2's length twice (:1855-1857)
    double thr = std::exp(floor_start + floor_per_base * (r2.lens[id] + r2.lens[id]));
    for (auto& x : e.second) {
      double p1 = r1.base_prob(id, x.edit);
      for (auto& y : other->second) {
        double p2 = r2.base_prob(id, y.edit);
        if (x.orient == y.orient) continue;
        int dist;
        if (x.pos < y.pos) {
          if (x.orient != 0 || y.orient != 1) continue;
          dist = y.pos - x.pos + r2.lens[id];
        } else {
          if (x.orient != 1 || y.orient != 0) continue;
          dist = x.pos - y.pos + r1.lens[id];
        }
        double ip = (size_t)dist < ins_tab.size() ? ins_tab[dist] : insert_prob(dist, ins_mean, ins_sd);
        if (p1 * p2 * ip > thr) {
          events.push_back(std::make_pair(std::max(x.pos, y.pos), 3));
          if (all_to_cov) events.push_back(std::make_pair(std::min(x.pos, y.pos), 3));
        }
        out.changes.push_back(std::make_pair(id, p1 * p2 * ip));
      }
    }
  }
  std::sort(events.begin(), events.end());  // :1893-1919
  int last_pos = 0, last_type = -1, last_begin = 0;
  for (auto& ev : events) {
    if (ev.second == 3) {
      if (ev.first - last_pos > cov_move && (last_type == 3 || last_type < 0) &&
          ev.first - last_begin > ins_mean + 5 * ins_sd)
        out.bad_bases += ev.first - last_pos;
    }
    if (ev.second == 1) last_begin = ev.first;
    last_pos = ev.first;
    last_type = ev.second;
  }
}

double total_prob_paired(const std::vector<double>& probs, int total_len, int& zero_reads,
                         double floor_per_base, double floor_start,
                         const ShortReadSet& r1, const ShortReadSet& r2) {
  double total = 0; int cnt = 0;  // graph.cc:1495-1516
  if (total_len == 0) total_len = 1;
  zero_reads = 0;
  for (size_t i = 0; i < probs.size(); i++) {
    double p = probs[i] / (2 * total_len);
    double thr = std::exp(floor_start + floor_per_base * (r1.lens[i] + r2.lens[i]));
    if (p < thr) { zero_reads++; p = thr; }
    total += std::log(p);
    cnt++;
  }
  return total / cnt;
}

double total_prob_single(const std::vector<double>& probs, int total_len, int& zero_reads,
                         double floor_per_base, double floor_start, const ShortReadSet& r) {
  double total = 0; int cnt = 0;  // graph.cc:1518-1537
  if (total_len == 0) total_len = 1;
  zero_reads = 0;
  for (size_t i = 0; i < probs.size(); i++) {
    double p = probs[i] / (2 * total_len);
    double thr = std::exp(floor_start + floor_per_base * (r.lens[i]));
    if (p < thr) { zero_reads++; p = thr; }
    total += std::log(p);
    cnt++;
  }
  return total / cnt;
}

double score_paired(const Graph& g, const std::vector<std::vector<int>>& paths, ShortReadSet& r1,
                    ShortReadSet& r2, double ins_mean, double ins_sd, int& zero_reads, int& total_len,
                    PairedState& st, double penalty, double cov_move, bool all_to_cov,
                    double floor_per_base, double floor_start) {
  std::vector<std::vector<int>> erased, added;  // graph.cc:1952-1989
  path_changes(paths, st.old_paths, erased, added);
  assert(r1.n() == r2.n());
  if (st.probs.empty()) st.probs.resize(r1.n());
  total_len = total_walk_len(g, paths);
  r1.precompute_for_paths(g, paths);
  r2.precompute_for_paths(g, paths);
  PathScore gone, fresh;
  for (auto& p : erased)
    score_path_paired(g, p, r1, r2, ins_mean, ins_sd, cov_move, all_to_cov, floor_per_base, floor_start, gone);
  for (auto& p : added)
    score_path_paired(g, p, r1, r2, ins_mean, ins_sd, cov_move, all_to_cov, floor_per_base, floor_start, fresh);
  st.bad_bases -= gone.bad_bases;                                 // :1936-1942
  for (auto& c : gone.changes) st.probs[c.first] -= c.second;
  st.bad_bases += fresh.bad_bases;                                // :1944-1950
  for (auto& c : fresh.changes) st.probs[c.first] += c.second;
  double tp = total_prob_paired(st.probs, total_len, zero_reads, floor_per_base, floor_start, r1, r2);
  st.old_paths = paths;
  return tp - st.bad_bases * penalty;
}

// ---------------------------------------------------------------------------
// The reference's OTHER paired scorer: the slow, non-incremental CalcScoreForPaths (graph.cc:1991-2127), unreachable
// from gaml (its call is commented out at prob_calculator.h:80-86, next to a printf("cmp ...") that compared it with
// the incremental scorer). Restated as a second, differently structured opinion on the paired value: positions are
// assembled the single-end way (AddPositions graph.cc:600-649: junction windows only, no position filter, absolute
// coordinates with the paths 1,000,000 apart), every position of mate 1 is paired with every position of mate 2
// whatever path they are on (graph.cc:2053-2088), coverage events are sorted and swept once over all paths
// (:2089-2117). It agrees with score_paired where the two definitions coincide: every node at most kWindowTail long
// (no whole-node windows), no gaps, penalty 0, and no record that the incremental scorer's position filter drops
// without an earlier window holding it (tests/test_oracle_golden.py).
// ---------------------------------------------------------------------------
double score_paired_slow(const Graph& g, const std::vector<std::vector<int>>& paths, ShortReadSet& r1, ShortReadSet& r2,
                         double ins_mean, double ins_sd, int& zero_reads, int& total_len, double penalty, double cov_move,
                         bool all_to_cov, double floor_per_base, double floor_start, std::vector<double>* probs_out,
                         int* bad_bases_out, std::vector<uint8_t>* several_out) {
  int tl1 = 0, tl2 = 0, st = 0;
  std::vector<double> probs(r1.n());
  r1.clear_positions();
  r2.clear_positions();
  r1.precompute_for_paths(g, paths);
  r2.precompute_for_paths(g, paths);
  std::vector<std::pair<int, int>> events;  // (position, type): 3 a well-aligned pair's end, 1 a contig start
  for (auto& path : paths) {
    std::vector<std::vector<int>> ctgs;
    std::vector<int> gaps;
    split_at_gaps(path, ctgs, gaps);
    events.push_back(std::make_pair(st + tl1, 1));
    for (size_t i = 0; i < ctgs.size(); i++) {
      if (i > 0) { tl1 += gaps[i - 1]; tl2 += gaps[i - 1]; events.push_back(std::make_pair(st + tl1, 1)); }
      const int at1 = st + tl1, at2 = st + tl2;  // arguments evaluated before the callee advances total_len (:2038-2039)
      r1.add_positions(g, ctgs[i], tl1, at1);
      r2.add_positions(g, ctgs[i], tl2, at2);
    }
    st += 1000000;
  }
  std::vector<double> ins_tab((size_t)(int)(ins_mean + 5 * ins_sd));
  for (size_t d = 0; d < ins_tab.size(); d++) ins_tab[d] = insert_prob((double)d, ins_mean, ins_sd);
  for (int i = 0; i < r1.n(); i++) {
    const double threshold = std::exp(floor_start + floor_per_base * (r1.lens[i] + r2.lens[i]));  // (:2051: both lengths here)
    for (auto& x : r1.positions[i]) {
      const double p1 = r1.base_prob(i, x.second.first);
      for (auto& y : r2.positions[i]) {
        const double p2 = r2.base_prob(i, y.second.first);
        if (x.second.second == y.second.second) continue;
        int dist;
        if (x.first < y.first) {
          if (x.second.second != 0 || y.second.second != 1) continue;
          dist = y.first - x.first + r2.lens[i];
        } else {
          if (x.second.second != 1 || y.second.second != 0) continue;
          dist = x.first - y.first + r1.lens[i];
        }
        const double ip = (size_t)dist < ins_tab.size() ? ins_tab[dist] : insert_prob((double)dist, ins_mean, ins_sd);
        if (p1 * p2 * ip > threshold) {
          events.push_back(std::make_pair(std::max(x.first, y.first), 3));
          if (all_to_cov) events.push_back(std::make_pair(std::min(x.first, y.first), 3));
        }
        probs[i] += p1 * p2 * ip;
      }
    }
  }
  std::sort(events.begin(), events.end());
  int last_pos = 0, last_type = -1, last_begin = 0, bad_bases = 0;
  for (auto& ev : events) {
    if (ev.second == 3 && ev.first - last_pos > cov_move && (last_type == 3 || last_type < 0) && ev.first - last_begin > ins_mean + 5 * ins_sd)
      bad_bases += ev.first - last_pos;
    if (ev.second == 1) last_begin = ev.first;
    last_pos = ev.first;
    last_type = ev.second;
  }
  const double tp = total_prob_paired(probs, tl1, zero_reads, floor_per_base, floor_start, r1, r2);
  total_len = tl1;
  if (probs_out) *probs_out = probs;
  if (bad_bases_out) *bad_bases_out = bad_bases;
  if (several_out) {  // reads with more than one distinct alignment on a mate: where the two paired scorers may part
    several_out->assign(r1.n(), 0);
    for (int i = 0; i < r1.n(); i++) (*several_out)[i] = r1.several[i] | r2.several[i];
  }
  return tp - bad_bases * penalty;
}

// ---------------------------------------------------------------------------
// Single-end scoring (graph.cc:1650-1743)
// ---------------------------------------------------------------------------
double score_single(const Graph& g, const std::vector<std::vector<int>>& paths, ShortReadSet& r,
                    int& zero_reads, int& total_len, double penalty, double cov_move,
                    double floor_per_base, double floor_start, std::vector<double>* probs_out,
                    int* bad_bases_out) {
  int tl = 0, st = 0;
  std::vector<double> probs(r.n());
  r.clear_positions();
  std::vector<std::pair<int, int>> events;
  for (auto& path : paths) {
    std::vector<std::vector<int>> ctgs;
    std::vector<int> gaps;
    split_at_gaps(path, ctgs, gaps);
    events.push_back(std::make_pair(st + tl, 1));
    for (size_t i = 0; i < ctgs.size(); i++) {
      if (i > 0) { tl += gaps[i - 1]; events.push_back(std::make_pair(st + tl, 1)); }
      int at = st + tl;  // argument evaluated before the callee advances total_len (:1683)
      r.add_positions(g, ctgs[i], tl, at);
    }
    st += 1000000;
  }
  for (int i = 0; i < r.n(); i++) {
    for (auto& x : r.positions[i]) {
      double p = r.base_prob(i, x.second.first);
      if (p > kCovEventMinProb) events.push_back(std::make_pair(x.first, r.lens[i]));
      probs[i] += p;
    }
  }
  std::sort(events.begin(), events.end());
  int last_fin = -1, last_type = -1, bad_bases = 0;  // :1701-1733
  for (auto& ev : events) {
    if (ev.second >= 3) {
      // last_type only ever takes the values -1 and 1 (:1722-1732), so this never fires;
      // restated as written.
      if (ev.first > last_fin && last_type >= 3) bad_bases += ev.first - last_fin;
      last_fin = std::max(last_fin, (int)(ev.first + ev.second * cov_move));
    }
    if (ev.second == 1) last_type = ev.second;
    if (ev.second < -1) last_type = ev.second;
  }
  double tp = total_prob_single(probs, tl, zero_reads, floor_per_base, floor_start, r);
  total_len = tl;
  if (probs_out) *probs_out = probs;
  if (bad_bases_out) *bad_bases_out = bad_bases;
  return tp - bad_bases * penalty;
}

// ---------------------------------------------------------------------------
// PacBio scoring
// ---------------------------------------------------------------------------
void LongReadSet::read_probabilities(const Graph& g, const std::vector<int>& path, int& total_len,
                                     std::vector<std::vector<std::pair<std::pair<int, int>, LogD>>>& out) {
  // graph.cc:2410-2503. node start/end offsets inside the path string; gap = run of N.
  std::vector<int> ends, begins;
  int len = path[0] >= 0 ? g.len(path[0]) : 0;  // the reference dereferences nodes[path[0]] (:2412)
  ends.push_back(len);
  begins.push_back(0);
  for (size_t i = 1; i < path.size(); i++) {
    begins.push_back(len);
    len += path[i] < 0 ? -path[i] : g.len(path[i]);
    ends.push_back(len);
  }
  total_len = len;
  out.assign(lens.size(), std::vector<std::pair<std::pair<int, int>, LogD>>());
  for (size_t i = 0; i < path.size(); i++) {
    std::vector<int> sub;
    for (size_t j = i; j < path.size(); j++) {
      sub.push_back(path[j]);
      int sub_len = ends[j] - begins[i], first_len = ends[i] - begins[i];
      auto it = cache.find(sub);
      if (it == cache.end()) cache_misses++;  // reference: BLASR run (:2455-2478), out of scope
      else
        for (const LongRec& al : it->second)
          out[al.read].push_back(std::make_pair(std::make_pair(begins[i] + al.pos, begins[i] + al.pos_end), al.prob));
      if (sub_len - first_len > max_len) break;
    }
  }
}

double total_prob_pacbio(const std::vector<LogD>& probs, int total_len, const LongReadSet& r,
                         int& zero_reads, double floor_per_base, double floor_start) {
  int cnt = 0;  // graph.cc:3062-3088 (the rp.dat dump at :3071-3086 is dropped on purpose)
  LogD total = LogD::from_linear(1);
  if (total_len == 0) total_len = 1;
  zero_reads = 0;
  for (size_t i = 0; i < probs.size(); i++) {
    LogD p = probs[i];
    LogD floor = ld_mul(LogD::from_linear(std::exp(floor_start)),
                        ld_pow(LogD::from_linear(std::exp(floor_per_base)), r.lens[i]));
    if (ld_lt(p, floor)) { zero_reads++; p = floor; }
    total = ld_mul(total, p);
    cnt++;
  }
  return total.lv / cnt - std::log(2 * total_len);
}

double score_pacbio(const Graph& g, std::vector<std::vector<int>> paths, LongReadSet& r, int& zero_reads,
                    int& total_len, double penalty, double cov_move, double floor_per_base,
                    double floor_start, std::vector<double>* logprobs_out, int* bad_bases_out) {
  std::vector<LogD> probs(r.n());  // graph.cc:3171-3261
  total_len = 0;
  int bad_bases = 0;
  for (auto& path : paths) {
    g.normalize_walk(path);
    // gaps do NOT split the path here (loop body commented out at :3188-3194)
    std::vector<std::pair<int, int>> events;
    events.push_back(std::make_pair(-1000, 1));
    events.push_back(std::make_pair(2000, -3000));
    int pp = 0;
    for (int e : path) {
      if (e >= 0) {
        int cl = g.len(e);
        events.push_back(std::make_pair(pp, 1));
        events.push_back(std::make_pair(pp + cl, -cl));
        pp += cl;
      } else pp += -e;
    }
    int tl = 0;
    std::vector<std::vector<std::pair<std::pair<int, int>, LogD>>> pos;
    r.read_probabilities(g, path, tl, pos);
    for (size_t i = 0; i < pos.size(); i++) {
      for (auto& p : pos[i]) {
        if (ld_lt(p.second, r.min_read_prob((int)i))) continue;
        events.push_back(std::make_pair(p.first.first, 1));
        events.push_back(std::make_pair(p.first.second, p.first.first - p.first.second));
      }
    }
    for (size_t i = 0; i < pos.size(); i++)  // AddPositionsToReadProbsPacbio :3052-3060
      for (auto& p : pos[i]) probs[i] = ld_add(probs[i], p.second);
    total_len += tl;
    std::sort(events.begin(), events.end());
    std::multiset<int> open;
    for (size_t j = 0; j < events.size(); j++) {  // :3228-3250
      if (events[j].second == 1) open.insert(events[j].first);
      if (events[j].second != 1) {
        auto it = open.find(events[j].first + events[j].second);
        if (it != open.end()) open.erase(it);
      }
      int good_start = tl - 250;
      if (!open.empty()) good_start = (int)(*open.begin() + cov_move);
      if (j + 1 < events.size()) good_start = std::min(events[j + 1].first, good_start);
      good_start = std::min(good_start, tl - 250);
      int from = std::max(2500, events[j].first);
      if (good_start > from) bad_bases += good_start - from;
    }
  }
  double tp = total_prob_pacbio(probs, total_len, r, zero_reads, floor_per_base, floor_start);
  if (logprobs_out) { logprobs_out->resize(probs.size()); for (size_t i = 0; i < probs.size(); i++) (*logprobs_out)[i] = probs[i].lv; }
  if (bad_bases_out) *bad_bases_out = bad_bases;
  return tp - bad_bases * penalty;
}

// ---------------------------------------------------------------------------
// SAM record -> alignment probability (cache-miss side of the PacBio path)
// ---------------------------------------------------------------------------
std::vector<std::pair<int, char>> LongReadSet::parse_cigar(const std::string& c) {  // :3023-3038
  std::vector<std::pair<int, char>> r;
  int start = 0;
  for (int i = 0; i < (int)c.size(); i++) {
    if (c[i] < '0' || c[i] > '9') {
      if (c[i] == 'M' || c[i] == 'I' || c[i] == 'D') {
        r.push_back(std::make_pair(atoi(c.substr(start, i - start).c_str()), c[i]));
        start = i + 1;
      }
    }
  }
  return r;
}

SamAlignment LongReadSet::parse_sam_line(const std::string& line, int total_len, bool do_reverse) {
  SamAlignment a;  // graph.cc:2945-3021
  std::vector<std::string> f;
  size_t s = 0;
  while (true) {
    size_t t = line.find('\t', s);
    if (t == std::string::npos) { f.push_back(line.substr(s)); break; }
    f.push_back(line.substr(s, t - s));
    s = t + 1;
  }
  size_t slash = 0;
  for (size_t i = 0; i < f[0].size(); i++) if (f[0][i] == '/') slash = i;
  a.name = f[0].substr(0, slash);
  int posstart = atoi(f[3].c_str());
  a.flags = atoi(f[1].c_str());
  a.len = atoi(f[8].c_str());
  int posend = posstart + a.len;
  a.sstart = 0; a.send = (int)f[9].size(); a.slen = (int)f[9].size(); a.edit_dist = 100000;
  for (size_t i = 11; i < f.size(); i++) {
    if (f[i].size() < 2) continue;
    int v = f[i].size() > 5 ? atoi(f[i].substr(5).c_str()) : 0;
    if (f[i][0] == 'X' && f[i][1] == 'S') a.sstart = v - 1;
    if (f[i][0] == 'X' && f[i][1] == 'E') a.send = v - 1;
    if (f[i][0] == 'X' && f[i][1] == 'Q') a.slen = v;
    if (f[i][0] == 'N' && f[i][1] == 'M') a.edit_dist = v;
  }
  a.tstart = posstart; a.tend = posend;
  a.cigar = parse_cigar(f[5]);
  if ((a.flags & 16) && do_reverse) {
    int l = posend - posstart;
    posstart = total_len - posend;
    posend = posstart + l;
    std::reverse(a.cigar.begin(), a.cigar.end());
  }
  if (a.send != a.slen) a.cigar.push_back(std::make_pair(a.slen - a.send, 'I'));
  if (a.sstart != 0) {
    int match = std::min(a.sstart, posstart), left = a.sstart - match;
    a.cigar.insert(a.cigar.begin(), std::make_pair(match, 'I'));
    if (left) a.cigar.insert(a.cigar.begin(), std::make_pair(left, 'I'));
  }
  a.posstart = posstart; a.posend = posend;
  return a;
}

LogD LongReadSet::pair_match(char a, char b) const {  // graph.h:555-564
  if (a == '\n' || b == '\n') return LogD::from_linear(0);  // kContigSeparator
  return a != b ? mismatch_p : match_p;
}

namespace {
// all (row, col) cells from min to max column of every row that appears (Uniquify :2153-2173)
void fill_rows(std::vector<std::pair<int, int>>& x) {
  if (x.empty()) return;
  int lo = x[0].first, hi = x[0].first;
  for (auto& e : x) { lo = std::min(lo, e.first); hi = std::max(hi, e.first); }
  std::vector<std::pair<int, int>> span(hi - lo + 1, std::make_pair(1000000, -1000000));
  for (auto& e : x) {
    span[e.first - lo].first = std::min(span[e.first - lo].first, e.second);
    span[e.first - lo].second = std::max(span[e.first - lo].second, e.second);
  }
  x.clear();
  for (int i = lo; i <= hi; i++)
    for (int j = span[i - lo].first; j <= span[i - lo].second; j++) x.push_back(std::make_pair(i, j));
}
}  // namespace

std::vector<std::pair<int, int>> LongReadSet::dp_cells(const SamAlignment& a, int band) {
  std::string cig;  // ExpandCigar :2129-2136
  for (auto& c : a.cigar) cig.append((size_t)std::max(0, c.first), c.second);
  int bl = 0, el = 0;  // GetCigarEnds :2138-2151 (values left untouched if the cigar is all 'I')
  for (int i = 0; i < (int)cig.size(); i++) if (cig[i] != 'I') { bl = i; break; }
  for (int i = (int)cig.size() - 1; i >= 0; i--) if (cig[i] != 'I') { el = (int)cig.size() - i; break; }
  bl = std::min(bl, 200); el = std::min(el, 200);
  std::vector<std::pair<int, int>> cells;  // :2183-2221
  int row = 0, col = 0;
  cells.push_back(std::make_pair(0, 0));
  for (int i = -bl; i < 3; i++) for (int j = 0; j < bl; j++) cells.push_back(std::make_pair(i, j));
  for (char c : cig) {
    if (c == 'M') { row++; col++; } else if (c == 'I') col++; else if (c == 'D') row++;
    cells.push_back(std::make_pair(row, col));
  }
  for (int i = row; i < row + el; i++) for (int j = col - el; j <= col; j++) cells.push_back(std::make_pair(i, j));
  fill_rows(cells);
  std::vector<std::pair<int, int>> halo;
  for (auto& e : cells)
    for (int i = -band; i <= band; i++) for (int j = -band; j <= band; j++) halo.push_back(std::make_pair(e.first + i, e.second + j));
  cells.insert(cells.end(), halo.begin(), halo.end());
  fill_rows(cells);
  return cells;
}

LogD LongReadSet::alignment_probability(const std::string& s1, const std::string& s2, const SamAlignment& a,
                                        int band) const {
  std::vector<std::pair<int, int>> cells = dp_cells(a, band);

  int off = cells[0].first, nrows = cells.back().first - off + 1;  // :2223-2235
  std::vector<int> row_lo(nrows, cells.back().second + 1000000);
  std::vector<std::vector<LogD>> val(nrows);
  for (auto& e : cells) row_lo[e.first - off] = std::min(row_lo[e.first - off], e.second);
  for (auto& e : cells) {
    size_t need = (size_t)(e.second - row_lo[e.first - off]) + 1;
    if (val[e.first - off].size() < need) val[e.first - off].resize(need);
  }
  LogD ret = LogD::from_linear(0);  // :2237
  const char gap = '-';
  for (auto& e : cells) if (e.second == 0) val[e.first - off][e.second - row_lo[e.first - off]] = LogD::from_linear(1);
  auto inside = [&](int r, int c) {
    return r - off >= 0 && c - row_lo[r - off] >= 0 && c - row_lo[r - off] < (int)val[r - off].size();
  };
  for (auto& e : cells) {  // :2245-2282
    int r = e.first, c = e.second;
    if (c == 0) continue;
    if (c - 1 < 0 || c - 1 >= (int)s2.size()) continue;
    int gi = r + a.posstart - 1;
    if (gi < 0 || gi >= (int)s1.size()) continue;
    LogD& cur = val[r - off][c - row_lo[r - off]];
    if (inside(r - 1, c - 1)) cur = ld_add(cur, ld_mul(val[r - 1 - off][c - 1 - row_lo[r - 1 - off]], pair_match(s1[gi], s2[c - 1])));
    if (inside(r - 1, c)) cur = ld_add(cur, ld_mul(val[r - 1 - off][c - row_lo[r - 1 - off]], pair_match(s1[gi], gap)));
    if (inside(r, c - 1)) cur = ld_add(cur, ld_mul(val[r - off][c - 1 - row_lo[r - off]], pair_match(gap, s2[c - 1])));
    if (c == (int)s2.size()) ret = ld_add(ret, cur);
  }
  return ret;
}

int LongReadSet::ingest_sam(const Graph& g, const std::vector<int>& path, const std::vector<std::string>& sam_lines) {
  // path string and node boundaries (graph.cc:2662-2688)
  std::string seq;
  std::vector<int> ends, begins;
  if (path[0] >= 0) seq = g.seq[path[0]]; else seq.assign((size_t)-path[0], 'N');
  ends.push_back((int)seq.size());
  begins.push_back(0);
  for (size_t i = 1; i < path.size(); i++) {
    begins.push_back((int)seq.size());
    if (path[i] < 0) seq.append((size_t)-path[i], 'N'); else seq += g.seq[path[i]];
    ends.push_back((int)seq.size());
  }
  const std::string seqall = seq + '\n' + revcomp(seq);  // kContigSeparator graph.cc:30, 2687-2688
  // sub-walks this call may file under (graph.cc:2724-2743)
  std::unordered_map<std::vector<int>, int, WalkHash> starts;
  std::unordered_set<std::vector<int>, WalkHash> dont_save;
  for (size_t i = 0; i < path.size(); i++) {
    std::vector<int> sub;
    for (size_t j = i; j < path.size(); j++) {
      sub.push_back(path[j]);
      int sub_len = ends[j] - begins[i], first_len = ends[i] - begins[i];
      if (cache.count(sub)) dont_save.insert(sub); else cache[sub].clear();
      starts[sub] = (int)i;
      if (sub_len - first_len > max_len) break;
    }
  }
  int filed = 0;
  for (const std::string& l : sam_lines) {  // graph.cc:2746-2786
    if (l.empty() || l[0] == '@') continue;
    SamAlignment a = parse_sam_line(l, (int)seqall.size());
    auto it = name_to_id.find(a.name);
    if (it == name_to_id.end()) continue;  // the reference asserts (:2751)
    const int read_id = it->second;
    LogD prob = alignment_probability(seqall, reads[read_id], a, 2);
    int it_begin = (int)(std::lower_bound(ends.begin(), ends.end(), std::max(0, a.tstart - 5)) - ends.begin());
    int it_end = (int)(std::lower_bound(ends.begin(), ends.end(), std::min(a.tstart + a.len + 5, (int)seq.size())) - ends.begin());
    if (it_begin >= (int)path.size() || it_end >= (int)path.size()) continue;  // asserted in the reference (:2767-2770)
    std::vector<int> sub(path.begin() + it_begin, path.begin() + it_end + 1);
    int pos_begin = it_begin > 0 ? ends[it_begin - 1] : 0;
    auto st = starts.find(sub);
    if (st != starts.end() && st->second == it_begin && dont_save.count(sub) == 0) {
      cache[sub].push_back(LongRec{a.tstart - pos_begin, a.tend - pos_begin, read_id, prob});
      filed++;
    }
  }
  return filed;
}

// ---------------------------------------------------------------------------
// Aggregation + config
// ---------------------------------------------------------------------------
double Calculator::calc_prob(const std::vector<std::vector<int>>& paths, std::vector<std::pair<int, int>>& zeros,
                             int& total_len, bool fresh) {
  zeros.clear();  // prob_calculator.h:63-109
  if (paired_state.size() != paired.size()) paired_state.resize(paired.size());
  double prob = 0;
  for (auto& e : single) {
    int zero = 0;
    prob += score_single(*g, paths, *e.second, zero, total_len, e.first.penalty_constant, e.first.step,
                         e.first.min_prob_per_base, e.first.min_prob_start) * e.first.weight;
    zeros.push_back(std::make_pair(zero, e.second->n()));
  }
  int ind = 0;
  for (auto& e : paired) {
    int zero = 0;
    if (fresh) paired_state[ind] = PairedState();
    double s = score_paired(*g, paths, *e.second.first, *e.second.second, e.first.insert_mean, e.first.insert_std,
                            zero, total_len, paired_state[ind], e.first.penalty_constant, e.first.step, true,
                            e.first.min_prob_per_base, e.first.min_prob_start) * e.first.weight;
    zeros.push_back(std::make_pair(zero, e.second.first->n()));
    prob += s;
    ind++;
  }
  for (auto& e : pacbio) {
    int zero = 0;
    prob += score_pacbio(*g, paths, *e.second, zero, total_len, e.first.penalty_constant, e.first.step,
                         e.first.min_prob_per_base, e.first.min_prob_start) * e.first.weight;
    zeros.push_back(std::make_pair(zero, e.second->n()));
  }
  return prob;
}

bool load_config(const std::string& file, KV& global, std::unordered_map<std::string, KV>& sets) {
  std::ifstream f(file.c_str());  // gaml.cc:748-780
  if (f.fail()) return false;
  std::string cur, l;
  while (std::getline(f, l)) {
    if (l.empty()) continue;
    if (l[0] == '[') cur = l.substr(1, l.size() - 2);
    else if (l[0] >= 'a' && l[0] <= 'z') {
      size_t eq = l.find('=');
      if (eq == std::string::npos) return false;
      if (cur.empty()) global[l.substr(0, eq)] = l.substr(eq + 1);
      else sets[cur][l.substr(0, eq)] = l.substr(eq + 1);
    }
  }
  return true;
}

static double get_d(KV& kv, const char* k, double def) {  // ExtractDouble gaml.cc:32-37
  auto it = kv.find(k);
  return it == kv.end() ? def : atof(it->second.c_str());
}

std::vector<ReadSetSpec> readsets_from_config(std::unordered_map<std::string, KV>& sets) {
  std::vector<ReadSetSpec> out;  // gaml.cc:783-872
  for (auto& e : sets) {         // hash order of the unordered_map, as the reference (:788)
    KV& kv = e.second;
    if (!kv.count("type")) continue;
    ReadSetSpec s;
    s.name = e.first; s.type = kv["type"];
    double weight = get_d(kv, "weight", 1);
    bool advice = kv.count("advice") > 0;
    if (s.type == "single" || s.type == "pacbio") {
      if (!kv.count("filename")) continue;
      s.file1 = kv["filename"];
      s.mismatch = get_d(kv, "mismatch_prob", 0.01);
      s.match = 1.0 - 4 * s.mismatch;
      s.scfg.min_prob_per_base = get_d(kv, "min_prob_per_base", -0.7);
      s.scfg.min_prob_start = get_d(kv, "min_prob_start", -10);
      s.scfg.penalty_constant = get_d(kv, "penalty_constant", 0);
      s.scfg.step = get_d(kv, "penalty_step", 50);
      s.scfg.weight = weight; s.scfg.advice = advice;
    } else if (s.type == "paired") {
      if (!kv.count("filename1") || !kv.count("filename2") || !kv.count("insert_mean") || !kv.count("insert_std")) continue;
      s.file1 = kv["filename1"]; s.file2 = kv["filename2"];
      s.pcfg.insert_mean = atof(kv["insert_mean"].c_str());
      s.pcfg.insert_std = atof(kv["insert_std"].c_str());
      s.mismatch = get_d(kv, "mismatch_prob", 0.01);
      s.match = 1.0 - 4 * s.mismatch;
      s.pcfg.min_prob_per_base = get_d(kv, "min_prob_pre_base", -0.7);  // sic, gaml.cc:855
      s.pcfg.min_prob_start = get_d(kv, "min_prob_start", -10);
      s.pcfg.penalty_constant = get_d(kv, "penalty_constant", 0);
      s.pcfg.step = s.pcfg.insert_mean - get_d(kv, "penalty_step", 50);  // gaml.cc:860
      s.pcfg.weight = weight; s.pcfg.advice = advice;
    } else continue;
    out.push_back(s);
  }
  return out;
}

}  // namespace orc
