// gaml_oracle_c.cc -- TEST INFRASTRUCTURE ONLY: flat C interface over gaml_oracle.hpp so
// tests/ and bench.py's cpu_baseline leg can drive the oracle through ctypes.
#include <cstring>
#include <memory>

#include "gaml_oracle.hpp"

using namespace orc;

namespace {
struct Session {
  Graph g;
  std::vector<std::unique_ptr<ShortReadSet>> shorts;
  std::vector<std::unique_ptr<LongReadSet>> longs;
  Calculator calc;
  // order of creation -> (kind, index inside calc vector); kind 0 single, 1 paired, 2 pacbio
  std::vector<std::pair<int, int>> sets;
  Session() { calc.g = &g; }
};
Session* S(void* p) { return (Session*)p; }

std::vector<std::vector<int>> unflatten(const int32_t* flat, const int64_t* offs, int n) {
  std::vector<std::vector<int>> r(n);
  for (int i = 0; i < n; i++) r[i].assign(flat + offs[i], flat + offs[i + 1]);
  return r;
}
ShortReadSet* short_set(Session* s, int set, int mate) {
  auto k = s->sets[set];
  if (k.first == 0) return s->calc.single[k.second].second;
  if (k.first == 1) return mate == 0 ? s->calc.paired[k.second].second.first : s->calc.paired[k.second].second.second;
  return nullptr;
}
}  // namespace

extern "C" {

void* orc_new() { return new Session(); }
void orc_free(void* p) { delete S(p); }

int orc_load_graph(void* p, const char* file) {
  return load_lastgraph(file, S(p)->g) ? (int)S(p)->g.seq.size() : -1;
}
int orc_set_graph(void* p, int n_nodes, const char* bases, const int64_t* offs) {
  Graph& g = S(p)->g;
  g.seq.resize(n_nodes);
  for (int i = 0; i < n_nodes; i++) g.seq[i].assign(bases + offs[i], bases + offs[i + 1]);
  g.calc_normalize_map();
  return n_nodes;
}
int orc_num_nodes(void* p) { return (int)S(p)->g.seq.size(); }
int orc_node_len(void* p, int node) { return S(p)->g.len(node); }
int orc_normalize_node(void* p, int node) { return S(p)->g.normalize_map[node]; }

static ShortReadSet* new_short(Session* s, double mismatch) {
  s->shorts.emplace_back(new ShortReadSet());
  ShortReadSet* r = s->shorts.back().get();
  r->mismatch_p = mismatch;
  r->match_p = 1.0 - 4 * mismatch;  // gaml.cc:813,854
  return r;
}

// cfg = {penalty_constant, step, min_prob_per_base, min_prob_start, weight}
int orc_add_single_fastq(void* p, const char* file, double mismatch, const double* cfg) {
  Session* s = S(p);
  ShortReadSet* r = new_short(s, mismatch);
  if (!r->load_fastq(file)) return -1;
  SingleCfg c; c.penalty_constant = cfg[0]; c.step = cfg[1]; c.min_prob_per_base = cfg[2]; c.min_prob_start = cfg[3]; c.weight = cfg[4];
  s->calc.single.push_back(std::make_pair(c, r));
  s->sets.push_back(std::make_pair(0, (int)s->calc.single.size() - 1));
  return (int)s->sets.size() - 1;
}

// cfg = {penalty_constant, step, insert_mean, insert_std, min_prob_per_base, min_prob_start, weight}
int orc_add_paired_fastq(void* p, const char* f1, const char* f2, double mismatch, const double* cfg) {
  Session* s = S(p);
  ShortReadSet* a = new_short(s, mismatch);
  ShortReadSet* b = new_short(s, mismatch);
  if (!a->load_fastq(f1) || !b->load_fastq(f2)) return -1;
  if (a->n() != b->n()) return -2;
  PairedCfg c; c.penalty_constant = cfg[0]; c.step = cfg[1]; c.insert_mean = cfg[2]; c.insert_std = cfg[3];
  c.min_prob_per_base = cfg[4]; c.min_prob_start = cfg[5]; c.weight = cfg[6];
  s->calc.paired.push_back(std::make_pair(c, std::make_pair(a, b)));
  s->calc.paired_state.resize(s->calc.paired.size());
  s->sets.push_back(std::make_pair(1, (int)s->calc.paired.size() - 1));
  return (int)s->sets.size() - 1;
}

// reads given as concatenated bases + offsets (n+1)
static std::vector<std::string> unpack_reads(int n, const char* bases, const int64_t* offs) {
  std::vector<std::string> r(n);
  for (int i = 0; i < n; i++) r[i].assign(bases + offs[i], bases + offs[i + 1]);
  return r;
}
int orc_add_single(void* p, int n, const char* bases, const int64_t* offs, double mismatch, const double* cfg) {
  Session* s = S(p);
  ShortReadSet* r = new_short(s, mismatch);
  r->set_reads(unpack_reads(n, bases, offs));
  SingleCfg c; c.penalty_constant = cfg[0]; c.step = cfg[1]; c.min_prob_per_base = cfg[2]; c.min_prob_start = cfg[3]; c.weight = cfg[4];
  s->calc.single.push_back(std::make_pair(c, r));
  s->sets.push_back(std::make_pair(0, (int)s->calc.single.size() - 1));
  return (int)s->sets.size() - 1;
}
int orc_add_paired(void* p, int n, const char* b1, const int64_t* o1, const char* b2, const int64_t* o2,
                   double mismatch, const double* cfg) {
  Session* s = S(p);
  ShortReadSet* a = new_short(s, mismatch);
  ShortReadSet* b = new_short(s, mismatch);
  a->set_reads(unpack_reads(n, b1, o1));
  b->set_reads(unpack_reads(n, b2, o2));
  PairedCfg c; c.penalty_constant = cfg[0]; c.step = cfg[1]; c.insert_mean = cfg[2]; c.insert_std = cfg[3];
  c.min_prob_per_base = cfg[4]; c.min_prob_start = cfg[5]; c.weight = cfg[6];
  s->calc.paired.push_back(std::make_pair(c, std::make_pair(a, b)));
  s->calc.paired_state.resize(s->calc.paired.size());
  s->sets.push_back(std::make_pair(1, (int)s->calc.paired.size() - 1));
  return (int)s->sets.size() - 1;
}
int orc_add_pacbio(void* p, int n, const int32_t* lens, double mismatch, const double* cfg) {
  Session* s = S(p);
  s->longs.emplace_back(new LongReadSet());
  LongReadSet* r = s->longs.back().get();
  r->set_params(1.0 - 4 * mismatch, mismatch);
  r->lens.assign(lens, lens + n);
  r->finalize();
  SingleCfg c; c.penalty_constant = cfg[0]; c.step = cfg[1]; c.min_prob_per_base = cfg[2]; c.min_prob_start = cfg[3]; c.weight = cfg[4];
  s->calc.pacbio.push_back(std::make_pair(c, r));
  s->sets.push_back(std::make_pair(2, (int)s->calc.pacbio.size() - 1));
  return (int)s->sets.size() - 1;
}
// append alignment records (pos, pos_end, read) + log-probabilities for one sub-walk
int orc_pacbio_put(void* p, int set, const int32_t* walk, int walk_len_, const int32_t* rec3, const double* logp, int nrec) {
  Session* s = S(p);
  auto k = s->sets[set];
  if (k.first != 2) return -1;
  LongReadSet* r = s->calc.pacbio[k.second].second;
  std::vector<int> key(walk, walk + walk_len_);
  auto& dst = r->cache[key];
  for (int i = 0; i < nrec; i++) dst.push_back(LongRec{rec3[3 * i], rec3[3 * i + 1], rec3[3 * i + 2], LogD::from_log(logp[i])});
  return (int)dst.size();
}
// PacBio set with read bases + names (needed for SAM ingestion)
int orc_add_pacbio_reads(void* p, int n, const char* bases, const int64_t* offs, const char* names_nl, double mismatch, const double* cfg) {
  Session* s = S(p);
  s->longs.emplace_back(new LongReadSet());
  LongReadSet* r = s->longs.back().get();
  r->set_params(1.0 - 4 * mismatch, mismatch);
  const char* nm = names_nl;
  for (int i = 0; i < n; i++) {
    r->reads.emplace_back(bases + offs[i], bases + offs[i + 1]);
    r->lens.push_back((int)(offs[i + 1] - offs[i]));
    const char* e = strchr(nm, '\n');
    std::string name = e ? std::string(nm, e) : std::string(nm);
    nm = e ? e + 1 : nm + name.size();
    r->name_to_id[name] = i;
  }
  r->finalize();
  SingleCfg c; c.penalty_constant = cfg[0]; c.step = cfg[1]; c.min_prob_per_base = cfg[2]; c.min_prob_start = cfg[3]; c.weight = cfg[4];
  s->calc.pacbio.push_back(std::make_pair(c, r));
  s->sets.push_back(std::make_pair(2, (int)s->calc.pacbio.size() - 1));
  return (int)s->sets.size() - 1;
}
int orc_pacbio_ingest_sam(void* p, int set, const int32_t* path, int n, const char* sam_text) {
  Session* s = S(p);
  auto k = s->sets[set];
  if (k.first != 2) return -1;
  std::vector<std::string> lines;
  const char* q = sam_text;
  while (*q) { const char* e = strchr(q, '\n'); if (!e) { lines.emplace_back(q); break; } lines.emplace_back(q, e); q = e + 1; }
  std::vector<int> w(path, path + n);
  s->g.normalize_walk(w);  // CalcScoreForPacbio normalises before GetReadProbabilities (graph.cc:3180)
  return s->calc.pacbio[k.second].second->ingest_sam(s->g, w, lines);
}
// dump the PacBio cache: for one sub-walk, (pos, pos_end, read) triples + log-probabilities; -1 if absent
int orc_pacbio_records(void* p, int set, const int32_t* walk, int n, int32_t* rec3, double* logp, int cap) {
  Session* s = S(p);
  LongReadSet* r = s->calc.pacbio[s->sets[set].second].second;
  auto it = r->cache.find(std::vector<int>(walk, walk + n));
  if (it == r->cache.end()) return -1;
  int cnt = (int)it->second.size();
  for (int i = 0; i < cnt && i < cap; i++) { rec3[3 * i] = it->second[i].pos; rec3[3 * i + 1] = it->second[i].pos_end; rec3[3 * i + 2] = it->second[i].read; logp[i] = it->second[i].prob.lv; }
  return cnt;
}
long orc_pacbio_keys(void* p, int set, int32_t* out, long cap) {
  LongReadSet* r = S(p)->calc.pacbio[S(p)->sets[set].second].second;
  long need = 0;
  for (auto& e : r->cache) {
    if (need + 1 + (long)e.first.size() <= cap) { out[need] = (int32_t)e.first.size(); for (size_t i = 0; i < e.first.size(); i++) out[need + 1 + i] = e.first[i]; }
    need += 1 + (long)e.first.size();
  }
  return need;
}
long orc_pacbio_misses(void* p, int set) { return S(p)->calc.pacbio[S(p)->sets[set].second].second->cache_misses; }

int orc_num_sets(void* p) { return (int)S(p)->sets.size(); }
int orc_set_kind(void* p, int set) { return S(p)->sets[set].first; }
int orc_set_reads(void* p, int set) {
  Session* s = S(p);
  auto k = s->sets[set];
  if (k.first == 0) return s->calc.single[k.second].second->n();
  if (k.first == 1) return s->calc.paired[k.second].second.first->n();
  return s->calc.pacbio[k.second].second->n();
}

// CalcProb; zeros_out gets 2 ints per read set in reference order (single, paired, pacbio)
double orc_calc_prob(void* p, const int32_t* flat, const int64_t* offs, int n_paths, int fresh,
                     int32_t* zeros_out, int32_t* total_len_out) {
  Session* s = S(p);
  std::vector<std::pair<int, int>> zeros;
  int tl = 0;
  double v = s->calc.calc_prob(unflatten(flat, offs, n_paths), zeros, tl, fresh != 0);
  if (zeros_out) for (size_t i = 0; i < zeros.size(); i++) { zeros_out[2 * i] = zeros[i].first; zeros_out[2 * i + 1] = zeros[i].second; }
  if (total_len_out) *total_len_out = tl;
  return v;
}

// per-read state after the last paired CalcProb of that set
int orc_paired_probs(void* p, int set, double* out, int32_t* bad_bases) {
  Session* s = S(p);
  auto k = s->sets[set];
  if (k.first != 1) return -1;
  PairedState& st = s->calc.paired_state[k.second];
  memcpy(out, st.probs.data(), st.probs.size() * sizeof(double));
  if (bad_bases) *bad_bases = st.bad_bases;
  return (int)st.probs.size();
}
// the slow non-incremental paired scorer (graph.cc:1991-2127) with per-read output; out3 = {zero_reads, total_len, bad_bases}
double orc_paired_slow(void* p, int set, const int32_t* flat, const int64_t* offs, int n_paths, double* probs_out, int32_t* out3, uint8_t* several_out) {
  Session* s = S(p);
  auto k = s->sets[set];
  if (k.first != 1) return 0.0;
  auto& e = s->calc.paired[k.second];
  std::vector<double> probs; int zero = 0, tl = 0, bad = 0;
  std::vector<uint8_t> several;
  double v = score_paired_slow(s->g, unflatten(flat, offs, n_paths), *e.second.first, *e.second.second, e.first.insert_mean, e.first.insert_std,
                               zero, tl, e.first.penalty_constant, e.first.step, true /* use_all_to_cov: prob_calculator.h:84 */,
                               e.first.min_prob_per_base, e.first.min_prob_start, &probs, &bad, &several);
  if (probs_out) memcpy(probs_out, probs.data(), probs.size() * sizeof(double));
  if (several_out) memcpy(several_out, several.data(), several.size());
  if (out3) { out3[0] = zero; out3[1] = tl; out3[2] = bad; }
  return v;
}
// single-end / pacbio scorers recomputed with per-read output (stateless in the reference)
double orc_single_detail(void* p, int set, const int32_t* flat, const int64_t* offs, int n_paths, double* probs_out,
                         int32_t* out3 /*zero_reads,total_len,bad_bases*/) {
  Session* s = S(p);
  auto k = s->sets[set];
  auto& e = s->calc.single[k.second];
  std::vector<double> probs; int zero = 0, tl = 0, bad = 0;
  double v = score_single(s->g, unflatten(flat, offs, n_paths), *e.second, zero, tl, e.first.penalty_constant, e.first.step,
                          e.first.min_prob_per_base, e.first.min_prob_start, &probs, &bad);
  if (probs_out) memcpy(probs_out, probs.data(), probs.size() * sizeof(double));
  if (out3) { out3[0] = zero; out3[1] = tl; out3[2] = bad; }
  return v;
}
double orc_pacbio_detail(void* p, int set, const int32_t* flat, const int64_t* offs, int n_paths, double* logprobs_out,
                         int32_t* out3) {
  Session* s = S(p);
  auto k = s->sets[set];
  auto& e = s->calc.pacbio[k.second];
  std::vector<double> lp; int zero = 0, tl = 0, bad = 0;
  double v = score_pacbio(s->g, unflatten(flat, offs, n_paths), *e.second, zero, tl, e.first.penalty_constant, e.first.step,
                          e.first.min_prob_per_base, e.first.min_prob_start, &lp, &bad);
  if (logprobs_out) memcpy(logprobs_out, lp.data(), lp.size() * sizeof(double));
  if (out3) { out3[0] = zero; out3[1] = tl; out3[2] = bad; }
  return v;
}

// ---- stage access: alignment window cache --------------------------------------------
int orc_window_count(void* p, int set, int mate) { return (int)short_set(S(p), set, mate)->cache.size(); }
long orc_windows_aligned(void* p, int set, int mate) { return short_set(S(p), set, mate)->windows_aligned; }
// records of one cached window as int32 quadruples (pos, edit, read, orient); -1 if absent
int orc_window_records(void* p, int set, int mate, const int32_t* walk, int n, int32_t* out, int cap) {
  ShortReadSet* r = short_set(S(p), set, mate);
  auto it = r->cache.find(std::vector<int>(walk, walk + n));
  if (it == r->cache.end()) return -1;
  int cnt = (int)it->second.size();
  for (int i = 0; i < cnt && i < cap; i++) {
    out[4 * i] = it->second[i].pos; out[4 * i + 1] = it->second[i].edit;
    out[4 * i + 2] = it->second[i].read; out[4 * i + 3] = it->second[i].orient;
  }
  return cnt;
}
int orc_align_window(void* p, int set, int mate, const int32_t* walk, int n) {
  ShortReadSet* r = short_set(S(p), set, mate);
  std::vector<std::vector<int>> w(1, std::vector<int>(walk, walk + n));
  r->align_windows(S(p)->g, w);
  return (int)r->cache[w[0]].size();
}
// dump every cached window key: out = [len, ids..., len, ids...]; returns ints needed
long orc_window_keys(void* p, int set, int mate, int32_t* out, long cap) {
  ShortReadSet* r = short_set(S(p), set, mate);
  long need = 0;
  for (auto& e : r->cache) {
    if (need + 1 + (long)e.first.size() <= cap) {
      out[need] = (int32_t)e.first.size();
      for (size_t i = 0; i < e.first.size(); i++) out[need + 1 + i] = e.first[i];
    }
    need += 1 + (long)e.first.size();
  }
  return need;
}
int orc_window_string(void* p, int set, int mate, const int32_t* walk, int n, char* out, int cap, int32_t* offset) {
  int off = 0;
  std::string s = short_set(S(p), set, mate)->window_string(S(p)->g, std::vector<int>(walk, walk + n), &off);
  if ((int)s.size() < cap) memcpy(out, s.c_str(), s.size() + 1);
  if (offset) *offset = off;
  return (int)s.size();
}
// assembled positions of one contig (GetPositionsOnlyPath): out quadruples in per-read
// order of first appearance sorted by read id for a stable dump
int orc_positions_only_path(void* p, int set, int mate, const int32_t* ctg, int n, int st, int32_t* out, int cap) {
  std::unordered_map<int, std::vector<Rec>> acc;
  short_set(S(p), set, mate)->positions_only_path(S(p)->g, std::vector<int>(ctg, ctg + n), st, acc);
  std::map<int, std::vector<Rec>> sorted(acc.begin(), acc.end());
  int cnt = 0;
  for (auto& e : sorted)
    for (auto& r : e.second) {
      if (cnt < cap) { out[4 * cnt] = r.pos; out[4 * cnt + 1] = r.edit; out[4 * cnt + 2] = r.read; out[4 * cnt + 3] = r.orient; }
      cnt++;
    }
  return cnt;
}
// CalcScoreForPathInc on one path: per-read summed contribution + bad_bases
int orc_score_path_paired(void* p, int set, const int32_t* path, int n, double* probs_out /*N, zero-filled by callee*/,
                          int32_t* bad_bases, int64_t* n_terms) {
  Session* s = S(p);
  auto k = s->sets[set];
  if (k.first != 1) return -1;
  auto& e = s->calc.paired[k.second];
  ShortReadSet& a = *e.second.first; ShortReadSet& b = *e.second.second;
  std::vector<std::vector<int>> one(1, std::vector<int>(path, path + n));
  a.precompute_for_paths(s->g, one);
  b.precompute_for_paths(s->g, one);
  PathScore ps;
  score_path_paired(s->g, one[0], a, b, e.first.insert_mean, e.first.insert_std, e.first.step, true,
                    e.first.min_prob_per_base, e.first.min_prob_start, ps);
  for (int i = 0; i < a.n(); i++) probs_out[i] = 0;
  for (auto& c : ps.changes) probs_out[c.first] += c.second;
  if (bad_bases) *bad_bases = ps.bad_bases;
  if (n_terms) *n_terms = (int64_t)ps.changes.size();
  return a.n();
}

// ---- stage access: primitives ---------------------------------------------------------
int orc_extend_hit(int win_pos, int read_pos, const char* read, const char* win, int32_t* out3) {
  HitResult h = extend_hit(win_pos, read_pos, read, win);
  out3[0] = h.errs; out3[1] = h.begin; out3[2] = h.end;
  return h.errs;
}
double orc_insert_prob(double len, double mean, double sd) { return insert_prob(len, mean, sd); }
uint64_t orc_max_hash(const char* s) { return MaxHashIndex::max_hash(s); }
// window_hashes for a given read length: out pairs (hash, pos)
int orc_window_hashes(const char* s, int read_len, uint64_t* hashes, int32_t* poses, int cap) {
  MaxHashIndex ix; ix.read_len = read_len;
  std::vector<std::pair<uint64_t, int>> v;
  ix.window_hashes(s, v);
  for (size_t i = 0; i < v.size() && (int)i < cap; i++) { hashes[i] = v[i].first; poses[i] = v[i].second; }
  return (int)v.size();
}
double orc_ld_from_linear(double x) { return LogD::from_linear(x).lv; }
double orc_ld_add(double a, double b) { return ld_add(LogD::from_log(a), LogD::from_log(b)).lv; }
double orc_ld_mul(double a, double b) { return ld_mul(LogD::from_log(a), LogD::from_log(b)).lv; }
double orc_ld_pow(double a, double e) { return ld_pow(LogD::from_log(a), e).lv; }
double orc_ld_div(double a, double b) { return ld_div(LogD::from_log(a), LogD::from_log(b)).lv; }
int orc_invert_walk(const int32_t* w, int n, int32_t* out) {
  std::vector<int> r = invert_walk(std::vector<int>(w, w + n));
  for (int i = 0; i < n; i++) out[i] = r[i];
  return n;
}
// SAM line -> banded alignment log-probability (cache-miss side of the PacBio path)
double orc_sam_alignment_logprob(const char* sam_line, const char* target_all, const char* read, double mismatch, int band,
                                 int32_t* out_tstart_tend /*2*/) {
  LongReadSet r; r.set_params(1.0 - 4 * mismatch, mismatch);
  std::string t(target_all);
  SamAlignment a = LongReadSet::parse_sam_line(sam_line, (int)t.size(), true);
  if (out_tstart_tend) { out_tstart_tend[0] = a.tstart; out_tstart_tend[1] = a.tend; }
  return r.alignment_probability(t, read, a, band).lv;
}

// parsed SAM fields {flags,len,posstart,posend,sstart,send,slen,tstart,tend,edit_dist} + the DP cell
// set as one column interval per row; returns the number of rows (needed size when > cap)
int orc_sam_band(const char* sam_line, int total_len, int32_t* fields10, int32_t* row0, int32_t* lo, int32_t* hi, int cap) {
  SamAlignment a = LongReadSet::parse_sam_line(sam_line, total_len, true);
  int32_t f[10] = {a.flags, a.len, a.posstart, a.posend, a.sstart, a.send, a.slen, a.tstart, a.tend, a.edit_dist};
  memcpy(fields10, f, sizeof(f));
  std::vector<std::pair<int, int>> cells = LongReadSet::dp_cells(a, 2);
  int r0 = cells.front().first, n = cells.back().first - r0 + 1;
  *row0 = r0;
  if (n > cap) return n;
  for (int i = 0; i < n; i++) { lo[i] = 1 << 30; hi[i] = -(1 << 30); }
  for (auto& e : cells) { lo[e.first - r0] = std::min(lo[e.first - r0], e.second); hi[e.first - r0] = std::max(hi[e.first - r0], e.second); }
  return n;
}

// ---- config-file driven set-up (gaml.cc main :935-1017 minus the optimiser) ------------
// returns number of read sets, <0 on error; read sets appear in the reference's order:
// hash order of the config map, grouped by CalcProb as single, paired, pacbio.
int orc_load_config(void* p, const char* cfg_file) {
  Session* s = S(p);
  KV global; std::unordered_map<std::string, KV> sets;
  if (!load_config(cfg_file, global, sets)) return -1;
  if (!global.count("graph")) return -2;
  if (!load_lastgraph(global["graph"], s->g)) return -3;
  for (auto& spec : readsets_from_config(sets)) {
    if (spec.type == "single") {
      ShortReadSet* r = new_short(s, spec.mismatch);
      if (!r->load_fastq(spec.file1)) return -4;
      s->calc.single.push_back(std::make_pair(spec.scfg, r));
      s->sets.push_back(std::make_pair(0, (int)s->calc.single.size() - 1));
    } else if (spec.type == "paired") {
      ShortReadSet* a = new_short(s, spec.mismatch);
      ShortReadSet* b = new_short(s, spec.mismatch);
      if (!a->load_fastq(spec.file1) || !b->load_fastq(spec.file2)) return -4;
      s->calc.paired.push_back(std::make_pair(spec.pcfg, std::make_pair(a, b)));
      s->sets.push_back(std::make_pair(1, (int)s->calc.paired.size() - 1));
    } else {
      // pacbio from a config needs the FASTQ only for read lengths
      s->longs.emplace_back(new LongReadSet());
      LongReadSet* r = s->longs.back().get();
      r->set_params(spec.match, spec.mismatch);
      ShortReadSet tmp; tmp.match_p = spec.match; tmp.mismatch_p = spec.mismatch;
      std::ifstream f(spec.file1.c_str());
      if (!f.is_open()) return -4;
      std::string l, q;
      while (std::getline(f, l)) { std::getline(f, q); r->lens.push_back((int)q.size()); r->reads.push_back(q); std::getline(f, l); std::getline(f, l); }
      r->finalize();
      s->calc.pacbio.push_back(std::make_pair(spec.scfg, r));
      s->sets.push_back(std::make_pair(2, (int)s->calc.pacbio.size() - 1));
    }
  }
  s->calc.paired_state.resize(s->calc.paired.size());
  return (int)s->sets.size();
}
// name order of the read sets as the config loader iterates them (for tests of the hash-order quirk)
int orc_config_order(const char* cfg_file, char* out, int cap) {
  KV global; std::unordered_map<std::string, KV> sets;
  if (!load_config(cfg_file, global, sets)) return -1;
  std::string joined;
  for (auto& spec : readsets_from_config(sets)) { joined += spec.name; joined += ","; }
  if ((int)joined.size() < cap) memcpy(out, joined.c_str(), joined.size() + 1);
  return (int)joined.size();
}
// parsed paired config values: out = {penalty_constant, step, insert_mean, insert_std, min_prob_per_base, min_prob_start, weight, mismatch}
int orc_config_paired_values(const char* cfg_file, const char* name, double* out8) {
  KV global; std::unordered_map<std::string, KV> sets;
  if (!load_config(cfg_file, global, sets)) return -1;
  for (auto& spec : readsets_from_config(sets))
    if (spec.name == name && spec.type == "paired") {
      out8[0] = spec.pcfg.penalty_constant; out8[1] = spec.pcfg.step; out8[2] = spec.pcfg.insert_mean; out8[3] = spec.pcfg.insert_std;
      out8[4] = spec.pcfg.min_prob_per_base; out8[5] = spec.pcfg.min_prob_start; out8[6] = spec.pcfg.weight; out8[7] = spec.mismatch;
      return 0;
    }
  return -2;
}

}  // extern "C"
