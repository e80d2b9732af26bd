// kernels.hip.h -- hand-written gfx950 kernels of the GAML likelihood path.
//
// All three scorers are HBM/L2-streaming reductions (no MFMA: SURVEY.md 8d): one lane owns one
// read (pair), loads its 16-B alignment records with coalesced dwordx4 loads from the READ-MAJOR
// record table, resolves each record's window against the per-evaluation window-occurrence table
// (a few KB..MB, L2 resident), accumulates the read's probability in f64, applies the reference's
// per-read floor + log and reduces (sum of logs, floored reads) wave -> block -> grid inside the
// same launch. f64 throughout, compiled with -ffp-contract=off so that products and sums round
// exactly like the reference's scalar C++ (the per-read probabilities are bit-identical to the
// oracle's; only log() and the order of the final sum differ).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gaml {

constexpr int kBlock = 256;      // 4 waves of 64
constexpr int kMaxBlocks = 2048; // 256 CUs x 8 blocks, grid-stride beyond (guide G11)
// device counters of a paired set's delta store (delta_dev.hip.h): delta pairs, long lists, used spill records per mate,
// overflow flag, sequence number of the last maintenance launch
enum { kDsDirty = 0, kDsSpill = 1, kDsTop0 = 2, kDsTop1 = 3, kDsOverflow = 4, kDsSeq = 5, kDsInts = 8 };

// Occ12 (host_model.h): {lo, hi, rank}, 4-byte aligned: one dwordx3 load (register classes) or a dwordx2 of the first two words (class 0)
__device__ __forceinline__ unsigned long long occ8_of(const Occ12* t, unsigned w) { const Occ12* e = t + w; return (unsigned long long)e->lo | ((unsigned long long)e->hi << 32); }

struct MateView {
  const int4* first;      // per read: {wid | -1, pos, edit | orient<<8 | extra_count<<9, extra_start}
  const int4* extra;      // further records of reads that have more than one
  const int4* occ;        // per window: {shift, min_pos, path | -1, rank | -(list+1)} -- single-end sets; null for paired sets, which use
  const Occ12* occ12;     //   {occ8 = shift:32 | min_pos:16 | path:15 | general:1 (all ones: does not occur), rank or -(list+1)}: 12 bytes
  const int* multi_off;   // windows occurring several times: list bounds
  const int4* multi;      //   ... and entries
  const double* mism_pow; // mismatch_prob^e
  const double* match_pow;// match_prob^k
};

struct PairedArgs {
  // ---- what the blocks of the static part of the compact class read (paired_score_kernel's lean entry), together at
  // the front: two wide scalar fetches, one wait, and the first records can be requested
  int total_blocks, blocks0a, n0a, n_codes;   // grid size; static part: blocks [0, blocks0a), slots [0, n0a); length combinations (<= 256)
  int blocks0;               // blocks [0, blocks0): compact path; [blocks0, blocks01): <= 2 records; [blocks01, blocks012): <= 4;
  int blocks01, blocks012;   // [blocks012, main_blocks): delta pairs (lane per pair)
  int main_blocks;           // partial slots [0, main_blocks) main, then the wave-per-pair blocks (total_blocks in all)
  const double2* memo;                        // (documented below)
  uint32_t* cov_bits;                         // coverage marks (only when penalty_constant > 0), else null
  const unsigned long long* rec8[2];          // class 0 (compact): slots [0, n0): 8-byte records
  const double2* static_val;                  // [n0a] the static pairs' memo entries {t, log t} (static_values_kernel)
  const Occ12* occ12[2];                      // = m[mt].occ12
  double* probs;                              // out: per-pair summed probability (ScoringState::probs)
  double* part_sum;                           // per-block partials: main kernel blocks, then overflow kernel blocks
  int* part_zero;
  double log_two_T;                           // log(2T), host libm, per call
  double tfloor0, logfloor0;                  // tfloor_c[0] / logfloor_c[0] by value (sets with one length combination)
  uint32_t len_combo0; int pad0_;             // len_combo[0] by value
  // ---- everything else
  MateView m[2];
  const uint32_t* len12;     // L1 | L2<<16 per pair
  const double* ins_tab;     // insert-size Gaussian (graph.cc:1593-1598, 1801-1804) for d in [0, ins_n); the
  int ins_n;                 // host extends the table until the f64 value is exactly 0.0, so beyond it: 0.0
  const double* floor_tab;   // exp(c + k*s), s = L1+L2 (graph.cc:1506-1507), host libm
  const double* logfloor_tab;// log(floor_tab[s]), host libm
  const double* covthr_tab;  // exp(c + k*2*L2) indexed by L2 (graph.cc:1855-1857 quirk), or null
  double two_T;              // 2 * total_len as double (graph.cc:1505)
  int n;                     // pairs in this shard
  const int* path_base;      // bit offset of each path in cov_bits
  // class 0 (compact): slots [0, n0): 8-byte records, 1-byte length code, 8-byte occurrence entries
  const unsigned char* len_code;
  const uint32_t* len_combo;
  // per length-combination tables of the compact path (host libm, indexed by len_code):
  const double* pe[2];       // [code*64 + e] = mismatch^e * match^(L-e)  (the product of graph.cc:1859-1863)
  const double* floor_c;     // [code] exp(c + k (L1+L2)); logfloor_c = log of it; covthr_c = exp(c + k 2 L2)
  const double* logfloor_c;
  const double* covthr_c;
  // memoised term + log of single-term pairs: the value of a pair with one alignment per mate depends only on
  // (length code, edit 1, edit 2, insert distance) -- a few 10^4 combinations. logterm_kernel tabulates
  // memo[((code*7 + e1)*7 + e2)*ins_n + dist] = {pair term t, log(t)} once per table build: nothing in it depends on
  // the path set. GetTotalProb's per-read step (graph.cc:1504-1513: p = t / 2T; p < floor ? floored : log(p)) then is
  //   floored  <=>  t < tfloor_c[code]   (tfloor = the smallest double whose quotient by 2T reaches the floor, found on the
  //                                       host with the same division: the SAME decision as the reference, exactly)
  //   log(p)    =   log(t) - log(2T)     (within an ulp or two of log(fl(t / 2T)): ~1e-16 relative on the mean)
  // so a change of total_len costs two doubles per length code per call instead of a table rebuild. null: off
  int lt_codes;              // memo: codes covered (< lt_codes), edits < 7
  const double* tfloor_c;    // [code] per call: see above
  const int4* inl[2];        // inline records of the register classes: [2t + k] (class 1), [2 n1 + 4 t2 + k] (class 2)
  int n0;                    // first[] / extra[] / len12[] hold slots >= n0, indexed slot - n0
  // class 0 in two parts (PairTables::n0a): slots [0, n0a) carry a memo index that does not depend on the path set
  // (static_idx, both records in one window), blocks [0, blocks0a); slots [n0a, n0) are resolved per call, blocks [blocks0a, blocks0)
  const int* static_idx;
  int n01;                   // slots [n0, n01): class 1 (<= 2 records per mate); [n01, n_main): class 2 (<= 4)
  int n_main;                // the lane-per-pair paths score slots [0, n_main)
  // (Pairs of the table classes with a record in a window that occurs several times -- or whose occurrence does not fit the
  // compact forms -- need the general loop: the GEN instantiation of the kernel scores them where it meets them, through
  // compact_general_call / general_pair_call.)
  unsigned long long* timeline;  // TL instantiation only: 8 wall-clock stamps (10 ns units) per wave of the grid
  // Delta: pairs whose record lists changed since the device tables were built (newly activated
  // windows). Their slots carry a DIRTY mark in the tables; their complete record lists travel with
  // every evaluation and the overflow path scores them.
  const int* dstate;          // the delta store's device counters (kDs*): [kDsDirty] delta pairs, [kDsSpill] long lists -- the lists are
                              // maintained by kernels (delta_dev.hip.h), the host only knows upper bounds when it sizes the grid
  const int* dirty_slots;     // [n_dirty] slot of the pair in the tables
  const int4* dirty_recs[2];  // [4 * n_dirty] per mate: the pair's complete record list {wid, pos, edit | orient<<8, 0}, padded with wid = -1
  const int* dirty_spill;     // [n_dirty] -1, or the pair's index in the spill lists (more than 4 records on a mate)
  const int2* spill_rng[2];   // long lists: {begin, count} in spill_recs per spill index (a list that changes is written anew behind the others)
  const int4* spill_recs[2];
  const int* spill_slot;      // [n_spill] slot of the pair (scored one WAVE per pair, behind the table pairs with long lists)
  unsigned* ticket;          // zero before first launch; the last block resets it
  double* out;               // the read set's 4 partials {sum of logs, floored reads, bad_bases, reads}
  double n_reads;
  double* status_out;        // sharded evaluations (stream-ordered, ticket finish): two status words written with the partials, or null
  double status_a, status_b;
};

struct Cand { int path, pos, edit, orient, rank, k; bool valid; };

// a window's occurrence entry in the 16-byte form from the paired sets' 8-byte entry + rank
__device__ __forceinline__ int4 occ_from_compact(unsigned long long o, int rk) {
  const int4 e = make_int4((int)(unsigned)o, (int)(short)(o >> 32), (int)((o >> 48) & 0x7fff), rk);
  return o == ~0ull ? make_int4(0, 0, -1, 0) : e;
}
// ... from whichever image the read set has (uniform branch: for the loops that are not latency-critical; the
// register classes read the compact image directly, their loads must not sit behind branches)
__device__ __forceinline__ int4 mate_occ(const MateView& v, int w) {
  if (v.occ) return v.occ[w];
  const Occ12 e = v.occ12[w];
  return occ_from_compact((unsigned long long)e.lo | ((unsigned long long)e.hi << 32), e.rank);
}

__device__ __forceinline__ Cand make_cand(const int4& r, const int4& o, int k) {
  Cand c;
  c.path = o.z; c.pos = r.y + o.x; c.edit = r.z & 0xff; c.orient = (r.z >> 8) & 1;
  c.rank = o.w; c.k = k; c.valid = r.y >= o.y;
  return c;
}

// where a read's records come from: the record tables (first record + link to the rest) ...
struct TableSrc {
  const MateView* v;
  int4 r0;
  __device__ __forceinline__ int count() const { return r0.x < 0 ? 0 : 1 + (int)((unsigned)r0.z >> 9); }
  __device__ __forceinline__ int4 get(int k) const { return k == 0 ? r0 : v->extra[r0.w + k - 1]; }
};
// ... or an explicit list (delta pairs)
struct ListSrc {
  const int4* recs;
  int cnt;
  __device__ __forceinline__ int count() const { return cnt; }
  __device__ __forceinline__ int4 get(int k) const { return recs[k]; }
};

// visit every (record, occurrence) combination of one read of one mate, in (record, occurrence) order
template <class Src, class F>
__device__ __forceinline__ void for_each_cand_src(const MateView& v, const Src& src, F f) {
  const int cnt = src.count();
  for (int k = 0; k < cnt; k++) {
    int4 r = src.get(k);
    if (r.x < 0) continue;
    int4 o = mate_occ(v, r.x);
    if (o.z < 0) continue;  // window not part of the scored paths
    if (o.w >= 0) {
      f(make_cand(r, o, k));
    } else {
      const int s = -o.w - 1;
      for (int q = v.multi_off[s]; q < v.multi_off[s + 1]; q++) f(make_cand(r, v.multi[q], k));
    }
  }
}
template <class F>
__device__ __forceinline__ void for_each_cand(const MateView& v, const int4& r0, F f) {
  for_each_cand_src(v, TableSrc{&v, r0}, f);
}

// The reference keeps, per path and read, one alignment per path position: a later record at the
// same position overwrites the earlier one (graph.cc:583-592). "Later" = visited later = larger
// (occurrence rank, record index). A candidate is live iff it is valid and no later valid
// candidate of the same read/mate shares its (path, position).
template <class Src>
__device__ __forceinline__ bool is_live_src(const MateView& v, const Src& src, const Cand& c) {
  if (!c.valid) return false;
  bool live = true;
  for_each_cand_src(v, src, [&](const Cand& d) {
    if (d.valid && d.path == c.path && d.pos == c.pos && (d.rank > c.rank || (d.rank == c.rank && d.k > c.k))) live = false;
  });
  return live;
}
__device__ __forceinline__ bool is_live(const MateView& v, const int4& r0, const Cand& c) {
  return is_live_src(v, TableSrc{&v, r0}, c);
}

__device__ __forceinline__ void mark_bit(uint32_t* bits, int bit) {
  atomicOr(&bits[bit >> 5], 1u << (bit & 31));
}

// one pair term (graph.cc:1858-1889); returns 0 when the orientation rule rejects the pair
__device__ __forceinline__ double pair_term(const PairedArgs& a, const Cand& x, const Cand& y, int L1, int L2) {
  if (x.orient == y.orient) return 0.0;
  int dist;
  if (x.pos < y.pos) {
    if (x.orient != 0 || y.orient != 1) return 0.0;
    dist = y.pos - x.pos + L2;
  } else {
    if (x.orient != 1 || y.orient != 0) return 0.0;
    dist = x.pos - y.pos + L1;
  }
  // beyond the insert-size table the Gaussian is exactly 0.0 in f64 (PairedArgs::ins_tab), and so is the term: two finite
  // products times 0.0 -- no need to fetch the factors (two occurrences of a repeat are thousands of bases apart: most
  // combinations of a pair in a 5-copy repeat end here)
  if ((unsigned)dist >= (unsigned)a.ins_n) return 0.0;
  double p1 = a.m[0].mism_pow[x.edit] * a.m[0].match_pow[L1 - x.edit];
  double p2 = a.m[1].mism_pow[y.edit] * a.m[1].match_pow[L2 - y.edit];
  double ip = a.ins_tab[dist];
  double t = p1 * p2 * ip;
  if (a.cov_bits && t > a.covthr_tab[L2]) {  // coverage events at both ends (use_all_to_cov, graph.cc:1883-1888)
    int base = a.path_base[x.path];
    mark_bit(a.cov_bits, base + max(x.pos, y.pos));
    mark_bit(a.cov_bits, base + min(x.pos, y.pos));
  }
  return t;
}

// reads with several records and/or windows that occur several times: fully general, one
// thread, quadratic; only the last resort of the overflow kernel (more than kOvfCap candidates)
template <class Src>
__device__ __forceinline__ double paired_general_src_slow(const PairedArgs& a, const Src& s1, const Src& s2, int L1, int L2) {
  double acc = 0.0;
  for_each_cand_src(a.m[0], s1, [&](const Cand& x) {
    if (!is_live_src(a.m[0], s1, x)) return;
    for_each_cand_src(a.m[1], s2, [&](const Cand& y) {
      if (y.path != x.path) return;
      if (!is_live_src(a.m[1], s2, y)) return;
      acc += pair_term(a, x, y, L1, L2);
    });
  });
  return acc;
}
// Which candidates of one mate are live, as a bit per candidate in visiting order; false when there are more than 64.
// One pass per candidate over the candidates (n^2 derivations), ONCE per mate -- the loop above re-derives the liveness
// of every candidate of mate 2 for every candidate of mate 1 (n1 n2^2 derivations, each a chain of dependent loads: a
// read in a 5-copy repeat seen through 4 windows took 8,000 of them, and its lane 50 us).
template <class Src>
__device__ __forceinline__ bool live_mask_src(const MateView& v, const Src& src, unsigned long long& mask) {
  mask = 0;
  int n = 0;
  for_each_cand_src(v, src, [&](const Cand& c) {
    if (n < 64 && is_live_src(v, src, c)) mask |= 1ull << n;
    n++;
  });
  return n <= 64;
}
// Same terms in the same order as the loop above (general_pair_call's fallback for a pair with more candidates than it stages).
template <class Src>
__device__ __forceinline__ double paired_general_src_masks(const PairedArgs& a, const Src& s1, const Src& s2, int L1, int L2) {
  unsigned long long live1, live2;
  if (!live_mask_src(a.m[0], s1, live1) || !live_mask_src(a.m[1], s2, live2)) return paired_general_src_slow(a, s1, s2, L1, L2);
  double acc = 0.0;
  if (!live1 || !live2) return acc;
  int ix = 0;
  for_each_cand_src(a.m[0], s1, [&](const Cand& x) {
    const bool lx = (live1 >> ix) & 1ull;
    ix++;
    if (!lx) return;
    int iy = 0;
    for_each_cand_src(a.m[1], s2, [&](const Cand& y) {
      const bool ly = (live2 >> iy) & 1ull;
      iy++;
      if (!ly || y.path != x.path) return;
      acc += pair_term(a, x, y, L1, L2);
    });
  });
  return acc;
}
template <class Src>
__device__ __forceinline__ double paired_general_src(const PairedArgs& a, const Src& s1, const Src& s2, int L1, int L2) {
  return paired_general_src_slow(a, s1, s2, L1, L2);
}
__device__ __forceinline__ double paired_general(const PairedArgs& a, const int4& r1, const int4& r2, int L1, int L2) {
  return paired_general_src_masks(a, TableSrc{&a.m[0], r1}, TableSrc{&a.m[1], r2}, L1, L2);
}

// wave (64 lanes) + block reduction of (double, int); result valid in thread 0
__device__ __forceinline__ void block_reduce(double& s, int& z, double* sh_s, int* sh_z) {
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off, 64);
    z += __shfl_down(z, off, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sh_s[wave] = s; sh_z[wave] = z; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0; int tz = 0;
    for (int w = 0; w < kBlock / 64; w++) { ts += sh_s[w]; tz += sh_z[w]; }
    s = ts; z = tz;
  }
}

// grid-level finish: every block publishes its partial; the last block to arrive sums
// n_partials partials in index order (deterministic), writes out[0..1] and resets the ticket.
constexpr int kTicketWords = 17;  // [0] groups done, [1 + g] blocks of group g (= slot mod 16) done
__device__ __forceinline__ void grid_finish(double s, int z, int my_slot, int n_partials, double* part_sum, int* part_zero,
                                            unsigned* ticket, double* out, double bad_bases, double n_reads, double* sh_s, int* sh_z,
                                            double* status_out = nullptr, double status_a = 0.0, double status_b = 0.0) {
  // n_partials = number of blocks (of ALL kernels sharing this ticket) = number of partial slots
  // Hand-off without a release fence: a full __threadfence() here writes back AND invalidates the
  // XCD's L2 once per block, which evicts the shared tables for every block still streaming. The
  // partials are stored write-through at agent scope (sc1), drained with vmcnt(0), then the
  // ticket is taken with a relaxed agent-scope add; the last block reads them back with
  // agent-scope loads (guide section 6, Guideline 16, "sc1 slab stores").
  // Tickets in two levels: a thousand blocks' atomics on ONE word took ~20 us (they are served one after the other); a block
  // draws from its group's word (slot mod 16), the block that completes a group draws from the top word -- ~60 atomics per
  // word. The block that completes the last group sums the partials in index order (deterministic) and resets the words.
  __shared__ bool is_last;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&part_sum[my_slot], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&part_zero[my_slot], z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int g = my_slot & 15;
    const unsigned in_group = (unsigned)((n_partials - g + 15) >> 4), groups = (unsigned)min(16, n_partials);
    bool last = false;
    if (__hip_atomic_fetch_add(&ticket[1 + g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_group - 1)
      last = __hip_atomic_fetch_add(&ticket[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1;
    is_last = last;
  }
  __syncthreads();
  if (!is_last) return;
  double ts = 0; int tz = 0;
  for (int b = threadIdx.x; b < n_partials; b += kBlock) {
    ts += __hip_atomic_load(&part_sum[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tz += __hip_atomic_load(&part_zero[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  block_reduce(ts, tz, sh_s, sh_z);
  if (threadIdx.x < kTicketWords) __hip_atomic_store(&ticket[threadIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) {
    // the read set's 4 partials {sum of logs, floored reads, bad_bases, reads}: ready for one D2H / all-reduce
    out[0] = ts;
    out[1] = (double)tz;
    if (bad_bases >= 0) out[2] = bad_bases;  // < 0: a later kernel (coverage sweep) fills it
    out[3] = n_reads;
    if (status_out) { status_out[0] = status_a; status_out[1] = status_b; }  // (a sharded evaluation's status words ride along: no dispatch of their own)
  }
}

// What a pair contributes before the path set's 2T and floor threshold enter: a batch of path sets resolves a pair
// once and finishes it per set (finish_val) wherever the sets' tables agree on the pair's windows.
// kind 0: nothing here (scored elsewhere, or not at all); 1: a memo entry {t, log t}, key = length code;
// 2: class 0 outside the memo, key = length code; 3: table classes, key = L1 + L2
struct PairVal { double t, logt; int key, kind; };

// per-read floor + log (GetTotalProb graph.cc:1504-1513)
__device__ __forceinline__ void finish_read(const PairedArgs& a, int i, double acc, int L1, int L2, double& lsum, int& zeros, PairVal* cap = nullptr) {
  if (cap) *cap = PairVal{acc, 0.0, L1 + L2, 3};
  a.probs[i] = acc;
  const double p = acc / a.two_T;
  const int s = L1 + L2;
  if (p < a.floor_tab[s]) { zeros++; lsum += a.logfloor_tab[s]; }
  else lsum += log(p);
}

// up to K records per mate held in registers (static indexing only)
template <int K>
struct RegCands {
  int path[K], pos[K], ef[K], rank[K];
  bool valid[K], live[K];
};

// K inline records -> register candidates incl. the overwrite rule (all K loads independent)
template <int K>
__device__ __forceinline__ bool cands_from_records(const MateView& v, const int4 (&r)[K], RegCands<K>& c) {
  int4 o[K];
  bool multi = false;
  Occ12 e[K];
#pragma unroll
  for (int k = 0; k < K; k++) e[k] = v.occ12[r[k].x >= 0 ? r[k].x : 0];  // unconditional (entry 0 always exists): the K loads go out back to back (paired sets only)
#pragma unroll
  for (int k = 0; k < K; k++) o[k] = occ_from_compact((unsigned long long)e[k].lo | ((unsigned long long)e[k].hi << 32), e[k].rank);
#pragma unroll
  for (int k = 0; k < K; k++) if (r[k].x < 0) o[k] = make_int4(0, 0, -1, 0);
#pragma unroll
  for (int k = 0; k < K; k++) {
    multi |= (o[k].z >= 0 && o[k].w < 0);
    c.path[k] = o[k].z; c.pos[k] = r[k].y + o[k].x; c.ef[k] = r[k].z & 0x1ff; c.rank[k] = o[k].w;
    c.valid[k] = r[k].x >= 0 && o[k].z >= 0 && r[k].y >= o[k].y;
  }
#pragma unroll
  for (int i = 0; i < K; i++) {
    bool lv = c.valid[i];
#pragma unroll
    for (int j = 0; j < K; j++)
      if (j != i) lv = lv && !(c.valid[j] && c.path[j] == c.path[i] && c.pos[j] == c.pos[i] &&
                               (c.rank[j] > c.rank[i] || (c.rank[j] == c.rank[i] && j > i)));
    c.live[i] = lv;
  }
  return multi;
}
template <int K>
__device__ __forceinline__ bool load_cands_inline(const MateView& v, const int4* rec, RegCands<K>& c) {
  int4 r[K];
#pragma unroll
  for (int k = 0; k < K; k++) r[k] = rec[k];
  return cands_from_records<K>(v, r, c);
}

template <int K>
__device__ __forceinline__ double score_regs(const PairedArgs& a, const RegCands<K>& c1, const RegCands<K>& c2, int L1, int L2) {
  // common case: the overwrite rule leaves one alignment per mate (junction duplicates) -> one
  // pair term, selected without branches; the general double loop only otherwise
  int n1 = 0, n2 = 0;
  Cand x, y;
  x.path = -1; x.pos = 0; x.edit = 0; x.orient = 0; y = x;
#pragma unroll
  for (int i = 0; i < K; i++) {
    if (c1.live[i]) { n1++; x.path = c1.path[i]; x.pos = c1.pos[i]; x.edit = c1.ef[i] & 0xff; x.orient = c1.ef[i] >> 8; }
    if (c2.live[i]) { n2++; y.path = c2.path[i]; y.pos = c2.pos[i]; y.edit = c2.ef[i] & 0xff; y.orient = c2.ef[i] >> 8; }
  }
  if (n1 == 0 || n2 == 0) return 0.0;
  if (n1 == 1 && n2 == 1) return x.path == y.path ? pair_term(a, x, y, L1, L2) : 0.0;
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < K; i++) {
#pragma unroll
    for (int j = 0; j < K; j++) {
      if (c1.live[i] && c2.live[j] && c1.path[i] == c2.path[j]) {
        Cand x, y;
        x.path = c1.path[i]; x.pos = c1.pos[i]; x.edit = c1.ef[i] & 0xff; x.orient = c1.ef[i] >> 8;
        y.path = c2.path[j]; y.pos = c2.pos[j]; y.edit = c2.ef[j] & 0xff; y.orient = c2.ef[j] >> 8;
        acc += pair_term(a, x, y, L1, L2);
      }
    }
  }
  return acc;
}

// Main kernel: one lane per read pair, pairs pre-sorted by record-count class so that a wave is
// homogeneous; covers the pairs with at most 4 records per mate (slots [0, n_main)). A pair that
// touches a window occurring several times in the path set (any record, either mate, whose window entry
// carries the general flag / rank < 0) takes the general path (GEN instantiation: compact_general_call, general_pair_call).
constexpr unsigned long long kNone8 = ~0ull;
constexpr unsigned long long kDirty8 = ~0ull - 1;  // class-0 slot whose records moved to the delta lists
constexpr int kDirtyWid = -2;                      // same mark in the 16-byte tables

__device__ __forceinline__ int4 rec8_to_quad(unsigned long long r) {  // 8-byte record -> the 16-byte form
  if (r == kNone8 || r == kDirty8) return make_int4(r == kDirty8 ? kDirtyWid : -1, 0, 0, 0);
  return make_int4((int)(r & 0xffffff), (int)((r >> 24) & 0xfffffff), (int)((r >> 52) & 63) | ((int)((r >> 58) & 1) << 8), 0);
}

// Class 0: at most one record per mate, packed to 8 bytes; pairs are ordered by window id, so the
// occurrence entries of a wave's lanes are mostly the same address (broadcast).
struct Compact1 {  // one class-0 pair in flight
  unsigned long long r1, r2, o1, o2;
  int L1, L2, lc;
};

// one thread per (length code, edit 1, edit 2, distance): the pair term (graph.cc:1858-1882) for every value a
// single-term pair can take, and its log. Same products in the same order as the per-pair path, so the looked-up
// terms are the per-pair terms, bit for bit.
__global__ __launch_bounds__(kBlock) void logterm_kernel(const double* pe0, const double* pe1, const double* ins_tab, int ins_n, int codes,
                                                        double2* memo) {
  const int total = codes * 49 * ins_n;
  for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < total; idx += gridDim.x * kBlock) {
    const int dist = idx % ins_n;
    const int q = idx / ins_n;
    const int e2 = q % 7, e1 = (q / 7) % 7, code = q / 49;
    const double t = pe0[code * 64 + e1] * pe1[code * 64 + e2] * ins_tab[dist];  // as compact_term_tables
    memo[idx] = make_double2(t, log(t));  // t = 0 (distance beyond the Gaussian's f64 range): -inf, never used (0 < tfloor: floored)
  }
}

__device__ __forceinline__ void finish_read_compact(const PairedArgs& a, int i, double acc, int lc, double& lsum, int& zeros, PairVal* cap = nullptr) {
  if (cap) *cap = PairVal{acc, 0.0, lc, 2};
  a.probs[i] = acc;
  if (acc == 0.0 && a.floor_c[lc] > 0.0) { zeros++; lsum += a.logfloor_c[lc]; return; }  // 0 / 2T < floor
  const double p = acc / a.two_T;
  if (p < a.floor_c[lc]) { zeros++; lsum += a.logfloor_c[lc]; }
  else lsum += log(p);
}

// Everything about a class-0 pair that needs no table: does it score at all (both mates occur on
// the same path, position filter graph.cc:577, orientation rule graph.cc:1864-1876), the
// candidates, the insert distance and the memo index (-1: not covered by the memo).
struct CompactPrep {
  Cand x, y;
  int dist, memo_idx;
  bool skip, scores;
};
__device__ __forceinline__ void compact_prep(const PairedArgs& a, const Compact1& c, CompactPrep& q) {
  q.memo_idx = -1; q.dist = -1; q.scores = false;
  // a record in a window that needs the general path (occurs several times, ...): compact_general
  q.skip = (c.o1 != kNone8 && (c.o1 >> 63)) || (c.o2 != kNone8 && (c.o2 >> 63));
  if (q.skip || c.o1 == kNone8 || c.o2 == kNone8 || ((c.o1 ^ c.o2) >> 48) != 0) return;  // both occur, same path
  const int p1 = (int)((c.r1 >> 24) & 0xfffffff), p2 = (int)((c.r2 >> 24) & 0xfffffff);
  if (p1 < (int)(short)(c.o1 >> 32) || p2 < (int)(short)(c.o2 >> 32)) return;  // position filter (graph.cc:577)
  Cand& x = q.x; Cand& y = q.y;
  x.path = (int)(c.o1 >> 48); x.pos = p1 + (int)(unsigned)c.o1; x.edit = (int)((c.r1 >> 52) & 63); x.orient = (int)((c.r1 >> 58) & 1);
  y.path = x.path; y.pos = p2 + (int)(unsigned)c.o2; y.edit = (int)((c.r2 >> 52) & 63); y.orient = (int)((c.r2 >> 58) & 1);
  if (x.orient == y.orient) return;
  if (x.pos < y.pos) {
    if (x.orient != 0 || y.orient != 1) return;
    q.dist = y.pos - x.pos + c.L2;
  } else {
    if (x.orient != 1 || y.orient != 0) return;
    q.dist = x.pos - y.pos + c.L1;
  }
  q.scores = true;
  if (a.memo && q.dist >= 0 && q.dist < a.ins_n && c.lc < a.lt_codes && x.edit < 7 && y.edit < 7)
    q.memo_idx = ((c.lc * 7 + x.edit) * 7 + y.edit) * a.ins_n + q.dist;
}

// the pair term when the memo does not cover it: same arithmetic as pair_term, the per-read error
// probability comes from the per-length-combination product table
__device__ __forceinline__ double compact_term_tables(const PairedArgs& a, const Compact1& c, const CompactPrep& q) {
  const double p1 = a.pe[0][c.lc * 64 + q.x.edit];
  const double p2 = a.pe[1][c.lc * 64 + q.y.edit];
  const double ip = (unsigned)q.dist < (unsigned)a.ins_n ? a.ins_tab[q.dist] : 0.0;
  return p1 * p2 * ip;
}

__device__ __forceinline__ void compact_cover(const PairedArgs& a, const Compact1& c, const CompactPrep& q, double t) {
  if (a.cov_bits && t > a.covthr_c[c.lc]) {  // coverage events at both ends (use_all_to_cov, graph.cc:1883-1888)
    const int base = a.path_base[q.x.path];
    mark_bit(a.cov_bits, base + max(q.x.pos, q.y.pos));
    mark_bit(a.cov_bits, base + min(q.x.pos, q.y.pos));
  }
}

// A class-0 pair with a record in a window that occurs several times in this path set (or whose occurrence
// does not fit the 8-byte form): fully general loop over (record, occurrence) candidates, one lane per pair. Inlined into
// the streaming loop this cost the loop 1.2 us of 12 at cfg3 even when no such pair exists (registers / code size): it
// exists in the GEN instantiation only, and there as a function (compact_general_call).
// One record per mate: every valid candidate is live (the overwrite rule only ever decides between DIFFERENT records of a
// read that land on the same path position), so the pair's probability is the plain double sum over the two windows'
// occurrences -- each window's list bounds found once, its entries read in order. Terms and their order are those of
// paired_general.
struct OccList { const int4* e; int n; int4 one; };
__device__ __forceinline__ OccList occ_list(const MateView& v, int wid) {
  OccList l{nullptr, 0, make_int4(0, 0, -1, 0)};
  if (wid < 0) return l;
  const int4 o = mate_occ(v, wid);
  if (o.z < 0) return l;
  if (o.w >= 0) { l.one = o; l.n = 1; return l; }
  const int s = -o.w - 1, b = v.multi_off[s];
  l.e = v.multi + b; l.n = v.multi_off[s + 1] - b;
  return l;
}
__device__ __forceinline__ void compact_general(const PairedArgs& a, int i, double& lsum, int& zeros) {
  const int lc = a.len_code[i];
  const uint32_t l12 = a.len_combo[lc];
  const int L1 = l12 & 0xffff, L2 = l12 >> 16;
  const int4 r1 = rec8_to_quad(a.rec8[0][i]), r2 = rec8_to_quad(a.rec8[1][i]);
  const OccList l1 = occ_list(a.m[0], r1.x), l2 = occ_list(a.m[1], r2.x);
  double acc = 0.0;
  constexpr int K = 8;
  if (l1.n <= K && l2.n <= K) {
    // both lists in registers: 2 K independent loads (one round trip) instead of a load-use chain per combination
    int4 e1[K], e2[K];
#pragma unroll
    for (int p = 0; p < K; p++) {
      e1[p] = l1.e ? l1.e[p < l1.n ? p : 0] : l1.one;
      e2[p] = l2.e ? l2.e[p < l2.n ? p : 0] : l2.one;
    }
#pragma unroll
    for (int p = 0; p < K; p++) {
      const Cand x = make_cand(r1, e1[p], 0);
      if (p >= l1.n || !x.valid) continue;
#pragma unroll
      for (int q = 0; q < K; q++) {
        const Cand y = make_cand(r2, e2[q], 0);
        if (q >= l2.n || !y.valid || y.path != x.path) continue;
        acc += pair_term(a, x, y, L1, L2);
      }
    }
  } else {
    for (int p = 0; p < l1.n; p++) {
      const Cand x = make_cand(r1, l1.e[p], 0);
      if (!x.valid) continue;
      for (int q = 0; q < l2.n; q++) {
        const Cand y = make_cand(r2, l2.e[q], 0);
        if (!y.valid || y.path != x.path) continue;
        acc += pair_term(a, x, y, L1, L2);
      }
    }
  }
  finish_read_compact(a, i, acc, lc, lsum, zeros);
}

// The pairs on windows that occur several times are few or clustered, their code is long (compact_general: 64 unrolled
// combinations; general_pair_staged) and was inlined wherever a class meets such a pair: the GEN kernels came to 550 KB
// and 480 KB of code, a wave that took the rare path fetched instructions from memory line by line (40-80 us for a chain of
// 9 us, in-kernel stamps). They are FUNCTIONS now (one copy each, called by every class of both kernels -- which also makes
// a batch's arithmetic the single call's by construction): the callee reads the kernel's argument block itself, through its
// address (the intrinsic that returns it is null outside a kernel: the caller passes it; set >= 0: path set `set` of a
// multi-set launch), and returns what the caller adds to its running sums.
struct GenOut { double add; int zeros; };
__device__ GenOut compact_general_call(unsigned long long kernargs, int i, int set);
// table pair of class 1 / 2 at slot i (dj < 0) or delta pair dj at slot i; lds: room for 2 * kGenCands candidates, or null
__device__ GenOut general_pair_call(unsigned long long kernargs, int i, int dj, int set, int4* lds);
__device__ __forceinline__ unsigned long long kernel_args_address() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
#else
  return 0;
#endif
}

// one scored class-0 pair: per-read probability out, floor / log into the running sums
__device__ __forceinline__ void compact_finish(const PairedArgs& a, int i, const Compact1& c, const CompactPrep& q, double2 m,
                                               double& lsum, int& zeros, PairVal* cap = nullptr) {
  if (q.memo_idx >= 0) {
    const double t = m.x;
    if (cap) *cap = PairVal{t, m.y, c.lc, 1};
    compact_cover(a, c, q, t);
    __builtin_nontemporal_store(t, &a.probs[i]);  // written once, read by nobody on the hot path: keep it out of the caches
    const bool floored = t < a.tfloor_c[c.lc];    // <=> t / 2T < floor (PairedArgs::memo)
    lsum += floored ? a.logfloor_c[c.lc] : m.y - a.log_two_T;
    zeros += (int)floored;
    return;
  }
  double t = 0.0;
  if (q.scores) { t = compact_term_tables(a, c, q); compact_cover(a, c, q, t); }
  finish_read_compact(a, i, t, c.lc, lsum, zeros, cap);
}

// a captured pair under another path set's 2T / thresholds (`a`: that set's view): exactly what the finishers above do
__device__ __forceinline__ void finish_val(const PairedArgs& a, int i, const PairVal& v, bool store, double& lsum, int& zeros) {
  if (v.kind == 0) return;
  if (store) __builtin_nontemporal_store(v.t, &a.probs[i]);
  if (v.kind == 1) {
    const bool floored = v.t < a.tfloor_c[v.key];
    lsum += floored ? a.logfloor_c[v.key] : v.logt - a.log_two_T;
    zeros += (int)floored;
  } else if (v.kind == 2) {
    if (v.t == 0.0 && a.floor_c[v.key] > 0.0) { zeros++; lsum += a.logfloor_c[v.key]; return; }
    const double p = v.t / a.two_T;
    if (p < a.floor_c[v.key]) { zeros++; lsum += a.logfloor_c[v.key]; }
    else lsum += log(p);
  } else {
    const double p = v.t / a.two_T;
    if (p < a.floor_tab[v.key]) { zeros++; lsum += a.logfloor_tab[v.key]; }
    else lsum += log(p);
  }
}

__device__ __forceinline__ void compact_load(const PairedArgs& a, int i, bool ok, Compact1& c) {
  c.r1 = ok ? a.rec8[0][i] : kNone8;
  c.r2 = ok ? a.rec8[1][i] : kNone8;
  c.lc = ok ? a.len_code[i] : 0;
}

#if defined(GAML_HIP_DEV) && defined(GAML_GEN_X)  // timing experiments of the development build (results WRONG), tools/build_variant.sh -DGAML_GEN_X=..: of the pairs on windows that occur several times: 1 static part of class 0, 8 its second part, 2 classes 1 / 2, 4 delta pairs, 16 no wave-per-pair blocks
#define GAML_GEN_OFF(bit) ((GAML_GEN_X) & (bit))
#else
#define GAML_GEN_OFF(bit) 0
#endif
// One part of class 0 and the blocks that score it: slots [lo, hi), `blocks` blocks of which this is number `lb`
struct SlotRange { int lo, hi, lb, blocks; };
__device__ __forceinline__ SlotRange compact_range(const PairedArgs& a, int lb) {
  return lb < a.blocks0a ? SlotRange{0, a.n0a, lb, a.blocks0a} : SlotRange{a.n0a, a.n0, lb - a.blocks0a, a.blocks0 - a.blocks0a};
}

template <bool GEN>
__device__ __forceinline__ void paired_compact_body(const PairedArgs& a, const SlotRange rg, double& lsum, int& zeros) {
  // Class 0 in the general form (coverage marks to set, or no memo): two pairs per lane and iteration, software
  // pipelined -- the record loads of iteration k+1 are issued before iteration k's occurrence lookups and arithmetic.
  const int stride = rg.blocks * kBlock, hi = rg.hi;
  int i0 = rg.lo + rg.lb * kBlock + threadIdx.x;
  if (i0 >= hi) return;
  Compact1 c0, c1;
  compact_load(a, i0, true, c0);
  compact_load(a, i0 + stride, i0 + stride < hi, c1);
  while (true) {
    const int i1 = i0 + stride;
    const bool two = i1 < hi;
    const int j0 = i0 + 2 * stride;
    const bool more = j0 < hi;
    Compact1 n0v, n1v;  // next iteration's records: issued now, consumed after this iteration's work
    compact_load(a, j0, more, n0v);
    compact_load(a, j0 + stride, j0 + stride < hi, n1v);
    const uint32_t l0 = a.len_combo[c0.lc], l1 = a.len_combo[c1.lc];
    c0.L1 = l0 & 0xffff; c0.L2 = l0 >> 16; c1.L1 = l1 & 0xffff; c1.L2 = l1 >> 16;
    const bool d0 = c0.r1 == kDirty8, d1 = c1.r1 == kDirty8;  // scored from the delta lists (paired_delta_body)
    if (d0) { c0.r1 = kNone8; c0.r2 = kNone8; }
    if (d1) { c1.r1 = kNone8; c1.r2 = kNone8; }
    c0.o1 = c0.r1 != kNone8 ? occ8_of(a.occ12[0], (unsigned)(c0.r1 & 0xffffff)) : kNone8;
    c0.o2 = c0.r2 != kNone8 ? occ8_of(a.occ12[1], (unsigned)(c0.r2 & 0xffffff)) : kNone8;
    c1.o1 = c1.r1 != kNone8 ? occ8_of(a.occ12[0], (unsigned)(c1.r1 & 0xffffff)) : kNone8;
    c1.o2 = c1.r2 != kNone8 ? occ8_of(a.occ12[1], (unsigned)(c1.r2 & 0xffffff)) : kNone8;
    CompactPrep q0, q1;
    compact_prep(a, c0, q0);
    compact_prep(a, c1, q1);
    // both memo entries are requested before anything is stored
    const double2 m0 = q0.memo_idx >= 0 ? a.memo[q0.memo_idx] : make_double2(0.0, 0.0);
    const double2 m1 = q1.memo_idx >= 0 ? a.memo[q1.memo_idx] : make_double2(0.0, 0.0);
    if (!d0 && !q0.skip) compact_finish(a, i0, c0, q0, m0, lsum, zeros);
    else if (GEN && q0.skip) { const GenOut o = compact_general_call(kernel_args_address(), i0, -1); lsum += o.add; zeros += o.zeros; }
    if (two && !d1 && !q1.skip) compact_finish(a, i1, c1, q1, m1, lsum, zeros);
    else if (GEN && two && q1.skip) { const GenOut o = compact_general_call(kernel_args_address(), i1, -1); lsum += o.add; zeros += o.zeros; }
    if (!more) break;
    c0 = n0v; c1 = n1v;
    i0 = j0;
  }
}

// Class 0 without coverage marks, memo present: FOUR pairs per lane, stage by stage -- all record loads, then all
// occurrence lookups, then all memo entries, then the stores -- every load unconditional (clamped index), so a
// stage is one round trip per lane whatever the pair count. Between the stages a pair is one word of state:
//   >= 0: memo index; kPairZero - code: scores nothing (no alignment of a mate in this path set, filtered, wrong
//   orientation): probability 0, floored; kPairOther: everything else (window occurring several times, dirty slot,
//   term outside the memo), settled after the stores from the tables again (rare).
// Pairs and their order per lane are those of paired_compact_body (slot = base + k * stride), so both give the
// same sums bit for bit.
constexpr int kPairZero = -1, kPairOther = -(1 << 20);
// compact_prep without branches, on the 32-bit halves of the 8-byte record / occurrence words (the scoring kernel
// is issue-bound at cfg3: the branchy 64-bit form was 156 instructions per pair, a third of a wave's lifetime)
__device__ __forceinline__ int compact_state(const PairedArgs& a, uint2 r1, uint2 r2, uint2 o1, uint2 o2, unsigned lc, uint32_t l12,
                                             bool in_range, bool& skip) {
  const bool v1 = r1.y != ~0u, v2 = r2.y != ~0u;           // a record (not kNone8 / kDirty8)
  const bool w1 = v1 & (o1.y != ~0u), w2 = v2 & (o2.y != ~0u);  // ... whose window occurs in this path set
  const bool here = in_range & !(r1.y == ~0u && r1.x == 0xfffffffeu);  // a pair, and not a dirty slot (those: paired_delta_body)
  skip = here & ((w1 & ((int)o1.y < 0)) | (w2 & ((int)o2.y < 0)));  // a window that needs the general path
  const int L1 = l12 & 0xffff, L2 = l12 >> 16;
  const int p1 = (int)(__funnelshift_r(r1.x, r1.y, 24) & 0xfffffffu), p2 = (int)(__funnelshift_r(r2.x, r2.y, 24) & 0xfffffffu);
  const int x = p1 + (int)o1.x, y = p2 + (int)o2.x;
  const unsigned or1 = (r1.y >> 26) & 1u, or2 = (r2.y >> 26) & 1u;
  const bool fwd = x < y;
  const int dist = fwd ? y - x + L2 : x - y + L1;            // graph.cc:1864-1876
  const bool scores = w1 & w2 & !skip & (((o1.y ^ o2.y) >> 16) == 0)  // both occur, same path
                      & (p1 >= (int)(short)(o1.y & 0xffffu)) & (p2 >= (int)(short)(o2.y & 0xffffu))  // position filter (graph.cc:577)
                      & (or1 != or2) & (or1 == (fwd ? 0u : 1u));
  const unsigned e1 = (r1.y >> 20) & 63u, e2 = (r2.y >> 20) & 63u;
  const bool in_memo = scores & ((unsigned)dist < (unsigned)a.ins_n) & (lc < (unsigned)a.lt_codes) & (e1 < 7u) & (e2 < 7u);
  const int idx = (int)((lc * 7u + e1) * 7u + e2) * a.ins_n + dist;
  int state = in_memo ? idx : ((skip | scores) ? kPairOther : kPairZero - (int)lc);
  if (!here) state = kPairOther - 1;
  return state;
}

// ONE: every pair has the same length combination (n_codes == 1, the usual case): no length-code loads, no LDS tables --
// the combination and its log-floor are two uniform values.
template <bool GEN, bool TL = false, bool ONE = false>
__device__ __forceinline__ void paired_compact4_body(const PairedArgs& a, const SlotRange rg, double& lsum, int& zeros) {
  const unsigned stride = (unsigned)rg.blocks * kBlock, n0 = (unsigned)rg.hi;  // (n0: end of this part's slots)
  unsigned long long* tl = TL ? a.timeline + ((size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 8 : nullptr;
#define GAML_STAMP(slot, dep) if (TL) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if ((threadIdx.x & 63) == 0) tl[slot] = (unsigned long long)wall_clock64() + ((dep) == 0x12345u ? 1 : 0); }
  // 32-bit byte offsets from uniform bases: one address register per load instead of a 64-bit add (tables < 4 GB)
  const char* const rec0 = (const char*)a.rec8[0];
  const char* const rec1 = (const char*)a.rec8[1];
  const char* const occ0 = (const char*)a.occ12[0];
  const char* const occ1 = (const char*)a.occ12[1];
  const char* const memo = (const char*)a.memo;
  char* const probs = (char*)a.probs;
  const uint32_t l12_one = ONE ? a.len_combo[0] : 0u;
  const double logfloor_one = ONE ? a.logfloor_c[0] : 0.0, tfloor_one = ONE ? a.tfloor_c[0] : 0.0, log2T = a.log_two_T;
  for (unsigned base = (unsigned)rg.lo + (unsigned)rg.lb * kBlock + threadIdx.x; base < n0; base += 4 * stride) {
    uint2 r1[4], r2[4], o1[4], o2[4];
    unsigned lc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned ic = base + k * stride < n0 ? base + k * stride : base;
      r1[k] = *(const uint2*)(rec0 + ic * 8u); r2[k] = *(const uint2*)(rec1 + ic * 8u); lc[k] = ONE ? 0u : (unsigned)a.len_code[ic];
    }
    GAML_STAMP(2, r1[0].x ^ r1[1].x ^ r1[2].x ^ r1[3].x ^ r2[0].x ^ r2[3].x)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const Occ12* e1 = (const Occ12*)(occ0 + (r1[k].y != ~0u ? (r1[k].x & 0xffffffu) : 0u) * 12u);  // the entry's first two words
      const Occ12* e2 = (const Occ12*)(occ1 + (r2[k].y != ~0u ? (r2[k].x & 0xffffffu) : 0u) * 12u);
      o1[k] = make_uint2(e1->lo, e1->hi); o2[k] = make_uint2(e2->lo, e2->hi);
    }
    GAML_STAMP(3, o1[0].x ^ o1[1].x ^ o1[2].x ^ o1[3].x ^ o2[0].x ^ o2[3].x)
    int state[4];
    unsigned skip_bits = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      bool skip;
      state[k] = compact_state(a, r1[k], r2[k], o1[k], o2[k], lc[k], ONE ? l12_one : a.len_combo[lc[k]], base + k * stride < n0, skip);
      skip_bits |= (unsigned)skip << k;
    }
    double2 m[4];
#pragma unroll
    for (int k = 0; k < 4; k++) m[k] = *(const double2*)(memo + (unsigned)max(state[k], 0) * 16u);
    GAML_STAMP(4, (unsigned)(__double2loint(m[0].x) ^ __double2loint(m[1].x) ^ __double2loint(m[2].x) ^ __double2loint(m[3].x)))
    bool other = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      double* const out = (double*)(probs + (base + k * stride) * 8u);
      if (state[k] >= 0) {  // as compact_finish
        __builtin_nontemporal_store(m[k].x, out);
        const bool floored = m[k].x < (ONE ? tfloor_one : a.tfloor_c[lc[k]]);  // a memo index was built from this pair's length code
        lsum += floored ? (ONE ? logfloor_one : a.logfloor_c[lc[k]]) : m[k].y - log2T;
        zeros += (int)floored;
      } else if (state[k] > kPairOther) {  // as finish_read_compact(acc = 0)
        __builtin_nontemporal_store(0.0, out);
        zeros++;
        lsum += ONE ? logfloor_one : a.logfloor_c[kPairZero - state[k]];
      } else other |= state[k] == kPairOther && !((skip_bits >> k) & 1u);
    }
    GAML_STAMP(5, 0u)
    if (__any(other)) {
      // scores, but outside the memo (edit count or insert distance beyond the table): from the tables, as paired_compact_body
#pragma unroll 1
      for (int k = 0; k < 4; k++) {
        if (state[k] != kPairOther || ((skip_bits >> k) & 1u)) continue;
        const int i = (int)(base + k * stride);
        Compact1 d;
        compact_load(a, i, true, d);
        d.o1 = d.r1 != kNone8 ? occ8_of(a.occ12[0], (unsigned)(d.r1 & 0xffffff)) : kNone8;
        d.o2 = d.r2 != kNone8 ? occ8_of(a.occ12[1], (unsigned)(d.r2 & 0xffffff)) : kNone8;
        const uint32_t l = a.len_combo[d.lc];
        d.L1 = l & 0xffff; d.L2 = l >> 16;
        CompactPrep q;
        compact_prep(a, d, q);
        compact_finish(a, i, d, q, make_double2(0.0, 0.0), lsum, zeros);
      }
    }
    if (GEN && !GAML_GEN_OFF(8) && __any(skip_bits != 0)) {  // pairs on a window that occurs several times: here, after the round's other pairs
#pragma unroll 1
      for (int k = 0; k < 4; k++)
        if ((skip_bits >> k) & 1u) { const GenOut o = compact_general_call(kernel_args_address(), (int)(base + k * stride), -1); lsum += o.add; zeros += o.zeros; }
    }
  }
#undef GAML_STAMP
}

// The first part of class 0 (slots [0, n0a), PairTables::static_idx): both records of a pair sit in the SAME window, so
// the pair's memo index was computed when the tables were built. Stage by stage like paired_compact4_body, but ONE round
// trip shorter: the records and the index come in together, then the occurrence entries AND the memo entries are requested
// together -- the occurrence entries only decide whether the pair scores at all (both windows occur, once, at the same
// place; position filter graph.cc:577). State between the stages: 1 scores (memo entry), 0 scores nothing (probability 0,
// floored), 2 takes the general path (GEN), 3 nothing here (dirty slot, out of range), 4 the entries of the two
// mates disagree about where the window sits (cannot happen; the block's partial is poisoned). An index < 0 marks a pair
// with a mate that has no alignment: state 0 whatever the tables say.
// Pairs, lanes and the order of additions are those of paired_compact_body over the same range.
#ifndef GAML_LEAN_ENTRY
#define GAML_LEAN_ENTRY 1
#endif
// A fresh, branch-local view of the kernel's argument block: re-read through a pointer the optimiser cannot see through
// (kernel-argument segment, constant address space: scalar loads), so that a class's code fetches the words IT needs where
// it runs instead of every wave fetching all ~150 at kernel entry (see paired_score_kernel).
#if GAML_LEAN_ENTRY && defined(__HIP_DEVICE_COMPILE__)
#define GAML_FRESH_ARGS(name, fallback)                                                                                   \
  const __attribute__((address_space(4))) PairedArgs* name##_p =                                                          \
      (const __attribute__((address_space(4))) PairedArgs*)__builtin_amdgcn_kernarg_segment_ptr();                        \
  asm volatile("" : "+s"(name##_p));                                                                                      \
  const PairedArgs name = *name##_p;
#else
#define GAML_FRESH_ARGS(name, fallback) const PairedArgs& name = fallback;
#endif
#ifndef GAML_STATIC_P
#define GAML_STATIC_P 4
#endif
#ifndef GAML_STATIC_PIPE
#define GAML_STATIC_PIPE 0
#endif
// Timing experiments of the development build (results WRONG where a part is switched off; tools/build_variant.sh -DGAML_STATIC_X=..):
// 1 no occurrence lookups, 2 no stores, 4 no value loads, 8 only the static part of class 0, 16 everything but it. The
// release library is compiled without any of these arms, whatever is passed on its command line.
#if defined(GAML_HIP_DEV) && defined(GAML_STATIC_X)
#define GAML_TIMING_X(bit) ((GAML_STATIC_X) & (bit))
#else
#define GAML_TIMING_X(bit) 0
#endif
// static_val[slot] = the memo entry {t, log t} of a static pair ({0, 0}: a mate without alignment -- floored like any
// term below the threshold), copied out of the memo once per table build: the scoring launch then STREAMS the pair's
// value with its records instead of gathering it. (A gather of 16 bytes per pair out of a 1 MB table moves a 128-byte
// line from L2 per pair: 100 MB of line traffic per launch at BASELINE config 3 for 12 MB of values, ~3 us of the L1s'
// fill bandwidth -- what kept the launch at 9 us whatever was done to its chain of dependent loads.)
__global__ __launch_bounds__(kBlock) void static_values_kernel(const int* static_idx, int n0a, const double2* memo, double2* static_val) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n0a; i += gridDim.x * kBlock) {
    const int mi = static_idx[i];
    static_val[i] = mi >= 0 ? memo[mi] : make_double2(0.0, 0.0);
  }
}

// Rounds of P pairs per lane, software-pipelined when a lane takes several: a round's occurrence entries are requested,
// then the NEXT round's records and values, then the round is finished -- its arithmetic and stores run under the next
// round's loads.
template <bool GEN, bool TL = false, bool ONE = false>
__device__ __forceinline__ void paired_static4_body(const PairedArgs& a, const SlotRange rg, double& lsum, int& zeros) {
  constexpr int P = GAML_STATIC_P;
  const unsigned stride = (unsigned)rg.blocks * kBlock, n0 = (unsigned)rg.hi;
  unsigned long long* tl = TL ? a.timeline + ((size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 8 : nullptr;
#define GAML_STAMP(slot, dep) if (TL && first_round) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if ((threadIdx.x & 63) == 0) tl[slot] = (unsigned long long)wall_clock64() + ((dep) == 0x12345u ? 1 : 0); }
  const char* const rec0 = (const char*)a.rec8[0];
  const char* const rec1 = (const char*)a.rec8[1];
  const char* const sval = (const char*)a.static_val;
  const char* const occ0 = (const char*)a.occ12[0];
  const char* const occ1 = (const char*)a.occ12[1];
  char* const probs = (char*)a.probs;
  const double logfloor_one = ONE ? a.logfloor0 : 0.0, tfloor_one = ONE ? a.tfloor0 : 0.0, log2T = a.log_two_T;
  unsigned base = (unsigned)rg.lo + (unsigned)rg.lb * kBlock + threadIdx.x;
  if (base >= n0) return;
  uint2 r1[P], r2[P];
  double2 m[P];
  unsigned lc[P];
#pragma unroll
  for (int k = 0; k < P; k++) {
    const unsigned ic = base + k * stride < n0 ? base + k * stride : base;
    r1[k] = *(const uint2*)(rec0 + ic * 8u); r2[k] = *(const uint2*)(rec1 + ic * 8u); m[k] = GAML_TIMING_X(4) ? make_double2(1e-9 * (double)ic, -20.0) : *(const double2*)(sval + ic * 16u);
    lc[k] = ONE ? 0u : (unsigned)a.len_code[ic];
  }
  bool first_round = true;
  while (true) {
    GAML_STAMP(2, r1[0].x ^ r2[0].x ^ (unsigned)__double2loint(m[0].x))
    uint2 o1[P], o2[P];
#pragma unroll
    for (int k = 0; k < P; k++) {  // (no record, or a dirty slot: entry 0, ignored; a mask, not a select: no branches here)
      const unsigned w1 = (r1[k].x & 0xffffffu) & (0u - (unsigned)(r1[k].y != ~0u)), w2 = (r2[k].x & 0xffffffu) & (0u - (unsigned)(r2[k].y != ~0u));
      const Occ12* e1 = (const Occ12*)(occ0 + w1 * 12u);
      const Occ12* e2 = (const Occ12*)(occ1 + w2 * 12u);
      o1[k] = make_uint2(e1->lo, e1->hi); o2[k] = make_uint2(e2->lo, e2->hi);
      if (GAML_TIMING_X(1)) { o1[k] = make_uint2(r1[k].x >> 8, 0u); o2[k] = make_uint2(r1[k].x >> 8, 0u); }
    }
    // the next round's records (wave-uniform branch; a lane without a next round asks for its first slot again)
    const unsigned nbase = base + P * stride;
    const bool more = nbase < n0;
    uint2 nr1[P], nr2[P];
    double2 nm[P];
    unsigned nlc[P];
    if (GAML_STATIC_PIPE && __any(more)) {
#pragma unroll
      for (int k = 0; k < P; k++) {
        const unsigned ic = nbase + k * stride < n0 ? nbase + k * stride : base;
        nr1[k] = *(const uint2*)(rec0 + ic * 8u); nr2[k] = *(const uint2*)(rec1 + ic * 8u); nm[k] = *(const double2*)(sval + ic * 16u);
        nlc[k] = ONE ? 0u : (unsigned)a.len_code[ic];
      }
    }
    GAML_STAMP(3, o1[0].x ^ o2[0].x)
    GAML_STAMP(4, 0u)  // (no memo stage here: the values came in with the records)
    // Everything below is straight-line: bit logic on the 32-bit halves, selects instead of branches, every loaded word
    // used unconditionally (a value that is only used inside a branch gets its LOAD moved into that branch by the
    // compiler -- behind a full wait, a round trip of its own; it did that to the fourth pair's records and values).
    unsigned skip_bits = 0;
#pragma unroll
    for (int k = 0; k < P; k++) {
      const bool none1 = r1[k].y == ~0u, none2 = r2[k].y == ~0u;          // no record (or, mate 1, the dirty mark)
      const bool here = (base + k * stride < n0) & !(none1 & (r1[k].x == 0xfffffffeu));  // a pair, and not a dirty slot (those: paired_delta_body)
      const bool w1 = !none1 & (o1[k].y != ~0u), w2 = !none2 & (o2[k].y != ~0u);  // a record whose window occurs in this path set
      // ... several times: general path (as compact_state: also when the other mate has no record -- which block counts the
      // pair decides the last bit of the sum, and batches must agree with calls)
      const bool gen = here & ((w1 & ((int)o1[k].y < 0)) | (w2 & ((int)o2[k].y < 0)));
      const bool same = (o1[k].x == o2[k].x) & (((o1[k].y ^ o2[k].y) >> 16) == 0);      // same shift, same path
      const int p1 = (int)(__funnelshift_r(r1[k].x, r1[k].y, 24) & 0xfffffffu), p2 = (int)(__funnelshift_r(r2[k].x, r2[k].y, 24) & 0xfffffffu);
      const bool kept = (p1 >= (int)(short)(o1[k].y & 0xffffu)) & (p2 >= (int)(short)(o2[k].y & 0xffffu));  // position filter (graph.cc:577)
      const bool both = here & !gen & w1 & w2;          // (a mate without alignment: w false -- scores nothing whatever the tables say)
      const bool scores = both & same & kept;           // the pair's value counts: as compact_finish
      // `both & !same` cannot happen: both mates register the same walks under the same rules, a walk's occurrences depend on
      // the path set alone, and a pair is only here when its two windows are linked (link_mate_windows). No second code path
      // for it (its registers would be every wave's): the block's partial is poisoned instead, and the host reports
      // GAML_HIP_ESTATE rather than a likelihood (combine()).
      const bool poison = both & !same;
      const bool counted = here & !gen;                 // scores, or scores nothing: probability written, floor / log added
      skip_bits |= (unsigned)gen << k;
      const double t = scores ? m[k].x : 0.0;
      const bool floored = counted & (!scores | (t < (ONE ? tfloor_one : a.tfloor_c[lc[k]])));  // (0 < tfloor: as finish_read_compact(acc = 0))
      const double lf = ONE ? logfloor_one : a.logfloor_c[lc[k]];
      double add = floored ? lf : m[k].y - log2T;
      add = counted ? add : 0.0;
      add = poison ? __builtin_nan("") : add;
      lsum += add;                                       // (adding 0.0 changes no bit of a sum of negative logs)
      zeros += (int)floored;
      if (counted && !GAML_TIMING_X(2)) __builtin_nontemporal_store(t, (double*)(probs + (base + k * stride) * 8u));
    }
    if (GEN && !GAML_GEN_OFF(1) && __any(skip_bits != 0)) {  // pairs on a window that occurs several times: here, after the round's other pairs
#pragma unroll 1
      for (int k = 0; k < P; k++)
        if ((skip_bits >> k) & 1u) { const GenOut o = compact_general_call(kernel_args_address(), (int)(base + k * stride), -1); lsum += o.add; zeros += o.zeros; }
    }
    GAML_STAMP(5, 0u)
    first_round = false;
    if (!more) break;
    base = nbase;
    if (GAML_STATIC_PIPE) {
#pragma unroll
      for (int k = 0; k < P; k++) { r1[k] = nr1[k]; r2[k] = nr2[k]; m[k] = nm[k]; lc[k] = nlc[k]; }
    } else {
#pragma unroll
      for (int k = 0; k < P; k++) {
        const unsigned ic = base + k * stride < n0 ? base + k * stride : base;
        r1[k] = *(const uint2*)(rec0 + ic * 8u); r2[k] = *(const uint2*)(rec1 + ic * 8u); m[k] = GAML_TIMING_X(4) ? make_double2(1e-9 * (double)ic, -20.0) : *(const double2*)(sval + ic * 16u);
        lc[k] = ONE ? 0u : (unsigned)a.len_code[ic];
      }
    }
  }
#undef GAML_STAMP
}

// One pair with up to four records per mate (table classes 1 and 2 through their inline copies, delta pairs at the fixed
// stride) on windows that occur several times, in a lane: every load of a stage asked for before any is used -- the eight
// records, their occurrence entries, the bounds of their occurrence lists, the lists' entries -- four dependent trips, where
// the general loop (paired_general_src_masks) derives each candidate anew for every liveness test and every term: hundreds of
// dependent chains for a read in a 5-copy repeat seen through two windows. The candidates sit in (private) arrays; more than
// kGenCands on a mate: false, the caller takes the loop. Same candidates, same liveness rule, same terms in the same order.
#ifndef GAML_GEN_CANDS
#define GAML_GEN_CANDS 32
#endif
constexpr int kGenCands = GAML_GEN_CANDS;  // (<= 32: the liveness masks are one word)
// timing builds only (tools/build_variant.sh NAME -DGAML_GEN_STAMPS): per stage of general_pair_staged, the time the waves that
// score such a pair spend in it (summed over waves, 10 ns units) and [15] the number of waves; printed by gaml_hip_destroy
#ifdef GAML_GEN_STAMPS
__device__ unsigned long long g_gen_stamp[32];
#define GEN_STAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = wall_clock64(); \
  if (__ffsll((long long)__ballot(1)) - 1 == (int)(threadIdx.x & 63)) atomicAdd(&g_gen_stamp[k], now_ - gen_t_); atomicMax(&g_gen_stamp[16 + k], now_ - gen_t_); if (now_ - gen_t_ > 1000) atomicAdd(&g_gen_stamp[24 + k], 1ull); gen_t_ = now_; } while (0)
#define GEN_STAMP_BEGIN unsigned long long gen_t_ = wall_clock64(); const unsigned long long gen_t0_ = gen_t_; if (__ffsll((long long)__ballot(1)) - 1 == (int)(threadIdx.x & 63)) atomicAdd(&g_gen_stamp[15], 1ull);
#else
#define GEN_STAMP(k) do { } while (0)
#define GEN_STAMP_BEGIN
#endif
// `cand`: room for 2 * kGenCands candidates -- {path, position on the path, edit | orient << 8 | valid << 9 | record << 10, rank}, index
// = visiting order, mate 2's behind mate 1's -- in LDS where the wave has a free slot (gen_wave_slot), else in a private array:
// the liveness tests and the terms are chains of loads from it, 30 ns a link from LDS, 270 ns from scratch (a pair with 2 + 5
// candidates spent 8 us of a 15 us chain there, in-kernel stamps).
constexpr int kGenListBatch = 8;  // entries of an occurrence list requested together
template <int K>  // records per mate the caller holds (2: the class of pairs with at most two -- half the loads, half the code)
__device__ __forceinline__ bool general_pair_staged(const PairedArgs& a, const int4 (&r1)[K], const int4 (&r2)[K], int L1, int L2, double& acc_out, int4* cand) {
  int n[2] = {0, 0};
  unsigned live[2] = {0, 0};
  Occ12 e[2][K];
  int lb[2][K], le[2][K];
  GEN_STAMP_BEGIN
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int k = 0; k < K; k++) { const int4& r = m == 0 ? r1[k] : r2[k]; e[m][k] = a.m[m].occ12[r.x >= 0 ? r.x : 0]; }
  GEN_STAMP(0);
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int4& r = m == 0 ? r1[k] : r2[k];
      const bool occurs = r.x >= 0 && !(e[m][k].lo == ~0u && e[m][k].hi == ~0u);
      const bool list = occurs && e[m][k].rank < 0;
      lb[m][k] = le[m][k] = 0;
      if (list) { const int s = -e[m][k].rank - 1; lb[m][k] = a.m[m].multi_off[s]; le[m][k] = a.m[m].multi_off[s + 1]; }
      else if (occurs) le[m][k] = -1;  // one occurrence, described by the entry itself
    }
  GEN_STAMP(1);
  bool fits = true;
#pragma unroll
  for (int m = 0; m < 2; m++) {
    int4* const cm = cand + m * kGenCands;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int4& r = m == 0 ? r1[k] : r2[k];
      const bool single = le[m][k] == -1;
      const int cnt = single ? 1 : le[m][k] - lb[m][k];
      if (!__any(cnt > 0)) continue;  // (wave-uniform: nobody's record k of this mate occurs)
      if (n[m] + cnt > kGenCands) fits = false;
      const int4 one = occ_from_compact((unsigned long long)e[m][k].lo | ((unsigned long long)e[m][k].hi << 32), e[m][k].rank);
      const int n_before = n[m];  // candidates of the earlier records
      for (int c0 = 0; __any(fits && c0 < cnt); c0 += kGenListBatch) {  // (wave-uniform trip count; a 5-copy repeat: one round)
        const int left = fits ? cnt - c0 : 0;
        // the record's occurrences: the list's entries all at once (one trip, not one per entry)
        int4 t[kGenListBatch];
#pragma unroll
        for (int q = 0; q < kGenListBatch; q++) t[q] = (!single && left > 0) ? a.m[m].multi[lb[m][k] + c0 + min(q, left - 1)] : one;
        // liveness (graph.cc:583-592): valid, and no LATER valid candidate of the mate at the same (path, position) -- later: larger
        // (rank, record). Candidates of one record sit at different places (its window's occurrences are different places), so only
        // the earlier records' candidates are looked at: whichever of the two is earlier dies.
        unsigned nl = 0;
        int pos[kGenListBatch];
#pragma unroll
        for (int q = 0; q < kGenListBatch; q++) { pos[q] = r.y + t[q].x; nl |= (unsigned)(q < left && r.y >= t[q].y) << q; }
        const unsigned valid = nl;
        for (int y = 0; y < n_before && left > 0; y++) {
          const int4 ot = cm[y];
          if (!(ot.z & 0x200)) continue;
#pragma unroll
          for (int q = 0; q < kGenListBatch; q++)
            if (((valid >> q) & 1u) && ot.x == t[q].z && ot.y == pos[q]) { if (t[q].w >= ot.w) live[m] &= ~(1u << y); else nl &= ~(1u << q); }
        }
#pragma unroll
        for (int q = 0; q < kGenListBatch; q++)
          if (q < left) cm[n[m] + q] = make_int4(t[q].z, pos[q], (r.z & 0x1ff) | (((valid >> q) & 1u) ? 0x200 : 0) | (k << 10), t[q].w);
        if (left > 0) { live[m] |= nl << n[m]; n[m] += min(left, kGenListBatch); }
      }
    }
  }
  GEN_STAMP(2);
#ifdef GAML_GEN_STAMPS
  if (!fits) atomicAdd(&g_gen_stamp[12], 1ull);
#endif
  if (!fits) return false;
  GEN_STAMP(3);
  double acc = 0.0;
  for (int x = 0; x < n[0]; x++) {
    if (!((live[0] >> x) & 1u)) continue;
    const int4 cx = cand[x];
    Cand X; X.path = cx.x; X.pos = cx.y; X.edit = cx.z & 0xff; X.orient = (cx.z >> 8) & 1;
    for (int y = 0; y < n[1]; y++) {
      if (!((live[1] >> y) & 1u)) continue;
      const int4 cy = cand[kGenCands + y];
      if (cy.x != cx.x) continue;
      Cand Y; Y.path = cy.x; Y.pos = cy.y; Y.edit = cy.z & 0xff; Y.orient = (cy.z >> 8) & 1;
      acc += pair_term(a, X, Y, L1, L2);
    }
  }
  GEN_STAMP(4);
#ifdef GAML_GEN_STAMPS
  if (__ffsll((long long)__ballot(1)) - 1 == (int)(threadIdx.x & 63)) { atomicAdd(&g_gen_stamp[8], (unsigned long long)n[0]); atomicAdd(&g_gen_stamp[9], (unsigned long long)n[1]); atomicAdd(&g_gen_stamp[10], (unsigned long long)__popc(live[0])); atomicAdd(&g_gen_stamp[11], (unsigned long long)__popcll(__ballot(1))); }
  atomicMax(&g_gen_stamp[13], wall_clock64() - gen_t0_); atomicMax(&g_gen_stamp[14], (unsigned long long)(n[0] * 1000 + n[1])); atomicMax(&g_gen_stamp[7], (unsigned long long)(__popc(live[0]) * 1000 + __popc(live[1])));
#endif
  acc_out = acc;
  return true;
}
// The candidate stores of a lane-per-pair block of paired_score_kernel: the LDS its wave would stage a wave-per-pair item's
// candidates in (4 KB a wave, idle in these blocks) holds kGenWaveSlots of them; the wave's lanes that meet such a pair in
// one round take them in lane order, the others use a private array (general_pair_call).
constexpr int kOvfCap = 128;     // candidates per mate the wave-per-pair blocks hold in LDS per wave (paired_overflow_body)
constexpr int kGenWaveSlots = 2 * kOvfCap / (2 * kGenCands);
__device__ __forceinline__ int4* gen_wave_slot(int4* wave_lds, bool mine) {
  const unsigned long long m = __ballot(mine);
  const int rank = __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
  return (wave_lds && mine && rank < kGenWaveSlots) ? wave_lds + rank * 2 * kGenCands : nullptr;
}

// up to K live candidates per mate in registers -> per-read probability, floor / log, running sums
template <int K>
__device__ __forceinline__ void score_cands_and_finish(const PairedArgs& a, int i, uint32_t l12, const RegCands<K>& x, const RegCands<K>& y,
                                                       double& lsum, int& zeros, PairVal* cap = nullptr) {
  const int L1 = l12 & 0xffff, L2 = l12 >> 16;
    // Junction duplicates: the overwrite rule usually leaves one alignment per mate, i.e. one pair
    // term -- the value the compact path looks up in the memo (same table, same index).
    if (a.memo) {
      int n1 = 0, n2 = 0;
      Compact1 c;
      CompactPrep q;
      q.x.path = -1; q.x.pos = 0; q.x.edit = 0; q.x.orient = 0; q.y = q.x;
#pragma unroll
      for (int k = 0; k < K; k++) {
        if (x.live[k]) { n1++; q.x.path = x.path[k]; q.x.pos = x.pos[k]; q.x.edit = x.ef[k] & 0xff; q.x.orient = x.ef[k] >> 8; }
        if (y.live[k]) { n2++; q.y.path = y.path[k]; q.y.pos = y.pos[k]; q.y.edit = y.ef[k] & 0xff; q.y.orient = y.ef[k] >> 8; }
      }
      // the pair's length code: combination 0 travels by value (the usual set has one), the others are looked up
      int code = (a.lt_codes > 0 && l12 == a.len_combo0) ? 0 : -1;
      if (code < 0 && a.n_codes > 1) {
#pragma unroll
        for (int k = 3; k >= 1; k--) if (k < a.lt_codes && a.len_combo[k] == l12) code = k;
      }
      if (n1 == 1 && n2 == 1 && q.x.path == q.y.path && code >= 0 && q.x.orient != q.y.orient && q.x.edit < 7 && q.y.edit < 7) {
        const bool fwd = q.x.pos < q.y.pos;  // orientation rule and insert distance (graph.cc:1864-1876)
        const bool ok = fwd ? (q.x.orient == 0) : (q.x.orient == 1);
        const int dist = fwd ? q.y.pos - q.x.pos + L2 : q.x.pos - q.y.pos + L1;
        if (ok && dist >= 0 && dist < a.ins_n) {
          c.lc = code; c.L1 = L1; c.L2 = L2;
          q.dist = dist; q.scores = true; q.skip = false;
          q.memo_idx = ((code * 7 + q.x.edit) * 7 + q.y.edit) * a.ins_n + dist;
          // the two per-code values are requested WITH the memo entry, not behind the store that needs it (the store may
          // alias them for all the compiler knows: they were a round trip of their own at the end of every such wave)
          const double tfl = code == 0 ? a.tfloor0 : a.tfloor_c[code], lfl = code == 0 ? a.logfloor0 : a.logfloor_c[code];
          const double2 m = a.memo[q.memo_idx];
          if (cap) *cap = PairVal{m.x, m.y, code, 1};
          compact_cover(a, c, q, m.x);
          __builtin_nontemporal_store(m.x, &a.probs[i]);
          const bool floored = m.x < tfl;  // as compact_finish
          lsum += floored ? lfl : m.y - a.log_two_T;
          zeros += (int)floored;
          return;
        }
      }
    }
    const double acc = score_regs<K>(a, x, y, L1, L2);
    finish_read(a, i, acc, L1, L2, lsum, zeros, cap);
}

// Classes 1 and 2: at most K = 2 / 4 records per mate, 16-byte records, overwrite rule in registers.
// Slots [slot_lo, slot_hi), blocks [block_lo, block_hi).
// (tl: the in-kernel timeline's slot of this wave, stamps [2] records in, [3] candidates (occurrence entries) in, [5]
// first pair finished -- tools/kernel_timeline.py)
template <int K, bool GEN>
__device__ __forceinline__ void paired_regs_body(const PairedArgs& a, int lb, int slot_lo, int slot_hi, int block_lo, int block_hi,
                                                 double& lsum, int& zeros, unsigned long long* tl = nullptr, int4* wave_lds = nullptr) {
  bool first_pair = true;
#ifdef GAML_GEN_STAMPS
  const unsigned long long tb_ = wall_clock64();
#endif
  for (int i = slot_lo + (lb - block_lo) * kBlock + threadIdx.x; i < slot_hi; i += (block_hi - block_lo) * kBlock) {
    const int t = i - a.n0;
    const uint32_t l12 = a.len12[t];
    const size_t at = K == 2 ? (size_t)2 * (i - a.n0) : (size_t)2 * (a.n01 - a.n0) + (size_t)4 * (i - a.n01);
    // all 2K records first, then all 2K occurrence entries: two round trips, not one per record (a conditional
    // load per record, each behind its own wait, made this class's blocks the longest chain of the launch)
    int4 r1[K], r2[K];
#pragma unroll
    for (int k = 0; k < K; k++) { r1[k] = a.inl[0][at + k]; r2[k] = a.inl[1][at + k]; }
    if (tl && first_pair) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if ((threadIdx.x & 63) == 0) tl[2] = (unsigned long long)wall_clock64() + (r1[0].x == 0x12345 ? 1 : 0); }
    RegCands<K> x, y;
    const bool m1 = cands_from_records<K>(a.m[0], r1, x), m2 = cands_from_records<K>(a.m[1], r2, y);
    if (tl && first_pair) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if ((threadIdx.x & 63) == 0) tl[3] = (unsigned long long)wall_clock64() + (x.pos[0] == 0x12345 ? 1 : 0); }
    const bool dirty = r1[0].x == kDirtyWid;  // scored from the delta lists (paired_delta_body)
    const bool general = !dirty && (m1 || m2);  // a window that occurs several times: general_pair_call
    if (!dirty && !general) score_cands_and_finish<K>(a, i, l12, x, y, lsum, zeros);
    if (tl && first_pair) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if ((threadIdx.x & 63) == 0) tl[5] = (unsigned long long)wall_clock64() + (lsum == 0.12345 ? 1 : 0); }
    first_pair = false;
    if (GEN && !GAML_GEN_OFF(2) && __any(general)) {
      int4* const slot = gen_wave_slot(wave_lds, general);
#ifdef GAML_GEN_STAMPS
      const unsigned long long tc0_ = wall_clock64();
#endif
      if (general) {
        // (the class of pairs with two records a mate: staged here, from the records and in the registers this lane already has -- a
        // late annealing walk brings 25 such lanes a wave, the call cost each 19 us; without a slot, or with more candidates than
        // a slot holds: the function)
        double acc;
        const int L1 = l12 & 0xffff, L2 = l12 >> 16;
        int4 priv[K == 2 ? 2 * kGenCands : 1];
        if (K == 2 && general_pair_staged<K>(a, r1, r2, L1, L2, acc, slot ? slot : priv)) finish_read(a, i, acc, L1, L2, lsum, zeros);
        else { const GenOut o = general_pair_call(kernel_args_address(), i, -1, -1, slot); lsum += o.add; zeros += o.zeros; }
      }
#ifdef GAML_GEN_STAMPS  // (the call as the caller sees it, per LANE that makes it; and how long after the class body's entry it begins)
      if (general) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long tc1_ = wall_clock64(); atomicAdd(&g_gen_stamp[5], tc1_ - tc0_); atomicAdd(&g_gen_stamp[6], 1ull); atomicAdd(&g_gen_stamp[21], tc0_ - tb_); }
#endif
    }
  }
}

// Delta pairs (pairs that gained records since the tables were built): one LANE per pair, records from
// the delta lists. Up to 4 records per mate go through the register path of class 2; longer lists and
// pairs touching a window that occurs several times take the fully general per-lane loop (rare).
template <bool GEN>
__device__ __forceinline__ void paired_delta_body(const PairedArgs& a, int db, int delta_blocks, double& lsum, int& zeros, int4* wave_lds = nullptr) {
  // (a delta pair touching a window that occurs several times goes through general_pair_call -- GEN launches: the general
  // loop inlined here, one lane re-deriving every candidate's liveness for every candidate, made a late annealing walk's launch 60 us)
  for (int dj = db * kBlock + threadIdx.x; dj < a.dstate[kDsDirty]; dj += delta_blocks * kBlock) {
    // Everything a delta pair needs sits at index dj (no chain through the pair's slot): the first record of mate 1
    // carries the two read lengths in its spare word, the first record of mate 2 the two list lengths (paired_upload_delta)
    const int i = a.dirty_slots[dj];
    const int sp = a.dirty_spill[dj];
    int4 r0[4], r1[4];
#pragma unroll
    for (int k = 0; k < 2; k++) { r0[k] = a.dirty_recs[0][4 * (size_t)dj + k]; r1[k] = a.dirty_recs[1][4 * (size_t)dj + k]; }
    const uint32_t l12 = (uint32_t)r0[0].w;
    const int c0 = r1[0].w & 0xff, c1 = (r1[0].w >> 8) & 0xff;  // the two lists' lengths
    bool general = false;
    if (__all(sp >= 0 || (c0 <= 2 && c1 <= 2))) {  // the usual delta pair: a second window per mate (a node and its twin, a junction)
      if (sp < 0) {
        RegCands<2> x, y;
        const int4 q0[2] = {r0[0], r0[1]}, q1[2] = {r1[0], r1[1]};
        const bool m0 = cands_from_records<2>(a.m[0], q0, x), m1 = cands_from_records<2>(a.m[1], q1, y);
        general = m0 || m1;
        if (!general) score_cands_and_finish<2>(a, i, l12, x, y, lsum, zeros);
      }
    } else if (sp < 0) {  // (sp >= 0, a long list: one WAVE scores it, paired_overflow_body -- a lane looping over it alone was the launch's tail)
#pragma unroll
      for (int k = 2; k < 4; k++) { r0[k] = a.dirty_recs[0][4 * (size_t)dj + k]; r1[k] = a.dirty_recs[1][4 * (size_t)dj + k]; }
      RegCands<4> x, y;
      const bool m0 = cands_from_records<4>(a.m[0], r0, x), m1 = cands_from_records<4>(a.m[1], r1, y);
      general = m0 || m1;
      if (!general) score_cands_and_finish<4>(a, i, l12, x, y, lsum, zeros);
    }
    if (GEN) {
      if (!GAML_GEN_OFF(4) && __any(general)) {
        int4* const slot = gen_wave_slot(wave_lds, general);
        if (general) { const GenOut o = general_pair_call(kernel_args_address(), i, dj, -1, slot); lsum += o.add; zeros += o.zeros; }
      }
    } else if (general) {  // cannot happen (a launch without notes has no such window): the partial is poisoned, combine() reports it
      lsum += __builtin_nan("");
    }
  }
}

// delta store maintenance: one thread per patched pair writes its slot, spill index and padded records
struct DeltaPatch { int dj, slot, spill, pad; int4 rec[2][4]; };
// the same with room for two records per mate -- the usual delta pair: a node and its twin, a node and a junction -- 80 bytes
// instead of 144 for the host to write and the device to fetch over PCIe (a patch of a few thousand pairs is what a call costs
// that activates a window aligned earlier)
struct DeltaPatch2 { int dj, slot, spill, pad; int4 rec[2][2]; };
// the class tables' "this pair lives on the delta lists now" marks (one per new delta pair); kDirty8 / kDirtyWid
__device__ __forceinline__ void mark_dirty_slot(int s, unsigned long long* rec8_0, int n0, int4* inl_0, int n01, int n_main, int4* first_0) {
  if (s < n0) rec8_0[s] = ~0ull - 1;
  else {
    if (s < n01) inl_0[(size_t)2 * (s - n0)].x = -2;
    else if (s < n_main) inl_0[(size_t)2 * (n01 - n0) + (size_t)4 * (s - n01)].x = -2;
    first_0[s - n0].x = -2;
  }
}
// `patch` is read where the host wrote it (mapped pinned memory): no copy kernel in front; delta pairs numbered
// mark_from and up are new with this patch and get their marks here (mark_from < 0: none) -- one dispatch where there
// were three: copy, patch, marks
template <class Patch, int K>
__global__ __launch_bounds__(kBlock) void apply_delta_patch_kernel(const Patch* patch, int n, int* slots, int* spill, int4* rec0, int4* rec1, int mark_from,
                                                                   unsigned long long* rec8_0, int n0, int4* inl_0, int n01, int n_main, int4* first_0) {
  for (int t = blockIdx.x * kBlock + threadIdx.x; t < n; t += gridDim.x * kBlock) {
    const Patch p = patch[t];
    slots[p.dj] = p.slot;
    spill[p.dj] = p.spill;
#pragma unroll
    for (int k = 0; k < 4; k++) {  // (the store always holds four records per mate: the short form's missing ones are "none")
      rec0[4 * (size_t)p.dj + k] = k < K ? p.rec[0][k < K ? k : 0] : make_int4(-1, 0, 0, 0);
      rec1[4 * (size_t)p.dj + k] = k < K ? p.rec[1][k < K ? k : 0] : make_int4(-1, 0, 0, 0);
    }
    if (mark_from >= 0 && p.dj >= mark_from) mark_dirty_slot(p.slot, rec8_0, n0, inl_0, n01, n_main, first_0);
  }
}

// TL: the in-kernel timeline of tools/kernel_timeline.py (a separate instantiation: the product kernels carry none of it)
template <bool TICKET, bool GEN, bool TL>
__device__ __forceinline__ void paired_main_body(const PairedArgs& a, int lb, double* sh_s, int* sh_z, int4* gen_lds) {
  // `a`: the argument block as the kernel received it; only its leading words (grid layout, partial slots) are read
  // here -- every class takes a fresh view of its own (GAML_FRESH_ARGS)
  double lsum = 0.0;
  int zeros = 0;
  unsigned long long* tl = TL ? a.timeline + ((size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 8 : nullptr;
  if (TL && (threadIdx.x & 63) == 0) { tl[0] = wall_clock64(); tl[7] = lb < a.blocks0 ? 0 : (lb < a.blocks01 ? 1 : 2); }
  if (GAML_TIMING_X(8) && lb >= a.blocks0a) {  // timing experiment: only the static part of class 0 does anything
  } else if (GAML_TIMING_X(16) && lb < a.blocks0a) {  // ... or everything but it
  } else if (lb < a.blocks0) {
    GAML_FRESH_ARGS(c, a)
    const bool wide = c.memo && !c.cov_bits;  // block-uniform: memo present, no coverage marks to set
    const SlotRange rg = compact_range(c, lb);
    const bool stat = wide && lb < c.blocks0a;  // the part of class 0 whose memo indices came with the tables
    if (!TL && wide && c.n_codes == 1) {  // one length combination: no tables, no barrier
      if (stat) paired_static4_body<GEN, false, true>(c, rg, lsum, zeros);
      else paired_compact4_body<GEN, false, true>(c, rg, lsum, zeros);
    } else {
      // the per-length-combination tables of the compact class (<= 256 entries each) are looked up once or twice
      // per pair, each time behind another load: from LDS they cost an LDS access instead of an L2 round trip
      __shared__ uint32_t sh_combo[256];
      __shared__ double sh_floor[256], sh_logfloor[256], sh_tfloor[256];
      for (int k = threadIdx.x; k < c.n_codes; k += kBlock) { sh_combo[k] = c.len_combo[k]; sh_floor[k] = c.floor_c[k]; sh_logfloor[k] = c.logfloor_c[k]; sh_tfloor[k] = c.tfloor_c ? c.tfloor_c[k] : 0.0; }
      __syncthreads();
      PairedArgs b = c;
      b.len_combo = sh_combo; b.floor_c = sh_floor; b.logfloor_c = sh_logfloor; b.tfloor_c = sh_tfloor;
      if (TL && (threadIdx.x & 63) == 0) tl[1] = wall_clock64();
      if (stat) paired_static4_body<GEN, TL>(b, rg, lsum, zeros);
      else if (wide) paired_compact4_body<GEN, TL>(b, rg, lsum, zeros);
      else paired_compact_body<GEN>(b, rg, lsum, zeros);
    }
  } else {
    // (GEN: the classes with several records per pair stage a repeated window's candidates in their wave's LDS, gen_wave_slot)
    int4* const wave_lds = GEN ? gen_lds + (threadIdx.x >> 6) * 2 * kOvfCap : nullptr;
    if (lb < a.blocks01) {
      GAML_FRESH_ARGS(c, a)
      paired_regs_body<2, GEN>(c, lb, c.n0, c.n01, c.blocks0, c.blocks01, lsum, zeros, tl, wave_lds);
    } else if (lb < a.blocks012) {
      GAML_FRESH_ARGS(c, a)
      paired_regs_body<4, GEN>(c, lb, c.n01, c.n_main, c.blocks01, c.blocks012, lsum, zeros, nullptr, wave_lds);
    } else {
      GAML_FRESH_ARGS(c, a)
      paired_delta_body<GEN>(c, lb - c.blocks012, c.main_blocks - c.blocks012, lsum, zeros, wave_lds);
    }
  }
  block_reduce(lsum, zeros, sh_s, sh_z);
  if (TL && (threadIdx.x & 63) == 0) tl[6] = wall_clock64();
  if (TICKET) {
    GAML_FRESH_ARGS(c, a)
    grid_finish(lsum, zeros, lb, c.total_blocks, c.part_sum, c.part_zero, c.ticket, c.out,
                c.cov_bits ? -1.0 : 0.0, c.n_reads, sh_s, sh_z, c.status_out, c.status_a, c.status_b);
  } else if (threadIdx.x == 0) {
    a.part_sum[lb] = lsum;
    a.part_zero[lb] = zeros;
  }
}

// experiment: separate one-block finisher (sums n_partials partials in index order)
__global__ __launch_bounds__(kBlock) void finish_partials_kernel(const double* part_sum, const int* part_zero, int n_partials,
                                                                 double* out, double bad_bases, double n_reads,
                                                                 double* status_out = nullptr, double status_a = 0.0, double status_b = 0.0) {
  __shared__ double sh_s[kBlock / 64];
  __shared__ int sh_z[kBlock / 64];
  double ts = 0; int tz = 0;
  for (int b = threadIdx.x; b < n_partials; b += kBlock) { ts += part_sum[b]; tz += part_zero[b]; }
  block_reduce(ts, tz, sh_s, sh_z);
  if (threadIdx.x == 0) {
    out[0] = ts; out[1] = (double)tz; if (bad_bases >= 0) out[2] = bad_bases; out[3] = n_reads;
    if (status_out) { status_out[0] = status_a; status_out[1] = status_b; }  // (a sharded evaluation's status words ride along: no dispatch of their own)
  }
}

// Overflow body: one WAVE per pair, for slots [n_main, n) (more than 4 records on a mate).
// The wave gathers every (record, occurrence) candidate of both mates into LDS,
// settles the overwrite rule in parallel and spreads the x * y pair terms over its lanes;
// deterministic lane-strided + butterfly summation. It reads nothing the main kernel writes, so
// both run concurrently (two streams); they share one ticket, and whichever block finishes last
// folds all per-block partials in index order.
constexpr int kOvfMaxBlocks = 1024;

template <class Src>
__device__ __forceinline__ int wave_gather(const MateView& v, const Src& src, int4* lds, int lane) {
  // returns the number of candidates, or -1 if they do not fit
  const int cnt = src.count();
  int total = 0;
  for (int base = 0; base < cnt; base += 64) {
    const int k = base + lane;
    int mine = 0;
    int4 r = make_int4(-1, 0, 0, 0), o = make_int4(0, 0, -1, 0);
    if (k < cnt) {
      r = src.get(k);
      if (r.x >= 0) { const Occ12 e = v.occ12[r.x]; o = occ_from_compact((unsigned long long)e.lo | ((unsigned long long)e.hi << 32), e.rank); }  // (paired sets only)
      if (o.z >= 0) mine = o.w >= 0 ? 1 : v.multi_off[-o.w] - v.multi_off[-o.w - 1];
    }
    // exclusive prefix sum of `mine` over the wave
    int incl = mine;
    for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    const int wave_total = __shfl(incl, 63, 64);
    const int at = total + incl - mine;
    if (total + wave_total > kOvfCap) return -1;
    if (mine == 1 && o.w >= 0) {
      lds[at] = make_int4(o.z, r.y + o.x, (r.z & 0x1ff) | ((r.y >= o.y) ? 0x200 : 0), o.w);
    } else if (mine > 0) {
      const int s = -o.w - 1;
      for (int q = 0; q < mine; q++) {
        const int4 oo = v.multi[v.multi_off[s] + q];
        lds[at + q] = make_int4(oo.z, r.y + oo.x, (r.z & 0x1ff) | ((r.y >= oo.y) ? 0x200 : 0), oo.w);
      }
    }
    total += wave_total;
  }
  return total;
}

// one pair, one wave: candidates of both mates staged in LDS, overwrite rule and pair terms spread over the lanes
template <class Src>
__device__ __forceinline__ double wave_score_pair(const PairedArgs& a, const Src& s1, const Src& s2, int L1, int L2, int4* c1, int4* c2, int lane) {
  const int n1 = wave_gather(a.m[0], s1, c1, lane);
  const int n2 = wave_gather(a.m[1], s2, c2, lane);
  double acc = 0.0;
  if (n1 < 0 || n2 < 0) {
    // more candidates than the LDS staging holds: fully general per-lane loop
    if (lane == 0) acc = paired_general_src(a, s1, s2, L1, L2);
    return acc;
  }
  __builtin_amdgcn_wave_barrier();
  // overwrite rule: candidate is live iff valid and no later-ranked valid twin (same path, pos)
  for (int m = 0; m < 2; m++) {
    int4* c = m == 0 ? c1 : c2;
    const int n = m == 0 ? n1 : n2;
    for (int x = lane; x < n; x += 64) {
      int4 me = c[x];
      bool live = (me.z & 0x200) != 0;
      for (int y = 0; y < n && live; y++) {
        const int4 ot = c[y];
        if (y != x && (ot.z & 0x200) && ot.x == me.x && ot.y == me.y && (ot.w > me.w || (ot.w == me.w && y > x))) live = false;
      }
      if (live) me.z |= 0x400;
      c[x].z = me.z;  // readers only test the valid bit 0x200; the live bit 0x400 is written once per slot
    }
  }
  __builtin_amdgcn_wave_barrier();
  const int pairs = n1 * n2;
  for (int idx = lane; idx < pairs; idx += 64) {
    const int4 xr = c1[idx / n2];
    const int4 yr = c2[idx % n2];
    if ((xr.z & 0x400) && (yr.z & 0x400) && xr.x == yr.x) {
      Cand x, y;
      x.path = xr.x; x.pos = xr.y; x.edit = xr.z & 0xff; x.orient = (xr.z >> 8) & 1;
      y.path = yr.x; y.pos = yr.y; y.edit = yr.z & 0xff; y.orient = (yr.z >> 8) & 1;
      acc += pair_term(a, x, y, L1, L2);
    }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  __builtin_amdgcn_wave_barrier();
  return acc;
}

template <bool TICKET>
__device__ __forceinline__ void paired_overflow_body(const PairedArgs& a, int ovf_block, int ovf_blocks, double* sh_s, int* sh_z,
                                                     int4 (*cand)[2][kOvfCap]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_global = ovf_block * (kBlock / 64) + wave;
  const int n_waves = ovf_blocks * (kBlock / 64);
  const int n_table = a.n - a.n_main;  // table pairs with more than 4 records on a mate
  const int n_items = n_table + a.dstate[kDsSpill];  // ... then the delta pairs with long lists
  double lsum = 0.0;
  int zeros = 0;
  int4* c1 = cand[wave][0];
  int4* c2 = cand[wave][1];
  for (int item = wave_global; item < n_items; item += n_waves) {  // fixed item -> wave assignment
    if (item < n_table) {
      const int i = a.n_main + item;
      const int4 r1 = a.m[0].first[i - a.n0], r2 = a.m[1].first[i - a.n0];
      const uint32_t l12 = a.len12[i - a.n0];
      if (r1.x == kDirtyWid) continue;  // a class-3 pair that is on the delta list: scored there
      const int L1 = l12 & 0xffff, L2 = l12 >> 16;
      const double acc = wave_score_pair(a, TableSrc{&a.m[0], r1}, TableSrc{&a.m[1], r2}, L1, L2, c1, c2, lane);
      if (lane == 0) finish_read(a, i, acc, L1, L2, lsum, zeros);
    } else {
      const int sp = item - n_table;
      const int i = a.spill_slot[sp];
      const uint32_t l12 = i < a.n0 ? a.len_combo[a.len_code[i]] : a.len12[i - a.n0];
      const int L1 = l12 & 0xffff, L2 = l12 >> 16;
      const int2 g0 = a.spill_rng[0][sp], g1 = a.spill_rng[1][sp];
      const double acc = wave_score_pair(a, ListSrc{a.spill_recs[0] + g0.x, g0.y}, ListSrc{a.spill_recs[1] + g1.x, g1.y}, L1, L2, c1, c2, lane);
      if (lane == 0) finish_read(a, i, acc, L1, L2, lsum, zeros);
    }
  }
  block_reduce(lsum, zeros, sh_s, sh_z);
  if (TICKET) {
    grid_finish(lsum, zeros, a.main_blocks + ovf_block, a.total_blocks, a.part_sum, a.part_zero, a.ticket, a.out,
                a.cov_bits ? -1.0 : 0.0, a.n_reads, sh_s, sh_z, a.status_out, a.status_a, a.status_b);
  } else if (threadIdx.x == 0) {
    a.part_sum[a.main_blocks + ovf_block] = lsum;
    a.part_zero[a.main_blocks + ovf_block] = zeros;
  }
}

// ONE launch for a paired read set: blocks [0, main_blocks) run the lane-per-pair path, blocks
// [main_blocks, total_blocks) the wave-per-pair path (block-uniform branch). They share nothing
// but read-only tables, so no ordering between them is needed.
// GEN: the path set has windows that occur several times -- their pairs are scored where a lane meets them, by function
// calls (compact_general_call, general_pair_call); four waves per SIMD instead of five (the callees' registers). Without such
// windows none of this is compiled in (rounds 2-3 noted the pairs and scored them in a second launch: 20-25 us of its own).
#ifndef GAML_GEN_WAVES
#define GAML_GEN_WAVES 4
#endif
template <bool TICKET, bool GEN = false, bool TL = false>
__global__ __launch_bounds__(kBlock, GEN ? GAML_GEN_WAVES : 5) void paired_score_kernel(PairedArgs a) {
  __shared__ double sh_s[kBlock / 64];
  __shared__ int sh_z[kBlock / 64];
  __shared__ int4 cand[kBlock / 64][2][kOvfCap];
  // Logical block ids are [compact][<= 2 records][<= 4 records][overflow]; they are handed out in
  // REVERSE dispatch order so that the few long-latency blocks (overflow, multi-record classes)
  // start first and hide under the compact stream instead of forming a tail.
  const int lb = a.total_blocks - 1 - (int)blockIdx.x;
#if GAML_LEAN_ENTRY && defined(__HIP_DEVICE_COMPILE__)
  // The argument block is ~600 bytes. Left to itself the compiler fetches ALL of it at kernel entry -- more than the
  // scalar registers hold, so in six dependent fetch-wait-spill rounds (~90 v_writelane) before any wave can request
  // its first record -- although the blocks that score nearly all pairs (static part of the compact class, one length
  // combination) need two dozen of its words. Those blocks read `a` directly (its leading words: PairedArgs); every other
  // class takes a fresh view of the block where its code starts (GAML_FRESH_ARGS).
  if (!TL && lb < a.blocks0a && a.memo && !a.cov_bits && a.n_codes == 1) {
    double lsum = 0.0;
    int zeros = 0;
    if (!GAML_TIMING_X(16))
    paired_static4_body<GEN, false, true>(a, SlotRange{0, a.n0a, lb, a.blocks0a}, lsum, zeros);
    block_reduce(lsum, zeros, sh_s, sh_z);
    if (TICKET) grid_finish(lsum, zeros, lb, a.total_blocks, a.part_sum, a.part_zero, a.ticket, a.out, 0.0, a.n_reads, sh_s, sh_z, a.status_out, a.status_a, a.status_b);
    else if (threadIdx.x == 0) { a.part_sum[lb] = lsum; a.part_zero[lb] = zeros; }
    return;
  }
  if (lb < a.main_blocks) paired_main_body<TICKET, GEN, TL>(a, lb, sh_s, sh_z, &cand[0][0][0]);
  else if (GAML_GEN_OFF(16)) { if (threadIdx.x == 0) { a.part_sum[lb] = 0.0; a.part_zero[lb] = 0; } }
  else {
    GAML_FRESH_ARGS(c, a)
    paired_overflow_body<TICKET>(c, lb - c.main_blocks, c.total_blocks - c.main_blocks, sh_s, sh_z, cand);
  }
#else
  if (lb < a.main_blocks) paired_main_body<TICKET, GEN, TL>(a, lb, sh_s, sh_z, &cand[0][0][0]);
  else paired_overflow_body<TICKET>(a, lb - a.main_blocks, a.total_blocks - a.main_blocks, sh_s, sh_z, cand);
#endif
}

// ---------------------------------------------------------------------------------------------------------
// Several path sets in ONE pass over the records (gaml_hip_calc_prob_batch; the move generators compare a handful of
// near-identical candidate assemblies: moves.cc:107-113 LocalChange2, 694-800 FixGapLength, 1156-1305 FixRepForNode2).
// A path set only changes WHERE windows sit (its occurrence tables, 12 B per window) and 2T; the records, the
// memo of pair terms and the grid are the same for all of them. So: the compact class (90 % of the pairs) loads
// its 8-byte records once and resolves them against every set's tables (S x 95 KB at cfg3: L2 resident) in an inner
// loop; the other classes (10 % of the pairs, a few records each, L2 hits after the first set) simply run their body
// once per set. Every (block, set) writes its own partial: same lane -> pair mapping and reduction order as the
// single-set kernel, so a batch gives bit for bit what the sets give one by one.
// ---------------------------------------------------------------------------------------------------------
constexpr int kMaxSets = 8;
struct SetDev {  // what differs between the path sets of one batch
  const Occ12* occ12[2];
  const int* multi_off[2];
  const int4* multi[2];
  const double* tfloor_c;          // [code] for this set's 2T
  double tfloor0;                  // = tfloor_c[0], by value
  double two_T, log_two_T;
  double* part_sum;                // this set's per-block partials
  int* part_zero;
};
// chg[mt][w]: bit s set = window w's table entry in set s of this launch may differ from set 0's (s >= 1; a batch whose
// sets' tables were built from patches knows, batch_tables_kernel). A pair none of whose records touch such a window
// resolves to the same candidates in set s as in set 0: its set-0 result is finished again under set s's 2T and
// thresholds, no table is read. The bits are cumulative (bit s implies bit s + 1: a set's tables are its
// predecessor's plus a patch). Null: unknown, every set resolves every pair. Used by the classes with several records
// per pair and the delta / wave-per-pair blocks; the compact class resolves every set (see its body).
struct MultiSets { int n; int pad_; const unsigned char* chg[2]; SetDev set[kMaxSets]; };

constexpr int kTfCodes = 16;  // length codes whose per-set thresholds a multi-set block keeps in LDS
__device__ __forceinline__ PairedArgs with_set(const PairedArgs& a, const SetDev& sd, const double* tfloor_lds = nullptr) {
  PairedArgs b = a;
#pragma unroll
  for (int mt = 0; mt < 2; mt++) { b.m[mt].occ12 = sd.occ12[mt]; b.occ12[mt] = sd.occ12[mt]; b.m[mt].multi_off = sd.multi_off[mt]; b.m[mt].multi = sd.multi[mt]; }
  b.tfloor_c = tfloor_lds ? tfloor_lds : sd.tfloor_c; b.tfloor0 = sd.tfloor0; b.two_T = sd.two_T; b.log_two_T = sd.log_two_T;
  b.part_sum = sd.part_sum; b.part_zero = sd.part_zero;
  return b;
}

// the kernel's argument block, read where it is needed (a callee's view of it); a multi-set launch's: {PairedArgs, MultiSets}
struct MultiKernArgs { PairedArgs a; MultiSets ms; };
#if defined(__HIP_DEVICE_COMPILE__)
#define GAML_CALLEE_ARGS(name, set)                                                                                                  \
  const unsigned long long name##_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(kernargs >> 32)) << 32) |  \
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)kernargs); /* (uniform: scalar loads) */         \
  const __attribute__((address_space(4))) MultiKernArgs* name##_p = (const __attribute__((address_space(4))) MultiKernArgs*)name##_u; \
  const PairedArgs name##_0 = name##_p->a;                                                                                           \
  const PairedArgs name = (set) >= 0 ? with_set(name##_0, name##_p->ms.set[set]) : name##_0;
#else
#define GAML_CALLEE_ARGS(name, set) const PairedArgs& name = *(const PairedArgs*)nullptr;
#endif
__device__ __noinline__ GenOut compact_general_call(unsigned long long kernargs, int i, int set) {
  GAML_CALLEE_ARGS(a, set)
  GenOut o{0.0, 0};
  compact_general(a, i, o.add, o.zeros);
  return o;
}
__device__ __noinline__ GenOut general_pair_call(unsigned long long kernargs, int i, int dj, int set, int4* lds) {
  GAML_CALLEE_ARGS(a, set)
  GenOut o{0.0, 0};
  int4 priv[2 * kGenCands];
  int4* const cand = lds ? lds : priv;
  const int4 none = make_int4(-1, 0, 0, 0);
  int4 r1[4], r2[4];
  double acc;
  if (dj < 0) {  // a table pair of class 1 / 2: its inline copies
    const bool four = i >= a.n01;
    const size_t at = four ? (size_t)2 * (a.n01 - a.n0) + (size_t)4 * (i - a.n01) : (size_t)2 * (i - a.n0);
#pragma unroll
    for (int k = 0; k < 4; k++) { const bool has = four || k < 2; r1[k] = has ? a.inl[0][at + (has ? k : 0)] : none; r2[k] = has ? a.inl[1][at + (has ? k : 0)] : none; }
    const uint32_t l12 = a.len12[i - a.n0];
    const int L1 = l12 & 0xffff, L2 = l12 >> 16;
    if (!general_pair_staged<4>(a, r1, r2, L1, L2, acc, cand)) acc = paired_general(a, a.m[0].first[i - a.n0], a.m[1].first[i - a.n0], L1, L2);
    finish_read(a, i, acc, L1, L2, o.add, o.zeros);
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) { r1[k] = a.dirty_recs[0][4 * (size_t)dj + k]; r2[k] = a.dirty_recs[1][4 * (size_t)dj + k]; }
    const uint32_t l12 = (uint32_t)r1[0].w;
    const int L1 = l12 & 0xffff, L2 = l12 >> 16;
    const int c0 = r2[0].w & 0xff, c1 = (r2[0].w >> 8) & 0xff;
    if (!general_pair_staged<4>(a, r1, r2, L1, L2, acc, cand))
      acc = paired_general_src_masks(a, ListSrc{a.dirty_recs[0] + 4 * (size_t)dj, c0}, ListSrc{a.dirty_recs[1] + 4 * (size_t)dj, c1}, L1, L2);
    finish_read(a, i, acc, L1, L2, o.add, o.zeros);
  }
  return o;
}

// paired_compact4_body with the path sets in the inner loop. acc_s / acc_z: one running sum per (set, thread) in LDS
// (a lane may take several rounds of four pairs; registers cannot be indexed by the set number).
template <bool GEN, bool ONE>
__device__ __forceinline__ void paired_compact4_multi_body(const PairedArgs& a, const MultiSets& ms, const SlotRange rg, double* acc_s, int* acc_z, const double* tf) {
  const unsigned stride = (unsigned)rg.blocks * kBlock, n0 = (unsigned)rg.hi;  // (n0: end of this part's slots)
  const char* const rec0 = (const char*)a.rec8[0];
  const char* const rec1 = (const char*)a.rec8[1];
  const char* const memo = (const char*)a.memo;
  char* const probs = (char*)a.probs;
  const uint32_t l12_one = ONE ? a.len_combo[0] : 0u;
  const double logfloor_one = ONE ? a.logfloor_c[0] : 0.0;
  for (unsigned base = (unsigned)rg.lo + (unsigned)rg.lb * kBlock + threadIdx.x; base < n0; base += 4 * stride) {
    uint2 r1[4], r2[4];
    unsigned lc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {  // the records: ONCE for all sets
      const unsigned ic = base + k * stride < n0 ? base + k * stride : base;
      r1[k] = *(const uint2*)(rec0 + ic * 8u); r2[k] = *(const uint2*)(rec1 + ic * 8u); lc[k] = ONE ? 0u : (unsigned)a.len_code[ic];
    }
    // (Every set resolves every pair here. Finishing a pair from its set-0 result where no window of it changed -- as the
    // other classes do -- does not pay in this class: the launch lasts as long as its slowest wavefront, and some
    // wavefront always holds a pair on a changed window; the bookkeeping only costs registers. Measured: 26.1 us for
    // four sets this way, 28.7 us with the capture, tools/batch_ablate.py.)
#pragma unroll 1
    for (int s = 0; s < ms.n; s++) {
      const SetDev& sd = ms.set[s];
      const char* const occ0 = (const char*)sd.occ12[0];
      const char* const occ1 = (const char*)sd.occ12[1];
      const double* const tfs = tf ? tf + s * kTfCodes : sd.tfloor_c;  // this set's thresholds per length code (LDS copy when it fits)
      const double log2T = sd.log_two_T, tfloor_one = ONE ? tfs[0] : 0.0;
      const bool last_set = s == ms.n - 1;  // per-read probabilities: those of the last set, as after a sequence of calls
      uint2 o1[4], o2[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const Occ12* e1 = (const Occ12*)(occ0 + (r1[k].y != ~0u ? (r1[k].x & 0xffffffu) : 0u) * 12u);
        const Occ12* e2 = (const Occ12*)(occ1 + (r2[k].y != ~0u ? (r2[k].x & 0xffffffu) : 0u) * 12u);
        o1[k] = make_uint2(e1->lo, e1->hi); o2[k] = make_uint2(e2->lo, e2->hi);
      }
      int state[4];
      unsigned skip_bits = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        bool skip;
        state[k] = compact_state(a, r1[k], r2[k], o1[k], o2[k], lc[k], ONE ? l12_one : a.len_combo[lc[k]], base + k * stride < n0, skip);
        skip_bits |= (unsigned)skip << k;
      }
      double2 m[4];
#pragma unroll
      for (int k = 0; k < 4; k++) m[k] = *(const double2*)(memo + (unsigned)max(state[k], 0) * 16u);
      double lsum = 0.0;
      int zeros = 0;
      bool other = false;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        double* const out = (double*)(probs + (base + k * stride) * 8u);
        if (state[k] >= 0) {
          if (last_set) __builtin_nontemporal_store(m[k].x, out);
          const bool floored = m[k].x < (ONE ? tfloor_one : tfs[lc[k]]);
          lsum += floored ? (ONE ? logfloor_one : a.logfloor_c[lc[k]]) : m[k].y - log2T;
          zeros += (int)floored;
        } else if (state[k] > kPairOther) {
          if (last_set) __builtin_nontemporal_store(0.0, out);
          zeros++;
          lsum += ONE ? logfloor_one : a.logfloor_c[kPairZero - state[k]];
        } else other |= state[k] == kPairOther && !((skip_bits >> k) & 1u);
      }
      if (__any(other)) {  // scores, but outside the memo: from the tables (rare)
        const PairedArgs b = with_set(a, sd, tf ? tfs : nullptr);
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
          if (state[k] != kPairOther || ((skip_bits >> k) & 1u)) continue;
          const int i = (int)(base + k * stride);
          Compact1 d;
          compact_load(b, i, true, d);
          d.o1 = d.r1 != kNone8 ? occ8_of(b.occ12[0], (unsigned)(d.r1 & 0xffffff)) : kNone8;
          d.o2 = d.r2 != kNone8 ? occ8_of(b.occ12[1], (unsigned)(d.r2 & 0xffffff)) : kNone8;
          const uint32_t l = b.len_combo[d.lc];
          d.L1 = l & 0xffff; d.L2 = l >> 16;
          CompactPrep q;
          compact_prep(b, d, q);
          compact_finish(b, i, d, q, make_double2(0.0, 0.0), lsum, zeros);
        }
      }
      if (GEN && __any(skip_bits != 0)) {  // as paired_compact4_body
#pragma unroll 1
        for (int k = 0; k < 4; k++)
          if ((skip_bits >> k) & 1u) { const GenOut o = compact_general_call(kernel_args_address(), (int)(base + k * stride), s); lsum += o.add; zeros += o.zeros; }
      }
      acc_s[s * kBlock + threadIdx.x] += lsum;
      acc_z[s * kBlock + threadIdx.x] += zeros;
    }
  }
}

// paired_static4_body with the path sets in the inner loop: the static pairs' records AND values come in once for all sets
// (the per-set arithmetic above goes records -> occurrence entries -> memo index -> memo entry, two dependent trips per set;
// here a set costs one: its occurrence entries). Lanes, pairs, the values added and their order are the single-set
// kernel's: a batch gives bit for bit what the sets give one by one.
template <bool GEN, bool ONE>
__device__ __forceinline__ void paired_static4_multi_body(const PairedArgs& a, const MultiSets& ms, const SlotRange rg, double* acc_s, int* acc_z, const double* tf) {
  const unsigned stride = (unsigned)rg.blocks * kBlock, n0 = (unsigned)rg.hi;
  const char* const rec0 = (const char*)a.rec8[0];
  const char* const rec1 = (const char*)a.rec8[1];
  const char* const sval = (const char*)a.static_val;
  char* const probs = (char*)a.probs;
  const double logfloor_one = ONE ? a.logfloor_c[0] : 0.0;
  for (unsigned base = (unsigned)rg.lo + (unsigned)rg.lb * kBlock + threadIdx.x; base < n0; base += 4 * stride) {
    uint2 r1[4], r2[4];
    double2 m[4];
    unsigned lc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {  // records and values: ONCE for all sets
      const unsigned ic = base + k * stride < n0 ? base + k * stride : base;
      r1[k] = *(const uint2*)(rec0 + ic * 8u); r2[k] = *(const uint2*)(rec1 + ic * 8u); m[k] = *(const double2*)(sval + ic * 16u);
      lc[k] = ONE ? 0u : (unsigned)a.len_code[ic];
    }
#pragma unroll 1
    for (int s = 0; s < ms.n; s++) {
      const SetDev& sd = ms.set[s];
      const char* const occ0 = (const char*)sd.occ12[0];
      const char* const occ1 = (const char*)sd.occ12[1];
      const double* const tfs = tf ? tf + s * kTfCodes : sd.tfloor_c;
      const double log2T = sd.log_two_T, tfloor_one = ONE ? tfs[0] : 0.0;
      const bool last_set = s == ms.n - 1;
      uint2 o1[4], o2[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const unsigned w1 = (r1[k].x & 0xffffffu) & (0u - (unsigned)(r1[k].y != ~0u)), w2 = (r2[k].x & 0xffffffu) & (0u - (unsigned)(r2[k].y != ~0u));
        const Occ12* e1 = (const Occ12*)(occ0 + w1 * 12u);
        const Occ12* e2 = (const Occ12*)(occ1 + w2 * 12u);
        o1[k] = make_uint2(e1->lo, e1->hi); o2[k] = make_uint2(e2->lo, e2->hi);
      }
      double lsum = acc_s[s * kBlock + threadIdx.x];  // (the running sum of this lane and set: additions in the single-set kernel's order)
      int zeros = acc_z[s * kBlock + threadIdx.x];
      unsigned skip_bits = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {  // as paired_static4_body, statement for statement
        const bool none1 = r1[k].y == ~0u, none2 = r2[k].y == ~0u;
        const bool here = (base + k * stride < n0) & !(none1 & (r1[k].x == 0xfffffffeu));
        const bool w1 = !none1 & (o1[k].y != ~0u), w2 = !none2 & (o2[k].y != ~0u);
        const bool gen = here & ((w1 & ((int)o1[k].y < 0)) | (w2 & ((int)o2[k].y < 0)));
        const bool same = (o1[k].x == o2[k].x) & (((o1[k].y ^ o2[k].y) >> 16) == 0);
        const int p1 = (int)(__funnelshift_r(r1[k].x, r1[k].y, 24) & 0xfffffffu), p2 = (int)(__funnelshift_r(r2[k].x, r2[k].y, 24) & 0xfffffffu);
        const bool kept = (p1 >= (int)(short)(o1[k].y & 0xffffu)) & (p2 >= (int)(short)(o2[k].y & 0xffffu));
        const bool both = here & !gen & w1 & w2;
        const bool scores = both & same & kept;
        const bool poison = both & !same;
        const bool counted = here & !gen;
        skip_bits |= (unsigned)gen << k;
        const double t = scores ? m[k].x : 0.0;
        const bool floored = counted & (!scores | (t < (ONE ? tfloor_one : tfs[lc[k]])));
        const double lf = ONE ? logfloor_one : a.logfloor_c[lc[k]];
        double add = floored ? lf : m[k].y - log2T;
        add = counted ? add : 0.0;
        add = poison ? __builtin_nan("") : add;
        lsum += add;
        zeros += (int)floored;
        if (counted && last_set) __builtin_nontemporal_store(t, (double*)(probs + (base + k * stride) * 8u));
      }
      if (GEN && __any(skip_bits != 0)) {  // as paired_static4_body
#pragma unroll 1
        for (int k = 0; k < 4; k++)
          if ((skip_bits >> k) & 1u) { const GenOut o = compact_general_call(kernel_args_address(), (int)(base + k * stride), s); lsum += o.add; zeros += o.zeros; }
      }
      acc_s[s * kBlock + threadIdx.x] = lsum;
      acc_z[s * kBlock + threadIdx.x] = zeros;
    }
  }
}

// Classes 1 and 2 with the path sets in the inner loop (paired_regs_body's pairs, lane -> pair mapping and order of
// additions): records once, set 0 resolved and captured, later sets finished from the capture unless one of the
// pair's windows changed.
template <int K, bool GEN>
__device__ __forceinline__ void paired_regs_multi_body(const PairedArgs& a, const MultiSets& ms, int lb, int slot_lo, int slot_hi, int block_lo,
                                                       int block_hi, double* acc_s, int* acc_z, const double* tf) {
  for (int i = slot_lo + (lb - block_lo) * kBlock + threadIdx.x; i < slot_hi; i += (block_hi - block_lo) * kBlock) {
    const uint32_t l12 = a.len12[i - a.n0];
    const size_t at = K == 2 ? (size_t)2 * (i - a.n0) : (size_t)2 * (a.n01 - a.n0) + (size_t)4 * (i - a.n01);
    int4 r1[K], r2[K];
#pragma unroll
    for (int k = 0; k < K; k++) { r1[k] = a.inl[0][at + k]; r2[k] = a.inl[1][at + k]; }
    const bool dirty = r1[0].x == kDirtyWid;  // scored from the delta lists
    unsigned chg = ms.chg[0] ? 0u : 0xffu;
    if (ms.chg[0] && !dirty) {
#pragma unroll
      for (int k = 0; k < K; k++) chg |= (r1[k].x >= 0 ? ms.chg[0][r1[k].x] : 0u) | (r2[k].x >= 0 ? ms.chg[1][r2[k].x] : 0u);
    }
    PairVal val{0.0, 0.0, 0, 0};
    bool general = false;
#pragma unroll 1
    for (int s = 0; s < ms.n; s++) {
      const PairedArgs b = with_set(a, ms.set[s], tf ? tf + s * kTfCodes : nullptr);
      double lsum = acc_s[s * kBlock + threadIdx.x];
      int zeros = acc_z[s * kBlock + threadIdx.x];
      if (s == 0 || ((chg >> s) & 1u)) {
        RegCands<K> x, y;
        const bool m1 = cands_from_records<K>(b.m[0], r1, x), m2 = cands_from_records<K>(b.m[1], r2, y);
        general = !dirty && (m1 || m2);  // a window that occurs several times: general_pair_call
        val.kind = 0;
        if (!dirty && !general) score_cands_and_finish<K>(b, i, l12, x, y, lsum, zeros, &val);
      } else {
        finish_val(b, i, val, s == ms.n - 1, lsum, zeros);
      }
      if (GEN && general) { const GenOut o = general_pair_call(kernel_args_address(), i, -1, s, nullptr); lsum += o.add; zeros += o.zeros; }  // (in every set: val holds nothing of such a pair)
      acc_s[s * kBlock + threadIdx.x] = lsum;
      acc_z[s * kBlock + threadIdx.x] = zeros;
    }
  }
}

// paired_delta_body with the path sets in the inner loop
template <bool GEN>
__device__ __forceinline__ void paired_delta_multi_body(const PairedArgs& a, const MultiSets& ms, int db, int delta_blocks, double* acc_s, int* acc_z, const double* tf) {
  for (int dj = db * kBlock + threadIdx.x; dj < a.dstate[kDsDirty]; dj += delta_blocks * kBlock) {
    const int i = a.dirty_slots[dj];
    const int sp = a.dirty_spill[dj];
    const bool mine = sp < 0;  // (sp >= 0, a long list: one WAVE scores it. Such a lane stays in the loop: the note words are
                               // written by the wave's lane 0 from a ballot over ALL its lanes)
    int4 r0[4], r1[4];
    int c0 = 0, c1 = 0;
    unsigned chg = ms.chg[0] ? 0u : 0xffu;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      r0[k] = a.dirty_recs[0][4 * (size_t)dj + k];
      r1[k] = a.dirty_recs[1][4 * (size_t)dj + k];
      c0 += r0[k].x >= 0; c1 += r1[k].x >= 0;
      if (ms.chg[0]) chg |= (r0[k].x >= 0 ? ms.chg[0][r0[k].x] : 0u) | (r1[k].x >= 0 ? ms.chg[1][r1[k].x] : 0u);
    }
    const uint32_t l12 = (uint32_t)r0[0].w;  // the pair's read lengths travel with its first record (paired_upload_delta)
    PairVal val{0.0, 0.0, 0, 0};
    bool general = false;  // as resolved last: a set whose tables agree with its predecessor's on this pair's windows inherits it
#pragma unroll 1
    for (int s = 0; s < ms.n; s++) {
      const PairedArgs b = with_set(a, ms.set[s], tf ? tf + s * kTfCodes : nullptr);
      double lsum = acc_s[s * kBlock + threadIdx.x];
      int zeros = acc_z[s * kBlock + threadIdx.x];
      if (!mine) {
      } else if (s == 0 || ((chg >> s) & 1u)) {
        RegCands<4> x, y;
        const bool m0 = cands_from_records<4>(b.m[0], r0, x), m1 = cands_from_records<4>(b.m[1], r1, y);
        general = m0 || m1;  // a window that occurs several times in this path set: general_pair_call (as paired_delta_body)
        val.kind = 0;
        if (!general) score_cands_and_finish<4>(b, i, l12, x, y, lsum, zeros, &val);
        else if (!GEN) lsum += __builtin_nan("");  // cannot happen (a launch without notes has no such window): poisoned, reported by combine()
      } else {
        finish_val(b, i, val, s == ms.n - 1, lsum, zeros);
      }
      if (GEN && mine && general) { const GenOut o = general_pair_call(kernel_args_address(), i, dj, s, nullptr); lsum += o.add; zeros += o.zeros; }
      acc_s[s * kBlock + threadIdx.x] = lsum;
      acc_z[s * kBlock + threadIdx.x] = zeros;
    }
  }
}

// the sets (bits) in which one of a pair's windows changed, over all its records of one mate: lanes stride, wave OR
template <class Src>
__device__ __forceinline__ unsigned wave_changed(const Src& src, const unsigned char* chg, int lane) {
  unsigned m = 0;
  const int cnt = src.count();
  for (int k = lane; k < cnt; k += 64) { const int4 r = src.get(k); if (r.x >= 0) m |= chg[r.x]; }
  for (int off = 32; off > 0; off >>= 1) m |= __shfl_xor(m, off, 64);
  return m;
}

// paired_overflow_body with the path sets in the inner loop: wave w keeps its running sums per set in acc (lane 0's)
__device__ __forceinline__ void paired_overflow_multi_body(const PairedArgs& a, const MultiSets& ms, int ovf_block, int ovf_blocks,
                                                           int4 (*cand)[2][kOvfCap], double* acc_s, int* acc_z, const double* tf) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_global = ovf_block * (kBlock / 64) + wave;
  const int n_waves = ovf_blocks * (kBlock / 64);
  const int n_table = a.n - a.n_main;
  const int n_items = n_table + a.dstate[kDsSpill];
  int4* c1 = cand[wave][0];
  int4* c2 = cand[wave][1];
  for (int item = wave_global; item < n_items; item += n_waves) {  // fixed item -> wave assignment
    int i, L1, L2;
    TableSrc t1{&a.m[0], make_int4(-1, 0, 0, 0)}, t2{&a.m[1], make_int4(-1, 0, 0, 0)};
    ListSrc l1{nullptr, 0}, l2{nullptr, 0};
    const bool table = item < n_table;
    if (table) {
      i = a.n_main + item;
      t1.r0 = a.m[0].first[i - a.n0]; t2.r0 = a.m[1].first[i - a.n0];
      const uint32_t l12 = a.len12[i - a.n0];
      if (t1.r0.x == kDirtyWid) continue;  // a class-3 pair that is on the delta list: scored there
      L1 = l12 & 0xffff; L2 = l12 >> 16;
    } else {
      const int sp = item - n_table;
      i = a.spill_slot[sp];
      const uint32_t l12 = i < a.n0 ? a.len_combo[a.len_code[i]] : a.len12[i - a.n0];
      L1 = l12 & 0xffff; L2 = l12 >> 16;
      const int2 g0 = a.spill_rng[0][sp], g1 = a.spill_rng[1][sp];
      l1 = ListSrc{a.spill_recs[0] + g0.x, g0.y};
      l2 = ListSrc{a.spill_recs[1] + g1.x, g1.y};
    }
    unsigned chg = 0xffu;
    if (ms.chg[0]) chg = table ? (wave_changed(t1, ms.chg[0], lane) | wave_changed(t2, ms.chg[1], lane)) : (wave_changed(l1, ms.chg[0], lane) | wave_changed(l2, ms.chg[1], lane));
    PairVal val{0.0, 0.0, 0, 0};
    for (int s = 0; s < ms.n; s++) {
      const PairedArgs b = with_set(a, ms.set[s], tf ? tf + s * kTfCodes : nullptr);
      double lsum = 0.0;
      int zeros = 0;
      if (s == 0 || ((chg >> s) & 1u)) {  // wave-uniform
        TableSrc u1{&b.m[0], t1.r0}, u2{&b.m[1], t2.r0};
        const double acc = table ? wave_score_pair(b, u1, u2, L1, L2, c1, c2, lane) : wave_score_pair(b, l1, l2, L1, L2, c1, c2, lane);
        if (lane == 0) finish_read(b, i, acc, L1, L2, lsum, zeros, &val);
      } else if (lane == 0) {
        finish_val(b, i, val, s == ms.n - 1, lsum, zeros);
      }
      if (lane == 0) { acc_s[s * (kBlock / 64) + wave] += lsum; acc_z[s * (kBlock / 64) + wave] += zeros; }
    }
  }
}

template <bool GEN>
__global__ __launch_bounds__(kBlock, 5) void paired_score_multi_kernel(PairedArgs a, MultiSets ms) {
  __shared__ double sh_s[kBlock / 64];
  __shared__ int sh_z[kBlock / 64];
  // the lane-per-pair blocks keep one running sum per (set, thread) here (8 sets x 256 threads x (8 + 4) B = 24 KB);
  // the wave-per-pair blocks stage candidates here (16 KB) and keep one running sum per (set, wave) behind them:
  // block-uniform roles, one buffer
  __shared__ __align__(16) unsigned char sh_raw[kMaxSets * kBlock * 12];
  constexpr size_t kCandBytes = sizeof(int4) * (kBlock / 64) * 2 * kOvfCap;
  static_assert(kCandBytes + kMaxSets * (kBlock / 64) * 12 <= sizeof(sh_raw), "candidate staging + per-wave sums must fit");
  const int lb = a.total_blocks - 1 - (int)blockIdx.x;
  // every set's thresholds per length code: read once per block (they sit in host-written device memory, a round trip
  // each), not once per set and pair
  __shared__ double sh_tf[kMaxSets * kTfCodes];
  const double* tf = a.n_codes <= kTfCodes ? sh_tf : nullptr;
  if (tf && threadIdx.x < kMaxSets * kTfCodes) {
    const int s = threadIdx.x / kTfCodes, k = threadIdx.x % kTfCodes;
    sh_tf[threadIdx.x] = s < ms.n && k < a.n_codes ? ms.set[s].tfloor_c[k] : 0.0;
  }
  if (lb < a.main_blocks) {
    double* acc_s = (double*)sh_raw;
    int* acc_z = (int*)(sh_raw + kMaxSets * kBlock * 8);
    for (int s = 0; s < ms.n; s++) { acc_s[s * kBlock + threadIdx.x] = 0.0; acc_z[s * kBlock + threadIdx.x] = 0; }
    __syncthreads();
    const int cls = lb < a.blocks0 ? 0 : lb < a.blocks01 ? 1 : lb < a.blocks012 ? 2 : 3;
    if ((ms.pad_ >> cls) & 1) {  // timing experiments (tools/): a class of blocks left out, results wrong
      if (threadIdx.x == 0) for (int s = 0; s < ms.n; s++) { ms.set[s].part_sum[lb] = 0.0; ms.set[s].part_zero[lb] = 0; }
      return;
    }
    if (lb < a.blocks0) {
      // (both parts of class 0 resolve every pair per set here: the lanes, pairs and order of additions are the single-set
      // kernel's, and so are the values -- a static memo index is the index the per-call arithmetic arrives at)
      const SlotRange rg = compact_range(a, lb);
      // the static part streams its values like the single-set kernel (same condition as there: memo present, no coverage marks)
      const bool stat = lb < a.blocks0a && a.memo && !a.cov_bits && a.static_val;
      if (a.n_codes == 1) {
        if (stat) paired_static4_multi_body<GEN, true>(a, ms, rg, acc_s, acc_z, tf);
        else paired_compact4_multi_body<GEN, true>(a, ms, rg, acc_s, acc_z, tf);
      } else {
        __shared__ uint32_t sh_combo[256];
        __shared__ double sh_logfloor[256];
        for (int k = threadIdx.x; k < a.n_codes; k += kBlock) { sh_combo[k] = a.len_combo[k]; sh_logfloor[k] = a.logfloor_c[k]; }
        __syncthreads();
        PairedArgs b = a;
        b.len_combo = sh_combo; b.logfloor_c = sh_logfloor;
        if (stat) paired_static4_multi_body<GEN, false>(b, ms, rg, acc_s, acc_z, tf);
        else paired_compact4_multi_body<GEN, false>(b, ms, rg, acc_s, acc_z, tf);
      }
    } else if (lb < a.blocks01) paired_regs_multi_body<2, GEN>(a, ms, lb, a.n0, a.n01, a.blocks0, a.blocks01, acc_s, acc_z, tf);
    else if (lb < a.blocks012) paired_regs_multi_body<4, GEN>(a, ms, lb, a.n01, a.n_main, a.blocks01, a.blocks012, acc_s, acc_z, tf);
    else paired_delta_multi_body<GEN>(a, ms, lb - a.blocks012, a.main_blocks - a.blocks012, acc_s, acc_z, tf);
    for (int s = 0; s < ms.n; s++) {
      double lsum = acc_s[s * kBlock + threadIdx.x];
      int zeros = acc_z[s * kBlock + threadIdx.x];
      block_reduce(lsum, zeros, sh_s, sh_z);
      if (threadIdx.x == 0) { ms.set[s].part_sum[lb] = lsum; ms.set[s].part_zero[lb] = zeros; }
      __syncthreads();  // sh_s / sh_z are reused by the next set
    }
    return;
  }
  if ((ms.pad_ >> 4) & 1) {
    if (threadIdx.x == 0) for (int s = 0; s < ms.n; s++) { ms.set[s].part_sum[lb] = 0.0; ms.set[s].part_zero[lb] = 0; }
    return;
  }
  double* acc_s = (double*)(sh_raw + kCandBytes);
  int* acc_z = (int*)(sh_raw + kCandBytes + kMaxSets * (kBlock / 64) * 8);
  if (threadIdx.x < kMaxSets * (kBlock / 64)) { acc_s[threadIdx.x] = 0.0; acc_z[threadIdx.x] = 0; }
  __syncthreads();
  paired_overflow_multi_body(a, ms, lb - a.main_blocks, a.total_blocks - a.main_blocks, (int4(*)[2][kOvfCap])sh_raw, acc_s, acc_z, tf);
  __syncthreads();
  for (int s = 0; s < ms.n; s++) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double lsum = lane == 0 ? acc_s[s * (kBlock / 64) + wave] : 0.0;
    int zeros = lane == 0 ? acc_z[s * (kBlock / 64) + wave] : 0;
    block_reduce(lsum, zeros, sh_s, sh_z);
    if (threadIdx.x == 0) { ms.set[s].part_sum[lb] = lsum; ms.set[s].part_zero[lb] = zeros; }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// coverage penalty sweep (graph.cc:1893-1919) over the bitmap of marked path positions.
// A marked position p (not itself a contig start) adds p - q to bad_bases, q = previous marked
// position of the same path, when no contig start lies in (q, p], p - q > cov_move and
// p - (contig start before p) > mean + 5 sd.
// ---------------------------------------------------------------------------------------
struct CovArgs {
  const uint32_t* bits;
  const int* path_base;     // [n_paths+1] bit offsets (multiples of 32)
  const int* start_off;     // [n_paths+1] into starts
  const int* starts;        // contig start positions (path coordinates), ascending per path
  int n_paths;
  int total_words;
  double cov_move;
  double far;               // insert_mean + 5*insert_std
  unsigned long long* bad;  // out
};

__global__ __launch_bounds__(kBlock) void coverage_sweep_kernel(CovArgs a) {
  for (int w = blockIdx.x * kBlock + threadIdx.x; w < a.total_words; w += gridDim.x * kBlock) {
    uint32_t word = a.bits[w];
    if (!word) continue;
    // locate the path of this word (few paths: binary search)
    int lo = 0, hi = a.n_paths - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (a.path_base[mid] <= w * 32) lo = mid; else hi = mid - 1; }
    const int base = a.path_base[lo], base_word = base >> 5;
    const int* st = a.starts + a.start_off[lo];
    const int nst = a.start_off[lo + 1] - a.start_off[lo];
    unsigned long long local = 0;
    uint32_t rest = word;
    while (rest) {
      const int b = __ffs(rest) - 1;
      rest &= rest - 1;
      const int p = w * 32 + b - base;
      // previous marked position q
      int q = -1;
      uint32_t below = word & ((1u << b) - 1);
      if (below) q = w * 32 + (31 - __clz(below)) - base;
      else {
        for (int v = w - 1; v >= base_word; v--) {
          uint32_t x = a.bits[v];
          if (x) { q = v * 32 + (31 - __clz(x)) - base; break; }
        }
      }
      if (q < 0) continue;  // previous event is the path start (type 1)
      // largest contig start <= p
      int l2 = 0, h2 = nst - 1;
      while (l2 < h2) { int mid = (l2 + h2 + 1) >> 1; if (st[mid] <= p) l2 = mid; else h2 = mid - 1; }
      const int lb = st[l2];
      if (lb > q) continue;  // a contig start in (q, p]: previous event has type 1 (or p is a start)
      if ((double)(p - q) > a.cov_move && (double)(p - lb) > a.far) local += (unsigned long long)(p - q);
    }
    if (local) atomicAdd(a.bad, local);
  }
}

// ---------------------------------------------------------------------------------------
// single-end scorer (graph.cc:1650-1743): probs_i = sum over distinct absolute positions of
// m^e * M^(L-e); later record at the same position overwrites (graph.cc:633-644).
// ---------------------------------------------------------------------------------------
struct SingleArgs {
  MateView m;
  const int* lens;
  const double* floor_tab;     // exp(c + k*L)
  const double* logfloor_tab;
  double two_T;
  int n;
  double* probs;
  double* part_sum; int* part_zero; unsigned* ticket; double* out;
  double n_reads;
};

__global__ __launch_bounds__(kBlock) void single_score_kernel(SingleArgs a) {
  __shared__ double sh_s[kBlock / 64];
  __shared__ int sh_z[kBlock / 64];
  double lsum = 0.0;
  int zeros = 0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.n; i += gridDim.x * kBlock) {
    const int4 r0 = a.m.first[i];
    const int L = a.lens[i];
    double acc = 0.0;
    // One record whose window occurs at most once in the scored paths -- nearly every read of an assembly without
    // repeats -- is its own only candidate: nothing can overwrite it (graph.cc:631-641), no second pass over the candidates.
    bool single_cand = false;
    if (r0.x >= 0 && ((unsigned)r0.z >> 9) == 0) {
      const int4 o = mate_occ(a.m, r0.x);
      if (o.z < 0) single_cand = true;  // the window is not part of the scored paths: the read scores nothing here
      else if (o.w >= 0) { single_cand = true; const int e = r0.z & 0xff; acc = a.m.mism_pow[e] * a.m.match_pow[L - e]; }
    }
    if (r0.x >= 0 && !single_cand) {
      for_each_cand(a.m, r0, [&](const Cand& x) {
        // positions are absolute here (path index * 1e6 folded into shift); all paths share one map
        bool live = true;
        for_each_cand(a.m, r0, [&](const Cand& d) {
          if (d.pos == x.pos && (d.rank > x.rank || (d.rank == x.rank && d.k > x.k))) live = false;
        });
        if (live) acc += a.m.mism_pow[x.edit] * a.m.match_pow[L - x.edit];
      });
    }
    a.probs[i] = acc;
    const double p = acc / a.two_T;  // GetTotalProb single (graph.cc:1526-1534)
    if (p < a.floor_tab[L]) { zeros++; lsum += a.logfloor_tab[L]; }
    else lsum += log(p);
  }
  block_reduce(lsum, zeros, sh_s, sh_z);
  // bad_bases of the single-end scorer is identically 0 (graph.cc:1701-1733)
  grid_finish(lsum, zeros, blockIdx.x, gridDim.x, a.part_sum, a.part_zero, a.ticket, a.out, 0.0, a.n_reads, sh_s, sh_z);
}

// ---------------------------------------------------------------------------------------
// PacBio scorer (graph.cc:3052-3088, 3223): per read log-sum-exp over its cached alignments,
// each counted once per occurrence of its sub-walk in the scored paths; floor; sum.
// One WAVE per read: lanes stride over the read's records (coalesced 8-B logprob loads),
// each lane folds its share with the reference's pairwise rule max + log1p(exp(min-max)),
// then the 64 partial values are combined by a butterfly of the same rule.
// ---------------------------------------------------------------------------------------
struct PacbioArgs {
  const int* rec_off;        // [n+1] read-major CSR
  const int* rec_walk;       // sub-walk id per record
  const double* rec_logp;    // log probability per record
  const int* walk_count;     // occurrences of each sub-walk in the scored paths
  const int* lens;
  double floor_a, floor_b;   // log(exp(min_prob_start)), log(exp(min_prob_per_base)) (graph.cc:3075-3076)
  int n;
  double* logprobs;          // out: per-read log probability (-inf when no alignment)
  double* part_sum; int* part_zero; unsigned* ticket; double* out;
  double n_reads, bad_bases;
};

__device__ __forceinline__ double lse2(double a, double b) {  // logdouble operator+ (logdouble.hpp:37-47)
  const double ninf = -__builtin_huge_val();
  if (a == ninf) return b;
  if (b == ninf) return a;
  const double hi = fmax(a, b), lo = fmin(a, b);
  return hi + log1p(exp(lo - hi));
}

__global__ __launch_bounds__(kBlock) void pacbio_score_kernel(PacbioArgs a) {
  __shared__ double sh_s[kBlock / 64];
  __shared__ int sh_z[kBlock / 64];
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * kBlock) >> 6;
  double lsum = 0.0;
  int zeros = 0;
  for (int i = wave_global; i < a.n; i += n_waves) {
    const int b = a.rec_off[i], e = a.rec_off[i + 1];
    double v = -__builtin_huge_val();
    for (int k = b + lane; k < e; k += 64) {
      const int c = a.walk_count[a.rec_walk[k]];
      const double lp = a.rec_logp[k];
      for (int t = 0; t < c; t++) v = lse2(v, lp);
    }
    for (int off = 32; off > 0; off >>= 1) v = lse2(v, __shfl_xor(v, off, 64));
    if (lane == 0) {
      a.logprobs[i] = v;
      const double floor_lp = a.floor_a + a.floor_b * (double)a.lens[i];
      if (v < floor_lp) { zeros++; v = floor_lp; }
      lsum += v;
    }
  }
  block_reduce(lsum, zeros, sh_s, sh_z);
  grid_finish(lsum, zeros, blockIdx.x, gridDim.x, a.part_sum, a.part_zero, a.ticket, a.out, a.bad_bases, a.n_reads, sh_s, sh_z);
}

// mark the slots of delta pairs in the record tables (idempotent; runs before the scoring kernel)
__global__ __launch_bounds__(kBlock) void mark_dirty_kernel(const int* slots, int n, unsigned long long* rec8_0, int n0, int4* inl_0,
                                                            int n01, int n_main, int4* first_0) {
  for (int t = blockIdx.x * kBlock + threadIdx.x; t < n; t += gridDim.x * kBlock) mark_dirty_slot(slots[t], rec8_0, n0, inl_0, n01, n_main, first_0);
}

// ---------------------------------------------------------------------------------------------------------
// A batch's per-set occurrence tables, built on the device: set g = the resident tables (the path set of the
// previous call) + the entries that changed from set to set, patches of sets 0..g applied in order. The host
// writes only the patches (a few dozen 16-byte entries per candidate) instead of every set's whole tables.
// ---------------------------------------------------------------------------------------------------------
struct BatchPatch { int32_t w; uint32_t lo, hi; int32_t rank; };  // table entry w := {lo, hi, rank}
struct BatchTabArgs {
  const char* base;      // the resident tables
  char* regions;         // set g at regions + g * stride, laid out like the resident tables
  size_t stride;
  size_t off_occ[2], bytes_occ[2], off_lo[2], bytes_lo[2], off_m[2], bytes_m[2];  // per mate; byte counts are multiples of 4
  const BatchPatch* patches;
  const int* patch_off;  // patches of (set g, mate mt): [patch_off[2 g + mt], patch_off[2 g + mt + 1])
  int first;             // first set of this launch
  int n_sets;            // sets of this launch
  unsigned char* chg[2]; // per mate: MultiSets::chg of this launch (one byte per table entry), written by the last blocks
  size_t chg_bytes[2];   // multiples of 16
};

__device__ __forceinline__ void block_copy_words(char* dst, const char* src, size_t bytes) {  // both 16-byte aligned
  const size_t n16 = bytes / 16;
  for (size_t i = threadIdx.x; i < n16; i += blockDim.x) ((int4*)dst)[i] = ((const int4*)src)[i];
  const size_t done = n16 * 16;
  for (size_t i = done / 4 + threadIdx.x; i < bytes / 4; i += blockDim.x) ((int*)dst)[i] = ((const int*)src)[i];
}

__global__ __launch_bounds__(1024) void batch_tables_kernel(BatchTabArgs a) {  // grid (sets of this launch + 1, 2 mates)
  const int mt = (int)blockIdx.y;
  if ((int)blockIdx.x == a.n_sets) {
    // which of this launch's sets may differ from its first one, per table entry: set s differs in the entries its own
    // patch and the patches of the sets between them name -- bits s .. n-1 for every entry of patch first + s
    int4* z = (int4*)a.chg[mt];
    for (size_t i = threadIdx.x; i < a.chg_bytes[mt] / 16; i += blockDim.x) z[i] = make_int4(0, 0, 0, 0);
    __syncthreads();
    for (int sidx = 1; sidx < a.n_sets; sidx++) {
      const unsigned bits = (0xffu << sidx) & 0xffu;
      for (int t = a.patch_off[2 * (a.first + sidx) + mt] + (int)threadIdx.x; t < a.patch_off[2 * (a.first + sidx) + mt + 1]; t += blockDim.x) {
        const int w = a.patches[t].w;
        atomicOr((unsigned*)(a.chg[mt] + (w & ~3)), bits << (8 * (w & 3)));
      }
    }
    return;
  }
  const int g = a.first + (int)blockIdx.x;
  char* region = a.regions + (size_t)g * a.stride;
  block_copy_words(region + a.off_occ[mt], a.base + a.off_occ[mt], a.bytes_occ[mt]);
  block_copy_words(region + a.off_lo[mt], a.base + a.off_lo[mt], a.bytes_lo[mt]);
  block_copy_words(region + a.off_m[mt], a.base + a.off_m[mt], a.bytes_m[mt]);
  int* occ = (int*)(region + a.off_occ[mt]);
  for (int j = 0; j <= g; j++) {  // later sets override earlier ones
    __syncthreads();
    for (int t = a.patch_off[2 * j + mt] + (int)threadIdx.x; t < a.patch_off[2 * j + mt + 1]; t += blockDim.x) {
      const BatchPatch pt = a.patches[t];
      int* e = occ + 3 * (size_t)pt.w;
      e[0] = (int)pt.lo; e[1] = (int)pt.hi; e[2] = pt.rank;
    }
  }
}

// bad_bases of a paired set with coverage penalty: u64 counter of the sweep -> its partial slot
// (scale 0: a rank other than 0 of a sharded evaluation -- the all-reduce(sum) of the partials must
// count the value once)
__global__ void store_bad_bases_kernel(const unsigned long long* bad, double* out4, double scale) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out4[2] = scale * (double)*bad;
}

// union of the coverage maps of all ranks: own |= maps[0] | maps[1] | ...
__global__ __launch_bounds__(kBlock) void or_maps_kernel(uint32_t* own, const uint32_t* maps, int n_maps, int words) {
  for (int w = blockIdx.x * kBlock + threadIdx.x; w < words; w += gridDim.x * kBlock) {
    uint32_t v = own[w];
    for (int k = 0; k < n_maps; k++) v |= maps[(size_t)k * words + w];
    own[w] = v;
  }
}

}  // namespace gaml
