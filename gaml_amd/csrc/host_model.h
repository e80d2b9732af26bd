// host_model.h -- host side of libgaml_hip: flat data model for the assembly graph, the read
// sets, the alignment-window cache and the per-evaluation window-occurrence tables that the
// HIP kernels consume. Plain C++17, no HIP types: everything the GPU needs leaves this file as
// POD arrays.
//
// Reference behaviour mirrored here (file:line under the reference tree):
//   * which windows get aligned, and when           graph.cc:447-533
//   * what a window's string is                     graph.cc:846-857
//   * max-hash candidate lookup                     graph.cc:1243-1348
//   * seed extension accept set / edit counts       graph.cc:730-837
//   * how window records become path positions      graph.cc:535-649
// The data structures are NOT the reference's (hash maps of vectors per call); see DESIGN.md.
#pragma once
#include <climits>
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/gaml_hip.h"

namespace gaml {

constexpr int kTail = 300;  // window tail length, kMinSubpathLength graph.cc:27
constexpr int kSeed = 15;   // kIndexKmer graph.cc:33

struct GraphStore {
  std::string bases;
  std::vector<int64_t> off;   // n+1
  std::vector<int32_t> norm;  // normalize_map graph.h:247-266
  int32_t n() const { return (int32_t)off.size() - 1; }
  int32_t len(int32_t node) const { return (int32_t)(off[node + 1] - off[node]); }
  const char* seq(int32_t node) const { return bases.data() + off[node]; }
  void finish();
  bool load_lastgraph(const std::string& file, std::string* err);
};

struct WalkHasher {
  size_t operator()(const std::vector<int32_t>& w) const {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ w.size();
    for (int32_t x : w) { h ^= (uint32_t)x; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 29; }
    return (size_t)h;
  }
};
using Walk = std::vector<int32_t>;

// One cached window: records live in ShortMate::pool[first, first+count), sorted by
// (position, read_id) like the reference's per-window vector.
struct Window {
  int64_t first = 0;          // where its records start in ShortMate::pool; -1: they exist on the device only (dfirst)
  int64_t dfirst = -1;        // ... in the device pool of a paired set (MateDev::pool); -1: not there (yet)
  int32_t count = 0;
  int32_t max_pos = INT_MIN;  // largest record position among THIS shard's reads (INT_MIN when empty)
  int32_t global_max_pos = INT_MIN;  // ... among all shards' reads (== max_pos until exchanged)
  bool active = false;        // has occurred in a scored path set: its records are on the device
  bool pending = false;       // registered, records not computed yet (batched alignment)
  // what build_pair_tables needs of the walk (set at registration, ShortMate::tag_window):
  int32_t solo = -1;          // single-node window: the node
  int32_t head = -1;          // window of several nodes whose first node is longer than kTail: that node. Wherever this
                              // window is looked up, the node's own window is looked up right after it (graph.cc:563-566)
  int32_t peer = -2;          // id of the window with the SAME node walk in the other mate of a paired set (-1: it has none
                              // yet, -2: not looked up yet): link_mate_windows, read by build_pair_tables
};

// One occurrence of a window in the path set being scored.
struct Occ {
  int32_t wid;
  int32_t shift;    // add to a record's window position to get the path position
  int32_t min_pos;  // keep a record only if its window position >= min_pos (graph.cc:577)
  int32_t path;     // index of the path (single-end: 0)
  int32_t rank;     // order in which the reference would visit this occurrence
};

// reads of one FASTQ file = the reference's ReadSet (graph.h:344-442)
struct ShortMate {
  double match = 0, mismatch = 0;
  std::vector<double> match_pow, mismatch_pow;  // pow tables graph.cc:1448-1453
  int64_t n_global = 0;
  int64_t lo = 0, hi = 0;                // shard: global read ids [lo, hi)
  std::string bases;                     // shard reads, concatenated
  std::vector<int64_t> roff;             // hi-lo+1
  std::vector<int32_t> lens;             // per shard read
  int32_t max_len = 0;
  // flat max-hash index over the shard's reads
  std::vector<uint64_t> bucket_hash;     // sorted unique
  std::vector<int32_t> bucket_off;       // size+1
  std::vector<int32_t> bucket_reads;     // local read ids, ascending inside a bucket
  int32_t index_read_len = 0;            // length of the LAST indexed read (graph.cc:1286)
  // window cache
  std::unordered_map<Walk, int32_t, WalkHasher> win_id;
  std::vector<Window> wins;
  std::vector<const Walk*> win_walk;     // window id -> its node ids (keys of win_id are stable)
  std::unordered_map<int32_t, int32_t> solo_of_node;  // node -> id of its single-node window (tag_window)
  std::vector<gaml_aligment> pool;       // read_id = LOCAL id inside the shard (host-filed windows; a paired set on a device keeps the
                                         // records of windows its kernels filed in the device pool only)
  std::vector<int32_t> filed;            // windows whose records were appended to `pool`, in pool order (mirrored to the device)
  std::vector<int32_t> unsynced;         // windows added since the last max-position exchange (sharded runs)
  std::vector<int32_t> added_log;        // windows added since the planner last looked (memo invalidation)
  std::vector<int32_t> activated_log;    // windows activated since the device tables last took them in
  uint64_t generation = 0;               // bumped whenever a window is added
  uint64_t active_generation = 0;        // bumped whenever a window is activated (device table stale)
  int64_t active_records = 0;            // records of activated windows
  int64_t windows_aligned = 0;
  // The alignment cache also holds windows no scored path has used yet (the reference aligns the
  // inverse of every junction window and the twin of every long node up front, graph.cc:474-481).
  // Only windows that occurred in some scored path set are kept in the device record table.
  void activate(int32_t wid) {
    Window& w = wins[wid];
    if (!w.active) { w.active = true; active_records += w.count; if (w.count) { active_generation++; activated_log.push_back(wid); } }
  }

  // A window that occurs in a scored path set: on the device (activated if it was not) and marked as used. At a
  // table rebuild the windows no path set has used since the previous rebuild leave the device tables again
  // (retire_unused): a long run does not drag the records of every junction it ever tried through every evaluation.
  std::vector<uint8_t> used;             // per window: occurred in a path set since the last rebuild
  void touch(int32_t wid) {
    if (!wins[wid].active) activate(wid);
    if (used.size() <= (size_t)wid) used.resize(wins.size(), 0);
    used[wid] = 1;
  }
  int64_t retire_unused();               // returns the number of windows retired; clears the marks

  int64_t n_local() const { return hi - lo; }
  const char* read(int64_t local) const { return bases.data() + roff[local]; }
  void set_reads(int64_t n_global_, int64_t lo_, int64_t hi_, const char* b, const int64_t* offs_global);
  void build_index();
  int32_t find(const Walk& w) const { auto it = win_id.find(w); return it == win_id.end() ? -1 : it->second; }
  int32_t add_window(const GraphStore& g, const Walk& w, std::vector<gaml_aligment>& recs_sorted_local);
  void tag_window(const GraphStore& g, int32_t wid, const Walk& w);
  // the library's own aligner (AlignSubpathInternal graph.cc:839-899). With defer_alignment the
  // window is only registered (cache membership is what the registration rules look at); its
  // records are computed later for the whole batch: flush_pending_cpu, or the GPU aligner.
  int32_t align(const GraphStore& g, const Walk& w);
  bool defer_alignment = false;
  std::vector<int32_t> pending;          // registered windows without records, in id order
  void cpu_align_records(const GraphStore& g, const Walk& w, std::vector<gaml_aligment>& recs) const;
  void finalize_window(int32_t wid, std::vector<gaml_aligment>& recs_sorted_local);
  void flush_pending_cpu(const GraphStore& g);
  std::string window_string(const GraphStore& g, const Walk& w, int32_t* offset) const;
};

// seed extension; returns false when no alignment within 3+3 errors exists (ProcessHit semantics)
struct Extension { int32_t errs, begin, end; };
bool extend_seed(int32_t win_pos, int32_t read_pos, const char* read, int32_t rlen, const char* win, int32_t wlen,
                 Extension* out);

// sliding maximum of the seed code over read-length spans (graph.cc:1289-1323)
void span_maxima(const char* s, int32_t n, int32_t read_len, std::vector<std::pair<uint64_t, int32_t>>& out);
uint64_t read_max_hash(const char* s, int32_t n);  // graph.cc:1254-1269

// window registration rules
void register_for_paths(const GraphStore& g, ShortMate& m, const std::vector<Walk>& paths);  // graph.cc:447-493
void register_for_contig(const GraphStore& g, ShortMate& m, const int32_t* ctg, int32_t n);  // graph.cc:495-533, 538-542

// occurrence lists
// paired, pass 1 (structure): which cached windows sit where on a contig placed at path coordinate
// `st` (GetPositionsOnlyPath graph.cc:544-573). Every cached window is listed, also one without
// records in this shard: its global_max_pos still feeds the position filter.
struct Placement { int32_t wid, shift, path, contig, node; };
void placements_paired_contig(const GraphStore& g, const ShortMate& m, const int32_t* ctg, int32_t n, int32_t st,
                              int32_t path, int32_t contig_serial, std::vector<Placement>& out);
// paired, pass 2 (thresholds): the max_pos - 5 filter of graph.cc:577 as a prefix maximum over the
// nodes of each contig, from the windows' GLOBAL largest positions; activates the windows used.
void occurrences_from_placements(ShortMate& m, const std::vector<Placement>& pl, std::vector<Occ>& out);
// single: contig at absolute coordinate `st` (AddPositions graph.cc:611-647)
void occurrences_single_contig(const GraphStore& g, ShortMate& m, const int32_t* ctg, int32_t n, int32_t st,
                               int32_t* rank, std::vector<Occ>& out);

// read-major record table for the device: every shard read's records in (window id, index in
// window) order. first[i] holds the read's first record (wid = -1 when it has none) plus the
// count / start of its remaining records in extra[].
struct RecQuad { int32_t wid, pos, flags, link; };  // flags = edit | orient<<8 | extra_count<<9 ; link = extra start
struct ReadMajor {
  std::vector<RecQuad> first;  // n_local
  std::vector<RecQuad> extra;
  uint64_t built_generation = ~0ull;
  int64_t total_records = 0;
};
// slot_of_read maps a local read id to its device slot (identity when null)
void build_read_major(const ShortMate& m, const std::vector<int32_t>* slot_of_read, ReadMajor& out);
// Compact tables of class 0 (at most one record per mate, everything within the packed ranges):
//   rec8  = window id (24 bits) | position in window (28) | edit distance (6) | orientation (1);
//           all ones = the mate has no record
//   occ8  = shift (int32) | min_pos clamped to int16 | path (15 bits) | "use the general path" (1)
// Class-0 pairs are additionally ordered by (window of mate 1, window of mate 2): the lanes of a wave
// then mostly share their window, so the occurrence lookups are wave-broadcasts.
// (The tables themselves are built and kept on the device: table_build.hip.h. PairTables / build_pair_tables below are the
// host restatement the DEVELOPMENT build checks them against -- gaml_hip_debug_tables_check -- and are not in the product.)
constexpr uint64_t kNoRec8 = ~0ull;
inline bool rec8_fits(int32_t wid, int32_t pos, int32_t edit) { return wid >= 0 && wid < (1 << 24) - 1 && pos >= 0 && pos < (1 << 28) && edit >= 0 && edit < 64; }
inline uint64_t rec8_pack(int32_t wid, int32_t pos, int32_t edit, int32_t orient) {
  return (uint64_t)(uint32_t)wid | ((uint64_t)(uint32_t)pos << 24) | ((uint64_t)(uint32_t)edit << 52) | ((uint64_t)(orient & 1) << 58);
}
#ifndef GAML_FOLD_CLASS2_BELOW
#define GAML_FOLD_CLASS2_BELOW 1024
#endif
constexpr int64_t kFoldClass2Below = GAML_FOLD_CLASS2_BELOW;  // fewer pairs than this with 3-4 records per mate: scored one wave per pair (class 3)
constexpr int kMemoCodes = 4;
constexpr int32_t kStaticZero = -2;  // PairTables::static_idx of a pair that never scores
constexpr size_t kMemoMaxEntries = (size_t)1 << 24;
#ifdef GAML_HIP_DEV
struct PairTables {
  std::vector<int32_t> slot_of_read, read_of_slot;
  int64_t class_count[4] = {0, 0, 0, 0};  // 0: compact; 1: <= 2 records; 2: <= 4; 3: more
  // Class 0 comes in two parts. Slots [0, n0a): both mates' records sit in the SAME window (same node walk) and the pair's
  // term is covered by the memo of pair terms -- wherever that window occurs in a path set the two alignments get the same
  // shift, so orientation rule, insert distance and with them the memo index (graph.cc:1864-1882) do not depend on the path
  // set: static_idx[slot] holds it, computed here once. The path set only decides WHETHER the pair scores (the window
  // occurs, position filter graph.cc:577). A pair with a mate that has no alignment at all (the max-hash aligner finds
  // ~85 % of the reads at 1 % substitutions) is there too, index kStaticZero: it scores nothing whatever the path set.
  // Slots [n0a, class_count[0]): every other class-0 pair (resolved per call).
  int64_t n0a = 0;
  std::vector<int32_t> static_idx;        // [n0a]
  std::vector<uint64_t> rec8[2];          // [n0]
  std::vector<uint8_t> len_code;          // [n0] index into len_combo
  std::vector<uint32_t> len_combo;        // distinct L1 | L2<<16 values (<= 256)
  std::vector<uint32_t> len12;            // [n - n0] for the 16-byte classes
  ReadMajor rm[2];                        // 16-byte tables of slots >= n0 (indexed slot - n0)
  // the same records once more, inline per pair, for the register paths: class 1 holds 2 slots per
  // pair and mate at [2t + k], class 2 holds 4 at [2 n1 + 4 t2 + k] (wid = -1: no record)
  std::vector<RecQuad> inl[2];
  int64_t dropped_records[2] = {0, 0};    // records of active windows left out of the tables (dominated_records)
};
// fold = false keeps the records that can never survive the overwrite rule (A/B and tests; same values either way).
// ins_n: length of the insert-size table the memo of pair terms is built over (memo index = ((code * 7 + edit 1) * 7 + edit 2)
// * ins_n + distance, first kMemoCodes length combinations, edits < 7); 0: no static indices (n0a = 0).
void build_pair_tables(const ShortMate& a, const ShortMate& b, PairTables& out, bool fold = true, int ins_n = 0);
#endif  // GAML_HIP_DEV
// which windows of the two mates of a paired set hold the same node walk (Window::peer); only windows not linked yet are
// looked up. Call before build_pair_tables (on the thread that owns the window caches).
void link_mate_windows(ShortMate& a, ShortMate& b);
// The same rule for ONE window that joins the device tables later (delta lists): keep[k] = 0 for the records of window
// `wid` that its first node's own window -- active -- always overwrites. Returns the number of records to keep.
#ifdef GAML_HIP_DEV
int64_t undominated_records(const ShortMate& m, int32_t wid, std::vector<uint8_t>& keep);
#endif


// direct-mapped occurrence table for the device: one 16-B entry per window.
//   path < 0            : the window does not occur in the current path set
//   path >= 0, rank >= 0: exactly one occurrence, described by the entry
//   rank < 0            : several; they are multi[multi_off[-rank-1] .. multi_off[-rank])
struct OccQuad { int32_t shift, min_pos, path, rank; };
struct OccTable {
  std::vector<OccQuad> direct;      // per window
  std::vector<int32_t> multi_off;
  std::vector<OccQuad> multi;
};
void build_occ_table(size_t n_windows, const std::vector<Occ>& occs, OccTable& out);

void split_contigs(const Walk& path, std::vector<std::pair<int32_t, int32_t>>& ctg_ranges, std::vector<int32_t>& gaps);

struct PlanView;
// Host image of one mate's device occurrence tables (8-byte entry + 4-byte rank per window, lists for windows
// that occur several times), kept between evaluations and patched: only the entries of windows that occurred in
// the previous or occur in the current path set are touched, O(occurrences) per call instead of O(windows).
//   occ12 = {occ8, rank} interleaved; occ8: shift:32 | min_pos:16 | path:15 | general:1 (all ones: the window does not occur);
//   rank: visiting rank of the one occurrence, or -(list + 1) for a general window (its occurrences are
//         multi[multi_off[list] .. multi_off[list + 1])).
// Together they carry what the 16-byte OccQuad of the single-end path carries, in 12 bytes (the per-call upload is
// what the scoring launch waits for).
struct Occ12 { uint32_t lo, hi; int32_t rank; };  // occ8 = lo | hi << 32, then the rank: ONE 12-byte load per occurrence
struct OccImage {
  std::vector<Occ12> occ12;
  std::vector<int32_t> multi_off;
  std::vector<OccQuad> multi;
  std::vector<int32_t> general_wids;  // windows whose entry sends their reads to the general path
  // whole path set at once (every path changed, first call, coverage penalty): O(occurrences)
  void build(size_t n_windows, const PlanView& view, int mate);
  // one path's occurrences in or out (a call that shares most paths with the previous one): O(its occurrences).
  // `slot` is what the entries carry as their path (stable while the path stays in the set; only equality of the
  // two mates' paths matters to the scorers), the rank is path-local (it only ever orders occurrences of one path).
  void add_path(size_t n_windows, const std::vector<Occ>& occ, int32_t slot);
  void remove_path(const std::vector<Occ>& occ, int32_t slot);
  // after the adds / removes of a call: the lists of windows that occur several times, entries in visiting order
  // (position of the path in the set, then path-local rank) -- the order build() gives
  void finalize(const std::vector<int32_t>& pos_of_slot);
  // table entries that changed since the last take_changed() (window ids, each once); `all` = everything may have
  std::vector<int32_t> changed;
  bool changed_all = true, lists_changed = true;
  // every occurrence the tables describe, as {window, shift, min_pos (as stored: clamped at -32768 in the 8-byte form), slot, rank}
  void dump(std::vector<Occ>& out) const;
  template <class F> void for_each_present(F f) const { for (int32_t w : touched_) if (cnt_[w] > 0) f(w); }
  void take_changed() { changed.clear(); changed_all = false; lists_changed = false; if (++mark_serial_ == 0) { std::fill(mark_.begin(), mark_.end(), 0); mark_serial_ = 1; } }
 private:
  void grow(size_t n_windows);
  void mark(int32_t w) { if (mark_[w] != mark_serial_) { mark_[w] = mark_serial_; changed.push_back(w); } }
  void set_direct(int32_t w, const OccQuad& q);
  std::vector<int32_t> touched_, stale_, cnt_, list_of_;
  std::vector<uint32_t> stamp_, mark_;
  uint32_t serial_ = 0, mark_serial_ = 1;
  struct Pending { int32_t wid; OccQuad q; };
  std::vector<Pending> pending_;
  // windows that need lists (several occurrences, or one that the 8-byte form cannot hold): all their occurrences
  std::unordered_map<int32_t, std::vector<OccQuad>> gen_;
  bool lists_dirty_ = false;
};

// ---------------------------------------------------------------------------------------------
// PairedPlanner: everything the host does for one paired read set per CalcProb, memoised per
// distinct path. Replaces the reference's per-call GetChanges + hash-map position assembly
// (graph.cc:1745-1764, 535-598) with an exact scheme:
//   * a path's window placements depend only on the path, the graph and on WHICH of the window keys
//     it looks up are cached. Keys it looked up and missed are remembered; when such a key is later
//     aligned, the memos that missed it are invalidated (and only those).
//   * registration is idempotent, so a memoised path only has to redo the one rule position whose
//     outcome depends on the previous path in the list (`last_end`, graph.cc:449,471-472).
// ---------------------------------------------------------------------------------------------
// An occurrence as its 12-byte table entry without the path slot: what a whole-set build writes per occurrence
// (hi gets `slot << 16`). wid < 0 (~wid): the 8-byte form cannot hold the threshold -- the build takes the Occ instead.
struct OccPre { int32_t wid; uint32_t lo, hi; int32_t rank; };

struct PathMemo {
  Walk path;
  bool registered[2] = {false, false};   // PrecomputeAlignmentForPaths rule has run for this path
  bool first_pending[2] = {false, false};// ... but its first position was skipped because last_end == cur_end
  bool valid[2] = {false, false};        // placements reflect the cache
  bool occ_valid[2] = {false, false};    // occurrence list reflects placements + window maxima
  std::vector<Placement> pl[2];          // shifts relative to the path start, path = 0
  std::vector<Occ> occ[2];               // rank path-local, path = 0
  std::vector<OccPre> pre[2];            // the same occurrences as table entries minus the path slot (OccImage::build)
  int64_t assembled[2] = {0, 0};         // records of the occurring windows
  std::vector<int32_t> starts;           // contig start coordinates (events of type 1, graph.cc:1826,1835)
  int32_t length = 0;                    // incl. gaps
  int32_t first_idx = -1, first_end = -1;// first non-gap position and the end index of its junction window
  int32_t final_last_end = -2;           // last_end after the path (-2: path has no node, passes through)
  int32_t use_count = 0;                 // instances in the current path set (a memo in use is never evicted)
  uint32_t touched_serial = 0;           // windows retired after this: the lists' windows must be (re)activated when the path enters a set
  uint64_t last_used = 0;
  uint32_t serial = 0;                   // bumps when the slot is reused (stale ids in the miss index)
};

struct PlanView {  // what one evaluation needs from the planner
  std::vector<const PathMemo*> paths;    // in path-set order
};

class PairedPlanner {
 public:
  // pass 1: registration + placements for the path set given in the ABI's flat form. When the set shares a prefix
  // and / or suffix of paths with the previous call's (what an annealing move leaves: it edits one or two paths), only
  // the paths in between are looked at: hashing, registration replay, placements, occurrence lists and the table
  // entries that follow from them are O(changed paths). Returns false (and sets *err) on a node id outside the graph.
  // allow_incremental = false: everything from scratch (first call, coverage penalty, after the window maxima changed).
  bool begin(const GraphStore& g, ShortMate mate[2], const int32_t* flat, const int64_t* offs, int32_t n_paths,
             bool allow_incremental, std::string* err);
  void finish(ShortMate mate[2]);          // pass 2 (needs the windows' records / global maxima): occurrence lists
  // the occurrence tables of this call: whole-set rebuild or per-path adds / removes (what begin() decided)
  void apply(ShortMate mate[2], OccImage image[2]);
  // table rebuilds retire windows no path set has used since the previous rebuild: mark_used() marks the windows of
  // the current set and of every path that entered a set since the last call of note_rebuild()
  void mark_used(ShortMate mate[2], const OccImage image[2]);
  void note_rebuild(bool retired_some) { rebuild_clock_ = clock_; if (retired_some) retire_serial_++; }
  void invalidate_thresholds();          // window maxima changed (sharded cold path)
  void forget_previous() { have_prev_ = false; }  // the next begin() starts from scratch
  const PlanView& view();                // memos of the current set, in path order (built on demand)
  // flat occurrence list of the current path set (path = index in the set, rank global, visiting order)
  void flat_occurrences(int mate, std::vector<Occ>& out);
  size_t memo_count() const { return memos_.size(); }
  int32_t total_len() const { return total_len_; }          // incl. gaps (GetTotalLen graph.cc:1775-1781)
  int64_t assembled(int mate) const { return assembled_[mate]; }
  int32_t n_paths() const { return (int32_t)cur_ids_.size(); }
  bool last_was_incremental() const { return incremental_; }
  const std::vector<int32_t>& slots() const { return cur_slots_; }  // slot of every path of the current set
  uint64_t hits = 0, misses = 0, incremental_calls = 0, full_calls = 0;
 private:
  int32_t lookup_or_create(const GraphStore& g, const int32_t* p, int32_t len, std::string* err);
  void drain(ShortMate mate[2]);
  void build_placements(const GraphStore& g, ShortMate& m, int mt, PathMemo& pm, int32_t id);
  void registration_chain(const GraphStore& g, ShortMate mate[2], int32_t from, int32_t to);
  std::vector<std::unique_ptr<PathMemo>> memos_;
  std::unordered_map<Walk, int32_t, WalkHasher> by_path_;
  // window key looked up and missed -> memos to invalidate when it gets aligned
  std::unordered_map<Walk, std::vector<std::pair<int32_t, uint32_t>>, WalkHasher> missed_[2];
  PlanView view_;
  bool view_valid_ = false;
  std::vector<int32_t> cur_ids_;         // memo of every path of the current set
  std::vector<int32_t> cur_slots_;       // ... and its slot (the `path` its table entries carry)
  std::vector<int32_t> free_slots_, pos_of_slot_;
  int32_t next_slot_ = 0;
  // previous call, for the diff
  std::vector<int32_t> prev_flat_;
  std::vector<int64_t> prev_offs_;
  bool have_prev_ = false;
  // what this call changes
  bool incremental_ = false;
  struct Removed { int32_t slot; int64_t assembled[2]; std::vector<Occ> occ[2]; };
  std::vector<Removed> removed_;         // instances whose table entries go: their occurrence lists as they were when they went in
  std::vector<int32_t> work_;            // path indices whose occurrences go (back) in after pass 2, ascending
  std::vector<int32_t> stale_;           // memos invalidated while in use: refreshed at the next begin()
  std::vector<char> redo_;
  struct Edit { int32_t p0, p1, c0, c1; };  // paths [p0, p1) of the previous set are replaced by paths [c0, c1) of this one
  std::vector<Edit> edits_;              // this call's edit script, ascending
  std::vector<char> is_new_;             // per path of this call: entered with this call
  int32_t total_len_ = 0;
  int64_t assembled_[2] = {0, 0};
  uint64_t clock_ = 0, rebuild_clock_ = 0;
  uint32_t retire_serial_ = 0;
  static constexpr size_t kMaxMemos = 2048;
};
int32_t walk_length(const GraphStore& g, const Walk& w);

bool read_fastq(const std::string& file, std::string& bases, std::vector<int64_t>& offs, std::string* err,
                std::vector<std::string>* names = nullptr);

// ---------------------------------------------------------------------------------------
// PacBio cache-miss side: SAM record -> banded-DP job (host part; the DP itself runs on the GPU)
// ---------------------------------------------------------------------------------------
// One SAM alignment as the reference reads it (PacbioAligmentData graph.h:499-514, ParseAligment
// graph.cc:2945-3021): QNAME up to its last '/', FLAG, POS, CIGAR (M/I/D only), column 9 length,
// SEQ length, tags XS/XE/XQ/NM. Reverse-strand records (FLAG & 16) are moved to the second half of
// "path + separator + reverse complement" (total_len = 2|path| + 1); soft-clipped read ends become
// insertions so that the DP covers the whole read.
struct SamRecord {
  std::string name;
  int32_t flags = 0, len = 0, posstart = 0, posend = 0, sstart = 0, send = 0, slen = 0, tstart = 0, tend = 0, edit_dist = 0;
  std::vector<std::pair<int32_t, char>> cigar;
};
// false: fewer than 10 tab-separated columns (the reference would index past the end)
bool parse_sam_record(const char* begin, const char* end, int32_t total_len, SamRecord& out);

// Cell set of AligmentProbability (graph.cc:2183-2235) as one column interval per DP row:
// rows row0 .. row0+lo.size()-1, row r covers columns lo[r-row0] .. hi[r-row0]. The set is the
// CIGAR path plus the two clip boxes (<= 200), closed to full row intervals, widened by 2 in both
// directions and closed again.
struct DpBand {
  int32_t row0 = 0, max_width = 0;
  std::vector<int32_t> lo, hi;
};
void pacbio_dp_band(const std::vector<std::pair<int32_t, char>>& cigar, DpBand& out);

// What the DP kernel needs instead of the materialised band: the run-length CIGAR (appended to
// `ops` as (length << 2) | code, code 0 = M, 1 = I, 2 = D; consecutive insertions merged, empty
// operations dropped), the end of the CIGAR path, the clip boxes, and an upper bound of the row
// width (the kernel derives the band itself).
struct DpShape {
  int32_t n_ops = 0, row_f = 0, col_f = 0, bl = 0, el = 0, max_width = 0;
};
void pacbio_dp_ops(const std::vector<std::pair<int32_t, char>>& cigar, std::vector<uint32_t>& ops, DpShape& out);

}  // namespace gaml
