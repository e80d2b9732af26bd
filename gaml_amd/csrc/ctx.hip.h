// ctx.hip.h -- the library context: device / pinned buffers, per-read-set device state, gaml_hip_ctx.
// Shared by gaml_hip.hip (the C ABI, launches) and multi.hip (several device shards in one process, RCCL).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "radix_sort.hip.h"

#include <algorithm>
#include <atomic>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <climits>
#include <limits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <immintrin.h>
#include <memory>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "aligner.hip.h"
#include "aligner_small.hip.h"
#include "aligner_file.hip.h"
#include "host_model.h"
#include "kernels.hip.h"
#include "table_build.hip.h"
#include "delta_dev.hip.h"
#include "pacbio_dp.hip.h"
#include "pacbio_sweep.hip.h"
#include "internal.h"


// A/B switches and tuning knobs (gaml_hip_debug_set_knob) exist in development builds (-DGAML_HIP_DEV) only; the release
// library is compiled with every one of them at its default, 0.
#ifdef GAML_HIP_DEV
#define KNOB(c, i) ((c)->knobs[i])
#else
#define KNOB(c, i) 0
#endif

namespace gaml {
namespace detail {

inline double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// development aid (development build, GAML_HIP_PROBE=1): a 4-byte fill + stream synchronise at a few places of an evaluation; reports
// how long the device took to answer -- how the 12-25 ms stall after the cold alignment batch was found (AlignScratch::stage)
inline void gpu_probe(hipStream_t st, void* p4, const char* label) {
#ifdef GAML_HIP_DEV
  static const bool on = getenv("GAML_HIP_PROBE") != nullptr;
  if (!on || !p4) return;
  const double t0 = now_us();
  (void)hipMemsetAsync(p4, 0, 4, st);
  (void)hipStreamSynchronize(st);
  const double dt = now_us() - t0;
  fprintf(stderr, "probe %-28s %8.2f ms%s\n", label, dt * 1e-3, dt > 1000.0 ? "   <-- stall" : "");
#else
  (void)st; (void)p4; (void)label;
#endif
}

// grow-only device / pinned buffers
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
    size_t want = std::max(bytes + bytes / 4, (size_t)256);
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return (T*)p; }
};
struct PinBuf {
  void* p = nullptr;
  void* dev = nullptr;  // the same memory as the device sees it
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { (void)hipHostFree(p); p = nullptr; dev = nullptr; cap = 0; }
    size_t want = std::max(bytes + bytes / 4, (size_t)4096);
    // device-visible (kernels read / write it directly) and coherent: the host polls words the device writes (per-block
    // partials, sequence words) without any runtime call in between
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer(&dev, p, 0);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; dev = nullptr; cap = 0; }
};

constexpr int kRing = 4;  // staging slots, so that async callers may run ahead of the device

struct Staging {
  PinBuf host[kRing];
  hipEvent_t done[kRing] = {};
  bool armed[kRing] = {};
  int next = 0;
};

// Per-call tables of a paired set: a ring of device slots. On a large-BAR device the host writes a slot directly
// (fine-grained device memory behind the PCIe BAR); otherwise through the pinned twin + a copy (paired_launch.hip.h).
struct Arena {
  void* dev[kRing] = {};
  size_t cap[kRing] = {};
  bool direct[kRing] = {};
  PinBuf host[kRing];
  hipEvent_t done[kRing] = {};
  bool armed[kRing] = {};
  int next = 0;
  void release() {
    for (int k = 0; k < kRing; k++) {
      if (dev[k]) (void)hipFree(dev[k]);
      dev[k] = nullptr; cap[k] = 0;
      host[k].release();
      if (done[k]) (void)hipEventDestroy(done[k]);
      done[k] = nullptr; armed[k] = false;
    }
  }
};

struct Reducer {  // per read set: partials + ticket + 2-double result
  DevBuf part_sum, part_zero, ticket, out;
  hipError_t init() {
    hipError_t e;
    if ((e = part_sum.reserve((4 * kMaxBlocks + kOvfMaxBlocks) * sizeof(double))) != hipSuccess) return e;
    if ((e = part_zero.reserve((4 * kMaxBlocks + kOvfMaxBlocks) * sizeof(int))) != hipSuccess) return e;
    if ((e = ticket.reserve(kTicketWords * sizeof(unsigned))) != hipSuccess) return e;
    if ((e = out.reserve(4 * sizeof(double))) != hipSuccess) return e;
    if ((e = hipMemset(ticket.p, 0, kTicketWords * sizeof(unsigned))) != hipSuccess) return e;
    if ((e = hipMemset(out.p, 0, 4 * sizeof(double))) != hipSuccess) return e;
    return hipDeviceSynchronize();  // the scoring stream is non-blocking: make the zeroes land first
  }
  void release() { part_sum.release(); part_zero.release(); ticket.release(); out.release(); }
};

struct AlignDev {  // device copies the GPU aligner needs: reads (1 byte per base) + the max-hash index
  DevBuf reads, read_off, bucket_hash, bucket_top, bucket_off, bucket_reads;  // bucket_top: first bucket per upper half of the key (aln_find_bucket)
  DevBuf htab;     // key -> bucket, open addressing (AlnHashSlot, 2^hbits slots; hbits = 0: none)
  int hbits = 0;
  bool uploaded = false;
  void release() { reads.release(); read_off.release(); bucket_hash.release(); bucket_top.release(); bucket_off.release(); bucket_reads.release(); htab.release(); }
};
struct AlignScratch {  // per context, grown on demand
  DevBuf wstr, wins, blk, spans, cands, hits, counters;
  DevBuf sort_keys, sort_idx, sort_tmp, hits_sorted, file_tmp;  // large batches: hits ordered (and filed) on the device
  // What a large batch sends and receives travels through pinned memory of the library's own. (Copies straight from / to the
  // job's vectors made the runtime register those pages with the device -- and FREEING such a buffer afterwards makes the
  // driver evict the process's queues: the first dispatch after the cold batch started 12-25 ms late, the device idle.)
  PinBuf stage;
  void release() { wstr.release(); wins.release(); blk.release(); spans.release(); cands.release(); hits.release(); counters.release();
                   sort_keys.release(); sort_idx.release(); sort_tmp.release(); hits_sorted.release(); file_tmp.release(); stage.release(); }
};

// small batches (aln_small_*): one set of buffers per mate, so that the two mates' pipelines run side by side. Input
// block written by the host (device memory behind the BAR, or its pinned twin), counters + hits published in mapped
// pinned memory behind a sequence word.
struct AlignSmall {
  DevBuf spans, cands, hits, counters, wcopy;
  void* in_dev = nullptr; size_t in_cap = 0; bool in_direct = false;
  PinBuf in_host, out_host;
  AlnStrArgs str_args;  // a small batch's window strings as they travel in the launches' argument segments
  unsigned long long out_seq = 0;
  double seen_us = 0;   // host clock when the last batch's sequence word was seen (timing builds)
  hipEvent_t probe_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // timing probes of the development build (GAML_ALN_WAIT)
  void release() {
    spans.release(); cands.release(); hits.release(); counters.release(); wcopy.release();
    for (int k = 0; k < 4; k++) if (probe_ev[k]) { (void)hipEventDestroy(probe_ev[k]); probe_ev[k] = nullptr; }
    if (in_dev) (void)hipFree(in_dev);
    in_dev = nullptr; in_cap = 0; in_host.release(); out_host.release();
  }
};
// one batch of pending windows of one mate: strings built once, possibly already in flight on the device
struct AlnJob {
  bool prepared = false, enqueued = false;
  std::string wstr;
  std::vector<AlnWindow> wins;
  std::vector<int32_t> blk;  // blk[w] = first block of window w in span_maxima_kernel's grid; blk[n] = the grid size
  unsigned long long seq = 0;
  void* stream = nullptr;  // where the small-batch pipeline was enqueued
};

struct MateDev {
  DevBuf first, extra, pows;  // pows = mismatch_pow | match_pow (first / extra: the read-major tables of a single-end set)
  AlignDev aln;
  uint64_t uploaded_generation = ~0ull;
  size_t pow_n = 0;
  // paired sets: the alignment-window cache's records ON THE DEVICE, window-major (a window's records contiguous, ordered by
  // (position, read)): int4 {window id, position, edit | orient << 8, read}. Window::dfirst points into it. Filled by the
  // aligner's filing kernels, or mirrored from the host pool for windows the host filed (host aligner, caller-supplied records)
  DevBuf pool;
  int64_t pool_n = 0;         // records in use
  size_t filed_done = 0;      // ShortMate::filed[0 .. filed_done) are on the device
  DevBuf lens;                // read lengths
};

// the device side of one build of the record tables: everything a rebuild replaces as a whole (table_build.hip.h)
struct TableDev {
  DevBuf rec8[2], first[2], extra[2], inl[2], len_code, len12, static_idx, static_val;
  DevBuf slot_of_read, read_of_slot, dirty_of_slot;  // read -> slot, slot -> read, slot -> index on the delta lists (-1: none)
  DevBuf cnt;              // the build's device counters (kTb*)
  bool built = false;
  // what the host knows of it once the build is through
  int64_t class_count[4] = {0, 0, 0, 0}, n0a = 0, extras[2] = {0, 0}, dropped[2] = {0, 0};
  bool keep_dominated = false;   // knob 16 when it was built
  std::vector<int32_t> read_of_slot_host;  // fetched on demand (gaml_hip_read_probs)
  bool ros_valid = false;
  void release() {
    for (int m = 0; m < 2; m++) { rec8[m].release(); first[m].release(); extra[m].release(); inl[m].release(); }
    len_code.release(); len12.release(); static_idx.release(); static_val.release(); slot_of_read.release(); read_of_slot.release(); dirty_of_slot.release(); cnt.release();
  }
};

// scratch of a table build (one build at a time per set)
struct BuildScratch {
  DevBuf k_in, k_out, k_tmp, v_in, v_tmp, hist, v_sorted[2], rstart[2], rend[2], one[2], cl, sidx, more[2], start[2], tiles, wins[2], peer0;
  PinBuf h_wins[2], h_peer, h_cnt;
  void release() {
    k_in.release(); k_out.release(); k_tmp.release(); v_in.release(); v_tmp.release(); hist.release(); cl.release(); sidx.release(); tiles.release(); peer0.release();
    for (int m = 0; m < 2; m++) { v_sorted[m].release(); rstart[m].release(); rend[m].release(); one[m].release(); more[m].release(); start[m].release(); wins[m].release(); h_wins[m].release(); }
    h_peer.release(); h_cnt.release();
  }
};

// A rebuild of the record tables beside the evaluations: the build is a chain of kernels on a stream of its own over the
// device pool (a few hundred microseconds at 833 k pairs); evaluations go on over the old tables + delta lists, and the new
// tables take over a FIXED number of evaluations later (equal inputs must give equal outputs run to run, and a rebuild
// changes the order of the final sum). What was activated in between is applied to the new tables' (empty) delta lists by
// the same kernel that maintains the live ones.
// what the slices of one build share (paired_build_enqueue)
struct BuildPlan { int64_t A[2] = {0, 0}; int n_act[2] = {0, 0}; unsigned tiles = 0, none1 = 0, none2 = 0; int ins_n = 0, units = 1; bool empty = true; };

struct TableRebuild {
  bool active = false;
  bool retired = false;        // a rebuild beside the evaluations was decided and the unused windows have left: the next evaluation starts the build
  int next_slice = 0;          // slices of the build's launch chain enqueued so far
  int64_t start_eval = 0;      // the set's evaluation count when the rebuild was decided
  TableDev tab;                // the spare set of buffers: being built, or idle
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr, mark = nullptr;
  std::vector<std::pair<int32_t, int32_t>> after;  // (mate, window) activated since the build started: not in the new tables
};

// what the host keeps of the live tables (the tables themselves exist on the device only)
struct PairInfo {
  int64_t class_count[4] = {0, 0, 0, 0};  // 0: compact; 1: <= 2 records; 2: <= 4; 3: more
  int64_t n0a = 0;                        // static part of the compact class
  std::vector<uint32_t> len_combo;        // distinct L1 | L2 << 16 values (<= 256), in order of first appearance: fixed per read set
  int64_t dropped_records[2] = {0, 0};
};

struct PairedSet {
  gaml_paired_cfg cfg;
  ShortMate mate[2];
  PairInfo pt;
  MateDev dev[2];                     // (pows, aligner index, record pool, generation)
  TableDev tab;                       // the live record tables
  TableRebuild rebuild;
  BuildScratch scratch;
  BuildPlan build_plan;
  int64_t async_rebuilds = 0, retired_windows = 0, eval_count = 0;
  // per set, fixed: length-combination code per pair, the per-combination tables, the memo of pair terms
  DevBuf lcode, len_combo_dev, combo_tabs, memo;
  int memo_codes = 0;
  bool statics_uploaded = false, kernels_warm = false, prereserved = false;
  // Delta store (device): pairs whose record lists gained records of windows activated after the tables were built. Their
  // complete lists sit at a fixed stride (4 records per mate; longer ones in the spill area); maintained by
  // delta_apply_kernel, read by the scoring launch. The host knows upper bounds (and the exact counts once a blocking call
  // has returned: h_dstate, written by the kernels).
  DevBuf dl_slot, dl_spill, dl_rec[2], sp_rng[2], sp_rec[2], sp_slot, dstate;
  DevBuf dl_bins, dl_bin_count, dl_blk_tot, dl_wlist, dl_stamps;   // multi-block maintenance launches (delta_dev.hip.h)
  PinBuf h_dstate;
  size_t delta_cap = 0, cap_spill = 0, cap_sprec = 0;
  int64_t nd_est = 0, ns_est = 0;     // delta pairs (upper bound) / long lists (last exact count) as the host knows them
  bool spill_may_grow = false;        // activations since the counts were last read back
  int dl_seq = 0;                     // sequence number of the last maintenance launch (h_dstate[kDsSeq] == dl_seq: the counts are current)
  int64_t full_rebuilds = 0, delta_updates = 0;
  int quiet_calls = 0;       // evaluations since the last window activation
  int64_t delta_left_out = 0, delta_left_out_base = 0;  // records of later windows that never reached the delta lists (always overwritten); the device counts them per list generation
  bool compact_requested = false;  // gaml_hip_compact_tables: fold the delta lists into the tables at the next evaluation
  PinBuf h_timeline; int timeline_waves = 0;  // ablation 8 (tools/kernel_timeline.py)
  PinBuf h_part_sum, h_part_zero;     // per-block partials written straight to pinned host memory (blocking calls): [set][block]
  size_t host_part_stride = 0;        // entries per set
  int last_total_blocks = 0, last_sets = 1;
  int last_blocks[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // partials written per path set of the last launch(es)
  bool last_host_partials = false;
  DevBuf probs, tabs, cov_bits, bad;
  Arena arena;                        // per-call tables (occurrence images, thresholds, coverage layout): ring, whole tables per call
  // Blocking calls on a large-BAR device keep ONE resident copy of the tables and patch it in place (the kernel
  // of the previous call is done when the call returns): a call that shares most paths with the previous one writes
  // a few dozen 12-byte entries through the BAR instead of the whole image. Capacities leave room to grow.
  struct Persist {
    void* dev = nullptr; size_t bytes = 0;
    size_t cap_w[2] = {0, 0}, cap_lo[2] = {0, 0}, cap_m[2] = {0, 0};  // table entries / list bounds / list entries per mate
    size_t off_tfloor = 0, off_occ[2] = {0, 0}, off_lo[2] = {0, 0}, off_m[2] = {0, 0};
    bool valid = false;                // the copy mirrors the host images as of their last take_changed()
    void release() { if (dev) (void)hipFree(dev); dev = nullptr; bytes = 0; valid = false; }
  } persist;
  int64_t batches_patched = 0, batches_full = 0;  // gaml_hip_calc_prob_batch chunks whose per-set tables were built on the device from patches / written whole
  size_t batch_slack = 0;             // extra bytes per path set region of a batch (grows when a set's tables did not fit)
  hipEvent_t ev_tables = nullptr, ev_ovf = nullptr;
  PairedPlanner planner;
  OccImage image[2];                 // persistent host images of the occurrence tables, patched per call
  std::vector<Occ> scratch_occ[2];   // debug dumps only
  Reducer red;
  Staging stage_pool;                // host-filed windows on their way into the device pool
  std::vector<double> ins_tab, floor_tab, logfloor_tab, covthr_tab;
  bool floor_positive = true;  // every floor exp(c + k s) > 0 (else "probability 0 is floored" does not hold: no memo / shortcut paths)
  bool tabs_uploaded = false;
  int64_t last_bad_bases = 0;
};

struct SingleSet {
  gaml_single_cfg cfg;
  ShortMate mate;
  ReadMajor rm;
  MateDev dev;
  DevBuf lens, probs, tabs, occ_arena;
  std::vector<Occ> last_occ;
  Reducer red;
  std::vector<double> floor_tab, logfloor_tab;
  bool tabs_uploaded = false;
  Staging stage;
};

struct DpDev {  // device buffers of the PacBio banded DP
  DevBuf path, jobs, ops, scratch, out, dbg;
  void release() { path.release(); jobs.release(); ops.release(); scratch.release(); out.release(); dbg.release(); }
};

// coverage sweep of a PacBio set with a penalty (pacbio_sweep.hip.h)
struct PbSweepDev {
  // follows the record cache: per sub-walk the intervals {position, position_end} of its records that clear GetMinReadProb
  DevBuf iv_off, iv;
  std::vector<int32_t> iv_off_host;
  uint64_t generation = ~0ull;
  // per evaluation: contig lengths | node intervals | occurrences, staged as one block; then the evaluation's interval
  // list: node intervals first, the records' after them (this rank's; all ranks' after the exchange)
  DevBuf in, all;
  DevBuf key_begin, key_end, pos, key_begin_s, key_end_s, pos_s, end_max, tmp, bad;
  Staging stage;
  void release() {
    iv_off.release(); iv.release(); in.release(); all.release(); key_begin.release(); key_end.release(); pos.release();
    key_begin_s.release(); key_end_s.release(); pos_s.release(); end_max.release(); tmp.release(); bad.release();
    for (int k = 0; k < kRing; k++) { stage.host[k].release(); if (stage.done[k]) (void)hipEventDestroy(stage.done[k]); stage.done[k] = nullptr; stage.armed[k] = false; }
  }
};

struct PacbioSet {
  gaml_single_cfg cfg;
  int64_t n_global = 0, lo = 0, hi = 0;
  std::vector<int32_t> lens;  // shard
  double log_match = 0, log_mismatch = 0;
  std::unordered_map<Walk, int32_t, WalkHasher> walk_id;
  std::vector<std::vector<gaml_pacbio_aligment>> recs;  // per sub-walk, local read ids
  uint64_t generation = 0, uploaded_generation = ~0ull;
  int32_t max_len = 0;
  int64_t misses = 0;
  DevBuf d_lens, rec_off, rec_walk, rec_logp, walk_count, logprobs;
  Reducer red;
  int64_t last_bad_bases = 0;
  Staging stage;
  PbSweepDev sweep;
  // cache-miss side (SAM ingestion): bases of this shard's reads and the name -> global id map
  bool have_reads = false;
  std::string bases;
  std::vector<int64_t> base_off;  // local read i = bases[base_off[i], base_off[i+1])
  std::unordered_map<std::string, int32_t> name_id;
  DevBuf d_bases;
  DpDev dp;
  bool bases_uploaded = false;
  double dp_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

struct PairedPrep {
  int64_t assembled_records = 0;  // records the reference would touch in GetPositionsOnlyPath
  std::vector<int32_t> path_base, start_off, starts;
  int32_t total_bits = 0, n_paths = 0;
  bool general = false;           // some window occurs several times in this path set: the GEN instantiation of the scoring kernels
};

struct SetRef { int kind, idx; };

}  // namespace detail
}  // namespace gaml

using namespace gaml;          // (this header is private to the library's two HIP translation units)
using namespace gaml::detail;

struct gaml_hip_ctx {
  gaml::MultiState* multi = nullptr;  // several device shards behind this context (gaml_hip_create_multi): every call is forwarded
  gaml::CommState* comm = nullptr;    // RCCL communicator of a sharded context (gaml_hip_comm_init_rank / the shards of a multi context)
  int device = -1;
  DevBuf warm_buf; bool general_warm = false;  // warm_general_kernels (gaml_hip.hip)
  AlnJob aln_job[2];  // the two mates' pending alignment batches (align_pending_pair): buffers kept from call to call
  hipStream_t stream = nullptr;
  hipStream_t aux_stream = nullptr;  // the overflow kernel runs beside the main kernel
  GraphStore g;
  bool have_graph = false;
  std::vector<std::unique_ptr<SingleSet>> singles;
  std::vector<std::unique_ptr<PairedSet>> paireds;
  std::vector<std::unique_ptr<PacbioSet>> pacbios;
  std::vector<SetRef> handles;  // creation order -> (kind, index)
  int32_t rank = 0, world = 1;
  AlignScratch aln_scratch;
  AlignSmall aln_small[2];
  int64_t aln_windows = 0, aln_candidates = 0;  // GPU aligner statistics
  double aln_us = 0;
  double aln_stage_us[5] = {0, 0, 0, 0, 0};  // window strings + upload, spans + candidates, extension, D2H of hits, sort + finalize
  int64_t aln_batches = 0;
  int knobs[24] = {0};  // tuning experiments and A/B switches (gaml_hip_debug.h), development builds only: read through KNOB()
  bool direct_write = false;  // large-BAR device: the host writes per-call tables straight into device memory (Arena)
  int32_t peers = 1;  // contexts (incl. this one) that hold reads of the same read sets: >1 => window maxima must be exchanged
  std::string err;
  // timing
  bool event_timing = false;
  int event_every = 1;    // time every k-th scoring launch (attached events cost ~4 us of host time per launch)
  int64_t event_tick = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;  // one pair per scoring launch of a call
  std::vector<uint64_t> ev_call;  // ... and the evaluation (eval_serial) it belongs to
  uint64_t eval_serial = 0;
  size_t ev_used = 0;
  double t_host_us = 0, t_dev_wall_us = 0, t_kernel_us = 0;
  int64_t stat_launches = 0;
  double stat_device_us = 0, stat_algo_bytes = 0;
  DevBuf packed;  // 4 doubles per read set
  PinBuf packed_host;
  // gaml_hip_shm_exchange_*: the ranks of one node add up their (host-resident) partials through a POSIX shared-memory block
  char* shm_base = nullptr; size_t shm_bytes = 0; int shm_rank = 0, shm_world = 0, shm_cap = 0; unsigned long long shm_step = 0; std::string shm_name;
  PinBuf fetch_host;            // gaml_hip_fetch_async / _wait: [sequence word | 63 x pad | doubles]
  unsigned long long fetch_seq = 0;
  hipStream_t fetch_stream = nullptr;
  DevBuf batch_dev;  // gaml_hip_calc_prob_batch: 4 doubles per read set and path set
  PinBuf batch_host;
  // evaluation in progress (between eval_begin and eval_finish)
  bool pending_open = false;
  std::vector<Walk> pending_paths;
  bool pending_paths_valid = false;  // contexts of paired sets only never materialise the vectors
  int32_t pending_total_len = 0;
  std::vector<std::unique_ptr<PairedPrep>> pending_prep;  // per paired set
  double pending_host_us = 0;
  double prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // phase stamps of the last blocking call (us): see gaml_hip_debug_profile
  bool used_default_stream = false;  // an *_async entry point was handed NULL (the legacy default stream) since the last gaml_hip_sync
  // sharded evaluations: two status words a scoring launch writes with its partials (ticket finish), so that the exchange needs no
  // dispatch of its own for them (multi.hip: ctx_set_status / ctx_status_done)
  double* status_dst = nullptr; double status_a = 0, status_b = 0; bool status_done = false;
  bool host_results = false;  // blocking call: kernels write their results into pinned host memory, no D2H copy
  // sharded evaluation with a coverage penalty: sweeps wait for the other ranks' coverage maps
  bool defer_cov = false;
  struct PendingCov { int paired_idx; CovArgs args; double* out4; };
  std::vector<PendingCov> pending_cov;
  // same for a PacBio set: the alignment intervals of the other ranks' reads are missing (device lists: PbSweepDev::all)
  struct PendingPacbio {
    int pacbio_idx;
    double* out4;
    int32_t n_paths;
    int64_t n_node, n_own;  // node intervals (every rank has them); this rank's record intervals behind them
  };
  std::vector<PendingPacbio> pending_pb;
};
