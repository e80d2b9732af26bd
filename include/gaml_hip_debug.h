/* gaml_hip_debug.h -- test, tuning and tracing entry points of the DEVELOPMENT build, libgaml_hip_dev.so
 * (gaml_amd/csrc/Makefile: the same sources compiled with -DGAML_HIP_DEV).
 *
 * NOT part of the drop-in boundary (include/gaml_hip.h), and not in the product: libgaml_hip.so exports none of these
 * symbols, carries none of the A/B switches and tuning knobs they set (every knob is compiled in at its default), no
 * kernel instantiation with in-kernel time stamps and no crash hook. These entry points expose intermediate state to
 * the parity tests (tests/) and the tuning tools (tools/); a caller that only wants ProbCalculator::CalcProb never
 * includes this file.
 */
#ifndef GAML_HIP_DEBUG_H_
#define GAML_HIP_DEBUG_H_

#include "gaml_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- PacBio SAM ingestion, piece by piece ------------------------------------------------------ */
/* host-only introspection (no device needed): how one SAM line is parsed
 * ({flags,len,posstart,posend,sstart,send,slen,tstart,tend,edit_dist}) and which DP cells it gets
 * (rows row0.., one column interval per row). Returns the number of rows, <0 on a malformed line. */
/* the banded DP of one SAM line against an explicit target string ("path + '\n' + reverse
 * complement") and read, on the GPU: what AligmentProbability (graph.cc:2175-2297) returns, as a
 * log.  Returns the number of DP rows (<0: error); when band_lo/band_hi hold at least that many
 * entries they receive the column interval per row that the kernel derived from the CIGAR. */
int gaml_hip_debug_sam_logprob(gaml_hip_ctx* ctx, const char* target, int32_t target_len, const char* read, int32_t read_len,
                               const char* sam_line, int64_t sam_len, double mismatch_prob, double* logprob_out,
                               int32_t* band_lo, int32_t* band_hi, int32_t band_cap);
/* host-only: what the DP kernel is given for one SAM line: {n_ops, row_f, col_f, bl, el, max_width}
 * and the run-length CIGAR ((length << 2) | code, 0 = M, 1 = I, 2 = D). Returns n_ops, <0 on a malformed line. */
int gaml_hip_debug_sam_shape(const char* sam_line, int64_t len, int32_t total_len, int32_t* out6, uint32_t* ops, int32_t cap);
int32_t gaml_hip_debug_sam_band(const char* sam_line, int64_t len, int32_t total_len, int32_t* fields10, int32_t* row0,
                                int32_t* lo, int32_t* hi, int32_t cap);

/* ---- host-side half of an evaluation ----------------------------------------------------------- */
/* Host-side half of an evaluation WITHOUT touching the device (works on a host-only context):
 * window registration + alignment of missing windows + window-occurrence lists for `paths`,
 * exactly what gaml_hip_calc_* does before it launches. For tests of the host logic. */
int gaml_hip_debug_prepare(gaml_hip_ctx* ctx, const int32_t* paths, const int64_t* path_offs, int32_t n_paths);
/* occurrence list of the last prepare/evaluation: 5 ints per entry {window id, shift, min_pos,
 * path, rank}; returns the number of entries. */
int64_t gaml_hip_debug_occurrences(gaml_hip_ctx* ctx, int readset, int mate, int32_t* out5, int64_t cap);
/* what the occurrence TABLES of the last prepare/evaluation hold (the host image that the device copy mirrors), as the
 * same 5 ints per entry, path = position of the path in the set, rank = path-local; min_pos as stored (clamped at
 * -32768 in the 8-byte entries). Sorted by (path, rank). For tests of the incremental table maintenance: must describe
 * the same occurrences as gaml_hip_debug_occurrences. info3 (may be NULL): {1 if the last planning was incremental,
 * incremental calls so far, whole-set calls so far}. */
int64_t gaml_hip_debug_table_occurrences(gaml_hip_ctx* ctx, int readset, int mate, int32_t* out5, int64_t cap, int64_t* info3);
/* node ids of a cached window (by id); returns its length, -1 if the id is unknown */
int32_t gaml_hip_debug_window_walk(gaml_hip_ctx* ctx, int readset, int mate, int32_t window_id, int32_t* out, int32_t cap);
/* ---- tuning ------------------------------------------------------------------------------------- */
/* tuning experiments and A/B switches (tools/, tests): 0 = compact-class blocks, 1 = dynamic LDS bytes, 2 = finish mode of
 * stream-ordered calls (1 two-level tickets in the kernel, 2 finisher kernel = the default), 3 = 8: in-kernel timeline,
 * 4 = 1: no floor/log memo, 5 = window aligner: 1 host, 2 hits sorted on the host, 3 always the general route, 4 one
 * small-batch pipeline per mate, 5 window strings through the input block, 6 hits filed on the host (default: on the
 * device), 6 = 1: no delta lists (every newly activated window rebuilds the tables), 2: no quiet-spell rebuild, 7 = 1:
 * always wait with hipStreamSynchronize (no spinning on the pinned partials), 8 = 1: no direct writes through the BAR
 * (staging slots + copies), 9 = 1: aligner stage times with device syncs, 10 = blocks of the <=2-record class, 11 =
 * batches: 1 one launch per path set, 2 whole tables per set, 3 no capture of unchanged pairs, 32 + mask: classes of
 * blocks left out (TIMING ONLY, results wrong), 12 = 1: every path set planned from scratch, 13 = 1: whole per-call tables
 * through the ring (no resident copy), 14 = 1: every table build on the calling stream, k > 1: a build beside the
 * evaluations takes over k evaluations after its start (default 96), 15 = 1: rebuilds never retire unused windows,
 * 16 = 1: record tables keep the records that can never survive the overwrite rule (takes effect at the next table build;
 * same values either way), 18 = d: tables are rebuilt when the delta lists pass pairs / d (default 8), 19 = 1: no static
 * memo indices (takes effect at the next table build; same values either way), 20 = blocks of the compact class's second
 * part, 22 = 1: delta maintenance by one-block launches only (default: multi-block
 * above 3,000 records). Environment (development build): GAML_DL_STAMPS=1 prints the delta kernel's stage times. */
/* Ablation 8 (knob 3 = 8) of the last evaluation of paired read set rs: 8 wall-clock stamps (10 ns units) per wave,
 * [kernel entry, tables in LDS, records in, occurrences in, memo in, stores issued, block reduced, class]. Returns the
 * number of waves copied. Tuning aid (tools/kernel_timeline.py). */
int gaml_hip_debug_timeline(gaml_hip_ctx* ctx, int rs, unsigned long long* out, int64_t cap_waves);

int gaml_hip_debug_set_knob(gaml_hip_ctx* ctx, int knob, int value);
/* Environment (read once): GAML_HIP_TRACE_HOST=1 -- host-side phase times of slow calls, table builds and rebuilds on
 * stderr; GAML_HIP_TRACE_ALIGNER=1 -- aligner stage times with gaml_hip_aligner_stats; GAML_HIP_BACKTRACE=1 -- a
 * backtrace on stderr when the process aborts or faults (also after the HIP runtime reports a GPU memory fault). */
/* host-only (works without a device): the record tables of the windows that are active now, built with and without the
 * rule "a junction record that the first node's own record always overwrites stays out" (knob 16), compared pair by
 * pair. out6 = {records left out mate 1, mate 2, compact-class pairs with / without the rule, records checked,
 * violations}; GAML_HIP_ESTATE if a record was left out that the rule does not cover. */
int gaml_hip_debug_fold_check(gaml_hip_ctx* ctx, int readset, int64_t* out6);
/* The library's own stable radix sort (gaml_amd/csrc/radix_sort.hip.h: the aligner's hit ordering, graph.cc:841, 895-897, and
   the PacBio coverage sweep, graph.cc:3198-3250) on caller data: keys[n] and, unless null, vals[n] are sorted in place on
   bits [begin_bit, end_bit); run_max (or null) receives the inclusive running maximum of the sorted payload (of the sorted
   keys without one). */
int gaml_hip_debug_radix_sort(gaml_hip_ctx* ctx, uint64_t* keys, uint64_t* vals, int64_t n, int begin_bit, int end_bit, uint64_t* run_max);
/* host-only: the static memo indices of the compact class (both records of a pair in windows with the same node walk:
 * orientation rule, insert distance and memo index do not depend on the path set) recomputed from the window cache.
 * out8 = {pairs with an index, other compact-class pairs, violations, then why those others have none: a mate without
 * record, records in different windows, orientation rule, distance outside the insert-size table, edit count or length
 * code outside the memo}; GAML_HIP_ESTATE on a violation. */
int gaml_hip_debug_static_check(gaml_hip_ctx* ctx, int readset, int64_t* out8);
/* The device table build (gaml_amd/csrc/table_build.hip.h: the read-major join the reference does per call through hash maps,
 * graph.cc:535-598) against the host restatement build_pair_tables on the windows that are active now: a fresh build into
 * scratch buffers, every array compared entry by entry. out8 = {pairs, compact class, its static part, <= 2 records, <= 4,
 * more, entries compared, mismatches}; GAML_HIP_ESTATE when they differ. */
int gaml_hip_debug_tables_check(gaml_hip_ctx* ctx, int readset, int64_t* out8);
/* per-block partial sums / floored counts of the last blocking evaluation of paired set `readset` (path set `set` of a
 * batch launch, 0 for a single call), in block order; layout8 = {blocks of the compact class's static part, of the
 * compact class, up to the <= 2-record class, up to the <= 4-record class, lane-per-pair blocks, all scoring blocks,
 * 0, partials}. Returns the number of partials. For bit-equality hunts between routes. */
int32_t gaml_hip_debug_block_partials(gaml_hip_ctx* ctx, int readset, int32_t set, double* sums, int32_t* zeros, int32_t cap, int32_t* layout8);

#ifdef __cplusplus
}
#endif
#endif /* GAML_HIP_DEBUG_H_ */
