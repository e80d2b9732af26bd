// internal.h -- what gaml_hip.hip (single-device context, C ABI) and multi.hip (several device shards in one
// process; the RCCL communicator of a sharded context) know about each other. Not installed, not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/gaml_hip.h"
#include "../../include/gaml_hip_debug.h"

namespace gaml {

struct CommState;   // RCCL communicator + device buffers of one sharded context (multi.hip)
struct MultiState;  // N device shards behind one context (multi.hip)

// ---- implemented in gaml_hip.hip ---------------------------------------------------------------------------
int ctx_fail(gaml_hip_ctx* c, int code, const std::string& msg);  // stores the text, returns the code
hipStream_t ctx_stream(const gaml_hip_ctx* c);
int ctx_device(const gaml_hip_ctx* c);
int ctx_rank(const gaml_hip_ctx* c);
int ctx_world(const gaml_hip_ctx* c);
int ctx_peers(const gaml_hip_ctx* c);
MultiState* ctx_multi(const gaml_hip_ctx* c);
void ctx_set_multi(gaml_hip_ctx* c, MultiState* m);
CommState* ctx_comm(const gaml_hip_ctx* c);
void ctx_set_comm(gaml_hip_ctx* c, CommState* s);
// a context that holds no read set and no device: the parent of a multi-device context
gaml_hip_ctx* ctx_new_parent();
// bad_bases bookkeeping after a reduced (sharded) evaluation: 4 doubles per read set, reference order
void ctx_note_reduced(gaml_hip_ctx* c, const double* partials);
// does any paired / PacBio read set carry a coverage penalty (the non-separable piece)?
bool ctx_has_penalty(const gaml_hip_ctx* c);
// gaml_hip_fetch_wait that gives up: polls the fetch's sequence word for at most timeout_s seconds (never an unbounded
// stream synchronise -- a collective whose peer died never completes). Returns 1 on time-out, < 0 on error, 0 when done.
int ctx_fetch_wait_bounded(gaml_hip_ctx* c, double* out, int32_t n_doubles, double timeout_s);
// a sharded evaluation's status words {a, b} at `dst` (device): the next stream-ordered scoring launch that finishes its own
// partials writes them too; ctx_status_done tells whether one did (else the caller dispatches them itself)
void ctx_set_status(gaml_hip_ctx* c, double* dst, double a, double b);
bool ctx_status_done(const gaml_hip_ctx* c);
// close an evaluation that gaml_hip_eval_begin opened and that will not be finished (another shard failed)
void ctx_eval_abandon(gaml_hip_ctx* c);


// ---- implemented in multi.hip ------------------------------------------------------------------------------
void multi_destroy(MultiState* m);
void comm_destroy(CommState* s);
// sharded context with a communicator: the whole evaluation incl. the exchanges, reduced partials on the host
int comm_eval_reduced(gaml_hip_ctx* c, const int32_t* paths, const int64_t* offs, int32_t n_paths, double* partials_out,
                      int32_t* total_len_out);
// the same for a batch of path sets: ONE all-reduce(sum) over all sets' partials (set_offs as gaml_hip_calc_prob_batch)
int comm_eval_reduced_batch(gaml_hip_ctx* c, int32_t n_sets, const int32_t* paths, const int64_t* offs, const int32_t* set_offs,
                            double* partials_out /* n_sets * 4 * n_readsets */, int32_t* total_lens_out);

// forwarded entry points of a multi-device context (same meaning as the C ABI function of the same name)
int multi_set_graph(MultiState* m, int32_t n_nodes, const char* bases, const int64_t* offs);
int multi_load_graph(MultiState* m, const char* file);
int multi_add_single(MultiState* m, const gaml_single_cfg* cfg, int32_t n, const char* bases, const int64_t* offs);
int multi_add_paired(MultiState* m, const gaml_paired_cfg* cfg, int32_t n, const char* b1, const int64_t* o1, const char* b2,
                     const int64_t* o2);
int multi_add_pacbio(MultiState* m, const gaml_single_cfg* cfg, int32_t n, const int32_t* lens);
int multi_add_pacbio_reads(MultiState* m, const gaml_single_cfg* cfg, int32_t n, const char* bases, const int64_t* offs,
                           const char* names);
int multi_put_window_records(MultiState* m, int rs, int mate, const int32_t* sub, int32_t len, const gaml_aligment* recs, int64_t n);
int multi_put_pacbio_records(MultiState* m, int rs, const int32_t* sub, int32_t len, const gaml_pacbio_aligment* recs, int64_t n);
int32_t multi_pacbio_missing(MultiState* m, int rs, const int32_t* path, int32_t n, int32_t* ranges, int32_t cap);
int multi_pacbio_ingest_sam(MultiState* m, int rs, const int32_t* path, int32_t n, const char* sam, int64_t sam_len, int64_t* filed);
int64_t multi_pacbio_records(MultiState* m, int rs, const int32_t* sub, int32_t len, gaml_pacbio_aligment* out, int64_t cap);
int multi_calc_partials(MultiState* m, const int32_t* paths, const int64_t* offs, int32_t n_paths, double* partials_out,
                        int32_t* total_len_out);
int multi_calc_prob_batch(MultiState* m, int32_t n_sets, const int32_t* paths, const int64_t* offs, const int32_t* set_offs,
                          double* probs_out, int32_t* zeros_out, int32_t* total_lens_out);
int multi_combine(MultiState* m, const double* partials, int32_t total_len, double* prob_out, int32_t* zeros_out);
int multi_num_readsets(const MultiState* m);
int multi_readset_kind(const MultiState* m, int rs);
int64_t multi_readset_reads(const MultiState* m, int rs);
int32_t multi_num_nodes(const MultiState* m);
int32_t multi_node_len(const MultiState* m, int32_t node);
int multi_read_probs(MultiState* m, int rs, double* out, int64_t n);
int multi_bad_bases(MultiState* m, int rs, int64_t* out);
int64_t multi_window_count(const MultiState* m, int rs, int mate);
int64_t multi_window_records(MultiState* m, int rs, int mate, const int32_t* sub, int32_t len, gaml_aligment* out, int64_t cap);
int64_t multi_align_window(MultiState* m, int rs, int mate, const int32_t* sub, int32_t len);
int multi_compact_tables(MultiState* m);
int multi_sync(MultiState* m);
int multi_set_event_timing(MultiState* m, int on);
int multi_kernel_stats(MultiState* m, int reset, int64_t* launches, double* device_us, double* algo_bytes);
int multi_last_timing(const MultiState* m, double* out3);
gaml_hip_ctx* multi_shard(const MultiState* m, int i);
int multi_num_shards(const MultiState* m);
const char* multi_last_error(const MultiState* m);

}  // namespace gaml
