/* gaml_hip.h -- C ABI of libgaml_hip.so: the MI355X (gfx950) implementation of GAML's
 * assembly-likelihood hot path.
 *
 * This is the drop-in boundary.  GAML has no plugin/FFI layer; the path sits behind the
 * header-only class ProbCalculator (reference prob_calculator.h:37-124) whose CalcProb()
 * calls the three per-read-set scorers of graph.cc.  A maintainer replaces
 * prob_calculator.h by include/gaml_hip_prob_calculator.h; that
 * header forwards to the functions below.  Plain C types only, caller-owned input
 * buffers (copied during the call), library-owned host and device state, integer status
 * returns (0 = ok, <0 = error, text via gaml_hip_last_error).  No exceptions cross.
 *
 * Every entry point cites the reference interface it replaces (file:line under the
 * reference tree).  All citations: graph.cc / graph.h / gaml.cc / prob_calculator.h.
 */
#ifndef GAML_HIP_H_
#define GAML_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gaml_hip_ctx gaml_hip_ctx; /* opaque: graph + read sets + device state */

/* Aligment record, 4 x int32 = 16 B (reference graph.h:211-231). */
typedef struct gaml_aligment {
  int32_t position;    /* 1-based start inside the window's node string (graph.cc:890) */
  int32_t edit_dist;
  int32_t read_id;
  int32_t orientation; /* 0 forward, 1 reverse complement */
} gaml_aligment;

/* PacbioAligment record (reference graph.h:516-535): 3 x int32 + logdouble (f64 log). */
typedef struct gaml_pacbio_aligment {
  int32_t position;
  int32_t position_end;
  int32_t read_id;
  int32_t pad_;
  double logprob;
} gaml_pacbio_aligment;

/* SingleReadConfig (prob_calculator.h:7-18) + the read set's error model (gaml.cc:812-813). */
typedef struct gaml_single_cfg {
  double penalty_constant;
  double step;              /* penalty_step for single / pacbio (gaml.cc:818) */
  double min_prob_per_base;
  double min_prob_start;
  double weight;
  double mismatch_prob;     /* match_prob = 1 - 4*mismatch_prob */
} gaml_single_cfg;

/* PairedReadConfig (prob_calculator.h:20-35). step = insert_mean - penalty_step (gaml.cc:860). */
typedef struct gaml_paired_cfg {
  double penalty_constant;
  double step;
  double insert_mean;
  double insert_std;
  double min_prob_per_base;
  double min_prob_start;
  double weight;
  double mismatch_prob;
} gaml_paired_cfg;

/* ---- life cycle ------------------------------------------------------------------- */
/* device >= 0: HIP device ordinal (required for scoring). device = -1: host-only context:
 * graph / read sets / window alignment / occurrence tables work, every scoring call fails
 * with GAML_HIP_ENODEVICE (there is no CPU scoring path in this library). */
int gaml_hip_create(gaml_hip_ctx** out, int device);
void gaml_hip_destroy(gaml_hip_ctx* ctx);
const char* gaml_hip_last_error(const gaml_hip_ctx* ctx);
const char* gaml_hip_version(void);

#define GAML_HIP_OK 0
#define GAML_HIP_EINVAL (-1)
#define GAML_HIP_ENODEVICE (-2)
#define GAML_HIP_EHIP (-3)
#define GAML_HIP_ESTATE (-4)

/* ---- several GPUs -------------------------------------------------------------------
 * The reference's caller is ONE process holding ONE ProbCalculator (gaml.cc:1010, prob_calculator.h:37-124);
 * nothing in the reference is parallel (the sharding is new design). Two ways to more than one GPU:
 *
 * (1) gaml_hip_create_multi: one context that owns n_devices shards, one per listed HIP device (a device may be
 *     listed more than once, or be -1 for a host-only shard: rehearsals and tests). Reads are split by contiguous
 *     id range (shard i of n keeps [N*i/n, N*(i+1)/n)), graph / paths / window registration are replicated, each
 *     shard is driven by its own host thread. Every entry point of this header then works on the whole set:
 *     gaml_hip_calc_prob returns the value over all reads. Per evaluation the shards exchange ONE sum of 4 f64 per
 *     read set -- by an RCCL all-reduce over xGMI on the shards' streams (one communicator per shard,
 *     ncclCommInitAll; the default when every shard has its own GPU) or, as the measured alternative, by adding the
 *     shards' pinned-host partials in rank order on the calling thread (gaml_hip_set_exchange). The cold-path
 *     maximum of new windows' record positions and, with penalty_constant > 0, the union of the coverage maps /
 *     PacBio alignment intervals are merged in the library too (device lists). Environment: GAML_HIP_EXCHANGE=host|rccl picks the
 *     initial exchange ("rccl": creation fails when RCCL cannot be used).
 * (2) one process per GPU (torch.distributed.run, MPI): every process creates a plain context, calls
 *     gaml_hip_set_shard(rank, world) (or gaml_hip_set_presharded) and then gaml_hip_comm_init_rank with an id
 *     that rank 0 made with gaml_hip_comm_unique_id and sent to the others by any means (128 bytes). From then on
 *     gaml_hip_calc_prob / gaml_hip_calc_prob_batch are collective calls -- same paths on every rank, same value
 *     back on every rank -- with the same in-library RCCL exchanges as (1). */
int gaml_hip_create_multi(gaml_hip_ctx** out, const int32_t* devices, int32_t n_devices);
/* What the ProbCalculator adapters call: environment GAML_HIP_DEVICES unset -> gaml_hip_create(out, 0); "all" -> every
 * visible HIP device; "0,1,2,3" -> those (one entry: a plain context; several: gaml_hip_create_multi). */
int gaml_hip_create_from_env(gaml_hip_ctx** out);
int gaml_hip_num_shards(const gaml_hip_ctx* ctx);  /* 1 for a plain context */
#define GAML_HIP_EXCHANGE_HOST 0
#define GAML_HIP_EXCHANGE_RCCL 1            /* ncclAllGather of the ranks' partials, summed in RANK ORDER by every rank: the same
                                             * doubles in the same order on every rank and in every run (default) */
#define GAML_HIP_EXCHANGE_RCCL_ALLREDUCE 2  /* ncclAllReduce(sum): the order of the additions is RCCL's choice */
int gaml_hip_set_exchange(gaml_hip_ctx* ctx, int32_t mode);  /* multi-device contexts only; RCCL needs distinct devices */
int gaml_hip_get_exchange(const gaml_hip_ctx* ctx);
#define GAML_HIP_COMM_ID_BYTES 128
int gaml_hip_comm_unique_id(void* id_out /* GAML_HIP_COMM_ID_BYTES */);
int gaml_hip_comm_init_rank(gaml_hip_ctx* ctx, const void* id /* GAML_HIP_COMM_ID_BYTES */, int32_t rank, int32_t world);

/* ---- inputs ----------------------------------------------------------------------- */
/* Graph node sequences, what Graph::nodes[i]->s holds after LoadGraph (graph.cc:52-106):
 * n_nodes = 2 * velvet nodes, twin of i is i^1. bases = concatenation, offs[n_nodes+1]. */
int gaml_hip_set_graph(gaml_hip_ctx* ctx, int32_t n_nodes, const char* bases, const int64_t* offs);
/* Same, parsed from a Velvet LastGraph file by the library (graph.cc:52-106). */
int gaml_hip_load_graph(gaml_hip_ctx* ctx, const char* lastgraph_file);

/* Read sets. Returns the read-set handle (>= 0). Reads = concatenated bases + offs[n+1];
 * what ReadSet::PreprocessReads + PrepareReadIndex leave in a ReadSet (graph.cc:1366-1415).
 * Order of creation within a kind = order of ProbCalculator's constructor vectors. */
int gaml_hip_add_single(gaml_hip_ctx* ctx, const gaml_single_cfg* cfg, int32_t n_reads,
                        const char* bases, const int64_t* offs);
int gaml_hip_add_paired(gaml_hip_ctx* ctx, const gaml_paired_cfg* cfg, int32_t n_pairs,
                        const char* bases1, const int64_t* offs1, const char* bases2, const int64_t* offs2);
/* PacBio: only read lengths are needed by the scoring side (graph.cc:3062-3088, graph.h:478-481). */
int gaml_hip_add_pacbio(gaml_hip_ctx* ctx, const gaml_single_cfg* cfg, int32_t n_reads, const int32_t* read_lens);
/* FASTQ-file variants (read ids by first appearance of the name, graph.h:410-420). */
int gaml_hip_add_single_fastq(gaml_hip_ctx* ctx, const gaml_single_cfg* cfg, const char* fastq);
int gaml_hip_add_paired_fastq(gaml_hip_ctx* ctx, const gaml_paired_cfg* cfg, const char* fastq1, const char* fastq2);
int gaml_hip_add_pacbio_fastq(gaml_hip_ctx* ctx, const gaml_single_cfg* cfg, const char* fastq);

/* Read sharding for multi-GPU runs (new design): this context keeps and
 * scores only reads [n*rank/world, n*(rank+1)/world) of every read set. Call before adding
 * read sets. Partial results are combined with gaml_hip_combine_partials. */
int gaml_hip_set_shard(gaml_hip_ctx* ctx, int32_t rank, int32_t world);
/* Alternative: the caller already hands each process only its own reads (world processes in
 * all). Nothing is partitioned; the context only knows that window maxima must be exchanged
 * with the other processes (gaml_hip_eval_* below). */
int gaml_hip_set_presharded(gaml_hip_ctx* ctx, int32_t world);

/* Alignment records computed outside the library (the reference's external-aligner branch
 * graph.cc:924-1033 produces exactly these per sub-walk; BLASR output for PacBio,
 * graph.cc:2776-2782). mate = 0/1 for paired sets, 0 otherwise. read_id is global. */
int gaml_hip_put_window_records(gaml_hip_ctx* ctx, int readset, int mate, const int32_t* subpath, int32_t subpath_len,
                                const gaml_aligment* recs, int64_t n);
int gaml_hip_put_pacbio_records(gaml_hip_ctx* ctx, int readset, const int32_t* subpath, int32_t subpath_len,
                                const gaml_pacbio_aligment* recs, int64_t n);

/* ---- PacBio cache-miss side: BLASR's SAM output -> cached alignment records -------------------
 * The reference, on a path with an uncached sub-walk, writes the path to a file, runs BLASR and
 * turns every SAM line into a cached record (PacbioReadSet::GetReadProbabilitiesSlow
 * graph.cc:2650-2795): ParseAligment (:2945-3021), the banded sum-over-alignments probability
 * AligmentProbability (:2175-2297, on the GPU here) and the filing rule (:2760-2782).  Running
 * BLASR stays with the caller; these three entry points are the rest.
 *
 * gaml_hip_add_pacbio_reads: like gaml_hip_add_pacbio but keeps the read bases (read_seq_,
 *   PacbioReadSet::PreprocessReads graph.cc:1416-1439) and the read names (GetReadId graph.h:410-420);
 *   `names` = n names, each followed by '\n'.  Reads may not hold '\n' or '-'.
 * gaml_hip_pacbio_missing: the index ranges [begin, end] of `path` the reference would hand to
 *   BLASR (merged ranges of uncached sub-walks, GetReadProbabilities graph.cc:2438-2478); writes
 *   up to `cap` pairs to ranges_out and returns the number of ranges.
 * gaml_hip_pacbio_ingest_sam: `path` is the sub-path that was aligned (as given to BLASR:
 *   path string + separator + reverse complement), `sam` the SAM text ('@' lines are skipped).
 *   Every sub-walk of the path that was not cached before gets a cache entry; a record is filed
 *   under the sub-walk it spans exactly as the reference does.  Unknown read names and lines
 *   with fewer than 10 columns are errors (the reference asserts / reads out of range). */
int gaml_hip_add_pacbio_reads(gaml_hip_ctx* ctx, const gaml_single_cfg* cfg, int32_t n_reads, const char* bases,
                              const int64_t* offsets, const char* names);
int32_t gaml_hip_pacbio_missing(gaml_hip_ctx* ctx, int readset, const int32_t* path, int32_t path_len, int32_t* ranges_out,
                                int32_t cap);
int gaml_hip_pacbio_ingest_sam(gaml_hip_ctx* ctx, int readset, const int32_t* path, int32_t path_len, const char* sam,
                               int64_t sam_len, int64_t* filed_out);
/* cached records of one sub-walk (global read ids); -1 when the sub-walk is not cached */
int64_t gaml_hip_pacbio_records(gaml_hip_ctx* ctx, int readset, const int32_t* subpath, int32_t subpath_len,
                                gaml_pacbio_aligment* out, int64_t cap);
/* last ingest: {SAM records, DP jobs, DP rows, DP cells, kernel ms (HIP events), host parse+prepare ms,
 * upload+kernel+download ms, scratch bytes} */
int gaml_hip_pacbio_dp_stats(gaml_hip_ctx* ctx, int readset, double* out8);
/* ---- the hot path ------------------------------------------------------------------ */
/* ProbCalculator::CalcProb(paths, zeros, total_len) (prob_calculator.h:63-109):
 * paths = flattened node ids (>= 0) / gaps (< 0), path_offs[n_paths+1].
 * zeros_out: 2 ints per read set (low-probability reads, reads) in the reference's order:
 * single sets, then paired, then pacbio, each in creation order. May be NULL.
 * The value is a pure function of (paths, alignment-window cache): every call rescoring all
 * reads from scratch on the GPU (= the reference evaluated with a fresh ScoringState). */
int gaml_hip_calc_prob(gaml_hip_ctx* ctx, const int32_t* paths, const int64_t* path_offs, int32_t n_paths,
                       double* prob_out, int32_t* zeros_out, int32_t* total_len_out);

/* Sharded form: per read set 4 doubles {sum of log-probabilities over this shard's reads,
 * floored reads, bad_bases (valid on every rank), reads in shard}. After an all-reduce(sum)
 * of elements 0, 1 and 3 (element 2 is replicated: take any rank's), gaml_hip_combine_partials
 * gives CalcProb's value. n_sets = gaml_hip_num_readsets. */
int gaml_hip_calc_partials(gaml_hip_ctx* ctx, const int32_t* paths, const int64_t* path_offs, int32_t n_paths,
                           double* partials_out /* 4 * n_sets */, int32_t* total_len_out);
int gaml_hip_combine_partials(gaml_hip_ctx* ctx, const double* partials /* 4 * n_sets, reduced */,
                              int32_t total_len, double* prob_out, int32_t* zeros_out);

/* Two-phase form, REQUIRED for sharded contexts whenever new windows get aligned (cold path):
 * the reference drops an alignment whose path position is more than 5 below the largest position
 * seen at earlier nodes of the contig (graph.cc:577) -- a maximum over ALL reads, so shards must
 * exchange the largest record position of every newly aligned window once.
 *   gaml_hip_eval_begin          registration + alignment + window placement; *pending_out = number
 *                                of windows whose maxima need exchanging (always 0 when world == 1,
 *                                and 0 in a sharded run once the window cache is warm)
 *   gaml_hip_eval_pending_maxpos the local maxima, in an order that is identical on every rank
 *   (caller: all-reduce(max) over ranks)
 *   gaml_hip_eval_apply_maxpos   install the reduced maxima
 *   gaml_hip_eval_finish[_async] thresholds, device tables, kernels; partials as gaml_hip_calc_partials */
int gaml_hip_eval_begin(gaml_hip_ctx* ctx, const int32_t* paths, const int64_t* path_offs, int32_t n_paths,
                        int64_t* pending_out, int32_t* total_len_out);
int64_t gaml_hip_eval_pending_maxpos(gaml_hip_ctx* ctx, int32_t* out, int64_t cap);
int gaml_hip_eval_apply_maxpos(gaml_hip_ctx* ctx, const int32_t* reduced, int64_t n);
int gaml_hip_eval_finish(gaml_hip_ctx* ctx, double* partials_out /* 4 * n_sets */);
int gaml_hip_eval_finish_async(gaml_hip_ctx* ctx, void* d_partials, void* hip_stream);
/* wait for everything enqueued on the library's private stream */
int gaml_hip_sync(gaml_hip_ctx* ctx);

/* Single-node exchange of the partials through POSIX shared memory (new; the reference is one process). After a
 * BLOCKING evaluation (gaml_hip_eval_finish / gaml_hip_calc_partials) a rank's partials are host values; the ranks of
 * one node add them up here in ~1 us instead of sending them back through the device for an RCCL all-reduce of 32
 * bytes (~30 us of dependent dispatches). Every rank opens the same `name` (e.g. "/gaml_<port>") with its rank and
 * the world size -- rank 0 FIRST (it replaces whatever block of that name exists), the others after a barrier of the
 * caller's; _allreduce_sum is collective, sums in rank order (identical bits on every rank) and waits at most
 * 30 s for the others. Not for read sets with penalty_constant > 0 on a sharded context (those need the coverage
 * exchange of gaml_hip_eval_score_async). _close(unlink_name = 1) on one rank removes the name. */
int gaml_hip_shm_exchange_open(gaml_hip_ctx* ctx, const char* name, int32_t rank, int32_t world, int32_t cap_doubles);
int gaml_hip_shm_allreduce_sum(gaml_hip_ctx* ctx, double* inout, int32_t n_doubles);
int gaml_hip_shm_exchange_close(gaml_hip_ctx* ctx, int32_t unlink_name);

/* Device values -> host at the end of a stream-ordered sequence (after the all-reduce of the partials, say) without a
 * D2H copy command and without the runtime's completion wake-up: _async enqueues a one-block kernel on `hip_stream`
 * (NULL: the library's stream) that writes n_doubles values from d_src and then a sequence word into mapped pinned
 * memory; _wait polls that word (bounded; then hipStreamSynchronize) and copies the values out. One fetch in flight per
 * context. Nothing in the reference corresponds to it (single process, host arithmetic). */
int gaml_hip_fetch_async(gaml_hip_ctx* ctx, const void* d_src, int32_t n_doubles, void* hip_stream);
int gaml_hip_fetch_wait(gaml_hip_ctx* ctx, double* out, int32_t n_doubles);

/* Maintenance hint.  Windows aligned after the device record tables of a paired set were built are
 * scored from delta lists (slightly slower per pair); the library folds them into the tables by
 * itself when they grow past 1/8 of the pairs or after 64 evaluations without a new window.  This
 * call asks for the fold at the next evaluation -- e.g. after a warm-up phase, before a long run of
 * re-scoring.  Results do not change (only the order of the final sum, i.e. last bits). */
int gaml_hip_compact_tables(gaml_hip_ctx* ctx);

/* Sharded evaluation with a coverage penalty (penalty_constant > 0 on a paired set; SURVEY 8e "the one
 * non-separable piece").  bad_bases (graph.cc:1893-1919) is a function of the union of every rank's
 * well-aligned pair positions, so the sweep has to wait for the other ranks' coverage maps:
 *   n = gaml_hip_eval_score_async(ctx, d_partials, stream)   instead of gaml_hip_eval_finish_async:
 *       scoring kernels of all read sets; returns how many coverage maps await the exchange (0: done);
 *   for i < n:
 *     gaml_hip_eval_coverage_export_async(ctx, i, dst, cap, &bytes, stream)   copy this rank's map (one bit
 *         per path base, same size on every rank) into caller memory, e.g. a torch tensor (dst NULL: size only);
 *     all-gather the maps (RCCL);
 *     gaml_hip_eval_coverage_finish_async(ctx, i, maps, n_maps, contribute, stream)   OR the n_maps gathered
 *         maps into this rank's, run the sweep, store bad_bases into the partials: the value where
 *         `contribute` is non-zero (exactly one rank, e.g. rank 0), 0 elsewhere, so that the
 *         all-reduce(sum) of the partials counts it once.
 * Single-end sets need no exchange (their bad_bases is identically 0 in the reference, graph.cc:1701-1733).
 * PacBio sets with a penalty (graph.cc:3198-3250: sweep over the alignment intervals of ALL reads): after
 * gaml_hip_eval_score_async, for i < gaml_hip_eval_pacbio_pending(ctx):
 *     n = gaml_hip_eval_pacbio_intervals(ctx, i)   how many alignment intervals this rank's reads contribute
 *         (known on the host, no synchronisation);
 *     gaml_hip_eval_pacbio_export_async(ctx, i, dst, cap, stream)   copy them -- 16 bytes each: int32 {contig,
 *         begin, end, 0} in path coordinates -- into caller DEVICE memory with room for `cap` intervals;
 *     all-gather them (counts differ per rank: gather the counts first); then
 *     gaml_hip_eval_pacbio_finish_async(ctx, i, intervals, n_intervals, contribute, stream)   `intervals`: the
 *         n_intervals gathered intervals of all ranks (this rank's among them) in device memory; sorts them
 *         together with the contigs' node intervals, runs the sweep on the device and stores bad_bases into the
 *         partials like the paired form. No interval ever visits the host.
 * `stream` everywhere here: the stream the caller orders its own copies and collectives on. NULL is the legacy default
 * stream (what torch.cuda.current_stream().cuda_stream is on torch's default stream) and is used as such -- the
 * library's work is ordered against the caller's other work on it like on any other handle. */
int32_t gaml_hip_eval_score_async(gaml_hip_ctx* ctx, void* d_partials, void* stream);
int gaml_hip_eval_coverage_export_async(gaml_hip_ctx* ctx, int32_t i, void* dst, int64_t cap, int64_t* bytes_out, void* stream);
int gaml_hip_eval_coverage_finish_async(gaml_hip_ctx* ctx, int32_t i, const void* maps, int32_t n_maps, int32_t contribute,
                                        void* stream);
int32_t gaml_hip_eval_pacbio_pending(gaml_hip_ctx* ctx);
int64_t gaml_hip_eval_pacbio_intervals(gaml_hip_ctx* ctx, int32_t i);
int gaml_hip_eval_pacbio_export_async(gaml_hip_ctx* ctx, int32_t i, void* dst, int64_t cap, void* stream);
int gaml_hip_eval_pacbio_finish_async(gaml_hip_ctx* ctx, int32_t i, const void* intervals, int64_t n_intervals, int32_t contribute,
                                      void* stream);

/* Device-resident form for callers that already own a HIP stream (e.g. torch): enqueue the
 * whole evaluation on `stream` and leave the 4*n_sets partials in device memory at
 * d_partials (f64). No host synchronisation. `hip_stream`: the stream the caller orders its own work
 * on; NULL is the legacy default stream (gaml_hip_sync waits for it as well as for the library's own). */
int gaml_hip_calc_partials_async(gaml_hip_ctx* ctx, const int32_t* paths, const int64_t* path_offs, int32_t n_paths,
                                 void* d_partials, void* hip_stream, int32_t* total_len_out);

/* Batched speculative scoring.  The move generators evaluate several near-identical
 * path sets and keep the best (moves.cc:107-113 LocalChange2, 694-800 FixGapLength, 1156-1305
 * FixRepForNode2): n_sets independent CalcProb calls whose results are only compared afterwards.
 * This entry point takes them in one call: the evaluations are enqueued back to back on the
 * library's stream, the host prepares set i+1 while the device scores set i, and there is ONE
 * synchronisation and ONE device->host copy for the whole batch.  Results (and the window cache
 * afterwards) are those of n_sets gaml_hip_calc_prob calls in the same order.
 *   paths / path_offs : all paths of all sets, concatenated (path k = paths[path_offs[k] .. path_offs[k+1]))
 *   set_offs[n_sets+1]: set i = paths set_offs[i] .. set_offs[i+1]-1
 *   probs_out[n_sets]; zeros_out[n_sets * 2 * num_readsets] and total_lens_out[n_sets] may be NULL.
 * Not for sharded contexts (GAML_HIP_ESTATE): there the batch is a loop over the gaml_hip_eval_*
 * protocol with one all-reduce over all sets' partials (gaml_amd/dist.py: ShardedScorer.calc_prob_batch). */
int gaml_hip_calc_prob_batch(gaml_hip_ctx* ctx, int32_t n_sets, const int32_t* paths, const int64_t* path_offs,
                             const int32_t* set_offs, double* probs_out, int32_t* zeros_out, int32_t* total_lens_out);


/* ---- introspection (tests, bench, logging) ----------------------------------------- */
int gaml_hip_num_readsets(const gaml_hip_ctx* ctx);
int gaml_hip_readset_kind(const gaml_hip_ctx* ctx, int readset);   /* 0 single, 1 paired, 2 pacbio */
int64_t gaml_hip_readset_reads(const gaml_hip_ctx* ctx, int readset); /* global read (pair) count */
int32_t gaml_hip_num_nodes(const gaml_hip_ctx* ctx);
int32_t gaml_hip_node_len(const gaml_hip_ctx* ctx, int32_t node);
/* per-read values of the last evaluation of a read set: probs (linear; log for pacbio),
 * n = reads in this shard. ScoringState::probs (graph.h:612-619) for paired sets. */
int gaml_hip_read_probs(gaml_hip_ctx* ctx, int readset, double* out, int64_t n);
/* bad_bases of the last evaluation (graph.cc:1893-1919 / 1701-1733 / 3226-3250). */
int gaml_hip_bad_bases(gaml_hip_ctx* ctx, int readset, int64_t* out);
/* alignment-window cache (aligment_cache_, graph.h:427): number of windows, and the sorted
 * records of one window (returns count, -1 if the window is not cached). */
int64_t gaml_hip_window_count(const gaml_hip_ctx* ctx, int readset, int mate);
int64_t gaml_hip_window_records(gaml_hip_ctx* ctx, int readset, int mate, const int32_t* subpath, int32_t subpath_len,
                                gaml_aligment* out, int64_t cap);
/* force alignment of one window with the library's own aligner (AlignSubpathInternal graph.cc:839-899) */
int64_t gaml_hip_align_window(gaml_hip_ctx* ctx, int readset, int mate, const int32_t* subpath, int32_t subpath_len);
/* GPU window aligner (cold path): windows aligned on the device so far, seed candidates extended, wall time.
 * (Development builds: knob 5 = 1, gaml_hip_debug.h, forces the host aligner.) */
int gaml_hip_aligner_stats(gaml_hip_ctx* ctx, int64_t* windows, int64_t* candidates, double* microseconds);
/* the same time by stage, host clock, cumulative: out6 = {window strings + upload, spans + candidates (small batches: the
 * whole device pipeline up to the published headers), extension, hits to the host, ordering + filing on the host, batches} */
int gaml_hip_aligner_stages(gaml_hip_ctx* ctx, double* out6);
/* ---- monitoring (what bench.py prints beside its numbers; all cheap, host-side) --------------------------------- */
/* pairs per record-count class of a paired set's device tables {<= 1 record per mate, <= 2, <= 4, more} */
int gaml_hip_pair_classes(gaml_hip_ctx* ctx, int readset, int64_t* out4);
/* device record tables of a paired set (built and maintained by kernels: the records never leave the device): {table
 * builds, delta updates, pairs currently on the delta lists, builds that ran BESIDE the evaluations on a stream of their
 * own and took over a fixed number of evaluations later (of the table builds), gaml_hip_calc_prob_batch chunks whose per-set tables were built on
 * the device from patches, chunks whose tables were written whole, records of mate 1 / mate 2 that the current tables
 * leave out because another record of the same read always overwrites them, such records of windows that joined later
 * and therefore never reached the delta lists (since creation), pairs of the compact class whose pair term came with
 * the tables (both records in one window, or a mate without alignment)} */
int gaml_hip_table_stats(gaml_hip_ctx* ctx, int readset, int64_t* out10);
/* host-side phase times of the last blocking paired evaluation, microseconds: [0] pass 1 (planner; includes [2]),
 * [1] thresholds + occurrence tables, [2] alignment of newly registered windows (inside pass 1), [3] per-call tables
 * written, [4] record tables / delta lists brought up to date, [5] kernel launches, [6] bytes of per-call tables
 * written, [7] wait for the device */
int gaml_hip_last_phases(gaml_hip_ctx* ctx, double* out8);
/* timing of the last scoring call, microseconds: [0] host preparation (window registration,
 * alignment of new windows, occurrence tables), [1] H2D + kernels + D2H wall, [2] device time
 * of the scoring kernels measured with HIP events on the library's stream (0 if events off). */
int gaml_hip_last_timing(const gaml_hip_ctx* ctx, double* out3);
/* HIP event timing of the dominant kernel (default off): on = 1 times every scoring launch, on = k > 1
 * every k-th. The two events are attached to the dispatch itself (hipExtLaunchKernelGGL), so they hold the
 * kernel's own begin / end stamps -- what rocprofv3's kernel trace reports -- at ~4 us of extra host time per
 * timed launch; sampling keeps that out of a throughput measurement. gaml_hip_kernel_stats then describes the
 * timed launches (count, summed device time, summed algorithmic bytes). */
int gaml_hip_set_event_timing(gaml_hip_ctx* ctx, int on);
/* cumulative kernel statistics since the last reset: launches, total device microseconds
 * (HIP events), algorithmic bytes streamed (SURVEY.md 8d accounting). */
int gaml_hip_kernel_stats(gaml_hip_ctx* ctx, int reset, int64_t* launches, double* device_us, double* algo_bytes);

#ifdef __cplusplus
}
#endif
#endif /* GAML_HIP_H_ */
