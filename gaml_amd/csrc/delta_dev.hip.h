// delta_dev.hip.h -- the delta lists of a paired set, kept ON THE DEVICE: pairs that gained records since the record tables
// were built (windows activated later: an annealing move's new junctions, a reversed path's twins). The reference finds a
// read's alignments through per-call hash maps (GetPositionsOnlyPath graph.cc:535-598) over its window cache (graph.cc:
// 911-922); here a window's records join the read-major tables without leaving HBM and without a table rebuild:
//
//   delta_apply_kernel   ONE block per launch, up to kDlMaxRecs records of up to kDlMaxWins newly activated windows:
//     1. a lane per record: left out when the first node's own window always overwrites it (graph.cc:563-592; the same
//        test the table build applies), else keyed by (slot of its pair, mate, position in the launch's record sequence)
//     2. the keys sorted in LDS (bitonic; they are distinct, so the order is a function of the input alone)
//     3. per touched pair, in slot order: its present list -- from the delta store, or from the tables when the pair is new
//        to the lists -- merged with the new records in (window id, position) order (the tables' order), written back at a
//        fixed stride (up to 4 records per mate) or to the spill area (longer lists: scored one wave per pair); new pairs
//        are numbered in slot order and their table slots marked.
// Every number handed out (delta index, spill index, spill space) is a prefix sum in slot order: equal inputs give equal
// lists and equal numbering, run to run (the order of the final sum depends on the numbering; SURVEY 8b: ties matter).
// Larger activations are cut into several launches on the stream; the lists compose (a launch merges into what is there).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hip.h"
#include "table_build.hip.h"

namespace gaml {

constexpr int kDlThreads = 1024, kDlMaxRecs = 8192, kDlMaxWins = 64;
constexpr int kDlBins = 32, kDlBinCap = kDlThreads * 8, kDlMbMaxRecs = 49152;  // multi-block launches: bins of keys by a hash of the pair's slot
// (the store's device counters, kDs*: kernels.hip.h)

struct DlWin { int mate, wid, first, count, dom_first, dom_count, start; };
struct DlArgs {
  int n_wins, n_total;
  const int4* pool[2];
  // the live tables
  unsigned long long* rec8[2];
  int4* first[2];
  const int4* extra[2];
  int4* inl0;
  const unsigned char* len_code;
  const unsigned* len_combo;
  const unsigned* len12;
  const int* slot_of_read;
  int* dirty_of_slot;
  int n0, n01, n_main;
  // the delta store
  int* dl_slot;
  int* dl_spill;
  int4* dl_rec[2];
  int2* sp_rng[2];
  int4* sp_rec[2];
  int* sp_slot;
  int* state;       // kDs*
  int* host_state;  // the same words in pinned host memory, written at the end of every launch
  int cap_pairs, cap_spill, cap_sprec, seq;
  int slot_bits;    // bits a slot number takes (the sort's passes)
  // multi-block launches (delta_mb_*): the bins, how many keys each holds, per block what it hands out + the counters' values before the launch
  unsigned long long* bins;
  int* bin_count;
  int* blk_tot;     // [kDlBins][4], then [4] base values, then [1] records left out, [1] a bin overflowed
  unsigned long long* stamps;  // timing probes (development): wall-clock stamps (10 ns) of thread 0 at the stage boundaries, or null
  const DlWin* wlist;  // multi-block launches of more windows than w[] holds: the list in device memory (else null)
  DlWin w[kDlMaxWins];
};

// the window that holds position p of the launch's record sequence: from the argument block's copy in LDS, or -- launches
// of many windows -- by bisection of the list in device memory
__device__ __forceinline__ DlWin dl_window_of(const DlArgs& a, const DlWin* sw, int p) {
  if (a.wlist) {
    int lo = 0, hi = a.n_wins - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (a.wlist[mid].start <= p) lo = mid; else hi = mid - 1; }
    return a.wlist[lo];
  }
  int k = 0;
  while (k + 1 < a.n_wins && sw[k + 1].start <= p) k++;
  return sw[k];
}

// where a pair's present list of one mate comes from
struct DlOld { int kind, n; const int4* p; int4 first; };  // kind 0: p[k] (delta store / spill area); 1: tables (first + extra)
__device__ __forceinline__ int4 dl_old_get(const DlOld& o, int k) {
  int4 r = o.kind == 0 ? o.p[k] : (k == 0 ? o.first : o.p[o.first.w + k - 1]);
  r.z &= 0x1ff; r.w = 0;
  return r;
}
__device__ __forceinline__ bool dl_before_eq(const int4& x, const int4& y) { return x.x != y.x ? x.x < y.x : x.y <= y.y; }  // x goes first (an old record before an equal new one: upper_bound)

// Bitonic sort of N (a power of two >= 64) distinct keys in LDS, ascending. A thread takes compare-exchange t, t + nthr, ...;
// with strides below 64 both elements of every exchange of a wave's 64 threads lie in the 128 keys that wave owns, so only
// the steps with larger strides need the block's barrier -- 15 of the 66 steps at 2,048 keys (a barrier over sixteen waves
// per step was 16 of a 40 us launch).
__device__ __forceinline__ void dl_bitonic_sort(unsigned long long* keys, int N, int tid, int nthr) {
  __syncthreads();
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (N >> 1); t += nthr) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
        const unsigned long long x = keys[i], y = keys[l];
        const bool up = (i & k) == 0;
        if ((x > y) == up) { keys[i] = y; keys[l] = x; }
      }
      // the next step's partners: within the wave's own keys when its stride is below 64 and so was this one's
      const int jn = j > 1 ? j >> 1 : k;  // (the next stage starts at stride k)
      if (j >= 64 || jn >= 64) __syncthreads();
      else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
  __syncthreads();
}

// Grouping without a sort (one-block launches of up to 4,096 records): the keys of one pair brought together, pairs in the
// order of their first record in the launch's sequence, a pair's records in sequence order -- the same lists as a sort by
// (slot, mate, sequence) gives, the pairs numbered by first touch instead of by slot (either is a function of the input
// alone). A hash table in LDS keyed by slot takes every record (compare-and-swap on the key, then count / smallest
// sequence number / a chain through the slot's records by atomics: their order is arbitrary, what is read off them is
// not); the record whose number is the slot's smallest is its head; a scan over the heads in sequence order places the
// groups; a record's place in its group is the number of chain members before it in the sequence. Four dependent LDS
// steps and a scan per 1,024 records, where the bitonic network took 55-78 (12 us at 1,024 keys, 31 us at 4,096).
struct DlGroupLds { int* tkey; int* thead; int* tcnt; int* tmin; int* next; int* scan /* [nthr / 64 + 1] */; };
template <int PER>
__device__ __forceinline__ void dl_group(const unsigned long long* keys, unsigned long long* out, int N, int tid, int nthr, const DlGroupLds& L) {
  const int H = N;  // (a power of two >= 64; at most N distinct slots)
  const int lane = tid & 63, wave = tid >> 6, nwav = nthr >> 6;
  for (int h = tid; h < H; h += nthr) { L.tkey[h] = -1; L.thead[h] = -1; L.tcnt[h] = 0; L.tmin[h] = 0x7fffffff; }
  for (int p = tid; p < N; p += nthr) out[p] = ~0ull;
  __syncthreads();
  int hp[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const int p = tid + j * nthr;
    hp[j] = -1;
    if (p >= N || keys[p] == ~0ull) continue;
    const int slot = (int)(keys[p] >> 33);
    int h = (int)(((unsigned)slot * 2654435761u) >> 7) & (H - 1);
    for (;;) {
      const int k = atomicCAS(&L.tkey[h], -1, slot);
      if (k == -1 || k == slot) break;
      h = (h + 1) & (H - 1);
    }
    hp[j] = h;
    atomicAdd(&L.tcnt[h], 1);
    atomicMin(&L.tmin[h], p);
    L.next[p] = atomicExch(&L.thead[h], p);
  }
  __syncthreads();
  // the groups' places: heads in sequence order (round by round: p = tid + j * nthr), each followed by its slot's records
  bool is_head[PER];
  int size[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) { is_head[j] = hp[j] >= 0 && L.tmin[hp[j]] == tid + j * nthr; size[j] = is_head[j] ? L.tcnt[hp[j]] : 0; }
  __syncthreads();  // (every head flag is settled before tmin is reused for the groups' places)
  int carry = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    int incl = size[j];
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d, 64); if (lane >= d) incl += u; }
    if (lane == 63) L.scan[wave] = incl;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wave; w++) before += L.scan[w];
    int total = 0;
    for (int w = 0; w < nwav; w++) total += L.scan[w];
    if (is_head[j]) L.tmin[hp[j]] = before + incl - size[j];
    carry += total;
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < PER; j++) {
    if (hp[j] < 0) continue;
    const int p = tid + j * nthr;
    int rank = 0;
    for (int q = L.thead[hp[j]]; q >= 0; q = L.next[q]) rank += q < p;
    out[L.tmin[hp[j]] + rank] = keys[p];
  }
  __syncthreads();
}

// PER: sorted positions per thread (kDlThreads * PER >= the launch's records, rounded up to a power of two). A thread's
// PER items go through every stage side by side (their loads in flight together): one lane working off eight chains of
// dependent loads one after the other was the kernel's duration.
// MODE 0: one block does everything. MODE 1 / 2: a block per bin of a multi-block launch (keys by delta_mb_keys_kernel) --
// 1 only counts what the block will hand out (new pairs, spill entries, spill records), 2 writes, its numbers starting
// behind those of the blocks before it.
template <int PER, int MODE = 0>
__global__ __launch_bounds__(kDlThreads) void delta_apply_kernel(DlArgs a) {
  constexpr bool kGroup = PER <= 4 && MODE == 0;  // keys grouped through a hash table (dl_group); else sorted (bitonic network)
  __shared__ unsigned long long keys_a[kDlThreads * PER], keys_b[kGroup ? kDlThreads * PER : 1];
  __shared__ __attribute__((aligned(16))) int g_tab[kGroup ? 5 * kDlThreads * PER : 4];
  __shared__ int g_scan[kDlThreads / 64 + 1];
  unsigned long long* keys = keys_a;
  constexpr bool kCache = kGroup;  // the launch's records sit in LDS for the merge (where the grouping's tables were: 20 bytes per key, a record takes 16)
  int4* const rec_lds = (int4*)g_tab;
  __shared__ int sc[4][kDlThreads / 64];
  __shared__ int tot[4];
  __shared__ int n_left_out;
  __shared__ DlWin sw[kDlMaxWins];  // (indexed by a lane's own window number: from LDS, not from a scratch copy of the argument block)
  if (threadIdx.x == 0) n_left_out = 0;
  if (threadIdx.x < kDlMaxWins) sw[threadIdx.x] = a.w[(int)threadIdx.x < a.n_wins && !a.wlist ? threadIdx.x : 0];
  __syncthreads();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = (int)blockDim.x, nwav = nthr >> 6;  // (PER == 1: as many threads as the records' power of two, 64 .. 1,024)
#define DL_STAMP(k) do { if (a.stamps && tid == 0 && blockIdx.x == 0) a.stamps[k] = wall_clock64(); } while (0)
  DL_STAMP(0);
  const int n_here = MODE == 0 ? a.n_total : min(a.bin_count[blockIdx.x], kDlBinCap);
  const bool mb_bad = MODE != 0 && (a.blk_tot[4 * kDlBins + 5] != 0);  // a bin overflowed: the launch changes nothing
  int N = 64;
  while (N < n_here) N <<= 1;
  int4 r[PER];
  if (MODE != 0) {
    for (int p = tid; p < N; p += nthr) keys[p] = p < n_here ? a.bins[(size_t)blockIdx.x * kDlBinCap + p] : ~0ull;
  } else
  // ---- 1. keys: record j of this thread is position tid + j * kDlThreads of the launch's record sequence
  {
    int kw[PER], lo[PER], hi[PER];
    bool live[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const int p = tid + j * nthr;
      live[j] = p < a.n_total;
      int k = 0;
      if (live[j]) while (k + 1 < a.n_wins && sw[k + 1].start <= p) k++;
      kw[j] = k;
      r[j] = live[j] ? a.pool[sw[k].mate][sw[k].first + (p - sw[k].start)] : make_int4(0, 0, 0, 0);
      lo[j] = 0; hi[j] = live[j] ? sw[k].dom_count : 0;
    }
    // is the record among the dominating window's (ordered by (position, read))? Lower bound, sixteen-way: a round asks for
    // fifteen pivots at once (a binary search through a long node's 2,800 records was eleven dependent trips: most of this
    // kernel's duration for an annealing move's junction windows), the PER searches in step
    bool more = true;
    while (more) {
      more = false;
#pragma unroll
      for (int j = 0; j < PER; j++) {
        if (lo[j] >= hi[j]) continue;
        const int4* base = a.pool[sw[kw[j]].mate] + sw[kw[j]].dom_first;
        const int span = hi[j] - lo[j];
        int4 pv[15];
#pragma unroll
        for (int q = 0; q < 15; q++) pv[q] = base[lo[j] + (int)(((long long)span * (q + 1)) >> 4)];  // (pivot q: < hi, distinct positions may repeat when span < 16)
        int nlo = lo[j], nhi = hi[j];
#pragma unroll
        for (int q = 0; q < 15; q++) {
          const int at = lo[j] + (int)(((long long)span * (q + 1)) >> 4);
          if (tb_rec_before(pv[q], r[j].y, r[j].w)) nlo = max(nlo, at + 1); else nhi = min(nhi, at);
        }
        lo[j] = nlo; hi[j] = nhi;
        more = more || lo[j] < hi[j];
      }
    }
    int4 at[PER];
    int slot[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const bool in = live[j] && lo[j] < sw[kw[j]].dom_count;
      at[j] = in ? a.pool[sw[kw[j]].mate][sw[kw[j]].dom_first + lo[j]] : make_int4(0, -1, 0, -1);
      slot[j] = live[j] ? a.slot_of_read[r[j].w] : 0;
    }
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const int p = tid + j * nthr;
      if (p >= N) continue;
      unsigned long long key = ~0ull;
      if (live[j]) {
        const bool drop = sw[kw[j]].dom_count > 0 && lo[j] < sw[kw[j]].dom_count && at[j].y == r[j].y && at[j].w == r[j].w;
        if (!drop) key = ((unsigned long long)(unsigned)slot[j] << 33) | ((unsigned long long)sw[kw[j]].mate << 32) | (unsigned)p;
        else atomicAdd(&n_left_out, 1);
      }
      keys[p] = key;
    }
  }
  __syncthreads();
  // ---- 2. bitonic sort, ascending (a launch of few records runs with as many threads as records: a barrier over sixteen
  // waves per step of the network, 36-55 steps, was most of a small launch's duration)
  DL_STAMP(1);
  if (kGroup) {
    __syncthreads();
    const DlGroupLds G{g_tab, g_tab + N, g_tab + 2 * N, g_tab + 3 * N, g_tab + 4 * N, g_scan};
    dl_group<PER>(keys_a, keys_b, N, tid, nthr, G);
    keys = keys_b;
#pragma unroll
    for (int j = 0; j < PER; j++) { const int p = tid + j * nthr; if (p < N) rec_lds[p] = r[j]; }  // (the tables are done with: dl_group ended on a barrier)
    __syncthreads();
  } else dl_bitonic_sort(keys, N, tid, nthr);
  DL_STAMP(2);
  // ---- 3. per touched pair. Thread t owns the sorted positions [t * per, (t + 1) * per): numbers are handed out in that order.
  const int per = N > nthr ? N / nthr : 1;
  const int p_lo = tid * per;
  const int* base4 = MODE == 0 ? a.state : a.blk_tot + 4 * kDlBins;  // (multi-block: the counters as the count launch found them)
  const int nd0 = base4[kDsDirty], ns0 = base4[kDsSpill], top0 = base4[kDsTop0], top1 = base4[kDsTop1];
  // what a head finds: where its pair's present lists are and how long, how many records join them
  bool head[PER];
  int slot[PER], dj[PER], sp_old[PER], c0[PER], c1[PER], add0[PER], add1[PER], q_end[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const int p = p_lo + j;
    head[j] = j < per && p < N && keys[p] != ~0ull && (p == 0 || (keys[p - 1] >> 33) != (keys[p] >> 33));
    slot[j] = 0; add0[j] = add1[j] = 0; q_end[j] = p; dj[j] = -1; sp_old[j] = -1; c0[j] = c1[j] = 0;
    if (head[j]) {
      slot[j] = (int)(keys[p] >> 33);
      int q = p;
      while (q < N && keys[q] != ~0ull && (int)(keys[q] >> 33) == slot[j]) { if ((keys[q] >> 32) & 1ull) add1[j]++; else add0[j]++; q++; }
      q_end[j] = q;
    }
  }
#pragma unroll
  for (int j = 0; j < PER; j++) if (head[j]) dj[j] = a.dirty_of_slot[slot[j]];
  unsigned long long r8[2][PER];  // a compact-class pair's two records (what its lists start from when it is new to them)
  unsigned l12v[PER];            // the pair's read lengths
  {
    int v0[PER], v1[PER];
    int code[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
      v0[j] = v1[j] = 0; code[j] = -1; l12v[j] = 0; r8[0][j] = r8[1][j] = ~0ull;
      if (!head[j]) continue;
      if (slot[j] < a.n0) code[j] = a.len_code[slot[j]]; else l12v[j] = a.len12[slot[j] - a.n0];
      if (dj[j] >= 0) { v0[j] = a.dl_spill[dj[j]]; v1[j] = a.dl_rec[1][4 * (size_t)dj[j]].w; }
      else if (slot[j] < a.n0) { r8[0][j] = a.rec8[0][slot[j]]; r8[1][j] = a.rec8[1][slot[j]]; v0[j] = r8[0][j] != ~0ull ? 1 : 0; v1[j] = r8[1][j] != ~0ull ? 1 : 0; }
      else {
        const int4 f0 = a.first[0][slot[j] - a.n0], f1 = a.first[1][slot[j] - a.n0];
        v0[j] = f0.x < 0 ? 0 : 1 + (int)((unsigned)f0.z >> 9);
        v1[j] = f1.x < 0 ? 0 : 1 + (int)((unsigned)f1.z >> 9);
      }
    }
#pragma unroll
    for (int j = 0; j < PER; j++) {
      if (!head[j]) continue;
      if (dj[j] >= 0) {
        sp_old[j] = v0[j];
        if (sp_old[j] >= 0) { c0[j] = a.sp_rng[0][sp_old[j]].y; c1[j] = a.sp_rng[1][sp_old[j]].y; }
        else { c0[j] = v1[j] & 0xff; c1[j] = (v1[j] >> 8) & 0xff; }
      } else { c0[j] = v0[j]; c1[j] = v1[j]; }
      if (code[j] >= 0) l12v[j] = a.len_combo[code[j]];
    }
  }
  DL_STAMP(3);
  int my[4] = {0, 0, 0, 0};  // new pairs, new spill entries, spill records of mate 0 / 1
#pragma unroll
  for (int j = 0; j < PER; j++) {
    if (!head[j]) continue;
    const bool lng = c0[j] + add0[j] > 4 || c1[j] + add1[j] > 4;
    my[0] += dj[j] < 0;
    my[1] += lng && sp_old[j] < 0;
    if (lng) { my[2] += c0[j] + add0[j]; my[3] += c1[j] + add1[j]; }
  }
  int excl[4];
  for (int v = 0; v < 4; v++) {
    int incl = my[v];
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d, 64); if (lane >= d) incl += u; }
    if (lane == 63) sc[v][wave] = incl;
    excl[v] = incl - my[v];
  }
  __syncthreads();
  if (tid < 4) { int s = 0; for (int w = 0; w < nwav; w++) { const int t = sc[tid][w]; sc[tid][w] = s; s += t; } tot[tid] = s; }
  __syncthreads();
  for (int v = 0; v < 4; v++) excl[v] += sc[v][wave];
  if (MODE == 1) {  // what this block will hand out; block 0 also notes the counters' present values for the apply launch
    if (tid < 4) { a.blk_tot[4 * blockIdx.x + tid] = tot[tid]; if (blockIdx.x == 0) a.blk_tot[4 * kDlBins + tid] = a.state[tid]; }
    return;
  }
  int all4[4] = {tot[0], tot[1], tot[2], tot[3]};
  if (MODE == 2) {
    __shared__ int before_sh[4], all_sh[4];
    if (tid < 4) { int bf = 0, al = 0; for (int b = 0; b < kDlBins; b++) { const int t = a.blk_tot[4 * b + tid]; al += t; if (b < (int)blockIdx.x) bf += t; } before_sh[tid] = bf; all_sh[tid] = al; }
    __syncthreads();
    for (int v = 0; v < 4; v++) { excl[v] += before_sh[v]; all4[v] = all_sh[v]; }
  }
  DL_STAMP(4);
  const bool overflow = mb_bad || nd0 + all4[0] > a.cap_pairs || ns0 + all4[1] > a.cap_spill || top0 + all4[2] > a.cap_sprec || top1 + all4[3] > a.cap_sprec;
  if (!overflow) {
    int at[4] = {nd0 + excl[0], ns0 + excl[1], top0 + excl[2], top1 + excl[3]};
#pragma unroll
    for (int j = 0; j < PER; j++) {
      if (!head[j]) continue;
      const int p = p_lo + j;
      const int cnt_old[2] = {c0[j], c1[j]}, cnt_add[2] = {add0[j], add1[j]};
      const bool lng = c0[j] + add0[j] > 4 || c1[j] + add1[j] > 4;
      const bool fresh = dj[j] < 0;
      const int sl = slot[j];
      int d = dj[j];
      if (fresh) d = at[0]++;
      int sp = sp_old[j];
      if (lng && sp < 0) sp = at[1]++;
      const unsigned l12 = l12v[j];
      // the present lists, requested together: up to four records per mate at the fixed stride / one compact record
      int4 reg[2][4];
      if (!fresh && sp_old[j] < 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) { reg[0][k] = a.dl_rec[0][4 * (size_t)d + k]; reg[1][k] = a.dl_rec[1][4 * (size_t)d + k]; }
      } else if (fresh && sl < a.n0) {
        const unsigned long long r0 = r8[0][j], r1 = r8[1][j];
        reg[0][0] = make_int4((int)(r0 & 0xffffff), (int)((r0 >> 24) & 0xfffffff), (int)((r0 >> 52) & 63) | ((int)((r0 >> 58) & 1) << 8), 0);
        reg[1][0] = make_int4((int)(r1 & 0xffffff), (int)((r1 >> 24) & 0xfffffff), (int)((r1 >> 52) & 63) | ((int)((r1 >> 58) & 1) << 8), 0);
      }
      int q = p;  // the pair's new records: mate 0's, then mate 1's, each in (window id, position) order
      for (int mt = 0; mt < 2; mt++) {
        DlOld o;
        o.n = cnt_old[mt];
        o.kind = 2;
        if (!fresh && sp_old[j] >= 0) { o.kind = 0; o.p = a.sp_rec[mt] + a.sp_rng[mt][sp_old[j]].x; }
        else if (fresh && sl >= a.n0) { o.kind = 1; o.first = a.first[mt][sl - a.n0]; o.p = a.extra[mt]; }
        auto old_get = [&](int k) -> int4 {
          if (o.kind == 2) { int4 r = k == 0 ? reg[mt][0] : k == 1 ? reg[mt][1] : k == 2 ? reg[mt][2] : reg[mt][3]; r.z &= 0x1ff; r.w = 0; return r; }
          return dl_old_get(o, k);
        };
        const int n_new = cnt_old[mt] + cnt_add[mt];
        int4* out = lng ? a.sp_rec[mt] + at[2 + mt] : a.dl_rec[mt] + 4 * (size_t)d;
        int io = 0, w = 0;
        int4 nr = make_int4(0, 0, 0, 0);
        bool have_new = false;
        auto next_new = [&]() {
          have_new = false;
          if (q < q_end[j] && (int)((keys[q] >> 32) & 1ull) == mt) {
            const int pp = (int)(unsigned)keys[q];
            int4 r;
            if (kCache) r = rec_lds[pp];
            else {
              const DlWin wn = dl_window_of(a, sw, pp);
              r = a.pool[mt][wn.first + (pp - wn.start)];
            }
            nr = make_int4(r.x, r.y, r.z & 0x1ff, 0);
            have_new = true;
            q++;
          }
        };
        next_new();
        int4 head0 = make_int4(-1, 0, 0, 0);
        while (w < n_new) {
          int4 put;
          if (io < o.n) {
            const int4 ov = old_get(io);
            if (!have_new || dl_before_eq(ov, nr)) { put = ov; io++; }
            else { put = nr; next_new(); }
          } else { put = nr; next_new(); }
          if (w == 0) head0 = put;
          if (lng || w > 0) out[w] = put;  // (the fixed stride's first word is written below, with its spare word)
          w++;
        }
        if (lng) {
          a.sp_rng[mt][sp] = make_int2(at[2 + mt], n_new);
          at[2 + mt] += n_new;
          for (int k = 0; k < 4; k++) a.dl_rec[mt][4 * (size_t)d + k] = make_int4(-1, 0, 0, k == 0 ? (mt == 0 ? (int)l12 : 0) : 0);
        } else {
          for (int k = n_new; k < 4; k++) if (k > 0) out[k] = make_int4(-1, 0, 0, 0);
          head0.w = mt == 0 ? (int)l12 : ((c0[j] + add0[j]) | ((c1[j] + add1[j]) << 8));
          out[0] = head0;
        }
      }
      a.dl_slot[d] = sl;
      a.dl_spill[d] = lng ? sp : -1;
      if (lng) a.sp_slot[sp] = sl;
      if (fresh) {
        a.dirty_of_slot[sl] = d;
        // the tables' "this pair lives on the delta lists now" marks (kDirty8 / kDirtyWid)
        if (sl < a.n0) a.rec8[0][sl] = ~0ull - 1;
        else {
          if (sl < a.n01) a.inl0[(size_t)2 * (sl - a.n0)].x = -2;
          else if (sl < a.n_main) a.inl0[(size_t)2 * (a.n01 - a.n0) + (size_t)4 * (sl - a.n01)].x = -2;
          a.first[0][sl - a.n0].x = -2;
        }
      }
    }
  }
  __syncthreads();
  DL_STAMP(5);
  if (MODE == 2 && tid == 0) a.bin_count[blockIdx.x] = 0;  // (for the next multi-block launch)
  if (tid == 0 && (MODE == 0 || blockIdx.x == 0)) {  // (multi-block: nobody reads the counters in the apply launch)
    int st[kDsInts];
    if (MODE == 2) n_left_out = a.blk_tot[4 * kDlBins + 4];
    st[kDsDirty] = overflow ? nd0 : nd0 + all4[0]; st[kDsSpill] = overflow ? ns0 : ns0 + all4[1];
    st[kDsTop0] = overflow ? top0 : top0 + all4[2]; st[kDsTop1] = overflow ? top1 : top1 + all4[3];
    st[kDsOverflow] = a.state[kDsOverflow] | (overflow ? 1 : 0); st[kDsSeq] = a.seq; st[6] = a.state[6] + n_left_out; st[7] = 0;
    for (int k = 0; k < kDsInts; k++) a.state[k] = st[k];
    if (a.host_state) {
      for (int k = 0; k < kDsInts; k++) if (k != kDsSeq) __hip_atomic_store(&a.host_state[k], st[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&a.host_state[kDsSeq], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  DL_STAMP(6);
#undef DL_STAMP
}

// multi-block launches, first dispatch: a lane per record -- left out or keyed as in the one-block kernel -- and the key
// appended to the bin its pair's slot hashes to (the order inside a bin is whatever the atomics make it: the bin is sorted
// by the block that takes it). Resets what the launch's other dispatches accumulate.
__global__ __launch_bounds__(256) void delta_mb_keys_kernel(DlArgs a) {
  __shared__ DlWin sw[kDlMaxWins];
  if (threadIdx.x < kDlMaxWins) sw[threadIdx.x] = a.w[(int)threadIdx.x < a.n_wins && !a.wlist ? threadIdx.x : 0];
  __syncthreads();
  int left_out = 0;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < a.n_total; p += gridDim.x * 256) {
    const DlWin wn = dl_window_of(a, sw, p);
    const int mate = wn.mate;
    const int4 r = a.pool[mate][wn.first + (p - wn.start)];
    int lo = 0, hi = wn.dom_count;
    const int4* base = a.pool[mate] + wn.dom_first;
    while (lo < hi) {  // sixteen-way lower bound over the dominating window's records
      const int span = hi - lo;
      int4 pv[15];
#pragma unroll
      for (int q = 0; q < 15; q++) pv[q] = base[lo + (int)(((long long)span * (q + 1)) >> 4)];
      int nlo = lo, nhi = hi;
#pragma unroll
      for (int q = 0; q < 15; q++) {
        const int at = lo + (int)(((long long)span * (q + 1)) >> 4);
        if (tb_rec_before(pv[q], r.y, r.w)) nlo = max(nlo, at + 1); else nhi = min(nhi, at);
      }
      lo = nlo; hi = nhi;
    }
    bool drop = false;
    if (wn.dom_count > 0 && lo < wn.dom_count) { const int4 at = base[lo]; drop = at.y == r.y && at.w == r.w; }
    if (drop) { left_out++; continue; }
    const unsigned slot = (unsigned)a.slot_of_read[r.w];
    const unsigned bin = (slot * 2654435761u) >> 27;  // (kDlBins = 32)
    const int at = atomicAdd(&a.bin_count[bin], 1);
    if (at < kDlBinCap) a.bins[(size_t)bin * kDlBinCap + at] = ((unsigned long long)slot << 33) | ((unsigned long long)mate << 32) | (unsigned)p;
    else a.blk_tot[4 * kDlBins + 5] = 1;
  }
  for (int off = 32; off > 0; off >>= 1) left_out += __shfl_down(left_out, off, 64);
  if ((threadIdx.x & 63) == 0 && left_out) atomicAdd(&a.blk_tot[4 * kDlBins + 4], left_out);
}
__global__ void delta_mb_begin_kernel(int* blk_tot) { if (threadIdx.x < 2) blk_tot[4 * kDlBins + 4 + threadIdx.x] = 0; }

// the delta store back to empty (a table build took the lists in): counters only -- the marks sit in the OLD tables
__global__ void delta_reset_kernel(int* state, int* host_state, int seq) {
  if (threadIdx.x == 0) {
    for (int k = 0; k < kDsInts; k++) state[k] = k == kDsSeq ? seq : 0;
    if (host_state) { for (int k = 0; k < kDsInts; k++) host_state[k] = k == kDsSeq ? seq : 0; }
  }
}

}  // namespace gaml
