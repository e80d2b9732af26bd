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

constexpr int kDlThreads = 1024, kDlMaxRecs = 8192, kDlMaxWins = 32;
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
  DlWin w[kDlMaxWins];
};

// where a pair's present list of one mate comes from
struct DlOld { int kind, n; const int4* p; int4 first; };  // kind 0: p[k] (delta store / spill area); 1: tables (first + extra)
__device__ __forceinline__ int4 dl_old_get(const DlOld& o, int k) {
  int4 r = o.kind == 0 ? o.p[k] : (k == 0 ? o.first : o.p[o.first.w + k - 1]);
  r.z &= 0x1ff; r.w = 0;
  return r;
}
__device__ __forceinline__ bool dl_before_eq(const int4& x, const int4& y) { return x.x != y.x ? x.x < y.x : x.y <= y.y; }  // x goes first (an old record before an equal new one: upper_bound)

__global__ __launch_bounds__(kDlThreads) void delta_apply_kernel(DlArgs a) {
  __shared__ unsigned long long keys[kDlMaxRecs];
  __shared__ int sc[4][kDlThreads / 64];
  __shared__ int tot[4];
  __shared__ int n_left_out;
  if (threadIdx.x == 0) n_left_out = 0;
  __syncthreads();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int N = 64;
  while (N < a.n_total) N <<= 1;
  // ---- 1. keys
  for (int p = tid; p < N; p += kDlThreads) {
    unsigned long long key = ~0ull;
    if (p < a.n_total) {
      int k = 0;
      while (k + 1 < a.n_wins && a.w[k + 1].start <= p) k++;
      const int mate = a.w[k].mate;
      const int4 r = a.pool[mate][a.w[k].first + (p - a.w[k].start)];
      const bool drop = a.w[k].dom_count > 0 && tb_holds(a.pool[mate], a.w[k].dom_first, a.w[k].dom_count, r.y, r.w);
      if (!drop) key = ((unsigned long long)(unsigned)a.slot_of_read[r.w] << 33) | ((unsigned long long)mate << 32) | (unsigned)p;
      else atomicAdd(&n_left_out, 1);
    }
    keys[p] = key;
  }
  __syncthreads();
  // ---- 2. bitonic sort, ascending
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (N >> 1); t += kDlThreads) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
        const unsigned long long x = keys[i], y = keys[l];
        const bool up = (i & k) == 0;
        if ((x > y) == up) { keys[i] = y; keys[l] = x; }
      }
      __syncthreads();
    }
  }
  // ---- 3. per touched pair. Thread t owns the sorted positions [t * per, (t + 1) * per): numbers are handed out in that order.
  const int per = N > kDlThreads ? N / kDlThreads : 1;
  const int p_lo = tid * per, p_hi = min(N, p_lo + per);
  const int nd0 = a.state[kDsDirty], ns0 = a.state[kDsSpill], top0 = a.state[kDsTop0], top1 = a.state[kDsTop1];
  // what a head finds: counts of the present lists, and what the new ones will be
  auto head_info = [&](int p, int& slot, int& dj, int& sp_old, int& c0, int& c1, int& add0, int& add1, int& q_end) {
    slot = (int)(keys[p] >> 33);
    add0 = add1 = 0;
    int q = p;
    while (q < N && keys[q] != ~0ull && (int)(keys[q] >> 33) == slot) { if ((keys[q] >> 32) & 1ull) add1++; else add0++; q++; }
    q_end = q;
    dj = a.dirty_of_slot[slot];
    sp_old = -1;
    if (dj >= 0) {
      sp_old = a.dl_spill[dj];
      if (sp_old >= 0) { c0 = a.sp_rng[0][sp_old].y; c1 = a.sp_rng[1][sp_old].y; }
      else { const int cc = a.dl_rec[1][4 * (size_t)dj].w; c0 = cc & 0xff; c1 = (cc >> 8) & 0xff; }
    } else if (slot < a.n0) {
      c0 = a.rec8[0][slot] != ~0ull ? 1 : 0;
      c1 = a.rec8[1][slot] != ~0ull ? 1 : 0;
    } else {
      const int4 f0 = a.first[0][slot - a.n0], f1 = a.first[1][slot - a.n0];
      c0 = f0.x < 0 ? 0 : 1 + (int)((unsigned)f0.z >> 9);
      c1 = f1.x < 0 ? 0 : 1 + (int)((unsigned)f1.z >> 9);
    }
  };
  int my[4] = {0, 0, 0, 0};  // new pairs, new spill entries, spill records of mate 0 / 1
  for (int p = p_lo; p < p_hi; p++) {
    if (keys[p] == ~0ull) break;
    if (p > 0 && (keys[p - 1] >> 33) == (keys[p] >> 33)) continue;
    int slot, dj, sp_old, c0, c1, add0, add1, q_end;
    head_info(p, slot, dj, sp_old, c0, c1, add0, add1, q_end);
    const bool lng = c0 + add0 > 4 || c1 + add1 > 4;
    my[0] += dj < 0;
    my[1] += lng && sp_old < 0;
    if (lng) { my[2] += c0 + add0; my[3] += c1 + add1; }
  }
  int excl[4];
  for (int v = 0; v < 4; v++) {
    int incl = my[v];
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d, 64); if (lane >= d) incl += u; }
    if (lane == 63) sc[v][wave] = incl;
    excl[v] = incl - my[v];
  }
  __syncthreads();
  if (tid < 4) { int s = 0; for (int w = 0; w < kDlThreads / 64; w++) { const int t = sc[tid][w]; sc[tid][w] = s; s += t; } tot[tid] = s; }
  __syncthreads();
  for (int v = 0; v < 4; v++) excl[v] += sc[v][wave];
  const bool overflow = nd0 + tot[0] > a.cap_pairs || ns0 + tot[1] > a.cap_spill || top0 + tot[2] > a.cap_sprec || top1 + tot[3] > a.cap_sprec;
  if (!overflow) {
    int at[4] = {nd0 + excl[0], ns0 + excl[1], top0 + excl[2], top1 + excl[3]};
    for (int p = p_lo; p < p_hi; p++) {
      if (keys[p] == ~0ull) break;
      if (p > 0 && (keys[p - 1] >> 33) == (keys[p] >> 33)) continue;
      int slot, dj, sp_old, c0, c1, add0, add1, q_end;
      head_info(p, slot, dj, sp_old, c0, c1, add0, add1, q_end);
      const int cnt_old[2] = {c0, c1}, cnt_add[2] = {add0, add1};
      const bool lng = c0 + add0 > 4 || c1 + add1 > 4;
      const bool fresh = dj < 0;
      if (fresh) dj = at[0]++;
      int sp = sp_old;
      if (lng && sp < 0) sp = at[1]++;
      const unsigned l12 = slot < a.n0 ? a.len_combo[a.len_code[slot]] : a.len12[slot - a.n0];
      int q = p;  // the pair's new records: mate 0's, then mate 1's, each in (window id, position) order
      for (int mt = 0; mt < 2; mt++) {
        DlOld o;
        o.n = cnt_old[mt];
        int4 reg[4];
        if (!fresh && sp_old < 0) {  // at the fixed stride: into registers first (the output may land on the same words)
          for (int k = 0; k < 4; k++) reg[k] = a.dl_rec[mt][4 * (size_t)dj + k];
          o.kind = 2;
        } else if (!fresh) { o.kind = 0; o.p = a.sp_rec[mt] + a.sp_rng[mt][sp_old].x; }
        else if (slot < a.n0) {
          const unsigned long long r = a.rec8[mt][slot];
          reg[0] = make_int4((int)(r & 0xffffff), (int)((r >> 24) & 0xfffffff), (int)((r >> 52) & 63) | ((int)((r >> 58) & 1) << 8), 0);
          o.kind = 2;
        } else { o.kind = 1; o.first = a.first[mt][slot - a.n0]; o.p = a.extra[mt]; }
        auto old_get = [&](int k) -> int4 {
          if (o.kind == 2) { int4 r = k == 0 ? reg[0] : k == 1 ? reg[1] : k == 2 ? reg[2] : reg[3]; r.z &= 0x1ff; r.w = 0; return r; }
          return dl_old_get(o, k);
        };
        const int n_new = cnt_old[mt] + cnt_add[mt];
        int4* out = lng ? a.sp_rec[mt] + at[2 + mt] : a.dl_rec[mt] + 4 * (size_t)dj;
        int io = 0, w = 0;
        int4 nr = make_int4(0, 0, 0, 0);
        bool have_new = false;
        auto next_new = [&]() {
          have_new = false;
          if (q < q_end && (int)((keys[q] >> 32) & 1ull) == mt) {
            const int pp = (int)(unsigned)keys[q];
            int k = 0;
            while (k + 1 < a.n_wins && a.w[k + 1].start <= pp) k++;
            const int4 r = a.pool[mt][a.w[k].first + (pp - a.w[k].start)];
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
          for (int k = 0; k < 4; k++) a.dl_rec[mt][4 * (size_t)dj + k] = make_int4(-1, 0, 0, k == 0 ? (mt == 0 ? (int)l12 : 0) : 0);
        } else {
          for (int k = n_new; k < 4; k++) if (k > 0) out[k] = make_int4(-1, 0, 0, 0);
          head0.w = mt == 0 ? (int)l12 : ((c0 + add0) | ((c1 + add1) << 8));
          out[0] = head0;
        }
      }
      a.dl_slot[dj] = slot;
      a.dl_spill[dj] = lng ? sp : -1;
      if (lng) a.sp_slot[sp] = slot;
      if (fresh) {
        a.dirty_of_slot[slot] = dj;
        // the tables' "this pair lives on the delta lists now" marks (kDirty8 / kDirtyWid)
        if (slot < a.n0) a.rec8[0][slot] = ~0ull - 1;
        else {
          if (slot < a.n01) a.inl0[(size_t)2 * (slot - a.n0)].x = -2;
          else if (slot < a.n_main) a.inl0[(size_t)2 * (a.n01 - a.n0) + (size_t)4 * (slot - a.n01)].x = -2;
          a.first[0][slot - a.n0].x = -2;
        }
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    int st[kDsInts];
    st[kDsDirty] = overflow ? nd0 : nd0 + tot[0]; st[kDsSpill] = overflow ? ns0 : ns0 + tot[1];
    st[kDsTop0] = overflow ? top0 : top0 + tot[2]; st[kDsTop1] = overflow ? top1 : top1 + tot[3];
    st[kDsOverflow] = a.state[kDsOverflow] | (overflow ? 1 : 0); st[kDsSeq] = a.seq; st[6] = a.state[6] + n_left_out; st[7] = 0;
    for (int k = 0; k < kDsInts; k++) a.state[k] = st[k];
    if (a.host_state) {
      for (int k = 0; k < kDsInts; k++) if (k != kDsSeq) __hip_atomic_store(&a.host_state[k], st[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&a.host_state[kDsSeq], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// the delta store back to empty (a table build took the lists in): counters only -- the marks sit in the OLD tables
__global__ void delta_reset_kernel(int* state, int* host_state, int seq) {
  if (threadIdx.x == 0) {
    for (int k = 0; k < kDsInts; k++) state[k] = k == kDsSeq ? seq : 0;
    if (host_state) { for (int k = 0; k < kDsInts; k++) host_state[k] = k == kDsSeq ? seq : 0; }
  }
}

}  // namespace gaml
