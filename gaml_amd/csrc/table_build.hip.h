// table_build.hip.h -- the read-major record tables of a paired set, built ON THE DEVICE from the window-major record
// pool (the device counterpart of the reference's per-call hash maps, GetPositionsOnlyPath graph.cc:535-598, done once per
// build instead of once per call; the alignment cache it starts from is graph.cc:911-922's aligment_cache_).
//
//   pool[mate]   int4 {window id, position in window, edit | orient << 8, read}: a window's records are contiguous, ordered by
//                (position, read) -- the order of the reference's per-window set (graph.h:227-230, graph.cc:841, 895-897)
//
// One build = a chain of dispatches on one stream, no host round trip in between (counts the later kernels need are read
// from device memory, grids are sized by upper bounds):
//   tb_keys_kernel      a lane per record of an ACTIVE window: key = its read, or "left out" when the record can never survive
//                       the overwrite rule (a junction record that the first node's own window also holds: graph.cc:563-592)
//   rs_sort             stable by read: a read's records end up contiguous, in (window id, position) order
//   tb_segments_kernel  where each read's run starts and ends
//   tb_class_kernel     per pair: record counts, the compact form of a single record, the class, the static memo index
//   tb_pairkey_kernel   (class, window of mate 1, window of mate 2) as a sort key; class counts
//   rs_sort             stable: the device order of the pairs (ties keep read order)
//   tb_compact_kernel   the compact class's tables, the other classes' lengths and list sizes
//   tb_scan_*           where each pair's further records go
//   tb_fill16_kernel    16-byte tables + inline copies for the register classes
// Every step is a deterministic function of the pool and the window list: equal inputs give equal tables, bit for bit (the
// development build checks them against the host restatement build_pair_tables).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "radix_sort.hip.h"

namespace gaml {

struct TbWin { int first, count, wid, dom_first, dom_count, astart; };  // dom: the window whose records overwrite this one's (count 0: none)

// device counters of one build (ints): what the host reads back when the build is done
enum { kTbN3 = 0, kTbClass0 = 1 /* .. 4 */, kTbN0a = 5, kTbExtras0 = 6, kTbExtras1 = 7, kTbDropped0 = 8, kTbDropped1 = 9, kTbInts = 16 };

constexpr unsigned long long kTbNoRec = ~0ull;        // the mate has no record (= kNoRec8)
constexpr unsigned long long kTbNoFit = ~0ull - 1;    // its first record does not fit the 8-byte form

__device__ __forceinline__ bool tb_rec_before(const int4& x, int pos, int read) { return x.y != pos ? x.y < pos : x.w < read; }

// is (pos, read) among the records [first, first + count) (ordered by (position, read))?
__device__ __forceinline__ bool tb_holds(const int4* pool, int first, int count, int pos, int read) {
  int lo = 0, hi = count;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (tb_rec_before(pool[first + mid], pos, read)) lo = mid + 1; else hi = mid;
  }
  if (lo >= count) return false;
  const int4 r = pool[first + lo];
  return r.y == pos && r.w == read;
}

__global__ __launch_bounds__(256) void tb_keys_kernel(const int4* pool, const TbWin* wins, int n_wins, int total, int n_reads, rs_u64* keys, unsigned* vals, int* dropped) {
  int mine = 0;
  for (int g = blockIdx.x * 256 + threadIdx.x; g < total; g += gridDim.x * 256) {
    int lo = 0, hi = n_wins - 1;  // the last window whose astart <= g
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (wins[mid].astart <= g) lo = mid; else hi = mid - 1; }
    const TbWin w = wins[lo];
    const int at = w.first + (g - w.astart);
    const int4 r = pool[at];
    const bool drop = w.dom_count > 0 && tb_holds(pool, w.dom_first, w.dom_count, r.y, r.w);
    keys[g] = drop ? (rs_u64)n_reads : (rs_u64)(unsigned)r.w;
    vals[g] = (unsigned)at;
    mine += drop;
  }
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(dropped, mine);
}

__global__ __launch_bounds__(256) void tb_segments_kernel(const rs_u64* keys, int total, int n_reads, int* rstart, int* rend) {
  for (int g = blockIdx.x * 256 + threadIdx.x; g < total; g += gridDim.x * 256) {
    const rs_u64 k = keys[g];
    if (k >= (rs_u64)n_reads) continue;
    if (g == 0 || keys[g - 1] != k) rstart[k] = g;
    if (g == total - 1 || keys[g + 1] != k) rend[k] = g + 1;
  }
}

struct TbClassArgs {
  const int4* pool[2];
  const unsigned* vals[2];      // pool index of the g-th record in read order
  const int* rstart[2];
  const int* rend[2];
  const int* lens[2];
  const short* lcode;           // length-combination code per pair, -1: beyond the 256 the compact class knows
  const int* peer0;             // per window of mate 1: the window of mate 2 with the same node walk, or < 0
  int n, ins_n, memo_codes, memo_fits;
  unsigned long long* one[2];
  unsigned char* cl;
  int* sidx;
  int* cnt;
};

__device__ __forceinline__ bool tb_rec8_fits(int wid, int pos, int edit) { return wid >= 0 && wid < (1 << 24) - 1 && pos >= 0 && pos < (1 << 28) && edit >= 0 && edit < 64; }
__device__ __forceinline__ unsigned long long tb_rec8_pack(int wid, int pos, int edit, int orient) {
  return (unsigned long long)(unsigned)wid | ((unsigned long long)(unsigned)pos << 24) | ((unsigned long long)(unsigned)edit << 52) | ((unsigned long long)(orient & 1) << 58);
}

// internal classes: 0 = compact with a static memo index, 1 = the rest of the compact class, 2..4 = up to 2 / up to 4 / more records
__global__ __launch_bounds__(256) void tb_class_kernel(TbClassArgs a) {
  int n3 = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n; i += gridDim.x * 256) {
    int k[2];
    unsigned long long one[2];
    for (int mt = 0; mt < 2; mt++) {
      const int b = a.rstart[mt][i], e = a.rend[mt][i];
      k[mt] = e - b;
      one[mt] = kTbNoRec;
      if (k[mt] > 0) {
        const int4 r = a.pool[mt][a.vals[mt][b]];
        const int edit = r.z & 0xff, orient = (r.z >> 8) & 1;
        one[mt] = tb_rec8_fits(r.x, r.y, edit) ? tb_rec8_pack(r.x, r.y, edit, orient) : kTbNoFit;
      }
      a.one[mt][i] = one[mt];
    }
    const int lc = a.lcode[i];
    const int m = max(k[0], k[1]);
    int cls, sidx = -1;
    if (m <= 1 && lc >= 0 && one[0] != kTbNoFit && one[1] != kTbNoFit) {
      // the memo index of a pair whose two records sit in the same window (same node walk): orientation rule and insert
      // distance as the scorers apply them per call (graph.cc:1864-1876), on window positions
      if (a.memo_fits) {
        if (one[0] == kTbNoRec || one[1] == kTbNoRec) sidx = -2;  // kStaticZero: a mate without alignment never scores
        else if (lc < a.memo_codes) {
          const int w1 = (int)(one[0] & 0xffffff), w2 = (int)(one[1] & 0xffffff);
          const int p1 = (int)((one[0] >> 24) & 0xfffffff), p2 = (int)((one[1] >> 24) & 0xfffffff);
          const int e1 = (int)((one[0] >> 52) & 63), e2 = (int)((one[1] >> 52) & 63);
          const int o1 = (int)((one[0] >> 58) & 1), o2 = (int)((one[1] >> 58) & 1);
          if (a.peer0[w1] == w2 && o1 != o2 && e1 < 7 && e2 < 7) {
            const bool fwd = p1 < p2;
            if (o1 == (fwd ? 0 : 1)) {
              const int dist = fwd ? p2 - p1 + a.lens[1][i] : p1 - p2 + a.lens[0][i];
              if (dist >= 0 && dist < a.ins_n) sidx = ((lc * 7 + e1) * 7 + e2) * a.ins_n + dist;
            }
          }
        }
      }
      cls = sidx != -1 ? 0 : 1;
    } else cls = m <= 2 ? 2 : m <= 4 ? 3 : 4;
    a.cl[i] = (unsigned char)cls;
    a.sidx[i] = sidx;
    n3 += cls == 3;
  }
  for (int off = 32; off > 0; off >>= 1) n3 += __shfl_down(n3, off, 64);
  if ((threadIdx.x & 63) == 0 && n3) atomicAdd(&a.cnt[kTbN3], n3);
}

// sort key (class, window of mate 1, window of mate 2); "no record / does not fit" sorts behind every window
__global__ __launch_bounds__(256) void tb_pairkey_kernel(const unsigned long long* one0, const unsigned long long* one1, unsigned char* cl, int n, int fold_below,
                                                        unsigned none1, unsigned none2, int bits1, int bits2, rs_u64* keys, unsigned* reads, int* cnt) {
  __shared__ int sh[5];
  if (threadIdx.x < 5) sh[threadIdx.x] = 0;
  __syncthreads();
  const int n3 = cnt[kTbN3];
  const bool fold = n3 > 0 && n3 <= fold_below;  // a handful of pairs with 3-4 records per mate go one wave per pair
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    int c = cl[i];
    if (fold && c == 3) { c = 4; cl[i] = 4; }
    const unsigned w1 = one0[i] >= kTbNoFit ? none1 : (unsigned)(one0[i] & 0xffffff);
    const unsigned w2 = one1[i] >= kTbNoFit ? none2 : (unsigned)(one1[i] & 0xffffff);
    keys[i] = ((rs_u64)c << (bits1 + bits2)) | ((rs_u64)w1 << bits2) | (rs_u64)w2;
    reads[i] = (unsigned)i;
    for (int q = 0; q < 5; q++) {  // (a count per wave and class: one LDS atomic per wave, not per lane)
      const unsigned long long b = __ballot(c == q);
      if ((threadIdx.x & 63) == 0 && b) atomicAdd(&sh[q], (int)__popcll(b));
    }
  }
  __syncthreads();
  if (threadIdx.x < 5 && sh[threadIdx.x]) atomicAdd(&cnt[threadIdx.x == 0 ? kTbN0a : kTbClass0 + threadIdx.x - 1], sh[threadIdx.x]);
  if (threadIdx.x == 0 && sh[0]) atomicAdd(&cnt[kTbClass0], sh[0]);  // (internal class 0 is part of the compact class)
}

struct TbCompactArgs {
  const unsigned* order;        // read of each slot
  const unsigned long long* one[2];
  const short* lcode;
  const int* sidx;
  const int* lens[2];
  const int* rstart[2];
  const int* rend[2];
  const int* cnt;
  int n;
  int* slot_of_read;
  int* dirty_of_slot;
  unsigned long long* rec8[2];
  unsigned char* len_code;
  int* static_idx;
  unsigned* len12;
  int* more[2];                 // per 16-byte slot: records beyond the first
};

__global__ __launch_bounds__(256) void tb_compact_kernel(TbCompactArgs a) {
  const int n0 = a.cnt[kTbClass0], n0a = a.cnt[kTbN0a];
  for (int s = blockIdx.x * 256 + threadIdx.x; s < a.n; s += gridDim.x * 256) {
    const int r = (int)a.order[s];
    a.slot_of_read[r] = s;
    a.dirty_of_slot[s] = -1;
    if (s < n0) {
      a.rec8[0][s] = a.one[0][r];
      a.rec8[1][s] = a.one[1][r];
      a.len_code[s] = (unsigned char)a.lcode[r];
      if (s < n0a) a.static_idx[s] = a.sidx[r];
    } else {
      const int t = s - n0;
      a.len12[t] = (unsigned)a.lens[0][r] | ((unsigned)a.lens[1][r] << 16);
      for (int mt = 0; mt < 2; mt++) { const int k = a.rend[mt][r] - a.rstart[mt][r]; a.more[mt][t] = k > 1 ? k - 1 : 0; }
    }
  }
}

// ---- exclusive prefix sum of n ints (n itself read on the device: *n_ptr - *n_sub), three small dispatches --------------------
constexpr int kTbScanTile = 4096;
__device__ __forceinline__ int tb_scan_n(const int* cnt, int n) { return n - cnt[kTbClass0]; }
__global__ __launch_bounds__(256) void tb_scan_tiles_kernel(const int* in, const int* cnt, int n_all, int* tile_sum) {
  const int n = tb_scan_n(cnt, n_all);
  __shared__ int sh[4];
  const int lo = blockIdx.x * kTbScanTile;
  int s = 0;
  for (int i = lo + threadIdx.x; i < min(n, lo + kTbScanTile); i += 256) s += in[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(1024) void tb_scan_top_kernel(int* tile_sum, int n_tiles, int* total_out) {
  __shared__ int wave_sum[16];
  __shared__ int carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int c0 = 0; c0 < n_tiles; c0 += 1024) {
    const int i = c0 + threadIdx.x;
    const int v = i < n_tiles ? tile_sum[i] : 0;
    int incl = v;
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d, 64); if (lane >= d) incl += u; }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wave; w++) before += wave_sum[w];
    if (i < n_tiles) tile_sum[i] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = carry;
}
__global__ __launch_bounds__(256) void tb_scan_apply_kernel(const int* in, const int* cnt, int n_all, const int* tile_before, int* out) {
  const int n = tb_scan_n(cnt, n_all);
  __shared__ int wave_sum[4];
  __shared__ int carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lo = blockIdx.x * kTbScanTile;
  if (lo >= n) return;
  if (threadIdx.x == 0) carry = tile_before[blockIdx.x];
  __syncthreads();
  for (int c0 = lo; c0 < min(n, lo + kTbScanTile); c0 += 256) {
    const int i = c0 + threadIdx.x;
    const int v = i < n ? in[i] : 0;
    int incl = v;
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d, 64); if (lane >= d) incl += u; }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wave; w++) before += wave_sum[w];
    if (i < n) out[i] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry = before + incl;
    __syncthreads();
  }
}

struct TbFillArgs {
  const int4* pool;
  const unsigned* vals;
  const int* rstart;
  const int* rend;
  const unsigned* order;
  const int* start;             // exclusive prefix of `more`
  const int* cnt;
  int n;
  int4* first;
  int4* extra;
  int4* inl;
};

// 16-byte tables of the slots behind the compact class: first[t] = the pair's first record {wid, pos, edit | orient << 8 |
// (count - 1) << 9, where its further records start in extra[]}; inline copies (2 / 4 per pair) for the register classes
__global__ __launch_bounds__(256) void tb_fill16_kernel(TbFillArgs a) {
  const int n0 = a.cnt[kTbClass0], n1 = a.cnt[kTbClass0 + 1], n2 = a.cnt[kTbClass0 + 2];
  const int n16 = a.n - n0;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < n16; t += gridDim.x * 256) {
    const int r = (int)a.order[n0 + t];
    const int b = a.rstart[r], k = a.rend[r] - b;
    const int st = a.start[t];
    int4* dst = t < n1 ? a.inl + (size_t)2 * t : (t < n1 + n2 ? a.inl + (size_t)2 * n1 + (size_t)4 * (t - n1) : nullptr);
    const int room = t < n1 ? 2 : (t < n1 + n2 ? 4 : 0);
    int4 f = make_int4(-1, 0, 0, 0);
    for (int q = 0; q < k; q++) {
      const int4 p = a.pool[a.vals[b + q]];
      int4 rq = make_int4(p.x, p.y, p.z & 0x1ff, 0);
      if (q == 0) { f = rq; f.z |= (k - 1) << 9; f.w = st; rq.w = st; }
      else a.extra[st + q - 1] = rq;
      if (q < room) dst[q] = rq;
    }
    for (int q = k; q < room; q++) dst[q] = make_int4(-1, 0, 0, 0);
    a.first[t] = f;
  }
}

// the static pairs' memo entries, streamed with their records by the scoring launch (n0a read on the device)
__global__ __launch_bounds__(256) void tb_static_values_kernel(const int* static_idx, const int* cnt, const double2* memo, double2* static_val) {
  const int n0a = cnt[kTbN0a];
  for (int s = blockIdx.x * 256 + threadIdx.x; s < n0a; s += gridDim.x * 256) {
    const int ix = static_idx[s];
    static_val[s] = ix >= 0 ? memo[ix] : make_double2(0.0, 0.0);
  }
}

}  // namespace gaml
