// pacbio_sweep.hip.h -- the coverage sweep of CalcScoreForPacbio (graph.cc:3198-3250) on the device.
//
// The reference sorts (position, +1 | begin - end) events of every contig's node intervals, the fixed interval
// (-1000, 2000) and the alignment intervals of all records that clear GetMinReadProb, walks them with a multiset of
// the open intervals' begins, and after every event j adds
//     max(0, min(next event's position, tl - 250, open.empty() ? tl - 250 : int(min(open) + step)) - max(2500, position_j))
// to bad_bases. Every interval has end > begin (an event with value +1 is an opening; closing an interval that is not
// open is undefined in the reference, graph.cc:3229-3231), so closings at a position come before its openings, every
// closing finds its opening, and only the LAST event of a position contributes (for the others the next event is at
// the same position). After all events at position p the open intervals are exactly those with begin <= p < end, and
// with the intervals ordered by begin the one with the smallest begin among them is the first whose running maximum
// of `end` exceeds p (all before it have closed; if it begins after p nothing is open). Hence, without a multiset:
//   1. sort the intervals by (contig, begin); running maximum of (contig, end) over that order (segmented by contig
//      for free: the contig is the key's high word);
//   2. sort all begins and ends by (contig, position);
//   3. one thread per sorted position: skip it when the next one is equal; binary search of the running maxima for
//      the open interval with the smallest begin; the term above; block sums, one atomic add per block.
// Integer arithmetic throughout (int(min + step) truncates a double exactly as the reference's int = int + double):
// bad_bases equals the reference's sweep bit for bit. HBM traffic: 16 B per interval in, ~100 B per interval through
// the two radix sorts -- a few hundred KB per evaluation; the launches' latency is what it costs.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace gaml {

// one occurrence of a cached sub-walk in a path of the evaluation
struct PbOcc {
  int32_t walk;  // cache id: intervals iv[iv_off[walk] .. iv_off[walk + 1])
  int32_t base;  // path position of the sub-walk's first base (graph.cc:2498)
  int32_t path;  // contig number in this evaluation
  int32_t out;   // first slot of its intervals in the evaluation's interval list
};

// (contig, position) as one ascending unsigned key; positions may be negative (-1000)
__device__ __forceinline__ unsigned long long pb_key(int path, int pos) {
  return ((unsigned long long)(unsigned)path << 32) | (unsigned long long)((unsigned)pos ^ 0x80000000u);
}
__device__ __forceinline__ int pb_key_path(unsigned long long k) { return (int)(unsigned)(k >> 32); }
__device__ __forceinline__ int pb_key_pos(unsigned long long k) { return (int)(((unsigned)k) ^ 0x80000000u); }

// the alignment intervals of every occurrence, in path coordinates: {contig, begin, end, 0}. One wavefront per occurrence.
__global__ __launch_bounds__(256) void pacbio_intervals_kernel(const PbOcc* __restrict__ occ, int n_occ, const int* __restrict__ iv_off,
                                                                const int2* __restrict__ iv, int4* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int waves = (gridDim.x * blockDim.x) >> 6;
  for (int o = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; o < n_occ; o += waves) {
    const PbOcc oc = occ[o];
    const int first = iv_off[oc.walk], n = iv_off[oc.walk + 1] - first;
    for (int t = lane; t < n; t += 64) {
      const int2 v = iv[first + t];
      out[oc.out + t] = make_int4(oc.path, oc.base + v.x, oc.base + v.y, 0);
    }
  }
}

__global__ __launch_bounds__(256) void pacbio_sweep_keys_kernel(const int4* __restrict__ iv, int n, unsigned long long* __restrict__ key_begin,
                                                                 unsigned long long* __restrict__ key_end, unsigned long long* __restrict__ pos) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const int4 v = iv[k];
    const unsigned long long b = pb_key(v.x, v.y), e = pb_key(v.x, v.z);
    key_begin[k] = b;
    key_end[k] = e;
    pos[2 * k] = b;
    pos[2 * k + 1] = e;
  }
}

// pos: the 2n begins and ends sorted; key_begin: the n intervals' begins sorted; end_max: running maximum of their
// (contig, end) keys in that order; tl: contig lengths (GetReadProbabilities' total_len, graph.cc:2431)
__global__ __launch_bounds__(256) void pacbio_sweep_kernel(const unsigned long long* __restrict__ pos, int n2,
                                                            const unsigned long long* __restrict__ key_begin,
                                                            const unsigned long long* __restrict__ end_max, int n, const int* __restrict__ tl,
                                                            double step, unsigned long long* __restrict__ bad) {
  __shared__ unsigned long long sh[4];
  unsigned long long mine = 0;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n2; j += gridDim.x * blockDim.x) {
    const unsigned long long here = pos[j];
    const bool has_next = j + 1 < n2;
    const unsigned long long nxt = has_next ? pos[j + 1] : 0ull;
    if (has_next && nxt == here) continue;  // not the last event of its position
    const int q = pb_key_path(here), p = pb_key_pos(here);
    int good = tl[q] - 250;
    // the open interval with the smallest begin: first interval whose running maximum of `end` exceeds p
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (end_max[mid] > here) hi = mid; else lo = mid + 1;
    }
    if (lo < n) {
      const unsigned long long kb = key_begin[lo];
      if (pb_key_path(kb) == q && pb_key_pos(kb) <= p) good = (int)((double)pb_key_pos(kb) + step);  // int = int + double (graph.cc:3236)
    }
    if (has_next && pb_key_path(nxt) == q) good = min(pb_key_pos(nxt), good);
    good = min(good, tl[q] - 250);
    const int from = max(2500, p);
    if (good > from) mine += (unsigned long long)(good - from);
  }
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long total = sh[0] + sh[1] + sh[2] + sh[3];
    if (total) atomicAdd(bad, total);
  }
}

}  // namespace gaml
