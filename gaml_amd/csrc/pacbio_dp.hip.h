// pacbio_dp.hip.h -- banded sum-over-alignments DP of a PacBio read against a path string
// (PacbioReadSet::AligmentProbability, graph.cc:2175-2297) for gfx950.
//
// The reference walks a sorted cell list (row = path base, column = read base) and for every
// cell adds, in logdouble arithmetic, the three predecessors (diagonal, up, left) that belong to
// the cell set; column 0 is a free start, the result is the sum of column |read| over all rows.
// The cell set is one column interval per row (host: pacbio_dp_band), typically 5-9 cells wide
// and a few thousand rows long: far too narrow for a wavefront per alignment and a long serial
// chain per thread.  Mapping used here:
//   * G lanes (8 or 16) own one alignment; a 64-wide wavefront carries 64/G alignments.
//   * a row is processed in chunks of G columns.  The diagonal and up terms of all cells of a
//     chunk are independent (2 log-sum-exp steps in parallel); the left dependence
//     V[c] = A[c] (+) V[c-1]*g is a linear recurrence with the constant gap factor g and is
//     solved by a log2(G)-step scan across the lanes (V[c] (+)= V[c-o]*g^o for o = 1,2,4,..).
//     The scan changes the association order of the logdouble additions relative to the
//     reference's left-to-right chain, so results agree to rounding (tests: 1e-9 relative on the
//     log probability), not bit for bit.
//   * the previous and current row live in a per-alignment scratch of 2*max_width doubles in
//     global memory (L2 resident: a row is rewritten every iteration); rows can be as wide as the
//     read (soft-clipped ends), so LDS cannot hold them in general.
// Compute bound: ~(2 + log2 G) dependent exp+log1p pairs per row; HBM traffic is negligible.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace gaml {

struct DpJob {
  int64_t read_off;     // first base of the read in DpArgs::reads
  int64_t band_off;     // first row of this job in DpArgs::lo / hi
  int64_t scratch_off;  // doubles; 2*max_width are reserved
  int32_t read_len;
  int32_t posstart;     // path position of DP row 1 (PacbioAligmentData::posstart)
  int32_t row0;         // first DP row (may be negative)
  int32_t n_rows;
  int32_t max_width;
  int32_t pad;
};

struct DpArgs {
  const unsigned char* path;  // "path + '\n' + reverse complement" (graph.cc:2687-2688)
  int32_t path_len;
  const unsigned char* reads;
  const DpJob* jobs;
  const int32_t* lo;
  const int32_t* hi;
  double* scratch;
  double* out;                // log probability per job
  int32_t n_jobs;
  double log_match, log_mismatch;
};

__device__ __forceinline__ double dp_lse2(double a, double b) {  // logdouble operator+ (logdouble.hpp:37-47)
  const double ninf = -__builtin_huge_val();
  if (a == ninf) return b;
  if (b == ninf) return a;
  const double hi = fmax(a, b), lo = fmin(a, b);
  return hi + log1p(exp(lo - hi));
}

constexpr unsigned char kDpSeparator = '\n';  // kContigSeparator graph.cc:30
constexpr unsigned char kDpGap = '-';

template <int G>
__global__ __launch_bounds__(256) void pacbio_dp_kernel(DpArgs a) {
  const int job = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) / G);
  const int j = (int)(threadIdx.x % G);
  if (job >= a.n_jobs) return;  // whole groups leave together
  const DpJob jb = a.jobs[job];
  const double ninf = -__builtin_huge_val();
  const unsigned char* rd = a.reads + jb.read_off;
  const int32_t* LO = a.lo + jb.band_off;
  const int32_t* HI = a.hi + jb.band_off;
  double* prev = a.scratch + jb.scratch_off;
  double* cur = prev + jb.max_width;
  const int n = jb.read_len;
  const double g = a.log_mismatch;  // MatchProbability('-', base): a read base is never '-' or the separator
  double ret = ninf;
  int plo = 0, phi = -1;  // column interval of the previous row (empty before the first row)
  for (int ri = 0; ri < jb.n_rows; ri++) {
    const int r = jb.row0 + ri;
    const int lo = LO[ri], hi = HI[ri];
    const int gi = r + jb.posstart - 1;  // path base of this row (graph.cc:2252)
    const bool row_ok = gi >= 0 && gi < a.path_len;
    const unsigned char pc = row_ok ? a.path[gi] : (unsigned char)0;
    // MatchProbability (graph.h:555-564) of the path base against a gap
    const double up_w = pc == kDpSeparator ? ninf : (pc == kDpGap ? a.log_match : a.log_mismatch);
    double seed = ninf;  // value left of the chunk
    for (int base = lo; base <= hi; base += G) {
      const int c = base + j;
      const bool in = c <= hi;
      const bool comp = in && row_ok && c >= 1 && c <= n;
      double v = ninf;
      if (comp) {
        const unsigned char rc = rd[c - 1];
        if (c - 1 >= plo && c - 1 <= phi) {
          const double w = pc == kDpSeparator ? ninf : (pc == rc ? a.log_match : a.log_mismatch);
          v = prev[c - 1 - plo] + w;  // (-inf) + finite stays -inf; both -inf stays -inf
        }
        if (c >= plo && c <= phi) v = dp_lse2(v, prev[c - plo] + up_w);
      } else if (in && c == 0) {
        v = 0.0;  // free start in column 0 (graph.cc:2238-2243)
      }
      if (row_ok) {
        if (j == 0) v = dp_lse2(v, seed + g);
        double step = g;
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
          const double t = __shfl_up(v, o, G);
          if (j >= o) v = dp_lse2(v, t + step);
          step += step;
        }
        seed = __shfl(v, G - 1, G);
        if (n >= 1 && n >= base && n < base + G && n <= hi)  // the cell in column |read| (graph.cc:2279-2281)
          ret = dp_lse2(ret, __shfl(v, n - base, G));
      }
      if (in) cur[c - lo] = c == 0 ? 0.0 : (comp ? v : ninf);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    double* t = prev; prev = cur; cur = t;
    plo = lo; phi = hi;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (j == 0) a.out[job] = ret;
}

}  // namespace gaml
