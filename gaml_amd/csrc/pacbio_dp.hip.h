// pacbio_dp.hip.h -- banded sum-over-alignments DP of a PacBio read against a path string
// (PacbioReadSet::AligmentProbability, graph.cc:2175-2297) for gfx950.
//
// The reference walks a sorted cell list (row = path base, column = read base) and for every
// cell adds, in logdouble arithmetic, the three predecessors (diagonal, up, left) that belong to
// the cell set; column 0 is a free start, the result is the sum of column |read| over all rows.
// The cell set is one column interval per row: the CIGAR path, the two clip boxes, closed to
// row intervals, widened by 2 and closed again (graph.cc:2183-2221) -- typically 5-9 cells wide
// and a few thousand rows long: far too narrow for a wavefront per alignment, and a long serial
// chain per thread.  Mapping:
//   * G = 16 lanes own one alignment; a 64-wide wavefront carries 4 alignments.
//   * the band is derived on the fly from the run-length CIGAR (a 5-row sliding window of the
//     path's first/last column per row), so the host ships ~4 bytes per CIGAR operation instead
//     of 8 bytes per DP row.
//   * a row is processed in chunks of G-2 columns.  A chunk reads its inputs (the previous row's
//     cells above it, the current row's cell to its left) as logs, subtracts their maximum and
//     exponentiates once per lane; then the whole cell update -- diagonal and up products, the
//     left recurrence V[c] = A[c] + g V[c-1] as a log2(G)-step multiply-add scan across lanes --
//     is plain f64 arithmetic, and one log per lane brings the cells back.  1 exp + 1 log per
//     cell instead of the reference's 3 dependent exp + log1p pairs.
//     Values more than e^1300 below the largest input of their chunk flush to zero (they could
//     only matter if every larger neighbour within 14 columns died later, which the band
//     excludes); everything else differs from the reference's logdouble chain by rounding only
//     (tests: 1e-9 relative on the log probability).
//   * rows up to 32 cells wide stay in LDS; wider rows (soft-clipped read ends) go through a
//     per-alignment scratch in global memory.
// Compute bound (f64 exp/log latency per row); HBM traffic is negligible.
#pragma once
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>

namespace gaml {

// run-length CIGAR operation: (length << 2) | code; consecutive insertions are merged into one
// run and zero-length operations dropped by the host
constexpr int kOpM = 0, kOpI = 1, kOpD = 2;

struct DpJob {
  int64_t read_off;     // first base of the read in DpArgs::reads
  int64_t ops_off;      // first CIGAR operation in DpArgs::ops
  int64_t scratch_off;  // doubles; 2*max_width are reserved
  int32_t read_len;
  int32_t posstart;     // path position of DP row 1 (PacbioAligmentData::posstart)
  int32_t n_ops;
  int32_t row_f, col_f; // end of the CIGAR path
  int32_t bl, el;       // clip boxes (GetCigarEnds graph.cc:2138-2151, capped at 200)
  int32_t max_width;    // upper bound of the row width (scratch sizing)
};

struct DpArgs {
  const unsigned char* path;  // "path + '\n' + reverse complement" (graph.cc:2687-2688)
  int32_t path_len;
  const unsigned char* reads;
  const DpJob* jobs;
  const uint32_t* ops;
  double* scratch;
  double* out;                // log probability per job
  long long* cells;           // DP cells per job (statistics)
  int32_t* dbg_lo;            // optional: the derived band of job 0 (tests), else null
  int32_t* dbg_hi;
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

// first/last column of the cell set before widening, for successive rows (call with r = r_first,
// r_first + 1, ...).  Every lane of a group runs the same copy.
struct BandWalk {
  const uint32_t* ops;
  int n_ops, k, used, col;
  int row_f, col_f, bl, el;
  __device__ __forceinline__ void row(int r, int& lo1, int& hi1) {
    lo1 = INT_MAX; hi1 = INT_MIN;
    if (r == 0) { lo1 = 0; hi1 = 0; }                               // the cell (0,0), graph.cc:2186
    if (bl > 0 && r >= -bl && r <= 2) { lo1 = min(lo1, 0); hi1 = max(hi1, bl - 1); }  // :2187-2191
    if (r >= 0 && r <= row_f) {                                      // the CIGAR path, :2192-2207
      const int enter = col;
      if (k < n_ops && (int)(ops[k] & 3u) == kOpI) { col += (int)(ops[k] >> 2); k++; }
      lo1 = min(lo1, enter); hi1 = max(hi1, col);
      if (r < row_f) {  // one M or D step into the next row
        const uint32_t op = ops[k];
        if ((int)(op & 3u) == kOpM) col++;
        if (++used == (int)(op >> 2)) { k++; used = 0; }
      }
    }
    if (r >= row_f && r < row_f + el) { lo1 = min(lo1, col_f - el); hi1 = max(hi1, col_f); }  // :2208-2212
  }
};

template <int G>
__global__ __launch_bounds__(256) void pacbio_dp_kernel(DpArgs a) {
  static_assert(G == 8 || G == 16, "the scan uses g^1..g^8");
  constexpr int kLdsWidth = 32;
  constexpr double kShift = 600.0;  // chunk maximum maps to e^600: sums of 16 stay finite, e^-1300 below it is still normal
  __shared__ double lds_rows[256 / G][2][kLdsWidth];
  const int job = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) / G);
  const int grp = (int)(threadIdx.x / G);
  const int j = (int)(threadIdx.x % G);
  if (job >= a.n_jobs) return;  // whole groups leave together
  const DpJob jb = a.jobs[job];
  const double ninf = -__builtin_huge_val();
  const unsigned char* rd = a.reads + jb.read_off;
  double* const gbuf0 = a.scratch + jb.scratch_off;
  double* const lbuf0 = &lds_rows[grp][0][0];
  const int n = jb.read_len;
  const double match = exp(a.log_match), mismatch = exp(a.log_mismatch);  // linear MatchProbability values (graph.h:555-564)
  // g^(1,2,4,8), g = MatchProbability('-', base) = mismatch: a read base is never '-' or the separator
  const double g1 = mismatch, g2 = g1 * g1, g4 = g2 * g2, g8 = g4 * g4;

  BandWalk bw;
  bw.ops = a.ops + jb.ops_off; bw.n_ops = jb.n_ops; bw.k = 0; bw.used = 0; bw.col = 0;
  bw.row_f = jb.row_f; bw.col_f = jb.col_f; bw.bl = jb.bl; bw.el = jb.el;
  const int r_first = jb.bl > 0 ? -jb.bl : 0;
  const int r_last = max(max(jb.row_f, jb.row_f + jb.el - 1), jb.bl > 0 ? 2 : 0);
  // sliding window over the first-pass rows r-2 .. r+2
  int wl0 = INT_MAX, wl1 = INT_MAX, wl2 = INT_MAX, wl3 = INT_MAX, wl4;
  int wh0 = INT_MIN, wh1 = INT_MIN, wh2 = INT_MIN, wh3 = INT_MIN, wh4;
  bw.row(r_first, wl4, wh4);

  double ret = ninf;
  long long cells = 0;
  const double* prev = gbuf0;
  int plo = 0, phi = -1;  // column interval of the previous row (empty before the first row)
  for (int r = r_first - 2, ri = 0; r <= r_last + 2; r++, ri++) {
    const int lo = min(min(min(wl0, wl1), min(wl2, wl3)), wl4) - 2;
    const int hi = max(max(max(wh0, wh1), max(wh2, wh3)), wh4) + 2;
    wl0 = wl1; wl1 = wl2; wl2 = wl3; wl3 = wl4; wh0 = wh1; wh1 = wh2; wh2 = wh3; wh3 = wh4;
    if (r + 3 <= r_last) bw.row(r + 3, wl4, wh4);
    else { wl4 = INT_MAX; wh4 = INT_MIN; }
    if (a.dbg_lo && job == 0 && j == 0) { a.dbg_lo[ri] = lo; a.dbg_hi[ri] = hi; }
    cells += hi - lo + 1;
    double* cur = (hi - lo + 1 <= kLdsWidth) ? lbuf0 + (ri & 1) * kLdsWidth : gbuf0 + (ri & 1) * (int64_t)jb.max_width;
    const int gi = r + jb.posstart - 1;  // path base of this row (graph.cc:2252)
    const bool row_ok = gi >= 0 && gi < a.path_len;
    const unsigned char pc = row_ok ? a.path[gi] : (unsigned char)0;
    const int c0 = lo > 1 ? lo : 1;
    const int c1 = row_ok ? (hi < n ? hi : n) : c0 - 1;  // computed cells [c0, c1] (graph.cc:2246-2255)
    // cells that are never computed: column 0 is the free start (graph.cc:2238-2243), the rest stay zero probability
    for (int c = lo + j; c <= hi; c += G)
      if (c < c0 || c > c1) cur[c - lo] = c == 0 ? 0.0 : ninf;
    if (c1 >= c0) {
      const double up_w = pc == kDpSeparator ? 0.0 : (pc == kDpGap ? match : mismatch);
      double seed = lo <= 0 ? 0.0 : ninf;  // V[c0-1]: the free start when it is in the band
      for (int base = c0; base <= c1; base += G - 2) {
        const int cc = base + j - 2;  // lane 1: base-1, lanes >= 2: the chunk's columns
        double x = ninf;
        if (j == 0) x = seed;
        else if (cc >= plo && cc <= phi && cc <= c1) x = prev[cc - plo];
        double mx = x;
#pragma unroll
        for (int o = 1; o < G; o <<= 1) mx = fmax(mx, __shfl_xor(mx, o, G));
        const bool active = j >= 2 && cc <= c1;
        double out = ninf;
        if (mx != ninf) {  // group-uniform
          const double sh = mx - kShift;
          const double e = x == ninf ? 0.0 : exp(x - sh);
          const double e_left = __shfl_up(e, 1, G);
          double v = 0.0;
          if (active) {
            const unsigned char rc = rd[cc - 1];
            const double diag_w = pc == kDpSeparator ? 0.0 : (pc == rc ? match : mismatch);
            v = e_left * diag_w + e * up_w;
          }
          const double seed_lin = __shfl(e, 0, G);
          if (j == 1) v = seed_lin;
#pragma unroll
          for (int o = 1; o < G; o <<= 1) {
            const double t = __shfl_up(v, o, G);
            const double gp = o == 1 ? g1 : o == 2 ? g2 : o == 4 ? g4 : g8;
            if (j >= o) v = v + t * gp;
          }
          if (v > 0.0) out = log(v) + sh;
        }
        if (active) cur[cc - lo] = out;
        seed = __shfl(out, G - 1, G);
        if (n >= base && n <= base + G - 3 && n <= c1)  // the cell in column |read| (graph.cc:2279-2281)
          ret = dp_lse2(ret, __shfl(out, n - base + 2, G));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    prev = cur;
    plo = lo; phi = hi;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (j == 0) { a.out[job] = ret; a.cells[job] = cells; }
}

}  // namespace gaml
