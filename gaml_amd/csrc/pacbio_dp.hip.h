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
//   * a row is processed in chunks of G-1 columns. A cell is a double times a power of two shared by its chunk: the
//     whole cell update -- diagonal and up products, the left recurrence V[c] = A[c] + g V[c-1] as a log2(G)-step
//     multiply-add scan across the group's DPP row -- is f64 multiply-adds; no exp / log on the row-to-row chain
//     (the reference: 3 dependent exp + log1p pairs per cell). Only the cells of column |read| (the result) go
//     through a log.
//   * rows up to 32 cells wide stay in LDS; wider rows (soft-clipped read ends) go through a
//     per-alignment scratch in global memory.
// Latency bound (one row after another per alignment: LDS round trip + multiply-add scan); HBM traffic is negligible.
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
  // ops[k .. k+3], fetched ahead (0 past the end): a row consumes at most two operations, so what it reads was
  // requested at least a row earlier -- no load latency on the row-to-row chain
  uint32_t q0, q1, q2, q3;
  __device__ __forceinline__ void start() {
    q0 = n_ops > 0 ? ops[0] : 0u; q1 = n_ops > 1 ? ops[1] : 0u; q2 = n_ops > 2 ? ops[2] : 0u; q3 = n_ops > 3 ? ops[3] : 0u;
  }
  __device__ __forceinline__ void advance() {
    k++;
    q0 = q1; q1 = q2; q2 = q3;
    q3 = k + 3 < n_ops ? ops[k + 3] : 0u;
  }
  __device__ __forceinline__ void row(int r, int& lo1, int& hi1) {
    lo1 = INT_MAX; hi1 = INT_MIN;
    if (r == 0) { lo1 = 0; hi1 = 0; }                               // the cell (0,0), graph.cc:2186
    if (bl > 0 && r >= -bl && r <= 2) { lo1 = min(lo1, 0); hi1 = max(hi1, bl - 1); }  // :2187-2191
    if (r >= 0 && r <= row_f) {                                      // the CIGAR path, :2192-2207
      const int enter = col;
      if (k < n_ops && (int)(q0 & 3u) == kOpI) { col += (int)(q0 >> 2); advance(); }
      lo1 = min(lo1, enter); hi1 = max(hi1, col);
      if (r < row_f) {  // one M or D step into the next row
        const uint32_t op = q0;
        if ((int)(op & 3u) == kOpM) col++;
        if (++used == (int)(op >> 2)) { advance(); used = 0; }
      }
    }
    if (r >= row_f && r < row_f + el) { lo1 = min(lo1, col_f - el); hi1 = max(hi1, col_f); }  // :2208-2212
  }
};

// A 16-lane DPP row is one alignment's group: shifts / rotations inside it are VALU moves, no LDS crossbar.
// Lanes without a source lane read 0.0.
template <int CTRL>
__device__ __forceinline__ double dp_row_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
constexpr int kDppRowShr = 0x110, kDppRowRor = 0x120;  // + the lane count

// doubles of scratch per DP row of an alignment whose rows may be `max_width` cells wide: the cells, then their exponents
__host__ __device__ inline int64_t dp_row_stride(int32_t max_width) { return (int64_t)max_width + ((int64_t)max_width + 1) / 2; }

template <int G>
__global__ __launch_bounds__(256) void pacbio_dp_kernel(DpArgs a) {
  static_assert(G == 16, "a group is one DPP row");
  constexpr int kLdsWidth = 32;
  constexpr int kNone = INT_MIN / 2;  // exponent of a zero
  __shared__ double lds_val[256 / G][2][kLdsWidth];
  __shared__ int lds_exp[256 / G][2][kLdsWidth];
  const int job = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) / G);
  const int grp = (int)(threadIdx.x / G);
  const int j = (int)(threadIdx.x % G);
  if (job >= a.n_jobs) return;  // whole groups leave together
  const DpJob jb = a.jobs[job];
  const double ninf = -__builtin_huge_val();
  const unsigned char* rd = a.reads + jb.read_off;
  const int64_t gstride = dp_row_stride(jb.max_width);
  double* const gbuf0 = a.scratch + jb.scratch_off;
  const int n = jb.read_len;
  const double match = exp(a.log_match), mismatch = exp(a.log_mismatch);  // linear MatchProbability values (graph.h:555-564)
  // g^(1,2,4,8), g = MatchProbability('-', base) = mismatch: a read base is never '-' or the separator
  const double g1 = mismatch, g2 = g1 * g1, g4 = g2 * g2, g8 = g4 * g4;

  BandWalk bw;
  bw.ops = a.ops + jb.ops_off; bw.n_ops = jb.n_ops; bw.k = 0; bw.used = 0; bw.col = 0;
  bw.row_f = jb.row_f; bw.col_f = jb.col_f; bw.bl = jb.bl; bw.el = jb.el;
  bw.start();
  const int r_first = jb.bl > 0 ? -jb.bl : 0;
  const int r_last = max(max(jb.row_f, jb.row_f + jb.el - 1), jb.bl > 0 ? 2 : 0);
  // sliding window over the first-pass rows r-2 .. r+2
  int wl0 = INT_MAX, wl1 = INT_MAX, wl2 = INT_MAX, wl3 = INT_MAX, wl4;
  int wh0 = INT_MIN, wh1 = INT_MIN, wh2 = INT_MIN, wh3 = INT_MIN, wh4;
  bw.row(r_first, wl4, wh4);

  // A cell is a double times 2^exponent, the exponent shared by the (up to G-1) cells a chunk writes. A chunk brings
  // its inputs -- the previous row's cells above it, the carry from the chunk to its left -- to the largest of their
  // binary exponents, then the reference's logdouble sum of three products per cell is two multiply-adds and a scan:
  // no exp / log on the row-to-row chain. An input more than 2^-1022 below the largest input of its chunk is flushed
  // towards zero (it could only matter if every larger neighbour within 15 columns died later, which the band
  // excludes); everything else differs from the logdouble chain by rounding only (tests: 1e-9 relative on the log).
  double ret = ninf;
  long long cells = 0;
  const double* prev_v = gbuf0;
  const int* prev_e = (const int*)(gbuf0 + jb.max_width);
  int plo = 0, phi = -1;  // column interval of the previous row (empty before the first row)
  // this row's column interval; its path base and the read bases of its first chunk were requested a row earlier
  int lo = min(min(min(wl0, wl1), min(wl2, wl3)), wl4) - 2;
  int hi = max(max(max(wh0, wh1), max(wh2, wh3)), wh4) + 2;
  auto path_base = [&](int r) -> unsigned char {
    const int gi = r + jb.posstart - 1;  // path base of row r (graph.cc:2252)
    return gi >= 0 && gi < a.path_len ? a.path[gi] : (unsigned char)0;
  };
  auto read_base = [&](int lo_r, int hi_r) -> unsigned char {  // lane j's base in the first chunk of a row [lo_r, hi_r]
    if (lo_r > hi_r) return 0;  // (past the last row the window is empty: INT_MAX / INT_MIN)
    const int cc = (lo_r > 1 ? lo_r : 1) + j - 1;
    return j >= 1 && cc <= hi_r && cc <= n ? rd[cc - 1] : (unsigned char)0;
  };
  unsigned char pc_ahead = path_base(r_first - 2), rc_ahead = read_base(lo, hi);
  for (int r = r_first - 2, ri = 0; r <= r_last + 2; r++, ri++) {
    wl0 = wl1; wl1 = wl2; wl2 = wl3; wl3 = wl4; wh0 = wh1; wh1 = wh2; wh2 = wh3; wh3 = wh4;
    if (r + 3 <= r_last) bw.row(r + 3, wl4, wh4);
    else { wl4 = INT_MAX; wh4 = INT_MIN; }
    const int lo_next = min(min(min(wl0, wl1), min(wl2, wl3)), wl4) - 2;
    const int hi_next = max(max(max(wh0, wh1), max(wh2, wh3)), wh4) + 2;
    const unsigned char pc = pc_ahead, rc_first = rc_ahead;
    pc_ahead = path_base(r + 1);
    rc_ahead = read_base(lo_next, hi_next);
    if (a.dbg_lo && job == 0 && j == 0) { a.dbg_lo[ri] = lo; a.dbg_hi[ri] = hi; }
    cells += hi - lo + 1;
    const bool in_lds = hi - lo + 1 <= kLdsWidth;
    double* cur_v = in_lds ? &lds_val[grp][ri & 1][0] : gbuf0 + (ri & 1) * gstride;
    int* cur_e = in_lds ? &lds_exp[grp][ri & 1][0] : (int*)(gbuf0 + (ri & 1) * gstride + jb.max_width);
    const int gi = r + jb.posstart - 1;  // path base of this row (graph.cc:2252)
    const bool row_ok = gi >= 0 && gi < a.path_len;
    const int c0 = lo > 1 ? lo : 1;
    const int c1 = row_ok ? (hi < n ? hi : n) : c0 - 1;  // computed cells [c0, c1] (graph.cc:2246-2255)
    // cells that are never computed: column 0 is the free start (graph.cc:2238-2243), the rest stay zero probability
    for (int c = lo + j; c <= hi; c += G)
      if (c < c0 || c > c1) { cur_v[c - lo] = c == 0 ? 1.0 : 0.0; cur_e[c - lo] = 0; }
    if (c1 >= c0) {
      const double up_w = pc == kDpSeparator ? 0.0 : (pc == kDpGap ? match : mismatch);
      double carry = lo <= 0 ? 1.0 : 0.0;  // V[c0-1]: the free start when it is in the band
      int carry_e = 0;
      for (int base = c0; base <= c1; base += G - 1) {
        const int cc = base + j - 1;  // lane 0: base-1 (the cell above-left of the chunk + the carry), lanes >= 1: the chunk's columns
        double x = 0.0;               // the previous row's cell in column cc
        int xe = 0;
        if (cc >= plo && cc <= phi && cc <= c1) { x = prev_v[cc - plo]; xe = prev_e[cc - plo]; }
        // the chunk's scale: the largest binary exponent among its inputs
        int m = x > 0.0 ? xe + ilogb(x) : kNone;
        if (j == 0 && carry > 0.0) m = max(m, carry_e + ilogb(carry));
        m = max(m, __builtin_amdgcn_update_dpp(kNone, m, kDppRowRor + 1, 0xf, 0xf, false));
        m = max(m, __builtin_amdgcn_update_dpp(kNone, m, kDppRowRor + 2, 0xf, 0xf, false));
        m = max(m, __builtin_amdgcn_update_dpp(kNone, m, kDppRowRor + 4, 0xf, 0xf, false));
        m = max(m, __builtin_amdgcn_update_dpp(kNone, m, kDppRowRor + 8, 0xf, 0xf, false));
        const int S = m;  // group-uniform
        x = ldexp(x, max(xe - S, -4096));
        const double x_left = dp_row_move<kDppRowShr + 1>(x);
        const bool active = j >= 1 && cc <= c1;
        double v = 0.0;
        if (active) {
          const unsigned char rc = base == c0 ? rc_first : rd[cc - 1];
          const double diag_w = pc == kDpSeparator ? 0.0 : (pc == rc ? match : mismatch);
          v = x_left * diag_w + x * up_w;
        }
        if (j == 0) v = ldexp(carry, max(carry_e - S, -4096));  // V[base-1] of this row
        // the left recurrence V[c] = A[c] + g V[c-1] as a multiply-add scan across the group
        v = fma(dp_row_move<kDppRowShr + 1>(v), g1, v);
        v = fma(dp_row_move<kDppRowShr + 2>(v), g2, v);
        v = fma(dp_row_move<kDppRowShr + 4>(v), g4, v);
        v = fma(dp_row_move<kDppRowShr + 8>(v), g8, v);
        if (active) { cur_v[cc - lo] = v; cur_e[cc - lo] = S; }
        carry = dp_row_move<kDppRowRor + 1>(v);  // lane 0 <- lane 15: the next chunk's V[base-1]
        carry_e = S;
        if (n >= base && n <= base + G - 2 && n <= c1) {  // the cell in column |read| (graph.cc:2279-2281)
          const double cell = __shfl(v, n - base + 1, G);
          if (cell > 0.0) ret = dp_lse2(ret, log(cell) + (double)S * 0.693147180559945309417);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    prev_v = cur_v; prev_e = cur_e;
    plo = lo; phi = hi;
    lo = lo_next; hi = hi_next;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (j == 0) { a.out[job] = ret; a.cells[job] = cells; }
}

}  // namespace gaml
