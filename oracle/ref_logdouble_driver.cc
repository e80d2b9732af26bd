// ref_logdouble_driver.cc -- TEST INFRASTRUCTURE ONLY.
// Our own driver TU. It #includes two of the reference's header-only files IN PLACE from
// /root/reference (logdouble.hpp, utility.h -- they depend only on libm / libstdc++), so
// the compiled object carries the reference's real logdouble arithmetic and InvertPath.
// Built by oracle/Makefile into oracle/_ref/ (git-ignored; never committed). Nothing from
// the reference is copied into this repository.
#include <cstdint>
#include <vector>
#include "logdouble.hpp"   // /root/reference/logdouble.hpp
#include "utility.h"       // /root/reference/utility.h

extern "C" {
double ref_ld_from_linear(double x) { logdouble a(x); return a.logval; }
double ref_ld_default() { logdouble a; return a.logval; }
static logdouble mk(double l) { logdouble a; a.logval = l; return a; }
double ref_ld_add(double a, double b) { return (mk(a) + mk(b)).logval; }
double ref_ld_add_assign(double a, double b) { logdouble x = mk(a); x += mk(b); return x.logval; }
double ref_ld_mul(double a, double b) { return (mk(a) * mk(b)).logval; }
double ref_ld_mul_assign(double a, double b) { logdouble x = mk(a); x *= mk(b); return x.logval; }
double ref_ld_pow(double a, double e) { return (mk(a) ^ e).logval; }
double ref_ld_div(double a, double b) { return (mk(a) / mk(b)).logval; }
int ref_ld_lt(double a, double b) { return mk(a) < mk(b); }
int ref_ld_gt(double a, double b) { return mk(a) > mk(b); }
int ref_invert_path(const int32_t* w, int n, int32_t* out) {
  std::vector<int> r = InvertPath(std::vector<int>(w, w + n));
  for (int i = 0; i < n; i++) out[i] = r[i];
  return (int)r.size();
}
int ref_reverse_path(int32_t* w, int n) {
  std::vector<int> v(w, w + n);
  ReversePath(v);
  for (int i = 0; i < n; i++) w[i] = v[i];
  return n;
}
}
