// tests/mock_ref/logdouble.hpp -- TEST-ONLY stand-in for the one thing the adapter reads from the reference's
// logdouble (logdouble.hpp:13-35): a public `logval` holding log(x) of the constructor's linear argument.
#ifndef MOCK_REF_LOGDOUBLE_HPP__
#define MOCK_REF_LOGDOUBLE_HPP__
#include <cmath>
struct logdouble {
  double logval;
  logdouble() : logval(-INFINITY) {}
  logdouble(double x) : logval(std::log(x)) {}
};
#endif
