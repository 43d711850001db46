// pih_counted_real.h -- TEST / TOOL ONLY: a `real` that counts the floating-point operations executed on it.  Building the
// product's algorithm headers with PIH_REAL = CountedReal (tests/emul, libpih_emul_cnt.so) turns one env-step of the host build
// into an exact count of the algorithm's arithmetic, per phase: tools/count_flops.py -> profiles/flops_latest.json (the figure
// bench.py quotes next to the roofline instead of an estimate).  Counted: add/sub, mul, fused or separate multiply-add as
// written in the source (the compiler's contraction is not modelled: a*b+c counts 2), div, sqrt, transcendental (sin, cos,
// atan2, acos, asin), comparisons and min/max/abs are counted separately and NOT included in the flop total.
#pragma once
#include <math.h>
#include <stdint.h>
#include <type_traits>

struct FlopCounters { uint64_t add, mul, div, sqrt_, trans, cmp; };
inline FlopCounters& flop_counters() { static thread_local FlopCounters c = {0, 0, 0, 0, 0, 0}; return c; }

struct CountedReal {
  double v;
  CountedReal() = default;
  template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type> CountedReal(T x) : v((double)x) {}
  explicit operator double() const { return v; }
  explicit operator float() const { return (float)v; }
  explicit operator int() const { return (int)v; }
  explicit operator unsigned() const { return (unsigned)v; }
  explicit operator long() const { return (long)v; }
  explicit operator unsigned long() const { return (unsigned long)v; }
  explicit operator unsigned long long() const { return (unsigned long long)v; }
  explicit operator bool() const { return v != 0; }
  CountedReal& operator+=(CountedReal o) { flop_counters().add++; v += o.v; return *this; }
  CountedReal& operator-=(CountedReal o) { flop_counters().add++; v -= o.v; return *this; }
  CountedReal& operator*=(CountedReal o) { flop_counters().mul++; v *= o.v; return *this; }
  CountedReal& operator/=(CountedReal o) { flop_counters().div++; v /= o.v; return *this; }
  CountedReal operator-() const { CountedReal r; r.v = -v; return r; }
};
#define PIH_CR_BIN(op, ctr) \
  inline CountedReal operator op(CountedReal a, CountedReal b) { flop_counters().ctr++; CountedReal r; r.v = a.v op b.v; return r; } \
  template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type> inline CountedReal operator op(CountedReal a, T b) { return a op CountedReal(b); } \
  template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type> inline CountedReal operator op(T a, CountedReal b) { return CountedReal(a) op b; }
PIH_CR_BIN(+, add) PIH_CR_BIN(-, add) PIH_CR_BIN(*, mul) PIH_CR_BIN(/, div)
#define PIH_CR_CMP(op) \
  inline bool operator op(CountedReal a, CountedReal b) { flop_counters().cmp++; return a.v op b.v; } \
  template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type> inline bool operator op(CountedReal a, T b) { flop_counters().cmp++; return a.v op (double)b; } \
  template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type> inline bool operator op(T a, CountedReal b) { flop_counters().cmp++; return (double)a op b.v; }
PIH_CR_CMP(<) PIH_CR_CMP(>) PIH_CR_CMP(<=) PIH_CR_CMP(>=) PIH_CR_CMP(==) PIH_CR_CMP(!=)
inline CountedReal sqrt(CountedReal a) { flop_counters().sqrt_++; return CountedReal(::sqrt(a.v)); }
inline CountedReal sin(CountedReal a) { flop_counters().trans++; return CountedReal(::sin(a.v)); }
inline CountedReal cos(CountedReal a) { flop_counters().trans++; return CountedReal(::cos(a.v)); }
inline CountedReal acos(CountedReal a) { flop_counters().trans++; return CountedReal(::acos(a.v)); }
inline CountedReal asin(CountedReal a) { flop_counters().trans++; return CountedReal(::asin(a.v)); }
inline CountedReal atan2(CountedReal a, CountedReal b) { flop_counters().trans++; return CountedReal(::atan2(a.v, b.v)); }
inline CountedReal fabs(CountedReal a) { return CountedReal(::fabs(a.v)); }
