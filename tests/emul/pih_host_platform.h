// pih_host_platform.h -- TEST-ONLY platform section for compiling the product's algorithm headers on the host (see pih_math.h):
// `real` is double or float (PIH_REAL), functions are plain inline, tables are static const, intrinsics become portable C.
#pragma once
#include <math.h>
#include <stdint.h>
#define PIH_PLATFORM_DEFINED 1
#ifdef PIH_COUNT_FLOPS
#include "pih_counted_real.h"     // real = a double that counts the operations executed on it (tools/count_flops.py)
typedef CountedReal real;
#else
#ifndef PIH_REAL
#define PIH_REAL float
#endif
typedef PIH_REAL real;
#endif
#define PIH_HD inline
#define PIH_NOINL inline
#define PIH_CONST static const
namespace pih {
inline bool finite_small(double x) { return x == x && (x < 0 ? -x : x) <= 1e15; }
inline bool finite_small(float x) { return x == x && (x < 0 ? -x : x) <= 1e15f; }
#ifdef PIH_COUNT_FLOPS
inline bool finite_small(CountedReal x) { return finite_small(x.v); }
inline void sincos_(CountedReal a, CountedReal* s, CountedReal* c) { flop_counters().trans += 2; s->v = ::sin(a.v); c->v = ::cos(a.v); }
inline CountedReal acos_(CountedReal a) { return acos(a); }
#endif
inline real med3_(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline real max_(real a, real b) { return a > b ? a : b; }
inline void sincos_(double a, double* s, double* c) { *s = sin(a); *c = cos(a); }
inline void sincos_(float a, float* s, float* c) { *s = sinf(a); *c = cosf(a); }
inline double acos_(double a) { return acos(a); }
inline float acos_(float a) { return acosf(a); }
}  // namespace pih
