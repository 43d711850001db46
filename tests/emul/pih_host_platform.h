// pih_host_platform.h -- TEST-ONLY platform section for compiling the product's algorithm headers on the host (see pih_math.h):
// `real` is double or float (PIH_REAL), functions are plain inline, tables are static const, intrinsics become portable C.
#pragma once
#include <math.h>
#include <stdint.h>
#define PIH_PLATFORM_DEFINED 1
#ifndef PIH_REAL
#define PIH_REAL float
#endif
typedef PIH_REAL real;
#define PIH_HD inline
#define PIH_NOINL inline
#define PIH_CONST static const
namespace pih {
inline bool finite_small(double x) { return x == x && (x < 0 ? -x : x) <= 1e15; }
inline bool finite_small(float x) { return x == x && (x < 0 ? -x : x) <= 1e15f; }
inline real med3_(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline real max_(real a, real b) { return a > b ? a : b; }
inline void sincos_(double a, double* s, double* c) { *s = sin(a); *c = cos(a); }
inline void sincos_(float a, float* s, float* c) { *s = sinf(a); *c = cosf(a); }
inline double acos_(double a) { return acos(a); }
inline float acos_(float a) { return acosf(a); }
}  // namespace pih
