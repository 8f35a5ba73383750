// tests/sanitize/hip/hip_runtime.h — stand-in for <hip/hip_runtime.h> so that the DEVICE headers of the solvers
// (ros2_mono_vo_amd/csrc/geom_linalg.h, geom_models.h) compile as plain x86 C++ for the ASan / UBSan harness.
// Only what those two headers use: the function-space qualifiers.  Test infrastructure, not part of the product.
#pragma once
#include <cmath>
#include <cstdio>
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __constant__
#include <cfenv>
static inline int __double2int_rn(double v) { return (int)std::nearbyint(v); }   // round half to even (default rounding mode)
static inline int __float2int_rn(float v) { return (int)std::nearbyintf(v); }
