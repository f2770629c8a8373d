// Host-emulator stand-in for open3d-ml_amd/csrc/gfx950_ops.h (TEST INFRASTRUCTURE ONLY): the same operations in
// portable C++.  This directory precedes the product sources on the emulator's include path.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#define ML3D_WAVES_PER_SIMD(n)

namespace ml3d {
static inline void key_minmax(double a, double b, double& lo, double& hi) { lo = std::fmin(a, b); hi = std::fmax(a, b); }
static inline double key_min(double a, double b) { return std::fmin(a, b); }
static inline void key_min_if(bool take, double& acc, double key) { if (take) acc = std::fmin(acc, key); }
static inline void key_insert_step(double& slot, double& x) { const double t = std::fmax(slot, x); slot = std::fmin(slot, x); x = t; }
// (lanes are independent fibers here: a ballot in divergent loops cannot rendezvous; the lane's own predicate is a valid answer
//  for every caller -- see the product header)
static inline bool wave_any_active(bool pred) { return pred; }
}  // namespace ml3d
