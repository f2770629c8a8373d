// hipemu stand-in for <hip/hip_fp16.h> (TEST INFRASTRUCTURE ONLY): IEEE binary16 via the compiler's _Float16.
#pragma once
typedef _Float16 __half;
static inline __half __float2half(float f) { return (__half)f; }
static inline float __half2float(__half h) { return (float)h; }
