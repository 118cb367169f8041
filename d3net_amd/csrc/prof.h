// prof.h -- launch timing shared by the convolution kernels (bench.py's `roofline` object): with profiling
// enabled, a launch is bracketed by two HIP events on its own stream and tagged with its algorithmic bytes.
#pragma once
#include <hip/hip_runtime.h>
// family: 0 = forward / data-gradient convolution, wave-per-tile kernel (big levels); 2 = the same contraction,
// workgroup-per-tile kernel (few-row levels); 1 = weight-gradient kernels.  Returns an opaque
// record (NULL when profiling is off) to pass to d3_prof_end after the launch.
void *d3_prof_begin(int family, double bytes, double flops, hipStream_t s);
void d3_prof_end(void *rec, hipStream_t s);
