// prof.h -- launch timing shared by the hot kernels (bench.py's `roofline` object): with profiling enabled, a sampled launch
// is bracketed by two HIP events on its own stream and tagged with its algorithmic bytes / flops and its shape.
#pragma once
#include <hip/hip_runtime.h>
// family: 0 = forward / data-gradient convolution, wave-per-tile kernel (big levels); 2 = the same contraction,
// workgroup-per-tile kernel (few-row levels); 1 = weight-gradient kernels; 3 = hg_gemm* (dense layers of the heads);
// 4 = td_* (captioner / language-encoder recurrence); 5 = cl_bfs2 (cluster replay); 6 = un_bn_* (BatchNorm passes).
// Returns an opaque record (NULL when profiling is off or this launch is not sampled) to pass to d3_prof_end after the launch.
#define D3_PROF_TAGS 12
void *d3_prof_begin(int family, double bytes, double flops, hipStream_t s);
void d3_prof_end(void *rec, hipStream_t s);
// shape / kernel-instance tags of a record (no-op on NULL): convolutions {Min, Mout, K, Cin, Cout, NT, WLDS, XBF, NW, F32M, KT, ST}
void d3_prof_tag(void *rec, int idx, int value);
// a device-side value the record's byte count depends on (BFS: the number of edge records actually streamed is known only on
// the device): returns a device pointer to one double the caller's kernel fills (NULL: no slot); d3_prof_dump folds
// `bytes += scale * value` into the record when it resolves it
double *d3_prof_dev_slot(void *rec, double scale);
