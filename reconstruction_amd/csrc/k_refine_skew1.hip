// k_refine_skew1.hip -- k_refine_skew1, the one-wave-per-strip form of DisparityRefine's time-skewed sweeps
// (CStereoMatching.cpp:572-680; option refine_skew_variant = 64), as a translation unit of its own: the kernel's source is in
// k_refine.hip beside the device functions it shares with the other refine kernels (the specified exp, the update, the data term,
// the miss service).  What differs is the compiler's scheduling strategy (Makefile: -mllvm -amdgpu-sched-strategy=max-ilp): the
// kernel advances four independent rows per step and wants their fp64 chains side by side; the default strategy finishes one
// row's chain before it starts the next (lowest register pressure), and fencing the stages by hand costs registers.
#define RF_TU 1
#include "k_refine.hip"
