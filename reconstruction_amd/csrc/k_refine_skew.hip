// k_refine_skew.hip -- k_refine_skew, DisparityRefine's time-skewed sweeps (CStereoMatching.cpp:572-680), the path's dominant kernel,
// as a translation unit of its own: its source is in k_refine.hip beside the device functions it shares with the other refine
// kernels.  What differs is the compiler's scheduling strategy (Makefile: -mllvm -amdgpu-sched-strategy=max-memory-clause):
// measured on C2 with the whole of k_refine.hip under each strategy, that one ran this kernel's launches 8 % faster beside the
// other pairs' kernels (0.365 against 0.398 ms) and the single-sweep kernels slower -- so only this kernel gets it.
#define RF_TU 2
#include "k_refine.hip"
