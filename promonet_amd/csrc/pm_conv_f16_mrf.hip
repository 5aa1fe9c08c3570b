// The whole-MRF launchers for ElemF16 in a translation unit of their own: these
// kernels alternate short MFMA loops with VALU-bound epilogues, and the
// max-ilp scheduling strategy (Makefile: MRF_FLAGS) is worth 2.4 % on them
// (profiles/r03/ab_skew.txt) while it costs the other conv kernels up to 1 %.
#define PM_INSTANTIATE
#include "pm_launch.h"
template hipError_t pm_launch_mrf<ElemF16>(int, const Block3Args (&)[3], hipStream_t);
