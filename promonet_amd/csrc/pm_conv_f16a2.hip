// Explicit instantiation of the MFMA convolution launchers for ElemF16A2 (f16
// operands, the ACTIVATIONS split into hi + lo: two MFMAs per k16 step sharing
// one weight fragment - pm_common.h).
#define PM_INSTANTIATE
#include "pm_launch.h"
template hipError_t pm_launch_pair<ElemF16A2>(int, int, const PairArgs&, hipStream_t);
template int pm_pair_tile_len<ElemF16A2>(int, int);
template hipError_t pm_launch_single<ElemF16A2>(int, int, int, const SingleArgs&, hipStream_t);
template hipError_t pm_launch_block3<ElemF16A2>(int, int, const Block3Args&, hipStream_t);
template hipError_t pm_launch_mrf<ElemF16A2>(int, const Block3Args (&)[3], hipStream_t);
template int pm_pair_chunk<ElemF16A2>(int);
template bool pm_block3_supported<ElemF16A2>(int, int);
