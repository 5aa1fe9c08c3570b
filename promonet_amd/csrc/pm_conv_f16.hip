// Explicit instantiation of the MFMA convolution launchers for ElemF16.
#define PM_INSTANTIATE
#include "pm_launch.h"
template hipError_t pm_launch_pair<ElemF16>(int, int, const PairArgs&, hipStream_t);
template int pm_pair_tile_len<ElemF16>(int, int);
template hipError_t pm_launch_single<ElemF16>(int, int, int, const SingleArgs&, hipStream_t);
template hipError_t pm_launch_block3<ElemF16>(int, int, const Block3Args&, hipStream_t);
template int pm_pair_chunk<ElemF16>(int);
template bool pm_block3_supported<ElemF16>(int, int);
