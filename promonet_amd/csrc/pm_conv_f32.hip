// Explicit instantiation of the MFMA convolution launchers for ElemF32.
#define PM_INSTANTIATE
#define PM_INSTANTIATE_STFT
#include "pm_launch.h"
template hipError_t pm_launch_pair<ElemF32>(int, int, const PairArgs&, hipStream_t);
template int pm_pair_tile_len<ElemF32>(int, int);
template hipError_t pm_launch_single<ElemF32>(int, int, int, const SingleArgs&, hipStream_t);
template hipError_t pm_launch_block3<ElemF32>(int, int, const Block3Args&, hipStream_t);
template hipError_t pm_launch_mrf<ElemF32>(int, const Block3Args (&)[3], hipStream_t);
template int pm_pair_chunk<ElemF32>(int);
template bool pm_block3_supported<ElemF32>(int, int);
