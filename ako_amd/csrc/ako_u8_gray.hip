// ako_u8_gray.hip -- the native u8 level-0 kernels for one- and two-channel images (ako_u8_gray.hip.h), a translation unit of
// their own so that they build in parallel with the others; launchers declared in ako_u8.h
#include "ako_stream.hip.h"
#include "ako_u8_gray.hip.h"
#include "ako_u8.h"

namespace ako
{

void akoLaunchForwardU8_gray(int kind, int channels, const LevelParams& P, const StreamGeom& G, uint32_t blocks, uint32_t threads, hipStream_t st)
{
#define AKO_GRAY_FWD(K, C) hipLaunchKernelGGL((k_forward_u8_gray<K, C>), dim3(blocks), dim3(threads), 0, st, P, G)
	if (kind == K_DD137)
	{
		if (channels == 1) AKO_GRAY_FWD(K_DD137, 1); else AKO_GRAY_FWD(K_DD137, 2);
	}
	else
	{
		if (channels == 1) AKO_GRAY_FWD(K_CDF53, 1); else AKO_GRAY_FWD(K_CDF53, 2);
	}
#undef AKO_GRAY_FWD
}

void akoLaunchInverseU8_gray(int kind, int channels, const LevelParams& P, const StreamGeom& G, uint32_t blocks, uint32_t threads, hipStream_t st)
{
#define AKO_GRAY_INV(K, C) hipLaunchKernelGGL((k_inverse_u8_gray<K, C>), dim3(blocks), dim3(threads), 0, st, P, G)
	if (kind == K_DD137)
	{
		if (channels == 1) AKO_GRAY_INV(K_DD137, 1); else AKO_GRAY_INV(K_DD137, 2);
	}
	else
	{
		if (channels == 1) AKO_GRAY_INV(K_CDF53, 1); else AKO_GRAY_INV(K_CDF53, 2);
	}
#undef AKO_GRAY_INV
}

}  // namespace ako
