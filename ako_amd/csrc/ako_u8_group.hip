// ako_u8_group.hip -- the u8 level-0 forward kernel in column groups (k_forward_group_u8 of ako_stream.hip.h), a
// translation unit of its own so that it builds in parallel with the others (see ako_u8.h)
#include "ako_stream.hip.h"
#include "ako_u8.h"

namespace ako
{

void akoLaunchForwardGroupU8_rgba(int kind, const LevelParams& P, const StreamGeom& G, uint32_t blocks, hipStream_t st)
{
	if (kind == K_DD137)
		hipLaunchKernelGGL((k_forward_group_u8<K_DD137, 4>), dim3(blocks), dim3(64 * GRP_WAVES), 0, st, P, G);
	else
		hipLaunchKernelGGL((k_forward_group_u8<K_CDF53, 4>), dim3(blocks), dim3(64 * GRP_WAVES), 0, st, P, G);
}

}  // namespace ako
