// ako_fused.hip -- the two-level workgroup kernels (ako_fused.hip.h) as a translation unit of their own, so that
// they build in parallel with ako_plan.hip; the plan code calls the launchers declared in ako_fused.h.
#include "ako_fused.hip.h"

#include <stdlib.h>

namespace ako
{

#ifdef AKO_MEASURE
static uint32_t f2_dbg()
{
	static const uint32_t v = getenv("AKO_F2_DBG") ? (uint32_t)atoi(getenv("AKO_F2_DBG")) : 0u;
	return v;
}
#define F2_WITH_DBG(P) F2Params P##_d = P; P##_d.dbg = f2_dbg(); const F2Params& P##_use = P##_d
#else
#define F2_WITH_DBG(P) const F2Params& P##_use = P
#endif

void akoFused2ForwardLaunch(int kind, const F2Params& P_in, hipStream_t st)
{
	F2_WITH_DBG(P_in);
	const F2Params& P = P_in_use;
	const uint32_t blocks = P.groups * P.segs * P.n_tiles * P.batch;
	if (kind == K_DD137)
		hipLaunchKernelGGL((k_fused2_forward<K_DD137>), dim3(blocks), dim3(F2_THREADS), 0, st, P);
	else
		hipLaunchKernelGGL((k_fused2_forward<K_CDF53>), dim3(blocks), dim3(F2_THREADS), 0, st, P);
}

void akoFused2InverseLaunch(int kind, const F2Params& P_in, hipStream_t st)
{
	F2_WITH_DBG(P_in);
	const F2Params& P = P_in_use;
	static bool raised = false;  // 143 KiB of the CU's 160 KiB: beyond HIP's 64 KiB default
	if (!raised)
	{
		(void)hipFuncSetAttribute((const void*)k_fused2_inverse<K_DD137>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F2I_LDS_BYTES);
		(void)hipFuncSetAttribute((const void*)k_fused2_inverse<K_CDF53>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F2I_LDS_BYTES);
		raised = true;
	}
	const uint32_t blocks = P.groups * P.segs * P.n_tiles * P.batch;
	if (kind == K_DD137)
		hipLaunchKernelGGL((k_fused2_inverse<K_DD137>), dim3(blocks), dim3(F2_THREADS), F2I_LDS_BYTES, st, P);
	else
		hipLaunchKernelGGL((k_fused2_inverse<K_CDF53>), dim3(blocks), dim3(F2_THREADS), F2I_LDS_BYTES, st, P);
}

}  // namespace ako
