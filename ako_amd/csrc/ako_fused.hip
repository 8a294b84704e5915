// ako_fused.hip -- the two-level workgroup kernels (ako_fused.hip.h) as a translation unit of their own; the plan code calls
// the launchers declared in ako_fused.h.  EXPERIMENTAL: built only with AKO_BUILD_EXPERIMENTAL=1 (ako_amd/build.py); the
// default library does not hold these kernels (measured slower than the level-per-kernel launches).
#include "ako_fused.hip.h"

#include <atomic>
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

int akoFused2ForwardLaunch(int kind, const F2Params& P_in, hipStream_t st)
{
	F2_WITH_DBG(P_in);
	const F2Params& P = P_in_use;
	const uint32_t blocks = P.groups * P.segs * P.n_tiles * P.batch;
	if (kind == K_DD137)
		hipLaunchKernelGGL((k_fused2_forward<K_DD137>), dim3(blocks), dim3(F2_THREADS), 0, st, P);
	else
		hipLaunchKernelGGL((k_fused2_forward<K_CDF53>), dim3(blocks), dim3(F2_THREADS), 0, st, P);
	return 0;
}

int akoFused2InverseLaunch(int kind, const F2Params& P_in, hipStream_t st)
{
	F2_WITH_DBG(P_in);
	const F2Params& P = P_in_use;
	// 143 KiB of the CU's 160 KiB: beyond HIP's 64 KiB default.  The attribute belongs to the device and function: raised once
	// per device (a plan's launches come from one thread at a time, but several plans may start at once: atomics)
	static std::atomic<int> raised[64];
	int dev = 0;
	(void)hipGetDevice(&dev);
	if (dev >= 0 && dev < 64 && raised[dev].load(std::memory_order_acquire) == 0)
	{
		const hipError_t e1 = hipFuncSetAttribute((const void*)k_fused2_inverse<K_DD137>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F2I_LDS_BYTES);
		const hipError_t e2 = hipFuncSetAttribute((const void*)k_fused2_inverse<K_CDF53>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F2I_LDS_BYTES);
		if (e1 != hipSuccess || e2 != hipSuccess)
			return 1;  // (the caller reports it: the launch would fail for want of LDS)
		raised[dev].store(1, std::memory_order_release);
	}
	const uint32_t blocks = P.groups * P.segs * P.n_tiles * P.batch;
	if (kind == K_DD137)
		hipLaunchKernelGGL((k_fused2_inverse<K_DD137>), dim3(blocks), dim3(F2_THREADS), F2I_LDS_BYTES, st, P);
	else
		hipLaunchKernelGGL((k_fused2_inverse<K_CDF53>), dim3(blocks), dim3(F2_THREADS), F2I_LDS_BYTES, st, P);
	return 0;
}

}  // namespace ako
