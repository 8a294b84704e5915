// ako_tail3.hip -- the line-engine tail kernels (ako_tail3.hip.h) as a translation unit of their own.
#include "ako_tail3.hip.h"

namespace ako
{

static void t3_raise_lds_limit()
{
	static bool raised = false;  // up to all 160 KiB of a CU: beyond HIP's 64 KiB default
	if (!raised)
	{
		(void)hipFuncSetAttribute((const void*)k_forward_tail3, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute((const void*)k_inverse_tail3, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		raised = true;
	}
}

void akoTail3ForwardLaunch(const TailParams& P, uint32_t blocks, uint32_t threads, uint32_t lds_bytes, hipStream_t st)
{
	t3_raise_lds_limit();
	hipLaunchKernelGGL(k_forward_tail3, dim3(blocks), dim3(threads), lds_bytes, st, P);
}

void akoTail3InverseLaunch(const TailParams& P, uint32_t blocks, uint32_t threads, uint32_t lds_bytes, hipStream_t st)
{
	t3_raise_lds_limit();
	hipLaunchKernelGGL(k_inverse_tail3, dim3(blocks), dim3(threads), lds_bytes, st, P);
}

}  // namespace ako
