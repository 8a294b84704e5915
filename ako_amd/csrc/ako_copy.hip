// ako_copy.hip -- the device's practical copy rate, the yardstick bench.py reports beside the 8 TB/s spec peak.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ako
{

// ---- the device's practical copy rate (the yardstick bench.py reports beside the 8 TB/s spec peak) ----
// Four independent 16-byte loads in flight per lane before the first store, non-temporal both ways, every workgroup on
// consecutive 16 KiB pieces: 5.3-5.9 TB/s read + write on MI355X where a one-load-in-flight grid-stride copy and
// hipMemcpyAsync reach 4.6-5.1 (profiles/r3_hbm_rates.txt).
typedef uint32_t copy_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_tuned_copy(const copy_u32x4* __restrict__ in, copy_u32x4* __restrict__ out, size_t n)
{
	constexpr int U = 4;
	const size_t per_block = (size_t)blockDim.x * U;
	for (size_t base = blockIdx.x * per_block; base < n; base += (size_t)gridDim.x * per_block)
	{
		copy_u32x4 v[U];
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
			v[u] = (i < n) ? __builtin_nontemporal_load(&in[i]) : copy_u32x4{0, 0, 0, 0};
		}
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
			if (i < n)
				__builtin_nontemporal_store(v[u], &out[i]);
		}
	}
}

}  // namespace ako

// read + write GB/s of a device-to-device copy of `bytes` (two buffers of that size are allocated and freed); 0 on failure
extern "C" __attribute__((visibility("default"))) double akoHipTunedCopyGBps(size_t bytes, int repeats)
{
	void *a = nullptr, *b = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	double rate = 0.0;
	if (bytes < 4096 || repeats < 1)
		return 0.0;
	if (hipMalloc(&a, bytes) == hipSuccess && hipMalloc(&b, bytes) == hipSuccess && hipMemset(a, 1, bytes) == hipSuccess &&
	    hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess)
	{
		const size_t n = bytes / 16;
		auto launch = [&] { hipLaunchKernelGGL(ako::k_tuned_copy, dim3(16384), dim3(256), 0, 0, (const ako::copy_u32x4*)a, (ako::copy_u32x4*)b, n); };
		launch();
		(void)hipEventRecord(e0, 0);
		for (int i = 0; i < repeats; i++)
			launch();
		(void)hipEventRecord(e1, 0);
		float ms = 0.0f;
		if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.0f)
			rate = 2.0 * (double)(n * 16) * repeats / (ms * 1e-3) / 1e9;
	}
	if (e0)
		(void)hipEventDestroy(e0);
	if (e1)
		(void)hipEventDestroy(e1);
	(void)hipFree(a);
	(void)hipFree(b);
	return rate;
}
