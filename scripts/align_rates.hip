// Do dword loads / stores that are only 2-byte aligned cost anything?  The [head][C][B][D] groups of a coefficient stream
// start one int16 after the previous group, so every other plane's sub-band rows are 2-byte aligned only, and the
// streaming kernels read / write them as dwords (two int16 columns per lane, 256 B per wave and row).  Row-strided
// pattern of the level kernels: a wave touches 256 B of a row, then the same columns of the next row.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/align_rates.hip -o /tmp/align && /tmp/align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <bool STORE, int SYNC = 0>
__global__ void k_rows(uint8_t* base, uint32_t* sink, uint32_t pitch, int rows_per_wave, int strips, uint32_t shift)
{
	const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	const size_t strip = wave % strips, seg = wave / strips;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + seg * rows_per_wave * (size_t)pitch, 0, (int)0xFFFFFFFFu, 0x00020000);
	const uint32_t lane_off = (uint32_t)(strip * 240 + lane * 4) + shift;  // strips of 120 columns like the kernels'
	uint32_t acc = 0;
	for (int r = 0; r < rows_per_wave; r++)
	{
		if (SYNC && (r % SYNC) == 0)
			__builtin_amdgcn_s_barrier();  // the waves of a workgroup (neighbouring strips) stay on the same rows
		if (STORE)
			__builtin_amdgcn_raw_buffer_store_b32(acc + r, rs, lane_off, r * pitch, 0);
		else
			acc ^= __builtin_amdgcn_raw_buffer_load_b32(rs, lane_off, r * pitch, 0);
	}
	if (!STORE && acc == 0x12345678u)
		sink[0] = acc;
}

// the same rows fetched with wider per-lane loads: a wave covers W * 64 bytes of a row per instruction (256 B, 512 B, 1 KB)
template <int W>
__global__ void k_rows_wide(uint8_t* base, uint32_t* sink, uint32_t pitch, int rows_per_wave, int strips, uint32_t shift)
{
	const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	const size_t strip = wave % strips, seg = wave / strips;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + seg * rows_per_wave * (size_t)pitch, 0, (int)0xFFFFFFFFu, 0x00020000);
	const uint32_t lane_off = (uint32_t)(strip * 64 * W + lane * W) + shift;
	uint32_t acc = 0;
	for (int r = 0; r < rows_per_wave; r++)
	{
		if (W == 4)
			acc ^= __builtin_amdgcn_raw_buffer_load_b32(rs, lane_off, r * pitch, 0);
		else if (W == 8)
		{
			const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_off, r * pitch, 0);
			acc ^= v[0] ^ v[1];
		}
		else
		{
			const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, r * pitch, 0);
			acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
		}
	}
	if (acc == 0x12345678u)
		sink[0] = acc;
}

// 8 B per lane, but even lanes read one array and odd lanes another (a sub-band 'other' bytes further on): per array the
// wave still covers 256 B of a row -- is it the bytes per lane or the length of the run that the rate follows?
template <bool STORE = false>
__global__ void k_rows_split(uint8_t* base, uint32_t* sink, uint32_t pitch, int rows_per_wave, int strips, uint32_t shift, uint32_t other)
{
	const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	const size_t strip = wave % strips, seg = wave / strips;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + seg * rows_per_wave * (size_t)pitch, 0, (int)0xFFFFFFFFu, 0x00020000);
	const uint32_t lane_off = (uint32_t)(strip * 256 + (lane >> 1) * 8) + shift + ((lane & 1) ? other : 0u);
	uint32_t acc = 0;
	for (int r = 0; r < rows_per_wave; r++)
	{
		if (STORE)
		{
			typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
			__builtin_amdgcn_raw_buffer_store_b64(u32x2{acc + r, acc}, rs, lane_off, r * pitch, 0);
		}
		else
		{
			const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_off, r * pitch, 0);
			acc ^= v[0] ^ v[1];
		}
	}
	if (!STORE && acc == 0x12345678u)
		sink[0] = acc;
}

template <typename F> static double time_ms(F&& launch)
{
	hipEvent_t a, b;
	(void)hipEventCreate(&a), (void)hipEventCreate(&b);
	launch();
	(void)hipEventRecord(a, 0);
	for (int i = 0; i < 10; i++)
		launch();
	(void)hipEventRecord(b, 0);
	(void)hipEventSynchronize(b);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, a, b);
	return ms / 10;
}

int main()
{
	const uint32_t pitch = 8192;          // level-0 sub-band row of the 8192 x 8192 image: 4096 int16
	const int rows = 4096 * 12, strips = 34, rows_per_wave = 48;
	const size_t bytes = (size_t)rows * pitch + 4096;
	uint8_t* buf;
	uint32_t* sink;
	(void)hipMalloc(&buf, bytes), (void)hipMalloc(&sink, 64);
	(void)hipMemset(buf, 1, bytes);
	const size_t waves = (size_t)strips * (rows / rows_per_wave);
	const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
	const double moved = (double)waves * rows_per_wave * 256;
	for (int rep = 0; rep < 2; rep++)
		for (uint32_t shift : {0u, 2u, 4u, 6u, 64u, 66u})
		{
			const double l = time_ms([&] { hipLaunchKernelGGL(k_rows<false>, grid, block, 0, 0, buf, sink, pitch, rows_per_wave, strips, shift); });
			const double s = time_ms([&] { hipLaunchKernelGGL(k_rows<true>, grid, block, 0, 0, buf, sink, pitch, rows_per_wave, strips, shift); });
			printf("byte shift %3u: dword loads %.3f ms (%.0f GB/s)   dword stores %.3f ms (%.0f GB/s)\n", shift, l, moved / l / 1e6, s, moved / s / 1e6);
		}
	{
		// same bytes per launch: 32 strips of 256 B, 16 of 512 B, 8 of 1 KB per 8 KB row
		const size_t segs = rows / rows_per_wave;
		const double mv = (double)segs * rows_per_wave * 8192;
		for (uint32_t shift : {0u, 2u})
		{
			const double t4 = time_ms([&] { hipLaunchKernelGGL((k_rows_wide<4>), dim3((unsigned)(segs * 32 / 4)), block, 0, 0, buf, sink, pitch, rows_per_wave, 32, shift); });
			const double t8 = time_ms([&] { hipLaunchKernelGGL((k_rows_wide<8>), dim3((unsigned)(segs * 16 / 4)), block, 0, 0, buf, sink, pitch, rows_per_wave, 16, shift); });
			const double t16 = time_ms([&] { hipLaunchKernelGGL((k_rows_wide<16>), dim3((unsigned)(segs * 8 / 4)), block, 0, 0, buf, sink, pitch, rows_per_wave, 8, shift); });
			printf("whole rows, byte shift %u: 4 B per lane %.0f GB/s, 8 B per lane %.0f GB/s, 16 B per lane %.0f GB/s\n", shift, mv / t4 / 1e6, mv / t8 / 1e6, mv / t16 / 1e6);
		}
	}
	{
		// the first half of the buffer is array X, the second half array Y: 32 strips of 256 B per row of each, 2 x 256 B per instruction
		const size_t segs = rows / 2 / rows_per_wave;
		const double mv = (double)segs * rows_per_wave * 8192 * 2;
		const uint32_t other = (uint32_t)((size_t)rows / 2 * pitch);
		for (uint32_t shift : {0u, 2u})
		{
			const double t = time_ms([&] { hipLaunchKernelGGL(k_rows_split<false>, dim3((unsigned)(segs * 32 / 4)), block, 0, 0, buf, sink, pitch, rows_per_wave, 32, shift, other); });
			const double ts = time_ms([&] { hipLaunchKernelGGL(k_rows_split<true>, dim3((unsigned)(segs * 32 / 4)), block, 0, 0, buf, sink, pitch, rows_per_wave, 32, shift, other); });
			printf("8 B per lane, even / odd lanes on two arrays (256 B runs), byte shift %u: loads %.0f GB/s, stores %.0f GB/s\n", shift, mv / t / 1e6, mv / ts / 1e6);
		}
	}
	// neighbouring strips kept on the same rows: workgroups of 4, 8, 16 waves with a barrier every 1 / 6 rows
	for (int wg : {256, 512, 1024})
	{
		const dim3 g2((unsigned)((waves * 64 + wg - 1) / wg)), b2(wg);
		const double l0 = time_ms([&] { hipLaunchKernelGGL((k_rows<false, 0>), g2, b2, 0, 0, buf, sink, pitch, rows_per_wave, strips, 2u); });
		const double l1 = time_ms([&] { hipLaunchKernelGGL((k_rows<false, 1>), g2, b2, 0, 0, buf, sink, pitch, rows_per_wave, strips, 2u); });
		const double l6 = time_ms([&] { hipLaunchKernelGGL((k_rows<false, 6>), g2, b2, 0, 0, buf, sink, pitch, rows_per_wave, strips, 2u); });
		const double s0 = time_ms([&] { hipLaunchKernelGGL((k_rows<true, 0>), g2, b2, 0, 0, buf, sink, pitch, rows_per_wave, strips, 2u); });
		const double s6 = time_ms([&] { hipLaunchKernelGGL((k_rows<true, 6>), g2, b2, 0, 0, buf, sink, pitch, rows_per_wave, strips, 2u); });
		printf("workgroup of %2d waves: loads free %.0f, barrier every row %.0f, every 6 rows %.0f GB/s;  stores free %.0f, every 6 rows %.0f GB/s\n",
		       wg / 64, moved / l0 / 1e6, moved / l1 / 1e6, moved / l6 / 1e6, moved / s0 / 1e6, moved / s6 / 1e6);
	}
	return 0;
}
