// Practical HBM rates of the box: linear read-only, write-only and copy kernels over 1 GiB (16 bytes per lane and
// access, grid-stride), and the same with the row-strided pattern of the level kernels (1 KiB per wave and row).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/hbm_rates.hip -o /tmp/hbm && /tmp/hbm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_read(const uint4* __restrict__ in, uint32_t* out, size_t n)
{
	uint32_t acc = 0;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
	{
		const uint4 v = in[i];
		acc ^= v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678u)
		out[0] = acc;
}
__global__ void k_write(uint4* __restrict__ out, size_t n)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		out[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		out[i] = in[i];
}
// tuned copy: U independent 16-byte loads in flight per lane before the first store, 16-byte stores,
// each workgroup on a contiguous chunk (no grid-stride hop between a lane's accesses: consecutive 4 KiB pieces)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ void k_copy_tuned(const uint4* __restrict__ in4, uint4* __restrict__ out4, size_t n)
{
	const u32x4* in = reinterpret_cast<const u32x4*>(in4);
	u32x4* out = reinterpret_cast<u32x4*>(out4);
	const size_t per_block = (size_t)blockDim.x * U;
	for (size_t base = blockIdx.x * per_block; base < n; base += (size_t)gridDim.x * per_block)
	{
		u32x4 v[U];
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
			v[u] = (i < n) ? __builtin_nontemporal_load(&in[i]) : u32x4{0, 0, 0, 0};
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
template <int U>
__global__ void k_copy_tuned_plain(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n)
{
	const size_t per_block = (size_t)blockDim.x * U;
	for (size_t base = blockIdx.x * per_block; base < n; base += (size_t)gridDim.x * per_block)
	{
		uint4 v[U];
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
			v[u] = (i < n) ? in[i] : make_uint4(0, 0, 0, 0);
		}
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
			if (i < n)
				out[i] = v[u];
		}
	}
}

// one wave reads 1 KiB of a row, then the same 1 KiB of the next row (pitch bytes further), like a strip walk
__global__ void k_read_strips(const uint4* __restrict__ in, uint32_t* out, size_t pitch16, int rows_per_wave, int strips)
{
	const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	const size_t strip = wave % strips, seg = wave / strips;
	const uint4* p = in + (seg * rows_per_wave) * pitch16 + strip * 64 + lane;
	uint32_t acc = 0;
	for (int r = 0; r < rows_per_wave; r++)
	{
		const uint4 v = p[(size_t)r * pitch16];
		acc ^= v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678u)
		out[0] = acc;
}

template <typename F> static double time_ms(F&& launch)
{
	hipEvent_t a, b;
	hipEventCreate(&a), hipEventCreate(&b);
	launch();
	hipEventRecord(a, 0);
	for (int i = 0; i < 10; i++)
		launch();
	hipEventRecord(b, 0);
	hipEventSynchronize(b);
	float ms = 0;
	hipEventElapsedTime(&ms, a, b);
	return ms / 10;
}

int main()
{
	const size_t bytes = (size_t)1 << 30, n = bytes / 16;
	uint4 *a, *b;
	uint32_t* o;
	hipMalloc(&a, bytes), hipMalloc(&b, bytes), hipMalloc(&o, 64);
	hipMemset(a, 1, bytes), hipMemset(b, 2, bytes);
	for (int blocks : {2048, 8192, 32768})
	{
		const double r = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, o, n); });
		const double w = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, b, n); });
		const double c = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); });
		printf("blocks %6d  read %7.1f GB/s  write %7.1f GB/s  copy %7.1f GB/s (read + write)\n", blocks, bytes / r / 1e6, bytes / w / 1e6,
		       2.0 * bytes / c / 1e6);
	}
	for (int blocks : {1024, 2048, 4096, 16384})
	{
		const double c4 = time_ms([&] { hipLaunchKernelGGL(k_copy_tuned_plain<4>, dim3(blocks), dim3(256), 0, 0, a, b, n); });
		const double c8 = time_ms([&] { hipLaunchKernelGGL(k_copy_tuned_plain<8>, dim3(blocks), dim3(256), 0, 0, a, b, n); });
		const double n4 = time_ms([&] { hipLaunchKernelGGL(k_copy_tuned<4>, dim3(blocks), dim3(256), 0, 0, a, b, n); });
		const double n8 = time_ms([&] { hipLaunchKernelGGL(k_copy_tuned<8>, dim3(blocks), dim3(256), 0, 0, a, b, n); });
		printf("tuned copy, blocks %6d: 4 loads in flight %7.1f, 8 loads %7.1f; non-temporal 4 loads %7.1f, 8 loads %7.1f GB/s (read + write)\n", blocks,
		       2.0 * bytes / c4 / 1e6, 2.0 * bytes / c8 / 1e6, 2.0 * bytes / n4 / 1e6, 2.0 * bytes / n8 / 1e6);
	}
	{
		const double m = time_ms([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
		printf("hipMemcpyAsync device to device: %7.1f GB/s (read + write)\n", 2.0 * bytes / m / 1e6);
	}
	// strip walk: an "image" of 8192 rows x 32 KiB (256 MiB), 32 strips of 1 KiB, R rows per wave
	for (int rows : {16, 64, 256})
	{
		const int strips = 32, waves = strips * (8192 / rows);
		const double t = time_ms([&] { hipLaunchKernelGGL(k_read_strips, dim3(waves / 4), dim3(256), 0, 0, a, o, (size_t)2048, rows, strips); });
		printf("strip walk, %3d rows per wave (%5d waves): read %7.1f GB/s\n", rows, waves, 8192.0 * 32768.0 / t / 1e6);
	}
	return 0;
}
