// FETCH_SIZE / WRITE_SIZE calibration: kernels that move a KNOWN number of bytes in the access widths
// of the level kernels, each launched once, to be run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE  (and WRITE_SIZE):
// factor = bytes moved / (counter KiB * 1024).  MI355X_MICROARCH.md calibrates 16 bytes per lane only (FETCH_SIZE reports
// half); the inverse kernels read a dword per lane in 256-byte runs of row-strided sub-bands.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/fetch_calib.hip -o /tmp/fc && /tmp/fc
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <typename T>
__global__ void k_calib_read(const T* __restrict__ in, uint32_t* sink, size_t n)  // linear, sizeof(T) bytes per lane
{
	uint32_t acc = 0;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
	{
		const T v = in[i];
		const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
		for (unsigned k = 0; k < sizeof(T) / 4; k++)
			acc ^= w[k];
	}
	if (acc == 0x12345678u)
		sink[0] = acc;
}
// a wave reads 256 bytes (a dword per lane) of a row, then of the next row (pitch bytes further): the inverse kernels' pattern
__global__ void k_calib_read_rows_dword(const uint8_t* __restrict__ in, uint32_t* sink, uint32_t pitch, int rows_per_wave, int strips, uint32_t shift)
{
	const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	const size_t strip = wave % strips, seg = wave / strips;
	const uint8_t* p = in + shift + (seg * rows_per_wave) * (size_t)pitch + strip * 256 + lane * 4;
	uint32_t acc = 0;
	for (int r = 0; r < rows_per_wave; r++)
		acc ^= *reinterpret_cast<const uint32_t*>(p + (size_t)r * pitch);
	if (acc == 0x12345678u)
		sink[0] = acc;
}
template <typename T>
__global__ void k_calib_write(T* __restrict__ out, size_t n)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
	{
		T v;
		uint32_t* w = reinterpret_cast<uint32_t*>(&v);
		for (unsigned k = 0; k < sizeof(T) / 4; k++)
			w[k] = (uint32_t)i + k;
		out[i] = v;
	}
}
__global__ void k_calib_write_rows_dword(uint8_t* __restrict__ out, uint32_t pitch, int rows_per_wave, int strips, uint32_t shift)
{
	const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	const size_t strip = wave % strips, seg = wave / strips;
	uint8_t* p = out + shift + (seg * rows_per_wave) * (size_t)pitch + strip * 256 + lane * 4;
	for (int r = 0; r < rows_per_wave; r++)
		*reinterpret_cast<uint32_t*>(p + (size_t)r * pitch) = (uint32_t)(wave + r);
}

int main()
{
	const size_t bytes = (size_t)1 << 30;
	uint8_t* a;
	uint32_t* o;
	hipMalloc(&a, bytes + 4096), hipMalloc(&o, 64);
	hipMemset(a, 1, bytes + 4096);
	hipDeviceSynchronize();
	// every kernel touches exactly 1 GiB once (a buffer four times the Infinity Cache: nothing is served on-die)
	hipLaunchKernelGGL(k_calib_read<uint4>, dim3(8192), dim3(256), 0, 0, (const uint4*)a, o, bytes / 16);
	hipLaunchKernelGGL(k_calib_read<uint2>, dim3(8192), dim3(256), 0, 0, (const uint2*)a, o, bytes / 8);
	hipLaunchKernelGGL(k_calib_read<uint32_t>, dim3(8192), dim3(256), 0, 0, (const uint32_t*)a, o, bytes / 4);
	// rows of 8 KiB (32 strips of 256 bytes), 64 rows per wave: 1 GiB = 131072 rows -> 2048 segments x 32 strips waves
	for (uint32_t shift : {0u, 2u})
		hipLaunchKernelGGL(k_calib_read_rows_dword, dim3(2048 * 32 / 4), dim3(256), 0, 0, a, o, 8192u, 64, 32, shift);
	hipLaunchKernelGGL(k_calib_write<uint4>, dim3(8192), dim3(256), 0, 0, (uint4*)a, bytes / 16);
	hipLaunchKernelGGL(k_calib_write<uint32_t>, dim3(8192), dim3(256), 0, 0, (uint32_t*)a, bytes / 4);
	for (uint32_t shift : {0u, 2u})
		hipLaunchKernelGGL(k_calib_write_rows_dword, dim3(2048 * 32 / 4), dim3(256), 0, 0, a, 8192u, 64, 32, shift);
	hipDeviceSynchronize();
	printf("each kernel moved %zu bytes\n", bytes);
	return 0;
}
