// raw buffer stores on gfx950: are out-of-range lanes dropped, do 2-byte aligned dword stores work,
// what happens to a dword that straddles num_records?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint16_t* p, unsigned bytes)
{
	__amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000);
	const unsigned lane = threadIdx.x;
	unsigned off;
	if (lane < 10) off = lane * 4 + 2;            // 2-byte aligned dwords
	else if (lane == 10) off = bytes - 2;          // straddles the end
	else if (lane == 11) off = bytes;              // first byte out of range
	else if (lane == 12) off = bytes - 4;          // last dword fully inside
	else off = 0xFFFFFFFFu;                        // far out of range
	__builtin_amdgcn_raw_buffer_store_b32(0x11110000u * 0 + (0xB000u + lane) | ((0xC000u + lane) << 16), r, off, 0, 0);
}
int main()
{
	const unsigned n = 64;  // uint16 inside the resource; 16 more behind it as a guard zone
	uint16_t* d; uint16_t h[n + 16];
	(void)hipMalloc(&d, sizeof h);
	(void)hipMemset(d, 0xAA, sizeof h);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n * 2);
	(void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
	for (unsigned i = 0; i < n + 16; i++) printf("%04x%s", h[i], (i % 16 == 15) ? "\n" : " ");
	return 0;
}
