// what v_cvt_pk_u8_f32 does with out-of-range / fractional / special inputs on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* a, unsigned* o, int n)
{
	int i = threadIdx.x;
	if (i < n)
		o[i] = __builtin_amdgcn_cvt_pk_u8_f32(a[i], 1, 0xAABBCCDDu);
}
int main()
{
	const float h[] = {0.f, 1.f, 254.f, 255.f, 256.f, 300.f, 1e9f, -1.f, -0.f, -300.f, -1e9f, 0.5f, 1.5f, 2.5f, 254.5f, 255.5f, INFINITY, -INFINITY, NAN, 32767.f, -32768.f};
	const int n = sizeof h / sizeof h[0];
	float* d; unsigned* o; unsigned r[64];
	(void)hipMalloc(&d, sizeof h); (void)hipMalloc(&o, sizeof r);
	(void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
	(void)hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
	for (int i = 0; i < n; i++) printf("%12g -> byte %3u  (dword %08x)\n", h[i], (r[i] >> 8) & 255, r[i]);
	return 0;
}
