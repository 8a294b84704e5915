// Third issue-rate probe (gfx950): packed fp32 (v_pk_add / mul / fma on register pairs) against the plain forms, with
// operands that keep changing (the second probe showed 2.6 cycles for forms whose result soon stops changing and
// 4.3-5 for the others).  Decides whether the two planes a wave of the u8 kernels carries should travel as float2.
// build: hipcc --offload-arch=gfx950 -O3 scripts/valu_rates3.hip -o /tmp/valu_rates3 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X X X X X X X X
#define ALL8(S) REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#define ALL4(S) REP8(S(d0) S(d1) S(d2) S(d3) S(d0) S(d1) S(d2) S(d3))

template <int OP> __global__ void __launch_bounds__(256) k_rate(uint32_t* out, int iters)
{
	float a0 = 1.0f + threadIdx.x, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f;
	float a4 = a0 * 2.0f, a5 = a1 * 2.0f, a6 = a2 * 2.0f, a7 = a3 * 2.0f;
	double d0 = a0, d1 = a1, d2 = a2, d3 = a3;  // register pairs: two floats each as far as the packed ops care
	double dm = 1.0000001, dc = 0.5000001;
	float m = 1.0001f, c = 0.37f;
	for (int i = 0; i < iters; i++)
	{
		if (OP == 0)
		{
#define S(r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 1)
		{
#define S(r) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r) : "v"(dc));
			ALL4(S)
#undef S
		}
		else if (OP == 2)
		{
#define S(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(m), "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 3)
		{
#define S(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(dm), "v"(dc));
			ALL4(S)
#undef S
		}
		else if (OP == 4)
		{
#define S(r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r) : "v"(m));
			ALL8(S)
#undef S
		}
		else if (OP == 5)
		{
#define S(r) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r) : "v"(dm));
			ALL4(S)
#undef S
		}
		else if (OP == 6)
		{
#define S(r) asm volatile("v_trunc_f32 %0, %0\n v_add_f32 %0, %0, %1" : "+v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 7)
		{
#define S(r) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 8)
		{
#define S(r) asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]" : "+v"(r) : "v"(dc));
			ALL4(S)
#undef S
		}
		else if (OP == 9)
		{
#define S(r) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1" : "+v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 10)
		{
#define S(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 11)
		{
#define S(r) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
	}
	float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3);
	if (s == 12345.678f)
		out[threadIdx.x] = (uint32_t)s;
}

template <int OP> void run(const char* name, uint32_t* d_out, int waves_per_simd)
{
	const int iters = 4000;
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	const int blocks = 256 * waves_per_simd;
	hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 10);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const int per_iter = (OP == 6) ? 128 : 64;
	const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * per_iter * waves_per_simd);
	printf("%-28s waves/SIMD=%d  %.3f ms  ~%.2f cycles per wave-instruction (at 2.4 GHz)\n", name, waves_per_simd, ms, cyc);
}

int main()
{
	uint32_t* d_out;
	hipMalloc(&d_out, 4096);
	for (int w = 2; w <= 4; w *= 2)
	{
		run<0>("v_add_f32", d_out, w);
		run<1>("v_pk_add_f32", d_out, w);
		run<7>("v_sub_f32", d_out, w);
		run<2>("v_fma_f32", d_out, w);
		run<3>("v_pk_fma_f32", d_out, w);
		run<4>("v_mul_f32", d_out, w);
		run<5>("v_pk_mul_f32", d_out, w);
		run<6>("v_trunc_f32 + v_add_f32", d_out, w);
		run<8>("v_pk_mov_b32", d_out, w);
		run<9>("v_mov_b32_dpp wave_shr", d_out, w);
		run<10>("v_add_u32", d_out, w);
		run<11>("v_pk_add_i16", d_out, w);
	}
	return 0;
}
