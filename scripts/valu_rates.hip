// Issue-rate probe for the VALU forms the lifting kernels could be built on (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 scripts/valu_rates.hip -o /tmp/valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X X X X X X X X

template <int OP> __global__ void __launch_bounds__(256) k_rate(uint32_t* out, int iters)
{
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 a0 = {1.0f + threadIdx.x, 2.0f}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f;
	f2 a4 = a0 * 2.0f, a5 = a1 * 2.0f, a6 = a2 * 2.0f, a7 = a3 * 2.0f;
	f2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
	for (int i = 0; i < iters; i++)
	{
		if (OP == 0)  // v_fma_f32 (one per half: 16 instr)
		{
#define S(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(m.x), "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 1)  // v_pk_fma_f32
		{
#define S(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(m), "v"(c));
			REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
		}
		else if (OP == 2)  // v_pk_add_f32
		{
#define S(r) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r) : "v"(c));
			REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
		}
		else if (OP == 3)  // v_pk_add_i16
		{
#define S(r) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(r) : "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 4)  // v_pk_ashrrev_i16
		{
#define S(r) asm volatile("v_pk_ashrrev_i16 %0, 1, %0" : "+v"(r));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 5)  // v_pk_mad_i16
		{
#define S(r) asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(r) : "v"(m.x), "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 6)  // v_add_u32
		{
#define S(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 7)  // v_mov_b32 dpp wave_shr
		{
#define S(r) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 8)  // v_add_u32 with dpp row_shr (fused shift+add)
		{
#define S(r) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r) : "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 9)  // v_lshl_add_u32
		{
#define S(r) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(r) : "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 10)  // v_add3_u32
		{
#define S(r) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(c.x), "v"(m.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 11)  // v_trunc_f32
		{
#define S(r) asm volatile("v_trunc_f32 %0, %0" : "+v"(r));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 12)  // v_pk_mul_f32
		{
#define S(r) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r) : "v"(m));
			REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))
#undef S
		}
		else if (OP == 13)  // v_cvt_i32_f32
		{
#define S(r) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 14)  // v_pk_max_i16
		{
#define S(r) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r) : "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 15)  // v_pk_mul_lo_u16
		{
#define S(r) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(r) : "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 16)  // v_mov_b32 dpp row_shr (no cross-row)
		{
#define S(r) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 17)
		{
#define S(r) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(r));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 18)
		{
#define S(r) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r) : "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 19)
		{
#define S(r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 20)
		{
#define S(r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r) : "v"(m.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 21)
		{
#define S(r) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(r) : "v"(m.x), "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 22)
		{
#define S(r) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(r) : "v"(m.x), "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 23)
		{
#define S(r) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 24)
		{
#define S(r) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r) : "v"(m.x), "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 25)
		{
#define S(r) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r) : "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 26)
		{
#define S(r) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 27)
		{
#define S(r) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 28)
		{
#define S(r) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(m.x), "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 29)
		{
#define S(r) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(r) : "v"(m.x), "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 30)
		{
#define S(r) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r) : "v"(c.x));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
		else if (OP == 31)
		{
#define S(r) asm volatile("v_add_f32 %0, 0x41000000, %0" : "+v"(r));
			REP8(S(a0.x) S(a1.x) S(a2.x) S(a3.x) S(a4.x) S(a5.x) S(a6.x) S(a7.x))
#undef S
		}
	}
	f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
	if (s.x == 12345.678f)
		out[threadIdx.x] = (uint32_t)s.y;
}

template <int OP> void run(const char* name, uint32_t* d_out, int waves_per_simd)
{
	const int iters = 4000;
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	const int blocks = 256 * waves_per_simd;  // 256-thread blocks: 4 waves = 1 per SIMD of a CU
	hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 10);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double instr_per_wave = (double)iters * 64.0;
	const double clk = 2.4e9;  // nominal; the ratio between rows is what matters
	const double cyc = ms * 1e-3 * clk / (instr_per_wave * waves_per_simd);
	printf("%-22s waves/SIMD=%d  %.3f ms  ~%.2f cycles per wave-instruction (at 2.4 GHz)\n", name, waves_per_simd, ms, cyc);
}

int main()
{
	uint32_t* d_out;
	hipMalloc(&d_out, 4096);
	for (int w = 1; w <= 8; w *= 2)
	{
		run<0>("v_fma_f32", d_out, w);
		run<1>("v_pk_fma_f32", d_out, w);
		run<2>("v_pk_add_f32", d_out, w);
		run<12>("v_pk_mul_f32", d_out, w);
		run<3>("v_pk_add_i16", d_out, w);
		run<4>("v_pk_ashrrev_i16", d_out, w);
		run<5>("v_pk_mad_i16", d_out, w);
		run<14>("v_pk_max_i16", d_out, w);
		run<15>("v_pk_mul_lo_u16", d_out, w);
		run<6>("v_add_u32", d_out, w);
		run<9>("v_lshl_add_u32", d_out, w);
		run<10>("v_add3_u32", d_out, w);
		run<7>("v_mov_dpp wave_shr", d_out, w);
		run<16>("v_mov_dpp row_shr", d_out, w);
		run<8>("v_add_u32_dpp row_shr", d_out, w);
		run<11>("v_trunc_f32", d_out, w);
		run<13>("v_cvt_i32_f32", d_out, w);
		run<17>("v_ashrrev_i32", d_out, w);
		run<18>("v_max_i32", d_out, w);
		run<19>("v_add_f32", d_out, w);
		run<20>("v_mul_f32", d_out, w);
		run<21>("v_mad_i32_i24", d_out, w);
		run<22>("v_med3_i32", d_out, w);
		run<23>("v_and_b32", d_out, w);
		run<24>("v_fmac_f32", d_out, w);
		run<25>("v_add_f32_dpp row_shr", d_out, w);
		run<26>("v_cvt_f32_i32", d_out, w);
		run<27>("v_mul_lo_u32", d_out, w);
		run<28>("v_perm_b32", d_out, w);
		run<29>("v_fma_f32 neg mod", d_out, w);
		run<30>("v_sub_f32", d_out, w);
		run<31>("v_add_f32 sdwa-free literal", d_out, w);
	}
	return 0;
}
