// Second issue-rate probe (gfx950): the forms left open by scripts/valu_rates.hip -- compares, selects, min/max,
// byte conversions, SDWA conversions, packs -- to decide how the gate / quantize / pack tail and the colour
// transform of the streaming kernels should be spelled.
// build: hipcc --offload-arch=gfx950 -O3 scripts/valu_rates2.hip -o scripts/valu_rates2.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X X X X X X X X
#define ALL8(S) REP8(S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7))

template <int OP> __global__ void __launch_bounds__(256) k_rate(uint32_t* out, int iters)
{
	float a0 = 1.0f + threadIdx.x, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f;
	float a4 = a0 * 2.0f, a5 = a1 * 2.0f, a6 = a2 * 2.0f, a7 = a3 * 2.0f;
	float m = 1.0001f, c = 0.5f;
	for (int i = 0; i < iters; i++)
	{
		if (OP == 0)
		{
#define S(r) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 1)
		{
#define S(r) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(r), "v"(c) : "vcc");
			ALL8(S)
#undef S
		}
		else if (OP == 2)
		{
#define S(r) asm volatile("v_cmp_gt_f32_e64 vcc, |%0|, %1" : : "v"(r), "v"(c) : "vcc");
			ALL8(S)
#undef S
		}
		else if (OP == 3)
		{
#define S(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(c) : "vcc");
			ALL8(S)
#undef S
		}
		else if (OP == 4)
		{
#define S(r) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(r));
			ALL8(S)
#undef S
		}
		else if (OP == 5)
		{
#define S(r) asm volatile("v_cvt_f32_ubyte2 %0, %0" : "+v"(r));
			ALL8(S)
#undef S
		}
		else if (OP == 6)
		{
#define S(r) asm volatile("v_floor_f32 %0, %0" : "+v"(r));
			ALL8(S)
#undef S
		}
		else if (OP == 7)
		{
#define S(r) asm volatile("v_rndne_f32 %0, %0" : "+v"(r));
			ALL8(S)
#undef S
		}
		else if (OP == 8)
		{
#define S(r) asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 9)
		{
#define S(r) asm volatile("v_cvt_i32_f32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 10)
		{
#define S(r) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r));
			ALL8(S)
#undef S
		}
		else if (OP == 11)
		{
#define S(r) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 12)
		{
#define S(r) asm volatile("v_bfe_i32 %0, %0, 4, 16" : "+v"(r));
			ALL8(S)
#undef S
		}
		else if (OP == 13)
		{
#define S(r) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(m));
			ALL8(S)
#undef S
		}
		else if (OP == 14)
		{
#define S(r) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(r) : "v"(c), "v"(m));
			ALL8(S)
#undef S
		}
		else if (OP == 15)
		{
#define S(r) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(r));
			ALL8(S)
#undef S
		}
		else if (OP == 16)
		{
#define S(r) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 17)
		{
#define S(r) asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 18)
		{
#define S(r) asm volatile("v_cvt_f32_i32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 19)
		{
#define S(r) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(m));
			ALL8(S)
#undef S
		}
		else if (OP == 20)
		{
#define S(r) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(r) : "v"(m));
			ALL8(S)
#undef S
		}
		else if (OP == 21)
		{
#define S(r) asm volatile("v_fma_f32 %0, %0, 0.5, 0.5 clamp" : "+v"(r));
			ALL8(S)
#undef S
		}
		else if (OP == 22)
		{
#define S(r) asm volatile("v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 23)
		{
#define S(r) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 24)
		{
#define S(r) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 25)
		{
#define S(r) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 26)
		{
#define S(r) asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(r));
			ALL8(S)
#undef S
		}
		else if (OP == 27)
		{
#define S(r) asm volatile("v_fma_f32 %0, %0, %1, %2 mul:2" : "+v"(r) : "v"(m), "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 28)
		{
#define S(r) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 29)
		{
#define S(r) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r) : "v"(m), "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 30)
		{
#define S(r) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r) : "v"(m), "v"(c));
			ALL8(S)
#undef S
		}
		else if (OP == 31)
		{
#define S(r) asm volatile("v_cvt_i32_f32_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD" : "=v"(r) : "v"(c));
			ALL8(S)
#undef S
		}
	}
	float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
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
	const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 64.0 * waves_per_simd);
	printf("%-28s waves/SIMD=%d  %.3f ms  ~%.2f cycles per wave-instruction (at 2.4 GHz)\n", name, waves_per_simd, ms, cyc);
}

int main()
{
	uint32_t* d_out;
	hipMalloc(&d_out, 4096);
	for (int w = 2; w <= 4; w *= 2)
	{
		run<0>("v_max_f32", d_out, w);
		run<24>("v_min_f32", d_out, w);
		run<1>("v_cmp_gt_f32 e32", d_out, w);
		run<2>("v_cmp_gt_f32 e64 |abs|", d_out, w);
		run<3>("v_cndmask_b32 vcc", d_out, w);
		run<4>("v_cvt_f32_ubyte0", d_out, w);
		run<5>("v_cvt_f32_ubyte2", d_out, w);
		run<6>("v_floor_f32", d_out, w);
		run<7>("v_rndne_f32", d_out, w);
		run<8>("v_cvt_pk_i16_i32", d_out, w);
		run<9>("v_cvt_i32_f32_sdwa WORD_1", d_out, w);
		run<31>("v_cvt_i32_f32_sdwa WORD_0", d_out, w);
		run<18>("v_cvt_f32_i32_sdwa srcW1", d_out, w);
		run<10>("v_lshlrev_b32", d_out, w);
		run<26>("v_lshrrev_b32", d_out, w);
		run<11>("v_or_b32", d_out, w);
		run<25>("v_xor_b32", d_out, w);
		run<12>("v_bfe_i32", d_out, w);
		run<13>("v_med3_f32", d_out, w);
		run<14>("v_max3_f32 abs", d_out, w);
		run<15>("v_mul_f32 inline 0.5", d_out, w);
		run<16>("v_sub_u32", d_out, w);
		run<17>("v_mov_b32", d_out, w);
		run<19>("v_and_or_b32", d_out, w);
		run<20>("v_mul_legacy_f32", d_out, w);
		run<21>("v_fma_f32 clamp", d_out, w);
		run<22>("v_add_f32_sdwa", d_out, w);
		run<23>("v_cvt_pk_u8_f32", d_out, w);
		run<27>("v_fma_f32 omod mul:2", d_out, w);
		run<28>("v_add_f32_dpp wave_shr", d_out, w);
		run<29>("v_fmac_f32_dpp row_shr", d_out, w);
		run<30>("v_bfi_b32", d_out, w);
	}
	return 0;
}
