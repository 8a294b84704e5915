// Issue-rate probe 4 (gfx950): does a VALU instruction cost more when its source registers share a register bank (index mod 4)?
// The level-0 kernels complete ~0.25 VALU instructions per cycle and SIMD with three or four waves resident (profiles/r4_queue_mode.txt),
// half of what the probes with one or two distinct sources measured (profiles/r2_valu_issue_rates3.txt).
// build: hipcc --offload-arch=gfx950 -O3 scripts/valu_banks.hip -o scripts/valu_banks.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define R8(X) X X X X X X X X
// explicit registers: v[8..23] accumulators, sources picked per variant
#define BODY(ASM) asm volatile(R8(R8(ASM)) ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31")

template <int OP> __global__ void __launch_bounds__(256) k(float* out, int iters)
{
	asm volatile("v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n"
	             "v_mov_b32 v16, 0.5\n v_mov_b32 v17, 0.5\n v_mov_b32 v18, 0.5\n v_mov_b32 v19, 0.5\n v_mov_b32 v20, 0.5\n v_mov_b32 v21, 0.5\n v_mov_b32 v22, 0.5\n v_mov_b32 v23, 0.5\n"
	             "v_mov_b32 v24, 0.25\n v_mov_b32 v25, 0.25\n v_mov_b32 v26, 0.25\n v_mov_b32 v27, 0.25\n v_mov_b32 v28, 0.25\n v_mov_b32 v29, 0.25\n v_mov_b32 v30, 0.25\n v_mov_b32 v31, 0.25\n" ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
	for (int i = 0; i < iters; i++)
	{
		if (OP == 0)  // add: two sources, different banks (8 = bank 0, 17 = bank 1)
			BODY("v_add_f32 v8, v8, v17\n v_add_f32 v9, v9, v18\n v_add_f32 v10, v10, v19\n v_add_f32 v11, v11, v16\n");
		else if (OP == 1)  // add: two sources, same bank (8, 16 = bank 0)
			BODY("v_add_f32 v8, v8, v16\n v_add_f32 v9, v9, v17\n v_add_f32 v10, v10, v18\n v_add_f32 v11, v11, v19\n");
		else if (OP == 2)  // fma: three sources, three banks
			BODY("v_fma_f32 v8, v8, v17, v26\n v_fma_f32 v9, v9, v18, v27\n v_fma_f32 v10, v10, v19, v24\n v_fma_f32 v11, v11, v16, v25\n");
		else if (OP == 3)  // fma: three sources, one bank
			BODY("v_fma_f32 v8, v8, v16, v24\n v_fma_f32 v9, v9, v17, v25\n v_fma_f32 v10, v10, v18, v26\n v_fma_f32 v11, v11, v19, v27\n");
		else if (OP == 4)  // fma: two of three in one bank
			BODY("v_fma_f32 v8, v8, v16, v25\n v_fma_f32 v9, v9, v17, v26\n v_fma_f32 v10, v10, v18, v27\n v_fma_f32 v11, v11, v19, v24\n");
		else if (OP == 5)  // dependent chain of adds (one accumulator): latency
			BODY("v_add_f32 v8, v8, v17\n v_add_f32 v8, v8, v18\n v_add_f32 v8, v8, v19\n v_add_f32 v8, v8, v16\n");
		else if (OP == 6)  // two accumulators alternating
			BODY("v_add_f32 v8, v8, v17\n v_add_f32 v9, v9, v18\n v_add_f32 v8, v8, v19\n v_add_f32 v9, v9, v16\n");
		else if (OP == 7)  // fmac (VOP2) three banks
			BODY("v_fmac_f32 v8, v17, v26\n v_fmac_f32 v9, v18, v27\n v_fmac_f32 v10, v19, v24\n v_fmac_f32 v11, v16, v25\n");
		else if (OP == 8)  // trunc (VOP1)
			BODY("v_trunc_f32 v8, v17\n v_trunc_f32 v9, v18\n v_trunc_f32 v10, v19\n v_trunc_f32 v11, v16\n");
		else if (OP == 9)  // VOP3 add with neg modifier (as the lifting steps use)
			BODY("v_sub_f32 v8, v8, v17\n v_fma_f32 v9, v9, 0.5, v18\n v_mul_f32 v10, 0.0625, v19\n v_trunc_f32 v11, v16\n");
	}
	float r;
	asm volatile("v_add_f32 %0, v8, v9\n v_add_f32 %0, %0, v10\n v_add_f32 %0, %0, v11" : "=v"(r)::"v8", "v9", "v10", "v11");
	if (r == 12345.678f)
		out[threadIdx.x] = r;
}

template <int OP> static void run(const char* name, int wps)
{
	float* d;
	hipMalloc(&d, 4096);
	const int iters = 2000, blocks = 256 * wps;  // 256 CUs x wps workgroups of 4 waves = wps waves per SIMD
	hipEvent_t a, b;
	hipEventCreate(&a), hipEventCreate(&b);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10);
	hipEventRecord(a);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	const double inst = (double)iters * 256.0 * wps;  // wave-instructions per SIMD
	printf("%-58s waves/SIMD=%d  %.3f ms  %.2f cycles per wave-instruction per SIMD at 2.3 GHz\n", name, wps, ms, ms * 1e-3 * 2.3e9 / inst);
	hipFree(d);
}

int main()
{
	for (int wps : {1, 2, 4})
	{
		run<0>("v_add_f32, sources in two banks", wps);
		run<1>("v_add_f32, both sources in one bank", wps);
		run<2>("v_fma_f32, sources in three banks", wps);
		run<3>("v_fma_f32, all three sources in one bank", wps);
		run<4>("v_fma_f32, two of three sources in one bank", wps);
		run<5>("v_add_f32, one dependent chain", wps);
		run<6>("v_add_f32, two chains alternating", wps);
		run<7>("v_fmac_f32 (VOP2), three banks", wps);
		run<8>("v_trunc_f32", wps);
		run<9>("sub / fma with constant / mul with constant / trunc mix", wps);
	}
	return 0;
}
