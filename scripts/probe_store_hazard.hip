// probe_store_hazard.hip -- does a VALU write of a buffer_store_dwordx4's data register, issued AT ONCE behind the store,
// reach memory instead of the stored value on gfx950?  (DESIGN.md 4.1 "a hardware hazard the compiler does not know":
// seen once in round 3 with a scalar-offset store, worked around by s_nop 1 behind every 16-byte store.)  LLVM's hazard
// table has the case only for stores WITHOUT a scalar offset register.  Both forms, with and without the guard:
//   hipcc --offload-arch=gfx950 -O2 scripts/probe_store_hazard.hip -o scripts/probe_store_hazard.bin ; run once on the box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int GUARD, int SOFF>  // GUARD: s_nop 1 between store and overwrite; SOFF: row offset in an SGPR (1) or folded into voffset (0)
__global__ void k(uint32_t* out, int rows)
{
	const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	u32x4 rs;
	const uint64_t base = (uint64_t)(out + (size_t)wave * rows * 256);
	rs.x = __builtin_amdgcn_readfirstlane((uint32_t)base), rs.y = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32) & 0xffff);
	rs.z = 0xffffffffu, rs.w = 0x00020000u;
	for (int r = 0; r < rows; r++)
	{
		const uint32_t val = (wave << 16) ^ (r << 8) ^ lane ^ 0x5a000000u;
		const uint32_t so = SOFF ? (uint32_t)r * 1024u : 0u, vo = lane * 16u + (SOFF ? 0u : (uint32_t)r * 1024u);
#define AKO_PROBE(G, SO)                                                                                                            \
	asm volatile("v_mov_b32 v20, %0\n\tv_add_u32 v21, 1, %0\n\tv_add_u32 v22, 2, %0\n\tv_add_u32 v23, 3, %0\n\ts_nop 4\n\t"           \
	             "buffer_store_dwordx4 v[20:23], %1, %2, " SO " offen\n\t" G                                                          \
	             "v_mov_b32 v20, 0xdeadbeef\n\tv_mov_b32 v21, 0xdeadbeef\n\tv_mov_b32 v22, 0xdeadbeef\n\tv_mov_b32 v23, 0xdeadbeef" \
	             ::"v"(val), "v"(vo), "s"(rs), "s"(so) : "v20", "v21", "v22", "v23", "memory")
		if constexpr (GUARD && SOFF)
			AKO_PROBE("s_nop 1\n\t", "%3");
		else if constexpr (SOFF)
			AKO_PROBE("", "%3");
		else if constexpr (GUARD)
			AKO_PROBE("s_nop 1\n\t", "0");
		else
			AKO_PROBE("", "0");
	}
}
template <int GUARD, int SOFF>
static void run(const char* name)
{
	const int blocks = 2048, threads = 256, rows = 64;  // 8192 waves x 64 rows x 1 KiB
	const size_t n = (size_t)blocks * (threads / 64) * rows * 256;
	uint32_t* d;
	hipMalloc(&d, n * 4);
	size_t bad = 0, beef = 0;
	for (int rep = 0; rep < 8; rep++)
	{
		hipMemset(d, 0, n * 4);
		hipLaunchKernelGGL((k<GUARD, SOFF>), dim3(blocks), dim3(threads), 0, 0, d, rows);
		std::vector<uint32_t> h(n);
		hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
		for (size_t i = 0; i < n; i++)
		{
			const uint32_t wave = (uint32_t)(i / (rows * 256)), r = (uint32_t)(i / 256 % rows), lane = (uint32_t)(i % 256 / 4), j = (uint32_t)(i % 4);
			const uint32_t want = ((wave << 16) ^ (r << 8) ^ lane ^ 0x5a000000u) + j;
			bad += h[i] != want, beef += h[i] == 0xdeadbeefu;
		}
	}
	printf("%-58s wrong dwords %zu of %zu (0xdeadbeef: %zu)\n", name, bad, n * 8, beef);
	hipFree(d);
}
int main()
{
	run<0, 1>("scalar row offset, overwrite at once (no guard)");
	run<1, 1>("scalar row offset, s_nop 1 between");
	run<0, 0>("no scalar offset, overwrite at once (LLVM's known hazard)");
	run<1, 0>("no scalar offset, s_nop 1 between");
	return 0;
}
