// What the level-0 forward kernel's LOAD pattern alone can reach, and which of its properties costs what: a wave reads
// 1 KiB (16 bytes per lane) of two adjacent image rows per row slot, walks `slots` slots down a strip, keeps RING slots of
// loads in flight; strips are `stride` bytes apart (the kernel: 960, overlapping by 64 bytes; aligned: 1024); `dup` waves
// read the same strip (the kernel's pair of waves); the grid is `rounds` rounds of 4096 resident waves.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/l0_read_pattern.hip -o /tmp/l0r && /tmp/l0r
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int RING, bool NT>
__global__ __launch_bounds__(256) void k_walk(const uint8_t* __restrict__ img, uint32_t* sink, uint32_t pitch, int strips, int segs, int slots,
                                              uint32_t stride, int dup, int seg_rows)
{
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	uint32_t u = wave / dup;
	const uint32_t strip = u % strips, seg = u / strips;
	if (seg >= (uint32_t)segs)
		return;
	const uint8_t* p = img + (size_t)seg * seg_rows * 2 * pitch + (size_t)strip * stride + lane * 16;
	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	u32x4 ring[RING][2];
	auto fetch = [&](int v, u32x4* r) {
		for (int par = 0; par < 2; par++)
		{
			const u32x4* q = reinterpret_cast<const u32x4*>(p + (size_t)(2 * v + par) * pitch);
			r[par] = NT ? __builtin_nontemporal_load(q) : *q;
		}
	};
#pragma unroll
	for (int k = 0; k < RING; k++)
		fetch(k, ring[k]);
	uint32_t acc = 0;
	for (int base = 0; base < slots; base += RING)
	{
#pragma unroll
		for (int k = 0; k < RING; k++)
		{
			const u32x4 a = ring[k][0], b = ring[k][1];
			acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
			fetch(base + k + RING, ring[k]);  // (reads up to RING slots past the segment: rows of the next segment)
		}
	}
	if (acc == 0x12345678u)
		sink[0] = acc;
}

// The same walk WITH the kernel's stores: per slot and wave 2 planes x (LL, C, B, D) rows of 256 bytes (a dword per lane),
// sub-bands 32 MiB apart.  WIDE = 0: eight 4-byte-per-lane stores (the kernel); 1: two 16-byte-per-lane stores whose lanes
// 4k + j write sub-band j (what a 4 x 4 lane transpose in front of the stores would give); 2: no stores.
template <int RING, int WIDE, bool LOADS, int SHAPE = 0>
__global__ __launch_bounds__(256) void k_walk_rw(const uint8_t* __restrict__ img, uint8_t* __restrict__ out, uint32_t pitch, int strips, int segs, int slots,
                                                 uint32_t stride, int seg_rows)
{
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	const uint32_t u = wave >> 1, role = wave & 1;
	const uint32_t strip = u % strips, seg = u / strips;
	if (seg >= (uint32_t)segs)
		return;
	const uint8_t* p = img + (size_t)seg * seg_rows * 2 * pitch + (size_t)strip * stride + lane * 16;
	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	u32x4 ring[RING][2];
	auto fetch = [&](int v, u32x4* r) {
		for (int par = 0; par < 2; par++)
			r[par] = LOADS ? *reinterpret_cast<const u32x4*>(p + (size_t)(2 * v + par) * pitch) : u32x4{(uint32_t)v, 1u, 2u, 3u};
	};
#pragma unroll
	for (int k = 0; k < RING; k++)
		fetch(k, ring[k]);
	const size_t sub = (size_t)4096 * 4096 * 2;  // bytes of a sub-band (4096 x 4096 int16)
	const uint32_t row_b = 4096 * 2;
	for (int base = 0; base < slots; base += RING)
	{
#pragma unroll
		for (int k = 0; k < RING; k++)
		{
			const u32x4 a = ring[k][0], b = ring[k][1];
			fetch(base + k + RING, ring[k]);
			const size_t r = (size_t)seg * seg_rows + base + k;
			for (int pl = 0; pl < 2; pl++)
			{
				uint8_t* plane = out + (size_t)(2 * pl + role) * 4 * sub + r * row_b + strip * (SHAPE == 2 ? 256 : 240) + ((SHAPE == 1 && role) ? 2 : 0);
				const u32x4 v = pl ? b : a;
				if (SHAPE == 1 && lane >= 60)
					continue;
				if (SHAPE == 3)
				{
					// what an LDS exchange inside a two-strip workgroup would give: the wave of the even strip stores sub-bands
					// 0, 1 of BOTH strips (480 contiguous bytes, 8 bytes per lane), the wave of the odd strip sub-bands 2, 3
					if (lane < 60)
					{
						uint8_t* pp = out + (size_t)(2 * pl + role) * 4 * sub + r * row_b + (strip & ~1u) * 240 + (role ? 2 : 0) + lane * 8;
						const int s0 = (strip & 1) ? 2 : 0;
						{ const uint2 w = make_uint2(v.x, v.y); __builtin_memcpy(pp + (size_t)s0 * sub, &w, 8); }
						{ const uint2 w = make_uint2(v.z, v.w); __builtin_memcpy(pp + (size_t)(s0 + 1) * sub, &w, 8); }
					}
					continue;
				}
				if (WIDE == 0)
				{
					*reinterpret_cast<uint32_t*>(plane + 0 * sub + lane * 4) = v.x;
					*reinterpret_cast<uint32_t*>(plane + 1 * sub + lane * 4) = v.y;
					*reinterpret_cast<uint32_t*>(plane + 2 * sub + lane * 4) = v.z;
					*reinterpret_cast<uint32_t*>(plane + 3 * sub + lane * 4) = v.w;
				}
				else if (WIDE == 1)
					*reinterpret_cast<u32x4*>(plane + (size_t)(lane & 3) * sub + (lane >> 2) * 16) = v;
			}
		}
	}
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

template <int RING, bool NT>
static void run(const uint8_t* img, uint32_t* sink, const char* what, uint32_t stride, int dup, int rounds)
{
	// an 8192 x 8192 RGBA image: 8192 rows of 32 KiB = 4096 slots; strips across 32 KiB; segments so that waves = rounds x 4096
	const uint32_t pitch = 32768;
	const int strips = (stride == 1024) ? 32 : 35;
	const int waves = rounds * 4096;
	const int segs = waves / (strips * dup);
	const int seg_rows = 4096 / segs, slots = seg_rows;  // (no halo slots: unique bytes only)
	const double bytes = (double)strips * segs * slots * 2048.0;
	const int blocks = (strips * dup * segs * 64 + 255) / 256;
	const double t = time_ms([&] { hipLaunchKernelGGL((k_walk<RING, NT>), dim3(blocks), dim3(256), 0, 0, img, sink, pitch, strips, segs, slots, stride, dup, seg_rows); });
	printf("%-34s stride %4u  dup %d  rounds %d  ring %d%s: %2d slots per wave  %.3f ms  %7.1f GB/s of distinct bytes\n", what, stride, dup, rounds, RING,
	       NT ? " nt" : "", slots, t, bytes / t / 1e6);
}

template <int RING, int WIDE, bool LOADS, int SHAPE = 0>
static void run_rw(const uint8_t* img, uint8_t* out, const char* what)
{
	const uint32_t pitch = 32768;
	const int strips = (SHAPE == 2) ? 32 : 35, rounds = 2, waves = rounds * 4096, segs = waves / (strips * 2), seg_rows = 4096 / segs, slots = seg_rows;
	const double rd = (double)strips * segs * slots * 2048.0, wr = (double)strips * segs * slots * 2.0 * 8 * 256;
	const int blocks = (strips * 2 * segs * 64 + 255) / 256;
	const double t = time_ms([&] { hipLaunchKernelGGL((k_walk_rw<RING, WIDE, LOADS, SHAPE>), dim3(blocks), dim3(256), 0, 0, img, out, pitch, strips, segs, slots, 960u, seg_rows); });
	printf("%-44s ring %d: %.3f ms  (%.0f MB read, %.0f MB written: %7.1f GB/s together)\n", what, RING, t, LOADS ? rd / 1e6 : 0.0, WIDE == 2 ? 0.0 : wr / 1e6,
	       ((LOADS ? rd : 0.0) + (WIDE == 2 ? 0.0 : wr)) / t / 1e6);
}

int main()
{
	uint8_t* img;
	uint32_t* sink;
	const size_t bytes = (size_t)8192 * 32768 + (1 << 20);
	(void)hipMalloc(&img, bytes), (void)hipMalloc(&sink, 64);
	(void)hipMemset(img, 1, bytes);
	run<2, false>(img, sink, "aligned strips, one wave per strip", 1024, 1, 1);
	run<2, false>(img, sink, "aligned strips, one wave per strip", 1024, 1, 2);
	run<6, false>(img, sink, "aligned strips, one wave per strip", 1024, 1, 2);
	run<2, false>(img, sink, "960-byte strips (64 B overlap)", 960, 1, 2);
	run<6, false>(img, sink, "960-byte strips (64 B overlap)", 960, 1, 2);
	run<2, false>(img, sink, "960-byte strips, pair of waves", 960, 2, 2);
	run<3, false>(img, sink, "960-byte strips, pair of waves", 960, 2, 2);
	run<6, false>(img, sink, "960-byte strips, pair of waves", 960, 2, 2);
	run<2, true>(img, sink, "960-byte strips, pair of waves", 960, 2, 2);
	run<6, true>(img, sink, "960-byte strips, one wave per strip", 960, 1, 2);
	run<2, false>(img, sink, "960-byte strips, pair, 4 rounds", 960, 2, 4);
	run<2, false>(img, sink, "960-byte strips, pair, 1 round", 960, 2, 1);
	uint8_t* out;
	(void)hipMalloc(&out, (size_t)16 * 4096 * 4096 * 2 + (1 << 20));
	run_rw<2, 2, true>(img, out, "pair walk, loads only");
	run_rw<2, 0, false>(img, out, "pair walk, 8 dword stores per slot only");
	run_rw<2, 1, false>(img, out, "pair walk, 2 x 16-byte stores per slot only");
	run_rw<2, 0, true>(img, out, "pair walk, loads + 8 dword stores");
	run_rw<6, 0, true>(img, out, "pair walk, loads + 8 dword stores");
	run_rw<2, 1, true>(img, out, "pair walk, loads + 2 x 16-byte stores");
	run_rw<6, 1, true>(img, out, "pair walk, loads + 2 x 16-byte stores");
	run_rw<2, 0, false, 1>(img, out, "stores only, 240-byte runs, odd planes +2 B (kernel)");
	run_rw<2, 0, false, 2>(img, out, "stores only, 256-byte runs on 256-byte boundaries");
	run_rw<2, 0, true, 1>(img, out, "loads + stores, 240-byte runs (kernel)");
	run_rw<2, 0, true, 2>(img, out, "loads + stores, aligned 256-byte runs");
	run_rw<2, 0, false, 3>(img, out, "stores only, 480-byte runs (two strips merged)");
	run_rw<2, 0, true, 3>(img, out, "loads + stores, 480-byte runs (two strips merged)");
	return 0;
}
