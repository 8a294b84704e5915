// Which alignment property of the level-0 sub-band runs costs what (round 3).  The level-0 kernels move, per row slot and
// wave, 2 planes x 4 sub-band rows of `NL` lanes x 4 bytes (forward: stores, inverse: loads) against two 16-byte-per-lane
// pixel rows on the other side.  In the product the runs are 240 bytes (60 net lanes) at byte offsets 90 + 2 p + 240 s
// (plane p, strip s): the stream format puts a one-value head in front of every plane's sub-bands (library/misc.c:245-285),
// so no run starts on a 4-, 16- or 64-byte boundary.  This walk has the kernels' shape (a pair of waves per strip, planes
// two and two, 4096 x 4096 sub-bands 32 MiB apart, 8192 resident waves) with net lanes, strip stride and per-plane
// byte offsets as parameters.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/store_align.hip -o /tmp/sa && /tmp/sa
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <type_traits>

struct Shape
{
	int nl;           // net lanes per strip (the strip stride, in lanes)
	int first;        // first net lane inside the wave's 64-lane window
	uint32_t off[4];  // byte offset of plane p's sub-bands
	int all_lanes;    // 1: all 64 lanes move data (inverse loads: halo lanes included)
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// DIR 0: forward (pixel loads, sub-band stores); 1: inverse (sub-band loads, pixel stores).  PIX / SUB switch a side off.
template <int DIR, bool PIX, bool SUB>
__global__ __launch_bounds__(256) void k_walk(uint8_t* __restrict__ img, uint8_t* __restrict__ bands, uint32_t* sink, int strips, int segs, int slots, Shape S)
{
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	const uint32_t u = wave >> 1, role = wave & 1;
	const uint32_t strip = u % strips, seg = u / strips;
	if (seg >= (uint32_t)segs)
		return;
	const uint32_t pitch = 32768;
	const long col_lane = (long)strip * S.nl - S.first + lane;  // this lane's dword column (pair of coefficient columns)
	const bool net = lane >= S.first && lane < S.first + S.nl && col_lane < 2048;
	const bool in = col_lane >= 0 && col_lane < 2048;
	const size_t sub = (size_t)4096 * 4096 * 2;
	uint8_t* px = img + (size_t)seg * slots * 2 * pitch + col_lane * 16;
	uint32_t acc = 0;
	u32x4 ring[2][2];
	uint32_t bring[2][8];
	auto fetch = [&](int v, int k) {
		if (DIR == 0)
		{
			for (int par = 0; par < 2; par++)
				ring[k][par] = (PIX && in) ? *reinterpret_cast<const u32x4*>(px + (size_t)(2 * v + par) * pitch) : u32x4{(uint32_t)v, 1u, 2u, 3u};
		}
		else
		{
			const size_t r = (size_t)seg * slots + v;
			for (int pl = 0; pl < 2; pl++)
				for (int sb = 0; sb < 4; sb++)
				{
					const uint8_t* q = bands + (size_t)((2 * role + pl) * 4 + sb) * sub + S.off[2 * role + pl] + r * 8192 + col_lane * 4;
					uint32_t w = (uint32_t)v;
					if (SUB && in && (S.all_lanes || net))
						__builtin_memcpy(&w, q, 4);
					bring[k][pl * 4 + sb] = w;
				}
		}
	};
	fetch(0, 0), fetch(1, 1);
	for (int base = 0; base < slots; base += 2)
	{
#pragma unroll
		for (int k = 0; k < 2; k++)
		{
			const int v = base + k;
			const size_t r = (size_t)seg * slots + v;
			if (DIR == 0)
			{
				const u32x4 a = ring[k][0], b = ring[k][1];
				fetch(v + 2 < slots ? v + 2 : v, k);
				for (int pl = 0; pl < 2; pl++)
				{
					const u32x4 val = pl ? b : a;
					const uint32_t w[4] = {val.x, val.y, val.z, val.w};
					for (int sb = 0; sb < 4; sb++)
					{
						uint8_t* q = bands + (size_t)((2 * pl + role) * 4 + sb) * sub + S.off[2 * pl + role] + r * 8192 + col_lane * 4;
						if (SUB && net)
							__builtin_memcpy(q, &w[sb], 4);
						else
							acc ^= w[sb];
					}
				}
			}
			else
			{
				uint32_t w[8];
				for (int i = 0; i < 8; i++)
					w[i] = bring[k][i];
				fetch(v + 2 < slots ? v + 2 : v, k);
				const u32x4 a = {w[0], w[1], w[2], w[3]}, b = {w[4], w[5], w[6], w[7]};
				// each wave of the pair finishes one pixel row of the slot
				if (PIX && net)
					*reinterpret_cast<u32x4*>(px + (size_t)(2 * v + role) * pitch) = a ^ b;
				else
					acc ^= a.x ^ b.y ^ a.z ^ b.w ^ a.y ^ b.x ^ a.w ^ b.z;
			}
		}
	}
	if (acc == 0x12345678u)
		sink[0] = acc;
}

// The same traffic with the stores re-shaped inside a workgroup: 8 waves = 4 neighbouring strips x the pair of waves; every
// slot each wave leaves its 8 sub-band dwords per lane in an LDS row buffer (16 row kinds x 448 columns = 7 lines each),
// the workgroup meets at ONE barrier (double buffer), and wave w then stores row kinds 2 w, 2 w + 1 as full 128-byte lines,
// 16 bytes per lane.  A workgroup nets 448 of its 480 columns (the owned lines of the four planes are skewed by one column
// each).  SLOTS_PER_BARRIER row slots share a barrier.
template <bool PIX, int SPB, bool BARRIER>
__global__ __launch_bounds__(512) void k_group(uint8_t* __restrict__ img, uint8_t* __restrict__ bands, uint32_t* sink, int groups, int segs, int slots, uint32_t phase_b)
{
	__shared__ __attribute__((aligned(16))) uint32_t lds[2 * SPB * 16 * 224];
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
	const uint32_t g = blockIdx.x % groups, seg = blockIdx.x / groups;
	if (seg >= (uint32_t)segs)
		return;
	const int strip = wave >> 1, role = (wave & 1) ^ ((wave >> 2) & 1);
	const uint32_t pitch = 32768;
	const long dw0 = (long)g * 224;                          // first owned dword column of the group
	const long col_lane = dw0 - 8 + strip * 60 - 2 + lane;   // this lane's dword column
	const bool in = col_lane >= 0 && col_lane < 2048;
	const bool net = lane >= 2 && lane < 62 && col_lane >= dw0 && col_lane < dw0 + 224 && col_lane < 2048;
	const size_t sub = (size_t)4096 * 4096 * 2;
	const uint8_t* px = img + (size_t)seg * slots * 2 * pitch + col_lane * 16;
	u32x4 ring[2][2];
	auto fetch = [&](int v, int k) {
		for (int par = 0; par < 2; par++)
			ring[k][par] = (PIX && in) ? *reinterpret_cast<const u32x4*>(px + (size_t)(2 * v + par) * pitch) : u32x4{(uint32_t)v, 1u, 2u, 3u};
	};
	fetch(0, 0), fetch(1, 1);
	const int wr = (int)(col_lane - dw0);  // dword index inside a row buffer
	for (int base = 0; base < slots; base += 2 * SPB)
	{
#pragma unroll
		for (int h = 0; h < 2; h++)
		{
			uint32_t* buf = lds + h * SPB * 16 * 224;
#pragma unroll
			for (int j = 0; j < SPB; j++)
			{
				const int k = (h * SPB + j) & 1;
				const int v = base + h * SPB + j;
				const u32x4 a = ring[k][0], b = ring[k][1];
				fetch(v + 2 < slots ? v + 2 : v, k);
				if (net)
					for (int pl = 0; pl < 2; pl++)
					{
						const u32x4 val = pl ? b : a;
						uint32_t* row = buf + (j * 16 + (2 * pl + role) * 4) * 224 + wr;
						row[0] = val.x, row[224] = val.y, row[448] = val.z, row[672] = val.w;
					}
			}
			if (BARRIER)
				__syncthreads();
#pragma unroll
			for (int j = 0; j < SPB; j++)
			{
				const size_t r = (size_t)seg * slots + base + h * SPB + j;
#pragma unroll
				for (int i = 0; i < 2; i++)
				{
					const int kind = 2 * wave + i;
					if (lane < 56 && dw0 + lane * 4 < 2048)
					{
						const u32x4 val = *reinterpret_cast<const u32x4*>(buf + (j * 16 + kind) * 224 + lane * 4);
						*reinterpret_cast<u32x4*>(bands + phase_b + (size_t)kind * sub + r * 8192 + (dw0 + lane * 4) * 4) = val;
					}
				}
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

static uint8_t *img, *bands;
static uint32_t* sink;

template <int DIR, bool PIX, bool SUB> static double run_one(const Shape& S)
{
	const int strips = (2048 + S.nl - 1) / S.nl;
	int segs = 8192 / (strips * 2);
	while (4096 % segs)
		segs--;
	const int slots = 4096 / segs;
	const int blocks = (strips * 2 * segs * 64 + 255) / 256;
	return time_ms([&] { hipLaunchKernelGGL((k_walk<DIR, PIX, SUB>), dim3(blocks), dim3(256), 0, 0, img, bands, sink, strips, segs, slots, S); });
}

static void run(const char* what, Shape S)
{
	const double f_sub = run_one<0, false, true>(S), f_all = run_one<0, true, true>(S);
	const double i_sub = run_one<1, false, true>(S), i_all = run_one<1, true, true>(S);
	printf("%-58s nl %2d first %d off %3u %3u %3u %3u | fwd: sub-band stores %.3f, + pixel loads %.3f ms | inv: sub-band loads %.3f, + pixel stores %.3f ms\n", what, S.nl,
	       S.first, S.off[0], S.off[1], S.off[2], S.off[3], f_sub, f_all, i_sub, i_all);
}

int main()
{
	(void)hipMalloc(&img, (size_t)8192 * 32768 + (1 << 20));
	(void)hipMalloc(&bands, (size_t)16 * 4096 * 4096 * 2 + (1 << 20));
	(void)hipMalloc(&sink, 64);
	(void)hipMemset(img, 1, (size_t)8192 * 32768 + (1 << 20));
	(void)hipMemset(bands, 1, (size_t)16 * 4096 * 4096 * 2 + (1 << 20));
	{
		Shape S{64, 0, {0, 0, 0, 0}, 1};
		printf("pixel side alone: forward loads %.3f ms, inverse stores %.3f ms (aligned 1 KiB strips)\n", run_one<0, true, false>(S), run_one<1, true, false>(S));
		Shape K{60, 2, {0, 0, 0, 0}, 1};
		printf("pixel side alone: forward loads %.3f ms, inverse stores %.3f ms (960-byte strips)\n", run_one<0, true, false>(K), run_one<1, true, false>(K));
	}
	{
		const int groups = 10, segs = 102, slots = 40;
		auto go = [&](auto pix, auto spb, auto bar, uint32_t phase) {
			return time_ms([&] {
				hipLaunchKernelGGL((k_group<decltype(pix)::value, decltype(spb)::value, decltype(bar)::value>), dim3(groups * segs), dim3(512), 0, 0, img, bands, sink, groups, segs,
				                   slots, phase);
			});
		};
		using T = std::true_type;
		using F = std::false_type;
		using I1 = std::integral_constant<int, 1>;
		using I2 = std::integral_constant<int, 2>;
		printf("stores through LDS, full lines (4080 of 4096 rows): sub-band stores %.3f, + pixel loads %.3f ms | two slots per barrier %.3f, %.3f | no barrier %.3f, %.3f | lines at +64: %.3f\n",
		       go(F{}, I1{}, T{}, 0), go(T{}, I1{}, T{}, 0), go(F{}, I2{}, T{}, 0), go(T{}, I2{}, T{}, 0), go(F{}, I1{}, F{}, 0), go(T{}, I1{}, F{}, 0), go(T{}, I1{}, T{}, 64));
	}
	run("the product: 240-byte runs at 90 + 2 p", {60, 2, {90, 92, 94, 96}, 1});
	run("240-byte runs, every plane on a 16-byte boundary", {60, 2, {0, 0, 0, 0}, 1});
	run("240-byte runs, a wave's planes 16-byte / +4", {60, 2, {0, 0, 4, 4}, 1});
	run("240-byte runs, all planes +4 (dword aligned only)", {60, 2, {4, 4, 4, 4}, 1});
	run("240-byte runs, all planes +2", {60, 2, {2, 2, 2, 2}, 1});
	run("224-byte runs on 32-byte boundaries (/ +4)", {56, 4, {0, 0, 4, 4}, 1});
	run("224-byte runs on 32-byte boundaries, all planes", {56, 4, {0, 0, 0, 0}, 1});
	run("192-byte runs on 64-byte boundaries (/ +4)", {48, 8, {0, 0, 4, 4}, 1});
	run("192-byte runs on 64-byte boundaries, all planes", {48, 8, {0, 0, 0, 0}, 1});
	run("192-byte runs, window on a 64-byte boundary, at 90 + 2 p", {48, 8, {90, 92, 94, 96}, 1});
	run("128-byte runs on 128-byte boundaries", {32, 16, {0, 0, 0, 0}, 1});
	run("256-byte runs on 256-byte boundaries (no halo: ideal)", {64, 0, {0, 0, 0, 0}, 1});
	run("256-byte runs +2", {64, 0, {2, 2, 2, 2}, 1});
	run("256-byte runs +4", {64, 0, {4, 4, 4, 4}, 1});
	run("256-byte runs +16", {64, 0, {16, 16, 16, 16}, 1});
	run("256-byte runs +32", {64, 0, {32, 32, 32, 32}, 1});
	run("256-byte runs +64", {64, 0, {64, 64, 64, 64}, 1});
	run("256-byte runs at 90 + 2 p", {64, 0, {90, 92, 94, 96}, 1});
	run("240-byte runs, net lanes only load (inverse)", {60, 2, {90, 92, 94, 96}, 0});
	return 0;
}
