// ako_requant.hip.h -- gate + quantize an UNQUANTIZED coefficient stream for another quantization factor.
//
// The quantization factor touches the forward path in exactly two places: the stores of the C / B / D sub-bands
// (s2dMemcpy, library/lifting.c:154-168: out = (|v| > g) ? v / q : 0) and the lift heads that record q
// (library/lifting.c:253-267).  Lifting, colour and the low-pass chain never see it.  A ratio search that tries one
// quantization after the other (tools/akoenc.cpp:130-214) therefore needs the transform ONCE, with q = 1 and g = 0 --
// which stores the coefficients as they are -- and per candidate only this pass over the stream: every
// [head C B D] group gets its head rewritten and its coefficients gated and divided.  Bit-identical to encoding with
// that quantization from the pixels, because the values s2dMemcpy divides are the int16 coefficients stored here.
#pragma once

#include "ako_kernels.hip.h"

namespace ako
{

constexpr int RQ_THREADS = 256;
constexpr int RQ_PER_THREAD = 8;                       // int16 values per thread: one 16-byte load / store
constexpr int RQ_CHUNK = RQ_THREADS * RQ_PER_THREAD;   // values per workgroup

struct RqSegment  // one [head C B D] group of one tile, level and plane; or a run of values copied as they are
{
	uint64_t start;   // int16 index inside the image's stream (of the head for a group)
	uint64_t count;   // values incl. the head
	int32_t q, g;     // q = 0: plain copy (low-pass sections)
	float rq;
	uint32_t first_block;
};

__global__ __launch_bounds__(RQ_THREADS) void k_requantize(const int16_t* __restrict__ in, int16_t* __restrict__ out,
                                                          const RqSegment* __restrict__ segs, uint32_t n_segs,
                                                          uint64_t image_stride, uint32_t blocks_per_image)
{
	const uint32_t image = blockIdx.x / blocks_per_image, blk = blockIdx.x % blocks_per_image;
	// segment of this block: binary search over first_block (wave-uniform)
	uint32_t lo = 0, hi = n_segs - 1;
	while (lo < hi)
	{
		const uint32_t mid = (lo + hi + 1) >> 1;
		if (segs[mid].first_block <= blk)
			lo = mid;
		else
			hi = mid - 1;
	}
	const RqSegment s = segs[lo];
	const uint64_t base = (uint64_t)image * image_stride + s.start;
	const uint64_t off0 = (uint64_t)(blk - s.first_block) * RQ_CHUNK + (uint64_t)threadIdx.x * RQ_PER_THREAD;
	const int16_t* src = in + base;
	int16_t* dst = out + base;
#pragma unroll
	for (int k = 0; k < RQ_PER_THREAD; k++)
	{
		const uint64_t i = off0 + k;
		if (i >= s.count)
			break;
		int v = src[i];
		if (s.q != 0)
			v = (i == 0) ? s.q : quantize(v, s.q, s.g, s.rq);  // the head holds the q in use (lifting.c:266-267)
		dst[i] = (int16_t)v;
	}
}

}  // namespace ako
