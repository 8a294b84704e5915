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

// A thread owns one 16-byte ALIGNED window of eight values (aligned in the stream buffer, whatever the segment's first
// index is): windows that lie inside the segment are one 16-byte load and one 16-byte store with the values re-quantized
// in registers; the window a segment starts or ends in is shared with its neighbour and goes value by value.  The host
// sizes a segment's blocks for the worst alignment (count + 7 values).
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
	const uint64_t seg_lo = (uint64_t)image * image_stride + s.start, seg_hi = seg_lo + s.count;  // absolute value indices
	const bool aligned = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
	const uint64_t w0 = (aligned ? (seg_lo & ~7ull) : seg_lo) + (uint64_t)(blk - s.first_block) * RQ_CHUNK + (uint64_t)threadIdx.x * RQ_PER_THREAD;
	if (w0 >= seg_hi)
		return;
	auto requant = [&](int v, bool head) {
		if (s.q == 0)
			return v;  // low-pass section: copied
		return head ? s.q : quantize(v, s.q, s.g, s.rq);  // the head holds the q in use (lifting.c:266-267)
	};
	if (aligned && w0 >= seg_lo && w0 + RQ_PER_THREAD <= seg_hi)
	{
		const uint4 raw = *reinterpret_cast<const uint4*>(in + w0);
		const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
		uint32_t r[4];
#pragma unroll
		for (int k = 0; k < 4; k++)
		{
			const int a = requant((int)(int16_t)(w[k] & 0xFFFFu), k == 0 && w0 == seg_lo);
			const int b = requant((int)w[k] >> 16, false);
			r[k] = ((uint32_t)a & 0xFFFFu) | ((uint32_t)b << 16);
		}
		*reinterpret_cast<uint4*>(out + w0) = make_uint4(r[0], r[1], r[2], r[3]);
		return;
	}
#pragma unroll
	for (int k = 0; k < RQ_PER_THREAD; k++)
	{
		const uint64_t i = w0 + k;
		if (i >= seg_lo && i < seg_hi)
			out[i] = (int16_t)requant(in[i], i == seg_lo);
	}
}

}  // namespace ako
