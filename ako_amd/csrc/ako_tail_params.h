// ako_tail_params.h -- parameters of the in-LDS tail kernels (ako_tail.hip.h: window / segment engines;
// ako_tail3.hip.h: line engine), shared by the translation units that launch and define them.
#pragma once

#include "ako_kernels.hip.h"

namespace ako
{

constexpr int TAIL_LEVELS = 10;

struct TailLevel
{
	uint32_t cw, ch, tw, th;  // full and sub-band extents
	int32_t kind;
	int32_t q[2], g[2];       // [0] plane 0, [1] the other planes
	float rq[2];
	uint64_t grp0;            // int16 offset of plane 0's [head C B D] group inside the tile stream
	uint32_t gsize;           // 1 + 3 * tw * th
};

struct TailParams
{
	uint32_t nlev;            // levels handled here, largest first
	TailLevel lv[TAIL_LEVELS];
	int32_t wrap;
	uint32_t channels;
	const TileDesc* tiles;
	uint32_t n_tiles, batch;
	uint32_t pitch;           // segment engine: LDS row pitch (elements) = 2 * ceil(lv[0].cw / 2)
	uint32_t win_elems;       // window engine: elements of the first (largest) level's window; the dense LL array follows
	// int16 plane side: the LL plane handed over by / to the level kernels (or PLANES_I16 images).
	// The u8 side (colour transform across planes) never runs here: level 0 of a u8 image is always
	// a level kernel.
	int16_t* plane;
	uint64_t plane_inst_stride, plane_plane_stride;
	uint32_t plane_pitch, plane_tiled;
	// stream
	int16_t* stream;
	uint64_t stream_stride;
	uint32_t fw, fh;          // final low-pass extent; plane p's low-pass sits at p * fw * fh
};


// ---- line engine (ako_tail3.hip.h): geometry shared with the host ----
constexpr int T3_MAX = 256;  // largest level extent in samples, either direction

// int16 elements per row of a level with tc coefficient columns: a multiple of four (rows stay 8-byte aligned for the
// lanes' 8-byte accesses); ZERO borders keep four zero samples behind the row
__host__ __device__ inline uint32_t t3_pitch(uint32_t tc, int wrap)
{
	return ((2 * tc + 3u) & ~3u) + ((wrap == W_ZERO) ? 4u : 0u);
}
__host__ __device__ inline uint32_t t3_level_elems(uint32_t tc, uint32_t tr, int wrap)
{
	return 2 * tr * t3_pitch(tc, wrap);
}
// int16 elements of LDS in front of buffer A: a row of zeros for ZERO borders
__host__ __device__ inline uint32_t t3_zero_elems(uint32_t tc0, int wrap)
{
	return (wrap == W_ZERO) ? t3_pitch(tc0, wrap) : 0u;
}

// line engine (ako_tail3.hip): launchers
void akoTail3ForwardLaunch(const TailParams& P, uint32_t blocks, uint32_t threads, uint32_t lds_bytes, hipStream_t st);
void akoTail3InverseLaunch(const TailParams& P, uint32_t blocks, uint32_t threads, uint32_t lds_bytes, hipStream_t st);

}  // namespace ako
