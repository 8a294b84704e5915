// ako_tail_params.h -- parameters of the in-LDS tail kernels (ako_tail.hip.h).
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


}  // namespace ako
