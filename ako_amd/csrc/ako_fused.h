// ako_fused.h -- host / device interface of the two-level workgroup kernels (ako_fused.hip): levels 0 and 1 of a
// u8 RGBA plan in ONE launch per direction, the level-0 low-pass plane handed over through LDS instead of HBM.
//
// Reference loops whose level order is kept: library/lifting.c:182-247 (forward: level k's LL is level k+1's
// input), library/misc.c:229-288 + library/lifting.c:104-148 (inverse: level k+1's output is level k's LL).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ako
{

struct TileDesc;

// geometry shared by both directions (see ako_fused.hip.h for the picture)
#ifndef AKO_F2_STRIPS
#define AKO_F2_STRIPS 4
#endif
constexpr int F2_STRIPS = AKO_F2_STRIPS;     // level-0 strips of 120 net coefficient columns per workgroup (4 or 6)
constexpr int F2_WAVES = 2 * F2_STRIPS;      // a pair of waves per strip (the four planes split two and two)
constexpr int F2_THREADS = 64 * F2_WAVES;    // 768
constexpr int F2_COLS = F2_STRIPS * 120;     // level-0 low-pass columns a workgroup produces / consumes per row
constexpr int F2_OVERLAP = 16;               // ... of which the outer 8 on either side are level-1 halo
constexpr int F2_GNET = F2_COLS - F2_OVERLAP;  // net level-0 coefficient columns per workgroup (704)
constexpr int F2I_OVERLAP = 8;               // inverse: the level-0 strips need their halo lanes filled from the level-1 output
constexpr int F2I_GNET = F2_COLS - F2I_OVERLAP;  // ... so a workgroup nets 712 level-0 columns there
constexpr int F2_L1STRIPS = F2_STRIPS / 2;     // level-1 strips (120 net level-1 columns = 240 low-pass columns each)
constexpr int F2_PITCH = F2_COLS + 32;                // floats per low-pass row and plane in LDS (>= 2 * 240 + 256)

struct F2Level
{
	uint32_t grp_off[4];   // int16 offset of plane p's [head C B D] group inside the tile stream
	float gate[2], rq[2];  // [0] plane 0, [1] the other planes: gate threshold, (1/q)(1 + 1e-6)
	int32_t q[2];
};

struct F2Params
{
	uint8_t* img;            // u8 RGBA images (forward: source, inverse: destination)
	uint64_t img_stride;     // bytes per image
	uint32_t img_pitch;      // pixels per image row
	int16_t* stream;         // coefficient streams
	uint64_t stream_stride;  // int16 per image
	const TileDesc* tiles;
	uint32_t n_tiles, batch;
	uint32_t Tc, Tr;         // level-0 sub-band extent (= level-1 input extent); level 1: Tc / 2, Tr / 2
	int32_t wrap, color;
	F2Level lv[2];
	// level-1 low-pass plane (forward: written, inverse: read): dense int16, [tile instance][plane][Tr/2 x Tc/2]
	int16_t* ll1;
	uint64_t ll1_inst_stride;   // int16 per tile instance
	uint32_t ll1_plane_stride;  // int16 per plane
	uint32_t ll1_pitch;
	uint32_t ll1_in_stream;     // 1: level 1 is the last level, its low-pass lives in the stream's LP section
	uint32_t lp_off[4];         // ... at these int16 offsets
	// row segments: the first one covers rows [0, edge_rows), the last one [last_lo, Tr), the ones between them seg_rows
	// each (the last of those possibly shorter); edge_rows == 0: all of them seg_rows.  Every boundary is a multiple of 6.
	// (The segments at the top / bottom border run the register-hungry border bodies: kept short, their workgroups end
	// with everybody else's.)
	uint32_t groups, segs, seg_rows, edge_rows, last_lo;
	int32_t* ovf_flag;          // inverse, optimistic pipeline: raised when a value may have left int16
	int32_t ovf_gen;
	uint32_t dbg;               // measurement builds (-DAKO_MEASURE) only: AKO_F2_DBG experiment bits
};

// rows [r_lo, r_hi) of a segment
__host__ __device__ inline void f2_segment_rows(const F2Params& P, uint32_t seg, int& r_lo, int& r_hi)
{
	const int Tr = (int)P.Tr, S = (int)P.seg_rows, E = (int)P.edge_rows;
	if (E == 0)
	{
		r_lo = (int)seg * S;
		r_hi = (r_lo + S < Tr) ? r_lo + S : Tr;
	}
	else if (seg == 0)
		r_lo = 0, r_hi = E;
	else if (seg + 1 == P.segs)
		r_lo = (int)P.last_lo, r_hi = Tr;
	else
	{
		r_lo = E + ((int)seg - 1) * S;
		r_hi = (r_lo + S < (int)P.last_lo) ? r_lo + S : (int)P.last_lo;
	}
}

// launchers (ako_fused.hip); kind = K_DD137 or K_CDF53
// (0, or nonzero when the launch could not be prepared; without AKO_EXPERIMENTAL the library holds stubs that are never called)
int akoFused2ForwardLaunch(int kind, const F2Params& P, hipStream_t st);
int akoFused2InverseLaunch(int kind, const F2Params& P, hipStream_t st);

}  // namespace ako
