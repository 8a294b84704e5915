// ako_tail.hip.h -- the small end of the pyramid in ONE launch per direction.
//
// Once a level is no larger than TAIL_MAX x TAIL_MAX samples, every further level of that plane
// fits in LDS.  One workgroup per (tile instance, plane) then runs all remaining levels back to
// back: the level's samples sit in an LDS window with the usual halo (same in-place lifting engine
// and the same boundary rules as ako_kernels.hip.h, here with ONE window covering the whole
// plane), the C / B / D sub-bands go straight to their places in the coefficient stream (gate +
// quantization fused, lift head written by one lane), the LL band is compacted into a small dense
// LDS array and becomes the next level's input -- it never travels to HBM.
//
// This replaces, per direction, 6-7 dependent launches of almost empty grids (each one a fixed
// ~5-40 us) by a single launch; it is also the "one tile = one workgroup" regime of the north star
// for tiles of up to 128 x 128 pixels.
//
// reference: library/lifting.c:171-292 (forward level loop), library/misc.c:229-288 +
// library/lifting.c:104-148 (inverse level loop)
#pragma once

#include "ako_kernels.hip.h"
#include "ako_stream.hip.h"  // lift_add / sum_p / sum_u / shift_p / shift_u (segment engine)

namespace ako
{

constexpr int TAIL_MAX = 128;     // largest level extent handled here
constexpr int TAIL_THREADS = 1024;
constexpr int TAIL_LEVELS = 10;
// window of the largest tail level (sub-band extent TAIL_MAX / 2 = 64): (2 * (64 + 6)) rows of
// 2 * (64 + 8) samples; the dense LL array holds at most 64 x 64 samples
constexpr int TAIL_WROWS = 2 * (TAIL_MAX / 2 + 6);
constexpr int TAIL_WCOLS = 2 * (TAIL_MAX / 2 + 8);
constexpr int TAIL_LDS_BYTES = (TAIL_WROWS * TAIL_WCOLS + (TAIL_MAX / 2) * (TAIL_MAX / 2)) * 2;

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

// exact idx / d for idx < 2^16, d < 2^16 with one multiply-high: M = floor(2^32 / d) + 1
struct FastDiv
{
	uint32_t d, m;
	__device__ __forceinline__ explicit FastDiv(int dd) : d((uint32_t)dd), m(dd > 1 ? (0xFFFFFFFFu / (uint32_t)dd + 1u) : 0u) {}
	__device__ __forceinline__ int div(int idx) const
	{
		return (d > 1) ? (int)__umulhi((uint32_t)idx, m) : idx;
	}
};

// The four lifting phases of one level on a window that covers the whole plane (origin slot
// -ORG_R / -ORG_C, Tc x Tr sub-band coefficients, element pitch wp).
template <int KIND, int SGN>
__device__ __forceinline__ void tail_level_passes(int16_t* W, int wp, int Tc, int Tr, int wrap, int tid, int nthreads)
{
	const int nrows = 2 * (Tr + 2 * ORG_R);
	const FastDiv dA(Tc + 3), dB(Tc), dC(2 * Tc), dD(2 * (Tc + 2 * ORG_C));
	if (SGN > 0)
	{
		// rows: predict over slots [-2, Tc], every window row; then update over [0, Tc)
		for (int idx = tid; idx < nrows * (Tc + 3); idx += nthreads)
		{
			const int wr = dA.div(idx), jj = idx - wr * (Tc + 3);
			lift_step<KIND, true, +1, true>(W + wr * wp, 1, jj + 2, jj - 2, -ORG_C, Tc, wrap);
		}
		__syncthreads();
		if (KIND != K_HAAR)
		{
			for (int idx = tid; idx < nrows * Tc; idx += nthreads)
			{
				const int wr = dB.div(idx), jj = idx - wr * Tc;
				lift_step<KIND, false, +1, false>(W + wr * wp, 1, jj + ORG_C, jj, -ORG_C, Tc, wrap);
			}
			__syncthreads();
		}
		// columns: predict over row slots [-2, Tr], the 2*Tc net columns; then update over [0, Tr)
		for (int idx = tid; idx < (Tr + 3) * (2 * Tc); idx += nthreads)
		{
			const int ii = dC.div(idx), x = idx - ii * (2 * Tc);
			lift_step<KIND, true, +1, true>(W + 2 * ORG_C + x, wp, ii + 1, ii - 2, -ORG_R, Tr, wrap);
		}
		__syncthreads();
		if (KIND != K_HAAR)
		{
			for (int idx = tid; idx < Tr * (2 * Tc); idx += nthreads)
			{
				const int ii = dC.div(idx), x = idx - ii * (2 * Tc);
				lift_step<KIND, false, +1, false>(W + 2 * ORG_C + x, wp, ii + ORG_R, ii, -ORG_R, Tr, wrap);
			}
			__syncthreads();
		}
	}
	else
	{
		const int ncols = 2 * (Tc + 2 * ORG_C);
		// columns: evens over row slots [-1, Tr+1], every window column; then odds over [0, Tr)
		if (KIND != K_HAAR)
		{
			for (int idx = tid; idx < (Tr + 3) * ncols; idx += nthreads)
			{
				const int ii = dD.div(idx), x = idx - ii * ncols;
				lift_step<KIND, false, -1, true>(W + x, wp, ii + 2, ii - 1, -ORG_R, Tr, wrap);
			}
			__syncthreads();
		}
		for (int idx = tid; idx < Tr * ncols; idx += nthreads)
		{
			const int ii = dD.div(idx), x = idx - ii * ncols;
			lift_step<KIND, true, -1, false>(W + x, wp, ii + ORG_R, ii, -ORG_R, Tr, wrap);
		}
		__syncthreads();
		// rows: evens over slots [-1, Tc+1], the 2*Tr net rows; then odds over [0, Tc)
		if (KIND != K_HAAR)
		{
			for (int idx = tid; idx < (2 * Tr) * (Tc + 3); idx += nthreads)
			{
				const int y = dA.div(idx), jj = idx - y * (Tc + 3);
				lift_step<KIND, false, -1, true>(W + (2 * ORG_R + y) * wp, 1, jj + 3, jj - 1, -ORG_C, Tc, wrap);
			}
			__syncthreads();
		}
		for (int idx = tid; idx < (2 * Tr) * Tc; idx += nthreads)
		{
			const int y = dB.div(idx), jj = idx - y * Tc;
			lift_step<KIND, true, -1, false>(W + (2 * ORG_R + y) * wp, 1, jj + ORG_C, jj, -ORG_C, Tc, wrap);
		}
		__syncthreads();
	}
}

template <int SGN>
__device__ __forceinline__ void tail_level_dispatch(int kind, int16_t* W, int wp, int Tc, int Tr, int wrap, int tid, int nthreads)
{
	if (kind == K_DD137)
		tail_level_passes<K_DD137, SGN>(W, wp, Tc, Tr, wrap, tid, nthreads);
	else if (kind == K_CDF53)
		tail_level_passes<K_CDF53, SGN>(W, wp, Tc, Tr, wrap, tid, nthreads);
	else
		tail_level_passes<K_HAAR, SGN>(W, wp, Tc, Tr, wrap, tid, nthreads);
}

__global__ __launch_bounds__(TAIL_THREADS) void k_forward_tail(const TailParams P)
{
	extern __shared__ __attribute__((aligned(16))) int16_t smem[];
	int16_t* W = smem;
	int16_t* dense = smem + P.win_elems;
	const int tid = threadIdx.x, nthreads = blockDim.x;
	const uint32_t p = blockIdx.x % P.channels;
	const uint64_t inst = blockIdx.x / P.channels;
	const uint32_t tile = (uint32_t)(inst % P.n_tiles), image = (uint32_t)(inst / P.n_tiles);
	const TileDesc td = P.tiles[tile];
	int16_t* tile_stream = P.stream + (uint64_t)image * P.stream_stride + td.stream_off;
	const int wrap = P.wrap;
	const int m = (p == 0) ? 0 : 1;

	for (uint32_t l = 0; l < P.nlev; l++)
	{
		const TailLevel& L = P.lv[l];
		const int Tc = (int)L.tw, Tr = (int)L.th, cw = (int)L.cw, chh = (int)L.ch;
		const int wcols = 2 * (Tc + 2 * ORG_C), wrows = 2 * (Tr + 2 * ORG_R), wp = wcols;

		// ---- fill the window (halo by index map, phantom last row / column by clamping) ---------
		// 8 elements per thread and round so that the global loads of the first level overlap
		const FastDiv dW(wcols);
		const int16_t* gsrc = nullptr;
		if (l == 0)
		{
			gsrc = P.plane + (P.plane_tiled ? (uint64_t)image : inst) * P.plane_inst_stride +
			       (uint64_t)p * P.plane_plane_stride;
			if (P.plane_tiled)
				gsrc += (uint64_t)td.y0 * P.plane_pitch + td.x0;
		}
		for (int base = tid; base < wrows * wcols; base += 8 * nthreads)
		{
			int16_t val[8];
#pragma unroll
			for (int k = 0; k < 8; k++)
			{
				const int idx = base + k * nthreads;
				val[k] = 0;
				if (idx < wrows * wcols)
				{
					const int wr = dW.div(idx), wc = idx - wr * wcols;
					const int mr = map_index((wr >> 1) - ORG_R, Tr, wrap);
					const int mc = map_index((wc >> 1) - ORG_C, Tc, wrap);
					if (mr >= 0 && mc >= 0)
					{
						const int y = min(2 * mr + (wr & 1), chh - 1), x = min(2 * mc + (wc & 1), cw - 1);
						val[k] = (l != 0) ? dense[y * cw + x] : gsrc[(uint64_t)y * P.plane_pitch + x];
					}
				}
			}
#pragma unroll
			for (int k = 0; k < 8; k++)
			{
				const int idx = base + k * nthreads;
				if (idx < wrows * wcols)
					W[idx] = val[k];
			}
		}
		__syncthreads();

		tail_level_dispatch<+1>(L.kind, W, wp, Tc, Tr, wrap, tid, nthreads);

		// ---- sub-bands out: C, B, D to the stream, LL to the dense array / final low-pass --------
		int16_t* grp = tile_stream + L.grp0 + (uint64_t)p * L.gsize;
		const int nsub = Tc * Tr;
		const int q = L.q[m], g = L.g[m];
		const float rq = L.rq[m];
		const bool last = (l + 1 == P.nlev);
		int16_t* lp_out = tile_stream + (uint64_t)p * P.fw * P.fh;
		if (tid == 0)
			grp[0] = (int16_t)q;
		const FastDiv dT(Tc);
		for (int idx = tid; idx < nsub; idx += nthreads)
		{
			const int r = dT.div(idx), c = idx - r * Tc;
			const int16_t* cell = W + (2 * (r + ORG_R)) * wp + 2 * (c + ORG_C);
			if (last)
				lp_out[idx] = cell[0];
			else
				dense[idx] = cell[0];
			grp[1 + idx] = quantize(cell[wp], q, g, rq);
			grp[1 + nsub + idx] = quantize(cell[1], q, g, rq);
			grp[1 + 2 * nsub + idx] = quantize(cell[wp + 1], q, g, rq);
		}
		__syncthreads();
	}
}

__global__ __launch_bounds__(TAIL_THREADS) void k_inverse_tail(const TailParams P)
{
	extern __shared__ __attribute__((aligned(16))) int16_t smem[];
	int16_t* W = smem;
	int16_t* dense = smem + P.win_elems;
	const int tid = threadIdx.x, nthreads = blockDim.x;
	const uint32_t p = blockIdx.x % P.channels;
	const uint64_t inst = blockIdx.x / P.channels;
	const uint32_t tile = (uint32_t)(inst % P.n_tiles), image = (uint32_t)(inst / P.n_tiles);
	const TileDesc td = P.tiles[tile];
	const int16_t* tile_stream = P.stream + (uint64_t)image * P.stream_stride + td.stream_off;
	const int wrap = P.wrap;

	// smallest level first (misc.c:257-285); its LL is the stream's low-pass section
	for (int l = (int)P.nlev - 1; l >= 0; l--)
	{
		const TailLevel& L = P.lv[l];
		const int Tc = (int)L.tw, Tr = (int)L.th, ow = (int)L.cw, oh = (int)L.ch;
		const int wcols = 2 * (Tc + 2 * ORG_C), wrows = 2 * (Tr + 2 * ORG_R), wp = wcols;
		const int16_t* grp = tile_stream + L.grp0 + (uint64_t)p * L.gsize;
		const int nsub = Tc * Tr;
		const int q = grp[0];
		const int16_t* ll = (l + 1 == (int)P.nlev) ? (tile_stream + (uint64_t)p * P.fw * P.fh) : dense;

		const FastDiv dW(wcols);
		for (int base = tid; base < wrows * wcols; base += 8 * nthreads)
		{
			int16_t val[8];
			bool hp[8];
#pragma unroll
			for (int k = 0; k < 8; k++)
			{
				const int idx = base + k * nthreads;
				val[k] = 0, hp[k] = false;
				if (idx < wrows * wcols)
				{
					const int wr = dW.div(idx), wc = idx - wr * wcols;
					const int mr = map_index((wr >> 1) - ORG_R, Tr, wrap);
					const int mc = map_index((wc >> 1) - ORG_C, Tc, wrap);
					if (mr >= 0 && mc >= 0)
					{
						const int quad = (wr & 1) * 2 + (wc & 1);  // 0 LL, 1 B, 2 C, 3 D
						const int o = mr * Tc + mc;
						if (quad == 0)
							val[k] = ll[o];
						else
						{
							const int sel = (quad == 2) ? 0 : ((quad == 1) ? 1 : 2);  // stream order C, B, D
							val[k] = grp[1 + sel * nsub + o];
							hp[k] = true;
						}
					}
				}
			}
#pragma unroll
			for (int k = 0; k < 8; k++)
			{
				const int idx = base + k * nthreads;
				if (idx < wrows * wcols)
					W[idx] = (hp[k] && q > 1) ? (int16_t)((int)val[k] * q) : val[k];  // lifting.c:30-40
			}
		}
		__syncthreads();

		tail_level_dispatch<-1>(L.kind, W, wp, Tc, Tr, wrap, tid, nthreads);

		// ---- the level's output: next level's LL (dense LDS), or the plane / image --------------
		const FastDiv dO(ow);
		for (int idx = tid; idx < ow * oh; idx += nthreads)
		{
			const int y = dO.div(idx), x = idx - y * ow;
			const int16_t v = W[(2 * ORG_R + y) * wp + 2 * ORG_C + x];
			if (l != 0)
				dense[idx] = v;
			else
			{
				int16_t* base = P.plane + (P.plane_tiled ? (uint64_t)image : inst) * P.plane_inst_stride +
				                (uint64_t)p * P.plane_plane_stride;
				if (P.plane_tiled)
					base += (uint64_t)td.y0 * P.plane_pitch + td.x0;
				base[(uint64_t)y * P.plane_pitch + x] = v;
			}
		}
		__syncthreads();
	}
}


// =============================================================================================
// Second tail engine: "segments in registers" (throughput regime: many small planes, e.g. tiled images)
//
//   * the plane sits dense in LDS (row pitch P = 2 * ceil(w0 / 2)); every level works IN PLACE on the
//     top-left corner and leaves its sub-bands de-interleaved there (LL | B over C | D)
//   * a row pass gives each thread one segment of SEG coefficient pairs of a row: it reads the SEG + 6
//     even and SEG + 3 odd samples it needs into registers (halo by the SURVEY A.2 index map), computes
//     the high-pass values incl. their halo (stage A), then the low-pass values, waits at ONE barrier
//     (all segments of a row are in flight together) and writes [LP | HP] back; a column pass does the
//     same down a column with lanes on adjacent columns (conflict-free LDS)
//   * about 50 instructions per sample instead of the window engine's ~160, planes up to 256 x 256
//     (128 KiB of LDS); a single small plane has a longer critical path than on the window engine, so
//     the plan uses this engine when a launch has many planes (run_tail in ako_plan.hip)
// =============================================================================================
constexpr int SEG_TAIL_MAX = 256;
constexpr int SEGT_THREADS = 1024;
constexpr int SEG = 8;  // coefficient pairs per thread and pass

// ---- one segment of a 1-D forward lift ---------------------------------------------------------
// EV(m) / OD(m) read the even / odd source coefficient m in [0, T).  Produces LP and HP for
// coefficients c_a .. c_a + SEG - 1 (those < T are valid).
template <int KIND, typename FE, typename FO>
__device__ __forceinline__ void seg_forward(FE EV, FO OD, int c_a, int T, int wrap, int lp[SEG], int hp[SEG])
{
	int e[SEG + 6], o[SEG + 3], h[SEG + 3];
#pragma unroll
	for (int j = 0; j < SEG + 6; j++)
	{
		const int m = map_index(c_a - 3 + j, T, wrap);
		e[j] = (m < 0) ? 0 : EV(m);
	}
	if (KIND == K_HAAR)
	{
#pragma unroll
		for (int k = 0; k < SEG; k++)
		{
			const int m = map_index(c_a + k, T, wrap);
			lp[k] = e[k + 3];
			hp[k] = (int)(int16_t)(((m < 0) ? 0 : OD(m)) - e[k + 3]);
		}
		return;
	}
#pragma unroll
	for (int j = 0; j < SEG + 3; j++)
	{
		const int m = map_index(c_a - 2 + j, T, wrap);
		o[j] = (m < 0) ? 0 : OD(m);
	}
	// stage A: high-pass for v = c_a - 2 + j, halo slots included
#pragma unroll
	for (int j = 0; j < SEG + 3; j++)
	{
		const int v = c_a - 2 + j;
		int p2 = e[j + 3];
		if (KIND == K_DD137 && wrap == W_MIRROR && v + 2 >= T)
			p2 = e[j];
		h[j] = lift_add<true>(o[j], sum_p<KIND, +1>(e[j], e[j + 1], e[j + 2], p2), shift_p<KIND>());
	}
	if (wrap != W_REPEAT)
	{
		// halo of the high-pass sequence: nearest in-range value (CLAMP / MIRROR) or zero
		int h_first = 0, h_last = 0;
#pragma unroll
		for (int j = 0; j < SEG + 3; j++)
		{
			const int v = c_a - 2 + j;
			if (v == 0)
				h_first = h[j];
			if (v == T - 1)
				h_last = h[j];
		}
		if (wrap == W_ZERO)
			h_first = 0, h_last = 0;
#pragma unroll
		for (int j = 0; j < SEG + 3; j++)
		{
			const int v = c_a - 2 + j;
			if (v < 0)
				h[j] = h_first;
			if (v >= T)
				h[j] = h_last;
		}
	}
#pragma unroll
	for (int k = 0; k < SEG; k++)
	{
		const int c = c_a + k;
		int l2 = h[k];
		if (KIND == K_DD137 && wrap == W_MIRROR && c < 2)
			l2 = h[k + 3];
		lp[k] = lift_add<true>(e[k + 3], sum_u<KIND, +1>(l2, h[k + 1], h[k + 2], h[k + 3]), shift_u<KIND>());
		hp[k] = h[k + 2];
	}
}

// ---- one segment of a 1-D inverse lift ---------------------------------------------------------
// LPF(m) / HPF(m) read low-pass / high-pass coefficient m in [0, T).  Produces the even and odd
// samples of coefficients c_a .. c_a + SEG - 1.
template <int KIND, typename FL, typename FH>
__device__ __forceinline__ void seg_inverse(FL LPF, FH HPF, int c_a, int T, int wrap, int ev_out[SEG], int od_out[SEG])
{
	int hv[SEG + 6], lv[SEG + 3], ev[SEG + 3];
#pragma unroll
	for (int j = 0; j < SEG + 6; j++)
	{
		const int m = map_index(c_a - 3 + j, T, wrap);
		hv[j] = (m < 0) ? 0 : HPF(m);
	}
#pragma unroll
	for (int j = 0; j < SEG + 3; j++)
	{
		const int m = map_index(c_a - 1 + j, T, wrap);
		lv[j] = (m < 0) ? 0 : LPF(m);
	}
	if (KIND == K_HAAR)
	{
#pragma unroll
		for (int k = 0; k < SEG; k++)
		{
			ev_out[k] = lv[k + 1];
			od_out[k] = (int)(int16_t)(lv[k + 1] + hv[k + 3]);
		}
		return;
	}
	// stage A: evens for v = c_a - 1 + j, halo slots included
#pragma unroll
	for (int j = 0; j < SEG + 3; j++)
	{
		const int v = c_a - 1 + j;
		int l2 = hv[j];
		if (KIND == K_DD137 && wrap == W_MIRROR && v < 2)
			l2 = hv[j + 3];
		ev[j] = lift_add<true>(lv[j], sum_u<KIND, -1>(l2, hv[j + 1], hv[j + 2], hv[j + 3]), shift_u<KIND>());
	}
	if (wrap != W_REPEAT)
	{
		int e_first = 0, e_last = 0;
#pragma unroll
		for (int j = 0; j < SEG + 3; j++)
		{
			const int v = c_a - 1 + j;
			if (v == 0)
				e_first = ev[j];
			if (v == T - 1)
				e_last = ev[j];
		}
		if (wrap == W_ZERO)
			e_first = 0, e_last = 0;
#pragma unroll
		for (int j = 0; j < SEG + 3; j++)
		{
			const int v = c_a - 1 + j;
			if (v < 0)
				ev[j] = e_first;
			if (v >= T)
				ev[j] = e_last;
		}
	}
#pragma unroll
	for (int k = 0; k < SEG; k++)
	{
		const int c = c_a + k;
		int p2 = ev[k + 3];
		if (KIND == K_DD137 && wrap == W_MIRROR && c + 2 >= T)
			p2 = ev[k];
		ev_out[k] = ev[k + 1];
		od_out[k] = lift_add<true>(hv[k + 3], sum_p<KIND, -1>(ev[k], ev[k + 1], ev[k + 2], p2), shift_p<KIND>());
	}
}

// ---- 2-D level passes on the LDS plane ----------------------------------------------------------

template <int KIND>
__device__ __forceinline__ void tail_forward_level(int16_t* A, int P, int cw, int chh, int Tc, int Tr, int wrap, int tid)
{
	// rows: all segments of a row share a round (one barrier between its loads and stores)
	{
		const int segs = (Tc + SEG - 1) / SEG;
		const int rows_per_round = max(1, SEGT_THREADS / segs);
		const FastDiv dseg(segs);
		for (int row0 = 0; row0 < chh; row0 += rows_per_round)
		{
			const int rr = dseg.div(tid), sg = tid - rr * segs;
			const int r = row0 + rr;
			const bool active = (rr < rows_per_round) && (r < chh);
			int lp[SEG], hp[SEG];
			int16_t* row = A + r * P;
			if (active)
				seg_forward<KIND>([&](int m) { return (int)row[2 * m]; },
				                  [&](int m) { return (int)row[min(2 * m + 1, cw - 1)]; },  // phantom odd = last even
				                  sg * SEG, Tc, wrap, lp, hp);
			__syncthreads();
			if (active)
			{
#pragma unroll
				for (int k = 0; k < SEG; k++)
				{
					const int c = sg * SEG + k;
					if (c < Tc)
						row[c] = (int16_t)lp[k], row[Tc + c] = (int16_t)hp[k];
				}
			}
		}
		__syncthreads();
	}
	// columns (2 * Tc of them); the phantom last row of an odd level is the lifted last row
	{
		const int ncols = 2 * Tc;
		const int segs = (Tr + SEG - 1) / SEG;
		const int cols_per_round = max(1, min(ncols, SEGT_THREADS / segs));
		const FastDiv dcol(cols_per_round);
		for (int col0 = 0; col0 < ncols; col0 += cols_per_round)
		{
			const int sg = dcol.div(tid), xx = tid - sg * cols_per_round;
			const int x = col0 + xx;
			const bool active = (sg < segs) && (x < ncols);
			int lp[SEG], hp[SEG];
			int16_t* col = A + x;
			if (active)
				seg_forward<KIND>([&](int m) { return (int)col[(2 * m) * P]; },
				                  [&](int m) { return (int)col[min(2 * m + 1, chh - 1) * P]; }, sg * SEG, Tr, wrap, lp,
				                  hp);
			__syncthreads();
			if (active)
			{
#pragma unroll
				for (int k = 0; k < SEG; k++)
				{
					const int r = sg * SEG + k;
					if (r < Tr)
						col[r * P] = (int16_t)lp[k], col[(Tr + r) * P] = (int16_t)hp[k];
				}
			}
		}
		__syncthreads();
	}
}

template <int KIND>
__device__ __forceinline__ void tail_inverse_level(int16_t* A, int P, int Tc, int Tr, int wrap, int tid)
{
	// columns first (lifting.c:137-138): (LL over C) and (B over D) -> interleaved rows
	{
		const int ncols = 2 * Tc;
		const int segs = (Tr + SEG - 1) / SEG;
		const int cols_per_round = max(1, min(ncols, SEGT_THREADS / segs));
		const FastDiv dcol(cols_per_round);
		for (int col0 = 0; col0 < ncols; col0 += cols_per_round)
		{
			const int sg = dcol.div(tid), xx = tid - sg * cols_per_round;
			const int x = col0 + xx;
			const bool active = (sg < segs) && (x < ncols);
			int ev[SEG], od[SEG];
			int16_t* col = A + x;
			if (active)
				seg_inverse<KIND>([&](int m) { return (int)col[m * P]; }, [&](int m) { return (int)col[(Tr + m) * P]; },
				                  sg * SEG, Tr, wrap, ev, od);
			__syncthreads();
			if (active)
			{
#pragma unroll
				for (int k = 0; k < SEG; k++)
				{
					const int r = sg * SEG + k;
					if (r < Tr)
						col[(2 * r) * P] = (int16_t)ev[k], col[(2 * r + 1) * P] = (int16_t)od[k];
				}
			}
		}
		__syncthreads();
	}
	// rows (2 * Tr of them; a phantom last row is computed and ignored by the next stage)
	{
		const int nrows = 2 * Tr;
		const int segs = (Tc + SEG - 1) / SEG;
		const int rows_per_round = max(1, SEGT_THREADS / segs);
		const FastDiv dseg(segs);
		for (int row0 = 0; row0 < nrows; row0 += rows_per_round)
		{
			const int rr = dseg.div(tid), sg = tid - rr * segs;
			const int r = row0 + rr;
			const bool active = (rr < rows_per_round) && (r < nrows);
			int ev[SEG], od[SEG];
			int16_t* row = A + r * P;
			if (active)
				seg_inverse<KIND>([&](int m) { return (int)row[m]; }, [&](int m) { return (int)row[Tc + m]; }, sg * SEG,
				                  Tc, wrap, ev, od);
			__syncthreads();
			if (active)
			{
#pragma unroll
				for (int k = 0; k < SEG; k++)
				{
					const int c = sg * SEG + k;
					if (c < Tc)
						row[2 * c] = (int16_t)ev[k], row[2 * c + 1] = (int16_t)od[k];
				}
			}
		}
		__syncthreads();
	}
}

// ---- kernels ------------------------------------------------------------------------------------

__device__ __forceinline__ int16_t* tail_plane_base(const TailParams& P, uint32_t image, uint64_t inst, uint32_t p,
                                                    const TileDesc& td)
{
	int16_t* base = P.plane + (P.plane_tiled ? (uint64_t)image : inst) * P.plane_inst_stride +
	                (uint64_t)p * P.plane_plane_stride;
	if (P.plane_tiled)
		base += (uint64_t)td.y0 * P.plane_pitch + td.x0;
	return base;
}

__global__ __launch_bounds__(SEGT_THREADS) void k_forward_tail_seg(const TailParams P)
{
	extern __shared__ __attribute__((aligned(16))) int16_t A[];
	const int tid = threadIdx.x;
	const uint32_t p = blockIdx.x % P.channels;
	const uint64_t inst = blockIdx.x / P.channels;
	const uint32_t tile = (uint32_t)(inst % P.n_tiles), image = (uint32_t)(inst / P.n_tiles);
	const TileDesc td = P.tiles[tile];
	int16_t* tile_stream = P.stream + (uint64_t)image * P.stream_stride + td.stream_off;
	const int wrap = P.wrap, LP = (int)P.pitch;
	const int m = (p == 0) ? 0 : 1;

	// plane -> LDS (8 loads in flight per thread)
	{
		const int cw = (int)P.lv[0].cw, chh = (int)P.lv[0].ch;
		const int16_t* src = tail_plane_base(P, image, inst, p, td);
		const FastDiv dW(cw);
		for (int base = tid; base < cw * chh; base += 8 * SEGT_THREADS)
		{
			int16_t val[8];
#pragma unroll
			for (int k = 0; k < 8; k++)
			{
				const int idx = base + k * SEGT_THREADS;
				val[k] = 0;
				if (idx < cw * chh)
				{
					const int y = dW.div(idx), x = idx - y * cw;
					val[k] = src[(uint64_t)y * P.plane_pitch + x];
				}
			}
#pragma unroll
			for (int k = 0; k < 8; k++)
			{
				const int idx = base + k * SEGT_THREADS;
				if (idx < cw * chh)
				{
					const int y = dW.div(idx), x = idx - y * cw;
					A[y * LP + x] = val[k];
				}
			}
		}
		__syncthreads();
	}

	for (uint32_t l = 0; l < P.nlev; l++)
	{
		const TailLevel& L = P.lv[l];
		const int Tc = (int)L.tw, Tr = (int)L.th;
		if (L.kind == K_DD137)
			tail_forward_level<K_DD137>(A, LP, (int)L.cw, (int)L.ch, Tc, Tr, wrap, tid);
		else if (L.kind == K_CDF53)
			tail_forward_level<K_CDF53>(A, LP, (int)L.cw, (int)L.ch, Tc, Tr, wrap, tid);
		else
			tail_forward_level<K_HAAR>(A, LP, (int)L.cw, (int)L.ch, Tc, Tr, wrap, tid);

		// sub-bands out: head, C, B, D (gate + quantize); the LL stays in place for the next level
		int16_t* grp = tile_stream + L.grp0 + (uint64_t)p * L.gsize;
		const int nsub = Tc * Tr;
		const int q = L.q[m], g = L.g[m];
		const float rq = L.rq[m];
		if (tid == 0)
			grp[0] = (int16_t)q;
		const FastDiv dT(Tc);
		for (int idx = tid; idx < nsub; idx += SEGT_THREADS)
		{
			const int r = dT.div(idx), c = idx - r * Tc;
			grp[1 + idx] = quantize(A[(Tr + r) * LP + c], q, g, rq);
			grp[1 + nsub + idx] = quantize(A[r * LP + Tc + c], q, g, rq);
			grp[1 + 2 * nsub + idx] = quantize(A[(Tr + r) * LP + Tc + c], q, g, rq);
		}
		// the next level rewrites the LL corner in place -- and one column / row beyond it when that
		// extent is odd, i.e. the first column of B / first row of C: finish reading them first
		__syncthreads();
	}

	// final low-pass
	{
		const int fw = (int)P.fw, fh = (int)P.fh;
		int16_t* lp_out = tile_stream + (uint64_t)p * fw * fh;
		for (int idx = tid; idx < fw * fh; idx += SEGT_THREADS)
			lp_out[idx] = A[(idx / fw) * LP + (idx % fw)];
	}
}

__global__ __launch_bounds__(SEGT_THREADS) void k_inverse_tail_seg(const TailParams P)
{
	extern __shared__ __attribute__((aligned(16))) int16_t A[];
	const int tid = threadIdx.x;
	const uint32_t p = blockIdx.x % P.channels;
	const uint64_t inst = blockIdx.x / P.channels;
	const uint32_t tile = (uint32_t)(inst % P.n_tiles), image = (uint32_t)(inst / P.n_tiles);
	const TileDesc td = P.tiles[tile];
	const int16_t* tile_stream = P.stream + (uint64_t)image * P.stream_stride + td.stream_off;
	const int wrap = P.wrap, LP = (int)P.pitch;

	// final low-pass -> LL corner
	{
		const int fw = (int)P.fw, fh = (int)P.fh;
		const int16_t* lp_in = tile_stream + (uint64_t)p * fw * fh;
		for (int idx = tid; idx < fw * fh; idx += SEGT_THREADS)
			A[(idx / fw) * LP + (idx % fw)] = lp_in[idx];
	}

	// smallest level first (misc.c:257-285)
	for (int l = (int)P.nlev - 1; l >= 0; l--)
	{
		const TailLevel& L = P.lv[l];
		const int Tc = (int)L.tw, Tr = (int)L.th;
		const int16_t* grp = tile_stream + L.grp0 + (uint64_t)p * L.gsize;
		const int nsub = Tc * Tr;
		const int q = grp[0];  // the decoder trusts the lift head (misc.c:266-272, lifting.c:114-116)

		// C, B, D in (de-quantized: lifting.c:30-40), 8 loads in flight per thread
		const FastDiv dT(Tc);
		for (int base = tid; base < 3 * nsub; base += 8 * SEGT_THREADS)
		{
			int16_t val[8];
#pragma unroll
			for (int k = 0; k < 8; k++)
			{
				const int idx = base + k * SEGT_THREADS;
				val[k] = (idx < 3 * nsub) ? grp[1 + idx] : (int16_t)0;
			}
#pragma unroll
			for (int k = 0; k < 8; k++)
			{
				const int idx = base + k * SEGT_THREADS;
				if (idx < 3 * nsub)
				{
					const int sel = (idx >= 2 * nsub) ? 2 : ((idx >= nsub) ? 1 : 0);  // stream order C, B, D
					const int o = idx - sel * nsub;
					const int r = dT.div(o), c = o - r * Tc;
					const int16_t v = (q > 1) ? (int16_t)((int)val[k] * q) : val[k];
					const int rr = (sel == 1) ? r : Tr + r, cc = (sel == 0) ? c : Tc + c;
					A[rr * LP + cc] = v;
				}
			}
		}
		__syncthreads();

		if (L.kind == K_DD137)
			tail_inverse_level<K_DD137>(A, LP, Tc, Tr, wrap, tid);
		else if (L.kind == K_CDF53)
			tail_inverse_level<K_CDF53>(A, LP, Tc, Tr, wrap, tid);
		else
			tail_inverse_level<K_HAAR>(A, LP, Tc, Tr, wrap, tid);
	}

	// LDS -> plane (the level's true extent: phantom row / column dropped, lifting.c:111-112,141)
	{
		const int ow = (int)P.lv[0].cw, oh = (int)P.lv[0].ch;
		int16_t* dst = tail_plane_base(P, image, inst, p, td);
		const FastDiv dW(ow);
		for (int idx = tid; idx < ow * oh; idx += SEGT_THREADS)
		{
			const int y = dW.div(idx), x = idx - y * ow;
			dst[(uint64_t)y * P.plane_pitch + x] = A[y * LP + x];
		}
	}
}

}  // namespace ako
