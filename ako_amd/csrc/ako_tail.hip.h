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
#include "ako_tail_params.h"
#include "ako_stream.hip.h"  // lift_add / sum_p / sum_u / shift_p / shift_u (segment engine)

namespace ako
{

constexpr int TAIL_MAX = 128;     // largest level extent handled here
constexpr int TAIL_THREADS = 1024;
// window of the largest tail level (sub-band extent TAIL_MAX / 2 = 64): (2 * (64 + 6)) rows of
// 2 * (64 + 8) samples; the dense LL array holds at most 64 x 64 samples
constexpr int TAIL_WROWS = 2 * (TAIL_MAX / 2 + 6);
constexpr int TAIL_WCOLS = 2 * (TAIL_MAX / 2 + 8);
constexpr int TAIL_LDS_BYTES = (TAIL_WROWS * TAIL_WCOLS + (TAIL_MAX / 2) * (TAIL_MAX / 2)) * 2;

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


}  // namespace ako
