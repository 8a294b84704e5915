// ako_kernels.hip.h -- gfx950 kernels of the tile-wise transform path (generic "window" engine).
//
// One workgroup (256 threads = 4 wave64) transforms ONE 2-D level of a TWxTH block of
// sub-band coefficients of one plane (or of up to four interleaved u8 channels on the first /
// last level, so that the colour transform sees R, G and B together):
//
//   forward  (reference: library/format.c:64-134, library/lifting.c:43-76,154-168,171-292)
//     HBM -> LDS window (2*(TW+8) x 2*(TH+6) samples incl. halo, colour transform fused on level 0)
//     rows:    predict, barrier, update      (wavelet-dd137.c:57 / wavelet-cdf53.c:57 / wavelet-haar.c:30)
//     columns: predict, barrier, update      (wavelet-dd137.c:212 / wavelet-cdf53.c:126 / wavelet-haar.c:58)
//     LDS -> HBM: LL to the next level's plane, C / B / D gated + quantized straight to their
//     final offsets in the coefficient stream, lift head written by one lane
//   inverse  (reference: library/lifting.c:86-148, library/format.c:138-311)
//     stream (de-quantized on load) + LL -> LDS window, columns: even, odd; rows: even, odd;
//     LDS -> HBM dense plane, or on the last level colour inverse + saturate + u8 interleave
//
// Lifting is done IN PLACE in the LDS window: coefficient slot j of a line keeps its even sample at
// element 2j and its odd sample at 2j+1; a predict-type step rewrites odd elements from even taps,
// an update-type step rewrites even elements from odd taps.  Boundary handling follows SURVEY A.2:
//   * the window halo is filled at load time by an index map (CLAMP / MIRROR: nearest index,
//     REPEAT: modulo, ZERO: zeros); the phantom last odd sample / row of an odd-sized level is a
//     copy of the last even one (wavelet-dd137.c:128-132, lifting.c:70-72)
//   * the first step of a pass ("stage A") also fills the halo slots of the sequence it produces:
//     CLAMP / MIRROR recompute the nearest in-range coefficient, REPEAT computes plainly on the
//     periodic data, ZERO stores 0
//   * MIRROR's far taps take the opposite near tap
// All arithmetic is int32 with truncating division, narrowed to int16 on every store, exactly as
// the reference's int16_t helper functions do (wavelet-dd137.c:36-54, wavelet-cdf53.c:36-54).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ako
{

constexpr int TW = 32;           // sub-band columns per workgroup
constexpr int TH = 32;           // sub-band rows per workgroup
constexpr int NSC = TW + 8;      // column slots in the window (origin c0 - 4)
constexpr int NSR = TH + 6;      // row slots in the window (origin r0 - 3)
constexpr int WCOLS = 2 * NSC;   // 80 samples
constexpr int WROWS = 2 * NSR;   // 76 samples
constexpr int WPITCH = WCOLS;    // int16 elements per window row
constexpr int WPLANE = WROWS * WPITCH;
constexpr int ORG_C = 4;         // window column slot 0 is coefficient c0 - ORG_C
constexpr int ORG_R = 3;
constexpr int THREADS = 256;
constexpr int MAX_CH = 16;

enum : int { K_DD137 = 0, K_CDF53 = 1, K_HAAR = 2 };
enum : int { W_CLAMP = 0, W_MIRROR = 1, W_REPEAT = 2, W_ZERO = 3 };
enum : int { C_YCOCG = 0, C_SUBG = 1, C_NONE = 2, C_YCOCG_Q = 3 };

struct TileDesc
{
	uint32_t x0, y0;      // tile origin in the image (pixels)
	uint64_t stream_off;  // int16 offset of the tile's stream inside the image's stream
};

struct LevelParams
{
	// geometry of this level: "full" = the larger side (forward input / inverse output),
	// "sub" = the four sub-bands
	uint32_t full_w, full_h;
	uint32_t sub_w, sub_h;
	int32_t wrap;
	uint32_t channels;
	uint32_t planes_per_wg;  // 1, or up to 4 on the u8 side
	uint32_t plane_groups;   // ceil(channels / planes_per_wg)
	uint32_t grid_x, grid_y; // workgroups per plane
	// tiles of this group, per image
	const TileDesc* tiles;
	uint32_t n_tiles;
	uint32_t batch;
	// u8 image side (first forward / last inverse level)
	uint8_t* img;
	uint64_t img_stride;     // bytes per image
	uint32_t img_pitch;      // pixels per image row
	int32_t color;
	int32_t discard;
	// dense int16 plane side.  element address =
	//   base + inst * inst_stride + plane * plane_stride + tile_off(x0,y0) + y * pitch + x
	const int16_t* src;      // forward: level input; inverse: LL input
	uint64_t src_inst_stride, src_plane_stride;
	uint32_t src_pitch;
	uint32_t src_tiled;      // 1: add y0 * pitch + x0 (PLANES_I16 mode on the image side)
	int16_t* dst;            // forward: LL output; inverse: level output
	uint64_t dst_inst_stride, dst_plane_stride;
	uint32_t dst_pitch;
	uint32_t dst_tiled;
	// coefficient stream
	int16_t* stream;
	uint64_t stream_stride;  // int16 per image
	uint32_t ll_in_stream;   // inverse: LL comes from the stream's low-pass section
	uint32_t ll_out_stream;  // forward: LL goes to the stream's low-pass section
	uint64_t lp_off[MAX_CH];   // int16 offset of plane p's final low-pass inside the tile stream
	uint64_t grp_off[MAX_CH];  // int16 offset of plane p's [head C B D] group of this level
	int32_t q_luma, g_luma, q_chroma, g_chroma;
	float rq_luma, rq_chroma;  // (1/q) * (1 + 1e-6): see quantize()
	uint32_t dbg;              // AKO_HIP_DBG: bit 2 switches the XCD-aware workgroup order of the streaming kernels off
	int32_t* ovf_flag;         // optimistic-float inverse: set when a value may have left int16 (see ako_stream.hip.h)
	int32_t ovf_gen;           // ... to this launch's generation number (> every earlier one: the flag is never reset)
};

// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ int tdiv(int s, int shift)  // C truncating division by 2^shift
{
	return (s + ((s >> 31) & ((1 << shift) - 1))) >> shift;
}

// load-time halo map of a coefficient index: returns -1 for "reads zero".
// Only v in [-3, T + 2] is ever consumed; anything further out just has to stay a valid index.
// REPEAT is done with two conditional adds / subtracts (enough for that range with T >= 2) instead
// of an integer modulo, which costs ~40 instructions on this hardware.
__device__ __forceinline__ int map_index(int v, int T, int wrap)
{
	if ((unsigned)v < (unsigned)T)
		return v;
	if (wrap == W_ZERO)
		return -1;
	if (wrap == W_REPEAT)
	{
		if (v < 0)
		{
			v += T;
			if (v < 0)
				v += T;
		}
		else
		{
			v -= T;
			if (v >= T)
				v -= T;
		}
		return min(max(v, 0), T - 1);
	}
	return (v < 0) ? 0 : T - 1;
}

// One lifting step on slot j (global coefficient index v) of a line whose element stride is S.
//   PTYPE  : true  = predict-like (rewrites the odd element from even taps  c-1, c, c+1, c+2)
//            false = update-like  (rewrites the even element from odd taps  c-2, c-1, c, c+1)
//   SGN    : +1 forward, -1 inverse
//   STAGEA : first step of a pass: also produces halo slots (see file header)
template <int KIND, bool PTYPE, int SGN, bool STAGEA>
__device__ __forceinline__ void lift_step(int16_t* line, int S, int j, int v, int org, int T, int wrap)
{
	int16_t* tgt = line + (2 * j + (PTYPE ? 1 : 0)) * S;
	int ve = v;
	if (STAGEA)
	{
		if ((unsigned)v >= (unsigned)T)
		{
			if (wrap == W_ZERO)
			{
				*tgt = 0;
				return;
			}
			if (wrap != W_REPEAT)
				ve = (v < 0) ? 0 : T - 1;
		}
	}
	else if ((unsigned)v >= (unsigned)T)
		return;

	const int je = ve - org;
	const int16_t* taps = line + (PTYPE ? 0 : 1) * S;  // tap k of the other parity: taps[2 * k * S]
#define TAP(k) ((int)taps[2 * (k)*S])
	int delta;
	if (KIND == K_HAAR)
	{
		delta = PTYPE ? -TAP(je) : 0;
		if (!PTYPE)
			return;
	}
	else if (KIND == K_CDF53)
	{
		if (PTYPE)
			delta = -tdiv(TAP(je) + TAP(je + 1), 1);
		else
			delta = tdiv(TAP(je - 1) + TAP(je), 2);
	}
	else
	{
		const bool mirror = (wrap == W_MIRROR);
		if (PTYPE)
		{
			const int l1 = TAP(je - 1), e = TAP(je), p1 = TAP(je + 1);
			const int p2 = (mirror && ve + 2 >= T) ? l1 : TAP(je + 2);
			delta = tdiv(l1 + p2 - 9 * (e + p1), 4);
		}
		else
		{
			const int l1 = TAP(je - 1), h = TAP(je), p1 = TAP(je + 1);
			const int l2 = (mirror && ve < 2) ? p1 : TAP(je - 2);
			delta = tdiv(-l2 - p1 + 9 * (l1 + h), 5);
		}
	}
#undef TAP
	*tgt = (int16_t)((int)*tgt + SGN * delta);
}

// gate + quantize (lifting.c:163): exact truncating x / q through one float multiply.
// rq = (1/q)(1 + 1e-6): for |x| <= 32768, 1 <= q <= 32765 the product lies in [k, k+1) whenever
// trunc(|x| / q) = k  (error analysis in DESIGN.md; swept in tests/test_host_logic.py).
__device__ __forceinline__ int16_t quantize(int v, int q, int g, float rq)
{
	const int a = (v < 0) ? -v : v;
	int r = v;
	if (q > 1)
		r = (int)((float)v * rq);
	return (int16_t)((a > g) ? r : 0);
}

__device__ __forceinline__ int sat8(int v)
{
	return (v > 0) ? ((v < 255) ? v : 255) : 0;
}

// decode the 1-D block index: (bx, by, plane group, tile, image)
struct BlockId
{
	uint32_t bx, by, pg, tile, image;
};

__device__ __forceinline__ BlockId decode_block(const LevelParams& P)
{
	uint64_t b = blockIdx.x;
	BlockId id;
	id.bx = (uint32_t)(b % P.grid_x);
	b /= P.grid_x;
	id.by = (uint32_t)(b % P.grid_y);
	b /= P.grid_y;
	id.pg = (uint32_t)(b % P.plane_groups);
	b /= P.plane_groups;
	id.tile = (uint32_t)(b % P.n_tiles);
	id.image = (uint32_t)(b / P.n_tiles);
	return id;
}

// ---------------------------------------------------------------------------------------------
// Forward level
// ---------------------------------------------------------------------------------------------

template <int KIND, bool FIRST_U8>
__global__ __launch_bounds__(THREADS) void k_forward_level(const LevelParams P)
{
	extern __shared__ __attribute__((aligned(16))) int16_t smem[];
	const int tid = threadIdx.x;
	const BlockId id = decode_block(P);
	const TileDesc td = P.tiles[id.tile];
	const uint64_t inst = (uint64_t)id.image * P.n_tiles + id.tile;

	const int Tc = (int)P.sub_w, Tr = (int)P.sub_h;
	const int cw = (int)P.full_w, chh = (int)P.full_h;
	const int c0 = (int)id.bx * TW, r0 = (int)id.by * TH;
	const int wrap = P.wrap;
	const int p_first = (int)(id.pg * P.planes_per_wg);
	const int npl = min((int)P.planes_per_wg, (int)P.channels - p_first);

	// ---- P0: HBM -> LDS window ---------------------------------------------------------------
	for (int idx = tid; idx < WROWS * WCOLS; idx += THREADS)
	{
		const int wr = idx / WCOLS, wc = idx - wr * WCOLS;
		const int mr = map_index(r0 - ORG_R + (wr >> 1), Tr, wrap);
		const int mc = map_index(c0 - ORG_C + (wc >> 1), Tc, wrap);
		const bool zero = (mr < 0) || (mc < 0);
		int y = 2 * mr + (wr & 1), x = 2 * mc + (wc & 1);
		if (y >= chh)
			y = chh - 1;  // phantom last row = copy of the last row
		if (x >= cw)
			x = cw - 1;   // phantom last odd sample = last even sample

		if (FIRST_U8)
		{
			int v[4] = {0, 0, 0, 0};
			if (!zero)
			{
				const uint8_t* px = P.img + (uint64_t)id.image * P.img_stride +
				                    ((uint64_t)(td.y0 + y) * P.img_pitch + (td.x0 + x)) * P.channels + p_first;
				if (P.channels == 4)
				{
					const uchar4 t = *reinterpret_cast<const uchar4*>(px);
					v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
				}
				else
					for (int k = 0; k < npl; k++)
						v[k] = px[k];

				if (P.discard && (P.channels == 2 || P.channels == 4) && v[P.channels - 1] == 0)
					for (int k = 0; k + 1 < (int)P.channels; k++)
						v[k] = 0;  // format.c:38-49

				if (id.pg == 0 && P.channels >= 3 && P.color != C_NONE)
				{
					const int r = v[0], g = v[1], b = v[2];
					if (P.color == C_SUBG)
					{
						v[0] = g, v[1] = r - g, v[2] = b - g;
					}
					else
					{
						const int co = r - b;
						const int t = b + tdiv(co, 1);
						const int cg = g - t;
						const int yy = t + tdiv(cg, 1);
						v[0] = (P.color == C_YCOCG_Q) ? yy * 2 : yy;
						v[1] = co, v[2] = cg;
					}
				}
			}
			for (int k = 0; k < npl; k++)
				smem[k * WPLANE + idx] = (int16_t)v[k];
		}
		else
		{
			int16_t s = 0;
			if (!zero)
			{
				// scratch planes are per tile instance; in PLANES_I16 mode the source is the image itself
				const int16_t* base = P.src + (P.src_tiled ? (uint64_t)id.image : inst) * P.src_inst_stride +
				                      (uint64_t)p_first * P.src_plane_stride;
				if (P.src_tiled)
					base += (uint64_t)td.y0 * P.src_pitch + td.x0;
				s = base[(uint64_t)y * P.src_pitch + x];
			}
			smem[idx] = s;
		}
	}
	__syncthreads();

	for (int k = 0; k < npl; k++)
	{
		int16_t* W = smem + k * WPLANE;

		// ---- rows: predict over slots [c0-2, c0+TW], every window row --------------------------
		for (int idx = tid; idx < WROWS * (TW + 3); idx += THREADS)
		{
			const int wr = idx / (TW + 3), jj = idx - wr * (TW + 3);
			lift_step<KIND, true, +1, true>(W + wr * WPITCH, 1, jj + 2, c0 - 2 + jj, c0 - ORG_C, Tc, wrap);
		}
		__syncthreads();
		// ---- rows: update over [c0, c0+TW) -----------------------------------------------------
		if (KIND != K_HAAR)
		{
			for (int idx = tid; idx < WROWS * TW; idx += THREADS)
			{
				const int wr = idx / TW, jj = idx - wr * TW;
				lift_step<KIND, false, +1, false>(W + wr * WPITCH, 1, jj + ORG_C, c0 + jj, c0 - ORG_C, Tc, wrap);
			}
			__syncthreads();
		}
		// ---- columns: predict over row slots [r0-2, r0+TH], the 2*TW net columns ---------------
		for (int idx = tid; idx < (TH + 3) * (2 * TW); idx += THREADS)
		{
			const int ii = idx / (2 * TW), x = idx - ii * (2 * TW);
			lift_step<KIND, true, +1, true>(W + 2 * ORG_C + x, WPITCH, ii + 1, r0 - 2 + ii, r0 - ORG_R, Tr, wrap);
		}
		__syncthreads();
		// ---- columns: update over [r0, r0+TH) --------------------------------------------------
		if (KIND != K_HAAR)
		{
			for (int idx = tid; idx < TH * (2 * TW); idx += THREADS)
			{
				const int ii = idx / (2 * TW), x = idx - ii * (2 * TW);
				lift_step<KIND, false, +1, false>(W + 2 * ORG_C + x, WPITCH, ii + ORG_R, r0 + ii, r0 - ORG_R, Tr,
				                                  wrap);
			}
			__syncthreads();
		}

		// ---- LDS -> HBM ------------------------------------------------------------------------
		const int p = p_first + k;
		const int q = (p == 0) ? P.q_luma : P.q_chroma;
		const int g = (p == 0) ? P.g_luma : P.g_chroma;
		const float rq = (p == 0) ? P.rq_luma : P.rq_chroma;
		int16_t* tile_stream = P.stream + (uint64_t)id.image * P.stream_stride + td.stream_off;
		int16_t* grp = tile_stream + P.grp_off[p];
		const uint64_t nsub = (uint64_t)Tc * Tr;

		int16_t* ll;
		uint32_t ll_pitch = P.sub_w;
		if (P.ll_out_stream)
			ll = tile_stream + P.lp_off[p];
		else
		{
			ll = P.dst + inst * P.dst_inst_stride + (uint64_t)p * P.dst_plane_stride;
			ll_pitch = P.dst_pitch;
		}

		if (id.bx == 0 && id.by == 0 && tid == 0)
			grp[0] = (int16_t)q;  // lift head (lifting.c:266-267)

		for (int idx = tid; idx < TH * TW; idx += THREADS)
		{
			const int ii = idx / TW, jj = idx - ii * TW;
			const int r = r0 + ii, c = c0 + jj;
			if (r < Tr && c < Tc)
			{
				const int16_t* cell = W + (2 * (ii + ORG_R)) * WPITCH + 2 * (jj + ORG_C);
				const uint64_t o = (uint64_t)r * Tc + c;
				ll[(uint64_t)r * ll_pitch + c] = cell[0];
				grp[1 + o] = quantize(cell[WPITCH], q, g, rq);                // C = (HP rows, LP cols)
				grp[1 + nsub + o] = quantize(cell[1], q, g, rq);              // B = (LP rows, HP cols)
				grp[1 + 2 * nsub + o] = quantize(cell[WPITCH + 1], q, g, rq); // D
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------
// Inverse level
// ---------------------------------------------------------------------------------------------

template <int KIND, bool LAST_U8>
__global__ __launch_bounds__(THREADS) void k_inverse_level(const LevelParams P)
{
	extern __shared__ __attribute__((aligned(16))) int16_t smem[];
	const int tid = threadIdx.x;
	const BlockId id = decode_block(P);
	const TileDesc td = P.tiles[id.tile];
	const uint64_t inst = (uint64_t)id.image * P.n_tiles + id.tile;

	const int Tc = (int)P.sub_w, Tr = (int)P.sub_h;
	const int ow = (int)P.full_w, oh = (int)P.full_h;
	const int c0 = (int)id.bx * TW, r0 = (int)id.by * TH;
	const int wrap = P.wrap;
	const int p_first = (int)(id.pg * P.planes_per_wg);
	const int npl = min((int)P.planes_per_wg, (int)P.channels - p_first);
	const int16_t* tile_stream = P.stream + (uint64_t)id.image * P.stream_stride + td.stream_off;
	const uint64_t nsub = (uint64_t)Tc * Tr;

	for (int k = 0; k < npl; k++)
	{
		int16_t* W = smem + k * WPLANE;
		const int p = p_first + k;
		const int16_t* grp = tile_stream + P.grp_off[p];
		const int q = grp[0];  // the decoder trusts the lift head (misc.c:266-272, lifting.c:114-116)

		const int16_t* ll;
		uint32_t ll_pitch = P.sub_w;
		if (P.ll_in_stream)
			ll = tile_stream + P.lp_off[p];
		else
		{
			ll = P.src + inst * P.src_inst_stride + (uint64_t)p * P.src_plane_stride;
			ll_pitch = P.src_pitch;
		}

		// ---- P0: stream + LL -> LDS window (interleaved quadrants, de-quantized) ---------------
		for (int idx = tid; idx < WROWS * WCOLS; idx += THREADS)
		{
			const int wr = idx / WCOLS, wc = idx - wr * WCOLS;
			const int mr = map_index(r0 - ORG_R + (wr >> 1), Tr, wrap);
			const int mc = map_index(c0 - ORG_C + (wc >> 1), Tc, wrap);
			int16_t s = 0;
			if (mr >= 0 && mc >= 0)
			{
				const int quad = (wr & 1) * 2 + (wc & 1);  // 0 LL, 1 B, 2 C, 3 D
				if (quad == 0)
					s = ll[(uint64_t)mr * ll_pitch + mc];
				else
				{
					const uint64_t sel = (quad == 2) ? 0 : ((quad == 1) ? 1 : 2);  // stream order C, B, D
					const int cv = grp[1 + sel * nsub + (uint64_t)mr * Tc + mc];
					s = (q > 1) ? (int16_t)(cv * q) : (int16_t)cv;  // lifting.c:30-40
				}
			}
			W[idx] = s;
		}
	}
	__syncthreads();

	for (int k = 0; k < npl; k++)
	{
		int16_t* W = smem + k * WPLANE;
		// ---- columns: evens over row slots [r0-1, r0+TH+1], every window column ----------------
		if (KIND != K_HAAR)
		{
			for (int idx = tid; idx < (TH + 3) * WCOLS; idx += THREADS)
			{
				const int ii = idx / WCOLS, x = idx - ii * WCOLS;
				lift_step<KIND, false, -1, true>(W + x, WPITCH, ii + 2, r0 - 1 + ii, r0 - ORG_R, Tr, wrap);
			}
			__syncthreads();
		}
		// ---- columns: odds over [r0, r0+TH) ----------------------------------------------------
		for (int idx = tid; idx < TH * WCOLS; idx += THREADS)
		{
			const int ii = idx / WCOLS, x = idx - ii * WCOLS;
			lift_step<KIND, true, -1, false>(W + x, WPITCH, ii + ORG_R, r0 + ii, r0 - ORG_R, Tr, wrap);
		}
		__syncthreads();
		// ---- rows: evens over slots [c0-1, c0+TW+1], the 2*TH net rows -------------------------
		if (KIND != K_HAAR)
		{
			for (int idx = tid; idx < (2 * TH) * (TW + 3); idx += THREADS)
			{
				const int y = idx / (TW + 3), jj = idx - y * (TW + 3);
				lift_step<KIND, false, -1, true>(W + (2 * ORG_R + y) * WPITCH, 1, jj + 3, c0 - 1 + jj, c0 - ORG_C, Tc,
				                                 wrap);
			}
			__syncthreads();
		}
		// ---- rows: odds over [c0, c0+TW) -------------------------------------------------------
		for (int idx = tid; idx < (2 * TH) * TW; idx += THREADS)
		{
			const int y = idx / TW, jj = idx - y * TW;
			lift_step<KIND, true, -1, false>(W + (2 * ORG_R + y) * WPITCH, 1, jj + ORG_C, c0 + jj, c0 - ORG_C, Tc,
			                                 wrap);
		}
		__syncthreads();
	}

	// ---- LDS -> HBM ----------------------------------------------------------------------------
	for (int idx = tid; idx < (2 * TH) * (2 * TW); idx += THREADS)
	{
		const int yy = idx / (2 * TW), xx = idx - yy * (2 * TW);
		const int y = 2 * r0 + yy, x = 2 * c0 + xx;
		if (y >= oh || x >= ow)
			continue;  // phantom row / column dropped (lifting.c:111-112,141)
		const int16_t* cell = smem + (2 * ORG_R + yy) * WPITCH + 2 * ORG_C + xx;

		if (LAST_U8)
		{
			int v[4];
			for (int k = 0; k < npl; k++)
				v[k] = cell[k * WPLANE];
			if (id.pg == 0 && P.channels >= 3 && P.color != C_NONE)
			{
				int r, g, b;
				if (P.color == C_SUBG)
				{
					r = (int16_t)(v[1] + v[0]), g = v[0], b = (int16_t)(v[2] + v[0]);
				}
				else
				{
					const int yv = (P.color == C_YCOCG_Q) ? tdiv(v[0], 1) : v[0];
					const int t = (int16_t)(yv - tdiv(v[2], 1));
					g = (int16_t)(v[2] + t);
					b = (int16_t)(t - tdiv(v[1], 1));
					r = (int16_t)(b + v[1]);
				}
				v[0] = r, v[1] = g, v[2] = b;
			}
			uint8_t* px = P.img + (uint64_t)id.image * P.img_stride +
			              ((uint64_t)(td.y0 + y) * P.img_pitch + (td.x0 + x)) * P.channels + p_first;
			if (P.channels == 4)
				*reinterpret_cast<uchar4*>(px) =
				    make_uchar4((uint8_t)sat8(v[0]), (uint8_t)sat8(v[1]), (uint8_t)sat8(v[2]), (uint8_t)sat8(v[3]));
			else
				for (int k = 0; k < npl; k++)
					px[k] = (uint8_t)sat8(v[k]);
		}
		else
		{
			int16_t* out = P.dst + (P.dst_tiled ? (uint64_t)id.image : inst) * P.dst_inst_stride +
			               (uint64_t)p_first * P.dst_plane_stride;
			if (P.dst_tiled)
				out += (uint64_t)td.y0 * P.dst_pitch + td.x0;
			out[(uint64_t)y * P.dst_pitch + x] = cell[0];
		}
	}
}

// ---------------------------------------------------------------------------------------------
// Format-only kernels: wavelet NONE, and tiles too small to have a single lift
// (stream = planar int16, plane pitch w*h: encode.c:127-128,151; misc.c:245-252)
// ---------------------------------------------------------------------------------------------

struct FormatParams
{
	uint32_t tile_w, tile_h, channels;
	const TileDesc* tiles;
	uint32_t n_tiles, batch;
	uint8_t* img;
	uint64_t img_stride;
	uint32_t img_pitch;
	int32_t color, discard;
	int16_t* stream;
	uint64_t stream_stride;
	// planar != 0: 'stream' is a planar int16 IMAGE instead ([image][plane][h][w], row pitch img_pitch) --
	// the staging form in front of / behind the int16 streaming kernels for channel counts other than 4
	uint32_t planar;
	uint64_t plane_stride;
};

// where pixel i of a tile lives on the int16 side, and the distance between its channels
__device__ __forceinline__ int16_t* format_plane_address(const FormatParams& P, const TileDesc& td, uint32_t image,
                                                         uint64_t i, uint32_t x, uint32_t y, uint64_t npx, uint64_t& kstride)
{
	if (P.planar)
	{
		kstride = P.plane_stride;
		return P.stream + (uint64_t)image * P.channels * P.plane_stride + (uint64_t)(td.y0 + y) * P.img_pitch + td.x0 + x;
	}
	kstride = npx;
	return P.stream + (uint64_t)image * P.stream_stride + td.stream_off + i;
}

template <int UNUSED = 0>  // (a template so that every translation unit including this header may hold it)
__global__ __launch_bounds__(THREADS) void k_format_forward(const FormatParams P)
{
	const uint64_t npx = (uint64_t)P.tile_w * P.tile_h;
	const uint64_t blocks_per_tile = (npx + THREADS - 1) / THREADS;
	const uint64_t inst = blockIdx.x / blocks_per_tile;
	const uint64_t i = (blockIdx.x % blocks_per_tile) * THREADS + threadIdx.x;
	if (i >= npx)
		return;
	const uint32_t tile = (uint32_t)(inst % P.n_tiles), image = (uint32_t)(inst / P.n_tiles);
	const TileDesc td = P.tiles[tile];
	const uint32_t y = (uint32_t)(i / P.tile_w), x = (uint32_t)(i % P.tile_w);
	const uint8_t* px = P.img + (uint64_t)image * P.img_stride +
	                    ((uint64_t)(td.y0 + y) * P.img_pitch + (td.x0 + x)) * P.channels;
	uint64_t kstride;
	int16_t* out = format_plane_address(P, td, image, i, x, y, npx, kstride);

	int v[MAX_CH];
	for (uint32_t k = 0; k < P.channels; k++)
		v[k] = px[k];
	if (P.discard && (P.channels == 2 || P.channels == 4) && v[P.channels - 1] == 0)
		for (uint32_t k = 0; k + 1 < P.channels; k++)
			v[k] = 0;
	if (P.channels >= 3 && P.color != C_NONE)
	{
		const int r = v[0], g = v[1], b = v[2];
		if (P.color == C_SUBG)
		{
			v[0] = g, v[1] = r - g, v[2] = b - g;
		}
		else
		{
			const int co = r - b;
			const int t = b + tdiv(co, 1);
			const int cg = g - t;
			const int yy = t + tdiv(cg, 1);
			v[0] = (P.color == C_YCOCG_Q) ? yy * 2 : yy;
			v[1] = co, v[2] = cg;
		}
	}
	for (uint32_t k = 0; k < P.channels; k++)
		out[k * kstride] = (int16_t)v[k];
}

template <int UNUSED = 0>
__global__ __launch_bounds__(THREADS) void k_format_inverse(const FormatParams P)
{
	const uint64_t npx = (uint64_t)P.tile_w * P.tile_h;
	const uint64_t blocks_per_tile = (npx + THREADS - 1) / THREADS;
	const uint64_t inst = blockIdx.x / blocks_per_tile;
	const uint64_t i = (blockIdx.x % blocks_per_tile) * THREADS + threadIdx.x;
	if (i >= npx)
		return;
	const uint32_t tile = (uint32_t)(inst % P.n_tiles), image = (uint32_t)(inst / P.n_tiles);
	const TileDesc td = P.tiles[tile];
	const uint32_t y = (uint32_t)(i / P.tile_w), x = (uint32_t)(i % P.tile_w);
	uint8_t* px = P.img + (uint64_t)image * P.img_stride +
	              ((uint64_t)(td.y0 + y) * P.img_pitch + (td.x0 + x)) * P.channels;
	uint64_t kstride;
	const int16_t* in = format_plane_address(P, td, image, i, x, y, npx, kstride);

	int v[MAX_CH];
	for (uint32_t k = 0; k < P.channels; k++)
		v[k] = in[k * kstride];
	if (P.channels >= 3 && P.color != C_NONE)
	{
		int r, g, b;
		if (P.color == C_SUBG)
		{
			r = (int16_t)(v[1] + v[0]), g = v[0], b = (int16_t)(v[2] + v[0]);
		}
		else
		{
			const int yv = (P.color == C_YCOCG_Q) ? tdiv(v[0], 1) : v[0];
			const int t = (int16_t)(yv - tdiv(v[2], 1));
			g = (int16_t)(v[2] + t);
			b = (int16_t)(t - tdiv(v[1], 1));
			r = (int16_t)(b + v[1]);
		}
		v[0] = r, v[1] = g, v[2] = b;
	}
	for (uint32_t k = 0; k < P.channels; k++)
		px[k] = (uint8_t)sat8(v[k]);
}

// ---- planar staging, vectorised: four pixels per thread (1, 2 or 3 channels, tile width a multiple of 4) ----
// Same arithmetic as k_format_forward / k_format_inverse with P.planar set; 4 * CH bytes in, one 8 byte
// store per plane out (and the reverse).  Row starts may sit at any byte / element offset (odd image
// widths): the dword accesses are made through the unaligned-capable global path like the stream stores.

__device__ __forceinline__ uint32_t pack_i16x2(int lo, int hi)
{
	return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16);
}

template <int CH>
__global__ __launch_bounds__(THREADS) void k_planes_forward4(const FormatParams P)
{
	const uint32_t qw = P.tile_w / 4;
	const uint64_t nq = (uint64_t)qw * P.tile_h;
	const uint64_t blocks_per_tile = (nq + THREADS - 1) / THREADS;
	const uint64_t inst = blockIdx.x / blocks_per_tile;
	const uint64_t i = (blockIdx.x % blocks_per_tile) * THREADS + threadIdx.x;
	if (i >= nq)
		return;
	const uint32_t tile = (uint32_t)(inst % P.n_tiles), image = (uint32_t)(inst / P.n_tiles);
	const TileDesc td = P.tiles[tile];
	const uint32_t y = (uint32_t)(i / qw), x = (uint32_t)(i % qw) * 4;
	const uint64_t at = (uint64_t)(td.y0 + y) * P.img_pitch + td.x0 + x;
	const uint32_t* src = reinterpret_cast<const uint32_t*>(P.img + (uint64_t)image * P.img_stride + at * CH);
	uint32_t w[CH];
#pragma unroll
	for (int k = 0; k < CH; k++)
		w[k] = src[k];
	int v[4][CH];
#pragma unroll
	for (int p = 0; p < 4; p++)
#pragma unroll
		for (int c = 0; c < CH; c++)
		{
			const int j = p * CH + c;
			v[p][c] = (int)((w[j >> 2] >> (8 * (j & 3))) & 255u);
		}
#pragma unroll
	for (int p = 0; p < 4; p++)
	{
		if constexpr (CH == 2)
			if (P.discard && v[p][1] == 0)  // format.c:38-49,76-79
				v[p][0] = 0;
		if constexpr (CH == 3)
		if (P.color != C_NONE)
		{
			const int r = v[p][0], g = v[p][1], b = v[p][2];
			if (P.color == C_SUBG)
				v[p][0] = g, v[p][1] = r - g, v[p][CH - 1] = b - g;
			else
			{
				const int co = r - b;
				const int t = b + tdiv(co, 1);
				const int cg = g - t;
				const int yy = t + tdiv(cg, 1);
				v[p][0] = (P.color == C_YCOCG_Q) ? yy * 2 : yy;
				v[p][1] = co, v[p][CH - 1] = cg;
			}
		}
	}
	int16_t* out = P.stream + (uint64_t)image * CH * P.plane_stride + at;
#pragma unroll
	for (int c = 0; c < CH; c++)
		*reinterpret_cast<uint2*>(out + (uint64_t)c * P.plane_stride) =
		    make_uint2(pack_i16x2(v[0][c], v[1][c]), pack_i16x2(v[2][c], v[3][c]));
}

template <int CH>
__global__ __launch_bounds__(THREADS) void k_planes_inverse4(const FormatParams P)
{
	const uint32_t qw = P.tile_w / 4;
	const uint64_t nq = (uint64_t)qw * P.tile_h;
	const uint64_t blocks_per_tile = (nq + THREADS - 1) / THREADS;
	const uint64_t inst = blockIdx.x / blocks_per_tile;
	const uint64_t i = (blockIdx.x % blocks_per_tile) * THREADS + threadIdx.x;
	if (i >= nq)
		return;
	const uint32_t tile = (uint32_t)(inst % P.n_tiles), image = (uint32_t)(inst / P.n_tiles);
	const TileDesc td = P.tiles[tile];
	const uint32_t y = (uint32_t)(i / qw), x = (uint32_t)(i % qw) * 4;
	const uint64_t at = (uint64_t)(td.y0 + y) * P.img_pitch + td.x0 + x;
	const int16_t* in = P.stream + (uint64_t)image * CH * P.plane_stride + at;
	int v[4][CH];
#pragma unroll
	for (int c = 0; c < CH; c++)
	{
		const uint2 q = *reinterpret_cast<const uint2*>(in + (uint64_t)c * P.plane_stride);
		v[0][c] = (int)(int16_t)(q.x & 0xFFFFu), v[1][c] = (int)q.x >> 16;
		v[2][c] = (int)(int16_t)(q.y & 0xFFFFu), v[3][c] = (int)q.y >> 16;
	}
	uint32_t w[CH];
#pragma unroll
	for (int k = 0; k < CH; k++)
		w[k] = 0;
#pragma unroll
	for (int p = 0; p < 4; p++)
	{
		if constexpr (CH == 3)
		if (P.color != C_NONE)
		{
			int r, g, b;
			if (P.color == C_SUBG)
				r = (int16_t)(v[p][1] + v[p][0]), g = v[p][0], b = (int16_t)(v[p][CH - 1] + v[p][0]);
			else
			{
				const int yv = (P.color == C_YCOCG_Q) ? tdiv(v[p][0], 1) : v[p][0];
				const int t = (int16_t)(yv - tdiv(v[p][CH - 1], 1));
				g = (int16_t)(v[p][CH - 1] + t);
				b = (int16_t)(t - tdiv(v[p][1], 1));
				r = (int16_t)(b + v[p][1]);
			}
			v[p][0] = r, v[p][1] = g, v[p][CH - 1] = b;
		}
#pragma unroll
		for (int c = 0; c < CH; c++)
		{
			const int j = p * CH + c;
			w[j >> 2] |= (uint32_t)sat8(v[p][c]) << (8 * (j & 3));
		}
	}
	uint32_t* dst = reinterpret_cast<uint32_t*>(P.img + (uint64_t)image * P.img_stride + at * CH);
#pragma unroll
	for (int k = 0; k < CH; k++)
		dst[k] = w[k];
}

}  // namespace ako
