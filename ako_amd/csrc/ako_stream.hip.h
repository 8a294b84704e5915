// ako_stream.hip.h -- register-streaming level kernels for gfx950 (the fast path of large levels).
//
// Same arithmetic as the generic window engine (ako_kernels.hip.h), different data movement:
//
//   * NO LDS and NO barriers.  One wave64 owns a vertical strip of 128 coefficient columns
//     (120 net + 4 halo columns on either side) and walks it top to bottom.
//   * every lane owns two adjacent coefficient columns (four samples of a row): on the u8 side
//     that is one 16-byte load of four RGBA pixels per lane and row, on int16 planes one 8-byte load
//   * the horizontal pass takes its neighbour taps from the adjacent lanes with DPP whole-wave
//     shifts (wave_shr:1 / wave_shl:1) -- 6 per row for DD13/7, 2 for CDF5/3
//   * the vertical pass is a software pipeline in registers: each lane keeps, per column, the last
//     three even rows, two odd rows and three high-pass rows; a new row pair in, one finished
//     low-pass row and one high-pass row out
//   * colour transform is fused in front (forward) / behind (inverse); gate + quantization and
//     the stream packing are fused into the stores (forward), de-quantization into the loads
//
// Boundary rules are the same closed form as in the window engine:
//   rows   : the row slot fed to the pipeline is map_index(v) (CLAMP/MIRROR nearest, REPEAT modulo,
//            ZERO zeros); the high-pass halo slots are patched in the pipeline (nearest / zero)
//   columns: REPEAT wraps the lane's load address; CLAMP/MIRROR/ZERO overwrite the out-of-range
//            lanes of the even sequence and then of the high-pass sequence (v_readlane broadcast)
//   MIRROR : the far taps take the opposite near tap
// Levels with an odd width, and everything small, stay on the window engine.
#pragma once

#include "ako_kernels.hip.h"

namespace ako
{

constexpr int SNET = 120;  // net coefficient columns per wave
constexpr int SORG = 4;    // lane 0 holds coefficient columns strip * SNET - SORG, +1

// lane i <- lane i-1 / lane i+1 over the whole wave64 (gfx9 DPP wave shifts)
__device__ __forceinline__ int from_prev_lane(int x)
{
	return __builtin_amdgcn_update_dpp(0, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ int from_next_lane(int x)
{
	return __builtin_amdgcn_update_dpp(0, x, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

template <bool NARROW>
__device__ __forceinline__ int nrw(int v)
{
	return NARROW ? (int)(int16_t)v : v;
}

// what a wave needs to know about the left / right tile border
struct HEdge
{
	bool left, right;  // this strip holds out-of-range lanes on that side (never set for REPEAT)
	bool oob_l, oob_r; // this lane is such a lane
	bool last;         // this lane holds columns T-2, T-1
	bool first;        // this lane holds columns 0, 1
	int lane_first, lane_last;
	int wrap;
};

// Horizontal forward lift of one row: samples (E0 O0 E1 O1) of this lane's two coefficient
// columns -> (L0 L1 H0 H1).  Valid in lanes 2..61.
template <int KIND, bool NARROW>
__device__ __forceinline__ void hlift_forward(int E0, int O0, int E1, int O1, const HEdge& ed, int& L0, int& L1,
                                              int& H0, int& H1)
{
	if (KIND == K_HAAR)
	{
		L0 = E0, L1 = E1;
		H0 = nrw<true>(O0 - E0), H1 = nrw<true>(O1 - E1);
		return;
	}
	if (ed.left)
	{
		const int f = (ed.wrap == W_ZERO) ? 0 : __builtin_amdgcn_readlane(E0, ed.lane_first);
		if (ed.oob_l)
			E0 = f, E1 = f;
	}
	if (ed.right)
	{
		const int f = (ed.wrap == W_ZERO) ? 0 : __builtin_amdgcn_readlane(E1, ed.lane_last);
		if (ed.oob_r)
			E0 = f, E1 = f;
	}

	if (KIND == K_CDF53)
	{
		const int eR0 = from_next_lane(E0);
		H0 = nrw<NARROW>(O0 - tdiv(E0 + E1, 1));
		H1 = nrw<NARROW>(O1 - tdiv(E1 + eR0, 1));
	}
	else
	{
		const int eL = from_prev_lane(E1);
		const int eR0 = from_next_lane(E0);
		const int eR1 = from_next_lane(E1);
		int p2_0 = eR0, p2_1 = eR1;
		if (ed.right && ed.wrap == W_MIRROR && ed.last)
			p2_0 = eL, p2_1 = E0;  // far tap := opposite near tap
		H0 = nrw<NARROW>(O0 + tdiv(eL + p2_0 - 9 * (E0 + E1), 4));
		H1 = nrw<NARROW>(O1 + tdiv(E0 + p2_1 - 9 * (E1 + eR0), 4));
	}

	if (ed.left)
	{
		const int f = (ed.wrap == W_ZERO) ? 0 : __builtin_amdgcn_readlane(H0, ed.lane_first);
		if (ed.oob_l)
			H0 = f, H1 = f;
	}
	if (ed.right)
	{
		const int f = (ed.wrap == W_ZERO) ? 0 : __builtin_amdgcn_readlane(H1, ed.lane_last);
		if (ed.oob_r)
			H0 = f, H1 = f;
	}

	if (KIND == K_CDF53)
	{
		const int hL1 = from_prev_lane(H1);
		L0 = nrw<NARROW>(E0 + tdiv(hL1 + H0, 2));
		L1 = nrw<NARROW>(E1 + tdiv(H0 + H1, 2));
	}
	else
	{
		const int hL0 = from_prev_lane(H0);
		const int hL1 = from_prev_lane(H1);
		const int hR0 = from_next_lane(H0);
		int l2_0 = hL0, l2_1 = hL1;
		if (ed.left && ed.wrap == W_MIRROR && ed.first)
			l2_0 = H1, l2_1 = hR0;
		L0 = nrw<NARROW>(E0 + tdiv(-l2_0 - H1 + 9 * (hL1 + H0), 5));
		L1 = nrw<NARROW>(E1 + tdiv(-l2_1 - hR0 + 9 * (H0 + H1), 5));
	}
}

// vertical pipeline state of one column
struct VCol
{
	int eA, eB, eC;  // E[v-3], E[v-2], E[v-1]
	int oA, oB;      // O[v-2], O[v-1]
	int hA, hB, hC;  // HP[v-5], HP[v-4], HP[v-3]
};

// what a wave needs to know about the top / bottom tile border at slot v
struct VEdge
{
	int wrap, T;
};

// Feed row slot v (even value E, odd value O); returns LP[v-3] and HP[v-3].
template <int KIND, bool NARROW>
__device__ __forceinline__ void vstep_forward(VCol& s, int E, int O, int v, const VEdge& ed, int& lp_out, int& hp_out)
{
	const int u = v - 2;  // high-pass slot produced now
	const int r = v - 3;  // row finished now
	int H;
	if (KIND == K_HAAR)
		H = nrw<true>(s.oA - s.eB);
	else if (KIND == K_CDF53)
		H = nrw<NARROW>(s.oA - tdiv(s.eB + s.eC, 1));
	else
	{
		int p2 = E;
		if (ed.wrap == W_MIRROR && u + 2 >= ed.T)
			p2 = s.eA;
		H = nrw<NARROW>(s.oA + tdiv(s.eA + p2 - 9 * (s.eB + s.eC), 4));
	}
	// halo slots of the high-pass sequence
	if (KIND != K_HAAR && ed.wrap != W_REPEAT)
	{
		if (u >= ed.T)
			H = (ed.wrap == W_ZERO) ? 0 : s.hC;  // HP[T-1] again (u == T is the only such slot consumed)
		if (u < 0 && ed.wrap == W_ZERO)
			H = 0;
		if (u == 0 && ed.wrap != W_ZERO)
			s.hB = H, s.hC = H;  // HP[-2] = HP[-1] = HP[0]
	}

	int L;
	if (KIND == K_HAAR)
		L = s.eA;
	else if (KIND == K_CDF53)
		L = nrw<NARROW>(s.eA + tdiv(s.hB + s.hC, 2));
	else
	{
		int l2 = s.hA;
		if (ed.wrap == W_MIRROR && r < 2)
			l2 = H;
		L = nrw<NARROW>(s.eA + tdiv(-l2 - H + 9 * (s.hB + s.hC), 5));
	}
	lp_out = L;
	hp_out = s.hC;

	s.eA = s.eB, s.eB = s.eC, s.eC = E;
	s.oA = s.oB, s.oB = O;
	s.hA = s.hB, s.hB = s.hC, s.hC = H;
}

struct StreamGeom
{
	uint32_t strips, segs, seg_rows;
};

// unit -> (strip, segment, plane group, tile instance)
struct UnitId
{
	uint32_t strip, seg, pg, tile, image;
	bool valid;
};

__device__ __forceinline__ UnitId decode_unit(const LevelParams& P, const StreamGeom& G)
{
	UnitId id;
	uint64_t u = (uint64_t)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
	const uint64_t total = (uint64_t)G.strips * G.segs * P.plane_groups * P.n_tiles * P.batch;
	id.valid = u < total;
	id.strip = (uint32_t)(u % G.strips);
	u /= G.strips;
	id.seg = (uint32_t)(u % G.segs);
	u /= G.segs;
	id.pg = (uint32_t)(u % P.plane_groups);
	u /= P.plane_groups;
	id.tile = (uint32_t)(u % P.n_tiles);
	id.image = (uint32_t)(u / P.n_tiles);
	return id;
}

__device__ __forceinline__ uint32_t pack2(int lo, int hi)
{
	return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16);
}

// ---------------------------------------------------------------------------------------------
// Forward.  NPL = planes handled by one wave: 4 with U8 (RGBA pixels), 1 on int16 planes.
// ---------------------------------------------------------------------------------------------

template <int KIND, int NPL, bool U8, bool NARROW>
__global__ __launch_bounds__(THREADS) void k_forward_stream(const LevelParams P, const StreamGeom G)
{
	const UnitId id = decode_unit(P, G);
	if (!id.valid)
		return;
	const int lane = threadIdx.x & 63;
	const TileDesc td = P.tiles[id.tile];
	const uint64_t inst = (uint64_t)id.image * P.n_tiles + id.tile;

	const int Tc = (int)P.sub_w, Tr = (int)P.sub_h;
	const int cw = (int)P.full_w, chh = (int)P.full_h;
	const int wrap = P.wrap;
	const int p_first = U8 ? 0 : (int)id.pg;

	// columns of this lane
	const int c_base = (int)id.strip * SNET - SORG;
	const int c0 = c_base + 2 * lane;
	HEdge he;
	he.wrap = wrap;
	he.left = (wrap != W_REPEAT) && (c_base < 0);
	he.right = (wrap != W_REPEAT) && (c_base + 128 > Tc);
	he.oob_l = c0 < 0;
	he.oob_r = c0 >= Tc;
	he.lane_first = SORG / 2;
	he.lane_last = (Tc - 2 - c_base) / 2;
	he.first = (c0 == 0);
	he.last = (c0 == Tc - 2);

	int xs;  // first of this lane's four samples
	if (wrap == W_REPEAT)
	{
		int cm = c0 % Tc;
		if (cm < 0)
			cm += Tc;
		xs = 2 * cm;
	}
	else
		xs = min(max(2 * c0, 0), cw - 4);

	// rows of this wave
	const int r_lo = (int)id.seg * (int)G.seg_rows;
	const int r_hi = min(r_lo + (int)G.seg_rows, Tr);
	VEdge ve;
	ve.wrap = wrap, ve.T = Tr;

	// sources
	const uint8_t* img = nullptr;
	const int16_t* src = nullptr;
	if (U8)
		img = P.img + (uint64_t)id.image * P.img_stride + ((uint64_t)td.y0 * P.img_pitch + td.x0 + xs) * 4;
	else
	{
		src = P.src + (P.src_tiled ? (uint64_t)id.image : inst) * P.src_inst_stride +
		      (uint64_t)p_first * P.src_plane_stride + xs;
		if (P.src_tiled)
			src += (uint64_t)td.y0 * P.src_pitch + td.x0;
	}

	// destinations
	int16_t* tile_stream = P.stream + (uint64_t)id.image * P.stream_stride + td.stream_off;
	const uint64_t nsub = (uint64_t)Tc * Tr;
	const bool store_lane = (lane >= 2) && (lane < 62) && (c0 < Tc);

	VCol st[NPL][4];
#pragma unroll
	for (int p = 0; p < NPL; p++)
#pragma unroll
		for (int k = 0; k < 4; k++)
			st[p][k] = VCol{0, 0, 0, 0, 0, 0, 0, 0};

	if (id.strip == 0 && id.seg == 0 && lane == 0)
#pragma unroll
		for (int p = 0; p < NPL; p++)
			tile_stream[P.grp_off[p_first + p]] = (int16_t)((p_first + p == 0) ? P.q_luma : P.q_chroma);

	for (int v = r_lo - 3; v < r_hi + 3; v++)
	{
		// ---- fetch the row pair of slot v ------------------------------------------------------
		int smp[2][NPL][4];
		const int m = map_index(v, Tr, wrap);
#pragma unroll
		for (int par = 0; par < 2; par++)
		{
			if (m < 0)
			{
#pragma unroll
				for (int p = 0; p < NPL; p++)
#pragma unroll
					for (int k = 0; k < 4; k++)
						smp[par][p][k] = 0;
				continue;
			}
			const int y = min(2 * m + par, chh - 1);  // phantom last row = copy of the last row
			if (U8)
			{
				const uint4 raw = *reinterpret_cast<const uint4*>(img + (uint64_t)y * P.img_pitch * 4);
				const uint32_t px[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
				for (int k = 0; k < 4; k++)
				{
					int r = px[k] & 255, g = (px[k] >> 8) & 255, b = (px[k] >> 16) & 255, a = px[k] >> 24;
					if (P.discard && a == 0)
						r = g = b = 0;
					int c0v = r, c1v = g, c2v = b;
					if (P.color == C_SUBG)
						c0v = g, c1v = r - g, c2v = b - g;
					else if (P.color != C_NONE)
					{
						const int co = r - b;
						const int t = b + tdiv(co, 1);
						const int cg = g - t;
						const int yy = t + tdiv(cg, 1);
						c0v = (P.color == C_YCOCG_Q) ? yy * 2 : yy;
						c1v = co, c2v = cg;
					}
					smp[par][0][k] = c0v;
					if (NPL > 1)
					{
						smp[par][1 % NPL][k] = c1v;
						smp[par][2 % NPL][k] = c2v;
						smp[par][3 % NPL][k] = a;
					}
				}
			}
			else
			{
				const uint2 raw = *reinterpret_cast<const uint2*>(src + (uint64_t)y * P.src_pitch);
				smp[par][0][0] = (int)(int16_t)(raw.x & 0xFFFF);
				smp[par][0][1] = (int)raw.x >> 16;
				smp[par][0][2] = (int)(int16_t)(raw.y & 0xFFFF);
				smp[par][0][3] = (int)raw.y >> 16;
			}
		}

		// ---- rows: horizontal lift of both rows, then columns: one pipeline step ---------------
		const int r = v - 3;
		const bool store_row = (r >= r_lo) && (r < r_hi) && store_lane;
#pragma unroll
		for (int p = 0; p < NPL; p++)
		{
			int e[4], o[4];  // columns: 0,1 = row low-pass of c0, c1; 2,3 = row high-pass of c0, c1
			hlift_forward<KIND, NARROW>(smp[0][p][0], smp[0][p][1], smp[0][p][2], smp[0][p][3], he, e[0], e[1], e[2],
			                            e[3]);
			hlift_forward<KIND, NARROW>(smp[1][p][0], smp[1][p][1], smp[1][p][2], smp[1][p][3], he, o[0], o[1], o[2],
			                            o[3]);
			int lp[4], hp[4];
#pragma unroll
			for (int k = 0; k < 4; k++)
				vstep_forward<KIND, NARROW>(st[p][k], e[k], o[k], v, ve, lp[k], hp[k]);

			if (store_row)
			{
				const int pl = p_first + p;
				const int q = (pl == 0) ? P.q_luma : P.q_chroma;
				const int g = (pl == 0) ? P.g_luma : P.g_chroma;
				const float rq = (pl == 0) ? P.rq_luma : P.rq_chroma;
				int16_t* grp = tile_stream + P.grp_off[pl] + 1 + (uint64_t)r * Tc + c0;
				int16_t* ll;
				if (P.ll_out_stream)
					ll = tile_stream + P.lp_off[pl] + (uint64_t)r * Tc + c0;
				else
					ll = P.dst + inst * P.dst_inst_stride + (uint64_t)pl * P.dst_plane_stride +
					     (uint64_t)r * P.dst_pitch + c0;
				// LL = (LP rows, LP cols), C = (HP rows, LP cols), B = (LP rows, HP cols), D = (HP, HP)
				*reinterpret_cast<uint32_t*>(ll) = pack2(lp[0], lp[1]);
				*reinterpret_cast<uint32_t*>(grp) = pack2(quantize(hp[0], q, g, rq), quantize(hp[1], q, g, rq));
				*reinterpret_cast<uint32_t*>(grp + nsub) =
				    pack2(quantize(lp[2], q, g, rq), quantize(lp[3], q, g, rq));
				*reinterpret_cast<uint32_t*>(grp + 2 * nsub) =
				    pack2(quantize(hp[2], q, g, rq), quantize(hp[3], q, g, rq));
			}
		}
	}
}

}  // namespace ako
