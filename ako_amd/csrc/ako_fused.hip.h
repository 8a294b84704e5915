// ako_fused.hip.h -- levels 0 and 1 of a u8 RGBA plan in one workgroup walk (gfx950).
//
// The level-per-kernel path writes the level-0 low-pass plane (a quarter of the image's samples, 134 MB for an
// 8192 x 8192 RGBA image) to HBM and reads it straight back, in both directions.  Here it never leaves the CU:
//
//   * a workgroup is 12 waves = 6 neighbouring level-0 strips (120 net coefficient columns each, the register-streaming
//     strip walk of ako_stream.hip.h, a PAIR of waves per strip splitting the four planes two and two) walking one
//     row segment together
//   * forward: every row slot a wave finishes leaves its low-pass row (2 planes x 2 columns per lane) in an LDS row
//     buffer as fp32; every second slot the workgroup meets at ONE barrier and each wave then runs one LEVEL-1 task:
//     one plane over one level-1 strip (2 level-1 coefficient columns = 4 low-pass columns per lane, one 16-byte LDS
//     read per row) -- 4 planes x 3 level-1 strips = 12 tasks = 12 waves.  Level 1 runs in the ordinary two-columns-
//     per-lane layout, so neither level pays for the other's halo lanes: a workgroup nets 704 of its 720 level-0
//     columns (the outer 8 low-pass columns per side are level-1 halo; neighbouring workgroups overlap by 16)
//   * inverse: the mirror image.  Each wave first runs its level-1 task one slot ahead and leaves the two low-pass rows
//     it reconstructs in the LDS row buffer; behind the barrier the level-0 waves read their LL pairs from there
//     instead of from HBM.  The same barrier covers the plane swap of the u8 pairs (two slots at a time)
//   * the row buffer is double buffered: one barrier per two row slots (the level-per-kernel inverse has two)
//
// Arithmetic, border rules and stream layout are the functions of ako_stream.hip.h unchanged (exact fp32 pipeline
// forward: |level-1 output| <= 17097; optimistic fp32 + overflow proof inverse).  Eligibility (host side,
// fused2_eligible): 4 channels, YCoCg / YCoCg_Q without the discard rule, CLAMP / MIRROR / ZERO, level-0 extent a
// multiple of 8 x 4 pixels (no phantom column or row at either level), DD13/7 or CDF5/3 at both levels.
//
// Reference: library/lifting.c:171-292 (forward level loop), library/lifting.c:295-... + library/misc.c:229-288
// (inverse level walk), library/format.c:64-134,138-311 (colour / interleave fused at level 0).
#pragma once

#include "ako_fused.h"
#include "ako_stream.hip.h"

namespace ako
{

// Row slots fetched ahead of the one being worked on (2, 3 or 6: the ring index repeats with the 6-slot unrolled loop).
// A workgroup of eight waves runs two waves per SIMD, with 256 registers each: memory latency is hidden by each wave's
// own loads in flight rather than by other waves, so the rings are as deep as the register file allows.
#ifndef AKO_F2_FWD_RING
#define AKO_F2_FWD_RING 6
#endif
#ifndef AKO_F2_INV_RING
#define AKO_F2_INV_RING 6
#endif
constexpr int F2_FWD_RING = AKO_F2_FWD_RING;
constexpr int F2_INV_RING = AKO_F2_INV_RING;
static_assert(6 % F2_FWD_RING == 0 && 6 % F2_INV_RING == 0 && F2_INV_RING >= 3, "ring indices must repeat with the unrolled loop");
constexpr int F2_ROW = 4 * F2_PITCH;   // floats per low-pass row (four planes)
constexpr int F2_HALF = 2 * F2_ROW;    // ... per half of the double buffer (an even and an odd row)
constexpr int F2_LL_FLOATS = 2 * F2_HALF;

struct F2Unit
{
	uint32_t group, seg, tile, image;
};

__device__ __forceinline__ F2Unit f2_unit(const F2Params& P)
{
	// XCD-aware order as in decode_unit(): every XCD gets a contiguous range of logical workgroups, so that the
	// workgroups which share halo columns / rows meet in one L2
	uint32_t blk = blockIdx.x;
	const uint32_t per_xcd = gridDim.x >> 3;
	if (blk < (per_xcd << 3))
		blk = (blk & 7) * per_xcd + (blk >> 3);
	F2Unit u;
	u.group = blk % P.groups;
	blk /= P.groups;
	u.seg = blk % P.segs;
	blk /= P.segs;
	u.tile = blk % P.n_tiles;
	u.image = blk / P.n_tiles;
	return u;
}

// three gated / quantized pairs of one plane's row + the pair of another value stream -> packed words (no low-pass
// word: the fused kernels keep LL in LDS)
__device__ __forceinline__ void f2_pack_cbd(const float lp[4], const float hp[4], float gf, float rq, uint32_t& w_c, uint32_t& w_b)
{
	pack2x2_f(gate_scale_f(hp[0], gf, rq), gate_scale_f(hp[1], gf, rq), gate_scale_f(lp[2], gf, rq), gate_scale_f(lp[3], gf, rq), w_c, w_b);
}

// ---------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------
template <int KIND, bool HEDGE, bool VEDGE>
__device__ __forceinline__ void f2_forward_body(const F2Params& P, const F2Unit& id, const int wave, const int lane, float* llbuf)
{
	const TileDesc td = P.tiles[id.tile];
	const uint64_t inst = (uint64_t)id.image * P.n_tiles + id.tile;
	const int Tc = (int)P.Tc, Tr = (int)P.Tr, T1c = Tc >> 1, T1r = Tr >> 1;
	const int wrap = P.wrap;
	constexpr uint32_t OOB = 0xFFFFFFFFu;
	constexpr int RSRC_FLAGS = 0x00020000;

	// ---- level 0: strip and role of this wave ----
	const int strip = wave >> 1;
	const int role = (wave & 1) ^ ((wave >> 2) & 1);  // planes (role, role + 2); mixed over the SIMDs (wave w runs on SIMD w % 4)
	const int Cg = (int)id.group * F2_GNET - F2_OVERLAP / 2;  // first low-pass column of the workgroup's row buffer
	const int c_base = Cg + strip * SNET - SORG;
	const bool l0_active = c_base + SORG < Tc;  // the strip has net columns inside the level
	const LaneCols lc = lane_columns_at(c_base, (strip == 0) ? 2 + F2_OVERLAP / 4 : 2, (strip == F2_STRIPS - 1) ? 62 - F2_OVERLAP / 4 : 62, lane, Tc, wrap);
	const int c0 = lc.c0;

	int r_lo, r_hi;
	f2_segment_rows(P, id.seg, r_lo, r_hi);
	const int r1_lo = r_lo >> 1, r1_hi = r_hi >> 1;

	const uint8_t* src_base = P.img + (uint64_t)id.image * P.img_stride + ((uint64_t)td.y0 * P.img_pitch + td.x0) * 4;
	const uint32_t row_pitch_b = P.img_pitch * 4u;
	const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src_base), 0, (int)0xFFFFFFFFu, RSRC_FLAGS);
	const uint32_t src_lane_off = (uint32_t)lc.xs * 4u;

	int16_t* tile_stream = P.stream + (uint64_t)id.image * P.stream_stride + td.stream_off;
	const uint64_t stream_left = (P.stream_stride - td.stream_off) * 2;
	const __amdgpu_buffer_rsrc_t rs_stream = __builtin_amdgcn_make_buffer_rsrc(
	    tile_stream, 0, (int)(uint32_t)(stream_left < 0xFFFFFFFFull ? stream_left : 0xFFFFFFFFull), RSRC_FLAGS);
	const bool store_lane = l0_active && lc.net && (c0 >= 0) && (c0 < Tc);
	const uint32_t lane_off = store_lane ? (uint32_t)(c0 * 2) : OOB;
	const uint32_t nsub_b = (uint32_t)((uint64_t)Tc * Tr * 2);
	uint32_t grp_off[2];
	float gf[2], rq[2];
#pragma unroll
	for (int p = 0; p < 2; p++)
	{
		const int pl = role + 2 * p;
		grp_off[p] = (P.lv[0].grp_off[pl] + 1u) * 2u;
		gf[p] = P.lv[0].gate[pl != 0], rq[p] = P.lv[0].rq[pl != 0];
		if (id.group == 0 && id.seg == 0 && strip == 0 && lane == 0)
			tile_stream[P.lv[0].grp_off[pl]] = (int16_t)P.lv[0].q[pl != 0];
	}
	// this wave's low-pass pairs in a row of the LDS buffer: planes (role, role + 2), columns 120 * strip + 2 * (lane - 2), +1
	const bool ll_lane = l0_active && (lane >= 2) && (lane < 62);
	const int ll_wr = role * F2_PITCH + strip * SNET + 2 * lane - 4;

	// ---- level 1: the task of this wave (plane p1 over level-1 strip j1) ----
	const int p1 = wave & 3, j1 = wave >> 2;
	const int c_base1 = (Cg >> 1) + SNET * j1;  // level-1 coefficient column of lane 0 (Cg is even; -8 >> 1 = -4)
	const bool l1_active = c_base1 + SORG < T1c;
	const LaneCols lc1 = lane_columns_at(c_base1, 2, (j1 == F2_L1STRIPS - 1) ? 62 - F2_OVERLAP / 4 : 62, lane, T1c, wrap);
	const int ll_rd = p1 * F2_PITCH + 2 * SNET * j1 + 4 * lane;
	const bool store_lane1 = l1_active && lc1.net && (lc1.c0 >= 0) && (lc1.c0 < T1c);
	const uint32_t lane1_off = store_lane1 ? (uint32_t)(lc1.c0 * 2) : OOB;
	const uint32_t nsub1_b = (uint32_t)((uint64_t)T1c * T1r * 2);
	const uint32_t grp1_off = (P.lv[1].grp_off[p1] + 1u) * 2u;
	const float gf1 = P.lv[1].gate[p1 != 0], rq1 = P.lv[1].rq[p1 != 0];
	if (id.group == 0 && id.seg == 0 && j1 == 0 && lane == 0)
		tile_stream[P.lv[1].grp_off[p1]] = (int16_t)P.lv[1].q[p1 != 0];
	int16_t* ll1_root = P.ll1_in_stream ? tile_stream : (P.ll1 + inst * P.ll1_inst_stride);
	const uint64_t ll1_left = P.ll1_in_stream ? stream_left : (uint64_t)4 * P.ll1_plane_stride * 2;
	const __amdgpu_buffer_rsrc_t rs_ll1 = __builtin_amdgcn_make_buffer_rsrc(
	    ll1_root, 0, (int)(uint32_t)(ll1_left < 0xFFFFFFFFull ? ll1_left : 0xFFFFFFFFull), RSRC_FLAGS);
	const uint32_t ll1_off = (P.ll1_in_stream ? P.lp_off[p1] : (uint32_t)p1 * P.ll1_plane_stride) * 2u;
	const uint32_t ll1_pitch_b = (P.ll1_in_stream ? (uint32_t)T1c : P.ll1_pitch) * 2u;

	// ---- pipeline state ----
	VFwd<float> st[2][4];
#pragma unroll
	for (int p = 0; p < 2; p++)
#pragma unroll
		for (int k = 0; k < 4; k++)
			st[p][k] = VFwd<float>{{0, 0, 0}, {0, 0}, {0, 0, 0}};
	VFwd3<float> st1[4];
	float holdE[4] = {0, 0, 0, 0}, holdO[4] = {0, 0, 0, 0};
#pragma unroll
	for (int k = 0; k < 4; k++)
		st1[k] = VFwd3<float>{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
	(void)holdE, (void)holdO;

	using Raw = FwdRaw<true>;
	auto fetch = [&](int v, Raw& raw) {
		const int m = VEDGE ? map_index(v, Tr, wrap) : v;
#pragma unroll
		for (int par = 0; par < 2; par++)
		{
			const uint32_t row_off = (uint32_t)(2 * max(m, 0) + par) * row_pitch_b;
			raw.a[par] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, src_lane_off, row_off, 0));
		}
	};

	// one level-0 row slot of this wave's two planes: C / B / D to the stream, the low-pass pairs to the LDS row 'll_row'
	auto l0_slot = [&](auto kc, const int v, Raw& raw, float* ll_row) {
		constexpr int K = decltype(kc)::value;
		const bool zero_row = VEDGE && (wrap == W_ZERO) && ((unsigned)v >= (unsigned)Tr);
		float smp[2][2][4];
#pragma unroll
		for (int par = 0; par < 2; par++)
		{
			const uint32_t px[4] = {raw.a[par].x, raw.a[par].y, raw.a[par].z, raw.a[par].w};
			float v0[4], v1[4];
			decode_pixels_ycocg<float>(px, (P.color == C_YCOCG_Q) ? 2.0f : 1.0f, role, v0, v1);
#pragma unroll
			for (int k = 0; k < 4; k++)
			{
				smp[par][0][k] = zero_row ? 0.0f : v0[k];
				smp[par][1][k] = zero_row ? 0.0f : v1[k];
			}
		}
		fetch(v + F2_FWD_RING, raw);  // the registers the slot's pixels just left take the rows of F2_FWD_RING slots ahead

		const int r = v - 3;
#ifdef AKO_MEASURE
		const bool row_ok = (r >= r_lo) && (r < r_hi) && !(P.dbg & 4);  // bit 2: no level-0 stores
#else
		const bool row_ok = (r >= r_lo) && (r < r_hi);  // wave-uniform
#endif
		const uint32_t row_grp = (uint32_t)r * (uint32_t)Tc * 2u;
		float dsub[2][2];
#pragma unroll
		for (int p = 0; p < 2; p++)
		{
			float e[4], o[4];
			hlift_forward<KIND, false, HEDGE, float>(smp[0][p][0], smp[0][p][1], smp[0][p][2], smp[0][p][3], lc.he, e[0], e[1], e[2], e[3]);
			hlift_forward<KIND, false, HEDGE, float>(smp[1][p][0], smp[1][p][1], smp[1][p][2], smp[1][p][3], lc.he, o[0], o[1], o[2], o[3]);
			float lp[4], hp[4];
#pragma unroll
			for (int k = 0; k < 4; k++)
				vstep_forward<KIND, false, VEDGE, K, float>(st[p][k], e[k], o[k], v, wrap, Tr, lp[k], hp[k]);
			if (ll_lane)
				*reinterpret_cast<float2*>(ll_row + ll_wr + 2 * p * F2_PITCH) = make_float2(lp[0], lp[1]);
			uint32_t w_c, w_b;
			f2_pack_cbd(lp, hp, gf[p], rq[p], w_c, w_b);
			dsub[p][0] = gate_scale_f(hp[2], gf[p], rq[p]), dsub[p][1] = gate_scale_f(hp[3], gf[p], rq[p]);
			const uint32_t s_c = row_ok ? grp_off[p] + row_grp : OOB;
			const uint32_t s_b = row_ok ? grp_off[p] + row_grp + nsub_b : OOB;
			__builtin_amdgcn_raw_buffer_store_b32(w_c, rs_stream, lane_off, s_c, 0);
			__builtin_amdgcn_raw_buffer_store_b32(w_b, rs_stream, lane_off, s_b, 0);
		}
		uint32_t w_d0, w_d1;
		pack2x2_f(dsub[0][0], dsub[0][1], dsub[1][0], dsub[1][1], w_d0, w_d1);
		__builtin_amdgcn_raw_buffer_store_b32(w_d0, rs_stream, lane_off, row_ok ? grp_off[0] + row_grp + 2u * nsub_b : OOB, 0);
		__builtin_amdgcn_raw_buffer_store_b32(w_d1, rs_stream, lane_off, row_ok ? grp_off[1] + row_grp + 2u * nsub_b : OOB, 0);
	};

	// one level-1 row slot of this wave's task: the two low-pass rows of 'll_half' are its even and its odd input row
	auto l1_slot = [&](auto jc, const int v1, const float* ll_half) {
		constexpr int K1 = decltype(jc)::value;
		const float4 ev = *reinterpret_cast<const float4*>(ll_half + ll_rd);
		const float4 od = *reinterpret_cast<const float4*>(ll_half + F2_ROW + ll_rd);
		float e[4], o[4];
		hlift_forward<KIND, false, HEDGE, float>(ev.x, ev.y, ev.z, ev.w, lc1.he, e[0], e[1], e[2], e[3]);
		hlift_forward<KIND, false, HEDGE, float>(od.x, od.y, od.z, od.w, lc1.he, o[0], o[1], o[2], o[3]);
		if constexpr (VEDGE)
		{
			// slots beyond the top / bottom border are fed as the level-per-kernel fetch maps them (map_index): the
			// nearest slot (CLAMP, MIRROR) or zeros (ZERO); the three slots above the top border in one burst as soon as
			// slot 0 exists, the ones below the bottom border from the held last slot
			const bool zero = (wrap == W_ZERO);
			if (v1 < 0)
				return;
			if (v1 == 0)
			{
				float x0, x1;
				static_for<3>([&](auto bc) {
					constexpr int B = decltype(bc)::value;
#pragma unroll
					for (int k = 0; k < 4; k++)
						vstep_forward3<KIND, true, B, float>(st1[k], zero ? 0.0f : e[k], zero ? 0.0f : o[k], B - 3, wrap, T1r, x0, x1);
				});
			}
			if (v1 == T1r - 1)
#pragma unroll
				for (int k = 0; k < 4; k++)
					holdE[k] = e[k], holdO[k] = o[k];
			if (v1 >= T1r)
#pragma unroll
				for (int k = 0; k < 4; k++)
					e[k] = zero ? 0.0f : holdE[k], o[k] = zero ? 0.0f : holdO[k];
		}
		float lp[4], hp[4];
#pragma unroll
		for (int k = 0; k < 4; k++)
			vstep_forward3<KIND, VEDGE, K1, float>(st1[k], e[k], o[k], v1, wrap, T1r, lp[k], hp[k]);
		uint32_t w_ll, w_c, w_b, w_d;
		pack_row_f(lp, hp, gf1, rq1, w_ll, w_c, w_b, w_d);
		const int r1 = v1 - 3;
		const bool ok1 = (r1 >= r1_lo) && (r1 < r1_hi);  // wave-uniform
		const uint32_t row1 = (uint32_t)r1 * (uint32_t)T1c * 2u;
		__builtin_amdgcn_raw_buffer_store_b32(w_ll, rs_ll1, lane1_off, ok1 ? ll1_off + (uint32_t)r1 * ll1_pitch_b : OOB, 0);
		__builtin_amdgcn_raw_buffer_store_b32(w_c, rs_stream, lane1_off, ok1 ? grp1_off + row1 : OOB, 0);
		__builtin_amdgcn_raw_buffer_store_b32(w_b, rs_stream, lane1_off, ok1 ? grp1_off + row1 + nsub1_b : OOB, 0);
		__builtin_amdgcn_raw_buffer_store_b32(w_d, rs_stream, lane1_off, ok1 ? grp1_off + row1 + 2u * nsub1_b : OOB, 0);
	};

	// Row slots walked: level 1 stores rows r1_lo .. r1_hi - 1, i.e. needs the low-pass rows r_lo - 6 .. r_hi + 5, i.e.
	// level-0 slots r_lo - 9 .. r_hi + 8.  r_lo is a multiple of 6, so the walk starts on a slot = 3 (mod 6): row r = v - 3
	// has the parity of the unroll position K, and level-1 slot (r - 1) / 2 the ring position (K - 1) / 2.
	const int v_begin = r_lo - 9;
	const int n_slots = (r_hi - r_lo) + 18;
	Raw ring[F2_FWD_RING];
	if (l0_active)
		static_for<F2_FWD_RING>([&](auto kc) { fetch(v_begin + decltype(kc)::value, ring[decltype(kc)::value]); });
	int half = 0;
	for (int base = 0; base < n_slots; base += 6)
	{
		static_for<3>([&](auto jc) {
			constexpr int J = decltype(jc)::value;
			const int v = v_begin + base + 2 * J;
			float* ll_half = llbuf + half * F2_HALF;
			if (l0_active)
			{
				l0_slot(std::integral_constant<int, 2 * J>{}, v, ring[(2 * J) % F2_FWD_RING], ll_half);
				l0_slot(std::integral_constant<int, 2 * J + 1>{}, v + 1, ring[(2 * J + 1) % F2_FWD_RING], ll_half + F2_ROW);
			}
#ifdef AKO_MEASURE
			if (!(P.dbg & 1))  // bit 0: no barrier (races: wrong output, timing only)
#endif
			__syncthreads();
			// (the other half is rewritten only behind the NEXT barrier, which no wave passes before every wave has
			// finished reading this one)
			const int v1 = (v - 3) >> 1;  // = (r_odd - 1) / 2 with r_odd = v + 1 - 3 (v is odd)
#ifdef AKO_MEASURE
			if (!(P.dbg & 2))  // bit 1: no level-1 work
#endif
			if (l1_active && (VEDGE || v1 >= r1_lo - 3))
				l1_slot(jc, v1, ll_half);
			half ^= 1;
		});
	}
}

template <int KIND>
__global__ __launch_bounds__(F2_THREADS) void k_fused2_forward(const F2Params P)
{
	__shared__ __attribute__((aligned(16))) float llbuf[F2_LL_FLOATS];
	const F2Unit id = f2_unit(P);
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int lane = threadIdx.x & 63;
	const int Tc = (int)P.Tc, Tr = (int)P.Tr;
	const int Cg = (int)id.group * F2_GNET - F2_OVERLAP / 2;
	const int c_base = Cg + (wave >> 1) * SNET - SORG;
	const int c_base1 = (Cg >> 1) + SNET * (wave >> 2);
	// border code for this wave: its level-0 strip or its level-1 strip holds out-of-range lanes
	const bool hedge = (c_base < 0) || (c_base + 128 > Tc) || (c_base1 < 0) || (c_base1 + 128 > (Tc >> 1));
	int r_lo, r_hi;
	f2_segment_rows(P, id.seg, r_lo, r_hi);
	// any slot of the walk (its lead-in, the slots the unrolled loop runs past the segment, the prefetch) outside the level?
	const int trips = (r_hi - r_lo + 18 + 5) / 6;
	const bool vedge = (r_lo < 18) || (r_lo - 10 + 6 * trips + F2_FWD_RING > Tr - 1);  // workgroup-uniform
	if (__builtin_expect(vedge, 0))
	{
		if (hedge)
			f2_forward_body<KIND, true, true>(P, id, wave, lane, llbuf);
		else
			f2_forward_body<KIND, false, true>(P, id, wave, lane, llbuf);
	}
	else
	{
		if (hedge)
			f2_forward_body<KIND, true, false>(P, id, wave, lane, llbuf);
		else
			f2_forward_body<KIND, false, false>(P, id, wave, lane, llbuf);
	}
}

// ---------------------------------------------------------------------------------------------
// Inverse
// ---------------------------------------------------------------------------------------------
// Geometry: the level-1 tasks of a workgroup reconstruct the low-pass columns [A, A + 720) (3 strips x 120 net level-1
// columns); the six level-0 strips need their own halo lanes filled from those, so the first one nets from lane 4 and the
// last one up to lane 59: 712 net level-0 columns per workgroup, A = 712 * group - 4.  The row buffer holds column
// A - 4 + i at index i.
// plane swap of the u8 pairs, two row slots at a time: [buffer][strip][destination wave of the pair][slot][plane][lane]
constexpr int F2I_XBUF_VEC = 2 * F2_STRIPS * 2 * 2 * 2 * 64;  // uint4 entries (96 KiB)
constexpr uint32_t F2I_LDS_BYTES = F2_LL_FLOATS * sizeof(float) + F2I_XBUF_VEC * sizeof(uint4);

struct F2InvRaw0  // level 0: C, B, D of this wave's two planes (two coefficients each); LL comes through LDS
{
	uint32_t c[2], b[2], d[2];
};

template <int KIND, bool HEDGE, bool VEDGE>
__device__ __forceinline__ void f2_inverse_body(const F2Params& P, const F2Unit& id, const int wave, const int lane, float* llbuf, uint4* xbuf)
{
	const TileDesc td = P.tiles[id.tile];
	const uint64_t inst = (uint64_t)id.image * P.n_tiles + id.tile;
	const int Tc = (int)P.Tc, Tr = (int)P.Tr, T1c = Tc >> 1, T1r = Tr >> 1;
	const int wrap = P.wrap;
	constexpr uint32_t OOB = 0xFFFFFFFFu;
	constexpr int RSRC_FLAGS = 0x00020000;
	float peak_in = 0.0f, peak_out = 0.0f;

	// ---- level 0: strip and role of this wave ----
	const int strip = wave >> 1;
	const int role = (wave & 1) ^ ((wave >> 2) & 1);  // planes (2 role, 2 role + 1); this wave finishes pixel row 'role' of a slot
	const int A = (int)id.group * F2I_GNET - F2I_OVERLAP / 2;
	const int c_base = A - SORG + strip * SNET;
	const bool l0_active = c_base + SORG < Tc;
	const LaneCols lc = lane_columns_at(c_base, (strip == 0) ? 4 : 2, (strip == F2_STRIPS - 1) ? 60 : 62, lane, Tc, wrap);
	const int c0 = lc.c0;

	int r_lo, r_hi;
	f2_segment_rows(P, id.seg, r_lo, r_hi);
	const int r1_lo = r_lo >> 1;
	const int half_rows = (r_hi - r_lo) >> 1;

	const int16_t* tile_stream = P.stream + (uint64_t)id.image * P.stream_stride + td.stream_off;
	const __amdgpu_buffer_rsrc_t rs_stream = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(tile_stream), 0, (int)0xFFFFFFFFu, RSRC_FLAGS);
	const uint32_t lane_in_off = (uint32_t)lc.cs * 2u;
	const uint32_t nsub_b = (uint32_t)((uint64_t)Tc * Tr * 2);
	uint32_t grp_off[2];
	float qf[2];
#pragma unroll
	for (int p = 0; p < 2; p++)
	{
		const int pl = 2 * role + p;
		grp_off[p] = (P.lv[0].grp_off[pl] + 1u) * 2u;
		// the decoder trusts the lift head (misc.c:266-272, lifting.c:114-116); wave-uniform, so kept in a scalar register
		qf[p] = (float)__builtin_amdgcn_readfirstlane((int)tile_stream[P.lv[0].grp_off[pl]]);
	}
	uint8_t* img = P.img + (uint64_t)id.image * P.img_stride + ((uint64_t)td.y0 * P.img_pitch + td.x0) * 4;
	const uint32_t out_pitch = P.img_pitch * 4u;
	const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc(img, 0, (int)0xFFFFFFFFu, RSRC_FLAGS);
	const bool store_lane = l0_active && lc.net && (c0 >= 0) && (c0 < Tc);
	const uint32_t px_lane_off = store_lane ? (uint32_t)(2 * c0) * 4u : OOB;
	const int ll_rd = 2 * role * F2_PITCH + strip * SNET + 2 * lane;  // + p * F2_PITCH

	// ---- level 1: the task of this wave (plane p1 over level-1 strip j1) ----
	const int p1 = wave & 3, j1 = wave >> 2;
	const int c_base1 = (A >> 1) + SNET * j1 - SORG;  // (A is even)
	const bool l1_active = c_base1 + SORG < T1c;
	const LaneCols lc1 = lane_columns_at(c_base1, 2, 62, lane, T1c, wrap);
	const bool ll_lane = (lane >= 2) && (lane < 62);
	const int ll_wr = p1 * F2_PITCH + 2 * SNET * j1 - 4 + 4 * lane;
	const uint32_t lane1_in_off = (uint32_t)lc1.cs * 2u;
	const uint32_t nsub1_b = (uint32_t)((uint64_t)T1c * T1r * 2);
	const uint32_t grp1_off = (P.lv[1].grp_off[p1] + 1u) * 2u;
	const float qf1 = (float)__builtin_amdgcn_readfirstlane((int)tile_stream[P.lv[1].grp_off[p1]]);
	const int16_t* ll1_root = P.ll1_in_stream ? tile_stream : (P.ll1 + inst * P.ll1_inst_stride);
	const __amdgpu_buffer_rsrc_t rs_ll1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(ll1_root), 0, (int)0xFFFFFFFFu, RSRC_FLAGS);
	const uint32_t ll1_off = (P.ll1_in_stream ? P.lp_off[p1] : (uint32_t)p1 * P.ll1_plane_stride) * 2u;
	const uint32_t ll1_pitch_b = (P.ll1_in_stream ? (uint32_t)T1c : P.ll1_pitch) * 2u;

	// ---- pipeline state ----
	VInv<float> st[2][4];
#pragma unroll
	for (int p = 0; p < 2; p++)
#pragma unroll
		for (int k = 0; k < 4; k++)
			st[p][k] = VInv<float>{{0, 0, 0}, {0, 0, 0}, 0};
	VInv<float> st1[4];
#pragma unroll
	for (int k = 0; k < 4; k++)
		st1[k] = VInv<float>{{0, 0, 0}, {0, 0, 0}, 0};
	float holdLL[2][2] = {{0, 0}, {0, 0}};
	(void)holdLL;

	using Raw1 = InvRaw<1>;
	auto fetch1 = [&](int v1, Raw1& raw) {
		const int m = max(VEDGE ? map_index(v1, T1r, wrap) : v1, 0);
		const uint32_t g = grp1_off + (uint32_t)m * (uint32_t)T1c * 2u;
		raw.ll[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_ll1, lane1_in_off, ll1_off + (uint32_t)m * ll1_pitch_b, 0);
		raw.c[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_stream, lane1_in_off, g, 0);
		raw.b[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_stream, lane1_in_off, g + nsub1_b, 0);
		raw.d[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_stream, lane1_in_off, g + 2u * nsub1_b, 0);
	};
	auto fetch0 = [&](int v, F2InvRaw0& raw) {
		const int m = max(VEDGE ? map_index(v, Tr, wrap) : v, 0);
		const uint32_t row_g = (uint32_t)m * (uint32_t)Tc * 2u;
#pragma unroll
		for (int p = 0; p < 2; p++)
		{
			const uint32_t g = grp_off[p] + row_g;
			raw.c[p] = __builtin_amdgcn_raw_buffer_load_b32(rs_stream, lane_in_off, g, 0);
			raw.b[p] = __builtin_amdgcn_raw_buffer_load_b32(rs_stream, lane_in_off, g + nsub_b, 0);
			raw.d[p] = __builtin_amdgcn_raw_buffer_load_b32(rs_stream, lane_in_off, g + 2u * nsub_b, 0);
		}
	};

	// one level-1 row slot: reconstructs the low-pass rows 2 (v1 - 3), + 1 of this wave's plane and strip into 'll_half'
	auto l1_slot = [&](auto jc, const int v1, const Raw1& raw, float* ll_half) {
		constexpr int K1 = decltype(jc)::value;
		const bool zero_row = VEDGE && (wrap == W_ZERO) && ((unsigned)v1 >= (unsigned)T1r);
		float lpv[4], hpv[4];
		unpack2_f(raw.ll[0], lpv[0], lpv[1]);
		unpack2_f(raw.b[0], lpv[2], lpv[3]);
		unpack2_f(raw.c[0], hpv[0], hpv[1]);
		unpack2_f(raw.d[0], hpv[2], hpv[3]);
		if (qf1 > 1.0f)  // lifting.c:30-40
		{
			lpv[2] *= qf1, lpv[3] *= qf1;
#pragma unroll
			for (int k = 0; k < 4; k++)
				hpv[k] *= qf1;
		}
#pragma unroll
		for (int k = 0; k < 4; k += 2)
		{
			absmax3(peak_in, lpv[k], lpv[k + 1]);
			absmax3(peak_in, hpv[k], hpv[k + 1]);
		}
		if (zero_row)
#pragma unroll
			for (int k = 0; k < 4; k++)
				lpv[k] = 0, hpv[k] = 0;
		float ev[4], od[4];
#pragma unroll
		for (int k = 0; k < 4; k++)
			vstep_inverse<KIND, VEDGE, K1, float>(st1[k], lpv[k], hpv[k], v1, wrap, T1r, ev[k], od[k]);
		float o0[4], o1[4];
		hlift_inverse<KIND, HEDGE, float>(ev[0], ev[1], ev[2], ev[3], lc1.he, o0[0], o0[1], o0[2], o0[3]);
		hlift_inverse<KIND, HEDGE, float>(od[0], od[1], od[2], od[3], lc1.he, o1[0], o1[1], o1[2], o1[3]);
		if (ll_lane)
		{
			*reinterpret_cast<float4*>(ll_half + ll_wr) = make_float4(o0[0], o0[1], o0[2], o0[3]);
			*reinterpret_cast<float4*>(ll_half + F2_ROW + ll_wr) = make_float4(o1[0], o1[1], o1[2], o1[3]);
		}
	};

	// one level-0 row slot of this wave's two planes: the lifted rows of slot v - 3; this wave's own pixel row stays in
	// 'mine', the other one goes to the partner through 'xw'
	auto l0_feed = [&](auto kc, const int v, const float lpv[2][4], const float hpv[2][4], float out[2][2][4]) {
		constexpr int K = decltype(kc)::value;
#pragma unroll
		for (int p = 0; p < 2; p++)
		{
			float ev[4], od[4];
#pragma unroll
			for (int k = 0; k < 4; k++)
				vstep_inverse<KIND, VEDGE, K, float>(st[p][k], lpv[p][k], hpv[p][k], v, wrap, Tr, ev[k], od[k]);
			hlift_inverse<KIND, HEDGE, float>(ev[0], ev[1], ev[2], ev[3], lc.he, out[0][p][0], out[0][p][1], out[0][p][2], out[0][p][3]);
			hlift_inverse<KIND, HEDGE, float>(od[0], od[1], od[2], od[3], lc.he, out[1][p][0], out[1][p][1], out[1][p][2], out[1][p][3]);
		}
	};
	auto l0_slot = [&](auto kc, const int v, const F2InvRaw0& raw, const float* ll_row, float mine[2][4], uint4* xw) {
		constexpr int K = decltype(kc)::value;
		if (VEDGE && v < 0)
			return;  // fed in a burst when slot 0 arrives (below)
		const bool zero_row = VEDGE && (wrap == W_ZERO) && (v >= Tr);
		float lpv[2][4], hpv[2][4];
#pragma unroll
		for (int p = 0; p < 2; p++)
		{
			float2 llp = *reinterpret_cast<const float2*>(ll_row + ll_rd + p * F2_PITCH);
			if (VEDGE)
			{
				// rows below the bottom border: the level-per-kernel load maps them to the last row (CLAMP, MIRROR)
				if (v == Tr - 1)
					holdLL[p][0] = llp.x, holdLL[p][1] = llp.y;
				if (v >= Tr)
					llp.x = holdLL[p][0], llp.y = holdLL[p][1];
			}
			lpv[p][0] = llp.x, lpv[p][1] = llp.y;
			unpack2_f(raw.b[p], lpv[p][2], lpv[p][3]);
			unpack2_f(raw.c[p], hpv[p][0], hpv[p][1]);
			unpack2_f(raw.d[p], hpv[p][2], hpv[p][3]);
			if (qf[p] > 1.0f)
			{
				lpv[p][2] *= qf[p], lpv[p][3] *= qf[p];
#pragma unroll
				for (int k = 0; k < 4; k++)
					hpv[p][k] *= qf[p];
			}
#pragma unroll
			for (int k = 0; k < 4; k += 2)
			{
				absmax3(peak_in, lpv[p][k], lpv[p][k + 1]);
				absmax3(peak_in, hpv[p][k], hpv[p][k + 1]);
			}
			if (zero_row)
#pragma unroll
				for (int k = 0; k < 4; k++)
					lpv[p][k] = 0, hpv[p][k] = 0;
		}
		float out[2][2][4];
		if constexpr (VEDGE)
		{
			if (v == 0)
			{
				// the three slots above the top border: slot 0's rows again (CLAMP, MIRROR: map_index) or zeros (ZERO)
				float zl[2][4], zh[2][4];
#pragma unroll
				for (int p = 0; p < 2; p++)
#pragma unroll
					for (int k = 0; k < 4; k++)
						zl[p][k] = (wrap == W_ZERO) ? 0.0f : lpv[p][k], zh[p][k] = (wrap == W_ZERO) ? 0.0f : hpv[p][k];
				static_for<3>([&](auto bc) {
					constexpr int B = decltype(bc)::value;
					l0_feed(std::integral_constant<int, (K + 3 + B) % 6>{}, B - 3, zl, zh, out);
				});
			}
		}
		l0_feed(kc, v, lpv, hpv, out);
#pragma unroll
		for (int par = 0; par < 2; par++)
#pragma unroll
			for (int p = 0; p < 2; p++)
			{
				absmax3(peak_out, out[par][p][0], out[par][p][1]);
				absmax3(peak_out, out[par][p][2], out[par][p][3]);
			}
		auto as_u4 = [](const float* f) { return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])); };
		// role 1 keeps pixel row 1 and hands over row 0; role 0 the other way round (both spelled out: no per-value selects)
		if (role)
		{
			xw[0 * 64 + lane] = as_u4(out[0][0]), xw[1 * 64 + lane] = as_u4(out[0][1]);
#pragma unroll
			for (int k = 0; k < 4; k++)
				mine[0][k] = out[1][0][k], mine[1][k] = out[1][1][k];
		}
		else
		{
			xw[0 * 64 + lane] = as_u4(out[1][0]), xw[1 * 64 + lane] = as_u4(out[1][1]);
#pragma unroll
			for (int k = 0; k < 4; k++)
				mine[0][k] = out[0][0][k], mine[1][k] = out[0][1][k];
		}
	};
	// behind the barrier: the partner's two planes of this wave's pixel row -> colour inverse, saturate, 16-byte store
	auto l0_finish = [&](const int v, const float mine[2][4], const uint4* xr) {
		const int r = v - 3;
		if (!((r >= r_lo) && (r < r_hi)))  // wave-uniform
			return;
		const uint4 g0 = xr[0 * 64 + lane], g1 = xr[1 * 64 + lane];
		const float his0[4] = {__uint_as_float(g0.x), __uint_as_float(g0.y), __uint_as_float(g0.z), __uint_as_float(g0.w)};
		const float his1[4] = {__uint_as_float(g1.x), __uint_as_float(g1.y), __uint_as_float(g1.z), __uint_as_float(g1.w)};
		uint32_t px[4];
		if (role)
		{
#pragma unroll
			for (int k = 0; k < 4; k++)
			{
				float rr, gg, bb;
				color_inverse_fast(P.color, his0[k], his1[k], mine[0][k], rr, gg, bb);
				px[k] = pixel_u8x4(rr, gg, bb, mine[1][k]);
			}
		}
		else
		{
#pragma unroll
			for (int k = 0; k < 4; k++)
			{
				float rr, gg, bb;
				color_inverse_fast(P.color, mine[0][k], mine[1][k], his0[k], rr, gg, bb);
				px[k] = pixel_u8x4(rr, gg, bb, his1[k]);
			}
		}
		typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
		__builtin_amdgcn_raw_buffer_store_b128(u32x4{px[0], px[1], px[2], px[3]}, rs_img, px_lane_off, (uint32_t)(2 * r + role) * out_pitch, 0);
		AKO_STORE_GUARD();  // (see store_b128_guarded in ako_stream.hip.h)
	};

	// Iteration i: [A] level-1 slot r1_lo - 7 + i leaves the low-pass rows of level-1 output slot m = r1_lo - 10 + i in half
	// i & 1 of the row buffer; [B] the level-0 slots 2 m', 2 m' + 1 of m' = m - 1 read theirs from the other half (written
	// one iteration earlier) and stage the plane swap; barrier; [C] the pixel rows of those two slots are finished.
	// Level 0 needs slots r_lo - 4 .. r_hi + 3 (an even start, so that slots pair up like level-1 output rows): m' from
	// r1_lo - 2, i.e. iterations 9 .. half_rows + 12; level 1 runs in iterations 0 .. half_rows + 11 (its first eight
	// slots are lead-in: two more than the filter needs, so that level 0 starts on unroll position 0).
	Raw1 ring1[3];
	F2InvRaw0 ring0[F2_INV_RING];
	if (l1_active)
	{
		fetch1(r1_lo - 7, ring1[0]);
		fetch1(r1_lo - 6, ring1[1]);
	}
	if (l0_active)
		static_for<F2_INV_RING - 1>([&](auto kc) { fetch0(r_lo - 4 + decltype(kc)::value, ring0[decltype(kc)::value]); });
	uint4* const xmine = xbuf + ((strip * 2 + (1 - role)) * 2) * 2 * 64;  // what this wave writes: destination = the partner
	const uint4* const xhis = xbuf + ((strip * 2 + role) * 2) * 2 * 64;    // what it reads
	constexpr int XBUF_HALF = F2_STRIPS * 2 * 2 * 2 * 64;
	auto iteration = [&](auto jc, const int i, auto with_l1, auto with_l0) {
		constexpr int J = decltype(jc)::value;
		const int buf = i & 1;
		if constexpr (decltype(with_l1)::value)
		{
			if (l1_active)
			{
				const int v1 = r1_lo - 7 + i;
				fetch1(v1 + 2, ring1[(J + 2) % 3]);
				l1_slot(jc, v1, ring1[J], llbuf + buf * F2_HALF);
			}
		}
		const int v = r_lo - 22 + 2 * i;  // = 2 (r1_lo - 11 + i)
		float mine[2][2][4];
		if constexpr (decltype(with_l0)::value)
		{
			if (l0_active)
			{
				const float* ll_prev = llbuf + (buf ^ 1) * F2_HALF;
				uint4* xw = xmine + buf * XBUF_HALF;
				constexpr int R = F2_INV_RING;
				fetch0(v + R - 1, ring0[(2 * J + R - 1) % R]);
				l0_slot(std::integral_constant<int, 2 * J>{}, v, ring0[(2 * J) % R], ll_prev, mine[0], xw);
				fetch0(v + R, ring0[(2 * J + R) % R]);
				l0_slot(std::integral_constant<int, 2 * J + 1>{}, v + 1, ring0[(2 * J + 1) % R], ll_prev + F2_ROW, mine[1], xw + 2 * 64);
			}
		}
		__syncthreads();
		if constexpr (decltype(with_l0)::value)
		{
			if (l0_active)
			{
				const uint4* xr = xhis + buf * XBUF_HALF;
				l0_finish(v, mine[0], xr);
				l0_finish(v + 1, mine[1], xr + 2 * 64);
			}
		}
	};
	for (int base = 0; base < 9; base += 3)  // level-1 lead-in
		static_for<3>([&](auto jc) { iteration(jc, base + decltype(jc)::value, std::true_type{}, std::false_type{}); });
	const int steady_end = 9 + ((half_rows + 3 + 2) / 3) * 3;
	for (int base = 9; base < steady_end; base += 3)
		static_for<3>([&](auto jc) { iteration(jc, base + decltype(jc)::value, std::true_type{}, std::true_type{}); });
	iteration(std::integral_constant<int, 0>{}, steady_end, std::false_type{}, std::true_type{});  // the last pair of level-0 slots
	const bool bad = !(peak_in <= OPT_INPUT_BOUND) || !(peak_out <= OPT_OUTPUT_BOUND);  // negated: NaN counts as bad
	if (__any(bad) && lane == 0)
		atomicMax(P.ovf_flag, P.ovf_gen);
}

template <int KIND>
__global__ __launch_bounds__(F2_THREADS) void k_fused2_inverse(const F2Params P)
{
	extern __shared__ __attribute__((aligned(16))) float f2_lds[];
	float* llbuf = f2_lds;
	uint4* xbuf = reinterpret_cast<uint4*>(f2_lds + F2_LL_FLOATS);
	// The row buffer starts as zeros: the columns no level-1 task of this workgroup writes (the four beyond either end, the
	// strips beyond the right border) are read by level-0 halo / out-of-range lanes, whose values never reach a stored
	// pixel but do enter the overflow proof -- they must not look like an overflow.
	for (int t = threadIdx.x; t < F2_LL_FLOATS / 4; t += F2_THREADS)
		reinterpret_cast<float4*>(llbuf)[t] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	__syncthreads();
	const F2Unit id = f2_unit(P);
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int lane = threadIdx.x & 63;
	const int Tc = (int)P.Tc, Tr = (int)P.Tr;
	const int A = (int)id.group * F2I_GNET - F2I_OVERLAP / 2;
	const int c_base = A - SORG + (wave >> 1) * SNET;
	const int c_base1 = (A >> 1) + SNET * (wave >> 2) - SORG;
	const bool hedge = (c_base < 0) || (c_base + 128 > Tc) || (c_base1 < 0) || (c_base1 + 128 > (Tc >> 1));
	int r_lo, r_hi;
	f2_segment_rows(P, id.seg, r_lo, r_hi);
	// any slot of either level (lead-in, prefetch) outside the level?
	const bool vedge = (r_lo < 18) || (r_hi + 14 + F2_INV_RING > Tr);  // workgroup-uniform
	if (__builtin_expect(vedge, 0))
	{
		if (hedge)
			f2_inverse_body<KIND, true, true>(P, id, wave, lane, llbuf, xbuf);
		else
			f2_inverse_body<KIND, false, true>(P, id, wave, lane, llbuf, xbuf);
	}
	else
	{
		if (hedge)
			f2_inverse_body<KIND, true, false>(P, id, wave, lane, llbuf, xbuf);
		else
			f2_inverse_body<KIND, false, false>(P, id, wave, lane, llbuf, xbuf);
	}
}

}  // namespace ako
