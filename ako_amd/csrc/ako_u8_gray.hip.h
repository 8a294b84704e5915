// ako_u8_gray.hip.h -- native u8 level-0 kernels for images of ONE or TWO channels (gray, gray + alpha), round 4.
//
// The reference takes 1..16 channels on one code path (library/format.c:30-84: de-interleave, no colour transform below three
// channels).  Until round 4 such images were STAGED here: a pass u8 -> planar int16 in front of the int16 streaming kernel and
// the reverse behind the inverse (ako_plan.hip: staged_level0), two extra trips of the image through memory.  These kernels
// read / write the pixels themselves, in the lean style of ako_u8_lean.hip.h (no private segment, border rules as selects,
// first trip without stores): one wave64 per strip of 128 coefficient columns carrying all CH planes, four pixels per lane and
// row = a 4- or 8-byte load / store.  No pair of waves, no LDS.
//   forward: exact fp32 pipeline (|sample| <= 255, every intermediate far below 2^24 and inside int16: as the RGBA level 0)
//   inverse: the exact int16-wrapping integer pipeline (streams are untrusted; there is no second, "exact" launch behind it)
// Same eligibility as the lean kernels (DD13/7 or CDF5/3, CLAMP / REPEAT / ZERO, level width a multiple of four, ordinary
// strips); the discard rule of gray + alpha (format.c:38-49) is applied at the pixel load.  Everything else stays staged.
#pragma once

namespace ako
{

// ---- forward ---------------------------------------------------------------------------------------------------------
template <int KIND, int CH, bool HEDGE, bool VEDGE>
__device__ __forceinline__ void forward_u8_gray(const LevelParams& P, const StreamGeom& G, const UnitId& id, const LaneCols& lc, int lane)
{
	static_assert(CH == 1 || CH == 2, "gray or gray + alpha");
	constexpr int NP = CH;
	constexpr uint32_t OOB = 0xFFFFFFFFu;
	constexpr int RSRC_FLAGS = 0x00020000;

	const TileDesc td = P.tiles[id.tile];
	const uint64_t inst = (uint64_t)id.image * P.n_tiles + id.tile;
	const int Tc = (int)P.sub_w, Tr = (int)P.sub_h;
	const int chh = (int)P.full_h;
	const int wrap = P.wrap;
	int r_lo, r_hi, seg_len;
	segment_rows(G, id.seg, Tr, r_lo, r_hi, seg_len);
	(void)seg_len;

	const uint8_t* src_base = P.img + (uint64_t)id.image * P.img_stride + ((uint64_t)td.y0 * P.img_pitch + td.x0) * CH;
	const uint32_t row_pitch_b = P.img_pitch * (uint32_t)CH;
	const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src_base), 0, (int)0xFFFFFFFFu, RSRC_FLAGS);
	const uint32_t src_lane_off = (uint32_t)lc.xs * (uint32_t)CH;

	int16_t* tile_stream = P.stream + (uint64_t)id.image * P.stream_stride + td.stream_off;
	const uint64_t stream_left = (P.stream_stride - td.stream_off) * 2;
	const __amdgpu_buffer_rsrc_t rs_stream = __builtin_amdgcn_make_buffer_rsrc(
	    tile_stream, 0, (int)(uint32_t)(stream_left < 0xFFFFFFFFull ? stream_left : 0xFFFFFFFFull), RSRC_FLAGS);
	int16_t* ll_root = P.ll_out_stream ? tile_stream : (P.dst + inst * P.dst_inst_stride);
	const uint64_t ll_left = P.ll_out_stream ? stream_left : (uint64_t)P.channels * P.dst_plane_stride * 2;
	const __amdgpu_buffer_rsrc_t rs_ll = __builtin_amdgcn_make_buffer_rsrc(
	    ll_root, 0, (int)(uint32_t)(ll_left < 0xFFFFFFFFull ? ll_left : 0xFFFFFFFFull), RSRC_FLAGS);
	const uint32_t ll_pitch_b = (P.ll_out_stream ? (uint32_t)Tc : P.dst_pitch) * 2u;
	const uint32_t sub_pitch_b = (uint32_t)Tc * 2u;
	const uint32_t nsub_b = (uint32_t)((uint64_t)Tc * Tr * 2);
	const bool store_lane = lc.net && (lc.c0 >= 0) && (lc.c0 < Tc);
	const uint32_t lane_off = store_lane ? (uint32_t)(lc.c0 * 2) : OOB;
	uint32_t ll_off[NP], grp_off[NP];
	float gf[NP], rq[NP];
#pragma unroll
	for (int p = 0; p < NP; p++)
	{
		grp_off[p] = (uint32_t)((P.grp_off[p] + 1) * 2);
		ll_off[p] = (uint32_t)((P.ll_out_stream ? P.lp_off[p] : (uint64_t)p * P.dst_plane_stride) * 2);
		gf[p] = (float)((p == 0) ? P.g_luma : P.g_chroma);  // lifting.c:202-211: every plane but the first is "chroma", alpha too
		rq[p] = (p == 0) ? P.rq_luma : P.rq_chroma;
		if (id.strip == 0 && id.seg == 0 && lane == 0)  // the lift head (lifting.c:266-267)
			tile_stream[P.grp_off[p]] = (int16_t)((p == 0) ? P.q_luma : P.q_chroma);
	}
	const bool discard = (CH == 2) && P.discard != 0;  // format.c:38-49, 76-79: gray := 0 where alpha == 0

	VFwd<float> st[NP][4];
#pragma unroll
	for (int p = 0; p < NP; p++)
#pragma unroll
		for (int k = 0; k < 4; k++)
			st[p][k] = VFwd<float>{{0, 0, 0}, {0, 0}, {0, 0, 0}};
	const HEdgeBF he = hedge_bf(lc.he);
	const VEdgeBF ve = {wrap != W_REPEAT, wrap == W_ZERO};

	struct Raw
	{
		uint32_t a[2][CH];  // the two pixel rows of a slot: four pixels = CH dwords
	};
	auto fetch = [&](int v, Raw& raw) {
		const int m = VEDGE ? map_index_bf(v, Tr, wrap) : v;
#pragma unroll
		for (int par = 0; par < 2; par++)
		{
			const int y = VEDGE ? min(2 * m + par, chh - 1) : (2 * m + par);  // phantom last row = copy of the last row
			const uint32_t row_off = (uint32_t)y * row_pitch_b;
			if constexpr (CH == 1)
				raw.a[par][0] = __builtin_amdgcn_raw_buffer_load_b32(rs_src, src_lane_off, row_off, AUX_FWD_PIXEL_LOAD);
			else
			{
				const uint2 t = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs_src, src_lane_off, row_off, AUX_FWD_PIXEL_LOAD));
				raw.a[par][0] = t.x, raw.a[par][1] = t.y;
			}
		}
	};
	auto lift_slot = [&](auto kc, const int v, Raw& raw, float (&lp)[NP][4], float (&hp)[NP][4]) {
		constexpr int K = decltype(kc)::value;
		const bool zero_row = VEDGE && ve.zero && ((unsigned)v >= (unsigned)Tr);
		float smp[2][NP][4];
#pragma unroll
		for (int par = 0; par < 2; par++)
		{
			if constexpr (CH == 1)
			{
				const uint32_t w = raw.a[par][0];
				smp[par][0][0] = (float)(w & 255), smp[par][0][1] = (float)((w >> 8) & 255);
				smp[par][0][2] = (float)((w >> 16) & 255), smp[par][0][3] = (float)(w >> 24);
			}
			else
			{
				const uint32_t w0 = raw.a[par][0], w1 = raw.a[par][1];  // g0 a0 g1 a1 | g2 a2 g3 a3
				smp[par][0][0] = (float)(w0 & 255), smp[par][0][1] = (float)((w0 >> 16) & 255);
				smp[par][0][2] = (float)(w1 & 255), smp[par][0][3] = (float)((w1 >> 16) & 255);
				smp[par][1][0] = (float)((w0 >> 8) & 255), smp[par][1][1] = (float)(w0 >> 24);
				smp[par][1][2] = (float)((w1 >> 8) & 255), smp[par][1][3] = (float)(w1 >> 24);
				if (discard)  // wave-uniform
#pragma unroll
					for (int k = 0; k < 4; k++)
						smp[par][0][k] = (smp[par][1][k] == 0.0f) ? 0.0f : smp[par][0][k];
			}
			if constexpr (VEDGE)
#pragma unroll
				for (int p = 0; p < NP; p++)
#pragma unroll
					for (int k = 0; k < 4; k++)
						smp[par][p][k] = zero_row ? 0.0f : smp[par][p][k];
		}
		__builtin_amdgcn_sched_barrier(0);
		fetch(v + 1, raw);
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int p = 0; p < NP; p++)
		{
			float e[4], o[4];
			hlift_forward_bf<KIND, HEDGE, float>(smp[0][p][0], smp[0][p][1], smp[0][p][2], smp[0][p][3], he, e[0], e[1], e[2], e[3]);
			hlift_forward_bf<KIND, HEDGE, float>(smp[1][p][0], smp[1][p][1], smp[1][p][2], smp[1][p][3], he, o[0], o[1], o[2], o[3]);
#pragma unroll
			for (int k = 0; k < 4; k++)
				vstep_forward_bf<KIND, VEDGE, K>(st[p][k], e[k], o[k], v, ve, Tr, lp[p][k], hp[p][k]);
		}
	};
	auto full_slot = [&](auto kc, const int v, Raw& raw) {
		float lp[NP][4], hp[NP][4];
		lift_slot(kc, v, raw, lp, hp);
		const int r = v - 3;
		const bool row_ok = (r >= r_lo) && (r < r_hi);
		const uint32_t row_grp = (uint32_t)r * sub_pitch_b, row_ll = (uint32_t)r * ll_pitch_b;
#pragma unroll
		for (int p = 0; p < NP; p++)
		{
			uint32_t w_ll, w_c, w_b, w_d;
			pack_row_f(lp[p], hp[p], gf[p], rq[p], w_ll, w_c, w_b, w_d);
			__builtin_amdgcn_raw_buffer_store_b32(w_ll, rs_ll, lane_off, row_ok ? ll_off[p] + row_ll : OOB, 0);
			__builtin_amdgcn_raw_buffer_store_b32(w_c, rs_stream, lane_off, row_ok ? grp_off[p] + row_grp : OOB, AUX_FWD_STREAM_STORE);
			__builtin_amdgcn_raw_buffer_store_b32(w_b, rs_stream, lane_off, row_ok ? grp_off[p] + row_grp + nsub_b : OOB, AUX_FWD_STREAM_STORE);
			__builtin_amdgcn_raw_buffer_store_b32(w_d, rs_stream, lane_off, row_ok ? grp_off[p] + row_grp + 2u * nsub_b : OOB, AUX_FWD_STREAM_STORE);
		}
	};

	const int v_begin = r_lo - 3;
	const int n_slots = r_hi + 3 - v_begin;
	Raw ring;
	fetch(v_begin, ring);
	static_for<6>([&](auto kc) {
		constexpr int K = decltype(kc)::value;
		float lp[NP][4], hp[NP][4];
		lift_slot(kc, v_begin + K, ring, lp, hp);
		(void)lp, (void)hp;
#pragma unroll
		for (int p = 0; p < NP; p++)  // (ties the slot's arithmetic to its place: see forward_u8_lean)
			asm volatile("" ::"v"(st[p][0].e[K % 3]), "v"(st[p][1].e[K % 3]), "v"(st[p][2].e[K % 3]), "v"(st[p][3].e[K % 3]),
			             "v"(st[p][0].o[K % 2]), "v"(st[p][1].o[K % 2]), "v"(st[p][2].o[K % 2]), "v"(st[p][3].o[K % 2]),
			             "v"(st[p][0].h[K % 3]), "v"(st[p][1].h[K % 3]), "v"(st[p][2].h[K % 3]), "v"(st[p][3].h[K % 3]));
		__builtin_amdgcn_sched_barrier(0);
		if constexpr (K == 5)
#pragma unroll
			for (int k = 0; k < 4 * NP; k++)
				__builtin_amdgcn_raw_buffer_store_b32(0u, rs_stream, OOB, 0, 0);
	});
	for (int base = 6; base < n_slots; base += 6)
		static_for<6>([&](auto kc) { full_slot(kc, v_begin + base + decltype(kc)::value, ring); });
}

template <int KIND, int CH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_forward_u8_gray(const LevelParams P, const StreamGeom G)
{
	const UnitId id = decode_unit(P, G);
	if (!id.valid)
		return;
	const int lane = threadIdx.x & 63;
	const LaneCols lc = lane_columns(id.strip, G.strips, false, lane, (int)P.sub_w, (int)P.full_w, P.wrap);
	const bool vedge = segment_needs_border_code(G, id.seg, (int)P.sub_h);
	if (__builtin_expect(vedge, 0))
	{
		if (lc.hedge)
			forward_u8_gray<KIND, CH, true, true>(P, G, id, lc, lane);
		else
			forward_u8_gray<KIND, CH, false, true>(P, G, id, lc, lane);
	}
	else
	{
		if (lc.hedge)
			forward_u8_gray<KIND, CH, true, false>(P, G, id, lc, lane);
		else
			forward_u8_gray<KIND, CH, false, false>(P, G, id, lc, lane);
	}
}

// ---- inverse (exact: the int16-wrapping integer pipeline of the reference, wavelet-dd137.c:36-54) ---------------------------
template <int KIND, int CH, bool HEDGE, bool VEDGE>
__device__ __forceinline__ void inverse_u8_gray(const LevelParams& P, const StreamGeom& G, const UnitId& id, const LaneCols& lc, int lane)
{
	static_assert(CH == 1 || CH == 2, "gray or gray + alpha");
	constexpr int NP = CH;
	constexpr uint32_t OOB = 0xFFFFFFFFu;
	constexpr int RSRC_FLAGS = 0x00020000;

	const TileDesc td = P.tiles[id.tile];
	const uint64_t inst = (uint64_t)id.image * P.n_tiles + id.tile;
	const int Tc = (int)P.sub_w, Tr = (int)P.sub_h;
	const int oh = (int)P.full_h;
	const int wrap = P.wrap;
	int r_lo, r_hi, seg_len;
	segment_rows(G, id.seg, Tr, r_lo, r_hi, seg_len);
	(void)seg_len;

	const int16_t* tile_stream = P.stream + (uint64_t)id.image * P.stream_stride + td.stream_off;
	const __amdgpu_buffer_rsrc_t rs_stream = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(tile_stream), 0, (int)0xFFFFFFFFu, RSRC_FLAGS);
	const int16_t* ll_root = P.ll_in_stream ? tile_stream : (P.src + inst * P.src_inst_stride);
	const __amdgpu_buffer_rsrc_t rs_ll = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(ll_root), 0, (int)0xFFFFFFFFu, RSRC_FLAGS);
	const uint32_t ll_pitch_b = (P.ll_in_stream ? (uint32_t)Tc : P.src_pitch) * 2u;
	const uint32_t sub_pitch_b = (uint32_t)Tc * 2u;
	const uint32_t nsub_b = (uint32_t)((uint64_t)Tc * Tr * 2);
	const uint32_t lane_in_off = (uint32_t)lc.cs * 2u;
	uint32_t ll_off[NP], grp_off[NP];
	int q[NP];
#pragma unroll
	for (int p = 0; p < NP; p++)
	{
		q[p] = __builtin_amdgcn_readfirstlane((int)tile_stream[P.grp_off[p]]);  // the decoder trusts the lift head (misc.c:266-272)
		grp_off[p] = (uint32_t)((P.grp_off[p] + 1) * 2);
		ll_off[p] = (uint32_t)((P.ll_in_stream ? P.lp_off[p] : (uint64_t)p * P.src_plane_stride) * 2);
	}
	uint8_t* img = P.img + (uint64_t)id.image * P.img_stride + ((uint64_t)td.y0 * P.img_pitch + td.x0) * CH;
	const uint32_t out_pitch_b = P.img_pitch * (uint32_t)CH;
	const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, (int)0xFFFFFFFFu, RSRC_FLAGS);
	const bool store_lane = lc.net && (lc.c0 >= 0) && (lc.c0 < Tc);
	const uint32_t px_lane_off = store_lane ? (uint32_t)(2 * lc.c0) * (uint32_t)CH : OOB;

	VInv<int> st[NP][4];
#pragma unroll
	for (int p = 0; p < NP; p++)
#pragma unroll
		for (int k = 0; k < 4; k++)
			st[p][k] = VInv<int>{{0, 0, 0}, {0, 0, 0}, 0};
	const HEdgeBF he = hedge_bf(lc.he);
	const VEdgeBF ve = {wrap != W_REPEAT, wrap == W_ZERO};

	struct Raw
	{
		uint32_t ll[NP], c[NP], b[NP], d[NP];
	};
	auto fetch = [&](int v, Raw& raw) {
		const int m = VEDGE ? map_index_bf(v, Tr, wrap) : v;
		const uint32_t row_g = (uint32_t)m * sub_pitch_b, row_l = (uint32_t)m * ll_pitch_b;
#pragma unroll
		for (int p = 0; p < NP; p++)
		{
			const uint32_t g = grp_off[p] + row_g;
			raw.ll[p] = __builtin_amdgcn_raw_buffer_load_b32(rs_ll, lane_in_off, ll_off[p] + row_l, 0);
			raw.c[p] = __builtin_amdgcn_raw_buffer_load_b32(rs_stream, lane_in_off, g, AUX_INV_STREAM_LOAD);
			raw.b[p] = __builtin_amdgcn_raw_buffer_load_b32(rs_stream, lane_in_off, g + nsub_b, AUX_INV_STREAM_LOAD);
			raw.d[p] = __builtin_amdgcn_raw_buffer_load_b32(rs_stream, lane_in_off, g + 2u * nsub_b, AUX_INV_STREAM_LOAD);
		}
	};
	auto column_pass = [&](auto kc, const int v, const Raw& raw, int (&ev)[NP][4], int (&od)[NP][4]) {
		constexpr int K = decltype(kc)::value;
		const bool zero_row = VEDGE && ve.zero && ((unsigned)v >= (unsigned)Tr);
#pragma unroll
		for (int p = 0; p < NP; p++)
		{
			int lpv[4] = {lo16(raw.ll[p]), hi16(raw.ll[p]), lo16(raw.b[p]), hi16(raw.b[p])};  // LL over C | B over D
			int hpv[4] = {lo16(raw.c[p]), hi16(raw.c[p]), lo16(raw.d[p]), hi16(raw.d[p])};
			if (q[p] > 1)  // lifting.c:30-40 (wave-uniform); int16 wrap as the reference's coeff_t
			{
				lpv[2] = (int16_t)(lpv[2] * q[p]), lpv[3] = (int16_t)(lpv[3] * q[p]);
#pragma unroll
				for (int k = 0; k < 4; k++)
					hpv[k] = (int16_t)(hpv[k] * q[p]);
			}
			if constexpr (VEDGE)
#pragma unroll
				for (int k = 0; k < 4; k++)
					lpv[k] = zero_row ? 0 : lpv[k], hpv[k] = zero_row ? 0 : hpv[k];
#pragma unroll
			for (int k = 0; k < 4; k++)
				vstep_inverse_bf<KIND, VEDGE, K, int>(st[p][k], lpv[k], hpv[k], v, ve, Tr, ev[p][k], od[p][k]);
		}
	};
	auto pack_row = [&](const int (&s)[NP][4], uint32_t (&w)[CH]) {  // sSaturateRgb + sInterleave (format.c:221-241)
		if constexpr (CH == 1)
			w[0] = (uint32_t)sat8(s[0][0]) | ((uint32_t)sat8(s[0][1]) << 8) | ((uint32_t)sat8(s[0][2]) << 16) | ((uint32_t)sat8(s[0][3]) << 24);
		else
		{
			w[0] = (uint32_t)sat8(s[0][0]) | ((uint32_t)sat8(s[1][0]) << 8) | ((uint32_t)sat8(s[0][1]) << 16) | ((uint32_t)sat8(s[1][1]) << 24);
			w[1] = (uint32_t)sat8(s[0][2]) | ((uint32_t)sat8(s[1][2]) << 8) | ((uint32_t)sat8(s[0][3]) << 16) | ((uint32_t)sat8(s[1][3]) << 24);
		}
	};
	auto full_slot = [&](auto kc, const int v, const Raw& raw) {
		int ev[NP][4], od[NP][4];
		column_pass(kc, v, raw, ev, od);
		int row[2][NP][4];
#pragma unroll
		for (int p = 0; p < NP; p++)
		{
			hlift_inverse_bf<KIND, HEDGE, int>(ev[p][0], ev[p][1], ev[p][2], ev[p][3], he, row[0][p][0], row[0][p][1], row[0][p][2], row[0][p][3]);
			hlift_inverse_bf<KIND, HEDGE, int>(od[p][0], od[p][1], od[p][2], od[p][3], he, row[1][p][0], row[1][p][1], row[1][p][2], row[1][p][3]);
		}
		const int r = v - 3;
#pragma unroll
		for (int par = 0; par < 2; par++)
		{
			const int y = 2 * r + par;
			const bool row_ok = (r >= r_lo) && (r < r_hi) && (y < oh);  // phantom last row dropped (lifting.c:112,141)
			const uint32_t s_row = row_ok ? (uint32_t)y * out_pitch_b : OOB;
			uint32_t w[CH];
			pack_row(row[par], w);
			if constexpr (CH == 1)
				__builtin_amdgcn_raw_buffer_store_b32(w[0], rs_img, px_lane_off, s_row, AUX_INV_PIXEL_STORE);
			else
			{
				typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
				__builtin_amdgcn_raw_buffer_store_b64(u32x2{w[0], w[1]}, rs_img, px_lane_off, s_row, AUX_INV_PIXEL_STORE);
			}
		}
	};

	const int v_begin = r_lo - 3;
	const int n_slots = r_hi + 3 - v_begin;
	Raw ring[2];
	fetch(v_begin, ring[0]);
	static_for<6>([&](auto kc) {
		constexpr int K = decltype(kc)::value;
		const int v = v_begin + K;
		fetch(v + 1, ring[(K + 1) & 1]);
		__builtin_amdgcn_sched_barrier(0);
		int ev[NP][4], od[NP][4];
		column_pass(kc, v, ring[K & 1], ev, od);
		(void)ev, (void)od;
		__builtin_amdgcn_sched_barrier(0);
		if constexpr (K == 5)
		{
			__builtin_amdgcn_raw_buffer_store_b32(0u, rs_img, OOB, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b32(0u, rs_img, OOB, 0, 0);
		}
	});
	for (int base = 6; base < n_slots; base += 6)
		static_for<6>([&](auto kc) {
			constexpr int K = decltype(kc)::value;
			const int v = v_begin + base + K;
			fetch(v + 1, ring[(K + 1) & 1]);
			__builtin_amdgcn_sched_barrier(0);
			full_slot(kc, v, ring[K & 1]);
		});
}

template <int KIND, int CH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_inverse_u8_gray(const LevelParams P, const StreamGeom G)
{
	const UnitId id = decode_unit(P, G);
	if (!id.valid)
		return;
	const int lane = threadIdx.x & 63;
	const LaneCols lc = lane_columns(id.strip, G.strips, false, lane, (int)P.sub_w, (int)P.full_w, P.wrap);
	const bool vedge = segment_needs_border_code(G, id.seg, (int)P.sub_h);
	if (__builtin_expect(vedge, 0))
	{
		if (lc.hedge)
			inverse_u8_gray<KIND, CH, true, true>(P, G, id, lc, lane);
		else
			inverse_u8_gray<KIND, CH, false, true>(P, G, id, lc, lane);
	}
	else
	{
		if (lc.hedge)
			inverse_u8_gray<KIND, CH, true, false>(P, G, id, lc, lane);
		else
			inverse_u8_gray<KIND, CH, false, false>(P, G, id, lc, lane);
	}
}

}  // namespace ako
