// ako_tail3.hip.h -- the small end of the pyramid, line engine: one workgroup per (tile instance, plane), the plane
// resident in LDS from 256 x 256 samples down, every remaining level in ONE launch per direction.
//
// The window engine (ako_tail.hip.h) works sample by sample with a window copy per level; the streaming kernels
// (ako_stream.hip.h) are a serial row pipeline whose six warm-up slots dominate a level this small and cost a launch per
// level.  Here a level is four data-parallel stages, one wave per row, nothing serial inside a stage:
//
//   forward  (library/lifting.c:43-76,171-292)          inverse  (library/lifting.c:86-148, library/misc.c:229-288)
//     A  row pass: every sample row, predict then          1  assemble: LL (previous level) + de-quantized C, B, D
//        update (wavelet-*.c LiftH)                            (lifting.c:30-40) into one interleaved plane
//     B  column predict: every coefficient row                2  column evens (wavelet-*.c InPlaceishUnliftV)
//        (LiftV), C and D go to the stream                    3  column odds
//     C  column update: LL to the next level's plane,         4  row pass: evens then odds (UnliftH) -> next level's LL
//        B to the stream (gate + quantize: lifting.c:163)        plane / the output plane
//
//   * a lane owns two coefficient columns = four samples of a row ([L H L H] once the row pass has run): 8 bytes, one LDS
//     read; column stages are element-wise over lanes, their taps are whole rows picked by a SCALAR row map
//   * row stages take their neighbour taps by 2-byte LDS gathers at per-lane byte offsets computed once per level: every
//     border rule of SURVEY A.2 -- CLAMP, MIRROR (far tap := opposite near tap), REPEAT, ZERO (a zero sample kept behind
//     every row / a zero row) -- and the phantom last sample / row of odd extents are nothing but entries of these maps;
//     the stages themselves have no border code.  A level may be at most 128 coefficient columns wide (64 lanes x 2)
//   * arithmetic: the wrapping int16 pipeline of the reference everywhere (int32 sums, truncating shifts, narrowing after
//     every step: wavelet-dd137.c:36-54), so untrusted streams decode exactly like the reference
//   * LDS: level l lives in buffer l % 2 (forward: A holds level 0, B a quarter of it) / in A with the hand-over plane in
//     B (inverse); 256 x 256 samples need 128 + 32 KiB = all of a CU's LDS
#pragma once

#include "ako_tail_params.h"
#include "ako_stream.hip.h"

namespace ako
{

// per lane and level: where the taps of the lane's two columns a = 2 * lane, b = a + 1 sit inside a row (byte offsets)
struct T3Cols
{
	int own;     // the lane's four samples
	bool va, vb; // column a / b exists
	int pe[6];   // even samples read by a predict-type step: l1, p1, p2 of column a, then of column b
	int uo[6];   // odd samples read by an update-type step: l2, l1, p1 of column a, then of column b
};

__device__ __forceinline__ T3Cols t3_columns(int lane, int Tc, int wrap)
{
	T3Cols t;
	const int a = 2 * lane;
	t.va = a < Tc, t.vb = a + 1 < Tc;
	t.own = t.va ? 8 * lane : 0;
	const int zero_off = 2 * ((int)t3_pitch((uint32_t)Tc, W_ZERO) - 4);  // the zero samples behind the row (ZERO borders only)
	auto even_off = [&](int m) { return (m < 0) ? zero_off : 4 * m; };
	auto odd_off = [&](int m) { return (m < 0) ? zero_off : 4 * m + 2; };
#pragma unroll
	for (int k = 0; k < 2; k++)
	{
		const int c = min(a + k, Tc - 1);  // (a lane without columns computes on the last one and stores nothing)
		const int l1 = map_index(c - 1, Tc, wrap), p1 = map_index(c + 1, Tc, wrap);
		int p2 = map_index(c + 2, Tc, wrap);
		if (wrap == W_MIRROR && c + 2 >= Tc)
			p2 = l1;  // far tap := opposite near tap
		t.pe[3 * k + 0] = even_off(l1), t.pe[3 * k + 1] = even_off(p1), t.pe[3 * k + 2] = even_off(p2);
		int l2 = map_index(c - 2, Tc, wrap);
		if (wrap == W_MIRROR && c < 2)
			l2 = p1;
		t.uo[3 * k + 0] = odd_off(l2), t.uo[3 * k + 1] = odd_off(l1), t.uo[3 * k + 2] = odd_off(p1);
	}
	return t;
}

__device__ __forceinline__ int t3_ld16(const int16_t* row, int byte_off)
{
	return (int)*reinterpret_cast<const int16_t*>(reinterpret_cast<const char*>(row) + byte_off);
}
__device__ __forceinline__ void t3_st_pair(int16_t* row, int byte_off, int lo, int hi)
{
	*reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(row) + byte_off) = pack2(lo, hi);
}

// Row pass, forward: samples (E O E O) of the lane's two columns -> (L H L H), in place.  One wave per row; the
// intermediate write makes the new odd samples visible to the neighbouring lanes (LDS operations of a wave stay in order).
template <int KIND>
__device__ __forceinline__ void t3_hpass_forward(int16_t* row, const T3Cols& t)
{
	const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(row) + t.own);
	const int Ea = lo16(w.x), Oa = hi16(w.x), Eb = lo16(w.y), Ob = hi16(w.y);
	int Ha, Hb;
	if constexpr (KIND == K_HAAR)
		Ha = nrw<true>(Oa - Ea), Hb = nrw<true>(Ob - Eb);
	else
	{
		const int l1a = (KIND == K_DD137) ? t3_ld16(row, t.pe[0]) : 0, p1a = t3_ld16(row, t.pe[1]);
		const int p2a = (KIND == K_DD137) ? t3_ld16(row, t.pe[2]) : 0;
		const int l1b = (KIND == K_DD137) ? t3_ld16(row, t.pe[3]) : 0, p1b = t3_ld16(row, t.pe[4]);
		const int p2b = (KIND == K_DD137) ? t3_ld16(row, t.pe[5]) : 0;
		Ha = lift_add<true>(Oa, sum_p<KIND, +1>(l1a, Ea, p1a, p2a), shift_p<KIND>());
		Hb = lift_add<true>(Ob, sum_p<KIND, +1>(l1b, Eb, p1b, p2b), shift_p<KIND>());
	}
	if (t.va)
		t3_st_pair(row, t.own, Ea, Ha);
	if (t.vb)
		t3_st_pair(row, t.own + 4, Eb, Hb);
	if constexpr (KIND != K_HAAR)
	{
		const int l2a = (KIND == K_DD137) ? t3_ld16(row, t.uo[0]) : 0, l1a = t3_ld16(row, t.uo[1]);
		const int p1a = (KIND == K_DD137) ? t3_ld16(row, t.uo[2]) : 0;
		const int l2b = (KIND == K_DD137) ? t3_ld16(row, t.uo[3]) : 0, l1b = t3_ld16(row, t.uo[4]);
		const int p1b = (KIND == K_DD137) ? t3_ld16(row, t.uo[5]) : 0;
		const int La = lift_add<true>(Ea, sum_u<KIND, +1>(l2a, l1a, Ha, p1a), shift_u<KIND>());
		const int Lb = lift_add<true>(Eb, sum_u<KIND, +1>(l2b, l1b, Hb, p1b), shift_u<KIND>());
		if (t.va)
			t3_st_pair(row, t.own, La, Ha);
		if (t.vb)
			t3_st_pair(row, t.own + 4, Lb, Hb);
	}
}

// Row pass, inverse: (L H L H) -> samples (E O E O); returns them (the caller stores: next level's plane or the output)
template <int KIND>
__device__ __forceinline__ void t3_hpass_inverse(int16_t* row, const T3Cols& t, int out[4])
{
	const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(row) + t.own);
	const int La = lo16(w.x), Ha = hi16(w.x), Lb = lo16(w.y), Hb = hi16(w.y);
	int Ea = La, Eb = Lb;
	if constexpr (KIND != K_HAAR)
	{
		const int l2a = (KIND == K_DD137) ? t3_ld16(row, t.uo[0]) : 0, l1a = t3_ld16(row, t.uo[1]);
		const int p1a = (KIND == K_DD137) ? t3_ld16(row, t.uo[2]) : 0;
		const int l2b = (KIND == K_DD137) ? t3_ld16(row, t.uo[3]) : 0, l1b = t3_ld16(row, t.uo[4]);
		const int p1b = (KIND == K_DD137) ? t3_ld16(row, t.uo[5]) : 0;
		Ea = lift_add<true>(La, sum_u<KIND, -1>(l2a, l1a, Ha, p1a), shift_u<KIND>());
		Eb = lift_add<true>(Lb, sum_u<KIND, -1>(l2b, l1b, Hb, p1b), shift_u<KIND>());
		if (t.va)
			t3_st_pair(row, t.own, Ea, Ha);
		if (t.vb)
			t3_st_pair(row, t.own + 4, Eb, Hb);
	}
	int Oa, Ob;
	if constexpr (KIND == K_HAAR)
		Oa = lift_add<true>(La, Ha, 0), Ob = lift_add<true>(Lb, Hb, 0);
	else
	{
		const int l1a = (KIND == K_DD137) ? t3_ld16(row, t.pe[0]) : 0, p1a = t3_ld16(row, t.pe[1]);
		const int p2a = (KIND == K_DD137) ? t3_ld16(row, t.pe[2]) : 0;
		const int l1b = (KIND == K_DD137) ? t3_ld16(row, t.pe[3]) : 0, p1b = t3_ld16(row, t.pe[4]);
		const int p2b = (KIND == K_DD137) ? t3_ld16(row, t.pe[5]) : 0;
		Oa = lift_add<true>(Ha, sum_p<KIND, -1>(l1a, Ea, p1a, p2a), shift_p<KIND>());
		Ob = lift_add<true>(Hb, sum_p<KIND, -1>(l1b, Eb, p1b, p2b), shift_p<KIND>());
	}
	out[0] = Ea, out[1] = Oa, out[2] = Eb, out[3] = Ob;
}

// the four values a lane holds of one row
__device__ __forceinline__ void t3_ld_row(const int16_t* row, int own, int v[4])
{
	const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(row) + own);
	v[0] = lo16(w.x), v[1] = hi16(w.x), v[2] = lo16(w.y), v[3] = hi16(w.y);
}
__device__ __forceinline__ void t3_st_row(int16_t* row, const T3Cols& t, const int v[4])
{
	if (t.vb)
		*reinterpret_cast<uint2*>(reinterpret_cast<char*>(row) + t.own) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
	else if (t.va)
		t3_st_pair(row, t.own, v[0], v[1]);
}

// rows of a column stage: the plane's row of coefficient index m and parity par (0 even / low, 1 odd / high), the zero
// row for a tap that reads zero
struct T3Plane
{
	int16_t* base;
	const int16_t* zero_row;
	int pitch;
	__device__ __forceinline__ const int16_t* row(int m, int par) const
	{
		return (m < 0) ? zero_row : base + (2 * m + par) * pitch;
	}
	__device__ __forceinline__ int16_t* wrow(int m, int par) const
	{
		return base + (2 * m + par) * pitch;
	}
};

// column predict-type step of coefficient row u: returns base[k] +/- P(even rows u-1 .. u+2)
template <int KIND, int SGN>
__device__ __forceinline__ void t3_vpredict(const T3Plane& pl, int u, int Tr, int wrap, int own, const int base[4], int res[4])
{
	int e[4], l1[4] = {0, 0, 0, 0}, p1[4] = {0, 0, 0, 0}, p2[4] = {0, 0, 0, 0};
	t3_ld_row(pl.row(u, 0), own, e);
	if constexpr (KIND != K_HAAR)
		t3_ld_row(pl.row(map_index(u + 1, Tr, wrap), 0), own, p1);
	if constexpr (KIND == K_DD137)
	{
		const int ml1 = map_index(u - 1, Tr, wrap);
		int mp2 = map_index(u + 2, Tr, wrap);
		if (wrap == W_MIRROR && u + 2 >= Tr)
			mp2 = ml1;
		t3_ld_row(pl.row(ml1, 0), own, l1);
		t3_ld_row(pl.row(mp2, 0), own, p2);
	}
#pragma unroll
	for (int k = 0; k < 4; k++)
		res[k] = lift_add<true>(base[k], sum_p<KIND, SGN>(l1[k], e[k], p1[k], p2[k]), shift_p<KIND>());
}
// column update-type step of coefficient row r: base[k] +/- U(odd rows r-2 .. r+1)
template <int KIND, int SGN>
__device__ __forceinline__ void t3_vupdate(const T3Plane& pl, int r, int Tr, int wrap, int own, const int base[4], int res[4])
{
	if constexpr (KIND == K_HAAR)
	{
#pragma unroll
		for (int k = 0; k < 4; k++)
			res[k] = base[k];
		return;
	}
	int h[4], l1[4], l2[4] = {0, 0, 0, 0}, p1[4] = {0, 0, 0, 0};
	t3_ld_row(pl.row(r, 1), own, h);
	const int ml1 = map_index(r - 1, Tr, wrap);
	t3_ld_row(pl.row(ml1, 1), own, l1);
	if constexpr (KIND == K_DD137)
	{
		const int mp1 = map_index(r + 1, Tr, wrap);
		int ml2 = map_index(r - 2, Tr, wrap);
		if (wrap == W_MIRROR && r < 2)
			ml2 = mp1;
		t3_ld_row(pl.row(ml2, 1), own, l2);
		t3_ld_row(pl.row(mp1, 1), own, p1);
	}
#pragma unroll
	for (int k = 0; k < 4; k++)
		res[k] = lift_add<true>(base[k], sum_u<KIND, SGN>(l2[k], l1[k], h[k], p1[k]), shift_u<KIND>());
}

struct T3Ctx
{
	int wave, nwaves, lane, wrap;
	int16_t* zero_row;
	int16_t* bufA;
	int16_t* bufB;
};

// a pair of coefficients -> the stream: 4 bytes per lane, 2 where the lane holds the last column of an odd count
__device__ __forceinline__ void t3_store_pair(const __amdgpu_buffer_rsrc_t& rs, uint32_t byte_off, const T3Cols& t, int lo, int hi)
{
	if (t.vb)
		__builtin_amdgcn_raw_buffer_store_b32(pack2(lo, hi), rs, byte_off, 0, 0);
	else if (t.va)
		__builtin_amdgcn_raw_buffer_store_b16((unsigned short)lo, rs, byte_off, 0, 0);
}
__device__ __forceinline__ void t3_load_pair(const __amdgpu_buffer_rsrc_t& rs, uint32_t byte_off, const T3Cols& t, int& lo, int& hi)
{
	lo = hi = 0;
	if (t.vb)
	{
		const uint32_t w = __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 0);
		lo = lo16(w), hi = hi16(w);
	}
	else if (t.va)
		lo = (int)(int16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, byte_off, 0, 0);
}

template <int KIND>
__device__ __forceinline__ void t3_forward_level(const T3Ctx& c, const TailLevel& L, int16_t* cur, int16_t* next, int next_pitch,
                                                 bool last, const __amdgpu_buffer_rsrc_t& rs, uint32_t grp_b, uint32_t lp_b, int m)
{
	const int Tc = (int)L.tw, Tr = (int)L.th, pitch = (int)t3_pitch(L.tw, c.wrap);
	const T3Cols t = t3_columns(c.lane, Tc, c.wrap);
	const T3Plane pl{cur, c.zero_row, pitch};
	const float gf = (float)L.g[m], rq = L.rq[m];
	const uint32_t nsub_b = (uint32_t)(Tc * Tr * 2);
	const uint32_t col_b = (uint32_t)(4 * c.lane);  // byte offset of column a inside a sub-band row

	// A: row pass over every sample row (phantom row included: it is a copy of the last row, so it lifts to a copy)
	for (int y = c.wave; y < 2 * Tr; y += c.nwaves)
		t3_hpass_forward<KIND>(cur + y * pitch, t);
	__syncthreads();
	// B: column predict; the odd rows become HP rows, C (low columns) and D (high columns) leave for the stream
	for (int u = c.wave; u < Tr; u += c.nwaves)
	{
		int o[4], hp[4];
		t3_ld_row(pl.row(u, 1), t.own, o);
		t3_vpredict<KIND, +1>(pl, u, Tr, c.wrap, t.own, o, hp);
		t3_st_row(pl.wrow(u, 1), t, hp);
		const uint32_t row_b = (uint32_t)(u * Tc * 2);
		t3_store_pair(rs, grp_b + row_b + col_b, t, quantize_f(hp[0], gf, rq), quantize_f(hp[2], gf, rq));                // C
		t3_store_pair(rs, grp_b + 2u * nsub_b + row_b + col_b, t, quantize_f(hp[1], gf, rq), quantize_f(hp[3], gf, rq));  // D
	}
	__syncthreads();
	// C: column update; LL becomes the next level's plane (phantom column / row materialised), B leaves for the stream
	for (int r = c.wave; r < Tr; r += c.nwaves)
	{
		int e[4], lp[4];
		t3_ld_row(pl.row(r, 0), t.own, e);
		t3_vupdate<KIND, +1>(pl, r, Tr, c.wrap, t.own, e, lp);
		const uint32_t row_b = (uint32_t)(r * Tc * 2);
		t3_store_pair(rs, grp_b + nsub_b + row_b + col_b, t, quantize_f(lp[1], gf, rq), quantize_f(lp[3], gf, rq));  // B
		if (last)
			t3_store_pair(rs, lp_b + row_b + col_b, t, lp[0], lp[2]);
		else if (t.va)
		{
			// the next level's sample row r, samples a and a + 1 (an odd column count ends in a phantom copy)
			const int hi = t.vb ? lp[2] : lp[0];
			t3_st_pair(next + r * next_pitch, 4 * c.lane, lp[0], hi);
			if (r == Tr - 1 && (Tr & 1))
				t3_st_pair(next + Tr * next_pitch, 4 * c.lane, lp[0], hi);  // phantom row of an odd row count
		}
	}
	__syncthreads();
}

__global__ __launch_bounds__(1024) void k_forward_tail3(const TailParams P)
{
	extern __shared__ __attribute__((aligned(16))) int16_t t3_lds[];
	const int wrap = P.wrap;
	T3Ctx c;
	c.wave = threadIdx.x >> 6, c.nwaves = blockDim.x >> 6, c.lane = threadIdx.x & 63, c.wrap = wrap;
	const uint32_t zero_elems = t3_zero_elems(P.lv[0].tw, wrap);
	c.zero_row = t3_lds;
	c.bufA = t3_lds + zero_elems;
	c.bufB = c.bufA + P.win_elems;
	const uint32_t p = blockIdx.x % P.channels;
	const uint64_t inst = blockIdx.x / P.channels;
	const uint32_t tile = (uint32_t)(inst % P.n_tiles), image = (uint32_t)(inst / P.n_tiles);
	const TileDesc td = P.tiles[tile];
	int16_t* tile_stream = P.stream + (uint64_t)image * P.stream_stride + td.stream_off;
	const uint64_t stream_left = (P.stream_stride - td.stream_off) * 2;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
	    tile_stream, 0, (int)(uint32_t)(stream_left < 0xFFFFFFFFull ? stream_left : 0xFFFFFFFFull), 0x00020000);
	const int m = (p == 0) ? 0 : 1;

	// the plane -> buffer A: sample rows 0 .. 2 Tr - 1 of 2 Tc samples, the phantom last column / row of an odd extent as a
	// copy of its neighbour (wavelet-dd137.c:128-132, lifting.c:70-72); ZERO borders: zero samples behind every row
	{
		const TailLevel& L = P.lv[0];
		const int Tc = (int)L.tw, Tr = (int)L.th, cw = (int)L.cw, chh = (int)L.ch, pitch = (int)t3_pitch(L.tw, wrap);
		const int16_t* gsrc = P.plane + (P.plane_tiled ? (uint64_t)image : inst) * P.plane_inst_stride + (uint64_t)p * P.plane_plane_stride;
		if (P.plane_tiled)
			gsrc += (uint64_t)td.y0 * P.plane_pitch + td.x0;
		for (uint32_t i = threadIdx.x; i < zero_elems; i += blockDim.x)
			t3_lds[i] = 0;
		for (int y = c.wave; y < 2 * Tr; y += c.nwaves)
		{
			const int16_t* srow = gsrc + (uint64_t)min(y, chh - 1) * P.plane_pitch;
			for (int x = c.lane; x < pitch; x += 64)
				c.bufA[y * pitch + x] = (x < 2 * Tc) ? srow[min(x, cw - 1)] : (int16_t)0;
		}
		__syncthreads();
	}
	for (uint32_t l = 0; l < P.nlev; l++)
	{
		const TailLevel& L = P.lv[l];
		int16_t* cur = (l & 1) ? c.bufB : c.bufA;
		int16_t* next = (l & 1) ? c.bufA : c.bufB;
		const bool last = (l + 1 == P.nlev);
		const int next_pitch = last ? 0 : (int)t3_pitch(P.lv[l + 1].tw, wrap);
		if (!last && wrap == W_ZERO)
		{
			// the zero samples behind the rows of the next level's plane
			const int rows = 2 * (int)P.lv[l + 1].th, tc2 = 2 * (int)P.lv[l + 1].tw;
			(void)tc2;
			for (int i = threadIdx.x; i < rows * 4; i += blockDim.x)
				next[(i >> 2) * next_pitch + next_pitch - 4 + (i & 3)] = 0;
		}
		const uint32_t grp_b = (uint32_t)((L.grp0 + (uint64_t)p * L.gsize + 1) * 2);
		const uint32_t lp_b = (uint32_t)((uint64_t)p * P.fw * P.fh * 2);
		if (threadIdx.x == 0)
			tile_stream[L.grp0 + (uint64_t)p * L.gsize] = (int16_t)L.q[m];
		if (L.kind == K_DD137)
			t3_forward_level<K_DD137>(c, L, cur, next, next_pitch, last, rs, grp_b, lp_b, m);
		else if (L.kind == K_CDF53)
			t3_forward_level<K_CDF53>(c, L, cur, next, next_pitch, last, rs, grp_b, lp_b, m);
		else
			t3_forward_level<K_HAAR>(c, L, cur, next, next_pitch, last, rs, grp_b, lp_b, m);
	}
}

template <int KIND>
__device__ __forceinline__ void t3_inverse_level(const T3Ctx& c, const TailLevel& L, int16_t* cur, const int16_t* ll, int ll_pitch,
                                                 bool ll_global, const __amdgpu_buffer_rsrc_t& rs, uint32_t grp_b, uint32_t lp_b,
                                                 int16_t* out, uint64_t out_pitch)
{
	const int Tc = (int)L.tw, Tr = (int)L.th, ow = (int)L.cw, oh = (int)L.ch, pitch = (int)t3_pitch(L.tw, c.wrap);
	const T3Cols t = t3_columns(c.lane, Tc, c.wrap);
	const T3Plane pl{cur, c.zero_row, pitch};
	const uint32_t nsub_b = (uint32_t)(Tc * Tr * 2);
	const uint32_t col_b = (uint32_t)(4 * c.lane);
	const int q = __builtin_amdgcn_readfirstlane((int)(int16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, 0, grp_b - 2u, 0));  // the lift head

	// 1: assemble the interleaved plane: even rows [LL B LL B], odd rows [C D C D]; de-quantization narrows (lifting.c:30-40)
	for (int r = c.wave; r < Tr; r += c.nwaves)
	{
		const uint32_t row_b = (uint32_t)(r * Tc * 2);
		int lo[4], hi[4];
		if (ll_global)
			t3_load_pair(rs, lp_b + row_b + col_b, t, lo[0], lo[2]);
		else
		{
			// (the hand-over plane is dense with an arbitrary pitch: element-wise)
			const int16_t* lrow = ll + r * ll_pitch + (t.va ? 2 * c.lane : 0);
			lo[0] = lrow[0], lo[2] = t.vb ? lrow[1] : 0;
		}
		t3_load_pair(rs, grp_b + row_b + col_b, t, hi[0], hi[2]);                // C
		t3_load_pair(rs, grp_b + nsub_b + row_b + col_b, t, lo[1], lo[3]);       // B
		t3_load_pair(rs, grp_b + 2u * nsub_b + row_b + col_b, t, hi[1], hi[3]);  // D
		if (q > 1)
		{
			lo[1] = (int16_t)(lo[1] * q), lo[3] = (int16_t)(lo[3] * q);
#pragma unroll
			for (int k = 0; k < 4; k++)
				hi[k] = (int16_t)(hi[k] * q);
		}
		t3_st_row(pl.wrow(r, 0), t, lo);
		t3_st_row(pl.wrow(r, 1), t, hi);
	}
	__syncthreads();
	// 2: column evens
	for (int r = c.wave; r < Tr; r += c.nwaves)
	{
		int lp[4], e[4];
		t3_ld_row(pl.row(r, 0), t.own, lp);
		t3_vupdate<KIND, -1>(pl, r, Tr, c.wrap, t.own, lp, e);
		t3_st_row(pl.wrow(r, 0), t, e);
	}
	__syncthreads();
	// 3: column odds
	for (int u = c.wave; u < Tr; u += c.nwaves)
	{
		int hp[4], o[4];
		t3_ld_row(pl.row(u, 1), t.own, hp);
		t3_vpredict<KIND, -1>(pl, u, Tr, c.wrap, t.own, hp, o);
		t3_st_row(pl.wrow(u, 1), t, o);
	}
	__syncthreads();
	// 4: row pass; the phantom last row / column of an odd extent is dropped (lifting.c:111-112,140-142)
	for (int y = c.wave; y < oh; y += c.nwaves)
	{
		int s[4];
		t3_hpass_inverse<KIND>(cur + y * pitch, t, s);
		int16_t* orow = out + (uint64_t)y * out_pitch + 4 * c.lane;
		const int nv = min(max(ow - 4 * c.lane, 0), 4);
#pragma unroll
		for (int k = 0; k < 4; k++)
			if (k < nv)
				orow[k] = (int16_t)s[k];
	}
	__syncthreads();
}

__global__ __launch_bounds__(1024) void k_inverse_tail3(const TailParams P)
{
	extern __shared__ __attribute__((aligned(16))) int16_t t3_lds[];
	const int wrap = P.wrap;
	T3Ctx c;
	c.wave = threadIdx.x >> 6, c.nwaves = blockDim.x >> 6, c.lane = threadIdx.x & 63, c.wrap = wrap;
	const uint32_t zero_elems = t3_zero_elems(P.lv[0].tw, wrap);
	c.zero_row = t3_lds;
	c.bufA = t3_lds + zero_elems;
	c.bufB = c.bufA + P.win_elems;
	const uint32_t p = blockIdx.x % P.channels;
	const uint64_t inst = blockIdx.x / P.channels;
	const uint32_t tile = (uint32_t)(inst % P.n_tiles), image = (uint32_t)(inst / P.n_tiles);
	const TileDesc td = P.tiles[tile];
	const int16_t* tile_stream = P.stream + (uint64_t)image * P.stream_stride + td.stream_off;
	const uint64_t stream_left = (P.stream_stride - td.stream_off) * 2;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
	    const_cast<int16_t*>(tile_stream), 0, (int)(uint32_t)(stream_left < 0xFFFFFFFFull ? stream_left : 0xFFFFFFFFull), 0x00020000);
	for (uint32_t i = threadIdx.x; i < zero_elems; i += blockDim.x)
		t3_lds[i] = 0;

	// smallest level first (misc.c:257-285); its LL is the stream's low-pass section.  Every level's plane lives in buffer
	// A, the plane it hands to the next (larger) level in buffer B.
	for (int l = (int)P.nlev - 1; l >= 0; l--)
	{
		const TailLevel& L = P.lv[l];
		const int pitch = (int)t3_pitch(L.tw, wrap);
		if (wrap == W_ZERO)
		{
			for (int i = threadIdx.x; i < 2 * (int)L.th * 4; i += blockDim.x)
				c.bufA[(i >> 2) * pitch + pitch - 4 + (i & 3)] = 0;
		}
		const bool smallest = (l + 1 == (int)P.nlev);
		const uint32_t grp_b = (uint32_t)((L.grp0 + (uint64_t)p * L.gsize + 1) * 2);
		const uint32_t lp_b = (uint32_t)((uint64_t)p * P.fw * P.fh * 2);
		int16_t* out;
		uint64_t out_pitch;
		if (l != 0)
			out = c.bufB, out_pitch = L.cw;
		else
		{
			out = P.plane + (P.plane_tiled ? (uint64_t)image : inst) * P.plane_inst_stride + (uint64_t)p * P.plane_plane_stride;
			if (P.plane_tiled)
				out += (uint64_t)td.y0 * P.plane_pitch + td.x0;
			out_pitch = P.plane_pitch;
		}
		if (L.kind == K_DD137)
			t3_inverse_level<K_DD137>(c, L, c.bufA, c.bufB, (int)L.tw, smallest, rs, grp_b, lp_b, out, out_pitch);
		else if (L.kind == K_CDF53)
			t3_inverse_level<K_CDF53>(c, L, c.bufA, c.bufB, (int)L.tw, smallest, rs, grp_b, lp_b, out, out_pitch);
		else
			t3_inverse_level<K_HAAR>(c, L, c.bufA, c.bufB, (int)L.tw, smallest, rs, grp_b, lp_b, out, out_pitch);
	}
}

}  // namespace ako
