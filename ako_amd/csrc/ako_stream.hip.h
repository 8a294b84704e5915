// ako_stream.hip.h -- register-streaming level kernels for gfx950 (the fast path of large levels).
//
// Same arithmetic as the generic window engine (ako_kernels.hip.h), different data movement:
//
//   * NO LDS and NO barriers.  One wave64 owns a vertical strip of 128 coefficient columns
//     (120 net + 4 halo columns on either side) and walks a segment of it top to bottom.
//   * every lane owns two adjacent coefficient columns (four samples of a row): on the u8 side
//     that is one 16-byte load / store of four RGBA pixels per lane and row.  To keep the register
//     file at >= 2 waves per SIMD the four planes of an RGBA strip are split over a PAIR of waves
//     (planes 0,1 and planes 2,3): forward, both waves load the same pixels (L1/L2 hits) and each
//     colour-transforms what it needs; inverse, the pair swaps two planes of one row through LDS
//     (16 bytes per lane and row slot, one barrier) so that each wave finishes one pixel row
//   * the horizontal pass takes its neighbour taps from the adjacent lanes with DPP whole-wave
//     shifts (wave_shr:1 / wave_shl:1) -- 6 per row and plane for DD13/7, 2 for CDF5/3
//   * the vertical pass is a software pipeline in registers: per column a lane keeps the last few
//     rows of each sequence in small rings; the row loop is unrolled by 6 (= lcm of the ring
//     periods) so that ring indices are compile-time constants and no register moves are needed
//   * row data is prefetched two row-slots ahead of its use
//   * colour transform is fused in front (forward) / behind (inverse); gate + quantization and the
//     stream packing are fused into the stores (forward), de-quantization into the loads (inverse)
//
// Boundary rules are the same closed form as in the window engine:
//   rows   : the row slot fed to the pipeline is map_index(v) (CLAMP/MIRROR nearest, REPEAT modulo,
//            ZERO zeros); the halo slots of the sequence the first step produces are patched in
//            the pipeline (nearest / zero)
//   columns: REPEAT wraps the lane's load address; CLAMP/MIRROR/ZERO overwrite the out-of-range
//            lanes of the input sequence and then of the produced sequence (v_readlane broadcast)
//   MIRROR : the far taps take the opposite near tap
// The boundary code is compiled out (template flags HEDGE / VEDGE) for waves whose strip / segment
// does not touch a tile border.  Levels whose width is not a multiple of 4, and everything small,
// stay on the window engine.
#pragma once

#include "ako_kernels.hip.h"

#include <type_traits>
#include <utility>

namespace ako
{

// compile-time unrolled loop: f(std::integral_constant<int, 0>{}) ... f(<N-1>)
template <int... Ks, typename F>
__device__ __forceinline__ void static_for_seq(std::integer_sequence<int, Ks...>, F&& f)
{
	(f(std::integral_constant<int, Ks>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
	static_for_seq(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// Cache policy (the aux operand of the raw buffer instructions; bit 1 = nt) of data that is touched ONCE: kept at
// the default everywhere.  Measured on the box (scripts/ab_libs.sh / scripts/ab_steps.py, round 2): non-temporal
// 4-byte-per-lane stream stores cost the level-0 forward kernel +15...35 % (although the level-1 kernel behind it
// then finds its input still in the memory-side cache, -25 %), non-temporal pixel loads +10 % (the second wave
// of a pair relies on finding the first one's lines in L2); on the inverse side non-temporal stream loads and
// pixel stores take 4-5 % off the level-0 kernel run alone but cost 5 % of the throughput with four steps in
// flight.  (-DAKO_INV_POLICY=2 rebuilds with the inverse hint for experiments.)
#ifndef AKO_PERM
#define AKO_PERM 0  // 1: the row pass takes its neighbours' values through ds_bpermute (LDS pipe) instead of DPP moves (VALU)
#endif
#ifndef AKO_CUT
#define AKO_CUT 0  // measurement builds: parts of the level-0 forward arithmetic compiled out (see forward_stream_body)
#endif
#ifndef AKO_INV_POLICY
#define AKO_INV_POLICY 0
#endif
constexpr int AUX_FWD_PIXEL_LOAD = 0;
constexpr int AUX_FWD_STREAM_STORE = 0;
constexpr int AUX_INV_STREAM_LOAD = AKO_INV_POLICY;
constexpr int AUX_INV_PIXEL_STORE = AKO_INV_POLICY;

constexpr int SNET = 120;  // net coefficient columns per wave
constexpr int SORG = 4;    // lane 0 holds coefficient columns strip * SNET - SORG, +1
constexpr int SWAVES = THREADS / 64;

// lane i <- lane i-1 / lane i+1 over the whole wave64 (gfx9 DPP wave shifts)
__device__ __forceinline__ int from_prev_lane(int x)
{
	return __builtin_amdgcn_mov_dpp(x, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}
__device__ __forceinline__ int from_next_lane(int x)
{
	return __builtin_amdgcn_mov_dpp(x, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
}

__device__ __forceinline__ float from_prev_lane(float x)
{
	return __int_as_float(from_prev_lane(__float_as_int(x)));
}
__device__ __forceinline__ float from_next_lane(float x)
{
	return __int_as_float(from_next_lane(__float_as_int(x)));
}
// the same shifts on the LDS pipe (ds_bpermute_b32: no LDS memory involved, no VALU cycles): idx = 4 * source lane
__device__ __forceinline__ int perm_lane(int idx, int x)
{
	return __builtin_amdgcn_ds_bpermute(idx, x);
}
__device__ __forceinline__ float perm_lane(int idx, float x)
{
	return __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(x)));
}
__device__ __forceinline__ int read_lane(int x, int lane)
{
	return __builtin_amdgcn_readlane(x, lane);
}
__device__ __forceinline__ float read_lane(float x, int lane)
{
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}

template <bool NARROW>
__device__ __forceinline__ int nrw(int v)
{
	return NARROW ? (int)(int16_t)v : v;
}

// ---- float pipeline -------------------------------------------------------------------------
// Where int16 narrowing is provably a no-op (forward levels 0 and 1 of u8 data, see the plan) the
// same integer arithmetic is carried out on the fp32 pipe: every value is an integer below 2^24 and
// every scaled sum a multiple of 2^-5 below 2^19, so each add / fma / multiply by 2^-k is exact and
// v_trunc_f32 is C's truncating division.  6 ops per lifting step instead of 7-8, u8 -> float
// conversion for free (v_cvt_f32_ubyteN), no int -> float conversion in front of the quantizer.
template <bool NARROW>
__device__ __forceinline__ float lift_add(float base, float sum, int k)
{
	static_assert(!NARROW, "the float pipeline cannot wrap to int16");
	return (k == 0) ? (base + sum) : (base + __builtin_truncf(sum * (1.0f / (float)(1 << k))));
}
template <int KIND, int SGN>
__device__ __forceinline__ float sum_p(float l1, float e, float p1, float p2)
{
	if (KIND == K_DD137)
		return (SGN > 0) ? __builtin_fmaf(e + p1, -9.0f, l1 + p2) : (__builtin_fmaf(e + p1, 9.0f, -(l1 + p2)));
	if (KIND == K_CDF53)
		return (SGN > 0) ? -(e + p1) : (e + p1);
	return (SGN > 0) ? -e : e;
}
template <int KIND, int SGN>
__device__ __forceinline__ float sum_u(float l2, float l1, float h, float p1)
{
	if (KIND == K_DD137)
		return (SGN > 0) ? __builtin_fmaf(l1 + h, 9.0f, -(l2 + p1)) : __builtin_fmaf(l1 + h, -9.0f, l2 + p1);
	if (KIND == K_CDF53)
		return (SGN > 0) ? (l1 + h) : -(l1 + h);
	return 0.0f;
}

// Lifting arithmetic (wavelet-dd137.c:36-54, wavelet-cdf53.c:36-54, wavelet-haar.c:41,68).
// A step is  base +/- trunc(sum / 2^k)  narrowed to int16.  With bias = (sum < 0) ? 2^k - 1 : 0,
//   base + trunc(sum / 2^k) == ((base << k) + sum + bias) >> k        (arithmetic shift)
// which costs: sign bit, shift-add, multiply-add (24 bit), bit-field extract (shift + int16 wrap).
// Subtraction is done by negating the sum (trunc is odd): base - trunc(s / 2^k) = base + trunc(-s / 2^k).
template <bool NARROW>
__device__ __forceinline__ int lift_add(int base, int sum, int k)
{
	const int neg = (int)((unsigned)sum >> 31);
	const int x = (base << k) + sum + __mul24(neg, (1 << k) - 1);
	return NARROW ? (int)(int16_t)(x >> k) : (x >> k);
}

// predict-like sum from the even taps l1 = c-1, e = c, p1 = c+1, p2 = c+2; SGN = -1 gives the negated sum
template <int KIND, int SGN>
__device__ __forceinline__ int sum_p(int l1, int e, int p1, int p2)
{
	if (KIND == K_DD137)
		return (SGN > 0) ? (l1 + p2 + __mul24(e + p1, -9)) : (__mul24(e + p1, 9) - (l1 + p2));
	if (KIND == K_CDF53)
		return (SGN > 0) ? -(e + p1) : (e + p1);
	return (SGN > 0) ? -e : e;
}
// update-like sum from the high-pass taps l2 = c-2, l1 = c-1, h = c, p1 = c+1
template <int KIND, int SGN>
__device__ __forceinline__ int sum_u(int l2, int l1, int h, int p1)
{
	if (KIND == K_DD137)
		return (SGN > 0) ? (__mul24(l1 + h, 9) - (l2 + p1)) : (l2 + p1 + __mul24(l1 + h, -9));
	if (KIND == K_CDF53)
		return (SGN > 0) ? (l1 + h) : -(l1 + h);
	return 0;
}
template <int KIND>
constexpr int shift_p()
{
	return KIND == K_DD137 ? 4 : (KIND == K_CDF53 ? 1 : 0);
}
template <int KIND>
constexpr int shift_u()
{
	return KIND == K_DD137 ? 5 : (KIND == K_CDF53 ? 2 : 0);
}

// what a lane needs to know about the left / right tile border
struct HEdge
{
	bool left, right;   // this strip holds out-of-range lanes on that side (never set for REPEAT)
	bool oob_l, oob_r;  // this lane is such a lane
	bool first, last;   // this lane holds columns 0,1 / T-2,T-1
	bool half;          // this lane holds column T-1 in its FIRST slot (odd T, strip not shifted): see lane_columns()
	bool drop_last;     // odd level width: this lane's fourth sample is the phantom one (lifting.c:111-112,140-142)
	bool nh_left, nh_right;  // "wide" strips: the border sits at lane 0 / lane 63, there are no out-of-range lanes on
	                         // that side to hold the border values (see edge_taps)
	int lane_first, lane_last;
	bool bf;         // the border rule of this strip is a plain tap substitution in its first / last lane (CLAMP / ZERO, no out-of-range
	                 // lanes to patch, no phantom sample): hlift_*() take the branch-free hlift_*_bf() (wave-uniform)
	bool rep_lanes;  // REPEAT over several tiles in one wave (lane_columns_pack): lane_first / lane_last are THIS lane's tile's, fetched by ds_bpermute
	int wrap;
	int perm_prev, perm_next;  // AKO_PERM: 4 * (lane - 1), 4 * (lane + 1) (mod 64): ds_bpermute addresses of the neighbours
};

// overwrite the out-of-range lanes of a two-column sequence (a0 = column c0, a1 = column c1) with
// its nearest in-range value (CLAMP / MIRROR) or zero
template <typename V>
__device__ __forceinline__ void fix_halo_lanes(V& a0, V& a1, const HEdge& ed)
{
	if (ed.left)
	{
		const V f = (ed.wrap == W_ZERO) ? (V)0 : read_lane(a0, ed.lane_first);
		if (ed.oob_l)
			a0 = f, a1 = f;
	}
	if (ed.right)
	{
		const V f = (ed.wrap == W_ZERO) ? (V)0 : read_lane(a1, ed.lane_last);
		if (ed.oob_r)
			a0 = f, a1 = f;
	}
}

// Wide strips (a tile of 121..128 coefficient columns in ONE wave, no halo lanes): what the neighbour
// shifts deliver at the two border lanes is replaced by the border values of the sequence (a0 = column c0,
// a1 = column c1): prev1 / prev2 = columns c0-1 / c0-2 as seen by the FIRST lane, next1 / next2 = columns
// c1+1 / c1+2 as seen by the LAST lane.  CLAMP and MIRROR take the nearest in-range value (MIRROR's far
// taps are substituted by the callers as everywhere else), ZERO takes 0, REPEAT the other end.
// the value lane 'src' holds (src: any lane, per lane)
__device__ __forceinline__ int from_lane(int x, int src)
{
	return __builtin_amdgcn_ds_bpermute(src * 4, x);
}
__device__ __forceinline__ float from_lane(float x, int src)
{
	return __int_as_float(__builtin_amdgcn_ds_bpermute(src * 4, __float_as_int(x)));
}

template <typename V>
struct BorderVals
{
	V prev1, prev2, next1, next2;
};
template <typename V>
__device__ __forceinline__ BorderVals<V> border_values(V a0, V a1, const HEdge& ed)
{
	BorderVals<V> b;
	if (ed.wrap == W_ZERO)
		b.prev1 = b.prev2 = b.next1 = b.next2 = (V)0;
	else if (ed.wrap == W_REPEAT && ed.rep_lanes)  // several tiles side by side: every lane asks its own tile's other end
	{
		b.prev1 = from_lane(a1, ed.lane_last), b.prev2 = from_lane(a0, ed.lane_last);
		b.next1 = from_lane(a0, ed.lane_first), b.next2 = from_lane(a1, ed.lane_first);
	}
	else if (ed.wrap == W_REPEAT)
	{
		b.prev1 = read_lane(a1, ed.lane_last), b.prev2 = read_lane(a0, ed.lane_last);
		b.next1 = read_lane(a0, ed.lane_first), b.next2 = read_lane(a1, ed.lane_first);
	}
	else
		b.prev1 = b.prev2 = a0, b.next1 = b.next2 = a1;  // only read in the first / last lane, where they are the border columns
	return b;
}

// ---- left / right tile border without control flow (CLAMP and ZERO) -----------------------------------------------
// fix_halo_lanes() (ako_stream.hip.h) overwrites the lanes BEYOND a border with the nearest in-range value (or zero), per
// sequence and side, behind a branch: eight row lifts per slot cut the slot into a hundred basic blocks, and a strip at a tile
// border ran 2.4-3 x the instructions of an interior one.  Its waves -- and, through the lockstep barriers, their whole
// workgroup -- were the tail of every launch.  The lanes beyond a border are only ever read by ONE lane, the first / last one
// inside (DD13/7 taps reach two columns = one lane), so here that lane's taps are substituted instead and the lanes beyond
// compute garbage that nobody reads or stores: three selects per side and row lift on per-lane masks that are simply empty in
// a strip without that border.  CLAMP: E[-1] := E[0]; HP[-1] = HP[-2] := HP[0]; E[T] = E[T+1] := E[T-1]; HP[T] := HP[T-1]
// (wavelet-dd137.c:76-79,110-125), ZERO: zeros.  REPEAT strips have no such lanes (they wrap their load addresses,
// lane_columns()) and run the bodies without border code; MIRROR stays on the general kernels.
struct HEdgeBF
{
	bool first, last;  // this lane holds columns 0,1 / T-2,T-1 of a tile whose border rule is CLAMP or ZERO
	bool zero;         // W_ZERO
};
__device__ __forceinline__ HEdgeBF hedge_bf(const HEdge& he)
{
	HEdgeBF e;
	e.first = (he.left || he.nh_left) && he.first, e.last = (he.right || he.nh_right) && he.last;
	e.zero = he.wrap == W_ZERO;
	return e;
}
// A neighbour tap that feeds TWO sums: kept as one DPP move whose result the compiler may not fold away again.  Left alone it
// folds the tap into one of its consumers (a DPP add issues at the rate of a DPP move, half that of a plain add) and still
// needs the move for the other.
template <typename V>
__device__ __forceinline__ V keep_tap(V x)
{
	asm("" : "+v"(x));
	return x;
}
// hlift_inverse() / hlift_forward() of ako_stream.hip.h with that border rule (HB = false: none at all)
template <int KIND, bool HB, typename V>
__device__ __forceinline__ void hlift_inverse_bf(V L0, V L1, V H0, V H1, const HEdgeBF& e, V& E0, V& O0, V& E1, V& O1)
{
	static_assert(KIND != K_HAAR, "lifting wavelets");
	constexpr bool NRW = std::is_same<V, int>::value;  // the integer pipe wraps to int16 after every step like the reference
	V hL1 = keep_tap(from_prev_lane(H1));  // (two sums)
	V hL0 = (V)0, hR0 = (V)0;
	if constexpr (KIND == K_DD137)
		hL0 = from_prev_lane(H0), hR0 = from_next_lane(H0);
	if constexpr (HB)
	{
		const V lo = e.zero ? (V)0 : H0, hi = e.zero ? (V)0 : H1;
		hL1 = e.first ? lo : hL1;
		if constexpr (KIND == K_DD137)
			hL0 = e.first ? lo : hL0, hR0 = e.last ? hi : hR0;
	}
	E0 = lift_add<NRW>(L0, sum_u<KIND, -1>(hL0, hL1, H0, H1), shift_u<KIND>());
	E1 = lift_add<NRW>(L1, sum_u<KIND, -1>(hL1, H0, H1, hR0), shift_u<KIND>());
	V eR0 = keep_tap(from_next_lane(E0));  // (two sums)
	V eL = (V)0, eR1 = (V)0;
	if constexpr (KIND == K_DD137)
		eL = from_prev_lane(E1), eR1 = from_next_lane(E1);
	if constexpr (HB)
	{
		const V lo = e.zero ? (V)0 : E0, hi = e.zero ? (V)0 : E1;
		eR0 = e.last ? hi : eR0;
		if constexpr (KIND == K_DD137)
			eL = e.first ? lo : eL, eR1 = e.last ? hi : eR1;
	}
	O0 = lift_add<NRW>(H0, sum_p<KIND, -1>(eL, E0, E1, eR0), shift_p<KIND>());
	O1 = lift_add<NRW>(H1, sum_p<KIND, -1>(E0, E1, eR0, eR1), shift_p<KIND>());
}
template <int KIND, bool HB, typename V, bool NARROW = false>
__device__ __forceinline__ void hlift_forward_bf(V E0, V O0, V E1, V O1, const HEdgeBF& e, V& L0, V& L1, V& H0, V& H1)
{
	static_assert(KIND != K_HAAR, "lifting wavelets");
	V eR0 = keep_tap(from_next_lane(E0));  // (two sums)
	V eL = (V)0, eR1 = (V)0;
	if constexpr (KIND == K_DD137)
		eL = from_prev_lane(E1), eR1 = from_next_lane(E1);
	if constexpr (HB)
	{
		const V lo = e.zero ? (V)0 : E0, hi = e.zero ? (V)0 : E1;
		eR0 = e.last ? hi : eR0;
		if constexpr (KIND == K_DD137)
			eL = e.first ? lo : eL, eR1 = e.last ? hi : eR1;
	}
	H0 = lift_add<NARROW>(O0, sum_p<KIND, +1>(eL, E0, E1, eR0), shift_p<KIND>());
	H1 = lift_add<NARROW>(O1, sum_p<KIND, +1>(E0, E1, eR0, eR1), shift_p<KIND>());
	V hL1 = keep_tap(from_prev_lane(H1));  // (two sums)
	V hL0 = (V)0, hR0 = (V)0;
	if constexpr (KIND == K_DD137)
		hL0 = from_prev_lane(H0), hR0 = from_next_lane(H0);
	if constexpr (HB)
	{
		const V lo = e.zero ? (V)0 : H0, hi = e.zero ? (V)0 : H1;
		hL1 = e.first ? lo : hL1;
		if constexpr (KIND == K_DD137)
			hL0 = e.first ? lo : hL0, hR0 = e.last ? hi : hR0;
	}
	L0 = lift_add<NARROW>(E0, sum_u<KIND, +1>(hL0, hL1, H0, H1), shift_u<KIND>());
	L1 = lift_add<NARROW>(E1, sum_u<KIND, +1>(hL1, H0, H1, hR0), shift_u<KIND>());
}

// Horizontal forward lift of one row: samples (E0 O0 E1 O1) of this lane's two coefficient
// columns -> (L0 L1 H0 H1).  Valid in lanes 2..61.
template <int KIND, bool NARROW, bool HEDGE, typename V>
__device__ __forceinline__ void hlift_forward(V E0, V O0, V E1, V O1, const HEdge& ed, V& L0, V& L1, V& H0, V& H1)
{
	if constexpr (KIND == K_HAAR)
	{
		L0 = E0, L1 = E1;
		H0 = nrw<true>(O0 - E0), H1 = nrw<true>(O1 - E1);
		return;
	}
	if constexpr (HEDGE && AKO_PERM == 0 && KIND != K_HAAR)
		if (ed.bf)  // (wave-uniform) the common border: three selects per side instead of the general rule's branches
		{
			hlift_forward_bf<KIND, true, V, NARROW>(E0, O0, E1, O1, hedge_bf(ed), L0, L1, H0, H1);
			return;
		}
	auto LP = [&](V x) { if constexpr (AKO_PERM != 0) return perm_lane(ed.perm_prev, x); else return from_prev_lane(x); };
	auto LN = [&](V x) { if constexpr (AKO_PERM != 0) return perm_lane(ed.perm_next, x); else return from_next_lane(x); };
	if (HEDGE)
		fix_halo_lanes(E0, E1, ed);

	V eR0 = LN(E0);
	V eL = (V)0, eR1 = (V)0;
	if (KIND == K_DD137)
		eL = LP(E1), eR1 = LN(E1);
	if (HEDGE && (ed.nh_left || ed.nh_right))
	{
		const BorderVals<V> b = border_values(E0, E1, ed);
		if (ed.nh_left && ed.first)
			eL = b.prev1;
		if (ed.nh_right && ed.last)
			eR0 = b.next1, eR1 = b.next2;
	}
	V p2_0 = eR0, p2_1 = eR1;
	if (HEDGE && KIND == K_DD137 && ed.wrap == W_MIRROR && ed.last)
		p2_0 = eL, p2_1 = E0;  // far tap := opposite near tap
	H0 = lift_add<NARROW>(O0, sum_p<KIND, +1>(eL, E0, E1, p2_0), shift_p<KIND>());
	H1 = lift_add<NARROW>(O1, sum_p<KIND, +1>(E0, E1, eR0, p2_1), shift_p<KIND>());

	if (HEDGE)
		fix_halo_lanes(H0, H1, ed);

	V hL1 = LP(H1);
	V hL0 = (V)0, hR0 = (V)0;
	if (KIND == K_DD137)
		hL0 = LP(H0), hR0 = LN(H0);
	if (HEDGE && (ed.nh_left || ed.nh_right))
	{
		const BorderVals<V> b = border_values(H0, H1, ed);
		if (ed.nh_left && ed.first)
			hL1 = b.prev1, hL0 = b.prev2;
		if (ed.nh_right && ed.last)
			hR0 = b.next1;
	}
	V l2_0 = hL0, l2_1 = hL1;
	if (HEDGE && KIND == K_DD137 && ed.wrap == W_MIRROR && ed.first)
		l2_0 = H1, l2_1 = hR0;
	L0 = lift_add<NARROW>(E0, sum_u<KIND, +1>(l2_0, hL1, H0, H1), shift_u<KIND>());
	L1 = lift_add<NARROW>(E1, sum_u<KIND, +1>(l2_1, H0, H1, hR0), shift_u<KIND>());
}

// Horizontal inverse lift of one row: (L0 L1 H0 H1) -> samples (E0 O0 E1 O1).  Valid in lanes 2..61.
template <int KIND, bool HEDGE, typename V>
__device__ __forceinline__ void hlift_inverse(V L0, V L1, V H0, V H1, const HEdge& ed, V& E0, V& O0, V& E1, V& O1)
{
	constexpr bool NRW = std::is_same<V, int>::value;  // the float pipeline never wraps (see OPT below)
	if constexpr (KIND == K_HAAR)
	{
		E0 = L0, E1 = L1;
		O0 = lift_add<NRW>(L0, H0, 0), O1 = lift_add<NRW>(L1, H1, 0);
		return;
	}
	if constexpr (HEDGE && AKO_PERM == 0 && KIND != K_HAAR)
		if (ed.bf)
		{
			hlift_inverse_bf<KIND, true, V>(L0, L1, H0, H1, hedge_bf(ed), E0, O0, E1, O1);
			return;
		}
	auto LP = [&](V x) { if constexpr (AKO_PERM != 0) return perm_lane(ed.perm_prev, x); else return from_prev_lane(x); };
	auto LN = [&](V x) { if constexpr (AKO_PERM != 0) return perm_lane(ed.perm_next, x); else return from_next_lane(x); };
	if (HEDGE)
		fix_halo_lanes(H0, H1, ed);

	V hL1 = LP(H1);
	V hL0 = (V)0, hR0 = (V)0;
	if (KIND == K_DD137)
		hL0 = LP(H0), hR0 = LN(H0);
	if (HEDGE && (ed.nh_left || ed.nh_right))
	{
		const BorderVals<V> b = border_values(H0, H1, ed);
		if (ed.nh_left && ed.first)
			hL1 = b.prev1, hL0 = b.prev2;
		if (ed.nh_right && ed.last)
			hR0 = b.next1;
	}
	V l2_0 = hL0, l2_1 = hL1;
	if (HEDGE && KIND == K_DD137 && ed.wrap == W_MIRROR && ed.first)
		l2_0 = H1, l2_1 = hR0;
	E0 = lift_add<NRW>(L0, sum_u<KIND, -1>(l2_0, hL1, H0, H1), shift_u<KIND>());
	E1 = lift_add<NRW>(L1, sum_u<KIND, -1>(l2_1, H0, H1, hR0), shift_u<KIND>());

	if (HEDGE)
		fix_halo_lanes(E0, E1, ed);

	V eR0 = LN(E0);
	V eL = (V)0, eR1 = (V)0;
	if (KIND == K_DD137)
		eL = LP(E1), eR1 = LN(E1);
	if (HEDGE && (ed.nh_left || ed.nh_right))
	{
		const BorderVals<V> b = border_values(E0, E1, ed);
		if (ed.nh_left && ed.first)
			eL = b.prev1;
		if (ed.nh_right && ed.last)
			eR0 = b.next1, eR1 = b.next2;
	}
	V p2_0 = eR0, p2_1 = eR1;
	if (HEDGE && KIND == K_DD137 && ed.wrap == W_MIRROR && ed.last)
		p2_0 = eL, p2_1 = E0;
	O0 = lift_add<NRW>(H0, sum_p<KIND, -1>(eL, E0, E1, p2_0), shift_p<KIND>());
	O1 = lift_add<NRW>(H1, sum_p<KIND, -1>(E0, E1, eR0, p2_1), shift_p<KIND>());
}

// ---- vertical pipelines ---------------------------------------------------------------------
// Rings are indexed with the unroll position K (0..5) of the row loop; all indices are constants.

template <typename V>
struct VFwd  // forward: E[v-3..v-1] in e[], O[v-2..v-1] in o[], HP[v-5..v-3] in h[]
{
	V e[3], o[2], h[3];
};

// Feed row slot v (even value E, odd value O); returns LP[v-3] and HP[v-3].
template <int KIND, bool NARROW, bool VEDGE, int K, typename V>
__device__ __forceinline__ void vstep_forward(VFwd<V>& s, V E, V O, int v, int wrap, int T, V& lp_out, V& hp_out)
{
	// ring slots at unroll position K
	V& eA = s.e[K % 3];        // E[v-3]   (overwritten by E[v] at the end)
	V& eB = s.e[(K + 1) % 3];  // E[v-2]
	V& eC = s.e[(K + 2) % 3];  // E[v-1]
	V& oA = s.o[K % 2];        // O[v-2]   (overwritten by O[v])
	V& hA = s.h[K % 3];        // HP[v-5]  (overwritten by HP[v-2])
	V& hB = s.h[(K + 1) % 3];  // HP[v-4]
	V& hC = s.h[(K + 2) % 3];  // HP[v-3]

	const int u = v - 2, r = v - 3;
	V p2 = E;
	if (VEDGE && KIND == K_DD137 && wrap == W_MIRROR && u + 2 >= T)
		p2 = eA;
	V H = lift_add<NARROW>(oA, sum_p<KIND, +1>(eA, eB, eC, p2), shift_p<KIND>());
	if (VEDGE && KIND != K_HAAR && wrap != W_REPEAT)
	{
		if (u >= T)
			H = (wrap == W_ZERO) ? (V)0 : hC;  // HP[T] := HP[T-1]
		if (u < 0 && wrap == W_ZERO)
			H = (V)0;
		if (u == 0 && wrap != W_ZERO)
			hB = H, hC = H;  // HP[-2] = HP[-1] := HP[0]
	}
	V l2 = hA;
	if (VEDGE && KIND == K_DD137 && wrap == W_MIRROR && r < 2)
		l2 = H;
	lp_out = lift_add<NARROW>(eA, sum_u<KIND, +1>(l2, hB, hC, H), shift_u<KIND>());
	hp_out = hC;
	eA = E, oA = O, hA = H;
}

template <typename V>
struct VInv  // inverse: HP[v-3..v-1] in h[], LP[v-1] in l, E[v-4..v-2] in e[]
{
	V h[3], e[3], l;
};

// Feed quadrant row slot v (low-pass value LP, high-pass value HP); returns the even and the odd
// sample row of slot v-3.
template <int KIND, bool VEDGE, int K, typename V>
__device__ __forceinline__ void vstep_inverse(VInv<V>& s, V LP, V HP, int v, int wrap, int T, V& even_out, V& odd_out)
{
	constexpr bool NRW = std::is_same<V, int>::value;
	V& hA = s.h[K % 3];        // HP[v-3]  (overwritten by HP[v])
	V& hB = s.h[(K + 1) % 3];  // HP[v-2]
	V& hC = s.h[(K + 2) % 3];  // HP[v-1]
	V& eA = s.e[K % 3];        // E[v-4]   (overwritten by E[v-1])
	V& eB = s.e[(K + 1) % 3];  // E[v-3]
	V& eC = s.e[(K + 2) % 3];  // E[v-2]

	const int re = v - 1, ro = v - 3;
	V l2 = hA;
	if (VEDGE && KIND == K_DD137 && wrap == W_MIRROR && re < 2)
		l2 = HP;
	V Ev = lift_add<NRW>(s.l, sum_u<KIND, -1>(l2, hB, hC, HP), shift_u<KIND>());
	if (VEDGE && KIND != K_HAAR && wrap != W_REPEAT)
	{
		if (re >= T)
			Ev = (wrap == W_ZERO) ? (V)0 : eC;  // E[T] := E[T-1]
		if (re < 0 && wrap == W_ZERO)
			Ev = (V)0;
		if (ro == 0)
			eA = (wrap == W_ZERO) ? (V)0 : eB;  // E[-1] := E[0]
	}
	V p2 = Ev;
	if (VEDGE && KIND == K_DD137 && wrap == W_MIRROR && ro + 2 >= T)
		p2 = eA;
	even_out = eB;
	odd_out = lift_add<NRW>(hA, sum_p<KIND, -1>(eA, eB, eC, p2), shift_p<KIND>());
	hA = HP, s.l = LP, eA = Ev;
}

struct StreamGeom
{
	uint32_t strips, segs, seg_rows;
	uint32_t wide;  // one strip without halo lanes covers the whole tile width (121..128 coefficient columns)
	// edge_rows != 0: the first segment and the last two are only edge_rows long, the segments between them seg_rows
	// (the last of those possibly shorter).  The segments that touch the top / bottom border run the register-hungry
	// border bodies (spilling in the 128-VGPR u8 kernels) and sit at the end of the grid: kept short, their waves
	// finish with everybody else's instead of being the tail of the launch.
	uint32_t edge_rows;
	// lockstep bit 0: the waves of a workgroup meet at a barrier every six row slots.  They work on neighbouring strips
	// of the same rows (the u8 pair even on the same pixels): kept together, what one wave brought into L2 is still
	// there when its neighbour asks for it -- left alone they drift apart by more rows than the L2 holds lines for
	// (512 waves per XCD x 2-3 rows in flight x 2 KB ~ the 4 MB of an XCD's L2) and the shared lines come from HBM
	// twice (level 0 forward: 451 -> 360 MB fetched, 0.233 -> 0.217 ms).
	// bit 1 (int16 kernels): units in strip-major order, so that a workgroup is four neighbouring strips of one plane
	// instead of the four planes of one strip.
	uint32_t lockstep;
};

// packs of k tiles (StreamGeom::wide = k > 1: several small tiles side by side in one wave, see lane_columns_pack)
__host__ __device__ __forceinline__ uint32_t packs_of(uint32_t n_tiles, uint32_t k)
{
	return (n_tiles + k - 1) / k;
}

// rows [r_lo, r_hi) of a segment and its nominal length
__host__ __device__ __forceinline__ void segment_rows(const StreamGeom& G, uint32_t seg, int Tr, int& r_lo, int& r_hi, int& len)
{
	if (G.edge_rows == 0)
	{
		r_lo = (int)seg * (int)G.seg_rows, len = (int)G.seg_rows;
		r_hi = (r_lo + len < Tr) ? r_lo + len : Tr;
		return;
	}
	const int E = (int)G.edge_rows, mid = (int)G.segs - 3;  // segments 1 .. mid are the long ones
	if (seg == 0)
		r_lo = 0, len = E, r_hi = E;
	else if ((int)seg <= mid)
	{
		r_lo = E + ((int)seg - 1) * (int)G.seg_rows, len = (int)G.seg_rows;
		r_hi = (r_lo + len < Tr - 2 * E) ? r_lo + len : Tr - 2 * E;
	}
	else
		r_lo = Tr - ((int)G.segs - (int)seg) * E, len = E, r_hi = r_lo + E;
}

// unit -> (strip, segment, plane group, tile instance)
struct UnitId
{
	uint32_t strip, seg, pg, tile, image;
	bool valid;
#if defined(AKO_STAMPS) && AKO_STAMPS == 2
	unsigned long long t_top, t_dec, t_lc;  // (measurement builds: s_memtime at the kernel's first instruction, after decode_unit(), after lane_columns())
#endif
};

template <typename T>
__device__ __forceinline__ void split_unit(T u, uint32_t blk, const LevelParams& P, const StreamGeom& G, UnitId& id)
{
	if ((G.lockstep & 2) && P.plane_groups != 2)
	{
		id.strip = (uint32_t)(u % (T)G.strips);
		u /= (T)G.strips;
		id.pg = (uint32_t)(u % (T)P.plane_groups);
		u /= (T)P.plane_groups;
	}
	else
	{
		id.pg = (uint32_t)(u % (T)P.plane_groups);
		// the two waves of a u8 pair do unequal work (forward: Y + Cg against Co + alpha) and wave w of a
		// workgroup runs on SIMD w % 4: swap the roles in every other workgroup (by bit parity, which does not
		// correlate with any round-robin placement) so that every SIMD gets both kinds
		if (P.plane_groups == 2)
			id.pg ^= (uint32_t)(__builtin_popcount(blk) & 1);
		u /= (T)P.plane_groups;
		id.strip = (uint32_t)(u % (T)G.strips);
		u /= (T)G.strips;
	}
	id.seg = (uint32_t)(u % (T)G.segs);
	u /= (T)G.segs;
	if (G.wide > 1)  // packed small tiles / row strips: the unit is a pack of consecutive tiles of the group / a row of them
	{
		const uint32_t k = (G.wide >> 31) ? (G.wide & 0xFFFFu) : G.wide;
		const T packs = (T)packs_of(P.n_tiles, k);
		id.tile = (uint32_t)(u % packs) * k;
		id.image = (uint32_t)(u / packs);
		return;
	}
	id.tile = (uint32_t)(u % (T)P.n_tiles);
	id.image = (uint32_t)(u / (T)P.n_tiles);
}

__device__ __forceinline__ UnitId decode_unit(const LevelParams& P, const StreamGeom& G)
{
	UnitId id;
	// the wave index is wave-uniform, but the compiler only knows that after a readfirstlane: with it,
	// strip / segment / plane and every base address derived from them live in SGPRs
	const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	// XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so
	// physical block b runs on XCD b % 8.  Give every XCD a CONTIGUOUS range of logical blocks: then
	// neighbouring strips / row segments -- which share halo reads and the partially written cache
	// lines at strip borders -- meet in one L2.  (Placement only affects speed, never results.)
	uint32_t blk = blockIdx.x;
	const uint32_t per_xcd = gridDim.x >> 3;
	if (!(P.dbg & 4) && blk < (per_xcd << 3))
		blk = (blk & 7) * per_xcd + (blk >> 3);
	const uint64_t u = (uint64_t)blk * (blockDim.x >> 6) + wave;
	const uint64_t total = (uint64_t)G.strips * G.segs * P.plane_groups * (G.wide > 1 ? packs_of(P.n_tiles, (G.wide >> 31) ? (G.wide & 0xFFFFu) : G.wide) : P.n_tiles) * P.batch;
	id.valid = u < total;
	// 64-bit divisions are loops of several hundred scalar instructions each on this target -- a few microseconds of
	// every wave's start, which the small levels (a wave there does ~1500 instructions of real work) feel; unit
	// counts practically always fit 32 bits, where a division is ~25 instructions
	if (total <= 0xFFFFFFFFull)
		split_unit<uint32_t>((uint32_t)u, blk, P, G, id);
	else
		split_unit<uint64_t>(u, blk, P, G, id);
	return id;
}

__device__ __forceinline__ uint32_t pack2(int lo, int hi)
{
	return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16);
}
// same for two values known to lie in the int16 range: one v_cvt_pk_i16_i32 (it saturates, which
// never triggers here) instead of and + shift-or
__device__ __forceinline__ uint32_t pack2_inrange(int lo, int hi)
{
	typedef short short2v __attribute__((ext_vector_type(2)));
	const short2v v = __builtin_amdgcn_cvt_pk_i16(lo, hi);
	return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ int lo16(uint32_t w)
{
	return (int)(int16_t)(w & 0xFFFFu);
}
__device__ __forceinline__ int hi16(uint32_t w)
{
	return (int)w >> 16;
}

// per lane column geometry shared by both directions
struct LaneCols
{
	int c0;   // first of the lane's two coefficient columns
	int xs;   // first of the lane's four samples (clamped / wrapped so that the access is in range)
	int cs;   // first of the lane's two coefficient columns, clamped / wrapped likewise
	HEdge he;
	bool hedge;
	bool rot;  // odd level width: the four samples were fetched one sample early (see lane_columns)
	bool net;  // this lane's columns belong to the strip's net range (it stores them)
	int tile_in_pack;  // packed small tiles (lane_columns_pack): which of the wave's tiles this lane works on
};

// An ODD number of coefficient columns (level width = 2 mod 4) would leave the last column alone in the
// first slot of a lane.  Instead the LAST strip starts one column earlier: its lanes then hold (odd, even)
// pairs, the last two columns share a lane exactly as with an even count, and all the border logic
// applies unchanged.  The strip re-computes the last net column of its left neighbour and stores the same
// values over it.  Needs a left neighbour (two strips at least) and a wrap mode that does not pair up
// columns across the border (REPEAT does): see stream_width_ok() in ako_plan.hip.
//
// An odd level WIDTH W = 2 Tc - 1 has a phantom last sample, O[Tc-1] := E[Tc-1] (wavelet-dd137.c:128-132).  The
// lane of the last pair (and every lane whose clamped / wrapped pair is the last one) must not read sample W:
// it fetches samples W-4 .. W-1 instead and the callers rotate them into place (E0 O0 E1 | E1); on the way
// back the lane stores three samples instead of four.
//
// A tile of 121..128 coefficient columns would need a second strip for a handful of columns.  Both its borders
// are tile borders, where nothing real lies beyond: the WIDE strip starts at column 0 in lane 0, stores from
// all its in-range lanes, and takes the taps across the borders from border_values() instead of from
// out-of-range lanes (right side: only when the 128 columns leave no lane over).
// first coefficient column of lane 0 of a strip
__host__ __device__ __forceinline__ int strip_base_column(uint32_t strip, uint32_t strips, bool wide, int Tc)
{
	return wide ? 0 : ((int)strip * SNET - SORG - (((Tc & 1) && strip + 1 == strips) ? 1 : 0));
}
// does any lane of the strip need the border code (HEDGE variants)?  Shared with the host, which counts the
// interior units of a launch with it.
__host__ __device__ __forceinline__ bool strip_needs_border_code(uint32_t strip, uint32_t strips, bool wide, int Tc, int W, int wrap)
{
	const int c_base = strip_base_column(strip, strips, wide, Tc);
	const bool left = (wrap != W_REPEAT) && (c_base < 0), right = (wrap != W_REPEAT) && (c_base + 128 > Tc);
	const bool phantom = (W & 1) != 0;
	return wide || left || right || (phantom && ((c_base + 128 >= Tc) || (wrap == W_REPEAT && c_base < 0)));
}
// does a row segment touch the top / bottom border (VEDGE variants)?  (the unrolled row loop may run up to 5 slots
// past the segment and prefetches 2 further)
__host__ __device__ __forceinline__ bool segment_needs_border_code(const StreamGeom& G, uint32_t seg, int Tr)
{
	int r_lo, r_hi, len;
	segment_rows(G, seg, Tr, r_lo, r_hi, len);
	return (r_lo < 3) || (r_lo + len + 12 > Tr);
}

__device__ __forceinline__ LaneCols lane_columns(uint32_t strip, uint32_t strips, bool wide, int lane, int Tc, int W, int wrap)
{
	LaneCols lc;
	const int c_base = strip_base_column(strip, strips, wide, Tc);
	lc.c0 = c_base + 2 * lane;
	lc.tile_in_pack = 0;
	lc.net = wide || ((lane >= 2) && (lane < 62));
	lc.he.wrap = wrap;
	lc.he.perm_prev = ((lane + 63) & 63) * 4, lc.he.perm_next = ((lane + 1) & 63) * 4;
	lc.he.left = (wrap != W_REPEAT) && (c_base < 0);
	lc.he.right = (wrap != W_REPEAT) && (c_base + 128 > Tc);
	lc.he.nh_left = wide;
	// (REPEAT needs real values from the other end, and a single out-of-range lane could not compute them:
	// its own far taps would lie beyond the wave)
	lc.he.nh_right = wide && (Tc == 128 || wrap == W_REPEAT);
	// Round 4: an ordinary strip at a tile border takes the border rule the way a wide strip does -- the taps of the first /
	// last lane INSIDE the tile are substituted (border_values), the lanes beyond it compute garbage that nobody reads (a
	// DD13/7 tap reaches one lane) -- instead of patching those lanes per sequence with v_readlane + selects behind a branch
	// per side (fix_halo_lanes): that made a border strip 2.4-3 x as expensive as an interior one, and its waves the tail of
	// every launch (profiles/r4_border_rule_cost.txt).  Even numbers of coefficient columns without a phantom sample only:
	// the odd geometries move values between the slots of the border lanes and keep the old scheme.
	if (!wide && wrap != W_REPEAT && (W & 3) == 0)
	{
		lc.he.nh_left = lc.he.left, lc.he.nh_right = lc.he.right;
		lc.he.left = lc.he.right = false;
	}
	lc.he.oob_l = lc.c0 < 0;
	lc.he.oob_r = lc.c0 >= Tc;
	lc.he.lane_first = wide ? 0 : SORG / 2;
	lc.he.lane_last = (Tc - 2 - c_base) / 2;
	lc.he.rep_lanes = false;
	lc.he.first = (lc.c0 == 0);
	lc.he.last = (lc.c0 == Tc - 2);
	// Odd Tc, in a strip that is not the (shifted) last one: the lane at c0 == Tc - 1 would straddle the border.
	// Its loads are clamped to the pair (Tc - 2, Tc - 1); the callers then move the second value into the first
	// slot.  The lane's second slot (column Tc) is never needed: stream_width_ok() only admits widths whose
	// last strip holds three columns or more, which keeps this strip's net outputs clear of the border.
	lc.he.half = (lc.c0 == Tc - 1);
	const bool phantom = (W & 1) != 0;
	lc.he.drop_last = phantom && lc.he.last;
	// (a half lane only exists where he.right is set; the phantom fix-ups also live in the border variants)
	lc.hedge = strip_needs_border_code(strip, strips, wide, Tc, W, wrap);  // (phantom: some lane maps to the last pair)
	if (wrap == W_REPEAT)
		lc.cs = max(map_index(lc.c0, Tc, W_REPEAT) & ~1, 0);  // pairs stay together: c0 and Tc are even
	else
		lc.cs = min(max(lc.c0, 0), Tc - 2);
	lc.rot = phantom && (lc.cs == Tc - 2);
	lc.xs = lc.rot ? (W - 4) : (2 * lc.cs);
	lc.he.bf = (wrap == W_CLAMP || wrap == W_ZERO) && !lc.he.left && !lc.he.right && !phantom && (Tc & 1) == 0;
	return lc;
}

// SEVERAL SMALL TILES IN ONE WAVE (round 3).  A level of 8, 16, 32 or 64 coefficient columns would leave most of a
// wave's 128 columns idle (16384 x 16384 in 256-pixel tiles: levels 1, 2, 3 use 50, 25, 12.5 % of their lanes).  Such a
// level is a "wide" strip in miniature -- both its borders are tile borders, where nothing real lies beyond -- so k = 128 /
// Tc tiles of one group sit side by side in one wave (StreamGeom::wide = k): lane l works on tile l / (Tc / 2), columns
// 2 * (l % (Tc / 2)), +1; the taps across EVERY tile border come from border_values() in the tile's first / last lane,
// exactly as a wide strip takes them (CLAMP, MIRROR, ZERO: per-lane values; REPEAT would need the other end of the
// tile and stays unpacked); tile origin, stream offset and lift head become per-lane values.  int16 levels only (the
// source of a packed level is the scratch plane of each tile instance, not the image).
// StreamGeom::wide: 0 ordinary strips; 1 one wide strip; 2..64 that many small tiles per wave (int16 levels); bit 31 set:
// ROW STRIPS over the tiles of a tile row, low 16 bits = tiles per row (u8 level 0, below)
__host__ __device__ __forceinline__ bool geom_wide(const StreamGeom& G)
{
	return G.wide == 1u;
}
__host__ __device__ __forceinline__ uint32_t geom_pack(const StreamGeom& G)
{
	return (G.wide > 1u && !(G.wide >> 31)) ? G.wide : 0u;
}
__host__ __device__ __forceinline__ uint32_t geom_row_tiles(const StreamGeom& G)
{
	return (G.wide >> 31) ? (G.wide & 0xFFFFu) : 0u;
}

// ROW STRIPS (round 3): level 0 of a u8 image in 512-pixel tiles has 256 coefficient columns per tile -- three strips of
// 120, the third one for 16 columns.  The tiles of a tile row lie side by side in the image, so the strips are laid over
// the whole ROW of tiles instead (ntx * Tc columns, 120 net each: 69 strips for 32 tiles instead of 96): a lane's pixels
// are just the image columns it covers, its tile is (global column) / Tc, and a tile border that falls inside a strip is
// handled where it falls, through border_values() in the tile's first / last lane, as in a wide strip or a pack of small
// tiles.  Stream offset, low-pass plane and lift head are per-lane values.  Not for REPEAT.
__device__ __forceinline__ LaneCols lane_columns_row(uint32_t strip, int lane, int Tc, int ntx, int wrap)
{
	LaneCols lc;
	const int total = ntx * Tc;
	const int gcol = (int)strip * SNET - SORG + 2 * lane;  // column inside the row of tiles (even)
	const int gin = min(max(gcol, 0), total - 2);           // ... clamped into it for the loads of the halo lanes at the two ends
	lc.tile_in_pack = gin / Tc;
	lc.c0 = gin - lc.tile_in_pack * Tc;
	lc.net = (lane >= 2) && (lane < 62) && (gcol >= 0) && (gcol < total);
	lc.he.wrap = wrap;
	lc.he.left = lc.he.right = false;
	lc.he.nh_left = lc.he.nh_right = true;
	lc.he.oob_l = lc.he.oob_r = false;
	lc.he.lane_first = 0, lc.he.lane_last = 0;
	lc.he.rep_lanes = false;
	lc.he.perm_prev = ((lane + 63) & 63) * 4, lc.he.perm_next = ((lane + 1) & 63) * 4;
	lc.he.first = (lc.c0 == 0);
	lc.he.last = (lc.c0 == Tc - 2);
	lc.he.half = false, lc.he.drop_last = false;
	lc.hedge = true;
	lc.cs = lc.c0;
	lc.rot = false;
	lc.xs = 2 * gin;  // the image side: samples counted from the row's first tile
	lc.he.bf = (wrap == W_CLAMP || wrap == W_ZERO);
	return lc;
}

__device__ __forceinline__ LaneCols lane_columns_pack(int lane, int Tc, int wrap, int k_valid)
{
	LaneCols lc;
	const int lpt = Tc >> 1;  // lanes per tile
	lc.tile_in_pack = lane / lpt;
	lc.c0 = 2 * (lane - lc.tile_in_pack * lpt);
	lc.net = lc.tile_in_pack < k_valid;
	lc.he.wrap = wrap;
	lc.he.left = lc.he.right = false;
	lc.he.nh_left = lc.he.nh_right = true;
	lc.he.oob_l = lc.he.oob_r = false;
	// REPEAT: the other end of the lane's own tile (border_values through ds_bpermute)
	lc.he.lane_first = lc.tile_in_pack * lpt, lc.he.lane_last = lc.he.lane_first + lpt - 1;
	lc.he.rep_lanes = (wrap == W_REPEAT);
	lc.he.perm_prev = ((lane + 63) & 63) * 4, lc.he.perm_next = ((lane + 1) & 63) * 4;
	lc.he.first = (lc.c0 == 0);
	lc.he.last = (lc.c0 == Tc - 2);
	lc.he.half = false, lc.he.drop_last = false;
	lc.hedge = true;
	lc.cs = lc.c0;
	lc.rot = false;
	lc.xs = 2 * lc.c0;
	lc.he.bf = (wrap == W_CLAMP || wrap == W_ZERO);
	return lc;
}
// ---- helpers of the two-level workgroup kernels (ako_fused.hip.h) ----
// lane_columns() for an arbitrary first column and net lane range (even Tc, no phantom column)
__device__ __forceinline__ LaneCols lane_columns_at(int c_base, int net_lo, int net_hi, int lane, int Tc, int wrap)
{
	LaneCols lc;
	lc.c0 = c_base + 2 * lane;
	lc.tile_in_pack = 0;
	lc.net = (lane >= net_lo) && (lane < net_hi);
	lc.he.wrap = wrap;
	lc.he.perm_prev = ((lane + 63) & 63) * 4, lc.he.perm_next = ((lane + 1) & 63) * 4;
	lc.he.left = (wrap != W_REPEAT) && (c_base < 0);
	lc.he.right = (wrap != W_REPEAT) && (c_base + 128 > Tc);
	lc.he.nh_left = lc.he.nh_right = false;
	lc.he.oob_l = lc.c0 < 0;
	lc.he.oob_r = lc.c0 >= Tc;
	lc.he.lane_first = (c_base < 0) ? (-c_base) / 2 : 0;
	lc.he.lane_last = (Tc - 2 - c_base) / 2;
	lc.he.rep_lanes = false;
	lc.he.first = (lc.c0 == 0);
	lc.he.last = (lc.c0 == Tc - 2);
	lc.he.half = false, lc.he.drop_last = false;
	lc.hedge = lc.he.left || lc.he.right;
	if (wrap == W_REPEAT)
		lc.cs = max(map_index(lc.c0, Tc, W_REPEAT) & ~1, 0);
	else
		lc.cs = min(max(lc.c0, 0), Tc - 2);
	lc.rot = false;
	lc.xs = 2 * lc.cs;
	lc.he.bf = false;
	return lc;
}
// vstep_forward() with a ring of THREE for the odd samples as well, so that every ring index has period 3: the level-1
// pipeline of the two-level kernels advances every second row slot, three times per trip of the 6-slot unrolled loop.
template <typename V>
struct VFwd3
{
	V e[3], o[3], h[3];
};
template <int KIND, bool VEDGE, int K, typename V>
__device__ __forceinline__ void vstep_forward3(VFwd3<V>& s, V E, V O, int v, int wrap, int T, V& lp_out, V& hp_out)
{
	V& eA = s.e[K % 3];        // E[v-3]   (overwritten by E[v] at the end)
	V& eB = s.e[(K + 1) % 3];  // E[v-2]
	V& eC = s.e[(K + 2) % 3];  // E[v-1]
	V& oR = s.o[(K + 1) % 3];  // O[v-2]
	V& oW = s.o[K % 3];        // O[v-3]   (dead: overwritten by O[v])
	V& hA = s.h[K % 3];        // HP[v-5]  (overwritten by HP[v-2])
	V& hB = s.h[(K + 1) % 3];  // HP[v-4]
	V& hC = s.h[(K + 2) % 3];  // HP[v-3]
	const int u = v - 2, r = v - 3;
	V p2 = E;
	if (VEDGE && KIND == K_DD137 && wrap == W_MIRROR && u + 2 >= T)
		p2 = eA;
	V H = lift_add<false>(oR, sum_p<KIND, +1>(eA, eB, eC, p2), shift_p<KIND>());
	if (VEDGE && KIND != K_HAAR && wrap != W_REPEAT)
	{
		if (u >= T)
			H = (wrap == W_ZERO) ? (V)0 : hC;
		if (u < 0 && wrap == W_ZERO)
			H = 0;
		if (u == 0 && wrap != W_ZERO)
			hB = H, hC = H;
	}
	V l2 = hA;
	if (VEDGE && KIND == K_DD137 && wrap == W_MIRROR && r < 2)
		l2 = H;
	lp_out = lift_add<false>(eA, sum_u<KIND, +1>(l2, hB, hC, H), shift_u<KIND>());
	hp_out = hC;
	eA = E, oW = O, hA = H;
}

// forward colour transform of one pixel (format.c:87-134)
__device__ __forceinline__ void color_forward(int color, int r, int g, int b, int& c0, int& c1, int& c2)
{
	c0 = r, c1 = g, c2 = b;
	if (color == C_SUBG)
		c0 = g, c1 = r - g, c2 = b - g;
	else if (color != C_NONE)
	{
		const int co = r - b;
		const int t = b + tdiv(co, 1);
		const int cg = g - t;
		const int yy = t + tdiv(cg, 1);
		c0 = (color == C_YCOCG_Q) ? yy * 2 : yy;
		c1 = co, c2 = cg;
	}
}

// the two planes a wave of a pair needs from one pixel: pair 0 -> planes 0,1; pair 1 -> planes 2,3
__device__ __forceinline__ void color_forward_pair(int color, int pair, int r, int g, int b, int a, int& v0,
                                                   int& v1)
{
	int c0, c1, c2;
	color_forward(color, r, g, b, c0, c1, c2);
	v0 = pair ? c2 : c0;
	v1 = pair ? a : c1;
}

// C's x / 2 (truncating) on either pipe
__device__ __forceinline__ int half_trunc(int x)
{
	return tdiv(x, 1);
}
__device__ __forceinline__ float half_trunc(float x)
{
	return __builtin_truncf(x * 0.5f);
}

// Four RGBA pixels -> the two planes of this wave's pair, the colour mode switch hoisted out of the
// pixel loop (one wave-uniform branch per row instead of three per pixel).  V = int or float.
// Forward pairs are (plane 0, plane 2) and (plane 1, plane 3): with YCoCg the first wave then owns
// Y and Cg (which share t = b + Co/2) and the second only Co = r - b and alpha, so nothing is computed
// twice and neither wave converts a channel it does not use.
template <typename V>
__device__ __forceinline__ void decode_pixels_pair(const uint32_t px[4], int color, int pair, bool discard, V v0[4],
                                                   V v1[4])
{
	V r[4], g[4], b[4], a[4];
#pragma unroll
	for (int k = 0; k < 4; k++)
	{
		r[k] = (V)(px[k] & 255), g[k] = (V)((px[k] >> 8) & 255);  // float: v_cvt_f32_ubyte0..3
		b[k] = (V)((px[k] >> 16) & 255), a[k] = (V)(px[k] >> 24);
	}
	if (discard)  // wave-uniform; format.c:38-49
	{
#pragma unroll
		for (int k = 0; k < 4; k++)
			if (a[k] == (V)0)
				r[k] = g[k] = b[k] = (V)0;
	}
	if (color == C_YCOCG || color == C_YCOCG_Q)
	{
		if (pair == 0)  // wave-uniform
		{
			const V ymul = (color == C_YCOCG_Q) ? (V)2 : (V)1;
#pragma unroll
			for (int k = 0; k < 4; k++)
			{
				const V co = r[k] - b[k];
				const V t = b[k] + half_trunc(co);
				const V cg = g[k] - t;
				v0[k] = (t + half_trunc(cg)) * ymul, v1[k] = cg;
			}
		}
		else
		{
#pragma unroll
			for (int k = 0; k < 4; k++)
				v0[k] = r[k] - b[k], v1[k] = a[k];
		}
	}
	else if (color == C_SUBG)
	{
#pragma unroll
		for (int k = 0; k < 4; k++)
		{
			v0[k] = pair ? (r[k] - g[k]) : g[k];
			v1[k] = pair ? a[k] : (b[k] - g[k]);
		}
	}
	else
	{
#pragma unroll
		for (int k = 0; k < 4; k++)
		{
			v0[k] = pair ? g[k] : r[k];
			v1[k] = pair ? a[k] : b[k];
		}
	}
}

// ---- 16-bit packing on the float pipe through SDWA conversions ------------------------------------
// Two / four integer-valued floats inside the int16 range -> packed int16 pairs: v_cvt_i32_f32 writing straight
// into the low / high word of the destination (dst_sel), instead of two full conversions + v_cvt_pk_i16_i32.
// gfx940+ need one wait state between an instruction with dst_sel != DWORD and a VALU instruction that reads
// its destination (here: the second conversion into the same register, which preserves the word the first one
// wrote); compilers insert it for their own code, not inside inline assembly, so the conversions of TWO words
// are interleaved and a trailing s_nop covers whatever VALU instruction comes next.
__device__ __forceinline__ void pack2x2_f(float a_lo, float a_hi, float b_lo, float b_hi, uint32_t& a, uint32_t& b)
{
	asm("v_cvt_i32_f32_sdwa %0, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
	    "v_cvt_i32_f32_sdwa %1, %4 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
	    "v_cvt_i32_f32_sdwa %0, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
	    "v_cvt_i32_f32_sdwa %1, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
	    "s_nop 0"
	    : "=&v"(a), "=&v"(b)
	    : "v"(a_lo), "v"(a_hi), "v"(b_lo), "v"(b_hi));
}
// a packed int16 pair -> two floats: sign-extending word selects on the source of the conversion
__device__ __forceinline__ void unpack2_f(uint32_t w, float& lo, float& hi)
{
	asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(lo) : "v"(w));
	asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(hi) : "v"(w));
}

// Gate as a factor on the float pipe: for integer-valued f and integer g, clamp(|f| - g, 0, 1) is exactly 1 where
// |f| > g and exactly 0 elsewhere (one v_sub_f32 with the abs and clamp modifiers), so the gated, scaled value is
// f * rq * that -- two plain multiplies instead of a compare and a select around the conversion.
__device__ __forceinline__ float gate_scale_f(float f, float gf, float rq)
{
	const float open = __builtin_fminf(__builtin_fmaxf(__builtin_fabsf(f) - gf, 0.0f), 1.0f);
	return (f * rq) * open;
}

// the row of one plane: LL pair + the three gated / quantized high-pass pairs -> four packed words
__device__ __forceinline__ void pack_row_f(const float lp[4], const float hp[4], float gf, float rq, uint32_t& w_ll,
                                           uint32_t& w_c, uint32_t& w_b, uint32_t& w_d)
{
	pack2x2_f(lp[0], lp[1], gate_scale_f(hp[0], gf, rq), gate_scale_f(hp[1], gf, rq), w_ll, w_c);
	pack2x2_f(gate_scale_f(lp[2], gf, rq), gate_scale_f(lp[3], gf, rq), gate_scale_f(hp[2], gf, rq),
	          gate_scale_f(hp[3], gf, rq), w_b, w_d);
}
__device__ __forceinline__ void pack_row_f(const int lp[4], const int hp[4], float gf, float rq, uint32_t& w_ll,
                                           uint32_t& w_c, uint32_t& w_b, uint32_t& w_d)
{
	const float l2 = (float)lp[2], l3 = (float)lp[3];
	const float h0 = (float)hp[0], h1 = (float)hp[1], h2 = (float)hp[2], h3 = (float)hp[3];
	uint32_t unused;
	w_ll = pack2_inrange(lp[0], lp[1]);
	pack2x2_f(gate_scale_f(h0, gf, rq), gate_scale_f(h1, gf, rq), gate_scale_f(l2, gf, rq), gate_scale_f(l3, gf, rq), w_c, w_b);
	pack2x2_f(gate_scale_f(h2, gf, rq), gate_scale_f(h3, gf, rq), 0.0f, 0.0f, w_d, unused);
}

// The common case of the u8 kernels spelled out: YCoCg / YCoCg_Q without the discard-transparent rule.  No colour
// mode dispatch per row, and each wave of the pair converts only the three channels its two planes are made of
// (Y and Cg: r, g, b; Co and alpha: r, b, a).
template <typename V>
__device__ __forceinline__ void decode_pixels_ycocg(const uint32_t px[4], V ymul, int pair, V v0[4], V v1[4])
{
	if (pair == 0)  // wave-uniform
	{
#pragma unroll
		for (int k = 0; k < 4; k++)
		{
			const V r = (V)(px[k] & 255), g = (V)((px[k] >> 8) & 255), b = (V)((px[k] >> 16) & 255);
			const V co = r - b;
			const V t = b + half_trunc(co);
			const V cg = g - t;
			v0[k] = (t + half_trunc(cg)) * ymul, v1[k] = cg;
		}
	}
	else
	{
#pragma unroll
		for (int k = 0; k < 4; k++)
		{
			const V r = (V)(px[k] & 255), b = (V)((px[k] >> 16) & 255);
			v0[k] = r - b, v1[k] = (V)(px[k] >> 24);
		}
	}
}

// gate + quantize on the float pipe (see quantize() in ako_kernels.hip.h for the exactness argument;
// q == 1 needs no special case: trunc(v * fl(1 + 1e-6)) == v for |v| <= 32768)
__device__ __forceinline__ int quantize_f(int v, float gf, float rq)
{
	const float f = (float)v;
	const int r = (int)(f * rq);
	return (fabsf(f) > gf) ? r : 0;
}

__device__ __forceinline__ int quantize_f(float f, float gf, float rq)
{
	const int r = (int)(f * rq);  // v_cvt_i32_f32 truncates toward zero
	return (fabsf(f) > gf) ? r : 0;
}
__device__ __forceinline__ int to_int(int v)
{
	return v;
}
__device__ __forceinline__ int to_int(float v)
{
	return (int)v;
}

// inverse colour transform of one pixel (format.c:138-218), int16 wrap after every step
__device__ __forceinline__ void color_inverse(int color, int v0, int v1, int v2, int& r, int& g, int& b)
{
	r = v0, g = v1, b = v2;
	if (color == C_SUBG)
		r = (int16_t)(v1 + v0), g = v0, b = (int16_t)(v2 + v0);
	else if (color != C_NONE)
	{
		const int yv = (color == C_YCOCG_Q) ? tdiv(v0, 1) : v0;
		const int t = (int16_t)(yv - tdiv(v2, 1));
		g = (int16_t)(v2 + t);
		b = (int16_t)(t - tdiv(v1, 1));
		r = (int16_t)(b + v1);
	}
}

__device__ __forceinline__ uint32_t sat8f(float v)
{
	return (uint32_t)(int)__builtin_fminf(__builtin_fmaxf(v, 0.0f), 255.0f);
}
// four integer-valued floats -> one RGBA pixel, each clamped to 0..255: v_cvt_pk_u8_f32 converts,
// saturates and inserts the byte in one instruction (vs. clamp + convert + shift-or per channel)
__device__ __forceinline__ uint32_t pixel_u8x4(float r, float g, float b, float a)
{
	uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(r, 0, 0);
	w = __builtin_amdgcn_cvt_pk_u8_f32(g, 1, w);
	w = __builtin_amdgcn_cvt_pk_u8_f32(b, 2, w);
	return __builtin_amdgcn_cvt_pk_u8_f32(a, 3, w);
}
// four RGBx pixels (fourth byte zero) -> twelve bytes of RGB
__device__ __forceinline__ auto rgb_pack4(const uint32_t px[4])
{
	typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
	return u32x3{px[0] | (px[1] << 24), (px[1] >> 8) | (px[2] << 16), (px[2] >> 16) | (px[3] << 8)};
}
// acc = max(acc, |a|, |b|) in one instruction (fmaxf() would first canonicalise each operand)
__device__ __forceinline__ void absmax3(float& acc, float a, float b)
{
	asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(acc) : "v"(a), "v"(b));
}
// inverse colour on the float pipe, intermediates untracked (the caller bounds them through its inputs)
__device__ __forceinline__ void color_inverse_fast(int color, float v0, float v1, float v2, float& r, float& g, float& b)
{
	r = v0, g = v1, b = v2;
	if (color == C_SUBG)
		r = v1 + v0, g = v0, b = v2 + v0;
	else if (color != C_NONE)
	{
		const float yv = (color == C_YCOCG_Q) ? half_trunc(v0) : v0;
		const float t = yv - half_trunc(v2);
		g = v2 + t;
		b = t - half_trunc(v1);
		r = b + v1;
	}
}

// ---------------------------------------------------------------------------------------------
// Column groups (round 3): the level-0 stores of big tiles re-shaped into whole cache lines.
//
// The stream format puts a one-value head in front of every plane's sub-bands (misc.c:245-285), so sub-band rows start
// at arbitrary 2-byte offsets and a strip's 240-byte runs cut cache lines at both ends.  The memory system takes a store
// that fills whole 128-byte lines at about twice the rate of one that leaves lines partly written (0.10 against 0.20 ms
// for the 537 MB of the 8192 x 8192 level; a 64-byte shift already costs half of that: scripts/store_align.hip,
// profiles/r3_store_alignment.txt).  So: a workgroup is 8 waves = 4 neighbouring strips x the pair of waves; it OWNS, in
// every row of every sub-band of plane p, the 7 cache lines (448 columns) starting at column S - p, where S is chosen per
// (image, tile) from the actual address of the plane-0 sub-bands; the planes' heads shift their lines by one column each,
// which the 480 net columns of four strips cover (451 needed).  Every row slot each wave leaves its 2 x 4 packed words
// per lane in an LDS row buffer (16 row kinds = 4 planes x LL C B D), the workgroup meets at one barrier (double buffer),
// and wave w then stores row kinds 2 w, 2 w + 1 with one 16-byte-per-lane store each: 56 lanes = 7 whole lines.  Lines cut
// by the tile's left / right border are stored value by value (first and last group only).  The low-pass planes in the
// scratch buffer are shifted by the same phase (plane p by phase + p values), so that their lines coincide with the
// stream's; the level-1 kernel reads them with the same shift (LevelParams::src_tiled).
// Needs: sub-band width a multiple of 64 (every row and all three sub-bands of a plane then share the phase) and an even
// level width.  A group nets 448 of 480 columns, i.e. 7 % more strips than the plain kernel walks.
// ---------------------------------------------------------------------------------------------
constexpr int GRP_STRIPS = 4, GRP_WAVES = 2 * GRP_STRIPS;
constexpr int GRP_OWN = 448;  // columns a workgroup owns per row: 7 lines of int16
constexpr int GRP_PAD = 8;    // values in front of / behind them in a row buffer (pairs cut by the ownership border land there)
constexpr int GRP_ROW = GRP_OWN + 2 * GRP_PAD;
constexpr int GRP_KINDS = 16;  // row kinds: plane * 4 + (0 LL, 1 C, 2 B, 3 D)
constexpr int GRP_LDS_VALUES = 2 * GRP_KINDS * GRP_ROW;

__host__ __device__ __forceinline__ uint32_t group_count(uint32_t Tc)
{
	return (Tc + 67u + (uint32_t)GRP_OWN - 1u) / (uint32_t)GRP_OWN;  // whatever the phase, plane 3 of the last group reaches Tc
}

// position of plane 0's sub-band rows inside a cache line, in values (plane p: + p); konst = grp_off[0] + 1 of level 0
__device__ __forceinline__ uint32_t stream_phase(const LevelParams& P, uint32_t image, const TileDesc& td, uint32_t konst)
{
	return ((uint32_t)(reinterpret_cast<uintptr_t>(P.stream) >> 1) + image * (uint32_t)P.stream_stride + (uint32_t)td.stream_off + konst) & 63u;
}

// A 16-byte-per-lane buffer store followed at once by a VALU write of one of its data registers stores the NEW value in
// part of the lanes on gfx950 (seen on the box: the next instruction's address computation showing up in the stream, four
// lanes in sixteen).  The compiler's hazard table only knows this for stores without a scalar offset; ours always have
// one.  Two wait states behind every such store.
#define AKO_STORE_GUARD()                      \
	do                                         \
	{                                          \
		__builtin_amdgcn_sched_barrier(0);     \
		asm volatile("s_nop 1");               \
		__builtin_amdgcn_sched_barrier(0);     \
	} while (0)
__device__ __forceinline__ void store_b128_guarded(__attribute__((ext_vector_type(4))) uint32_t v, __amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff, int aux = 0)
{
	__builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, 0);
	AKO_STORE_GUARD();
	(void)aux;
}

struct __attribute__((packed, aligned(2))) U32A2
{
	uint32_t v;
};
// A packed pair into a row buffer.  The position is only 2-byte aligned when the plane's phase is odd (wave-uniform), and
// the compiler must then be kept from fusing the halves back into ds_write_b32 / ds_write2_b32, which want a dword
// address (seen on the box: dwords landing 2 bytes off): volatile halves.
template <bool ODD>
__device__ __forceinline__ void grp_put(uint8_t* at, uint32_t w)
{
	typedef __attribute__((address_space(3))) volatile uint16_t lds_vu16;
	typedef __attribute__((address_space(3))) uint32_t lds_u32;
	if constexpr (ODD)
	{
		lds_vu16* h = (lds_vu16*)at;
		h[0] = (uint16_t)w;
		h[1] = (uint16_t)(w >> 16);
	}
	else
		*(lds_u32*)at = w;
}

// Register budget: the u8 kernels live on exactly 128 VGPRs, and a single spilled address register costs a scratch reload
// + s_waitcnt vmcnt(0) per row slot, i.e. the prefetch.  Every per-lane address of the group kernel is therefore derived
// from ONE register, lane16 = 16 * lane: the pixel load offset (16 bytes per lane; interior strips), the position of the
// lane's packed words in a row buffer (4 * lane + a wave constant) and the lane's share of the drain (16 bytes per lane).
struct GroupCtx
{
	int16_t* lds;
	int S;        // first owned column of plane 0
	// this wave's share of the stores: row kinds 2 * wave, 2 * wave + 1 (plane wave / 2: LL and C, or B and D); resources
	// based at the plane's first owned column of row 0 (which lies in front of the sub-band in group 0)
	__amdgpu_buffer_rsrc_t rs[2];
	uint32_t soff[2];     // byte offset of the sub-band
	uint32_t pitch_b[2];  // bytes per row
	uint32_t lds_k;       // byte offset of the first owned column in row kind 2 * wave of buffer 0
	int d_first;          // first owned column of the plane
	bool ragged;          // the owned range crosses column 0 or Tc (first / last group): lanes are checked one by one
	bool exists;          // (RGB: plane 3 does not)
};

// row_ok false: the same instructions with every store out of range (the slots in front of / behind a segment) -- a
// branch around them would make the compiler's count of memory operations in flight inexact, and the wait in front of a
// slot's pixels would then cover the NEXT slot's prefetch as well (stores count in vmcnt on gfx950).
__device__ __forceinline__ void group_drain(const GroupCtx& gc, uint32_t lane16, int buf, int r, bool row_ok, int Tc)
{
	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	constexpr uint32_t OOB = 0xFFFFFFFFu;
	if (!gc.exists)
		return;
	// 56 lanes x 16 bytes = the 7 owned lines; lanes 56..63 double lane 55 (same values, same address: cheaper than a mask)
	const uint32_t da = min(lane16, (uint32_t)(GRP_OWN / 8 - 1) * 16u);
	const uint8_t* rows = reinterpret_cast<const uint8_t*>(gc.lds) + gc.lds_k + (uint32_t)(buf * GRP_KINDS * GRP_ROW * 2);
#pragma unroll
	for (int i = 0; i < 2; i++)
	{
		const uint8_t* row = rows + i * GRP_ROW * 2;
		const u32x4 v = *reinterpret_cast<const u32x4*>(row + da);
		const uint32_t so = row_ok ? gc.soff[i] + (uint32_t)r * gc.pitch_b[i] : OOB;
		if (!gc.ragged)  // wave-uniform
			store_b128_guarded(v, gc.rs[i], da, so);
		else
		{
			// first / last group of a row: whole lanes inside the level as above, the columns of the lanes the border cuts
			// one by one (lanes 0..7: the columns in front of the first whole lane; lanes 8..15: the ones behind the last)
			const int first = gc.d_first;
			const int c = first + (int)(da >> 1);
			store_b128_guarded(v, gc.rs[i], (c >= 0 && c + 8 <= Tc) ? da : OOB, so);
			const int lane = (int)(lane16 >> 4);
			int col = -1;
			if (lane < 8)
			{
				const int cf = (first < 0) ? ((8 - ((-first) & 7)) & 7) : 0;
				if (lane < cf)
					col = lane;
			}
			else if (lane < 16 && first < Tc && first + GRP_OWN > Tc)
			{
				const int ce = Tc - ((Tc - first) & 7);
				if (ce + lane - 8 < Tc)
					col = ce + lane - 8;
			}
			const int at = (col >= 0) ? (col - first) * 2 : 0;
			__builtin_amdgcn_raw_buffer_store_b16(*reinterpret_cast<const int16_t*>(row + at), gc.rs[i], (col >= 0) ? (uint32_t)at : OOB, so, 0);
		}
	}
}

// ---------------------------------------------------------------------------------------------
// Forward.  NPL = planes handled by one wave: 2 with U8 (half of an RGBA pixel), 1 on int16 planes.
// ---------------------------------------------------------------------------------------------

template <bool U8>
struct FwdRaw
{
	using vec = uint4;  // U8: four RGBA pixels
	vec a[2];           // the two rows of a slot
};
template <>
struct FwdRaw<false>
{
	using vec = uint2;  // int16: four samples
	vec a[2];
};

// DEEP = 0: a wave prefetches two row slots ahead of the one it works on (ring of three).  DEEP = N > 0, for
// segments of at most N - 6 rows: ALL N row slots of the segment are fetched before the first one is used.  Small
// levels have too few waves to hide memory latency behind each other, and a wave's row slots are a dependent
// chain: with the running prefetch every slot of such a wave waits for its own round trip to memory, with all
// loads in flight at once the segment costs one round trip.  (A single pass over exactly N slots: the ring indices
// of the column pipeline stay compile-time constants for any N.)
#ifndef AKO_DEEP_FILL_CUT
#define AKO_DEEP_FILL_CUT 1  // experiments: 0 = the fill slots of the deep forward variants run gate, quantizer and (dropped) stores like every slot
#endif
template <int KIND, int NPL, bool U8, bool NARROW, bool HEDGE, bool VEDGE, int DEEP, bool CFAST = false, int PF = 2, int LATE = 0,
          bool MEMONLY = false, int CH = 4, bool GRP = false>
__device__ __forceinline__ void forward_stream_body(const LevelParams& P, const StreamGeom& G, const UnitId& id,
                                                    const LaneCols& lc, int lane, const GroupCtx* gcp = nullptr)
{
	static_assert(CH == 4 || (CH == 3 && U8 && NPL == 2), "CH = 3: the u8 kernels on RGB pixels");
	static_assert(!GRP || (U8 && NPL == 2 && LATE > 0 && !MEMONLY), "column groups: the u8 kernel");
	// CH = 3 (RGB): the pair's second wave owns planes 1 and 3, and plane 3 does not exist: it carries one plane
	const bool one_plane = (CH == 3) && (id.pg == 1);
	const TileDesc td = P.tiles[id.tile];
	const uint64_t inst = (uint64_t)id.image * P.n_tiles + id.tile;
	const int Tc = (int)P.sub_w, Tr = (int)P.sub_h;
	const int chh = (int)P.full_h;
	const int wrap = P.wrap;
	// planes of this wave: pg alone on int16 planes; on the u8 side the pair (pg, pg + 2), see
	// decode_pixels_pair()
	const int p_first = (int)id.pg;
	constexpr int P_STEP = (U8 && NPL == 2) ? 2 : 1;
	const int c0 = lc.c0;

	int r_lo, r_hi, seg_len;
	segment_rows(G, id.seg, Tr, r_lo, r_hi, seg_len);
	(void)seg_len;

	// Packed small tiles (lane_columns_pack, int16 levels only): the wave's resources are based at the IMAGE's first tile
	// instance / at the image's stream, and the lane adds its own tile's instance and stream offset
	const bool pack = U8 ? (geom_row_tiles(G) != 0) : (geom_pack(G) != 0);  // (u8: row strips over a row of tiles)
	const uint32_t lane_tile = pack ? min(id.tile + (uint32_t)lc.tile_in_pack, P.n_tiles - 1u) : 0u;  // per lane
	const uint64_t base_inst = pack ? (uint64_t)id.image * P.n_tiles : inst;                               // wave-uniform
	const uint32_t lane_stream_b = pack ? (uint32_t)(P.tiles[lane_tile].stream_off * 2) : 0u;              // per lane

	// Sources, read through a raw buffer resource like the stores below: the wave-uniform origin of the tile
	// (plane) is the resource's base, the lane's four samples are ONE register of byte offset for every load of the
	// wave, and the row travels as the scalar offset -- no 64-bit vector address arithmetic per row.  (All reads are
	// in range by construction; the plan keeps images and planes of 4 GiB and more away from these kernels.)
	const uint8_t* src_base;
	uint32_t row_pitch_b;  // bytes per row
	if (U8)
	{
		src_base = P.img + (uint64_t)id.image * P.img_stride + ((uint64_t)td.y0 * P.img_pitch + td.x0) * CH;
		row_pitch_b = P.img_pitch * (uint32_t)CH;
	}
	else
	{
		const int16_t* src = P.src + ((P.src_tiled == 1) ? (uint64_t)id.image : base_inst) * P.src_inst_stride + (uint64_t)p_first * P.src_plane_stride;
		if (P.src_tiled == 1)
			src += (uint64_t)td.y0 * P.src_pitch + td.x0;
		else if (P.src_tiled & 2)  // written by the column-group kernel: planes shifted to the phase of the stream's lines
			src += stream_phase(P, id.image, td, P.src_tiled >> 8) + (uint32_t)p_first;
		src_base = reinterpret_cast<const uint8_t*>(src);
		row_pitch_b = P.src_pitch * 2u;
	}
	const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src_base), 0, (int)0xFFFFFFFFu, 0x00020000);
	const uint32_t src_lane_off = (uint32_t)lc.xs * (U8 ? (uint32_t)CH : 2u) + ((!U8 && pack) ? lane_tile * (uint32_t)P.src_inst_stride * 2u : 0u);

	// destinations.  Stream and LL stores go through raw buffer resources: a lane or a row that must not
	// store gets an out-of-range offset and the hardware drops the write (scripts/probe_buffer_store.hip;
	// 2-byte aligned dwords are fine there too).  Every row slot therefore issues the same 4 * NPL stores
	// with no branch around them, which keeps the compiler's vmcnt bookkeeping exact: the wait in front of
	// a slot's pixels no longer covers the stores of the previous slot (stores count in vmcnt on gfx950).
	const uint64_t tile_off = pack ? 0 : td.stream_off;  // (packed: the lane's tile offset travels in its byte offset)
	int16_t* tile_stream = P.stream + (uint64_t)id.image * P.stream_stride + tile_off;
	const uint64_t nsub = (uint64_t)Tc * Tr;
	const bool store_lane = lc.net && (c0 >= 0) && (c0 < Tc);
	constexpr uint32_t OOB = 0xFFFFFFFFu;
	constexpr int RSRC_FLAGS = 0x00020000;  // gfx9 raw buffer, 32 bit data format
	const uint64_t stream_left = (P.stream_stride - tile_off) * 2;  // bytes up to the end of the image's stream
	const __amdgpu_buffer_rsrc_t rs_stream = __builtin_amdgcn_make_buffer_rsrc(
	    tile_stream, 0, (int)(uint32_t)(stream_left < 0xFFFFFFFFull ? stream_left : 0xFFFFFFFFull), RSRC_FLAGS);
	int16_t* ll_root = P.ll_out_stream ? tile_stream : (P.dst + base_inst * P.dst_inst_stride);
	const uint64_t ll_left = P.ll_out_stream ? stream_left : (uint64_t)P.channels * P.dst_plane_stride * 2 * (pack ? P.n_tiles : 1u);
	const __amdgpu_buffer_rsrc_t rs_ll = __builtin_amdgcn_make_buffer_rsrc(
	    ll_root, 0, (int)(uint32_t)(ll_left < 0xFFFFFFFFull ? ll_left : 0xFFFFFFFFull), RSRC_FLAGS);
	const uint32_t ll_pitch = P.ll_out_stream ? (uint32_t)Tc : P.dst_pitch;
	// Store addresses without vector arithmetic: the per-lane part (voffset) is the lane's column pair, ONE register
	// for every store of the wave, out of range once and for all in a lane that must not store; everything else --
	// plane, sub-band, row -- is wave-uniform and travels as the scalar offset of the store, 0xFFFFFFFF for a row
	// that must not be stored.  The hardware checks voffset + soffset (as a sum that does not wrap,
	// scripts/probe_buffer_soffset.hip) against the end of the buffer.
	const uint32_t lane_off = store_lane ? (uint32_t)(c0 * 2) + lane_stream_b : OOB;
	// (packed: the low-pass plane of the lane's tile instance sits elsewhere than its stream: a second register; unpacked,
	// and always on the u8 side, both are the same value)
	const uint32_t lane_ll_off = !pack ? lane_off
	                             : (store_lane ? (uint32_t)(c0 * 2) + (P.ll_out_stream ? lane_stream_b : lane_tile * (uint32_t)P.dst_inst_stride * 2u) : OOB);
	uint32_t ll_off[NPL], grp_off[NPL];  // scalar: byte offset of column 0, row 0 of this plane's LL / C sub-band
#pragma unroll
	for (int p = 0; p < NPL; p++)
	{
		const int pl = p_first + p * P_STEP;
		grp_off[p] = (uint32_t)((P.grp_off[pl] + 1) * 2);
		ll_off[p] = (uint32_t)((P.ll_out_stream ? P.lp_off[pl] : (uint64_t)pl * P.dst_plane_stride) * 2);
	}
	const uint32_t nsub_b = (uint32_t)(nsub * 2);
	// GRP: this lane's packed words go to the workgroup's row buffers instead (planes p_first, p_first + 2: the second one
	// sits 8 row kinds and 2 values further on); one predicate for both, whatever it lets through beyond the owned
	// columns lands in the padding.  Addresses from lane16 alone (see GroupCtx): c0 = c_base + 2 * lane in every lane.
	const uint32_t lane16 = (uint32_t)lane * 16u;
	uint32_t lds_b = 0, src_a = 0;
	bool lds_lane = false;
	(void)lane16, (void)lds_b, (void)src_a, (void)lds_lane;
	if constexpr (GRP)
	{
		const int idx0 = c0 - gcp->S + p_first;
		lds_lane = store_lane && (idx0 >= -3) && (idx0 < GRP_OWN);
		lds_b = (uint32_t)__builtin_amdgcn_readfirstlane((GRP_PAD + idx0 + 4 * p_first * GRP_ROW) * 2 - 4 * lane);
		if constexpr (!HEDGE)  // no clamped / wrapped lane: the pixel offsets are 16 * lane + a wave constant
			src_a = (uint32_t)__builtin_amdgcn_readfirstlane((int)(src_lane_off - lane16));
	}

	if (pack ? (id.seg == 0 && lc.net && lc.he.first) : (id.strip == 0 && id.seg == 0 && lane == 0))  // the lift head of every tile
#pragma unroll
		for (int p = 0; p < NPL; p++)
			if (!(one_plane && p == 1))
				(tile_stream + (lane_stream_b >> 1))[P.grp_off[p_first + p * P_STEP]] = (int16_t)((p_first + p * P_STEP == 0) ? P.q_luma : P.q_chroma);

	const float gf_luma = (float)P.g_luma, gf_chroma = (float)P.g_chroma;

	// narrowing kernels compute on the integer pipe, the others (narrowing provably a no-op) on fp32
	using V = std::conditional_t<NARROW, int, float>;
	VFwd<V> st[NPL][4];
#pragma unroll
	for (int p = 0; p < NPL; p++)
#pragma unroll
		for (int k = 0; k < 4; k++)
			st[p][k] = VFwd<V>{{0, 0, 0}, {0, 0}, {0, 0, 0}};

	using Raw = FwdRaw<U8>;
	using RawVec = typename Raw::vec;
	auto fetch = [&](int v, Raw& raw) {
		const int m = VEDGE ? map_index(v, Tr, wrap) : v;
#pragma unroll
		for (int par = 0; par < 2; par++)
		{
			const int y = min(2 * max(m, 0) + par, chh - 1);  // phantom last row = copy of the last row
			const uint32_t row_off = (uint32_t)y * row_pitch_b;
			if (MEMONLY && ((P.dbg & 64) || ((P.dbg & 128) && id.pg == 1)))  // bit 6: stores only; bit 7: one wave of a pair loads
			{
				raw.a[par] = RawVec{};
				continue;
			}
#ifdef AKO_MEASURE
			if (P.dbg & 8192)  // bit 13: the real arithmetic on constant pixels (no loads)
			{
				raw.a[par] = RawVec{};
				continue;
			}
#endif
			if constexpr (U8 && CH == 3)
			{
				typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
				const u32x3 t = __builtin_bit_cast(u32x3, __builtin_amdgcn_raw_buffer_load_b96(rs_src, src_lane_off, row_off, AUX_FWD_PIXEL_LOAD));
				raw.a[par] = RawVec{t.x, t.y, t.z, 0u};
			}
			else if constexpr (U8 && GRP && !HEDGE)
				raw.a[par] = __builtin_bit_cast(RawVec, __builtin_amdgcn_raw_buffer_load_b128(rs_src, lane16, row_off + src_a, AUX_FWD_PIXEL_LOAD));
			else if constexpr (U8)
				raw.a[par] = __builtin_bit_cast(RawVec, __builtin_amdgcn_raw_buffer_load_b128(rs_src, src_lane_off, row_off, AUX_FWD_PIXEL_LOAD));
			else
				raw.a[par] = __builtin_bit_cast(RawVec, __builtin_amdgcn_raw_buffer_load_b64(rs_src, src_lane_off, row_off, 0));
		}
	};

	const int v_begin = r_lo - 3, n_slots = r_hi + 3 - v_begin;
	// The two slots fetched ahead of the loop are followed by as many (dropped) stores as a slot issues,
	// so that the memory operations in flight look the same on entry as on every later trip.
	auto phantom_stores = [&]() {
#pragma unroll
		for (int k = 0; k < 4 * NPL; k++)
			__builtin_amdgcn_raw_buffer_store_b32(0u, rs_stream, OOB, 0, 0);
	};
	// one row slot: K = unroll position (ring indices of the column pipeline), v = slot, raw = its two fetched rows
	// (raw_consumed() is called as soon as the slot's fetched rows have been turned into samples: see LATE below)
	auto do_slot = [&](auto kc, const int v, const Raw& raw, auto&& raw_consumed) {
			constexpr int K = decltype(kc)::value;
			const bool zero_row = VEDGE && (wrap == W_ZERO) && ((unsigned)v >= (unsigned)Tr);

			if constexpr (MEMONLY)
			{
				// measurement aid (AKO_HIP_DBG bit 4): the slot's loads and stores with no arithmetic between them --
				// what the memory system alone takes for this access pattern (the output is garbage)
				raw_consumed();
				const int r = v - 3;
				const bool row_ok = (r >= r_lo) && (r < r_hi) && !(P.dbg & 32);  // bit 5: loads only
				const uint32_t rr = (uint32_t)r;
				const uint32_t row_grp = rr * (uint32_t)Tc * 2u, row_ll = rr * ll_pitch * 2u;
				if (P.dbg & 256)  // bit 8: the same bytes as ONE 16-byte-per-lane store per plane (what wider stores would cost)
				{
					typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
					for (int p = 0; p < NPL; p++)
					{
						const uint32_t so = row_ok ? (rr * G.strips + id.strip) * 2048u + (uint32_t)p * 1024u : OOB;
						__builtin_amdgcn_raw_buffer_store_b128(u32x4{raw.a[0].x, raw.a[0].y, raw.a[1].x, raw.a[1].y}, rs_stream, (uint32_t)lane * 16u, so, 0);
					}
					return;
				}
#pragma unroll
				for (int p = 0; p < NPL; p++)
				{
					uint32_t a, b, c, d;
					if constexpr (U8)
						a = raw.a[p].x, b = raw.a[p].y, c = raw.a[p].z, d = raw.a[p].w;
					else
						a = raw.a[0].x, b = raw.a[0].y, c = raw.a[1].x, d = raw.a[1].y;
					__builtin_amdgcn_raw_buffer_store_b32(a, rs_ll, lane_off, row_ok ? ll_off[p] + row_ll : OOB, 0);
					__builtin_amdgcn_raw_buffer_store_b32(b, rs_stream, lane_off, row_ok ? grp_off[p] + row_grp : OOB, 0);
					__builtin_amdgcn_raw_buffer_store_b32(c, rs_stream, lane_off, row_ok ? grp_off[p] + row_grp + nsub_b : OOB, 0);
					__builtin_amdgcn_raw_buffer_store_b32(d, rs_stream, lane_off, row_ok ? grp_off[p] + row_grp + 2u * nsub_b : OOB, 0);
				}
				return;
			}
			V smp[2][NPL][4];
#pragma unroll
			for (int par = 0; par < 2; par++)
			{
				if constexpr (U8)
				{
					uint32_t px[4] = {raw.a[par].x, raw.a[par].y, raw.a[par].z, raw.a[par].w};
					if constexpr (CH == 3)
					{
						// twelve bytes = four RGB pixels: each into the low three bytes of a dword (the fourth byte, read as
						// "alpha" by the decoders, belongs to the plane this wave does not carry)
						px[3] = raw.a[par].z >> 8;
						px[2] = __builtin_amdgcn_alignbit(raw.a[par].z, raw.a[par].y, 16);
						px[1] = __builtin_amdgcn_alignbit(raw.a[par].y, raw.a[par].x, 24);
					}
					V v0[4], v1[4];
#ifdef AKO_MEASURE
					if (AKO_CUT & 8 || (P.dbg & 524288))  // bit 19 / AKO_CUT 8: no pixel decoding (the raw bits taken as samples)
					{
#pragma unroll
						for (int k = 0; k < 4; k++)
							v0[k] = (V)__uint_as_float(px[k]), v1[k] = (V)__uint_as_float(px[k] ^ 0x00400000u);
					}
					else
#endif
					if constexpr (CFAST)
						decode_pixels_ycocg<V>(px, (P.color == C_YCOCG_Q) ? (V)2 : (V)1, (int)id.pg, v0, v1);
					else
						decode_pixels_pair<V>(px, P.color, (int)id.pg, (CH == 4) && P.discard != 0, v0, v1);
#pragma unroll
					for (int k = 0; k < 4; k++)
					{
						smp[par][0][k] = zero_row ? (V)0 : v0[k];
						smp[par][1 % NPL][k] = zero_row ? (V)0 : v1[k];
					}
				}
				else
				{
					smp[par][0][0] = zero_row ? (V)0 : (V)lo16(raw.a[par].x);
					smp[par][0][1] = zero_row ? (V)0 : (V)hi16(raw.a[par].x);
					smp[par][0][2] = zero_row ? (V)0 : (V)lo16(raw.a[par].y);
					smp[par][0][3] = zero_row ? (V)0 : (V)hi16(raw.a[par].y);
				}
				if (HEDGE && lc.rot)  // fetched as samples W-4 .. W-1: move W-3, W-2, W-1 into place, phantom := E1
#pragma unroll
					for (int p = 0; p < NPL; p++)
					{
						smp[par][p][0] = smp[par][p][1], smp[par][p][1] = smp[par][p][2];
						smp[par][p][2] = smp[par][p][3];
					}
				if (HEDGE && lc.he.half)
#pragma unroll
					for (int p = 0; p < NPL; p++)
						smp[par][p][0] = smp[par][p][2], smp[par][p][1] = smp[par][p][3];
			}
			raw_consumed();

			const int r = v - 3;
			uint32_t w_ll[NPL] = {}, w_c[NPL] = {}, w_b[NPL] = {}, w_d[NPL] = {};
#pragma unroll
			for (int p = 0; p < NPL; p++)
			{
				if (one_plane && p == 1)  // wave-uniform
					continue;
				V e[4], o[4];  // columns: 0,1 = row low-pass of c0, c1; 2,3 = row high-pass of c0, c1
#ifdef AKO_MEASURE
				if (AKO_CUT & 2 || (P.dbg & 262144))  // bit 18 / AKO_CUT 2: no row pass
				{
#pragma unroll
					for (int k = 0; k < 4; k++)
						e[k] = smp[0][p][k], o[k] = smp[1][p][k];
				}
				else
#endif
				{
					hlift_forward<KIND, NARROW, HEDGE, V>(smp[0][p][0], smp[0][p][1], smp[0][p][2], smp[0][p][3], lc.he,
					                                      e[0], e[1], e[2], e[3]);
					hlift_forward<KIND, NARROW, HEDGE, V>(smp[1][p][0], smp[1][p][1], smp[1][p][2], smp[1][p][3], lc.he,
					                                      o[0], o[1], o[2], o[3]);
				}
				V lp[4], hp[4];
#ifdef AKO_MEASURE
				if (AKO_CUT & 4 || (P.dbg & 1048576))  // bit 20 / AKO_CUT 4: no column pass
				{
#pragma unroll
					for (int k = 0; k < 4; k++)
						lp[k] = e[k], hp[k] = o[k];
				}
				else
#endif
#pragma unroll
				for (int k = 0; k < 4; k++)
					vstep_forward<KIND, NARROW, VEDGE, K, V>(st[p][k], e[k], o[k], v, wrap, Tr, lp[k], hp[k]);

				// LL = (LP rows, LP cols), C = (HP rows, LP cols), B = (LP rows, HP cols), D = (HP, HP)
				const float gf = (p_first + p * P_STEP == 0) ? gf_luma : gf_chroma;
				const float rq = (p_first + p * P_STEP == 0) ? P.rq_luma : P.rq_chroma;
#ifdef AKO_MEASURE
				if constexpr (!NARROW)
				if (AKO_CUT & 1 || (P.dbg & 131072))  // bit 17 / AKO_CUT 1: no gate / quantizer (values packed as they are)
				{
					pack2x2_f(lp[0], lp[1], hp[0], hp[1], w_ll[p], w_c[p]);
					pack2x2_f(lp[2], lp[3], hp[2], hp[3], w_b[p], w_d[p]);
					continue;
				}
#endif
				// the deep variants (no loop: slot K is the K-th of the segment) know their fill slots at compile time: the first
				// six finish rows of the segment above, nothing of theirs is ever stored -- no gate, no quantizer, no stores
				if constexpr (DEEP > 0 && K < 6 && AKO_DEEP_FILL_CUT)
					continue;
				pack_row_f(lp, hp, gf, rq, w_ll[p], w_c[p], w_b[p], w_d[p]);

			}
			if constexpr (DEEP > 0 && K < 6 && AKO_DEEP_FILL_CUT)
				return;
			if constexpr (GRP)
			{
				const bool row_ok = (r >= r_lo) && (r < r_hi);  // the same in every wave of the workgroup
#ifdef AKO_MEASURE  // AKO_HIP_DBG bits 9..11: no barrier (races: timing only) / no drain / no row-buffer writes
				const bool m_nobar = P.dbg & 512, m_nodrain = P.dbg & 1024, m_nolds = P.dbg & 2048;
#else
				constexpr bool m_nobar = false, m_nodrain = false, m_nolds = false;
#endif
				if (lds_lane && !m_nolds)
				{
					uint8_t* q = reinterpret_cast<uint8_t*>(gcp->lds) + ((lane16 >> 2) + lds_b) + (K & 1) * GRP_KINDS * GRP_ROW * 2;
					auto put_rows = [&](auto odd) {
#pragma unroll
						for (int p = 0; p < NPL; p++)
						{
							grp_put<decltype(odd)::value>(q + 0 * GRP_ROW * 2, w_ll[p]);
							grp_put<decltype(odd)::value>(q + 1 * GRP_ROW * 2, w_c[p]);
							grp_put<decltype(odd)::value>(q + 2 * GRP_ROW * 2, w_b[p]);
							grp_put<decltype(odd)::value>(q + 3 * GRP_ROW * 2, w_d[p]);
							q += (8 * GRP_ROW + 2) * 2;
						}
					};
					if (lds_b & 2u)  // wave-uniform: both planes of the wave sit at odd positions
						put_rows(std::true_type{});
					else
						put_rows(std::false_type{});
				}
				// Every slot, inside the segment or not: one barrier, then this wave's share of the stores (out of range for a
				// row outside the segment).  The other buffer is rewritten only behind the NEXT barrier, which no wave passes
				// before every wave has finished reading this one.
				if (!m_nobar)
					__syncthreads();
				if (!m_nodrain)
					group_drain(*gcp, lane16, K & 1, r, row_ok, Tc);
			}
			else
			{
#ifdef AKO_MEASURE
				const bool row_ok = (r >= r_lo) && (r < r_hi) && !(P.dbg & 4096);  // bit 12: every store dropped
#else
				const bool row_ok = (r >= r_lo) && (r < r_hi);  // wave-uniform
#endif
				const uint32_t rr = (uint32_t)r;
				const uint32_t row_grp = rr * (uint32_t)Tc * 2u, row_ll = rr * ll_pitch * 2u;
#pragma unroll
				for (int p = 0; p < NPL; p++)
				{
					if (one_plane && p == 1)
						continue;
#ifdef AKO_MEASURE
					if (P.dbg & 16384)  // bit 14: no store instructions at all
						continue;
#endif
					const uint32_t s_ll = row_ok ? ll_off[p] + row_ll : OOB;
					const uint32_t s_c = row_ok ? grp_off[p] + row_grp : OOB;
					const uint32_t s_b = row_ok ? grp_off[p] + row_grp + nsub_b : OOB;
					const uint32_t s_d = row_ok ? grp_off[p] + row_grp + 2u * nsub_b : OOB;
					__builtin_amdgcn_raw_buffer_store_b32(w_ll[p], rs_ll, lane_ll_off, s_ll, 0);
					__builtin_amdgcn_raw_buffer_store_b32(w_c[p], rs_stream, lane_off, s_c, AUX_FWD_STREAM_STORE);
					__builtin_amdgcn_raw_buffer_store_b32(w_b[p], rs_stream, lane_off, s_b, AUX_FWD_STREAM_STORE);
					__builtin_amdgcn_raw_buffer_store_b32(w_d[p], rs_stream, lane_off, s_d, AUX_FWD_STREAM_STORE);
				}
			}
	};

	if constexpr (DEEP > 0)
	{
		(void)n_slots, (void)phantom_stores;
		Raw all[DEEP];
		static_for<DEEP>([&](auto kc) { fetch(v_begin + decltype(kc)::value, all[decltype(kc)::value]); });
		static_for<DEEP>([&](auto kc) { do_slot(kc, v_begin + decltype(kc)::value, all[decltype(kc)::value], [] {}); });
	}
	else if constexpr (LATE > 0)
	{
		// LATE row slots ahead with a ring of LATE entries: the fetch for slot v + LATE is issued right after slot v's
		// rows have been decoded into samples, into the registers they just left (one entry fewer than a ring that
		// is refilled at the top of the slot).  A wave's loads in flight are what bounds these kernels -- memory
		// latency under load is several microseconds -- so the ring is as deep as the register budget allows.
		static_assert(6 % LATE == 0, "the ring index must repeat with the unrolled row loop");
		Raw ring[LATE];
		static_for<LATE>([&](auto kc) {
			fetch(v_begin + decltype(kc)::value, ring[decltype(kc)::value]);
			if constexpr (!GRP)
				phantom_stores();
		});
		for (int base = 0; base < n_slots; base += 6)
		{
			if constexpr (U8 && KIND == K_DD137 && CH == 4 && CFAST && !GRP && !MEMONLY)  // (scripts/isa_lint.py finds the loop by this comment)
				asm volatile("; AKO_LOOP fwd_u8_general_h%0_v%1" ::"n"((int)HEDGE), "n"((int)VEDGE));
			if (!GRP && (G.lockstep & 1))  // (column groups meet at every row slot)
				__builtin_amdgcn_s_barrier();
			static_for<6>([&](auto kc) {
				constexpr int K = decltype(kc)::value;
				const int v = v_begin + base + K;
				do_slot(kc, v, ring[K % LATE], [&] { fetch(v + LATE, ring[K % LATE]); });
			});
		}
	}
	else
	{
		// PF row slots are fetched ahead of the one being worked on (2; 1 in the kernels that trade the third ring
		// entry's registers for a fourth wave per SIMD)
		static_assert(PF == 1 || PF == 2, "prefetch distance");
		Raw ring[PF + 1];
		fetch(v_begin, ring[0]);
		phantom_stores();
		if constexpr (PF == 2)
		{
			fetch(v_begin + 1, ring[1]);
			phantom_stores();
		}
		for (int base = 0; base < n_slots; base += 6)
		{
			if constexpr (!U8 && KIND == K_DD137 && !NARROW && !MEMONLY)  // (scripts/isa_lint.py finds the loop by this comment)
				asm volatile("; AKO_LOOP fwd_i16_general_h%0_v%1" ::"n"((int)HEDGE), "n"((int)VEDGE));
			if (G.lockstep & 1)
				__builtin_amdgcn_s_barrier();
			static_for<6>([&](auto kc) {
				constexpr int K = decltype(kc)::value;
				const int v = v_begin + base + K;
				fetch(v + PF, ring[(K + PF) % (PF + 1)]);  // prefetch (clamped rows: always in range)
				do_slot(kc, v, ring[K % (PF + 1)], [] {});
			});
		}
	}
}

template <int KIND, int NPL, bool U8, bool NARROW, int DEEP = 0>
// (5 waves per SIMD = 102 VGPRs: the small-level launches are sized as one round of 5 x 1024 waves, ako_plan.hip)
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(5))) void k_forward_stream(const LevelParams P, const StreamGeom G)
{
	const UnitId id = decode_unit(P, G);
	if (!id.valid)
		return;
	const int lane = threadIdx.x & 63;
	const LaneCols lc = (!U8 && geom_pack(G)) ? lane_columns_pack(lane, (int)P.sub_w, P.wrap, (int)min(geom_pack(G), P.n_tiles - id.tile))
	                                        : lane_columns(id.strip, G.strips, geom_wide(G), lane, (int)P.sub_w, (int)P.full_w, P.wrap);
	// segment touches the top / bottom border (or wraps over it): needs the row boundary code
	const bool vedge = segment_needs_border_code(G, id.seg, (int)P.sub_h);
	// u8 side: the usual colour mode gets straight-line pixel decoding (decode_pixels_ycocg)
	const bool cfast = U8 && (P.color == C_YCOCG || P.color == C_YCOCG_Q) && P.discard == 0;
#define AKO_FWD_BODY(H, V)                                                                       \
	do                                                                                           \
	{                                                                                            \
		if constexpr (U8)                                                                        \
		{                                                                                        \
			if (cfast)                                                                           \
				forward_stream_body<KIND, NPL, U8, NARROW, H, V, DEEP, true>(P, G, id, lc, lane);  \
			else                                                                                 \
				forward_stream_body<KIND, NPL, U8, NARROW, H, V, DEEP, false>(P, G, id, lc, lane); \
		}                                                                                        \
		else                                                                                     \
			forward_stream_body<KIND, NPL, U8, NARROW, H, V, DEEP, false>(P, G, id, lc, lane);     \
	} while (0)
	(void)cfast;
	bool hedge_ = lc.hedge;
#ifdef AKO_EXP_NOHEDGE_I16  // experiment (timing only, wrong values at the left / right borders): border strips on the interior bodies
	hedge_ = false;
#endif
	if (hedge_)
	{
		if (vedge)
			AKO_FWD_BODY(true, true);
		else
			AKO_FWD_BODY(true, false);
	}
	else
	{
		if (vedge)
			AKO_FWD_BODY(false, true);
		else
			AKO_FWD_BODY(false, false);
	}
#undef AKO_FWD_BODY
}

// The u8 level kernel (level 0 of RGBA images).  The streaming kernels are bound by how many waves a SIMD can
// interleave (one wave issues a VALU instruction every ~5 cycles at best, and nothing else hides the latencies of
// the unrolled row slots), so this kernel is held to 128 VGPRs = 4 waves per SIMD: the interior body and the
// left / right border body fit (prefetch distance 1 instead of 2, every address a scalar offset on one per-lane
// register), the top / bottom border bodies -- three row segments in a hundred -- spill a few dozen registers to
// scratch and are the only ones that do.
#ifndef AKO_U8_RING
#define AKO_U8_RING 1
#endif
#ifndef AKO_U8_WAVES
#define AKO_U8_WAVES 4
#endif
constexpr int U8_RING = AKO_U8_RING;  // row slots the u8 forward kernel fetches ahead (1, 2, 3 or 6; round 3: 1 -- the kernel is bound by
                                      // instruction issue, a second slot in flight buys nothing and costs eight registers: -3 % alone)

// Measurement builds only (-DAKO_MEASURE, scripts/build_variant.sh): the shipped library neither holds these kernels nor
// reads AKO_HIP_DBG.
#ifdef AKO_MEASURE
// measurement aid: the interior body's loads and stores without its arithmetic, every unit (AKO_HIP_DBG bit 4)
template <int UNUSED = 0>  // (a template: every translation unit of a measurement build may hold it)
__global__ __launch_bounds__(THREADS) void k_forward_stream_u8_memonly(const LevelParams P, const StreamGeom G)
{
	const UnitId id = decode_unit(P, G);
	if (!id.valid)
		return;
	const int lane = threadIdx.x & 63;
	const LaneCols lc = lane_columns(id.strip, G.strips, geom_wide(G), lane, (int)P.sub_w, (int)P.full_w, P.wrap);
	forward_stream_body<K_DD137, 2, true, false, false, true, 0, true, 2, U8_RING, true>(P, G, id, lc, lane);
}

template <int UNUSED = 0>  // (a template: every translation unit of a measurement build may hold it)
__global__ __launch_bounds__(THREADS) void k_forward_stream_i16_memonly(const LevelParams P, const StreamGeom G)
{
	const UnitId id = decode_unit(P, G);
	if (!id.valid)
		return;
	const int lane = threadIdx.x & 63;
	const LaneCols lc = lane_columns(id.strip, G.strips, geom_wide(G), lane, (int)P.sub_w, (int)P.full_w, P.wrap);
	forward_stream_body<K_DD137, 1, false, false, false, true, 0, false, 2, 0, true>(P, G, id, lc, lane);
}
#endif  // AKO_MEASURE

template <int KIND, int CH = 4>  // CH: bytes per pixel = channels (4 RGBA, 3 RGB)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(AKO_U8_WAVES, AKO_U8_WAVES))) void k_forward_stream_u8(const LevelParams P, const StreamGeom G)
{
	const UnitId id = decode_unit(P, G);
	if (!id.valid)
		return;
	const int lane = threadIdx.x & 63;
	const uint32_t row_tiles = geom_row_tiles(G);
	const LaneCols lc = row_tiles ? lane_columns_row(id.strip, lane, (int)P.sub_w, (int)min(row_tiles, P.n_tiles - id.tile), P.wrap)
	                              : lane_columns(id.strip, G.strips, geom_wide(G), lane, (int)P.sub_w, (int)P.full_w, P.wrap);
	const bool vedge = segment_needs_border_code(G, id.seg, (int)P.sub_h);
	// the usual colour mode gets straight-line pixel decoding (decode_pixels_ycocg)
	const bool cfast = (P.color == C_YCOCG || P.color == C_YCOCG_Q) && P.discard == 0;
#define AKO_FWD_U8(H, V)                                                                               \
	do                                                                                                 \
	{                                                                                                  \
		if (cfast)                                                                                     \
			forward_stream_body<KIND, 2, true, false, H, V, 0, true, 2, U8_RING, false, CH>(P, G, id, lc, lane);  \
		else                                                                                           \
			forward_stream_body<KIND, 2, true, false, H, V, 0, false, 2, U8_RING, false, CH>(P, G, id, lc, lane); \
	} while (0)
	if (__builtin_expect(vedge, 0))
	{
		if (lc.hedge)
			AKO_FWD_U8(true, true);
		else
			AKO_FWD_U8(false, true);
	}
	else
	{
		if (lc.hedge)
			AKO_FWD_U8(true, false);
		else
			AKO_FWD_U8(false, false);
	}
#undef AKO_FWD_U8
}

// The u8 level kernel in column groups (see "Column groups" above).  StreamGeom::strips holds the number of GROUPS;
// grid = groups x segments x tile instances workgroups of 8 waves.
template <int KIND, int CH = 4>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(AKO_U8_WAVES, AKO_U8_WAVES))) void k_forward_group_u8(const LevelParams P, const StreamGeom G)
{
	__shared__ __attribute__((aligned(16))) int16_t rowbuf[GRP_LDS_VALUES];
	// XCD-aware order as in decode_unit(): neighbouring groups / segments meet in one L2
	uint32_t blk = blockIdx.x;
	const uint32_t per_xcd = gridDim.x >> 3;
	if (blk < (per_xcd << 3))
		blk = (blk & 7) * per_xcd + (blk >> 3);
	const uint32_t group = blk % G.strips;
	blk /= G.strips;
	UnitId id;
	id.seg = blk % G.segs;
	blk /= G.segs;
	id.tile = blk % P.n_tiles;
	id.image = blk / P.n_tiles;
	id.valid = true;
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int lane = threadIdx.x & 63;
	const int strip = wave >> 1;
	id.pg = (uint32_t)((wave & 1) ^ ((wave >> 2) & 1));  // both roles on every SIMD (wave w runs on SIMD w % 4)
	id.strip = group * GRP_STRIPS + (uint32_t)strip;
	const int Tc = (int)P.sub_w, Tr = (int)P.sub_h;
	const TileDesc td = P.tiles[id.tile];
	const uint64_t inst = (uint64_t)id.image * P.n_tiles + id.tile;

	const uint32_t phase = stream_phase(P, id.image, td, (uint32_t)P.grp_off[0] + 1u);
	GroupCtx gc;
	gc.lds = rowbuf;
	gc.S = (int)group * GRP_OWN + (int)((64u - phase) & 63u) - 64;
	const int c_base = ((gc.S - 8) & ~1) + strip * SNET;  // lane 0 of the strip (net from lane 2: the group nets S - 4 .. S + 475 at least)
	const bool active = c_base + SORG < Tc;
	const LaneCols lc = lane_columns_at(c_base, 2, 62, lane, Tc, P.wrap);

	// this wave's share of the stores
	{
		constexpr int RSRC_FLAGS = 0x00020000;
		const int dp = wave >> 1;
		gc.exists = dp < CH;
		gc.d_first = gc.S - dp;
		int16_t* tile_stream = P.stream + (uint64_t)id.image * P.stream_stride + td.stream_off;
		int16_t* sub = tile_stream + (int64_t)(P.grp_off[gc.exists ? dp : 0] + 1) + gc.d_first;
		int16_t* ll = P.dst + inst * P.dst_inst_stride + (uint64_t)dp * P.dst_plane_stride + phase + (uint32_t)dp + gc.d_first;
		const uint32_t nsub_b = (uint32_t)((uint64_t)Tc * Tr * 2);
		// (the streaming kernels' tiles stay below 0xFFF00000 bytes: every real offset is in range, and 0xFFFFFFFF + any
		// lane offset is not)
		constexpr int DRAIN_RANGE = (int)0xFFF00000u;
		const __amdgpu_buffer_rsrc_t rs_stream = __builtin_amdgcn_make_buffer_rsrc(sub, 0, DRAIN_RANGE, RSRC_FLAGS);
		if (wave & 1)  // B and D
		{
			gc.rs[0] = rs_stream, gc.rs[1] = rs_stream;
			gc.soff[0] = nsub_b, gc.soff[1] = 2u * nsub_b;
			gc.pitch_b[0] = gc.pitch_b[1] = (uint32_t)Tc * 2u;
		}
		else  // LL (plane p shifted by phase + p values: its lines coincide with the stream's) and C
		{
			gc.rs[0] = __builtin_amdgcn_make_buffer_rsrc(ll, 0, DRAIN_RANGE, RSRC_FLAGS), gc.rs[1] = rs_stream;
			gc.soff[0] = 0, gc.soff[1] = 0;
			gc.pitch_b[0] = P.dst_pitch * 2u, gc.pitch_b[1] = (uint32_t)Tc * 2u;
		}
		gc.lds_k = (uint32_t)((2 * wave * GRP_ROW + GRP_PAD) * 2);
		gc.ragged = (gc.d_first < 0) || (gc.d_first + GRP_OWN > Tc);
	}

	if (!active)
	{
		// a strip beyond the right border (last group): no pixels, but the wave stores its share of the rows
		int r_lo, r_hi, len;
		segment_rows(G, id.seg, Tr, r_lo, r_hi, len);
		const int n_slots = (r_hi - r_lo + 6 + 5) / 6 * 6;  // as forward_stream_body walks them
		for (int k = 0; k < n_slots; k++)
		{
			const int r = r_lo - 6 + k;  // slot v = r_lo - 3 + k stores row v - 3
			__syncthreads();
			group_drain(gc, (uint32_t)lane * 16u, k & 1, r, (r >= r_lo) && (r < r_hi), Tc);
		}
		return;
	}
	const bool vedge = segment_needs_border_code(G, id.seg, Tr);
	const bool cfast = (P.color == C_YCOCG || P.color == C_YCOCG_Q) && P.discard == 0;
#define AKO_FWD_GRP(H, V)                                                                                          \
	do                                                                                                             \
	{                                                                                                              \
		if (cfast)                                                                                                 \
			forward_stream_body<KIND, 2, true, false, H, V, 0, true, 2, U8_RING, false, CH, true>(P, G, id, lc, lane, &gc);  \
		else                                                                                                       \
			forward_stream_body<KIND, 2, true, false, H, V, 0, false, 2, U8_RING, false, CH, true>(P, G, id, lc, lane, &gc); \
	} while (0)
	// (REPEAT has no border lanes to patch, but its wrapped lanes fetch from the other end of the row: not the
	// 16 * lane + constant the interior bodies address their pixels with)
	const bool hedge = lc.hedge || (c_base < 0) || (c_base + 128 > Tc);
	if (__builtin_expect(vedge, 0))
	{
		if (hedge)
			AKO_FWD_GRP(true, true);
		else
			AKO_FWD_GRP(false, true);
	}
	else
	{
		if (hedge)
			AKO_FWD_GRP(true, false);
		else
			AKO_FWD_GRP(false, false);
	}
#undef AKO_FWD_GRP
}

// ---------------------------------------------------------------------------------------------
// Inverse.  NPL = planes handled by one wave: 2 with U8 (the workgroup is then exactly one PAIR of
// waves working on the same strip and segment, see the file header), 1 on int16 planes.
// ---------------------------------------------------------------------------------------------

template <int NPL>
struct InvRaw
{
	uint32_t ll[NPL], c[NPL], b[NPL], d[NPL];  // two coefficients each
};

// OPT = optimistic float pipeline (u8 side only).  The reference wraps every intermediate to int16
// (wavelet-dd137.c:36-54); streams are untrusted, so the exact kernels (OPT = false) wrap too.  For real
// images no wrap ever happens, and then the same integers can be computed on the fp32 pipe (6 ops per
// step, cheaper (de)quantize / pack).  OPT proves per wave that no wrap happened:
//   * M = max |input| over everything the wave loads (LL and the de-quantized C, B, D, halo included);
//     one inverse 1-D pass grows magnitudes by at most 3.03x (E <= M + 20M/32, O <= M + 20*1.625M/16),
//     two passes by 9.2x, so M <= 3560 bounds every lifting intermediate below 32768
//   * Mo = max |lifted sample| over the wave's own planes; the colour inverse (format.c:138-218) grows
//     magnitudes by at most 3x (t <= 1.5 Mo, g <= 2.5 Mo, b <= 2 Mo, r <= 3 Mo), so Mo <= 10921 bounds its
//     intermediates below 32768; the two waves of a pair cover the four planes between them
// If either test fails anywhere the wave raises P.ovf_flag and the exact kernel, launched right behind
// on the same stream, redoes the level (it returns at once when the flag is clear).
constexpr float OPT_INPUT_BOUND = 3560.0f;
constexpr float OPT_OUTPUT_BOUND = 10921.0f;

template <int KIND, int NPL, bool U8, bool OPT, bool HEDGE, bool VEDGE, int DEEP, int PF = 2, bool MEMONLY = false, int CH = 4>
__device__ __forceinline__ void inverse_stream_body(const LevelParams& P, const StreamGeom& G, const UnitId& id,
                                                    const LaneCols& lc, int lane, uint4 (*xbuf)[2][2][64])
{
	static_assert(!OPT || U8, "the optimistic pipeline is used on the u8 side only");
	static_assert(CH == 4 || (CH == 3 && U8 && NPL == 2), "CH = 3: the u8 kernels on RGB pixels");
	// CH = 3 (RGB): the pair's second wave owns planes 2 and 3, and plane 3 does not exist: it carries one plane
	const bool one_plane = (CH == 3) && (id.pg == 1);
	using V = std::conditional_t<OPT, float, int>;
	float peak_in = 0.0f, peak_out = 0.0f;
	const TileDesc td = P.tiles[id.tile];
	const uint64_t inst = (uint64_t)id.image * P.n_tiles + id.tile;
	const int Tc = (int)P.sub_w, Tr = (int)P.sub_h;
	const int ow = (int)P.full_w, oh = (int)P.full_h;
	const int wrap = P.wrap;
	const int pair = (int)id.pg;  // U8: which half of the pixel this wave reconstructs
	const int p_first = U8 ? 2 * pair : (int)id.pg;
	const int c0 = lc.c0;

	int r_lo, r_hi, seg_len;
	segment_rows(G, id.seg, Tr, r_lo, r_hi, seg_len);
	(void)seg_len;

	// packed small tiles (lane_columns_pack, int16 levels only): resources based at the image's first tile instance / at the
	// image's stream, the lane adds its own tile's instance and stream offset (and reads its own tile's lift head)
	const bool pack = U8 ? (geom_row_tiles(G) != 0) : (geom_pack(G) != 0);  // (u8: row strips over a row of tiles)
	const uint32_t lane_tile = pack ? min(id.tile + (uint32_t)lc.tile_in_pack, P.n_tiles - 1u) : 0u;  // per lane
	const uint64_t base_inst = pack ? (uint64_t)id.image * P.n_tiles : inst;                               // wave-uniform
	const uint32_t lane_stream_b = pack ? (uint32_t)(P.tiles[lane_tile].stream_off * 2) : 0u;              // per lane
	const int16_t* tile_stream = P.stream + (uint64_t)id.image * P.stream_stride + (pack ? 0 : td.stream_off);
	const uint64_t nsub = (uint64_t)Tc * Tr;
	// Loads through raw buffer resources, as in the forward kernels: ONE register of byte offset per lane (its
	// column pair) for every load of the wave; plane, sub-band and row travel as the scalar offset.
	constexpr int RSRC_FLAGS = 0x00020000;
	const __amdgpu_buffer_rsrc_t rs_stream = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(tile_stream), 0, (int)0xFFFFFFFFu, RSRC_FLAGS);
	const int16_t* ll_root = P.ll_in_stream ? tile_stream : (P.src + base_inst * P.src_inst_stride);
	const __amdgpu_buffer_rsrc_t rs_ll = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(ll_root), 0, (int)0xFFFFFFFFu, RSRC_FLAGS);
	const uint32_t ll_pitch = P.ll_in_stream ? (uint32_t)Tc : P.src_pitch;
	const uint32_t lane_in_off = (uint32_t)lc.cs * 2u + lane_stream_b;
	// (packed: the low-pass plane of the lane's tile instance sits elsewhere than its stream: a second register)
	const uint32_t lane_ll_in_off = !pack ? lane_in_off
	                                : (uint32_t)lc.cs * 2u + (P.ll_in_stream ? lane_stream_b : lane_tile * (uint32_t)P.src_inst_stride * 2u);
	const uint32_t nsub_b = (uint32_t)(nsub * 2);
	uint32_t ll_off[NPL], grp_off[NPL];  // scalar: byte offset of column 0, row 0 of this plane's LL / C sub-band
	int qv[NPL];
#pragma unroll
	for (int p = 0; p < NPL; p++)
	{
		const int pl = (one_plane && p == 1) ? p_first : p_first + p;  // (the absent plane: any valid offsets, never used)
		qv[p] = (tile_stream + (lane_stream_b >> 1))[P.grp_off[pl]];  // the decoder trusts the lift head (misc.c:266-272, lifting.c:114-116)
		grp_off[p] = (uint32_t)((P.grp_off[pl] + 1) * 2);
		ll_off[p] = (uint32_t)((P.ll_in_stream ? P.lp_off[pl] : (uint64_t)pl * P.src_plane_stride) * 2);
	}

	// destination
	uint8_t* img = nullptr;
	int16_t* dst = nullptr;
	uint64_t out_pitch;
	if (U8)
	{
		img = P.img + (uint64_t)id.image * P.img_stride + ((uint64_t)td.y0 * P.img_pitch + td.x0) * CH;  // tile origin
		out_pitch = (uint64_t)P.img_pitch * CH;
	}
	else
	{
		dst = P.dst + (P.dst_tiled ? (uint64_t)id.image : (base_inst + lane_tile)) * P.dst_inst_stride +
		      (uint64_t)p_first * P.dst_plane_stride + 2 * c0;
		if (P.dst_tiled)
			dst += (uint64_t)td.y0 * P.dst_pitch + td.x0;
		out_pitch = P.dst_pitch;
	}
	// the output width is 2 * Tc here (level widths that are multiples of 4), rows may be odd in number
	const bool store_lane = lc.net && (c0 >= 0) && (c0 < Tc);
	(void)ow;
	// u8 side: the pixel row goes out through a raw buffer store as well -- the lane's four pixels are a byte offset
	// that is out of range in a lane that must not store, the row is the scalar offset (0xFFFFFFFF: row dropped)
	constexpr uint32_t OOB = 0xFFFFFFFFu;
	const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc(U8 ? (void*)img : (void*)nullptr, 0, U8 ? (int)0xFFFFFFFFu : 0, RSRC_FLAGS);
	const int img_c0 = (U8 && pack) ? (lc.xs >> 1) : c0;  // row strips: the lane's column inside the row of tiles
	const uint32_t px_lane_off = store_lane ? (uint32_t)(2 * img_c0) * (uint32_t)CH : OOB;
	(void)rs_img, (void)px_lane_off;

	VInv<V> st[NPL][4];
#pragma unroll
	for (int p = 0; p < NPL; p++)
#pragma unroll
		for (int k = 0; k < 4; k++)
			st[p][k] = VInv<V>{{0, 0, 0}, {0, 0, 0}, 0};

	using Raw = InvRaw<NPL>;
	auto fetch = [&](int v, Raw& raw) {
		const int m = max(VEDGE ? map_index(v, Tr, wrap) : v, 0);
		const uint32_t row_g = (uint32_t)m * (uint32_t)Tc * 2u, row_l = (uint32_t)m * ll_pitch * 2u;
		if (MEMONLY && (P.dbg & 64))  // measurement aid: stores only
		{
#pragma unroll
			for (int p = 0; p < NPL; p++)
				raw.ll[p] = raw.c[p] = raw.b[p] = raw.d[p] = (uint32_t)v;
			return;
		}
#ifdef AKO_MEASURE
		if (P.dbg & 32768)  // bit 15: the real arithmetic on constant coefficients (no loads)
		{
#pragma unroll
			for (int p = 0; p < NPL; p++)
				raw.ll[p] = raw.c[p] = raw.b[p] = raw.d[p] = 0x00010002u;
			return;
		}
#endif
#pragma unroll
		for (int p = 0; p < NPL; p++)
		{
			if (one_plane && p == 1)  // wave-uniform
			{
				raw.ll[p] = raw.c[p] = raw.b[p] = raw.d[p] = 0u;
				continue;
			}
			const uint32_t g = grp_off[p] + row_g;
			raw.ll[p] = __builtin_amdgcn_raw_buffer_load_b32(rs_ll, lane_ll_in_off, ll_off[p] + row_l, 0);
			raw.c[p] = __builtin_amdgcn_raw_buffer_load_b32(rs_stream, lane_in_off, g, AUX_INV_STREAM_LOAD);
			raw.b[p] = __builtin_amdgcn_raw_buffer_load_b32(rs_stream, lane_in_off, g + nsub_b, AUX_INV_STREAM_LOAD);
			raw.d[p] = __builtin_amdgcn_raw_buffer_load_b32(rs_stream, lane_in_off, g + 2u * nsub_b, AUX_INV_STREAM_LOAD);
		}
	};

	const int v_begin = r_lo - 3;
	const int n_slots = r_hi + 3 - v_begin;
	auto do_slot = [&](auto kc, const int v, const Raw& raw) {
			constexpr int K = decltype(kc)::value;
			const bool zero_row = VEDGE && (wrap == W_ZERO) && ((unsigned)v >= (unsigned)Tr);

			const int r = v - 3;
			if constexpr (MEMONLY && U8 && NPL == 2)
			{
				// measurement aid (AKO_HIP_DBG bit 4): the slot's loads and its pixel store with no arithmetic and no
				// exchange between them (garbage output); bit 5 loads only, bit 6 stores only
				typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
				const int y = 2 * r + pair;
				const bool row_ok = (r >= r_lo) && (r < r_hi) && (y < oh) && !(P.dbg & 32);
				const uint32_t s_row = row_ok ? (uint32_t)y * (uint32_t)out_pitch : OOB;
				__builtin_amdgcn_raw_buffer_store_b128(u32x4{raw.ll[0] ^ raw.c[0], raw.b[0] ^ raw.d[0], raw.ll[1] ^ raw.c[1], raw.b[1] ^ raw.d[1]},
				                                       rs_img, px_lane_off, s_row, AUX_INV_PIXEL_STORE);
				if ((P.dbg & 32) && (raw.ll[0] ^ raw.c[0] ^ raw.b[1] ^ raw.d[1]) == 0x12345678u)  // keeps the loads alive
					__builtin_amdgcn_raw_buffer_store_b32(raw.b[0], rs_img, px_lane_off, 0, 0);
				return;
			}
			const bool store_row = (r >= r_lo) && (r < r_hi) && store_lane;
			V out[2][NPL][4] = {};  // [row parity][plane][E0 O0 E1 O1]
#pragma unroll
			for (int p = 0; p < NPL; p++)
			{
				if (one_plane && p == 1)  // wave-uniform
					continue;
				// columns: 0,1 = row low-pass columns c0, c1 (LL over C); 2,3 = row high-pass (B over D)
				V lpv[4], hpv[4];
				if constexpr (OPT)  // float pipe: sign-extending word selects on the conversions (unpack2_f)
				{
					unpack2_f(raw.ll[p], lpv[0], lpv[1]);
					unpack2_f(raw.b[p], lpv[2], lpv[3]);
					unpack2_f(raw.c[p], hpv[0], hpv[1]);
					unpack2_f(raw.d[p], hpv[2], hpv[3]);
				}
				else
				{
					lpv[0] = (V)lo16(raw.ll[p]), lpv[1] = (V)hi16(raw.ll[p]);
					lpv[2] = (V)lo16(raw.b[p]), lpv[3] = (V)hi16(raw.b[p]);
					hpv[0] = (V)lo16(raw.c[p]), hpv[1] = (V)hi16(raw.c[p]);
					hpv[2] = (V)lo16(raw.d[p]), hpv[3] = (V)hi16(raw.d[p]);
				}
				const int q = qv[p];
				if (q > 1)  // lifting.c:30-40 (int16 wrap on the exact pipe)
				{
					if constexpr (OPT)
					{
						const float qf = (float)q;
						lpv[2] *= qf, lpv[3] *= qf;
#pragma unroll
						for (int k = 0; k < 4; k++)
							hpv[k] *= qf;
					}
					else
					{
						lpv[2] = (int16_t)(lpv[2] * q), lpv[3] = (int16_t)(lpv[3] * q);
#pragma unroll
						for (int k = 0; k < 4; k++)
							hpv[k] = (int16_t)(hpv[k] * q);
					}
				}
				if constexpr (OPT)
				{
#pragma unroll
					for (int k = 0; k < 4; k += 2)
					{
						absmax3(peak_in, (float)lpv[k], (float)lpv[k + 1]);
						absmax3(peak_in, (float)hpv[k], (float)hpv[k + 1]);
					}
				}
				if (zero_row)
#pragma unroll
					for (int k = 0; k < 4; k++)
						lpv[k] = 0, hpv[k] = 0;
				if (HEDGE && lc.he.half)
					lpv[0] = lpv[1], lpv[2] = lpv[3], hpv[0] = hpv[1], hpv[2] = hpv[3];

				V ev[4], od[4];
#pragma unroll
				for (int k = 0; k < 4; k++)
					vstep_inverse<KIND, VEDGE, K, V>(st[p][k], lpv[k], hpv[k], v, wrap, Tr, ev[k], od[k]);

				hlift_inverse<KIND, HEDGE, V>(ev[0], ev[1], ev[2], ev[3], lc.he, out[0][p][0], out[0][p][1], out[0][p][2],
				                              out[0][p][3]);
				hlift_inverse<KIND, HEDGE, V>(od[0], od[1], od[2], od[3], lc.he, out[1][p][0], out[1][p][1], out[1][p][2],
				                              out[1][p][3]);
			}

			if constexpr (U8 && OPT)
			{
				// This wave finishes pixel row 'pair' of the slot.  Hand the other wave our two planes of
				// ITS row, take its two planes of OUR row (double buffered, one barrier per slot; every
				// wave of the workgroup runs the same number of slots).  The pair index is wave-uniform:
				// both roles are spelled out so that no per-value select is left, and floats travel as
				// they are (no convert / pack / unpack around the LDS hop).
#pragma unroll
				for (int par = 0; par < 2; par++)
#pragma unroll
					for (int pp = 0; pp < NPL; pp++)
					{
						absmax3(peak_out, out[par][pp][0], out[par][pp][1]);
						absmax3(peak_out, out[par][pp][2], out[par][pp][3]);
					}
				auto as_u4 = [](const float* f) {
					return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
				};
				if (pair)
				{
					xbuf[K & 1][0][0][lane] = as_u4(out[0][0]);
					xbuf[K & 1][0][1][lane] = as_u4(out[0][1]);
				}
				else
				{
					xbuf[K & 1][1][0][lane] = as_u4(out[1][0]);
					xbuf[K & 1][1][1][lane] = as_u4(out[1][1]);
				}
				__syncthreads();
				const int y = 2 * r + pair;
#ifdef AKO_MEASURE
				const bool row_ok = (r >= r_lo) && (r < r_hi) && (y < oh) && !(P.dbg & 65536);  // bit 16: no pixel stores
#else
				const bool row_ok = (r >= r_lo) && (r < r_hi) && (y < oh);  // wave-uniform; phantom last row dropped (lifting.c:112,141)
#endif
				if (row_ok)
				{
					const uint4 g0 = xbuf[K & 1][pair][0][lane], g1 = xbuf[K & 1][pair][1][lane];
					const float his0[4] = {__uint_as_float(g0.x), __uint_as_float(g0.y), __uint_as_float(g0.z), __uint_as_float(g0.w)};
					const float his1[4] = {__uint_as_float(g1.x), __uint_as_float(g1.y), __uint_as_float(g1.z), __uint_as_float(g1.w)};
					uint32_t px[4];
					if (pair)
					{
#pragma unroll
						for (int k = 0; k < 4; k++)
						{
							float rr, gg, bb;
							color_inverse_fast(P.color, his0[k], his1[k], out[1][0][k], rr, gg, bb);
							px[k] = pixel_u8x4(rr, gg, bb, (CH == 3) ? 0.0f : out[1][1][k]);
						}
					}
					else
					{
#pragma unroll
						for (int k = 0; k < 4; k++)
						{
							float rr, gg, bb;
							color_inverse_fast(P.color, out[0][0][k], out[0][1][k], his0[k], rr, gg, bb);
							px[k] = pixel_u8x4(rr, gg, bb, (CH == 3) ? 0.0f : his1[k]);
						}
					}
					const uint32_t s_row = (uint32_t)y * (uint32_t)out_pitch;
					typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
					typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
					if constexpr (CH == 3)  // four RGB pixels = twelve bytes (widths here are multiples of four pixels)
						__builtin_amdgcn_raw_buffer_store_b96(rgb_pack4(px), rs_img, px_lane_off, s_row, AUX_INV_PIXEL_STORE);
					else if (HEDGE && lc.he.drop_last)  // odd width: the fourth pixel does not exist
						__builtin_amdgcn_raw_buffer_store_b96(u32x3{px[0], px[1], px[2]}, rs_img, px_lane_off, s_row, AUX_INV_PIXEL_STORE);
					else
						__builtin_amdgcn_raw_buffer_store_b128(u32x4{px[0], px[1], px[2], px[3]}, rs_img, px_lane_off, s_row, AUX_INV_PIXEL_STORE);
					AKO_STORE_GUARD();  // (see store_b128_guarded)
				}
			}
			else if constexpr (U8)
			{
				// exact pipe: same exchange, int16 pairs
				uint4 send;
				send.x = pack2(to_int(pair ? out[0][0][0] : out[1][0][0]), to_int(pair ? out[0][0][1] : out[1][0][1]));
				send.y = pack2(to_int(pair ? out[0][0][2] : out[1][0][2]), to_int(pair ? out[0][0][3] : out[1][0][3]));
				send.z = pack2(to_int(pair ? out[0][1][0] : out[1][1][0]), to_int(pair ? out[0][1][1] : out[1][1][1]));
				send.w = pack2(to_int(pair ? out[0][1][2] : out[1][1][2]), to_int(pair ? out[0][1][3] : out[1][1][3]));
				xbuf[K & 1][1 - pair][0][lane] = send;
				__syncthreads();
				const uint4 got = xbuf[K & 1][pair][0][lane];
				const int y = 2 * r + pair;
				if (store_row && y < oh)
				{
					const uint32_t gw[4] = {got.x, got.y, got.z, got.w};
					uint32_t px[4];
#pragma unroll
					for (int k = 0; k < 4; k++)
					{
						const V mine0 = pair ? out[1][0][k] : out[0][0][k];
						const V mine1 = pair ? out[1][1][k] : out[0][1][k];
						const V his0 = (V)((k & 1) ? hi16(gw[k >> 1]) : lo16(gw[k >> 1]));
						const V his1 = (V)((k & 1) ? hi16(gw[2 + (k >> 1)]) : lo16(gw[2 + (k >> 1)]));
						const V v0 = pair ? his0 : mine0, v1 = pair ? his1 : mine1;
						const V v2 = pair ? mine0 : his0, v3 = pair ? mine1 : his1;
						int rr, gg, bb;
						color_inverse((int)P.color, (int)v0, (int)v1, (int)v2, rr, gg, bb);
						px[k] = (uint32_t)sat8(rr) | ((uint32_t)sat8(gg) << 8) | ((uint32_t)sat8(bb) << 16) |
						        ((CH == 3) ? 0u : ((uint32_t)sat8((int)v3) << 24));
					}
					uint8_t* row = img + (uint64_t)y * out_pitch + (int64_t)(2 * img_c0) * CH;
					if constexpr (CH == 3)
					{
						const auto w3 = rgb_pack4(px);
						uint32_t* o = reinterpret_cast<uint32_t*>(row);
						o[0] = w3.x, o[1] = w3.y, o[2] = w3.z;
					}
					else if (HEDGE && lc.he.drop_last)  // odd width: the fourth pixel does not exist
					{
						uint32_t* o = reinterpret_cast<uint32_t*>(row);
						o[0] = px[0], o[1] = px[1], o[2] = px[2];
					}
					else
						*reinterpret_cast<uint4*>(row) = make_uint4(px[0], px[1], px[2], px[3]);
				}
			}
			else if (store_row)
			{
#pragma unroll
				for (int par = 0; par < 2; par++)
				{
					const int y = 2 * r + par;
					if (y < oh)
					{
						int16_t* o = dst + (uint64_t)y * out_pitch;
						if (HEDGE && lc.he.drop_last)  // odd width: three samples
						{
							*reinterpret_cast<uint32_t*>(o) = pack2(to_int(out[par][0][0]), to_int(out[par][0][1]));
							o[2] = (int16_t)to_int(out[par][0][2]);
						}
						else
							*reinterpret_cast<uint2*>(o) = make_uint2(pack2(to_int(out[par][0][0]), to_int(out[par][0][1])),
							                                          pack2(to_int(out[par][0][2]), to_int(out[par][0][3])));
					}
				}
			}
	};

	if constexpr (DEEP > 0)
	{
		static_assert(!U8, "deep prefetch: int16 planes only (the u8 pair exchanges through LDS per slot)");
		(void)n_slots;
		Raw all[DEEP];
		static_for<DEEP>([&](auto kc) { fetch(v_begin + decltype(kc)::value, all[decltype(kc)::value]); });
		static_for<DEEP>([&](auto kc) { do_slot(kc, v_begin + decltype(kc)::value, all[decltype(kc)::value]); });
	}
	else
	{
		static_assert(PF == 1 || PF == 2, "prefetch distance");
		Raw ring[PF + 1];
		fetch(v_begin, ring[0]);
		if constexpr (PF == 2)
			fetch(v_begin + 1, ring[1]);
		for (int base = 0; base < n_slots; base += 6)
		{
			if constexpr (U8 && OPT && KIND == K_DD137 && CH == 4 && !MEMONLY)  // (scripts/isa_lint.py finds the loop by this comment)
				asm volatile("; AKO_LOOP inv_u8_general_h%0_v%1" ::"n"((int)HEDGE), "n"((int)VEDGE));
			if constexpr (!U8 && KIND == K_DD137 && !MEMONLY)
				asm volatile("; AKO_LOOP inv_i16_general_h%0_v%1" ::"n"((int)HEDGE), "n"((int)VEDGE));
			if constexpr (!U8)  // (the u8 pairs already meet at the barriers of their LDS exchange, every slot)
			{
				if (G.lockstep & 1)
					__builtin_amdgcn_s_barrier();
			}
			static_for<6>([&](auto kc) {
				constexpr int K = decltype(kc)::value;
				const int v = v_begin + base + K;
				fetch(v + PF, ring[(K + PF) % (PF + 1)]);
				do_slot(kc, v, ring[K % (PF + 1)]);
			});
		}
	}
	if constexpr (OPT)
	{
		const bool bad = !(peak_in <= OPT_INPUT_BOUND) || !(peak_out <= OPT_OUTPUT_BOUND);  // negated: NaN counts as bad
		if (__any(bad) && lane == 0)
			atomicMax(P.ovf_flag, P.ovf_gen);
	}
}

template <int KIND, int NPL, bool U8, bool OPT, int DEEP = 0>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(3))) void k_inverse_stream(const LevelParams P, const StreamGeom G)
{
	__shared__ uint4 xbuf[2][2][2][64];  // U8 only: [slot parity][destination wave of the pair][plane][lane]
	if (!OPT && P.ovf_flag != nullptr)
	{
		// exact re-run behind an optimistic launch: nothing to do unless that launch raised the flag
		if (__builtin_amdgcn_readfirstlane(*(volatile const int32_t*)P.ovf_flag) != P.ovf_gen)
			return;
	}
	const UnitId id = decode_unit(P, G);
	if (!id.valid)
		return;  // U8: units come in pairs and the workgroup is one pair, so both waves leave together
	const int lane = threadIdx.x & 63;
	const LaneCols lc = (!U8 && geom_pack(G)) ? lane_columns_pack(lane, (int)P.sub_w, P.wrap, (int)min(geom_pack(G), P.n_tiles - id.tile))
	                                        : lane_columns(id.strip, G.strips, geom_wide(G), lane, (int)P.sub_w, (int)P.full_w, P.wrap);
	const bool vedge = segment_needs_border_code(G, id.seg, (int)P.sub_h);
	bool hedge_ = lc.hedge;
#ifdef AKO_EXP_NOHEDGE_I16
	hedge_ = false;
#endif
	if (hedge_)
	{
		if (vedge)
			inverse_stream_body<KIND, NPL, U8, OPT, true, true, DEEP>(P, G, id, lc, lane, xbuf);
		else
			inverse_stream_body<KIND, NPL, U8, OPT, true, false, DEEP>(P, G, id, lc, lane, xbuf);
	}
	else
	{
		if (vedge)
			inverse_stream_body<KIND, NPL, U8, OPT, false, true, DEEP>(P, G, id, lc, lane, xbuf);
		else
			inverse_stream_body<KIND, NPL, U8, OPT, false, false, DEEP>(P, G, id, lc, lane, xbuf);
	}
}

// The u8 inverse level kernel at 4 waves per SIMD (see k_forward_stream_u8): the workgroup is 1, 2 or 4 pairs of
// waves -- neighbouring strips, which the barriers of the per-slot LDS exchange then keep on the same rows
// (StreamGeom::lockstep); 8 KB of dynamic LDS per pair.
// OPT = optimistic fp32 pipeline; the exact re-run behind it (OPT = false) returns at once unless flagged.
#ifndef AKO_U8_INV_PF
#define AKO_U8_INV_PF 1
#endif
constexpr int U8_INV_PF = AKO_U8_INV_PF;  // row slots the u8 inverse kernel fetches ahead (1 or 2; round 3: 1, -3 % alone, as U8_RING)
constexpr uint32_t INV_U8_LDS_PER_PAIR = 2 * 2 * 2 * 64 * sizeof(uint4);
template <int KIND, bool OPT, int CH = 4>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(AKO_U8_WAVES, AKO_U8_WAVES))) void k_inverse_stream_u8(const LevelParams P, const StreamGeom G)
{
	extern __shared__ uint4 xdyn[];  // [pair of the workgroup][slot parity][destination wave of the pair][plane][lane]
	uint4(*xbuf)[2][2][64] = reinterpret_cast<uint4(*)[2][2][64]>(xdyn + (threadIdx.x >> 7) * (2 * 2 * 2 * 64));
	if (!OPT && P.ovf_flag != nullptr)
	{
		if (__builtin_amdgcn_readfirstlane(*(volatile const int32_t*)P.ovf_flag) != P.ovf_gen)
			return;
	}
	const UnitId id = decode_unit(P, G);
	if (!id.valid)
		return;  // units come in pairs, so both waves of a pair leave together (a barrier does not wait for ended waves)
	const int lane = threadIdx.x & 63;
	const uint32_t row_tiles = geom_row_tiles(G);
	const LaneCols lc = row_tiles ? lane_columns_row(id.strip, lane, (int)P.sub_w, (int)min(row_tiles, P.n_tiles - id.tile), P.wrap)
	                              : lane_columns(id.strip, G.strips, geom_wide(G), lane, (int)P.sub_w, (int)P.full_w, P.wrap);
	const bool vedge = segment_needs_border_code(G, id.seg, (int)P.sub_h);
	if (__builtin_expect(vedge, 0))
	{
		if (lc.hedge)
			inverse_stream_body<KIND, 2, true, OPT, true, true, 0, U8_INV_PF, false, CH>(P, G, id, lc, lane, xbuf);
		else
			inverse_stream_body<KIND, 2, true, OPT, false, true, 0, U8_INV_PF, false, CH>(P, G, id, lc, lane, xbuf);
	}
	else
	{
		if (lc.hedge)
			inverse_stream_body<KIND, 2, true, OPT, true, false, 0, U8_INV_PF, false, CH>(P, G, id, lc, lane, xbuf);
		else
			inverse_stream_body<KIND, 2, true, OPT, false, false, 0, U8_INV_PF, false, CH>(P, G, id, lc, lane, xbuf);
	}
}

#ifdef AKO_MEASURE
// measurement aid (AKO_HIP_DBG bit 4): the u8 inverse level kernel's loads and stores alone
template <int UNUSED = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(AKO_U8_WAVES, AKO_U8_WAVES))) void k_inverse_stream_u8_memonly(const LevelParams P, const StreamGeom G)
{
	const UnitId id = decode_unit(P, G);
	if (!id.valid)
		return;
	const int lane = threadIdx.x & 63;
	const LaneCols lc = lane_columns(id.strip, G.strips, geom_wide(G), lane, (int)P.sub_w, (int)P.full_w, P.wrap);
	inverse_stream_body<K_DD137, 2, true, true, false, true, 0, 2, true>(P, G, id, lc, lane, nullptr);
}
#endif  // AKO_MEASURE

}  // namespace ako

#include "ako_u8_lean.hip.h"  // the lean u8 level-0 kernels (round 4)
