// ako_u8_lean.hip.h -- the LEAN u8 level-0 kernels (round 4): k_forward_u8_lean / k_inverse_u8_lean.
//
// Same strip walk, same arithmetic and same results as k_forward_stream_u8 / k_inverse_stream_u8 (ako_stream.hip.h: one
// wave64 per strip of 128 coefficient columns and pair of planes, rows in registers, taps through DPP shifts, column pass
// as a register pipeline), for the configurations that carry practically all pixels: YCoCg / YCoCg_Q colour, DD13/7 or
// CDF5/3, CLAMP / REPEAT / ZERO borders, level widths that are multiples of four, ordinary strips and (not REPEAT) the one
// wide strip of a tile of 121..128 columns (256-pixel tiles); no row strips over tiles.  Everything else -- MIRROR, Haar, other colour modes, the discard rule, odd widths, packed
// tiles -- stays on the general kernels, which the host picks per launch (ako_plan.hip: lean_u8_level()).
//
// Why a second pair of kernels (measured on one box in one call, profiles/r4_lean_ab.txt):
//   * NO SCRATCH.  The general kernels serve every geometry from one source; their border bodies spill (76 / 104 bytes of
//     private segment in round 3).  A private segment costs a launch time even when no wave ever touches it: the lean
//     inverse kernel alone 0.209 ms, the same kernel with 176 unused bytes of scratch 0.230 ms, with the general border
//     bodies linked in 0.250 ms.  Every body here fits its 128 registers without a spill, so the kernels have no private
//     segment at all (tests/test_isa_lint.py checks the kernel descriptors)
//   * NO CONTROL FLOW in the row loops.  ROLE (which wave of the pair), left / right border and top / bottom border are
//     template parameters; what is left of the border rules are selects on wave-uniform or per-lane masks
//     (fix_halo_lanes_bf, vstep_*_bf).  The general border bodies branch around every patch: a strip at a tile border ran
//     2.4-3 x the instructions of an interior one (6 000-7 900 against 2 500 per six slots, with up to 200 scratch accesses)
//     and its waves were the tail of the launch
//   * the first trip of six row slots only fills the column pipeline (its output rows belong to the segment above): it
//     runs without row pass / swap / colour / stores (inverse) and without gate, quantizer and stores (forward)
//   * inverse: the plane swap goes through LDS as [plane][sample][lane] dwords (no four-register staging for 16-byte LDS
//     accesses: 288 moves per six slots in round 3), de-quantization is an unconditional multiply, and ONE proof obligation
//     covers the optimistic fp32 pipeline: |input| <= OPT_INPUT_BOUND_TIGHT bounds every lifted sample by 9.2 x that
//     <= OPT_OUTPUT_BOUND, so the outputs need not be tracked (the exact kernel behind it is the general one, as before)
// VALU instructions per six row slots of an interior wave: inverse 2 166 -> 1 790, forward 2 162 -> see profiles/r4_isa_lint.txt.
//
// Reference: library/wavelet-dd137.c:57-702, wavelet-cdf53.c:57-362, format.c:87-229, lifting.c:30-40,154-168.
#pragma once

namespace ako
{

#ifndef AKO_U8L_LOCKSTEP
#define AKO_U8L_LOCKSTEP 1  // experiments: 0 = no lockstep barrier in the lean forward kernel
#endif
#ifndef AKO_U8L_INV_PF
#define AKO_U8L_INV_PF 1  // row slots the lean inverse kernel fetches ahead (1 or 2; 2 measured no faster)
#endif

// |input| <= 1187  =>  |lifted sample| <= 9.2 * 1187 < 10921 = OPT_OUTPUT_BOUND  (see OPT_INPUT_BOUND in ako_stream.hip.h)
constexpr float OPT_INPUT_BOUND_TIGHT = 1187.0f;

// LDS of one pair of waves for the plane swap: [slot parity][destination role][plane][sample][lane] floats = 8 KiB
constexpr int XI_PLANE = 4 * 64, XI_ROLE = 2 * XI_PLANE, XI_BUF = 2 * XI_ROLE;
static_assert(2 * XI_BUF * sizeof(float) == INV_U8_LDS_PER_PAIR, "same LDS per pair as the general kernel");

// ---- in-kernel phase stamps (measurement builds: scripts/build_rgba_variant.sh stamps -DAKO_STAMPS) --------------------
// s_memtime between the phases of a row slot, per-phase sums kept in scalars and added to ako_lean_stamps[direction][phase]
// by lane 0 when the wave ends (a row per wave: [9] counts waves, [8] sums wave lifetimes); scripts/issue_model.py reads them through
// akoHipLeanStamps().  Every stamp first pins the values of the phase it closes (AKO_PIN: otherwise instruction selection
// computes them where they are used, phases later).  The shipped library holds none of this.
#ifdef AKO_STAMPS
#if AKO_STAMPS != 2
#define AKO_STAMPS_PHASES 1  // (the slots' bodies split into phases; -DAKO_STAMPS=2 leaves them as shipped)
#endif
constexpr int STAMP_WAVES = 16384;  // (a row per wave: 8 000 waves adding to sixteen shared words serialise for a millisecond)
__device__ unsigned long long ako_lean_stamps[2][STAMP_WAVES][12];  // [10], [11]: birth and end of the row's latest wave (absolute)
// One slot in six is stamped (unroll position 0), and a slot's stamps are only read at its end: s_memtime answers through the
// scalar cache after several hundred cycles, and a wave that waited for each answer where it asked ran four times as long.
#define AKO_STAMP_DECL                                          \
	unsigned long long st_acc[8] = {};                          \
	unsigned long long st_t[8] = {};                            \
	unsigned long long st_prev = __builtin_amdgcn_s_memtime();  \
	const unsigned long long st_born = st_prev
#define AKO_STAMP_K(i, K_)                                       \
	do                                                           \
	{                                                            \
		if constexpr ((K_) == 0)                                 \
		{                                                        \
			__builtin_amdgcn_sched_barrier(0);                   \
			st_t[i] = __builtin_amdgcn_s_memtime();              \
			__builtin_amdgcn_sched_barrier(0);                   \
		}                                                        \
	} while (0)
#define AKO_STAMP(i) AKO_STAMP_K(i, K)
// the sampled slot's phases first .. last: st_prev = the stamp in front of phase `first`
#define AKO_STAMP_SLOT_BEGIN(K_)                                 \
	do                                                           \
	{                                                            \
		if constexpr ((K_) == 0)                                 \
		{                                                        \
			__builtin_amdgcn_sched_barrier(0);                   \
			st_prev = __builtin_amdgcn_s_memtime();              \
			__builtin_amdgcn_sched_barrier(0);                   \
		}                                                        \
	} while (0)
#define AKO_STAMP_SLOT_END(K_, first, last)                      \
	do                                                           \
	{                                                            \
		if constexpr ((K_) == 0)                                 \
		{                                                        \
			__builtin_amdgcn_sched_barrier(0);                   \
			unsigned long long p_ = st_prev;                     \
			for (int i_ = (first); i_ <= (last); i_++)           \
				st_acc[i_] += st_t[i_] - p_, p_ = st_t[i_];      \
			__builtin_amdgcn_sched_barrier(0);                   \
		}                                                        \
	} while (0)
#define AKO_STAMP_NOW(i)                                                 \
	do                                                                   \
	{                                                                    \
		__builtin_amdgcn_sched_barrier(0);                               \
		const unsigned long long st_now = __builtin_amdgcn_s_memtime();  \
		st_acc[i] += st_now - st_prev;                                   \
		st_prev = st_now;                                                \
		__builtin_amdgcn_sched_barrier(0);                               \
	} while (0)
#define AKO_STAMP_FLUSH(dir)                                                                                           \
	do                                                                                                                 \
	{                                                                                                                  \
		if (lane == 0)                                                                                                 \
		{                                                                                                              \
			unsigned long long* row_ = ako_lean_stamps[dir][(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % STAMP_WAVES]; \
			for (int i_ = 0; i_ < 8; i_++)                                                                             \
				row_[i_] += st_acc[i_];                                                                                \
			const unsigned long long st_end = __builtin_amdgcn_s_memtime();                                            \
			row_[8] += st_end - st_born;                                                                               \
			row_[9] += 1ull;                                                                                           \
			row_[10] = st_born, row_[11] = st_end;                                                                     \
		}                                                                                                              \
	} while (0)
#define AKO_PIN4(a) asm volatile("" ::"v"((a)[0]), "v"((a)[1]), "v"((a)[2]), "v"((a)[3]))
#define AKO_WAIT_VM(n, K_)                                                                   \
	do                                                                                       \
	{                                                                                        \
		if constexpr ((K_) == 0)                                                             \
			__builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14));             \
	} while (0)
#if AKO_STAMPS == 2
// -DAKO_STAMPS=2: only birth, end and PLACE of every wave (HW_ID: wave slot / SIMD / CU / SE, XCC_ID; workgroup and wave in it):
// two s_memtime per wave, the slots untouched (scripts/wave_places.py: who shared a SIMD with whom, and for how long)
#undef AKO_STAMP_DECL
#undef AKO_STAMP_K
#undef AKO_STAMP_SLOT_BEGIN
#undef AKO_STAMP_SLOT_END
#undef AKO_STAMP_NOW
#undef AKO_STAMP_FLUSH
#undef AKO_PIN4
#undef AKO_WAIT_VM
#define AKO_STAMP_DECL const unsigned long long st_born = __builtin_amdgcn_s_memtime()
#define AKO_STAMP_TOP const unsigned long long st_top_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#define AKO_STAMP_TOP_SET(id_) (id_).t_top = st_top_, (id_).t_dec = __builtin_amdgcn_s_memtime()
#define AKO_STAMP_LC_SET(id_) __builtin_amdgcn_sched_barrier(0); (id_).t_lc = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#define AKO_STAMP_K(i, K_)
#define AKO_STAMP_SLOT_BEGIN(K_)
#define AKO_STAMP_SLOT_END(K_, first, last)
#define AKO_STAMP_NOW(i)
#define AKO_PIN4(a)
#define AKO_WAIT_VM(n, K_)
#define AKO_STAMP_FLUSH(dir)                                                                                           \
	do                                                                                                                 \
	{                                                                                                                  \
		const unsigned hw_ = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc_ = __builtin_amdgcn_s_getreg(20 | (31 << 11)); \
		if (lane == 0)                                                                                                 \
		{                                                                                                              \
			unsigned long long* row_ = ako_lean_stamps[dir][(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % STAMP_WAVES]; \
			const unsigned long long st_end = __builtin_amdgcn_s_memtime();                                            \
			row_[0] = hw_, row_[1] = xcc_, row_[2] = blockIdx.x, row_[3] = threadIdx.x >> 6;                           \
			row_[4] = id.strip, row_[5] = id.seg | ((unsigned long long)(id.t_dec - id.t_top) << 32),                  \
			row_[6] = id.pg | ((unsigned long long)(id.t_lc - id.t_top) << 32), row_[7] = (HEDGE ? 1 : 0) | (VEDGE ? 2 : 0); \
			__builtin_amdgcn_s_waitcnt(0x0F70);  /* vmcnt(0): the wave's last stores acknowledged */                     \
			const unsigned long long st_drained = __builtin_amdgcn_s_memtime();                                        \
			row_[8] = id.t_top;                                                                                        \
			row_[9] = st_drained;                                                                                      \
			row_[10] = st_born, row_[11] = st_end;                                                                     \
		}                                                                                                              \
	} while (0)
#endif
#else
#define AKO_STAMP_DECL
#define AKO_STAMP(i)
#define AKO_STAMP_K(i, K_)
#define AKO_STAMP_SLOT_BEGIN(K_)
#define AKO_STAMP_SLOT_END(K_, first, last)
#define AKO_STAMP_NOW(i)
#define AKO_STAMP_FLUSH(dir)
#define AKO_PIN4(a)
#define AKO_WAIT_VM(n, K_)
#endif

// (HEdgeBF, hedge_bf(), keep_tap(), hlift_inverse_bf(), hlift_forward_bf(): in ako_stream.hip.h since the general kernels use them too)

// ---- top / bottom tile border without control flow (CLAMP, REPEAT, ZERO) -------------------------------------------
// vstep_forward() / vstep_inverse() of ako_stream.hip.h with the border patches as selects on wave-uniform conditions
// (the row slot v is a scalar): REPEAT patches nothing (its rows come in through map_index), CLAMP repeats the nearest
// row of the produced sequence, ZERO zeroes it.
struct VEdgeBF
{
	bool patch;  // CLAMP or ZERO
	bool zero;   // ZERO
};
// map_index() (ako_kernels.hip.h) as scalar selects: the row a slot v in [-3, T + 2] reads.  CLAMP: the nearest row; REPEAT: v mod T
// (two conditional steps each way, as there); ZERO: any valid row (the callers zero what they read).  Further out -- the slots
// a trip of six runs past the segment, the prefetch -- any valid row will do.
__device__ __forceinline__ int map_index_bf(int v, int T, int wrap)
{
	const int t = (wrap == W_REPEAT) ? T : 0;
	int m = v;
	m += (m < 0) ? t : 0;
	m += (m < 0) ? t : 0;
	m -= (m >= T) ? t : 0;
	m -= (m >= T) ? t : 0;
	return min(max(m, 0), T - 1);
}
template <int KIND, bool VEDGE, int K>
__device__ __forceinline__ void vstep_forward_bf(VFwd<float>& s, float E, float O, int v, const VEdgeBF& ve, int T, float& lp_out,
                                                 float& hp_out)
{
	float& eA = s.e[K % 3];        // E[v-3]   (overwritten by E[v] at the end)
	float& eB = s.e[(K + 1) % 3];  // E[v-2]
	float& eC = s.e[(K + 2) % 3];  // E[v-1]
	float& oA = s.o[K % 2];        // O[v-2]   (overwritten by O[v])
	float& hA = s.h[K % 3];        // HP[v-5]  (overwritten by HP[v-2])
	float& hB = s.h[(K + 1) % 3];  // HP[v-4]
	float& hC = s.h[(K + 2) % 3];  // HP[v-3]
	float H = lift_add<false>(oA, sum_p<KIND, +1>(eA, eB, eC, E), shift_p<KIND>());
	if constexpr (VEDGE)
	{
		const int u = v - 2;
		const bool beyond = ve.patch && (u >= T), before = ve.zero && (u < 0), first = ve.patch && !ve.zero && (u == 0);
		const float edge = ve.zero ? 0.0f : hC;  // HP[T] := HP[T-1]
		H = beyond ? edge : H;
		H = before ? 0.0f : H;
		hB = first ? H : hB, hC = first ? H : hC;  // HP[-2] = HP[-1] := HP[0]
	}
	lp_out = lift_add<false>(eA, sum_u<KIND, +1>(hA, hB, hC, H), shift_u<KIND>());
	hp_out = hC;
	eA = E, oA = O, hA = H;
}
template <int KIND, bool VEDGE, int K, typename V>
__device__ __forceinline__ void vstep_inverse_bf(VInv<V>& s, V LP, V HP, int v, const VEdgeBF& ve, int T, V& even_out, V& odd_out)
{
	constexpr bool NRW = std::is_same<V, int>::value;
	V& hA = s.h[K % 3];        // HP[v-3]  (overwritten by HP[v])
	V& hB = s.h[(K + 1) % 3];  // HP[v-2]
	V& hC = s.h[(K + 2) % 3];  // HP[v-1]
	V& eA = s.e[K % 3];        // E[v-4]   (overwritten by E[v-1])
	V& eB = s.e[(K + 1) % 3];  // E[v-3]
	V& eC = s.e[(K + 2) % 3];  // E[v-2]
	V Ev = lift_add<NRW>(s.l, sum_u<KIND, -1>(hA, hB, hC, HP), shift_u<KIND>());
	if constexpr (VEDGE)
	{
		const int re = v - 1, ro = v - 3;
		const bool beyond = ve.patch && (re >= T), before = ve.zero && (re < 0), first = ve.patch && (ro == 0);
		const V edge = ve.zero ? (V)0 : eC;  // E[T] := E[T-1]
		Ev = beyond ? edge : Ev;
		Ev = before ? (V)0 : Ev;
		const V lead = ve.zero ? (V)0 : eB;  // E[-1] := E[0]
		eA = first ? lead : eA;
	}
	even_out = eB;
	odd_out = lift_add<NRW>(hA, sum_p<KIND, -1>(eA, eB, eC, Ev), shift_p<KIND>());
	hA = HP, s.l = LP, eA = Ev;
}

// ---- CDF5/3 on a pipeline of its own depth -----------------------------------------------------------------------------
// The ring pipeline above is as deep as DD13/7 needs it (a row leaves three slots after it came in, six slots of every segment only
// fill it); CDF5/3 through it leaves half the taps unused.  Its own pipeline is one slot deep:
//   forward:  HP[v-1] = O[v-1] - trunc((E[v-1] + E[v]) / 2),  LP[v-1] = E[v-1] + trunc((HP[v-2] + HP[v-1]) / 4)
//   inverse:  E[v] = LP[v] - trunc((HP[v-1] + HP[v]) / 4),    O[v-1] = HP[v-1] + trunc((E[v-1] + E[v]) / 2)
// -- slot v finishes row v - 1, a segment of n rows takes n + 2 slots instead of n + 6, and a column keeps three (two) values
// instead of eight (seven).  LEAN_LAG is the number of slots between a row coming in and going out.
#ifndef AKO_CDF_SHALLOW
#define AKO_CDF_SHALLOW 1  // experiments: 0 = CDF5/3 through the ring pipeline, as before
#endif
template <int KIND>
constexpr int lean_lag()
{
	return (KIND == K_CDF53 && AKO_CDF_SHALLOW) ? 1 : 3;
}
struct VFwdC
{
	float e, o, h;  // E[v-1], O[v-1], HP[v-2]
};
template <typename V>
struct VInvC
{
	V h, e;  // HP[v-1], E[v-1]
};
// border rules as in vstep_forward_bf: the rows beyond the tile come in through map_index_bf (CLAMP: the nearest, REPEAT: wrapped,
// ZERO: zeroed by the caller); of the COMPUTED values only HP[-1] is ever asked for: CLAMP HP[-1] := HP[0], ZERO HP[-1] := 0
template <bool VEDGE>
__device__ __forceinline__ void vstep_forward_cdf(VFwdC& s, float E, float O, int v, const VEdgeBF& ve, int T, float& lp_out, float& hp_out)
{
	(void)T;
	float H = lift_add<false>(s.o, sum_p<K_CDF53, +1>(0.0f, s.e, E, 0.0f), shift_p<K_CDF53>());  // HP[v-1]
	float hp = s.h;                                                                             // HP[v-2]
	if constexpr (VEDGE)
	{
		const int u = v - 1;
		H = (ve.zero && u < 0) ? 0.0f : H;
		hp = (ve.patch && !ve.zero && u == 0) ? H : hp;
	}
	lp_out = lift_add<false>(s.e, sum_u<K_CDF53, +1>(0.0f, hp, H, 0.0f), shift_u<K_CDF53>());
	hp_out = H;
	s.e = E, s.o = O, s.h = H;
}
// ... and as in vstep_inverse_bf: of the computed values E[T] is asked for (CLAMP E[T] := E[T-1], ZERO E[T] := 0)
template <bool VEDGE, typename V>
__device__ __forceinline__ void vstep_inverse_cdf(VInvC<V>& s, V LP, V HP, int v, const VEdgeBF& ve, int T, V& even_out, V& odd_out)
{
	constexpr bool NRW = std::is_same<V, int>::value;
	V Ev = lift_add<NRW>(LP, sum_u<K_CDF53, -1>((V)0, s.h, HP, (V)0), shift_u<K_CDF53>());  // E[v]
	if constexpr (VEDGE)
	{
		const V edge = ve.zero ? (V)0 : s.e;
		Ev = (ve.patch && v >= T) ? edge : Ev;
		Ev = (ve.zero && v < 0) ? (V)0 : Ev;
	}
	even_out = s.e;                                                                                          // E[v-1]
	odd_out = lift_add<NRW>(s.h, sum_p<K_CDF53, -1>((V)0, s.e, Ev, (V)0), shift_p<K_CDF53>());              // O[v-1]
	s.h = HP, s.e = Ev;
}

// =====================================================================================================================
// Inverse.  Role 0 carries planes 0, 1 (Y, Co) and finishes the even pixel row of a slot, role 1 planes 2, 3 (Cg, alpha;
// RGB: Cg alone) and the odd row; the rows' other planes cross through LDS (one barrier per slot, double buffered).
// =====================================================================================================================
template <int KIND, int CH, int ROLE, bool HEDGE, bool VEDGE, bool ROWS = false>
__device__ __forceinline__ void inverse_u8_lean(const LevelParams& P, const StreamGeom& G, const UnitId& id, const LaneCols& lc, int lane,
                                                float* xb)
{
	static_assert(CH == 4 || CH == 3, "RGBA or RGB");
	constexpr int NP = (CH == 3 && ROLE == 1) ? 1 : 2;      // planes this wave carries
	constexpr int NP_HIS = (CH == 3 && ROLE == 0) ? 1 : 2;  // planes the partner sends
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

	// ROWS: strips laid over a whole row of tiles (lane_columns_row; level 0 of 512-pixel tiles): tile borders fall on any lane, the
	// border rule is the same tap substitution as at a strip's border; the resources are based at the image's stream / first
	// tile instance and the lane's byte offsets carry its own tile's stream offset and low-pass plane
	static_assert(!ROWS || HEDGE, "every strip over a row of tiles holds tile borders");
	const uint32_t lane_tile = ROWS ? min(id.tile + (uint32_t)lc.tile_in_pack, P.n_tiles - 1u) : 0u;  // per lane
	const uint64_t base_inst = ROWS ? (uint64_t)id.image * P.n_tiles : inst;
	const uint32_t lane_stream_b = ROWS ? (uint32_t)(P.tiles[lane_tile].stream_off * 2) : 0u;  // per lane
	const int16_t* tile_stream = P.stream + (uint64_t)id.image * P.stream_stride + (ROWS ? 0 : td.stream_off);
	const __amdgpu_buffer_rsrc_t rs_stream = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(tile_stream), 0, (int)0xFFFFFFFFu, RSRC_FLAGS);
	const int16_t* ll_root = P.ll_in_stream ? tile_stream : (P.src + base_inst * P.src_inst_stride);
	const __amdgpu_buffer_rsrc_t rs_ll = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(ll_root), 0, (int)0xFFFFFFFFu, RSRC_FLAGS);
	const uint32_t ll_pitch_b = (P.ll_in_stream ? (uint32_t)Tc : P.src_pitch) * 2u;
	const uint32_t sub_pitch_b = (uint32_t)Tc * 2u;
	const uint32_t nsub_b = (uint32_t)((uint64_t)Tc * Tr * 2);
	// (lanes beyond a tile border read their clamped / wrapped pair, lane_columns(); HEDGE patches them for every rule but REPEAT)
	const uint32_t lane_in_off = (uint32_t)lc.cs * 2u + lane_stream_b;
	const uint32_t lane_ll_in_off = !ROWS ? lane_in_off
	                                      : (uint32_t)lc.cs * 2u + (P.ll_in_stream ? lane_stream_b : lane_tile * (uint32_t)P.src_inst_stride * 2u);
	bool heads_differ = false;  // ROWS: the tiles of a wave carry one lift head each

	uint32_t ll_off[NP], grp_off[NP];
	float qf[NP];
#pragma unroll
	for (int p = 0; p < NP; p++)
	{
		const int pl = 2 * ROLE + p;
		// the decoder trusts the lift head (misc.c:266-272, lifting.c:114-116); one tile per wave: wave-uniform
		// ROWS: one head per tile.  An honest stream repeats the same value in every tile of a level and plane; where the tiles
		// of this wave disagree the wave gives up like on an input beyond its proof bound, and the exact kernel behind this one,
		// which de-quantizes per lane, does the level
		const int q_lane = (int)(tile_stream + (lane_stream_b >> 1))[P.grp_off[pl]];
		const int q = __builtin_amdgcn_readfirstlane(q_lane);
		if constexpr (ROWS)
			heads_differ = heads_differ || (q_lane != q);
		qf[p] = (q > 1) ? (float)q : 1.0f;  // lifting.c:30-40: multiply only when q > 1
		grp_off[p] = (uint32_t)((P.grp_off[pl] + 1) * 2);
		ll_off[p] = (uint32_t)((P.ll_in_stream ? P.lp_off[pl] : (uint64_t)pl * P.src_plane_stride) * 2);
	}

	uint8_t* img = P.img + (uint64_t)id.image * P.img_stride + ((uint64_t)td.y0 * P.img_pitch + td.x0) * CH;
	const uint32_t out_pitch_b = P.img_pitch * (uint32_t)CH;
	const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, (int)0xFFFFFFFFu, RSRC_FLAGS);
	const bool store_lane = lc.net && (lc.c0 >= 0) && (lc.c0 < Tc);
	const uint32_t px_lane_off = store_lane ? (uint32_t)(2 * (ROWS ? (lc.xs >> 1) : lc.c0)) * (uint32_t)CH : OOB;  // (ROWS: the column inside the row of tiles)
	const float ysc = (P.color == C_YCOCG_Q) ? 0.5f : 1.0f;  // format.c:170: y = in / 2 first

	constexpr int LAG = lean_lag<KIND>();  // slots between a quadrant row coming in and its two sample rows going out
	std::conditional_t<LAG == 1, VInvC<float>, VInv<float>> st[NP][4];
#pragma unroll
	for (int p = 0; p < NP; p++)
#pragma unroll
		for (int k = 0; k < 4; k++)
		{
			if constexpr (LAG == 1)
				st[p][k] = VInvC<float>{0, 0};
			else
				st[p][k] = VInv<float>{{0, 0, 0}, {0, 0, 0}, 0};
		}
	float peak_in = 0.0f;
	const HEdgeBF he = hedge_bf(lc.he);
	const VEdgeBF ve = {wrap != W_REPEAT, wrap == W_ZERO};
	AKO_STAMP_DECL;  // phases: 0 wait for the slot's coefficients, 1 unpack + column pass, 2 row pass, 3 LDS writes + barrier,
	                 // 4 LDS reads, 5 colour + pack, 6 store, 7 the first trip (pipeline fill)

	struct Raw
	{
		uint32_t ll[NP], c[NP], b[NP], d[NP];
	};
	auto fetch = [&](int v, Raw& raw) {
		const int m = VEDGE ? map_index_bf(v, Tr, wrap) : v;  // (ZERO's rows beyond the border: any row, zeroed in column_pass)
		const uint32_t row_g = (uint32_t)m * sub_pitch_b, row_l = (uint32_t)m * ll_pitch_b;
#pragma unroll
		for (int p = 0; p < NP; p++)
		{
			const uint32_t g = grp_off[p] + row_g;
			raw.ll[p] = __builtin_amdgcn_raw_buffer_load_b32(rs_ll, lane_ll_in_off, ll_off[p] + row_l, 0);
			raw.c[p] = __builtin_amdgcn_raw_buffer_load_b32(rs_stream, lane_in_off, g, AUX_INV_STREAM_LOAD);
			raw.b[p] = __builtin_amdgcn_raw_buffer_load_b32(rs_stream, lane_in_off, g + nsub_b, AUX_INV_STREAM_LOAD);
			raw.d[p] = __builtin_amdgcn_raw_buffer_load_b32(rs_stream, lane_in_off, g + 2u * nsub_b, AUX_INV_STREAM_LOAD);
		}
	};

	// unpack + de-quantize + column pass of one slot: even / odd sample rows of slot v - 3 as [plane][row low-pass c0 c1, row high-pass c0 c1]
	auto column_pass = [&](auto kc, const int v, const Raw& raw, float (&ev)[NP][4], float (&od)[NP][4]) {
		constexpr int K = decltype(kc)::value;
		const bool zero_row = VEDGE && ve.zero && ((unsigned)v >= (unsigned)Tr);  // wave-uniform
#pragma unroll
		for (int p = 0; p < NP; p++)
		{
			float lpv[4], hpv[4];  // columns: 0,1 = row low-pass (LL over C); 2,3 = row high-pass (B over D)
			unpack2_f(raw.ll[p], lpv[0], lpv[1]);
			unpack2_f(raw.b[p], lpv[2], lpv[3]);
			unpack2_f(raw.c[p], hpv[0], hpv[1]);
			unpack2_f(raw.d[p], hpv[2], hpv[3]);
			lpv[2] *= qf[p], lpv[3] *= qf[p];
#pragma unroll
			for (int k = 0; k < 4; k++)
				hpv[k] *= qf[p];
#pragma unroll
			for (int k = 0; k < 4; k += 2)
			{
				absmax3(peak_in, lpv[k], lpv[k + 1]);
				absmax3(peak_in, hpv[k], hpv[k + 1]);
			}
			if constexpr (VEDGE)
#pragma unroll
				for (int k = 0; k < 4; k++)
					lpv[k] = zero_row ? 0.0f : lpv[k], hpv[k] = zero_row ? 0.0f : hpv[k];
#pragma unroll
			for (int k = 0; k < 4; k++)
			{
				if constexpr (LAG == 1)
					vstep_inverse_cdf<VEDGE, float>(st[p][k], lpv[k], hpv[k], v, ve, Tr, ev[p][k], od[p][k]);
				else
					vstep_inverse_bf<KIND, VEDGE, K, float>(st[p][k], lpv[k], hpv[k], v, ve, Tr, ev[p][k], od[p][k]);
			}
		}
	};

	float* const xw = xb + (1 - ROLE) * XI_ROLE + lane;  // what this wave sends: the partner's row
	const float* const xr = xb + ROLE * XI_ROLE + lane;  // what it receives: the partner's planes of its own row
	auto full_slot = [&](auto kc, const int v, const Raw& raw) {
		constexpr int K = decltype(kc)::value;
		float ev[NP][4], od[NP][4];
		AKO_STAMP_SLOT_BEGIN(K);
		AKO_WAIT_VM(4 * NP * AKO_U8L_INV_PF + 1, K);  // (stamps: everything but the slots fetched ahead and the last pixel store)
		AKO_STAMP(0);
		column_pass(kc, v, raw, ev, od);
#ifdef AKO_STAMPS_PHASES
		for (int p = 0; p < NP; p++)
		{
			AKO_PIN4(ev[p]);
			AKO_PIN4(od[p]);
		}
#endif
		AKO_STAMP(1);
		// row pass: this wave finishes pixel row ROLE of the slot (role 0 the even one), the other row's planes go to the partner
		float mine[NP][4], send[NP][4];
#pragma unroll
		for (int p = 0; p < NP; p++)
		{
			float (&me)[4] = mine[p];
			float (&sd)[4] = send[p];
			if constexpr (ROLE == 0)
			{
				hlift_inverse_bf<KIND, HEDGE, float>(ev[p][0], ev[p][1], ev[p][2], ev[p][3], he, me[0], me[1], me[2], me[3]);
				hlift_inverse_bf<KIND, HEDGE, float>(od[p][0], od[p][1], od[p][2], od[p][3], he, sd[0], sd[1], sd[2], sd[3]);
			}
			else
			{
				hlift_inverse_bf<KIND, HEDGE, float>(ev[p][0], ev[p][1], ev[p][2], ev[p][3], he, sd[0], sd[1], sd[2], sd[3]);
				hlift_inverse_bf<KIND, HEDGE, float>(od[p][0], od[p][1], od[p][2], od[p][3], he, me[0], me[1], me[2], me[3]);
			}
		}
#ifdef AKO_STAMPS_PHASES
		for (int p = 0; p < NP; p++)
		{
			AKO_PIN4(mine[p]);
			AKO_PIN4(send[p]);
		}
#endif
		AKO_STAMP(2);
#pragma unroll
		for (int p = 0; p < NP; p++)
#pragma unroll
			for (int k = 0; k < 4; k++)
				xw[(K & 1) * XI_BUF + p * XI_PLANE + k * 64] = send[p][k];
		__syncthreads();  // (one barrier per slot, double buffered: see inverse_stream_body)
		AKO_STAMP(3);
		float his[2][4] = {};
#pragma unroll
		for (int p = 0; p < NP_HIS; p++)
#pragma unroll
			for (int k = 0; k < 4; k++)
				his[p][k] = xr[(K & 1) * XI_BUF + p * XI_PLANE + k * 64];
#ifdef AKO_STAMPS_PHASES
		for (int p = 0; p < NP_HIS; p++)
			AKO_PIN4(his[p]);
#endif
		AKO_STAMP(4);

		const int r = v - LAG, y = 2 * r + ROLE;
		const bool row_ok = (r >= r_lo) && (r < r_hi) && (y < oh);  // wave-uniform; phantom last row dropped (lifting.c:112,141)
		uint32_t px[4];
#pragma unroll
		for (int k = 0; k < 4; k++)
		{
			// format.c:138-191: t = y - v / 2; g = v + t; b = t - u / 2; r = b + u  (planes: y, u = Co, v = Cg)
			float yy, co, cg, al = 0.0f;
			if constexpr (ROLE == 0)
			{
				yy = mine[0][k], co = mine[1][k], cg = his[0][k];
				if constexpr (CH == 4)
					al = his[1][k];
			}
			else
			{
				yy = his[0][k], co = his[1][k], cg = mine[0][k];
				if constexpr (CH == 4)
					al = mine[1][k];
			}
			const float t = __builtin_truncf(yy * ysc) - half_trunc(cg);
			const float gg = cg + t;
			const float bb = t - half_trunc(co);
			const float rr = bb + co;
			px[k] = pixel_u8x4(rr, gg, bb, al);
		}
		AKO_PIN4(px);
		AKO_STAMP(5);
		const uint32_t s_row = row_ok ? (uint32_t)y * out_pitch_b : OOB;
		typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
		if constexpr (CH == 3)
			__builtin_amdgcn_raw_buffer_store_b96(rgb_pack4(px), rs_img, px_lane_off, s_row, AUX_INV_PIXEL_STORE);
		else
			__builtin_amdgcn_raw_buffer_store_b128(u32x4{px[0], px[1], px[2], px[3]}, rs_img, px_lane_off, s_row, AUX_INV_PIXEL_STORE);
		AKO_STORE_GUARD();  // (see store_b128_guarded)
		AKO_STAMP(6);
		AKO_STAMP_SLOT_END(K, 0, 6);
	};

	// slots v_begin .. r_hi + 2; slot v consumes quadrant row v and finishes sample rows 2 (v - 3), 2 (v - 3) + 1.  The first six
	// slots finish rows r_lo - 6 .. r_lo - 1, which belong to the segment above (or do not exist): they only fill the pipeline.
	const int v_begin = r_lo - LAG;
	const int n_slots = r_hi + LAG - v_begin;
	// PF row slots are fetched ahead of the one being worked on (the rows exist: segment_needs_border_code() keeps twelve
	// rows between a body without row border code and the bottom border; VEDGE maps every row into the tile)
	constexpr int PF = AKO_U8L_INV_PF;
	static_assert(PF == 1 || PF == 2, "the ring index must repeat with the unrolled row loop");
	Raw ring[PF + 1];
	static_for<PF>([&](auto jc) { fetch(v_begin + decltype(jc)::value, ring[decltype(jc)::value]); });
	// (the one-slot pipeline of CDF5/3 has no trip that only fills it: its first trip is a trip like every other, whose first two
	// slots finish rows above the segment and drop them)
	if constexpr (LAG == 3)
	static_for<6>([&](auto kc) {
		constexpr int K = decltype(kc)::value;
		const int v = v_begin + K;
		fetch(v + PF, ring[(K + PF) % (PF + 1)]);
		__builtin_amdgcn_sched_barrier(0);  // (the compiler would sink the loads to where their registers are free: half the prefetch distance)
		float ev[NP][4], od[NP][4];
		column_pass(kc, v, ring[K % (PF + 1)], ev, od);
		(void)ev, (void)od;
		__builtin_amdgcn_sched_barrier(0);
		// as many (dropped) stores behind the loads as a full slot issues: the memory operations in flight then look the same on
		// entry to the loop below as on every later trip, and the wait in front of a slot's coefficients stays a counted one
		if constexpr (K == 5)
			__builtin_amdgcn_raw_buffer_store_b32(0u, rs_img, OOB, 0, 0);
	});
	AKO_STAMP_NOW(7);
	for (int base = (LAG == 3) ? 6 : 0; base < n_slots; base += 6)
	{
		if constexpr (KIND == K_DD137 && CH == 4)  // (scripts/isa_lint.py finds the loop by this comment)
		{
			if constexpr (ROWS)
				asm volatile("; AKO_LOOP inv_u8_rows_role%0_h%1_v%2" ::"n"(ROLE), "n"((int)HEDGE), "n"((int)VEDGE));
			else
				asm volatile("; AKO_LOOP inv_u8_lean_role%0_h%1_v%2" ::"n"(ROLE), "n"((int)HEDGE), "n"((int)VEDGE));
		}
		static_for<6>([&](auto kc) {
			constexpr int K = decltype(kc)::value;
			const int v = v_begin + base + K;
			fetch(v + PF, ring[(K + PF) % (PF + 1)]);
			__builtin_amdgcn_sched_barrier(0);
			full_slot(kc, v, ring[K % (PF + 1)]);
		});
	}
	const bool bad = !(peak_in <= OPT_INPUT_BOUND_TIGHT) || heads_differ;  // negated: NaN counts as bad
	if (__any(bad) && lane == 0)
		atomicMax(P.ovf_flag, P.ovf_gen);
	AKO_STAMP_FLUSH(1);
}

// the workgroup is 1, 2 or 4 pairs of waves (neighbouring strips); 8 KiB of dynamic LDS per pair, as k_inverse_stream_u8
template <int KIND, int CH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_inverse_u8_lean(const LevelParams P, const StreamGeom G)
{
	extern __shared__ float xlean[];
	float* xb = xlean + (threadIdx.x >> 7) * (2 * XI_BUF);
#if defined(AKO_STAMPS) && AKO_STAMPS == 2
	AKO_STAMP_TOP;
	UnitId id = decode_unit(P, G);
	AKO_STAMP_TOP_SET(id);
#else
	const UnitId id = decode_unit(P, G);
#endif
	if (!id.valid)
		return;  // units come in pairs, so both waves of a pair leave together (a barrier does not wait for ended waves)
	const int lane = threadIdx.x & 63;
	const LaneCols lc = lane_columns(id.strip, G.strips, G.wide == 1u, lane, (int)P.sub_w, (int)P.full_w, P.wrap);
	bool vedge = segment_needs_border_code(G, id.seg, (int)P.sub_h);
	bool hedge_ = lc.hedge;
#if defined(AKO_STAMPS) && AKO_STAMPS == 2
	asm volatile("" ::"v"(lc.c0), "v"(lc.xs));
	AKO_STAMP_LC_SET(id);
#endif
#ifdef AKO_EXP_NOHEDGE  // experiments (timing only, wrong pixels at the borders): border strips / segments run the interior bodies
	hedge_ = false;
#endif
	// (strip and segment are the pair's: both its waves take the same way, and execute the same sequence of barriers)
#define AKO_INV_LEAN(H, V)                                                       \
	do                                                                           \
	{                                                                            \
		if (id.pg == 0)                                                          \
			inverse_u8_lean<KIND, CH, 0, H, V>(P, G, id, lc, lane, xb);          \
		else                                                                     \
			inverse_u8_lean<KIND, CH, 1, H, V>(P, G, id, lc, lane, xb);          \
	} while (0)
	if (__builtin_expect(vedge, 0))
	{
		if (hedge_)
			AKO_INV_LEAN(true, true);
		else
			AKO_INV_LEAN(false, true);
	}
	else
	{
		if (hedge_)
			AKO_INV_LEAN(true, false);
		else
			AKO_INV_LEAN(false, false);
	}
#undef AKO_INV_LEAN
}

// ROWS: level 0 in strips over whole rows of tiles (StreamGeom::wide bit 31, lane_columns_row) on the lean bodies.  Kernels of
// their own: what these bodies need beyond the single-tile ones (two per-lane offsets) must not cost the lean kernels a register
// -- or, worse, a private segment.
template <int KIND, int CH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_inverse_u8_rows(const LevelParams P, const StreamGeom G)
{
	extern __shared__ float xlean[];
	float* xb = xlean + (threadIdx.x >> 7) * (2 * XI_BUF);
	const UnitId id = decode_unit(P, G);
	if (!id.valid)
		return;
	const int lane = threadIdx.x & 63;
	const LaneCols lc = lane_columns_row(id.strip, lane, (int)P.sub_w, (int)geom_row_tiles(G), P.wrap);
	const bool vedge = segment_needs_border_code(G, id.seg, (int)P.sub_h);
	if (id.pg == 0)
	{
		if (vedge)
			inverse_u8_lean<KIND, CH, 0, true, true, true>(P, G, id, lc, lane, xb);
		else
			inverse_u8_lean<KIND, CH, 0, true, false, true>(P, G, id, lc, lane, xb);
	}
	else
	{
		if (vedge)
			inverse_u8_lean<KIND, CH, 1, true, true, true>(P, G, id, lc, lane, xb);
		else
			inverse_u8_lean<KIND, CH, 1, true, false, true>(P, G, id, lc, lane, xb);
	}
}

// =====================================================================================================================
// Forward.  Role 0 carries planes 0, 2 (Y, Cg: they share t = b + Co / 2), role 1 planes 1, 3 (Co, alpha; RGB: Co alone);
// both waves of a pair load the same pixels.
// =====================================================================================================================
template <int KIND, int CH, int ROLE, bool HEDGE, bool VEDGE, bool ROWS = false>
__device__ __forceinline__ void forward_u8_lean(const LevelParams& P, const StreamGeom& G, const UnitId& id, const LaneCols& lc, int lane)
{
	static_assert(CH == 4 || CH == 3, "RGBA or RGB");
	constexpr int NP = (CH == 3 && ROLE == 1) ? 1 : 2;
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

	// source: the tile's pixels through a raw buffer resource, the lane's four pixels ONE register of byte offset
	const uint8_t* src_base = P.img + (uint64_t)id.image * P.img_stride + ((uint64_t)td.y0 * P.img_pitch + td.x0) * CH;
	const uint32_t row_pitch_b = P.img_pitch * (uint32_t)CH;
	const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src_base), 0, (int)0xFFFFFFFFu, RSRC_FLAGS);
	const uint32_t src_lane_off = (uint32_t)lc.xs * (uint32_t)CH;

	// destinations (see forward_stream_body: out-of-range offsets drop the stores of lanes / rows that must not store)
	// ROWS (strips over a whole row of tiles, see inverse_u8_lean): based at the image's stream / first tile instance, the lane's
	// offsets carry its own tile's stream offset and low-pass plane
	static_assert(!ROWS || HEDGE, "every strip over a row of tiles holds tile borders");
	const uint32_t lane_tile = ROWS ? min(id.tile + (uint32_t)lc.tile_in_pack, P.n_tiles - 1u) : 0u;  // per lane
	const uint64_t base_inst = ROWS ? (uint64_t)id.image * P.n_tiles : inst;
	const uint32_t lane_stream_b = ROWS ? (uint32_t)(P.tiles[lane_tile].stream_off * 2) : 0u;  // per lane
	const uint64_t tile_off = ROWS ? 0 : td.stream_off;
	int16_t* tile_stream = P.stream + (uint64_t)id.image * P.stream_stride + tile_off;
	const uint64_t stream_left = (P.stream_stride - tile_off) * 2;  // bytes up to the end of the image's stream
	const __amdgpu_buffer_rsrc_t rs_stream = __builtin_amdgcn_make_buffer_rsrc(
	    tile_stream, 0, (int)(uint32_t)(stream_left < 0xFFFFFFFFull ? stream_left : 0xFFFFFFFFull), RSRC_FLAGS);
	int16_t* ll_root = P.ll_out_stream ? tile_stream : (P.dst + base_inst * P.dst_inst_stride);
	const uint64_t ll_left = P.ll_out_stream ? stream_left : (uint64_t)P.channels * P.dst_plane_stride * 2 * (ROWS ? P.n_tiles : 1u);
	const __amdgpu_buffer_rsrc_t rs_ll = __builtin_amdgcn_make_buffer_rsrc(
	    ll_root, 0, (int)(uint32_t)(ll_left < 0xFFFFFFFFull ? ll_left : 0xFFFFFFFFull), RSRC_FLAGS);
	const uint32_t ll_pitch_b = (P.ll_out_stream ? (uint32_t)Tc : P.dst_pitch) * 2u;
	const uint32_t sub_pitch_b = (uint32_t)Tc * 2u;
	const uint32_t nsub_b = (uint32_t)((uint64_t)Tc * Tr * 2);
	const bool store_lane = lc.net && (lc.c0 >= 0) && (lc.c0 < Tc);
	const uint32_t lane_off = store_lane ? (uint32_t)(lc.c0 * 2) + lane_stream_b : OOB;
	const uint32_t lane_ll_off = !ROWS ? lane_off
	                                   : (store_lane ? (uint32_t)(lc.c0 * 2) + (P.ll_out_stream ? lane_stream_b : lane_tile * (uint32_t)P.dst_inst_stride * 2u) : OOB);
	uint32_t ll_off[NP], grp_off[NP];
	float gf[NP], rq[NP];
#pragma unroll
	for (int p = 0; p < NP; p++)
	{
		const int pl = ROLE + 2 * p;
		grp_off[p] = (uint32_t)((P.grp_off[pl] + 1) * 2);
		ll_off[p] = (uint32_t)((P.ll_out_stream ? P.lp_off[pl] : (uint64_t)pl * P.dst_plane_stride) * 2);
		gf[p] = (float)((pl == 0) ? P.g_luma : P.g_chroma);  // lifting.c:202-211: every plane but the first is "chroma"
		rq[p] = (pl == 0) ? P.rq_luma : P.rq_chroma;
		if (ROWS ? (id.seg == 0 && lc.net && lc.he.first) : (id.strip == 0 && id.seg == 0 && lane == 0))  // the lift head of every tile (lifting.c:266-267)
			(tile_stream + (lane_stream_b >> 1))[P.grp_off[pl]] = (int16_t)((pl == 0) ? P.q_luma : P.q_chroma);
	}
	const float ymul = (P.color == C_YCOCG_Q) ? 2.0f : 1.0f;

	constexpr int LAG = lean_lag<KIND>();  // slots between a pixel row pair coming in and its sub-band row going out
	std::conditional_t<LAG == 1, VFwdC, VFwd<float>> st[NP][4];
#pragma unroll
	for (int p = 0; p < NP; p++)
#pragma unroll
		for (int k = 0; k < 4; k++)
		{
			if constexpr (LAG == 1)
				st[p][k] = VFwdC{0, 0, 0};
			else
				st[p][k] = VFwd<float>{{0, 0, 0}, {0, 0}, {0, 0, 0}};
		}
	const HEdgeBF he = hedge_bf(lc.he);
	const VEdgeBF ve = {wrap != W_REPEAT, wrap == W_ZERO};
	AKO_STAMP_DECL;  // phases: 0 wait for the slot's pixels, 1 pixels -> samples, 2 row pass, 3 column pass, 4 gate + quantizer + pack,
	                 // 5 stores, 6 the barrier of a trip, 7 the first trip (pipeline fill)

	struct Raw
	{
		uint4 a[2];  // the two pixel rows of a slot
	};
	auto fetch = [&](int v, Raw& raw) {
		const int m = VEDGE ? map_index_bf(v, Tr, wrap) : v;
#pragma unroll
		for (int par = 0; par < 2; par++)
		{
			const int y = VEDGE ? min(2 * m + par, chh - 1) : (2 * m + par);  // phantom last row = copy of the last row
			const uint32_t row_off = (uint32_t)y * row_pitch_b;
			if constexpr (CH == 3)
			{
				typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
				const u32x3 t = __builtin_bit_cast(u32x3, __builtin_amdgcn_raw_buffer_load_b96(rs_src, src_lane_off, row_off, AUX_FWD_PIXEL_LOAD));
				raw.a[par] = uint4{t.x, t.y, t.z, 0u};
			}
			else
				raw.a[par] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, src_lane_off, row_off, AUX_FWD_PIXEL_LOAD));
		}
	};

	// pixels -> samples -> row pass -> column pass of one slot: low-pass / high-pass rows of slot v - 3, [plane][row LP c0 c1, row HP c0 c1]
	auto lift_slot = [&](auto kc, const int v, Raw& raw, float (&lp)[NP][4], float (&hp)[NP][4]) {
		constexpr int K = decltype(kc)::value;
		const bool zero_row = VEDGE && ve.zero && ((unsigned)v >= (unsigned)Tr);
		float smp[2][2][4];
		AKO_STAMP_SLOT_BEGIN(K);
		AKO_WAIT_VM(4 * NP, K);  // (stamps: everything but the previous slot's stores)
		AKO_STAMP(0);
#pragma unroll
		for (int par = 0; par < 2; par++)
		{
			uint32_t px[4] = {raw.a[par].x, raw.a[par].y, raw.a[par].z, raw.a[par].w};
			if constexpr (CH == 3)
			{
				// twelve bytes = four RGB pixels: each into the low three bytes of a dword
				px[3] = raw.a[par].z >> 8;
				px[2] = __builtin_amdgcn_alignbit(raw.a[par].z, raw.a[par].y, 16);
				px[1] = __builtin_amdgcn_alignbit(raw.a[par].y, raw.a[par].x, 24);
			}
			decode_pixels_ycocg<float>(px, ymul, ROLE, smp[par][0], smp[par][1]);
			if constexpr (VEDGE)
#pragma unroll
				for (int k = 0; k < 4; k++)
					smp[par][0][k] = zero_row ? 0.0f : smp[par][0][k], smp[par][1][k] = zero_row ? 0.0f : smp[par][1][k];
		}
#ifdef AKO_STAMPS_PHASES
		for (int p = 0; p < NP; p++)
		{
			AKO_PIN4(smp[0][p]);
			AKO_PIN4(smp[1][p]);
		}
#endif
		AKO_STAMP(1);
		// the slot's pixels are samples now: the next slot's go into the registers they left
		__builtin_amdgcn_sched_barrier(0);
		fetch(v + 1, raw);
		__builtin_amdgcn_sched_barrier(0);
		float e[NP][4], o[NP][4];  // columns: 0,1 = row low-pass of c0, c1; 2,3 = row high-pass of c0, c1
#pragma unroll
		for (int p = 0; p < NP; p++)
		{
			hlift_forward_bf<KIND, HEDGE, float>(smp[0][p][0], smp[0][p][1], smp[0][p][2], smp[0][p][3], he, e[p][0], e[p][1], e[p][2], e[p][3]);
			hlift_forward_bf<KIND, HEDGE, float>(smp[1][p][0], smp[1][p][1], smp[1][p][2], smp[1][p][3], he, o[p][0], o[p][1], o[p][2], o[p][3]);
#ifndef AKO_STAMPS_PHASES
#pragma unroll
			for (int k = 0; k < 4; k++)
			{
				if constexpr (LAG == 1)
					vstep_forward_cdf<VEDGE>(st[p][k], e[p][k], o[p][k], v, ve, Tr, lp[p][k], hp[p][k]);
				else
					vstep_forward_bf<KIND, VEDGE, K>(st[p][k], e[p][k], o[p][k], v, ve, Tr, lp[p][k], hp[p][k]);
			}
#endif
		}
#ifdef AKO_STAMPS_PHASES
		for (int p = 0; p < NP; p++)
		{
			AKO_PIN4(e[p]);
			AKO_PIN4(o[p]);
		}
		AKO_STAMP(2);
		for (int p = 0; p < NP; p++)
			for (int k = 0; k < 4; k++)
			{
				if constexpr (LAG == 1)
					vstep_forward_cdf<VEDGE>(st[p][k], e[p][k], o[p][k], v, ve, Tr, lp[p][k], hp[p][k]);
				else
					vstep_forward_bf<KIND, VEDGE, K>(st[p][k], e[p][k], o[p][k], v, ve, Tr, lp[p][k], hp[p][k]);
			}
		for (int p = 0; p < NP; p++)
		{
			AKO_PIN4(lp[p]);
			AKO_PIN4(hp[p]);
		}
		AKO_STAMP(3);
#endif
	};
	auto full_slot = [&](auto kc, const int v, Raw& raw) {
		constexpr int K = decltype(kc)::value;
		(void)K;
		float lp[NP][4], hp[NP][4];
		lift_slot(kc, v, raw, lp, hp);
		const int r = v - LAG;
		const bool row_ok = (r >= r_lo) && (r < r_hi);  // wave-uniform
		const uint32_t row_grp = (uint32_t)r * sub_pitch_b, row_ll = (uint32_t)r * ll_pitch_b;
#pragma unroll
		for (int p = 0; p < NP; p++)
		{
			// LL = (LP rows, LP cols), C = (HP rows, LP cols), B = (LP rows, HP cols), D = (HP, HP); gate + quantizer: lifting.c:154-168
			uint32_t w_ll, w_c, w_b, w_d;
			pack_row_f(lp[p], hp[p], gf[p], rq[p], w_ll, w_c, w_b, w_d);
#ifdef AKO_STAMPS_PHASES
			asm volatile("" ::"v"(w_ll), "v"(w_c), "v"(w_b), "v"(w_d));
			if (p == NP - 1)
				AKO_STAMP(4);
#endif
			const uint32_t s_ll = row_ok ? ll_off[p] + row_ll : OOB;
			const uint32_t s_c = row_ok ? grp_off[p] + row_grp : OOB;
			const uint32_t s_b = row_ok ? grp_off[p] + row_grp + nsub_b : OOB;
			const uint32_t s_d = row_ok ? grp_off[p] + row_grp + 2u * nsub_b : OOB;
			__builtin_amdgcn_raw_buffer_store_b32(w_ll, rs_ll, lane_ll_off, s_ll, 0);
			__builtin_amdgcn_raw_buffer_store_b32(w_c, rs_stream, lane_off, s_c, AUX_FWD_STREAM_STORE);
			__builtin_amdgcn_raw_buffer_store_b32(w_b, rs_stream, lane_off, s_b, AUX_FWD_STREAM_STORE);
			__builtin_amdgcn_raw_buffer_store_b32(w_d, rs_stream, lane_off, s_d, AUX_FWD_STREAM_STORE);
		}
		AKO_STAMP(5);
		AKO_STAMP_SLOT_END(K, 0, 5);
	};
	// (dropped) stores: as many as a full slot issues, so that the memory operations in flight look the same on entry to the
	// main loop as on every later trip (stores count in vmcnt on gfx950)
	auto phantom_stores = [&]() {
#pragma unroll
		for (int k = 0; k < 4 * NP; k++)
			__builtin_amdgcn_raw_buffer_store_b32(0u, rs_stream, OOB, 0, 0);
	};

	// slots v_begin .. r_hi + 2; slot v consumes pixel rows 2 v, 2 v + 1 and finishes sub-band row v - 3.  The first six slots
	// finish rows r_lo - 6 .. r_lo - 1, which belong to the segment above (or do not exist): no gate, quantizer or stores.
	const int v_begin = r_lo - LAG;
	const int n_slots = r_hi + LAG - v_begin;
	Raw ring;
	fetch(v_begin, ring);
	// lockstep: the waves of a workgroup (neighbouring strips, the pair even the same pixels) meet every six slots, so that
	// what one brought into L2 is still there when its neighbour asks for it (StreamGeom::lockstep; always on here: a branch
	// around the barrier makes the block behind it a place to sink the first trip's arithmetic into, through scratch)
	// (the one-slot pipeline of CDF5/3 has no trip that only fills it: its first trip is a trip like every other, whose first two
	// slots finish rows above the segment and drop them)
	if constexpr (LAG == 3)
	{
	if constexpr (AKO_U8L_LOCKSTEP != 0)
		__builtin_amdgcn_s_barrier();
	static_for<6>([&](auto kc) {
		constexpr int K = decltype(kc)::value;
		float lp[NP][4], hp[NP][4];
		lift_slot(kc, v_begin + K, ring, lp, hp);
		(void)lp, (void)hp;
		// Slot by slot: with nothing to store, nothing ties a slot's arithmetic to its place -- instruction selection then
		// computes each value where it is first used, slots later, and parks what it needs until then in scratch.  The column
		// pipeline's new entries are "used" here, the low-pass update that nobody reads stays dead.
#pragma unroll
		for (int p = 0; p < NP; p++)
			asm volatile("" ::"v"(st[p][0].e[K % 3]), "v"(st[p][1].e[K % 3]), "v"(st[p][2].e[K % 3]), "v"(st[p][3].e[K % 3]),
			             "v"(st[p][0].o[K % 2]), "v"(st[p][1].o[K % 2]), "v"(st[p][2].o[K % 2]), "v"(st[p][3].o[K % 2]),
			             "v"(st[p][0].h[K % 3]), "v"(st[p][1].h[K % 3]), "v"(st[p][2].h[K % 3]), "v"(st[p][3].h[K % 3]));
		__builtin_amdgcn_sched_barrier(0);
		if constexpr (K == 5)
			phantom_stores();
	});
	}
	for (int base = (LAG == 3) ? 6 : 0; base < n_slots; base += 6)
	{
		if constexpr (KIND == K_DD137 && CH == 4)  // (scripts/isa_lint.py finds the loop by this comment)
		{
			if constexpr (ROWS)
				asm volatile("; AKO_LOOP fwd_u8_rows_role%0_h%1_v%2" ::"n"(ROLE), "n"((int)HEDGE), "n"((int)VEDGE));
			else
				asm volatile("; AKO_LOOP fwd_u8_lean_role%0_h%1_v%2" ::"n"(ROLE), "n"((int)HEDGE), "n"((int)VEDGE));
		}
		if (base == 6)
			AKO_STAMP_NOW(7);
		AKO_STAMP_SLOT_BEGIN(0);
		if constexpr (AKO_U8L_LOCKSTEP != 0)
			__builtin_amdgcn_s_barrier();  // (lockstep, see above)
		AKO_STAMP_NOW(6);
		static_for<6>([&](auto kc) { full_slot(kc, v_begin + base + decltype(kc)::value, ring); });
	}
	AKO_STAMP_FLUSH(0);
}

template <int KIND, int CH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_forward_u8_lean(const LevelParams P, const StreamGeom G)
{
#if defined(AKO_STAMPS) && AKO_STAMPS == 2
	AKO_STAMP_TOP;
	UnitId id = decode_unit(P, G);
	AKO_STAMP_TOP_SET(id);
#else
	const UnitId id = decode_unit(P, G);
#endif
	if (!id.valid)
		return;
	const int lane = threadIdx.x & 63;
	const LaneCols lc = lane_columns(id.strip, G.strips, G.wide == 1u, lane, (int)P.sub_w, (int)P.full_w, P.wrap);
	bool vedge = segment_needs_border_code(G, id.seg, (int)P.sub_h);
	bool hedge_ = lc.hedge;
#if defined(AKO_STAMPS) && AKO_STAMPS == 2
	asm volatile("" ::"v"(lc.c0), "v"(lc.xs));
	AKO_STAMP_LC_SET(id);
#endif
#ifdef AKO_EXP_NOHEDGE  // experiments (timing only, wrong pixels at the borders): border strips / segments run the interior bodies
	hedge_ = false;
#endif
#define AKO_FWD_LEAN(H, V)                                                   \
	do                                                                       \
	{                                                                        \
		if (id.pg == 0)                                                      \
			forward_u8_lean<KIND, CH, 0, H, V>(P, G, id, lc, lane);          \
		else                                                                 \
			forward_u8_lean<KIND, CH, 1, H, V>(P, G, id, lc, lane);          \
	} while (0)
	if (__builtin_expect(vedge, 0))
	{
		if (hedge_)
			AKO_FWD_LEAN(true, true);
		else
			AKO_FWD_LEAN(false, true);
	}
	else
	{
		if (hedge_)
			AKO_FWD_LEAN(true, false);
		else
			AKO_FWD_LEAN(false, false);
	}
#undef AKO_FWD_LEAN
}

template <int KIND, int CH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_forward_u8_rows(const LevelParams P, const StreamGeom G)
{
	const UnitId id = decode_unit(P, G);
	if (!id.valid)
		return;
	const int lane = threadIdx.x & 63;
	const LaneCols lc = lane_columns_row(id.strip, lane, (int)P.sub_w, (int)geom_row_tiles(G), P.wrap);
	const bool vedge = segment_needs_border_code(G, id.seg, (int)P.sub_h);
	if (id.pg == 0)
	{
		if (vedge)
			forward_u8_lean<KIND, CH, 0, true, true, true>(P, G, id, lc, lane);
		else
			forward_u8_lean<KIND, CH, 0, true, false, true>(P, G, id, lc, lane);
	}
	else
	{
		if (vedge)
			forward_u8_lean<KIND, CH, 1, true, true, true>(P, G, id, lc, lane);
		else
			forward_u8_lean<KIND, CH, 1, true, false, true>(P, G, id, lc, lane);
	}
}

}  // namespace ako
