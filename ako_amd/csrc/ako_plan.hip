// ako_plan.hip -- host side of the device path: plans, launch sequencing, C-ABI (include/ako_hip.h).
//
// The level loop that the reference runs per tile on the CPU (library/lifting.c:182-291 forward,
// library/misc.c:229-288 inverse) becomes, per GROUP of equally sized tiles, one kernel launch per
// level covering every tile of every image of the batch and every plane.  Geometry, stream offsets
// and the float quantizer / gate scalars (library/quantization.c:43-98) are prepared once per plan
// on the host.
#include "ako_kernels.hip.h"
#include "ako_stream.hip.h"
#include "ako_u8.h"
#include "ako_tail.hip.h"
#include "ako_kagari.hip.h"
#include "ako_requant.hip.h"
#include "ako_fused.h"

#include "../../include/ako_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

extern "C" {
// host C (ako_host.c): same float recipe as library/quantization.c
int16_t akoHostQuantStep(int factor, int mul, size_t tile_w, size_t tile_h, size_t cur_w, size_t cur_h);
int16_t akoHostGateStep(int factor, int mul, size_t tile_w, size_t tile_h, size_t cur_w, size_t cur_h);
}

using namespace ako;

namespace
{

thread_local std::string g_last_error;

int fail(enum akoStatus st, const char* fmt, const char* a = "", const char* b = "")
{
	char buf[512];
	snprintf(buf, sizeof buf, fmt, a, b);
	g_last_error = buf;
	return (int)st;
}

#define HIP_TRY(expr)                                                                   \
	do                                                                                  \
	{                                                                                   \
		hipError_t e_ = (expr);                                                         \
		if (e_ != hipSuccess)                                                           \
			return fail(AKO_ERROR, "HIP error: %s at %s", hipGetErrorString(e_), #expr); \
	} while (0)

struct LevelGeom
{
	uint32_t cw, ch, tw, th;  // current (input) and target (sub-band) extents: lifting.c:184-187
	int kind;                 // wavelet after the DD137 -> CDF53 fallback: lifting.c:58
	int q[2], g[2];           // [0] plane 0, [1] every other plane: lifting.c:202-211
	uint64_t grp_off[MAX_CH];
};

struct Group
{
	uint32_t tile_w = 0, tile_h = 0;
	uint32_t fw = 0, fh = 0;  // final low-pass extent
	std::vector<LevelGeom> levels;
	uint64_t lp_off[MAX_CH];
	uint64_t tile_values = 0;  // int16 per tile stream
	std::vector<TileDesc> tiles;
	TileDesc* d_tiles = nullptr;
	uint32_t row_tiles = 0;  // tiles per tile row when the group's tiles form full rows of horizontally adjacent tiles (0: they do not)
};

struct TileInfo
{
	size_t x, y, w, h, stream_off, stream_bytes;
	int group;
};

struct EventPair
{
	hipEvent_t a, b;
};

// buffers of the device Kagari encoder (ako_kagari.hip.h), created on first use
struct KagariState
{
	uint32_t n_tiles = 0, n_blocks = 0;
	KgTile* d_tiles = nullptr;
	uint64_t* d_tile_stage = nullptr;  // byte offset of every tile's staging area
	uint32_t* d_block_runs = nullptr;  // runs per block, then (scanned in place) first run of every block
	uint32_t* d_total_runs = nullptr;
	uint32_t* d_run_start = nullptr;
	uint32_t* d_run_bits = nullptr;
	size_t run_capacity = 0;
	uint64_t* d_block_bits = nullptr;  // bits per block of runs, then (scanned in place) bit offset
	size_t block_bits_capacity = 0;
	uint64_t* d_total_bits = nullptr;
	uint32_t* d_tile_first_run = nullptr;
	uint64_t* d_tile_bit_off = nullptr;
	uint64_t* d_tile_payload = nullptr;
	uint64_t* d_tile_dst = nullptr;
	KgResult* d_result = nullptr;
	uint8_t* d_stage = nullptr;
	size_t stage_bytes = 0;
	uint8_t* d_body = nullptr;
	size_t body_capacity = 0;
	size_t body_bytes = 0;  // of the last successful call
	bool encoder_ready = false;
	// decoder side
	int16_t* d_literals = nullptr;
	size_t literal_capacity = 0;
	KgRun* d_runs = nullptr;
	size_t run_record_capacity = 0;
};

}  // namespace

#ifndef AKO_EXPERIMENTAL
// The default library is built without ako_fused.hip and ako_u8_group.hip: the routes that lost stay in the
// source, parity-tested in an AKO_BUILD_EXPERIMENTAL=1 build, and out of libako.so.  Their launchers are never reached here
// (Tuning::from_env keeps fuse2 and group at 0).
namespace ako
{
int akoFused2ForwardLaunch(int, const F2Params&, hipStream_t) { return 1; }
int akoFused2InverseLaunch(int, const F2Params&, hipStream_t) { return 1; }
void akoLaunchForwardGroupU8_rgba(int, const LevelParams&, const StreamGeom&, uint32_t, hipStream_t) {}
}  // namespace ako
#endif

// Tuning / test knobs, read from the environment ONCE when a plan is created (a launch path that called getenv per
// level and call was neither cheap for small images nor safe against a concurrent setenv).
struct Tuning
{
	int path = 0;          // AKO_HIP_PATH: 0 auto, 1 generic (window engine), 2 stream wherever legal
	int tail = 1;          // AKO_HIP_TAIL: 0 no in-LDS tail, 1 window-engine tail
	int tail_max = 0;      // AKO_HIP_TAIL_MAX: hand only levels this small to the tail (0 = default)
	bool wide = true;      // AKO_HIP_WIDE=0: no halo-free 121..128 column strips
	int seg_rows = 0;      // AKO_HIP_SEG_ROWS: rows per segment of the streaming kernels (0 = chosen per level)
	int seg_rows_big = 0;  // AKO_HIP_SEG_ROWS_BIG: the same for levels of >= 1024 columns only
	int seg_rows_small = 0;  // AKO_HIP_SEG_ROWS_SMALL: shortest segment of levels below 1024 columns (0 = default)
	bool opt = true;       // AKO_HIP_OPT=0: exact int16-wrapping inverse alone (no optimistic fp32 launch)
	int staged = 1;        // AKO_HIP_STAGED=0: no planar staging of 1-3 / 5+ channel u8 images; 2: RGB images staged too (not the u8 kernels)
	bool deep = true;      // AKO_HIP_DEEP=0: small levels keep the running two-slot prefetch
	int seg_rows_mid = 0, seg_rows_mid_inv = 6;  // AKO_HIP_SEG_ROWS_MID / _MID_INV: same for int16 levels of 1024..2047
	                                             // columns, forward / inverse kernels
	int floor_big = 24;    // AKO_HIP_FLOOR_BIG: fewest rows per segment of levels with >= 2048 columns
	int tail_many = 4;     // AKO_HIP_TAIL_MANY: largest level the tail takes when a launch has many planes (tiled images;
	                       // 16384 x 16384 in 256-px tiles, round 2: 8 -> 89, 16 -> 95, 32 -> 90, 64 -> 51 Gpx/s; round 3, with
	                       // several small tiles per wave in the streaming kernels: 16 -> 116-118, 8 -> 119.4-122, 4 -> 123-128)
	int u8_waves = 0;      // AKO_HIP_U8_WAVES: waves a u8 level launch aims at (0 = two rounds of resident waves)
	int lockstep = 3;      // AKO_HIP_LOCKSTEP: StreamGeom::lockstep (bit 0 barrier every six slots, bit 1 strip-major units)
	int fwd_pairs = 2;     // AKO_HIP_FWD_PAIRS: pairs of waves (= neighbouring strips) per workgroup of the u8 forward kernel
	int inv_pairs = 2;     // AKO_HIP_INV_PAIRS: same for the u8 inverse kernel (1, 2 or 4)
	int interior = 1;      // AKO_HIP_LEAN=0: the general u8 level-0 kernels everywhere (not the lean ones of ako_u8_lean.hip.h)
	int fuse2 = 0;         // AKO_HIP_FUSE2: levels 0 and 1 of eligible RGBA plans in one workgroup walk, the level-0 low-pass plane
	                       // handed over through LDS (ako_fused.hip.h): bit 0 forward, bit 1 inverse.  Bit-exact and parity-tested,
	                       // off by default: measured slower than the level-per-kernel launches
	int row_strips = 1;    // AKO_HIP_ROW_STRIPS: level 0 of u8 images in 512-pixel tiles in strips over whole rows of tiles (0: per tile)
	int pack = 1;          // AKO_HIP_PACK: levels of 8..64 columns of tiled images run several tiles per wave (0: one tile per wave)
	int group = 0;         // AKO_HIP_GROUP: level 0 of big RGBA tiles in column groups, the stores re-shaped into whole cache lines
	                       // (k_forward_group_u8, ako_stream.hip.h).  Bit-exact and parity-tested, off by default: the level-0
	                       // kernels turned out to be bound by instruction issue, not by their stores
	int group_min = 1024;  // AKO_HIP_GROUP_MIN: narrowest level-0 sub-band (columns) that takes the group kernel
	int f2_rows = 0;       // AKO_HIP_F2_ROWS: rows per segment of those kernels (0 = one round of workgroups)
	int f2_edge = -1;      // AKO_HIP_F2_EDGE: rows of their first / last segment (-1 = chosen, 0 = like the others)
	uint32_t dbg = 0;      // AKO_HIP_DBG bits (kernel side experiments)

	static Tuning from_env()
	{
		Tuning t;
		auto num = [](const char* name, int dflt) {
			const char* e = getenv(name);
			return (e != nullptr && *e != '\0') ? atoi(e) : dflt;
		};
		if (const char* e = getenv("AKO_HIP_PATH"))
			t.path = (strcmp(e, "generic") == 0) ? 1 : ((strcmp(e, "stream") == 0) ? 2 : 0);
		t.tail = num("AKO_HIP_TAIL", 1) != 0;
		t.tail_max = num("AKO_HIP_TAIL_MAX", 0);
		t.wide = num("AKO_HIP_WIDE", 1) != 0;
		t.seg_rows = num("AKO_HIP_SEG_ROWS", 0);
		t.seg_rows_big = num("AKO_HIP_SEG_ROWS_BIG", 0);
		t.seg_rows_small = num("AKO_HIP_SEG_ROWS_SMALL", 0);
		t.opt = num("AKO_HIP_OPT", 1) != 0;
		t.staged = num("AKO_HIP_STAGED", 1);
		t.deep = num("AKO_HIP_DEEP", 1) != 0;
		t.u8_waves = num("AKO_HIP_U8_WAVES", 0);
		t.tail_many = num("AKO_HIP_TAIL_MANY", 4);
		if (t.tail_many < 4 || t.tail_many > TAIL_MAX)
			t.tail_many = 4;
		t.floor_big = num("AKO_HIP_FLOOR_BIG", 24);
		if (t.floor_big < 2)
			t.floor_big = 2;
		t.seg_rows_mid = num("AKO_HIP_SEG_ROWS_MID", 0);
		t.seg_rows_mid_inv = num("AKO_HIP_SEG_ROWS_MID_INV", 6);  // (6 rows = all twelve row slots fetched up front: level 2 of the 8192 x 8192 image 34.5 -> 32.3 us)
		t.lockstep = num("AKO_HIP_LOCKSTEP", 3) & 3;
		t.fwd_pairs = num("AKO_HIP_FWD_PAIRS", 2);
		t.inv_pairs = num("AKO_HIP_INV_PAIRS", 2);
		t.interior = num("AKO_HIP_LEAN", 1) != 0;
		if (t.fwd_pairs != 1 && t.fwd_pairs != 4)
			t.fwd_pairs = 2;
		if (t.inv_pairs != 1 && t.inv_pairs != 4)
			t.inv_pairs = 2;
		t.fuse2 = num("AKO_HIP_FUSE2", 0) & 3;
#ifndef AKO_EXPERIMENTAL  // the default library holds neither the two-level kernels nor the column-group kernel (ako_amd/build.py)
		t.fuse2 = 0;
#endif
		t.pack = num("AKO_HIP_PACK", 1) != 0;
		t.row_strips = num("AKO_HIP_ROW_STRIPS", 1) != 0;
		t.group = num("AKO_HIP_GROUP", 0);
#ifndef AKO_EXPERIMENTAL
		t.group = 0;
#endif
		t.group_min = num("AKO_HIP_GROUP_MIN", 1024);
		t.f2_rows = num("AKO_HIP_F2_ROWS", 0);
		t.f2_edge = num("AKO_HIP_F2_EDGE", -1);
#ifdef AKO_MEASURE  // measurement builds only: the shipped library does not read AKO_HIP_DBG
		t.dbg = (uint32_t)num("AKO_HIP_DBG", 0);
#endif
		return t;
	}
};

struct akoHipPlan
{
	int device = 0;
	Tuning tune;
	hipStream_t stream = nullptr;
	struct akoSettings s;
	size_t channels = 0, w = 0, h = 0, batch = 0;
	unsigned flags = 0;
	std::vector<Group> groups;
	std::vector<TileInfo> tiles;
	size_t stream_values = 0;  // int16 per image
	int16_t* scratch[2] = {nullptr, nullptr};
	size_t scratch_elems[2] = {0, 0};
	// host-variant staging
	void* d_img = nullptr;
	void* d_stream = nullptr;
	int32_t* d_flags = nullptr;  // overflow flags of the optimistic inverse launches
	int32_t ovf_gen = 0;         // generation number of the latest one
	int16_t* planes0 = nullptr;  // planar int16 image: staging for u8 images with 1-3 or 5+ channels (staged_level0)
	// profiling
	bool profiling = false;
	std::vector<EventPair> events[2];  // [0] encode launches, [1] decode launches
	size_t events_used[2] = {0, 0};
	struct Pending
	{
		akoHipKernelRecord rec;
		size_t ev;
	};
	std::vector<Pending> pending[2];
	KagariState* kg = nullptr;
	bool owns_stream = false;
	// ratio search (akoHipRequantize): the re-quantized streams and the segment table of the last candidate
	int16_t* d_requant = nullptr;
	void* d_rq_segments = nullptr;
	size_t rq_segment_capacity = 0;
	// chunked download of large images (download_chunked): two pinned staging buffers and their events
	void* pin[2] = {nullptr, nullptr};
	hipEvent_t pin_done[2] = {nullptr, nullptr};
};

namespace
{

size_t half_up(size_t v)  // misc.c:98
{
	return (v + 1) / 2;
}

size_t tile_extent(size_t pos, size_t image_d, size_t td)  // misc.c:152
{
	if (td == 0)
		return image_d;
	return (pos + td > image_d) ? (image_d % td) : td;
}

void build_group(Group& g, const akoSettings& s, size_t channels)
{
	size_t w = g.tile_w, h = g.tile_h;
	if (s.wavelet != AKO_WAVELET_NONE)
	{
		while (w > 2 && h > 2)  // lifting.c:182
		{
			LevelGeom L;
			L.cw = (uint32_t)w, L.ch = (uint32_t)h;
			w = half_up(w), h = half_up(h);
			L.tw = (uint32_t)w, L.th = (uint32_t)h;
			if (s.wavelet == AKO_WAVELET_HAAR)
				L.kind = K_HAAR;
			else if (s.wavelet == AKO_WAVELET_CDF53 || w < 8 || h < 8)
				L.kind = K_CDF53;
			else
				L.kind = K_DD137;
			for (int m = 0; m < 2; m++)
			{
				const int mul = (m == 0) ? 1 : s.chroma_loss + 1;
				L.q[m] = akoHostQuantStep(s.quantization, mul, g.tile_w, g.tile_h, L.cw, L.ch);
				L.g[m] = akoHostGateStep(s.gate, mul, g.tile_w, g.tile_h, L.cw, L.ch);
			}
			g.levels.push_back(L);
		}
	}
	g.fw = (uint32_t)w, g.fh = (uint32_t)h;

	// stream layout (misc.c:245-285): low-passes of all planes, then levels small -> large,
	// planes 0 .. N-1, each [head C B D]
	uint64_t off = 0;
	for (size_t p = 0; p < channels; p++)
	{
		g.lp_off[p] = off;
		off += (uint64_t)w * h;
	}
	for (size_t l = g.levels.size(); l-- > 0;)
		for (size_t p = 0; p < channels; p++)
		{
			g.levels[l].grp_off[p] = off;
			off += 1 + 3 * (uint64_t)g.levels[l].tw * g.levels[l].th;
		}
	g.tile_values = off;
}

struct Launch
{
	akoHipPlan* plan;
	int decode;
	hipStream_t on = nullptr;  // the stream the kernel is launched on (default: the plan's)
	size_t ev = (size_t)-1;

	hipStream_t stream() const
	{
		return on ? on : plan->stream;
	}

	int begin()
	{
		if (!plan->profiling)
			return 0;
		if (plan->events_used[decode] == plan->events[decode].size())
		{
			EventPair p;
			HIP_TRY(hipEventCreate(&p.a));
			HIP_TRY(hipEventCreate(&p.b));
			plan->events[decode].push_back(p);
		}
		ev = plan->events_used[decode]++;
		HIP_TRY(hipEventRecord(plan->events[decode][ev].a, stream()));
		return 0;
	}
	int end(const char* name, uint32_t level, uint32_t group, uint64_t units, uint64_t rd, uint64_t wr)
	{
		HIP_TRY(hipGetLastError());
		if (!plan->profiling)
			return 0;
		HIP_TRY(hipEventRecord(plan->events[decode][ev].b, stream()));
		akoHipPlan::Pending p;
		memset(&p.rec, 0, sizeof p.rec);
		snprintf(p.rec.name, sizeof p.rec.name, "%s", name);
		p.rec.level = level, p.rec.group = group, p.rec.units = units, p.rec.bytes_rd = rd, p.rec.bytes_wr = wr;
		p.ev = ev;
		plan->pending[decode].push_back(p);
		return 0;
	}
};

const char* kind_name(int k)
{
	return k == K_DD137 ? "dd137" : (k == K_CDF53 ? "cdf53" : "haar");
}

void fill_common(LevelParams& P, const akoHipPlan* pl, const Group& g, const LevelGeom& L)
{
	memset(&P, 0, sizeof P);
	P.full_w = L.cw, P.full_h = L.ch, P.sub_w = L.tw, P.sub_h = L.th;
	P.wrap = (int)pl->s.wrap;
	P.channels = (uint32_t)pl->channels;
	P.grid_x = (L.tw + TW - 1) / TW, P.grid_y = (L.th + TH - 1) / TH;
	P.tiles = g.d_tiles;
	P.n_tiles = (uint32_t)g.tiles.size();
	P.batch = (uint32_t)pl->batch;
	P.img_pitch = (uint32_t)pl->w;
	P.img_stride = (uint64_t)pl->w * pl->h * pl->channels;
	P.color = (int)pl->s.color;
	P.discard = pl->s.discard_non_visible;
	P.stream_stride = pl->stream_values;
	for (size_t p = 0; p < pl->channels; p++)
	{
		P.lp_off[p] = g.lp_off[p];
		P.grp_off[p] = L.grp_off[p];
	}
	P.q_luma = L.q[0], P.g_luma = L.g[0], P.q_chroma = L.q[1], P.g_chroma = L.g[1];
	P.rq_luma = (float)((1.0 / (double)(L.q[0] < 1 ? 1 : L.q[0])) * (1.0 + 1e-6));
	P.rq_chroma = (float)((1.0 / (double)(L.q[1] < 1 ? 1 : L.q[1])) * (1.0 + 1e-6));
	P.dbg = pl->tune.dbg;
}

template <bool U8>
void launch_forward(int kind, const LevelParams& P, uint32_t blocks, size_t smem, hipStream_t st)
{
	switch (kind)
	{
	case K_DD137: hipLaunchKernelGGL((k_forward_level<K_DD137, U8>), dim3(blocks), dim3(THREADS), smem, st, P); break;
	case K_CDF53: hipLaunchKernelGGL((k_forward_level<K_CDF53, U8>), dim3(blocks), dim3(THREADS), smem, st, P); break;
	default: hipLaunchKernelGGL((k_forward_level<K_HAAR, U8>), dim3(blocks), dim3(THREADS), smem, st, P); break;
	}
}

template <bool U8>
void launch_inverse(int kind, const LevelParams& P, uint32_t blocks, size_t smem, hipStream_t st)
{
	switch (kind)
	{
	case K_DD137: hipLaunchKernelGGL((k_inverse_level<K_DD137, U8>), dim3(blocks), dim3(THREADS), smem, st, P); break;
	case K_CDF53: hipLaunchKernelGGL((k_inverse_level<K_CDF53, U8>), dim3(blocks), dim3(THREADS), smem, st, P); break;
	default: hipLaunchKernelGGL((k_inverse_level<K_HAAR, U8>), dim3(blocks), dim3(THREADS), smem, st, P); break;
	}
}

// ---- choice between the window engine and the register-streaming kernels ----------------------
enum { PATH_AUTO = 0, PATH_GENERIC = 1, PATH_STREAM = 2 };

int path_mode(const akoHipPlan* pl)
{
	return pl->tune.path;
}

bool many_planes(const akoHipPlan* pl)
{
	uint64_t tiles = 0;
	for (const Group& g : pl->groups)
		tiles += g.tiles.size();
	return tiles * pl->batch * pl->channels >= 64;
}

// streaming kernels need a level width that is a multiple of 4 (no phantom column, and an even
// number of coefficient columns so that a lane's column pair is never split by the border);
// in AUTO mode they are used where they pay: wide levels
// Level widths the streaming kernels take (lane_columns() in ako_stream.hip.h).  With an even number of
// coefficient columns (width = 0 or 3 mod 4) every lane holds a full pair; an odd width adds the phantom
// last sample, fixed up in the lane of the last pair.  With an odd number of columns (width = 2 or 1
// mod 4) the last strip shifts by one column: that needs at least two strips, a last strip of three
// columns or more (with a single one the strip before it would need border values inside its own net
// range) and a wrap mode that does not pair up columns across the border (REPEAT does).
bool stream_width_ok(const akoHipPlan* pl, const LevelGeom& L)
{
	if (L.cw < 8)
		return false;
	if ((L.tw & 1) == 0)
		return true;
	return L.tw > (uint32_t)SNET && (L.tw % SNET) != 1 && pl->s.wrap != AKO_WRAP_REPEAT;
}

// RGB (three channel) u8 images take the u8 streaming kernels too (twelve-byte loads / stores of four pixels) when every
// tile row starts on a dword and no tile has a ragged four-pixel group: image width (hence every tile width) a multiple
// of four.  AKO_HIP_STAGED=2 keeps them on the staged route (planar int16 image in front of the int16 kernels).
bool rgb_native(const akoHipPlan* pl)
{
	return pl->channels == 3 && !(pl->flags & AKO_HIP_PLAN_PLANES_I16) && (pl->w % 4) == 0 && pl->tune.staged != 2 &&
	       (pl->s.tiles_dimension == 0 || (pl->s.tiles_dimension % 4) == 0);
}

// One- and two-channel u8 images on the native gray kernels (ako_u8_gray.hip.h) instead of the staged route?  The kernels are
// written in the lean style and take the lean kernels' geometry: DD13/7 or CDF5/3 at level 0, any border rule but MIRROR, a
// level width that is a multiple of four, ordinary strips (no wide strip, no row strips over tiles).  AKO_HIP_STAGED=2 and
// AKO_HIP_LEAN=0 keep such images staged.
uint32_t row_strips(const akoHipPlan* pl, const Group& g, const LevelGeom& L, bool u8);
bool gray_native(const akoHipPlan* pl, const Group& g)
{
	if ((pl->flags & AKO_HIP_PLAN_PLANES_I16) || (pl->channels != 1 && pl->channels != 2) || g.levels.empty())
		return false;
	if (!pl->tune.interior || pl->tune.staged == 2 || pl->s.wrap == AKO_WRAP_MIRROR)
		return false;
	const LevelGeom& L = g.levels[0];
	if ((L.kind != K_DD137 && L.kind != K_CDF53) || (L.cw & 3u) != 0 || L.cw != 2 * L.tw)
		return false;
	if (L.tw > (uint32_t)SNET && L.tw <= 128 && pl->tune.wide)
		return false;  // (one wide strip: general kernels only)
	return row_strips(pl, g, L, true) == 0;
}
bool gray_native_any(const akoHipPlan* pl)
{
	for (const Group& g : pl->groups)
		if (gray_native(pl, g))
			return true;
	return false;
}

bool stream_eligible(const akoHipPlan* pl, const LevelGeom& L, bool u8);
// ... and level 0 of that group is one the streaming kernels take at all
bool gray_level0(const akoHipPlan* pl, const Group& g)
{
	return gray_native(pl, g) && stream_eligible(pl, g.levels[0], false);
}

bool stream_eligible(const akoHipPlan* pl, const LevelGeom& L, bool u8)
{
	const int mode = path_mode(pl);
	if (mode == PATH_GENERIC)
		return false;
	if (!stream_width_ok(pl, L) || L.th < 2)
		return false;
	if (u8 && !(pl->channels == 4 || rgb_native(pl)))
		return false;  // (one / two channels: gray_level0() below)
	// the forward streaming kernels address a tile's stream and the LL scratch planes with 32-bit byte
	// offsets (raw buffer stores): tiles of 4 GiB and more stay on the window engine
	for (const Group& g : pl->groups)
		if (g.tile_values * 2 >= 0xFFF00000ull || (uint64_t)g.tile_w * g.tile_h * pl->channels * 2 >= 0xFFF00000ull)
			return false;
	// ... and their sources with a 32-bit row offset inside the image / plane (raw buffer loads)
	if ((uint64_t)pl->w * pl->h * (u8 ? 4 : 2) >= 0xFFF00000ull)
		return false;
	if (mode == PATH_STREAM)
		return true;
	// a launch over many planes (tiled images, batches of tiles) fills the chip at any level size, and the
	// register kernels do ~3x fewer instructions per sample than the in-LDS tail: stream all the way down
	if (many_planes(pl))
		return true;
	return L.tw >= 64 && L.th >= 12;
}

uint64_t scratch_plane_elems(const Group& g, int which);

// Small tiles side by side in one wave (lane_columns_pack in ako_stream.hip.h): int16 levels of 8, 16, 32 or 64 coefficient
// columns with an even width, any border rule (REPEAT since round 4: every lane fetches its own tile's other end by
// ds_bpermute), more than one tile in the group; returns the tiles per wave (0: no)
uint32_t tile_pack(const akoHipPlan* pl, const Group& g, const LevelGeom& L, bool u8)
{
	if (!pl->tune.pack || u8 || g.tiles.size() < 2)
		return 0;
	if (L.tw < 4 || L.tw > 64 || (L.tw & (L.tw - 1)) != 0 || L.cw != 2 * L.tw || L.th < 2)
		return 0;
	// the lanes address their tile instance and their tile's stream with 32-bit byte offsets from the image's first
	if (pl->stream_values * 2 >= 0xFFF00000ull || (uint64_t)g.tiles.size() * pl->channels * scratch_plane_elems(g, 0) * 2 >= 0xFFF00000ull)
		return 0;
	return 128u / L.tw;
}

// Level 0 of a u8 image in strips laid over whole ROWS of tiles (lane_columns_row in ako_stream.hip.h): at least two tiles per
// row, an even level width, any border rule but REPEAT, and fewer strips than tile by tile (512-pixel tiles: 256 columns
// = three strips each; tiles of 128 pixels and less: a quarter to half a strip each; not 256-pixel tiles, which are exactly
// one wide strip); returns the tiles per row (0: no)
uint32_t row_strips(const akoHipPlan* pl, const Group& g, const LevelGeom& L, bool u8)
{
	if (!pl->tune.row_strips || !u8 || pl->s.wrap == AKO_WRAP_REPEAT || g.row_tiles < 2 || g.row_tiles > 0xFFFF)
		return 0;
	if (L.tw < 4 || (L.tw & 1) != 0 || L.cw != 2 * L.tw)
		return 0;
	if (pl->stream_values * 2 >= 0xFFF00000ull || (uint64_t)g.tiles.size() * pl->channels * scratch_plane_elems(g, 0) * 2 >= 0xFFF00000ull)
		return 0;
	// only where it saves strips
	const bool one_wide = L.tw > (uint32_t)SNET && L.tw <= 128 && pl->tune.wide;  // (a tile of 121..128 columns is ONE wide strip)
	const uint32_t per_tile = one_wide ? 1u : (L.tw + SNET - 1) / SNET, per_row = (g.row_tiles * L.tw + SNET - 1) / SNET;
	return (per_row < per_tile * g.row_tiles) ? g.row_tiles : 0u;
}

// row slots a "deep prefetch" wave fetches up front and then works through (exactly that many, no loop):
// segments of <= 6 rows + 6 halo slots, or of <= 2 rows
constexpr int DEEP_SLOTS = 12, DEEP_SLOTS_SHORT = 8;

// (groups != 0: the column-group kernel -- StreamGeom::strips then counts groups, waves_per_row_unit = 8 per tile instance)
// (pack != 0: that many small tiles side by side in one wave, lane_columns_pack; waves_per_row_unit then counts packs)
StreamGeom stream_geometry(const akoHipPlan* pl, const LevelGeom& L, uint64_t waves_per_row_unit, bool u8, bool inverse, uint32_t groups = 0,
                           uint32_t pack = 0, uint32_t row_tiles = 0)
{
	StreamGeom G;
	G.strips = groups ? groups : (L.tw + SNET - 1) / SNET;
	G.wide = 0;
	G.lockstep = (uint32_t)pl->tune.lockstep;
	G.edge_rows = 0;
	// 121..128 coefficient columns (an even number): one strip without halo lanes instead of two
	if (!groups && L.tw > (uint32_t)SNET && L.tw <= 128 && (L.tw & 1) == 0 && pl->tune.wide)
		G.strips = 1, G.wide = 1;
	if (pack)
		G.strips = 1, G.wide = pack;
	if (row_tiles)  // strips over a whole row of tiles (waves_per_row_unit then counts tile ROWS)
		G.strips = (row_tiles * L.tw + SNET - 1) / SNET, G.wide = 0x80000000u | row_tiles;
	uint32_t seg_rows = (uint32_t)pl->tune.seg_rows;
	if (pl->tune.seg_rows_big != 0 && L.tw >= 1024)  // tuning aid: levels with >= 1024 columns only
		seg_rows = (uint32_t)pl->tune.seg_rows_big;
	if (L.tw >= 1024 && L.tw < 2048 && !u8 && seg_rows == 0)  // ... 1024..2047 columns, per direction
		seg_rows = (uint32_t)(inverse ? pl->tune.seg_rows_mid_inv : pl->tune.seg_rows_mid);
	if (seg_rows == 0)
	{
		// aim at two rounds of resident waves: the u8 kernels run 4 waves per SIMD (8192 waves), the int16
		// ones 5 (10240).  Every segment re-computes 6 halo row slots, so big levels keep segments of >= 24
		// rows; small levels are latency bound (a wave's row slots are a dependent chain) and prefer many
		// short segments.  Rounded DOWN: a handful of waves over the target would cost a third round
		const uint64_t per_seg = (uint64_t)G.strips * waves_per_row_unit;
		// (row strips over many tile rows: one segment per tile row would be a little more than ONE round of resident waves --
		// 4416 for 16384 x 16384 in 512-pixel tiles -- whose stragglers run alone; five rounds of shorter segments instead:
		// 256 -> 64 rows, level 0 1.22 / 1.11 -> 0.98 / 0.84 ms)
		const uint64_t u8_target = row_tiles ? 5 * 4096 : 8192;
		uint64_t segs = (u8 ? (uint64_t)(pl->tune.u8_waves > 0 ? pl->tune.u8_waves : u8_target) : 10240) / per_seg;
		if (segs < 1)
			segs = 1;
		seg_rows = (uint32_t)((L.th + segs - 1) / segs);
		// (level 2 of the 8192 x 8192 image, 1024 columns: 24 -> 12 rows took 42 -> 37 us off each direction)
		const uint32_t floor_rows = (L.tw >= 2048) ? (uint32_t)pl->tune.floor_big
		                            : (L.tw >= 1024) ? 12
		                                             : (pl->tune.seg_rows_small > 0 ? (uint32_t)pl->tune.seg_rows_small : 2);
		if (seg_rows < floor_rows)
		{
			seg_rows = floor_rows;
			// the floor decides: if the launch then is a little more than ONE round of resident waves (int16 kernels:
			// 5 per SIMD = 5120), its second round would run almost empty -- make it exactly one round instead
			// (level 1 of the 8192 x 8192 image: 86 segments = 6192 waves -> 71 = 5112: 88 / 86 -> 82 / 79 us)
			const uint64_t resident = u8 ? 4096 : 5120;
			const uint64_t n_segs = (L.th + seg_rows - 1) / seg_rows;
			if (n_segs * per_seg > resident && n_segs * per_seg < resident * 8 / 5 && resident / per_seg >= 1)
				seg_rows = (uint32_t)((L.th + resident / per_seg - 1) / (resident / per_seg));
		}
	}
	if (seg_rows > L.th)
		seg_rows = L.th;
	G.seg_rows = seg_rows;
	G.segs = (L.th + seg_rows - 1) / seg_rows;
	G.edge_rows = 0;
	// u8 kernels (128 VGPRs, the top / bottom border bodies spill): short segments at the two borders, see StreamGeom
	constexpr uint32_t EDGE_ROWS = 12;
	// (tried for the big int16 levels too, round 4: no gain -- level 1 of the 8192 x 8192 image 86.6-87.6 us with, 85.0-85.2
	// without; profiles/r4_lean_ab.txt)
	if (u8 && pl->tune.seg_rows == 0 && seg_rows > EDGE_ROWS && L.th >= 3 * EDGE_ROWS + seg_rows)
	{
		G.edge_rows = EDGE_ROWS;
		G.segs = 3 + (L.th - 3 * EDGE_ROWS + seg_rows - 1) / seg_rows;
		// the short border segments are extra units: keep the launch within the rounds of resident waves it was sized for
		// (column groups, 10 per row of the 8192 x 8192 level: 103 segments = 1030 workgroups would be a third round of 6)
		const uint64_t resident = 4096, units_per_seg = (uint64_t)G.strips * waves_per_row_unit;
		const uint64_t rounds = ((uint64_t)((L.th + seg_rows - 1) / seg_rows) * units_per_seg + resident - 1) / resident;
		while (groups && (uint64_t)G.segs * units_per_seg > rounds * resident && seg_rows < L.th)
		{
			seg_rows++;
			G.seg_rows = seg_rows;
			G.segs = 3 + (L.th - 3 * EDGE_ROWS + seg_rows - 1) / seg_rows;
		}
	}
	return G;
}

// Levels 0 and 1 of a u8 RGBA plan in one workgroup walk per direction (ako_fused.hip.h): shapes without a phantom
// column or row at either level, the usual colour mode (YCoCg / YCoCg_Q without the discard rule), no REPEAT border (a
// wrapped tap would need the other end of the row / column, which another workgroup holds), the same wavelet at both
// levels (DD13/7 or CDF5/3), and both levels outside the in-LDS tail.
bool fused2_eligible(const akoHipPlan* pl, const Group& g, bool u8_level0, size_t lt, int direction_bit)
{
	if (!(pl->tune.fuse2 & direction_bit) || !u8_level0 || pl->channels != 4 || g.levels.size() < 2 || lt < 2)
		return false;
	const LevelGeom &L0 = g.levels[0], &L1 = g.levels[1];
	if (pl->s.wrap == AKO_WRAP_REPEAT || pl->s.discard_non_visible)
		return false;
	if (pl->s.color != AKO_COLOR_YCOCG && pl->s.color != AKO_COLOR_YCOCG_Q)
		return false;
	if ((L0.cw % 8) != 0 || (L0.ch % 4) != 0 || L0.kind != L1.kind || L0.kind == K_HAAR)
		return false;
	if (!stream_eligible(pl, L0, true) || !stream_eligible(pl, L1, false))
		return false;
	if (g.tile_values * 2 >= 0xFFF00000ull)
		return false;
	// narrow levels leave most of a workgroup's six strips without columns
	const uint32_t min_cols = (path_mode(pl) == PATH_STREAM) ? 16 : 512;
	return L0.tw >= min_cols && L0.th >= 12;
}

void fill_f2(F2Params& F, const akoHipPlan* pl, const Group& g, void* d_images, void* d_streams, int net_cols)
{
	const LevelGeom &L0 = g.levels[0], &L1 = g.levels[1];
	memset(&F, 0, sizeof F);
	F.img = (uint8_t*)d_images;
	F.img_stride = (uint64_t)pl->w * pl->h * 4;
	F.img_pitch = (uint32_t)pl->w;
	F.stream = (int16_t*)d_streams;
	F.stream_stride = pl->stream_values;
	F.tiles = g.d_tiles, F.n_tiles = (uint32_t)g.tiles.size(), F.batch = (uint32_t)pl->batch;
	F.Tc = L0.tw, F.Tr = L0.th;
	F.wrap = (int)pl->s.wrap, F.color = (int)pl->s.color;
	for (int l = 0; l < 2; l++)
	{
		const LevelGeom& L = g.levels[l];
		for (int p = 0; p < 4; p++)
			F.lv[l].grp_off[p] = (uint32_t)L.grp_off[p];
		for (int m = 0; m < 2; m++)
		{
			F.lv[l].q[m] = L.q[m];
			F.lv[l].gate[m] = (float)L.g[m];
			F.lv[l].rq[m] = (float)((1.0 / (double)(L.q[m] < 1 ? 1 : L.q[m])) * (1.0 + 1e-6));
		}
	}
	if (g.levels.size() == 2)
	{
		F.ll1_in_stream = 1;
		for (int p = 0; p < 4; p++)
			F.lp_off[p] = (uint32_t)g.lp_off[p];
	}
	else
	{
		F.ll1 = pl->scratch[1];
		F.ll1_pitch = L1.tw;
		F.ll1_plane_stride = (uint32_t)scratch_plane_elems(g, 1);
		F.ll1_inst_stride = (uint64_t)F.ll1_plane_stride * 4;
	}
	F.groups = (L0.tw + (uint32_t)net_cols - 1) / (uint32_t)net_cols;
	// row segments (every boundary a multiple of 6): one round of workgroups -- a workgroup fills a CU -- unless told otherwise
	const uint64_t insts = (uint64_t)g.tiles.size() * pl->batch;
	uint32_t S = (uint32_t)pl->tune.f2_rows;
	if (S == 0)
	{
		uint64_t segs = 256 / ((uint64_t)F.groups * insts);
		if (segs < 1)
			segs = 1;
		S = (uint32_t)((L0.th + segs - 1) / segs);
	}
	S = ((S + 5) / 6) * 6;
	if (S < 24)
		S = 24;
	uint32_t E = (pl->tune.f2_edge >= 0) ? (uint32_t)pl->tune.f2_edge : ((S * 5 / 8 + 5) / 6) * 6;
	E = ((E + 5) / 6) * 6;
	F.seg_rows = S;
	if (E == 0 || E >= S || L0.th < 2 * E + 24)
	{
		F.edge_rows = 0, F.last_lo = 0;
		F.segs = (L0.th + S - 1) / S;
	}
	else
	{
		F.edge_rows = E;
		F.last_lo = ((L0.th - E) / 6) * 6;
		F.segs = 2 + (F.last_lo - E + S - 1) / S;
	}
}

// int16 levels whose segments fit: every row slot of a wave's segment is fetched before the first is used
int deep_prefetch(const akoHipPlan* pl, const StreamGeom& G, bool u8)
{
	if (u8 || !pl->tune.deep || G.seg_rows + 6 > (uint32_t)DEEP_SLOTS)
		return 0;
	return (G.seg_rows + 6 <= (uint32_t)DEEP_SLOTS_SHORT) ? DEEP_SLOTS_SHORT : DEEP_SLOTS;
}

template <int NPL, bool U8, int DEEP>
void launch_forward_stream(int kind, bool narrow, const LevelParams& P, const StreamGeom& G, uint32_t blocks,
                           hipStream_t st)
{
#define AKO_FS(K, N)                                                                                         \
	hipLaunchKernelGGL((k_forward_stream<K, NPL, U8, N, DEEP>), dim3(blocks), dim3(THREADS), 0, st, P, G)
	if (kind == K_DD137)
	{
		if (narrow) AKO_FS(K_DD137, true); else AKO_FS(K_DD137, false);
	}
	else if (kind == K_CDF53)
	{
		if (narrow) AKO_FS(K_CDF53, true); else AKO_FS(K_CDF53, false);
	}
	else
		AKO_FS(K_HAAR, true);
#undef AKO_FS
}

template <int NPL, bool U8, bool OPT, int DEEP>
void launch_inverse_stream(int kind, const LevelParams& P, const StreamGeom& G, uint32_t blocks, hipStream_t st)
{
	const dim3 threads(U8 ? 128 : THREADS);  // U8: the workgroup is one pair of waves (LDS plane swap)
	if (kind == K_DD137)
		hipLaunchKernelGGL((k_inverse_stream<K_DD137, NPL, U8, OPT, DEEP>), dim3(blocks), threads, 0, st, P, G);
	else if (kind == K_CDF53)
		hipLaunchKernelGGL((k_inverse_stream<K_CDF53, NPL, U8, OPT, DEEP>), dim3(blocks), threads, 0, st, P, G);
	else
		hipLaunchKernelGGL((k_inverse_stream<K_HAAR, NPL, U8, OPT, DEEP>), dim3(blocks), threads, 0, st, P, G);
}

// May this u8 level launch run on the lean kernels (ako_u8_lean.hip.h)?  They hold what carries practically all pixels --
// YCoCg / YCoCg_Q without the discard rule, DD13/7 or CDF5/3, CLAMP / REPEAT / ZERO, a level width that is a multiple of
// four (even number of coefficient columns, no phantom sample), ordinary strips or ONE wide strip (a tile of 121..128 columns;
// not with REPEAT, which needs the other end of the wave) or strips over whole rows of tiles (never REPEAT: row_strips()) -- and
// nothing else; MIRROR, Haar, other colour modes, odd widths and packed small tiles stay on the general kernels.
bool lean_u8_level(const akoHipPlan* pl, const LevelParams& P, const StreamGeom& G, int kind, bool forward)
{
	return pl->tune.interior && (kind == K_DD137 || kind == K_CDF53) && (P.color == C_YCOCG || P.color == C_YCOCG_Q) && !(forward && P.discard != 0) &&
	       P.wrap != W_MIRROR && (P.full_w & 3u) == 0 && P.full_w == 2 * P.sub_w && (G.wide == 0 || (G.wide == 1 && P.wrap != W_REPEAT) || (G.wide >> 31) != 0);
}

void launch_inverse_u8(const akoHipPlan* pl, bool opt, int kind, const LevelParams& P, const StreamGeom& G, uint32_t blocks)
{
	const bool lean = lean_u8_level(pl, P, G, kind, false);
	if (pl->channels == 3)
		akoLaunchInverseU8_rgb(kind, opt, lean, P, G, blocks, (uint32_t)pl->tune.inv_pairs, pl->stream);
	else
		akoLaunchInverseU8_rgba(kind, opt, lean, P, G, blocks, (uint32_t)pl->tune.inv_pairs, pl->stream);
}

int check_blocks(uint64_t blocks)
{
	if (blocks == 0 || blocks > 0x7FFFFFFFull)
		return fail(AKO_ERROR, "launch too large for one grid%s%s");
	return 0;
}

// int16 planes per tile instance held by scratch buffer 'which' for group g
uint64_t scratch_plane_elems(const Group& g, int which)
{
	if ((int)g.levels.size() <= which)
		return 0;
	// a multiple of a cache line, with room for the phase shift of the column-group kernel (plane p starts up to 66 values in)
	return ((uint64_t)g.levels[which].tw * g.levels[which].th + 128 + 63) & ~(uint64_t)63;
}

// Level 0 of a u8 RGBA plan in column groups (k_forward_group_u8, ako_stream.hip.h: "Column groups"): big tiles whose
// sub-band width is a multiple of a cache line of values, an even level width, and a level 1 that runs on the int16
// streaming kernel (which reads the phase-shifted low-pass planes).
bool group_eligible(const akoHipPlan* pl, const Group& g, bool u8_level0, size_t lt)
{
	if (!pl->tune.group || !u8_level0 || pl->channels != 4 || g.levels.size() < 2 || lt < 2)
		return false;
	const LevelGeom &L0 = g.levels[0], &L1 = g.levels[1];
	if ((L0.tw % 64) != 0 || L0.cw != 2 * L0.tw || L0.tw < (uint32_t)pl->tune.group_min || L0.kind == K_HAAR)
		return false;
	return stream_eligible(pl, L0, true) && stream_eligible(pl, L1, false);
}

// The in-LDS tail (window engine, ako_tail.hip.h): AKO_HIP_TAIL=0 switches it off.  (Two other engines were built and
// measured equal or slower on every workload -- a register-blocked segment engine in round 2, a row-parallel "line"
// engine with planes resident from 256 x 256 down in round 3 -- and removed again: DESIGN.md 4.3.)
int tail_engine(const akoHipPlan* pl, const Group& g)
{
	(void)g;
	return pl->tune.tail;
}

// first level handled by the fused in-LDS tail kernel (nl = none).  Level 0 of a u8 image never is.
size_t tail_start(const akoHipPlan* pl, const Group& g)
{
	const size_t nl = g.levels.size();
	const int engine = tail_engine(pl, g);
	if (engine == 0)
		return nl;
	uint32_t lim = (uint32_t)TAIL_MAX;
	if (many_planes(pl) && path_mode(pl) != PATH_GENERIC)
		lim = (uint32_t)pl->tune.tail_many;  // see stream_eligible(): only what the streaming kernels cannot take
	if (pl->tune.tail_max >= 4 && (uint32_t)pl->tune.tail_max < lim)  // tuning aid: hand smaller levels only to the tail
		lim = (uint32_t)pl->tune.tail_max;
	const bool planes = (pl->flags & AKO_HIP_PLAN_PLANES_I16) != 0;
	// Window engine: a launch with few planes (one workgroup each) leaves most of the chip idle, and the tail's first
	// level is its most expensive: when that level (65..128 samples) can run as a streaming launch of its own, the tail
	// starts one level later (8192 x 8192 RGBA, 4 planes: levels >= 6 take 113 -> 91 us for both directions)
	const uint64_t n_planes = (uint64_t)pl->channels * g.tiles.size() * pl->batch;
	for (size_t l = planes ? 0 : 1; l < nl; l++)
		if (g.levels[l].cw <= lim && g.levels[l].ch <= lim)
		{
			if (engine == 1 && pl->tune.tail_max == 0 && n_planes <= 32 && l + 1 < nl && l > 0 &&
			    (g.levels[l].cw > 64 || g.levels[l].ch > 64) && stream_eligible(pl, g.levels[l], false))
				l++;
			return (nl - l <= (size_t)TAIL_LEVELS) ? l : nl;
		}
	return nl;
}

int run_tail(akoHipPlan* pl, int gi, size_t lt, int decode, void* d_images, int16_t* d_streams)
{
	const Group& g = pl->groups[gi];
	const size_t nl = g.levels.size();
	TailParams T;
	memset(&T, 0, sizeof T);
	T.nlev = (uint32_t)(nl - lt);
	uint64_t samples = 0;
	for (size_t l = lt; l < nl; l++)
	{
		const LevelGeom& L = g.levels[l];
		TailLevel& t = T.lv[l - lt];
		t.cw = L.cw, t.ch = L.ch, t.tw = L.tw, t.th = L.th, t.kind = L.kind;
		for (int m = 0; m < 2; m++)
		{
			t.q[m] = L.q[m], t.g[m] = L.g[m];
			t.rq[m] = (float)((1.0 / (double)(L.q[m] < 1 ? 1 : L.q[m])) * (1.0 + 1e-6));
		}
		t.grp0 = L.grp_off[0];
		t.gsize = 1 + 3 * L.tw * L.th;
		samples += (uint64_t)L.cw * L.ch;
	}
	T.wrap = (int)pl->s.wrap;
	T.channels = (uint32_t)pl->channels;
	T.tiles = g.d_tiles, T.n_tiles = (uint32_t)g.tiles.size(), T.batch = (uint32_t)pl->batch;
	T.stream = d_streams, T.stream_stride = pl->stream_values;
	T.fw = g.fw, T.fh = g.fh;
	if (lt == 0)
	{
		// PLANES_I16 mode, small tile: the planes themselves are the input / output
		T.plane = (int16_t*)d_images;
		T.plane_tiled = 1;
		T.plane_pitch = (uint32_t)pl->w;
		T.plane_plane_stride = (uint64_t)pl->w * pl->h;
		T.plane_inst_stride = T.plane_plane_stride * pl->channels;
	}
	else
	{
		const int which = (int)((lt - 1) & 1);  // LL of level lt-1
		T.plane = pl->scratch[which];
		T.plane_pitch = g.levels[lt].cw;
		T.plane_plane_stride = scratch_plane_elems(g, which);
		T.plane_inst_stride = T.plane_plane_stride * pl->channels;
	}
	const uint64_t blocks = (uint64_t)g.tiles.size() * pl->batch * pl->channels;
	if (int rc = check_blocks(blocks))
		return rc;
	T.pitch = 2 * g.levels[lt].tw;
	// LDS and threads by the size of the first (largest) level -- a tiled image has thousands of tiny planes here
	// (16384 x 16384 in 256-pixel tiles: 16384 planes of 8 x 8), and at the full 48 KB / 1024 threads only three of them
	// fit a CU at a time
	const LevelGeom& L0t = g.levels[lt];
	T.win_elems = 2 * (L0t.th + 6) * 2 * (L0t.tw + 8);
	const size_t lds_bytes = ((size_t)T.win_elems + (size_t)L0t.tw * L0t.th) * sizeof(int16_t);
	if (lds_bytes > (size_t)TAIL_LDS_BYTES)
		return fail(AKO_ERROR, "tail level larger than the tail kernel's window%s%s");
	uint32_t tail_threads = 64;
	while (tail_threads < (uint32_t)TAIL_THREADS && (uint64_t)tail_threads * 4 < (uint64_t)L0t.cw * L0t.ch)
		tail_threads *= 2;
	Launch LA{pl, decode};
	if (int rc = LA.begin())
		return rc;
	if (decode)
		hipLaunchKernelGGL(k_inverse_tail, dim3((uint32_t)blocks), dim3(tail_threads), lds_bytes, pl->stream, T);
	else
		hipLaunchKernelGGL(k_forward_tail, dim3((uint32_t)blocks), dim3(tail_threads), lds_bytes, pl->stream, T);
	const uint64_t units = samples * blocks;
	return LA.end(decode ? "inv_tail" : "fwd_tail", (uint32_t)lt, (uint32_t)gi, units, units * 2, units * 2);
}

int run_format(akoHipPlan* pl, int gi, int decode, uint8_t* img, int16_t* stream, bool planar = false)
{
	const Group& g = pl->groups[gi];
	FormatParams F;
	memset(&F, 0, sizeof F);
	F.tile_w = g.tile_w, F.tile_h = g.tile_h, F.channels = (uint32_t)pl->channels;
	F.tiles = g.d_tiles, F.n_tiles = (uint32_t)g.tiles.size(), F.batch = (uint32_t)pl->batch;
	F.img = img, F.img_stride = (uint64_t)pl->w * pl->h * pl->channels, F.img_pitch = (uint32_t)pl->w;
	F.color = (int)pl->s.color, F.discard = pl->s.discard_non_visible;
	F.stream = stream, F.stream_stride = pl->stream_values;
	F.planar = planar ? 1 : 0, F.plane_stride = (uint64_t)pl->w * pl->h;
	const uint64_t npx = (uint64_t)g.tile_w * g.tile_h;
	const uint64_t blocks = ((npx + THREADS - 1) / THREADS) * g.tiles.size() * pl->batch;
	if (int rc = check_blocks(blocks))
		return rc;
	Launch L{pl, decode};
	if (int rc = L.begin())
		return rc;
	const bool quads = planar && pl->channels <= 3 && (g.tile_w % 4) == 0;  // four pixels per thread (else one)
	if (quads)
	{
		const uint64_t nq = (uint64_t)(g.tile_w / 4) * g.tile_h;
		const uint32_t qb = (uint32_t)(((nq + THREADS - 1) / THREADS) * g.tiles.size() * pl->batch);
#define AKO_PLANES4(CH)                                                                                 \
	do                                                                                                  \
	{                                                                                                   \
		if (decode)                                                                                     \
			hipLaunchKernelGGL(k_planes_inverse4<CH>, dim3(qb), dim3(THREADS), 0, pl->stream, F);       \
		else                                                                                            \
			hipLaunchKernelGGL(k_planes_forward4<CH>, dim3(qb), dim3(THREADS), 0, pl->stream, F);       \
	} while (0)
		if (pl->channels == 1)
			AKO_PLANES4(1);
		else if (pl->channels == 2)
			AKO_PLANES4(2);
		else
			AKO_PLANES4(3);
#undef AKO_PLANES4
	}
	else if (decode)
		hipLaunchKernelGGL(k_format_inverse<>, dim3((uint32_t)blocks), dim3(THREADS), 0, pl->stream, F);
	else
		hipLaunchKernelGGL(k_format_forward<>, dim3((uint32_t)blocks), dim3(THREADS), 0, pl->stream, F);
	const uint64_t units = npx * pl->channels * g.tiles.size() * pl->batch;
	return L.end(decode ? (planar ? "planes_to_u8" : "format_inverse") : (planar ? "u8_to_planes" : "format_forward"), 0,
	             (uint32_t)gi, units, decode ? units * 2 : units, decode ? units : units * 2);
}

// u8 images whose channel count is not 4 have no u8 streaming kernel (those split RGBA over a pair of
// waves).  Where level 0 can stream at all they take two cheap extra passes instead of the window
// engine: u8 -> planar int16 (deinterleave + colour) in front of the int16 streaming kernel, and the
// reverse behind it -- the same route a PLANES_I16 plan takes, with a plan-owned staging image.
bool staged_level0(const akoHipPlan* pl, const Group& g)
{
	if ((pl->flags & AKO_HIP_PLAN_PLANES_I16) || pl->channels == 4 || g.levels.empty())
		return false;
	if (rgb_native(pl) && stream_eligible(pl, g.levels[0], true))
		return false;  // the u8 kernels take RGB themselves
	if (gray_level0(pl, g))
		return false;  // ... and the gray kernels one / two channels
	if (pl->s.wavelet == AKO_WAVELET_NONE || path_mode(pl) == PATH_GENERIC)
		return false;
	if (!pl->tune.staged)
		return false;
	const LevelGeom& L = g.levels[0];
	return stream_width_ok(pl, L) && L.th >= 2 && L.tw >= 64 && L.th >= 12;
}

int ensure_planes0(akoHipPlan* pl)
{
	if (pl->planes0)
		return 0;
	if (hipMalloc((void**)&pl->planes0, pl->w * pl->h * pl->channels * pl->batch * sizeof(int16_t)) != hipSuccess)
		return fail(AKO_NO_ENOUGH_MEMORY, "hipMalloc(planar staging image) failed%s%s");
	return 0;
}

int run_encode(akoHipPlan* pl, const void* d_images, void* d_streams)
{
	const bool planes = (pl->flags & AKO_HIP_PLAN_PLANES_I16) != 0;
	for (size_t gi = 0; gi < pl->groups.size(); gi++)
	{
		const Group& g = pl->groups[gi];
		if (pl->s.wavelet == AKO_WAVELET_NONE)
		{
			if (planes)
				return fail(AKO_ERROR, "PLANES_I16 mode needs a wavelet%s%s");
			if (int rc = run_format(pl, (int)gi, 0, (uint8_t*)d_images, (int16_t*)d_streams))
				return rc;
			continue;
		}
		if (g.levels.empty())
			return fail(AKO_ERROR, "tile with an extent <= 2 cannot be encoded (reference mis-reads it: lifting.c:285)%s%s");

		const size_t nl = g.levels.size();
		const size_t lt = tail_start(pl, g);
		const uint64_t insts = (uint64_t)g.tiles.size() * pl->batch;
		const bool staged = staged_level0(pl, g) && lt > 0;
		if (staged)
		{
			if (int rc = ensure_planes0(pl))
				return rc;
			if (int rc = run_format(pl, (int)gi, 0, (uint8_t*)d_images, pl->planes0, true))
				return rc;
		}
		for (size_t l = 0; l < lt; l++)
		{
			const LevelGeom& L = g.levels[l];
			LevelParams P;
			fill_common(P, pl, g, L);
			P.stream = (int16_t*)d_streams;
			const bool u8 = (l == 0) && !planes && !staged;
			if (l == 0 && fused2_eligible(pl, g, u8, lt, 1))
			{
				// levels 0 and 1 in one workgroup walk: level 0's low-pass plane goes from wave to wave through LDS
				const LevelGeom& L1 = g.levels[1];
				F2Params F;
				fill_f2(F, pl, g, (void*)d_images, d_streams, F2_GNET);
				if (int rc = check_blocks((uint64_t)F.groups * F.segs * insts))
					return rc;
				Launch LF{pl, 0};
				if (int rc = LF.begin())
					return rc;
				if (akoFused2ForwardLaunch(L.kind, F, pl->stream) != 0)
					return fail(AKO_ERROR, "two-level forward launch could not be prepared%s%s");
				char fname[48];
				snprintf(fname, sizeof fname, "fwd_fused2_%s_u8", kind_name(L.kind));
				const uint64_t smp0 = (uint64_t)L.cw * L.ch * pl->channels * insts;
				const uint64_t out0 = ((uint64_t)3 * L.tw * L.th + 1) * pl->channels * insts;
				const uint64_t out1 = ((uint64_t)4 * L1.tw * L1.th + 1) * pl->channels * insts;
				if (int rc = LF.end(fname, 0, (uint32_t)gi, smp0, smp0, (out0 + out1) * 2))
					return rc;
				l = 1;  // level 1 is done
				continue;
			}
			const bool grouped0 = !planes && !staged && group_eligible(pl, g, true, lt) && !fused2_eligible(pl, g, true, lt, 1);
			if (u8)
			{
				P.img = (uint8_t*)d_images;
				P.planes_per_wg = (uint32_t)(pl->channels < 4 ? pl->channels : 4);
			}
			else
			{
				P.planes_per_wg = 1;
				if (l == 0)
				{
					P.src = staged ? pl->planes0 : (const int16_t*)d_images;
					P.src_tiled = 1;
					P.src_pitch = (uint32_t)pl->w;
					P.src_plane_stride = (uint64_t)pl->w * pl->h;
					P.src_inst_stride = P.src_plane_stride * pl->channels;
				}
				else
				{
					const int which = (int)((l - 1) & 1);
					P.src = pl->scratch[which];
					P.src_pitch = L.cw;
					P.src_plane_stride = scratch_plane_elems(g, which);
					P.src_inst_stride = P.src_plane_stride * pl->channels;
					if (l == 1 && grouped0)  // low-pass planes shifted to the phase of the stream's cache lines
						P.src_tiled = 2u | ((uint32_t)((g.levels[0].grp_off[0] + 1) & 63) << 8);
				}
			}
			P.plane_groups = (uint32_t)((pl->channels + P.planes_per_wg - 1) / P.planes_per_wg);
			if (l + 1 == nl)
				P.ll_out_stream = 1;
			else
			{
				const int which = (int)(l & 1);
				P.dst = pl->scratch[which];
				P.dst_pitch = L.tw;
				P.dst_plane_stride = scratch_plane_elems(g, which);
				P.dst_inst_stride = P.dst_plane_stride * pl->channels;
			}

			const bool gray = u8 && pl->channels <= 2 && gray_level0(pl, g);  // one / two channels on the native gray kernels
			const bool streaming = gray || stream_eligible(pl, L, u8);
			Launch LA{pl, 0};
			char name[48];
			if (streaming)
			{
				if (gray)
					P.planes_per_wg = (uint32_t)pl->channels, P.plane_groups = 1;  // one wave per strip carries every plane
				else if (u8)
					P.planes_per_wg = 2, P.plane_groups = 2;  // a pair of waves per RGBA strip
				const bool grouped = u8 && grouped0;
				// small tiles side by side in one wave: the tile instances of a launch then count in packs
				// (not level 0 of a planar / staged plan: its source is the image, tile by tile, not a plane per tile instance)
				// (nor level 1 behind a column-group level 0: its low-pass planes are shifted by a phase that differs per tile)
				const uint32_t pack = (grouped || l == 0 || (l == 1 && grouped0)) ? 0u : tile_pack(pl, g, L, u8);
				const uint32_t rowt = (grouped || l != 0) ? 0u : row_strips(pl, g, L, u8);
				const uint64_t pinsts = pack   ? (uint64_t)packs_of((uint32_t)g.tiles.size(), pack) * pl->batch
				                        : rowt ? (uint64_t)(g.tiles.size() / rowt) * pl->batch
				                               : insts;
				const StreamGeom G = grouped ? stream_geometry(pl, L, (uint64_t)GRP_WAVES * insts, u8, false, group_count(L.tw))
				                             : stream_geometry(pl, L, (uint64_t)P.plane_groups * pinsts, u8, false, 0, pack, rowt);
				const uint64_t units = (uint64_t)G.strips * G.segs * P.plane_groups * pinsts;
				const uint32_t waves_per_block = gray ? 4u : (u8 ? 2u * (uint32_t)pl->tune.fwd_pairs : (uint32_t)(THREADS / 64));
				const uint64_t blocks = grouped ? (uint64_t)G.strips * G.segs * insts : (units + waves_per_block - 1) / waves_per_block;
				if (int rc = check_blocks(blocks))
					return rc;
				// int16 narrowing after every step is a no-op where the worst-case growth of u8-sourced
				// data provably stays inside int16: levels 0 and 1 (|Y| <= 510 -> level 0 output <= 2953 ->
				// level 1 output <= 17097).  Those levels run on the exact fp32 pipeline.
				const bool narrow = planes || l >= 2;
				const int deep = deep_prefetch(pl, G, u8);
				if (int rc = LA.begin())
					return rc;
#ifdef AKO_MEASURE
				if (u8 && (pl->tune.dbg & 16))
					hipLaunchKernelGGL(k_forward_stream_u8_memonly<>, dim3((uint32_t)blocks), dim3(64 * waves_per_block), 0, pl->stream, P, G);
				else if (!u8 && (pl->tune.dbg & 16))
					hipLaunchKernelGGL(k_forward_stream_i16_memonly<>, dim3((uint32_t)blocks), dim3(THREADS), 0, pl->stream, P, G);
				else
#endif
				if (gray)
					akoLaunchForwardU8_gray(L.kind, (int)pl->channels, P, G, (uint32_t)blocks, 64 * waves_per_block, pl->stream);
				else if (grouped)
					akoLaunchForwardGroupU8_rgba(L.kind, P, G, (uint32_t)blocks, pl->stream);
				else if (u8 && pl->channels == 3)
					akoLaunchForwardU8_rgb(L.kind, lean_u8_level(pl, P, G, L.kind, true), P, G, (uint32_t)blocks, 64 * waves_per_block, pl->stream);
				else if (u8)
					akoLaunchForwardU8_rgba(L.kind, lean_u8_level(pl, P, G, L.kind, true), P, G, (uint32_t)blocks, 64 * waves_per_block, pl->stream);
				else if (deep == DEEP_SLOTS_SHORT)
					launch_forward_stream<1, false, DEEP_SLOTS_SHORT>(L.kind, narrow, P, G, (uint32_t)blocks, pl->stream);
				else if (deep)
					launch_forward_stream<1, false, DEEP_SLOTS>(L.kind, narrow, P, G, (uint32_t)blocks, pl->stream);
				else
					launch_forward_stream<1, false, 0>(L.kind, narrow, P, G, (uint32_t)blocks, pl->stream);
				snprintf(name, sizeof name, "fwd_%s_%s%s%s", grouped ? "group" : "stream", kind_name(L.kind), u8 ? "_u8" : "", deep ? "_deep" : "");
			}
			else
			{
				const uint64_t blocks = (uint64_t)P.grid_x * P.grid_y * P.plane_groups * insts;
				if (int rc = check_blocks(blocks))
					return rc;
				const size_t smem = (size_t)P.planes_per_wg * WPLANE * sizeof(int16_t);
				if (int rc = LA.begin())
					return rc;
				if (u8)
					launch_forward<true>(L.kind, P, (uint32_t)blocks, smem, pl->stream);
				else
					launch_forward<false>(L.kind, P, (uint32_t)blocks, smem, pl->stream);
				snprintf(name, sizeof name, "fwd_level_%s%s", kind_name(L.kind), u8 ? "_u8" : "");
			}
			const uint64_t samples = (uint64_t)L.cw * L.ch * pl->channels * insts;
			const uint64_t outs = ((uint64_t)4 * L.tw * L.th + 1) * pl->channels * insts;
			if (int rc = LA.end(name, (uint32_t)l, (uint32_t)gi, samples, samples * (u8 ? 1 : 2), outs * 2))
				return rc;
		}
		if (lt < nl)
			if (int rc = run_tail(pl, (int)gi, lt, 0, (void*)d_images, (int16_t*)d_streams))
				return rc;
	}
	return 0;
}

int run_decode(akoHipPlan* pl, const void* d_streams, void* d_images)
{
	const bool planes = (pl->flags & AKO_HIP_PLAN_PLANES_I16) != 0;
	for (size_t gi = 0; gi < pl->groups.size(); gi++)
	{
		const Group& g = pl->groups[gi];
		if (pl->s.wavelet == AKO_WAVELET_NONE || g.levels.empty())
		{
			if (planes)
				return fail(AKO_ERROR, "PLANES_I16 mode needs at least one lift%s%s");
			if (int rc = run_format(pl, (int)gi, 1, (uint8_t*)d_images, (int16_t*)d_streams))
				return rc;
			continue;
		}

		const size_t nl = g.levels.size();
		const size_t lt = tail_start(pl, g);
		const uint64_t insts = (uint64_t)g.tiles.size() * pl->batch;
		if (lt < nl)
			if (int rc = run_tail(pl, (int)gi, lt, 1, d_images, (int16_t*)d_streams))
				return rc;
		const bool staged = staged_level0(pl, g) && lt > 0;
		if (staged)
			if (int rc = ensure_planes0(pl))
				return rc;
		// levels 1 and 0 in one workgroup walk (optimistic fp32 pipeline, ako_fused.hip.h); the exact kernels of both levels
		// are launched behind it and return at once unless it raised the overflow flag
		const bool f2inv = pl->tune.opt && fused2_eligible(pl, g, !planes && !staged, lt, 2);
		int32_t* f2_flag = nullptr;
		int32_t f2_gen = 0;
		for (size_t l = lt; l-- > 0;)
		{
			const LevelGeom& L = g.levels[l];
			LevelParams P;
			fill_common(P, pl, g, L);
			P.stream = (int16_t*)d_streams;
			if (l == 1 && f2inv)
			{
				F2Params F;
				fill_f2(F, pl, g, d_images, (void*)d_streams, F2I_GNET);
				if (int rc = check_blocks((uint64_t)F.groups * F.segs * insts))
					return rc;
				if (pl->ovf_gen == INT32_MAX)
				{
					HIP_TRY(hipMemsetAsync(pl->d_flags, 0, 64 * sizeof(int32_t), pl->stream));
					pl->ovf_gen = 0;
				}
				F.ovf_flag = f2_flag = pl->d_flags + (gi * 8) % 64;
				F.ovf_gen = f2_gen = ++pl->ovf_gen;
				Launch LF{pl, 1};
				if (int rc = LF.begin())
					return rc;
				if (akoFused2InverseLaunch(L.kind, F, pl->stream) != 0)
					return fail(AKO_ERROR, "two-level inverse launch could not be prepared (dynamic LDS limit)%s%s");
				char fname[48];
				snprintf(fname, sizeof fname, "inv_fused2_%s_u8", kind_name(L.kind));
				const LevelGeom& L0 = g.levels[0];
				const uint64_t smp0 = (uint64_t)L0.cw * L0.ch * pl->channels * insts;
				const uint64_t in0 = ((uint64_t)3 * L0.tw * L0.th + 1) * pl->channels * insts;
				const uint64_t in1 = ((uint64_t)4 * L.tw * L.th + 1) * pl->channels * insts;
				if (int rc = LF.end(fname, 0, (uint32_t)gi, smp0, (in0 + in1) * 2, smp0))
					return rc;
			}
			const bool u8 = (l == 0) && !planes && !staged;
			P.planes_per_wg = u8 ? (uint32_t)(pl->channels < 4 ? pl->channels : 4) : 1;
			P.plane_groups = (uint32_t)((pl->channels + P.planes_per_wg - 1) / P.planes_per_wg);

			if (l + 1 == nl)
				P.ll_in_stream = 1;
			else
			{
				const int which = (int)(l & 1);
				P.src = pl->scratch[which];
				P.src_pitch = L.tw;
				P.src_plane_stride = scratch_plane_elems(g, which);
				P.src_inst_stride = P.src_plane_stride * pl->channels;
			}
			if (u8)
				P.img = (uint8_t*)d_images;
			else if (l == 0)
			{
				P.dst = staged ? pl->planes0 : (int16_t*)d_images;
				P.dst_tiled = 1;
				P.dst_pitch = (uint32_t)pl->w;
				P.dst_plane_stride = (uint64_t)pl->w * pl->h;
				P.dst_inst_stride = P.dst_plane_stride * pl->channels;
			}
			else
			{
				const int which = (int)((l - 1) & 1);
				P.dst = pl->scratch[which];
				P.dst_pitch = L.cw;
				P.dst_plane_stride = scratch_plane_elems(g, which);
				P.dst_inst_stride = P.dst_plane_stride * pl->channels;
			}

			const bool gray = u8 && pl->channels <= 2 && gray_level0(pl, g);  // one / two channels on the native gray kernels
			const bool streaming = gray || stream_eligible(pl, L, u8);
			Launch LA{pl, 1};
			char name[48];
			if (streaming)
			{
				if (gray)
					P.planes_per_wg = (uint32_t)pl->channels, P.plane_groups = 1;  // one wave per strip carries every plane
				else if (u8)
					P.planes_per_wg = 2, P.plane_groups = 2;  // a pair of waves per RGBA strip = one workgroup
				const uint32_t pack = (l == 0) ? 0u : tile_pack(pl, g, L, u8);
				const uint32_t rowt = (l != 0 || f2inv) ? 0u : row_strips(pl, g, L, u8);
				const uint64_t pinsts = pack   ? (uint64_t)packs_of((uint32_t)g.tiles.size(), pack) * pl->batch
				                        : rowt ? (uint64_t)(g.tiles.size() / rowt) * pl->batch
				                               : insts;
				const StreamGeom G = stream_geometry(pl, L, (uint64_t)P.plane_groups * pinsts, u8, true, 0, pack, rowt);
				const uint64_t units = (uint64_t)G.strips * G.segs * P.plane_groups * pinsts;
				const uint32_t waves_per_block = gray ? 4u : (u8 ? 2u * (uint32_t)pl->tune.inv_pairs : (uint32_t)(THREADS / 64));
				const int deep = deep_prefetch(pl, G, u8);
				const uint64_t blocks = (units + waves_per_block - 1) / waves_per_block;
				if (int rc = check_blocks(blocks))
					return rc;
				// u8 side: optimistic fp32 launch, then the exact kernel which only works if the first one
				// raised the overflow flag (AKO_HIP_OPT=0 runs the exact kernel alone)
				const bool behind_fused = f2inv && l <= 1;  // exact kernel only, behind the two-level launch
				if (behind_fused)
					P.ovf_flag = f2_flag, P.ovf_gen = f2_gen;
				const bool optimistic = u8 && !gray && pl->tune.opt;  // (the gray kernel IS the exact pipeline: one launch)
				if (optimistic && !behind_fused)
				{
					const size_t slot = (gi * 8 + l) % 64;
					// the flag is never reset: an optimistic launch raises it to its generation number (they only
					// grow), the exact kernel behind it works only if it finds exactly that number
					if (pl->ovf_gen == INT32_MAX)
					{
						HIP_TRY(hipMemsetAsync(pl->d_flags, 0, 64 * sizeof(int32_t), pl->stream));
						pl->ovf_gen = 0;
					}
					P.ovf_flag = pl->d_flags + slot;
					P.ovf_gen = ++pl->ovf_gen;
					Launch LO{pl, 1};
					if (int rc = LO.begin())
						return rc;
#ifdef AKO_MEASURE
					if (pl->tune.dbg & 16)
						hipLaunchKernelGGL(k_inverse_stream_u8_memonly<>, dim3((uint32_t)blocks), dim3(128 * (uint32_t)pl->tune.inv_pairs), 0, pl->stream, P, G);
					else
#endif
						launch_inverse_u8(pl, true, L.kind, P, G, (uint32_t)blocks);
					snprintf(name, sizeof name, "inv_stream_%s_u8", kind_name(L.kind));
					const uint64_t smp = (uint64_t)L.cw * L.ch * pl->channels * insts;
					const uint64_t ins = ((uint64_t)4 * L.tw * L.th + 1) * pl->channels * insts;
					if (int rc = LO.end(name, (uint32_t)l, (uint32_t)gi, smp, ins * 2, smp))
						return rc;
				}
				if (int rc = LA.begin())
					return rc;
				if (gray)
					akoLaunchInverseU8_gray(L.kind, (int)pl->channels, P, G, (uint32_t)blocks, 64 * waves_per_block, pl->stream);
				else if (u8)
					launch_inverse_u8(pl, false, L.kind, P, G, (uint32_t)blocks);
				else if (deep == DEEP_SLOTS_SHORT)
					launch_inverse_stream<1, false, false, DEEP_SLOTS_SHORT>(L.kind, P, G, (uint32_t)blocks, pl->stream);
				else if (deep)
					launch_inverse_stream<1, false, false, DEEP_SLOTS>(L.kind, P, G, (uint32_t)blocks, pl->stream);
				else
					launch_inverse_stream<1, false, false, 0>(L.kind, P, G, (uint32_t)blocks, pl->stream);
				snprintf(name, sizeof name, "inv_stream_%s%s%s", kind_name(L.kind), u8 ? "_u8" : (deep ? "_deep" : ""),
				         (optimistic || behind_fused) ? "_exact_if_flagged" : "");
			}
			else
			{
				const uint64_t blocks = (uint64_t)P.grid_x * P.grid_y * P.plane_groups * insts;
				if (int rc = check_blocks(blocks))
					return rc;
				const size_t smem = (size_t)P.planes_per_wg * WPLANE * sizeof(int16_t);
				if (int rc = LA.begin())
					return rc;
				if (u8)
					launch_inverse<true>(L.kind, P, (uint32_t)blocks, smem, pl->stream);
				else
					launch_inverse<false>(L.kind, P, (uint32_t)blocks, smem, pl->stream);
				snprintf(name, sizeof name, "inv_level_%s%s", kind_name(L.kind), u8 ? "_u8" : "");
			}
			const uint64_t samples = (uint64_t)L.cw * L.ch * pl->channels * insts;
			const uint64_t ins = ((uint64_t)4 * L.tw * L.th + 1) * pl->channels * insts;
			if (int rc = LA.end(name, (uint32_t)l, (uint32_t)gi, samples, ins * 2, samples * (u8 ? 1 : 2)))
				return rc;
		}
		if (staged)
			if (int rc = run_format(pl, (int)gi, 1, (uint8_t*)d_images, pl->planes0, true))
				return rc;
	}
	return 0;
}

}  // namespace

// nothing may unwind through the C callers of the C-ABI: a failed host allocation (std::bad_alloc from a vector)
// becomes AKO_NO_ENOUGH_MEMORY
template <typename F>
static int guarded(F&& body)
{
	try
	{
		return body();
	}
	catch (...)
	{
		return fail(AKO_NO_ENOUGH_MEMORY, "out of host memory%s%s");
	}
}

// ---------------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------------

#pragma GCC visibility push(default)
extern "C" {

// 1 when the library was built with AKO_BUILD_EXPERIMENTAL=1: AKO_HIP_FUSE2 / AKO_HIP_GROUP then select the two-level workgroup
// kernels (ako_fused.hip.h) and the column-group level-0 kernel (k_forward_group_u8); 0: those knobs are ignored
int akoHipHasExperimental(void)
{
#ifdef AKO_EXPERIMENTAL
	return 1;
#else
	return 0;
#endif
}

int akoHipDeviceCount(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
	{
		(void)hipGetLastError();
		return 0;
	}
	return n;
}

const char* akoHipLastError(void)
{
	return g_last_error.c_str();
}

// What the tuning knobs of the environment say right now, folded into one word: the drivers keep a thread's plan
// between calls and must not reuse it once a knob has changed (plans read the knobs when they are created).
uint64_t akoHipTuningSignature(void)
{
	const Tuning t = Tuning::from_env();
	const int v[] = {t.path, t.tail, t.tail_max, t.wide, t.seg_rows, t.seg_rows_big, t.seg_rows_small, t.opt, t.staged, t.deep, t.u8_waves, t.lockstep, t.fwd_pairs, t.inv_pairs, t.fuse2, t.pack, t.row_strips, t.group, t.group_min, t.f2_rows, t.f2_edge, t.seg_rows_mid, t.seg_rows_mid_inv, t.floor_big, t.tail_many, t.interior, (int)t.dbg};
	uint64_t h = 1469598103934665603ull;
	for (int x : v)
		h = (h ^ (uint64_t)(uint32_t)x) * 1099511628211ull;
	return h;
}

enum akoColor akoHipEffectiveColor(const struct akoSettings* s)  // encode.c:59-64
{
	if (s->color == AKO_COLOR_YCOCG && (s->quantization > 0 || s->gate > 0))
		return AKO_COLOR_YCOCG_Q;
	if (s->color == AKO_COLOR_YCOCG_Q && (s->quantization <= 0 && s->gate <= 0))
		return AKO_COLOR_YCOCG;
	return s->color;
}

akoHipPlan* akoHipPlanCreate(int device, const struct akoSettings* settings, size_t channels, size_t image_w,
                             size_t image_h, size_t batch, void* hip_stream, unsigned flags,
                             enum akoStatus* out_status)
{
	enum akoStatus st = AKO_OK;
	akoHipPlan* pl = nullptr;
	g_last_error.clear();

#define PLAN_FAIL(code, msg)      \
	do                            \
	{                             \
		st = (code);              \
		g_last_error = (msg);     \
		goto failed;              \
	} while (0)

	try  // nothing may unwind through the C callers: std::bad_alloc becomes AKO_NO_ENOUGH_MEMORY
	{
		if (settings == nullptr || batch == 0)
			PLAN_FAIL(AKO_INVALID_INPUT, "null settings or empty batch");
		if (channels == 0 || channels > AKO_MAX_CHANNELS)
			PLAN_FAIL(AKO_INVALID_CHANNELS_NO, "channels must be 1..16");
		if (image_w == 0 || image_h == 0 || image_w > AKO_MAX_WIDTH || image_h > AKO_MAX_HEIGHT)
			PLAN_FAIL(AKO_INVALID_DIMENSIONS, "invalid image dimensions");
		const size_t td = settings->tiles_dimension;
		if (td != 0 && (td < AKO_MIN_TILES_DIMENSION || td > AKO_MAX_TILES_DIMENSION || (td & (td - 1)) != 0))
			PLAN_FAIL(AKO_INVALID_TILES_DIMENSIONS, "tiles dimension must be 0 or a power of two >= 8");
		if ((int)settings->wrap < 0 || (int)settings->wrap > 3)
			PLAN_FAIL(AKO_INVALID_WRAP_MODE, "invalid wrap mode");
		if ((int)settings->wavelet < 0 || (int)settings->wavelet > 3)
			PLAN_FAIL(AKO_INVALID_WAVELET_TRANSFORMATION, "invalid wavelet");
		if ((int)settings->color < 0 || (int)settings->color > 3)
			PLAN_FAIL(AKO_INVALID_COLOR_TRANSFORMATION, "invalid colour transformation");

		// sizes that do not fit: 64-bit products must not wrap (w * h * channels, the streams, the scratch planes),
		// and the tile table must stay a table (a forged head could ask for 10^17 tiles of 8 pixels)
		{
			size_t samples = 0, bytes = 0, n_tiles = 0;
			const size_t tiles_x = td ? (image_w + td - 1) / td : 1, tiles_y = td ? (image_h + td - 1) / td : 1;
			if (__builtin_mul_overflow(image_w, image_h, &samples) || __builtin_mul_overflow(samples, channels, &samples) ||
			    __builtin_mul_overflow(samples, batch, &samples) || __builtin_mul_overflow(samples, (size_t)4, &bytes) ||
			    bytes > ((size_t)1 << 47) || __builtin_mul_overflow(tiles_x, tiles_y, &n_tiles) || n_tiles > ((size_t)1 << 28))
				PLAN_FAIL(AKO_NO_ENOUGH_MEMORY, "image too large for a device plan");
		}

		int ndev = 0;
		if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
		{
			(void)hipGetLastError();
			PLAN_FAIL(AKO_ERROR, "no usable HIP device: this library has no CPU fallback");
		}
		if (device < 0 || device >= ndev)
			PLAN_FAIL(AKO_ERROR, "HIP device index out of range");
		if (hipSetDevice(device) != hipSuccess)
			PLAN_FAIL(AKO_ERROR, "hipSetDevice failed");

		pl = new akoHipPlan();
		pl->device = device;
		pl->stream = (hipStream_t)hip_stream;
		if (hip_stream == nullptr && (flags & AKO_HIP_PLAN_OWN_STREAM))
		{
			if (hipStreamCreateWithFlags(&pl->stream, hipStreamNonBlocking) != hipSuccess)
				PLAN_FAIL(AKO_ERROR, "hipStreamCreate failed");
			pl->owns_stream = true;
		}
		pl->s = *settings;
		pl->tune = Tuning::from_env();
		pl->channels = channels, pl->w = image_w, pl->h = image_h, pl->batch = batch, pl->flags = flags;
		if (flags & AKO_HIP_PLAN_PLANES_I16)
			pl->s.color = AKO_COLOR_NONE;

		// tiles in raster order (encode.c:115-204), grouped by extent
		size_t off = 0;
		for (size_t ty = 0; ty < image_h; ty += (td ? td : image_h))
			for (size_t tx = 0; tx < image_w; tx += (td ? td : image_w))
			{
				const size_t tw = tile_extent(tx, image_w, td), th = tile_extent(ty, image_h, td);
				int gi = -1;
				for (size_t k = 0; k < pl->groups.size(); k++)
					if (pl->groups[k].tile_w == tw && pl->groups[k].tile_h == th)
						gi = (int)k;
				if (gi < 0)
				{
					Group g;
					g.tile_w = (uint32_t)tw, g.tile_h = (uint32_t)th;
					build_group(g, pl->s, channels);
					pl->groups.push_back(g);
					gi = (int)pl->groups.size() - 1;
				}
				Group& g = pl->groups[gi];
				TileDesc d;
				d.x0 = (uint32_t)tx, d.y0 = (uint32_t)ty, d.stream_off = off;
				g.tiles.push_back(d);
				TileInfo ti{tx, ty, tw, th, off * 2, (size_t)g.tile_values * 2, gi};
				pl->tiles.push_back(ti);
				off += g.tile_values;
			}
		pl->stream_values = off;

		for (Group& g : pl->groups)
		{
			// rows of horizontally adjacent tiles, all rows equally long, in raster order (the interior group, the bottom edge)
			{
				size_t ntx = 1;
				while (ntx < g.tiles.size() && g.tiles[ntx].y0 == g.tiles[0].y0)
					ntx++;
				bool ok = ntx >= 2 && g.tiles.size() % ntx == 0;
				for (size_t i = 0; ok && i < g.tiles.size(); i++)
					ok = (i % ntx == 0) ? (g.tiles[i].x0 == g.tiles[0].x0 && (i == 0 || g.tiles[i].y0 > g.tiles[i - ntx].y0))
					                    : (g.tiles[i].y0 == g.tiles[i - 1].y0 && g.tiles[i].x0 == g.tiles[i - 1].x0 + g.tile_w);
				g.row_tiles = ok ? (uint32_t)ntx : 0u;
			}
			if (hipMalloc((void**)&g.d_tiles, g.tiles.size() * sizeof(TileDesc)) != hipSuccess)
				PLAN_FAIL(AKO_NO_ENOUGH_MEMORY, "hipMalloc(tile table) failed");
			if (hipMemcpy(g.d_tiles, g.tiles.data(), g.tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice) !=
			    hipSuccess)
				PLAN_FAIL(AKO_ERROR, "hipMemcpy(tile table) failed");
			for (int which = 0; which < 2; which++)
			{
				const size_t need = (size_t)scratch_plane_elems(g, which) * channels * g.tiles.size() * batch;
				if (need > pl->scratch_elems[which])
					pl->scratch_elems[which] = need;
			}
		}
		for (int which = 0; which < 2; which++)
			if (pl->scratch_elems[which] != 0 &&
			    hipMalloc((void**)&pl->scratch[which], pl->scratch_elems[which] * sizeof(int16_t)) != hipSuccess)
				PLAN_FAIL(AKO_NO_ENOUGH_MEMORY, "hipMalloc(low-pass scratch) failed");
		// everything a first call would otherwise allocate inside its (possibly timed) launch sequence
		if (hipMalloc((void**)&pl->d_flags, 64 * sizeof(int32_t)) != hipSuccess)
			PLAN_FAIL(AKO_NO_ENOUGH_MEMORY, "hipMalloc(overflow flags) failed");
		if (hipMemset(pl->d_flags, 0, 64 * sizeof(int32_t)) != hipSuccess)
			PLAN_FAIL(AKO_ERROR, "hipMemset(overflow flags) failed");
		for (const Group& g : pl->groups)
			if (pl->planes0 == nullptr && staged_level0(pl, g) && tail_start(pl, g) > 0 && ensure_planes0(pl) != 0)
				PLAN_FAIL(AKO_NO_ENOUGH_MEMORY, "hipMalloc(planar staging image) failed");
	}
	catch (...)
	{
		st = AKO_NO_ENOUGH_MEMORY;
		g_last_error = "out of host memory while building the plan";
		goto failed;
	}

	if (out_status)
		*out_status = AKO_OK;
	return pl;

failed:
	(void)hipGetLastError();
	if (pl)
		akoHipPlanDestroy(pl);
	if (out_status)
		*out_status = st;
	return nullptr;
#undef PLAN_FAIL
}

void akoHipPlanDestroy(akoHipPlan* pl)
{
	if (!pl)
		return;
	(void)hipSetDevice(pl->device);
	for (Group& g : pl->groups)
		if (g.d_tiles)
			(void)hipFree(g.d_tiles);
	for (int which = 0; which < 2; which++)
		if (pl->scratch[which])
			(void)hipFree(pl->scratch[which]);
	if (pl->d_img)
		(void)hipFree(pl->d_img);
	if (pl->d_stream)
		(void)hipFree(pl->d_stream);
	if (pl->d_flags)
		(void)hipFree(pl->d_flags);
	if (pl->planes0)
		(void)hipFree(pl->planes0);
	for (int k = 0; k < 2; k++)
	{
		if (pl->pin[k])
			(void)hipHostFree(pl->pin[k]);
		if (pl->pin_done[k])
			(void)hipEventDestroy(pl->pin_done[k]);
	}
	if (pl->d_requant)
		(void)hipFree(pl->d_requant);
	if (pl->d_rq_segments)
		(void)hipFree(pl->d_rq_segments);
	if (pl->kg)
	{
		KagariState* k = pl->kg;
		void* bufs[] = {k->d_tiles, k->d_tile_stage, k->d_block_runs, k->d_total_runs, k->d_run_start, k->d_run_bits,
		                k->d_block_bits, k->d_total_bits, k->d_tile_first_run, k->d_tile_bit_off, k->d_tile_payload,
		                k->d_tile_dst, k->d_result, k->d_stage, k->d_body, k->d_literals, k->d_runs};
		for (void* b : bufs)
			if (b)
				(void)hipFree(b);
		delete k;
	}
	for (int d = 0; d < 2; d++)
		for (EventPair& e : pl->events[d])
		{
			(void)hipEventDestroy(e.a);
			(void)hipEventDestroy(e.b);
		}
	if (pl->owns_stream)
	{
		(void)hipStreamSynchronize(pl->stream);
		(void)hipStreamDestroy(pl->stream);
	}
	delete pl;
}

size_t akoHipPlanImageBytes(const akoHipPlan* pl)
{
	return pl->w * pl->h * pl->channels * ((pl->flags & AKO_HIP_PLAN_PLANES_I16) ? 2 : 1);
}

size_t akoHipPlanStreamBytes(const akoHipPlan* pl)
{
	return pl->stream_values * 2;
}

size_t akoHipPlanTiles(const akoHipPlan* pl)
{
	return pl->tiles.size();
}

size_t akoHipPlanBatch(const akoHipPlan* pl)
{
	return pl->batch;
}

int akoHipPlanTileInfo(const akoHipPlan* pl, size_t t, size_t* x, size_t* y, size_t* w, size_t* h, size_t* so,
                       size_t* sb)
{
	if (t >= pl->tiles.size())
		return fail(AKO_INVALID_INPUT, "tile index out of range%s%s");
	const TileInfo& ti = pl->tiles[t];
	if (x) *x = ti.x;
	if (y) *y = ti.y;
	if (w) *w = ti.w;
	if (h) *h = ti.h;
	if (so) *so = ti.stream_off;
	if (sb) *sb = ti.stream_bytes;
	return 0;
}

int akoHipPlanLevels(const akoHipPlan* pl, size_t t)
{
	if (t >= pl->tiles.size())
		return -1;
	return (int)pl->groups[pl->tiles[t].group].levels.size();
}

int akoHipPlanQuant(const akoHipPlan* pl, size_t t, size_t level, size_t channel, int* q, int* g)
{
	if (t >= pl->tiles.size())
		return fail(AKO_INVALID_INPUT, "tile index out of range%s%s");
	const Group& gr = pl->groups[pl->tiles[t].group];
	if (level >= gr.levels.size() || channel >= pl->channels)
		return fail(AKO_INVALID_INPUT, "level / channel out of range%s%s");
	const int m = (channel == 0) ? 0 : 1;
	if (q) *q = gr.levels[level].q[m];
	if (g) *g = gr.levels[level].g[m];
	return 0;
}

int akoHipEncode(akoHipPlan* pl, const void* d_images, void* d_streams)
{
	if (!pl || !d_images || !d_streams)
		return fail(AKO_INVALID_INPUT, "null argument%s%s");
	HIP_TRY(hipSetDevice(pl->device));
	return guarded([&] { return run_encode(pl, d_images, d_streams); });
}

int akoHipDecode(akoHipPlan* pl, const void* d_streams, void* d_images)
{
	if (!pl || !d_images || !d_streams)
		return fail(AKO_INVALID_INPUT, "null argument%s%s");
	HIP_TRY(hipSetDevice(pl->device));
	return guarded([&] { return run_decode(pl, d_streams, d_images); });
}

void* akoHipHostAlloc(size_t bytes)
{
	void* p = nullptr;
	if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess)
	{
		(void)hipGetLastError();
		return nullptr;
	}
	return p;
}

void akoHipHostFree(void* p)
{
	if (p)
		(void)hipHostFree(p);
}

int akoHipHostIsPinned(const void* p)
{
	hipPointerAttribute_t attr;
	if (p == nullptr || hipPointerGetAttributes(&attr, p) != hipSuccess)
	{
		(void)hipGetLastError();
		return 0;
	}
	return attr.type == hipMemoryTypeHost;
}

int akoHipSynchronize(akoHipPlan* pl)
{
	HIP_TRY(hipSetDevice(pl->device));
	HIP_TRY(hipStreamSynchronize(pl->stream));
	return 0;
}

// Device -> pageable host memory for LARGE results.  One hipMemcpy into pageable memory is staged by the runtime
// through its own pinned buffers and copied out by ONE host thread (about 19 GB/s for a 268 MB image); here the
// device writes 16 MB chunks into two pinned buffers of the plan in turn (link rate), and while chunk k + 1 is on
// the link a few host threads copy chunk k to its destination.
constexpr size_t DL_CHUNK = (size_t)16 << 20;
constexpr int DL_THREADS = 4;

static int download_chunked(akoHipPlan* pl, void* h_dst, const void* d_src, size_t bytes)
{
	hipPointerAttribute_t attr;
	const bool pinned = (hipPointerGetAttributes(&attr, h_dst) == hipSuccess && attr.type == hipMemoryTypeHost);
	if (!pinned)
		(void)hipGetLastError();  // plain malloc'ed memory: the query fails on purpose
	if (pinned || bytes < 4 * DL_CHUNK)
	{
		HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, pl->stream));
		HIP_TRY(hipStreamSynchronize(pl->stream));
		return 0;
	}
	for (int k = 0; k < 2; k++)
	{
		if (!pl->pin[k] && hipHostMalloc(&pl->pin[k], DL_CHUNK, hipHostMallocDefault) != hipSuccess)
			return fail(AKO_NO_ENOUGH_MEMORY, "hipHostMalloc(download staging) failed%s%s");
		if (!pl->pin_done[k])
			HIP_TRY(hipEventCreateWithFlags(&pl->pin_done[k], hipEventDisableTiming));
	}
	const size_t n = (bytes + DL_CHUNK - 1) / DL_CHUNK;
	auto issue = [&](size_t k) -> int {
		const size_t off = k * DL_CHUNK, len = (off + DL_CHUNK <= bytes) ? DL_CHUNK : bytes - off;
		HIP_TRY(hipMemcpyAsync(pl->pin[k & 1], (const uint8_t*)d_src + off, len, hipMemcpyDeviceToHost, pl->stream));
		HIP_TRY(hipEventRecord(pl->pin_done[k & 1], pl->stream));
		return 0;
	};
	if (int rc = issue(0))
		return rc;
	if (n > 1)
		if (int rc = issue(1))
			return rc;
	for (size_t k = 0; k < n; k++)
	{
		HIP_TRY(hipEventSynchronize(pl->pin_done[k & 1]));
		const size_t off = k * DL_CHUNK, len = (off + DL_CHUNK <= bytes) ? DL_CHUNK : bytes - off;
		uint8_t* dst = (uint8_t*)h_dst + off;
		const uint8_t* src = (const uint8_t*)pl->pin[k & 1];
		const size_t part = ((len / DL_THREADS) + 63) & ~(size_t)63;
		std::thread helpers[DL_THREADS - 1];
		int started = 0;
		for (int t = 1; t < DL_THREADS; t++)
		{
			const size_t lo = (size_t)t * part;
			if (lo >= len)
				break;
			const size_t cnt = (lo + part <= len) ? part : len - lo;
			try
			{
				helpers[started] = std::thread([=] { memcpy(dst + lo, src + lo, cnt); });
				started++;
			}
			catch (...)
			{
				memcpy(dst + lo, src + lo, cnt);  // no thread to be had: the caller copies this part too
			}
		}
		memcpy(dst, src, part < len ? part : len);
		for (int t = 0; t < started; t++)
			helpers[t].join();
		if (k + 2 < n)
			if (int rc = issue(k + 2))
				return rc;
	}
	return 0;
}

// Host -> device for a large PAGEABLE source, the mirror image of download_chunked(): a few host threads fill one pinned
// 16 MB buffer while the other one is on the link (a plain hipMemcpy from pageable memory runs at about a quarter of the
// link rate).  The copies are ordered on the plan's stream; returns when the last one has been issued AND has left the
// staging buffers.
static int upload_chunked(akoHipPlan* pl, void* d_dst, const void* h_src, size_t bytes)
{
	hipPointerAttribute_t attr;
	const bool pinned = (hipPointerGetAttributes(&attr, h_src) == hipSuccess && attr.type == hipMemoryTypeHost);
	if (!pinned)
		(void)hipGetLastError();
	if (pinned || bytes < 4 * DL_CHUNK)
	{
		HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, pl->stream));
		return 0;
	}
	for (int k = 0; k < 2; k++)
	{
		if (!pl->pin[k] && hipHostMalloc(&pl->pin[k], DL_CHUNK, hipHostMallocDefault) != hipSuccess)
			return fail(AKO_NO_ENOUGH_MEMORY, "hipHostMalloc(upload staging) failed%s%s");
		if (!pl->pin_done[k])
			HIP_TRY(hipEventCreateWithFlags(&pl->pin_done[k], hipEventDisableTiming));
	}
	const size_t n = (bytes + DL_CHUNK - 1) / DL_CHUNK;
	for (size_t k = 0; k < n; k++)
	{
		HIP_TRY(hipEventSynchronize(pl->pin_done[k & 1]));  // whatever copy last used this buffer has left it
		const size_t off = k * DL_CHUNK, len = (off + DL_CHUNK <= bytes) ? DL_CHUNK : bytes - off;
		uint8_t* dst = (uint8_t*)pl->pin[k & 1];
		const uint8_t* src = (const uint8_t*)h_src + off;
		const size_t part = ((len / DL_THREADS) + 63) & ~(size_t)63;
		std::thread helpers[DL_THREADS - 1];
		int started = 0;
		for (int t = 1; t < DL_THREADS; t++)
		{
			const size_t lo = (size_t)t * part;
			if (lo >= len)
				break;
			const size_t cnt = (lo + part <= len) ? part : len - lo;
			try
			{
				helpers[started] = std::thread([=] { memcpy(dst + lo, src + lo, cnt); });
				started++;
			}
			catch (...)
			{
				memcpy(dst + lo, src + lo, cnt);
			}
		}
		memcpy(dst, src, part < len ? part : len);
		for (int t = 0; t < started; t++)
			helpers[t].join();
		HIP_TRY(hipMemcpyAsync((uint8_t*)d_dst + off, dst, len, hipMemcpyHostToDevice, pl->stream));
		HIP_TRY(hipEventRecord(pl->pin_done[k & 1], pl->stream));
	}
	return 0;
}

static int ensure_staging(akoHipPlan* pl)
{
	if (!pl->d_img)
		if (hipMalloc(&pl->d_img, akoHipPlanImageBytes(pl) * pl->batch) != hipSuccess)
			return fail(AKO_NO_ENOUGH_MEMORY, "hipMalloc(image staging) failed%s%s");
	if (!pl->d_stream)
		if (hipMalloc(&pl->d_stream, akoHipPlanStreamBytes(pl) * pl->batch) != hipSuccess)
			return fail(AKO_NO_ENOUGH_MEMORY, "hipMalloc(stream staging) failed%s%s");
	return 0;
}

int akoHipEncodeHost(akoHipPlan* pl, const void* h_images, void* h_streams)
{
	if (!pl || !h_images || !h_streams)
		return fail(AKO_INVALID_INPUT, "null argument%s%s");
	HIP_TRY(hipSetDevice(pl->device));
	if (int rc = ensure_staging(pl))
		return rc;
	HIP_TRY(hipMemcpyAsync(pl->d_img, h_images, akoHipPlanImageBytes(pl) * pl->batch, hipMemcpyHostToDevice,
	                       pl->stream));
	if (int rc = akoHipEncode(pl, pl->d_img, pl->d_stream))
		return rc;
	HIP_TRY(hipMemcpyAsync(h_streams, pl->d_stream, akoHipPlanStreamBytes(pl) * pl->batch, hipMemcpyDeviceToHost,
	                       pl->stream));
	HIP_TRY(hipStreamSynchronize(pl->stream));
	return 0;
}

int akoHipDecodeHost(akoHipPlan* pl, const void* h_streams, void* h_images)
{
	if (!pl || !h_images || !h_streams)
		return fail(AKO_INVALID_INPUT, "null argument%s%s");
	HIP_TRY(hipSetDevice(pl->device));
	if (int rc = ensure_staging(pl))
		return rc;
	HIP_TRY(hipMemcpyAsync(pl->d_stream, h_streams, akoHipPlanStreamBytes(pl) * pl->batch, hipMemcpyHostToDevice,
	                       pl->stream));
	if (int rc = akoHipDecode(pl, pl->d_stream, pl->d_img))
		return rc;
	return download_chunked(pl, h_images, pl->d_img, akoHipPlanImageBytes(pl) * pl->batch);
}

int akoHipPlanSetProfiling(akoHipPlan* pl, int enabled)
{
	pl->profiling = enabled != 0;
	pl->pending[0].clear();
	pl->pending[1].clear();
	pl->events_used[0] = pl->events_used[1] = 0;
	return 0;
}

size_t akoHipPlanKernelRecords(akoHipPlan* pl, int decode, struct akoHipKernelRecord* out, size_t capacity)
{
	const int d = decode ? 1 : 0;
	const auto& v = pl->pending[d];
	size_t n = 0;
	for (const auto& p : v)
	{
		if (n == capacity)
			break;
		out[n] = p.rec;
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, pl->events[d][p.ev].a, pl->events[d][p.ev].b) != hipSuccess)
		{
			(void)hipGetLastError();
			ms = -1.f;
		}
		out[n].ms = ms;
		n++;
	}
	return n;
}

// ---- device Kagari encoder ------------------------------------------------------------------------

static int kagari_state(akoHipPlan* pl)
{
	if (pl->kg && pl->kg->encoder_ready)
		return 0;
	if (pl->flags & AKO_HIP_PLAN_PLANES_I16)
		return fail(AKO_INVALID_INPUT, "the entropy stage works on coefficient streams, not on PLANES_I16 plans%s%s");
	if (pl->stream_values == 0 || pl->stream_values > 0xFFFFFFF0ull)
		return fail(AKO_ERROR, "stream too large for the device entropy stage%s%s");
	if (pl->kg && pl->kg->d_tiles)
		return fail(AKO_NO_ENOUGH_MEMORY, "the device entropy stage could not be set up earlier%s%s");
	if (!pl->kg)
		pl->kg = new KagariState;
	KagariState* k = pl->kg;
	std::vector<KgTile> tiles(pl->tiles.size());
	std::vector<uint64_t> stage(pl->tiles.size());
	uint64_t blocks = 0, expect = 0;
	for (size_t t = 0; t < pl->tiles.size(); t++)
	{
		const TileInfo& ti = pl->tiles[t];
		if (ti.stream_off != expect)  // the run logic relies on tiles being contiguous and in order
			return fail(AKO_ERROR, "internal: tile streams are not contiguous%s%s");
		expect += ti.stream_bytes;
		tiles[t].off = ti.stream_off / 2, tiles[t].n = ti.stream_bytes / 2;
		tiles[t].first_block = (uint32_t)blocks, tiles[t].pad = 0;
		blocks += (tiles[t].n + KG_RUN_CHUNK - 1) / KG_RUN_CHUNK;
		stage[t] = ((ti.stream_off + 7) & ~(uint64_t)7) + 16 * t;  // 8-byte aligned, 8+ bytes of slack each
	}
	k->n_tiles = (uint32_t)tiles.size(), k->n_blocks = (uint32_t)blocks;
	k->stage_bytes = pl->stream_values * 2 + 16 * tiles.size() + 32;
	k->body_capacity = pl->stream_values * 2 + 4 * tiles.size();
#define KG_ALLOC(ptr, bytes)                                                                \
	if (hipMalloc((void**)&(ptr), (bytes)) != hipSuccess)                                   \
		return fail(AKO_NO_ENOUGH_MEMORY, "hipMalloc(device entropy stage) failed%s%s")
	KG_ALLOC(k->d_tiles, tiles.size() * sizeof(KgTile));
	KG_ALLOC(k->d_tile_stage, tiles.size() * 8);
	KG_ALLOC(k->d_block_runs, (blocks + 1) * 4);
	KG_ALLOC(k->d_total_runs, 8);
	KG_ALLOC(k->d_total_bits, 8);
	KG_ALLOC(k->d_tile_first_run, tiles.size() * 4);
	KG_ALLOC(k->d_tile_bit_off, tiles.size() * 8);
	KG_ALLOC(k->d_tile_payload, tiles.size() * 8);
	KG_ALLOC(k->d_tile_dst, tiles.size() * 8);
	KG_ALLOC(k->d_result, sizeof(KgResult));
	KG_ALLOC(k->d_stage, k->stage_bytes);
	KG_ALLOC(k->d_body, k->body_capacity);
	HIP_TRY(hipMemcpy(k->d_tiles, tiles.data(), tiles.size() * sizeof(KgTile), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(k->d_tile_stage, stage.data(), stage.size() * 8, hipMemcpyHostToDevice));
	k->encoder_ready = true;
	return 0;
}

int akoHipEncodeUpload(akoHipPlan* pl, const void* h_images)
{
	if (!pl || !h_images)
		return fail(AKO_INVALID_INPUT, "null argument%s%s");
	HIP_TRY(hipSetDevice(pl->device));
	if (int rc = ensure_staging(pl))
		return rc;
	HIP_TRY(hipMemcpyAsync(pl->d_img, h_images, akoHipPlanImageBytes(pl) * pl->batch, hipMemcpyHostToDevice,
	                       pl->stream));
	return akoHipEncode(pl, pl->d_img, pl->d_stream);
}

static int kagari_encode(akoHipPlan* pl, const void* d_streams, size_t image, size_t* body_bytes, size_t* failed_tile);
int akoHipKagariEncode(akoHipPlan* pl, const void* d_streams, size_t image, size_t* body_bytes, size_t* failed_tile)
{
	return guarded([&] { return kagari_encode(pl, d_streams, image, body_bytes, failed_tile); });
}
static int kagari_encode(akoHipPlan* pl, const void* d_streams, size_t image, size_t* body_bytes, size_t* failed_tile)
{
	if (!pl || image >= pl->batch)
		return fail(AKO_INVALID_INPUT, "null plan or image index out of range%s%s");
	if (d_streams == nullptr)
		d_streams = pl->d_stream;
	if (d_streams == nullptr)
		return fail(AKO_INVALID_INPUT, "no coefficient streams: pass a device pointer or call akoHipEncodeUpload first%s%s");
	HIP_TRY(hipSetDevice(pl->device));
	if (int rc = kagari_state(pl))
		return rc;
	KagariState* k = pl->kg;
	k->body_bytes = 0;
	hipStream_t st = pl->stream;
	const int16_t* s = static_cast<const int16_t*>(d_streams) + image * pl->stream_values;

	// 1-2: run starts per block, first run of every block
	hipLaunchKernelGGL(k_kg_starts<false>, dim3(k->n_blocks), dim3(KG_THREADS), 0, st, s, k->d_tiles, k->n_tiles,
	                   k->d_block_runs, (const uint32_t*)nullptr, (uint32_t*)nullptr);
	hipLaunchKernelGGL(k_kg_scan<uint32_t>, dim3(1), dim3(KG_SCAN_THREADS), 0, st, k->d_block_runs, k->d_block_runs,
	                   (uint64_t)k->n_blocks, k->d_total_runs);
	uint32_t n_runs = 0;
	HIP_TRY(hipMemcpyAsync(&n_runs, k->d_total_runs, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	const uint32_t run_blocks = (n_runs + KG_THREADS - 1) / KG_THREADS;
	if (n_runs > k->run_capacity)
	{
		if (k->d_run_start)
			(void)hipFree(k->d_run_start), k->d_run_start = nullptr;
		if (k->d_run_bits)
			(void)hipFree(k->d_run_bits), k->d_run_bits = nullptr;
		k->run_capacity = 0;
		const size_t want = (size_t)n_runs + n_runs / 4 + 1024;
		KG_ALLOC(k->d_run_start, want * 4);
		KG_ALLOC(k->d_run_bits, want * 4);
		k->run_capacity = want;
	}
	if (run_blocks > k->block_bits_capacity)
	{
		if (k->d_block_bits)
			(void)hipFree(k->d_block_bits), k->d_block_bits = nullptr;
		k->block_bits_capacity = 0;
		const size_t want = (size_t)run_blocks + run_blocks / 4 + 64;
		KG_ALLOC(k->d_block_bits, want * 8);
		k->block_bits_capacity = want;
	}
#undef KG_ALLOC
	// 3-6: runs, their bits, bit offsets, tile layout
	hipLaunchKernelGGL(k_kg_starts<true>, dim3(k->n_blocks), dim3(KG_THREADS), 0, st, s, k->d_tiles, k->n_tiles,
	                   (uint32_t*)nullptr, (const uint32_t*)k->d_block_runs, k->d_run_start);
	hipLaunchKernelGGL(k_kg_bits, dim3(run_blocks), dim3(KG_THREADS), 0, st, s, (const uint32_t*)k->d_run_start, n_runs,
	                   (uint64_t)pl->stream_values, k->d_run_bits, k->d_block_bits);
	hipLaunchKernelGGL(k_kg_scan<uint64_t>, dim3(1), dim3(KG_SCAN_THREADS), 0, st, k->d_block_bits, k->d_block_bits,
	                   (uint64_t)run_blocks, k->d_total_bits);
	hipLaunchKernelGGL(k_kg_tiles, dim3(k->n_tiles), dim3(64), 0, st, (const KgTile*)k->d_tiles, k->n_tiles,
	                   (const uint32_t*)k->d_block_runs, (const uint32_t*)k->d_run_bits, (const uint64_t*)k->d_block_bits,
	                   k->d_tile_first_run, k->d_tile_bit_off);
	hipLaunchKernelGGL(k_kg_layout, dim3(1), dim3(KG_SCAN_THREADS), 0, st, (const KgTile*)k->d_tiles, k->n_tiles,
	                   (const uint64_t*)k->d_tile_bit_off, (const uint64_t*)k->d_total_bits, k->d_tile_payload, k->d_tile_dst,
	                   k->d_result);
	// 7: the bits (into zeroed staging areas)
	HIP_TRY(hipMemsetAsync(k->d_stage, 0, k->stage_bytes, st));
	hipLaunchKernelGGL(k_kg_write, dim3(run_blocks), dim3(KG_THREADS), 0, st, s, (const uint32_t*)k->d_run_start,
	                   (const uint32_t*)k->d_run_bits, n_runs, (uint64_t)pl->stream_values, (const uint64_t*)k->d_block_bits,
	                   (const uint32_t*)k->d_tile_first_run, (const uint64_t*)k->d_tile_bit_off,
	                   (const uint64_t*)k->d_tile_stage, k->n_tiles, (const KgResult*)k->d_result, k->d_stage);
	KgResult res;
	HIP_TRY(hipMemcpyAsync(&res, k->d_result, sizeof res, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	if (res.failed_tile != 0xFFFFFFFFu)
	{
		if (failed_tile)
			*failed_tile = res.failed_tile;
		return fail(AKO_ERROR, "a tile did not shrink under Kagari (library/encode.c:159-164)%s%s");
	}
	// 8: blob body
	const uint32_t gather_blocks = (uint32_t)((res.body_bytes + KG_GATHER_BYTES - 1) / KG_GATHER_BYTES);
	hipLaunchKernelGGL(k_kg_gather, dim3(gather_blocks), dim3(KG_THREADS), 0, st, (const uint8_t*)k->d_stage,
	                   (const uint64_t*)k->d_tile_stage, (const uint64_t*)k->d_tile_payload, (const uint64_t*)k->d_tile_dst,
	                   k->n_tiles, (uint64_t)res.body_bytes, k->d_body);
	HIP_TRY(hipGetLastError());
	k->body_bytes = res.body_bytes;
	if (body_bytes)
		*body_bytes = res.body_bytes;
	return 0;
}

// ---- ratio search support: one transform, many quantizations -------------------------------------

void* akoHipPlanDeviceImages(akoHipPlan* pl)
{
	if (!pl || hipSetDevice(pl->device) != hipSuccess || ensure_staging(pl) != 0)
		return nullptr;
	return pl->d_img;
}

void* akoHipPlanDeviceStreams(akoHipPlan* pl)
{
	if (!pl || hipSetDevice(pl->device) != hipSuccess || ensure_staging(pl) != 0)
		return nullptr;
	return pl->d_stream;
}

static int requantize(akoHipPlan* pl, int quantization, int gate, const void* d_unquantized, void** d_out)
{
	if (!pl)
		return fail(AKO_INVALID_INPUT, "null plan%s%s");
	if ((pl->flags & AKO_HIP_PLAN_PLANES_I16) || pl->s.wavelet == AKO_WAVELET_NONE)
		return fail(AKO_INVALID_INPUT, "nothing to re-quantize: the plan has no coefficient groups%s%s");
	if (pl->s.quantization > 0 || pl->s.gate > 0)
		return fail(AKO_INVALID_INPUT, "akoHipRequantize works on the streams of a plan created with quantization 0 and gate 0%s%s");
	HIP_TRY(hipSetDevice(pl->device));
	if (d_unquantized == nullptr)
	{
		if (int rc = ensure_staging(pl))
			return rc;
		d_unquantized = pl->d_stream;
	}
	if (!pl->d_requant)
		if (hipMalloc((void**)&pl->d_requant, akoHipPlanStreamBytes(pl) * pl->batch) != hipSuccess)
			return fail(AKO_NO_ENOUGH_MEMORY, "hipMalloc(re-quantized streams) failed%s%s");

	// segment table: per tile the low-pass section (copied) and one group per level and plane, in stream order
	std::vector<RqSegment> segs;
	uint32_t blocks = 0;
	auto push = [&](uint64_t start, uint64_t count, int q, int g) {
		RqSegment s;
		s.start = start, s.count = count, s.q = q, s.g = g;
		s.rq = (float)((1.0 / (double)(q < 1 ? 1 : q)) * (1.0 + 1e-6));
		s.first_block = blocks;
		blocks += (uint32_t)((count + 7 + RQ_CHUNK - 1) / RQ_CHUNK);  // + 7: the kernel's windows are 16-byte aligned in the buffer
		segs.push_back(s);
	};
	for (const TileInfo& ti : pl->tiles)
	{
		const Group& g = pl->groups[ti.group];
		const uint64_t t0 = ti.stream_off / 2;
		push(t0, (uint64_t)g.fw * g.fh * pl->channels, 0, 0);
		for (size_t l = g.levels.size(); l-- > 0;)
		{
			const LevelGeom& L = g.levels[l];
			for (size_t p = 0; p < pl->channels; p++)
			{
				const int mul = (p == 0) ? 1 : pl->s.chroma_loss + 1;  // lifting.c:202-211
				const int q = akoHostQuantStep(quantization, mul, g.tile_w, g.tile_h, L.cw, L.ch);
				const int gt = akoHostGateStep(gate, mul, g.tile_w, g.tile_h, L.cw, L.ch);
				push(t0 + L.grp_off[p], 1 + 3 * (uint64_t)L.tw * L.th, q < 1 ? 1 : q, gt);
			}
		}
	}
	if ((uint64_t)blocks * pl->batch > 0x7FFFFFFFull)
		return fail(AKO_ERROR, "launch too large for one grid%s%s");
	if (segs.size() > pl->rq_segment_capacity)
	{
		if (pl->d_rq_segments)
			(void)hipFree(pl->d_rq_segments), pl->d_rq_segments = nullptr;
		pl->rq_segment_capacity = 0;
		if (hipMalloc(&pl->d_rq_segments, segs.size() * sizeof(RqSegment)) != hipSuccess)
			return fail(AKO_NO_ENOUGH_MEMORY, "hipMalloc(segment table) failed%s%s");
		pl->rq_segment_capacity = segs.size();
	}
	HIP_TRY(hipMemcpyAsync(pl->d_rq_segments, segs.data(), segs.size() * sizeof(RqSegment), hipMemcpyHostToDevice, pl->stream));
	Launch LA{pl, 0};
	if (int rc = LA.begin())
		return rc;
	hipLaunchKernelGGL(k_requantize, dim3(blocks * (uint32_t)pl->batch), dim3(RQ_THREADS), 0, pl->stream,
	                   (const int16_t*)d_unquantized, pl->d_requant, (const RqSegment*)pl->d_rq_segments, (uint32_t)segs.size(),
	                   (uint64_t)pl->stream_values, blocks);
	const uint64_t vals = (uint64_t)pl->stream_values * pl->batch;
	if (int rc = LA.end("requantize", 0, 0, vals, vals * 2, vals * 2))
		return rc;
	HIP_TRY(hipStreamSynchronize(pl->stream));  // the host table may go away
	if (d_out)
		*d_out = pl->d_requant;
	return 0;
}

int akoHipRequantize(akoHipPlan* pl, int quantization, int gate, const void* d_unquantized, void** d_out)
{
	return guarded([&] { return requantize(pl, quantization, gate, d_unquantized, d_out); });
}

int akoHipKagariFetch(akoHipPlan* pl, void* h_body)
{
	if (!pl || !pl->kg || !h_body || pl->kg->body_bytes == 0)
		return fail(AKO_INVALID_INPUT, "no entropy-coded body to fetch%s%s");
	HIP_TRY(hipSetDevice(pl->device));
	HIP_TRY(hipMemcpyAsync(h_body, pl->kg->d_body, pl->kg->body_bytes, hipMemcpyDeviceToHost, pl->stream));
	HIP_TRY(hipStreamSynchronize(pl->stream));
	return 0;
}

const void* akoHipKagariBody(const akoHipPlan* pl)
{
	return (pl && pl->kg && pl->kg->body_bytes) ? pl->kg->d_body : nullptr;
}

static int kagari_expand(akoHipPlan* pl, const int16_t* h_literals, size_t n_literals, const struct akoHipKagariRun* h_runs,
                         size_t n_runs, void* d_streams, size_t image);
int akoHipKagariExpand(akoHipPlan* pl, const int16_t* h_literals, size_t n_literals, const struct akoHipKagariRun* h_runs,
                       size_t n_runs, void* d_streams, size_t image)
{
	return guarded([&] { return kagari_expand(pl, h_literals, n_literals, h_runs, n_runs, d_streams, image); });
}
static int kagari_expand(akoHipPlan* pl, const int16_t* h_literals, size_t n_literals, const struct akoHipKagariRun* h_runs,
                         size_t n_runs, void* d_streams, size_t image)
{
	static_assert(sizeof(KgRun) == sizeof(struct akoHipKagariRun), "record layout");
	if (!pl || !h_literals || n_literals == 0 || (n_runs != 0 && !h_runs) || image >= pl->batch)
		return fail(AKO_INVALID_INPUT, "bad argument%s%s");
	if (pl->flags & AKO_HIP_PLAN_PLANES_I16)
		return fail(AKO_INVALID_INPUT, "the entropy stage works on coefficient streams, not on PLANES_I16 plans%s%s");
	if (pl->stream_values > 0xFFFFFFF0ull || n_runs > 0xFFFFFFF0ull)
		return fail(AKO_ERROR, "stream too large for the device entropy stage%s%s");
	HIP_TRY(hipSetDevice(pl->device));
	if (d_streams == nullptr)
	{
		if (int rc = ensure_staging(pl))
			return rc;
		d_streams = pl->d_stream;
	}
	// the records must describe exactly this image's stream: literals + repeats == values, runs in order
	// and inside the stream (a forged record must not make the kernel read or write out of range)
	{
		uint64_t repeats = 0, prev_end = 0;
		for (size_t k = 0; k < n_runs; k++)
		{
			const struct akoHipKagariRun& r = h_runs[k];
			if (r.count == 0 || r.after == 0 || r.after > n_literals || r.out_start < prev_end ||
			    (uint64_t)r.out_start + r.count > pl->stream_values || (uint64_t)r.out_start != (uint64_t)r.after + repeats)
				return fail(AKO_BROKEN_INPUT, "inconsistent run records%s%s");
			repeats += r.count;
			prev_end = (uint64_t)r.out_start + r.count;
		}
		if (n_literals + repeats != pl->stream_values)
			return fail(AKO_BROKEN_INPUT, "run records do not add up to the stream length%s%s");
	}
	if (!pl->kg)
		pl->kg = new KagariState;  // decoder side needs none of the encoder's buffers
	KagariState* k = pl->kg;
	if (n_literals > k->literal_capacity)
	{
		if (k->d_literals)
			(void)hipFree(k->d_literals), k->d_literals = nullptr;
		k->literal_capacity = 0;
		if (hipMalloc((void**)&k->d_literals, (n_literals + n_literals / 4 + 1024) * 2) != hipSuccess)
			return fail(AKO_NO_ENOUGH_MEMORY, "hipMalloc(device entropy stage) failed%s%s");
		k->literal_capacity = n_literals + n_literals / 4 + 1024;
	}
	if (n_runs > k->run_record_capacity)
	{
		if (k->d_runs)
			(void)hipFree(k->d_runs), k->d_runs = nullptr;
		k->run_record_capacity = 0;
		if (hipMalloc((void**)&k->d_runs, (n_runs + n_runs / 4 + 1024) * sizeof(KgRun)) != hipSuccess)
			return fail(AKO_NO_ENOUGH_MEMORY, "hipMalloc(device entropy stage) failed%s%s");
		k->run_record_capacity = n_runs + n_runs / 4 + 1024;
	}
	hipStream_t st = pl->stream;
	if (int rc = upload_chunked(pl, k->d_literals, h_literals, n_literals * 2))
		return rc;
	if (n_runs)
		if (int rc = upload_chunked(pl, k->d_runs, h_runs, n_runs * sizeof(KgRun)))
			return rc;
	int16_t* out = static_cast<int16_t*>(d_streams) + image * pl->stream_values;
	const uint32_t blocks = (uint32_t)((pl->stream_values + KG_CHUNK - 1) / KG_CHUNK);
	hipLaunchKernelGGL(k_kg_expand, dim3(blocks), dim3(KG_THREADS), 0, st, (const int16_t*)k->d_literals,
	                   (const KgRun*)k->d_runs, (uint32_t)n_runs, (uint64_t)pl->stream_values, out);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(st));  // the host arrays may be released by the caller
	return 0;
}

int akoHipDecodeDownload(akoHipPlan* pl, void* h_images)
{
	if (!pl || !h_images)
		return fail(AKO_INVALID_INPUT, "null argument%s%s");
	HIP_TRY(hipSetDevice(pl->device));
	if (int rc = ensure_staging(pl))
		return rc;
	if (int rc = akoHipDecode(pl, pl->d_stream, pl->d_img))
		return rc;
	return download_chunked(pl, h_images, pl->d_img, akoHipPlanImageBytes(pl) * pl->batch);
}

}  // extern "C"
#pragma GCC visibility pop
