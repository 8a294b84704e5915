/*
 * ako_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see ako_oracle.h).
 *
 * Plain scalar C restatement of the reference's tile-wise transform path, written from the
 * closed-form description of the algorithm rather than from the
 * reference's loop structure:
 *
 *   - every 1-D lifting pass is "compute all T high-pass values, then all T low-pass values"
 *     with out-of-range taps resolved by ONE index function (orc_tap) instead of the
 *     reference's first / middle / last loop splits          (wavelet-dd137.c:57-702,
 *                                                              wavelet-cdf53.c:57-362,
 *                                                              wavelet-haar.c:30-113)
 *   - a 2-D level works on four dense quadrant arrays, not on the reference's in-place
 *     pitch-2w scratch layout                                 (lifting.c:43-76, 104-148)
 *   - the coefficient stream is written forwards from a table of offsets, not backwards
 *     from the end                                            (lifting.c:171-292, misc.c:229-288)
 *
 * Parity: PINNED against the compiled reference (oracle/_ref) and tests/golden/.
 */
#include "ako_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int16_t i16;

/* ------------------------------------------------------------------------------------------
 * Geometry
 * ---------------------------------------------------------------------------------------- */

size_t orcHalfUp(size_t v) /* "divide plus one" rule: misc.c:98 */
{
	return (v + 1) / 2;
}

size_t orcLevels(size_t w, size_t h) /* lifting.c:182 */
{
	size_t n = 0;
	while (w > 2 && h > 2)
	{
		w = orcHalfUp(w);
		h = orcHalfUp(h);
		n++;
	}
	return n;
}

size_t orcTileStreamBytes(size_t w, size_t h) /* misc.c:117-149 */
{
	size_t values = 0;
	while (w > 2 && h > 2)
	{
		w = orcHalfUp(w);
		h = orcHalfUp(h);
		values += 3 * w * h + 1; /* C, B, D and the lift head */
	}
	values += w * h; /* final low-pass */
	return values * sizeof(i16);
}

size_t orcTileExtent(size_t pos, size_t image_d, size_t tiles_d) /* misc.c:152-161 */
{
	if (tiles_d == 0)
		return image_d;
	return (pos + tiles_d > image_d) ? (image_d % tiles_d) : tiles_d;
}

size_t orcTilesNo(size_t image_w, size_t image_h, size_t tiles_d) /* misc.c:192-203 */
{
	if (tiles_d == 0)
		return 1;
	return ((image_w + tiles_d - 1) / tiles_d) * ((image_h + tiles_d - 1) / tiles_d);
}

/* Largest per-plane stream over the tiles of an image: misc.c:174-189 */
static size_t orc_max_tile_stream(size_t image_w, size_t image_h, size_t td)
{
	if (td == 0 || (td >= image_w && td >= image_h))
		return orcTileStreamBytes(image_w, image_h);
	if (image_w % td == 0 && image_h % td == 0)
		return orcTileStreamBytes(td, td);

	size_t rem_w = image_w % td, rem_h = image_h % td;
	size_t a = orcTileStreamBytes(td, td);
	size_t b = orcTileStreamBytes((rem_w < td) ? rem_w : td, td);
	size_t c = orcTileStreamBytes(td, (rem_h < td) ? rem_h : td);
	size_t m = a > b ? a : b;
	return m > c ? m : c;
}

/* ------------------------------------------------------------------------------------------
 * Quantizer and gate steps (float, host only): quantization.c:43-98.
 * The order of float operations is kept as in the reference so libm rounding agrees.
 * ---------------------------------------------------------------------------------------- */

static float orc_step_curve(float factor, float tile_w, float tile_h, float cur_w, float cur_h)
{
	const float scale = (512.0F * 0.73F);
	const float highs = 6.0F;

	const float side0 = sqrtf(tile_w * tile_h);
	const float side = sqrtf(cur_w * cur_h);
	const float total = log2f(side0) - 1.0F;
	const float lift = log2f(side) - 1.0F;

	const float linear = (lift / total);
	const float fade = powf(linear + 1.0F, highs) / powf(2.0F, highs);
	const float curve = powf(2.0F, (lift - 1.0F)) * fade;
	return roundf(curve * (factor / scale));
}

int16_t orcQuantStep(int factor, int mul, size_t tw, size_t th, size_t cw, size_t ch) /* quantization.c:84 */
{
	if (factor <= 0)
		return 1;
	float q = orc_step_curve((float)factor * (float)mul, (float)tw, (float)th, (float)cw, (float)ch);
	if (q < 1.0F)
		q = 1.0F;
	if (q > 32765.0F)
		q = 32765.0F;
	return (int16_t)q;
}

int16_t orcGateStep(int factor, int mul, size_t tw, size_t th, size_t cw, size_t ch) /* quantization.c:67 */
{
	if (factor <= 0)
		return 0;
	float g = orc_step_curve((float)factor * (float)mul, (float)tw, (float)th, (float)cw, (float)ch);
	if (g < 0.0F)
		g = 0.0F;
	if (g > 32765.0F)
		g = 32765.0F;
	return (int16_t)g;
}

/* ------------------------------------------------------------------------------------------
 * 1-D lifting
 *
 * Tap resolution (derived from wavelet-dd137.c:70-80,104-126,145-167,193-203 and
 * wavelet-cdf53.c:77-84,101-108): index i outside [0,T) becomes
 *   CLAMP  : nearest valid index
 *   MIRROR : nearest valid index for the near taps; the far taps (DD137's "+2" in the predict
 *            step and "-2" in the update step) take the index of the opposite near tap ('alt')
 *   REPEAT : i modulo T
 *   ZERO   : no index, the tap reads 0
 * ---------------------------------------------------------------------------------------- */

static inline ptrdiff_t orc_tap(ptrdiff_t i, ptrdiff_t T, int wrap, ptrdiff_t alt)
{
	if (i >= 0 && i < T)
		return i;
	switch (wrap)
	{
	case AKO_WRAP_CLAMP: return (i < 0) ? 0 : T - 1;
	case AKO_WRAP_MIRROR: return (alt >= 0) ? alt : ((i < 0) ? 0 : T - 1);
	case AKO_WRAP_REPEAT: return ((i % T) + T) % T;
	default: return -1;
	}
}

/* C truncating division, written out so the intent is visible */
static inline int orc_div(int a, int b)
{
	return a / b;
}

/* DD13/7 steps: wavelet-dd137.c:36-54 */
static inline int dd_p(int l1, int e, int p1, int p2) /* predict term, before /16 */
{
	return l1 + p2 - 9 * (e + p1);
}

static inline int dd_u(int l2, int l1, int h, int p1) /* update term, before /32 */
{
	return -l2 - p1 + 9 * (l1 + h);
}

#define SEQ(ptr, stride, idx) ((ptr)[(ptrdiff_t)(idx) * (stride)])

void orcLift1d(int wavelet, int wrap, size_t Tn, int fake_last, const i16* src, ptrdiff_t ss, i16* lp, ptrdiff_t ls,
               i16* hp, ptrdiff_t hs)
{
	const ptrdiff_t T = (ptrdiff_t)Tn;

/* even / odd source coefficients; the phantom last odd is a copy of the last even
 * (wavelet-dd137.c:128-132, wavelet-cdf53.c:86-90, wavelet-haar.c:44-53) */
#define EVEN(i) ((int)SEQ(src, ss, 2 * (i)))
#define ODD(i) ((fake_last && (i) == T - 1) ? EVEN(i) : (int)SEQ(src, ss, 2 * (i) + 1))
#define EVEN_TAP(i, alt) (((k = orc_tap((i), T, wrap, (alt))) < 0) ? 0 : EVEN(k))
#define HP_TAP(i, alt) (((k = orc_tap((i), T, wrap, (alt))) < 0) ? 0 : (int)SEQ(hp, hs, k))

	ptrdiff_t k;

	if (wavelet == AKO_WAVELET_HAAR)
	{
		for (ptrdiff_t c = 0; c < T; c++)
		{
			SEQ(hp, hs, c) = (i16)(ODD(c) - EVEN(c));
			SEQ(lp, ls, c) = (i16)EVEN(c);
		}
	}
	else if (wavelet == AKO_WAVELET_CDF53)
	{
		for (ptrdiff_t c = 0; c < T; c++)
			SEQ(hp, hs, c) = (i16)(ODD(c) - orc_div(EVEN(c) + EVEN_TAP(c + 1, -1), 2));
		for (ptrdiff_t c = 0; c < T; c++)
			SEQ(lp, ls, c) = (i16)(EVEN(c) + orc_div(HP_TAP(c - 1, -1) + (int)SEQ(hp, hs, c), 4));
	}
	else
	{
		for (ptrdiff_t c = 0; c < T; c++)
		{
			const int l1 = EVEN_TAP(c - 1, -1);
			const int p1 = EVEN_TAP(c + 1, -1);
			const int p2 = EVEN_TAP(c + 2, c - 1);
			SEQ(hp, hs, c) = (i16)(ODD(c) + orc_div(dd_p(l1, EVEN(c), p1, p2), 16));
		}
		for (ptrdiff_t c = 0; c < T; c++)
		{
			const int l2 = HP_TAP(c - 2, c + 1);
			const int l1 = HP_TAP(c - 1, -1);
			const int p1 = HP_TAP(c + 1, -1);
			SEQ(lp, ls, c) = (i16)(EVEN(c) + orc_div(dd_u(l2, l1, (int)SEQ(hp, hs, c), p1), 32));
		}
	}

#undef EVEN
#undef ODD
#undef EVEN_TAP
#undef HP_TAP
}

void orcUnlift1d(int wavelet, int wrap, size_t Tn, const i16* lp, ptrdiff_t ls, const i16* hp, ptrdiff_t hs, i16* ev,
                 ptrdiff_t es, i16* od, ptrdiff_t os)
{
	const ptrdiff_t T = (ptrdiff_t)Tn;
	ptrdiff_t k;

	/* The outputs may alias the inputs element-for-element (ev == lp, od == hp), exactly like
	 * the reference's "in-place-ish" vertical pass (wavelet-dd137.c:540): evens only read
	 * high-pass values, odds only read finished evens and their own high-pass value. */

#define HP_TAP(i, alt) (((k = orc_tap((i), T, wrap, (alt))) < 0) ? 0 : (int)SEQ(hp, hs, k))
#define EV_TAP(i, alt) (((k = orc_tap((i), T, wrap, (alt))) < 0) ? 0 : (int)SEQ(ev, es, k))

	if (wavelet == AKO_WAVELET_HAAR)
	{
		for (ptrdiff_t c = 0; c < T; c++)
		{
			const int l = SEQ(lp, ls, c), h = SEQ(hp, hs, c);
			SEQ(ev, es, c) = (i16)l;
			SEQ(od, os, c) = (i16)(l + h);
		}
	}
	else if (wavelet == AKO_WAVELET_CDF53)
	{
		for (ptrdiff_t c = 0; c < T; c++)
			SEQ(ev, es, c) = (i16)((int)SEQ(lp, ls, c) - orc_div(HP_TAP(c - 1, -1) + (int)SEQ(hp, hs, c), 4));
		for (ptrdiff_t c = 0; c < T; c++)
			SEQ(od, os, c) = (i16)((int)SEQ(hp, hs, c) + orc_div((int)SEQ(ev, es, c) + EV_TAP(c + 1, -1), 2));
	}
	else
	{
		for (ptrdiff_t c = 0; c < T; c++)
		{
			const int l2 = HP_TAP(c - 2, c + 1);
			const int l1 = HP_TAP(c - 1, -1);
			const int p1 = HP_TAP(c + 1, -1);
			SEQ(ev, es, c) = (i16)((int)SEQ(lp, ls, c) - orc_div(dd_u(l2, l1, (int)SEQ(hp, hs, c), p1), 32));
		}
		for (ptrdiff_t c = 0; c < T; c++)
		{
			const int l1 = EV_TAP(c - 1, -1);
			const int p1 = EV_TAP(c + 1, -1);
			const int p2 = EV_TAP(c + 2, c - 1);
			SEQ(od, os, c) = (i16)((int)SEQ(hp, hs, c) - orc_div(dd_p(l1, (int)SEQ(ev, es, c), p1, p2), 16));
		}
	}

#undef HP_TAP
#undef EV_TAP
}

/* ------------------------------------------------------------------------------------------
 * One 2-D level on dense quadrants
 * ---------------------------------------------------------------------------------------- */

static int orc_level_kind(int wavelet, size_t tgt_w, size_t tgt_h) /* lifting.c:49,58,118,126 */
{
	if (wavelet == AKO_WAVELET_HAAR)
		return AKO_WAVELET_HAAR;
	if (wavelet == AKO_WAVELET_CDF53 || tgt_w < 8 || tgt_h < 8)
		return AKO_WAVELET_CDF53;
	return AKO_WAVELET_DD137;
}

void orcLevelForward(int kind, int wrap, size_t cw, size_t ch, const i16* src, size_t pitch, i16* ll, i16* b, i16* c,
                     i16* d)
{
	const size_t tw = orcHalfUp(cw), th = orcHalfUp(ch);
	const int fake_col = (2 * tw != cw), fake_row = (2 * th != ch);

	/* rows first: each source row becomes [LP(tw) | HP(tw)]; an odd height repeats the last
	 * lifted row so that the column pass sees 2*th rows (lifting.c:69-74) */
	i16* mid = malloc(sizeof(i16) * 2 * tw * 2 * th);
	for (size_t r = 0; r < ch; r++)
		orcLift1d(kind, wrap, tw, fake_col, src + r * pitch, 1, mid + r * 2 * tw, 1, mid + r * 2 * tw + tw, 1);
	if (fake_row)
		memcpy(mid + ch * 2 * tw, mid + (ch - 1) * 2 * tw, sizeof(i16) * 2 * tw);

	/* then columns: left half (row low-pass) gives LL over C, right half gives B over D */
	for (size_t x = 0; x < tw; x++)
	{
		orcLift1d(kind, wrap, th, 0, mid + x, (ptrdiff_t)(2 * tw), ll + x, (ptrdiff_t)tw, c + x, (ptrdiff_t)tw);
		orcLift1d(kind, wrap, th, 0, mid + tw + x, (ptrdiff_t)(2 * tw), b + x, (ptrdiff_t)tw, d + x, (ptrdiff_t)tw);
	}
	free(mid);
}

void orcLevelInverse(int kind, int wrap, size_t sw, size_t sh, size_t tgt_w, size_t tgt_h, const i16* ll, const i16* c,
                     const i16* b, const i16* d, i16* out)
{
	/* columns first (lifting.c:137-138): (LL, C) -> left half rows, (B, D) -> right half rows */
	i16* left = malloc(sizeof(i16) * sw * 2 * sh);
	i16* right = malloc(sizeof(i16) * sw * 2 * sh);
	for (size_t x = 0; x < sw; x++)
	{
		orcUnlift1d(kind, wrap, sh, ll + x, (ptrdiff_t)sw, c + x, (ptrdiff_t)sw, left + x, (ptrdiff_t)(2 * sw),
		            left + sw + x, (ptrdiff_t)(2 * sw));
		orcUnlift1d(kind, wrap, sh, b + x, (ptrdiff_t)sw, d + x, (ptrdiff_t)sw, right + x, (ptrdiff_t)(2 * sw),
		            right + sw + x, (ptrdiff_t)(2 * sw));
	}

	/* then rows (lifting.c:140-142); the phantom last row / column is dropped */
	i16* row = malloc(sizeof(i16) * 2 * sw);
	for (size_t y = 0; y < tgt_h; y++)
	{
		orcUnlift1d(kind, wrap, sw, left + y * sw, 1, right + y * sw, 1, row, 2, row + 1, 2);
		memcpy(out + y * tgt_w, row, sizeof(i16) * tgt_w);
	}
	free(row);
	free(left);
	free(right);
}

/* ------------------------------------------------------------------------------------------
 * Pyramid of one plane <-> its share of the coefficient stream
 *
 * Stream layout (lifting.c:179-285 writes it backwards, misc.c:245-285 reads it):
 *   LP[ch 0] .. LP[ch N-1], then for level = smallest .. largest, for ch = 0 .. N-1:
 *   head(q)  C  B  D
 * ---------------------------------------------------------------------------------------- */

struct orc_level
{
	size_t cur_w, cur_h, tgt_w, tgt_h;
};

static size_t orc_level_table(size_t w, size_t h, struct orc_level* out /* >= 40 entries */)
{
	size_t n = 0;
	while (w > 2 && h > 2)
	{
		out[n].cur_w = w;
		out[n].cur_h = h;
		w = orcHalfUp(w);
		h = orcHalfUp(h);
		out[n].tgt_w = w;
		out[n].tgt_h = h;
		n++;
	}
	return n;
}

/* offset (in int16 units) of the [head C B D] group of (level index from largest = 0, channel) */
static size_t orc_group_offset(const struct orc_level* lv, size_t levels, size_t channels, size_t final_w,
                               size_t final_h, size_t level, size_t ch)
{
	size_t off = final_w * final_h * channels;
	for (size_t l = levels; l-- > level + 1;)
		off += channels * (3 * lv[l].tgt_w * lv[l].tgt_h + 1);
	return off + ch * (3 * lv[level].tgt_w * lv[level].tgt_h + 1);
}

static inline i16 orc_quantize(i16 v, i16 q, i16 g) /* lifting.c:163 */
{
	return (v < -g || v > g) ? (i16)(v / q) : 0;
}

/* Forward pyramid of one plane.  q_of / g_of give the steps per level (NULL = lossless). */
static void orc_plane_forward(int wavelet, int wrap, size_t channels, size_t ch, size_t w, size_t h, i16* plane,
                              const i16* q_of, const i16* g_of, i16* stream)
{
	struct orc_level lv[48];
	const size_t levels = orc_level_table(w, h, lv);
	const size_t fw = levels ? lv[levels - 1].tgt_w : w;
	const size_t fh = levels ? lv[levels - 1].tgt_h : h;

	i16* cur = plane; /* dense cur_w x cur_h */
	i16* ll = malloc(sizeof(i16) * (orcHalfUp(w) * orcHalfUp(h) + 1));
	i16* quad = malloc(sizeof(i16) * (3 * orcHalfUp(w) * orcHalfUp(h) + 1));

	for (size_t l = 0; l < levels; l++)
	{
		const size_t n = lv[l].tgt_w * lv[l].tgt_h;
		const int kind = orc_level_kind(wavelet, lv[l].tgt_w, lv[l].tgt_h);
		orcLevelForward(kind, wrap, lv[l].cur_w, lv[l].cur_h, cur, lv[l].cur_w, ll, quad + n, quad, quad + 2 * n);

		i16 q = q_of ? q_of[l] : 1;
		const i16 g = g_of ? g_of[l] : 0;
		i16* grp = stream + orc_group_offset(lv, levels, channels, fw, fh, l, ch);
		grp[0] = q; /* lift head stores what akoQuantization returned: lifting.c:267 */
		if (q < 1)
			q = 1; /* lifting.c:157 */
		for (size_t i = 0; i < 3 * n; i++)
			grp[1 + i] = orc_quantize(quad[i], q, g);

		memcpy(plane, ll, sizeof(i16) * n); /* LL becomes the next level's dense input */
		cur = plane;
	}

	memcpy(stream + fw * fh * ch, cur, sizeof(i16) * fw * fh);
	free(ll);
	free(quad);
}

static void orc_plane_inverse(int wavelet, int wrap, size_t channels, size_t ch, size_t w, size_t h, const i16* stream,
                              i16* plane)
{
	struct orc_level lv[48];
	const size_t levels = orc_level_table(w, h, lv);
	const size_t fw = levels ? lv[levels - 1].tgt_w : w;
	const size_t fh = levels ? lv[levels - 1].tgt_h : h;

	i16* ll = malloc(sizeof(i16) * (w * h + 1));
	i16* quad = malloc(sizeof(i16) * (3 * orcHalfUp(w) * orcHalfUp(h) + 1));
	memcpy(ll, stream + fw * fh * ch, sizeof(i16) * fw * fh);

	for (size_t l = levels; l-- > 0;)
	{
		const size_t n = lv[l].tgt_w * lv[l].tgt_h;
		const i16* grp = stream + orc_group_offset(lv, levels, channels, fw, fh, l, ch);
		const i16 q = grp[0];
		for (size_t i = 0; i < 3 * n; i++) /* lifting.c:30-40: multiply only when q > 1, wraps to int16 */
			quad[i] = (q > 1) ? (i16)(grp[1 + i] * q) : grp[1 + i];

		const int kind = orc_level_kind(wavelet, lv[l].tgt_w, lv[l].tgt_h);
		orcLevelInverse(kind, wrap, lv[l].tgt_w, lv[l].tgt_h, lv[l].cur_w, lv[l].cur_h, ll, quad, quad + n,
		                quad + 2 * n, plane);
		memcpy(ll, plane, sizeof(i16) * lv[l].cur_w * lv[l].cur_h);
	}

	memcpy(plane, ll, sizeof(i16) * w * h);
	free(ll);
	free(quad);
}

int orcLiftPlane(int wavelet, int wrap, size_t w, size_t h, const i16* plane, i16* stream)
{
	i16* work = malloc(sizeof(i16) * w * h);
	if (!work)
		return 1;
	memcpy(work, plane, sizeof(i16) * w * h);
	orc_plane_forward(wavelet, wrap, 1, 0, w, h, work, NULL, NULL, stream);
	free(work);
	return 0;
}

int orcUnliftPlane(int wavelet, int wrap, size_t w, size_t h, const i16* stream, i16* plane)
{
	orc_plane_inverse(wavelet, wrap, 1, 0, w, h, stream, plane);
	return 0;
}

/* ------------------------------------------------------------------------------------------
 * Tile: u8 window <-> stream
 * ---------------------------------------------------------------------------------------- */

static inline i16 orc_sat8(i16 v) /* format.c:155-157 */
{
	return (v > 0) ? ((v < 255) ? v : 255) : 0;
}

int orcEncodeTile(const struct akoSettings* s, size_t channels, size_t tw, size_t th, size_t image_w, const uint8_t* in,
                  i16* stream)
{
	const size_t n = tw * th;
	i16* planes = malloc(sizeof(i16) * n * channels);
	if (!planes)
		return 1;

	/* de-interleave (+ optional "discard pixels under a zero alpha": format.c:38-49, only 2 / 4 channels) */
	const int discard = s->discard_non_visible && (channels == 2 || channels == 4);
	for (size_t y = 0; y < th; y++)
		for (size_t x = 0; x < tw; x++)
		{
			const uint8_t* px = in + (y * image_w + x) * channels;
			const int hide = discard && px[channels - 1] == 0;
			for (size_t ch = 0; ch < channels; ch++)
				planes[ch * n + y * tw + x] = (hide && ch != channels - 1) ? 0 : px[ch];
		}

	/* forward colour: format.c:87-134 */
	if (channels >= 3 && s->color != AKO_COLOR_NONE)
		for (size_t i = 0; i < n; i++)
		{
			const i16 r = planes[i], g = planes[n + i], b = planes[2 * n + i];
			if (s->color == AKO_COLOR_SUBTRACT_G)
			{
				planes[i] = g;
				planes[n + i] = (i16)(r - g);
				planes[2 * n + i] = (i16)(b - g);
			}
			else
			{
				const i16 co = (i16)(r - b);
				const i16 t = (i16)(b + (r - b) / 2);
				const i16 cg = (i16)(g - t);
				const int y = t + (g - t) / 2;
				planes[i] = (s->color == AKO_COLOR_YCOCG_Q) ? (i16)(y * 2) : (i16)y;
				planes[n + i] = co;
				planes[2 * n + i] = cg;
			}
		}

	if (s->wavelet == AKO_WAVELET_NONE)
	{
		memcpy(stream, planes, sizeof(i16) * n * channels); /* encode.c:127-128,151 */
		free(planes);
		return 0;
	}

	/* per level steps: plane 0 uses multiplier 1, every other plane chroma_loss + 1 (lifting.c:202-211) */
	struct orc_level lv[48];
	const size_t levels = orc_level_table(tw, th, lv);
	i16 q[48], g[48];
	for (size_t ch = 0; ch < channels; ch++)
	{
		const int mul = (ch == 0) ? 1 : s->chroma_loss + 1;
		for (size_t l = 0; l < levels; l++)
		{
			q[l] = orcQuantStep(s->quantization, mul, tw, th, lv[l].cur_w, lv[l].cur_h);
			g[l] = orcGateStep(s->gate, mul, tw, th, lv[l].cur_w, lv[l].cur_h);
		}
		orc_plane_forward(s->wavelet, s->wrap, channels, ch, tw, th, planes + ch * n, q, g, stream);
	}

	free(planes);
	return 0;
}

int orcDecodeTile(const struct akoSettings* s, size_t channels, size_t tw, size_t th, size_t image_w, const i16* stream,
                  uint8_t* out)
{
	const size_t n = tw * th;
	i16* planes = malloc(sizeof(i16) * n * channels);
	if (!planes)
		return 1;

	if (s->wavelet == AKO_WAVELET_NONE)
		memcpy(planes, stream, sizeof(i16) * n * channels);
	else
		for (size_t ch = 0; ch < channels; ch++)
			orc_plane_inverse(s->wavelet, s->wrap, channels, ch, tw, th, stream, planes + ch * n);

	/* inverse colour + saturation: format.c:138-229 */
	for (size_t i = 0; i < n; i++)
	{
		if (channels >= 3 && s->color != AKO_COLOR_NONE)
		{
			i16 y = planes[i];
			const i16 u = planes[n + i], v = planes[2 * n + i];
			i16 r, g, b;
			if (s->color == AKO_COLOR_SUBTRACT_G)
			{
				r = (i16)(u + y);
				g = y;
				b = (i16)(v + y);
			}
			else
			{
				if (s->color == AKO_COLOR_YCOCG_Q)
					y = (i16)(y / 2);
				const i16 t = (i16)(y - (v / 2));
				g = (i16)(v + t);
				b = (i16)(t - (u / 2));
				r = (i16)(b + u);
			}
			planes[i] = r;
			planes[n + i] = g;
			planes[2 * n + i] = b;
		}
	}

	for (size_t y = 0; y < th; y++)
		for (size_t x = 0; x < tw; x++)
			for (size_t ch = 0; ch < channels; ch++)
				out[(y * image_w + x) * channels + ch] = (uint8_t)orc_sat8(planes[ch * n + y * tw + x]);

	free(planes);
	return 0;
}

/* ------------------------------------------------------------------------------------------
 * Header: head.c:34-169
 * ---------------------------------------------------------------------------------------- */

static enum akoStatus orc_validate(size_t channels, size_t w, size_t h, size_t td, int wrap, int wavelet, int color,
                                   int compression)
{
	if (channels > AKO_MAX_CHANNELS)
		return AKO_INVALID_CHANNELS_NO;
	if (w == 0 || h == 0 || w > AKO_MAX_WIDTH || h > AKO_MAX_HEIGHT)
		return AKO_INVALID_DIMENSIONS;
	if (td != 0 && (td < AKO_MIN_TILES_DIMENSION || td > AKO_MAX_TILES_DIMENSION))
		return AKO_INVALID_TILES_DIMENSIONS;
	if (wrap < AKO_WRAP_CLAMP || wrap > AKO_WRAP_ZERO)
		return AKO_INVALID_WRAP_MODE;
	if (wavelet < AKO_WAVELET_DD137 || wavelet > AKO_WAVELET_NONE)
		return AKO_INVALID_WAVELET_TRANSFORMATION;
	if (color < AKO_COLOR_YCOCG || color > AKO_COLOR_YCOCG_Q)
		return AKO_INVALID_COLOR_TRANSFORMATION;
	if (compression < AKO_COMPRESSION_KAGARI || compression > AKO_COMPRESSION_NONE)
		return AKO_INVALID_COMPRESSION_METHOD;
	return AKO_OK;
}

enum akoStatus orcHeadWrite(size_t channels, size_t w, size_t h, const struct akoSettings* s, void* out16)
{
	size_t td_field = 0;
	if (s->tiles_dimension != 0)
	{
		size_t lg = 0;
		while (((size_t)1 << (lg + 1)) <= s->tiles_dimension)
			lg++;
		if (((size_t)1 << lg) != s->tiles_dimension)
			return AKO_INVALID_TILES_DIMENSIONS;
		td_field = lg - 2; /* underflows for 1, 2: matches the reference, caught by validation below */
	}

	const enum akoStatus v = orc_validate(channels, w, h, s->tiles_dimension, (int)s->wrap, (int)s->wavelet,
	                                      (int)s->color, (int)s->compression);
	if (v != AKO_OK)
		return v;

	uint8_t* o = out16;
	const uint32_t w32 = (uint32_t)w, h32 = (uint32_t)h;
	const uint32_t flags = (uint32_t)(channels - 1) | ((uint32_t)s->wrap << 4) | ((uint32_t)s->wavelet << 6) |
	                       ((uint32_t)s->color << 8) | ((uint32_t)s->compression << 10) | ((uint32_t)td_field << 12);
	o[0] = 'A';
	o[1] = 'k';
	o[2] = 'o';
	o[3] = AKO_FORMAT_VERSION;
	memcpy(o + 4, &w32, 4);
	memcpy(o + 8, &h32, 4);
	memcpy(o + 12, &flags, 4);
	return AKO_OK;
}

enum akoStatus orcHeadRead(const void* in16, size_t* channels, size_t* w, size_t* h, struct akoSettings* s)
{
	const uint8_t* i = in16;
	uint32_t w32, h32, flags;
	if (i[0] != 'A' || i[1] != 'k' || i[2] != 'o')
		return AKO_INVALID_MAGIC;
	if (i[3] != AKO_FORMAT_VERSION)
		return AKO_UNSUPPORTED_VERSION;
	memcpy(&w32, i + 4, 4);
	memcpy(&h32, i + 8, 4);
	memcpy(&flags, i + 12, 4);

	if ((flags >> 15) != 0) /* head.c:124 -- this also rejects tiles >= 1024 */
		return AKO_INVALID_FLAGS;

	const size_t ch = (flags & 15) + 1;
	const int wrap = (flags >> 4) & 3, wavelet = (flags >> 6) & 3, color = (flags >> 8) & 3;
	const int compression = (flags >> 10) & 3;
	size_t td = (flags >> 12) & 31;
	if (td != 0)
	{
		if (td >= 30)
			return AKO_INVALID_TILES_DIMENSIONS;
		td = (size_t)1 << (td + 2);
	}

	const enum akoStatus v = orc_validate(ch, w32, h32, td, wrap, wavelet, color, compression);
	if (v != AKO_OK)
		return v;

	if (channels)
		*channels = ch;
	if (w)
		*w = w32;
	if (h)
		*h = h32;
	if (s)
	{
		s->wrap = (enum akoWrap)wrap;
		s->wavelet = (enum akoWavelet)wavelet;
		s->color = (enum akoColor)color;
		s->compression = (enum akoCompression)compression;
		s->tiles_dimension = td;
	}
	return AKO_OK;
}

/* ------------------------------------------------------------------------------------------
 * Kagari: zig-zag + Elias-gamma values, run lengths after two repeats (kagari.c:59-366)
 *
 * Bit-stream facts restated from the reference:
 *  - a value v >= 1 of bit length L+1 is written as L zero bits followed by the L+1 bits of v
 *  - bits are packed MSB first through a 64 bit accumulator; it is flushed by whole bytes only
 *    when the next code would not fit (and more than 8 bits are pending)      (kagari.c:64-78)
 *  - coefficient c is sent as zigzag(c) + 1; after the 2nd repeat of a value the number of
 *    further repeats r is sent as r - 2 + 1 once the run ends                  (kagari.c:34,194-198)
 * ---------------------------------------------------------------------------------------- */

struct orc_bits
{
	uint64_t acc;
	int used;
	uint8_t* cur;
	const uint8_t* end;
	uint8_t* start;
};

static int orc_put_gamma(struct orc_bits* b, uint16_t v)
{
	int len = 0;
	for (uint16_t t = v; t > 1; t >>= 1)
		len++;
	const int total = 2 * len + 1;

	if (b->used > 8 && b->used + total > 64)
	{
		if (b->cur + (b->used / 8) >= b->end)
			return 0;
		do
		{
			b->used -= 8;
			*b->cur++ = (uint8_t)(b->acc >> b->used);
		} while (b->used + total > 64);
	}

	b->used += total;
	b->acc = (b->acc << total) | v;
	return total;
}

static size_t orc_put_end(struct orc_bits* b)
{
	while (b->used / 8 != 0)
	{
		if (b->cur + 1 >= b->end)
			return 0;
		b->used -= 8;
		*b->cur++ = (uint8_t)(b->acc >> b->used);
	}
	if (b->used != 0)
	{
		if (b->cur + 1 >= b->end)
			return 0;
		*b->cur++ = (uint8_t)(b->acc << (8 - b->used));
	}
	return (size_t)(b->cur - b->start);
}

static inline uint16_t orc_zigzag(i16 v) /* kagari.c:169-173 */
{
	return (uint16_t)(((uint32_t)(int32_t)v << 1) ^ (uint32_t)((int32_t)v >> 15));
}

static inline i16 orc_unzigzag(uint16_t v) /* kagari.c:175-178 */
{
	return (i16)((v >> 1) ^ (uint16_t)(~(v & 1) + 1));
}

size_t orcKagariEncode(size_t input_bytes, size_t capacity, const void* input, void* output)
{
	if (capacity == 0 || input_bytes == 0 || (input_bytes % 2) != 0)
		return 0;

	struct orc_bits b = {0, 0, output, (const uint8_t*)output + capacity, output};
	const i16* in = input;
	const size_t n = input_bytes / 2;

#define PUT_VALUE(v) \
	if (orc_put_gamma(&b, (uint16_t)(orc_zigzag(v) + 1)) == 0) \
	return 0
#define PUT_RUN(r) \
	if (orc_put_gamma(&b, (uint16_t)((r)-2 + 1)) == 0) \
	return 0

	PUT_VALUE(in[0]);
	i16 prev = in[0];
	uint16_t run = 0; /* repeats of 'prev' seen after its first occurrence */

	for (size_t i = 1; i < n; i++)
	{
		if (in[i] == prev)
		{
			run++;
			if (run <= 2)
			{
				PUT_VALUE(in[i]);
			}
			else if (run == 65535 - 1) /* run counter about to overflow: kagari.c:262 */
			{
				PUT_RUN(run);
				run = 0;
			}
		}
		else
		{
			if (run >= 2)
			{
				PUT_RUN(run);
			}
			PUT_VALUE(in[i]);
			prev = in[i];
			run = 0;
		}
	}
	if (run >= 2)
	{
		PUT_RUN(run);
	}
#undef PUT_VALUE
#undef PUT_RUN

	return orc_put_end(&b);
}

struct orc_rbits
{
	uint64_t acc;
	int used;
	const uint8_t* cur;
	const uint8_t* end;
};

static uint16_t orc_get_gamma(struct orc_rbits* b, int* bits) /* kagari.c:113-163 */
{
	if (b->acc == 0 || b->used < 32)
	{
		if (b->cur + ((64 - b->used) / 8) < b->end)
		{
			do
			{
				b->used += 8;
				b->acc |= (uint64_t)(*b->cur++) << (64 - b->used);
			} while (b->used < 56);
		}
		else
		{
			while (b->used < 56 && b->cur < b->end)
			{
				b->used += 8;
				b->acc |= (uint64_t)(*b->cur++) << (64 - b->used);
			}
		}
		if (b->acc == 0)
			return 0;
	}

	const uint32_t top = (uint32_t)(b->acc >> 32);
	const int zeros = top ? __builtin_clz(top) : 32;
	const int total = zeros * 2 + 1;
	if (total > b->used)
		return 0;

	*bits = total;
	const uint16_t v = (uint16_t)(b->acc >> (64 - total));
	b->acc <<= total;
	b->used -= total;
	return v;
}

size_t orcKagariDecode(size_t no, size_t input_bytes, size_t output_bytes, const void* input, void* output)
{
	if (output_bytes == 0 || input_bytes == 0 || no == 0 || (output_bytes % 2) != 0)
		return 0;

	struct orc_rbits b = {0, 0, input, (const uint8_t*)input + input_bytes};
	i16* out = output;
	const i16* out_end = (const i16*)((const uint8_t*)output + output_bytes);
	int bits = 0;

	uint16_t code = orc_get_gamma(&b, &bits);
	if (bits == 0)
		return 0;
	i16 prev = orc_unzigzag((uint16_t)(code - 1));
	*out++ = prev;
	no--;

	uint16_t run = 0;
	for (; no != 0; no--)
	{
		if (out == out_end)
			return 0;
		bits = 0;
		code = orc_get_gamma(&b, &bits);
		if (bits == 0)
			return 0;
		const i16 v = orc_unzigzag((uint16_t)(code - 1));

		*out++ = v;
		if (v == prev)
		{
			if (++run == 2)
			{
				bits = 0;
				code = orc_get_gamma(&b, &bits);
				if (bits == 0)
					return 0;
				const uint16_t len = (uint16_t)(code - 1);
				if (out + (size_t)len > out_end)
					return 0;
				for (uint16_t u = 0; u < len; u++)
					*out++ = prev;
				run = 0;
				no -= len;
			}
		}
		else
		{
			prev = v;
			run = 0;
		}
	}
	return (size_t)(b.cur - (const uint8_t*)input);
}

/* ------------------------------------------------------------------------------------------
 * Image level: encode.c:38-232, decode.c:38-250
 * ---------------------------------------------------------------------------------------- */

static __thread double orc_transform_seconds;

static double orc_now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double orcLastTransformSeconds(void)
{
	return orc_transform_seconds;
}

size_t orcEncodeImage(const struct akoSettings* s_in, size_t channels, size_t image_w, size_t image_h, const void* in,
                      void** out, enum akoStatus* status)
{
	struct akoSettings s = *s_in;
	enum akoStatus st = AKO_OK;
	uint8_t* blob = NULL;
	i16* stream = NULL;
	uint8_t* packed = NULL;
	size_t blob_size = 16;
	orc_transform_seconds = 0.0;

	/* colour fix-up: encode.c:59-64 */
	if (s.color == AKO_COLOR_YCOCG && (s.quantization > 0 || s.gate > 0))
		s.color = AKO_COLOR_YCOCG_Q;
	else if (s.color == AKO_COLOR_YCOCG_Q && (s.quantization <= 0 && s.gate <= 0))
		s.color = AKO_COLOR_YCOCG;

	if (in == NULL)
	{
		st = AKO_INVALID_INPUT;
		goto fail;
	}
	if ((blob = malloc(blob_size)) == NULL)
	{
		st = AKO_NO_ENOUGH_MEMORY;
		goto fail;
	}
	if ((st = orcHeadWrite(channels, image_w, image_h, &s, blob)) != AKO_OK)
		goto fail;

	const size_t td = s.tiles_dimension;
	const size_t tiles = orcTilesNo(image_w, image_h, td);
	size_t max_bytes = orc_max_tile_stream(image_w, image_h, td) * channels;
	{
		/* wavelet NONE streams are the raw planes */
		const size_t mw = orcTileExtent(0, image_w, td), mh = orcTileExtent(0, image_h, td);
		if (max_bytes < mw * mh * channels * 2)
			max_bytes = mw * mh * channels * 2;
	}
	stream = malloc(max_bytes + 16);
	packed = malloc(max_bytes + 16);
	if (!stream || !packed)
	{
		st = AKO_NO_ENOUGH_MEMORY;
		goto fail;
	}

	size_t tx = 0, ty = 0;
	for (size_t t = 0; t < tiles; t++)
	{
		const size_t tw = orcTileExtent(tx, image_w, td), th = orcTileExtent(ty, image_h, td);
		const size_t bytes = (s.wavelet != AKO_WAVELET_NONE) ? orcTileStreamBytes(tw, th) * channels
		                                                      : tw * th * channels * sizeof(i16);

		/* The reference mis-reads tiles that never enter the lift loop (w <= 2 or h <= 2): its final
		 * low-pass copy assumes a pitch of 2w (lifting.c:285).  Not reproducible
		 * (it reads uninitialised spacing), so this restatement and the product both refuse them. */
		if (s.wavelet != AKO_WAVELET_NONE && (tw <= 2 || th <= 2))
		{
			st = AKO_ERROR;
			goto fail;
		}

		const double t0 = orc_now();
		if (orcEncodeTile(&s, channels, tw, th, image_w, (const uint8_t*)in + (image_w * ty + tx) * channels,
		                  stream) != 0)
		{
			st = AKO_NO_ENOUGH_MEMORY;
			goto fail;
		}
		orc_transform_seconds += orc_now() - t0;

		const uint8_t* from = (const uint8_t*)stream;
		size_t out_bytes = bytes;
		if (s.compression != AKO_COMPRESSION_NONE)
		{
			/* compression.c:36-55: capacity equals the raw size, 4 byte block head in front.
			 * NB the reference always sizes the block as a wavelet stream (akoTileDataSize). */
			const size_t raw = orcTileStreamBytes(tw, th) * channels;
			const size_t c = orcKagariEncode(raw, raw - 4, stream, packed + 4);
			if (c == 0)
			{
				st = AKO_ERROR;
				goto fail;
			}
			const uint32_t c32 = (uint32_t)c;
			memcpy(packed, &c32, 4);
			from = packed;
			out_bytes = c + 4;
		}

		uint8_t* grown = realloc(blob, blob_size + out_bytes);
		if (!grown)
		{
			st = AKO_NO_ENOUGH_MEMORY;
			goto fail;
		}
		blob = grown;
		memcpy(blob + blob_size, from, out_bytes);
		blob_size += out_bytes;

		tx += td;
		if (tx >= image_w)
		{
			tx = 0;
			ty += td;
		}
	}

	free(stream);
	free(packed);
	if (status)
		*status = AKO_OK;
	if (out)
		*out = blob;
	else
		free(blob);
	return blob_size;

fail:
	free(stream);
	free(packed);
	free(blob);
	if (status)
		*status = st;
	return 0;
}

uint8_t* orcDecodeImage(size_t input_size, const void* input, struct akoSettings* out_s, size_t* out_channels,
                        size_t* out_w, size_t* out_h, enum akoStatus* status)
{
	struct akoSettings s;
	memset(&s, 0, sizeof(s));
	enum akoStatus st = AKO_OK;
	size_t channels = 0, image_w = 0, image_h = 0;
	uint8_t* image = NULL;
	i16* stream = NULL;
	orc_transform_seconds = 0.0;

	if (input == NULL)
	{
		st = AKO_INVALID_INPUT;
		goto fail;
	}
	if ((st = orcHeadRead(input, &channels, &image_w, &image_h, &s)) != AKO_OK)
		goto fail;

	const uint8_t* cur = (const uint8_t*)input + 16;
	const uint8_t* end = (const uint8_t*)input + input_size;
	const size_t td = s.tiles_dimension;
	const size_t tiles = orcTilesNo(image_w, image_h, td);
	size_t max_bytes = orc_max_tile_stream(image_w, image_h, td) * channels;
	{
		const size_t mw = orcTileExtent(0, image_w, td), mh = orcTileExtent(0, image_h, td);
		if (max_bytes < mw * mh * channels * 2)
			max_bytes = mw * mh * channels * 2;
	}

	image = malloc(image_w * image_h * channels);
	stream = malloc(max_bytes + 4 * (image_w + image_h) * channels + 64);
	if (!image || !stream)
	{
		st = AKO_NO_ENOUGH_MEMORY;
		goto fail;
	}

	size_t tx = 0, ty = 0;
	for (size_t t = 0; t < tiles; t++)
	{
		const size_t tw = orcTileExtent(tx, image_w, td), th = orcTileExtent(ty, image_h, td);
		const size_t bytes = (s.wavelet != AKO_WAVELET_NONE) ? orcTileStreamBytes(tw, th) * channels
		                                                      : tw * th * channels * sizeof(i16);
		const i16* src;

		if (s.compression != AKO_COMPRESSION_NONE)
		{
			uint32_t c32;
			memcpy(&c32, cur, 4);
			/* decode.c:152: capacity = stream bytes + planes spacing (2w + 2h, wavelet streams only) */
			const size_t spacing = (s.wavelet != AKO_WAVELET_NONE) ? (2 * tw + 2 * th) : 0;
			const size_t used = orcKagariDecode(bytes / 2, c32, bytes + spacing, cur + 4, stream);
			if (used == 0 || used != c32)
			{
				st = AKO_BROKEN_INPUT;
				goto fail;
			}
			cur += used + 4;
			src = stream;
		}
		else
		{
			if (cur + bytes > end)
			{
				st = AKO_BROKEN_INPUT;
				goto fail;
			}
			memcpy(stream, cur, bytes); /* alignment */
			cur += bytes;
			src = stream;
		}

		const double t0 = orc_now();
		if (orcDecodeTile(&s, channels, tw, th, image_w, src, image + (image_w * ty + tx) * channels) != 0)
		{
			st = AKO_NO_ENOUGH_MEMORY;
			goto fail;
		}
		orc_transform_seconds += orc_now() - t0;

		tx += td;
		if (tx >= image_w)
		{
			tx = 0;
			ty += td;
		}
	}

	free(stream);
	if (out_s)
		*out_s = s;
	if (out_channels)
		*out_channels = channels;
	if (out_w)
		*out_w = image_w;
	if (out_h)
		*out_h = image_h;
	if (status)
		*status = AKO_OK;
	return image;

fail:
	free(stream);
	free(image);
	if (status)
		*status = st;
	return NULL;
}

/* ------------------------------------------------------------------------------------------
 * Checksums and synthetic inputs
 * ---------------------------------------------------------------------------------------- */

uint32_t orcAdler32(const uint8_t* data, size_t len) /* same definition as tools/misc.hpp:59-82 (zlib's) */
{
	uint32_t a = 1, b = 0;
	while (len != 0)
	{
		size_t n = len > 5552 ? 5552 : len;
		len -= n;
		while (n--)
		{
			a += *data++;
			b += a;
		}
		a %= 65521;
		b %= 65521;
	}
	return (b << 16) | a;
}

static inline uint32_t orc_xorshift(uint32_t* x)
{
	*x ^= *x << 13;
	*x ^= *x >> 17;
	*x ^= *x << 5;
	return *x;
}

void orcGenImage(int generator, uint32_t seed, size_t w, size_t h, uint8_t* rgba) /* SURVEY 8d, G0 / G1 */
{
	uint32_t st = seed;
	for (size_t y = 0; y < h; y++)
		for (size_t x = 0; x < w; x++)
		{
			const uint32_t n = orc_xorshift(&st);
			uint8_t* p = rgba + (y * w + x) * 4;
			if (generator == 0)
			{
				p[0] = (uint8_t)((x * 255) / w + (n & 3));
				p[1] = (uint8_t)((y * 255) / h + ((n >> 2) & 3));
				p[2] = (uint8_t)(((x + y) * 255) / (w + h) + ((n >> 4) & 3));
				p[3] = (((x / 64 + y / 64) & 1) != 0) ? 255 : 200;
			}
			else
			{
				p[0] = (uint8_t)(n & 255);
				p[1] = (uint8_t)((n >> 8) & 255);
				p[2] = (uint8_t)((n >> 16) & 255);
				p[3] = (uint8_t)(n >> 24);
			}
		}
}

void orcGenPlane(uint32_t seed, size_t n, i16* plane) /* SURVEY 8d, G2 */
{
	uint32_t st = seed;
	for (size_t i = 0; i < n; i++)
		plane[i] = (i16)((int)(orc_xorshift(&st) & 0x3FF) - 512);
}
