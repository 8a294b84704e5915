/*
 * ako_head.c -- the 16 byte little-endian file header (reference: library/head.c:34-169,
 * layout library/ako.h:111-127).  Byte-wise (de)serialisation, so no struct aliasing.
 */
#include "ako_host.h"

#include <string.h>

static enum akoStatus check_fields(size_t channels, size_t w, size_t h, size_t td, unsigned wrap, unsigned wavelet,
                                   unsigned color, unsigned compression)
{
	/* same order of checks as head.c:34-64 so the first failing field decides the status */
	if (channels > AKO_MAX_CHANNELS)
		return AKO_INVALID_CHANNELS_NO;
	if (w == 0 || h == 0 || w > AKO_MAX_WIDTH || h > AKO_MAX_HEIGHT)
		return AKO_INVALID_DIMENSIONS;
	if (td != 0 && (td < AKO_MIN_TILES_DIMENSION || td > AKO_MAX_TILES_DIMENSION))
		return AKO_INVALID_TILES_DIMENSIONS;
	if (wrap > AKO_WRAP_ZERO)
		return AKO_INVALID_WRAP_MODE;
	if (wavelet > AKO_WAVELET_NONE)
		return AKO_INVALID_WAVELET_TRANSFORMATION;
	if (color > AKO_COLOR_YCOCG_Q)
		return AKO_INVALID_COLOR_TRANSFORMATION;
	if (compression > AKO_COMPRESSION_NONE)
		return AKO_INVALID_COMPRESSION_METHOD;
	return AKO_OK;
}

static void put32(uint8_t* p, uint32_t v)
{
	p[0] = (uint8_t)v, p[1] = (uint8_t)(v >> 8), p[2] = (uint8_t)(v >> 16), p[3] = (uint8_t)(v >> 24);
}

static uint32_t get32(const uint8_t* p)
{
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

enum akoStatus akoHostHeadWrite(size_t channels, size_t w, size_t h, const struct akoSettings* s, void* out16)
{
	/* tiles field = log2(dimension) - 2; the dimension must be an exact power of two (head.c:71-82) */
	size_t field = 0;
	const size_t td = s->tiles_dimension;
	if (td != 0)
	{
		size_t log2td = 0;
		for (size_t v = td; v > 1; v >>= 1)
			log2td++;
		if (((size_t)1 << log2td) != td)
			return AKO_INVALID_TILES_DIMENSIONS;
		field = log2td - 2;
	}

	const enum akoStatus st = check_fields(channels, w, h, td, (unsigned)s->wrap, (unsigned)s->wavelet,
	                                       (unsigned)s->color, (unsigned)s->compression);
	if (st != AKO_OK)
		return st;

	uint8_t* o = out16;
	memcpy(o, "Ako", 3);
	o[3] = AKO_FORMAT_VERSION;
	put32(o + 4, (uint32_t)w);
	put32(o + 8, (uint32_t)h);
	put32(o + 12, (uint32_t)(channels - 1) | ((uint32_t)s->wrap << 4) | ((uint32_t)s->wavelet << 6) |
	                  ((uint32_t)s->color << 8) | ((uint32_t)s->compression << 10) | ((uint32_t)field << 12));
	return AKO_OK;
}

enum akoStatus akoHostHeadRead(const void* in16, size_t* channels, size_t* w, size_t* h, struct akoSettings* out_s)
{
	const uint8_t* i = in16;
	if (memcmp(i, "Ako", 3) != 0)
		return AKO_INVALID_MAGIC;
	if (i[3] != AKO_FORMAT_VERSION)
		return AKO_UNSUPPORTED_VERSION;

	const uint32_t flags = get32(i + 12);
	/* Parity note: the reference rejects any flags word with bit 15 or above set (head.c:124),
	 * which makes tiles >= 1024 undecodable.  Reproduced on purpose. */
	if ((flags >> 15) != 0)
		return AKO_INVALID_FLAGS;

	const size_t ch = (flags & 0xF) + 1;
	size_t td = (flags >> 12) & 0x1F;
	if (td != 0)
	{
		if (td >= 30)
			return AKO_INVALID_TILES_DIMENSIONS;
		td = (size_t)1 << (td + 2);
	}

	const size_t iw = get32(i + 4), ih = get32(i + 8);
	const enum akoStatus st =
	    check_fields(ch, iw, ih, td, (flags >> 4) & 3, (flags >> 6) & 3, (flags >> 8) & 3, (flags >> 10) & 3);
	if (st != AKO_OK)
		return st;

	if (channels)
		*channels = ch;
	if (w)
		*w = iw;
	if (h)
		*h = ih;
	if (out_s)
	{
		out_s->wrap = (enum akoWrap)((flags >> 4) & 3);
		out_s->wavelet = (enum akoWavelet)((flags >> 6) & 3);
		out_s->color = (enum akoColor)((flags >> 8) & 3);
		out_s->compression = (enum akoCompression)((flags >> 10) & 3);
		out_s->tiles_dimension = td;
	}
	return AKO_OK;
}
