/*
 * ako.h -- public API of the MI355X-native Ako transform library.
 *
 * This header is the DROP-IN BOUNDARY: it declares the same ten C symbols, with
 * the same struct layouts and the same enumerator values, as the reference
 * codec's public header (reference: library/ako.h:21-145), so that a program
 * written against the reference (tools/akoenc.cpp:112-217, tools/akodec.cpp:100-154)
 * links against this library unchanged.  Everything behind akoEncodeExt /
 * akoDecodeExt -- colour transform, integer lifting DWT, quantization / noise
 * gate and coefficient-stream packing -- runs as hand-written HIP kernels on a
 * gfx950 device; only the entropy coder and the 16-byte file header stay on the
 * host.  There is no CPU fallback: without a usable HIP device the two entry
 * points fail with AKO_ERROR.
 *
 * The device-level C-ABI underneath (plans, device-resident encode / decode,
 * lifting-only entry points, the device entropy stage) is declared in ako_hip.h.
 *
 * Typical use (error handling elided):
 *
 *     struct akoSettings s = akoDefaultSettings();          // DD13/7, YCoCg, CLAMP, Kagari, q = 16
 *     s.quantization = 0;                                   // lossless
 *     void* blob = NULL;
 *     enum akoStatus st;
 *     size_t n = akoEncodeExt(NULL, &s, 4, w, h, rgba, &blob, &st);   // NULL callbacks = libc allocator
 *     ...
 *     size_t ch, dw, dh;
 *     uint8_t* pixels = akoDecodeExt(NULL, n, blob, &s, &ch, &dw, &dh, &st);
 *     akoDefaultFree(pixels);
 *     akoDefaultFree(blob);
 *
 * What happens where: the pixels go to the GPU once; colour transform, lifting, gate / quantization,
 * stream packing AND the Kagari entropy coder run there; only the compressed body comes back and is
 * placed behind the 16 byte head (decoder: the host parses the bit-streams -- tiles in parallel --
 * and the GPU expands the runs, inverts the transform and returns the pixels).  Process-wide knobs
 * are environment variables, listed in INTEGRATION.md: AKO_HIP_DEVICE, AKO_HIP_QUIET,
 * AKO_HIP_KAGARI, AKO_HIP_PLAN_CACHE, ...
 *
 * Thread safety: like the reference, the two entry points are re-entrant and keep no shared state
 * (each thread caches the device plan of its own previous call); the event callback runs on the
 * calling thread, per tile, in tile order.
 */
#ifndef AKO_H
#define AKO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Library and bit-stream versions (reference: library/ako.h:8-12). */
#define AKO_VERSION_MAJOR 0
#define AKO_VERSION_MINOR 2
#define AKO_VERSION_PATCH 0
#define AKO_FORMAT_VERSION 2

/* Limits enforced by the header validator (reference: library/ako.h:14-18, library/head.c:34-64). */
#define AKO_MAX_CHANNELS 16
#define AKO_MAX_WIDTH 4294967295
#define AKO_MAX_HEIGHT 4294967295
#define AKO_MIN_TILES_DIMENSION 8
#define AKO_MAX_TILES_DIMENSION 2147483648

/* Result codes.  Numeric values are part of the ABI (reference: library/ako.h:21-41). */
enum akoStatus
{
	AKO_OK = 0,
	AKO_ERROR = 1, /* generic failure; also: no HIP device, HIP runtime error, incompressible tile */

	AKO_INVALID_CHANNELS_NO = 2,
	AKO_INVALID_DIMENSIONS = 3,
	AKO_INVALID_TILES_DIMENSIONS = 4,
	AKO_INVALID_WRAP_MODE = 5,
	AKO_INVALID_WAVELET_TRANSFORMATION = 6,
	AKO_INVALID_COLOR_TRANSFORMATION = 7,
	AKO_INVALID_COMPRESSION_METHOD = 8,

	AKO_INVALID_INPUT = 9,
	AKO_INVALID_CALLBACKS = 10,
	AKO_INVALID_MAGIC = 11,
	AKO_UNSUPPORTED_VERSION = 12,
	AKO_NO_ENOUGH_MEMORY = 13,
	AKO_INVALID_FLAGS = 14,
	AKO_BROKEN_INPUT = 15
};

/* Wavelet family used by every lift of a tile (reference: library/ako.h:43-49).
 * DD137 silently drops to CDF53 on levels whose target extent is below 8
 * (reference: library/lifting.c:58,126). */
enum akoWavelet
{
	AKO_WAVELET_DD137 = 0, /* Deslauriers-Dubuc 13/7, integer lifting */
	AKO_WAVELET_CDF53 = 1, /* Cohen-Daubechies-Feauveau 5/3, integer lifting */
	AKO_WAVELET_HAAR = 2,  /* lazy split + difference */
	AKO_WAVELET_NONE = 3   /* no transform: the "stream" is the planar int16 image */
};

/* Reversible colour decorrelation applied when channels >= 3 (reference: library/ako.h:51-58). */
enum akoColor
{
	AKO_COLOR_YCOCG = 0,
	AKO_COLOR_SUBTRACT_G = 1,
	AKO_COLOR_NONE = 2,
	AKO_COLOR_YCOCG_Q = 3 /* chosen by the encoder itself whenever quantization or gate are on */
};

/* How lifting taps outside a tile are resolved (reference: library/ako.h:60-66). */
enum akoWrap
{
	AKO_WRAP_CLAMP = 0,
	AKO_WRAP_MIRROR = 1,
	AKO_WRAP_REPEAT = 2,
	AKO_WRAP_ZERO = 3
};

/* Host entropy stage (reference: library/ako.h:68-73). */
enum akoCompression
{
	AKO_COMPRESSION_KAGARI = 0,     /* Elias-gamma + run lengths, host side */
	AKO_COMPRESSION_MANBAVARAN = 1, /* placeholder in the reference; treated like KAGARI there */
	AKO_COMPRESSION_NONE = 2        /* blob = header + raw coefficient streams */
};

/* Progress / timing notifications (reference: library/ako.h:75-84). */
enum akoEvent
{
	AKO_EVENT_NONE = 0,
	AKO_EVENT_FORMAT_START = 1,
	AKO_EVENT_FORMAT_END = 2,
	AKO_EVENT_WAVELET_START = 3,
	AKO_EVENT_WAVELET_END = 4,
	AKO_EVENT_COMPRESSION_START = 5,
	AKO_EVENT_COMPRESSION_END = 6
};

/* Encoder knobs; field order and types are ABI (reference: library/ako.h:86-99). */
struct akoSettings
{
	enum akoWavelet wavelet;
	enum akoColor color;
	enum akoWrap wrap;
	enum akoCompression compression;
	size_t tiles_dimension; /* 0 = the whole image is one tile, else a power of two >= 8 */

	int quantization; /* 0 = lossless */
	int gate;         /* 0 = off */

	int chroma_loss; /* extra quantizer multiplier (value + 1) on every plane but the first */
	int discard_non_visible;
};

/* Caller supplied allocator and event sink (reference: library/ako.h:101-109). */
struct akoCallbacks
{
	void* (*malloc)(size_t);
	void* (*realloc)(void*, size_t);
	void (*free)(void*);

	void (*events)(size_t tile_no, size_t total_tiles, enum akoEvent, void* events_data);
	void* events_data;
};

/* The 16 byte file header, little endian (reference: library/ako.h:111-127).
 *   flags bits 0-3   channels - 1
 *         bits 4-5   wrap
 *         bits 6-7   wavelet
 *         bits 8-9   colour
 *         bits 10-11 compression
 *         bits 12-16 log2(tiles_dimension) - 2, or 0 when untiled
 *         bits 17-31 zero */
struct akoHead
{
	uint8_t magic[3]; /* 'A' 'k' 'o' */
	uint8_t version;  /* AKO_FORMAT_VERSION */
	uint32_t width;
	uint32_t height;
	uint32_t flags;
};

/*
 * Encode an interleaved 8 bit image (channels 1..16, row pitch image_w * channels).
 * Returns the blob size and stores the blob, allocated with callbacks.realloc, in *out
 * (or discards it when out == NULL); returns 0 and sets *out_status on failure.
 * NULL callbacks / settings select the defaults.  (reference: library/encode.c:38)
 */
size_t akoEncodeExt(const struct akoCallbacks*, const struct akoSettings*, size_t channels, size_t image_w,
                    size_t image_h, const void* in, void** out, enum akoStatus* out_status);

/*
 * Decode a blob made by akoEncodeExt.  Returns the interleaved 8 bit image allocated
 * with callbacks.malloc (to be released with callbacks.free), or NULL with *out_status set.
 * (reference: library/decode.c:38)
 */
uint8_t* akoDecodeExt(const struct akoCallbacks*, size_t input_size, const void* in, struct akoSettings* out_s,
                      size_t* out_channels, size_t* out_w, size_t* out_h, enum akoStatus* out_status);

/* Defaults and helpers (reference: library/misc.c:30-95, library/version.c). */
struct akoSettings akoDefaultSettings(void);
struct akoCallbacks akoDefaultCallbacks(void);
void akoDefaultFree(void*);

const char* akoStatusString(enum akoStatus);

int akoVersionMajor(void);
int akoVersionMinor(void);
int akoVersionPatch(void);
int akoFormatVersion(void);

#ifdef __cplusplus
}
#endif

#endif /* AKO_H */
