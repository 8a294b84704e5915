/*
 * ako_host.h -- internal declarations of the host (C11) side of libako.
 *
 * The host keeps what the reference keeps on the CPU besides the transform: the 16 byte header
 * (reference: library/head.c), the Kagari entropy coder (library/kagari.c, library/compression.c),
 * the float quantizer curve (library/quantization.c) and the per image drivers
 * (library/encode.c, library/decode.c).  The transform itself is reached ONLY through the device
 * C-ABI of include/ako_hip.h -- there is no CPU implementation of it in this library.
 */
#ifndef AKO_HOST_H
#define AKO_HOST_H

#include "../../../include/ako.h"
#include "../../../include/ako_hip.h"

#define AKO_API __attribute__((visibility("default")))

/* ako_quant.c */
AKO_API int16_t akoHostQuantStep(int factor, int mul, size_t tile_w, size_t tile_h, size_t cur_w, size_t cur_h);
AKO_API int16_t akoHostGateStep(int factor, int mul, size_t tile_w, size_t tile_h, size_t cur_w, size_t cur_h);

/* ako_head.c */
AKO_API enum akoStatus akoHostHeadWrite(size_t channels, size_t w, size_t h, const struct akoSettings* s, void* out16);
AKO_API enum akoStatus akoHostHeadRead(const void* in16, size_t* channels, size_t* w, size_t* h,
                                       struct akoSettings* out_s);

/* ako_kagari.c: block = uint32 payload size + payload (reference: library/compression.c:30-73) */
AKO_API size_t akoHostKagariEncode(size_t input_bytes, size_t output_capacity, const void* input, void* output);
AKO_API size_t akoHostKagariDecode(size_t values_no, size_t input_bytes, size_t output_bytes, const void* input,
                                   void* output);

/* parse without expanding the runs (device route of the decoder); records are what
 * akoHipKagariExpand() takes (include/ako_hip.h: struct akoHipKagariRun has the same layout) */
struct akoKagariRun
{
	uint32_t out_start; /* output index (int16 units from the start of the image's stream) of the first repeat */
	uint32_t count;     /* repeats to write */
	uint32_t after;     /* literal values decoded before the run; the repeated value is literals[after - 1] */
	uint32_t pad;
};
struct akoKagariTokens
{
	int16_t* literals;
	size_t n_literals, cap_literals;
	struct akoKagariRun* runs;
	size_t n_runs, cap_runs;
};
AKO_API size_t akoHostKagariTokenize(size_t values_no, size_t input_bytes, const void* input, uint64_t out_base,
                                     struct akoKagariTokens* tok);
/* blocks of >= 128 KiB are parsed by several threads (speculative starts that re-synchronise, ako_kagari.c);
 * counts of blocks they finished / handed back to the sequential loop */
AKO_API void akoHostKagariParallelStats(size_t* accepted, size_t* handed_back);
AKO_API size_t akoHostKagariTokenizeWith(size_t max_threads, size_t values_no, size_t input_bytes, const void* input,
                                         uint64_t out_base, struct akoKagariTokens* tok);
AKO_API void akoHostKagariThreadLimit(size_t max_threads); /* for the calling thread; 0 = no limit */
AKO_API size_t akoHostKagariThreadLimitGet(void);            /* the calling thread's current limit */
AKO_API int akoHostKagariTokensReserve(struct akoKagariTokens* tok, size_t literals, size_t runs); /* room for that many MORE */
AKO_API int akoHostKagariTokensAppend(struct akoKagariTokens* dst, const struct akoKagariTokens* src,
                                      uint32_t literal_base);
AKO_API void akoHostKagariTokensFree(struct akoKagariTokens* tok);

/* ako_batch.c: one image (or band of tile rows) through a plan, host pixels <-> blob body (shared by the batch lanes
 * and the multi-device route of akoEncodeExt / akoDecodeExt) */
enum akoStatus akoHostEncodeBody(akoHipPlan* plan, enum akoCompression compression, const void* pixels, size_t head_room,
                                 uint8_t** out, size_t* out_bytes);
enum akoStatus akoHostDecodeBody(akoHipPlan* plan, enum akoCompression compression, const uint8_t* body, size_t body_bytes,
                                 size_t* used, void* pixels);

/* ako_synth.c: synthetic benchmark inputs */
AKO_API void akoHostSynthImage(int generator, uint32_t seed, size_t w, size_t h, uint8_t* rgba);
AKO_API void akoHostSynthPlane(uint32_t seed, size_t n, int16_t* plane);

#endif
