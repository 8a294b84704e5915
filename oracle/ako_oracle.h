/*
 * ako_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference codec's tile-wise transform path, used
 * only as the parity checker by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  Nothing under ako_amd/ may include, link or
 * call this.  Parity status: PINNED -- checked byte-for-byte against the
 * compiled reference (oracle/_ref, built by oracle/Makefile from the sources
 * under /root/reference) by tests/test_oracle_vs_ref.py and against the committed
 * golden vectors in tests/golden/ (generated from oracle/_ref by
 * tests/golden/make_golden.py).
 */
#ifndef AKO_ORACLE_H
#define AKO_ORACLE_H

#include "../include/ako.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- geometry (reference: library/misc.c:98-203) ---- */
size_t orcHalfUp(size_t v);                                       /* misc.c:98  */
size_t orcTileStreamBytes(size_t tile_w, size_t tile_h);          /* misc.c:117 (one plane) */
size_t orcTileExtent(size_t pos, size_t image_d, size_t tiles_d); /* misc.c:152 */
size_t orcTilesNo(size_t image_w, size_t image_h, size_t tiles_d); /* misc.c:192 */
size_t orcLevels(size_t tile_w, size_t tile_h);                   /* lifting.c:182 loop count */

/* ---- quantizer / gate scalars (reference: library/quantization.c:43-98) ---- */
int16_t orcQuantStep(int factor, int mul, size_t tile_w, size_t tile_h, size_t cur_w, size_t cur_h);
int16_t orcGateStep(int factor, int mul, size_t tile_w, size_t tile_h, size_t cur_w, size_t cur_h);

/* ---- 1-D lifting on strided int16 sequences (reference: library/wavelet-*.c) ----
 * wavelet: AKO_WAVELET_DD137 / CDF53 / HAAR.  T = number of even (and odd) coefficients.
 * fake_last != 0 means the source has 2T-1 samples and the last odd equals the last even. */
void orcLift1d(int wavelet, int wrap, size_t T, int fake_last, const int16_t* src, ptrdiff_t src_stride, int16_t* lp,
               ptrdiff_t lp_stride, int16_t* hp, ptrdiff_t hp_stride);
void orcUnlift1d(int wavelet, int wrap, size_t T, const int16_t* lp, ptrdiff_t lp_stride, const int16_t* hp,
                 ptrdiff_t hp_stride, int16_t* even, ptrdiff_t even_stride, int16_t* odd, ptrdiff_t odd_stride);

/* ---- one 2-D level on dense planes (reference: library/lifting.c:43-76, 104-148) ----
 * Forward: src is cur_w x cur_h (pitch src_pitch); ll/b/c/d are dense tgt_w x tgt_h.  'kind' is the
 * wavelet actually used on this level (after the DD137 -> CDF53 fallback). */
void orcLevelForward(int kind, int wrap, size_t cur_w, size_t cur_h, const int16_t* src, size_t src_pitch, int16_t* ll,
                     int16_t* b, int16_t* c, int16_t* d);
/* Inverse: ll/c/b/d dense sub_w x sub_h (already de-quantized); out is dense tgt_w x tgt_h. */
void orcLevelInverse(int kind, int wrap, size_t sub_w, size_t sub_h, size_t tgt_w, size_t tgt_h, const int16_t* ll,
                     const int16_t* c, const int16_t* b, const int16_t* d, int16_t* out);

/* ---- whole-plane pyramid without colour / quantization: config "lifting only" ---- */
/* plane (w x h dense int16) -> stream (orcTileStreamBytes(w,h) bytes); returns 0 on success. */
int orcLiftPlane(int wavelet, int wrap, size_t w, size_t h, const int16_t* plane, int16_t* stream);
int orcUnliftPlane(int wavelet, int wrap, size_t w, size_t h, const int16_t* stream, int16_t* plane);

/* ---- tile level: u8 window <-> coefficient stream (reference: format.c:64, lifting.c:171, lifting.c:295,
 * format.c:244).  'in' / 'out' point at the tile origin inside an image whose row pitch is image_w pixels.
 * settings->color must already be the effective colour (YCOCG_Q fix-up applied).  Returns 0 on success. */
int orcEncodeTile(const struct akoSettings* s, size_t channels, size_t tile_w, size_t tile_h, size_t image_w,
                  const uint8_t* in, int16_t* stream);
int orcDecodeTile(const struct akoSettings* s, size_t channels, size_t tile_w, size_t tile_h, size_t image_w,
                  const int16_t* stream, uint8_t* out);

/* ---- header (reference: library/head.c:67-169) ---- */
enum akoStatus orcHeadWrite(size_t channels, size_t w, size_t h, const struct akoSettings* s, void* out16);
enum akoStatus orcHeadRead(const void* in16, size_t* channels, size_t* w, size_t* h, struct akoSettings* s);

/* ---- Kagari entropy coder (reference: library/kagari.c:228-366, compression.c:36-73) ---- */
size_t orcKagariEncode(size_t input_bytes, size_t output_capacity, const void* input, void* output);
size_t orcKagariDecode(size_t values_no, size_t input_bytes, size_t output_bytes, const void* input, void* output);

/* ---- image level, same contract as akoEncodeExt / akoDecodeExt with default callbacks
 * (reference: library/encode.c:38, library/decode.c:38).  Blob is malloc'ed; free with free(). ---- */
size_t orcEncodeImage(const struct akoSettings* s, size_t channels, size_t image_w, size_t image_h, const void* in,
                      void** out, enum akoStatus* status);
uint8_t* orcDecodeImage(size_t input_size, const void* in, struct akoSettings* out_s, size_t* out_channels,
                        size_t* out_w, size_t* out_h, enum akoStatus* status);

/* ---- stage timing for the cpu_baseline leg: transform-only seconds of the last
 * orcEncodeImage / orcDecodeImage call on this thread (format + wavelet stages, no entropy coding). */
double orcLastTransformSeconds(void);

/* ---- checksums and synthetic inputs shared by tests and bench (tools/misc.hpp:59) ---- */
uint32_t orcAdler32(const uint8_t* data, size_t len);
void orcGenImage(int generator, uint32_t seed, size_t w, size_t h, uint8_t* rgba); /* 0 = G0 smooth, 1 = G1 noise */
void orcGenPlane(uint32_t seed, size_t n, int16_t* plane);                         /* G2 */

#ifdef __cplusplus
}
#endif
#endif
