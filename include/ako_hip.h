/*
 * ako_hip.h -- C-ABI of the MI355X (gfx950) transform path of libako.
 *
 * This is the "internal seam" of the reference codec turned into a device contract.  Per tile the
 * reference runs (library/ako-private.h:47-51,81-84; call sites library/encode.c:132-148 and
 * library/decode.c:183-205):
 *
 *     akoFormatToPlanarI16Yuv(...)  +  akoLift(...)          u8 tile window  ->  coefficient stream
 *     akoUnlift(...)  +  akoFormatToInterleavedU8Rgb(...)    coefficient stream  ->  u8 tile window
 *
 * Here the same two mappings are offered for a whole BATCH of equally sized images at once, with
 * inputs and outputs resident in device memory (or in host memory through the *Host variants).
 * A "stream" is exactly the byte sequence the reference places in a blob when
 * compression == AKO_COMPRESSION_NONE (library/encode.c:151-152,177-182): for every tile in raster
 * order, akoTileDataSize(tile_w, tile_h) * channels bytes laid out as in library/lifting.c:171-292 /
 * library/misc.c:229-288.  The 16 byte file header and the entropy coder are NOT part of it.
 *
 * All functions are plain C: pointers, sizes, ints.  Functions returning int return 0 on success
 * and an enum akoStatus value otherwise; akoHipLastError() gives a human readable reason for the
 * calling thread.  Nothing here falls back to the CPU: without a HIP device every call fails.
 */
#ifndef AKO_HIP_H
#define AKO_HIP_H

#include "ako.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct akoHipPlan akoHipPlan;

/* Number of usable HIP devices (0 when there is none or the runtime cannot initialise). */
int akoHipDeviceCount(void);
/* 1 when built with AKO_BUILD_EXPERIMENTAL=1 (the routes AKO_HIP_FUSE2 / AKO_HIP_GROUP select exist), else 0 */
int akoHipHasExperimental(void);

/* Reason of the last failure on this thread ("" if none). */
const char* akoHipLastError(void);

/* The AKO_HIP_* tuning / test knobs of the environment (kernel path, tail engine, segment length ...) are read when a
 * plan is CREATED.  This folds their current values into one word, so that a cache of plans can tell when they moved. */
uint64_t akoHipTuningSignature(void);

/*
 * A plan fixes: device, settings, channel count, image size and batch size.  It owns the per level
 * geometry / quantizer tables (library/lifting.c:182-211 via library/quantization.c:67-98 -- computed
 * on the host), the tile table (library/misc.c:152-203) and the device scratch for the
 * intermediate low-pass planes.  settings->color is taken as given; use akoHipEffectiveColor() to
 * apply the encoder's YCOCG <-> YCOCG_Q rule (library/encode.c:59-64) first.
 *
 * hip_stream: a hipStream_t to launch on (e.g. the current PyTorch stream), or NULL for the
 * device's default stream.  flags: AKO_HIP_PLAN_* bits.
 */
#define AKO_HIP_PLAN_PLANES_I16 1u /* lifting-only mode: "images" are int16 planes (channels planes of w x h), no \
                                      colour transform, no saturation; decode writes int16 planes back */
#define AKO_HIP_PLAN_OWN_STREAM 2u /* with hip_stream == NULL: the plan creates (and owns) a non-blocking stream \
                                      instead of using the legacy default stream, so that plans of different host \
                                      threads run side by side; akoHipSynchronize() waits for it */

akoHipPlan* akoHipPlanCreate(int device, const struct akoSettings* settings, size_t channels, size_t image_w,
                             size_t image_h, size_t batch, void* hip_stream, unsigned flags,
                             enum akoStatus* out_status);
void akoHipPlanDestroy(akoHipPlan*);

enum akoColor akoHipEffectiveColor(const struct akoSettings* settings);

/* Sizes, per image of the batch. */
size_t akoHipPlanImageBytes(const akoHipPlan*);  /* w * h * channels (x2 in PLANES_I16 mode) */
size_t akoHipPlanStreamBytes(const akoHipPlan*); /* sum of the tile streams = blob size - 16 */
size_t akoHipPlanTiles(const akoHipPlan*);       /* library/misc.c:192 */
size_t akoHipPlanBatch(const akoHipPlan*);

/* Tile t (raster order): origin, extent and where its stream sits inside the image's stream. */
int akoHipPlanTileInfo(const akoHipPlan*, size_t tile, size_t* x, size_t* y, size_t* w, size_t* h,
                       size_t* stream_offset, size_t* stream_bytes);

/* Quantizer / gate actually used for (tile group of tile t, level, plane): what the reference's
 * akoQuantization / akoGate return (library/quantization.c:67,84).  level 0 = largest. */
int akoHipPlanLevels(const akoHipPlan*, size_t tile);
int akoHipPlanQuant(const akoHipPlan*, size_t tile, size_t level, size_t channel, int* q, int* g);

/*
 * Device resident transforms, asynchronous on the plan's stream.
 *   d_images : batch * akoHipPlanImageBytes   (interleaved u8, row pitch image_w * channels)
 *   d_streams: batch * akoHipPlanStreamBytes
 * Replaces akoFormatToPlanarI16Yuv + akoLift (encode) and akoUnlift + akoFormatToInterleavedU8Rgb
 * (decode) for every tile of every image of the batch.
 */
int akoHipEncode(akoHipPlan*, const void* d_images, void* d_streams);
int akoHipDecode(akoHipPlan*, const void* d_streams, void* d_images);
int akoHipSynchronize(akoHipPlan*);

/* The same with host buffers (pinned or pageable): H2D, kernels, D2H, synchronous. */
int akoHipEncodeHost(akoHipPlan*, const void* h_images, void* h_streams);
int akoHipDecodeHost(akoHipPlan*, const void* h_streams, void* h_images);

/*
 * Per-kernel timing for roofline reporting.  With profiling on, every kernel launch of
 * akoHipEncode / akoHipDecode is bracketed by hipEvents on the plan's stream; after
 * akoHipSynchronize the records of ALL encode (decode = 0) or decode (decode = 1) launches since
 * profiling was last switched on can be read back, in launch order.  akoHipPlanSetProfiling
 * (either value) forgets the records collected so far.
 */
struct akoHipKernelRecord
{
	char name[48];     /* e.g. "fwd_level_dd137_first" */
	float ms;          /* hipEventElapsedTime of this launch */
	uint32_t level;    /* 0 = largest */
	uint32_t group;    /* tile group: 0 interior, 1 right edge, 2 bottom edge, 3 corner */
	uint64_t units;    /* samples (w*h*planes*tiles*batch) the launch transformed */
	uint64_t bytes_rd; /* algorithmic bytes read by this launch */
	uint64_t bytes_wr; /* algorithmic bytes written by this launch */
};
int akoHipPlanSetProfiling(akoHipPlan*, int enabled);
size_t akoHipPlanKernelRecords(akoHipPlan*, int decode, struct akoHipKernelRecord* out, size_t capacity);

/* ---- device entropy stage: Kagari ENCODER on the GPU --------------------------------
 * Replaces the per-tile akoCompress() -> akoKagariEncode() calls of the encoder (library/encode.c:151-175,
 * library/compression.c:36-55, library/kagari.c:228-366) for all tiles of one image at once: the
 * coefficient streams never leave the GPU, only the compressed blob body does.
 *
 * akoHipEncodeUpload   copy a batch of host images to the device and transform them into the plan's own
 *                      stream buffer (akoHipEncodeHost without the copy back)
 * akoHipKagariEncode   entropy-code every tile stream of image `image` of d_streams (NULL = the plan's own
 *                      buffer).  On success *body_bytes is the size of the blob body: per tile, in tile
 *                      order, a little-endian uint32 payload size followed by the payload -- exactly what
 *                      follows the 16 byte head in a blob with compression KAGARI.  When a tile does not
 *                      shrink the reference encoder gives up (library/encode.c:159-164): returns AKO_ERROR
 *                      with *failed_tile set.  The body stays in plan-owned device memory
 * akoHipKagariFetch    copy that body to host memory (synchronous)
 * akoHipKagariBody     its device address, for device-resident callers */
int akoHipEncodeUpload(akoHipPlan*, const void* h_images);
int akoHipKagariEncode(akoHipPlan*, const void* d_streams, size_t image, size_t* body_bytes, size_t* failed_tile);
int akoHipKagariFetch(akoHipPlan*, void* h_body);
const void* akoHipKagariBody(const akoHipPlan*);

/* ---- device entropy stage, decoder side: run expansion on the GPU ---------------------------------
 * Kagari's bit-stream is sequential, so the host still parses it (library/kagari.c:296-366), but it no
 * longer writes the coefficient stream: it hands over the literal values it decoded plus one record per
 * run, and the GPU expands them into the stream (host work and the upload scale with the compressed size).
 *
 * akoHipKagariExpand   literals / runs of ALL tiles of one image (output positions are global over the
 *                      image's stream) -> d_streams (NULL = the plan's own buffer), image `image`
 * akoHipDecodeDownload inverse transform of the plan's own stream buffer + copy of the images to the host
 *                      (akoHipDecodeHost without the stream upload) */
struct akoHipKagariRun
{
	uint32_t out_start; /* stream index (int16 units) of the first repeated value */
	uint32_t count;     /* how many repeats */
	uint32_t after;     /* literals that precede the run; the repeated value is literals[after - 1] */
	uint32_t pad;
};
int akoHipKagariExpand(akoHipPlan*, const int16_t* h_literals, size_t n_literals, const struct akoHipKagariRun* h_runs,
                       size_t n_runs, void* d_streams, size_t image);
int akoHipDecodeDownload(akoHipPlan*, void* h_images);

/* ---- ratio search support: transform once, quantize per candidate ---------------------------
 * The quantization factor only enters the forward path where the C / B / D sub-bands are stored
 * (library/lifting.c:154-168) and in the lift heads (library/lifting.c:253-267).  A search over quantizations
 * (tools/akoenc.cpp:130-214) therefore transforms ONCE on a plan created with quantization 0 and gate 0 (and the colour
 * the candidates will have: see akoHipEffectiveColor), and per candidate calls
 *
 * akoHipRequantize      streams of that plan (d_unquantized, NULL = the plan's own buffer as left by akoHipEncodeUpload)
 *                       -> plan-owned streams (*d_out) that are bit-identical to encoding the pixels with
 *                       (quantization, gate); feed them to akoHipKagariEncode.  Synchronous.
 * akoHipPlanDeviceImages / akoHipPlanDeviceStreams   the plan's own staging buffers (device addresses)
 * akoEncodeRatioExt     the whole search of tools/akoenc.cpp:112-217 behind one call: same bracketing and bisection,
 *                       same chosen quantization and same blob as repeated akoEncodeExt calls, with one upload and
 *                       one transform per distinct colour transformation (at most two).  *out_quantization: the
 *                       factor it settled on; *out_encodes: how many candidate encodes it replaced.
 *                       Where it differs from calling akoEncodeExt in that loop: (1) the callbacks' EVENTS are not
 *                       fired during the search (the candidates are not per-tile encode passes; the reference tool
 *                       does not install its events callback for a ratio search either: tools/akoenc.cpp:271-277) --
 *                       allocation callbacks are honoured; (2) ratio <= 1 is a single lossless encode (the tool's
 *                       "ratio == 1" rule, tools/akoenc.cpp:118-126, extended to 0 and negatives instead of
 *                       dividing by them); (3) the bracket stops growing at quantization 2^28 (the tool's loop
 *                       would overflow its int there) */
int akoHipRequantize(akoHipPlan*, int quantization, int gate, const void* d_unquantized, void** d_out);
void* akoHipPlanDeviceImages(akoHipPlan*);
void* akoHipPlanDeviceStreams(akoHipPlan*);
size_t akoEncodeRatioExt(const struct akoCallbacks*, const struct akoSettings*, size_t channels, size_t image_w,
                         size_t image_h, const void* in, int ratio, void** out, int* out_quantization, int* out_encodes,
                         int* out_transforms, enum akoStatus* out_status);

/* akoEncodeExt / akoDecodeExt keep the device plan of a thread's previous call (same shape and settings: the next
 * call reuses it).  When a thread exits its plans are parked in a small process-wide pool, where the next thread
 * that needs the same shape finds them (what does not fit is destroyed by the next thread that enters the library).
 * akoHipThreadRelease() destroys the calling thread's plans and everything parked, at once. */
void akoHipThreadRelease(void);

/* ---- batched host API ---------------------------------------------------------------
 * Many equally shaped images from host memory to .ako blobs (and back) at link rate: the per-image loop of
 * tools/akoenc.cpp:112-217 / tools/akodec.cpp:100-154 over a whole array, BASELINE configs[3].  A batch object owns,
 * per device, a few LANES: a plan with its own HIP stream plus pinned staging buffers.  A call deals the images to
 * the lanes of all devices (worker threads of the library; the caller just waits), so that on every device the
 * upload of one image, the transform + entropy stage of another and the download of a third overlap.  Blobs are
 * byte-identical to akoEncodeExt's for the same image and settings.
 *
 * akoHipBatchCreate   devices: n_devices HIP device indices (NULL: device 0; a device may appear more than once);
 *                     lanes_per_device 0 = default (8)
 * akoHipEncodeBatch   images[i]: image_w * image_h * channels bytes each.  out_blobs[i] / out_sizes[i]: a malloc'ed
 *                     blob per image (release with akoDefaultFree), NULL / 0 where out_status[i] != AKO_OK
 *                     (out_status may be NULL).  Returns 0 when every image was encoded
 * akoHipDecodeBatch   blobs of images of the batch's shape -> images[i] (caller's buffers of image bytes each)
 * Images in pinned memory (akoHipHostAlloc) are copied to / from the device directly; pageable ones go through the
 * lane's pinned staging (one pass of a lane thread over the pixels).
 * ONE call at a time per batch object: the call's state, the lanes' plans and their staging belong to the call in
 * progress; a second thread that calls akoHipEncodeBatch / akoHipDecodeBatch on the same batch meanwhile gets AKO_ERROR
 * (several batch objects may of course be used side by side). */
typedef struct akoHipBatch akoHipBatch;
akoHipBatch* akoHipBatchCreate(const int* devices, size_t n_devices, size_t lanes_per_device,
                               const struct akoSettings* settings, size_t channels, size_t image_w, size_t image_h,
                               enum akoStatus* out_status);
void akoHipBatchDestroy(akoHipBatch*);
size_t akoHipBatchLanes(const akoHipBatch*);
/* what lane `lane` did in the last call: its device, the time it spent on its images, how many it took; 0: no such lane */
int akoHipBatchLaneStats(const akoHipBatch*, size_t lane, int* device, double* busy_seconds, size_t* images);
int akoHipEncodeBatch(akoHipBatch*, size_t n_images, const void* const* images, void** out_blobs, size_t* out_sizes,
                      enum akoStatus* out_status);
int akoHipDecodeBatch(akoHipBatch*, size_t n_blobs, const void* const* blobs, const size_t* blob_sizes, void** images,
                      enum akoStatus* out_status);

/* pinned host memory (hipHostMalloc): page-locked, so copies to / from the device run at link rate and asynchronously */
void* akoHipHostAlloc(size_t bytes);
void akoHipHostFree(void* p);
int akoHipHostIsPinned(const void* p); /* 1: page-locked memory the HIP runtime knows (akoHipHostAlloc, hipHostRegister) */

/* The bands of the calling thread's last akoEncodeExt / akoDecodeExt that was split over devices (AKO_HIP_DEVICES): device,
 * seconds and image rows of each, in band order (at most `cap` written); returns their number (0: the call was not split). */
size_t akoHipLastBands(int* devices, double* seconds, size_t* rows, size_t cap);
/* ... and how many band plans that call had to create: the band route keeps its plans (per device, band shape, settings and
 * direction) and its worker threads from call to call, so a repeated call reports 0 (AKO_HIP_PLAN_CACHE=0: one per band). */
size_t akoHipLastBandPlansCreated(void);

/* measurement aid: read + write GB/s of a tuned device-to-device copy of `bytes` on the current device (two temporary
 * buffers of that size); the practical memory rate bench.py reports beside the 8 TB/s spec peak.  0 on failure. */
double akoHipTunedCopyGBps(size_t bytes, int repeats);

#ifdef __cplusplus
}
#endif
#endif /* AKO_HIP_H */
