/*
 * ako_batch.c -- batched host API: many equally shaped images through the device path at link rate
 * (the caller it serves is the per-image loop of tools/akoenc.cpp:112-217 / tools/akodec.cpp:100-154,
 * BASELINE configs[3]: "batch of 64 x 4K images ... sharded across 8 MI355X via per-GPU streams").
 *
 * akoEncodeExt / akoDecodeExt work on one image from pageable memory and synchronise at every stage: the link, the
 * GPU and the host take turns.  A batch object owns LANES instead -- per device a few of them, each with its own
 * plan (own HIP stream), a pinned input staging buffer and a pinned output staging buffer.  One call hands a whole
 * array of images over; every lane runs on a worker thread of the library and pulls the next image as soon as it is
 * free, so that on each device the stages of consecutive images overlap across its lanes' streams: while one lane's
 * kernels and entropy stage run, another lane's pixels cross the link and a third copies its result out.  Images
 * are dealt to the lanes of ALL devices of the batch (image i of a batch is independent of every other: SURVEY 8e);
 * results land in the caller's arrays at the image's index, so the order of completion is invisible.
 *
 * The blobs are byte-identical to what akoEncodeExt writes for the same image and settings: same head
 * (ako_head.c), same device transform and device entropy stage (include/ako_hip.h), per image.
 */
#include "ako_host.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>

struct lane
{
	struct akoHipBatch* owner;
	int device;
	akoHipPlan* enc_plan;
	akoHipPlan* dec_plan;
	struct akoSettings dec_settings; /* what dec_plan was created for */
	uint8_t* pin_in;                 /* image_bytes: pixels on their way to the device */
	uint8_t* pin_out;                /* image_bytes: pixels on their way back (decode) */
	pthread_t thread;
	int started;
	double busy_s; /* the last call: time this lane spent on its images, and how many it took */
	size_t images;
};

struct akoHipBatch
{
	struct akoSettings s; /* colour already effective (encode.c:59-64) */
	size_t channels, w, h, image_bytes;
	size_t n_lanes, cap_lanes;
	struct lane* lanes;

	/* the call in progress */
	int decode;
	size_t n;
	const void* const* in;
	const size_t* in_sizes;
	void** out;
	size_t* out_sizes;
	enum akoStatus* status;
	size_t next;   /* next image index, atomically incremented */
	size_t failed; /* images that did not make it */
	int busy;      /* a call is in progress: the call state above, the lanes' plans and their staging belong to it */
};

static void set_status(struct akoHipBatch* b, size_t i, enum akoStatus st)
{
	if (b->status != NULL)
		b->status[i] = st;
	if (st != AKO_OK)
		__atomic_fetch_add(&b->failed, 1, __ATOMIC_RELAXED);
}

/* One image (or one band of tile rows of an image) through a plan: host pixels -> blob body.  `head_room` bytes are
 * left free in front of the body (the batch path writes the 16 byte head there).  *out is malloc'ed. */
enum akoStatus akoHostEncodeBody(akoHipPlan* plan, enum akoCompression compression, const void* pixels, size_t head_room,
                                 uint8_t** out, size_t* out_bytes)
{
	uint8_t* buf = NULL;
	size_t bytes = 0;
	if (compression == AKO_COMPRESSION_NONE)
	{
		bytes = akoHipPlanStreamBytes(plan);
		if ((buf = malloc(head_room + bytes)) == NULL)
			return AKO_NO_ENOUGH_MEMORY;
		const int rc = akoHipEncodeHost(plan, pixels, buf + head_room);
		if (rc != 0)
		{
			free(buf);
			return (enum akoStatus)rc;
		}
	}
	else
	{
		size_t bad = 0;
		int rc = akoHipEncodeUpload(plan, pixels);
		if (rc == 0)
			rc = akoHipKagariEncode(plan, NULL, 0, &bytes, &bad); /* AKO_ERROR: a tile did not shrink (encode.c:159-164) */
		if (rc != 0)
			return (enum akoStatus)rc;
		if ((buf = malloc(head_room + bytes)) == NULL)
			return AKO_NO_ENOUGH_MEMORY;
		if ((rc = akoHipKagariFetch(plan, buf + head_room)) != 0)
		{
			free(buf);
			return (enum akoStatus)rc;
		}
	}
	*out = buf, *out_bytes = bytes;
	return AKO_OK;
}

/* The inverse: the blob body of one image (or band) -> pixels.  *used = bytes of the body this plan's tiles took. */
enum akoStatus akoHostDecodeBody(akoHipPlan* plan, enum akoCompression compression, const uint8_t* body, size_t body_bytes,
                                 size_t* used, void* pixels)
{
	const uint8_t* cursor = body;
	const uint8_t* const end = body + body_bytes;
	const size_t stream_bytes = akoHipPlanStreamBytes(plan);
	int rc;
	if (compression == AKO_COMPRESSION_NONE)
	{
		if (body_bytes < stream_bytes)
			return AKO_BROKEN_INPUT;
		rc = akoHipDecodeHost(plan, body, pixels);
		cursor += stream_bytes;
	}
	else
	{
		/* parse every tile's bit-stream into ONE token list, expand on the GPU */
		struct akoKagariTokens tok;
		memset(&tok, 0, sizeof tok);
		const size_t tiles = akoHipPlanTiles(plan);
		enum akoStatus st = AKO_OK;
		for (size_t t = 0; t < tiles && st == AKO_OK; t++)
		{
			size_t off = 0, bytes = 0;
			akoHipPlanTileInfo(plan, t, NULL, NULL, NULL, NULL, &off, &bytes);
			uint32_t block = 0;
			if ((size_t)(end - cursor) < 4)
			{
				st = AKO_BROKEN_INPUT;
				break;
			}
			memcpy(&block, cursor, 4);
			if ((size_t)(end - cursor) - 4 < block)
			{
				st = AKO_BROKEN_INPUT;
				break;
			}
			const size_t took = akoHostKagariTokenize(bytes / 2, block, cursor + 4, off / 2, &tok);
			if (took == 0 || took != block) /* compression.c:69-70 */
				st = AKO_BROKEN_INPUT;
			cursor += (size_t)block + 4;
		}
		rc = (int)st;
		if (rc == 0)
			rc = akoHipKagariExpand(plan, tok.literals, tok.n_literals, (const struct akoHipKagariRun*)tok.runs, tok.n_runs,
			                        NULL, 0);
		akoHostKagariTokensFree(&tok);
		if (rc == 0)
			rc = akoHipDecodeDownload(plan, pixels);
	}
	if (rc != 0)
		return (enum akoStatus)rc;
	if (used != NULL)
		*used = (size_t)(cursor - body);
	return AKO_OK;
}

static enum akoStatus encode_one(struct lane* L, size_t i)
{
	struct akoHipBatch* b = L->owner;
	enum akoStatus st = AKO_OK;
	if (b->in[i] == NULL)
		return AKO_INVALID_INPUT;
	if (L->enc_plan == NULL &&
	    (L->enc_plan = akoHipPlanCreate(L->device, &b->s, b->channels, b->w, b->h, 1, NULL, AKO_HIP_PLAN_OWN_STREAM, &st)) == NULL)
		return st;

	const void* pixels = b->in[i];
	if (!akoHipHostIsPinned(pixels))
	{
		memcpy(L->pin_in, pixels, b->image_bytes); /* the only pass of the host over the pixels */
		pixels = L->pin_in;
	}
	uint8_t* blob = NULL;
	size_t body = 0;
	if ((st = akoHostEncodeBody(L->enc_plan, b->s.compression, pixels, sizeof(struct akoHead), &blob, &body)) != AKO_OK)
		return st;
	if ((st = akoHostHeadWrite(b->channels, b->w, b->h, &b->s, blob)) != AKO_OK)
	{
		free(blob);
		return st;
	}
	b->out[i] = blob;
	b->out_sizes[i] = sizeof(struct akoHead) + body;
	return AKO_OK;
}

static enum akoStatus decode_one(struct lane* L, size_t i)
{
	struct akoHipBatch* b = L->owner;
	const uint8_t* blob = b->in[i];
	const size_t size = b->in_sizes[i];
	if (blob == NULL || b->out[i] == NULL)
		return AKO_INVALID_INPUT;
	if (size < sizeof(struct akoHead))
		return AKO_BROKEN_INPUT;

	struct akoSettings hs;
	memset(&hs, 0, sizeof hs);
	size_t ch = 0, w = 0, h = 0;
	enum akoStatus st = akoHostHeadRead(blob, &ch, &w, &h, &hs);
	if (st != AKO_OK)
		return st;
	if (ch != b->channels || w != b->w || h != b->h)
		return AKO_INVALID_DIMENSIONS; /* not an image of this batch's shape */

	if (L->dec_plan != NULL && memcmp(&hs, &L->dec_settings, sizeof hs) != 0)
	{
		akoHipPlanDestroy(L->dec_plan);
		L->dec_plan = NULL;
	}
	if (L->dec_plan == NULL)
	{
		if ((L->dec_plan = akoHipPlanCreate(L->device, &hs, ch, w, h, 1, NULL, AKO_HIP_PLAN_OWN_STREAM, &st)) == NULL)
			return st;
		L->dec_settings = hs;
	}
	const int direct = akoHipHostIsPinned(b->out[i]);
	if ((st = akoHostDecodeBody(L->dec_plan, hs.compression, blob + sizeof(struct akoHead), size - sizeof(struct akoHead), NULL,
	                            direct ? b->out[i] : (void*)L->pin_out)) != AKO_OK)
		return st;
	if (!direct)
	{
		if (b->image_bytes >= ((size_t)4 << 20)) /* a fresh buffer: 2 MB faults instead of 4 KB ones where the system allows */
		{
			const uintptr_t lo = ((uintptr_t)b->out[i] + ((size_t)2 << 20) - 1) & ~(uintptr_t)(((size_t)2 << 20) - 1);
			const uintptr_t hi = ((uintptr_t)b->out[i] + b->image_bytes) & ~(uintptr_t)(((size_t)2 << 20) - 1);
			if (hi > lo)
				(void)madvise((void*)lo, (size_t)(hi - lo), MADV_HUGEPAGE);
		}
		memcpy(b->out[i], L->pin_out, b->image_bytes);
	}
	if (b->out_sizes != NULL)
		b->out_sizes[i] = b->image_bytes;
	return AKO_OK;
}

static void* lane_main(void* arg)
{
	struct lane* L = arg;
	struct akoHipBatch* b = L->owner;
	/* the lanes are the parallelism here: with more than two at work each parses its bit-streams alone */
	const size_t lanes = b->n_lanes < b->n ? b->n_lanes : b->n;
	akoHostKagariThreadLimit(lanes > 2 ? 1 : 0);
	for (;;)
	{
		const size_t i = __atomic_fetch_add(&b->next, 1, __ATOMIC_RELAXED);
		if (i >= b->n)
			return NULL;
		struct timespec t0, t1;
		clock_gettime(CLOCK_MONOTONIC, &t0);
		set_status(b, i, b->decode ? decode_one(L, i) : encode_one(L, i));
		clock_gettime(CLOCK_MONOTONIC, &t1);
		L->busy_s += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
		L->images++;
	}
}

static int run_call(struct akoHipBatch* b)
{
	b->next = 0, b->failed = 0;
	for (size_t k = 0; k < b->n_lanes; k++)
		b->lanes[k].busy_s = 0.0, b->lanes[k].images = 0;
	size_t lanes = b->n_lanes < b->n ? b->n_lanes : b->n;
	for (size_t k = 0; k < lanes; k++)
		b->lanes[k].started = (pthread_create(&b->lanes[k].thread, NULL, lane_main, &b->lanes[k]) == 0);
	int any = 0;
	for (size_t k = 0; k < lanes; k++)
		any |= b->lanes[k].started;
	if (!any && lanes != 0)
	{
		/* no thread could be started: the caller does the work (and gets its own parse-thread limit back afterwards) */
		const size_t callers_limit = akoHostKagariThreadLimitGet();
		lane_main(&b->lanes[0]);
		akoHostKagariThreadLimit(callers_limit);
	}
	for (size_t k = 0; k < lanes; k++)
		if (b->lanes[k].started)
			pthread_join(b->lanes[k].thread, NULL);
	return b->failed == 0 ? 0 : (int)AKO_ERROR;
}

AKO_API akoHipBatch* akoHipBatchCreate(const int* devices, size_t n_devices, size_t lanes_per_device,
                                       const struct akoSettings* settings, size_t channels, size_t image_w, size_t image_h,
                                       enum akoStatus* out_status)
{
	enum akoStatus st = AKO_OK;
	struct akoHipBatch* b = NULL;
	const int dev0 = 0;
	if (devices == NULL || n_devices == 0)
		devices = &dev0, n_devices = 1;
	if (lanes_per_device == 0)
		lanes_per_device = 8; /* 3840x2160 RGBA on one MI355X, encode / decode Gpx/s (link bound 13.8): 4 lanes 7.7 / 5.5,
		                       * 6 lanes 10.2-11.6 / 7.2-7.6, 8 lanes 10.3-11.0 / 10.2-10.5, 12 lanes 10.8-12.6 / 9.5-11.8 */
	if (settings == NULL || channels == 0 || image_w == 0 || image_h == 0 || n_devices > 64 || lanes_per_device > 16)
	{
		st = AKO_INVALID_INPUT;
		goto failed;
	}
	if ((b = calloc(1, sizeof *b)) == NULL || (b->lanes = calloc(n_devices * lanes_per_device, sizeof *b->lanes)) == NULL)
	{
		st = AKO_NO_ENOUGH_MEMORY;
		goto failed;
	}
	b->cap_lanes = n_devices * lanes_per_device;
	b->s = *settings;
	b->s.color = akoHipEffectiveColor(&b->s);
	b->channels = channels, b->w = image_w, b->h = image_h;
	if (__builtin_mul_overflow(image_w, image_h, &b->image_bytes) || __builtin_mul_overflow(b->image_bytes, channels, &b->image_bytes))
	{
		st = AKO_NO_ENOUGH_MEMORY;
		goto failed;
	}
	{
		uint8_t head[sizeof(struct akoHead)]; /* validates channels / dimensions / tiles / enums like the encoder does */
		if ((st = akoHostHeadWrite(channels, image_w, image_h, &b->s, head)) != AKO_OK)
			goto failed;
	}
	/* lanes interleave over the devices, so that a short batch still touches every device */
	for (size_t k = 0; k < n_devices * lanes_per_device; k++)
	{
		struct lane* L = &b->lanes[k];
		L->owner = b, L->device = devices[k % n_devices];
		/* the first plan of every lane is created here: a device that is not there fails the creation, loudly */
		if ((L->enc_plan = akoHipPlanCreate(L->device, &b->s, channels, image_w, image_h, 1, NULL, AKO_HIP_PLAN_OWN_STREAM, &st)) == NULL)
			goto failed;
		b->n_lanes = k + 1;
		if ((L->pin_in = akoHipHostAlloc(b->image_bytes)) == NULL || (L->pin_out = akoHipHostAlloc(b->image_bytes)) == NULL)
		{
			st = AKO_NO_ENOUGH_MEMORY;
			goto failed;
		}
	}
	if (out_status != NULL)
		*out_status = AKO_OK;
	return b;

failed:
	if (st != AKO_OK && getenv("AKO_HIP_QUIET") == NULL)
		fprintf(stderr, "libako (HIP): akoHipBatchCreate: %s\n", akoHipLastError());
	akoHipBatchDestroy(b);
	if (out_status != NULL)
		*out_status = st;
	return NULL;
}

AKO_API void akoHipBatchDestroy(akoHipBatch* b)
{
	if (b == NULL)
		return;
	for (size_t k = 0; b->lanes != NULL && k < b->cap_lanes; k++)
	{
		struct lane* L = &b->lanes[k]; /* a lane whose set-up failed half way still owns what it got */
		if (L->enc_plan != NULL)
			akoHipPlanDestroy(L->enc_plan);
		if (L->dec_plan != NULL)
			akoHipPlanDestroy(L->dec_plan);
		if (L->pin_in != NULL)
			akoHipHostFree(L->pin_in);
		if (L->pin_out != NULL)
			akoHipHostFree(L->pin_out);
	}
	free(b->lanes);
	free(b);
}

AKO_API size_t akoHipBatchLanes(const akoHipBatch* b)
{
	return b != NULL ? b->n_lanes : 0;
}

AKO_API int akoHipEncodeBatch(akoHipBatch* b, size_t n_images, const void* const* images, void** out_blobs,
                              size_t* out_sizes, enum akoStatus* out_status)
{
	if (b == NULL || (n_images != 0 && (images == NULL || out_blobs == NULL || out_sizes == NULL)))
		return (int)AKO_INVALID_INPUT;
	for (size_t i = 0; i < n_images; i++)
		out_blobs[i] = NULL, out_sizes[i] = 0;
	if (__atomic_exchange_n(&b->busy, 1, __ATOMIC_ACQUIRE))
		return (int)AKO_ERROR; /* one call at a time per batch (include/ako_hip.h) */
	b->decode = 0, b->n = n_images, b->in = images, b->in_sizes = NULL;
	b->out = out_blobs, b->out_sizes = out_sizes, b->status = out_status;
	const int rc = run_call(b);
	__atomic_store_n(&b->busy, 0, __ATOMIC_RELEASE);
	return rc;
}

AKO_API int akoHipDecodeBatch(akoHipBatch* b, size_t n_blobs, const void* const* blobs, const size_t* blob_sizes,
                              void** images, enum akoStatus* out_status)
{
	if (b == NULL || (n_blobs != 0 && (blobs == NULL || blob_sizes == NULL || images == NULL)))
		return (int)AKO_INVALID_INPUT;
	if (__atomic_exchange_n(&b->busy, 1, __ATOMIC_ACQUIRE))
		return (int)AKO_ERROR; /* one call at a time per batch (include/ako_hip.h) */
	b->decode = 1, b->n = n_blobs, b->in = blobs, b->in_sizes = blob_sizes;
	b->out = images, b->out_sizes = NULL, b->status = out_status;
	const int rc = run_call(b);
	__atomic_store_n(&b->busy, 0, __ATOMIC_RELEASE);
	return rc;
}

/* what lane `lane` did in the last call: its device, the time it spent on its images, how many it took (1: lane exists) */
AKO_API int akoHipBatchLaneStats(const akoHipBatch* b, size_t lane, int* device, double* busy_seconds, size_t* images)
{
	if (b == NULL || lane >= b->n_lanes)
		return 0;
	if (device)
		*device = b->lanes[lane].device;
	if (busy_seconds)
		*busy_seconds = b->lanes[lane].busy_s;
	if (images)
		*images = b->lanes[lane].images;
	return 1;
}
