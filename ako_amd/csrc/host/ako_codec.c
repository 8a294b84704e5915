/*
 * ako_codec.c -- akoEncodeExt / akoDecodeExt: the per image drivers behind the public API
 * (reference: library/encode.c:38-232, library/decode.c:38-250).
 *
 * Same validation order, status codes, ownership rules and event sequence as the reference, but
 * the tile loop's two transform calls (akoFormatToPlanarI16Yuv + akoLift, akoUnlift +
 * akoFormatToInterleavedU8Rgb) are replaced by ONE batched device call for all tiles of the image
 * through the C-ABI of include/ako_hip.h.  The host then walks the tiles in raster order for the
 * entropy stage and the blob assembly.
 *
 * Events: the reference fires FORMAT, WAVELET and COMPRESSION start/end pairs per tile, in tile
 * order, on the caller's thread (encode.c:132-184, decode.c:145-207; consumers such as
 * tools/benchmark.hpp:73-89 reset on tile 0 and print on the last tile).  That sequence is kept;
 * the whole device transform runs inside tile 0's WAVELET bracket (encode) or the last
 * tile's WAVELET bracket (decode), because it cannot start before every tile is decompressed.
 *
 * Device selection: environment variable AKO_HIP_DEVICE (default 0).  No CPU fallback.
 */
#include "ako_host.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static void fire(const struct akoCallbacks* c, size_t tile, size_t total, enum akoEvent e)
{
	if (c->events != NULL)
		c->events(tile, total, e, c->events_data);
}

static int chosen_device(void)
{
	const char* e = getenv("AKO_HIP_DEVICE");
	return (e != NULL && *e != '\0') ? atoi(e) : 0;
}

/* ---- plan cache ----------------------------------------------------------------------------
 * Creating a plan allocates the device buffers of the transform and of the entropy stage (a few dozen
 * hipMalloc / hipFree per call, about half a millisecond -- more than the work on a small image).
 * Each thread keeps the plan of its last call and reuses it when the next call has the same shape and
 * settings (the usual case: a stream of equally sized images, akoenc's ratio search).  The plan of a
 * thread is replaced on a miss and lives until then: AKO_HIP_PLAN_CACHE=0 switches the cache off and
 * restores create / destroy per call.  Re-entrancy is unaffected (nothing is shared between threads). */
struct plan_key
{
	struct akoSettings s;
	size_t channels, w, h;
	int device;
};

/* one slot per direction: an encoder's settings never equal a decoder's (quantization is not in the head) */
static __thread akoHipPlan* cached_plans[2] = {NULL, NULL};
static __thread struct plan_key cached_keys[2];

static int plan_cache_on(void)
{
	const char* e = getenv("AKO_HIP_PLAN_CACHE");
	return !(e != NULL && atoi(e) == 0);
}

static akoHipPlan* plan_acquire(int slot, const struct akoSettings* st, size_t channels, size_t w, size_t h,
                                enum akoStatus* status)
{
	struct plan_key key;
	memset(&key, 0, sizeof key); /* padding bytes too: the keys are compared with memcmp */
	key.s.wavelet = st->wavelet, key.s.color = st->color, key.s.wrap = st->wrap, key.s.compression = st->compression;
	key.s.tiles_dimension = st->tiles_dimension, key.s.quantization = st->quantization, key.s.gate = st->gate;
	key.s.chroma_loss = st->chroma_loss, key.s.discard_non_visible = st->discard_non_visible;
	key.channels = channels, key.w = w, key.h = h, key.device = chosen_device();

	if (cached_plans[slot] != NULL)
	{
		akoHipPlan* p = cached_plans[slot];
		cached_plans[slot] = NULL; /* taken out while in use: an event callback may call back into the library */
		if (plan_cache_on() && memcmp(&key, &cached_keys[slot], sizeof key) == 0)
			return p;
		akoHipPlanDestroy(p);
	}
	cached_keys[slot] = key;
	/* an own stream per plan: calls from different host threads overlap on the GPU instead of queueing up on
	 * the legacy default stream (every driver call below ends with a synchronisation of that stream) */
	return akoHipPlanCreate(key.device, st, channels, w, h, 1, NULL, AKO_HIP_PLAN_OWN_STREAM, status);
}

static void plan_release(int slot, akoHipPlan* plan, int healthy)
{
	if (plan == NULL)
		return;
	if (healthy && plan_cache_on() && cached_plans[slot] == NULL)
		cached_plans[slot] = plan; /* its key was stored by plan_acquire */
	else
		akoHipPlanDestroy(plan);
}

/* ---- tiles of a Kagari blob parsed in parallel -----------------------------------------------
 * A tile's bit-stream can only be walked sequentially, but tiles are independent (library/encode.c:115-205):
 * worker threads pull tile numbers and tokenize them into per-tile lists; the caller then merges the lists
 * in tile order (and reports the first broken tile exactly where the sequential loop would have). */
struct tile_job
{
	const uint8_t* payload; /* NULL: the block chain ran off the input before this tile */
	uint32_t block;
	size_t values, out_base;
	struct akoKagariTokens tok;
	size_t used;
};
struct tile_pool
{
	struct tile_job* jobs;
	size_t count;
	size_t next; /* atomically incremented */
};

static void* tile_worker(void* arg)
{
	struct tile_pool* pool = arg;
	for (;;)
	{
		const size_t t = __atomic_fetch_add(&pool->next, 1, __ATOMIC_RELAXED);
		if (t >= pool->count)
			return NULL;
		struct tile_job* j = &pool->jobs[t];
		j->used = (j->payload != NULL) ? akoHostKagariTokenize(j->values, j->block, j->payload, j->out_base, &j->tok) : 0;
	}
}

static void tokenize_tiles(struct tile_job* jobs, size_t count)
{
	struct tile_pool pool = {jobs, count, 0};
	long cores = sysconf(_SC_NPROCESSORS_ONLN);
	size_t workers = (cores > 1) ? (size_t)cores : 1;
	if (workers > 16)
		workers = 16;
	if (workers > count)
		workers = count;
	pthread_t th[16];
	size_t started = 0;
	for (size_t k = 1; k < workers; k++) /* the calling thread is worker 0 */
		if (pthread_create(&th[started], NULL, tile_worker, &pool) == 0)
			started++;
	tile_worker(&pool);
	for (size_t k = 0; k < started; k++)
		pthread_join(th[k], NULL);
}

static void complain(const char* where)
{
	/* failing loudly: a missing / broken HIP path must never pass for a working codec */
	if (getenv("AKO_HIP_QUIET") == NULL)
		fprintf(stderr, "libako (HIP): %s: %s\n", where, akoHipLastError());
}

AKO_API size_t akoEncodeExt(const struct akoCallbacks* c, const struct akoSettings* s, size_t channels,
                            size_t image_w, size_t image_h, const void* in, void** out, enum akoStatus* out_status)
{
	enum akoStatus status = AKO_OK;
	uint8_t* blob = NULL;
	size_t blob_size = 0;
	uint8_t* streams = NULL;
	uint8_t* packed = NULL;
	akoHipPlan* plan = NULL;

	const struct akoCallbacks cb = (c != NULL) ? *c : akoDefaultCallbacks();
	struct akoSettings st = (s != NULL) ? *s : akoDefaultSettings();

	if (cb.malloc == NULL || cb.realloc == NULL || cb.free == NULL)
	{
		status = AKO_INVALID_CALLBACKS;
		goto failure;
	}

	st.color = akoHipEffectiveColor(&st); /* encode.c:59-64 */

	if (in == NULL)
	{
		status = AKO_INVALID_INPUT;
		goto failure;
	}

	blob_size = sizeof(struct akoHead);
	if ((blob = cb.malloc(blob_size)) == NULL)
	{
		status = AKO_NO_ENOUGH_MEMORY;
		goto failure;
	}
	if ((status = akoHostHeadWrite(channels, image_w, image_h, &st, blob)) != AKO_OK)
		goto failure;

	if ((plan = plan_acquire(0, &st, channels, image_w, image_h, &status)) == NULL)
	{
		complain("akoEncodeExt");
		goto failure;
	}

	const size_t tiles = akoHipPlanTiles(plan);
	const size_t stream_bytes = akoHipPlanStreamBytes(plan);

	/* Entropy stage on the GPU (default): the coefficient streams stay on the device and only the
	 * compressed body comes back.  AKO_HIP_KAGARI=host keeps the streams-to-host + host Kagari route. */
	const char* kg_env = getenv("AKO_HIP_KAGARI");
	const int device_kagari = (st.compression != AKO_COMPRESSION_NONE) && !(kg_env != NULL && strcmp(kg_env, "host") == 0);

	if (device_kagari)
	{
		for (size_t t = 0; t < tiles; t++)
		{
			fire(&cb, t, tiles, AKO_EVENT_FORMAT_START);
			fire(&cb, t, tiles, AKO_EVENT_FORMAT_END);
			if (st.wavelet != AKO_WAVELET_NONE)
				fire(&cb, t, tiles, AKO_EVENT_WAVELET_START);
			if (t == 0)
			{
				int rc = akoHipEncodeUpload(plan, in);
				if (rc == 0)
					rc = akoHipSynchronize(plan);
				if (rc != 0)
				{
					status = (enum akoStatus)rc;
					complain("akoEncodeExt");
					goto failure;
				}
			}
			if (st.wavelet != AKO_WAVELET_NONE)
				fire(&cb, t, tiles, AKO_EVENT_WAVELET_END);

			fire(&cb, t, tiles, AKO_EVENT_COMPRESSION_START);
			if (t == 0)
			{
				size_t body = 0, bad_tile = 0;
				const int rc = akoHipKagariEncode(plan, NULL, 0, &body, &bad_tile);
				if (rc != 0)
				{
					status = (enum akoStatus)rc; /* AKO_ERROR when a tile did not shrink: encode.c:159-164 */
					if (rc != AKO_ERROR)
						complain("akoEncodeExt");
					goto failure;
				}
				uint8_t* grown = cb.realloc(blob, blob_size + body);
				if (grown == NULL)
				{
					status = AKO_NO_ENOUGH_MEMORY;
					goto failure;
				}
				blob = grown;
				if ((status = (enum akoStatus)akoHipKagariFetch(plan, blob + blob_size)) != AKO_OK)
				{
					complain("akoEncodeExt");
					goto failure;
				}
				blob_size += body;
			}
			fire(&cb, t, tiles, AKO_EVENT_COMPRESSION_END);
		}
		goto done;
	}

	if ((streams = cb.malloc(stream_bytes)) == NULL)
	{
		status = AKO_NO_ENOUGH_MEMORY;
		goto failure;
	}

	size_t max_tile_bytes = 0;
	for (size_t t = 0; t < tiles; t++)
	{
		size_t bytes = 0;
		akoHipPlanTileInfo(plan, t, NULL, NULL, NULL, NULL, NULL, &bytes);
		if (bytes > max_tile_bytes)
			max_tile_bytes = bytes;
	}
	if (st.compression != AKO_COMPRESSION_NONE && (packed = cb.malloc(max_tile_bytes + 8)) == NULL)
	{
		status = AKO_NO_ENOUGH_MEMORY;
		goto failure;
	}
	if (st.compression == AKO_COMPRESSION_NONE)
	{
		/* final size is known up front: one allocation instead of one realloc per tile */
		uint8_t* grown = cb.realloc(blob, blob_size + stream_bytes);
		if (grown == NULL)
		{
			status = AKO_NO_ENOUGH_MEMORY;
			goto failure;
		}
		blob = grown;
	}

	for (size_t t = 0; t < tiles; t++)
	{
		size_t off = 0, bytes = 0;
		akoHipPlanTileInfo(plan, t, NULL, NULL, NULL, NULL, &off, &bytes);

		fire(&cb, t, tiles, AKO_EVENT_FORMAT_START);
		fire(&cb, t, tiles, AKO_EVENT_FORMAT_END);

		if (st.wavelet != AKO_WAVELET_NONE || t == 0)
		{
			if (st.wavelet != AKO_WAVELET_NONE)
				fire(&cb, t, tiles, AKO_EVENT_WAVELET_START);
			if (t == 0)
			{
				const int rc = akoHipEncodeHost(plan, in, streams);
				if (rc != 0)
				{
					status = (enum akoStatus)rc;
					complain("akoEncodeExt");
					goto failure;
				}
			}
			if (st.wavelet != AKO_WAVELET_NONE)
				fire(&cb, t, tiles, AKO_EVENT_WAVELET_END);
		}

		fire(&cb, t, tiles, AKO_EVENT_COMPRESSION_START);
		if (st.compression != AKO_COMPRESSION_NONE)
		{
			/* compression.c:36-55: output capacity == input size, uint32 block size in front */
			const size_t payload = akoHostKagariEncode(bytes, bytes - 4, streams + off, packed + 4);
			if (payload == 0)
			{
				status = AKO_ERROR; /* tile did not shrink: encode.c:159-164 */
				goto failure;
			}
			const uint32_t p32 = (uint32_t)payload;
			memcpy(packed, &p32, 4);

			uint8_t* grown = cb.realloc(blob, blob_size + payload + 4);
			if (grown == NULL)
			{
				status = AKO_NO_ENOUGH_MEMORY;
				goto failure;
			}
			blob = grown;
			memcpy(blob + blob_size, packed, payload + 4);
			blob_size += payload + 4;
		}
		else
		{
			memcpy(blob + blob_size, streams + off, bytes);
			blob_size += bytes;
		}
		fire(&cb, t, tiles, AKO_EVENT_COMPRESSION_END);
	}

done:
	plan_release(0, plan, 1);
	if (streams != NULL)
		cb.free(streams);
	if (packed != NULL)
		cb.free(packed);

	if (out_status != NULL)
		*out_status = AKO_OK;
	if (out != NULL)
		*out = blob;
	else
		cb.free(blob); /* caller only wanted the size: encode.c:214-217 */
	return blob_size;

failure:
	plan_release(0, plan, 0); /* after an error the plan is not kept */
	if (cb.free != NULL)
	{
		if (streams != NULL)
			cb.free(streams);
		if (packed != NULL)
			cb.free(packed);
		if (blob != NULL)
			cb.free(blob);
	}
	if (out_status != NULL)
		*out_status = status;
	return 0;
}

AKO_API uint8_t* akoDecodeExt(const struct akoCallbacks* c, size_t input_size, const void* input,
                              struct akoSettings* out_s, size_t* out_channels, size_t* out_w, size_t* out_h,
                              enum akoStatus* out_status)
{
	enum akoStatus status = AKO_OK;
	struct akoSettings st;
	memset(&st, 0, sizeof st);
	size_t channels = 0, image_w = 0, image_h = 0;
	uint8_t* image = NULL;
	uint8_t* streams = NULL;
	akoHipPlan* plan = NULL;
	struct akoKagariTokens tokens;
	memset(&tokens, 0, sizeof tokens);
	struct tile_job* jobs = NULL;
	size_t n_jobs = 0;

	const struct akoCallbacks cb = (c != NULL) ? *c : akoDefaultCallbacks();
	if (cb.malloc == NULL || cb.realloc == NULL || cb.free == NULL)
	{
		status = AKO_INVALID_CALLBACKS;
		goto failure;
	}
	if (input == NULL)
	{
		status = AKO_INVALID_INPUT;
		goto failure;
	}
	if (input_size < sizeof(struct akoHead))
	{
		/* the reference's own bound check here is vacuous (decode.c:71); a short blob must not be read */
		status = AKO_BROKEN_INPUT;
		goto failure;
	}
	if ((status = akoHostHeadRead(input, &channels, &image_w, &image_h, &st)) != AKO_OK)
		goto failure;

	if ((plan = plan_acquire(1, &st, channels, image_w, image_h, &status)) == NULL)
	{
		complain("akoDecodeExt");
		goto failure;
	}

	const size_t tiles = akoHipPlanTiles(plan);
	const size_t stream_bytes = akoHipPlanStreamBytes(plan);
	const uint8_t* cursor = (const uint8_t*)input + sizeof(struct akoHead);
	const uint8_t* const end = (const uint8_t*)input + input_size;

	/* Entropy stage, device route (default): parse the Kagari bit-streams on the host, expand the runs on the
	 * GPU -- the coefficient streams never exist in host memory.  AKO_HIP_KAGARI=host keeps the old route. */
	const char* kg_env = getenv("AKO_HIP_KAGARI");
	if (st.compression != AKO_COMPRESSION_NONE && !(kg_env != NULL && strcmp(kg_env, "host") == 0) &&
	    stream_bytes / 2 <= 0xFFFFFFF0ull)
	{
		if ((image = cb.malloc(image_w * image_h * channels)) == NULL)
		{
			status = AKO_NO_ENOUGH_MEMORY;
			goto failure;
		}
		memset(&tokens, 0, sizeof tokens);

		/* several tiles: walk the chain of block sizes, then parse all tiles at once on worker threads */
		if (tiles > 1)
		{
			if ((jobs = calloc(tiles, sizeof *jobs)) == NULL)
			{
				status = AKO_NO_ENOUGH_MEMORY;
				goto failure;
			}
			n_jobs = tiles;
			const uint8_t* walk = cursor;
			for (size_t t = 0; t < tiles; t++)
			{
				size_t off = 0, bytes = 0;
				akoHipPlanTileInfo(plan, t, NULL, NULL, NULL, NULL, &off, &bytes);
				uint32_t block = 0;
				if ((size_t)(end - walk) < 4)
					break; /* this tile and every later one keep payload == NULL */
				memcpy(&block, walk, 4);
				if ((size_t)(end - walk) - 4 < block)
					break;
				jobs[t].payload = walk + 4, jobs[t].block = block;
				jobs[t].values = bytes / 2, jobs[t].out_base = off / 2;
				walk += (size_t)block + 4;
			}
			tokenize_tiles(jobs, tiles);
		}

		for (size_t t = 0; t < tiles; t++)
		{
			size_t off = 0, bytes = 0;
			akoHipPlanTileInfo(plan, t, NULL, NULL, NULL, NULL, &off, &bytes);

			fire(&cb, t, tiles, AKO_EVENT_COMPRESSION_START);
			uint32_t block = 0;
			if ((size_t)(end - cursor) < 4)
			{
				status = AKO_BROKEN_INPUT;
				goto failure;
			}
			memcpy(&block, cursor, 4);
			if ((size_t)(end - cursor) - 4 < block)
			{
				status = AKO_BROKEN_INPUT;
				goto failure;
			}
			if (jobs != NULL)
			{
				/* merge this tile's list: literals are appended, 'after' counts become global */
				struct tile_job* j = &jobs[t];
				if (j->used == 0 || j->used != block) /* compression.c:69-70 */
				{
					status = AKO_BROKEN_INPUT;
					goto failure;
				}
				const size_t base = tokens.n_literals;
				if (base + j->tok.n_literals > 0xFFFFFFF0ull ||
				    !akoHostKagariTokensAppend(&tokens, &j->tok, (uint32_t)base))
				{
					status = AKO_NO_ENOUGH_MEMORY;
					goto failure;
				}
				akoHostKagariTokensFree(&j->tok);
			}
			else
			{
				const size_t used = akoHostKagariTokenize(bytes / 2, block, cursor + 4, off / 2, &tokens);
				if (used == 0 || used != block) /* compression.c:69-70 */
				{
					status = AKO_BROKEN_INPUT;
					goto failure;
				}
			}
			cursor += (size_t)block + 4;
			if (t + 1 == tiles)
			{
				const int rc = akoHipKagariExpand(plan, tokens.literals, tokens.n_literals,
				                                  (const struct akoHipKagariRun*)tokens.runs, tokens.n_runs, NULL, 0);
				if (rc != 0)
				{
					status = (enum akoStatus)rc;
					complain("akoDecodeExt");
					goto failure;
				}
			}
			fire(&cb, t, tiles, AKO_EVENT_COMPRESSION_END);

			if (st.wavelet != AKO_WAVELET_NONE)
				fire(&cb, t, tiles, AKO_EVENT_WAVELET_START);
			if (t + 1 == tiles)
			{
				const int rc = akoHipDecodeDownload(plan, image);
				if (rc != 0)
				{
					status = (enum akoStatus)rc;
					complain("akoDecodeExt");
					goto failure;
				}
			}
			if (st.wavelet != AKO_WAVELET_NONE)
				fire(&cb, t, tiles, AKO_EVENT_WAVELET_END);
			fire(&cb, t, tiles, AKO_EVENT_FORMAT_START);
			fire(&cb, t, tiles, AKO_EVENT_FORMAT_END);
		}
		akoHostKagariTokensFree(&tokens);
		free(jobs); /* every per-tile list was released when it was merged */
		jobs = NULL;
		goto decoded;
	}

	image = cb.malloc(image_w * image_h * channels);
	streams = cb.malloc(stream_bytes);
	if (image == NULL || streams == NULL)
	{
		status = AKO_NO_ENOUGH_MEMORY;
		goto failure;
	}

	for (size_t t = 0; t < tiles; t++)
	{
		size_t off = 0, bytes = 0;
		akoHipPlanTileInfo(plan, t, NULL, NULL, NULL, NULL, &off, &bytes);

		fire(&cb, t, tiles, AKO_EVENT_COMPRESSION_START);
		if (st.compression != AKO_COMPRESSION_NONE)
		{
			uint32_t block = 0;
			if ((size_t)(end - cursor) < 4)
			{
				status = AKO_BROKEN_INPUT;
				goto failure;
			}
			memcpy(&block, cursor, 4);
			if ((size_t)(end - cursor) - 4 < block)
			{
				status = AKO_BROKEN_INPUT;
				goto failure;
			}
			const size_t used = akoHostKagariDecode(bytes / 2, block, bytes, cursor + 4, streams + off);
			if (used == 0 || used != block) /* compression.c:69-70 */
			{
				status = AKO_BROKEN_INPUT;
				goto failure;
			}
			cursor += (size_t)block + 4;
		}
		else
		{
			if ((size_t)(end - cursor) < bytes) /* decode.c:163-167 */
			{
				status = AKO_BROKEN_INPUT;
				goto failure;
			}
			memcpy(streams + off, cursor, bytes);
			cursor += bytes;
		}
		fire(&cb, t, tiles, AKO_EVENT_COMPRESSION_END);

		if (st.wavelet != AKO_WAVELET_NONE)
			fire(&cb, t, tiles, AKO_EVENT_WAVELET_START);
		if (t + 1 == tiles)
		{
			const int rc = akoHipDecodeHost(plan, streams, image);
			if (rc != 0)
			{
				status = (enum akoStatus)rc;
				complain("akoDecodeExt");
				goto failure;
			}
		}
		if (st.wavelet != AKO_WAVELET_NONE)
			fire(&cb, t, tiles, AKO_EVENT_WAVELET_END);

		fire(&cb, t, tiles, AKO_EVENT_FORMAT_START);
		fire(&cb, t, tiles, AKO_EVENT_FORMAT_END);
	}

decoded:
	plan_release(1, plan, 1);
	if (streams != NULL)
		cb.free(streams);

	if (out_s != NULL)
		*out_s = st;
	if (out_channels != NULL)
		*out_channels = channels;
	if (out_w != NULL)
		*out_w = image_w;
	if (out_h != NULL)
		*out_h = image_h;
	if (out_status != NULL)
		*out_status = AKO_OK;
	return image;

failure:
	akoHostKagariTokensFree(&tokens);
	if (jobs != NULL)
	{
		for (size_t t = 0; t < n_jobs; t++)
			akoHostKagariTokensFree(&jobs[t].tok);
		free(jobs);
	}
	plan_release(1, plan, 0);
	if (cb.free != NULL)
	{
		if (streams != NULL)
			cb.free(streams);
		if (image != NULL)
			cb.free(image);
	}
	if (out_status != NULL)
		*out_status = status;
	return NULL;
}
