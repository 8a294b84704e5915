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
#include <sys/mman.h>
#include <unistd.h>

/* AKO_HIP_TRACE=1: wall-clock of the stages of the drivers on stderr (where a call's time goes) */
#include <time.h>
static double now_ms(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double)t.tv_sec * 1e3 + (double)t.tv_nsec * 1e-6;
}
static int tracing(void)
{
	static int on = -1;
	if (on < 0)
		on = (getenv("AKO_HIP_TRACE") != NULL);
	return on;
}
#define TRACE(label, since)                                                                     \
	do                                                                                          \
	{                                                                                           \
		if (tracing())                                                                          \
		{                                                                                       \
			const double t_ = now_ms();                                                         \
			fprintf(stderr, "libako trace: %-28s %8.3f ms\n", (label), t_ - (since));           \
			(since) = t_;                                                                       \
		}                                                                                       \
	} while (0)

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
	uint64_t tuning; /* akoHipTuningSignature(): a plan created under other AKO_HIP_* knobs is not reused */
};

/* one slot per direction: an encoder's settings never equal a decoder's (quantization is not in the head).
 * The slots hang off a pthread key.  A thread that exits must hand its plans back -- device image and stream
 * staging, entropy-stage buffers, scratch planes and an owned HIP stream, around 1 GB for an 8192x8192 RGBA plan,
 * which a thread-per-request caller would otherwise leak once per thread -- but it cannot destroy them itself: by
 * the time key destructors run, the HIP runtime's own per-thread state is gone and its calls no longer free
 * anything reliably.  So the dying thread only PARKS its plans in a small process-wide pool (no HIP call); the next
 * thread that needs a plan of the same shape takes it from there (a thread-per-request server keeps reusing the
 * same few plans), and what does not fit in the pool is destroyed by the next live thread that enters the library. */
struct plan_slots
{
	akoHipPlan* plans[2];
	struct plan_key keys[2];
	/* the decoder's token lists of the previous call, emptied but not freed: for a large image they are tens of
	 * megabytes whose pages cost more to hand back and fault in again than the parse that fills them */
	struct akoKagariTokens spare;
};
#define SPARE_TOKENS_MAX ((size_t)256 << 20) /* bytes of list capacity a thread keeps for its next call (an 8192 x 8192 RGBA
                                              * stream's lists are ~150 MB; bigger ones are given back: ADVICE r2) */

#define POOL_PLANS 8
#define DOOMED_CHUNK 64
static pthread_mutex_t pool_mutex = PTHREAD_MUTEX_INITIALIZER;
static struct
{
	akoHipPlan* plan;
	struct plan_key key;
	int slot;
} pool[POOL_PLANS];
static akoHipPlan** doomed = NULL; /* grows on demand: a plan that cannot be parked is NEVER dropped (each holds device memory) */
static size_t n_doomed = 0, cap_doomed = 0;

static pthread_key_t slots_key;
static pthread_once_t slots_once = PTHREAD_ONCE_INIT;
static int slots_key_ok = 0;

/* thread exit: no HIP call in here */
static void slots_park(void* arg)
{
	struct plan_slots* sl = arg;
	if (sl == NULL)
		return;
	pthread_mutex_lock(&pool_mutex);
	for (int k = 0; k < 2; k++)
	{
		if (sl->plans[k] == NULL)
			continue;
		int placed = 0;
		for (int e = 0; e < POOL_PLANS && !placed; e++)
			if (pool[e].plan == NULL)
			{
				pool[e].plan = sl->plans[k], pool[e].key = sl->keys[k], pool[e].slot = k;
				placed = 1;
			}
		if (!placed)
		{
			if (n_doomed == cap_doomed)
			{
				akoHipPlan** grown = realloc(doomed, (cap_doomed + DOOMED_CHUNK) * sizeof *doomed); /* (no HIP call) */
				if (grown != NULL)
					doomed = grown, cap_doomed += DOOMED_CHUNK;
			}
			if (n_doomed < cap_doomed)
				doomed[n_doomed++] = sl->plans[k], placed = 1;
			/* (only a failed realloc of a few hundred bytes loses the plan to the process) */
		}
	}
	pthread_mutex_unlock(&pool_mutex);
	akoHostKagariTokensFree(&sl->spare);
	free(sl);
}

static void slots_make_key(void)
{
	slots_key_ok = (pthread_key_create(&slots_key, slots_park) == 0);
}

static struct plan_slots* thread_slots(int create)
{
	pthread_once(&slots_once, slots_make_key);
	if (!slots_key_ok)
		return NULL;
	struct plan_slots* sl = pthread_getspecific(slots_key);
	if (sl == NULL && create)
	{
		if ((sl = calloc(1, sizeof *sl)) == NULL)
			return NULL;
		if (pthread_setspecific(slots_key, sl) != 0)
		{
			free(sl);
			return NULL;
		}
	}
	return sl;
}

/* called by live threads: destroy what exited threads could not park */
static void reap_doomed(void)
{
	if (__atomic_load_n(&n_doomed, __ATOMIC_RELAXED) == 0)
		return;
	pthread_mutex_lock(&pool_mutex);
	akoHipPlan** mine = doomed;
	const size_t n = n_doomed;
	doomed = NULL, n_doomed = 0, cap_doomed = 0;
	pthread_mutex_unlock(&pool_mutex);
	for (size_t k = 0; k < n; k++)
		akoHipPlanDestroy(mine[k]);
	free(mine);
}

static akoHipPlan* pool_take(int slot, const struct plan_key* key)
{
	akoHipPlan* p = NULL;
	pthread_mutex_lock(&pool_mutex);
	for (int e = 0; e < POOL_PLANS && p == NULL; e++)
		if (pool[e].plan != NULL && pool[e].slot == slot && memcmp(&pool[e].key, key, sizeof *key) == 0)
			p = pool[e].plan, pool[e].plan = NULL;
	pthread_mutex_unlock(&pool_mutex);
	return p;
}

static void band_pool_release(void);
/* Explicit release: the calling thread's cached plans, every plan parked by threads that have exited, and the idle plans of
 * the band route. */
AKO_API void akoHipThreadRelease(void)
{
	struct plan_slots* sl = thread_slots(0);
	for (int k = 0; sl != NULL && k < 2; k++)
		if (sl->plans[k] != NULL)
		{
			akoHipPlanDestroy(sl->plans[k]);
			sl->plans[k] = NULL;
		}
	if (sl != NULL)
		akoHostKagariTokensFree(&sl->spare);
	akoHipPlan* parked[POOL_PLANS];
	pthread_mutex_lock(&pool_mutex);
	for (int e = 0; e < POOL_PLANS; e++)
		parked[e] = pool[e].plan, pool[e].plan = NULL;
	pthread_mutex_unlock(&pool_mutex);
	for (int e = 0; e < POOL_PLANS; e++)
		if (parked[e] != NULL)
			akoHipPlanDestroy(parked[e]);
	reap_doomed();
	band_pool_release();
}

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
	key.tuning = akoHipTuningSignature();

	reap_doomed();
	struct plan_slots* sl = thread_slots(1);
	if (sl != NULL && sl->plans[slot] != NULL)
	{
		akoHipPlan* p = sl->plans[slot];
		sl->plans[slot] = NULL; /* taken out while in use: an event callback may call back into the library */
		if (plan_cache_on() && memcmp(&key, &sl->keys[slot], sizeof key) == 0)
			return p;
		akoHipPlanDestroy(p);
	}
	if (sl != NULL)
		sl->keys[slot] = key;
	if (plan_cache_on())
	{
		akoHipPlan* parked = pool_take(slot, &key); /* left behind by a thread that has exited */
		if (parked != NULL)
			return parked;
	}
	/* an own stream per plan: calls from different host threads overlap on the GPU instead of queueing up on
	 * the legacy default stream (every driver call below ends with a synchronisation of that stream) */
	return akoHipPlanCreate(key.device, st, channels, w, h, 1, NULL, AKO_HIP_PLAN_OWN_STREAM, status);
}

static void plan_release(int slot, akoHipPlan* plan, int healthy)
{
	if (plan == NULL)
		return;
	struct plan_slots* sl = thread_slots(0);
	if (healthy && plan_cache_on() && sl != NULL && sl->plans[slot] == NULL)
		sl->plans[slot] = plan; /* its key was stored by plan_acquire */
	else
		akoHipPlanDestroy(plan);
}

/* the calling thread's spare token lists (empty ones if it has none); taken OUT while in use */
static void tokens_take(struct akoKagariTokens* tok)
{
	memset(tok, 0, sizeof *tok);
	struct plan_slots* sl = plan_cache_on() ? thread_slots(1) : NULL;
	if (sl != NULL)
	{
		*tok = sl->spare;
		memset(&sl->spare, 0, sizeof sl->spare);
		tok->n_literals = 0, tok->n_runs = 0;
	}
}
static void* tokens_free_main(void* arg)
{
	struct akoKagariTokens* t = arg;
	akoHostKagariTokensFree(t);
	free(t);
	return NULL;
}
static void tokens_give(struct akoKagariTokens* tok)
{
	struct plan_slots* sl = plan_cache_on() ? thread_slots(0) : NULL;
	const size_t held = tok->cap_literals * sizeof(int16_t) + tok->cap_runs * sizeof(struct akoKagariRun);
	if (sl != NULL && sl->spare.literals == NULL && sl->spare.runs == NULL && held <= SPARE_TOKENS_MAX)
	{
		sl->spare = *tok;
		memset(tok, 0, sizeof *tok);
	}
	else if (held >= ((size_t)64 << 20))
	{
		/* handing gigabytes of touched pages back takes the kernel as long as parsing them did (16384 x 16384 in
		 * 256-pixel tiles: 0.2 s): a detached thread does it while the caller already has its image */
		struct akoKagariTokens* boxed = malloc(sizeof *boxed);
		pthread_t th;
		pthread_attr_t at;
		int started = 0;
		if (boxed != NULL && pthread_attr_init(&at) == 0)
		{
			*boxed = *tok;
			pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
			started = (pthread_create(&th, &at, tokens_free_main, boxed) == 0);
			pthread_attr_destroy(&at);
		}
		if (started)
			memset(tok, 0, sizeof *tok);
		else
		{
			free(boxed);
			akoHostKagariTokensFree(tok);
		}
	}
	else
		akoHostKagariTokensFree(tok);
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
	size_t lit_base, run_base; /* where the tile's tokens go in the image's lists (merge_window) */
	int merged;
};
struct tile_pool
{
	struct tile_job* jobs;
	size_t count;
	size_t next; /* atomically incremented */
	struct akoKagariTokens* merge_into; /* NULL: parse; else: move the parsed lists there */
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
		if (pool->merge_into != NULL)
		{
			struct akoKagariTokens* dst = pool->merge_into;
			if (j->tok.n_literals)
				memcpy(dst->literals + j->lit_base, j->tok.literals, j->tok.n_literals * sizeof(int16_t));
			for (size_t k = 0; k < j->tok.n_runs; k++)
			{
				struct akoKagariRun r = j->tok.runs[k];
				r.after += (uint32_t)j->lit_base;
				dst->runs[j->run_base + k] = r;
			}
			akoHostKagariTokensFree(&j->tok);
			j->merged = 1;
			continue;
		}
		/* one thread per tile here: the tiles are what runs in parallel */
		j->used = (j->payload != NULL) ? akoHostKagariTokenizeWith(1, j->values, j->block, j->payload, j->out_base, &j->tok) : 0;
	}
}

static size_t tokenize_workers(void)
{
	long cores = sysconf(_SC_NPROCESSORS_ONLN);
	size_t workers = (cores > 1) ? (size_t)cores : 1;
	return workers > 16 ? 16 : workers;
}

static void run_tile_pool(struct tile_pool* pool, size_t workers)
{
	pthread_t th[16];
	size_t started = 0;
	for (size_t k = 1; k < workers; k++) /* the calling thread is worker 0 */
		if (pthread_create(&th[started], NULL, tile_worker, pool) == 0)
			started++;
	tile_worker(pool);
	for (size_t k = 0; k < started; k++)
		pthread_join(th[k], NULL);
}

/* The parsed lists of a window of tiles -> the image's lists, in tile order, by the same worker threads (one sequential
 * append per tile was a third of the decoder's time for a 8192 x 8192 image in 512-pixel tiles).  Only when every tile of
 * the window parsed cleanly: otherwise nothing moves and the per-tile loop of akoDecodeExt reports the first broken tile
 * exactly as before.  'remaining_tiles' sizes the lists for the windows still to come (address space, not memory). */
static void merge_window(struct tile_job* jobs, size_t count, struct akoKagariTokens* dst, size_t remaining_tiles)
{
	size_t lits = 0, runs = 0;
	for (size_t k = 0; k < count; k++)
	{
		if (jobs[k].payload == NULL || jobs[k].used == 0 || jobs[k].used != jobs[k].block)
			return;
		jobs[k].lit_base = dst->n_literals + lits, jobs[k].run_base = dst->n_runs + runs;
		lits += jobs[k].tok.n_literals, runs += jobs[k].tok.n_runs;
	}
	if (dst->n_literals + lits > 0xFFFFFFF0ull || lits < ((size_t)1 << 16))
		return; /* (small windows: the plain appends are faster than threads) */
	const size_t more = 1 + remaining_tiles / count;
	if (!akoHostKagariTokensReserve(dst, lits + lits / 8 * more * 9, runs + runs / 8 * more * 9) &&
	    !akoHostKagariTokensReserve(dst, lits, runs))
		return;
	struct tile_pool pool = {jobs, count, 0, dst};
	size_t workers = tokenize_workers();
	run_tile_pool(&pool, workers > count ? count : workers);
	dst->n_literals += lits, dst->n_runs += runs;
}

static void tokenize_tiles(struct tile_job* jobs, size_t count)
{
	struct tile_pool pool = {jobs, count, 0, NULL};
	size_t workers = tokenize_workers();
	if (workers > count)
		workers = count;
	/* threads only where they pay: a window of small tiles is parsed faster than a thread starts */
	size_t payload = 0;
	for (size_t k = 0; k < count; k++)
		payload += jobs[k].block;
	if (payload < 64 * 1024)
		workers = 1;
	run_tile_pool(&pool, workers);
}

/* ---- first touch of a large result buffer, off the critical path ---------------------------------------------
 * The decoded image goes into memory fresh from the caller's malloc.  Fresh pages are mapped on first touch, and
 * for a 268 MB image that costs the copy back from the device more than the copy itself (65536 page faults in the
 * copying thread).  The pages are touched here instead, by a few helper threads, while the caller's thread is still
 * busy parsing the entropy-coded input; the buffer is the library's own until it is returned, so writing zeros
 * into it is invisible. */
#define TOUCH_THREADS 16
struct toucher
{
	uint8_t* base;
	size_t bytes;
	pthread_t thread;
	int started;
};
static void* touch_main(void* arg)
{
	struct toucher* t = arg;
	for (size_t off = 0; off < t->bytes; off += 4096)
		((volatile uint8_t*)t->base)[off] = 0;
	return NULL;
}
static void touch_begin(struct toucher* ts, uint8_t* base, size_t bytes)
{
	memset(ts, 0, TOUCH_THREADS * sizeof *ts);
	if (bytes < ((size_t)16 << 20))
		return;
	/* AKO_HIP_TOUCH_THREADS: helper threads (default 4, 0 = none, 16 at most) */
	size_t threads = 4;
	const char* e = getenv("AKO_HIP_TOUCH_THREADS");
	if (e != NULL && e[0] != 0)
		threads = (size_t)atol(e);
	if (threads > TOUCH_THREADS)
		threads = TOUCH_THREADS;
	if (threads == 0)
		return;
	/* huge pages where the system hands them out on request: 128 faults instead of 65536 for a 268 MB image */
	{
		const uintptr_t lo = ((uintptr_t)base + ((size_t)2 << 20) - 1) & ~(uintptr_t)(((size_t)2 << 20) - 1);
		const uintptr_t hi = ((uintptr_t)base + bytes) & ~(uintptr_t)(((size_t)2 << 20) - 1);
		if (hi > lo)
			(void)madvise((void*)lo, (size_t)(hi - lo), MADV_HUGEPAGE);
	}
	const size_t part = ((bytes / threads) + 4095) & ~(size_t)4095;
	for (size_t k = 0; k < threads; k++)
	{
		const size_t lo = k * part;
		if (lo >= bytes)
			break;
		ts[k].base = base + lo, ts[k].bytes = (lo + part < bytes) ? part : bytes - lo;
		ts[k].started = (pthread_create(&ts[k].thread, NULL, touch_main, &ts[k]) == 0);
	}
}
static void touch_end(struct toucher* ts)
{
	for (size_t k = 0; k < TOUCH_THREADS; k++)
		if (ts[k].started)
		{
			pthread_join(ts[k].thread, NULL);
			ts[k].started = 0;
		}
}

/* ---- several devices: a tiled image split into bands of whole tile rows -------------------------------------
 * Tiles are independent and stored in raster order (library/encode.c:115-205), so a band of whole tile rows is
 * itself a valid image whose stream -- and whose blob body -- is exactly that slice of the whole image's.  With
 * AKO_HIP_DEVICES naming more than one device ("all", or a list such as "0,1,2,3"; a device may be named twice)
 * akoEncodeExt / akoDecodeExt give every device one band (own plan, own stream, a worker thread of the library
 * each) and join the bodies in band order on the calling thread.  No data moves between devices. */
#define MAX_BANDS 16

static size_t device_list(int* devs, size_t max)
{
	const char* e = getenv("AKO_HIP_DEVICES");
	if (e == NULL || *e == '\0')
		return 0;
	size_t n = 0;
	if (strcmp(e, "all") == 0)
	{
		const int count = akoHipDeviceCount();
		for (int d = 0; d < count && n < max; d++)
			devs[n++] = d;
		return n;
	}
	while (*e != '\0' && n < max)
	{
		char* stop = NULL;
		const long v = strtol(e, &stop, 10);
		if (stop == e)
			break;
		devs[n++] = (int)v;
		e = (*stop == ',') ? stop + 1 : stop;
	}
	return n;
}

struct band
{
	int device, decode;
	struct akoSettings st;
	size_t channels, w, y0, rows;
	akoHipPlan* plan;
	const uint8_t* pixels_in; /* encode */
	uint8_t* body_out;
	size_t body_out_bytes;
	const uint8_t* body_in;   /* decode */
	size_t body_in_bytes;
	uint8_t* pixels_out;
	enum akoStatus status;
	pthread_t thread;
	int started;
	double seconds; /* plan creation + transform + entropy stage + copies of this band */
};

/* the bands of this thread's last akoEncodeExt / akoDecodeExt that was split over devices (akoHipLastBands) */
static __thread struct
{
	size_t n;
	int device[MAX_BANDS];
	double seconds[MAX_BANDS];
	size_t rows[MAX_BANDS];
	size_t plans_created; /* by that call: 0 when every band found its plan in the pool (akoHipLastBandPlansCreated) */
} g_last_bands;

/* ---- plans and threads of the band route outlive the call --------------------------------------------------------
 * A band's plan holds the device buffers of a slice of a LARGE image (16384 x 16384 over eight devices: a quarter of a gigabyte
 * of staging per band); creating and destroying it per call, on a thread created per call, cost more than the band's work
 * (one-GPU rehearsal, round 3: 459 Mpx/s for the unsplit call, 148 cut in two).  So: a process-wide pool of band plans keyed
 * like the per-thread slots (device, band shape, settings, direction, tuning knobs), and one persistent worker thread per
 * band index, which keeps its HIP thread state and its spare token lists from call to call.  One split call at a time owns
 * the workers (a second concurrent one falls back to threads of its own); AKO_HIP_PLAN_CACHE=0 restores create / destroy
 * per call; akoHipThreadRelease() empties the pool. */
#define BAND_POOL 32
static struct
{
	akoHipPlan* plan;
	struct plan_key key;
	int decode, busy;
} band_pool[BAND_POOL];
static pthread_mutex_t band_pool_mutex = PTHREAD_MUTEX_INITIALIZER;
static size_t band_plans_created_total = 0; /* (atomic) */

struct band_worker
{
	pthread_t thread;
	int alive;
	pthread_mutex_t m;
	pthread_cond_t cv;
	struct band* job;
	int done;
};
static struct band_worker band_workers[MAX_BANDS];
static pthread_mutex_t band_route_mutex = PTHREAD_MUTEX_INITIALIZER;

static void band_key(struct plan_key* key, const struct band* b)
{
	memset(key, 0, sizeof *key);
	key->s.wavelet = b->st.wavelet, key->s.color = b->st.color, key->s.wrap = b->st.wrap, key->s.compression = b->st.compression;
	key->s.tiles_dimension = b->st.tiles_dimension, key->s.quantization = b->st.quantization, key->s.gate = b->st.gate;
	key->s.chroma_loss = b->st.chroma_loss, key->s.discard_non_visible = b->st.discard_non_visible;
	key->channels = b->channels, key->w = b->w, key->h = b->rows, key->device = b->device;
	key->tuning = akoHipTuningSignature();
}

static akoHipPlan* band_plan_take(struct band* b)
{
	struct plan_key key;
	band_key(&key, b);
	if (plan_cache_on())
	{
		akoHipPlan* p = NULL;
		pthread_mutex_lock(&band_pool_mutex);
		for (int e = 0; e < BAND_POOL && p == NULL; e++)
			if (band_pool[e].plan != NULL && !band_pool[e].busy && band_pool[e].decode == b->decode &&
			    memcmp(&band_pool[e].key, &key, sizeof key) == 0)
				p = band_pool[e].plan, band_pool[e].busy = 1;
		pthread_mutex_unlock(&band_pool_mutex);
		if (p != NULL)
			return p;
	}
	__atomic_add_fetch(&band_plans_created_total, 1, __ATOMIC_RELAXED);
	return akoHipPlanCreate(b->device, &b->st, b->channels, b->w, b->rows, 1, NULL, AKO_HIP_PLAN_OWN_STREAM, &b->status);
}

static void band_plan_give(struct band* b, akoHipPlan* plan, int healthy)
{
	if (plan == NULL)
		return;
	akoHipPlan* drop = plan;
	if (plan_cache_on())
	{
		struct plan_key key;
		band_key(&key, b);
		pthread_mutex_lock(&band_pool_mutex);
		int mine = -1, empty = -1, idle = -1;
		for (int e = 0; e < BAND_POOL; e++)
		{
			if (band_pool[e].plan == plan)
				mine = e;
			else if (band_pool[e].plan == NULL && empty < 0)
				empty = e;
			else if (band_pool[e].plan != NULL && !band_pool[e].busy && idle < 0)
				idle = e;
		}
		if (mine >= 0) /* taken from the pool: stays, or leaves if the call failed on it */
		{
			if (healthy)
				band_pool[mine].busy = 0, drop = NULL;
			else
				band_pool[mine].plan = NULL, band_pool[mine].busy = 0;
		}
		else if (healthy)
		{
			const int at = (empty >= 0) ? empty : idle; /* a full pool gives up an idle plan of some other shape */
			if (at >= 0)
			{
				drop = band_pool[at].plan; /* NULL for an empty entry */
				band_pool[at].plan = plan, band_pool[at].key = key, band_pool[at].decode = b->decode, band_pool[at].busy = 0;
			}
		}
		pthread_mutex_unlock(&band_pool_mutex);
	}
	if (drop != NULL)
		akoHipPlanDestroy(drop);
}

static void band_pool_release(void) /* akoHipThreadRelease */
{
	akoHipPlan* idle[BAND_POOL];
	int n = 0;
	pthread_mutex_lock(&band_pool_mutex);
	for (int e = 0; e < BAND_POOL; e++)
		if (band_pool[e].plan != NULL && !band_pool[e].busy)
			idle[n++] = band_pool[e].plan, band_pool[e].plan = NULL;
	pthread_mutex_unlock(&band_pool_mutex);
	for (int e = 0; e < n; e++)
		akoHipPlanDestroy(idle[e]);
}

/* akoHostDecodeBody() for a band of a tiled Kagari blob, with the single-device driver's parse: the tiles of a window are
 * tokenized on worker threads and their lists merged by the same threads (tokenize_tiles, merge_window), the token lists
 * stay with the thread for its next call (a band worker is persistent).  A band of 16384 x 8192 in 512-pixel tiles holds
 * 512 bit-streams, 85 MB of payload: parsed tile after tile on the band's own thread it took 1.2 s of the band's 1.3 s. */
static enum akoStatus band_decode_body(akoHipPlan* plan, enum akoCompression compression, const uint8_t* body, size_t body_bytes,
                                       void* pixels)
{
	const size_t tiles = akoHipPlanTiles(plan);
	if (compression == AKO_COMPRESSION_NONE || tiles < 2)
		return akoHostDecodeBody(plan, compression, body, body_bytes, NULL, pixels);
	struct akoKagariTokens tokens;
	tokens_take(&tokens);
	size_t n_jobs = tokenize_workers() * 8;
	if (n_jobs > tiles)
		n_jobs = tiles;
	struct tile_job* jobs = calloc(n_jobs, sizeof *jobs);
	if (jobs == NULL)
	{
		tokens_give(&tokens);
		return AKO_NO_ENOUGH_MEMORY;
	}
	enum akoStatus status = AKO_OK;
	const uint8_t* walk = body;
	const uint8_t* const end = body + body_bytes;
	for (size_t first = 0; first < tiles && status == AKO_OK;)
	{
		const size_t count = (tiles - first < n_jobs) ? tiles - first : n_jobs;
		for (size_t k = 0; k < count; k++)
		{
			size_t o = 0, b = 0;
			akoHipPlanTileInfo(plan, first + k, NULL, NULL, NULL, NULL, &o, &b);
			memset(&jobs[k], 0, sizeof jobs[k]); /* payload == NULL: the chain ran off the input before this tile */
			uint32_t blk = 0;
			if (walk == NULL || (size_t)(end - walk) < 4)
			{
				walk = NULL;
				continue;
			}
			memcpy(&blk, walk, 4);
			if ((size_t)(end - walk) - 4 < blk)
			{
				walk = NULL;
				continue;
			}
			jobs[k].payload = walk + 4, jobs[k].block = blk;
			jobs[k].values = b / 2, jobs[k].out_base = o / 2;
			walk += (size_t)blk + 4;
		}
		tokenize_tiles(jobs, count);
		merge_window(jobs, count, &tokens, tiles - first - count);
		for (size_t k = 0; k < count; k++)
		{
			struct tile_job* j = &jobs[k];
			if (status == AKO_OK && (j->payload == NULL || j->used == 0 || j->used != j->block)) /* compression.c:69-70 */
				status = AKO_BROKEN_INPUT;
			if (status == AKO_OK && !j->merged)
			{
				const size_t base = tokens.n_literals;
				if (base + j->tok.n_literals > 0xFFFFFFF0ull || !akoHostKagariTokensAppend(&tokens, &j->tok, (uint32_t)base))
					status = AKO_NO_ENOUGH_MEMORY;
			}
			if (!j->merged)
				akoHostKagariTokensFree(&j->tok);
		}
		first += count;
	}
	free(jobs);
	if (status == AKO_OK)
	{
		int rc = akoHipKagariExpand(plan, tokens.literals, tokens.n_literals, (const struct akoHipKagariRun*)tokens.runs, tokens.n_runs, NULL, 0);
		if (rc == 0)
			rc = akoHipDecodeDownload(plan, pixels);
		status = (enum akoStatus)rc;
	}
	tokens_give(&tokens);
	return status;
}

static void* band_main(void* arg)
{
	struct band* b = arg;
	akoHipPlan* plan = (b->plan != NULL) ? b->plan : band_plan_take(b); /* (the decoder of uncompressed blobs takes them beforehand) */
	b->plan = NULL;
	if (plan == NULL)
		return NULL;
	if (b->decode)
		b->status = band_decode_body(plan, b->st.compression, b->body_in, b->body_in_bytes, b->pixels_out);
	else
		b->status = akoHostEncodeBody(plan, b->st.compression, b->pixels_in, 0, &b->body_out, &b->body_out_bytes);
	band_plan_give(b, plan, b->status == AKO_OK);
	return NULL;
}

static void* band_main_timed(void* arg)
{
	struct band* b = arg;
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	band_main(arg);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	b->seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
	return NULL;
}

static void* band_worker_main(void* arg)
{
	struct band_worker* w = arg;
	for (;;)
	{
		pthread_mutex_lock(&w->m);
		while (w->job == NULL)
			pthread_cond_wait(&w->cv, &w->m);
		struct band* b = w->job;
		pthread_mutex_unlock(&w->m);
		band_main_timed(b);
		pthread_mutex_lock(&w->m);
		w->job = NULL, w->done = 1;
		pthread_cond_broadcast(&w->cv);
		pthread_mutex_unlock(&w->m);
	}
	return NULL;
}

/* hands band b to persistent worker k (created on first use; the caller holds band_route_mutex); 0 if that is not possible */
static int band_worker_post(size_t k, struct band* b)
{
	struct band_worker* w = &band_workers[k];
	if (!w->alive)
	{
		pthread_attr_t at;
		if (pthread_mutex_init(&w->m, NULL) != 0 || pthread_cond_init(&w->cv, NULL) != 0 || pthread_attr_init(&at) != 0)
			return 0;
		pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
		w->job = NULL, w->done = 0;
		const int rc = pthread_create(&w->thread, &at, band_worker_main, w);
		pthread_attr_destroy(&at);
		if (rc != 0)
			return 0;
		w->alive = 1;
	}
	pthread_mutex_lock(&w->m);
	w->done = 0, w->job = b;
	pthread_cond_broadcast(&w->cv);
	pthread_mutex_unlock(&w->m);
	return 1;
}
static void band_worker_wait(size_t k)
{
	struct band_worker* w = &band_workers[k];
	pthread_mutex_lock(&w->m);
	while (!w->done)
		pthread_cond_wait(&w->cv, &w->m);
	pthread_mutex_unlock(&w->m);
}

/* bands[0 .. n) filled in by the caller; runs them (band 0 on the calling thread) and returns the first failure */
static enum akoStatus run_bands(struct band* bands, size_t n)
{
	const size_t created_before = __atomic_load_n(&band_plans_created_total, __ATOMIC_RELAXED);
	/* started: 2 = a persistent worker has it, 1 = a thread of this call, 0 = nobody yet (run here afterwards) */
	const int pooled = plan_cache_on() && pthread_mutex_trylock(&band_route_mutex) == 0;
	for (size_t k = 1; k < n; k++)
	{
		if (pooled && band_worker_post(k, &bands[k]))
			bands[k].started = 2;
		else
			bands[k].started = (pthread_create(&bands[k].thread, NULL, band_main_timed, &bands[k]) == 0);
	}
	band_main_timed(&bands[0]);
	for (size_t k = 1; k < n; k++)
	{
		if (bands[k].started == 2)
			band_worker_wait(k);
		else if (bands[k].started)
			pthread_join(bands[k].thread, NULL);
		else
			band_main_timed(&bands[k]);
	}
	if (pooled)
		pthread_mutex_unlock(&band_route_mutex);
	g_last_bands.n = n < MAX_BANDS ? n : MAX_BANDS;
	for (size_t k = 0; k < g_last_bands.n; k++)
		g_last_bands.device[k] = bands[k].device, g_last_bands.seconds[k] = bands[k].seconds, g_last_bands.rows[k] = bands[k].rows;
	/* (exact unless another thread split a call at the same time: it is a diagnostic) */
	g_last_bands.plans_created = __atomic_load_n(&band_plans_created_total, __ATOMIC_RELAXED) - created_before;
	enum akoStatus st = AKO_OK;
	for (size_t k = 0; k < n; k++)
		if (st == AKO_OK && bands[k].status != AKO_OK)
			st = bands[k].status;
	return st;
}

/* how a tiled image of `h` rows is cut: n bands of whole tile rows, as equal as they come */
static size_t cut_bands(size_t h, size_t td, size_t n_devices, size_t* y0, size_t* rows)
{
	const size_t tile_rows = (h + td - 1) / td;
	size_t n = n_devices < tile_rows ? n_devices : tile_rows;
	if (n > MAX_BANDS)
		n = MAX_BANDS;
	const size_t per = (tile_rows + n - 1) / n;
	size_t used = 0;
	for (size_t k = 0; k < n; k++)
	{
		const size_t first = k * per;
		if (first >= tile_rows)
			break;
		const size_t last = (first + per < tile_rows) ? first + per : tile_rows;
		y0[used] = first * td;
		rows[used] = ((last * td < h) ? last * td : h) - first * td;
		used++;
	}
	return used;
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

	/* several devices (AKO_HIP_DEVICES) and a tiled image of two tile rows or more: one band per device */
	{
		int devs[MAX_BANDS];
		const size_t nd = device_list(devs, MAX_BANDS);
		const size_t td = st.tiles_dimension;
		g_last_bands.n = 0;
		if (nd > 1 && td != 0 && image_h > td)
		{
			struct band bands[MAX_BANDS];
			size_t y0[MAX_BANDS], rows[MAX_BANDS];
			memset(bands, 0, sizeof bands);
			const size_t nb = cut_bands(image_h, td, nd, y0, rows);
			const size_t tiles_all = ((image_w + td - 1) / td) * ((image_h + td - 1) / td);
			for (size_t k = 0; k < nb; k++)
			{
				bands[k].device = devs[k], bands[k].st = st, bands[k].channels = channels, bands[k].w = image_w;
				bands[k].y0 = y0[k], bands[k].rows = rows[k];
				bands[k].pixels_in = (const uint8_t*)in + y0[k] * image_w * channels;
			}
			/* the reference's event sequence, per tile in order (encode.c:132-184); all device work sits in tile 0's
			 * WAVELET bracket, as in the single-device driver */
			for (size_t t = 0; t < tiles_all; t++)
			{
				fire(&cb, t, tiles_all, AKO_EVENT_FORMAT_START);
				fire(&cb, t, tiles_all, AKO_EVENT_FORMAT_END);
				if (st.wavelet != AKO_WAVELET_NONE)
					fire(&cb, t, tiles_all, AKO_EVENT_WAVELET_START);
				if (t == 0 && (status = run_bands(bands, nb)) != AKO_OK)
				{
					if (status != AKO_ERROR)
						complain("akoEncodeExt (bands)");
					for (size_t k = 0; k < nb; k++)
						free(bands[k].body_out);
					goto failure;
				}
				if (st.wavelet != AKO_WAVELET_NONE)
					fire(&cb, t, tiles_all, AKO_EVENT_WAVELET_END);
				fire(&cb, t, tiles_all, AKO_EVENT_COMPRESSION_START);
				if (t == 0)
				{
					size_t total = 0;
					for (size_t k = 0; k < nb; k++)
						total += bands[k].body_out_bytes;
					uint8_t* grown = cb.realloc(blob, blob_size + total);
					if (grown == NULL)
					{
						for (size_t k = 0; k < nb; k++)
							free(bands[k].body_out);
						status = AKO_NO_ENOUGH_MEMORY;
						goto failure;
					}
					blob = grown;
					for (size_t k = 0; k < nb; k++)
					{
						memcpy(blob + blob_size, bands[k].body_out, bands[k].body_out_bytes);
						blob_size += bands[k].body_out_bytes;
						free(bands[k].body_out);
					}
				}
				fire(&cb, t, tiles_all, AKO_EVENT_COMPRESSION_END);
			}
			goto done;
		}
	}

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
	struct toucher touchers[TOUCH_THREADS];
	memset(touchers, 0, sizeof touchers);
	double t_call = tracing() ? now_ms() : 0.0;

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

	/* The head is untrusted: before anything is sized from it, it must be able to describe this blob.  Every tile
	 * costs at least 5 bytes under Kagari (block size + one payload byte) and its whole stream (>= 2 bytes per
	 * sample) without compression; products that leave 64 bits are not an image.  (A forged 16 byte blob claiming
	 * 2^32 x 2^32 pixels in 8 pixel tiles would otherwise ask for ~10^17 tile records.) */
	{
		const size_t td = st.tiles_dimension;
		const size_t tiles_x = (td != 0) ? (image_w + td - 1) / td : 1;
		const size_t tiles_y = (td != 0) ? (image_h + td - 1) / td : 1;
		size_t n_tiles = 0, samples = 0, raw_bytes = 0, need = 0;
		if (__builtin_mul_overflow(tiles_x, tiles_y, &n_tiles) || __builtin_mul_overflow(image_w, image_h, &samples) ||
		    __builtin_mul_overflow(samples, channels, &samples) || __builtin_mul_overflow(samples, (size_t)2, &raw_bytes) ||
		    raw_bytes > ((size_t)1 << 46))
		{
			status = AKO_NO_ENOUGH_MEMORY;
			goto failure;
		}
		if (st.compression != AKO_COMPRESSION_NONE)
		{
			if (__builtin_mul_overflow(n_tiles, (size_t)5, &need))
				need = (size_t)-1;
		}
		else
			need = raw_bytes;
		if (input_size - sizeof(struct akoHead) < need)
		{
			status = AKO_BROKEN_INPUT;
			goto failure;
		}
	}

	/* several devices (AKO_HIP_DEVICES) and a tiled image of two tile rows or more: one band per device */
	{
		int devs[MAX_BANDS];
		const size_t nd = device_list(devs, MAX_BANDS);
		const size_t td = st.tiles_dimension;
		g_last_bands.n = 0;
		if (nd > 1 && td != 0 && image_h > td)
		{
			struct band bands[MAX_BANDS];
			size_t y0[MAX_BANDS], rows[MAX_BANDS];
			memset(bands, 0, sizeof bands);
			const size_t nb = cut_bands(image_h, td, nd, y0, rows);
			const size_t tiles_x = (image_w + td - 1) / td;
			const size_t tiles_all = tiles_x * ((image_h + td - 1) / td);
			if ((image = cb.malloc(image_w * image_h * channels)) == NULL)
			{
				status = AKO_NO_ENOUGH_MEMORY;
				goto failure;
			}
			{
				/* a fresh buffer that the bands' downloads touch first: 2 MB faults instead of 4 KB ones where the system allows */
				const size_t image_bytes = image_w * image_h * channels;
				const uintptr_t lo = ((uintptr_t)image + ((size_t)2 << 20) - 1) & ~(uintptr_t)(((size_t)2 << 20) - 1);
				const uintptr_t hi = ((uintptr_t)image + image_bytes) & ~(uintptr_t)(((size_t)2 << 20) - 1);
				if (image_bytes >= ((size_t)16 << 20) && hi > lo)
					(void)madvise((void*)lo, (size_t)(hi - lo), MADV_HUGEPAGE);
			}
			/* where every band's body starts: fixed stream sizes without compression, the chain of block sizes with */
			const uint8_t* at = (const uint8_t*)input + sizeof(struct akoHead);
			const uint8_t* const stop = (const uint8_t*)input + input_size;
			for (size_t k = 0; k < nb && status == AKO_OK; k++)
			{
				struct band* b = &bands[k];
				b->device = devs[k], b->decode = 1, b->st = st, b->channels = channels, b->w = image_w;
				b->y0 = y0[k], b->rows = rows[k];
				b->pixels_out = image + y0[k] * image_w * channels;
				b->body_in = at;
				if (st.compression == AKO_COMPRESSION_NONE)
				{
					/* plans are taken (from the band pool, or created) here, one after the other, for the sizes of their streams */
					if ((b->plan = band_plan_take(b)) == NULL)
					{
						status = (b->status != AKO_OK) ? b->status : AKO_ERROR;
						break;
					}
					const size_t need = akoHipPlanStreamBytes(b->plan);
					if ((size_t)(stop - at) < need)
						status = AKO_BROKEN_INPUT;
					else
						at += need;
				}
				else
				{
					const size_t band_tiles = tiles_x * ((rows[k] + td - 1) / td);
					for (size_t t = 0; t < band_tiles; t++)
					{
						uint32_t block = 0;
						if ((size_t)(stop - at) < 4)
						{
							status = AKO_BROKEN_INPUT;
							break;
						}
						memcpy(&block, at, 4);
						if ((size_t)(stop - at) - 4 < block)
						{
							status = AKO_BROKEN_INPUT;
							break;
						}
						at += (size_t)block + 4;
					}
				}
				b->body_in_bytes = (size_t)(at - b->body_in);
			}
			if (status == AKO_OK)
				status = run_bands(bands, nb);
			else
				for (size_t k = 0; k < nb; k++)
					band_plan_give(&bands[k], bands[k].plan, 1);
			if (status != AKO_OK)
			{
				if (status != AKO_BROKEN_INPUT)
					complain("akoDecodeExt (bands)");
				goto failure;
			}
			for (size_t t = 0; t < tiles_all; t++) /* decode.c:145-207 */
			{
				fire(&cb, t, tiles_all, AKO_EVENT_COMPRESSION_START);
				fire(&cb, t, tiles_all, AKO_EVENT_COMPRESSION_END);
				if (st.wavelet != AKO_WAVELET_NONE)
				{
					fire(&cb, t, tiles_all, AKO_EVENT_WAVELET_START);
					fire(&cb, t, tiles_all, AKO_EVENT_WAVELET_END);
				}
				fire(&cb, t, tiles_all, AKO_EVENT_FORMAT_START);
				fire(&cb, t, tiles_all, AKO_EVENT_FORMAT_END);
			}
			goto decoded;
		}
	}

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
		double t_trace = t_call;
		TRACE("decode: head, plan", t_trace);
		if ((image = cb.malloc(image_w * image_h * channels)) == NULL)
		{
			status = AKO_NO_ENOUGH_MEMORY;
			goto failure;
		}
		touch_begin(touchers, image, image_w * image_h * channels);
		tokens_take(&tokens);

		/* several tiles: they are parsed on worker threads, a bounded WINDOW of tiles at a time (host memory stays
		 * proportional to the window, not to the tile count: an 8192x8192 image in 8 pixel tiles has a million),
		 * and merged in tile order as the loop below reaches them */
		const uint8_t* walk = cursor; /* runs ahead of `cursor` along the chain of block sizes */
		size_t win_first = 0, win_count = 0;
		if (tiles > 1)
		{
			n_jobs = tokenize_workers() * 8;
			if (n_jobs > tiles)
				n_jobs = tiles;
			if ((jobs = calloc(n_jobs, sizeof *jobs)) == NULL)
			{
				status = AKO_NO_ENOUGH_MEMORY;
				goto failure;
			}
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
			if (jobs != NULL && t == win_first + win_count)
			{
				/* next window: walk the chain of block sizes, then parse its tiles at once */
				win_first = t;
				win_count = (tiles - t < n_jobs) ? tiles - t : n_jobs;
				for (size_t k = 0; k < win_count; k++)
				{
					size_t o = 0, b = 0;
					akoHipPlanTileInfo(plan, t + k, NULL, NULL, NULL, NULL, &o, &b);
					memset(&jobs[k], 0, sizeof jobs[k]); /* payload == NULL: the chain ran off the input before this tile */
					uint32_t blk = 0;
					if (walk == NULL || (size_t)(end - walk) < 4)
					{
						walk = NULL;
						continue;
					}
					memcpy(&blk, walk, 4);
					if ((size_t)(end - walk) - 4 < blk)
					{
						walk = NULL;
						continue;
					}
					jobs[k].payload = walk + 4, jobs[k].block = blk;
					jobs[k].values = b / 2, jobs[k].out_base = o / 2;
					walk += (size_t)blk + 4;
				}
				tokenize_tiles(jobs, win_count);
				merge_window(jobs, win_count, &tokens, tiles - t - win_count);
			}
			if (jobs != NULL)
			{
				/* merge this tile's list: literals are appended, 'after' counts become global */
				struct tile_job* j = &jobs[t - win_first];
				if (j->used == 0 || j->used != block) /* compression.c:69-70 */
				{
					status = AKO_BROKEN_INPUT;
					goto failure;
				}
				if (!j->merged)
				{
					const size_t base = tokens.n_literals;
					if (base + j->tok.n_literals > 0xFFFFFFF0ull ||
					    !akoHostKagariTokensAppend(&tokens, &j->tok, (uint32_t)base))
					{
						status = AKO_NO_ENOUGH_MEMORY;
						goto failure;
					}
					akoHostKagariTokensFree(&j->tok);
				}
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
				TRACE("decode: parse bit-streams", t_trace);
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
				TRACE("decode: upload + expand runs", t_trace);
				touch_end(touchers);
				const int rc = akoHipDecodeDownload(plan, image);
				TRACE("decode: transform + download", t_trace);
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
		tokens_give(&tokens);
		free(jobs); /* every per-tile list was released when it was merged */
		jobs = NULL;
		TRACE("decode: release token lists", t_trace);
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
	TRACE("decode: the whole call", t_call);

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
	touch_end(touchers); /* nobody may still write into the image when it is freed */
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

/* ---- ratio search in one call ---------------------------------------------------------------
 * tools/akoenc.cpp:112-217 looks for the quantization that lands near ratio:1 by encoding the image again and again:
 * quantization 0 for the upper bracket, then x4 steps until the blob is small enough, then bisection, then once more
 * with the winner.  Every one of those encodes uploads the same pixels and runs the same lifting; only the stores of
 * the C / B / D sub-bands and the lift heads depend on the quantization (library/lifting.c:154-168,253-267).  Here the
 * pixels are uploaded once and transformed once per colour transformation (quantization 0 with gate 0 keeps plain
 * YCoCg, every other candidate runs YCoCg_Q: library/encode.c:59-64 -- two transforms at most, one when the colour does
 * not depend on it); a candidate then costs a pass over the unquantized coefficients (akoHipRequantize) and the
 * device entropy stage, and only the winner's body crosses the link.  Control flow and arithmetic of the search are
 * the reference's, so it settles on the same quantization and the blob is the one repeated akoEncodeExt calls give. */
struct ratio_state
{
	akoHipPlan* plan_q;    /* colour of the candidates with quantization > 0 (or gate > 0); settings q = 0, g = 0 */
	akoHipPlan* plan_0;    /* colour of the quantization 0 candidate when that differs, else NULL */
	struct akoSettings base;
	int have_q, have_0;    /* unquantized streams present on the device */
	const void* pixels;
	int encodes, transforms;
	enum akoStatus status;
};

/* size of the blob for `quantization`, 0 when it fails (a tile that does not shrink: encode.c:159-164); the body
 * stays on the device, in the entropy stage of the plan that is returned through *used */
static size_t ratio_candidate(struct ratio_state* rs, int quantization, akoHipPlan** used)
{
	struct akoSettings s = rs->base;
	s.quantization = quantization;
	/* plan_0 exists only where quantization 0 (with the caller's gate) keeps another colour than the rest */
	akoHipPlan* plan = (rs->plan_0 != NULL && quantization <= 0) ? rs->plan_0 : rs->plan_q;
	int* have = (plan == rs->plan_0) ? &rs->have_0 : &rs->have_q;
	rs->encodes++;
	int rc = 0;
	if (!*have)
	{
		/* pixels go up once: the second plan transforms the first plan's device copy */
		if (rs->have_q || rs->have_0)
		{
			akoHipPlan* other = (plan == rs->plan_0) ? rs->plan_q : rs->plan_0;
			void* d_img = akoHipPlanDeviceImages(other);
			void* d_str = akoHipPlanDeviceStreams(plan);
			rc = (d_img != NULL && d_str != NULL) ? akoHipEncode(plan, d_img, d_str) : (int)AKO_NO_ENOUGH_MEMORY;
		}
		else
			rc = akoHipEncodeUpload(plan, rs->pixels);
		if (rc == 0)
			rc = akoHipSynchronize(plan);
		if (rc != 0)
		{
			rs->status = (enum akoStatus)rc;
			return 0;
		}
		*have = 1;
		rs->transforms++;
	}
	void* d_q = NULL;
	size_t body = 0, bad = 0;
	if ((rc = akoHipRequantize(plan, quantization, s.gate, NULL, &d_q)) == 0)
		rc = akoHipKagariEncode(plan, d_q, 0, &body, &bad);
	if (rc != 0)
	{
		rs->status = (enum akoStatus)rc; /* AKO_ERROR: did not shrink -- the search goes on with size 0, as the tool's does */
		return 0;
	}
	rs->status = AKO_OK;
	*used = plan;
	return sizeof(struct akoHead) + body;
}

AKO_API size_t akoEncodeRatioExt(const struct akoCallbacks* c, const struct akoSettings* s, size_t channels, size_t image_w,
                                 size_t image_h, const void* in, int ratio, void** out, int* out_quantization,
                                 int* out_encodes, int* out_transforms, enum akoStatus* out_status)
{
	const struct akoCallbacks cb = (c != NULL) ? *c : akoDefaultCallbacks();
	struct akoSettings base = (s != NULL) ? *s : akoDefaultSettings();
	if (out_encodes != NULL)
		*out_encodes = 1;
	if (out_transforms != NULL)
		*out_transforms = 1;
	/* what the tool does without a search (tools/akoenc.cpp:116-128) */
	if (ratio <= 1 || base.wavelet == AKO_WAVELET_NONE || base.compression == AKO_COMPRESSION_NONE)
	{
		if (ratio == 1)
			base.quantization = 0, base.gate = 0; /* lossless */
		if (out_quantization != NULL)
			*out_quantization = base.quantization;
		return akoEncodeExt(c, &base, channels, image_w, image_h, in, out, out_status);
	}

	enum akoStatus status = AKO_OK;
	uint8_t* blob = NULL;
	struct ratio_state rs;
	memset(&rs, 0, sizeof rs);
	rs.base = base, rs.pixels = in;
	if (cb.malloc == NULL || cb.realloc == NULL || cb.free == NULL)
	{
		status = AKO_INVALID_CALLBACKS;
		goto failure;
	}
	if (in == NULL)
	{
		status = AKO_INVALID_INPUT;
		goto failure;
	}
	{
		uint8_t head[sizeof(struct akoHead)]; /* validation as in the encoder (head.c:34-64) */
		struct akoSettings probe = base;
		probe.color = akoHipEffectiveColor(&probe);
		if ((status = akoHostHeadWrite(channels, image_w, image_h, &probe, head)) != AKO_OK)
			goto failure;
	}
	{
		/* the unquantized plans: quantization 0, gate 0, colour fixed to what the candidates will have */
		struct akoSettings u = base;
		u.quantization = 1;
		const enum akoColor color_q = akoHipEffectiveColor(&u);
		u.quantization = 0;
		const enum akoColor color_0 = akoHipEffectiveColor(&u); /* with the caller's gate */
		u.gate = 0;
		u.color = color_q;
		if ((rs.plan_q = akoHipPlanCreate(chosen_device(), &u, channels, image_w, image_h, 1, NULL, AKO_HIP_PLAN_OWN_STREAM, &status)) == NULL)
			goto failure;
		if (color_0 != color_q && channels >= 3) /* below three channels no colour transformation runs (format.c:87) */
		{
			u.color = color_0;
			if ((rs.plan_0 = akoHipPlanCreate(chosen_device(), &u, channels, image_w, image_h, 1, NULL, AKO_HIP_PLAN_OWN_STREAM, &status)) == NULL)
				goto failure;
		}
	}

	/* ---- the search, step for step (tools/akoenc.cpp:130-214) ---- */
	const size_t target = (image_w * image_h * channels) / (size_t)ratio;
	const size_t margin = (target * 4) / 100;
	akoHipPlan* last_plan = NULL;
	int q = 0;
	size_t ceil_size = ratio_candidate(&rs, 0, &last_plan), floor_size = ceil_size;
	int ceil_q = 0, floor_q = 0;
	q = 1;
	do
	{
		q *= 4;
		ceil_size = floor_size, ceil_q = floor_q;
		floor_size = ratio_candidate(&rs, q, &last_plan), floor_q = q;
	} while (floor_size > target && q < (1 << 28));

	size_t last_size = floor_size;
	int last_q = floor_q;
#define DISTANCE(a, b) ((a) > (b) ? (a) - (b) : (b) - (a))
	while (DISTANCE(floor_size, ceil_size) > margin && abs(floor_q - ceil_q) > 1)
	{
		q = (ceil_q + floor_q) / 2;
		last_size = ratio_candidate(&rs, q, &last_plan), last_q = q;
		if (last_size > target)
			ceil_size = last_size, ceil_q = q;
		else
			floor_size = last_size, floor_q = q;
	}
	const int take_floor = DISTANCE(floor_size, target) < DISTANCE(ceil_size, target);
#undef DISTANCE
	const int best_q = take_floor ? floor_q : ceil_q;
	const size_t best_size = take_floor ? floor_size : ceil_size;
	/* tools/akoenc.cpp:207-213: the last encode is kept when its SIZE is the winner's (whatever its quantization) */
	if (last_size != best_size)
		last_size = ratio_candidate(&rs, best_q, &last_plan), last_q = best_q;
	if (last_size == 0 || last_plan == NULL)
	{
		status = (rs.status != AKO_OK) ? rs.status : AKO_ERROR;
		goto failure;
	}

	/* the winner: head + body out of the entropy stage */
	if ((blob = cb.malloc(last_size)) == NULL)
	{
		status = AKO_NO_ENOUGH_MEMORY;
		goto failure;
	}
	{
		struct akoSettings fin = base;
		fin.quantization = last_q;
		fin.color = akoHipEffectiveColor(&fin);
		if ((status = akoHostHeadWrite(channels, image_w, image_h, &fin, blob)) != AKO_OK)
			goto failure;
		if ((status = (enum akoStatus)akoHipKagariFetch(last_plan, blob + sizeof(struct akoHead))) != AKO_OK)
			goto failure;
	}
	akoHipPlanDestroy(rs.plan_q);
	if (rs.plan_0 != NULL)
		akoHipPlanDestroy(rs.plan_0);
	if (out_quantization != NULL)
		*out_quantization = last_q;
	if (out_encodes != NULL)
		*out_encodes = rs.encodes;
	if (out_transforms != NULL)
		*out_transforms = rs.transforms;
	if (out_status != NULL)
		*out_status = AKO_OK;
	if (out != NULL)
		*out = blob;
	else
		cb.free(blob);
	return last_size;

failure:
	if (status != AKO_ERROR && status != AKO_INVALID_INPUT && status != AKO_INVALID_CALLBACKS)
		complain("akoEncodeRatioExt");
	if (rs.plan_q != NULL)
		akoHipPlanDestroy(rs.plan_q);
	if (rs.plan_0 != NULL)
		akoHipPlanDestroy(rs.plan_0);
	if (blob != NULL && cb.free != NULL)
		cb.free(blob);
	if (out_status != NULL)
		*out_status = status;
	return 0;
}

/* The bands of the calling thread's last akoEncodeExt / akoDecodeExt that was split over devices (AKO_HIP_DEVICES): device,
 * seconds (plan, transform, entropy stage, copies) and image rows of each, in band order; returns their number. */
/* Plans the calling thread's last split call had to CREATE (0: every band found its plan in the pool). */
AKO_API size_t akoHipLastBandPlansCreated(void)
{
	return g_last_bands.n ? g_last_bands.plans_created : 0;
}

AKO_API size_t akoHipLastBands(int* devices, double* seconds, size_t* rows, size_t cap)
{
	const size_t n = g_last_bands.n < cap ? g_last_bands.n : cap;
	for (size_t k = 0; k < n; k++)
	{
		if (devices)
			devices[k] = g_last_bands.device[k];
		if (seconds)
			seconds[k] = g_last_bands.seconds[k];
		if (rows)
			rows[k] = g_last_bands.rows[k];
	}
	return g_last_bands.n;
}
