/*
 * ako_kagari.c -- host entropy stage "Kagari" (reference: library/kagari.c:59-366).
 *
 * Stays on the host by design (BASELINE north star); the GPU hands over the raw coefficient
 * stream and this file turns it into the bit-stream the reference writes:
 *
 *   - every int16 coefficient c is sent as the Elias-gamma code of zigzag(c) + 1
 *     (kagari.c:169-173,214-217): L zero bits, then the L+1 significant bits, MSB first
 *   - a value repeated more than twice is followed, when the run ends, by the gamma code of
 *     (repeats - 2) + 1; a run counter reaching 65534 is flushed early and restarts
 *     (kagari.c:34,194-198,254-283)
 *
 * The encoder below is organised around maximal RUNS of equal values rather than the reference's
 * per-value state machine; the bit-stream and the "does it still fit" failure rule (capacity is
 * checked whenever whole bytes leave the 64 bit accumulator: kagari.c:64-78,91-112) are the same.
 */
#include "ako_host.h"

#include <pthread.h>
#include <stdio.h>
#include <time.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define RUN_LIMIT 65534u /* AKO_ELIAS_MAX - 1 */

struct bit_sink
{
	uint64_t acc;
	int pending; /* valid low bits of acc */
	uint8_t* at;
	const uint8_t* limit;
};

static inline int gamma_bits(uint16_t v)
{
	int extra = 0;
	while (v > 1)
	{
		v >>= 1;
		extra++;
	}
	return 2 * extra + 1;
}

/* returns 0 when the output is exhausted */
static inline int sink_put(struct bit_sink* s, uint16_t code)
{
	const int bits = gamma_bits(code);
	if (s->pending > 8 && s->pending + bits > 64)
	{
		if (s->at + (s->pending / 8) >= s->limit)
			return 0;
		while (s->pending + bits > 64)
		{
			s->pending -= 8;
			*s->at++ = (uint8_t)(s->acc >> s->pending);
		}
	}
	s->acc = (s->acc << bits) | code;
	s->pending += bits;
	return 1;
}

static inline uint16_t value_code(int16_t c)
{
	const uint16_t zz = (uint16_t)(((uint32_t)(int32_t)c << 1) ^ (uint32_t)((int32_t)c >> 15));
	return (uint16_t)(zz + 1);
}

size_t akoHostKagariEncode(size_t input_bytes, size_t capacity, const void* input, void* output)
{
	if (capacity == 0 || input_bytes == 0 || (input_bytes & 1) != 0)
		return 0;

	const int16_t* in = input;
	const size_t n = input_bytes / 2;
	struct bit_sink s = {0, 0, output, (const uint8_t*)output + capacity};

	size_t i = 0;
	while (i < n)
	{
		const int16_t v = in[i++];
		const uint16_t code = value_code(v);
		if (!sink_put(&s, code))
			return 0;

		/* repeats of v that follow */
		size_t repeats = 0;
		while (i + repeats < n && in[i + repeats] == v)
			repeats++;
		i += repeats;

		while (repeats != 0)
		{
			const size_t chunk = repeats < RUN_LIMIT ? repeats : RUN_LIMIT;
			const size_t literal = chunk < 2 ? chunk : 2;
			for (size_t k = 0; k < literal; k++)
				if (!sink_put(&s, code))
					return 0;
			if (chunk >= 2 && !sink_put(&s, (uint16_t)(chunk - 2 + 1)))
				return 0;
			repeats -= chunk;
		}
	}

	/* drain: whole bytes, then the zero padded tail (kagari.c:91-112) */
	while (s.pending >= 8)
	{
		if (s.at + 1 >= s.limit)
			return 0;
		s.pending -= 8;
		*s.at++ = (uint8_t)(s.acc >> s.pending);
	}
	if (s.pending != 0)
	{
		if (s.at + 1 >= s.limit)
			return 0;
		*s.at++ = (uint8_t)(s.acc << (8 - s.pending));
	}
	return (size_t)(s.at - (uint8_t*)output);
}

/* ---- decoder ------------------------------------------------------------------------------- */

/* The reference's reader (kagari.c:113-163) refills a 64 bit accumulator EAGERLY, whole bytes at a time, and
 * reports the bytes it has FETCHED as the size it consumed -- which compression.c:69 then compares with the
 * block size.  For a stream the encoder wrote the two always agree; for a damaged one the verdict depends on
 * the fetch pattern, so the pattern is part of the format's observable behaviour and is kept as is:
 * refill when fewer than 32 bits are held (or the accumulator is all zero), up to at least 56 bits. */
struct bit_source
{
	uint64_t acc;
	int held; /* valid high bits of acc */
	const uint8_t* at;
	const uint8_t* end;
	const uint8_t* base;
	int saw_long; /* codes of more than 31 bits that came by: damaged input.  No value the ENCODER can write has one: the longest,
	               * 31 bits, belong to zigzag + 1 = 32768 .. 65535; -32768 would need 65536, which the reference's uint16_t code
	               * word (kagari.c:172-178) cannot hold -- it has no code at all, and a block with it fails to encode.  Whether
	               * an over-long code fits what is 'held' depends on the refill history, which a parser that joined the
	               * stream halfway does not share with the sequential reader -- the parallel tokenizer hands such blocks back */
};

/* Next gamma code, low 16 bits (kagari.c keeps it in a uint16_t, so an over-long code simply wraps).
 * Returns 0 when the stream ends or breaks, 1 otherwise. */
static inline int source_get(struct bit_source* s, uint16_t* value)
{
	if (s->acc == 0 || s->held < 32)
	{
		if (s->at + ((64 - s->held) / 8) < s->end)
		{
			do
			{
				s->held += 8;
				s->acc |= (uint64_t)(*s->at++) << ((64 - s->held) & 63);
			} while (s->held < 56);
		}
		else
		{
			while (s->held < 56 && s->at < s->end)
			{
				s->held += 8;
				s->acc |= (uint64_t)(*s->at++) << (64 - s->held);
			}
		}
		if (s->acc == 0)
			return 0;
	}
	const uint32_t top = (uint32_t)(s->acc >> 32);
	const int zeros = top ? __builtin_clz(top) : 32;
	const int bits = 2 * zeros + 1;
	if (bits > 31)
		s->saw_long++;
	if (bits > s->held)
		return 0;
	*value = (uint16_t)(s->acc >> (64 - bits));
	s->acc = (bits < 64) ? (s->acc << bits) : 0;
	s->held -= bits;
	return 1;
}

static inline size_t source_used(const struct bit_source* s)
{
	return (size_t)(s->at - s->base);
}

size_t akoHostKagariDecode(size_t values_no, size_t input_bytes, size_t output_bytes, const void* input, void* output)
{
	if (output_bytes == 0 || input_bytes == 0 || values_no == 0 || (output_bytes & 1) != 0)
		return 0;
	if (values_no > output_bytes / 2)
		return 0;

	struct bit_source s = {0, 0, input, (const uint8_t*)input + input_bytes, input, 0};
	int16_t* out = output;
	size_t done = 0;
	int16_t prev = 0;
	unsigned same = 0; /* repeats of prev seen since the last run code */

	while (done < values_no)
	{
		uint16_t code;
		if (!source_get(&s, &code))
			return 0;
		const uint16_t zz = (uint16_t)(code - 1);
		const int16_t v = (int16_t)((zz >> 1) ^ (uint16_t)(~(zz & 1) + 1)); /* kagari.c:175-178 */
		out[done++] = v;

		if (done > 1 && v == prev)
		{
			if (++same == 2)
			{
				uint16_t run;
				if (!source_get(&s, &run))
					return 0;
				const size_t extra = (uint16_t)(run - 1);
				/* the reference lets a run overshoot the value count (its counter wraps, kagari.c:345-353) and
				 * then fails further on, at the latest at the end of its output buffer: same verdict here */
				if (extra > values_no - done)
					return 0;
				for (size_t k = 0; k < extra; k++)
					out[done + k] = prev;
				done += extra;
				same = 0;
			}
		}
		else
		{
			prev = v;
			same = 0;
		}
	}
	return source_used(&s);
}


/* ---- tokenizer: the decoder's parse without the run expansion ------------------------------------
 * For the device route of the decoder (akoHipKagariExpand): the bit-stream is walked exactly as
 * akoHostKagariDecode() walks it, but repeated values are not written out -- each run becomes a record
 * (where it starts in the output, how long, after how many literal values) and the GPU expands them
 * straight into the coefficient stream.  Host work and the host -> device copy then scale with the
 * COMPRESSED size.  Output positions are global over the image's stream (tiles are contiguous). */

static int tokens_reserve(struct akoKagariTokens* tok, size_t literals, size_t runs)
{
	if (tok->n_literals + literals > tok->cap_literals)
	{
		size_t cap = tok->cap_literals ? tok->cap_literals * 2 : 256;
		while (cap < tok->n_literals + literals)
			cap *= 2;
		int16_t* p = realloc(tok->literals, cap * sizeof(int16_t));
		if (p == NULL)
			return 0;
		tok->literals = p, tok->cap_literals = cap;
	}
	if (tok->n_runs + runs > tok->cap_runs)
	{
		size_t cap = tok->cap_runs ? tok->cap_runs * 2 : 64;
		while (cap < tok->n_runs + runs)
			cap *= 2;
		struct akoKagariRun* p = realloc(tok->runs, cap * sizeof(struct akoKagariRun));
		if (p == NULL)
			return 0;
		tok->runs = p, tok->cap_runs = cap;
	}
	return 1;
}

/* ---- the same parse by several threads ----------------------------------------------------------------------
 * A block can only be walked from its first bit -- but gamma codes re-synchronise: a reader dropped at an arbitrary
 * byte soon falls onto the true code boundaries, and once it has also seen a value that differs from the one
 * before it, its state (previous value, repeats seen) is the true one.  So the block is cut into byte ranges, one
 * per thread:
 *   1. every thread but the first starts SPECULATIVELY at its range's first byte with an unknown previous value and
 *      notes the state in front of each of its first PT_TRAIL codes (bit position, previous value, repeats, how much
 *      it has produced so far);
 *   2. every thread parses to the end of its range and on into the next one until it stands on a bit position its
 *      right neighbour noted with the SAME state: from there on the neighbour's output is what the sequential
 *      reader would have produced, and the thread stops.  The last thread reads to the end of the block;
 *   3. what a thread produced before the point its left neighbour joined it is dropped, the rest is moved into the
 *      caller's token lists (positions and literal counts shifted by what the threads to its left contributed).
 * The result is accepted only if every join happened, no thread met a broken code behind its join point, the
 * values add up to exactly 'values_no' and the last code ends in the block's last byte -- then the sequential
 * reader would have fetched exactly 'input_bytes' as well.  In EVERY other case (damaged or padded blocks, no
 * re-synchronisation inside the trail, no threads) the block is parsed again by akoHostKagariTokenize's sequential
 * loop, whose verdict and fetch count are the reference's (kagari.c:113-163, compression.c:58-73). */

#define PT_MAX_THREADS 16
#define PT_TRAIL 16384
#define PT_UNKNOWN INT32_MIN

struct pt_mark
{
	uint64_t bit, done;
	int32_t prev;
	uint32_t same;
	size_t n_lit, n_runs;
	int long_codes; /* over-long codes seen so far (bit_source.saw_long) */
};

struct pt_sync
{
	pthread_mutex_t m;
	pthread_cond_t c;
	int parties, waiting, abort;
	unsigned phase;
};

struct pt_worker
{
	struct pt_sync* sync;
	struct pt_worker* right; /* NULL: last range */
	int first;
	const uint8_t* input;
	size_t input_bytes, begin, end;
	/* parser state (positions relative to the worker's start) */
	struct bit_source s;
	int32_t prev;
	uint32_t same;
	uint64_t done;
	struct akoKagariTokens tok;
	struct pt_mark* marks;
	size_t n_marks;
	uint64_t last_skip_bit; /* speculative phase only: end of the last stretch that could not be a code */
	int trail_failed;       /* ... its last step failed with no byte left to restart on */
	/* outcome */
	int ok;
	size_t join_mark;  /* index into right->marks where this worker stopped */
	uint64_t end_bit;  /* last range: bits consumed when the block ran out */
	/* phase 3 */
	int go;
	size_t lit0, run0, lit_base, run_base; /* first kept literal / run, where they go in dst */
	uint64_t done0, done_base;
	struct akoKagariTokens* dst;
	size_t dst_lit0; /* dst->n_literals before the call */
} __attribute__((aligned(256))); /* a worker's counters change with every code: no two workers in one cache line */

static inline uint64_t pt_bit(const struct bit_source* s)
{
	return (uint64_t)(s->at - s->base) * 8 - (uint64_t)s->held;
}

/* one value (and the run code behind it when it is the third of its kind): 1 done, 0 no further code, -1 broken */
static inline int pt_step(struct pt_worker* w)
{
	uint16_t code;
	if (!source_get(&w->s, &code))
		return 0;
	const uint16_t zz = (uint16_t)(code - 1);
	const int16_t v = (int16_t)((zz >> 1) ^ (uint16_t)(~(zz & 1) + 1));
	if (w->tok.n_literals == w->tok.cap_literals && !tokens_reserve(&w->tok, 1, 0))
		return -1;
	w->tok.literals[w->tok.n_literals++] = v;
	w->done++;
	if ((int32_t)v == w->prev)
	{
		if (++w->same == 2)
		{
			uint16_t run;
			if (!source_get(&w->s, &run))
				return -1;
			const uint32_t extra = (uint16_t)(run - 1);
			if (extra != 0)
			{
				if (w->tok.n_runs == w->tok.cap_runs && !tokens_reserve(&w->tok, 0, 1))
					return -1;
				struct akoKagariRun* r = &w->tok.runs[w->tok.n_runs++];
				r->out_start = (uint32_t)w->done; /* relative; fits: done < values_no + 65536 is checked at the merge */
				r->count = extra;
				r->after = (uint32_t)w->tok.n_literals;
				r->pad = 0;
				w->done += extra;
			}
			w->same = 0;
		}
	}
	else
	{
		w->prev = v;
		w->same = 0;
	}
	return 1;
}

/* all parties meet; returns 0 when the run was called off */
static int pt_meet(struct pt_sync* y)
{
	pthread_mutex_lock(&y->m);
	const unsigned phase = y->phase;
	if (++y->waiting == y->parties)
	{
		y->waiting = 0;
		y->phase++;
		pthread_cond_broadcast(&y->c);
	}
	else
		while (y->phase == phase && !y->abort)
			pthread_cond_wait(&y->c, &y->m);
	const int ok = !y->abort;
	pthread_mutex_unlock(&y->m);
	return ok;
}

static void pt_parse(struct pt_worker* w)
{
	w->ok = 0;
	if (w->trail_failed)
		return;
	const uint64_t end_bit_of_range = (uint64_t)w->end * 8;
	for (;;)
	{
		if (w->right != NULL && pt_bit(&w->s) >= end_bit_of_range)
			break;
		const uint64_t before = pt_bit(&w->s);
		const int r = pt_step(w);
		if (r < 0)
			return;
		if (r == 0)
		{
			if (w->right != NULL)
				return; /* the block ran out (or broke) inside a range that is not the last one */
			w->end_bit = before;
			w->ok = 1;
			return;
		}
	}
	/* in the right neighbour's range: walk on until standing on one of its marks in the same state */
	const struct pt_mark* marks = w->right->marks;
	const size_t n = w->right->n_marks;
	size_t j = 0;
	for (;;)
	{
		const uint64_t p = pt_bit(&w->s);
		while (j < n && marks[j].bit < p)
			j++;
		if (j == n)
			return;
		if (marks[j].bit == p && marks[j].prev == w->prev && marks[j].same == w->same && p >= w->right->last_skip_bit)
		{
			w->join_mark = j;
			w->ok = 1;
			return;
		}
		if (pt_step(w) <= 0)
			return;
	}
}

static void pt_move(struct pt_worker* w)
{
	struct akoKagariTokens* dst = w->dst;
	const size_t lits = w->tok.n_literals - w->lit0, runs = w->tok.n_runs - w->run0;
	if (lits)
		memcpy(dst->literals + w->dst_lit0 + w->lit_base, w->tok.literals + w->lit0, lits * sizeof(int16_t));
	struct akoKagariRun* out = dst->runs + w->run_base;
	for (size_t k = 0; k < runs; k++)
	{
		struct akoKagariRun r = w->tok.runs[w->run0 + k];
		/* the noted positions are the low 32 bits of counters that ran through a speculative stretch first */
		r.out_start = (uint32_t)w->done_base + (uint32_t)(r.out_start - (uint32_t)w->done0);
		r.after = (uint32_t)(w->dst_lit0 + w->lit_base) + (uint32_t)(r.after - (uint32_t)w->lit0);
		out[k] = r;
	}
}

static void* pt_main(void* arg)
{
	struct pt_worker* w = arg;
	if (!pt_meet(w->sync)) /* start: every thread exists */
		return NULL;
	if (!w->first)
	{
		/* the trail: states in front of the first PT_TRAIL codes, stretches that cannot be codes skipped bytewise */
		while (w->n_marks < PT_TRAIL)
		{
			struct pt_mark* m = &w->marks[w->n_marks++];
			m->bit = pt_bit(&w->s), m->done = w->done, m->prev = w->prev, m->same = w->same;
			m->n_lit = w->tok.n_literals, m->n_runs = w->tok.n_runs, m->long_codes = w->s.saw_long;
			const int r = pt_step(w);
			if (r > 0)
				continue;
			const size_t next_byte = (size_t)(m->bit / 8) + 1;
			if (next_byte >= w->input_bytes)
			{
				/* r == 0: the block ran out, the ordinary end of the last range.  r < 0: a code BROKE in the block's last
				 * byte (a third equal value whose run code is missing) and nothing is left to restart on -- the
				 * sequential reader calls such a block broken, and that verdict is not ours to give */
				if (r < 0)
					w->trail_failed = 1;
				break;
			}
			const struct bit_source restart = {0, 0, w->input + next_byte, w->input + w->input_bytes, w->input, 0};
			w->s = restart;
			w->prev = PT_UNKNOWN, w->same = 0;
			w->last_skip_bit = (uint64_t)next_byte * 8;
		}
	}
	if (!pt_meet(w->sync)) /* every trail is written */
		return NULL;
	pt_parse(w);
	if (!pt_meet(w->sync)) /* every range is parsed; the first thread decides */
		return NULL;
	if (!pt_meet(w->sync)) /* ... and has decided */
		return NULL;
	if (w->go)
		pt_move(w);
	return NULL;
}

static size_t pt_threads(size_t input_bytes)
{
	/* AKO_KAGARI_THREADS: threads to use (0 / 1 = sequential parse only; default: the cores, 16 at most);
	 * AKO_KAGARI_PAR_MIN: smallest block in bytes parsed in parallel (default 128 KiB; ranges are a quarter of it at least) */
	if (input_bytes < 64)
		return 1;
	size_t min_bytes = 128u << 10;
	const char* mb = getenv("AKO_KAGARI_PAR_MIN");
	if (mb != NULL && mb[0] != 0)
		min_bytes = (size_t)atol(mb);
	if (input_bytes < min_bytes)
		return 1;
	const char* e = getenv("AKO_KAGARI_THREADS");
	const long want = (e != NULL && e[0] != 0) ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
	const int configured = (int)(want < 1 ? 1 : (want > PT_MAX_THREADS ? PT_MAX_THREADS : want));
	if (configured < 2 || input_bytes < min_bytes || input_bytes < 64)
		return 1;
	size_t n = input_bytes / (min_bytes / 4 ? min_bytes / 4 : 1); /* ranges of a quarter of the threshold at least */
	if (n > (size_t)configured)
		n = (size_t)configured;
	if (n > input_bytes / 16)
		n = input_bytes / 16;
	return n < 2 ? 1 : n;
}

/* 1: tok extended, block fully and validly parsed; 0: nothing changed, parse it sequentially */
static int tokenize_in_parallel(size_t threads, size_t values_no, size_t input_bytes, const uint8_t* input, uint64_t out_base,
                                struct akoKagariTokens* tok)
{
	struct pt_sync sync;
	struct pt_worker* w = aligned_alloc(256, threads * sizeof *w);
	pthread_t th[PT_MAX_THREADS];
	if (w == NULL)
		return 0;
	memset(w, 0, threads * sizeof *w);
	pthread_mutex_init(&sync.m, NULL);
	pthread_cond_init(&sync.c, NULL);
	sync.parties = (int)threads, sync.waiting = 0, sync.abort = 0, sync.phase = 0;

	int ready = 1;
	for (size_t k = 0; k < threads; k++)
	{
		w[k].sync = &sync;
		w[k].right = (k + 1 < threads) ? &w[k + 1] : NULL;
		w[k].first = (k == 0);
		w[k].input = input, w[k].input_bytes = input_bytes;
		w[k].begin = input_bytes * k / threads, w[k].end = input_bytes * (k + 1) / threads;
		const struct bit_source start = {0, 0, input + w[k].begin, input + input_bytes, input, 0};
		w[k].s = start;
		w[k].prev = PT_UNKNOWN;
		w[k].dst = tok, w[k].dst_lit0 = tok->n_literals;
		if (k != 0 && (w[k].marks = malloc(PT_TRAIL * sizeof(struct pt_mark))) == NULL)
			ready = 0;
		/* a couple of literals per byte in practice; the lists grow on demand anyway */
		if (!tokens_reserve(&w[k].tok, (w[k].end - w[k].begin) * 2 + 1024, (w[k].end - w[k].begin) / 4 + 1024))
			ready = 0;
	}
	size_t started = 0;
	if (ready)
		for (size_t k = 1; k < threads; k++)
		{
			if (pthread_create(&th[started], NULL, pt_main, &w[k]) != 0)
				break;
			started++;
		}
	int accepted = 0;
	if (!ready || started + 1 != threads)
	{
		pthread_mutex_lock(&sync.m);
		sync.abort = 1;
		pthread_cond_broadcast(&sync.c);
		pthread_mutex_unlock(&sync.m);
	}
	else
	{
		/* the calling thread is the first worker; between the third and the fourth meeting it merges */
		struct pt_worker* me = &w[0];
		struct timespec t0, t1, t2, t3;
		clock_gettime(CLOCK_MONOTONIC, &t0);
		pt_meet(&sync);
		pt_meet(&sync);
		clock_gettime(CLOCK_MONOTONIC, &t1);
		pt_parse(me);
		clock_gettime(CLOCK_MONOTONIC, &t2);
		pt_meet(&sync);
		clock_gettime(CLOCK_MONOTONIC, &t3);
		if (getenv("AKO_KAGARI_TRACE"))
		{
			fprintf(stderr, "pt: trail %.2f ms, my parse %.2f ms, wait for the others %.2f ms; joins:", (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) / 1e6,
			        (t2.tv_sec - t1.tv_sec) * 1e3 + (t2.tv_nsec - t1.tv_nsec) / 1e6, (t3.tv_sec - t2.tv_sec) * 1e3 + (t3.tv_nsec - t2.tv_nsec) / 1e6);
			for (size_t k = 0; k < threads; k++)
				fprintf(stderr, " [%d %zu lit %zu]", w[k].ok, w[k].join_mark, w[k].tok.n_literals);
			fprintf(stderr, "\n");
		}

		int good = 1;
		uint64_t total = 0;
		size_t lits = 0, runs = 0;
		for (size_t k = 0; k < threads && good; k++)
		{
			/* (an over-long code in what a worker KEEPS -- everything behind the mark its left neighbour joined at)
			 * sends the block back; the speculative stretch in front of that mark is dropped anyway) */
			if (!w[k].ok || w[k].s.saw_long != (k != 0 ? w[k].marks[w[k - 1].join_mark].long_codes : 0))
				good = 0;
			else
			{
				if (k != 0)
				{
					const struct pt_mark* m = &w[k].marks[w[k - 1].join_mark];
					w[k].lit0 = m->n_lit, w[k].run0 = m->n_runs, w[k].done0 = m->done;
				}
				w[k].lit_base = lits, w[k].run_base = tok->n_runs + runs, w[k].done_base = out_base + total;
				lits += w[k].tok.n_literals - w[k].lit0;
				runs += w[k].tok.n_runs - w[k].run0;
				total += w[k].done - w[k].done0;
				if (total > values_no)
					good = 0;
			}
		}
		if (good && (total != values_no || (w[threads - 1].end_bit + 7) / 8 != input_bytes))
			good = 0;
		if (good && (tok->n_literals + lits > 0xFFFFFFF0ull || !tokens_reserve(tok, lits, runs)))
			good = 0;
		if (good)
		{
			for (size_t k = 0; k < threads; k++)
				w[k].go = 1;
			accepted = 1;
		}
		pt_meet(&sync);
		if (accepted)
		{
			pt_move(me);
			tok->n_literals += lits, tok->n_runs += runs;
		}
	}
	for (size_t k = 0; k < started; k++)
		pthread_join(th[k], NULL);
	for (size_t k = 0; k < threads; k++)
	{
		free(w[k].marks);
		akoHostKagariTokensFree(&w[k].tok);
	}
	free(w);
	pthread_mutex_destroy(&sync.m);
	pthread_cond_destroy(&sync.c);
	return accepted;
}

/* blocks the threads parsed / blocks they gave back to the sequential loop, since the library was loaded */
static size_t parallel_blocks[2];
void akoHostKagariParallelStats(size_t* accepted, size_t* handed_back)
{
	*accepted = __atomic_load_n(&parallel_blocks[0], __ATOMIC_RELAXED);
	*handed_back = __atomic_load_n(&parallel_blocks[1], __ATOMIC_RELAXED);
}

size_t akoHostKagariTokenize(size_t values_no, size_t input_bytes, const void* input, uint64_t out_base,
                             struct akoKagariTokens* tok)
{
	return akoHostKagariTokenizeWith(0, values_no, input_bytes, input, out_base, tok);
}

/* a caller that is itself one of several threads doing the same (batch lanes) keeps its parses to itself */
static __thread size_t thread_limit = 0;
void akoHostKagariThreadLimit(size_t max_threads)
{
	thread_limit = max_threads;
}
size_t akoHostKagariThreadLimitGet(void)
{
	return thread_limit;
}

/* max_threads: 0 = as many as pay (pt_threads) within the calling thread's limit, 1 = the sequential loop only */
size_t akoHostKagariTokenizeWith(size_t max_threads, size_t values_no, size_t input_bytes, const void* input,
                                 uint64_t out_base, struct akoKagariTokens* tok)
{
	if (input_bytes == 0 || values_no == 0 || out_base + values_no > 0xFFFFFFF0ull)
		return 0;

	if (max_threads == 0)
		max_threads = thread_limit;
	size_t threads = (max_threads == 1) ? 1 : pt_threads(input_bytes);
	if (max_threads != 0 && threads > max_threads)
		threads = max_threads;
	if (threads > 1)
	{
		const int accepted = tokenize_in_parallel(threads, values_no, input_bytes, input, out_base, tok);
		__atomic_fetch_add(&parallel_blocks[accepted ? 0 : 1], 1, __ATOMIC_RELAXED);
		if (accepted)
			return input_bytes;
	}

	struct bit_source s = {0, 0, input, (const uint8_t*)input + input_bytes, input, 0};
	size_t done = 0;
	int16_t prev = 0;
	unsigned same = 0;

	while (done < values_no)
	{
		uint16_t code;
		if (!source_get(&s, &code))
			return 0;
		const uint16_t zz = (uint16_t)(code - 1);
		const int16_t v = (int16_t)((zz >> 1) ^ (uint16_t)(~(zz & 1) + 1));
		if (tok->n_literals == tok->cap_literals && !tokens_reserve(tok, 1, 0))
			return 0;
		tok->literals[tok->n_literals++] = v;
		done++;

		if (done > 1 && v == prev)
		{
			if (++same == 2)
			{
				uint16_t run;
				if (!source_get(&s, &run))
					return 0;
				const size_t extra = (uint16_t)(run - 1);
				if (extra > values_no - done)
					return 0;
				if (extra != 0)
				{
					if (tok->n_runs == tok->cap_runs && !tokens_reserve(tok, 0, 1))
						return 0;
					struct akoKagariRun* r = &tok->runs[tok->n_runs++];
					r->out_start = (uint32_t)(out_base + done);
					r->count = (uint32_t)extra;
					r->after = (uint32_t)tok->n_literals;
					r->pad = 0;
					done += extra;
				}
				same = 0;
			}
		}
		else
		{
			prev = v;
			same = 0;
		}
	}
	return source_used(&s);
}

int akoHostKagariTokensReserve(struct akoKagariTokens* tok, size_t literals, size_t runs)
{
	return tokens_reserve(tok, literals, runs);
}

/* dst += src, with src's literal counts shifted by 'literal_base' (= dst->n_literals before the call) */
int akoHostKagariTokensAppend(struct akoKagariTokens* dst, const struct akoKagariTokens* src, uint32_t literal_base)
{
	if (!tokens_reserve(dst, src->n_literals, src->n_runs))
		return 0;
	if (src->n_literals)
		memcpy(dst->literals + dst->n_literals, src->literals, src->n_literals * sizeof(int16_t));
	for (size_t k = 0; k < src->n_runs; k++)
	{
		struct akoKagariRun r = src->runs[k];
		r.after += literal_base;
		dst->runs[dst->n_runs + k] = r;
	}
	dst->n_literals += src->n_literals;
	dst->n_runs += src->n_runs;
	return 1;
}

void akoHostKagariTokensFree(struct akoKagariTokens* tok)
{
	free(tok->literals);
	free(tok->runs);
	memset(tok, 0, sizeof *tok);
}
