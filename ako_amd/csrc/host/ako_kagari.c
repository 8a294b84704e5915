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

#include <stdlib.h>
#include <string.h>

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
	const uint16_t zz = (uint16_t)(((int)c << 1) ^ ((int)c >> 15));
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

	struct bit_source s = {0, 0, input, (const uint8_t*)input + input_bytes, input};
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

size_t akoHostKagariTokenize(size_t values_no, size_t input_bytes, const void* input, uint64_t out_base,
                             struct akoKagariTokens* tok)
{
	if (input_bytes == 0 || values_no == 0 || out_base + values_no > 0xFFFFFFF0ull)
		return 0;

	struct bit_source s = {0, 0, input, (const uint8_t*)input + input_bytes, input};
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
