// ako_kagari.hip.h -- Kagari entropy ENCODER on the GPU (reference: library/kagari.c:228-366,
// library/compression.c:36-55).  Included by ako_plan.hip only.
//
// Kagari codes every int16 coefficient as the Elias-gamma code of zigzag(c) + 1 and closes a run of more
// than two equal values with the gamma code of (repeats - 2) + 1, restarting at 65534 repeats (see
// host/ako_kagari.c, which is the host restatement that the tests compare this file against).  The
// bit-stream of a tile therefore is a concatenation of per-RUN bit strings whose lengths depend only on
// (value, run length): data parallel once the runs are known.
//
//   1. k_kg_starts<0> flag run starts (value differs from its predecessor, or first of a tile), count / block
//   2. k_kg_scan      one workgroup: exclusive scan of the block counts -> first run index of every block
//   3. k_kg_starts<1> write the start position of every run (compacted, in order)
//   4. k_kg_bits      bits of every run; per-block sums
//   5. k_kg_scan      scan of the block sums -> bit offset of every block of runs
//   6. k_kg_tiles     one wave per tile: bit offset of the tile's first run; k_kg_layout: sizes, the
//                     "did it shrink" rule, offsets of the tiles inside the blob body
//   7. k_kg_write     every run writes its codes at its bit offset (atomicOr on zeroed big-endian words)
//   8. k_kg_gather    [uint32 size][payload] per tile, contiguous: the blob body after the 16 byte head
//
// Tile streams are contiguous and in raster order inside an image's stream (ako_plan.hip, plan->tiles), a
// tile starts at a block boundary of step 1, and its first value always starts a run.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ako
{

constexpr int KG_THREADS = 256;
constexpr int KG_PER_THREAD = 8;
constexpr int KG_CHUNK = KG_THREADS * KG_PER_THREAD;  // values per workgroup in steps 1 and 3
constexpr uint32_t KG_RUN_LIMIT = 65534u;             // AKO_ELIAS_MAX - 1, kagari.c:34
constexpr int KG_SCAN_THREADS = 1024;

struct KgTile
{
	uint64_t off;          // first value of the tile, int16 units from the start of the image's stream
	uint64_t n;            // values
	uint32_t first_block;  // first workgroup of steps 1 / 3 that belongs to this tile
	uint32_t pad;
};

struct KgResult
{
	uint64_t body_bytes;   // 0 when a tile failed
	uint64_t total_bits;
	uint32_t failed_tile;  // first tile that did not shrink, or 0xFFFFFFFF
	uint32_t pad;
};

// ---- helpers -----------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t kg_value_code(int v)  // kagari.c:169-173: zigzag + 1, in 16 bits
{
	const uint32_t zz = (uint32_t)((v << 1) ^ (v >> 15)) & 0xFFFFu;
	return (zz + 1u) & 0xFFFFu;  // -32768 wraps to 0 like the reference's uint16_t arithmetic
}
__device__ __forceinline__ uint32_t kg_gamma_bits(uint32_t code)  // kagari.c:214-217
{
	return (code <= 1u) ? 1u : (uint32_t)(2 * (31 - __clz(code)) + 1);
}
// largest t with key(t) <= x; keys ascending, key(0) <= x
template <typename F>
__device__ __forceinline__ uint32_t kg_upper(uint32_t n, uint64_t x, F key)
{
	uint32_t lo = 0, hi = n;
	while (hi - lo > 1)
	{
		const uint32_t mid = (lo + hi) >> 1;
		if (key(mid) <= x)
			lo = mid;
		else
			hi = mid;
	}
	return lo;
}

// bits of one run (value code c of cb bits, 'len' equal values)
__device__ __forceinline__ uint64_t kg_run_bits(uint32_t cb, uint64_t len)
{
	uint64_t bits = cb;
	uint64_t repeats = len - 1;
	while (repeats != 0)
	{
		const uint32_t chunk = repeats < KG_RUN_LIMIT ? (uint32_t)repeats : KG_RUN_LIMIT;
		bits += (uint64_t)cb * (chunk < 2 ? chunk : 2);
		if (chunk >= 2)
			bits += kg_gamma_bits(chunk - 1);
		repeats -= chunk;
	}
	return bits;
}

// ---- 1 / 3: run starts ---------------------------------------------------------------------------

// value index handled by (thread, j) inside a chunk of the EXPAND kernel: consecutive threads, consecutive values
__device__ __forceinline__ uint32_t kg_slot(int j)
{
	return (uint32_t)j * KG_THREADS + threadIdx.x;
}

constexpr int KG_RUN_VALUES = 32;                          // consecutive values per thread in k_kg_starts
constexpr int KG_RUN_CHUNK = KG_THREADS * KG_RUN_VALUES;   // values per workgroup there (= KgTile::first_block unit)

// Every thread takes 32 consecutive values (four 16-byte loads; tile streams are only 2-byte aligned, which the
// global path tolerates) plus the one in front of them, and marks the values that differ from their
// predecessor.  Threads are in value order, so the rank of a run start is the exclusive scan of the per-thread
// counts plus the number of set bits below it.
template <bool EMIT>
__global__ __launch_bounds__(KG_THREADS) void k_kg_starts(const int16_t* __restrict__ stream, const KgTile* __restrict__ tiles,
                                                          uint32_t n_tiles, uint32_t* __restrict__ block_count,
                                                          const uint32_t* __restrict__ block_first_run,
                                                          uint32_t* __restrict__ run_start)
{
	__shared__ uint32_t wave_total[KG_THREADS / 64];
	const uint32_t b = blockIdx.x;
	const uint32_t t = kg_upper(n_tiles, b, [&](uint32_t k) { return (uint64_t)tiles[k].first_block; });
	const KgTile tile = tiles[t];
	const uint64_t i0 = (uint64_t)(b - tile.first_block) * KG_RUN_CHUNK + (uint64_t)threadIdx.x * KG_RUN_VALUES;
	const int16_t* s = stream + tile.off;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

	uint32_t mask = 0;
	if (i0 < tile.n)
	{
		int16_t v[KG_RUN_VALUES];
		const uint32_t have = (tile.n - i0 < KG_RUN_VALUES) ? (uint32_t)(tile.n - i0) : KG_RUN_VALUES;
		if (have == KG_RUN_VALUES)
		{
#pragma unroll
			for (int q = 0; q < KG_RUN_VALUES / 8; q++)
			{
				const uint4 u = *reinterpret_cast<const uint4*>(s + i0 + 8 * q);
				const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
				for (int j = 0; j < 4; j++)
					v[8 * q + 2 * j] = (int16_t)(w[j] & 0xFFFFu), v[8 * q + 2 * j + 1] = (int16_t)(w[j] >> 16);
			}
		}
		else
		{
#pragma unroll
			for (int j = 0; j < KG_RUN_VALUES; j++)
				v[j] = ((uint32_t)j < have) ? s[i0 + j] : (int16_t)0;
		}
		int16_t prev = (i0 == 0) ? (int16_t)0 : s[i0 - 1];
#pragma unroll
		for (int j = 0; j < KG_RUN_VALUES; j++)
		{
			const bool start = ((uint32_t)j < have) && ((i0 + j == 0) || (v[j] != prev));
			mask |= (start ? 1u : 0u) << j;
			prev = v[j];
		}
	}
	const uint32_t mine = (uint32_t)__popc(mask);

	// exclusive scan of 'mine' over the workgroup: inside the wave by shuffles, across the four waves through LDS
	uint32_t incl = mine;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1)
	{
		const uint32_t up = __shfl_up(incl, d, 64);
		if (lane >= d)
			incl += up;
	}
	if (lane == 63)
		wave_total[wave] = incl;
	__syncthreads();
	uint32_t before = 0, total = 0;
#pragma unroll
	for (int k = 0; k < KG_THREADS / 64; k++)
	{
		if (k < wave)
			before += wave_total[k];
		total += wave_total[k];
	}
	if (!EMIT)
	{
		if (threadIdx.x == 0)
			block_count[b] = total;
		return;
	}
	uint32_t rank = block_first_run[b] + before + incl - mine;
	const uint32_t at = (uint32_t)(tile.off + i0);
	while (mask)
	{
		const int j = __ffs(mask) - 1;
		mask &= mask - 1;
		run_start[rank++] = at + (uint32_t)j;
	}
}

// ---- 2 / 5: exclusive scan by ONE workgroup (n up to a few million) --------------------------------

template <typename T>
__global__ __launch_bounds__(KG_SCAN_THREADS) void k_kg_scan(const T* in, T* out, uint64_t n, T* total)  // in == out allowed
{
	__shared__ T sums[KG_SCAN_THREADS];
	const uint64_t per = (n + KG_SCAN_THREADS - 1) / KG_SCAN_THREADS;
	const uint64_t lo = per * threadIdx.x, hi = (lo + per < n) ? lo + per : n;
	T acc = 0;
	for (uint64_t i = lo; i < hi; i++)
		acc += in[i];
	sums[threadIdx.x] = acc;
	__syncthreads();
	for (int d = 1; d < KG_SCAN_THREADS; d <<= 1)  // Hillis-Steele over the 1024 partial sums
	{
		const T add = (threadIdx.x >= (unsigned)d) ? sums[threadIdx.x - d] : (T)0;
		__syncthreads();
		sums[threadIdx.x] += add;
		__syncthreads();
	}
	T run = sums[threadIdx.x] - acc;  // exclusive
	for (uint64_t i = lo; i < hi; i++)
	{
		const T v = in[i];
		out[i] = run;
		run += v;
	}
	if (threadIdx.x == KG_SCAN_THREADS - 1 && total)
		*total = sums[KG_SCAN_THREADS - 1];
}

// ---- 4: bits per run -----------------------------------------------------------------------------

__device__ __forceinline__ uint64_t kg_block_sum(uint64_t v, uint64_t* lds)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1)
		v += __shfl_down(v, d, 64);
	if ((threadIdx.x & 63) == 0)
		lds[threadIdx.x >> 6] = v;
	__syncthreads();
	uint64_t s = 0;
	for (int k = 0; k < KG_THREADS / 64; k++)
		s += lds[k];
	return s;
}

__global__ __launch_bounds__(KG_THREADS) void k_kg_bits(const int16_t* __restrict__ stream, const uint32_t* __restrict__ run_start,
                                                        uint32_t n_runs, uint64_t stream_values, uint32_t* __restrict__ run_bits,
                                                        uint64_t* __restrict__ block_bits)
{
	__shared__ uint64_t lds[KG_THREADS / 64];
	const uint32_t k = blockIdx.x * KG_THREADS + threadIdx.x;
	uint64_t bits = 0;
	if (k < n_runs)
	{
		const uint64_t from = run_start[k];
		const uint64_t to = (k + 1 < n_runs) ? (uint64_t)run_start[k + 1] : stream_values;  // tiles are contiguous
		bits = kg_run_bits(kg_gamma_bits(kg_value_code(stream[from])), to - from);
		run_bits[k] = (uint32_t)bits;
	}
	const uint64_t sum = kg_block_sum(bits, lds);
	if (threadIdx.x == 0)
		block_bits[blockIdx.x] = sum;
}

// ---- 6: per tile -----------------------------------------------------------------------------------

// one wave per tile: bit offset of the tile's first run
__global__ __launch_bounds__(64) void k_kg_tiles(const KgTile* __restrict__ tiles, uint32_t n_tiles,
                                                 const uint32_t* __restrict__ block_first_run, const uint32_t* __restrict__ run_bits,
                                                 const uint64_t* __restrict__ block_bit_off, uint32_t* __restrict__ tile_first_run,
                                                 uint64_t* __restrict__ tile_bit_off)
{
	const uint32_t t = blockIdx.x;
	if (t >= n_tiles)
		return;
	const uint32_t fr = block_first_run[tiles[t].first_block];
	const uint32_t blk = fr / KG_THREADS;
	uint64_t sum = 0;
	for (uint32_t k = blk * KG_THREADS + threadIdx.x; k < fr; k += 64)
		sum += run_bits[k];
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1)
		sum += __shfl_down(sum, d, 64);
	if (threadIdx.x == 0)
	{
		tile_first_run[t] = fr;
		tile_bit_off[t] = block_bit_off[blk] + sum;
	}
}

// one workgroup: payload size of every tile, the reference's failure rule, offsets inside the body
__global__ __launch_bounds__(KG_SCAN_THREADS) void k_kg_layout(const KgTile* __restrict__ tiles, uint32_t n_tiles,
                                                               const uint64_t* __restrict__ tile_bit_off,
                                                               const uint64_t* __restrict__ total_bits,
                                                               uint64_t* __restrict__ tile_payload, uint64_t* __restrict__ tile_dst,
                                                               KgResult* __restrict__ result)
{
	__shared__ uint64_t sums[KG_SCAN_THREADS];
	__shared__ uint32_t failed;
	if (threadIdx.x == 0)
		failed = 0xFFFFFFFFu;
	__syncthreads();
	const uint32_t per = (n_tiles + KG_SCAN_THREADS - 1) / KG_SCAN_THREADS;
	const uint32_t lo = per * threadIdx.x, hi = (lo + per < n_tiles) ? lo + per : n_tiles;
	uint64_t acc = 0;
	for (uint32_t t = lo; t < hi; t++)
	{
		const uint64_t end = (t + 1 < n_tiles) ? tile_bit_off[t + 1] : *total_bits;
		const uint64_t payload = (end - tile_bit_off[t] + 7) >> 3;
		// compression.c:36-43 hands Kagari a buffer of (tile bytes - 4) and kagari.c:64-112 never lets the
		// last byte of it be written: the tile "did not shrink" unless payload <= tile bytes - 5
		if (payload + 5 > tiles[t].n * 2)
			atomicMin(&failed, t);
		tile_payload[t] = payload;
		acc += payload + 4;
	}
	sums[threadIdx.x] = acc;
	__syncthreads();
	for (int d = 1; d < KG_SCAN_THREADS; d <<= 1)
	{
		const uint64_t add = (threadIdx.x >= (unsigned)d) ? sums[threadIdx.x - d] : 0;
		__syncthreads();
		sums[threadIdx.x] += add;
		__syncthreads();
	}
	uint64_t run = sums[threadIdx.x] - acc;
	for (uint32_t t = lo; t < hi; t++)
	{
		tile_dst[t] = run;
		run += tile_payload[t] + 4;
	}
	if (threadIdx.x == 0)
	{
		result->failed_tile = failed;
		result->total_bits = *total_bits;
		result->body_bytes = (failed == 0xFFFFFFFFu) ? sums[KG_SCAN_THREADS - 1] : 0;
	}
}

// ---- 7: bits out ---------------------------------------------------------------------------------

// OR 'bits' (<= 32) low bits of 'code' into a zeroed bit-stream at bit position 'at' (MSB first, as
// kagari.c:64-112 shifts them out); the stream is addressed as big-endian 32 bit words
__device__ __forceinline__ void kg_put(uint32_t* words, uint64_t at, uint32_t code, uint32_t bits)
{
	const uint64_t window = (uint64_t)code << (64 - bits - (uint32_t)(at & 31));
	const uint32_t hi = (uint32_t)(window >> 32), lo = (uint32_t)window;
	if (hi)
		atomicOr(&words[at >> 5], __builtin_bswap32(hi));
	if (lo)
		atomicOr(&words[(at >> 5) + 1], __builtin_bswap32(lo));
}

__global__ __launch_bounds__(KG_THREADS) void k_kg_write(const int16_t* __restrict__ stream, const uint32_t* __restrict__ run_start,
                                                         const uint32_t* __restrict__ run_bits, uint32_t n_runs,
                                                         uint64_t stream_values, const uint64_t* __restrict__ block_bit_off,
                                                         const uint32_t* __restrict__ tile_first_run,
                                                         const uint64_t* __restrict__ tile_bit_off,
                                                         const uint64_t* __restrict__ tile_stage, uint32_t n_tiles,
                                                         const KgResult* __restrict__ result, uint8_t* __restrict__ stage)
{
	__shared__ uint64_t scan[KG_THREADS];
	if (result->failed_tile != 0xFFFFFFFFu)
		return;  // a tile that does not shrink could overrun its staging area; the call fails anyway
	const uint32_t k = blockIdx.x * KG_THREADS + threadIdx.x;
	const uint64_t mine = (k < n_runs) ? run_bits[k] : 0;
	scan[threadIdx.x] = mine;
	__syncthreads();
	for (int d = 1; d < KG_THREADS; d <<= 1)
	{
		const uint64_t add = (threadIdx.x >= (unsigned)d) ? scan[threadIdx.x - d] : 0;
		__syncthreads();
		scan[threadIdx.x] += add;
		__syncthreads();
	}
	if (k >= n_runs)
		return;
	uint64_t at = block_bit_off[blockIdx.x] + scan[threadIdx.x] - mine;

	const uint32_t t = kg_upper(n_tiles, k, [&](uint32_t i) { return (uint64_t)tile_first_run[i]; });
	at -= tile_bit_off[t];
	uint32_t* words = reinterpret_cast<uint32_t*>(stage + tile_stage[t]);

	const uint64_t from = run_start[k];
	const uint64_t to = (k + 1 < n_runs) ? (uint64_t)run_start[k + 1] : stream_values;
	const uint32_t code = kg_value_code(stream[from]);
	const uint32_t cb = kg_gamma_bits(code);
	kg_put(words, at, code, cb), at += cb;
	uint64_t repeats = to - from - 1;
	while (repeats != 0)
	{
		const uint32_t chunk = repeats < KG_RUN_LIMIT ? (uint32_t)repeats : KG_RUN_LIMIT;
		kg_put(words, at, code, cb), at += cb;
		if (chunk >= 2)
		{
			kg_put(words, at, code, cb), at += cb;
			const uint32_t rc = chunk - 1, rb = kg_gamma_bits(rc);
			kg_put(words, at, rc, rb), at += rb;
		}
		repeats -= chunk;
	}
}

// ---- 8: blob body --------------------------------------------------------------------------------

constexpr int KG_GATHER_BYTES = 4096;  // body bytes per workgroup

__global__ __launch_bounds__(KG_THREADS) void k_kg_gather(const uint8_t* __restrict__ stage, const uint64_t* __restrict__ tile_stage,
                                                          const uint64_t* __restrict__ tile_payload,
                                                          const uint64_t* __restrict__ tile_dst, uint32_t n_tiles,
                                                          uint64_t body_bytes, uint8_t* __restrict__ body)
{
	const uint64_t lo = (uint64_t)blockIdx.x * KG_GATHER_BYTES;
	for (uint32_t i = threadIdx.x; i < KG_GATHER_BYTES; i += KG_THREADS)
	{
		const uint64_t at = lo + i;
		if (at >= body_bytes)
			return;
		const uint32_t t = kg_upper(n_tiles, at, [&](uint32_t k) { return tile_dst[k]; });
		const uint64_t rel = at - tile_dst[t];
		uint8_t v;
		if (rel < 4)
			v = (uint8_t)((uint32_t)tile_payload[t] >> (8 * rel));  // little-endian block size, compression.c:30-33,52
		else
			v = stage[tile_stage[t] + rel - 4];
		body[at] = v;
	}
}

// ---- decoder side: expand (literals, runs) into the coefficient stream -----------------------------

struct KgRun  // = struct akoHipKagariRun
{
	uint32_t out_start, count, after, pad;
};

__global__ __launch_bounds__(KG_THREADS) void k_kg_expand(const int16_t* __restrict__ literals, const KgRun* __restrict__ runs,
                                                          uint32_t n_runs, uint64_t n_out, int16_t* __restrict__ out)
{
	__shared__ uint32_t range[2];
	const uint64_t base = (uint64_t)blockIdx.x * KG_CHUNK;
	if (threadIdx.x < 2)  // runs that can matter to this block's outputs: [range[0], range[1])
	{
		const uint64_t probe = threadIdx.x ? ((base + KG_CHUNK - 1 < n_out) ? base + KG_CHUNK - 1 : n_out - 1) : base;
		uint32_t lo = 0, hi = n_runs;  // number of runs with out_start <= probe
		while (lo < hi)
		{
			const uint32_t mid = (lo + hi) >> 1;
			if ((uint64_t)runs[mid].out_start <= probe)
				lo = mid + 1;
			else
				hi = mid;
		}
		range[threadIdx.x] = threadIdx.x ? lo : (lo ? lo - 1 : 0);
	}
	__syncthreads();
	const uint32_t r_lo = range[0], r_hi = range[1];
#pragma unroll
	for (int j = 0; j < KG_PER_THREAD; j++)
	{
		const uint64_t i = base + kg_slot(j);
		if (i >= n_out)
			break;
		uint32_t lo = r_lo, hi = r_hi;  // number of runs with out_start <= i, searched inside the block's range
		while (lo < hi)
		{
			const uint32_t mid = (lo + hi) >> 1;
			if ((uint64_t)runs[mid].out_start <= i)
				lo = mid + 1;
			else
				hi = mid;
		}
		uint64_t lit = i;
		if (lo != 0)
		{
			const KgRun r = runs[lo - 1];
			const uint64_t run_end = (uint64_t)r.out_start + r.count;
			lit = (i < run_end) ? (uint64_t)r.after - 1 : i - (run_end - r.after);
		}
		out[i] = literals[lit];
	}
}

}  // namespace ako
