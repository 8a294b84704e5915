/*
 * ako_synth.c -- the synthetic inputs of the benchmark configurations (BASELINE.json "configs").
 *
 * One xorshift32 draw per pixel in raster order, state seeded per image:
 *   G0 "smooth"  R = x*255/w + (n & 3), G = y*255/h + ((n >> 2) & 3), B = (x+y)*255/(w+h) + ((n >> 4) & 3),
 *                A = ((x/64 + y/64) & 1) ? 255 : 200        (each truncated to 8 bits)
 *   G1 "noise"   R, G, B, A = the four bytes of n
 *   G2 "plane"   one draw per sample, (n & 0x3FF) - 512
 * bench.py and the tools generate their inputs with these; the oracle keeps its own copy for the parity
 * tests (tests/test_cabi_host.py checks that the two agree).
 */
#include "ako_host.h"

static inline uint32_t xorshift32(uint32_t* s)
{
	uint32_t x = *s;
	x ^= x << 13;
	x ^= x >> 17;
	x ^= x << 5;
	return *s = x;
}

AKO_API void akoHostSynthImage(int generator, uint32_t seed, size_t w, size_t h, uint8_t* rgba)
{
	uint32_t state = seed;
	for (size_t y = 0; y < h; y++)
	{
		const uint8_t gy = (uint8_t)((y * 255) / h);
		uint8_t* row = rgba + y * w * 4;
		for (size_t x = 0; x < w; x++)
		{
			const uint32_t n = xorshift32(&state);
			uint8_t* px = row + x * 4;
			if (generator == 0)
			{
				px[0] = (uint8_t)((x * 255) / w + (n & 3));
				px[1] = (uint8_t)(gy + ((n >> 2) & 3));
				px[2] = (uint8_t)(((x + y) * 255) / (w + h) + ((n >> 4) & 3));
				px[3] = (((x >> 6) + (y >> 6)) & 1) ? 255 : 200;
			}
			else
			{
				px[0] = (uint8_t)n, px[1] = (uint8_t)(n >> 8);
				px[2] = (uint8_t)(n >> 16), px[3] = (uint8_t)(n >> 24);
			}
		}
	}
}

AKO_API void akoHostSynthPlane(uint32_t seed, size_t n, int16_t* plane)
{
	uint32_t state = seed;
	for (size_t i = 0; i < n; i++)
		plane[i] = (int16_t)((int)(xorshift32(&state) & 0x3FF) - 512);
}
