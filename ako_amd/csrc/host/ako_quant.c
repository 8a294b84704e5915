/*
 * ako_quant.c -- quantizer / noise-gate step of one lift (reference: library/quantization.c:43-98).
 *
 * A handful of float scalars per tile; they stay on the host and are handed to the kernels as
 * integers, so that libm rounding is a host matter ("Float in quantization.c").
 * Formula: round( 2^(lift-1) * ((lift/total)+1)^6 / 2^6 * factor / (512 * 0.73) ) with
 * lift = log2(sqrt(cur_w * cur_h)) - 1, total = log2(sqrt(tile_w * tile_h)) - 1, evaluated in
 * float in the reference's operation order.
 */
#include "ako_host.h"

#include <math.h>

static float step_curve(float factor, float tile_w, float tile_h, float cur_w, float cur_h)
{
	const float norm = (512.0F * 0.73F);
	const float exponent = 6.0F;

	const float tile_side = sqrtf(tile_w * tile_h);
	const float cur_side = sqrtf(cur_w * cur_h);
	const float lifts_total = log2f(tile_side) - 1.0F;
	const float lift = log2f(cur_side) - 1.0F;

	const float ratio = (lift / lifts_total);
	const float highs = powf(ratio + 1.0F, exponent) / powf(2.0F, exponent);
	const float base = powf(2.0F, (lift - 1.0F)) * highs;
	return roundf(base * (factor / norm));
}

static int16_t clamp_step(float v, float lo)
{
	if (v < lo)
		v = lo;
	if (v > 32765.0F)
		v = 32765.0F;
	return (int16_t)v;
}

int16_t akoHostQuantStep(int factor, int mul, size_t tile_w, size_t tile_h, size_t cur_w, size_t cur_h)
{
	if (factor <= 0)
		return 1; /* lossless: divide by one (quantization.c:86-87) */
	return clamp_step(step_curve((float)factor * (float)mul, (float)tile_w, (float)tile_h, (float)cur_w, (float)cur_h),
	                  1.0F);
}

int16_t akoHostGateStep(int factor, int mul, size_t tile_w, size_t tile_h, size_t cur_w, size_t cur_h)
{
	if (factor <= 0)
		return 0; /* gate off (quantization.c:69-70) */
	return clamp_step(step_curve((float)factor * (float)mul, (float)tile_w, (float)tile_h, (float)cur_w, (float)cur_h),
	                  0.0F);
}
