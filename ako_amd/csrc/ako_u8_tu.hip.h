// ako_u8_tu.hip.h -- body of ako_u8_rgba.hip / ako_u8_rgb.hip: AKO_U8_CH = bytes per pixel, AKO_U8_NAME(x) = x ## _rgba / _rgb
#include "ako_stream.hip.h"
#include "ako_u8.h"

namespace ako
{

void AKO_U8_NAME(akoLaunchForwardU8)(int kind, bool lean, const LevelParams& P, const StreamGeom& G, uint32_t blocks, uint32_t threads, hipStream_t st)
{
	if (lean && geom_row_tiles(G) != 0 && kind == K_DD137)  // strips over whole rows of tiles: the ROWS bodies, kernels of their own
		hipLaunchKernelGGL((k_forward_u8_rows<K_DD137, AKO_U8_CH>), dim3(blocks), dim3(threads), 0, st, P, G);
	else if (lean && geom_row_tiles(G) != 0 && kind == K_CDF53)
		hipLaunchKernelGGL((k_forward_u8_rows<K_CDF53, AKO_U8_CH>), dim3(blocks), dim3(threads), 0, st, P, G);
	else if (lean && kind == K_DD137)
		hipLaunchKernelGGL((k_forward_u8_lean<K_DD137, AKO_U8_CH>), dim3(blocks), dim3(threads), 0, st, P, G);
	else if (lean && kind == K_CDF53)
		hipLaunchKernelGGL((k_forward_u8_lean<K_CDF53, AKO_U8_CH>), dim3(blocks), dim3(threads), 0, st, P, G);
	else if (kind == K_DD137)
		hipLaunchKernelGGL((k_forward_stream_u8<K_DD137, AKO_U8_CH>), dim3(blocks), dim3(threads), 0, st, P, G);
	else if (kind == K_CDF53)
		hipLaunchKernelGGL((k_forward_stream_u8<K_CDF53, AKO_U8_CH>), dim3(blocks), dim3(threads), 0, st, P, G);
	else
		hipLaunchKernelGGL((k_forward_stream_u8<K_HAAR, AKO_U8_CH>), dim3(blocks), dim3(threads), 0, st, P, G);
}

template <bool OPT>
static void launch_inverse(int kind, const LevelParams& P, const StreamGeom& G, uint32_t blocks, uint32_t pairs, hipStream_t st)
{
	const dim3 threads(128 * pairs);  // the workgroup is 'pairs' pairs of waves (LDS plane swap inside each pair)
	const uint32_t lds = pairs * INV_U8_LDS_PER_PAIR;
	if (kind == K_DD137)
		hipLaunchKernelGGL((k_inverse_stream_u8<K_DD137, OPT, AKO_U8_CH>), dim3(blocks), threads, lds, st, P, G);
	else if (kind == K_CDF53)
		hipLaunchKernelGGL((k_inverse_stream_u8<K_CDF53, OPT, AKO_U8_CH>), dim3(blocks), threads, lds, st, P, G);
	else
		hipLaunchKernelGGL((k_inverse_stream_u8<K_HAAR, OPT, AKO_U8_CH>), dim3(blocks), threads, lds, st, P, G);
}

void AKO_U8_NAME(akoLaunchInverseU8)(int kind, bool opt, bool lean, const LevelParams& P, const StreamGeom& G, uint32_t blocks, uint32_t pairs, hipStream_t st)
{
	if (opt && lean && kind != K_HAAR)  // (the lean kernel is the optimistic launch; the exact kernel behind it stays the general one)
	{
		const dim3 threads(128 * pairs);
		const uint32_t lds = pairs * INV_U8_LDS_PER_PAIR;
		if (geom_row_tiles(G) != 0 && kind == K_DD137)
			hipLaunchKernelGGL((k_inverse_u8_rows<K_DD137, AKO_U8_CH>), dim3(blocks), threads, lds, st, P, G);
		else if (geom_row_tiles(G) != 0)
			hipLaunchKernelGGL((k_inverse_u8_rows<K_CDF53, AKO_U8_CH>), dim3(blocks), threads, lds, st, P, G);
		else if (kind == K_DD137)
			hipLaunchKernelGGL((k_inverse_u8_lean<K_DD137, AKO_U8_CH>), dim3(blocks), threads, lds, st, P, G);
		else
			hipLaunchKernelGGL((k_inverse_u8_lean<K_CDF53, AKO_U8_CH>), dim3(blocks), threads, lds, st, P, G);
	}
	else if (opt)
		launch_inverse<true>(kind, P, G, blocks, pairs, st);
	else
		launch_inverse<false>(kind, P, G, blocks, pairs, st);
}

}  // namespace ako

#if defined(AKO_STAMPS) && AKO_U8_CH == 4
// measurement builds: the phase sums of the lean kernels (ako_u8_lean.hip.h) summed over the waves, 2 x 10 values; reset != 0
// clears them afterwards
extern "C" __attribute__((visibility("default"))) int akoHipLeanStamps(unsigned long long* out, int reset)
{
	const size_t n = (size_t)2 * ako::STAMP_WAVES * 12;
	unsigned long long* h = (unsigned long long*)calloc(n, sizeof(unsigned long long));
	if (h == nullptr || hipMemcpyFromSymbol(h, HIP_SYMBOL(ako::ako_lean_stamps), n * sizeof(unsigned long long)) != hipSuccess)
	{
		free(h);
		return 1;
	}
	for (int d = 0; d < 2; d++)
		for (int i = 0; i < 10; i++)
		{
			unsigned long long sum = 0;
			for (int w = 0; w < ako::STAMP_WAVES; w++)
				sum += h[((size_t)d * ako::STAMP_WAVES + w) * 12 + i];
			out[d * 10 + i] = sum;
		}
	// behind the sums: birth and end of every row's latest wave (2 x STAMP_WAVES x 2 values), if the caller left room for them
	if (reset & 2)
		for (int d = 0; d < 2; d++)
			for (int w = 0; w < ako::STAMP_WAVES; w++)
			{
				out[20 + ((size_t)d * ako::STAMP_WAVES + w) * 2 + 0] = h[((size_t)d * ako::STAMP_WAVES + w) * 12 + 10];
				out[20 + ((size_t)d * ako::STAMP_WAVES + w) * 2 + 1] = h[((size_t)d * ako::STAMP_WAVES + w) * 12 + 11];
			}
	// reset & 4 (-DAKO_STAMPS=2 builds): twelve values per row instead -- birth, end, HW_ID, XCC_ID, workgroup, wave in it, strip,
	// segment, role, border flags
	if (reset & 4)
		for (int d = 0; d < 2; d++)
			for (int w = 0; w < ako::STAMP_WAVES; w++)
			{
				const unsigned long long* r = h + ((size_t)d * ako::STAMP_WAVES + w) * 12;
				unsigned long long* o = out + 20 + ((size_t)d * ako::STAMP_WAVES + w) * 12;
				o[0] = r[10], o[1] = r[11];
				for (int i = 0; i < 10; i++)
					o[2 + i] = r[i];  // ... [10] kernel entry, [11] last stores acknowledged
			}
	int rc = 0;
	if (reset & 1)
	{
		for (size_t i = 0; i < n; i++)
			h[i] = 0;
		rc = hipMemcpyToSymbol(HIP_SYMBOL(ako::ako_lean_stamps), h, n * sizeof(unsigned long long)) != hipSuccess;
	}
	free(h);
	return rc;
}
#endif
