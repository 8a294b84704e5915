// akodec -- .ako -> PNG through the public ako.h API, on the MI355X transform path.
//
// Counterpart of the reference tool (tools/akodec.cpp): flags and defaults of :267-286, the one-line
// summary of :246-256, '-b' stage timing, '-ch' Adler-32 of the decoded pixels.  The PNG it writes
// holds the same pixels as the reference's; the file bytes differ (zlib instead of lodepng's deflate).
#include "ako.h"
#include "cli_common.hpp"

namespace
{

struct StageTimers
{
	cli::Timer format, wavelet, compression;
};

void on_event(size_t tile_no, size_t total_tiles, enum akoEvent e, void* user)
{
	auto* t = static_cast<StageTimers*>(user);
	const bool first = (tile_no == 0), last = (tile_no == total_tiles - 1);
	auto finish = [last](cli::Timer& w, const char* label) {
		const double ms = w.stop();
		if (last)
			std::printf(" - %s: %g ms\n", label, ms);
	};
	switch (e)
	{
	case AKO_EVENT_FORMAT_START: t->format.start(first); break;
	case AKO_EVENT_WAVELET_START: t->wavelet.start(first); break;
	case AKO_EVENT_COMPRESSION_START: t->compression.start(first); break;
	case AKO_EVENT_FORMAT_END: finish(t->format, "Format"); break;
	case AKO_EVENT_WAVELET_END: finish(t->wavelet, "Wavelet transformation"); break;
	case AKO_EVENT_COMPRESSION_END: finish(t->compression, "Compression"); break;
	default: break;
	}
}

void print_banner()
{
	std::printf("Ako decoding tool (MI355X transform path)\n");
	std::printf(" - libako v%i.%i.%i, format %i\n", akoVersionMajor(), akoVersionMinor(), akoVersionPatch(),
	            akoFormatVersion());
	std::printf(" - zlib %s\n", zlibVersion());
}

}  // namespace

int main(int argc, const char* argv[])
{
	cli::Options o;
	o.flag("-v", "--version", "Print program version.");
	o.flag("-h", "--help", "Print this help.");
	o.flag("-verbose", "--verbose", "Print all available information while decoding.");
	o.flag("-quiet", "--quiet", "Don't print anything.");
	o.text("-i", "--input", "Input filename (.ako).", "");
	o.text("-o", "--output", "Output filename (PNG). Without it everything runs and the result is discarded.", "");
	o.integer("-e", "--effort", "Computational effort spent on the PNG, from 1 to 10.", 7, 1, 10);
	o.flag("-b", "--benchmark", "Print the time spent per stage.");
	o.flag("-ch", "--checksum", "Print the Adler-32 of the decoded pixels.");
	o.integer("-dev", "--device", "[extra] HIP device to run on.", -1, -1, 1024);
	if (!o.parse(argc, argv))
		return 1;

	if (o.on("--help"))
	{
		std::printf("USAGE\n    akodec [options] -i <input.ako> -o <output.png>\n    akodec [options] -i <input.ako>\n\n");
		o.print_help();
		return 0;
	}
	if (o.on("--version"))
	{
		print_banner();
		return 0;
	}
	const bool verbose = o.on("--verbose"), quiet = o.on("--quiet");
	const bool benchmark = o.on("--benchmark"), checksum = o.on("--checksum");
	const int effort = (int)o.number("--effort");
	if (o.number("--device") >= 0)
		setenv("AKO_HIP_DEVICE", o.str("--device").c_str(), 1);

	try
	{
		const std::string in_name = o.str("--input"), out_name = o.str("--output");
		if (in_name.empty())
			throw cli::Failure("No input filename specified");
		if (verbose)
		{
			print_banner();
			std::printf("Opening input: '%s'...\n", in_name.c_str());
		}
		const std::vector<uint8_t> blob = cli::read_file(in_name);

		akoSettings s = akoDefaultSettings();
		size_t channels = 0, width = 0, height = 0;
		akoStatus status = AKO_ERROR;
		akoCallbacks cb = akoDefaultCallbacks();
		StageTimers stages;
		cli::Timer total;
		const bool timing = benchmark && !quiet;
		if (timing)
		{
			total.start(true);
			cb.events = on_event, cb.events_data = &stages;
			std::printf("Benchmark: \n");
		}
		uint8_t* pixels = akoDecodeExt(&cb, blob.size(), blob.data(), &s, &channels, &width, &height, &status);
		if (timing)
			std::printf(" - Total: %g ms\n", total.stop());
		if (pixels == nullptr)
			throw cli::Failure(std::string("Ako error: '") + akoStatusString(status) + "'");

		if (verbose)
			std::printf("Input data: %zu channels, %zux%zu px, wavelet: %i, color: %i, wrap: %i, compression: %i\n",
			            channels, width, height, (int)s.wavelet, (int)s.color, (int)s.wrap, (int)s.compression);

		uint32_t sum = 0;
		if (checksum)
			sum = cli::adler32_of(pixels, width * height * channels);

		if (verbose)
			std::printf("Encoding...\n");
		const std::vector<uint8_t> png = cli::png_encode(pixels, width, height, channels, effort);
		if (!out_name.empty())
		{
			if (verbose)
				std::printf("Writing output: '%s'...\n", out_name.c_str());
			cli::write_file(out_name, png.data(), png.size());
		}

		const double raw = (double)(width * height * channels), packed = (double)blob.size();
		const double bpp = packed / raw * 8.0 * (double)channels;
		if (!quiet)
		{
			if (checksum)
				std::printf("(%08x) ", sum);
			std::printf("%.2f kB <- %.2f kB, ratio: %.2f:1, %.4f bpp\n", raw / 1000.0, packed / 1000.0, raw / packed, bpp);
		}
		akoDefaultFree(pixels);
	}
	catch (const std::exception& e)
	{
		std::printf("%s\n", e.what());
		return 1;
	}
	return 0;
}
