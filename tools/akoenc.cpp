// akoenc -- PNG -> .ako through the public ako.h API, on the MI355X transform path.
//
// Counterpart of the reference tool (tools/akoenc.cpp): same flags, defaults and ranges
// (:340-395), the same one-line summary (:303-313), '-b' stage timing through the event callback
// (tools/benchmark.hpp:60-84), '-ch' Adler-32 of the input pixels, and the '-dev-r' ratio search
// (:111-214).  Opt-in extras that the reference does not have: '--tiles', '--device'.
#include "ako.h"
#include "ako_hip.h"  // akoEncodeRatioExt: the ratio search as one call
#include "cli_common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

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
	switch (e)
	{
	case AKO_EVENT_FORMAT_START: t->format.start(first); break;
	case AKO_EVENT_WAVELET_START: t->wavelet.start(first); break;
	case AKO_EVENT_COMPRESSION_START: t->compression.start(first); break;
	case AKO_EVENT_FORMAT_END:
	{
		const double ms = t->format.stop();
		if (last)
			std::printf(" - Format: %g ms\n", ms);
		break;
	}
	case AKO_EVENT_WAVELET_END:
	{
		const double ms = t->wavelet.stop();
		if (last)
			std::printf(" - Wavelet transformation: %g ms\n", ms);
		break;
	}
	case AKO_EVENT_COMPRESSION_END:
	{
		const double ms = t->compression.stop();
		if (last)
			std::printf(" - Compression: %g ms\n", ms);
		break;
	}
	default: break;
	}
}

// One encode, or several while looking for the quantization that lands near 'ratio':1.
// The search follows the reference step for step (tools/akoenc.cpp:111-214) so that the same input
// ends in the same quantization and therefore in the same file.
size_t encode_pass(bool verbose, int ratio, const akoCallbacks& cb, const akoSettings& base, const cli::Image& img,
                   void** blob, akoStatus* status, std::string* search_note = nullptr)
{
	auto run = [&](const akoSettings& s) {
		return akoEncodeExt(&cb, &s, img.channels, img.width, img.height, img.pixels.data(), blob, status);
	};
	if (ratio == 0 || base.wavelet == AKO_WAVELET_NONE || base.compression == AKO_COMPRESSION_NONE)
		return run(base);

	akoSettings s = base;
	if (ratio == 1)  // lossless
	{
		s.quantization = 0, s.gate = 0;
		return run(s);
	}

	// The search as ONE library call (include/ako_hip.h: akoEncodeRatioExt): same bracketing and bisection, same
	// winner, same file -- with the pixels uploaded once and the image transformed once per colour transformation,
	// each candidate costing a re-quantization pass and the device entropy stage.  '-verbose' keeps the loop below,
	// which prints every step as the reference tool does.
	if (!verbose && std::getenv("AKOENC_SEARCH_BY_REENCODING") == nullptr)
	{
		int q = 0, encodes = 0, transforms = 0;
		cli::Timer t;
		t.start(true);
		const size_t size = akoEncodeRatioExt(&cb, &base, img.channels, img.width, img.height, img.pixels.data(), ratio, blob,
		                                      &q, &encodes, &transforms, status);
		if (search_note != nullptr && size != 0)
		{
			char line[160];
			std::snprintf(line, sizeof line, "Ratio search: quantization %i after %i candidates on %i transform%s, %g ms", q,
			              encodes, transforms, transforms == 1 ? "" : "s", t.stop());
			*search_note = line;
		}
		return size;
	}

	const size_t target = (img.width * img.height * img.channels) / (size_t)ratio;
	const size_t margin = (target * 4) / 100;
	if (verbose)
		std::printf("Target: %.2f kB, error: %.2f kB...\n", (double)target / 1000.0, (double)margin / 1000.0);

	auto report = [&](int cq, int fq, size_t cs, size_t fs) {
		if (verbose)
			std::printf(" - Q: %i|%i, %.1f|%.1f kB\n", cq, fq, (double)cs / 1000.0, (double)fs / 1000.0);
	};
	auto distance = [](size_t a, size_t b) { return a > b ? a - b : b - a; };

	// upper bracket: quantization 0; lower bracket: grow by x4 until the blob is small enough
	s.quantization = 0;
	size_t ceil_size = run(s), floor_size = ceil_size;
	int ceil_q = 0, floor_q = 0;
	s.quantization = 1;
	do
	{
		s.quantization *= 4;
		ceil_size = floor_size, ceil_q = floor_q;
		floor_size = run(s), floor_q = s.quantization;
		report(ceil_q, floor_q, ceil_size, floor_size);
	} while (floor_size > target);

	// bisection
	size_t last_size = floor_size;
	while (distance(floor_size, ceil_size) > margin && std::abs(floor_q - ceil_q) > 1)
	{
		s.quantization = (ceil_q + floor_q) / 2;
		last_size = run(s);
		if (last_size > target)
			ceil_size = last_size, ceil_q = s.quantization;
		else
			floor_size = last_size, floor_q = s.quantization;
		report(ceil_q, floor_q, ceil_size, floor_size);
	}

	const bool take_floor = distance(floor_size, target) < distance(ceil_size, target);
	const int q = take_floor ? floor_q : ceil_q;
	const size_t q_size = take_floor ? floor_size : ceil_size;
	if (verbose)
		std::printf(" - Q: %i\n", q);
	if (last_size == q_size)
		return last_size;  // the blob in *blob already is that encode
	s.quantization = q;
	return run(s);
}

void print_banner(const char* what)
{
	std::printf("Ako %s tool (MI355X transform path)\n", what);
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
	o.flag("-verbose", "--verbose", "Print all available information while encoding.");
	o.flag("-quiet", "--quiet", "Don't print anything.");
	o.text("-i", "--input", "Input filename (PNG, 8 bit grey / grey+alpha / RGB / RGBA).", "");
	o.text("-o", "--output", "Output filename. Without it everything runs and the result is discarded.", "");
	o.integer("-q", "--quantization", "Loss through coarser wavelet coefficients; 0 = lossless.", 16, 0, 8192);
	o.integer("-g", "--noise-gate", "Loss through dropping coefficients under a threshold; 0 = off.", 0, 0, 8192);
	o.text("-w", "--wavelet", "Wavelet transformation.", "DD137", {"DD137", "CDF53", "HAAR", "NONE"});
	o.text("-c", "--color", "Colour transformation.", "YCOCG", {"YCOCG", "SUBTRACT-G", "NONE"});
	o.text("-wr", "--wrap", "How lifting wraps around tile borders.", "CLAMP", {"CLAMP", "MIRROR", "REPEAT", "ZERO"});
	o.integer("-chroma-loss", "--chroma-loss", "Extra loss on every channel but the first; 0 = none.", 1, 0, 8192);
	o.flag("-d", "--discard-non-visible", "Zero the colour of fully transparent pixels (not lossless).");
	o.flag("-b", "--benchmark", "Print the time spent per stage.");
	o.flag("-ch", "--checksum", "Print the Adler-32 of the input pixels.");
	o.integer("-dev-r", "--dev-ratio", "Search the quantization that gives about this compression ratio.", 0, 0, 4096);
	o.text("-dev-compression", "--dev-compression", "Entropy stage.", "KAGARI", {"KAGARI", "MANBAVARAN", "NONE"});
	o.integer("-t", "--tiles", "[extra] Tile size, a power of two >= 8; 0 = one tile.", 0, 0, 1 << 30);
	o.integer("-dev", "--device", "[extra] HIP device to run on.", -1, -1, 1024);
	if (!o.parse(argc, argv))
		return 1;

	if (o.on("--help"))
	{
		std::printf("USAGE\n    akoenc [options] -i <input.png> -o <output.ako>\n    akoenc [options] -i <input.png>\n\n");
		o.print_help();
		return 0;
	}
	if (o.on("--version"))
	{
		print_banner("encoding");
		return 0;
	}

	akoSettings s = akoDefaultSettings();
	s.quantization = (int)o.number("--quantization");
	s.gate = (int)o.number("--noise-gate");
	s.discard_non_visible = o.on("--discard-non-visible") ? 1 : 0;
	s.wavelet = (akoWavelet)o.choice("--wavelet");
	s.color = (akoColor)o.choice("--color");
	s.wrap = (akoWrap)o.choice("--wrap");
	s.chroma_loss = (int)o.number("--chroma-loss");
	s.compression = (akoCompression)o.choice("--dev-compression");
	s.tiles_dimension = (size_t)o.number("--tiles");
	const int ratio = (int)o.number("--dev-ratio");
	const bool verbose = o.on("--verbose"), quiet = o.on("--quiet");
	const bool benchmark = o.on("--benchmark"), checksum = o.on("--checksum");
	if (o.number("--device") >= 0)
		setenv("AKO_HIP_DEVICE", o.str("--device").c_str(), 1);

	try
	{
		const std::string in_name = o.str("--input"), out_name = o.str("--output");
		if (in_name.empty())
			throw cli::Failure("No input filename specified");
		if (verbose)
		{
			print_banner("encoding");
			std::printf("Opening input: '%s'...\n", in_name.c_str());
		}
		const cli::Image img = cli::png_decode(cli::read_file(in_name));
		if (verbose)
			std::printf("Input data: %zu channels, %zux%zu px\n", img.channels, img.width, img.height);

		uint32_t sum = 0;
		if (checksum)
			sum = cli::adler32_of(img.pixels.data(), img.pixels.size());

		if (verbose)
			std::printf("Encoding...\n[Wavelet: %i, color: %i, wrap: %i, compression %i, chroma loss: %i, discard "
			            "non-visible: %i]\n",
			            (int)s.wavelet, (int)s.color, (int)s.wrap, (int)s.compression, s.chroma_loss,
			            s.discard_non_visible);

		void* blob = nullptr;
		akoStatus status = AKO_ERROR;
		akoCallbacks cb = akoDefaultCallbacks();
		StageTimers stages;
		cli::Timer total;
		const bool timing = benchmark && !quiet;
		if (timing)
		{
			total.start(true);
			if (ratio == 0)
			{
				cb.events = on_event, cb.events_data = &stages;
				std::printf("Benchmark: \n");
			}
		}
		std::string search_note;
		const size_t blob_size = encode_pass(verbose, ratio, cb, s, img, &blob, &status, &search_note);
		if (timing)
		{
			if (ratio != 0)
				std::printf("Benchmark: \n");
			if (!search_note.empty())
				std::printf(" - %s\n", search_note.c_str());
			std::printf(" - Total: %g ms\n", total.stop());
		}
		if (blob_size == 0)
			throw cli::Failure(std::string("Ako error: '") + akoStatusString(status) + "'");

		if (!out_name.empty())
		{
			if (verbose)
				std::printf("Writing output: '%s'...\n", out_name.c_str());
			cli::write_file(out_name, blob, blob_size);
		}

		const double raw = (double)img.pixels.size(), packed = (double)blob_size;
		const double bpp = packed / raw * 8.0 * (double)img.channels;
		if (!quiet)
		{
			if (checksum)
				std::printf("(%08x) ", sum);
			std::printf("%.2f kB -> %.2f kB, ratio: %.2f:1, %.4f bpp\n", raw / 1000.0, packed / 1000.0, raw / packed, bpp);
		}
		akoDefaultFree(blob);
	}
	catch (const std::exception& e)
	{
		std::printf("%s\n", e.what());
		return 1;
	}
	return 0;
}
