// pngcheck -- exercises the PNG reader / writer of cli_common.hpp without touching the GPU:
//   pngcheck <in.png> <out.raw> [<out.png> [effort]]
// writes "<channels> <width> <height>\n" + the decoded pixels to out.raw and, when asked, re-encodes
// them.  Used by tests/test_cli.py on machines without a GPU.
#include "cli_common.hpp"

int main(int argc, const char* argv[])
{
	if (argc < 3)
	{
		std::printf("usage: pngcheck <in.png> <out.raw> [<out.png> [effort]]\n");
		return 2;
	}
	try
	{
		const cli::Image img = cli::png_decode(cli::read_file(argv[1]));
		std::string head = std::to_string(img.channels) + " " + std::to_string(img.width) + " " + std::to_string(img.height) + "\n";
		std::vector<uint8_t> raw(head.begin(), head.end());
		raw.insert(raw.end(), img.pixels.begin(), img.pixels.end());
		cli::write_file(argv[2], raw.data(), raw.size());
		if (argc >= 4)
		{
			const int effort = argc >= 5 ? std::atoi(argv[4]) : 7;
			const auto png = cli::png_encode(img.pixels.data(), img.width, img.height, img.channels, effort);
			cli::write_file(argv[3], png.data(), png.size());
		}
	}
	catch (const std::exception& e)
	{
		std::printf("%s\n", e.what());
		return 1;
	}
	return 0;
}
