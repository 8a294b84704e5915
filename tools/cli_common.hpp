// cli_common.hpp -- what akoenc / akodec share: option parsing, PNG I/O on zlib, Adler-32, timers.
//
// The reference tools (tools/akoenc.cpp, tools/akodec.cpp) sit on lodepng and an OptionsManager class;
// neither is part of the transform path, so this is a small independent implementation that keeps the
// command-line surface (flag names, defaults, value ranges: tools/akoenc.cpp:340-395,
// tools/akodec.cpp:267-286) and the PNG subset the reference accepts (8 bit grey / grey+alpha / RGB /
// RGBA, tools/akoenc.cpp:79-91).
#ifndef AKO_CLI_COMMON_HPP
#define AKO_CLI_COMMON_HPP

#include <zlib.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace cli
{

struct Failure : std::runtime_error
{
	using std::runtime_error::runtime_error;
};

// ---- options --------------------------------------------------------------------------------

class Options
{
  public:
	void flag(const std::string& s, const std::string& l, const std::string& help)
	{
		add({s, l, help, KIND_FLAG, "", 0, 0, {}});
	}
	void integer(const std::string& s, const std::string& l, const std::string& help, long def, long lo, long hi)
	{
		add({s, l, help, KIND_INT, std::to_string(def), lo, hi, {}});
	}
	void text(const std::string& s, const std::string& l, const std::string& help, const std::string& def,
	          const std::vector<std::string>& choices = {})
	{
		add({s, l, help, KIND_TEXT, def, 0, 0, choices});
	}

	// returns false (after printing why) on a malformed command line
	bool parse(int argc, const char* argv[])
	{
		for (int i = 1; i < argc; i++)
		{
			const std::string a = argv[i];
			Entry* e = find(a);
			if (e == nullptr)
			{
				std::printf("Unknown option '%s'\n", a.c_str());
				return false;
			}
			if (e->kind == KIND_FLAG)
			{
				e->value = "1";
				continue;
			}
			if (i + 1 >= argc)
			{
				std::printf("Option '%s' needs a value\n", a.c_str());
				return false;
			}
			const std::string v = argv[++i];
			if (e->kind == KIND_INT)
			{
				char* end = nullptr;
				const long n = std::strtol(v.c_str(), &end, 10);
				if (end == v.c_str() || *end != '\0' || n < e->lo || n > e->hi)
				{
					std::printf("Option '%s' takes an integer from %ld to %ld\n", a.c_str(), e->lo, e->hi);
					return false;
				}
				e->value = std::to_string(n);
			}
			else
			{
				if (!e->choices.empty())
				{
					bool ok = false;
					for (const auto& c : e->choices)
						ok = ok || (upper(c) == upper(v));
					if (!ok)
					{
						std::printf("Option '%s' takes one of:", a.c_str());
						for (const auto& c : e->choices)
							std::printf(" %s", c.c_str());
						std::printf("\n");
						return false;
					}
				}
				e->value = v;
			}
		}
		return true;
	}

	bool on(const std::string& l) const { return get(l).value == "1"; }
	long number(const std::string& l) const { return std::strtol(get(l).value.c_str(), nullptr, 10); }
	std::string str(const std::string& l) const { return get(l).value; }
	// position of the chosen value in the option's list: the tools cast it straight to the ako.h enums
	int choice(const std::string& l) const
	{
		const Entry& e = get(l);
		for (size_t k = 0; k < e.choices.size(); k++)
			if (upper(e.choices[k]) == upper(e.value))
				return (int)k;
		return 0;
	}
	void print_help() const
	{
		for (const auto& e : entries_)
		{
			std::printf("  %s, %s", e.s.c_str(), e.l.c_str());
			if (e.kind == KIND_INT)
				std::printf(" <%ld..%ld> (default %s)", e.lo, e.hi, e.value.c_str());
			if (e.kind == KIND_TEXT && !e.choices.empty())
			{
				std::printf(" <");
				for (size_t k = 0; k < e.choices.size(); k++)
					std::printf("%s%s", k ? "|" : "", e.choices[k].c_str());
				std::printf("> (default %s)", e.value.c_str());
			}
			std::printf("\n      %s\n", e.help.c_str());
		}
	}

  private:
	enum Kind
	{
		KIND_FLAG,
		KIND_INT,
		KIND_TEXT
	};
	struct Entry
	{
		std::string s, l, help;
		Kind kind;
		std::string value;
		long lo, hi;
		std::vector<std::string> choices;
	};
	std::vector<Entry> entries_;

	static std::string upper(std::string s)
	{
		for (auto& c : s)
			c = (char)std::toupper((unsigned char)c);
		return s;
	}
	void add(Entry e) { entries_.push_back(std::move(e)); }
	Entry* find(const std::string& name)
	{
		for (auto& e : entries_)
			if (e.s == name || e.l == name)
				return &e;
		return nullptr;
	}
	const Entry& get(const std::string& l) const
	{
		for (const auto& e : entries_)
			if (e.l == l)
				return e;
		throw Failure("internal: option " + l + " not declared");
	}
};

// ---- small helpers --------------------------------------------------------------------------

inline uint32_t adler32_of(const uint8_t* data, size_t len)  // what '-ch' prints (tools/misc.hpp:59-82)
{
	uLong a = adler32(0L, Z_NULL, 0);
	while (len != 0)
	{
		const uInt step = (uInt)(len > (1u << 30) ? (1u << 30) : len);
		a = adler32(a, data, step);
		data += step, len -= step;
	}
	return (uint32_t)a;
}

inline std::vector<uint8_t> read_file(const std::string& name)
{
	FILE* fp = std::fopen(name.c_str(), "rb");
	if (fp == nullptr)
		throw Failure("Error at opening file '" + name + "'");
	std::vector<uint8_t> out;
	uint8_t buf[1 << 16];
	size_t n;
	while ((n = std::fread(buf, 1, sizeof buf, fp)) != 0)
		out.insert(out.end(), buf, buf + n);
	const bool bad = std::ferror(fp) != 0;
	std::fclose(fp);
	if (bad)
		throw Failure("Error at reading file '" + name + "'");
	return out;
}

inline void write_file(const std::string& name, const void* data, size_t size)
{
	FILE* fp = std::fopen(name.c_str(), "wb");
	if (fp == nullptr || std::fwrite(data, 1, size, fp) != size || std::fclose(fp) != 0)
		throw Failure("Write error");
}

class Timer
{
  public:
	void start(bool fresh)
	{
		from_ = std::chrono::steady_clock::now();
		if (fresh)
			total_ms_ = 0.0;
	}
	double stop()
	{
		total_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - from_).count();
		return total_ms_;
	}

  private:
	std::chrono::steady_clock::time_point from_;
	double total_ms_ = 0.0;
};

// ---- PNG (8 bit, colour types 0 / 2 / 4 / 6) ---------------------------------------------------

struct Image
{
	size_t width = 0, height = 0, channels = 0;
	std::vector<uint8_t> pixels;  // interleaved, row pitch width * channels
};

namespace png_detail
{
inline uint32_t be32(const uint8_t* p)
{
	return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}
inline void put_be32(std::vector<uint8_t>& v, uint32_t x)
{
	v.push_back((uint8_t)(x >> 24)), v.push_back((uint8_t)(x >> 16)), v.push_back((uint8_t)(x >> 8)), v.push_back((uint8_t)x);
}
inline int paeth(int a, int b, int c)
{
	const int p = a + b - c;
	const int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
	return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
// undo the filter of one scanline in place; prev = the reconstructed line above (nullptr for the first)
inline void unfilter(int type, uint8_t* line, const uint8_t* prev, size_t bytes, size_t bpp)
{
	for (size_t i = 0; i < bytes; i++)
	{
		const int a = (i >= bpp) ? line[i - bpp] : 0;
		const int b = prev ? prev[i] : 0;
		const int c = (prev && i >= bpp) ? prev[i - bpp] : 0;
		int add = 0;
		switch (type)
		{
		case 0: add = 0; break;
		case 1: add = a; break;
		case 2: add = b; break;
		case 3: add = (a + b) >> 1; break;
		case 4: add = paeth(a, b, c); break;
		default: throw Failure("PNG error: unknown filter type");
		}
		line[i] = (uint8_t)(line[i] + add);
	}
}
inline void filter(int type, const uint8_t* line, const uint8_t* prev, size_t bytes, size_t bpp, uint8_t* out)
{
	for (size_t i = 0; i < bytes; i++)
	{
		const int a = (i >= bpp) ? line[i - bpp] : 0;
		const int b = prev ? prev[i] : 0;
		const int c = (prev && i >= bpp) ? prev[i - bpp] : 0;
		int sub = 0;
		switch (type)
		{
		case 1: sub = a; break;
		case 2: sub = b; break;
		case 3: sub = (a + b) >> 1; break;
		case 4: sub = paeth(a, b, c); break;
		default: break;
		}
		out[i] = (uint8_t)(line[i] - sub);
	}
}
// Adam7 pass geometry
static const int A7_X0[7] = {0, 4, 0, 2, 0, 1, 0}, A7_Y0[7] = {0, 0, 4, 0, 2, 0, 1};
static const int A7_DX[7] = {8, 8, 4, 4, 2, 2, 1}, A7_DY[7] = {8, 8, 8, 4, 4, 2, 2};
}  // namespace png_detail

inline Image png_decode(const std::vector<uint8_t>& file)
{
	using namespace png_detail;
	static const uint8_t SIG[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	if (file.size() < 8 + 25 || std::memcmp(file.data(), SIG, 8) != 0)
		throw Failure("PNG error: not a PNG file");

	Image img;
	int interlace = 0;
	bool have_head = false, ended = false;
	std::vector<uint8_t> idat;
	size_t at = 8;
	while (!ended)
	{
		if (at + 12 > file.size())
			throw Failure("PNG error: truncated file");
		const uint32_t len = be32(&file[at]);
		const uint8_t* type = &file[at + 4];
		if ((size_t)len > file.size() - at - 12)
			throw Failure("PNG error: truncated chunk");
		const uint8_t* body = &file[at + 8];
		if (be32(body + len) != (uint32_t)crc32(crc32(0L, Z_NULL, 0), type, (uInt)len + 4))
			throw Failure("PNG error: chunk checksum mismatch");

		if (std::memcmp(type, "IHDR", 4) == 0)
		{
			if (len != 13)
				throw Failure("PNG error: bad header");
			img.width = be32(body), img.height = be32(body + 4);
			const int depth = body[8], ctype = body[9];
			interlace = body[12];
			switch (ctype)  // tools/akoenc.cpp:79-86
			{
			case 0: img.channels = 1; break;
			case 4: img.channels = 2; break;
			case 2: img.channels = 3; break;
			case 6: img.channels = 4; break;
			default: throw Failure("Unsupported channels number (" + std::to_string(ctype) + ")");
			}
			if (depth != 8)  // tools/akoenc.cpp:88-90
				throw Failure("Unsupported bits per pixel-component (" + std::to_string(depth) + ")");
			if (img.width == 0 || img.height == 0 || body[10] != 0 || body[11] != 0 || interlace > 1)
				throw Failure("PNG error: bad header");
			have_head = true;
		}
		else if (std::memcmp(type, "IDAT", 4) == 0)
			idat.insert(idat.end(), body, body + len);
		else if (std::memcmp(type, "IEND", 4) == 0)
			ended = true;
		else if ((type[0] & 0x20) == 0 && std::memcmp(type, "PLTE", 4) != 0)
			throw Failure("PNG error: unknown critical chunk");
		at += (size_t)len + 12;
	}
	if (!have_head || idat.empty())
		throw Failure("PNG error: no image data");

	const size_t bpp = img.channels;
	// size of the filtered data
	size_t raw = 0;
	if (interlace == 0)
		raw = img.height * (1 + img.width * bpp);
	else
		for (int p = 0; p < 7; p++)
		{
			const size_t pw = (img.width + A7_DX[p] - 1 - A7_X0[p]) / A7_DX[p];
			const size_t ph = (img.height + A7_DY[p] - 1 - A7_Y0[p]) / A7_DY[p];
			if (pw != 0 && ph != 0)
				raw += ph * (1 + pw * bpp);
		}

	std::vector<uint8_t> data(raw);
	{
		z_stream z;
		std::memset(&z, 0, sizeof z);
		if (inflateInit(&z) != Z_OK)
			throw Failure("PNG error: zlib");
		size_t in_at = 0, out_at = 0;
		int rc = Z_OK;
		while (rc != Z_STREAM_END)
		{
			const size_t in_step = std::min<size_t>(idat.size() - in_at, 1u << 30);
			const size_t out_step = std::min<size_t>(data.size() - out_at, 1u << 30);
			z.next_in = idat.data() + in_at, z.avail_in = (uInt)in_step;
			z.next_out = data.data() + out_at, z.avail_out = (uInt)out_step;
			rc = inflate(&z, Z_NO_FLUSH);
			in_at += in_step - z.avail_in, out_at += out_step - z.avail_out;
			if (rc != Z_OK && rc != Z_STREAM_END)
				break;
			if (rc == Z_OK && in_step - z.avail_in == 0 && out_step - z.avail_out == 0)
				break;  // no progress: truncated input or too much output
		}
		inflateEnd(&z);
		if (rc != Z_STREAM_END || out_at != data.size())
			throw Failure("PNG error: broken image data");
	}

	img.pixels.resize(img.width * img.height * bpp);
	if (interlace == 0)
	{
		const size_t line = img.width * bpp;
		const uint8_t* prev = nullptr;
		for (size_t y = 0; y < img.height; y++)
		{
			uint8_t* src = &data[y * (line + 1)];
			unfilter(src[0], src + 1, prev, line, bpp);
			std::memcpy(&img.pixels[y * line], src + 1, line);
			prev = src + 1;
		}
	}
	else
	{
		size_t off = 0;
		for (int p = 0; p < 7; p++)
		{
			const size_t pw = (img.width + A7_DX[p] - 1 - A7_X0[p]) / A7_DX[p];
			const size_t ph = (img.height + A7_DY[p] - 1 - A7_Y0[p]) / A7_DY[p];
			if (pw == 0 || ph == 0)
				continue;
			const size_t line = pw * bpp;
			const uint8_t* prev = nullptr;
			for (size_t y = 0; y < ph; y++)
			{
				uint8_t* src = &data[off + y * (line + 1)];
				unfilter(src[0], src + 1, prev, line, bpp);
				prev = src + 1;
				for (size_t x = 0; x < pw; x++)
					std::memcpy(&img.pixels[((A7_Y0[p] + y * A7_DY[p]) * img.width + A7_X0[p] + x * A7_DX[p]) * bpp],
					            src + 1 + x * bpp, bpp);
			}
			off += ph * (line + 1);
		}
	}
	return img;
}

// effort 1..10 (akodec '-e', tools/akodec.cpp:43-71): higher = smaller file, slower
inline std::vector<uint8_t> png_encode(const uint8_t* pixels, size_t width, size_t height, size_t channels, int effort)
{
	using namespace png_detail;
	if (channels < 1 || channels > 4)
		throw Failure("Unsupported channels number (" + std::to_string(channels) + ")");
	static const uint8_t CTYPE[5] = {0, 0, 4, 2, 6};
	const size_t bpp = channels, line = width * bpp;

	std::vector<uint8_t> filtered(height * (line + 1));
	std::vector<uint8_t> trial(line);
	for (size_t y = 0; y < height; y++)
	{
		const uint8_t* cur = pixels + y * line;
		const uint8_t* prev = y ? pixels + (y - 1) * line : nullptr;
		uint8_t* dst = &filtered[y * (line + 1)];
		int best = 0;
		if (effort >= 2)  // minimum sum of absolute differences over the five filters
		{
			uint64_t best_sum = ~0ull;
			for (int f = 0; f < 5; f++)
			{
				filter(f, cur, prev, line, bpp, trial.data());
				uint64_t sum = 0;
				for (size_t i = 0; i < line; i++)
					sum += (uint64_t)std::abs((int)(int8_t)trial[i]);
				if (sum < best_sum)
					best_sum = sum, best = f;
			}
		}
		dst[0] = (uint8_t)best;
		filter(best, cur, prev, line, bpp, dst + 1);
	}

	const int level = effort <= 1 ? 1 : (effort >= 9 ? 9 : effort);
	std::vector<uint8_t> packed(compressBound((uLong)std::min<size_t>(filtered.size(), 1u << 30)) +
	                            filtered.size() / 1000 * 2 + (filtered.size() >> 30) * 64 + 64);
	{
		z_stream z;
		std::memset(&z, 0, sizeof z);
		if (deflateInit(&z, level) != Z_OK)
			throw Failure("PNG error: zlib");
		packed.resize(std::max<size_t>(packed.size(), filtered.size() + filtered.size() / 512 + 1024));
		size_t in_at = 0, out_at = 0;
		int rc = Z_OK;
		while (rc != Z_STREAM_END)
		{
			const size_t in_step = std::min<size_t>(filtered.size() - in_at, 1u << 30);
			const size_t out_step = std::min<size_t>(packed.size() - out_at, 1u << 30);
			z.next_in = filtered.data() + in_at, z.avail_in = (uInt)in_step;
			z.next_out = packed.data() + out_at, z.avail_out = (uInt)out_step;
			rc = deflate(&z, (in_at + in_step == filtered.size()) ? Z_FINISH : Z_NO_FLUSH);
			in_at += in_step - z.avail_in, out_at += out_step - z.avail_out;
			if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR)
				break;
			if (out_at == packed.size())
				packed.resize(packed.size() * 2);
		}
		deflateEnd(&z);
		if (rc != Z_STREAM_END)
			throw Failure("PNG error: zlib deflate");
		packed.resize(out_at);
	}

	std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	auto chunk = [&out](const char* type, const uint8_t* body, size_t len) {
		put_be32(out, (uint32_t)len);
		const size_t from = out.size();
		out.insert(out.end(), type, type + 4);
		out.insert(out.end(), body, body + len);
		put_be32(out, (uint32_t)crc32(crc32(0L, Z_NULL, 0), &out[from], (uInt)(len + 4)));
	};
	std::vector<uint8_t> head;
	put_be32(head, (uint32_t)width), put_be32(head, (uint32_t)height);
	head.push_back(8), head.push_back(CTYPE[channels]), head.push_back(0), head.push_back(0), head.push_back(0);
	chunk("IHDR", head.data(), head.size());
	for (size_t at = 0; at < packed.size() || at == 0; at += (1u << 30))  // IDAT bodies stay below 2^31
	{
		chunk("IDAT", packed.data() + at, std::min<size_t>(packed.size() - at, 1u << 30));
		if (packed.empty())
			break;
	}
	chunk("IEND", nullptr, 0);
	return out;
}

}  // namespace cli

#endif
