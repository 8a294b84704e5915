/*
 * ako_misc.c -- defaults, status strings and version getters of the public API
 * (reference: library/misc.c:30-95, library/version.c).
 */
#include "ako_host.h"

#include <stdlib.h>

AKO_API struct akoSettings akoDefaultSettings(void)
{
	/* reference defaults: library/misc.c:30-47 */
	struct akoSettings s;
	s.wavelet = AKO_WAVELET_DD137;
	s.color = AKO_COLOR_YCOCG;
	s.wrap = AKO_WRAP_CLAMP;
	s.compression = AKO_COMPRESSION_KAGARI;
	s.tiles_dimension = 0;
	s.quantization = 16;
	s.gate = 0;
	s.chroma_loss = 1;
	s.discard_non_visible = 0;
	return s;
}

AKO_API struct akoCallbacks akoDefaultCallbacks(void)
{
	struct akoCallbacks c;
	c.malloc = malloc;
	c.realloc = realloc;
	c.free = free;
	c.events = NULL;
	c.events_data = NULL;
	return c;
}

AKO_API void akoDefaultFree(void* p)
{
	free(p);
}

AKO_API const char* akoStatusString(enum akoStatus status)
{
	/* texts as printed by the reference tools (library/misc.c:71-95) */
	static const char* const text[] = {
	    "Everything Ok!",
	    "Something went wrong",
	    "Invalid channels number",
	    "Invalid dimensions",
	    "Invalid tiles dimensions",
	    "Invalid wrap mode",
	    "Invalid wavelet transformation",
	    "Invalid color transformation",
	    "Invalid compression method",
	    "Invalid input",
	    "Invalid callbacks",
	    "Invalid magic (not an Ako file)",
	    "Unsupported version",
	    "No enough memory",
	    "Invalid flags",
	    "Broken input/premature end",
	};
	if ((unsigned)status < sizeof(text) / sizeof(text[0]))
		return text[(unsigned)status];
	return "Unknown status code";
}

AKO_API int akoVersionMajor(void)
{
	return AKO_VERSION_MAJOR;
}

AKO_API int akoVersionMinor(void)
{
	return AKO_VERSION_MINOR;
}

AKO_API int akoVersionPatch(void)
{
	return AKO_VERSION_PATCH;
}

AKO_API int akoFormatVersion(void)
{
	return AKO_FORMAT_VERSION;
}
