// How gfx950 bounds-checks a raw buffer store that carries a scalar offset (soffset): is soffset part of the
// range check, and can voffset + soffset wrap back into range?  (The streaming kernels want to drop a LANE through
// an out-of-range voffset and put the ROW offset into soffset.)
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 scripts/probe_buffer_soffset.hip -o /tmp/pbs && /tmp/pbs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__global__ void k(uint32_t* buf, uint32_t num_records, uint32_t voff, uint32_t soff, int null_rsrc)
{
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, null_rsrc ? 0 : (int)num_records, 0x00020000);
	if (threadIdx.x == 0)
		__builtin_amdgcn_raw_buffer_store_b32(0xABCD1234u, rs, (int)voff, (int)soff, 0);
}

static uint32_t host[4096];

static void run(uint32_t* d, const char* what, uint32_t nr, uint32_t voff, uint32_t soff, int null_rsrc)
{
	hipMemset(d, 0, sizeof host);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, nr, voff, soff, null_rsrc);
	hipMemcpy(host, d, sizeof host, hipMemcpyDeviceToHost);
	int hit = -1;
	for (int i = 0; i < 4096; i++)
		if (host[i] == 0xABCD1234u)
			hit = i * 4;
	printf("%-64s num_records=%u voffset=0x%08x soffset=0x%08x -> %s", what, nr, voff, soff, hit < 0 ? "dropped\n" : "stored at byte ");
	if (hit >= 0)
		printf("%d\n", hit);
}

int main()
{
	uint32_t* d;
	hipMalloc(&d, sizeof host);
	run(d, "plain in range", 1024, 100, 0, 0);
	run(d, "voffset in range, soffset pushes the address past num_records", 1024, 100, 2000, 0);
	run(d, "voffset out of range, soffset 0", 1024, 0xFFFFFFFFu, 0, 0);
	run(d, "voffset 0xFFFFFFFF, soffset 17 (sum wraps to 16)", 1024, 0xFFFFFFFFu, 17, 0);
	run(d, "voffset 2000 (out of range), soffset -1500 (sum 500)", 1024, 2000, (uint32_t)-1500, 0);
	run(d, "voffset 0x80000000, soffset 0x80000010 (sum wraps to 16)", 1024, 0x80000000u, 0x80000010u, 0);
	run(d, "null descriptor (num_records 0), offsets in range", 1024, 100, 200, 1);
	run(d, "voffset in range, soffset in range, sum in range", 1024, 100, 200, 0);
	return 0;
}
