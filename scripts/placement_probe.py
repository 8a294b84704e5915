"""What decides the fast / slow level of the level-0 / level-1 kernel times (6-10 % apart, alternating between identical
processes: DESIGN.md 5.1)?  One process, several rounds; each round allocates the caller's buffers afresh -- through torch's
allocator or as physically contiguous device memory (hipExtMallocWithFlags, hipDeviceMallocContiguous) -- and a new plan,
and prints the kernel times.  usage: placement_probe.py torch|contig|contig-stream|contig-image [rounds]
(profiles/r4_placement.txt also holds the runs with the stream built from 2 MiB / 32 MiB physical chunks mapped in scrambled
order through the HIP virtual-memory API: no different from plain allocations, so that allocator was not kept)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ako_amd import api
from oracle import pyoracle as po
mode = sys.argv[1] if len(sys.argv) > 1 else "torch"   # torch | contig | contig-stream | contig-image
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w = h = 8192
host = torch.from_numpy(po.gen_image(0, w, h))
s = api.settings(wavelet=0, compression=2, q=16, g=16)
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]

class Raw:
    """a device allocation of our own, visible to torch through __cuda_array_interface__"""
    def __init__(self, shape, dtype):
        self.shape, self.np_dtype = shape, np.dtype(dtype)
        n = int(np.prod(shape)) * self.np_dtype.itemsize
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), n, 0x4)  # hipDeviceMallocContiguous
        assert rc == 0, f"hipExtMallocWithFlags(contiguous) failed: {rc}"
        self.ptr = p.value
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": self.np_dtype.str, "data": (self.ptr, False), "version": 2}
    def free(self):
        hip.hipFree(C.c_void_p(self.ptr))

def measure(plan, d, st, back):
    for _ in range(3):
        plan.encode(d, st); plan.decode(st, back)
    plan.synchronize(); plan.set_profiling(True)
    for _ in range(10):
        plan.encode(d, st); plan.decode(st, back)
    plan.synchronize()
    rec = {}
    for r in plan.kernel_records(False) + plan.kernel_records(True):
        rec.setdefault((r["name"], r["level"]), []).append(r["ms"])
    plan.set_profiling(False)
    g = lambda k: sum(rec[k]) / len(rec[k])
    return g(("fwd_stream_dd137_u8", 0)), g(("inv_stream_dd137_u8", 0)), g(("fwd_stream_dd137", 1)), g(("inv_stream_dd137", 1))

torch.cuda.init()
_ = torch.zeros(1, device="cuda")
for rnd in range(rounds):
    raws = []
    plan = api.Plan(s, 4, w, h)
    d = host.cuda().reshape(1, h, w, 4); st = plan.new_streams(); back = plan.new_images()
    if mode in ("contig", "contig-image"):
        raws += [Raw((1, h, w, 4), np.uint8), Raw((1, h, w, 4), np.uint8)]
        d, back = (torch.as_tensor(r, device="cuda") for r in raws[-2:])
        d.copy_(host.reshape(1, h, w, 4))
    if mode in ("contig", "contig-stream"):
        raws.append(Raw((1, plan.stream_bytes // 2), np.int16))
        st = torch.as_tensor(raws[-1], device="cuda")
    print(f"{mode} round {rnd}: fwd0 %.4f inv0 %.4f fwd1 %.4f inv1 %.4f ms" % measure(plan, d, st, back), flush=True)
    plan.close() if hasattr(plan, "close") else None
    del plan, d, st, back
    for r in raws:
        r.free()
    torch.cuda.empty_cache()
