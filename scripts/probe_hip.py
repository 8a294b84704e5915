import ctypes as C, sys, os
which = sys.argv[1]
if which == "system":
    L = C.CDLL("/opt/rocm/lib/libamdhip64.so.7")
else:
    import torch
    L = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
n = C.c_int(-1)
L.hipGetErrorString.restype = C.c_char_p
rc = L.hipGetDeviceCount(C.byref(n))
print(which, "rc", rc, L.hipGetErrorString(rc), "count", n.value)
v = C.c_int(0); L.hipRuntimeGetVersion(C.byref(v)); print("runtime version", v.value)
