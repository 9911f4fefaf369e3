"""Diagnostic: can uncached (fine-grained) device memory be exported over HIP IPC on this stack?"""
import ctypes as C
hip = C.CDLL("libamdhip64.so")
p = C.c_void_p()
for name, flag in (("finegrained", 0x1), ("uncached", 0x3)):
    e = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(1 << 20), C.c_uint(flag))
    h = (C.c_uint8 * 64)()
    e2 = hip.hipIpcGetMemHandle(h, p) if e == 0 else -1
    print(name, "malloc", e, "ipc_get", e2)
    if e == 0:
        hip.hipFree(p)
