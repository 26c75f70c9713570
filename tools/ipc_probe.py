"""How long does hipIpcOpenMemHandle take as a function of the allocation's size (two processes, one device)?  Development aid:
the particle-sharded sweep maps its peers' trace buffers (tens of GB each) this way once at setup."""
import ctypes as C, multiprocessing as mp, os, sys, time

KEEP = os.environ.get("KEEP") == "1"   # keep every mapping (and every allocation) alive, as the sharded sweep's setup does

class Handle(C.Structure):          # hipIpcMemHandle_t: 64 opaque bytes, passed BY VALUE to hipIpcOpenMemHandle
    _fields_ = [("reserved", C.c_char * 64)]


def hip():
    L = C.CDLL(os.environ.get("HIPLIB", "libamdhip64.so"))   # HIPLIB: e.g. the copy bundled with PyTorch
    L.hipIpcGetMemHandle.argtypes = [C.POINTER(Handle), C.c_void_p]
    L.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), Handle, C.c_uint]
    L.hipIpcCloseMemHandle.argtypes = [C.c_void_p]
    L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    L.hipFree.argtypes = [C.c_void_p]
    return L

def child(conn):
    L = hip()
    assert L.hipSetDevice(0) == 0
    while True:
        msg = conn.recv()
        if msg is None:
            break
        h = Handle.from_buffer_copy(msg)
        p = C.c_void_p()
        t0 = time.perf_counter()
        rc = L.hipIpcOpenMemHandle(C.byref(p), h, int(os.environ.get("IPCFLAGS", "1")))   # 1 = hipIpcMemLazyEnablePeerAccess
        if rc == 0:   # touch it: one kernel-free proof that the mapping is usable is a 4-byte copy
            buf = C.c_int(0); L.hipMemcpy(C.byref(buf), p, 4, 2)
        dt = time.perf_counter() - t0
        rc2 = (L.hipIpcCloseMemHandle(p) if rc == 0 else -1) if not KEEP else 0
        conn.send((rc, dt, rc2))

if __name__ == "__main__":
    mp.set_start_method("spawn")
    a, b = mp.Pipe()
    pr = mp.Process(target=child, args=(b,)); pr.start()
    L = hip()
    assert L.hipSetDevice(0) == 0
    for mb in [int(x) for x in sys.argv[1:]] or [64, 512, 2048, 8192]:
        p = C.c_void_p()
        t0 = time.perf_counter()
        assert L.hipMalloc(C.byref(p), mb << 20) == 0
        t_alloc = time.perf_counter() - t0
        h = Handle()
        assert L.hipIpcGetMemHandle(C.byref(h), p) == 0
        a.send(bytes(h))
        if not a.poll(int(os.environ.get("POLL_S", "120"))):
            print(f"{mb} MiB: hipIpcOpenMemHandle did not return within 120 s", flush=True)
            pr.terminate(); sys.exit(1)
        rc, dt, rc2 = a.recv()
        print(f"{mb:6d} MiB: hipMalloc {t_alloc*1e3:8.1f} ms, hipIpcOpenMemHandle rc={rc} {dt*1e3:8.1f} ms, close rc={rc2}", flush=True)
        if not KEEP:
            L.hipFree(p)
    a.send(None); pr.join(10)
