"""hipIpc between two processes on one GPU: which allocation kinds can be exported and opened on this box, with the
error text of every failing call.  Prints one line per variant (tools/calls/r6q.sh)."""
import ctypes
import multiprocessing as mp
import os
import sys


class H(ctypes.Structure):
    _fields_ = [("r", ctypes.c_char * 64)]


def hip():
    lib = ctypes.CDLL("libamdhip64.so")
    lib.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    lib.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    lib.hipIpcGetMemHandle.argtypes = [ctypes.POINTER(H), ctypes.c_void_p]
    lib.hipIpcOpenMemHandle.argtypes = [ctypes.POINTER(ctypes.c_void_p), H, ctypes.c_uint]
    lib.hipGetErrorString.restype = ctypes.c_char_p
    lib.hipGetErrorString.argtypes = [ctypes.c_int]
    lib.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    return lib


def exporter(kind, ptracer, conn):
    lib = hip()
    if ptracer:
        ctypes.CDLL(None).prctl(0x59616d61, ctypes.c_ulong(-1 & (2 ** 64 - 1)), 0, 0, 0)   # PR_SET_PTRACER, ANY
    lib.hipSetDevice(0)
    p = ctypes.c_void_p()
    nbytes = 4 << 20
    if kind == "malloc":
        e = lib.hipMalloc(ctypes.byref(p), nbytes)
    elif kind == "fine":
        e = lib.hipExtMallocWithFlags(ctypes.byref(p), nbytes, 0x1)
    elif kind == "uncached":
        e = lib.hipExtMallocWithFlags(ctypes.byref(p), nbytes, 0x3)
    if e:
        conn.send(("alloc", e, lib.hipGetErrorString(e).decode()))
        return
    src = (ctypes.c_int * 4)(11, 22, 33, 44)
    lib.hipMemcpy(p, src, 16, 1)
    lib.hipDeviceSynchronize()
    h = H()
    e = lib.hipIpcGetMemHandle(ctypes.byref(h), p)
    if e:
        conn.send(("get", e, lib.hipGetErrorString(e).decode()))
        return
    conn.send(("ok", bytes(h.r), os.getpid()))
    conn.recv()


def importer(raw, flags, conn):
    lib = hip()
    lib.hipSetDevice(0)
    lib.hipFree(None)
    h = H()
    ctypes.memmove(ctypes.byref(h), raw, 64)
    p = ctypes.c_void_p()
    e = lib.hipIpcOpenMemHandle(ctypes.byref(p), h, flags)
    if e:
        conn.send(("open", e, lib.hipGetErrorString(e).decode()))
        return
    dst = (ctypes.c_int * 4)()
    e = lib.hipMemcpy(dst, p, 16, 2)
    conn.send(("ok", list(dst), e))


def main():
    print("HSA_ENABLE_IPC_MODE_LEGACY =", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
    try:
        print("ptrace_scope =", open("/proc/sys/kernel/yama/ptrace_scope").read().strip())
    except OSError as e:
        print("ptrace_scope: ", e)
    print("CapEff:", [l.strip() for l in open("/proc/self/status") if l.startswith("CapEff")])
    ctx = mp.get_context("spawn")
    for kind in ("malloc", "fine", "uncached"):
        for ptracer in (0, 1):
            for flags in (1,):
                a, b = ctx.Pipe()
                pe = ctx.Process(target=exporter, args=(kind, ptracer, b))
                pe.start()
                r = a.recv() if a.poll(120) else ("timeout",)
                if r[0] != "ok":
                    print(kind, "ptracer", ptracer, "EXPORT FAILED", r)
                    pe.join(10)
                    continue
                c, d = ctx.Pipe()
                pi = ctx.Process(target=importer, args=(r[1], flags, d))
                pi.start()
                r2 = c.recv() if c.poll(120) else ("timeout",)
                print(kind, "ptracer", ptracer, "flags", flags, "->", r2)
                sys.stdout.flush()
                a.send("done")
                pi.join(10)
                pe.join(10)


if __name__ == "__main__":
    main()
