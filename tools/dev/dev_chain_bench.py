"""Developer scratch: cost of a dependent kernel boundary with plain vs write-through producer stores."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qwen3_tts_rs_amd import _lib
lib = _lib.load()
lib.qtts_bench_sync.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
ctx = C.c_void_p()
_lib.check(lib.qtts_init(0, C.byref(ctx)))
for kib in (64, 1024, 8192):
    for blocks in (64, 256):
        for mode, name in ((2, "plain"), (5, "alt2")):
            us, err = C.c_double(0), C.c_int(0)
            _lib.check(lib.qtts_bench_sync(ctx, mode, blocks, kib, 1000, C.byref(us), C.byref(err)))
            print(f"chain {kib:5d} KiB blocks={blocks:3d} {name:5s}: {us.value:6.2f} us/kernel wrong={err.value}", flush=True)
lib.qtts_shutdown(ctx)
