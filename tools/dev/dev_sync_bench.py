"""Developer scratch: grid-barrier and graph-node costs."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qwen3_tts_rs_amd import _lib
lib = _lib.load()
lib.qtts_bench_sync.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
ctx = C.c_void_p()
_lib.check(lib.qtts_init(0, C.byref(ctx)))
for blocks, threads in [(64, 256), (256, 256), (256, 512), (256, 1024), (512, 256), (512, 512)]:
    us, err = C.c_double(0), C.c_int(0)
    _lib.check(lib.qtts_bench_sync(ctx, 0, blocks, threads, 2000, C.byref(us), C.byref(err)))
    print(f"barrier blocks={blocks} threads={threads}: {us.value:.2f} us/barrier errors={err.value}", flush=True)
for n in (100, 1000):
    us, err = C.c_double(0), C.c_int(0)
    _lib.check(lib.qtts_bench_sync(ctx, 1, 1, 64, n, C.byref(us), C.byref(err)))
    print(f"graph of {n} empty kernels: {us.value:.2f} us/node", flush=True)
lib.qtts_shutdown(ctx)
