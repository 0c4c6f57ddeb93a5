import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qwen3_tts_rs_amd import _lib
lib = _lib.load()
ctx = C.c_void_p()
_lib.check(lib.qtts_init(0, C.byref(ctx)))
lib.qtts_bench_stream.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
for mb in (8, 16, 48, 192):
    nbytes = mb << 20
    for pattern, blocks, unroll, ld in [(0, 256, 8, 0), (0, 512, 8, 0), (0, 1024, 8, 0), (0, 2048, 4, 0), (0, 4096, 4, 0),
                                        (1, 256, 8, 4096), (1, 512, 8, 4096), (1, 1024, 4, 4096), (1, 512, 8, 2048), (1, 2048, 4, 4096)]:
        us = C.c_double(0)
        _lib.check(lib.qtts_bench_stream(ctx, nbytes, blocks, pattern, unroll, ld, 100, C.byref(us)))
        print(f"{mb:4d} MB pattern={pattern} blocks={blocks:5d} unroll={unroll} ld={ld:6d}: {us.value:8.2f} us {nbytes/us.value/1e3:8.0f} GB/s", flush=True)
