import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qwen3_tts_rs_amd import _lib
lib = _lib.load()
ctx = C.c_void_p()
_lib.check(lib.qtts_init(0, C.byref(ctx)))
lib.qtts_bench_gemm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
for M in (16, 64):
    for N, K in [(12288, 2048), (49152, 2048), (98304, 2048), (16384, 6144), (65536, 1024)]:
        for ks in (1, 2, 4):
            us = C.c_double(0)
            _lib.check(lib.qtts_bench_gemm(ctx, 1, M, N, K, ks, 50, C.byref(us)))
            print(f"M={M:4d} N={N:6d} K={K:5d} KS={ks} {us.value:8.2f} us  {N*K*2/us.value/1e3:8.0f} GB/s", flush=True)
