"""Developer scratch: GEMM micro-benchmark over the decode shapes."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qwen3_tts_rs_amd import _lib
lib = _lib.load()
ctx = C.c_void_p()
_lib.check(lib.qtts_init(0, C.byref(ctx)))
lib.qtts_bench_gemm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
shapes = [("t.qkv", 4096, 2048), ("t.o", 2048, 2048), ("t.gate_up", 12288, 2048), ("t.down", 2048, 6144),
          ("c.qkv", 4096, 1024), ("c.o", 1024, 2048), ("c.gate_up", 6144, 1024), ("c.down", 1024, 3072), ("head", 3072, 2048)]
Ms = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [64, 1, 16, 128]
for M in Ms:
    tot = 0.0
    for name, N, K in shapes:
        us = C.c_double(0)
        _lib.check(lib.qtts_bench_gemm(ctx, 1, M, N, K, 0, 200, C.byref(us)))
        tot += us.value
        print(f"M={M:4d} {name:10s} N={N:6d} K={K:5d} {us.value:8.2f} us  {N*K*2/us.value/1e3:8.0f} GB/s", flush=True)
    print(f"M={M} sum {tot:.1f} us")
