"""Developer scratch: aggregate rate of dependent kernels when 1..4 streams each replay a graph of them.
mode 1 = empty kernels, mode 2 = each kernel reads what the previous one wrote (256 blocks over 1 MiB)."""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qwen3_tts_rs_amd import _lib
lib = _lib.load()
lib.qtts_bench_sync.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
N = 4000
for mode, blocks, kib in ((1, 1, 64), (2, 256, 1024), (2, 64, 256)):
    for streams in (1, 2, 3, 4, 6):
        ctxs = []
        for _ in range(streams):
            c = C.c_void_p()
            _lib.check(lib.qtts_init(0, C.byref(c)))
            ctxs.append(c)
        res = [0.0] * streams
        lib.qtts_bench_sync_threads(streams)
        def work(i):
            us, err = C.c_double(), C.c_int()
            for _ in range(2):
                _lib.check(lib.qtts_bench_sync(ctxs[i], mode, blocks, kib, N, C.byref(us), C.byref(err)))
            res[i] = us.value
        th = [threading.Thread(target=work, args=(i,)) for i in range(streams)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        wall = time.perf_counter() - t0
        print(f"mode {mode} blocks {blocks} kib {kib} streams {streams}: per-node us per stream {['%.2f' % r for r in res]}  "
              f"aggregate (last rep, events) {sum(1.0 / r for r in res):.3f} kernels/us", flush=True)
        for c in ctxs:
            lib.qtts_shutdown(c)
