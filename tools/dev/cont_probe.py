"""Which launch widths give bit-equal rows between a plain batch and continuous batching (S17 bf16, seeded sampling)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from test_gpu_continuous import _model, _requests
from qwen3_tts_rs_amd import SamplingConfig
m, tmpl = _model("bf16", eos_gain=6.0, shape="S17")
sampling = SamplingConfig(temperature=1.0, top_k=0)
max_new = 48
reqs = _requests(tmpl, int(os.environ.get("PROBE_N", "40")))
def first_diff(a, b, e):
    d = np.argwhere(a[:e] != b[:e])
    return None if len(d) == 0 else tuple(int(v) for v in d[0])
combos = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(16, 16), (64, 64)]
n = len(reqs)
for width, slots in combos:
    cont = m.generate(reqs, sampling, max_new, slots=slots)
    bad = []
    for lo in range(0, n, width):
        grp = list(range(lo, min(lo + width, n)))
        grp += [g % n for g in range(grp[-1] + 1, grp[-1] + 1 + width - len(grp))]   # fillers keep the launch width
        plain = m.generate([reqs[u] for u in grp], sampling, max_new)
        for j, u in enumerate(grp[:min(width, n - lo)]):
            e = plain.effective_lengths[j]
            fd = first_diff(cont.codes[u], plain.codes[j], min(e, cont.effective_lengths[u]))
            if fd is not None or cont.effective_lengths[u] != e:
                bad.append((u, e, cont.effective_lengths[u], fd))
    print(f"plain batches of {width} vs slots {slots}: {len(bad)} of {n} utterances differ", bad[:6], flush=True)
m.close()
