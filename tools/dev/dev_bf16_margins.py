"""Prints how far the bf16 path sits from its parity bars at the benchmarked shapes (logit error, waveform SNR)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import snr_db
import test_gpu_parity_full_size as P
from oracle import bind

m, o, cfg, ccfg, tmpl = P._pair(bind, "S17", "bf16", talker=True, codec=True)
frames = 8
texts, reqs = P._batch64(tmpl)
out = m.generate(reqs, P._greedy(), frames, trace=True)
worst = 0
for b in (0, 7, 13, 31, 40, 52, 63):
    ref = P._oracle_row(bind, o, cfg, texts[b], frames)
    same = (ref["codes"] == out.codes[b]).all(axis=1)
    upto = min((int(np.argmin(same)) if not same.all() else frames) + 1, frames)
    errs = np.abs(ref["logits"][:upto] - out.logits[b][:upto]).max(axis=1)
    worst = max(worst, errs.max())
    cw = 0
    for t in range(upto):
        oc, ol = o.code_predictor(out.hidden[b][t], int(out.codes[b][t][0]), P.GREEDY)
        agree = oc == out.codes[b][t][1:]
        nj = min((int(np.argmin(agree)) if not agree.all() else len(oc)) + 1, len(oc))
        cw = max(cw, float(np.abs(ol[:nj] - out.cp_logits[b][t][:nj]).max()))
    print(f"row {b}: frames with equal history {upto}, max-abs logit error per frame {np.round(errs, 4)}, code predictor {cw:.4f}", flush=True)
print("worst", worst)
rng = np.random.default_rng(5)
codes = rng.integers(0, 2048, (3, 24, 16)).astype(np.uint32)
wavs = m.decode_codes(codes, [24, 12, 9])
for b, L in enumerate([24, 12, 9]):
    print(f"codec row {b} ({L} frames): {snr_db(o.codec_decode(codes[b, :L]), wavs[b]):.2f} dB", flush=True)
m.close()
