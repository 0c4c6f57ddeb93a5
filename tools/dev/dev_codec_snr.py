"""Developer scratch: bf16 against f32 codec decode on the GPU at the real codec shape (identical bf16-rounded weights)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from conftest import snr_db
from qwen3_tts_rs_amd import synthetic_model
codes = np.random.default_rng(3).integers(0, 2048, (2, 60, 16)).astype(np.uint32)
out = {}
for dt in ("f32", "bf16"):
    m, cfg, ccfg, tmpl = synthetic_model("S06", dtype=dt, codec=True, talker=False, round_to_bf16=True)
    out[dt] = m.decode_codes(codes)
    m.close()
for b in range(2):
    print("row", b, "SNR bf16 vs f32: %.1f dB" % snr_db(out["f32"][b], out["bf16"][b]), "peak", float(np.abs(out["f32"][b]).max()))
