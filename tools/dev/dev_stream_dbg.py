import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qwen3_tts_rs_amd import synthetic_model
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "t0_golden.npz"))["codec_codes"]
m, cfg, ccfg, tmpl = synthetic_model("T0", dtype="f32", codec=True, talker=False)
codes = np.stack([g[:300], g[300:600]])
whole = m.decode_codes(codes)
up = m.total_upsample()
print("up", up, "len", len(whole[0]), ccfg["decoder_config"])
for pushes in ([5, 1, 7, 40, 3, 64, 64, 2, 50, 64], [30] * 10, [100, 100, 100]):
    s = m.open_codec_stream(2, 100)
    t = 0; off = 0
    for n in pushes:
        w = s.push(codes[:, t:t + n])
        ref = whole[0][off:off + w.shape[1]]
        err = np.abs(w[0] - ref)
        bad = np.nonzero(err > 1e-5)[0]
        print(f"push n={n:3d} frames [{t},{t+n}) samples {w.shape[1]} maxerr {err.max():.3g} bad {len(bad)} first {bad[:1] / up if len(bad) else None} last {bad[-1:] / up if len(bad) else None}")
        t += n; off += w.shape[1]
    s.close()
m.close()
