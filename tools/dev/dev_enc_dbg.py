import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from golden.make_golden import test_audio as make_audio
from qwen3_tts_rs_amd import synthetic_model
for dt in ("f32", "bf16"):
    m, cfg, ccfg, tmpl = synthetic_model("T0", dtype=dt, codec=True, talker=False, encoder=True, round_to_bf16=False)
    clip = make_audio(72000, 5)
    a = m.encode_audio([clip], 16)[0]
    a2 = m.encode_audio([clip], 16)[0]
    b = m.encode_audio([clip, make_audio(24000, 6)], 16)[0]
    c = m.encode_audio([clip, make_audio(72000, 6)], 16)[0]
    print(dt, a.shape, a.max(), "rerun equal", np.array_equal(a, a2), "batch equal", np.array_equal(a, b), (a != b).mean(), "frames differing", np.nonzero((a != b).any(1))[0][:10],
          "same-len batch equal", np.array_equal(a, c))
    m.close()
