"""Developer scratch (not collected by pytest): codec decoder on the GPU against the oracle."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import bind
from qwen3_tts_rs_amd import synth, synthetic_model, round_bf16


def snr(ref, x):
    n = min(len(ref), len(x))
    e = ref[:n] - x[:n]
    return 10 * np.log10((ref[:n] ** 2).sum() / max((e ** 2).sum(), 1e-30))


def run(dtype):
    m, cfg, ccfg, tmpl = synthetic_model("T0", dtype=dtype, codec=True, talker=False, round_to_bf16=(dtype == "bf16"))
    o = bind.Oracle()
    for k, a in synth.iter_weights(synth.codec_keys(ccfg), 20261111):
        o.load_tensor(k, round_bf16(a) if dtype == "bf16" else a)
    o.set_codec_config(ccfg)
    rng = np.random.default_rng(0)
    lens = [20, 7, 33]
    T = max(lens)
    codes = rng.integers(0, 256, (len(lens), T, 4)).astype(np.uint32)
    t0 = time.time()
    wavs = m.decode_codes(codes, lens)
    print(dtype, "gpu decode", time.time() - t0)
    for b, L in enumerate(lens):
        ref = o.codec_decode(codes[b, :L])
        print(" row", b, "len", len(wavs[b]), len(ref), "snr dB", snr(ref, wavs[b]), "maxabs", np.abs(ref - wavs[b][:len(ref)]).max())
    # chunked decode: windows of 8 frames with 3 of left context
    w2 = m.decode_codes(codes[:1], [33], chunk=8, left_ctx=3)[0]
    ref2 = o.codec_decode(codes[0, :33], mode=1, chunk=8, left_ctx=3)
    print(" chunked len", len(w2), len(ref2), "snr dB", snr(ref2, w2))
    m.close()


if __name__ == "__main__":
    for dt in ("f32", "bf16"):
        run(dt)
