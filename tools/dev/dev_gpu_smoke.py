"""Developer scratch (not collected by pytest): first end-to-end GPU check against the oracle."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import bind
from qwen3_tts_rs_amd import synth, synthetic_model, GenerationRequest, SamplingConfig

def run(shape, dtype, B=2, steps=12):
    B = int(os.environ.get("SMOKE_B", B))
    m, cfg, ccfg, tmpl = synthetic_model(shape, dtype=dtype, codec=False, round_to_bf16=(dtype == "bf16"))
    o = bind.Oracle()
    o.set_model_config(cfg)
    from qwen3_tts_rs_amd import round_bf16
    for k, a in synth.iter_weights(synth.talker_keys(cfg), 20261111):
        o.load_tensor(k, round_bf16(a) if dtype == "bf16" else a)
    greedy = SamplingConfig.cli_greedy()
    reqs = [GenerationRequest(text_ids=synth.synthetic_text_ids(tmpl, 10 + 3 * b, b), language="english", speaker="speaker1") for b in range(B)]
    t0 = time.time()
    out = m.generate(reqs, greedy, steps, trace=True, use_graph=False)
    print(shape, dtype, "gpu gen", time.time() - t0, out.num_steps, out.effective_lengths, m.last_timing)
    for b in range(B):
        r = o.generate(reqs[b].text_ids, bind.resolve_codec_prefill(cfg, "english", "speaker1"), greedy.as_dict(), steps,
                       speaker_token=bind.resolve_speaker_token(cfg, "speaker1"), trace=True)
        n = min(r["num_steps"], out.num_steps)
        same = (r["codes"][:n] == out.codes[b, :n]).all()
        dl = np.abs(r["logits"][:n] - out.logits[b, :n]).max(axis=1)
        print(" row", b, "tokens equal:", bool(same), "min margin", r["min_margin"], "logit maxabs first/all", dl[0], dl.max())
        if not same:
            bad = np.argwhere(r["codes"][:n] != out.codes[b, :n])[0]
            print("  first mismatch at", bad, r["codes"][bad[0]], out.codes[b, bad[0]])
    out2 = m.generate(reqs, greedy, steps, use_graph=True)
    print(" graph == eager:", bool((out2.codes == out.codes).all()))
    samp = SamplingConfig()
    for b in range(B): reqs[b].seed = 42 + b
    out3 = m.generate(reqs, samp, steps, use_graph=True)
    for b in range(B):
        r = o.generate(reqs[b].text_ids, bind.resolve_codec_prefill(cfg, "english", "speaker1"), samp.as_dict(), steps, seed=42 + b,
                       speaker_token=bind.resolve_speaker_token(cfg, "speaker1"))
        n = min(r["num_steps"], out3.num_steps)
        print(" sampled row", b, "equal:", bool((r["codes"][:n] == out3.codes[b, :n]).all()), (r["codes"][:n] != out3.codes[b, :n]).sum())
    m.close()

if __name__ == "__main__":
    for shape, dtype in [("T0", "f32"), ("T0", "bf16"), ("M0", "f32"), ("M0", "bf16")]:
        run(shape, dtype)
