#!/usr/bin/env python3
"""Golden vectors for the talker forward and the code-predictor loop from an independent public implementation.

The talker and the code predictor are Qwen3 decoder stacks (per-head q/k RMSNorm, RoPE, GQA, SwiGLU; the talker's
multimodal RoPE degenerates to plain RoPE because the reference feeds the same position id in all three rows).
`transformers.Qwen3Model` (installed in the build container) is loaded with the synthetic T0 weights and run on seeded
input embeddings; everything around the stacks (codec head, the code predictor's input projection, heads, embedding
tables, greedy choice, the UNMASKED two-token prefill of nn/code_predictor.rs:455-460) is plain torch here.

    python tests/golden/make_qwen3_golden.py          # writes tests/golden/qwen3_golden.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
SEED = 20261111


def stack(c, W, prefix):
    import torch
    from transformers import Qwen3Config, Qwen3Model
    hc = Qwen3Config(vocab_size=8, hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"],
                     num_hidden_layers=c["num_hidden_layers"], num_attention_heads=c["num_attention_heads"],
                     num_key_value_heads=c["num_key_value_heads"], head_dim=c["head_dim"], hidden_act="silu",
                     max_position_embeddings=c["max_position_embeddings"], rms_norm_eps=c["rms_norm_eps"],
                     rope_parameters={"rope_theta": c["rope_theta"], "rope_type": "default"}, attention_bias=c["attention_bias"],
                     use_sliding_window=False, sliding_window=None)
    hc._attn_implementation = "eager"
    m = Qwen3Model(hc).eval()
    sd = m.state_dict()
    for k in sd:
        if k == "embed_tokens.weight":
            continue
        src = torch.from_numpy(W[prefix + k])
        assert tuple(src.shape) == tuple(sd[k].shape), k
        sd[k].copy_(src)
    m.load_state_dict(sd)
    return m


def main():
    import torch
    import torch.nn.functional as F
    from transformers import DynamicCache
    from qwen3_tts_rs_amd import synth
    cfg, ccfg, tmpl = synth.shape_set("T0")
    t = cfg["talker_config"]
    cp = t["code_predictor_config"]
    W = {k: a for k, a in synth.iter_weights(synth.talker_keys(cfg), SEED)}
    tw = lambda k: torch.from_numpy(W[k])
    talker = stack(t, W, "talker.model.")
    pred = stack(cp, W, "talker.code_predictor.model.")
    rng = np.random.default_rng(5)
    out = {}
    with torch.no_grad():
        # ---- talker forward over a 13-token sequence (nn/talker.rs:126-155, 392-404)
        emb = (rng.standard_normal((13, t["hidden_size"])) * 0.5).astype(np.float32)
        hid = talker(inputs_embeds=torch.from_numpy(emb)[None]).last_hidden_state[0]
        out["talker_embeds"] = emb
        out["talker_hidden"] = hid.numpy()
        out["talker_logits"] = F.linear(hid, tw("talker.codec_head.weight")).numpy()
        # ---- code predictor, greedy (nn/code_predictor.rs:433-531): [talker hidden, embed(cb0)] unmasked, then one token per pass
        G = t["num_code_groups"]
        cb0 = 37
        proj = lambda x: F.linear(x, tw("talker.code_predictor.small_to_mtp_projection.weight"), tw("talker.code_predictor.small_to_mtp_projection.bias"))
        x = torch.stack([hid[-1], tw("talker.model.codec_embedding.weight")[cb0]])[None]
        cache = DynamicCache(config=pred.config)
        h = pred(inputs_embeds=proj(x), attention_mask={"full_attention": torch.zeros(1, 1, 2, 2)}, past_key_values=cache, use_cache=True,
                 position_ids=torch.arange(2)[None]).last_hidden_state[0, -1]
        codes, logits = [], []
        for j in range(G - 1):
            lg = F.linear(h, tw(f"talker.code_predictor.lm_head.{j}.weight"))
            logits.append(lg.numpy())
            code = int(torch.argmax(lg))
            codes.append(code)
            if j + 1 < G - 1:
                e = tw(f"talker.code_predictor.model.codec_embedding.{j}.weight")[code][None, None]
                h = pred(inputs_embeds=proj(e), attention_mask={"full_attention": None}, past_key_values=cache, use_cache=True,
                         position_ids=torch.tensor([[j + 2]])).last_hidden_state[0, -1]
        out["cp_cb0"] = np.int64(cb0)
        out["cp_codes"] = np.asarray(codes, np.uint32)
        out["cp_logits"] = np.stack(logits)
    np.savez_compressed(os.path.join(HERE, "qwen3_golden.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
