"""Synthetic Qwen3-TTS model factory: named shape sets, config JSON, random-init weights, text ids.

No checkpoint can be fetched (no network), so tests and benchmarks run on random-init weights of
the architecture the reference loads.  Tensor names are the HF keys the reference's VarBuilder
paths read (qwen_tts/src/nn/talker.rs:32-59,253-277, nn/code_predictor.rs:40-63,199-226,
nn/attention/unified.rs:52-77, nn/mlp.rs:30-41,104-113, nn/speaker_encoder.rs:44-50,489-538,
audio/decoder/v2.rs:53-134, audio/quantizer.rs:40-41,204-221).

Weights come from a counter-based generator (SplitMix64 of key-hash + element index, four 16-bit
uniforms summed) that uses only exact float64 arithmetic, so every machine regenerates the same
bits and nothing needs to be stored.
"""
from __future__ import annotations

import copy
import json
import math

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def counter_normal(key: str, n: int, seed: int, start: int = 0) -> np.ndarray:
    """n approximately-normal float64 values (Irwin-Hall of four 16-bit uniforms), unit variance."""
    base = np.uint64((_fnv1a(key) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(start, start + n, dtype=np.uint64) + base
    z = _splitmix64(idx)
    m = np.uint64(0xFFFF)
    s = ((z & m).astype(np.float64) + ((z >> np.uint64(16)) & m).astype(np.float64)
         + ((z >> np.uint64(32)) & m).astype(np.float64) + ((z >> np.uint64(48)) & m).astype(np.float64))
    # sum of 4 U[0,65535]/65536 has mean 2*(65535/65536) and variance 4/12
    return (s / 65536.0 - 2.0 * (65535.0 / 65536.0)) * math.sqrt(3.0)


def make_tensor(key: str, shape, kind: str, seed: int) -> np.ndarray:
    n = int(np.prod(shape))
    x = counter_normal(key, n, seed)
    k = kind.split(":")
    if k[0] == "normal":          # normal:<sigma>
        x = x * float(k[1])
    elif k[0] == "fanin":         # fanin:<gain>  sigma = gain / sqrt(fan_in)
        fan_in = int(np.prod(shape[1:]))
        x = x * (float(k[1]) / math.sqrt(fan_in))
    elif k[0] == "fanin_t":       # transposed conv (Cin, Cout, k): fan_in = Cin * k / stride
        x = x * (float(k[1]) / math.sqrt(shape[0] * shape[2] / float(k[2])))
    elif k[0] == "ones":          # ones:<jitter>
        x = 1.0 + x * float(k[1])
    elif k[0] == "const":         # const:<value>:<jitter>
        x = float(k[1]) + x * float(k[2])
    elif k[0] == "usage":         # strictly positive cluster usage
        x = 1.0 + 0.5 * np.abs(x)
    else:
        raise ValueError(kind)
    return x.astype(np.float32).reshape(shape)


# ----------------------------------------------------------------------------- shape sets
def _talker(H, I, L, nh, nkv, d, V, Ht, text_vocab, G, theta, mrope, interleaved, cp, ids, sliding=None):
    t = dict(hidden_size=H, intermediate_size=I, num_hidden_layers=L, num_attention_heads=nh,
             num_key_value_heads=nkv, head_dim=d, vocab_size=V, text_hidden_size=Ht, text_vocab_size=text_vocab,
             num_code_groups=G, rope_theta=theta, rms_norm_eps=1e-6, hidden_act="silu", attention_bias=False,
             max_position_embeddings=32768,
             rope_scaling=dict(rope_type="default", mrope_section=mrope, interleaved=interleaved),
             code_predictor_config=cp, sliding_window=sliding, use_sliding_window=False)
    t.update(ids)
    return t


def _cp(Hc, Ic, Lc, nh, nkv, d, Vc, G, theta=1e6):
    return dict(hidden_size=Hc, intermediate_size=Ic, num_hidden_layers=Lc, num_attention_heads=nh,
                num_key_value_heads=nkv, head_dim=d, vocab_size=Vc, num_code_groups=G, rope_theta=theta,
                rms_norm_eps=1e-6, hidden_act="silu", attention_bias=False, max_position_embeddings=32768,
                use_sliding_window=False, sliding_window=4096, max_window_layers=28)


_IDS_REAL = dict(codec_eos_token_id=2150, codec_bos_id=2149, codec_pad_id=2148, codec_think_id=2154,
                 codec_nothink_id=2155, codec_think_bos_id=2156, codec_think_eos_id=2157,
                 spk_id={f"speaker{i}": 3000 + i for i in range(9)},
                 spk_is_dialect={f"speaker{i}": False for i in range(9)},
                 codec_language_id={"english": 2050, "chinese": 2055})
_TOP_REAL = dict(im_start_token_id=151644, im_end_token_id=151645, tts_pad_token_id=151671,
                 tts_bos_token_id=151672, tts_eos_token_id=151673, tokenizer_type="12hz")
_TEMPLATE_REAL = dict(prefix=[151644, 77091, 198], suffix=[151645, 198, 151644, 77091, 198], text_hi=151643)

_IDS_T0 = dict(codec_eos_token_id=300, codec_bos_id=299, codec_pad_id=298, codec_think_id=304,
               codec_nothink_id=305, codec_think_bos_id=306, codec_think_eos_id=307,
               spk_id={f"speaker{i}": 400 + i for i in range(3)},
               spk_is_dialect={f"speaker{i}": False for i in range(3)},
               codec_language_id={"english": 350, "chinese": 355})
_TOP_T0 = dict(im_start_token_id=500, im_end_token_id=501, tts_pad_token_id=505, tts_bos_token_id=506,
               tts_eos_token_id=507, tokenizer_type="12hz")
_TEMPLATE_T0 = dict(prefix=[500, 502, 503], suffix=[501, 503, 500, 502, 503], text_hi=480)

_SPK_E0 = dict(mel_dim=128, enc_channels=[512, 512, 512, 512, 1536], enc_kernel_sizes=[5, 3, 3, 3, 1],
               enc_dilations=[1, 2, 3, 4, 1], enc_attention_channels=128, enc_res2net_scale=8, enc_se_channels=128,
               sample_rate=24000)
_SPK_ET0 = dict(mel_dim=128, enc_channels=[64, 64, 64, 64, 192], enc_kernel_sizes=[5, 3, 3, 3, 1],
                enc_dilations=[1, 2, 3, 4, 1], enc_attention_channels=32, enc_res2net_scale=8, enc_se_channels=32,
                sample_rate=24000)

CODEC_C0 = dict(codebook_size=2048, hidden_size=1024, latent_dim=1024, codebook_dim=256, rope_theta=10000.0,
                num_attention_heads=16, num_key_value_heads=16, attention_bias=False, sliding_window=72,
                intermediate_size=3072, hidden_act="silu", layer_scale_initial_scale=0.01, rms_norm_eps=1e-5,
                num_hidden_layers=8, num_quantizers=16, upsample_rates=[8, 5, 4, 3], upsampling_ratios=[2, 2],
                decoder_dim=1536, max_position_embeddings=8000)
CODEC_CT0 = dict(codebook_size=256, hidden_size=128, latent_dim=64, codebook_dim=64, rope_theta=10000.0,
                 num_attention_heads=2, num_key_value_heads=2, attention_bias=False, sliding_window=8,
                 intermediate_size=128, hidden_act="silu", layer_scale_initial_scale=0.01, rms_norm_eps=1e-5,
                 num_hidden_layers=2, num_quantizers=4, upsample_rates=[3, 2, 2, 2], upsampling_ratios=[2, 2],
                 decoder_dim=512, max_position_embeddings=8000)


def shape_set(name: str, model_type: str = "custom_voice"):
    """Returns (config dict in the reference's config.json layout, codec config dict, text template)."""
    if name == "S06":
        cp = _cp(1024, 3072, 5, 16, 8, 128, 2048, 16)
        t = _talker(1024, 3072, 28, 16, 8, 128, 3072, 2048, 151936, 16, 1e6, [24, 20, 20], True, cp, _IDS_REAL)
        top, tmpl, spk, codec = _TOP_REAL, _TEMPLATE_REAL, _SPK_E0, CODEC_C0
    elif name == "S17":
        cp = _cp(1024, 3072, 5, 16, 8, 128, 2048, 16)
        t = _talker(2048, 6144, 28, 16, 8, 128, 3072, 2048, 151936, 16, 1e6, [24, 20, 20], True, cp, _IDS_REAL)
        top, tmpl, spk, codec = _TOP_REAL, _TEMPLATE_REAL, _SPK_E0, CODEC_C0
    elif name == "M0":  # mid-size set: real head_dim / vocab, few layers (GPU parity at realistic tile shapes)
        cp = _cp(512, 1024, 2, 8, 4, 128, 2048, 16)
        t = _talker(1024, 2048, 3, 8, 4, 128, 3072, 512, 4096, 16, 1e6, [24, 20, 20], True, cp,
                    dict(_IDS_REAL, spk_id={f"speaker{i}": 3000 + i for i in range(3)}))
        top = dict(im_start_token_id=4000, im_end_token_id=4001, tts_pad_token_id=4005, tts_bos_token_id=4006,
                   tts_eos_token_id=4007, tokenizer_type="12hz")
        tmpl = dict(prefix=[4000, 4002, 4003], suffix=[4001, 4003, 4000, 4002, 4003], text_hi=3990)
        spk, codec = _SPK_ET0, CODEC_CT0
    elif name == "T0":
        cp = _cp(64, 128, 2, 2, 1, 64, 64, 4)
        t = _talker(128, 256, 2, 4, 2, 64, 1280, 128, 512, 4, 1e4, [12, 10, 10], True, cp, _IDS_T0)
        top, tmpl, spk, codec = _TOP_T0, _TEMPLATE_T0, _SPK_ET0, CODEC_CT0
    else:
        raise ValueError(f"unknown shape set {name}")
    cfg = dict(talker_config=copy.deepcopy(t), tts_model_type=model_type, tts_model_size=name)
    cfg.update(top)
    if model_type == "base":
        cfg["speaker_encoder_config"] = dict(spk, enc_dim=t["hidden_size"])
    return cfg, dict(decoder_config=copy.deepcopy(codec), encoder_valid_num_quantizers=codec["num_quantizers"],
                     input_sample_rate=24000, output_sample_rate=24000, decode_upsample_rate=1920,
                     encode_downsample_rate=1920), dict(tmpl)


# ------------------------------------------------------------------------------ key lists
def _layer_keys(prefix, H, I, nh, nkv, d, bias=False):
    ks = [(prefix + "self_attn.q_proj.weight", (nh * d, H), "normal:0.02"),
          (prefix + "self_attn.k_proj.weight", (nkv * d, H), "normal:0.02"),
          (prefix + "self_attn.v_proj.weight", (nkv * d, H), "normal:0.02"),
          (prefix + "self_attn.o_proj.weight", (H, nh * d), "normal:0.02"),
          (prefix + "self_attn.q_norm.weight", (d,), "ones:0.1"),
          (prefix + "self_attn.k_norm.weight", (d,), "ones:0.1"),
          (prefix + "mlp.gate_proj.weight", (I, H), "normal:0.02"),
          (prefix + "mlp.up_proj.weight", (I, H), "normal:0.02"),
          (prefix + "mlp.down_proj.weight", (H, I), "normal:0.02"),
          (prefix + "input_layernorm.weight", (H,), "ones:0.1"),
          (prefix + "post_attention_layernorm.weight", (H,), "ones:0.1")]
    if bias:
        ks += [(prefix + f"self_attn.{n}_proj.bias", (s,), "normal:0.02")
               for n, s in (("q", nh * d), ("k", nkv * d), ("v", nkv * d), ("o", H))]
    return ks


def talker_keys(cfg):
    t = cfg["talker_config"]
    c = t["code_predictor_config"]
    H, Ht, G = t["hidden_size"], t["text_hidden_size"], t["num_code_groups"]
    ks = []
    for i in range(t["num_hidden_layers"]):
        ks += _layer_keys(f"talker.model.layers.{i}.", H, t["intermediate_size"], t["num_attention_heads"],
                          t["num_key_value_heads"], t["head_dim"], t.get("attention_bias", False))
    ks += [("talker.model.norm.weight", (H,), "ones:0.1"),
           ("talker.model.codec_embedding.weight", (t["vocab_size"], H), "normal:1.0"),
           ("talker.model.text_embedding.weight", (t["text_vocab_size"], Ht), "normal:1.0"),
           ("talker.codec_head.weight", (t["vocab_size"], H), "normal:0.02"),
           ("talker.text_projection.linear_fc1.weight", (Ht, Ht), "normal:0.02"),
           ("talker.text_projection.linear_fc1.bias", (Ht,), "normal:0.02"),
           ("talker.text_projection.linear_fc2.weight", (H, Ht), "normal:0.02"),
           ("talker.text_projection.linear_fc2.bias", (H,), "normal:0.02")]
    Hc = c["hidden_size"]
    for i in range(c["num_hidden_layers"]):
        ks += _layer_keys(f"talker.code_predictor.model.layers.{i}.", Hc, c["intermediate_size"],
                          c["num_attention_heads"], c["num_key_value_heads"], c["head_dim"],
                          c.get("attention_bias", False))
    ks.append(("talker.code_predictor.model.norm.weight", (Hc,), "ones:0.1"))
    for j in range(G - 1):
        ks.append((f"talker.code_predictor.model.codec_embedding.{j}.weight", (c["vocab_size"], H), "normal:1.0"))
        ks.append((f"talker.code_predictor.lm_head.{j}.weight", (c["vocab_size"], Hc), "normal:0.02"))
    if Hc != H:
        ks += [("talker.code_predictor.small_to_mtp_projection.weight", (Hc, H), "normal:0.02"),
               ("talker.code_predictor.small_to_mtp_projection.bias", (Hc,), "normal:0.02")]
    return ks


def speaker_keys(cfg):
    s = cfg.get("speaker_encoder_config")
    if not s:
        return []
    ch, ker = s["enc_channels"], s["enc_kernel_sizes"]
    P = "speaker_encoder."

    def conv(name, cout, cin, k):
        return [(P + name + ".weight", (cout, cin, k), "fanin:1.0"), (P + name + ".bias", (cout,), "normal:0.02")]

    ks = conv("blocks.0.conv", ch[0], s["mel_dim"], ker[0])
    sc = s["enc_res2net_scale"]
    for i in range(1, len(ch) - 1):
        ks += conv(f"blocks.{i}.tdnn1.conv", ch[i], ch[i - 1], 1)
        for j in range(sc - 1):
            ks += conv(f"blocks.{i}.res2net_block.blocks.{j}.conv", ch[i] // sc, ch[i] // sc, ker[i])
        ks += conv(f"blocks.{i}.tdnn2.conv", ch[i], ch[i], 1)
        ks += conv(f"blocks.{i}.se_block.conv1", s["enc_se_channels"], ch[i], 1)
        ks += conv(f"blocks.{i}.se_block.conv2", ch[i], s["enc_se_channels"], 1)
    ks += conv("mfa.conv", ch[-1], sum(ch[1:-1]), ker[-1])
    ks += conv("asp.tdnn.conv", s["enc_attention_channels"], ch[-1] * 3, 1)
    ks += conv("asp.conv", ch[-1], s["enc_attention_channels"], 1)
    ks += conv("fc", s["enc_dim"], ch[-1] * 2, 1)
    return ks


def codec_keys(codec_cfg):
    c = codec_cfg["decoder_config"]
    Dc, Dl, Hd, Dd = c["codebook_dim"], c["latent_dim"], c["hidden_size"], c["decoder_dim"]
    nh, nkv = c["num_attention_heads"], c["num_key_value_heads"]
    d = c.get("head_dim") or Hd // nh
    I, Q = c["intermediate_size"], c["num_quantizers"]
    half = Dc // 2
    P = "decoder."
    ks = []
    for grp, n in (("rvq_first", 1), ("rvq_rest", Q - 1)):
        ks.append((P + f"quantizer.{grp}.input_proj.weight", (half, Dc, 1), "fanin:1.0"))
        ks.append((P + f"quantizer.{grp}.output_proj.weight", (Dc, half, 1), "fanin:1.0"))
        for i in range(n):
            ks.append((P + f"quantizer.{grp}.vq.layers.{i}._codebook.cluster_usage", (c["codebook_size"],), "usage"))
            ks.append((P + f"quantizer.{grp}.vq.layers.{i}._codebook.embedding_sum", (c["codebook_size"], half),
                       "normal:1.0"))

    def conv(name, cout, cin_g, k, gain="1.0"):
        return [(P + name + ".conv.weight", (cout, cin_g, k), f"fanin:{gain}"),
                (P + name + ".conv.bias", (cout,), "normal:0.02")]

    ks += conv("pre_conv", Dl, Dc, 3)
    T = P + "pre_transformer."
    ks += [(T + "input_proj.weight", (Hd, Dl), "fanin:1.0"), (T + "input_proj.bias", (Hd,), "normal:0.02"),
           (T + "output_proj.weight", (Dl, Hd), "fanin:1.0"), (T + "output_proj.bias", (Dl,), "normal:0.02"),
           (T + "norm.weight", (Hd,), "ones:0.1")]
    for i in range(c["num_hidden_layers"]):
        L = T + f"layers.{i}."
        ks += [(L + "self_attn.q_proj.weight", (nh * d, Hd), "fanin:1.0"),
               (L + "self_attn.k_proj.weight", (nkv * d, Hd), "fanin:1.0"),
               (L + "self_attn.v_proj.weight", (nkv * d, Hd), "fanin:1.0"),
               (L + "self_attn.o_proj.weight", (Hd, nh * d), "fanin:1.0"),
               (L + "mlp.gate_proj.weight", (I, Hd), "fanin:1.0"),
               (L + "mlp.up_proj.weight", (I, Hd), "fanin:1.0"),
               (L + "mlp.down_proj.weight", (Hd, I), "fanin:1.0"),
               (L + "input_layernorm.weight", (Hd,), "ones:0.1"),
               (L + "post_attention_layernorm.weight", (Hd,), "ones:0.1"),
               (L + "self_attn_layer_scale.scale", (Hd,), "const:0.1:0.01"),
               (L + "mlp_layer_scale.scale", (Hd,), "const:0.1:0.01")]
    for i, r in enumerate(c["upsampling_ratios"]):
        U = f"upsample.{i}."
        ks += [(P + U + "0.conv.weight", (Dl, Dl, r), f"fanin_t:1.0:{r}"), (P + U + "0.conv.bias", (Dl,), "normal:0.02")]
        ks += conv(U + "1.dwconv", Dl, 1, 7)
        ks += [(P + U + "1.norm.weight", (Dl,), "ones:0.1"), (P + U + "1.norm.bias", (Dl,), "normal:0.02"),
               (P + U + "1.pwconv1.weight", (4 * Dl, Dl), "fanin:1.0"), (P + U + "1.pwconv1.bias", (4 * Dl,), "normal:0.02"),
               (P + U + "1.pwconv2.weight", (Dl, 4 * Dl), "fanin:1.0"), (P + U + "1.pwconv2.bias", (Dl,), "normal:0.02"),
               (P + U + "1.gamma", (Dl,), "const:0.1:0.01")]
    ks += conv("decoder.0", Dd, Dl, 7)
    cur = Dd
    for b, r in enumerate(c["upsample_rates"]):
        B = f"decoder.{b + 1}.block."
        nxt = cur // 2
        ks += [(P + B + "0.alpha", (cur,), "normal:0.3"), (P + B + "0.beta", (cur,), "normal:0.3"),
               (P + B + "1.conv.weight", (cur, nxt, 2 * r), f"fanin_t:1.0:{r}"), (P + B + "1.conv.bias", (nxt,), "normal:0.02")]
        for u in range(3):
            R = B + f"{u + 2}."
            ks += [(P + R + "act1.alpha", (nxt,), "normal:0.3"), (P + R + "act1.beta", (nxt,), "normal:0.3"),
                   (P + R + "act2.alpha", (nxt,), "normal:0.3"), (P + R + "act2.beta", (nxt,), "normal:0.3")]
            ks += conv(R + "conv1", nxt, nxt, 7, "0.7") + conv(R + "conv2", nxt, nxt, 1, "0.3")
        cur = nxt
    nb = len(c["upsample_rates"])
    ks += [(P + f"decoder.{nb + 1}.alpha", (cur,), "normal:0.3"), (P + f"decoder.{nb + 1}.beta", (cur,), "normal:0.3")]
    ks += conv(f"decoder.{nb + 2}", 1, cur, 7, "0.08")
    return ks


def encoder_keys():
    """Codec encoder tensors (fixed architecture: audio/encoder/v2.rs:30-49, seanet.rs:93-150, quantizer.rs:86-172)."""
    P = "encoder."
    ks = []

    def conv(name, cout, cin, k, gain="1.0", bias=True):
        out = [(P + name + ".weight", (cout, cin, k), f"fanin:{gain}")]
        if bias:
            out.append((P + name + ".bias", (cout,), "normal:0.02"))
        return out

    ks += conv("encoder.layers.0.conv", 64, 1, 7)
    C, idx = 64, 1
    for r in (4, 5, 6, 8):
        ks += conv(f"encoder.layers.{idx}.block.1.conv", C // 2, C, 3) + conv(f"encoder.layers.{idx}.block.3.conv", C, C // 2, 1, "0.5")
        ks += conv(f"encoder.layers.{idx + 2}.conv", 2 * C, C, 2 * r)
        C *= 2
        idx += 3
    ks += conv(f"encoder.layers.{idx + 1}.conv", 512, C, 3)
    for i in range(8):
        L = P + f"encoder_transformer.layers.{i}."
        ks += [(L + f"self_attn.{n}_proj.weight", (512, 512), "fanin:1.0") for n in "qkvo"]
        ks += [(L + "mlp.fc1.weight", (2048, 512), "fanin:1.0"), (L + "mlp.fc2.weight", (512, 2048), "fanin:1.0"),
               (L + "input_layernorm.weight", (512,), "ones:0.1"), (L + "input_layernorm.bias", (512,), "normal:0.02"),
               (L + "post_attention_layernorm.weight", (512,), "ones:0.1"), (L + "post_attention_layernorm.bias", (512,), "normal:0.02"),
               (L + "self_attn_layer_scale.scale", (512,), "const:0.1:0.01"), (L + "mlp_layer_scale.scale", (512,), "const:0.1:0.01")]
    ks += conv("downsample.conv", 512, 512, 4, bias=False)
    for grp, n in (("semantic", 1), ("acoustic", 31)):
        Q = P + f"quantizer.{grp}_residual_vector_quantizer."
        ks += [(Q + "input_proj.weight", (256, 512, 1), "fanin:1.0"), (Q + "output_proj.weight", (512, 256, 1), "fanin:1.0")]
        for i in range(n):
            ks += [(Q + f"layers.{i}.codebook.embed_sum", (2048, 256), "normal:1.0"), (Q + f"layers.{i}.codebook.cluster_usage", (2048,), "usage")]
    return ks


def iter_weights(keys, seed: int):
    for key, shape, kind in keys:
        yield key, make_tensor(key, shape, kind, seed)


# ----------------------------------------------------------------------------- text / misc
def synthetic_text_ids(template, n_content: int, seed: int) -> np.ndarray:
    """<|im_start|>assistant\\n + n_content ids + <|im_end|>\\n<|im_start|>assistant\\n
    (shape of qwen_tts/src/text/processing.rs:430-435 output)."""
    z = _splitmix64(np.arange(n_content, dtype=np.uint64) + np.uint64((seed * 0x51ED270B + 12345) & 0xFFFFFFFF))
    body = (z % np.uint64(template["text_hi"])).astype(np.uint32)
    return np.concatenate([np.array(template["prefix"], np.uint32), body, np.array(template["suffix"], np.uint32)])


def config_json(cfg) -> str:
    return json.dumps(cfg)


def weight_bytes_per_frame(cfg, bytes_per_elem: int = 2) -> dict:
    """Algorithmic weight bytes one generated frame must stream (talker + G-1 code-predictor passes)."""
    t = cfg["talker_config"]
    c = t["code_predictor_config"]
    H, I, L = t["hidden_size"], t["intermediate_size"], t["num_hidden_layers"]
    nh, nkv, d, V, G = (t["num_attention_heads"], t["num_key_value_heads"], t["head_dim"], t["vocab_size"],
                        t["num_code_groups"])
    Wt = L * (2 * H * nh * d + 2 * H * nkv * d + 3 * H * I) + H * V
    Hc, Ic, Lc = c["hidden_size"], c["intermediate_size"], c["num_hidden_layers"]
    Wc = Lc * (2 * Hc * c["num_attention_heads"] * c["head_dim"] + 2 * Hc * c["num_key_value_heads"] * c["head_dim"]
               + 3 * Hc * Ic)
    Wp = H * Hc if Hc != H else 0
    Wh = (G - 1) * c["vocab_size"] * Hc
    return dict(talker=Wt * bytes_per_elem, code_predictor=((G - 1) * (Wc + Wp) + Wh) * bytes_per_elem,
                total=(Wt + (G - 1) * (Wc + Wp) + Wh) * bytes_per_elem,
                kv_per_token=2 * L * nkv * d * bytes_per_elem)


def decode_bytes_per_frame(cfg, batch: int, ctx: float, bytes_per_elem: int = 2) -> float:
    """Algorithmic HBM bytes of one decode frame for the whole batch: every weight once, each row's KV cache read
    and appended once (talker at context `ctx`; code predictor contexts up to G), the G embedding rows per utterance."""
    t = cfg["talker_config"]
    c = t["code_predictor_config"]
    w = weight_bytes_per_frame(cfg, bytes_per_elem)
    G, H = t["num_code_groups"], t["hidden_size"]
    kv_t = w["kv_per_token"] * (ctx + 1)
    kv_c = c["num_hidden_layers"] * c["num_key_value_heads"] * c["head_dim"] * G * (G - 1) * bytes_per_elem
    return float(w["total"] + batch * (kv_t + kv_c + bytes_per_elem * G * H))


def decode_flops_per_frame(cfg, batch: int, ctx: float) -> float:
    """Algorithmic FLOPs of one decode frame for the whole batch: 2 per weight per row for every projection of the talker,
    the G-1 code-predictor passes and their heads, plus the talker's attention over `ctx` cached positions."""
    t = cfg["talker_config"]
    w = weight_bytes_per_frame(cfg, 1)
    attn = 4.0 * t["num_hidden_layers"] * t["num_attention_heads"] * t["head_dim"] * ctx
    return float(batch) * (2.0 * w["total"] + attn)


def codec_flops_per_frame(codec_cfg) -> float:
    """Algorithmic FLOPs (2 per multiply-add, GEMM/conv terms only) the codec decoder spends per 12.5 Hz frame."""
    c = codec_cfg["decoder_config"]
    Dc, Dl, Hd, Dd, I = c["codebook_dim"], c["latent_dim"], c["hidden_size"], c["decoder_dim"], c["intermediate_size"]
    nh, nkv = c["num_attention_heads"], c["num_key_value_heads"]
    d = c.get("head_dim") or Hd // nh
    f = 2.0 * 3 * Dc * Dl                                   # pre_conv k=3
    f += 2.0 * (Dl * Hd + Hd * Dl)                          # input / output projections
    per_layer = 2.0 * (Hd * (nh + 2 * nkv) * d + nh * d * Hd + 3 * Hd * I) + 4.0 * nh * d * min(c["sliding_window"], 72)
    f += c["num_hidden_layers"] * per_layer
    rate = 1
    for r in c["upsampling_ratios"]:
        f += rate * 2.0 * Dl * Dl * r                       # transposed conv, kernel == stride
        rate *= r
        f += rate * (2.0 * 7 * Dl + 2.0 * 2 * Dl * 4 * Dl)  # ConvNeXt: depthwise k=7 + two pointwise layers
    f += rate * 2.0 * 7 * Dl * Dd                           # decoder.0 k=7
    cur = Dd
    for r in c["upsample_rates"]:
        nxt = cur // 2
        f += rate * 2.0 * cur * nxt * 2 * r                 # transposed conv k = 2r (per input step)
        rate *= r
        f += rate * 3 * (2.0 * 7 * nxt * nxt + 2.0 * nxt * nxt)
        cur = nxt
    f += rate * 2.0 * 7 * cur
    return f


def iter_weights_torch(keys, seed: int, device="cuda", bf16: bool = False):
    """Random-init weights generated on the GPU with torch (benchmarks only: fast, but not the bit-exact
    counter-based stream of iter_weights).  Yields (key, tensor); matrices are bf16 when `bf16`."""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    for key, shape, kind in keys:
        k = kind.split(":")
        x = torch.randn(tuple(shape), generator=gen, device=device, dtype=torch.float32)
        if k[0] == "normal":
            x.mul_(float(k[1]))
        elif k[0] == "fanin":
            x.mul_(float(k[1]) / math.sqrt(float(np.prod(shape[1:]))))
        elif k[0] == "fanin_t":
            x.mul_(float(k[1]) / math.sqrt(shape[0] * shape[2] / float(k[2])))
        elif k[0] == "ones":
            x.mul_(float(k[1])).add_(1.0)
        elif k[0] == "const":
            x.mul_(float(k[2])).add_(float(k[1]))
        elif k[0] == "usage":
            x = x.abs().mul_(0.5).add_(1.0)
        if bf16 and len(shape) >= 2 and "_codebook" not in key and "dwconv" not in key:
            x = x.to(torch.bfloat16)
        yield key, x
