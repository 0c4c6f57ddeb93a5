#!/usr/bin/env python3
"""Golden waveforms for the codec DECODER from an independent public implementation of the same architecture.

The reference's 12.5 Hz decoder (qwen_tts/src/audio/decoder/v2.rs) has the layout of Qwen3-Omni's Code2Wav: a sliding-window
transformer with LayerScale, transposed-conv + ConvNeXt upsampling, and a SnakeBeta vocoder, under the same weight keys.
`transformers` (installed in the build container) ships that model (Qwen3OmniMoeCode2Wav).  Its transformer, upsample and
vocoder modules are loaded with the synthetic T0 codec weights; the front end the two models do not share (split RVQ
dequantisation, pre_conv, input / output projections) is written here with plain torch ops from audio/quantizer.rs:56-105,
290-318, 417-434 and audio/decoder/v2.rs:163-214.  One place follows the reference instead of Code2Wav: ConvNeXt uses the
tanh GELU (Candle's Tensor::gelu, audio/tokenizer/v2/convnext.rs:81), Code2Wav the erf one.

    python tests/golden/make_code2wav_golden.py          # writes tests/golden/code2wav_golden.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
SEED = 20261111


def main():
    import torch
    import torch.nn.functional as F
    from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import Qwen3OmniMoeCode2WavConfig
    from transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe import Qwen3OmniMoeCode2Wav
    from qwen3_tts_rs_amd import synth
    cfg, ccfg, tmpl = synth.shape_set("T0")
    d = ccfg["decoder_config"]
    W = {k: torch.from_numpy(a) for k, a in synth.iter_weights(synth.codec_keys(ccfg), SEED)}

    def hf(hidden):
        c = Qwen3OmniMoeCode2WavConfig(
            codebook_size=d["codebook_size"], hidden_size=hidden, max_position_embeddings=d["max_position_embeddings"],
            rope_parameters={"rope_theta": d["rope_theta"], "rope_type": "default"}, num_attention_heads=d["num_attention_heads"],
            num_key_value_heads=d["num_key_value_heads"], attention_bias=d["attention_bias"], sliding_window=d["sliding_window"],
            intermediate_size=d["intermediate_size"], hidden_act=d["hidden_act"], rms_norm_eps=d["rms_norm_eps"],
            num_hidden_layers=d["num_hidden_layers"], num_quantizers=d["num_quantizers"], upsample_rates=d["upsample_rates"],
            upsampling_ratios=d["upsampling_ratios"], decoder_dim=d["decoder_dim"])
        c._attn_implementation = "eager"
        return Qwen3OmniMoeCode2Wav(c).eval()

    def load(module, prefix):
        sd = module.state_dict()
        n = 0
        for k in sd:
            src = W[prefix + k]
            assert tuple(src.shape) == tuple(sd[k].shape), (k, src.shape, sd[k].shape)
            sd[k].copy_(src)
            n += 1
        module.load_state_dict(sd)
        return n

    torch.manual_seed(0)
    xf = hf(d["hidden_size"]).pre_transformer          # transformer at its own width
    voc = hf(d["latent_dim"])                          # upsample + vocoder at the latent width
    n = load(xf, "decoder.pre_transformer.") + load(voc.upsample, "decoder.upsample.") + load(voc.decoder, "decoder.decoder.")
    for blocks in voc.upsample:
        blocks[1].act = torch.nn.GELU(approximate="tanh")

    def codebook(prefix):
        return W[prefix + "embedding_sum"] / W[prefix + "cluster_usage"].clamp(min=1e-5)[:, None]

    def dequant(codes):  # codes (T, Q) -> (1, codebook_dim, T)
        first = codebook("decoder.quantizer.rvq_first.vq.layers.0._codebook.")[codes[:, 0]]
        rest = sum(codebook(f"decoder.quantizer.rvq_rest.vq.layers.{q - 1}._codebook.")[codes[:, q]] for q in range(1, codes.shape[1]))
        a = F.conv1d(first.T[None], W["decoder.quantizer.rvq_first.output_proj.weight"])
        b = F.conv1d(rest.T[None], W["decoder.quantizer.rvq_rest.output_proj.weight"])
        return a + b

    def decode(codes):
        with torch.no_grad():
            x = dequant(torch.from_numpy(codes.astype(np.int64)))
            x = F.conv1d(F.pad(x, (2, 0)), W["decoder.pre_conv.conv.weight"], W["decoder.pre_conv.conv.bias"])
            h = F.linear(x.transpose(1, 2), W["decoder.pre_transformer.input_proj.weight"], W["decoder.pre_transformer.input_proj.bias"])
            h = xf(inputs_embeds=h).last_hidden_state
            h = F.linear(h, W["decoder.pre_transformer.output_proj.weight"], W["decoder.pre_transformer.output_proj.bias"])
            y = h.permute(0, 2, 1)
            for blocks in voc.upsample:
                for blk in blocks:
                    y = blk(y)
            for blk in voc.decoder:
                y = blk(y)
            return y.clamp(min=-1, max=1)[0, 0].numpy()

    rng = np.random.default_rng(11)
    out = {"loaded_tensors": np.int64(n)}
    for name, T in (("a", 24), ("b", 77)):
        codes = rng.integers(0, d["codebook_size"], (T, d["num_quantizers"])).astype(np.uint32)
        out["codes_" + name] = codes
        out["wav_" + name] = decode(codes).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "code2wav_golden.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
