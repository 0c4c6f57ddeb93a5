#!/usr/bin/env python3
"""Golden vectors for the codec ENCODER from an independent public implementation of the same architecture.

The reference's encoder (qwen_tts/src/audio/encoder/*.rs) re-implements the Mimi codec encoder and reads Mimi's weight keys
under the prefix `encoder.`.  `transformers.MimiModel` (installed in the build container, default MimiConfig = the
12.5 Hz / 2048 x 256 / 32-codebook shape the reference hard-codes) is instantiated with the synthetic encoder weights of
synth.encoder_keys(), and its codes for two seeded clips are stored.  The oracle (oracle/oracle_encoder.cpp) is then pinned
against them in tests/test_oracle_golden.py.  Clips are shorter than Mimi's 250-frame attention window, inside which its
sliding-window attention equals the reference's full causal attention.  Two places where the reference departs from Mimi are
followed here: the downsample conv is zero-padded (Mimi: replicate), and clips are whole frames (Mimi pads a partial last
frame, the reference drops it).

    python tests/golden/make_mimi_golden.py          # writes tests/golden/mimi_encoder_golden.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
SEED = 20261111


def main():
    import torch
    from transformers import MimiConfig, MimiModel
    from golden.make_golden import test_audio
    from qwen3_tts_rs_amd import synth
    torch.manual_seed(0)
    model = MimiModel(MimiConfig()).eval()
    sd = model.state_dict()
    for k, a in synth.iter_weights(synth.encoder_keys(), SEED):
        kk = k[len("encoder."):]
        assert tuple(sd[kk].shape) == a.shape, kk
        sd[kk].copy_(torch.from_numpy(a))
    model.load_state_dict(sd)
    # the reference zero-pads the stride-2 downsample conv (audio/encoder/seanet.rs:244-248); Mimi replicates the first step
    assert model.downsample.pad_mode == "replicate"
    model.downsample.pad_mode = "constant"
    out = {}
    for name, n, seed in (("a", 48000, 1), ("b", 1920 * 9, 2)):  # whole frames: the reference, unlike Mimi, does not pad a partial last frame
        clip = test_audio(n, seed)
        with torch.no_grad():
            codes = model.encode(torch.from_numpy(clip)[None, None], num_quantizers=16).audio_codes[0].T.numpy()
        out["clip_" + name + "_len"] = np.int64(n)
        out["clip_" + name + "_seed"] = np.int64(seed)
        out["codes_" + name] = codes.astype(np.uint32)
    np.savez_compressed(os.path.join(HERE, "mimi_encoder_golden.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
