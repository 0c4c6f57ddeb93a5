"""qtts_mel_noalloc (the entry point that allocates no device memory: staging in a fixed per-GPU array) against qtts_mel,
through the C ABI: two calls in a row, the second with a smaller input, must equal the allocating entry point bit for bit
(same kernel, same inputs); a batch beyond the array's capacity is rejected with a clear error."""
import numpy as np
import pytest

from golden.make_golden import test_audio as make_audio

pytestmark = pytest.mark.gpu


def test_no_alloc_mel_equals_the_allocating_entry_point():
    from qwen3_tts_rs_amd import QttsError, synthetic_model
    m, cfg, ccfg, tmpl = synthetic_model("T0", dtype="f32", codec=False, model_type="base")
    try:
        big = [make_audio(48000 + 777 * i, i) for i in range(5)]      # 2.0 - 2.1 s clips
        small = [make_audio(9000 + 131 * i, 7 + i) for i in range(2)]  # shorter clips, smaller batch
        for clips in (big, small):
            ref = m.compute_mel_spectrogram(clips)
            got = m.compute_mel_spectrogram(clips, no_alloc=True)
            assert len(ref) == len(got)
            for a, b in zip(ref, got):
                assert a.shape == b.shape and np.array_equal(a, b)
        # no device allocation behind the call: the free-memory figure of the device does not move across it
        import torch
        m.compute_mel_spectrogram(small, no_alloc=True)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        m.compute_mel_spectrogram(big, no_alloc=True)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0
        # the first call's larger result must not leak into the second call's rows
        again = m.compute_mel_spectrogram(small, no_alloc=True)
        for a, b in zip(m.compute_mel_spectrogram(small), again):
            assert np.array_equal(a, b)
        cap = m.lib.qtts_mel_noalloc_capacity()
        assert cap == 16 << 20
        too_many = [make_audio(240000, i) for i in range(14)]          # 14 x 10 s: 13.4 MB of clips + 6.7 MB of mel
        with pytest.raises(QttsError, match="scratch"):
            m.compute_mel_spectrogram(too_many, no_alloc=True)
        assert len(m.compute_mel_spectrogram(too_many)) == 14          # the allocating entry point still takes it
    finally:
        m.close()
