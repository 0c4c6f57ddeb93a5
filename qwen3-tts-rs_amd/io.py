"""On-disk formats of the reference that sit next to the hot path, byte-compatible so that files move between the two
implementations: voice-clone prompt files (qwen_tts/src/io/voice_prompt.rs:12-151), audio-codes files
(qwen_tts/src/synthesis/tokenizer.rs:29-39,133-192) and the 16-bit mono WAV writer (io/voice_prompt.rs:154-191)."""
from __future__ import annotations

import base64
import json
import struct
from typing import Optional

import numpy as np

from ._lib import QttsError
from .model import VoiceClonePromptItem

VOICE_PROMPT_VERSION = 1
CODES_FILE_VERSION = 1


def save_voice_prompt(prompt: VoiceClonePromptItem, path: str, x_vector_only: bool, ref_text: Optional[str] = None):
    """JSON {version, x_vector_only, ref_text, prompt_data}; prompt_data = base64(safetensors{ref_spk_embedding f32
    [, ref_code i64 (codes_len, num_quantizers)]})."""
    from safetensors.numpy import save
    tensors = {"ref_spk_embedding": np.ascontiguousarray(prompt.ref_spk_embedding, np.float32)}
    if prompt.ref_code is not None:
        tensors["ref_code"] = np.ascontiguousarray(prompt.ref_code).astype(np.int64)
    doc = dict(version=VOICE_PROMPT_VERSION, x_vector_only=bool(x_vector_only), ref_text=ref_text,
               prompt_data=base64.b64encode(save(tensors)).decode("ascii"))
    with open(path, "w") as f:
        json.dump(doc, f, indent=2)


def load_voice_prompt(path: str, ref_text_ids=None) -> VoiceClonePromptItem:
    """Inverse of save_voice_prompt.  ICL mode needs ref_code, ref_text and x_vector_only == false (voice_prompt.rs:141-143);
    `ref_text_ids` is the tokenised transcript (the tokenizer stays outside this package)."""
    from safetensors.numpy import load
    with open(path) as f:
        doc = json.load(f)
    if doc.get("version") != VOICE_PROMPT_VERSION:
        raise QttsError(-5, f"Voice prompt file version mismatch: expected {VOICE_PROMPT_VERSION}, got {doc.get('version')}")
    try:
        tensors = load(base64.b64decode(doc["prompt_data"], validate=True))
    except Exception as e:
        raise QttsError(-5, f"Failed to decode voice prompt data: {e}")
    if "ref_spk_embedding" not in tensors:
        raise QttsError(-5, "Voice prompt file missing ref_spk_embedding")
    code = tensors.get("ref_code")
    icl = code is not None and doc.get("ref_text") is not None and not doc.get("x_vector_only", False)
    item = VoiceClonePromptItem(ref_spk_embedding=tensors["ref_spk_embedding"].astype(np.float32),
                                ref_code=None if code is None else code.astype(np.uint32),
                                x_vector_only_mode=not icl if doc.get("x_vector_only") is None else bool(doc["x_vector_only"]),
                                ref_text_ids=ref_text_ids if icl else None)
    item.ref_text = doc.get("ref_text")
    return item


def save_codes(codes: np.ndarray, path: str, sample_rate: int = 24000):
    """AudioCodesFile: {version, sample_rate, num_codebooks, num_steps, codes[steps][codebooks]}."""
    c = np.asarray(codes)
    if c.ndim != 2:
        raise QttsError(-4, "codes must be (num_steps, num_codebooks)")
    doc = dict(version=CODES_FILE_VERSION, sample_rate=int(sample_rate), num_codebooks=int(c.shape[1]), num_steps=int(c.shape[0]),
               codes=[[int(v) for v in row] for row in c])
    with open(path, "w") as f:
        json.dump(doc, f, indent=2)


def load_codes(path: str):
    with open(path) as f:
        doc = json.load(f)
    if doc.get("version") != CODES_FILE_VERSION:
        raise QttsError(-5, f"Unsupported codes file version: {doc.get('version')}")
    c = np.asarray(doc["codes"], np.int64).reshape(doc["num_steps"], doc["num_codebooks"])
    return c.astype(np.uint32), int(doc["sample_rate"])


def write_wav(path: str, audio, sample_rate: int):
    """Mono 16-bit PCM: clamp to [-1, 1], then truncate (x * 32767) toward zero like Rust's `as i16`."""
    x = np.asarray(audio, np.float32).reshape(-1)
    if x.size == 0:
        raise QttsError(-5, "No audio was generated. The model produced 0 samples.")
    pcm = np.trunc(np.clip(x, -1.0, 1.0) * np.float32(32767.0)).astype("<i2")
    data = pcm.tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, int(sample_rate), int(sample_rate) * 2, 2, 16))
        f.write(b"data" + struct.pack("<I", len(data)) + data)
