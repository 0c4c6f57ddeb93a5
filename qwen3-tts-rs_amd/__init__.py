"""MI355X-native Qwen3-TTS inference hot path: HIP kernels behind a C ABI (csrc/, include/qtts.h) plus the
host-side mirror of the reference's model API.  Importable as `qwen3_tts_rs_amd`."""
from .model import (CodecStream, GenerateConfig, GenerationRequest, GenerationResult, LoaderConfig, Model, ModelLoader, Output,
                    SamplingConfig, VoiceClonePromptItem, round_bf16, synthetic_model)
from ._lib import QttsError, QTTS_BF16, QTTS_F32
from . import io

__all__ = ["GenerateConfig", "GenerationRequest", "GenerationResult", "LoaderConfig", "Model", "ModelLoader", "Output",
           "SamplingConfig", "VoiceClonePromptItem", "round_bf16", "synthetic_model", "QttsError", "QTTS_BF16", "QTTS_F32"]
