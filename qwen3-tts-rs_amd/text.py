"""Host-side text front end in front of the hot path: the reference's TextProcessor (qwen_tts/src/text/processing.rs).
Nothing here runs on the GPU; it turns strings into the token ids `Model.generate*` takes.

The byte-level BPE itself is the `tokenizers` library reading the model directory's tokenizer.json (the reference uses the
Rust crate of the same project); without that file `tokenize` returns [] and `try_tokenize` raises, as in the reference."""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

IM_START, IM_END, TTS_PAD, TTS_BOS, TTS_EOS = 151644, 151645, 151671, 151672, 151673  # processing.rs:88-97


def normalize_text(text: str) -> str:
    """processing.rs:56-71: typographic quotes / dashes / ellipsis to ASCII, Unicode spaces to ' ', CR(LF) to LF,
    zero-width characters removed - in that order."""
    for a, b in (("’", "'"), ("‘", "'"), ("“", '"'), ("”", '"'), ("—", "--"), ("–", "-"),
                 ("…", "..."), (" ", " "), (" ", " "), ("　", " "), ("\r\n", "\n"), ("\r", "\n"),
                 ("​", ""), ("‌", ""), ("‍", ""), ("﻿", "")):
        text = text.replace(a, b)
    return text


@dataclass
class TokenizerOutput:  # processing.rs:101-109
    input_ids: List[List[int]] = field(default_factory=list)
    attention_mask: List[List[int]] = field(default_factory=list)
    lengths: List[int] = field(default_factory=list)


class TextProcessor:
    def __init__(self, tokenizer=None):
        self.tokenizer = tokenizer

    @classmethod
    def from_file(cls, path: str) -> "TextProcessor":  # processing.rs:401-409
        from tokenizers import Tokenizer
        return cls(Tokenizer.from_file(path))

    @classmethod
    def from_pretrained(cls, model_dir: str) -> "TextProcessor":  # processing.rs:172-214 (tokenizer.json only)
        p = os.path.join(model_dir, "tokenizer.json")
        if not os.path.exists(p):
            raise FileNotFoundError(f"No tokenizer found in {model_dir}. Expected tokenizer.json")
        return cls.from_file(p)

    def has_tokenizer(self) -> bool:
        return self.tokenizer is not None

    # chat templates (processing.rs:427-454)
    @staticmethod
    def build_assistant_text(text: str) -> str:
        return f"<|im_start|>assistant\n{text}<|im_end|>\n<|im_start|>assistant\n"

    @staticmethod
    def build_ref_text(text: str) -> str:
        return f"<|im_start|>assistant\n{text}<|im_end|>\n"

    @staticmethod
    def build_instruct_text(instruct: str) -> str:
        return f"<|im_start|>user\n{instruct}<|im_end|>\n"

    def tokenize(self, text: str) -> List[int]:  # processing.rs:468-493
        if self.tokenizer is None:
            return []
        return list(self.tokenizer.encode(normalize_text(text), add_special_tokens=False).ids)

    def try_tokenize(self, text: str) -> List[int]:  # processing.rs:505-519
        if self.tokenizer is None:
            raise ValueError("No tokenizer loaded")
        return self.tokenize(text)

    def tokenize_for_tts(self, text: str) -> List[int]:
        return self.tokenize(self.build_assistant_text(text))

    def tokenize_ref_text(self, text: str) -> List[int]:
        return self.tokenize(self.build_ref_text(text))

    def tokenize_instruct(self, instruct: str) -> List[int]:
        return self.tokenize(self.build_instruct_text(instruct))

    @staticmethod
    def add_tts_tokens(token_ids: Sequence[int]) -> List[int]:  # processing.rs:587-591
        return [TTS_BOS] + list(token_ids)

    def batch_tokenize(self, texts: Sequence[str]) -> List[List[int]]:
        return [self.tokenize(t) for t in texts]

    def try_batch_tokenize(self, texts: Sequence[str]) -> List[List[int]]:
        return [self.try_tokenize(t) for t in texts]

    def batch_tokenize_for_tts(self, texts: Sequence[str], padding_side: str = "left") -> TokenizerOutput:  # processing.rs:692-705
        if self.tokenizer is None:
            raise ValueError("No tokenizer loaded")
        return self.pad_sequences([self.tokenize_for_tts(t) for t in texts], padding_side, TTS_PAD)

    @staticmethod
    def pad_sequences(sequences: Sequence[Sequence[int]], padding_side: str = "left", pad_token_id: int = 0) -> TokenizerOutput:
        """processing.rs:711-766: pad to the longest sequence; mask 1 = real token."""
        out = TokenizerOutput()
        if not sequences:
            return out
        out.lengths = [len(s) for s in sequences]
        mx = max(out.lengths)
        for s in sequences:
            pad = mx - len(s)
            if padding_side == "left":
                out.input_ids.append([pad_token_id] * pad + list(s))
                out.attention_mask.append([0] * pad + [1] * len(s))
            else:
                out.input_ids.append(list(s) + [pad_token_id] * pad)
                out.attention_mask.append([1] * len(s) + [0] * pad)
        return out

    def decode(self, token_ids: Sequence[int]) -> Optional[str]:
        if self.tokenizer is None:
            return None
        return self.tokenizer.decode(list(token_ids), skip_special_tokens=True)
