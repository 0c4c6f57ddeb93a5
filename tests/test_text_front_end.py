"""Host-side text front end: the known-answer vectors of the reference's own unit tests
(qwen_tts/src/text/processing.rs:800-960) plus a round trip through a small byte-level BPE built here."""
import json
import os

import pytest

from qwen3_tts_rs_amd.text import TTS_BOS, TextProcessor, normalize_text


def test_normalize_text_reference_vectors():
    assert normalize_text("it’s") == "it's"
    assert normalize_text("‘quoted’") == "'quoted'"
    assert normalize_text("“Hello”") == '"Hello"'
    assert normalize_text("wait—what") == "wait--what"
    assert normalize_text("1–2") == "1-2"
    assert normalize_text("wait…") == "wait..."
    assert normalize_text("hello world") == "hello world"
    assert normalize_text("hello world") == "hello world"
    assert normalize_text("hello　world") == "hello world"
    assert normalize_text("a\r\nb") == "a\nb"
    assert normalize_text("a\rb") == "a\nb"
    for zw in ("​", "﻿", "‌", "‍"):
        assert normalize_text(f"hel{zw}lo") == "hello"
    assert normalize_text("He said, “It’s—well…”") == 'He said, "It\'s--well..."'
    assert normalize_text("Hello, world!") == "Hello, world!"


def test_chat_templates_and_special_tokens():
    p = TextProcessor()
    assert p.build_assistant_text("Hello world") == "<|im_start|>assistant\nHello world<|im_end|>\n<|im_start|>assistant\n"
    assert p.build_ref_text("Reference") == "<|im_start|>assistant\nReference<|im_end|>\n"
    assert p.build_instruct_text("Speak slowly") == "<|im_start|>user\nSpeak slowly<|im_end|>\n"
    assert TTS_BOS == 151672 and p.add_tts_tokens([1, 2, 3]) == [151672, 1, 2, 3]


def test_without_tokenizer():
    p = TextProcessor()
    assert not p.has_tokenizer() and p.tokenize("Hello") == [] and p.batch_tokenize(["Hello", "World"]) == [[], []]
    with pytest.raises(ValueError):
        p.try_batch_tokenize(["Hello", "World"])
    with pytest.raises(ValueError):
        p.batch_tokenize_for_tts(["Hello"])


def test_pad_sequences_reference_vectors():
    out = TextProcessor.pad_sequences([[1, 2], [3, 4, 5, 6], [7]], "left", 0)
    assert out.input_ids == [[0, 0, 1, 2], [3, 4, 5, 6], [0, 0, 0, 7]]
    assert out.attention_mask == [[0, 0, 1, 1], [1, 1, 1, 1], [0, 0, 0, 1]] and out.lengths == [2, 4, 1]
    out = TextProcessor.pad_sequences([[1, 2], [3, 4, 5]], "right", 99)
    assert out.input_ids == [[1, 2, 99], [3, 4, 5]] and out.attention_mask == [[1, 1, 0], [1, 1, 1]]
    out = TextProcessor.pad_sequences([], "left", 0)
    assert out.input_ids == [] and out.attention_mask == [] and out.lengths == []
    out = TextProcessor.pad_sequences([[1, 2, 3]], "left", 0)
    assert out.input_ids == [[1, 2, 3]] and out.attention_mask == [[1, 1, 1]] and out.lengths == [3]


def _toy_tokenizer(path):
    """A tiny byte-level BPE with the chat markers as special tokens, written in the tokenizer.json format."""
    from tokenizers import AddedToken, Tokenizer, decoders, models, pre_tokenizers, trainers
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    tr = trainers.BpeTrainer(vocab_size=400, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tok.train_from_iterator(["assistant user hello world speak slowly it's a test"] * 4, tr)
    tok.add_special_tokens([AddedToken("<|im_start|>", special=True), AddedToken("<|im_end|>", special=True)])
    tok.save(path)
    return tok


def test_tokenize_for_tts_with_a_tokenizer_file(tmp_path):
    path = str(tmp_path / "tokenizer.json")
    tok = _toy_tokenizer(path)
    p = TextProcessor.from_pretrained(str(tmp_path))
    a, b = tok.token_to_id("<|im_start|>"), tok.token_to_id("<|im_end|>")
    ids = p.tokenize_for_tts("hello world")
    # template shape the prompt builder relies on: 3 prefix tokens ... 5 suffix tokens
    assert ids[0] == a and ids[-5] == b and ids[-3] == a and ids[-2:] == ids[1:3]
    assert p.decode(ids[3:-5]) == "hello world"
    # normalisation happens before tokenisation
    assert p.tokenize("it’s") == p.tokenize("it's")
    out = p.batch_tokenize_for_tts(["hello", "hello world speak slowly"])
    assert out.lengths[0] < out.lengths[1] and len(out.input_ids[0]) == len(out.input_ids[1])
    assert out.input_ids[0][0] == 151671 and out.attention_mask[0][0] == 0      # left padding with tts_pad
    with pytest.raises(FileNotFoundError):
        TextProcessor.from_pretrained(str(tmp_path / "missing"))
