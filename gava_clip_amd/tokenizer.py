"""CLIP byte-pair tokenizer (host side, init-time only; integer output).

Behavioural mirror of /root/reference/training/VitaCLIP_text_encoder_utils.py:62-132
(``SimpleTokenizer``) and ``tokenize`` (/root/reference/training/VitaCLIP_text_encoder.py:27-65),
re-implemented around a rank-heap-free greedy merge.  The merge table is the public
OpenAI CLIP vocabulary file shipped as data under ``gava_clip_amd/data``.

``ftfy`` (used upstream only inside ``basic_clean``) is not installed in this image; it is
the identity on the ASCII class lists under ``data/classes``.  For other text we fall
back to NFC normalisation, which is what ftfy does last; mojibake repair is not done.
Token ids for both class lists are pinned by tests/golden/tokens_*.json.
"""
import gzip
import html
import os
import unicodedata
from functools import lru_cache

import numpy as np
import regex

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
SOT, EOT = "<|startoftext|>", "<|endoftext|>"


@lru_cache()
def _byte_alphabet():
    """Printable stand-in character for every byte value (GPT-2 convention)."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    table, extra = {}, 0
    for b in keep:
        table[b] = chr(b)
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    # vocabulary order is: kept bytes in ``keep`` order, then the remapped ones
    order = keep + [b for b in range(256) if b not in set(keep)]
    return table, [table[b] for b in order]


class ClipBPE:
    def __init__(self, vocab_path: str = None):
        vocab_path = vocab_path or os.path.join(_DATA, "bpe_simple_vocab_16e6.txt.gz")
        with gzip.open(vocab_path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        n_merges = 49152 - 256 - 2
        merges = [tuple(l.split()) for l in lines[1:1 + n_merges]]
        self.byte_map, alphabet = _byte_alphabet()
        vocab = alphabet + [c + "</w>" for c in alphabet] + ["".join(m) for m in merges] + [SOT, EOT]
        self.token_id = {tok: i for i, tok in enumerate(vocab)}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.splitter = regex.compile(
            r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
            regex.IGNORECASE)
        self._memo = {SOT: [SOT], EOT: [EOT]}

    def _merge_word(self, chars: str):
        if chars in self._memo:
            return self._memo[chars]
        parts = list(chars[:-1]) + [chars[-1] + "</w>"]
        while len(parts) > 1:
            best, best_rank = None, None
            for a, b in zip(parts, parts[1:]):
                r = self.rank.get((a, b))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (a, b), r
            if best is None:
                break
            merged, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == best[0] and parts[i + 1] == best[1]:
                    merged.append(parts[i] + parts[i + 1])
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        self._memo[chars] = parts
        return parts

    @staticmethod
    def _clean(text: str) -> str:
        if not text.isascii():
            text = unicodedata.normalize("NFC", text)
        text = html.unescape(html.unescape(text)).strip()
        return regex.sub(r"\s+", " ", text).strip().lower()

    def encode(self, text: str):
        ids = []
        for piece in self.splitter.findall(self._clean(text)):
            chars = "".join(self.byte_map[b] for b in piece.encode("utf-8"))
            ids.extend(self.token_id[t] for t in self._merge_word(chars))
        return ids


@lru_cache()
def _default_bpe():
    return ClipBPE()


def tokenize(texts, context_length: int = 77, truncate: bool = False) -> np.ndarray:
    """(len(texts), context_length) int32: [SOT] + bpe + [EOT], zero padded.  Raises
    RuntimeError when a text does not fit, like the reference (text_encoder.py:58-63)."""
    if isinstance(texts, str):
        texts = [texts]
    bpe = _default_bpe()
    sot, eot = bpe.token_id[SOT], bpe.token_id[EOT]
    out = np.zeros((len(texts), context_length), dtype=np.int32)
    for i, t in enumerate(texts):
        ids = [sot] + bpe.encode(t) + [eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {t} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = eot
        out[i, :len(ids)] = ids
    return out


def read_class_names(path: str):
    """Class phrases of a class-list file; lines starting with '*' are display labels and
    are dropped (/root/reference/training/VitaCLIP_model.py:203-205)."""
    with open(path, "r") as f:
        rows = f.read().strip().split("\n")
    return [r for r in rows if r[0] != "*"]


def prompt_texts(classnames, n_ctx: int):
    """'X X ... X <name>.' (/root/reference/training/VitaCLIP_text_encoder.py:239,264)."""
    prefix = " ".join(["X"] * n_ctx)
    return [prefix + " " + c.replace("_", " ") + "." for c in classnames]
