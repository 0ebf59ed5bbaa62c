"""WordPiece tokenizer of the sentence encoder: list[str] -> token ids, host side, integer work.

The reference tokenises inside the module: `SBertLang.encode` calls `self.model.tokenize(sentences_sorted)`
(hulc2/affordance/models/language_encoders/sbert_lang_encoder.py:45), i.e. sentence-transformers' `Transformer.tokenize`: strip each
sentence, then the checkpoint's BertTokenizer with `padding=True, truncation="longest_first", max_length=128`.  sentence-transformers
and the checkpoint (vocab.txt) are un-vendored, unpinned dependencies (requirements.txt:22) and there is no network here, so this file
restates the published BERT tokenisation algorithm — what `tokenizers`' BertNormalizer + BertPreTokenizer + WordPiece model do:

  normalise : drop NUL / U+FFFD / control characters (categories Cc, Cf, Cn, Co except \\t \\n \\r), every whitespace -> " ",
              spaces around CJK ideographs, NFD + drop combining marks (Mn) [strip_accents follows do_lower_case], lower-case
  pre-split : on whitespace, then every punctuation character (ASCII 33-47 58-64 91-96 123-126 or Unicode P*) is its own piece
  WordPiece : greedy longest-match-first over the vocabulary, continuation pieces prefixed "##", words > 100 characters or without a
              full cover -> [UNK]
  post      : [CLS] pieces [SEP], truncated to max_length, padded with [PAD] to the longest sentence of the batch

Parity: token ids are bit-exact against transformers' own BertTokenizer on tests/golden/wordpiece.npz (generated in the build container
by oracle/gen_golden.py on a synthetic vocabulary: the real vocab.txt is not available offline).  The vocabulary file is loaded from the
checkpoint directory (`vocab.txt`, one piece per line, line number = id).
"""
from __future__ import annotations

import os
import unicodedata
from typing import Dict, Iterable, List, Optional, Sequence

import torch

_SPECIALS = ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")
# Unicode White_Space beyond \t \n \r and " " (what Rust's char::is_whitespace accepts)
_WS = {0x0B, 0x0C, 0x85, 0xA0, 0x1680, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000, *range(0x2000, 0x200B)}


def _is_whitespace(ch: str) -> bool:
    return ch in " \t\n\r" or ord(ch) in _WS


def _is_control(ch: str) -> bool:
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch) in ("Cc", "Cf", "Cn", "Co")


def _is_punctuation(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or
            0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class WordPieceTokenizer:
    """BertTokenizer(do_lower_case=True) semantics; callable on a list of sentences -> {'input_ids', 'token_type_ids', 'attention_mask'}"""

    def __init__(self, vocab: Dict[str, int] | Sequence[str], do_lower_case: bool = True, max_length: int = 128,
                 max_input_chars_per_word: int = 100):
        if not isinstance(vocab, dict):
            vocab = {tok: i for i, tok in enumerate(vocab)}
        for s in _SPECIALS[:4]:
            if s not in vocab:
                raise ValueError(f"vocabulary has no {s} entry (not a BERT WordPiece vocab.txt)")
        self.vocab = vocab
        self.do_lower_case, self.max_length, self.max_chars = do_lower_case, max_length, max_input_chars_per_word
        self.pad_id, self.unk_id, self.cls_id, self.sep_id = (vocab[s] for s in _SPECIALS[:4])
        self._specials = [s for s in _SPECIALS if s in vocab]

    @classmethod
    def from_vocab_file(cls, path: str, **kw) -> "WordPieceTokenizer":
        with open(path, encoding="utf-8") as f:
            toks = [line.rstrip("\n") for line in f]
        return cls({t: i for i, t in enumerate(toks)}, **kw)     # a repeated piece keeps its LAST line number, as BertTokenizer's load_vocab does

    @classmethod
    def from_checkpoint_dir(cls, root: str, **kw) -> "WordPieceTokenizer":
        """the sentence-transformers checkpoint layout: vocab.txt at the top level (or under 0_Transformer/ in old releases)"""
        for rel in ("vocab.txt", os.path.join("0_Transformer", "vocab.txt")):
            p = os.path.join(root, rel)
            if os.path.isfile(p):
                return cls.from_vocab_file(p, **kw)
        raise FileNotFoundError(f"no vocab.txt under {root}")

    # ---- text -> pieces ------------------------------------------------------------------------------------------------------
    def _normalize(self, text: str) -> str:
        out = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            out.append(" " if _is_whitespace(ch) else ch)
        text = "".join(out)
        out = []
        for ch in text:
            if _is_cjk(ord(ch)):
                out += [" ", ch, " "]
            else:
                out.append(ch)
        text = "".join(out)
        if self.do_lower_case:                                  # strip_accents=None follows do_lower_case
            text = "".join(ch for ch in unicodedata.normalize("NFD", text) if unicodedata.category(ch) != "Mn")
            text = text.lower()
        return text

    @staticmethod
    def _pre_split(text: str) -> List[str]:
        words, cur = [], []
        for ch in text:
            if _is_whitespace(ch):
                if cur:
                    words.append("".join(cur))
                    cur = []
            elif _is_punctuation(ch):
                if cur:
                    words.append("".join(cur))
                    cur = []
                words.append(ch)
            else:
                cur.append(ch)
        if cur:
            words.append("".join(cur))
        return words

    def _wordpiece(self, word: str) -> List[int]:
        if len(word) > self.max_chars:
            return [self.unk_id]
        ids, start, n = [], 0, len(word)
        while start < n:
            end, hit = n, None
            while start < end:
                piece = word[start:end] if start == 0 else "##" + word[start:end]
                hit = self.vocab.get(piece)
                if hit is not None:
                    break
                end -= 1
            if hit is None:
                return [self.unk_id]
            ids.append(hit)
            start = end
        return ids

    def _split_specials(self, text: str) -> Iterable[tuple]:
        """special tokens written out in the text are matched before normalisation and kept whole (the added-tokens pass)"""
        i, n = 0, len(text)
        while i < n:
            nxt, which = n, None
            for s in self._specials:
                j = text.find(s, i)
                if j != -1 and j < nxt:
                    nxt, which = j, s
            if nxt > i:
                yield text[i:nxt], False
            if which is None:
                break
            yield which, True
            i = nxt + len(which)

    def encode_one(self, sentence: str) -> List[int]:
        """pieces of one sentence without [CLS] / [SEP]"""
        ids: List[int] = []
        for chunk, special in self._split_specials(sentence):
            if special:
                ids.append(self.vocab[chunk])
            else:
                for w in self._pre_split(self._normalize(chunk)):
                    ids += self._wordpiece(w)
        return ids

    def __call__(self, sentences: Sequence[str]) -> Dict[str, torch.Tensor]:
        rows = []
        for s in sentences:
            ids = self.encode_one(str(s).strip())[: self.max_length - 2]      # sentence-transformers strips; truncation keeps the head
            rows.append([self.cls_id] + ids + [self.sep_id])
        L = max((len(r) for r in rows), default=0)
        ids = torch.full((len(rows), L), self.pad_id, dtype=torch.int64)
        mask = torch.zeros(len(rows), L, dtype=torch.int64)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r, dtype=torch.int64)
            mask[i, :len(r)] = 1
        return {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask}


def find_checkpoint_dir(nlp_model: str) -> Optional[str]:
    """where the sentence encoder's files live: `nlp_model` itself when it is a directory, $HULC2_SBERT_DIR, or the sentence-transformers /
    huggingface cache entries of that model name — never a download"""
    cands = [nlp_model, os.environ.get("HULC2_SBERT_DIR", "")]
    home = os.path.expanduser("~")
    cands += [os.path.join(home, ".cache", "torch", "sentence_transformers", f"sentence-transformers_{nlp_model}")]
    hub = os.path.join(os.environ.get("HF_HOME", os.path.join(home, ".cache", "huggingface")), "hub",
                       f"models--sentence-transformers--{nlp_model}", "snapshots")
    if os.path.isdir(hub):
        cands += [os.path.join(hub, d) for d in sorted(os.listdir(hub))]
    for c in cands:
        if c and os.path.isdir(c) and any(os.path.isfile(os.path.join(c, r)) for r in ("vocab.txt", os.path.join("0_Transformer", "vocab.txt"))):
            return c
    return None
