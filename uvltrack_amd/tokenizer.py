"""WordPiece tokenisation of the language query.  The reference tracker builds
`BertTokenizer.from_pretrained(VOCAB_PATH, do_lower_case=True)` from the third-party package pytorch-pretrained-bert 0.6.2
(lib/test/tracker/uvltrack.py:16,40; uvltrack_env.yaml:302), which is not vendored in the reference tree, and turns a
sentence into (ids, mask) in `extract_token_from_nlp` (lib/test/tracker/uvltrack.py:196-233).  This module restates the
published BERT algorithm (BasicTokenizer: clean, CJK spacing, lower-case + accent stripping, punctuation split; then greedy
longest-match-first WordPiece with the "##" continuation prefix) and that function.  Host-side string work: no GPU involved.
tests/test_tokenizer.py pins it against `transformers.BertTokenizer` (same lineage, present in the image) on a synthetic
vocabulary; the real `bert-base-uncased` vocab.txt is a data file the user supplies (VOCAB_PATH), as for the reference.
"""
from __future__ import annotations

import collections
import os
import unicodedata
from typing import List, Sequence, Tuple

NEVER_SPLIT = ("[UNK]", "[SEP]", "[PAD]", "[CLS]", "[MASK]")


def load_vocab(vocab_file: str) -> "collections.OrderedDict[str, int]":
    vocab = collections.OrderedDict()
    with open(vocab_file, "r", encoding="utf-8") as reader:
        index = 0
        while True:
            token = reader.readline()
            if not token:
                break
            vocab[token.strip()] = index
            index += 1
    return vocab


def whitespace_tokenize(text: str) -> List[str]:
    text = text.strip()
    return text.split() if text else []


def _is_whitespace(ch: str) -> bool:
    if ch in (" ", "\t", "\n", "\r"):
        return True
    return unicodedata.category(ch) == "Zs"


def _is_control(ch: str) -> bool:
    if ch in ("\t", "\n", "\r"):
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch: str) -> bool:
    cp = ord(ch)
    if (33 <= cp <= 47) or (58 <= cp <= 64) or (91 <= cp <= 96) or (123 <= cp <= 126):
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return ((0x4E00 <= cp <= 0x9FFF) or (0x3400 <= cp <= 0x4DBF) or (0x20000 <= cp <= 0x2A6DF) or (0x2A700 <= cp <= 0x2B73F) or
            (0x2B740 <= cp <= 0x2B81F) or (0x2B820 <= cp <= 0x2CEAF) or (0xF900 <= cp <= 0xFAFF) or (0x2F800 <= cp <= 0x2FA1F))


class BasicTokenizer(object):
    def __init__(self, do_lower_case: bool = True, never_split: Sequence[str] = NEVER_SPLIT):
        self.do_lower_case = do_lower_case
        self.never_split = tuple(never_split)

    def tokenize(self, text: str) -> List[str]:
        text = self._clean_text(text)
        text = self._tokenize_chinese_chars(text)
        split_tokens = []
        for token in whitespace_tokenize(text):
            if self.do_lower_case and token not in self.never_split:
                token = self._run_strip_accents(token.lower())
            split_tokens.extend(self._run_split_on_punc(token))
        return whitespace_tokenize(" ".join(split_tokens))

    @staticmethod
    def _run_strip_accents(text: str) -> str:
        return "".join(ch for ch in unicodedata.normalize("NFD", text) if unicodedata.category(ch) != "Mn")

    def _run_split_on_punc(self, text: str) -> List[str]:
        if text in self.never_split:
            return [text]
        out, start_new = [], True
        for ch in text:
            if _is_punctuation(ch):
                out.append([ch])
                start_new = True
            else:
                if start_new:
                    out.append([])
                start_new = False
                out[-1].append(ch)
        return ["".join(x) for x in out]

    @staticmethod
    def _tokenize_chinese_chars(text: str) -> str:
        out = []
        for ch in text:
            if _is_cjk(ord(ch)):
                out.extend((" ", ch, " "))
            else:
                out.append(ch)
        return "".join(out)

    @staticmethod
    def _clean_text(text: str) -> str:
        out = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            out.append(" " if _is_whitespace(ch) else ch)
        return "".join(out)


class WordpieceTokenizer(object):
    def __init__(self, vocab, unk_token: str = "[UNK]", max_input_chars_per_word: int = 100):
        self.vocab, self.unk_token, self.max_input_chars_per_word = vocab, unk_token, max_input_chars_per_word

    def tokenize(self, text: str) -> List[str]:
        output = []
        for token in whitespace_tokenize(text):
            chars = list(token)
            if len(chars) > self.max_input_chars_per_word:
                output.append(self.unk_token)
                continue
            bad, start, subs = False, 0, []
            while start < len(chars):
                end, cur = len(chars), None
                while start < end:
                    sub = "".join(chars[start:end])
                    if start > 0:
                        sub = "##" + sub
                    if sub in self.vocab:
                        cur = sub
                        break
                    end -= 1
                if cur is None:
                    bad = True
                    break
                subs.append(cur)
                start = end
            output.extend([self.unk_token] if bad else subs)
        return output


class BertTokenizer(object):
    """`BertTokenizer(vocab_file, do_lower_case=True)` with `.tokenize`, `.convert_tokens_to_ids`, `.from_pretrained(path)`."""

    def __init__(self, vocab_file: str, do_lower_case: bool = True, never_split: Sequence[str] = NEVER_SPLIT):
        if not os.path.isfile(vocab_file):
            raise ValueError("Can't find a vocabulary file at path '%s'" % vocab_file)
        self.vocab = load_vocab(vocab_file)
        self.ids_to_tokens = collections.OrderedDict((i, t) for t, i in self.vocab.items())
        self.basic_tokenizer = BasicTokenizer(do_lower_case=do_lower_case, never_split=never_split)
        self.wordpiece_tokenizer = WordpieceTokenizer(vocab=self.vocab)

    @classmethod
    def from_pretrained(cls, path: str, do_lower_case: bool = True, **kw):
        """A directory holding vocab.txt or the file itself; model-name shortcuts would need the network and are not supported."""
        if os.path.isdir(path):
            path = os.path.join(path, "vocab.txt")
        return cls(path, do_lower_case=do_lower_case, **kw)

    def tokenize(self, text: str) -> List[str]:
        out = []
        for token in self.basic_tokenizer.tokenize(text):
            out.extend(self.wordpiece_tokenizer.tokenize(token))
        return out

    def convert_tokens_to_ids(self, tokens: Sequence[str]) -> List[int]:
        return [self.vocab[t] for t in tokens]

    def convert_ids_to_tokens(self, ids: Sequence[int]) -> List[str]:
        return [self.ids_to_tokens[i] for i in ids]


def extract_token_from_nlp(tokenizer, nlp: str, seq_length: int) -> Tuple[List[int], List[int]]:
    """lib/test/tracker/uvltrack.py:196-233: [CLS] + word pieces (cut to seq_length - 2) + [SEP], zero-padded to seq_length.
    Returns (input_ids, input_mask) as Python lists; mask is 1 for real tokens."""
    nlp_token = tokenizer.tokenize(nlp)
    if len(nlp_token) > seq_length - 2:
        nlp_token = nlp_token[0:(seq_length - 2)]
    tokens = ["[CLS]"] + list(nlp_token) + ["[SEP]"]
    input_ids = tokenizer.convert_tokens_to_ids(tokens)
    input_mask = [1] * len(input_ids)
    while len(input_ids) < seq_length:
        input_ids.append(0)
        input_mask.append(0)
    assert len(input_ids) == seq_length and len(input_mask) == seq_length
    return input_ids, input_mask
