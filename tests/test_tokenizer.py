"""WordPiece tokenizer pinned against transformers.BertTokenizer (slow Python tokenizer, same lineage as
pytorch-pretrained-bert's) on a synthetic vocabulary; plus extract_token_from_nlp (tracker:196-233)."""
import os

import pytest

from uvltrack_amd.tokenizer import BertTokenizer, extract_token_from_nlp

WORDS = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "the", "a", "man", "woman", "in", "red", "shirt", "walk", "##ing", "##s", "##ed",
         "on", "left", "right", "side", "of", "road", "car", "white", "dog", "run", "##ning", "black", "and", "with", ",", ".", "!", "?", "'",
         "-", "un", "##believ", "##able", "cafe", "naive", "person", "play", "##er", "ball", "basket", "##ball", "2", "##3", "x", "##y", "##z",
         "中", "国", "ride", "##r", "bike", "motor", "##cycle", "behind", "front", "is", "who", "wear", "##s"]
SENTENCES = [
    "The man in the red shirt walking on the left side of the road.",
    "a white dog running,   and a black   car!",
    "Unbelievable!! café naïve player's basketball 23 xyz",
    "who is the motorcycle rider behind the car?",
    "中国 bike\tride\nwalks  walked zzzz qqq",
    "[CLS] the [MASK] man [SEP]",
    "",
    "   ",
    "x" * 150 + " man",
    "the-man ... a--b 'car' \x00�\x07dog",
]


@pytest.fixture(scope="module")
def vocab_file(tmp_path_factory):
    p = tmp_path_factory.mktemp("vocab") / "vocab.txt"
    seen, lines = set(), []
    for w in WORDS:
        if w not in seen:
            seen.add(w)
            lines.append(w)
    p.write_text("\n".join(lines) + "\n", encoding="utf-8")
    return str(p)


def test_matches_transformers_slow_tokenizer(vocab_file):
    hf = pytest.importorskip("transformers")
    ref = hf.BertTokenizer(vocab_file, do_lower_case=True)
    mine = BertTokenizer(vocab_file, do_lower_case=True)
    for s in SENTENCES:
        want = ref.tokenize(s)
        got = mine.tokenize(s)
        assert got == want, (s, got, want)
        assert mine.convert_tokens_to_ids(got) == ref.convert_tokens_to_ids(want)


def test_extract_token_from_nlp(vocab_file):
    tok = BertTokenizer.from_pretrained(os.path.dirname(vocab_file))
    ids, mask = extract_token_from_nlp(tok, "the man in the red shirt", 40)
    assert len(ids) == len(mask) == 40
    n = sum(mask)
    assert n == 8 and ids[0] == tok.vocab["[CLS]"] and ids[n - 1] == tok.vocab["[SEP]"] and ids[n:] == [0] * (40 - n) and mask[:n] == [1] * n
    # truncation keeps [CLS] ... [SEP] within seq_length
    ids, mask = extract_token_from_nlp(tok, "the man " * 50, 16)
    assert len(ids) == 16 and sum(mask) == 16 and ids[-1] == tok.vocab["[SEP]"]
    with pytest.raises(ValueError):
        BertTokenizer("/nonexistent/vocab.txt")
