"""SBertLang on the MI355X kernels (encoder half) against the fixture from transformers' BertModel and the oracle."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pytestmark = pytest.mark.gpu

from oracle import hulc2_oracle as O  # noqa: E402  (checker only)
from tests.test_oracle_golden import _bert_sd  # noqa: E402

G = ROOT / "tests" / "golden"


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


@pytest.mark.parametrize("mode,tol", [("fp32", 2e-4), ("bf16", 3e-2)])
def test_sentence_embeddings_match_bert(dev, mode, tol):
    from hulc2_amd import kernels as kn
    from hulc2_amd.models.language_encoders import SBertLang

    fx = dict(np.load(G / "minilm.npz", allow_pickle=True))
    kn.set_compute(mode)
    try:
        enc = SBertLang().to(dev)
        enc.load_bert_state_dict(_bert_sd(int(fx["seed"])))
        ids, mask = torch.tensor(fx["input_ids"]), torch.tensor(fx["attention_mask"])
        emb = enc.encode_tokens(ids, mask).cpu()
        torch.cuda.synchronize()
    finally:
        kn.set_compute("bf16")
    want = torch.tensor(fx["sentence_embedding"])
    assert tuple(emb.shape) == (ids.shape[0], 384) and torch.isfinite(emb).all()
    assert (emb - want).abs().max().item() < tol * want.abs().max().item()


def test_pieces_against_the_oracle(dev):
    """ragged batch (one single-token sentence, S not a multiple of anything) in exact mode, and the error paths"""
    from hulc2_amd import kernels as kn
    from hulc2_amd.models.language_encoders import SBertLang

    kn.set_compute("fp32")
    try:
        sd = _bert_sd(5)
        enc = SBertLang().to(dev)
        enc.load_bert_state_dict({("0.auto_model." + k): v for k, v in sd.items()})      # sentence_transformers' prefix
        g = torch.Generator().manual_seed(3)
        ids = torch.randint(0, 30522, (5, 37), generator=g)
        lens = torch.tensor([37, 1, 20, 36, 5])
        mask = (torch.arange(37)[None] < lens[:, None]).long()
        emb = enc.encode_tokens(ids, mask).cpu()
        want = O.minilm_sentence_embedding(sd, ids, mask)
        assert (emb - want).abs().max().item() < 2e-4 * want.abs().max().item()
    finally:
        kn.set_compute("bf16")
    with pytest.raises(NotImplementedError):
        enc.encode(["open the drawer"])
    with pytest.raises(NotImplementedError):
        SBertLang("all-mpnet-base-v2")
    tok = lambda s: {"input_ids": ids[:len(s)], "attention_mask": mask[:len(s)]}
    enc2 = SBertLang(tokenizer=tok).to(dev)
    enc2.load_bert_state_dict(sd)
    out = enc2(["a", "b"])
    assert tuple(out.shape) == (2, 384)
    assert tuple(enc2.encode_text(["a", "b"])[0].shape) == (2, 1024)


def test_encode_sentences_from_a_checkpoint_directory(dev, tmp_path):
    """`model/language_encoder=sbert` as shipped: SBertLang(nlp_model) finds vocab.txt + model.safetensors in a checkpoint directory,
    `encode(list[str])` tokenises inside the module (length sort -> WordPiece -> forward -> un-sort, sbert_lang_encoder.py:38-62) and
    equals the oracle on the token ids transformers' BertTokenizer produced for the same sentences (tests/golden/wordpiece.npz)."""
    import numpy as np
    from safetensors.torch import save_file
    from hulc2_amd import kernels as kn
    from hulc2_amd.models.language_encoders import SBertLang

    G = ROOT / "tests" / "golden"
    fx = np.load(G / "wordpiece.npz", allow_pickle=False)
    sents = [str(s) for s in fx["sentences"]]
    root = tmp_path / "paraphrase-MiniLM-L3-v2"
    root.mkdir()
    (root / "vocab.txt").write_text((G / "wordpiece_vocab.txt").read_text())
    sd = _bert_sd(9)
    save_file({("0.auto_model." + k): v.contiguous() for k, v in sd.items()}, str(root / "model.safetensors"))
    kn.set_compute("fp32")
    try:
        enc = SBertLang(nlp_model=str(root)).to(dev)
        emb = enc(sents).cpu()                                                     # forward(list[str]) == encode
        ids, mask = torch.tensor(fx["input_ids"]), torch.tensor(fx["attention_mask"])
        want = O.minilm_sentence_embedding(sd, ids, mask)
        assert tuple(emb.shape) == (len(sents), 384)
        # row i of the output belongs to sentence i (the length sort is undone); padding to the batch maximum does not change a row
        assert (emb - want).abs().max().item() < 2e-4 * want.abs().max().item()
        one = enc([sents[3]]).cpu()
        assert (one[0] - want[3]).abs().max().item() < 2e-4 * want.abs().max().item()
    finally:
        kn.set_compute("bf16")
