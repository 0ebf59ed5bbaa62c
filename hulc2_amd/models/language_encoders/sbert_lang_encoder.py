"""SBertLang — the sentence encoder of `language_encoder: sbert` on the MI355X kernels (encoder half).

Mirrors hulc2/affordance/models/language_encoders/sbert_lang_encoder.py:13-71 (same class name, `encode` / `forward` / `encode_text`,
the unused-in-forward `text_fc` 384 -> 1024 head).  The reference wraps SentenceTransformer("paraphrase-MiniLM-L3-v2") = transformers'
BertModel (3 layers, hidden 384, 12 heads, GELU, LayerNorm eps 1e-12) + mean Pooling; here that arithmetic runs as
  hulc_embed_ln_fwd -> 3 x [hulc_gemm (fused QKV) -> hulc_mha_masked_fwd -> hulc_gemm -> hulc_ln_wide_fwd -> hulc_gemm (GELU epilogue)
  -> hulc_gemm -> hulc_ln_wide_fwd] -> hulc_masked_mean_fwd,
frozen and inference-only as in the reference (`freeze_backbone=True`).

`encode(list[str])` tokenises inside the module like the reference (sbert_lang_encoder.py:38-62: length-sort, tokenize, forward, un-sort)
with the WordPiece tokenizer of wordpiece.py (token ids bit-exact against transformers' BertTokenizer, tests/golden/wordpiece.npz).

What is NOT here: the trained checkpoint and its vocab.txt (no network in the build image).  They are read from a checkpoint directory —
`nlp_model` given as a path, $HULC2_SBERT_DIR, or the local sentence-transformers / huggingface cache — never downloaded; without one the
weights stay zero until `load_bert_state_dict` (a transformers BertModel state_dict, sentence_transformers' `0.auto_model.` prefix
stripped) and `encode` needs `tokenizer=`.  Parity is pinned on the arithmetic (tests/golden/minilm.npz from transformers' own BertModel
with seeded weights) and on the token ids, not on the checkpoint."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch
from torch import nn

import numpy as np

from ... import kernels as kn
from .wordpiece import WordPieceTokenizer, find_checkpoint_dir


class SBertLang(nn.Module):
    HIDDEN, LAYERS, HEADS, INTER, VOCAB, MAXPOS, EPS = 384, 3, 12, 1536, 30522, 512, 1e-12

    def __init__(self, nlp_model: str = "paraphrase-MiniLM-L3-v2", freeze_backbone: bool = True,
                 tokenizer: Optional[Callable[[List[str]], Dict[str, torch.Tensor]]] = None) -> None:
        super().__init__()
        import os
        ckpt_dir = find_checkpoint_dir(nlp_model)
        if os.path.basename(os.path.normpath(nlp_model)).replace("sentence-transformers_", "") != "paraphrase-MiniLM-L3-v2":
            raise NotImplementedError(f"{nlp_model}: only paraphrase-MiniLM-L3-v2 (conf/model/language_encoder/sbert.yaml) is built")
        if not freeze_backbone:
            raise NotImplementedError("the sentence encoder is inference-only (the reference trains with freeze_backbone=True)")
        self.freeze_backbone, self.tokenizer = freeze_backbone, tokenizer
        D, I = self.HIDDEN, self.INTER
        z = lambda *s: nn.Parameter(torch.zeros(*s), requires_grad=False)
        self.word, self.pos, self.tok_type = z(self.VOCAB, D), z(self.MAXPOS, D), z(2, D)
        self.emb_ln_w, self.emb_ln_b = z(D), z(D)
        for l in range(self.LAYERS):
            for name, shape in (("qkv_w", (3 * D, D)), ("qkv_b", (3 * D,)), ("ao_w", (D, D)), ("ao_b", (D,)), ("ln1_w", (D,)), ("ln1_b", (D,)),
                                ("in_w", (I, D)), ("in_b", (I,)), ("out_w", (D, I)), ("out_b", (D,)), ("ln2_w", (D,)), ("ln2_b", (D,))):
                setattr(self, f"l{l}_{name}", z(*shape))
        self.text_fc = nn.Linear(D, 1024)
        self._w16: Dict[str, torch.Tensor] = {}
        self.checkpoint_dir = ckpt_dir
        if ckpt_dir is not None:
            self.load_checkpoint_dir(ckpt_dir)

    def load_checkpoint_dir(self, root: str) -> None:
        """a sentence-transformers checkpoint directory: vocab.txt -> the tokenizer (unless one was injected), model.safetensors /
        pytorch_model.bin (top level or 0_Transformer/) -> the BertModel weights"""
        import os
        if self.tokenizer is None:
            self.tokenizer = WordPieceTokenizer.from_checkpoint_dir(root, max_length=128)
        for sub in ("", "0_Transformer"):
            st, pt = os.path.join(root, sub, "model.safetensors"), os.path.join(root, sub, "pytorch_model.bin")
            if os.path.isfile(st):
                from safetensors.torch import load_file
                self.load_bert_state_dict(load_file(st))
                return
            if os.path.isfile(pt):
                self.load_bert_state_dict(torch.load(pt, map_location="cpu", weights_only=True))
                return

    # ---- weights ---------------------------------------------------------------------------------------------------------
    def load_bert_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """sd: transformers BertModel.state_dict() (optionally with sentence_transformers' '0.auto_model.' prefix)"""
        sd = {k[len("0.auto_model."):] if k.startswith("0.auto_model.") else k: v for k, v in sd.items()}
        with torch.no_grad():
            self.word.copy_(sd["embeddings.word_embeddings.weight"]); self.pos.copy_(sd["embeddings.position_embeddings.weight"])
            self.tok_type.copy_(sd["embeddings.token_type_embeddings.weight"])
            self.emb_ln_w.copy_(sd["embeddings.LayerNorm.weight"]); self.emb_ln_b.copy_(sd["embeddings.LayerNorm.bias"])
            for l in range(self.LAYERS):
                q = f"encoder.layer.{l}."
                g = lambda n: getattr(self, f"l{l}_{n}")
                g("qkv_w").copy_(torch.cat([sd[q + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], 0))
                g("qkv_b").copy_(torch.cat([sd[q + f"attention.self.{n}.bias"] for n in ("query", "key", "value")], 0))
                g("ao_w").copy_(sd[q + "attention.output.dense.weight"]); g("ao_b").copy_(sd[q + "attention.output.dense.bias"])
                g("ln1_w").copy_(sd[q + "attention.output.LayerNorm.weight"]); g("ln1_b").copy_(sd[q + "attention.output.LayerNorm.bias"])
                g("in_w").copy_(sd[q + "intermediate.dense.weight"]); g("in_b").copy_(sd[q + "intermediate.dense.bias"])
                g("out_w").copy_(sd[q + "output.dense.weight"]); g("out_b").copy_(sd[q + "output.dense.bias"])
                g("ln2_w").copy_(sd[q + "output.LayerNorm.weight"]); g("ln2_b").copy_(sd[q + "output.LayerNorm.bias"])
        self._w16.clear()

    def _operand(self, name: str) -> torch.Tensor:
        """dense weight as the GEMM's B operand: the fp32 parameter in exact mode, a cached bf16 copy in bf16 mode"""
        w = getattr(self, name)
        if kn.get_compute() != "bf16":
            return w
        c = self._w16.get(name)
        if c is None or c.device != w.device:
            c = self._w16[name] = w.detach().to(torch.bfloat16).contiguous()
        return c

    # ---- forward ---------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_tokens(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        """(B, S) token ids + (B, S) attention mask -> (B, 384) sentence embeddings"""
        B, S = input_ids.shape
        if S > 128:
            raise ValueError("sentences longer than 128 tokens are not supported (the reference's model truncates at 128)")
        dev, D, I, T = self.word.device, self.HIDDEN, self.INTER, input_ids.numel()
        ids = input_ids.to(dev, torch.int64).contiguous()
        mask = attention_mask.to(dev, torch.int32).contiguous()
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        x = kn.embed_ln_fwd(ids.view(-1), self.word, self.pos, self.tok_type[0].contiguous(), self.emb_ln_w, self.emb_ln_b, self.EPS, T, S, D, f(T, D))
        qkv, ctx, t1, a, h, t2 = f(T, 3 * D), f(T, D), f(T, D), f(T, D), f(T, I), f(T, D)
        for l in range(self.LAYERS):
            g = lambda n: getattr(self, f"l{l}_{n}")
            kn.gemm(x, self._operand(f"l{l}_qkv_w"), qkv, T, 3 * D, D, D, D, 3 * D, bias=g("qkv_b"))
            kn.mha_masked_fwd(qkv, mask, B, S, self.HEADS, D // self.HEADS, ctx)
            kn.gemm(ctx, self._operand(f"l{l}_ao_w"), t1, T, D, D, D, D, D, bias=g("ao_b"))
            kn.ln_wide_fwd(t1, x, g("ln1_w"), g("ln1_b"), self.EPS, T, D, a)
            kn.gemm(a, self._operand(f"l{l}_in_w"), h, T, I, D, D, D, I, bias=g("in_b"), relu=2)          # exact GELU epilogue
            kn.gemm(h, self._operand(f"l{l}_out_w"), t2, T, D, I, I, I, D, bias=g("out_b"))
            x = kn.ln_wide_fwd(t2, a, g("ln2_w"), g("ln2_b"), self.EPS, T, D, f(T, D))
        return kn.masked_mean_fwd(x, mask, B, S, D, f(B, D))

    def encode(self, sentences: List[str], normalize_embeddings: bool = False) -> torch.Tensor:
        """sbert_lang_encoder.py:31-62: sort by text length (longest first), tokenize, forward, undo the sort, stack"""
        if self.tokenizer is None:
            raise NotImplementedError("no tokenizer: vocab.txt of paraphrase-MiniLM-L3-v2 is not shipped and no checkpoint directory was found "
                                      "(nlp_model=<dir>, $HULC2_SBERT_DIR, or the local sentence-transformers cache); pass "
                                      "tokenizer=callable(sentences) -> {'input_ids', 'attention_mask'} or call encode_tokens()")
        order = np.argsort([-len(sen) for sen in sentences])                 # SentenceTransformer._text_length of a str = len(str)
        feats = self.tokenizer([sentences[i] for i in order])
        emb = self.encode_tokens(feats["input_ids"], feats["attention_mask"])
        if normalize_embeddings:
            emb = torch.nn.functional.normalize(emb, p=2, dim=1)
        return emb[torch.as_tensor(np.argsort(order), device=emb.device)]

    def forward(self, x: List[str]) -> torch.Tensor:
        return self.encode(x)

    def encode_text(self, x: List[str]):
        return self.text_fc(self.encode(x)), None, None
