"""XLM-RoBERTa text tower on the device: class-name token ids -> the
L2-normalised [K, 768] bank the similarity GEMM consumes.

Reference: ``XLMRobertaLanguageBackbone`` (mm_backbone.py:341-390) = HF ``XLMRobertaModel`` ->
``last_hidden_state[:, 0]`` -> ``nn.Linear`` head -> ``F.normalize``.  The tokenizer (SentencePiece
blobs fetched by ``AutoTokenizer.from_pretrained``) is host-side and absent offline: the boundary
here is token ids + attention mask, exactly what the reference's tokenizer call returns.

The encoder is the post-LayerNorm BERT layout: embeddings (word + position + token type) -> LN ->
N x [QKV projection, softmax(QK^T / sqrt(d)) V, output projection + residual -> LN, 4x FFN with
exact-erf GELU + residual -> LN].  Dense layers run on the same GEMM kernels as the image tower
(``precision`` = "fp32" or "fp16x3"); the bank is built once per vocabulary, so nothing here is tuned."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import lib as L


def position_ids(input_ids: torch.Tensor, pad_id: int = 1) -> torch.Tensor:
    """HF ``create_position_ids_from_input_ids``: non-pad tokens count up from pad_id + 1, pads stay pad_id."""
    m = (input_ids != pad_id).to(torch.int32)
    return (torch.cumsum(m, dim=1).to(torch.int32) * m + pad_id).to(torch.int32)


class TextTower:
    """``state`` uses HF names: ``embeddings.*``, ``encoder.layer.N.*`` (an optional ``model.`` prefix is
    stripped), plus ``head.weight`` / ``head.bias`` of the 768-d projection."""

    def __init__(self, state: Dict[str, "np.ndarray | torch.Tensor"], num_heads: int, device="cuda",
                 precision: str = "fp16x3", pad_id: int = 1, eps: float = 1e-5):
        if precision not in ("fp32", "fp16x3"):
            raise ValueError("precision must be fp32 or fp16x3")
        self.dev = torch.device(device)
        self.precision, self.pad_id, self.eps = precision, pad_id, eps
        g = {}
        for k, v in state.items():
            k = k[6:] if k.startswith("model.") else k
            t = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).to(torch.float32)
            g[k] = t.to(self.dev).contiguous()
        self.w = g
        self.hidden = g["embeddings.word_embeddings.weight"].shape[1]
        self.heads = num_heads
        self.dh = self.hidden // num_heads
        self.layers = 1 + max(int(k.split(".")[2]) for k in g if k.startswith("encoder.layer."))
        self.out_dim = g["head.weight"].shape[0]
        # fused QKV projection: rows (Wq | Wk | Wv)
        for i in range(self.layers):
            p = f"encoder.layer.{i}.attention.self."
            g[p + "qkv.weight"] = torch.cat([g[p + "query.weight"], g[p + "key.weight"], g[p + "value.weight"]]).contiguous()
            g[p + "qkv.bias"] = torch.cat([g[p + "query.bias"], g[p + "key.bias"], g[p + "value.bias"]]).contiguous()
        self._split = {}

    def _gemm(self, a, wname, c, m, k, n, act=L.ACT_NONE, res=None):
        w, b = self.w[wname + ".weight"], self.w[wname + ".bias"]
        kw = dict(batch=1, hin=1, win=m, cin=k, lda=a.shape[1], n=n, ldc=c.shape[1], act=act)
        if res is not None:
            kw.update(res=res, ldres=res.shape[1])
        if self.precision == "fp16x3" and k % 8 == 0:
            ws = self._split.get(wname)
            if ws is None:
                ws = self._split[wname] = L.split_weights(w)
            L.conv_gemm(a, None, b, c, w_split=ws, **kw)
        else:
            L.conv_gemm(a, w, b, c, **kw)

    @torch.no_grad()
    def encode(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[n, L] token ids (+ mask, default ids != pad) -> [n, out_dim] L2-normalised class embeddings."""
        ids = input_ids.to(self.dev, torch.int32).contiguous()
        n, ln = ids.shape
        if ln > 64:
            raise L.WedetectHipError("sequences longer than 64 tokens are not supported by wd_attention_small")
        mask = (ids != self.pad_id).to(torch.int32) if attention_mask is None else attention_mask.to(self.dev, torch.int32).contiguous()
        pos = position_ids(ids, self.pad_id)
        h, m = self.hidden, n * ln
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.dev)
        x, y, qkv, att = f(m, h), f(m, h), f(m, 3 * h), f(m, h)
        inter = f(m, self.w["encoder.layer.0.intermediate.dense.weight"].shape[0])
        e = "embeddings."
        L.text_embed(ids.view(-1), pos.view(-1), self.w[e + "word_embeddings.weight"], self.w[e + "position_embeddings.weight"],
                     self.w[e + "token_type_embeddings.weight"][0].contiguous(), x)
        L.layernorm_rows(x, x, self.w[e + "LayerNorm.weight"], self.w[e + "LayerNorm.bias"], m, h, eps=self.eps)
        for i in range(self.layers):
            p = f"encoder.layer.{i}."
            self._gemm(x, p + "attention.self.qkv", qkv, m, h, 3 * h)
            L.attention_small(qkv, mask, att, n, ln, self.heads, self.dh)
            self._gemm(att, p + "attention.output.dense", y, m, h, h, res=x)
            L.layernorm_rows(y, y, self.w[p + "attention.output.LayerNorm.weight"], self.w[p + "attention.output.LayerNorm.bias"],
                             m, h, eps=self.eps)
            self._gemm(y, p + "intermediate.dense", inter, m, h, inter.shape[1], act=L.ACT_GELU)
            self._gemm(inter, p + "output.dense", x, m, inter.shape[1], h, res=y)
            L.layernorm_rows(x, x, self.w[p + "output.LayerNorm.weight"], self.w[p + "output.LayerNorm.bias"], m, h, eps=self.eps)
        cls = x.view(n, ln, h)[:, 0].contiguous()                 # last_hidden_state[:, 0]
        feats = f(n, self.out_dim)
        self._gemm(cls, "head", feats, n, h, self.out_dim)
        out = f(n, self.out_dim)
        L.l2norm_rows(feats, out)
        return out


class XLMRobertaLanguageBackbone:
    """Operator-surface mirror of the reference class (mm_backbone.py:330-390; constructor keywords ``model_name,
    model_size, frozen_modules, dropout, training_use_cache, init_cfg`` as the configs pass them):
    ``forward(texts)`` with ``texts`` = ``List[List[str]]`` returns ``[B, K, D]`` normalised text features.

    ``tokenizer`` is any callable ``(list_of_strings) -> {"input_ids": [n, L], "attention_mask": [n, L]}``.  When
    none is given, ``AutoTokenizer.from_pretrained(model_name)`` is tried at first use, like the reference's
    constructor does (its SentencePiece files must be on disk: there is no download path here); a failure raises.
    ``forward_ids`` takes the tokenizer's output directly.  Weights arrive through ``load_state_dict`` (the
    detector forwards the checkpoint's ``backbone.text_model.*`` tensors); the device tower is built at first use."""

    HEADS = {"tiny": 12, "base": 12, "large": 16, "xlarge": 16}

    def __init__(self, model_name: Optional[str] = None, model_size: str = "base", frozen_modules=(), dropout: float = 0.0,
                 training_use_cache: bool = False, init_cfg=None, *, tokenizer=None, precision: Optional[str] = None):
        if model_size not in self.HEADS:
            raise ValueError(f"model_size must be one of {sorted(self.HEADS)}")
        self.model_name, self.model_size, self.tokenizer = model_name, model_size, tokenizer
        self.precision = precision
        self._state: Optional[Dict[str, torch.Tensor]] = None
        self._tower: Optional[TextTower] = None
        self._device = None
        self.training = False

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = {(k[len("backbone.text_model."):] if k.startswith("backbone.text_model.") else k): v
              for k, v in state_dict.items()}
        need = ("model.embeddings.word_embeddings.weight", "head.weight", "head.bias")
        missing = [k for k in need if k not in sd and k[6:] not in sd]
        if missing:
            raise RuntimeError(f"cannot build the text tower, missing {missing}")
        self._state = {k: v for k, v in sd.items() if not k.endswith("position_ids")}
        self._tower = None
        return self

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return dict(self._state or {})

    def cuda(self, device=None):
        self._device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self._tower is not None and self._tower.dev != self._device:
            self._tower = None
        return self

    def eval(self):
        return self

    def _ensure(self) -> TextTower:
        if self._state is None:
            raise RuntimeError("load_state_dict() must be called before encoding text")
        if self._tower is None:
            import os
            prec = self.precision or os.environ.get("WEDETECT_PRECISION", "fp16x3")
            self._tower = TextTower(self._state, self.HEADS[self.model_size], device=self._device or "cuda", precision=prec)
        return self._tower

    def _tokenizer(self):
        if self.tokenizer is None:
            if not self.model_name:
                raise RuntimeError("no tokenizer: pass one to the constructor, give model_name=<directory with the XLM-R "
                                   "tokenizer files>, or use forward_ids")
            try:
                from transformers import AutoTokenizer
                hf = AutoTokenizer.from_pretrained(self.model_name)
            except Exception as e:                                 # missing files, no network
                raise RuntimeError(f"cannot load the tokenizer of {self.model_name!r} ({type(e).__name__}: {e}); tokenizer "
                                   "files are host-side data this package does not ship") from e
            self.tokenizer = lambda strings: hf(text=list(strings), return_tensors="pt", padding=True)   # mm_backbone.py:383
        return self.tokenizer

    def forward_ids(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self._ensure().encode(input_ids, attention_mask)

    def forward(self, text) -> torch.Tensor:
        num = [len(t) for t in text]
        if max(num) != min(num):
            raise AssertionError("number of sequences not equal in batch")      # mm_backbone.py:377-379
        tokenizer = self._tokenizer()
        flat = [s for t in text for s in t]
        tok = tokenizer(flat)
        feats = self.forward_ids(torch.as_tensor(tok["input_ids"]), torch.as_tensor(tok["attention_mask"]))
        return feats.reshape(-1, num[0], feats.shape[-1])

    __call__ = forward

    def encode_classes(self, names) -> torch.Tensor:
        """List[str] -> [K, D]: the callable ``YOLOWorldDetector(text_encoder=...)`` expects."""
        return self.forward([list(names)])[0]
