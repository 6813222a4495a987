"""Text tower parity (GPU): the device encoder against HuggingFace's own XLMRobertaModel (the class the
reference instantiates, mm_backbone.py:355) run on the CPU with the same seeded random weights, followed
by the reference's head + normalisation (mm_backbone.py:384-386)."""
import numpy as np
import pytest
import torch

from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _hf(cfg_kw, out_dim, seed):
    from transformers import XLMRobertaConfig, XLMRobertaModel
    torch.manual_seed(seed)
    cfg = XLMRobertaConfig(type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5, hidden_act="gelu",
                           hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **cfg_kw)
    model = XLMRobertaModel(cfg, add_pooling_layer=False).eval()
    head = torch.nn.Linear(cfg.hidden_size, out_dim)
    with torch.no_grad():                                  # trained-like spread instead of the 0.02-std init
        for n_, p_ in model.named_parameters():
            if p_.dim() == 2 and "embeddings" not in n_:
                p_.mul_(2.5)
    return cfg, model, head


def _ids(n, ln, vocab, seed):
    g = np.random.default_rng(seed)
    ids = np.full((n, ln), 1, np.int64)                    # pad
    for i in range(n):
        k = int(g.integers(1, ln - 1))
        ids[i, 0] = 0                                      # <s>
        ids[i, 1:1 + k] = g.integers(4, vocab, k)
        ids[i, 1 + k] = 2                                  # </s>
    return torch.from_numpy(ids)


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
@pytest.mark.parametrize("dims", [dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256),
                                  dict(hidden_size=768, num_hidden_layers=3, num_attention_heads=12, intermediate_size=3072)])
def test_text_tower_matches_huggingface(precision, dims):
    from wedetect_amd.text import TextTower, position_ids
    vocab, n, ln, out_dim = 1000, 37, 11, 768
    cfg, model, head = _hf(dict(vocab_size=vocab, max_position_embeddings=64, **dims), out_dim, seed=3)
    ids = _ids(n, ln, vocab, seed=4)
    mask = (ids != 1).long()
    with torch.no_grad():
        hs = model(input_ids=ids, attention_mask=mask)["last_hidden_state"][:, 0]
        ref = torch.nn.functional.normalize(head(hs), dim=-1)
    sd = dict(model.state_dict())
    sd["head.weight"], sd["head.bias"] = head.weight.detach(), head.bias.detach()
    tower = TextTower(sd, cfg.num_attention_heads, precision=precision)
    assert tower.layers == cfg.num_hidden_layers and tower.dh == cfg.hidden_size // cfg.num_attention_heads
    got = tower.encode(ids.cuda(), mask.cuda())
    assert_close(f"text bank {precision}", got, ref, 2e-5)
    # default mask (ids != pad) and HF's position ids
    from transformers.models.xlm_roberta.modeling_xlm_roberta import XLMRobertaEmbeddings
    assert torch.equal(position_ids(ids, 1).long(), XLMRobertaEmbeddings.create_position_ids_from_input_ids(ids, 1))
    assert torch.equal(tower.encode(ids.cuda()), got)


def test_attention_small_vs_torch():
    from wedetect_amd import lib as L
    n, ln, heads, dh = 5, 13, 3, 32
    g = torch.Generator(device="cuda").manual_seed(9)
    qkv = torch.randn(n * ln, 3 * heads * dh, device="cuda", generator=g)
    mask = (torch.rand(n, ln, device="cuda", generator=g) > 0.3).to(torch.int32)
    mask[:, 0] = 1
    out = torch.empty(n * ln, heads * dh, device="cuda")
    L.attention_small(qkv, mask, out, n, ln, heads, dh)
    q, k, v = [t.view(n, ln, heads, dh).permute(0, 2, 1, 3).double() for t in qkv.split(heads * dh, dim=1)]
    s = q @ k.transpose(-1, -2) / dh ** 0.5
    s = s.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
    ref = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(n * ln, heads * dh)
    assert_close("attention", out, ref, 2e-6, 1e-6)
    with pytest.raises(L.WedetectHipError):
        L.attention_small(qkv, mask, out, n, 65, heads, dh)


def test_language_backbone_surface_feeds_the_detector():
    """XLMRobertaLanguageBackbone(texts) -> [B, K, D] with a stand-in tokenizer, reference key prefixes,
    and YOLOWorldDetector.reparameterize using it as its text encoder."""
    from wedetect_amd.text import XLMRobertaLanguageBackbone
    vocab = 500
    cfg, model, head = _hf(dict(vocab_size=vocab, max_position_embeddings=32, hidden_size=768, num_hidden_layers=1,
                                num_attention_heads=12, intermediate_size=256), 768, seed=5)
    sd = {"backbone.text_model.model." + k: v for k, v in model.state_dict().items()}
    sd["backbone.text_model.head.weight"], sd["backbone.text_model.head.bias"] = head.weight.detach(), head.bias.detach()

    def tokenizer(strings):                                # deterministic toy tokenizer: bytes -> ids, padded
        rows = [[0] + [4 + (b % (vocab - 4)) for b in s.encode()][:10] + [2] for s in strings]
        ln = max(len(r) for r in rows)
        ids = torch.tensor([r + [1] * (ln - len(r)) for r in rows])
        return {"input_ids": ids, "attention_mask": (ids != 1).long()}

    tb = XLMRobertaLanguageBackbone(model_size="base", tokenizer=tokenizer).load_state_dict(sd)
    texts = [["person", "traffic light", "dog"], ["cat", "kite", "bicycle rack"]]
    feats = tb(texts)
    assert feats.shape == (2, 3, 768)
    tok = tokenizer([s for t in texts for s in t])
    with torch.no_grad():
        hs = model(input_ids=tok["input_ids"], attention_mask=tok["attention_mask"])["last_hidden_state"][:, 0]
        ref = torch.nn.functional.normalize(head(hs), dim=-1).reshape(2, 3, 768)
    assert_close("language backbone", feats, ref, 2e-5)
    assert_close("unit norm", feats.norm(dim=-1), torch.ones(2, 3), 1e-5)
    assert torch.equal(tb.encode_classes(texts[0]), feats[0])
    with pytest.raises(AssertionError):
        tb([["a", "b"], ["c"]])
    with pytest.raises(RuntimeError):
        XLMRobertaLanguageBackbone(model_size="base")([["a"]])
