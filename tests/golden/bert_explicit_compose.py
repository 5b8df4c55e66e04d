"""The reference's explicit BERT composite (lxt/explicit/models/bert.py), hand-composed over PLAIN weight tensors from any module
pair (lf, rules) that offers the lxt.explicit API -- the vendored model file itself cannot be imported under transformers 5.x
(SURVEY.md 8c), so the composition is restated here rule site by rule site:

  embeddings   add2(word, token_type.detach()); add2(., position); LayerNormEpsilon          ref :249-253
  attention    Linear -> EpsilonRule (q, k, v); lf.matmul(q, k^T); mul2(., 1/sqrt(d)); add2(., mask); lf.softmax;
               lf.matmul(p, v)   (BOTH contractions use lf.matmul: R/(2 O + eps))             ref :60-65, :338-373
  self-output  Linear; LayerNormEpsilon(add2(dense, input))                                    ref :396
  MLP          Linear; GELUActivation -> IdentityRule; Linear; LayerNormEpsilon(add2(., .))
  head         pooler Linear + Tanh -> IdentityRule; classifier Linear

Used (1) by make_golden_bert_explicit.py with the REAL reference modules to freeze the fixture, (2) by the GPU tests with
lxt_amd.explicit.{functional, rules}: the same code, the other backend -- the drop-in claim of the explicit API surface."""
import math

import torch


def weights_from_hf(model, dtype=torch.float32, device="cpu"):
    """plain tensors of a HF BertForSequenceClassification"""
    def t(x):
        return x.detach().to(device=device, dtype=dtype).clone()
    b = model.bert
    W = dict(word=t(b.embeddings.word_embeddings.weight), pos=t(b.embeddings.position_embeddings.weight),
             tt=t(b.embeddings.token_type_embeddings.weight), eln_w=t(b.embeddings.LayerNorm.weight), eln_b=t(b.embeddings.LayerNorm.bias),
             ln_eps=float(model.config.layer_norm_eps), heads=int(model.config.num_attention_heads), layers=[],
             pool_w=t(b.pooler.dense.weight), pool_b=t(b.pooler.dense.bias), cls_w=t(model.classifier.weight), cls_b=t(model.classifier.bias))
    for L in b.encoder.layer:
        a = L.attention
        W["layers"].append(dict(
            wq=t(a.self.query.weight), bq=t(a.self.query.bias), wk=t(a.self.key.weight), bk=t(a.self.key.bias),
            wv=t(a.self.value.weight), bv=t(a.self.value.bias), wo=t(a.output.dense.weight), bo=t(a.output.dense.bias),
            ln1_w=t(a.output.LayerNorm.weight), ln1_b=t(a.output.LayerNorm.bias), wi=t(L.intermediate.dense.weight),
            bi=t(L.intermediate.dense.bias), wd=t(L.output.dense.weight), bd=t(L.output.dense.bias),
            ln2_w=t(L.output.LayerNorm.weight), ln2_b=t(L.output.LayerNorm.bias)))
    return W


def forward(lf, rules, W, ids, emb=None, hidden_hook=None):
    """logits [B, num_labels] of the explicit composite; `emb` (word embeddings, requires_grad) is where the relevance is read"""
    B, S = ids.shape
    H, nh = W["word"].shape[1], W["heads"]
    d = H // nh
    dev = W["word"].device
    if emb is None:
        emb = W["word"][ids]

    def lin(x, w, b):      # nn.Linear -> rules.EpsilonRule (epsilon default 1e-8); lf.linear_epsilon is the same rule
        return lf.linear_epsilon(x, w, b, 1e-8)        # (bit-identical on the CPU) and exists in both backends

    def ident(fn, x):      # IdentityRule
        return rules.identity(fn, x)

    tt = W["tt"][torch.zeros(B, S, dtype=torch.long, device=dev)]
    h = lf.add2(emb, tt.detach())
    h = lf.add2(h, W["pos"][torch.arange(S, device=dev)][None].expand(B, S, H))
    h = lf.layer_norm(h, W["eln_w"], W["eln_b"], W["ln_eps"])
    mask = torch.zeros(B, 1, S, S, dtype=h.dtype, device=dev)                    # HF's extended mask without padding: zeros
    for li, L in enumerate(W["layers"]):
        q = lin(h, L["wq"], L["bq"]).view(B, S, nh, d).permute(0, 2, 1, 3)
        k = lin(h, L["wk"], L["bk"]).view(B, S, nh, d).permute(0, 2, 1, 3)
        v = lin(h, L["wv"], L["bv"]).view(B, S, nh, d).permute(0, 2, 1, 3)
        s = lf.matmul(q, k.transpose(-1, -2))
        s = lf.mul2(s, 1 / math.sqrt(d))
        s = lf.add2(s, mask.expand(B, nh, S, S))
        p = lf.softmax(s, dim=-1)
        c = lf.matmul(p, v)
        c = c.permute(0, 2, 1, 3).contiguous().view(B, S, H)
        a = lin(c, L["wo"], L["bo"])
        h = lf.layer_norm(lf.add2(a, h), L["ln1_w"], L["ln1_b"], W["ln_eps"])
        m = ident(torch.nn.functional.gelu, lin(h, L["wi"], L["bi"]))
        o = lin(m, L["wd"], L["bd"])
        h = lf.layer_norm(lf.add2(o, h), L["ln2_w"], L["ln2_b"], W["ln_eps"])
        if hidden_hook is not None:
            hidden_hook(li, h)
    pooled = ident(torch.tanh, lin(h[:, 0], W["pool_w"], W["pool_b"]))
    return lin(pooled, W["cls_w"], W["cls_b"])


def explain(lf, rules, W, ids, target=None):
    """the explicit protocol: logits[0, idx].backward(that logit) -> relevance = emb.grad (ref: examples/paper/llama.py:45-46)"""
    emb = W["word"][ids].detach().clone().requires_grad_()
    logits = forward(lf, rules, W, ids, emb)
    idx = int(logits[0].argmax()) if target is None else int(target)
    logits[0, idx].backward(logits[0, idx].detach())
    R = emb.grad[0]
    return dict(idx=idx, logit=float(logits[0, idx].detach()), logits=logits.detach()[0], R_tok=R.sum(-1), R_emb=R)
