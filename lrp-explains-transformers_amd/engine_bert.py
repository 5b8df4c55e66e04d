"""Whole-model AttnLRP engine for BERT-style encoders with a sequence-classification head (BASELINE config 2): fused forward +
LRP backward on the HIP kernels, one call per batch of equally long prompts, optionally replayed as ONE hipGraph.

Rule placement (gradient form):
  mode="explicit"   ref lxt/explicit/models/bert.py:60-65 (nn.Linear -> EpsilonRule 1e-8, GELU / Tanh -> IdentityRule),
                    :249-253 (embeddings: add2, add2, LayerNormEpsilon), :338-373 (lf.matmul on BOTH attention contractions,
                    mul2 by 1/sqrt(d), add2 with the mask, lf.softmax), :396 ff. (LayerNormEpsilon(add2(dense, input)))
  mode="efficient"  ref lxt/efficient/models/bert.py:86-90,339,380,476-488,581,790,806 (stop-gradient LayerNorm, identity rule on
                    GELU / Tanh, divide_gradient 4,4,2 around the attention) == the explicit composite with every eps = 0
Differences from the Llama driver (engine.py): biases everywhere (z of the eps-rule includes them), LayerNorm instead of
RMSNorm, non-causal attention with head dim 64 (the generic flash kernels of attention.hip: S x S scores are never
materialised, unlike the explicit drop-in path), the P.V rule of lf.matmul  o/(2 o + eps) = 1/2 * o/(o + eps/2)  (the uniform-rule
kernel with half the stabiliser), a pooler + classifier head on the [CLS] row.
Python only sequences kernel launches on the current stream (plus the embedding gather); with graph=True the whole
explanation of a (B, S) shape is captured once and replayed -- BERT-base at S = 128 is ~600 launches of a few microseconds each,
i.e. launch-bound without it.
"""
import math

import torch

from . import ops

EXPLICIT = dict(lin=1e-8, add=1e-8, qk=1e-8, mask=1e-8, pv=1e-8, ln=1e-6, act=0.0)
EFFICIENT = dict(lin=0.0, add=0.0, qk=0.0, mask=0.0, pv=0.0, ln=0.0, act=1e-10)


def weights_from_hf(model):
    """(cfg, W) of a HF BertForSequenceClassification: plain tensors, no copies"""
    c = model.config
    if getattr(c, "model_type", "bert") != "bert":
        raise NotImplementedError(f"BertLRP drives BERT encoders only (model_type={c.model_type!r})")
    if getattr(c, "position_embedding_type", "absolute") != "absolute":
        raise NotImplementedError("BertLRP: only absolute position embeddings are supported")
    act = c.hidden_act if isinstance(c.hidden_act, str) else "gelu"
    if act not in ("gelu", "gelu_new", "gelu_pytorch_tanh"):
        raise NotImplementedError(f"BertLRP: hidden_act {act!r} is not supported")
    cfg = dict(hidden=c.hidden_size, inter=c.intermediate_size, n_layers=c.num_hidden_layers, n_heads=c.num_attention_heads,
               ln_eps=float(c.layer_norm_eps), act="gelu" if act == "gelu" else "gelu_tanh", labels=model.classifier.weight.shape[0])
    b = model.bert
    d = lambda t: t.detach()                                              # noqa: E731
    W = dict(word=d(b.embeddings.word_embeddings.weight), pos=d(b.embeddings.position_embeddings.weight),
             tt=d(b.embeddings.token_type_embeddings.weight), eln_w=d(b.embeddings.LayerNorm.weight), eln_b=d(b.embeddings.LayerNorm.bias),
             pool_w=d(b.pooler.dense.weight), pool_b=d(b.pooler.dense.bias), cls_w=d(model.classifier.weight), cls_b=d(model.classifier.bias),
             layers=[])
    for L in b.encoder.layer:
        a = L.attention
        W["layers"].append(dict(
            wq=d(a.self.query.weight), bq=d(a.self.query.bias), wk=d(a.self.key.weight), bk=d(a.self.key.bias),
            wv=d(a.self.value.weight), bv=d(a.self.value.bias), wo=d(a.output.dense.weight), bo=d(a.output.dense.bias),
            ln1_w=d(a.output.LayerNorm.weight), ln1_b=d(a.output.LayerNorm.bias), wi=d(L.intermediate.dense.weight),
            bi=d(L.intermediate.dense.bias), wd=d(L.output.dense.weight), bd=d(L.output.dense.bias),
            ln2_w=d(L.output.LayerNorm.weight), ln2_b=d(L.output.LayerNorm.bias)))
    return cfg, W


class BertLRP:
    """Device-resident weights (forward layout only; ops.linear_dgrad derives the backward from it) + explain()."""

    def __init__(self, cfg, W, dtype=torch.float32, device="cuda", mode="efficient"):
        if not torch.cuda.is_available():
            raise RuntimeError("BertLRP needs a HIP device: the LRP kernels have no CPU fallback")
        self.cfg, self.dtype, self.device = dict(cfg), dtype, torch.device(device)
        self.set_mode(mode)
        t = lambda x: x.to(device=self.device, dtype=dtype).contiguous()     # noqa: E731
        self.word, self.eln_w, self.eln_b = t(W["word"]), t(W["eln_w"]), t(W["eln_b"])
        self.tt0 = t(W["tt"][0:1])                                            # token type 0 (single-segment inputs)
        self.pos = t(W["pos"])
        self.pool_w, self.pool_b, self.cls_w, self.cls_b = t(W["pool_w"]), t(W["pool_b"]), t(W["cls_w"]), t(W["cls_b"])
        self.layers = []
        for L in W["layers"]:
            wqkv = torch.cat([t(L["wq"]), t(L["wk"]), t(L["wv"])], 0)
            P = dict(wqkv=wqkv, bqkv=torch.cat([t(L["bq"]), t(L["bk"]), t(L["bv"])], 0), wo=t(L["wo"]), bo=t(L["bo"]),
                     wi=t(L["wi"]), bi=t(L["bi"]), wd=t(L["wd"]), bd=t(L["bd"]), ln1_w=t(L["ln1_w"]), ln1_b=t(L["ln1_b"]),
                     ln2_w=t(L["ln2_w"]), ln2_b=t(L["ln2_b"]))
            self.layers.append(P)
        self._graphs = {}

    @classmethod
    def from_hf(cls, model, dtype=None, device="cuda", mode="efficient"):
        cfg, W = weights_from_hf(model)
        return cls(cfg, W, dtype=dtype or next(model.parameters()).dtype, device=device, mode=mode)

    def set_mode(self, mode):
        if mode not in ("efficient", "explicit"):
            raise ValueError(f"mode must be 'efficient' or 'explicit', got {mode!r}")
        self.mode, self.eps = mode, dict(EXPLICIT if mode == "explicit" else EFFICIENT)
        self._graphs = {}

    # ------------------------------------------------------------------------------------------------ helpers
    def _scale(self, G, z, eps, out=None):
        """G * z/(z + eps): the eps-rule factor of one site (identity when eps == 0)"""
        if eps == 0.0:
            return G
        return ops.eps_scale(G, z, 1.0, eps, out=out)

    def _lin(self, x, w, b):
        out = torch.empty(x.shape[0], w.shape[0], device=x.device, dtype=x.dtype)
        return ops.gemm_nt_2d(x, w, out, b)

    def _dgrad(self, G, z, w, eps):
        """input gradient of z = x W^T + b under the eps rule: (G * z/(z+eps)) W, from the stored weight W [out, in] (bf16: NN form of
        the GEMM, no W^T copy; fp32 and odd shapes such as the 2-label classifier: a W^T copy cached on the weight)"""
        return ops.linear_dgrad(self._scale(G, z, eps), w)

    # ------------------------------------------------------------------------------------------------ one explanation
    def _run(self, ids, target, want_layers):
        cfg, E = self.cfg, self.eps
        B, S = ids.shape
        H, nh, I = cfg["hidden"], cfg["n_heads"], cfg["inter"]
        d = H // nh
        M = B * S
        scale = 1.0 / math.sqrt(d)
        act = cfg["act"]
        need_t = ops.attn_needs_transposed(self.word, d)
        # ---------------------------------------------------------------- forward
        word = self.word.index_select(0, ids.reshape(-1))                    # the only non-library launch: the embedding gather
        e1 = ops.add_bcast(word, self.tt0)
        e2 = ops.add_bcast(e1, self.pos[:S])
        h, _, rstd0 = ops.layernorm_fwd(e2, self.eln_w, self.eln_b, cfg["ln_eps"])
        h0 = h
        stash = []
        for P in self.layers:
            qkv = self._lin(h, P["wqkv"], P["bqkv"])                         # [M, 3H]; q / k / v are column slices, used in place
            q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
            v_t = ops.transpose_heads(v, B, S, nh, d) if need_t else None
            o = torch.empty(M, H, device=h.device, dtype=h.dtype)
            lse = torch.empty(B, nh, S, device=h.device, dtype=torch.float32)
            ops.attn_fwd(q, k, v, v_t, o, lse, B, S, nh, nh, d, scale, False, 0)
            a = self._lin(o, P["wo"], P["bo"])
            r1 = ops.add_bcast(a, h)
            h1, _, rstd1 = ops.layernorm_fwd(r1, P["ln1_w"], P["ln1_b"], cfg["ln_eps"])
            zi = self._lin(h1, P["wi"], P["bi"])
            m = ops.act_fwd(zi, act)
            dn = self._lin(m, P["wd"], P["bd"])
            r2 = ops.add_bcast(dn, h1)
            h2, _, rstd2 = ops.layernorm_fwd(r2, P["ln2_w"], P["ln2_b"], cfg["ln_eps"])
            stash.append(dict(h=h, qkv=qkv, o=o, lse=lse, a=a, r1=r1, h1=h1, rstd1=rstd1, zi=zi, dn=dn, r2=r2, h2=h2, rstd2=rstd2))
            h = h2
        cls_rows = h.view(B, S, H)[:, 0]                                     # [B, H] view, row stride S*H
        zp = self._lin(cls_rows, self.pool_w, self.pool_b)
        pooled = ops.act_fwd(zp, "tanh")
        logits = torch.empty(B, cfg["labels"], device=h.device, dtype=torch.float32)
        ops.gemm_nt_2d(pooled, self.cls_w, logits, self.cls_b)
        # ---------------------------------------------------------------- seed: unit gradient on the explained logit
        if target is None:
            idx, _ = ops.argmax_rows(logits)
            idx = idx.long()
        else:
            idx = target
        onehot = torch.zeros(B, cfg["labels"], device=h.device, dtype=h.dtype).scatter_(1, idx.view(B, 1), 1.0)
        logit = logits.gather(1, idx.view(B, 1)).view(B)
        # ---------------------------------------------------------------- backward (gradient form)
        Gpooled = self._dgrad(onehot, logits.to(h.dtype), self.cls_w, E["lin"])
        Gzp = ops.act_bwd(Gpooled, zp, "tanh", E["act"])
        Gh = torch.zeros(M, H, device=h.device, dtype=h.dtype)
        Gh.view(B, S, H)[:, 0].copy_(self._dgrad(Gzp, zp, self.pool_w, E["lin"]))
        layer_R = []
        if want_layers:
            layer_R.append(ops.readout(h, Gh).view(B, S).sum(1))
        for P, c in zip(reversed(self.layers), reversed(stash)):
            qkv = c["qkv"]
            q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
            # h2 = LN(add2(dn, h1))
            Gr2 = ops.layernorm_bwd(Gh, c["h2"], P["ln2_w"], c["rstd2"], E["ln"])
            Gs = self._scale(Gr2, c["r2"], E["add"])                         # add2: the same factor to both summands
            Gm = self._dgrad(Gs, c["dn"], P["wd"], E["lin"])
            Gzi = ops.act_bwd(Gm, c["zi"], act, E["act"])
            Gh1 = ops.add_bcast(self._dgrad(Gzi, c["zi"], P["wi"], E["lin"]), Gs)
            # h1 = LN(add2(a, h))
            Gr1 = ops.layernorm_bwd(Gh1, c["h1"], P["ln1_w"], c["rstd1"], E["ln"])
            Gs1 = self._scale(Gr1, c["r1"], E["add"])
            Go = self._dgrad(Gs1, c["a"], P["wo"], E["lin"])
            # attention: lf.matmul on P.V == the uniform-rule kernel with eps/2 (o/(2o+eps) = 1/2 o/(o+eps/2))
            Gho = torch.empty_like(Go)
            D = torch.empty(B, nh, S, device=h.device, dtype=torch.float32)
            ops.attn_bwd_prep(Go, c["o"], Gho, D, B, S, nh, d, 0.5 * E["pv"], 0.5)
            A = torch.empty(M, 3 * H, device=h.device, dtype=h.dtype)        # [dQ | dK | dV], then scaled in place
            dq, dk, dv = A[:, :H], A[:, H:2 * H], A[:, 2 * H:]
            q_t, Gho_t = (ops.transpose_heads(q, B, S, nh, d), ops.transpose_heads(Gho, B, S, nh, d)) if need_t else (None, None)
            ops.attn_bwd_dkv(q, k, v, q_t, Gho, Gho_t, c["lse"], D, dk, dv, B, S, nh, nh, d, scale, E["mask"], E["qk"], False, 0)
            k_t = ops.transpose_heads(k, B, S, nh, d) if need_t else None
            ops.attn_bwd_dq(q, k, v, k_t, Gho, c["lse"], D, dq, B, S, nh, nh, d, scale, E["mask"], E["qk"], False, 0)
            if E["lin"] != 0.0:
                ops.eps_scale2d(A, qkv, A, 1.0, E["lin"])
            Gx = ops.linear_dgrad(A, P["wqkv"])
            Gh = ops.add_bcast(Gx, Gs1)
            if want_layers:
                layer_R.append(ops.readout(c["h"], Gh).view(B, S).sum(1))
        Ge2 = ops.layernorm_bwd(Gh, h0, self.eln_w, rstd0, E["ln"])
        Ge1 = self._scale(Ge2, e2, E["add"])
        Gword = self._scale(Ge1, e1, E["add"])
        R_tok = ops.readout(word, Gword).view(B, S)
        out = dict(R_tok=R_tok, idx=idx, logit=logit, logits=logits)
        if want_layers:
            out["layer_R"] = torch.stack(layer_R[::-1], 1)                   # [B, L+1]: embeddings output ... last layer
        return out

    def explain(self, input_ids, target=None, layer_relevance=False, graph=False):
        """input_ids [B, S] (equally long prompts, token type 0, no padding) -> dict(R_tok [B,S] fp32, idx [B], logit [B],
        logits [B, labels], layer_R [B, L+1] if asked).  target: [B] class indices (default: arg-max logit per prompt).
        graph=True: capture the launches for this (B, S, target given?) once and replay them as one hipGraph; the returned
        tensors are then the graph's static outputs (copy them before the next call)."""
        ids = input_ids.to(self.device)
        if ids.dim() != 2:
            raise ValueError("input_ids must be [B, S]")
        if ids.shape[1] > self.pos.shape[0]:
            raise ValueError(f"sequence length {ids.shape[1]} exceeds the position table ({self.pos.shape[0]})")
        if target is not None:
            target = torch.as_tensor(target, device=self.device).long().view(-1)
            if target.numel() != ids.shape[0] or int(target.min()) < 0 or int(target.max()) >= self.cfg["labels"]:
                raise ValueError("target must hold one class index in [0, num_labels) per prompt")
        if not graph:
            return self._run(ids, target, layer_relevance)
        key = (tuple(ids.shape), target is not None, bool(layer_relevance))
        g = self._graphs.get(key)
        if g is None:
            s_ids = ids.clone()
            s_tgt = target.clone() if target is not None else None
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                                    # warm-up outside the capture (lazy module loads)
                self._run(s_ids, s_tgt, layer_relevance)
            torch.cuda.current_stream().wait_stream(side)
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg):
                out = self._run(s_ids, s_tgt, layer_relevance)
            g = self._graphs[key] = (cg, s_ids, s_tgt, out)
        cg, s_ids, s_tgt, out = g
        s_ids.copy_(ids)
        if s_tgt is not None:
            s_tgt.copy_(target)
        cg.replay()
        return out
