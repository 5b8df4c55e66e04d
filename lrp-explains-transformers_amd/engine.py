"""Whole-model AttnLRP engine for Llama-style decoders: fused forward + LRP backward on the HIP
kernels, one call per batch of prompts.

This is the hot path named by BASELINE.json: it reproduces the user protocol of the reference
(docs/source/quickstart.rst:120-141; examples/paper/llama.py:27-46 --  embed ids -> forward ->
pick max(logits[0,-1]) -> seed the backward -> sum_h emb (*) grad) with the rule placement of
lxt/efficient/models/llama.py:9-14 (mode="efficient") or of the explicit composite
lxt/explicit/models/llama.py:83-93,226-260,273-281,379-391,481-488 (mode="explicit").

MI355X-first decisions:
  * everything an eps-rule needs (every Linear output z, pre/post-RoPE q/k, attention o + lse,
    residual sums) is STASHED in HBM during the forward (~0.3 GB/layer at S=2048 bf16; 288 GB
    available) -- the backward never recomputes a GEMM, it runs one dgrad GEMM per Linear;
  * every weight is kept ONCE, in its forward layout W [out,in] (QKV and gate+up fused along the output dim; gate / up rows interleaved
    in blocks of 32 so that the gated-MLP rules run inside the GEMM epilogues): the forward is an NT
    GEMM, the backward the NN form  c = s W  of the same kernel (the transposed MFMA operand is gathered in LDS by
    ds_read_b64_tr_b16); only the fp32 parity engine falls back to a W^T copy (ops.linear_dgrad);
  * bf16 / head_dim 128 attention reads its transposed operands out of row-major LDS tiles as well; fp32 and the other head dims keep
    head-transposed copies of q/k/v/Gho (attention.hip, `attn_t`);
  * only the last token's logits are formed (the explained logit lives there), and the LM head
    + final norm backward is a single row per prompt.
Python only sequences kernel launches on the current stream; it performs no arithmetic.
"""
import torch

from . import ops

EXPLICIT = dict(lin=1e-8, add=1e-8, qk=1e-8, mask=1e-8, pv=1e-6, rope=1e-8, act=None)
EFFICIENT = dict(lin=0.0, add=0.0, qk=0.0, mask=0.0, pv=0.0, rope=0.0, act=1e-10)


_STATIC_ROPE = ("default", "linear", "llama3", "yarn")


def rope_kind(hf_cfg):
    """rope type of a HF config: transformers 5 keeps it in `rope_parameters`, transformers 4.x in `rope_scaling` (key 'rope_type', legacy
    key 'type') next to `rope_theta` -- Llama-3.1 / 3.2 there carry {'rope_type': 'llama3', ...}, which must not be dropped silently"""
    rp = getattr(hf_cfg, "rope_parameters", None)
    if isinstance(rp, dict):
        return rp.get("rope_type", rp.get("type", "default"))
    rs = getattr(hf_cfg, "rope_scaling", None)
    if rs is None:
        return "default"
    if isinstance(rs, dict) and (rs.get("rope_type") or rs.get("type")):
        return rs.get("rope_type") or rs.get("type")
    raise NotImplementedError(f"LlamaLRP: cannot interpret rope_scaling={rs!r}")


def config_from_hf(hf_cfg):
    """HF LlamaConfig -> engine cfg.  Everything the fused driver does not implement is refused LOUDLY here instead of
    being ignored: other model types (use lxt_amd.efficient.monkey_patch for those), attention / MLP biases, and rope
    types whose frequencies depend on the running sequence length ("dynamic", "longrope").  The rotary frequencies and
    the cos/sin post-scale are taken from HF's own initialiser for the config's rope_type (Llama-3.1/3.2: "llama3"),
    exactly what LlamaRotaryEmbedding.__init__ does -- the reference inherits them from HF unchanged."""
    mt = getattr(hf_cfg, "model_type", "llama")
    if mt != "llama":
        raise NotImplementedError(f"LlamaLRP drives Llama-architecture decoders only (model_type={mt!r}); "
                                  "other families run through lxt_amd.efficient.monkey_patch")
    if getattr(hf_cfg, "attention_bias", False) or getattr(hf_cfg, "mlp_bias", False):
        raise NotImplementedError("LlamaLRP: attention_bias / mlp_bias = True are not supported by the fused driver "
                                  "(use the monkey_patch drop-in path)")
    hd = getattr(hf_cfg, "head_dim", None) or hf_cfg.hidden_size // hf_cfg.num_attention_heads
    theta = None
    rp = getattr(hf_cfg, "rope_parameters", None)
    if isinstance(rp, dict):
        theta = rp.get("rope_theta")
    if theta is None:
        theta = getattr(hf_cfg, "rope_theta", 10000.0)
    cfg = dict(hidden=hf_cfg.hidden_size, inter=hf_cfg.intermediate_size, n_layers=hf_cfg.num_hidden_layers,
               n_heads=hf_cfg.num_attention_heads, n_kv=hf_cfg.num_key_value_heads, head_dim=hd,
               vocab=hf_cfg.vocab_size, rope_theta=float(theta), rms_eps=float(hf_cfg.rms_norm_eps),
               act=getattr(hf_cfg, "hidden_act", "silu"))
    kind = rope_kind(hf_cfg)
    if kind not in _STATIC_ROPE:
        raise NotImplementedError(f"LlamaLRP: rope_type {kind!r} (sequence-length dependent frequencies) is not supported")
    if kind != "default":
        from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
        inv_freq, att = ROPE_INIT_FUNCTIONS[kind](hf_cfg, "cpu")      # (the installed transformers' own initialiser for its own config class)
        cfg["inv_freq"], cfg["attention_scaling"] = inv_freq.float().cpu(), float(att)
    return cfg


def weights_from_hf(model):
    """plain (cfg, W) view of a HF LlamaForCausalLM (no copies; tensors stay where they are)"""
    m = model.model
    W = dict(embed=m.embed_tokens.weight.detach(), norm=m.norm.weight.detach(), lm_head=model.lm_head.weight.detach(),
             layers=[])
    for L in m.layers:
        a, mlp = L.self_attn, L.mlp
        W["layers"].append(dict(ln1=L.input_layernorm.weight.detach(), ln2=L.post_attention_layernorm.weight.detach(),
                                wq=a.q_proj.weight.detach(), wk=a.k_proj.weight.detach(), wv=a.v_proj.weight.detach(),
                                wo=a.o_proj.weight.detach(), wg=mlp.gate_proj.weight.detach(),
                                wu=mlp.up_proj.weight.detach(), wd=mlp.down_proj.weight.detach()))
    return config_from_hf(model.config), W


PITCH_PAD = True          # module attribute, not an environment knob: tools/r3_ab_bench.sh-style A/B scripts set engine.PITCH_PAD = False


def pitch_pad(cols, elem_size):
    """extra elements per row for a K-contiguous GEMM operand: a row pitch that is a multiple of 4 KiB puts the same K offset of every row
    on the same memory channels; measured on MI355X at M = 8192, N = 4096 (tools/pitch_probe.py, profiles/r03_gemm_experiments.txt):
    K = 28672 NT 1357 -> 1553 TFLOP/s with 128 bytes of padding on both operands, K = 14336 1460 -> 1545 (hipBLASLt gains too:
    1533 -> 1603); no effect at K <= 6144 or on the C pitch"""
    nbytes = cols * elem_size
    if not PITCH_PAD:
        return 0
    return 128 // elem_size if (nbytes >= 16384 and nbytes % 4096 == 0) else 0


def weight_pitch_pad(cols, elem_size, rows=0):
    """extra elements per ROW of a stored weight W [out, in] that is larger than the chip's caches: the dgrad c = s W reads it in the NN form --
    512-byte segments of consecutive contraction ROWS, i.e. with the weight's row pitch as the stride -- and a pitch that is a multiple of 4 KiB
    (in = 4096 in bf16) puts a column block of every row on the same memory channels.  Measured on MI355X, M = 8192 (tools/nn_pitch_probe.py,
    profiles/r05_gemm_experiments.txt): the [28672, 4096] gate/up weight 1418 -> 1274 us with 128 bytes of padding per row, 1237 us (1357 -> 1555
    TFLOP/s) with 256; the [6144, 4096] qkv weight 271 -> 267 us; the cache-resident [4096, 4096] o weight and the [4096, 14336] down weight:
    nothing.  256 bytes where the pitch is a multiple of 1 KiB and the weight is at least 32 MB."""
    nbytes = cols * elem_size
    # (round 5, Gemma-3-4B's [20480, 2560] gate/up weight: a pitch of 5 KiB -- a multiple of 1 KiB, not of 4 -- aliases too: dgrad 751 -> 650 us,
    # forward 612 -> 588 us with the same 256 bytes, tools/gemma_pitch_probe.py; the L2-resident qkv / o weights and SigLIP's: nothing)
    if not PITCH_PAD or nbytes % 1024 != 0 or rows * nbytes < (32 << 20):
        return 0
    return 256 // elem_size


class LlamaLRP:
    """Device-resident weights (each ONCE, forward layout, one flat buffer) + explain()."""

    def __init__(self, cfg, W, dtype=torch.bfloat16, device="cuda", mode="efficient", max_seq=4096, sparse_top=True, fold_norm=None):
        if not torch.cuda.is_available():
            raise RuntimeError("LlamaLRP needs a HIP device: the LRP kernels have no CPU fallback")
        self.cfg, self.dtype, self.device = dict(cfg), dtype, torch.device(device)
        self.set_mode(mode)
        # top-layer sparsity: above the last attention layer only the LAST token of a prompt feeds the
        # explained logit and carries relevance, so the last layer's o-proj / MLP (forward and backward) and
        # its attention rows are evaluated for one row per prompt (M = B instead of B*S)
        self.sparse_top = bool(sparse_top)
        self.act = cfg.get("act", "silu")
        dev = self.device
        # fold_norm (default: the bf16 engine): the weights of the two per-layer RMSNorms are multiplied into the COLUMNS of the Linears that
        # consume them (W'qkv = Wqkv diag(w1), W'gu = Wgu diag(w2)) and replaced by ones -- the same network (w (.) x rstd) W^T = (x rstd) W'^T,
        # and the same relevance under every rule of both placements: a contribution x_i w_ji of the eps-rule is unchanged, the norm's
        # identity rule does not see its weight.  What it buys: the norm is then a pure row scale, which commutes with the Linear --
        # rstd (.) (h W'^T) -- so for M = B S rows the efficient placement runs it inside the GEMM epilogues (K1n, _norm_fused below).
        self.folded = bool(dtype == torch.bfloat16 if fold_norm is None else fold_norm)
        self._nf_cache = {}                 # row count -> do the K1n entry points take every GEMM around the norms (_norm_fused)

        # ONE flat device buffer holds every weight in its forward layout (embedding, norms, LM head, per layer the fused
        # [q;k;v] and [gate;up] matrices, o, down): the tensors below are views into it, so the multi-GPU start-up is a
        # single broadcast of `self.flat` (lxt_amd.dist.broadcast_weights) followed by the LOCAL W^T copies
        H, I, nq, nk, hd, V = cfg["hidden"], cfg["inter"], cfg["n_heads"], cfg["n_kv"], cfg["head_dim"], cfg["vocab"]
        nqkv = (nq + 2 * nk) * hd
        up = lambda n: (n + 63) // 64 * 64                                   # noqa: E731  (every view starts 128-byte aligned)
        es = torch.empty(0, dtype=dtype).element_size()
        per_layer = (2 * up(H) + up(nqkv * (H + weight_pitch_pad(H, es, nqkv))) + up(H * nq * hd) + up(2 * I * (H + weight_pitch_pad(H, es, 2 * I)))
                     + up(H * (I + pitch_pad(I, es))))
        total = 2 * up(V * H) + up(H) + len(W["layers"]) * per_layer
        self.flat = torch.empty(total, device=dev, dtype=dtype)
        cursor = [0]

        def take(*shape):
            n = 1
            for s_ in shape:
                n *= s_
            v = self.flat[cursor[0]: cursor[0] + n].view(*shape)
            cursor[0] += up(n)
            return v

        def take_rows(rows, cols):
            # [rows, cols] view with a row pitch that is not a multiple of 4 KiB (pitch_pad): the K-contiguous operand of a long-K GEMM
            pad = pitch_pad(cols, es)
            return take(rows, cols + pad)[:, :cols]

        def take_weight(rows, cols):
            # stored weight whose NN (dgrad) reads stride over its rows: row pitch off the 4-KiB grid (weight_pitch_pad)
            pad = weight_pitch_pad(cols, es, rows)
            return take(rows, cols + pad)[:, :cols]

        def put(dst, *srcs):
            o = 0
            for t in srcs:
                rows = t.shape[0]
                dst[o: o + rows].copy_(t.to(device=dev, dtype=dtype, non_blocking=True))
                o += rows
            return dst

        def put_gu(dst, wg, wu):
            # gate / up rows interleaved in blocks of 32 (ops.interleave_gate_up): the gated-MLP rules then run inside the GEMM epilogues
            return ops.interleave_gate_up(wg.to(device=dev, dtype=dtype), wu.to(device=dev, dtype=dtype), out=dst)

        self.embed, self.lm_head = put(take(V, H), W["embed"]), put(take(V, H), W["lm_head"])
        self.norm = put(take(H), W["norm"])
        self.lm_head_t = None                        # [H, V] copy, made on the first dense-seed explanation
        self.layers = []

        def fold(w, ln):
            # W' = W diag(ln): product in fp32, ONE rounding to the storage dtype, block by block (no 4-byte copy of a whole weight)
            lnf = ln.to(device=dev, dtype=torch.float32)
            for r0 in range(0, w.shape[0], 4096):
                blk = w[r0: r0 + 4096]
                blk.copy_((blk.float() * lnf).to(dtype))
            return w

        for L in W["layers"]:
            Lw = dict(ln1=put(take(H), L["ln1"]), ln2=put(take(H), L["ln2"]),
                      wqkv=put(take_weight(nqkv, H), L["wq"], L["wk"], L["wv"]), wo=put(take(H, nq * hd), L["wo"]),
                      wgu=put_gu(take_weight(2 * I, H), L["wg"], L["wu"]), wd=put(take_rows(H, I), L["wd"]))
            if self.folded:
                fold(Lw["wqkv"], L["ln1"])
                fold(Lw["wgu"], L["ln2"])
                Lw["ln1"].fill_(1.0)
                Lw["ln2"].fill_(1.0)
            self.layers.append(Lw)
        self.attn_t = ops.attn_needs_transposed(self.embed, cfg["head_dim"])
        d = cfg["head_dim"]
        inv = cfg.get("inv_freq")                    # scaled rope types: HF's own frequencies (config_from_hf)
        if inv is None:
            inv = 1.0 / (cfg["rope_theta"] ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
        att = float(cfg.get("attention_scaling", 1.0))
        fr = torch.arange(max_seq, dtype=torch.float32)[:, None] * inv.float().cpu()[None, :]
        emb = torch.cat((fr, fr), dim=-1)
        # HF hands cos/sin to the layers in the model dtype; keep that rounding, store as fp32 tables
        self.cos = (emb.cos() * att).to(dtype).to(torch.float32).to(dev).contiguous()
        self.sin = (emb.sin() * att).to(dtype).to(torch.float32).to(dev).contiguous()
        self.max_seq = max_seq
        torch.cuda.synchronize(dev)

    # ops.linear_fwd / ops.linear_dgrad pick the kernel by row count: W-streaming small-M kernels and the split-K skinny path for the
    # one-row-per-prompt top layer and the LM head, the 256x256 ping-pong GEMM (NT forward, NN backward) for M = B*S rows
    def _lin_fwd(self, x, W, out):
        return ops.linear_fwd(x, W, out=out)

    def _lin_bwd(self, A, W, out):
        return ops.linear_dgrad(A, W, out=out)

    def _pitches(self):
        """row pitches (elements) of the backward's two wide GEMM operands, in ONE place: what backward() allocates is what the eligibility
        checks below are asked about"""
        c = self.cfg
        es = torch.empty(0, dtype=self.dtype).element_size()
        nqkv = (c["n_heads"] + 2 * c["n_kv"]) * c["head_dim"]
        aqkv_pad = 64 if ((nqkv * es) % 4096 == 0 and PITCH_PAD) else 0
        return dict(Aqkv=nqkv + aqkv_pad, Aqkv_pad=aqkv_pad, Agu=2 * c["inter"] + pitch_pad(2 * c["inter"], es), m=c["inter"] + pitch_pad(c["inter"], es))

    def _gated_coef(self, M):
        """the gated-MLP rules run as a coefficient stash inside the two GEMMs around them (ops.gemm_gated_fwd_coef / _bwd_coef) at this row count"""
        key = ("coef", M, PITCH_PAD, ops.GATED_FUSION)
        hit = self._nf_cache.get(key)
        if hit is None:
            c = self.cfg
            L0 = self.layers[0] if self.layers else None
            hit = bool(L0 is not None and ops.gated_coef_ok(M, c["inter"], c["hidden"], c["hidden"], L0["wgu"].stride(0), c["hidden"],
                                                            L0["wd"].stride(0), self.act, self.dtype))
            self._nf_cache[key] = hit
        return hit

    def _norm_fused(self, M, fwd_only=False):
        """K1n applies: folded norm weights, efficient placement (no stabiliser on the residual add / the Linears: eps = 0), and every GEMM on
        both sides of the two norms is a problem the ping-pong kernel's fused epilogues take (bf16, >= 190 tiles, N % 256 == 0).  fwd_only: the
        FORWARD half is the same computation under both placements (the explicit one additionally keeps each Linear's own output: round 6)"""
        if not (self.folded and ops.NORM_FUSION and (self.mode == "efficient" or fwd_only)):
            return False
        key = ("nf", M, PITCH_PAD, repr(ops.NORM_FUSION), ops.GATED_FUSION)
        hit = self._nf_cache.get(key)
        if hit is None:
            c = self.cfg
            H, I, d, nq, nk = c["hidden"], c["inter"], c["head_dim"], c["n_heads"], c["n_kv"]
            nqkv = (nq + 2 * nk) * d
            L0 = self.layers[0] if self.layers else None
            ok = L0 is not None and self._gated_coef(M)
            if ok:
                pt = self._pitches()
                ok = all(ops.norm_fused_ok(*a, self.dtype) for a in (
                    (M, H, nq * d, nq * d, L0["wo"].stride(0), False),              # h1 = h + o Wo^T
                    (M, H, I, pt["m"], L0["wd"].stride(0), False),                  # h' = h1 + m Wd^T
                    (M, nqkv, H, H, L0["wqkv"].stride(0), False),                   # qkv = rstd (h W'qkv^T)
                    (M, 2 * I, H, H, L0["wgu"].stride(0), False),                   # gate/up = rstd (h1 W'gu^T)
                    (M, H, nqkv, pt["Aqkv"], L0["wqkv"].stride(0), True),           # G_h = rstd (Aqkv W'qkv) + G_res
                    (M, H, 2 * I, pt["Agu"], L0["wgu"].stride(0), True)))
            self._nf_cache[key] = hit = bool(ok)
        return hit

    def build_transposes(self):
        """kept for callers of the round-2 API (bench.py, dist tests): the bf16 engine holds no W^T copies any more; the fp32 parity
        engine makes them lazily (ops.weight_t, cached on the weight) -- dropping the cache here makes a weight broadcast visible"""
        for L in self.layers:
            ops.clear_weight_cache(L["wqkv"], L["wo"], L["wgu"], L["wd"])
        self.lm_head_t = None

    @classmethod
    def from_hf(cls, model, **kw):
        cfg, W = weights_from_hf(model)
        kw.setdefault("dtype", next(model.parameters()).dtype)
        return cls(cfg, W, **kw)

    def set_mode(self, mode):
        if mode not in ("explicit", "efficient"):
            raise ValueError(f"mode must be 'explicit' or 'efficient', got {mode!r}")
        self.mode = mode
        self._graphs = {}
        self.eps = dict(EXPLICIT if mode == "explicit" else EFFICIENT)
        # identity rule (*) gate Linear eps: act/(g+eps_lin) in explicit form, act/(g+1e-10) efficient
        self.eps_g = self.eps["lin"] if mode == "explicit" else self.eps["act"]

    # ---------------------------------------------------------------------------------------------
    class _Arena:
        """Workspace arena: every activation stash and every temporary of one explanation lives in a flat buffer keyed by a tag; the
        buffers persist across calls (same B*S -> no allocator traffic in the per-layer loops, and stable addresses for hipGraph
        replay).  A tag is (name, layer or None); temporaries share one tag across layers."""

        def __init__(self, device):
            self.device, self.buf = device, {}
            self.gen = 0          # bumped whenever a buffer is REPLACED by a larger one: captured graphs hold the old address (engine drops them)

        def get(self, tag, shape, dtype, zero=False, pad=0):
            """pad: extra elements per row of a 2-D buffer (the view returned is [rows, cols] with row pitch cols + pad)"""
            full = tuple(shape[:-1]) + (shape[-1] + pad,) if pad else tuple(shape)
            n = 1
            for s_ in full:
                n *= s_
            t = self.buf.get((tag, dtype))
            if t is None or t.numel() < n:
                if t is not None:
                    self.gen += 1
                t = torch.empty(max(n, 1), device=self.device, dtype=dtype)
                self.buf[(tag, dtype)] = t
            v = t[:n].view(*full)
            if zero:
                v.zero_()
            return v[..., : shape[-1]] if pad else v

        def nbytes(self):
            return sum(t.numel() * t.element_size() for t in self.buf.values())

    def release(self):
        """drop the arena (and the captured graphs that point into it)"""
        self._arena, self._graphs = None, {}

    def forward(self, emb, B, S, row_iv=None):
        c = self.cfg
        H, I, d, nq, nk = c["hidden"], c["inter"], c["head_dim"], c["n_heads"], c["n_kv"]
        M = B * S
        dev, dt = self.device, self.dtype
        nqk, nqkv = (nq + nk) * d, (nq + 2 * nk) * d
        scale = d ** -0.5
        if getattr(self, "_arena", None) is None:
            self._arena = LlamaLRP._Arena(dev)
        ar = self._arena
        new = lambda tag, *s: ar.get(tag, s, dt)  # noqa: E731
        wide = lambda tag, r, c_: ar.get(tag, (r, c_), dt, pad=pitch_pad(c_, emb.element_size()))  # noqa: E731  (long-K GEMM operands)
        f32 = lambda tag, *s: ar.get(tag, s, torch.float32)  # noqa: E731
        stash = []
        h_prev, branch = emb, None
        last = torch.arange(B, device=dev) * S + (S - 1)
        coef = self._gated_coef(M)
        explicit = self.mode == "explicit"
        nf = self._norm_fused(M, fwd_only=True) and ops.norm_fusion_part("fwd")          # K1n: the two norms + residual sums of a layer inside the GEMM epilogues around them
        ready = None                      # (h, rstd1) of this layer, left by the previous layer's down-projection epilogue
        nL = len(self.layers)
        for li, Lw in enumerate(self.layers):
            st = {}
            rotated = False
            top = self.sparse_top and li == nL - 1
            if ready is not None:
                st["h"], st["rstd1"] = ready
                ready = None
                qkv = new(("qkv", li), M, nqkv)
                if not explicit and S <= self.max_seq and ops.gemm_nt_rs_rope_ok(st["h"], Lw["wqkv"], qkv, S, nqk, d):
                    # RoPE in the QKV GEMM's epilogue: q and k leave the kernel rotated (no rope_fwd pass, no second copy of q / k)
                    ops.gemm_nt_rs_rope(st["h"], Lw["wqkv"], st["rstd1"], self.cos, self.sin, qkv, S, nqk, d)
                    rotated = True
                else:
                    ops.gemm_nt_rs(st["h"], Lw["wqkv"], st["rstd1"], qkv)
            else:
                x, st["rstd1"] = new("x", M, H), f32(("rstd1", li), M)
                if branch is None:
                    st["h"] = h_prev
                    ops.add_rmsnorm_fwd(h_prev, None, Lw["ln1"], c["rms_eps"], y=x, rstd=st["rstd1"])
                else:
                    st["h"] = new(("h", li), M, H)
                    ops.add_rmsnorm_fwd(h_prev, branch, Lw["ln1"], c["rms_eps"], hsum_out=st["h"], y=x, rstd=st["rstd1"])
                qkv = self._lin_fwd(x, Lw["wqkv"], new(("qkv", li), M, nqkv))
            qkr = qkv[:, :nqk] if rotated else ops.rope_fwd(qkv, new(("qkr", li), M, nqk), self.cos, self.sin, S, nq + nk, d)
            v = qkv[:, nqk:]
            v_t = ops.transpose_heads(v, B, S, nk, d) if self.attn_t else None
            o = new(("o", li), M, nq * d)
            lse = f32(("lse", li), B, nq, S)
            if top:
                ops.attn_fwd(qkr[:, : nq * d], qkr[:, nq * d:], v, v_t, o, lse, B, S, nq, nk, d, scale, True, 0, q_begin=S - 1, row_iv=row_iv)
                o_l, h_l = o.index_select(0, last), st["h"].index_select(0, last)
                a_l = self._lin_fwd(o_l, Lw["wo"], new("a_l", B, H))
                h1_l = new("h1_l", B, H)
                x2_l, rstd2_l = ops.add_rmsnorm_fwd(h_l, a_l, Lw["ln2"], c["rms_eps"], hsum_out=h1_l)
                gu_l, m_l = ops.gemm_gated_fwd(x2_l, Lw["wgu"], new("gu_l", B, 2 * I), new("m_l", B, I), self.act)
                dn_l = self._lin_fwd(m_l, Lw["wd"], new("dn_l", B, H))
                st.update(top=True, qkv=qkv, qkr=qkr, lse=lse, o_l=o_l, a_l=a_l, h1_l=h1_l, rstd2_l=rstd2_l, gu_l=gu_l, dn_l=dn_l)
                stash.append(st)
                h_prev, branch = h1_l, dn_l
                break
            ops.attn_fwd(qkr[:, : nq * d], qkr[:, nq * d:], v, v_t, o, lse, B, S, nq, nk, d, scale, True, 0, row_iv=row_iv)
            if nf:
                # h1 = h + o Wo^T and its rows' sums of squares in ONE launch; the gate/up GEMM reads the un-normalised h1 and scales its output
                # rows by rstd2; the down projection leaves the next layer's input sum and ITS statistics the same way (the last layer's keeps
                # the stand-alone form: the tail below wants h1 and dn of the explained rows separately)
                h1, ssq = new(("h1", li), M, H), ar.get("ssq", (H // 64, M), torch.float32)
                a_raw = new(("a", li), M, H) if explicit else None      # explicit placement: the o-projection's own output (its stabiliser divides by it)
                ops.gemm_res_ssq(o, Lw["wo"], st["h"], h1, ssq, raw=a_raw)
                st["rstd2"] = ops.rms_rstd(ssq, M, H, c["rms_eps"], f32(("rstd2", li), M))
                gu, m = ops.gemm_gated_fwd_coef(h1, Lw["wgu"], new(("gu", li), M, 2 * I), wide("m", M, I), self.eps_g, self.eps["lin"], self.act,
                                                rs=st["rstd2"])
                st.update(qkv=qkv, qkr=qkr, o=o, lse=lse, a=a_raw, h1=h1, gu=gu, coef=True, dn=None)
                stash.append(st)
                if li + 1 < nL:
                    hn = new(("h", li + 1), M, H)
                    st["dn"] = new(("dn", li), M, H) if explicit else None
                    ops.gemm_res_ssq(m, Lw["wd"], h1, hn, ssq, raw=st["dn"])
                    ready = (hn, ops.rms_rstd(ssq, M, H, c["rms_eps"], f32(("rstd1", li + 1), M)))
                    h_prev, branch = hn, None
                else:
                    st["dn"] = self._lin_fwd(m, Lw["wd"], new(("dn", li), M, H))
                    h_prev, branch = h1, st["dn"]
                continue
            a = self._lin_fwd(o, Lw["wo"], new(("a", li), M, H))
            h1 = new(("h1", li), M, H)
            x2, st["rstd2"] = new("x2", M, H), f32(("rstd2", li), M)
            ops.add_rmsnorm_fwd(st["h"], a, Lw["ln2"], c["rms_eps"], hsum_out=h1, y=x2, rstd=st["rstd2"])
            if coef:      # the gated rules inside the two GEMMs around them: the backward's coefficients are stashed in gu's place (g, u never stored)
                gu, m = ops.gemm_gated_fwd_coef(x2, Lw["wgu"], new(("gu", li), M, 2 * I), wide("m", M, I), self.eps_g, self.eps["lin"], self.act)
            else:
                gu, m = ops.gemm_gated_fwd(x2, Lw["wgu"], new(("gu", li), M, 2 * I), wide("m", M, I), self.act)
            dn = self._lin_fwd(m, Lw["wd"], new(("dn", li), M, H))
            st.update(qkv=qkv, qkr=qkr, o=o, lse=lse, a=a, h1=h1, gu=gu, coef=coef, dn=dn)
            stash.append(st)
            h_prev, branch = h1, dn
        # last token only: final residual add + norm + LM head
        if h_prev.shape[0] == B and self.sparse_top and len(self.layers) > 0:
            h1_last, dn_last = h_prev, branch
        else:
            h1_last, dn_last = h_prev.index_select(0, last), branch.index_select(0, last)
        hL_last = new("hL_last", B, H)
        xn, rstd_f = ops.add_rmsnorm_fwd(h1_last, dn_last, self.norm, c["rms_eps"], hsum_out=hL_last)
        logits = self._lin_fwd(xn, self.lm_head, f32("logits", B, c["vocab"]))
        return dict(stash=stash, last=last, hL_last=hL_last, dn_last=dn_last, rstd_f=rstd_f, logits=logits, row_iv=row_iv)

    # ---------------------------------------------------------------------------------------------
    def backward(self, fw, emb, idx, B, S, layer_relevance=False, seed=None):
        c, E = self.cfg, self.eps
        H, I, d, nq, nk = c["hidden"], c["inter"], c["head_dim"], c["n_heads"], c["n_kv"]
        M, rep = B * S, nq // nk
        dev, dt = self.device, self.dtype
        nqk, nqkv = (nq + nk) * d, (nq + 2 * nk) * d
        scale = d ** -0.5
        ar = self._arena
        new = lambda tag, *s: ar.get(tag, s, dt)  # noqa: E731
        wide = lambda tag, r, c_: ar.get(tag, (r, c_), dt, pad=pitch_pad(c_, emb.element_size()))  # noqa: E731
        f32 = lambda tag, *s: ar.get(tag, s, torch.float32)  # noqa: E731
        zeros = lambda tag, *s: ar.get(tag, s, dt, zero=True)  # noqa: E731
        # LM head eps rule + final-norm identity rule on the single explained row of each prompt
        if seed is None:
            Gh_last = ops.head_seed(self.lm_head, fw["logits"], idx, self.norm, fw["rstd_f"], new("Gh_last", B, H), 0.0, E["lin"])
        else:
            # dense seed over the last-position logits (contrastive explanations): gradient in efficient mode, relevance
            # in explicit mode (coef = R / (z + eps)); G_xn = coef @ W_lm (ops.linear_dgrad: W streamed once from its stored layout),
            # then the final norm's identity rule (row scale, rmsnorm_bwd_add2 without a residual)
            coef = seed.to(device=dev, dtype=torch.float32).reshape(B, -1).contiguous()
            if E["lin"] != 0.0:
                coef = ops.eps_scale(coef, fw["logits"], 1.0, E["lin"], relevance=True)
            g_xn = ops.linear_dgrad(coef.to(dt), self.lm_head, out_dtype=torch.float32)
            Gh_last = ops.head_norm_bwd(g_xn, self.norm, fw["rstd_f"], new("Gh_last", B, H))
        # add2 at h_L = h1 + dn and the eps scale of the last down_proj, still one row per prompt
        Gs_last, A_last = new("Gs_last", B, H), new("A_last", B, H)
        rel_last = f32("rel_last", B) if layer_relevance else None
        ops.rmsnorm_bwd_add2(Gh_last, None, None, None, fw["hL_last"], fw["dn_last"], Gs_last, A_last, rel_last,
                             0.0, E["add"], E["lin"])
        last, row_iv = fw["last"], fw["row_iv"]
        top_sparse = bool(fw["stash"]) and fw["stash"][-1].get("top", False)
        if not top_sparse:
            Gs = zeros(("Gs", len(self.layers) & 1), M, H).index_copy_(0, last, Gs_last)
            Adn = zeros(("Adn", len(self.layers) & 1), M, H).index_copy_(0, last, A_last)
        layer_R = [rel_last] if layer_relevance else None
        plain_add = E["add"] == 0.0 and E["lin"] == 0.0          # efficient placement: add2 / Linear eps factors are exactly 1
        nfb = plain_add and self._norm_fused(M)                  # K1n in the backward (independent of what the forward ran: both need only rstd)
        # attn_bwd_prep folded away (efficient placement, the bf16 kernels that take every operand token-major, M = B S rows)
        fuse_prep = (plain_add and E["pv"] == 0.0 and E["mask"] == 0.0 and E["qk"] == 0.0 and not self.attn_t and ops.attn_dq_d_ok(dt, d)
                     and bool(self.layers) and ops.norm_fused_ok(M, nq * d, H, H, self.layers[0]["wo"].stride(0), True, dt))      # (Aa [M, H] contiguous)
        half = ar.get("half", (M,), torch.float32).fill_(0.5) if fuse_prep else None
        fuse_rope = fuse_prep and ops.ROPE_BWD_FUSION and E["rope"] == 0.0 and E["lin"] == 0.0 and d in (64, 128) and S <= self.max_seq

        for li in range(len(self.layers) - 1, -1, -1):
            Lw, st = self.layers[li], fw["stash"][li]
            qkv, qkr = st["qkv"], st["qkr"]
            q_begin = 0
            if st.get("top", False):
                # ---- one row per prompt through MLP, norm/add2 and o-proj; scatter into the dense attention inputs
                gu_l = st["gu_l"]
                Agu = ops.gemm_gated_bwd(A_last, Lw["wd"], gu_l, new("Agu_l", B, 2 * I), self.eps_g, E["lin"], self.act)
                Gx2 = self._lin_bwd(Agu, Lw["wgu"], new("Gx2_l", B, H))
                Gs1_l, Aa_l = new("Gs1_l", B, H), new("Aa_l", B, H)
                ops.rmsnorm_bwd_add2(Gs_last, Gx2, Lw["ln2"], st["rstd2_l"], st["h1_l"], st["a_l"], Gs1_l, Aa_l, None, 0.0,
                                     E["add"], E["lin"])
                Gof_l = self._lin_bwd(Aa_l, Lw["wo"], new("Gof_l", B, nq * d))
                Gho_l = new("Gho_l", B, nq * d)
                D_l = f32("D_l", B, nq, 1)
                ops.attn_bwd_prep(Gof_l, st["o_l"], Gho_l, D_l, B, 1, nq, d, E["pv"], 0.5)
                # scatter the one live row per prompt into the dense [M, .] operands of the attention backward (library launches only:
                # zero fill + row scatter; D's live column S-1 is a strided 2-D scatter of the same kind)
                Gho = zeros("Gho", M, nq * d).index_copy_(0, last, Gho_l)
                D = ar.get("D", (B, nq, S), torch.float32, zero=True)
                D.view(B * nq, S)[:, S - 1].copy_(D_l.view(B * nq))
                Gs1 = zeros("Gs1", M, H).index_copy_(0, last, Gs1_l)
                q_begin = S - 1
            else:
                gu = st["gu"]
                # ---- MLP
                if st.get("coef", False):
                    Agu = ops.gemm_gated_bwd_coef(Adn, Lw["wd"], gu, wide("Agu", M, 2 * I))
                else:
                    Agu = ops.gemm_gated_bwd(Adn, Lw["wd"], gu, wide("Agu", M, 2 * I), self.eps_g, E["lin"], self.act)
                Gs1 = new("Gs1", M, H)
                if nfb and ops.norm_fusion_part("bwd_gu"):               # K1n: Gs1 = rstd2 (.) (Agu W'gu) + Gs in the dgrad GEMM's epilogue
                    Aa = ops.gemm_nn_rs_res(Agu, Lw["wgu"], st["rstd2"], Gs, Gs1)
                else:
                    Gx2 = self._lin_bwd(Agu, Lw["wgu"], new("Gx2", M, H))
                    if plain_add:     # no stabiliser on the add / the branch's Linear: the branch gradient IS the residual gradient
                        ops.rmsnorm_bwd_add2(Gs, Gx2, Lw["ln2"], st["rstd2"], None, None, Gs1, None, None, 0.0, 0.0, 0.0)
                        Aa = Gs1
                    else:
                        Aa = new("Aa", M, H)
                        ops.rmsnorm_bwd_add2(Gs, Gx2, Lw["ln2"], st["rstd2"], st["h1"], st["a"], Gs1, Aa, None, 0.0, E["add"], E["lin"])
                # ---- attention
                Gho = new("Gho", M, nq * d)
                D = f32("D", B, nq, S)
                if fuse_prep:
                    # no attn_bwd_prep pass: Gho = 1/2 (Aa Wo) straight out of the o-projection's dgrad (row scale 1/2: exact), and the dQ
                    # kernel forms D = rowsum(Gho (*) o) from the rows it loads anyway and leaves it for the dK / dV kernel
                    ops.gemm_nn_rs(Aa, Lw["wo"], half, Gho)
                else:
                    Gof = self._lin_bwd(Aa, Lw["wo"], new("Gof", M, nq * d))
                    ops.attn_bwd_prep(Gof, st["o"], Gho, D, B, S, nq, d, E["pv"], 0.5)
            q, k, v = qkr[:, : nq * d], qkr[:, nq * d:], qkv[:, nqk:]
            k_t = q_t = Gho_t = None
            if self.attn_t:        # kernels that read head-transposed copies (fp32, head dims other than 128)
                k_t = ops.transpose_heads(k, B, S, nk, d)
                q_t = ops.transpose_heads(q, B, S, nq, d)
                Gho_t = ops.transpose_heads(Gho, B, S, nq, d)
            dk_h, dv_h = new("dk_h", M, nq * d), new("dv_h", M, nq * d)
            # (row pitch off the 4-KiB grid: the dQ kernel stores one row segment per lane straight into it, and 12 KiB would put them all on the
            # same channels -- dqk's 10 KiB never did)
            Aqkv = ar.get("Aqkv", (M, nqkv), dt, pad=self._pitches()["Aqkv_pad"])
            if fuse_prep and q_begin == 0 and fuse_rope:
                # RoPE's backward rides on the dQ store and on dK's group sum (no rope_bwd pass, no dqk round trip): both write Aqkv directly
                ops.attn_bwd_dq_d(q, k, v, Gho, st["o"], st["lse"], D, Aqkv[:, : nq * d], B, S, nq, nk, d, scale, row_iv=row_iv,
                                  rope=(self.cos, self.sin))
                ops.attn_bwd_dkv(q, k, v, None, Gho, None, st["lse"], D, dk_h, dv_h, B, S, nq, nk, d, scale, 0.0, 0.0, row_iv=row_iv)
                ops.gqa_reduce_rope(dk_h, Aqkv[:, nq * d: nqk], M, S, nk, rep, d, self.cos, self.sin)
                ops.gqa_reduce(dv_h, Aqkv[:, nqk:], M, nk, rep, d)
            else:
                dqk = new("dqk", M, nqk) if q_begin == 0 else zeros("dqk", M, nqk)
                if fuse_prep and q_begin == 0:
                    ops.attn_bwd_dq_d(q, k, v, Gho, st["o"], st["lse"], D, dqk[:, : nq * d], B, S, nq, nk, d, scale, row_iv=row_iv)
                else:
                    ops.attn_bwd_dq(q, k, v, k_t, Gho, st["lse"], D, dqk[:, : nq * d], B, S, nq, nk, d, scale, E["mask"], E["qk"],
                                    q_begin=q_begin, row_iv=row_iv)
                ops.attn_bwd_dkv(q, k, v, q_t, Gho, Gho_t, st["lse"], D, dk_h, dv_h, B, S, nq, nk, d, scale, E["mask"], E["qk"],
                                 q_begin=q_begin, row_iv=row_iv)
                ops.gqa_reduce(dk_h, dqk[:, nq * d:], M, nk, rep, d)
                if E["lin"] == 0.0:
                    ops.gqa_reduce(dv_h, Aqkv[:, nqk:], M, nk, rep, d)
                    ops.rope_bwd(dqk, None, None, Aqkv[:, :nqk], self.cos, self.sin, S, nq + nk, d, 0.0, 0.0)
                else:
                    dv = ops.gqa_reduce(dv_h, new("dv", M, nk * d), M, nk, rep, d)
                    ops.eps_scale2d(dv, v, Aqkv[:, nqk:], 1.0, E["lin"])
                    ops.rope_bwd(dqk, qkr, qkv[:, :nqk], Aqkv[:, :nqk], self.cos, self.sin, S, nq + nk, d, E["rope"], E["lin"])
            rel = f32(("rel", li), M) if layer_relevance else None
            if nfb and ops.norm_fusion_part("bwd_qkv"):
                # K1n: Gs = rstd1 (.) (Aqkv W'qkv) + Gs1 -- the input norm's identity rule and the residual add in the qkv dgrad's epilogue.  The
                # layer relevance sum_j h_j G_j (a diagnostic) is then one read-out pass over the stored gradient: the SAME kernels serve the call
                # with and without it (a prompt's relevance must not depend on the options of the call)
                Gs = ops.gemm_nn_rs_res(Aqkv, Lw["wqkv"], st["rstd1"], Gs1, new(("Gs", li & 1), M, H))
                Adn = Gs
                if layer_relevance:
                    layer_R.append(ops.readout(st["h"], Gs, out=rel))
                continue
            Gx = self._lin_bwd(Aqkv, Lw["wqkv"], new("Gx", M, H))
            # ---- input norm + the residual add below (or the embedding)
            if li > 0:
                prev = fw["stash"][li - 1]
                Gs = new(("Gs", li & 1), M, H)
                if plain_add:
                    ops.rmsnorm_bwd_add2(Gs1, Gx, Lw["ln1"], st["rstd1"], st["h"] if layer_relevance else None, None, Gs, None, rel, 0.0, 0.0, 0.0)
                    Adn = Gs
                else:
                    Adn = new(("Adn", li & 1), M, H)
                    ops.rmsnorm_bwd_add2(Gs1, Gx, Lw["ln1"], st["rstd1"], st["h"], prev["dn"], Gs, Adn, rel, 0.0, E["add"], E["lin"])
            else:
                Gs = new(("Gs", 0), M, H)
                ops.rmsnorm_bwd_add2(Gs1, Gx, Lw["ln1"], st["rstd1"], st["h"] if layer_relevance else None, None, Gs, None,
                                     rel, 0.0, 0.0, 0.0)
            if layer_relevance:
                layer_R.append(rel)
        return Gs, layer_R

    # ---------------------------------------------------------------------------------------------
    def _run(self, input_ids, emb, B, S, row_iv, idx, layer_relevance, return_G, seed):
        """forward + backward + read-out on the current stream: library launches only, no host synchronisation (capturable)"""
        if emb is None:
            emb = self.embed.index_select(0, input_ids.reshape(-1))
        fw = self.forward(emb, B, S, row_iv)
        if idx is None:
            # (a dense seed explains no single logit; idx / logit then report the arg-max for convenience)
            idx, _ = ops.argmax_rows(fw["logits"])
        G, layer_R = self.backward(fw, emb, idx, B, S, layer_relevance, seed=seed)
        R_tok = ops.readout(emb, G).view(B, S)
        logits = fw["logits"].clone()                            # (the arena's buffer is overwritten by the next call)
        out = dict(idx=idx, logit=logits.gather(1, idx.long()[:, None])[:, 0], R_tok=R_tok, logits=logits)
        if layer_relevance:
            rows = [layer_R[0]] + [r.view(B, S).sum(1) for r in layer_R[1:]]
            out["layer_R"] = torch.stack(rows[::-1], 0)          # [L+1, B], index 0 = embedding
        if return_G:
            out["G_emb"] = G.view(B, S, -1).clone()
            out["emb"] = emb.view(B, S, -1)
        return out

    @torch.no_grad()
    def explain(self, input_ids=None, inputs_embeds=None, target=None, layer_relevance=False, return_G=False, lengths=None,
                seed=None, graph=False):
        """input_ids [B,S] (or inputs_embeds [B,S,H]); target: None (arg-max of the last position) or
        int tensor [B].  Returns dict(idx [B], logit [B], R_tok [B,S] fp32, and optionally
        layer_R [L+1, B] (sum_h h (*) G_h at every residual-stream boundary) and G_emb [B,S,H]).
        lengths [B] (optional): prompts of different lengths in ONE call, LEFT-padded to S (prompt b occupies columns
        S-lengths[b] .. S-1, so every prompt's explained position is still the last column).  Pad keys are masked out
        through the attention kernels' per-row key intervals; RoPE is relative, so the result equals the un-padded
        single-prompt explanation up to rounding.  R_tok is exactly 0 at pad positions.
        seed [B,V] (optional, instead of target): what the reference's protocol passes to `logits[:, -1].backward(seed)` --
        a gradient over the last-position logits in efficient mode (contrastive explanations), a relevance over them
        in explicit mode.
        graph=True (input_ids only, no lengths / seed / return_G): the ~1500 launches of one explanation of this (B, S) are captured
        once into a hipGraph and replayed; the returned tensors are the graph's static outputs (copy them before the next call)."""
        if inputs_embeds is None:
            input_ids = input_ids.to(self.device)
            B, S = input_ids.shape
            emb = None
        else:
            B, S = inputs_embeds.shape[:2]
            emb = inputs_embeds.to(device=self.device, dtype=self.dtype).reshape(B * S, -1).contiguous()
        if S > self.max_seq:
            raise ValueError(f"sequence length {S} exceeds max_seq={self.max_seq}")
        row_iv = None
        if lengths is not None:
            lens = torch.as_tensor(lengths, device=self.device).to(torch.int32).reshape(B)
            if int(lens.min()) < 1 or int(lens.max()) > S:
                raise ValueError("lengths must lie in [1, S]")
            i = torch.arange(S, device=self.device, dtype=torch.int32)
            first = (S - lens)[:, None]
            lo = first.expand(B, S).contiguous()
            hi = torch.where(i[None] >= first, (i + 1)[None].expand(B, S), torch.zeros_like(lo)).contiguous()   # pad rows: empty
            row_iv = (lo, hi)
        V = self.cfg["vocab"]
        idx = None
        if target is not None:
            tgt = torch.as_tensor(target).reshape(-1).cpu().long()
            if tgt.numel() != B or int(tgt.min()) < 0 or int(tgt.max()) >= V:
                raise ValueError(f"target must hold {B} vocabulary indices in [0, {V})")
            idx = tgt.to(device=self.device, dtype=torch.int32).contiguous()
        if seed is not None:
            if target is not None:
                raise ValueError("pass either target or seed, not both")
            if tuple(seed.shape) != (B, V):
                raise ValueError(f"seed must have shape ({B}, {V}), got {tuple(seed.shape)}")
        if not graph:
            return self._run(input_ids, emb, B, S, row_iv, idx, layer_relevance, return_G, seed)
        if emb is not None or lengths is not None or seed is not None or return_G:
            raise ValueError("graph=True takes input_ids only (no inputs_embeds / lengths / seed / return_G)")
        if not hasattr(self, "_graphs") or self._graphs is None:
            self._graphs = {}
        key = (B, S, idx is not None, bool(layer_relevance), self.mode)
        g = self._graphs.get(key)
        # a graph captured before the arena re-allocated one of its buffers (a later, larger call) points into freed memory: drop it
        # and capture again against the arena as it is now
        if g is not None and g[4] != self._arena.gen:
            del self._graphs[key]
            g = None
        if g is None:
            s_ids = input_ids.clone()
            s_idx = idx.clone() if idx is not None else None
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):                        # warm-up outside the capture: sizes the arena, loads the kernels
                self._run(s_ids, None, B, S, None, s_idx, layer_relevance, False, None)
            torch.cuda.current_stream(self.device).wait_stream(side)
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg):
                out = self._run(s_ids, None, B, S, None, s_idx, layer_relevance, False, None)
            g = self._graphs[key] = (cg, s_ids, s_idx, out, self._arena.gen)
        cg, s_ids, s_idx, out, _ = g
        s_ids.copy_(input_ids)
        if s_idx is not None:
            s_idx.copy_(idx)
        cg.replay()
        return out
