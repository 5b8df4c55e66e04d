"""Seeded construction of the HuggingFace models the drop-in fixtures are defined on (shared by the
fixture generator and the tests; weights come from the seed, a checksum in the fixture pins them)."""
import torch


def wsum(model):
    return float(sum(p.detach().double().abs().sum() for p in model.parameters()))


def build_bert(seed=0, attn="eager"):
    from transformers import BertConfig, BertForSequenceClassification
    torch.manual_seed(seed)
    cfg = BertConfig(num_labels=2, attn_implementation=attn)            # BERT-base defaults: 12 L, H 768, 12 heads, I 3072
    return BertForSequenceClassification(cfg).eval()


def build_gemma3(seed=3, attn="eager"):
    from transformers import Gemma3TextConfig, Gemma3ForCausalLM
    torch.manual_seed(seed)
    cfg = Gemma3TextConfig(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=4, num_attention_heads=4,
                           num_key_value_heads=2, head_dim=32, sliding_window=32, max_position_embeddings=512,
                           layer_types=["sliding_attention", "sliding_attention", "sliding_attention", "full_attention"],
                           query_pre_attn_scalar=32, attn_implementation=attn, tie_word_embeddings=False)
    return Gemma3ForCausalLM(cfg).eval()


def build_gemma3_4bdims(layers=2, seed=5, attn="eager", vocab=4096):
    """Gemma-3-4B-it TEXT-tower layer dimensions (BASELINE config 4; public model card: H 2560, 8 query / 4 kv heads of d = 256, I 10240,
    sliding window 1024, query_pre_attn_scalar 256, tied embeddings), `layers` decoder layers alternating local / global, small
    vocabulary, seeded init with non-trivial norm weights (HF initialises Gemma's (1 + w) norm weights to zero)"""
    from transformers import Gemma3TextConfig, Gemma3ForCausalLM
    torch.manual_seed(seed)
    cfg = Gemma3TextConfig(vocab_size=vocab, hidden_size=2560, intermediate_size=10240, num_hidden_layers=layers, num_attention_heads=8,
                           num_key_value_heads=4, head_dim=256, sliding_window=1024, max_position_embeddings=4096,
                           layer_types=["sliding_attention", "full_attention"][:layers], query_pre_attn_scalar=256,
                           attn_implementation=attn, tie_word_embeddings=True)
    m = Gemma3ForCausalLM(cfg).eval()
    with torch.no_grad():
        g = torch.Generator().manual_seed(seed + 1)
        for n_, p_ in m.named_parameters():
            if "norm" in n_:
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.1)
    return m


def build_llama_8bdims(layers=8, seed=17, attn="sdpa", vocab=4096):
    """Llama-3-8B layer dimensions (H 4096, I 14336, 32 query / 8 kv heads of d = 128, rope theta 5e5), `layers` decoder layers, small
    vocabulary, HF's default seeded init: the depth / bf16 conditioning fixture (make_golden_llama_bf16_depth.py)"""
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(seed)
    kw = dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=layers, num_attention_heads=32, num_key_value_heads=8,
              vocab_size=vocab, rms_norm_eps=1e-5, max_position_embeddings=8192, tie_word_embeddings=False, attn_implementation=attn)
    try:
        cfg = LlamaConfig(rope_parameters=dict(rope_type="default", rope_theta=500000.0), **kw)
    except TypeError:
        cfg = LlamaConfig(rope_theta=500000.0, **kw)
    return LlamaForCausalLM(cfg).eval()


def build_gemma3_mm(seed=11, attn="eager"):
    """tiny Gemma3ForConditionalGeneration: text tower (sliding + global layers) + SigLIP tower (head_dim 72 like the real
    SigLIP-So400m: 1152 / 16) + multi-modal projector.  HF leaves the projector weight at zeros on random init -> seeded here."""
    from transformers import Gemma3Config, Gemma3ForConditionalGeneration
    torch.manual_seed(seed)
    text = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2, head_dim=32,
                vocab_size=300, sliding_window=24, layer_types=["sliding_attention", "sliding_attention", "full_attention"],
                max_position_embeddings=512, query_pre_attn_scalar=32)
    vision = dict(hidden_size=144, intermediate_size=192, num_hidden_layers=2, num_attention_heads=2, image_size=56, patch_size=14,
                  num_channels=3)
    cfg = Gemma3Config(text_config=text, vision_config=vision, mm_tokens_per_image=4, image_token_id=299, boi_token_id=297,
                       eoi_token_id=298, attn_implementation=attn)
    m = Gemma3ForConditionalGeneration(cfg).eval()
    with torch.no_grad():
        g = torch.Generator().manual_seed(seed + 1)
        w = m.model.multi_modal_projector.mm_input_projection_weight
        w.copy_(torch.randn(w.shape, generator=g) * 0.05)
    return m


def build_gemma3_mm_fulldims(vision_layers=2, text_layers=2, seed=13, attn="sdpa"):
    """Gemma3ForConditionalGeneration at the released Gemma-3-4B-it dimensions (BASELINE config 4, image + text): SigLIP-So400m tower
    (H 1152, 16 heads of d = 72, I 4304, 896 x 896 images, patch 14 -> 4096 patches, 256 image tokens) with `vision_layers` encoder layers,
    text tower (H 2560, 8 / 4 heads of d = 256, I 10240, window 1024) with `text_layers` layers, small vocabulary; seeded, non-trivial norms"""
    from transformers import Gemma3Config, Gemma3ForConditionalGeneration
    torch.manual_seed(seed)
    text = dict(vocab_size=4096, hidden_size=2560, intermediate_size=10240, num_hidden_layers=text_layers, num_attention_heads=8,
                num_key_value_heads=4, head_dim=256, sliding_window=1024, max_position_embeddings=4096, query_pre_attn_scalar=256,
                layer_types=["sliding_attention", "full_attention"][:text_layers])
    vision = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=vision_layers, num_attention_heads=16, image_size=896,
                  patch_size=14, num_channels=3)
    cfg = Gemma3Config(text_config=text, vision_config=vision, mm_tokens_per_image=256, image_token_id=4095, boi_token_id=4093,
                       eoi_token_id=4094, attn_implementation=attn)
    m = Gemma3ForConditionalGeneration(cfg).eval()
    with torch.no_grad():
        g = torch.Generator().manual_seed(seed + 1)
        w = m.model.multi_modal_projector.mm_input_projection_weight
        w.copy_(torch.randn(w.shape, generator=g) * 0.03)
        for n_, p_ in m.named_parameters():
            if "norm" in n_ and p_.dim() == 1:               # Gemma (1 + w) weights and LayerNorm biases start at 0, LayerNorm weights at 1
                p_.add_(torch.randn(p_.shape, generator=g) * 0.1)
    return m


def gemma3_mm_fulldims_inputs(S=384):
    """one image inside a text prompt: text, <boi>, 256 image tokens, <eoi>, text; token_type_ids; pixel values"""
    ids = torch.randint(0, 4000, (1, S), generator=torch.Generator().manual_seed(5))
    ids[0, 20] = 4093
    ids[0, 21: 21 + 256] = 4095
    ids[0, 21 + 256] = 4094
    tt = (ids == 4095).long()
    pv = torch.randn(1, 3, 896, 896, generator=torch.Generator().manual_seed(6))
    return ids, tt, pv


def gemma3_mm_inputs():
    """ids with one image (boi, 4 image tokens, eoi) inside a 48-token prompt, token_type_ids, pixel values"""
    S = 48
    ids = torch.randint(0, 290, (1, S), generator=torch.Generator().manual_seed(3))
    ids[0, 10] = 297
    ids[0, 11:15] = 299
    ids[0, 15] = 298
    tt = (ids == 299).long()
    pv = torch.randn(1, 3, 56, 56, generator=torch.Generator().manual_seed(4))
    return ids, tt, pv


class MiniViTBlock(torch.nn.Module):
    """pre-LN encoder block with torch.nn.MultiheadAttention(batch_first) -- the structure of torchvision's EncoderBlock"""

    def __init__(self, dim, heads, mlp_dim):
        super().__init__()
        nn = torch.nn
        self.ln_1 = nn.LayerNorm(dim, eps=1e-6)
        self.self_attention = nn.MultiheadAttention(dim, heads, dropout=0.0, batch_first=True)
        self.dropout = nn.Dropout(0.0)
        self.ln_2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = nn.Sequential(nn.Linear(dim, mlp_dim), nn.GELU(), nn.Dropout(0.0), nn.Linear(mlp_dim, dim), nn.Dropout(0.0))

    def forward(self, x):
        y = self.ln_1(x)
        y, _ = self.self_attention(y, y, y, need_weights=False)
        x = x + self.dropout(y)
        return x + self.mlp(self.ln_2(x))


class MiniViT(torch.nn.Module):
    """patchify Conv2d (stride = kernel) + class token + learned positions + encoder blocks + LayerNorm + linear head"""

    def __init__(self, image=32, patch=8, dim=96, heads=4, mlp_dim=192, layers=3, classes=10):
        super().__init__()
        nn = torch.nn
        self.conv_proj = nn.Conv2d(3, dim, patch, patch)
        n = (image // patch) ** 2
        self.class_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embedding = nn.Parameter(torch.empty(1, n + 1, dim).normal_(std=0.02))
        self.layers = nn.ModuleList([MiniViTBlock(dim, heads, mlp_dim) for _ in range(layers)])
        self.ln = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Linear(dim, classes)

    def forward(self, x):
        x = self.conv_proj(x).flatten(2).transpose(1, 2)
        x = torch.cat([self.class_token.expand(x.shape[0], -1, -1), x], 1) + self.pos_embedding
        for blk in self.layers:
            x = blk(x)
        return self.head(self.ln(x)[:, 0])


def build_mini_vit(seed=31):
    torch.manual_seed(seed)
    m = MiniViT().eval()
    with torch.no_grad():
        for p in m.parameters():          # random biases / class token too: nothing in the explanation is trivially zero
            if p.dim() == 1 or p is m.class_token:
                p.normal_(0, 0.05)
        for blk in m.layers:
            blk.ln_1.weight.add_(1.0); blk.ln_2.weight.add_(1.0)
        m.ln.weight.add_(1.0)
    for p in m.parameters():
        p.requires_grad_(False)
    return m


def build_llama_from_weights(cfg, W, attn="eager", dtype=torch.float32, rotary_fp32=False):
    """HF LlamaForCausalLM carrying the oracle-format weights W of config cfg (oracle/llama.py: random_weights) -- the model the Llama fixtures
    (tests/golden/llama_*.npz) were captured on, as a user of the drop-in APIs would hold it.  rotary_fp32: `model.to(bfloat16)` also rounds the
    rotary embedding's inv_freq BUFFER to bf16 (positions up to S x 2^-9 relative off in phase: ~4 rad at S = 2048 for the fastest pair) -- the
    model then computes a different network; `from_pretrained(..., dtype=bfloat16)` keeps that buffer in fp32, and True restores it the same way."""
    from transformers import LlamaConfig, LlamaForCausalLM
    hc = LlamaConfig(hidden_size=cfg["hidden"], intermediate_size=cfg["inter"], num_hidden_layers=cfg["n_layers"],
                     num_attention_heads=cfg["n_heads"], num_key_value_heads=cfg["n_kv"], head_dim=cfg["head_dim"], vocab_size=cfg["vocab"],
                     rms_norm_eps=cfg["rms_eps"], max_position_embeddings=4096,
                     rope_parameters=dict(rope_type="default", rope_theta=cfg["rope_theta"]), tie_word_embeddings=False,
                     attn_implementation=attn)
    model = LlamaForCausalLM(hc).eval()
    with torch.no_grad():
        model.model.embed_tokens.weight.copy_(W["embed"])
        model.model.norm.weight.copy_(W["norm"])
        model.lm_head.weight.copy_(W["lm_head"])
        for L, Lw in zip(model.model.layers, W["layers"]):
            L.input_layernorm.weight.copy_(Lw["ln1"]); L.post_attention_layernorm.weight.copy_(Lw["ln2"])
            L.self_attn.q_proj.weight.copy_(Lw["wq"]); L.self_attn.k_proj.weight.copy_(Lw["wk"])
            L.self_attn.v_proj.weight.copy_(Lw["wv"]); L.self_attn.o_proj.weight.copy_(Lw["wo"])
            L.mlp.gate_proj.weight.copy_(Lw["wg"]); L.mlp.up_proj.weight.copy_(Lw["wu"]); L.mlp.down_proj.weight.copy_(Lw["wd"])
    for p_ in model.parameters():
        p_.requires_grad_(False)
    inv = model.model.rotary_emb.inv_freq.detach().clone()
    model = model.to(dtype)
    if rotary_fp32:
        model.model.rotary_emb.inv_freq = inv.to(model.model.rotary_emb.inv_freq.device)
    return model


def build_llama(seed=5, attn="eager"):
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(seed)
    cfg = LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                      head_dim=32, vocab_size=256, max_position_embeddings=512, attn_implementation=attn, tie_word_embeddings=False)
    return LlamaForCausalLM(cfg).eval()


def build_qwen2(seed=6, attn="eager"):
    from transformers import Qwen2Config, Qwen2ForCausalLM
    torch.manual_seed(seed)
    cfg = Qwen2Config(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                      vocab_size=256, max_position_embeddings=512, attn_implementation=attn, tie_word_embeddings=False,
                      use_sliding_window=False)
    return Qwen2ForCausalLM(cfg).eval()


def build_qwen3(seed=7, attn="eager"):
    from transformers import Qwen3Config, Qwen3ForCausalLM
    torch.manual_seed(seed)
    cfg = Qwen3Config(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                      head_dim=32, vocab_size=256, max_position_embeddings=512, attn_implementation=attn, tie_word_embeddings=False,
                      use_sliding_window=False)
    return Qwen3ForCausalLM(cfg).eval()


def build_gpt2(seed=8, attn="eager"):
    from transformers import GPT2Config, GPT2LMHeadModel
    torch.manual_seed(seed)
    cfg = GPT2Config(n_embd=128, n_layer=3, n_head=4, vocab_size=256, n_positions=512, attn_implementation=attn,
                     resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    return GPT2LMHeadModel(cfg).eval()


BUILDERS = dict(llama_cp=build_llama, qwen2=build_qwen2, qwen3=build_qwen3, gpt2=build_gpt2)
