import sys, warnings, torch
sys.path.insert(0, '/root/repo')
import lxt_amd, lxt_amd.engine as E
import lxt_amd.efficient.patches as P
from lxt_amd.efficient import monkey_patch
from transformers.models.llama import modeling_llama
from oracle import llama as ol
from tests.golden.hf_models import build_llama_from_weights
from tests.util import nmax
cfg = dict(hidden=2048, inter=5632, n_layers=3, n_heads=16, n_kv=4, head_dim=128, vocab=1024, rope_theta=1e4, rms_eps=1e-5)
W = ol.random_weights(cfg, seed=77)
g = torch.Generator().manual_seed(78)
for L in W["layers"]:
    L["ln1"] = (0.25 + 1.5 * torch.rand(cfg["hidden"], generator=g))
    L["ln2"] = (0.25 + 1.5 * torch.rand(cfg["hidden"], generator=g))
B, S = 3, 2048
ids = torch.randint(0, cfg["vocab"], (B, S), generator=g).cuda()
ref = E.LlamaLRP(cfg, W, dtype=torch.float32, mode="efficient", max_seq=S).explain(ids)
tgt = ref["idx"]
eb = E.LlamaLRP(cfg, W, dtype=torch.bfloat16, mode="efficient", max_seq=S, sparse_top=False).explain(ids, target=tgt)
print("engine bf16 vs engine fp32", nmax(eb["R_tok"], ref["R_tok"]), "logit", ref["logit"], eb["logit"])
with warnings.catch_warnings():
    warnings.simplefilter("ignore"); monkey_patch(modeling_llama)
hs_ref = None
for dtype, fuse in ((torch.float32, True), (torch.bfloat16, True), (torch.bfloat16, False)):
    model = build_llama_from_weights(cfg, W, attn="sdpa", dtype=dtype, rotary_fp32=True).cuda()
    P.FUSE_LAYER = fuse
    e = model.get_input_embeddings()(ids).detach().requires_grad_()
    out = model(inputs_embeds=e, use_cache=False, output_hidden_states=True)
    last = out.logits[:, -1]
    last[torch.arange(B), tgt.long()].sum().backward()
    R = (e * e.grad).float().sum(-1)
    hs = [h.detach().float() for h in out.hidden_states]
    if hs_ref is None:
        hs_ref = hs
    else:
        print("   hidden states rel L2 vs fp32:", [f"{float((a - b).norm() / b.norm()):.2e}" for a, b in zip(hs, hs_ref)], "last-token rows:",
              [f"{float((a[:, -1] - b[:, -1]).norm() / b[:, -1].norm()):.2e}" for a, b in zip(hs, hs_ref)])
    print(dtype, "fuse", fuse, "vs engine fp32", nmax(R, ref["R_tok"]), "vs engine bf16", nmax(R, eb["R_tok"]), "logit", last[torch.arange(B), tgt.long()].float().tolist())
