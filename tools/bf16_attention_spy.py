import os, sys, warnings, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
warnings.simplefilter("ignore")
from transformers import LlamaConfig, LlamaForCausalLM
from transformers.models.llama import modeling_llama
from lxt_amd.efficient import monkey_patch
import lxt_amd.efficient.patches as P
monkey_patch(modeling_llama)
S, V = 2048, 4096
cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=8,
                  vocab_size=V, rms_norm_eps=1e-5, max_position_embeddings=8192, tie_word_embeddings=False,
                  rope_parameters=dict(rope_type="default", rope_theta=500000.0), attn_implementation="sdpa")
torch.manual_seed(0)
with torch.device("cuda"):
    m = LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
ids = torch.randint(0, V, (1, S), generator=torch.Generator().manual_seed(1234)).cuda()
reg = modeling_llama.ALL_ATTENTION_FUNCTIONS
orig = reg["sdpa"]
def spy(module, query, key, value, attention_mask=None, scaling=None, dropout=0.0, **kw):
    out, _ = orig(module, query, key, value, attention_mask=attention_mask, scaling=scaling, dropout=dropout, **kw)
    am = attention_mask
    print("q", tuple(query.shape), query.dtype, query.stride(), "| k", tuple(key.shape), key.stride(), "| v", tuple(value.shape), value.stride())
    print("mask", None if am is None else (tuple(am.shape), am.dtype, am.stride()), "scaling", scaling, "kw", {k: (v if not torch.is_tensor(v) else tuple(v.shape)) for k, v in kw.items()})
    print("plan", [None if x is None else (x if not isinstance(x, tuple) else "intervals") for x in P._mask_plan(am, query.shape[2], module, 0)])
    qf, kf, vf = query.double(), key.double().repeat_interleave(4, 1), value.double().repeat_interleave(4, 1)
    s = (qf @ kf.transpose(-1, -2)) * (scaling if scaling is not None else query.shape[-1] ** -0.5)
    s = s.masked_fill(~torch.ones(S, S, dtype=torch.bool, device="cuda").tril(), float("-inf"))
    ref = (torch.softmax(s, -1) @ vf).transpose(1, 2)
    err = (out.double() - ref).abs().reshape(S, -1).max(1).values / ref.abs().max()
    print("HIP attention vs fp64 eager on the SAME q,k,v: max", float(err.max()), "worst row", int(err.argmax()), "| |s| max", float(s[s > -1e30].abs().max()))
    return out, None
for k_ in list(reg.keys()):
    reg[k_] = spy
with torch.no_grad():
    m(input_ids=ids, use_cache=False)
