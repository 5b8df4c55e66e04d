import os, sys, warnings, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
warnings.simplefilter("ignore")
from transformers import LlamaConfig, LlamaForCausalLM
from transformers.models.llama import modeling_llama
from lxt_amd.efficient import monkey_patch
monkey_patch(modeling_llama)
S, V = 2048, 4096
cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=8,
                  vocab_size=V, rms_norm_eps=1e-5, max_position_embeddings=8192, tie_word_embeddings=False,
                  rope_parameters=dict(rope_type="default", rope_theta=500000.0), attn_implementation="sdpa")
torch.manual_seed(0)
with torch.device("cuda"):
    m = LlamaForCausalLM(cfg).eval()
for p in m.parameters():
    p.requires_grad_(False)
ids = torch.randint(0, V, (1, S), generator=torch.Generator().manual_seed(1234)).cuda()
names = ["model.layers.0.input_layernorm", "model.layers.0.self_attn.q_proj", "model.layers.0.self_attn.k_proj", "model.layers.0.self_attn.v_proj",
         "model.layers.0.self_attn.o_proj", "model.layers.0.post_attention_layernorm", "model.layers.0.mlp.gate_proj", "model.layers.0.mlp.up_proj",
         "model.layers.0.mlp.down_proj", "model.norm", "lm_head"]
def run(model):
    rec = {}
    hs = []
    mods = dict(model.named_modules())
    for n in names:
        hs.append(mods[n].register_forward_hook(lambda mod, inp, out, n=n: rec.__setitem__(n, (inp[0].detach().float().clone(), out.detach().float().clone()))))
    with torch.no_grad():
        model(input_ids=ids, use_cache=False)
    for h in hs: h.remove()
    return rec
r32 = run(m)
r16 = run(m.to(torch.bfloat16))
for n in names:
    (i32, o32), (i16, o16) = r32[n], r16[n]
    ei = float((i16 - i32).abs().max() / i32.abs().max()); eo = float((o16 - o32).abs().max() / o32.abs().max())
    rowerr = (o16 - o32).abs().reshape(-1, o32.shape[-1]).max(1).values
    print(f"{n:42s} input err {ei:.2e}  output err {eo:.2e}  worst row {int(rowerr.argmax())} of {rowerr.numel()}")
