"""dev, CPU only, plain HuggingFace (no lxt_amd, no HIP): bf16 vs fp32 of ONE Llama-3-8B-shaped layer at S=2048, module by module.
Shows that the jump at the attention output seen in the bf16 drop-in path is a property of HF's bf16 modeling arithmetic on
random-init weights, not of the HIP kernels."""
import torch, warnings
warnings.simplefilter("ignore")
from transformers import LlamaConfig, LlamaForCausalLM
torch.set_num_threads(8)
S, V = 2048, 4096
cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=8,
                  vocab_size=V, rms_norm_eps=1e-5, max_position_embeddings=8192, tie_word_embeddings=False,
                  rope_parameters=dict(rope_type="default", rope_theta=500000.0), attn_implementation="eager")
torch.manual_seed(0)
m = LlamaForCausalLM(cfg).eval()
ids = torch.randint(0, V, (1, S), generator=torch.Generator().manual_seed(1234))
names = ["model.layers.0.input_layernorm", "model.layers.0.self_attn.q_proj", "model.layers.0.self_attn.o_proj", "model.layers.0.post_attention_layernorm",
         "model.layers.0.mlp.down_proj", "lm_head"]
def run(model):
    rec = {}; hs = []
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
    rows = (i16 - i32).abs().reshape(-1, i32.shape[-1]).max(1).values / i32.abs().max()
    print(f"{n:42s} input err {ei:.2e} (median row {float(rows.median()):.1e})  output err {eo:.2e}")
