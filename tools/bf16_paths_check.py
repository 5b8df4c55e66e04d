import os, sys, warnings, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
warnings.simplefilter("ignore")
from transformers import LlamaConfig, LlamaForCausalLM
from transformers.models.llama import modeling_llama
from lxt_amd.efficient import monkey_patch
import lxt_amd.engine as E
from tests.util import nmax
monkey_patch(modeling_llama)
L = int(os.environ.get("LAYERS", 2)); S = int(os.environ.get("SEQ", 2048)); V = int(os.environ.get("VOCAB", 128256))
cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=L, num_attention_heads=32, num_key_value_heads=8,
                  vocab_size=V, rms_norm_eps=1e-5, max_position_embeddings=8192, tie_word_embeddings=False,
                  rope_parameters=dict(rope_type="default", rope_theta=500000.0), attn_implementation="sdpa")
torch.manual_seed(0)
with torch.device("cuda"):
    m32 = LlamaForCausalLM(cfg).eval()
for p in m32.parameters():
    p.requires_grad_(False)
ids = torch.randint(0, V, (1, S), generator=torch.Generator().manual_seed(1234)).cuda()
eng32 = E.LlamaLRP.from_hf(m32, mode="efficient", max_seq=S)
o32 = eng32.explain(ids, return_G=True)
tgt = o32["idx"].long()
G32, R32 = o32["G_emb"][0].float(), o32["R_tok"][0]
del eng32
m16 = m32.to(torch.bfloat16)
eng16 = E.LlamaLRP.from_hf(m16, mode="efficient", max_seq=S)
o16 = eng16.explain(ids, target=tgt, return_G=True)
e = m16.get_input_embeddings()(ids).detach().requires_grad_()
hs = {}
logits = m16(inputs_embeds=e, use_cache=False, logits_to_keep=1, output_hidden_states=True)
lg = logits.logits
lg[0, -1, tgt[0]].backward()
Rd, Gd = (e * e.grad).float().sum(-1)[0], e.grad[0].float()
print(f"L={L} S={S}: logit fp32 {float(o32['logit'][0]):.5f} | bf16 engine {float(o16['logit'][0]):.5f} | bf16 drop-in {float(lg[0,-1,tgt[0]]):.5f}")
print(f"  G_emb vs fp32 engine: bf16 engine {nmax(o16['G_emb'][0].float(), G32):.2e} | bf16 drop-in {nmax(Gd, G32):.2e}")
print(f"  R_tok vs fp32 engine: bf16 engine {nmax(o16['R_tok'][0], R32):.2e} | bf16 drop-in {nmax(Rd, R32):.2e}")
rowerr = (Gd - G32).abs().max(1).values / G32.abs().max()
print("  drop-in G row error: last 4 rows", [f"{float(v):.1e}" for v in rowerr[-4:]], " first 4 rows", [f"{float(v):.1e}" for v in rowerr[:4]], " worst row", int(rowerr.argmax()))
