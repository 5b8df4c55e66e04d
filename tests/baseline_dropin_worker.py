"""Worker for tests/test_baseline_size_gpu.py::test_dropin_fp32_full_width_vs_oracle (one fresh process: monkey_patch is
class-level).  Builds a HF LlamaForCausalLM at the BASELINE layer width from the oracle's synthetic weights, patches
transformers.models.llama.modeling_llama with lxt_amd.efficient.monkey_patch and runs the reference's user protocol
(docs/source/quickstart.rst:120-141) on the GPU; compares the per-token relevance with the fp64 oracle result handed over
in an .npz by the parent test."""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
warnings.simplefilter("ignore")
from oracle import llama as ol  # noqa: E402
from tests.util import nmax  # noqa: E402


def build_hf(cfg, W, attn):
    from transformers import LlamaConfig, LlamaForCausalLM
    hc = LlamaConfig(hidden_size=cfg["hidden"], intermediate_size=cfg["inter"], num_hidden_layers=cfg["n_layers"],
                     num_attention_heads=cfg["n_heads"], num_key_value_heads=cfg["n_kv"], head_dim=cfg["head_dim"],
                     vocab_size=cfg["vocab"], rms_norm_eps=cfg["rms_eps"], rope_theta=cfg["rope_theta"],
                     max_position_embeddings=8192, tie_word_embeddings=False, attention_bias=False, mlp_bias=False)
    hc._attn_implementation = attn
    with torch.device("meta"):
        model = LlamaForCausalLM(hc)
    model = model.to_empty(device="cuda")
    sd = {"model.embed_tokens.weight": W["embed"], "model.norm.weight": W["norm"], "lm_head.weight": W["lm_head"]}
    for i, L in enumerate(W["layers"]):
        pre = f"model.layers.{i}."
        sd.update({pre + "input_layernorm.weight": L["ln1"], pre + "post_attention_layernorm.weight": L["ln2"],
                   pre + "self_attn.q_proj.weight": L["wq"], pre + "self_attn.k_proj.weight": L["wk"],
                   pre + "self_attn.v_proj.weight": L["wv"], pre + "self_attn.o_proj.weight": L["wo"],
                   pre + "mlp.gate_proj.weight": L["wg"], pre + "mlp.up_proj.weight": L["wu"], pre + "mlp.down_proj.weight": L["wd"]})
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary" in m or "inv_freq" in m for m in missing), (missing, unexpected)
    # buffers created on meta (rotary inv_freq) must be re-materialised
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding
    model.model.rotary_emb = LlamaRotaryEmbedding(hc).to("cuda")
    for p in model.parameters():
        p.requires_grad_(False)
    return model.eval()


def main(path):
    fx = np.load(path)
    cfg = {k: (float(v) if k in ("rope_theta", "rms_eps") else int(v)) for k, v in zip(fx["cfg_keys"].tolist(), fx["cfg_vals"].tolist())}
    W = ol.random_weights(cfg, seed=int(fx["wseed"]))
    ids = torch.from_numpy(fx["ids"])
    from transformers.models.llama import modeling_llama
    from lxt_amd.efficient import monkey_patch
    monkey_patch(modeling_llama)
    worst = 0.0
    for impl in ("eager", "sdpa"):
        model = build_hf(cfg, W, impl)
        e = model.get_input_embeddings()(ids[None].cuda()).requires_grad_()
        last = model(inputs_embeds=e, use_cache=False).logits[0, -1]
        idx = int(last.argmax())
        assert idx == int(fx["idx"]), (idx, int(fx["idx"]))
        assert abs(float(last[idx]) - float(fx["logit"])) < 1e-4 * max(1.0, abs(float(fx["logit"])))
        last[idx].backward()
        R = (e * e.grad)[0].float().sum(-1)
        err = nmax(R, fx["R_tok"])
        print(f"[drop-in H{cfg['hidden']}/S{ids.numel()} fp32 {impl}] token relevance vs fp64 oracle {err:.2e}")
        worst = max(worst, err)
        del model, e, last, R
        torch.cuda.empty_cache()
    print(f"WORST {worst:.3e}")
    return 0 if worst < 1e-4 else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
