"""Worker (one process per model family -- the patches are class-level and process-global, like the
reference's): patch a HF modeling module with lxt_amd, run the quickstart protocol on the GPU and print
the normalised max error against the fixture captured from the real reference."""
import importlib
import os
import sys
import warnings

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
warnings.simplefilter("ignore")
from tests.golden.hf_models import BUILDERS, wsum  # noqa: E402
from tests.util import load, t, nmax  # noqa: E402


def main(which):
    fx = load(f"hf_{which}.npz")
    fam = which.replace("_cp", "")
    mod = importlib.import_module(f"transformers.models.{fam}.modeling_{fam}")
    from lxt_amd.efficient import monkey_patch
    if which.endswith("_cp"):
        monkey_patch(mod, importlib.import_module(f"lxt_amd.efficient.models.{fam}").cp_LRP)
    else:
        monkey_patch(mod)
    ids = t(fx["ids"])
    worst = 0.0
    for impl in ("eager", "sdpa"):
        model = BUILDERS[which](attn=impl)
        assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"]), "weights did not reproduce"
        for p in model.parameters():
            p.requires_grad_(False)
        model = model.cuda()
        e = model.get_input_embeddings()(ids[None].cuda()).requires_grad_()
        last = model(inputs_embeds=e, use_cache=False).logits[0, -1]
        idx = int(last.argmax())
        assert idx == int(fx["idx"]), (idx, int(fx["idx"]))
        last[idx].backward()
        R = (e * e.grad)[0].sum(-1)
        err = max(nmax(R, fx["R_tok"]), nmax(R, fx["R_tok_fp64"]))
        print(f"[{which}/{impl}] tok vs reference {nmax(R, fx['R_tok']):.2e} | vs oracle fp64 {nmax(R, fx['R_tok_fp64']):.2e}")
        worst = max(worst, err)
    print(f"WORST {worst:.3e}")
    return 0 if worst < 1e-4 else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
