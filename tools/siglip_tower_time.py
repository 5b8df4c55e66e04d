#!/usr/bin/env python3
"""Dev tool (GPU box): the SigLIP tower of bench.py's config-4 image probe alone (27 layers, 4 images of 896 x 896): wall time of forward + backward,
host time to issue them, and the per-phase split."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from lxt_amd.engine_gemma3_mm import SiglipLRP  # noqa: E402


def main():
    dev, dtype = torch.device("cuda:0"), torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(7)
    Lv, Hv, Iv, nh, img, pt, T, Ht = 27, 1152, 4304, 16, 896, 14, 256, 2560
    rn = lambda sd, *s: (torch.randn(*s, generator=g, device=dev) * sd).to(dtype)  # noqa: E731
    P = (img // pt) ** 2
    W = dict(patch_w=rn(0.02, Hv, 3, pt, pt), patch_b=rn(0.02, Hv), pos=rn(0.02, P, Hv), post_w=1 + rn(0.1, Hv), post_b=rn(0.1, Hv),
             proj_norm=rn(0.1, Hv), proj_w=rn(0.03, Hv, Ht), layers=[
        dict(ln1_w=1 + rn(0.1, Hv), ln1_b=rn(0.1, Hv), ln2_w=1 + rn(0.1, Hv), ln2_b=rn(0.1, Hv), wq=rn(0.02, Hv, Hv), bq=rn(0.02, Hv),
             wk=rn(0.02, Hv, Hv), bk=rn(0.02, Hv), wv=rn(0.02, Hv, Hv), bv=rn(0.02, Hv), wo=rn(0.02, Hv, Hv), bo=rn(0.02, Hv),
             w1=rn(0.02, Iv, Hv), b1=rn(0.02, Iv), w2=rn(0.02, Hv, Iv), b2=rn(0.02, Hv)) for _ in range(Lv)])
    vcfg = dict(hidden=Hv, inter=Iv, n_layers=Lv, n_heads=nh, image=img, patch=pt, channels=3, ln_eps=1e-6, act="gelu_tanh",
                tokens_per_image=T, text_hidden=Ht, image_token_id=1)
    vis = SiglipLRP(vcfg, W, dtype=dtype, device=dev, vision_attn_rule=True)
    del W
    pix = torch.randn(4, 3, img, img, generator=g, device=dev).to(dtype)
    G = torch.randn(4 * T, Ht, generator=g, device=dev).to(dtype)
    for _ in range(2):
        fw = vis.forward(pix)
        vis.backward(fw, G)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        fw = vis.forward(pix)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        vis.backward(fw, G)
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print(f"tower forward: issue {1e3 * (t1 - t0):6.1f} ms, done {1e3 * (t2 - t0):6.1f} ms | backward: issue {1e3 * (t3 - t2):6.1f} ms, done {1e3 * (t4 - t2):6.1f} ms "
              f"| total {1e3 * (t4 - t0):6.1f} ms", flush=True)


if __name__ == "__main__":
    main()
