"""dev: a few BertLRP explanations (for rocprofv3 --kernel-trace --stats): python tools/bert_engine_run.py <B> <dtype> <n>"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from tests.golden.hf_models import build_bert
from lxt_amd.engine_bert import BertLRP
B, dtype, n = int(sys.argv[1]), getattr(torch, sys.argv[2]), int(sys.argv[3])
eng = BertLRP.from_hf(build_bert(seed=0, attn="eager"), dtype=dtype, mode="efficient")
ids = torch.randint(0, 30522, (B, 128), generator=torch.Generator().manual_seed(1)).cuda()
for _ in range(n):
    eng.explain(ids)
torch.cuda.synchronize()
