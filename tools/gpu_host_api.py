"""host-pointer API rates (PCIe inclusive) on a 64 MiB sample: python tools/gpu_host_api.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.init()
import datagen, bench
host = datagen.rep_text(64 << 20)
print(json.dumps(bench.host_api_rates("chameleon", host, 4 << 20, 64 << 20), indent=1))
