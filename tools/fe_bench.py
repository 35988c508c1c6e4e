import sys, time, os, numpy as np
sys.path.insert(0, '.')
import torch
from apr_amd import model
from apr_amd._capi import FLAVOUR_GO
dev = torch.device("cuda", 0)
fe = model.Engine(None, 0, 1024)
n = 1024
pcm = np.concatenate([model.synth_pcm(5, i, 65) for i in range(16)]).view(np.uint8)
d_pcm = torch.from_numpy(pcm).to(dev)
d_out = torch.empty((n, 128, 188), dtype=torch.float32, device=dev)
fe.prepare_batch_device(d_pcm, d_pcm.numel(), n, 96000, d_out, FLAVOUR_GO); fe.sync()
fe.profile_reset(); fe.profile_enable(True)
for _ in range(10): fe.prepare_batch_device(d_pcm, d_pcm.numel(), n, 96000, d_out, FLAVOUR_GO)
p = fe.profile_read()
print(os.environ.get("APR_AMD_LIB", "default"), os.environ.get("APR_FE_FEW", ""), [(s["name"], round(1e3 * s["ms"] / s["launches"], 1)) for s in p if s["launches"]], float(d_out.double().sum().item()))
