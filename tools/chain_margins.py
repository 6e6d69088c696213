"""Print how far the free-running sample chains sit from their test bounds
(tests/test_gpu_parity.py::test_sample_chain_external_noise), so the bounds
can be chosen from measurements.  GPU box only."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from test_gpu_parity import load_golden, sub, cu, diff_kernels, den_kernels, orc
from timediff_amd.ops import SamplerKernels

dev = torch.device("cuda:0")
for name in ["eicu_lstm_T8", "k3mix_gru_T6", "hirid_lstm_uni_T5"]:
    g = load_golden("sample_chain_" + name)
    L, H, N, T, bidir, gru = [int(v) for v in g["cfg"]]
    tables = {k: g["sd." + k] for k in orc._TABLE_MAP.values()}
    sd = sub(g, "sd.model.")
    c_io = len(g["num_idx"]) + int(g["K"].sum())
    dk = diff_kernels(g, L, T, dev, tables=tables, data_channels=c_io - len(g["cat_idx"]))
    sampler = SamplerKernels(den_kernels(sd, dev), dk)
    noise = np.concatenate([g["z_T"][None], g["noise"]])
    uniforms = np.concatenate([g["u_prior"][None], g["uniforms"]])
    out = sampler.sample(g["z_T"].shape[0], cu(noise, dev), cu(uniforms, dev)).cpu().numpy()
    od = orc.Diffusion(len(g["num_idx"]), g["K"], L, T, tables=tables)
    o16 = orc.Denoiser(sd, bf16=True)
    z, idx = od.sample_chain(o16, g["z_T"], od.prior_categorical(g["u_prior"]), g["noise"], g["uniforms"])
    want = od.decode(z, idx, g["num_idx"], g["cat_idx"], channels=c_io)
    cat, num = list(g["cat_idx"]), list(g["num_idx"])
    print(name, "cells", out[:, cat].size,
          "cat-vs-oracle16 %.5f" % (out[:, cat] != want[:, cat]).mean(),
          "num-vs-oracle16 %.3e" % np.abs(out[:, num] - want[:, num]).max(),
          "cat-vs-ref32 %.5f" % (out[:, cat] != g["sample"][:, cat]).mean(),
          "num-vs-ref32 %.3e" % np.abs(out[:, num] - g["sample"][:, num]).max(), flush=True)
