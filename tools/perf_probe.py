"""Quick timing probe of the denoiser forward at the eICU shape (not part of the product)."""
import sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from timediff_amd.ops import DenoiserKernels

def make_sd(H=256, C=14, N=3, D=2, dev="cuda"):
    g = torch.Generator().manual_seed(0)
    k = 1.0 / np.sqrt(H)
    sd = {}
    for l in range(N):
        for d in range(D):
            sfx = "_reverse" if d else ""
            In = C if l == 0 else D * H
            sd[f"rnn.rnn.weight_ih_l{l}{sfx}"] = (torch.rand(4 * H, In, generator=g) * 2 - 1) * k
            sd[f"rnn.rnn.weight_hh_l{l}{sfx}"] = (torch.rand(4 * H, H, generator=g) * 2 - 1) * k
            sd[f"rnn.rnn.bias_ih_l{l}{sfx}"] = (torch.rand(4 * H, generator=g) * 2 - 1) * k
            sd[f"rnn.rnn.bias_hh_l{l}{sfx}"] = (torch.rand(4 * H, generator=g) * 2 - 1) * k
    sd["time_emb.1.weight"] = torch.randn(256, 64, generator=g) / 8; sd["time_emb.1.bias"] = torch.zeros(256)
    sd["time_emb.3.weight"] = torch.randn(256, 256, generator=g) / 16; sd["time_emb.3.bias"] = torch.zeros(256)
    sd["rnn.time_mlp.1.weight"] = torch.randn(2 * D * H, 256, generator=g) / 16; sd["rnn.time_mlp.1.bias"] = torch.zeros(2 * D * H)
    sd["rnn.norm.weight"] = torch.ones(D * H); sd["rnn.norm.bias"] = torch.zeros(D * H)
    sd["fc.weight"] = torch.randn(C, D * H, generator=g) / np.sqrt(D * H); sd["fc.bias"] = torch.zeros(C)
    return {k_: v.to(dev) for k_, v in sd.items()}

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    L, C = 272, 14
    den = DenoiserKernels(make_sd())
    for B in [int(a) for a in sys.argv[1:]] or [4096]:
        x = torch.randn(B, C, L, device=dev)
        t = torch.randint(0, 1000, (B,), device=dev)
        film = den.time_film(t)
        y = den.forward(x, film=film)
        torch.cuda.synchronize()
        n = 3
        t0 = time.time()
        for _ in range(n):
            y = den.forward(x, film=film)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / n
        flops = 2.017e9 * B
        print(f"B={B} fwd {dt*1e3:.2f} ms  {flops/dt/1e12:.1f} TFLOP/s  -> {B/dt/1000:.1f} seq/s at T=1000", flush=True)
