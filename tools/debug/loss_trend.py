"""Sanity: the loss of the ETDiff trainer over a few hundred optimizer steps on synthetic eICU-shaped data (graph-replayed steps)."""
import sys
import torch
sys.path.insert(0, ".")
from bench import EICU
from models.ETDiff.blocks import RNN
from models.ETDiff.et_diff import ETDiff
from models.ETDiff.mixed_diffusion import MixedDiffusion
from timediff_amd.data import TensorSeriesDataset
torch.manual_seed(1)
cols, L, H = EICU, 272, 256
width = len(cols["num"]) + sum(cols["K"])
n_ch = len(cols["num"]) + len(cols["cat"])
rnn = RNN(input_channels=width, hidden_channels=H, output_channels=width, layers=3, model="lstm", bidirectional=True)
diff = MixedDiffusion(rnn, seq_length=L, channels=width, numerical_features_indices=cols["num"], categorical_features_indices=cols["cat"],
                      categorical_num_classes=cols["K"], timesteps=1000, loss_lambda=0.8)
g = torch.Generator().manual_seed(3)
tt = torch.linspace(0, 6.28, L)
data = 0.5 + 0.4 * torch.sin(tt[None, None, :] * (1 + torch.rand(512, n_ch, 1, generator=g)) + 6.28 * torch.rand(512, n_ch, 1, generator=g))
for c in cols["cat"]:
    data[:, c] = (data[:, c] > 0.5).float()
et = ETDiff(diff, TensorSeriesDataset(data, cols["cat"]), train_batch_size=32, gradient_accumulate_every=2, diff_lr=8e-4,
            diff_num_steps=400, ema_update_every=10, check_point_path="/tmp/trend.pt")
losses = []
for i in range(400):
    losses.append(et.train_step())
losses = torch.stack(losses).cpu()
for a in range(0, 400, 50):
    print(f"steps {a:3d}-{a + 49:3d}: mean loss {losses[a:a + 50].mean():.4f}")
print("graphs:", len(et._graphs), "finite:", bool(torch.isfinite(losses).all()))
out = et.ema.ema_model.sample(batch_size=8)
print("sample range", float(out.min()), float(out.max()))
