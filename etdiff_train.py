#!/usr/bin/env python3
"""Train ETDiff and write 20 000 synthetic patients -- the reference's CLI (etdiff_train.py) on the MI355X-native
kernels.  Flags, defaults, dataset column maps, checkpoint layout and the `.npy` output location are the
reference's; launch with `python -m torch.distributed.run --nproc-per-node N etdiff_train.py ...` for data-parallel
training and patient-sharded sampling (one process per GPU).

Extra flags (not in the reference): --num_samples, --sample_batch_size, --synthetic N (train on a synthetic
tensor of the dataset's shape when the credentialed data is not available).
"""
import argparse
import os

import numpy as np
import torch

from helpers.utils import (create_id, is_file_on_disk, is_positive_float, is_positive_integer, reverse_normalize,
                           seed_everything)
from models.ETDiff.blocks import RNN, EncoderDecoderRNN, NeuralCDE  # noqa: F401  (names the reference imports)
from models.ETDiff.et_diff import ETDiff
from models.ETDiff.gaussian_diffusion import GaussianDiffusion
from models.ETDiff.mixed_diffusion import MixedDiffusion
from models.ETDiff.utils import TimeSeriesDataset
from timediff_amd.data import TensorSeriesDataset
from timediff_amd.parallel import patient_shard

# numerical / missingness-indicator channel maps per dataset (reference etdiff_train.py:24-29)
COLUMNS_DICT = {
    "mimiciv": {"numerical": [0, 2, 4, 6, 8], "categorical": [1, 3, 5, 7, 9, 10], "categorical_num_classes": [2] * 6},
    "mimiciii": {"numerical": [0, 2, 4, 6, 8, 10, 12], "categorical": [1, 3, 5, 7, 9, 11, 13, 14],
                 "categorical_num_classes": [2] * 8},
    "eicu": {"numerical": [0, 2, 4, 6], "categorical": [1, 3, 5, 7, 8], "categorical_num_classes": [2] * 5},
    "hirid": {"numerical": [0, 1, 2, 3, 4, 5, 6], "categorical": [7], "categorical_num_classes": [2]},
}
WIDE_HIDDEN = ["stock", "energy", "mimiciv", "mimiciii", "hirid"]      # hidden = 4 * channels (etdiff_train.py:239)


def parse_arguments(argv=None):
    prs = argparse.ArgumentParser(prog="etdiff_train.py", description="Train our model",
                                  formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    prs.add_argument("--load_path", type=str, default="data/eicu-extract/TRAIN_irregular_all_patients_0_1440_5.pt")
    prs.add_argument("--data_name", type=str, default="eicu")
    prs.add_argument("--check_point_path", required=True)
    prs.add_argument("--cut_length", type=int, default=272)
    prs.add_argument("--hidden", type=int, default=256)
    prs.add_argument("--num_layers", type=int, default=3)
    prs.add_argument("--model", type=str, default="lstm", choices=["lstm", "gru"])
    prs.add_argument("--bidirectional", action="store_true")
    prs.add_argument("--batch_size", type=is_positive_integer, default=32)
    prs.add_argument("--learning_rate", type=is_positive_float, default=8e-5)
    prs.add_argument("--embed_dim", type=int, default=64)
    prs.add_argument("--time_dim", type=int, default=256)
    prs.add_argument("--dropout", type=float, default=0)
    prs.add_argument("--num_steps", type=is_positive_integer, default=700000)
    prs.add_argument("--gradient_accumulate_every", type=is_positive_integer, default=2)
    prs.add_argument("--ema_decay", type=is_positive_float, default=0.995)
    prs.add_argument("--timesteps", type=is_positive_integer, default=1000)
    prs.add_argument("--auto_normalize", type=bool, default=True)
    prs.add_argument("--record_every", type=is_positive_integer, default=10)
    prs.add_argument("--wandb", action="store_true")
    prs.add_argument("--project_name", type=str, default="Tony-results")
    prs.add_argument("--entity", type=str, default="gen-ehr")
    prs.add_argument("--eval", action="store_true")
    prs.add_argument("--eval_path", type=is_file_on_disk, required=False)
    prs.add_argument("--seed", type=int, default=2023)
    prs.add_argument("--positive_only", action="store_true", required=False)
    prs.add_argument("--loss_lambda", type=float, default=0.8)
    prs.add_argument("--diff_type", type=str, default="mixed", choices=["mixed", "gaussian"])
    prs.add_argument("--dim", type=int, default=64)
    # not in the reference
    prs.add_argument("--num_samples", type=is_positive_integer, default=20000)
    prs.add_argument("--sample_batch_size", type=is_positive_integer, default=100)
    prs.add_argument("--synthetic", type=int, default=0, help="train on N synthetic sequences of the dataset's shape")
    return prs.parse_args(argv)


@torch.no_grad()
def collect_samples(model, num_samples, batch_size=20, min=None, max=None, positive_only: bool = False):
    """reference etdiff_train.py:150-188: `num_samples // batch_size` chains, concatenated, de-normalised."""
    model.eval()
    samples_list = []
    if not positive_only:
        # The reference runs num_samples // batch_size chains of batch_size patients one after the other (200 x 100 by
        # default).  Chains never interact (mixed_diffusion.py:437-446 has no cross-batch operation), so the same
        # patients are drawn in as few chains as fit the GPU: at the reference's batch of 100 a chain keeps 4 of the
        # MI355X's 256 CUs busy (11 sequences/s), at 16 384 all of them (480 sequences/s).
        total = (num_samples // batch_size) * batch_size
        chunk = int(os.environ.get("TIMEDIFF_SAMPLE_CHUNK", "16384"))
        done = 0
        while done < total:
            n = builtins_min(chunk, total - done)
            samples_list.append(model.sample(batch_size=n))
            done += n
    else:
        print("*** COLLECT POSITIVE SAMPLES ***")
        collected = 0
        while collected < num_samples:
            samples = model.sample(batch_size=batch_size)
            mask = (torch.round(samples[:, -1, :]) == 1).any(dim=1)
            if mask.any():
                samples_list.append(samples[mask])
                collected += int(mask.sum())
    samples = torch.cat(samples_list, dim=0).squeeze().cpu().numpy()
    if min is not None and max is not None:
        samples = reverse_normalize(samples, min, max)
    return samples


def _sample_dims(model):
    """(channels, seq_len) of what model.sample returns: create_samples (mixed_diffusion.py:173-179) sizes the
    result `channels - n_cat`; the Gaussian process returns all its channels."""
    n_cat = len(getattr(model, "categorical_features_indices", []))
    return model.channels - n_cat, model.seq_length


@torch.no_grad()
def collect_samples_to_file(model, num_samples, batch_size, min, max, path, rank=0, world=1, seed=None):
    """collect_samples + reverse_normalize + np.save (reference etdiff_train.py:150-166, :190-193, :286) as one
    streaming, patient-sharded pass: rank r draws the patients [start, stop) of the global range in GPU-filling
    chunks (Philox streams keyed by the global patient id, so the file does not depend on the number of ranks or
    on the chunking) and writes its rows of the `.npy` itself (timediff_amd.parallel.SampleWriter): no gather, no
    host concatenation.  The D2H copy and the write of chunk i run while chunk i + 1's chain is on the GPU.
    `seed`: the Philox seed every rank must share (drawn from torch's generator when None: seed_everything makes
    that equal on all ranks)."""
    from timediff_amd.parallel import SampleWriter
    model.eval()
    total = (num_samples // batch_size) * batch_size
    channels, seq_len = _sample_dims(model)
    writer = SampleWriter(path, total, channels, seq_len, min, max, rank, world)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        if world > 1 and torch.distributed.is_initialized():     # rank 0's draw, whatever the others consumed
            dev = "cuda" if torch.distributed.get_backend() == "nccl" else "cpu"
            box = torch.tensor([seed], dtype=torch.int64, device=dev)
            torch.distributed.broadcast(box, src=0)
            seed = int(box.item())
    import inspect
    keep_on_device = "to_cpu" in inspect.signature(model.sample).parameters   # MixedDiffusion.sample moves to the CPU
    chunk = int(os.environ.get("TIMEDIFF_SAMPLE_CHUNK", "16384"))
    start, stop = patient_shard(total, rank, world)
    pending = None            # (first row, pinned host tensor, copy-done event) of the previous chunk
    pos = start
    while pos < stop or pending is not None:
        nxt = None
        if pos < stop:
            n = builtins_min(chunk, stop - pos)
            kw = dict(batch_size=n, seed=seed, patient0=pos)
            if keep_on_device:
                kw["to_cpu"] = False
            dev_out = model.sample(**kw)                    # enqueued; the GPU is busy for the whole chain
            if dev_out.is_cuda:
                host = torch.empty(dev_out.shape, dtype=dev_out.dtype, pin_memory=True)
                host.copy_(dev_out, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            else:
                host, ev = dev_out, None
            nxt = (pos, host, ev)
            pos += n
        if pending is not None:                             # write the previous chunk under this chunk's chain
            p0, host, ev = pending
            if ev is not None:
                ev.synchronize()
            writer.put(p0, host)
        pending = nxt
    return writer.close()


builtins_min = min


def model_path_to_sample_path(model_path) -> str:
    return model_path.replace(".pt", "_samples.npy").replace("models", "samples")


def count_params(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def synthetic_dataset(data_name, n, cut_length, seed):
    cols = COLUMNS_DICT[data_name]
    C = len(cols["numerical"]) + len(cols["categorical"])
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, C, cut_length, generator=g)
    for c, k in zip(cols["categorical"], cols["categorical_num_classes"]):
        x[:, c, :] = torch.randint(0, k, (n, cut_length), generator=g).float()
    return TensorSeriesDataset(x, cols["categorical"])


def main(argv=None):
    args = parse_arguments(argv)
    seed_everything(args.seed)
    run_id = create_id()
    wandb = None
    if args.wandb and int(os.environ.get("RANK", "0")) == 0:
        import wandb as _wandb
        wandb = _wandb
        wandb.init(project=args.project_name, entity=args.entity, name=f"ETDiff_{run_id}", config=args)

    known = args.data_name in COLUMNS_DICT
    if args.synthetic:
        dataset = synthetic_dataset(args.data_name, args.synthetic, args.cut_length, args.seed)
    else:
        dataset = TimeSeriesDataset(categorical_cols=COLUMNS_DICT[args.data_name]["categorical"] if known else None,
                                    data_name=args.data_name, load_path=args.load_path, cut_length=args.cut_length)
    if args.diff_type == "gaussian":
        print("=== Gaussian Diffusion ===")
        model = RNN(input_channels=dataset.channels, hidden_channels=args.hidden, output_channels=dataset.channels,
                    layers=args.num_layers, model=args.model, dropout=args.dropout, bidirectional=args.bidirectional,
                    self_condition=False, embed_dim=args.embed_dim, time_dim=args.time_dim)
        diffusion = GaussianDiffusion(model=model, channels=dataset.channels, seq_length=dataset.seq_length,
                                      timesteps=args.timesteps, auto_normalize=args.auto_normalize)
    else:
        print("=== Mixed Diffusion ===")
        width = dataset.channels + len(dataset.categorical_cols)
        model = RNN(input_channels=width, hidden_channels=width * 4 if args.data_name in WIDE_HIDDEN else args.hidden,
                    output_channels=width, layers=args.num_layers, model=args.model, dropout=args.dropout,
                    bidirectional=args.bidirectional, self_condition=False, embed_dim=args.embed_dim, time_dim=args.time_dim)
        cols = COLUMNS_DICT[args.data_name]
        diffusion = MixedDiffusion(model=model, channels=width, seq_length=dataset.seq_length, timesteps=args.timesteps,
                                   auto_normalize=args.auto_normalize, numerical_features_indices=cols["numerical"],
                                   categorical_features_indices=cols["categorical"],
                                   categorical_num_classes=cols["categorical_num_classes"], loss_lambda=args.loss_lambda)

    etdiff = ETDiff(diffusion_model=diffusion, dataset=dataset, sample_every=args.record_every,
                    train_batch_size=args.batch_size, diff_lr=args.learning_rate, diff_num_steps=args.num_steps,
                    gradient_accumulate_every=args.gradient_accumulate_every, ema_decay=args.ema_decay, amp=False,
                    wandb=wandb, check_point_path=args.check_point_path, run_id=run_id)
    params = count_params(model)
    print(f"Number of parameters: {params}")
    if wandb is not None:
        wandb.log({"total_params": params})

    if not args.eval:
        etdiff.train()
        sample_path = model_path_to_sample_path(args.check_point_path)
        print("Creating synthetic samples...")
    else:
        print("Collecting...")
        etdiff.load(args.eval_path)
        sample_path = model_path_to_sample_path(args.eval_path)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ema_model = etdiff.ema.ema_model
    # every rank holds the same EMA weights after training (rank 0's after --eval loads on every rank) and the same
    # torch seed, so the shared Philox seed needs no exchange
    collect_samples_to_file(ema_model, args.num_samples, args.sample_batch_size, etdiff.sample_min, etdiff.sample_max,
                            sample_path, etdiff.rank, world)
    if etdiff.rank == 0:
        print(f"Saved synthetic samples to {sample_path}")

if __name__ == "__main__":
    main()
