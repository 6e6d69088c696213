"""End to end on the GPU through the reference's CLI surface (etdiff_train.py:198-299): BASELINE.json configs[0]
-- train + sample on a synthetic MIMIC-III-shaped tensor (7 continuous + 8 binary channels, 24-step sequences,
batch 256, hidden = 4 x width = 92) -- then the --eval path from the checkpoint it wrote."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cli_train_sample_eval_mimiciii_shape(tmp_path):
    assert torch.cuda.is_available(), "this test needs the MI355X"
    import etdiff_train as cli
    ckpt = str(tmp_path / "models" / "etdiff_mimiciii.pt")
    os.makedirs(os.path.dirname(ckpt))
    common = ["--data_name", "mimiciii", "--cut_length", "24", "--bidirectional", "--batch_size", "256",
              "--timesteps", "30", "--check_point_path", ckpt, "--synthetic", "1024", "--num_samples", "700",
              "--sample_batch_size", "100"]
    cli.main(common + ["--num_steps", "3", "--gradient_accumulate_every", "2", "--record_every", "2"])
    data = torch.load(ckpt, map_location="cpu", weights_only=False)     # our own file
    assert set(data) == {"diff_step", "diff_model", "opt", "ema", "scaler"} and data["diff_step"] == 3
    assert data["diff_model"]["model.rnn.rnn.weight_hh_l0"].shape == (4 * 92, 92)
    cols = cli.COLUMNS_DICT["mimiciii"]
    path = cli.model_path_to_sample_path(ckpt)
    samples = np.load(path)
    assert samples.shape == (700, 15, 24) and samples.dtype == np.float32 and np.isfinite(samples).all()
    assert set(np.unique(samples[:, cols["categorical"]])) <= {0.0, 1.0}
    num = samples[:, cols["numerical"]]
    assert num.min() >= 0.0 and num.max() <= 1.0 and num.std() > 0.01
    # --eval: restore the checkpoint and sample again (etdiff_train.py:288-299)
    os.remove(path)
    cli.main(common + ["--eval", "--eval_path", ckpt])
    again = np.load(path)
    assert again.shape == (700, 15, 24) and np.isfinite(again).all()
