"""Data-parallel training on the GPU: two ranks (two processes on this MI355X, gloo moving the CUDA gradient slices).
Collected last (file name): it is the only GPU test that starts other processes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _dp_worker(rank, world, port, overlap, out_path):
    """One data-parallel rank (both ranks share GPU 0; gloo moves the CUDA gradient slices)."""
    import faulthandler
    import os
    import sys
    faulthandler.dump_traceback_later(90, exit=True)   # a stuck rank reports where and ends itself
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), TIMEDIFF_DIST_BACKEND="gloo", TIMEDIFF_DP_OVERLAP=str(overlap),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from models.ETDiff.blocks import RNN
    from models.ETDiff.et_diff import ETDiff
    from models.ETDiff.mixed_diffusion import MixedDiffusion
    from timediff_amd.data import TensorSeriesDataset
    torch.manual_seed(0)          # same initial weights on every rank
    rnn = RNN(input_channels=7, hidden_channels=32, output_channels=7, layers=3, model="lstm", bidirectional=True)
    diff = MixedDiffusion(rnn, seq_length=12, channels=7, numerical_features_indices=[0, 2, 3],
                          categorical_features_indices=[1, 4], categorical_num_classes=[2, 2], timesteps=20, loss_lambda=0.8)
    data = torch.rand(64, 5, 12)
    data[:, [1, 4]] = (data[:, [1, 4]] > 0.5).float()
    et = ETDiff(diff, TensorSeriesDataset(data, [1, 4]), train_batch_size=16, gradient_accumulate_every=2, diff_lr=1e-3,
                diff_num_steps=2, check_point_path=out_path + ".ckpt")
    torch.manual_seed(100 + rank)  # different noise / time steps per rank
    init = et.flat_param.clone()
    losses = [et.train_step().item() for _ in range(2)]
    torch.cuda.synchronize()
    mine = et.flat_param.detach().cpu()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    assert torch.equal(both[0], both[1]), "ranks diverged: the gradient exchange missed a slice"
    assert (mine != init.cpu()).float().mean() > 0.99 and all(l == l for l in losses)
    if rank == 0:
        torch.save({"param": mine, "losses": losses}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def _run_dp(tmp_path, overlap, port):
    import torch.multiprocessing as mp
    out = str(tmp_path / f"dp_overlap{overlap}.pt")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, overlap, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(110)
    codes = [p.exitcode for p in procs]
    for p in procs:
        if p.exitcode is None:      # never leave a rank behind on the GPU
            p.kill()
            p.join(10)
    assert codes == [0, 0], f"data-parallel ranks exited with {codes} (None = still running after 110 s, killed)"
    return torch.load(out, weights_only=True)


def test_data_parallel_two_ranks_overlap_equals_plain(dev, tmp_path):
    """World size 2 (two processes on this GPU, gloo): after two optimizer steps both ranks hold identical
    weights, and the per-layer all-reduces started from inside the backward pass (td_rnn_backward_cb) give the
    same weights as one all-reduce of the whole flat buffer after it (same seeds; float atomics in the
    weight-gradient kernels make the last bits run-dependent, hence a tolerance)."""
    import os
    port = 29600 + os.getpid() % 1000
    a = _run_dp(tmp_path, 1, port)
    b = _run_dp(tmp_path, 0, port + 1)
    assert torch.allclose(a["param"], b["param"], rtol=0, atol=2e-5), float((a["param"] - b["param"]).abs().max())
    np.testing.assert_allclose(a["losses"], b["losses"], rtol=1e-4)
