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


def _dp_worker(rank, world, port, overlap, out_path, backend="gloo", force=0, hidden=32, layers=3):
    """One data-parallel rank (both ranks share GPU 0; gloo moves the CUDA gradient slices).  backend="nccl" with
    world = 1 and force = 1: the production RCCL branch on a single-rank communicator."""
    import faulthandler
    import os
    import sys
    faulthandler.dump_traceback_later(90, exit=True)   # a stuck rank reports where and ends itself
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), TIMEDIFF_DIST_BACKEND=backend, TIMEDIFF_DP_OVERLAP=str(overlap),
                      TIMEDIFF_DP_FORCE=str(force), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from models.ETDiff.blocks import RNN
    from models.ETDiff.et_diff import ETDiff
    from models.ETDiff.mixed_diffusion import MixedDiffusion
    from timediff_amd.data import TensorSeriesDataset
    torch.manual_seed(0)          # same initial weights on every rank
    rnn = RNN(input_channels=7, hidden_channels=hidden, output_channels=7, layers=layers, model="lstm", bidirectional=True)
    diff = MixedDiffusion(rnn, seq_length=12, channels=7, numerical_features_indices=[0, 2, 3],
                          categorical_features_indices=[1, 4], categorical_num_classes=[2, 2], timesteps=20, loss_lambda=0.8)
    data = torch.rand(64, 5, 12)
    data[:, [1, 4]] = (data[:, [1, 4]] > 0.5).float()
    et = ETDiff(diff, TensorSeriesDataset(data, [1, 4]), train_batch_size=16, gradient_accumulate_every=2, diff_lr=1e-3,
                diff_num_steps=2, check_point_path=out_path + ".ckpt")
    torch.manual_seed(100 + rank)  # different noise / time steps per rank
    init = et.flat_param.clone()
    reduced = []
    if force:      # record which slices went through the per-layer path
        plain = et._reduce_slice
        et._reduce_slice = lambda a, b: (reduced.append((a, b)), plain(a, b))[1]
    losses = [et.train_step().item() for _ in range(2)]
    torch.cuda.synchronize()
    mine = et.flat_param.detach().cpu()
    if world > 1:
        both = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        assert torch.equal(both[0], both[1]), "ranks diverged: the gradient exchange missed a slice"
    assert (mine != init.cpu()).float().mean() > 0.99 and all(l == l for l in losses)
    if force:
        assert dist.is_initialized() and dist.get_backend() == backend and et._dp
        want = 2 * (layers + 1 if overlap else 1)      # per optimizer step: a slice per layer + the rest, or one whole-buffer reduce
        assert len(reduced) >= want, (len(reduced), want)
    if rank == 0:
        torch.save({"param": mine, "losses": losses}, out_path)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def _run_dp(tmp_path, overlap, port, world=2, backend="gloo", force=0, hidden=32, layers=3):
    import torch.multiprocessing as mp
    out = str(tmp_path / f"dp_{backend}_w{world}_f{force}_overlap{overlap}_h{hidden}.pt")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, overlap, out, backend, force, hidden, layers)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(110)
    codes = [p.exitcode for p in procs]
    for p in procs:
        if p.exitcode is None:      # never leave a rank behind on the GPU
            p.kill()
            p.join(10)
    assert codes == [0] * world, f"data-parallel ranks exited with {codes} (None = still running after 110 s, killed)"
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


def test_nccl_branch_single_rank_equals_plain(dev, tmp_path):
    """The production branch of the gradient exchange -- backend "nccl" (= RCCL), device-side all-reduce of each
    recurrent layer's slice on the communication stream behind an event recorded from td_rnn_backward_cb, then the
    join -- executed on the hardware at hand: WORLD_SIZE = 1 with the data-parallel path forced on (RCCL accepts a
    single-rank communicator).  The weights after two optimizer steps must equal the world == 1 fast path's (same
    seeds; float atomics in the weight-gradient kernels make the last bits run-dependent, hence the tolerance)."""
    import os
    port = 29800 + os.getpid() % 1000
    forced = _run_dp(tmp_path, 1, port, world=1, backend="nccl", force=1)
    plain = _run_dp(tmp_path, 1, port + 1, world=1, backend="nccl", force=0)
    assert torch.allclose(forced["param"], plain["param"], rtol=0, atol=2e-5), float((forced["param"] - plain["param"]).abs().max())
    np.testing.assert_allclose(forced["losses"], plain["losses"], rtol=1e-4)


def test_data_parallel_two_ranks_wide_hidden(dev, tmp_path):
    """BASELINE.json configs[4] in miniature: data-parallel training of a WIDE-hidden denoiser (H = 512: rnn_wide.hip, the wide
    BPTT sweep, 13 M parameters in the flat gradient buffer), two ranks on this GPU, per-layer all-reduces started from inside
    the backward pass.  The worker asserts that both ranks hold bit-identical weights after two optimizer steps; here the
    overlapped exchange must also give the weights of the single all-reduce after the backward pass."""
    import os
    port = 29300 + os.getpid() % 1000
    a = _run_dp(tmp_path, 1, port, hidden=512, layers=2)
    b = _run_dp(tmp_path, 0, port + 1, hidden=512, layers=2)
    assert torch.isfinite(a["param"]).all()
    assert torch.allclose(a["param"], b["param"], rtol=0, atol=2e-5), float((a["param"] - b["param"]).abs().max())
    np.testing.assert_allclose(a["losses"], b["losses"], rtol=1e-4)
