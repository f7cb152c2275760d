"""CPU, world_size 2, gloo: the batch-shard / all_gather path used for N > 1 GPUs.  The per-rank op is
the CPU oracle (the sharding logic is op-agnostic) and, in the last test, the product's own Python layer running on the host
build of the kernels (tests/emu) with the oracle as the checker."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, batch, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from _util import flagship_homographies

        from kornia_amd.distributed import shard_bounds, sharded_apply

        g = torch.Generator().manual_seed(0)
        x = torch.rand(batch, 3, 32, 40, generator=g)
        M = flagship_homographies(batch, 32, 40, 32, 40, g, jitter=2.0)
        shared_A = torch.tensor([[[1.0, 0.0, 1.5], [0.0, 1.0, -0.5]]])

        def op(xs, Ms):
            return oracle.gaussian_blur2d(oracle.warp_perspective(xs, Ms, (32, 40)), (5, 5), (1.5, 1.5))

        full = op(x, M)
        got = sharded_apply(op, x, M)
        assert got.shape == full.shape and torch.equal(got, full), "gathered result differs from the unsharded one"
        local = sharded_apply(op, x, M, gather=False)
        lo, hi = shard_bounds(batch, world, rank)
        assert torch.equal(local, full[lo:hi])
        # direct peer exchange and the chunked (compute / exchange overlapped) form give the same tensor
        assert torch.equal(sharded_apply(op, x, M, mode="p2p"), full)
        for k in (2, 3):
            assert torch.equal(sharded_apply(op, x, M, chunks=k), full), f"chunks={k}"
        # a (batch,) vector is only sliced when it is declared batched; a (3,) fill value with batch == 3 never is by shape
        scale = torch.arange(1.0, batch + 1.0)
        got_s = sharded_apply(lambda xs, s: xs * s.view(-1, 1, 1, 1), x, scale, batched=(0, 1))
        assert torch.equal(got_s, x * scale.view(-1, 1, 1, 1))
        got_f = sharded_apply(lambda xs, f: xs + f.view(1, -1, 1, 1), x, torch.tensor([1.0, 2.0, 3.0]))
        assert torch.equal(got_f, x + torch.tensor([1.0, 2.0, 3.0]).view(1, -1, 1, 1))
        # a shared (1,2,3) matrix is replicated, not sliced
        got2 = sharded_apply(lambda xs, A: oracle.warp_affine(xs, A, (32, 40)), x, shared_A)
        assert torch.equal(got2, oracle.warp_affine(x, shared_A, (32, 40)))
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [6, 5, 3])  # even and uneven split; 3 = the channel count (a (3,) fill value is not a batch)
def test_sharded_apply_world2_gloo(tmp_path, batch):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, batch, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_shard_bounds():
    from kornia_amd.distributed import shard_batch, shard_bounds

    assert [shard_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [shard_bounds(2, 4, r) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)
    x, m = torch.zeros(8, 3), torch.zeros(1, 3, 3)
    xs, ms, n = shard_batch([x, m, None], 8, 4, 1)
    assert xs.shape[0] == 2 and ms is m and n is None
    v = torch.zeros(8)
    with pytest.raises(ValueError, match="ambiguous"):  # a (B,) vector is neither sliced nor silently replicated by the shape rule ...
        shard_batch([x, v], 8, 4, 1)
    assert shard_batch([x, v], 8, 4, 1, batched=(0, 1))[1].shape[0] == 2  # ... it is sliced by declaration ...
    assert shard_batch([x, v], 8, 4, 1, batched=(0,))[1] is v  # ... or replicated by declaration
    f3 = torch.zeros(3)
    assert shard_batch([torch.zeros(3, 3, 4, 4), f3], 3, 3, 1)[1] is f3  # channel-like sizes (a (3,) fill value with batch 3) pass
    with pytest.raises(ValueError):
        shard_batch([x, m], 8, 4, 1, batched=(0, 1))


def _worker_product(rank, world, port, batch, out_dir):
    """Same sharding, but the per-rank op is the product's own Python layer on the host build of the kernels (tests/emu)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from _util import flagship_homographies
        from mode import emulated_device

        import kornia_amd as K
        from kornia_amd.distributed import shard_bounds, sharded_apply

        g = torch.Generator().manual_seed(0)
        x = torch.rand(batch, 3, 32, 40, generator=g)
        M = flagship_homographies(batch, 32, 40, 32, 40, g, jitter=2.0)
        with emulated_device():
            step = lambda xs, Ms: K.gaussian_blur2d(K.warp_perspective(xs, Ms, (32, 40)), (5, 5), (1.5, 1.5))  # noqa: E731
            got = sharded_apply(step, x, M)
            assert torch.equal(got, oracle.gaussian_blur2d(oracle.warp_perspective(x, M, (32, 40)), (5, 5), (1.5, 1.5)))
            pyr = sharded_apply(lambda xs: K.geometry.transform.pyrdown(xs), x)
            assert torch.equal(pyr, oracle.pyrdown(x))
            # gradients stay with their shard: each rank back-propagates its own slice, nothing is exchanged
            lo, hi = shard_bounds(batch, world, rank)
            xs, Ms = x[lo:hi].clone().requires_grad_(True), M[lo:hi].clone().requires_grad_(True)
            go = torch.rand(batch, 3, 32, 40, generator=g)[lo:hi]
            step(xs, Ms).backward(go)
            gw = oracle.gaussian_blur2d_backward(go, oracle.warp_perspective(x[lo:hi], M[lo:hi], (32, 40)), (5, 5), (1.5, 1.5))
            gx, _ = oracle.warp_perspective_backward(gw, x[lo:hi], M[lo:hi], (32, 40))
            assert torch.allclose(xs.grad, gx, atol=1e-5) and Ms.grad.shape == (hi - lo, 3, 3)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_sharded_product_ops_world2_gloo(tmp_path):
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("host build of the kernels needs ROCm's clang++")
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu

    build_emu.build()  # once, before the ranks start
    port = _free_port()
    mp.spawn(_worker_product, args=(2, port, 5, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
