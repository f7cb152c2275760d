"""CPU: the GPU parity tests, unchanged, against a host build of the shipped kernel sources.

tests/emu/ compiles kornia_amd/csrc/*.hip for x86 (same -ffp-contract=off; work-items are fibers, so LDS, barriers, wave
shuffles and atomics are modelled) into a library with the same C ABI; inside `emulated_device()` the unchanged Python
host layer calls it and "cuda" means host memory.  What this tier proves without a GPU: the arithmetic of the shipped
kernels - fp32 operation order, border / padding index maps, tile ownership, fixed-point accumulation, reductions -
reproduces the reference's fixtures and the oracle bit for bit (or within the same tolerances as on the device).
What it cannot prove: anything about the gfx950 code generator, memory model or speed - the `-m gpu` run does that.
The package itself never uses this path (tests/test_abi_and_host.py::test_no_cpu_fallback)."""
import importlib
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))

# GPU test modules whose cases are small enough for the emulator (full-size configs and HIP-graph capture stay
# device-only)
MODULES = os.environ.get("KM_EMU_MODULES", "test_gpu_golden test_gpu_aug_modules test_gpu_config_parity test_gpu_warp test_gpu_warp_fused test_gpu_warp_gm_box test_gpu_warp_blur test_gpu_augmentation test_gpu_filters test_gpu_edge_cases test_gpu_grid_sample test_gpu_color test_gpu_fuzz test_gpu_pyramid test_zz_gpu_registration test_zz_gpu_canny test_zz_gpu_fuzz_pyramid test_zz_gpu_nonfinite_paths").split()
# cases that are about the device itself, not about kernel arithmetic
SKIP = {
    ("test_gpu_warp", "test_identity_is_exact_and_errors"),      # asserts that host tensors are refused
    ("test_gpu_color", "test_half_precision_and_errors"),        # same
    ("test_gpu_edge_cases", "test_mixed_dtypes_and_streams"),    # HIP streams
    ("test_gpu_edge_cases", "test_two_threads_two_streams_share_no_launch_state"),  # HIP streams, host threads
}

for _m in MODULES:
    _mod = importlib.import_module(_m)
    for _name in dir(_mod):
        if _name.endswith("_at_full_size"):  # BASELINE-size property tests: device RNG, minutes of emulation
            continue
        if _name.startswith("test_") and callable(getattr(_mod, _name)) and (_m, _name) not in SKIP:
            globals()[f"{_name}__{_m.replace('test_zz_gpu_', 'zz_').replace('test_gpu_', '')}"] = getattr(_mod, _name)


if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):  # pragma: no cover
    pytest.skip("ROCm clang++ (host compiler of the emulated build) not found", allow_module_level=True)


def test_emulated_library_exports_the_c_abi():
    import emu_lib

    from kornia_amd import _native

    h = emu_lib.lib()
    for name in _native.exported_symbols():
        assert hasattr(h, name), name
    assert h.km_abi_version() == _native.ABI_VERSION


def test_emulator_schedules_expose_a_missing_barrier():
    """The emulator's own teeth: a hand-off through LDS without a barrier is right by luck in launch order and wrong as soon
    as the waves resume in another order; with the barrier every order gives the same answer; a wave operation that only part
    of a wave reaches is reported as a launch failure instead of hanging."""
    import ctypes
    import subprocess

    import build_emu

    if build_emu.UBSAN:
        pytest.skip("the toy kernels are not linked against the sanitizer runtime")
    build_emu.build()
    so = os.path.join(build_emu.OUT, f"libselftest{os.environ.get('PYTEST_XDIST_WORKER', '')}.so")  # (one per worker process of a parallel run)
    subprocess.run([build_emu.CXX, *build_emu._flags(), "-shared", "-Wl,-Bsymbolic", os.path.join(build_emu.EMU, "selftest_kernels.hip"),
                    os.path.join(build_emu.EMU, "emu_runtime.cpp"), "-o", so], check=True)
    h = ctypes.CDLL(so)
    h.emu_set_schedule.argtypes = [ctypes.c_int, ctypes.c_ulonglong]
    out = (ctypes.c_int * 512)()
    expect = [((t + 192) % 256) + 1000 * b for b in range(2) for t in range(256)]
    for mode in (0, 1, 2, 3):
        h.emu_set_schedule(mode, 11)
        assert h.selftest_handoff(out, 2, 1) == 0 and list(out) == expect
    h.emu_set_schedule(0, 1)
    assert h.selftest_handoff(out, 2, 0) == 0
    in_launch_order = list(out)
    h.emu_set_schedule(1, 1)
    assert h.selftest_handoff(out, 2, 0) == 0
    assert list(out) != in_launch_order, "reversing the wave order must change the result of the racy kernel"
    assert h.selftest_divergent_wave_op(out) != 0
    # LDS-DMA models: a request lands at once (immediate) or at the requesting lane's KM_VMCNT0 (deferred) - one that is never waited for never lands
    fo = (ctypes.c_float * 2)()
    h.emu_set_schedule(0, 1)
    h.emu_set_glds(0)
    assert h.selftest_glds(fo) == 0 and list(fo) == [1.0, 1.0]
    h.emu_set_glds(1)
    assert h.selftest_glds(fo) == 0 and list(fo) == [1.0, 0.0]
    h.emu_set_glds(0)


@pytest.mark.parametrize("schedule", ["reverse", "random", "lanes"])
def test_lds_kernels_do_not_depend_on_wave_order(oracle, schedule):
    """The kernels that communicate through LDS (tile-owner scatter, matrix gradient, fused loss, colour statistics, large
    separable filter, box forward) give the same results when their waves / work-items resume in another order."""
    import emu_lib
    import test_gpu_color
    import test_gpu_filters
    import test_gpu_warp
    import test_zz_gpu_registration as reg

    emu_lib.set_schedule(schedule, 5)  # (the autouse fixture below has already entered the emulated device)
    try:
        test_gpu_warp.test_tiled_backward_affine_and_homography(oracle, False)
        test_gpu_warp.test_tiled_backward_vanishing_line_inside_image(oracle)
        reg.test_single_level_loss_vs_reference("l1", torch.nn.functional.l1_loss)
        test_gpu_color.test_all_orders_vs_restatement((3, 3, 37, 53))
        test_gpu_filters.test_large_separable_kernels(oracle, "reflect", (23, 23), (2, 3, 70, 90))
        if schedule != "lanes":  # (its gather rows exchange registers between lanes - wave_shl / wave_shr - which only means something in wave order)
            test_gpu_warp.test_box_forward_is_bit_identical(oracle, 1, torch.float32)  # fill -> barrier -> sample through LDS, three tile attempts per block
    finally:
        emu_lib.set_schedule("forward")


@pytest.mark.parametrize("schedule", ["forward", "reverse"])
def test_lds_dma_kernels_under_the_latest_legal_landing(oracle, schedule):
    """The three kernels that fill LDS by LDS-DMA (global_load_lds: one-read backward - the NEXT tile's source tile into the second buffer while the
    current one is read -, box forward, bicubic forward) with every request landing as LATE as the hardware may land it: at the requesting lane's
    own `s_waitcnt vmcnt(0)`, in shuffled order (the default model lands it at the request: the earliest).  A missing wait, or a missing barrier
    between the wait and another wave's read, is stale LDS - a wrong result - here (test_emulator_schedules_expose_a_missing_barrier shows that the model bites)."""
    import emu_lib
    import test_gpu_warp
    import test_gpu_warp_fused as wf

    emu_lib.set_glds(True)
    emu_lib.set_schedule(schedule, 3)
    try:
        wf.test_source_tile_staging_does_not_depend_on_the_address_or_the_tile_shape(oracle, "zeros")
        wf.test_source_tile_staging_does_not_depend_on_the_address_or_the_tile_shape(oracle, "fill")
        wf.test_both_gradients_from_one_read_match_the_oracle(oracle, (3, 200, 130, 150, 170), "zeros", 3)
        wf.test_both_gradients_from_one_read_match_the_oracle(oracle, (2, 128, 192, 128, 192), "fill", 3)
        test_gpu_warp.test_box_forward_is_bit_identical(oracle, 3, torch.float32)
        test_gpu_warp.test_bicubic_lds_staged_kernel_is_bit_identical(oracle, 1, torch.float32)
    finally:
        emu_lib.set_glds(False)
        emu_lib.set_schedule("forward")


def test_smoke_entry_point_on_the_host_build(monkeypatch):
    """__graft_entry__.smoke() (the driver's device check) end to end against the host build of the kernels."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as entry

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    entry.smoke()


def test_odd_storage_offsets_never_reach_the_vector_paths(oracle):
    """Contiguous tensors whose storage starts 4 (fp32) / 2 (bf16) bytes into an allocation: the launchers must route them to
    kernels without 8 / 16-byte accesses.  The host build checks the alignment of every vector access (KM_CHECK_ALIGNED) and
    fails the launch otherwise; results stay bit-identical to the oracle."""
    import emu_lib

    import kornia_amd as K

    def shifted(shape, dtype=torch.float32, seed=0):
        n = 1
        for d in shape:
            n *= d
        base = torch.rand(n + 1, generator=torch.Generator().manual_seed(seed)).to(dtype)
        t = base[1:].view(*shape)
        assert t.is_contiguous() and t.data_ptr() % 8 != 0
        return t

    before = emu_lib.stats()["misaligned_vector_accesses"]
    x = shifted((2, 3, 32, 48))
    T = K.geometry.transform
    assert torch.equal(K.gaussian_blur2d(x.cuda(), (5, 5), (1.5, 1.5)), oracle.gaussian_blur2d(x, (5, 5), (1.5, 1.5)))
    k = torch.rand(1, 5, 5, generator=torch.Generator().manual_seed(1))
    assert torch.equal(K.filter2d(x.cuda(), k), oracle.filter2d(x, k))
    assert torch.equal(K.spatial_gradient(x.cuda()), oracle.spatial_gradient(x))
    assert torch.equal(T.pyrdown(x.cuda()), oracle.pyrdown(x))
    assert torch.equal(T.pyrup(x.cuda()), oracle.pyrup(x))
    xg = shifted((2, 3, 32, 48), seed=2).requires_grad_(True)
    K.gaussian_blur2d(xg.cuda(), (5, 5), (1.5, 1.5)).sum().backward()
    assert torch.isfinite(xg.grad).all()
    xb = shifted((2, 3, 32, 48), torch.bfloat16)
    assert (K.gaussian_blur2d(xb.cuda(), (5, 5), (1.5, 1.5)).float() - oracle.gaussian_blur2d(xb.float(), (5, 5), (1.5, 1.5))).abs().max() <= 1e-2
    assert (T.pyrdown(xb.cuda()).float() - oracle.pyrdown(xb.float())).abs().max() <= 1e-2
    big = shifted((1, 2, 40, 64), seed=4)
    assert torch.equal(K.gaussian_blur2d(big.cuda(), (23, 23), (4.0, 4.0)), oracle.gaussian_blur2d(big, (23, 23), (4.0, 4.0)))  # LDS sliding-window kernel
    M = torch.tensor([[[1.02, 0.03, 1.5], [-0.02, 0.98, -2.0], [1e-4, 0.0, 1.0]]]).repeat(2, 1, 1)
    xw = shifted((2, 3, 64, 64), seed=5).requires_grad_(True)
    go = shifted((2, 3, 64, 64), seed=6)
    K.warp_perspective(xw.cuda(), M.cuda(), (64, 64)).backward(go.cuda())  # tile-owner scatter reading an odd-offset grad_out
    gref, _ = oracle.warp_perspective_backward(go, xw.detach(), M, (64, 64))
    assert torch.allclose(xw.grad, gref, atol=1e-5, rtol=1e-5)
    f = [torch.full((2,), v) for v in (1.1, 0.9, 1.2, 0.05)]
    assert torch.isfinite(K.enhance.color_jitter(x.cuda(), *f)).all()
    assert emu_lib.stats()["misaligned_vector_accesses"] == before


@pytest.fixture(autouse=True)
def _emulated():
    from mode import emulated_device

    with emulated_device():
        yield
