"""Where the HOST's time goes in config 3 as it is written (kornia_amd.augmentation.AugmentationSequential(RandomAffine, ColorJitter,
RandomGaussianBlur)(x), 256x3x224x224 bf16, sampling inside the call): cProfile over 300 calls on the GPU box, top entries by own time, plus the
wall time per call with and without the profiler.   python profiles/prof_cfg3_host.py"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kornia_amd.augmentation as A
dev = torch.device("cuda")
with torch.no_grad():
    x = torch.rand(256, 3, 224, 224, device=dev).bfloat16()
    aug = A.AugmentationSequential(A.RandomAffine(degrees=15.0, translate=(0.1, 0.1), scale=(0.8, 1.2), shear=5.0, p=1.0),
                                   A.ColorJitter(0.2, 0.2, 0.2, 0.1, p=1.0), A.RandomGaussianBlur((5, 5), (0.1, 2.0), p=1.0))
    for _ in range(50): aug(x)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(300): aug(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"300 calls: host issue {1e3 * (t1 - t0) / 300:.4f} ms per call, with the device drained {1e3 * (t2 - t0) / 300:.4f} ms per call", flush=True)
    t0 = time.perf_counter()
    for _ in range(300): aug.forward_parameters(x.shape)
    print(f"forward_parameters alone: {1e3 * (time.perf_counter() - t0) / 300:.4f} ms per call", flush=True)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300): aug(x)
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(40)
