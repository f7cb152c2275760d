import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, kornia_amd as K
for shape, dt in (((256,3,224,224), torch.bfloat16), ((256,3,224,224), torch.float32), ((128,3,256,256), torch.float32), ((64,3,512,512), torch.float32), ((256,3,512,512), torch.float32)):
    x = torch.rand(*shape, device="cuda").to(dt)
    for rows in ("32", "8", ""):
        if rows: os.environ["KM_BLUR_ROWS"] = rows
        else: os.environ.pop("KM_BLUR_ROWS", None)
        with torch.no_grad():
            ms = bench.event_time_ms(lambda: K.gaussian_blur2d(x, (5,5), (1.5,1.5)), 30)
        print(shape, dt, "rows", rows or "auto", round(ms*1000,1), "us", flush=True)
