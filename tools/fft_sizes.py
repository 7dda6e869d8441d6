import torch, time
dev = "cuda"
for n in (160, 192, 200, 216, 224, 240, 250, 256, 288, 320):
    x = torch.randn(n, n, n, device=dev, dtype=torch.float32)
    for _ in range(3):
        y = torch.fft.rfftn(x); z = torch.fft.irfftn(y, s=(n, n, n))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
    tf = tb = 0.0
    for _ in range(10):
        e0.record(); y = torch.fft.rfftn(x); e1.record(); z = torch.fft.irfftn(y, s=(n, n, n)); e2.record()
        torch.cuda.synchronize(); tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
    print(f"n={n}: rfftn {100*tf:.0f} us  irfftn {100*tb:.0f} us  total {100*(tf+tb):.0f} us  ({n**3/1e6:.1f} M points)")
