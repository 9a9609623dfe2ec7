import torch, time, torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = 'cuda'
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
# (N, Cin, H, W, Cout) typical ResNet-50 @512^2 input, batch 32
shapes = [(32, 64, 128, 128, 256), (32, 256, 128, 128, 64), (32, 512, 64, 64, 128), (32, 128, 64, 64, 512),
          (32, 1024, 32, 32, 256), (32, 256, 32, 32, 1024), (32, 2048, 32, 32, 512), (32, 512, 32, 32, 2048)]
for (N, Ci, H, W, Co) in shapes:
    x = torch.randn(N, Ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, Ci, 1, 1, device=dev).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * N * H * W * Ci * Co
    tc = t(lambda: F.conv2d(x, w))
    x2 = x.permute(0, 2, 3, 1).reshape(-1, Ci)          # NHWC view, no copy
    w2 = w.view(Co, Ci).t().contiguous()
    tm = t(lambda: torch.matmul(x2, w2))
    print(f"{(N,Ci,H,W,Co)} conv {tc:.3f} ms {fl/tc/1e9:.1f} TF | matmul {tm:.3f} ms {fl/tm/1e9:.1f} TF")
# 3x3 conv for reference
for (N, C, H, W) in [(32, 64, 128, 128), (32, 128, 64, 64), (32, 256, 32, 32), (32, 512, 32, 32)]:
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, C, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * N * H * W * C * C * 9
    tc = t(lambda: F.conv2d(x, w, padding=1))
    print(f"3x3 {(N,C,H,W)} conv {tc:.3f} ms {fl/tc/1e9:.1f} TF")
