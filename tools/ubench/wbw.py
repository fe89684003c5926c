import torch, time
def bw(nbytes, iters=50, op="fill"):
    x = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    for _ in range(3):
        x.fill_(1.0) if op == "fill" else y.copy_(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        x.fill_(1.0) if op == "fill" else y.copy_(x)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    moved = nbytes * (1 if op == "fill" else 2)
    return moved / ms / 1e6
for mb in (32, 64, 128, 192, 512, 2048):
    print(f"{mb:5d} MB  fill {bw(mb << 20):8.0f} GB/s   copy(r+w) {bw(mb << 20, op='copy'):8.0f} GB/s")
