import torch, time
dev = "cuda:0"
def bench(m, n, k, reps=10):
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
    for _ in range(3): torch.matmul(a, b.t())
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); torch.matmul(a, b.t()); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in evs)[len(evs) // 2] * 1e-3
    print(f"gemm M{m} N{n} K{k}: {t*1e6:9.1f} us  {2*m*n*k/t/1e12:8.1f} TFLOP/s", flush=True)
for (m, n, k) in [(65536, 512, 4608), (1048576, 512, 4608), (16384, 512, 4608), (8192, 8192, 8192), (65536, 2048, 512), (1048576, 128, 1152), (262144, 256, 2304)]:
    bench(m, n, k)
