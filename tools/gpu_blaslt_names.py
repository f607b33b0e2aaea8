import torch
for (M, N, K) in [(2528, 28672, 4096), (2528, 4096, 28672), (2528, 4096, 4096), (12000, 4096, 1024), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    for _ in range(3): torch.matmul(a, b.t())
torch.cuda.synchronize()
