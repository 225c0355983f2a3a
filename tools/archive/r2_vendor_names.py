import torch
d = torch.device("cuda")
R, D, F = 100864, 768, 3072
X, H = torch.randn(R, D, device=d).half(), torch.randn(R, F, device=d).half()
Wq, Wo, W1, W2 = (torch.randn(n, k, device=d).half() for n, k in ((3 * D, D), (D, D), (F, D), (D, F)))
for _ in range(5):
    torch.matmul(X, Wq.t()); torch.matmul(X, Wo.t()); torch.matmul(X, W1.t()); torch.matmul(H, W2.t())
torch.cuda.synchronize()
