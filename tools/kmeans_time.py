import os, sys, time, random, torch
sys.path.insert(0, os.getcwd())
from cogstream_amd import kmeans as km
dev = torch.device("cuda:0")
torch.manual_seed(1)
T, P, D, K = 256, 50, 3584, 18
base = torch.randn(1, P * D)
cent = base + 0.5 * torch.randn(K, P * D)
feats = (cent[torch.arange(T) // 15 % K] + 0.2 * torch.randn(T, P * D)).bfloat16().to(dev).view(T, P, D)
ts = torch.arange(T, dtype=torch.float32)
def run():
    random.seed(0); torch.manual_seed(0)
    return km.kmeans_with_time_min_max(feats, ts, K)
run(); torch.cuda.synchronize()
tt = []
for _ in range(7):
    t0 = time.perf_counter(); r = run(); torch.cuda.synchronize(); tt.append(time.perf_counter() - t0)
print("kmeans ms", [round(x * 1e3, 3) for x in sorted(tt)], km.last_stats)
