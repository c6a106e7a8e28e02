"""What the memory system gives a 1:1 read/write mix at the size of config 2's [N_seg][L] array (k_post_z reads and rewrites it in place): a device copy
and an in-place scale through torch's elementwise kernels.  python tools/hbm_mix_bw.py"""
import torch, time
n = 11_300_000_000 // 8
a = torch.empty(n, dtype=torch.float64, device="cuda"); b = torch.ones(n, dtype=torch.float64, device="cuda")
for _ in range(2): a.copy_(b)
torch.cuda.synchronize(); t=time.time()
for _ in range(5): a.copy_(b)
torch.cuda.synchronize(); dt=(time.time()-t)/5
print("copy 11.3 GB: %.2f ms, %.2f TB/s (read+write)" % (dt*1e3, 2*n*8/dt/1e12))
for _ in range(2): a.mul_(1.0000001)
torch.cuda.synchronize(); t=time.time()
for _ in range(5): a.mul_(1.0000001)
torch.cuda.synchronize(); dt=(time.time()-t)/5
print("in-place scale 11.3 GB: %.2f ms, %.2f TB/s (read+write)" % (dt*1e3, 2*n*8/dt/1e12))
