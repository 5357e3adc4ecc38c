cd $GRAFT_REPO_ROOT
for t in "" 4 16; do
  echo "== OMP_NUM_THREADS=$t"; OMP_NUM_THREADS=$t N_FRAMES=24 timeout 600 python tools/api_pcie_bench.py 2>&1 | grep "batch rep"
done
python - <<'PY'
import torch, time, numpy as np
x = torch.randn(268, 721, 1440)
pin = torch.empty_like(x).pin_memory()
for nt in (1, 4, 16, 64, 128):
    torch.set_num_threads(nt)
    pin.copy_(x); t0 = time.perf_counter()
    for _ in range(3): pin.copy_(x)
    dt = (time.perf_counter() - t0) / 3
    print(f"pageable -> pinned copy, {nt} threads: {dt*1e3:.1f} ms = {x.numel()*4/dt/1e9:.1f} GB/s")
d = torch.empty_like(x, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): d.copy_(pin, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"pinned -> device: {dt*1e3:.1f} ms = {x.numel()*4/dt/1e9:.1f} GB/s")
t0 = time.perf_counter()
for _ in range(2): d.copy_(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
print(f"pageable -> device directly: {dt*1e3:.1f} ms = {x.numel()*4/dt/1e9:.1f} GB/s")
PY
