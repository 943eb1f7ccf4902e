import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_e2e as t
ref, yard, gpu, _, (lr, ly, lg) = t._three_way(0.1)
names = [n for n, _ in ref.named_parameters()]
print("logits rel", t._rel(lg.float(), lr), t._rel(ly.float(), lr))
for n, pg, py, pr in zip(names, gpu.parameters(), yard.parameters(), ref.parameters()):
    c, cy = t._cos(pg.grad, pr.grad), t._cos(py.grad, pr.grad)
    if n.startswith("layer1") or n.startswith("conv1") or n.startswith("bn1") or "layer2.0" in n:
        print(f"{n:40s} gpu {c:.4f} stock-bf16 {cy:.4f} {'<<<' if c < cy - 0.05 else ''}")
