"""Checksum of the parameters after three deterministic training steps (ResNet-18, ResNet-50 slice): two builds whose
kernels compute the same sums in the same order print the same lines.   python bench/det_checksum.py"""
import hashlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
from pytorch_distributed_training_amd import ops
from pytorch_distributed_training_amd.models import build_model
from pytorch_distributed_training_amd.optim import build_optimizer
dev = torch.device("cuda", 0)
ops.set_deterministic(True)
for name, bs, size in (("resnet18", 32, 64), ("resnet50", 16, 96)):
    torch.manual_seed(0)
    m = build_model(name, 10, size).to(dev).to(memory_format=torch.channels_last)
    x = ops.to_device_layout(torch.randn(bs, 3, size, size, device=dev))
    y = torch.randint(0, 10, (bs,), device=dev)
    raw = ops.prepare_module(m)
    opt = build_optimizer([p for p in raw.parameters() if p.requires_grad], "sgd", 0.05, 1e-4, 0.9)
    opt.zero_grad(set_to_none=False)
    for _ in range(3):
        loss = ops.softmax_cross_entropy(m(x), y)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=False)
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1).float() for p in raw.parameters()]).cpu().numpy().tobytes()
    print(name, hashlib.sha256(flat).hexdigest()[:16], float(loss))
