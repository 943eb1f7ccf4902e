"""Does the fused GPU path learn?  A 10-class task with a visible cue (a bright 16x16 patch whose position encodes the
class, on top of noise), ResNet-50 / ViT-B/16 at reduced resolution, 200 optimisation steps; prints the loss every
20 steps and the accuracy on the training images.  A correct training path drives the loss towards zero.

    python bench/learn_check.py [resnet50|vit_b_16] [image size] [batch]

At 224 x 224 with a batch of 128 the ResNet run goes through the kernels that only large launches select (the
shared-rows 3x3 and stem kernels, the ring GEMMs).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pytorch_distributed_training_amd import ops  # noqa: E402
from pytorch_distributed_training_amd.models import build_model  # noqa: E402
from pytorch_distributed_training_amd.optim import build_optimizer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
size = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if name.startswith("resnet") else 224)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 128
n = 16 * bs
y_all = torch.randint(0, 10, (n,))
x_all = 0.3 * torch.rand(n, 3, size, size)
p = size // 4
for i in range(n):
    c = int(y_all[i])
    r, q = (c // 4) * p, (c % 4) * p
    x_all[i, :, r:r + p, q:q + p] += 0.7
model = build_model(name, 10, size).to(dev).to(memory_format=torch.channels_last)
raw = ops.prepare_module(model)
opt = build_optimizer(list(raw.parameters()), "sgd" if name.startswith("resnet") else "adamw",
                      0.05 if name.startswith("resnet") else 3e-4, 1e-4, 0.9)
opt.zero_grad(set_to_none=False)
xd = [ops.to_device_layout(x_all[i:i + bs].to(dev), torch.bfloat16) for i in range(0, n, bs)]
yd = [y_all[i:i + bs].to(dev) for i in range(0, n, bs)]
model.train()
for step in range(200):
    k = step % len(xd)
    loss = ops.softmax_cross_entropy(model(xd[k]), yd[k])
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=False)
    if step % 20 == 19:
        print(f"step {step + 1}: loss {float(loss.detach()):.4f}", flush=True)
model.eval()
with torch.no_grad():
    hits = sum(int((model(a).argmax(1) == b).sum()) for a, b in zip(xd, yd))
print(f"training-set accuracy {100.0 * hits / n:.1f}%", flush=True)
