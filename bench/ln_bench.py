"""LayerNorm forward / backward pass bandwidth on the ViT-B/16 and GPT-2 medium activation shapes.

    python bench/ln_bench.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from attn_bench import _time  # noqa: E402


def main():
    from pytorch_distributed_training_amd.ops import hip_transformer as ht

    dev = torch.device("cuda", 0)
    print(f"{'rows':>8}{'D':>6}{'fwd us':>9}{'TB/s':>7}{'bwd us':>9}{'TB/s':>7}   (bytes: 2 tensors forward, 3 backward)")
    for rows, d in ((1024 * 197, 768), (32 * 1024, 1024), (128 * 1024, 1024), (8192, 2048)):
        x = torch.randn(rows, d, device=dev).to(torch.bfloat16).requires_grad_(True)
        w = torch.ones(d, device=dev, requires_grad=True)
        b = torch.zeros(d, device=dev, requires_grad=True)
        g = torch.randn(rows, d, device=dev).to(torch.bfloat16)
        with torch.no_grad():
            tf = _time(lambda: ht.layer_norm(x, w, b, 1e-5, None))
        y = ht.layer_norm(x, w, b, 1e-5, None)

        def bwd():
            x.grad = None
            y.backward(g, retain_graph=True)

        tb = _time(bwd)
        nbytes = rows * d * 2
        print(f"{rows:8d}{d:6d}{tf * 1e6:9.1f}{2 * nbytes / tf / 1e12:7.2f}{tb * 1e6:9.1f}{3 * nbytes / tb / 1e12:7.2f}", flush=True)


if __name__ == "__main__":
    main()
