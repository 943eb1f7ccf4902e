"""Attention time against the batch size at the GPT-2 shape (S 1024, 16 heads, causal): a time that does not halve
with the batch is launch ramp and tail (the last workgroups running on a mostly idle chip), not per-tile work.

    python bench/attn_batch_sweep.py [S]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from attn_bench import _time  # noqa: E402


def main():
    from pytorch_distributed_training_amd.ops import hip_transformer as ht

    dev = torch.device("cuda", 0)
    h, s = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    print(f"{'B':>5}{'fwd us':>9}{'TF/s':>7}{'bwd us':>9}{'TF/s':>7}{'fwd us/B':>10}{'bwd us/B':>10}")
    for b in (4, 8, 16, 32, 64, 128):
        qkv = torch.randn(b, s, 3 * h * 64, device=dev).to(torch.bfloat16).requires_grad_(True)
        g = torch.randn(b, s, h * 64, device=dev).to(torch.bfloat16)
        flops = 4.0 * b * h * s * s * 64 * 0.5
        with torch.no_grad():
            tf = _time(lambda: ht.attention_qkv(qkv, h, True))
        out = ht.attention_qkv(qkv, h, True)

        def bwd():
            qkv.grad = None
            out.backward(g, retain_graph=True)

        tb = _time(bwd)
        print(f"{b:5d}{tf * 1e6:9.1f}{flops / tf / 1e12:7.0f}{tb * 1e6:9.1f}{2.5 * flops / tb / 1e12:7.0f}{tf * 1e6 / b:10.2f}{tb * 1e6 / b:10.2f}", flush=True)


if __name__ == "__main__":
    main()
