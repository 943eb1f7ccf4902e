"""Attention forward / backward time against sequence length at a constant token count (B * S = 32768, 16 heads).

    python bench/attn_sweep.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from attn_bench import _time  # noqa: E402


def main():
    from pytorch_distributed_training_amd.ops import hip_transformer as ht

    dev = torch.device("cuda", 0)
    h = 16
    print(f"{'S':>6}{'causal':>7}{'fwd us':>9}{'TF/s':>7}{'bwd us':>9}{'TF/s':>7}{'us per 64-key tile step of a workgroup (fwd)':>50}")
    for causal in (False, True):
        for s in (128, 256, 512, 1024, 2048, 4096):
            b = 32768 // s
            qkv = torch.randn(b, s, 3 * h * 64, device=dev).to(torch.bfloat16).requires_grad_(True)
            g = torch.randn(b, s, h * 64, device=dev).to(torch.bfloat16)
            flops = 4.0 * b * h * s * s * 64 * (0.5 if causal else 1.0)
            with torch.no_grad():
                tf = _time(lambda: ht.attention_qkv(qkv, h, causal))
            out = ht.attention_qkv(qkv, h, causal)

            def bwd():
                qkv.grad = None
                out.backward(g, retain_graph=True)

            tb = _time(bwd)
            nblk = (s + 127) // 128
            tiles = sum(min(s, 128 * (i + 1)) // 64 if causal else s // 64 for i in range(nblk)) * b * h  # workgroup tile steps
            per = tf * 1e6 / (tiles / 768.0)  # 768 workgroups resident at a time
            print(f"{s:6d}{int(causal):7d}{tf * 1e6:9.1f}{flops / tf / 1e12:7.0f}{tb * 1e6:9.1f}{2.5 * flops / tb / 1e12:7.0f}{per:20.2f}", flush=True)


if __name__ == "__main__":
    main()
