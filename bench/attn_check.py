"""Attention kernels against fp32 PyTorch on awkward shapes (ragged sequence ends, one key, rows whose scores
differ by hundreds of octaves), then the timings of bench/attn_bench.py.

    python bench/attn_check.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def rel(a, b):
    a, b = a.detach().float(), b.detach().float()
    return float((a - b).norm() / b.norm().clamp_min(1e-3 * a.numel() ** 0.5))  # a zero reference is compared absolutely


def main():
    from pytorch_distributed_training_amd.ops import hip_transformer as ht
    from pytorch_distributed_training_amd.ops import reference as ref_ops

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    bad = 0
    for b, s, h, causal, amp in [(2, 1, 2, True, 1), (2, 77, 3, False, 1), (2, 77, 3, True, 1), (3, 197, 12, False, 1), (2, 128, 2, True, 1),
                                 (2, 130, 2, True, 1), (1, 1024, 4, True, 1), (1, 1000, 4, False, 1), (2, 300, 2, True, 6), (2, 300, 2, False, 12),
                                 (1, 520, 2, True, 30)]:
        d = h * 64
        qkv = (torch.randn(b, s, 3 * d, device=dev) * amp).to(torch.bfloat16).requires_grad_(True)
        out = ht.attention_qkv(qkv, h, causal)
        qr = qkv.detach().float().requires_grad_(True)
        ref = ref_ops.attention_qkv(qr, h, causal)
        g = torch.randn_like(ref)
        out.backward(g.to(torch.bfloat16))
        ref.backward(g.to(torch.bfloat16).float())
        e = [rel(out, ref)] + [rel(qkv.grad.view(b, s, 3, d)[:, :, i], qr.grad.view(b, s, 3, d)[:, :, i]) for i in range(3)]
        ok = e[0] < 1e-2 and max(e[1:]) < 2.5e-2 and bool(torch.isfinite(out.float()).all()) and bool(torch.isfinite(qkv.grad.float()).all())
        bad += not ok
        print(f"B{b} S{s} H{h} causal={int(causal)} amp={amp}: out {e[0]:.2e} dq {e[1]:.2e} dk {e[2]:.2e} dv {e[3]:.2e} {'ok' if ok else 'BAD'}", flush=True)
    print("bad", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
