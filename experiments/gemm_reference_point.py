"""Reference point for the ToMP encoder GEMMs: the vendor library (hipBLASLt / rocBLAS through torch.mm, fp32, no TF32)
on the same shapes, timed with events over 50 back-to-back calls.  Measurement aid only -- not part of the product."""
import json
import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda"
M = 1944
out = {}
for name, (N, K) in {"qkv": (768, 256), "out_proj": (256, 256), "ffn1": (2048, 256), "ffn2": (256, 2048)}.items():
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev)
    for _ in range(5):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        c = a @ w.t()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    out[name] = {"us": round(us, 2), "TFLOPs": round(2.0 * M * N * K / us / 1e6, 1)}
q = torch.randn(2, 8, 972, 32, device=dev)
for _ in range(3):
    o = torch.nn.functional.scaled_dot_product_attention(q, q, q)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    o = torch.nn.functional.scaled_dot_product_attention(q, q, q)
e1.record()
torch.cuda.synchronize()
out["sdpa_fp32_2x8x972x32"] = {"us": round(e0.elapsed_time(e1) * 1e3 / 50, 2)}
print(json.dumps(out))
