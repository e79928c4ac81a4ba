"""Times the prefill / ViT attention kernel on the bench shapes through the C ABI (GPU box only) and compares the output
with a stored one: ORC_ATTN_QB=<variant> python tools/attn_bench.py [ref.pt]  (the first run writes ref.pt, later runs compare)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle_amd import engine as E

d = torch.device('cuda:0')
ref_path = sys.argv[1] if len(sys.argv) > 1 else None
ref = torch.load(ref_path, weights_only=True) if ref_path and os.path.exists(ref_path) else None
outs = {}
g = torch.Generator(device='cpu').manual_seed(5)
for name, (B, H, S, hd, causal) in {'prefill': (8, 32, 739, 128, True), 'vit': (24, 16, 577, 64, False)}.items():
    q, k, v = (torch.randn(B, H, S, hd, generator=g).to(torch.bfloat16).to(d) for _ in range(3))
    kbeg = torch.zeros(B, dtype=torch.int32, device=d) if causal else None
    for _ in range(5):
        o = E.test_attention(q, k, v, kbeg=kbeg, causal=causal)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 100
    e0.record()
    for _ in range(n):
        o = E.test_attention(q, k, v, kbeg=kbeg, causal=causal)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    flops = 4.0 * B * H * S * S * hd * (0.5 if causal else 1.0)
    outs[name] = o.cpu()
    line = f'variant {os.environ.get("ORC_ATTN_QB", "default")} {name}: {us:7.1f} us  {flops / us * 1e-6:6.1f} TFLOP/s'
    if ref is not None:
        df = (outs[name].float() - ref[name].float()).abs().max().item()
        line += f'  max|diff vs ref| {df:.3e}  bit-equal {torch.equal(outs[name], ref[name])}'
    print(line, flush=True)
if ref_path and ref is None:
    torch.save(outs, ref_path)
