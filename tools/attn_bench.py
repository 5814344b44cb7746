"""Time the attention kernel alone at the 32 x 32 level of the ch = 64 networks (one head, 1024 tokens).
    MCEDM_ATTN_SPLIT=1 python tools/attn_bench.py     # LDS-staged kernel
    MCEDM_ATTN_SPLIT=2 python tools/attn_bench.py     # key-split kernel"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = importlib.import_module("m-cedm_amd.lib")
for B, heads, T in ((32, 1, 1024), (8, 1, 1024), (32, 4, 256), (32, 2, 256), (8, 2, 256)):
    qkv = torch.randn(B, heads * 192, T, 1, device="cuda")
    for _ in range(3):
        lib.op_attention(qkv, heads)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20):
        lib.op_attention(qkv, heads)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 20
    fl = 4.0 * B * heads * T * T * 64
    print(f"B={B} heads={heads} T={T}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s")
