"""nsd_kv_reorder against the HBM roofline: Llama-3.2-3B-shaped cache (28 layers, 8 kv heads, 128-wide bf16 rows),
1024 hypotheses, 64 positions of which t are copied; the same copy with torch.index_select for comparison."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from silent_speech_amd import lib

Lh = lib.load()
dev = torch.device("cuda", 0)
L, N, B, H, T, hd = 28, 64, 16, 8, 64, 128
M = N * B
src = torch.randn(L, M, H, T, hd, device=dev, dtype=torch.bfloat16)
dst = torch.empty_like(src)
bp = torch.randint(0, B, (M,), device=dev, dtype=torch.uint8)
parent = (torch.arange(N, device=dev) * B).repeat_interleave(B) + bp.long()
st = torch.cuda.current_stream().cuda_stream
for t in (8, 32, 64):
    def hip():
        lib.check(Lh.nsd_kv_reorder(src.data_ptr(), dst.data_ptr(), bp.data_ptr(), L, M, B, H, T, t, hd * 2, st), "kv")
    def tor():
        for l in range(L):
            dst[l, :, :, :t] = src[l].index_select(0, parent)[:, :, :t]
    for name, fn in (("nsd_kv_reorder", hip), ("torch index_select + copy", tor)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        byts = 2 * L * M * H * t * hd * 2                      # read + write
        print(f"t={t:2d} {name:28s} {us:9.1f} us  {byts / us / 1e3:7.0f} GB/s ({byts / us / 8e6 * 100:.0f}% of 8 TB/s)")
