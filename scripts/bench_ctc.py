"""CTC greedy proposal + LM rerank (the CTC + rerank benchmark configuration) timing: HIP events around nsd_ctc_greedy."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from silent_speech_amd.decoder import ctc_greedy_batch

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Tp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = torch.Generator(device=dev); g.manual_seed(0)
x = torch.randn(N, Tp, 40, device=dev, generator=g)
# planted frame-repeated targets with blanks
tok = torch.randint(0, 40, (N, Tp // 4 + 1), device=dev, generator=g).repeat_interleave(4, dim=1)[:, :Tp]
x.scatter_add_(2, tok.unsqueeze(-1), torch.full((N, Tp, 1), 6.0, device=dev))
for _ in range(3):
    ids, lens = ctc_greedy_batch(x)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a, b in ev:
    a.record(); ids, lens = ctc_greedy_batch(x); b.record()
torch.cuda.synchronize()
ms = np.array([a.elapsed_time(b) for a, b in ev])
byts = N * Tp * 160 + int(lens.sum().item()) + 4 * N
print(f"nsd_ctc_greedy N={N} T'={Tp}: median {np.median(ms)*1e3:.1f} us  min {ms.min()*1e3:.1f} us  "
      f"{byts/np.median(ms)/1e6:.0f} GB/s ({byts/np.median(ms)/1e6/8000:.1%} of 8 TB/s)  mean len {lens.float().mean().item():.1f}")
