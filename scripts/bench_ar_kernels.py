"""Timing of the elementwise kernels of the AR step at the Llama-3.2-3B shape (1 024 hypotheses, hidden 3 072,
intermediate 8 192, 24 + 2 x 8 heads of 128, bf16): nsd_add_rmsnorm, nsd_rope_kv_append, nsd_swiglu.  HIP events around
runs of 20 launches (buffers cycle through several copies so that a launch does not find its input in the caches its
predecessor left).  usage: bench_ar_kernels.py [rows]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from silent_speech_amd import lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H, I, heads, kvh, hd = 3072, 8192, 24, 8, 128
dev = torch.device("cuda", 0)
L = lib.load()
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16
K = 6                                              # copies of every buffer
x = [torch.randn(M, H, device=dev).to(bf) for _ in range(K)]
d = [torch.randn(M, H, device=dev).to(bf) for _ in range(K)]
w = torch.ones(H, device=dev, dtype=bf)
h = [torch.empty(M, H, device=dev, dtype=bf) for _ in range(K)]
gu = [torch.randn(M, 2 * I, device=dev).to(bf) for _ in range(K)]
act = [torch.empty(M, I, device=dev, dtype=bf) for _ in range(K)]
qkv = [torch.randn(M, (heads + 2 * kvh) * hd, device=dev).to(bf) for _ in range(K)]
q = [torch.empty(M, heads, hd, device=dev, dtype=bf) for _ in range(K)]
ko = [torch.empty(M, kvh, hd, device=dev, dtype=bf) for _ in range(K)]
vo = [torch.empty(M, kvh, hd, device=dev, dtype=bf) for _ in range(K)]
cos = torch.rand(hd, device=dev); sin = torch.rand(hd, device=dev)


def timed(name, fn, nbytes, reps=20, rounds=5):
    for i in range(K):
        fn(i)
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for r in range(reps):
            fn(r % K)
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / reps)
    us = float(np.median(out))
    print(f"{name:34s} {us:7.2f} us per launch  {nbytes / us / 1e6:6.2f} TB/s of {nbytes / 1e6:.1f} MB", flush=True)


timed("nsd_add_rmsnorm (add + norm)", lambda i: L.nsd_add_rmsnorm(x[i].data_ptr(), d[i].data_ptr(), w.data_ptr(), 1e-5, M, H, 1,
                                                                  x[i].data_ptr(), h[i].data_ptr(), st), 4 * M * H * 2)
timed("nsd_add_rmsnorm (norm only)", lambda i: L.nsd_add_rmsnorm(x[i].data_ptr(), None, w.data_ptr(), 1e-5, M, H, 1, None,
                                                                 h[i].data_ptr(), st), 2 * M * H * 2)
timed("nsd_rope_kv_append", lambda i: L.nsd_rope_kv_append(qkv[i].data_ptr(), cos.data_ptr(), sin.data_ptr(), M, heads, kvh, hd, 1,
                                                           q[i].data_ptr(), ko[i].data_ptr(), vo[i].data_ptr(), st),
      2 * M * (heads + 2 * kvh) * hd * 2)
timed("nsd_swiglu", lambda i: L.nsd_swiglu(gu[i].data_ptr(), M, I, 1, act[i].data_ptr(), st), 3 * M * I * 2)
