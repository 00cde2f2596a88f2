"""Timing of the LM regulariser (forward + gradient) on a training-sized batch, next to the CPU restatement
of the reference's per-position Python loop (train_emg_llm.py:254-316)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from silent_speech_amd import synthetic as S
from silent_speech_amd.prior import ns_lm_regularizer, tables_from_lm
from silent_speech_amd.tables import NgramCounts

B, L, order = (int(a) for a in (sys.argv[1:4] + ["64", "128", "3"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0)
words = S.make_lexicon(200, 0)
corpus = S.make_sentences(words, 20000, 1)
tab = tables_from_lm(NgramCounts(order).fit(corpus))
rs = np.random.RandomState(0)
logits = (rs.standard_normal((B, L, 40)) * 3).astype(np.float32)
out_ids = np.full((B, L), -100, np.int64)
code = {c: i for i, c in enumerate("abcdefghijklmnopqrstuvwxyz0123456789 ")}
for i, s in enumerate(S.make_sentences(words, B, 55, 8, 20)):
    toks = [code[c] for c in s][:L - 1] + [38]
    out_ids[i, :len(toks)] = toks
x = torch.from_numpy(logits).to(dev).requires_grad_(True)
ids = torch.from_numpy(out_ids).to(dev)
for _ in range(5):
    x.grad = None
    ns_lm_regularizer(x, None, ids, tab).backward()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    x.grad = None
    loss = ns_lm_regularizer(x, None, ids, tab)
    loss.backward()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
valid = int((out_ids != -100).sum())
byts = B * L * 40 * 4 * 3 + B * L * 8 * 2          # logits read twice, gradient written, targets read twice
print(f"GPU forward+gradient: {us:.1f} us per batch  (B {B}, L {L}, {valid} valid positions, "
      f"{byts / us / 1e3:.1f} GB/s of logits/gradient traffic)  loss {float(loss.detach()):.6f}")
try:
    from oracle import ns_oracle as O
    lm = O.NgramLM(order).fit(corpus)
    t0 = time.perf_counter()
    wl, _ = O.lm_regularizer(lm, logits, out_ids)
    dt = time.perf_counter() - t0
    print(f"CPU restatement (1 core): {dt * 1e3:.1f} ms per batch, loss {float(wl):.6f}  -> {dt * 1e6 / us:.0f}x")
except ImportError:
    pass
