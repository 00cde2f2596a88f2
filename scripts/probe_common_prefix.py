"""Diagnostic: how long is the part of the generated history that all hypotheses of an utterance share?  For every
step t of a decode, from the scorer's back-pointers: per utterance the number of positions s < t at which every
hypothesis of the beam descends from the same slot, against t.  Two workloads: the synthetic AR head of bench.py
(planted targets) and a random-weight Llama-shaped head (--llm: the scorer-share benchmark's)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from silent_speech_amd import synthetic as S
from silent_speech_amd.decoder import DeviceTables, NSDecoder
from silent_speech_amd.tables import NgramCounts

llm = "--llm" in sys.argv
variant = "I"
N, B, T = 64, 16, 64
dev = torch.device("cuda", 0)
words = S.make_lexicon(200, 0)
lmc = NgramCounts(5, floor_vsz=False).fit(S.make_sentences(words, 20000, 1))
dec = NSDecoder(DeviceTables.from_words(words, lmc), B, T, variant=variant, lm_weight=0.05)
ws = dec.workspace(N)
dec.init(ws)
if llm:
    from silent_speech_amd.ar_head import HipLlamaStep, LlamaDims
    dims = LlamaDims(512, 4, 8, 2, 64, 1408)
    head = HipLlamaStep(dims, dtype=torch.bfloat16, device=dev, seed=0)
    head.encode_context(torch.randn(N, 32, dims.hidden, device=dev, dtype=torch.bfloat16) * 0.5, B, T)
    fn = head.step_fn_batched()
else:
    tg, tl = S.pack_targets(S.targets_for_range(words, 0, N))
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
rows = []
for t in range(T):
    if llm:
        dec.step(ws, fn(None, None, ws).contiguous())
    else:
        dec.synth_logits(ws, 20260403, 0, tgd, tld, 60.0)
        dec.step(ws)
    bp = ws.backptr[:t + 1].cpu().numpy().reshape(t + 1, N, B).astype(np.int64)
    nb = (ws.utt[:N].cpu().numpy() & 0xFF)
    # ancestors of the current hypotheses at every earlier position (slot at step s that hypothesis r descends from)
    anc = np.tile(np.arange(B), (N, 1))
    common = np.zeros(N, np.int64)
    shared = []
    for s in range(t, -1, -1):
        anc = np.take_along_axis(np.minimum(bp[s], B - 1), anc, axis=1)       # slot at step s-1 ... (parent ranks)
        live = np.arange(B)[None, :] < nb[:, None]
        first = anc[:, :1]
        shared.append(((anc == first) | ~live).all(1))
    shared = np.array(shared[::-1])                                            # [t+1, N]: position s shared by all?
    # length of the shared prefix (positions 0..k-1 all shared)
    pref = np.where(shared.all(0), t + 1, (~shared).argmax(0))
    rows.append((t + 1, pref.mean(), np.percentile(pref, 10), pref.min()))
for t1, m, p10, mn in rows[::4]:
    print(f"after step {t1:2d}: shared prefix mean {m:5.1f}  p10 {p10:5.1f}  min {mn:3d}  (of {t1} generated positions)")
