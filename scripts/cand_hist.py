"""Diagnostic: how many candidates does an utterance have per step (lexicon-allowed children of its live hypotheses)?
Prints the distribution over (utterance, step) for the benchmark workload; the lanes of a wave are 64."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from silent_speech_amd import synthetic as S
from silent_speech_amd.decoder import DeviceTables, NSDecoder
from silent_speech_amd.tables import NgramCounts

variant = sys.argv[1] if len(sys.argv) > 1 else "I"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
BEAM = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda", 0)
words = S.make_lexicon(200, 0)
lm = NgramCounts(5, floor_vsz=(variant == "U")).fit(S.make_sentences(words, 20000, 1))
tg, tl = S.pack_targets(S.targets_for_range(words, 0, N))
dec = NSDecoder(DeviceTables.from_words(words, lm), BEAM, 64, variant=variant, lm_weight=0.05, word_bonus=0.35)
ws = dec.workspace(N, with_masks=True)
dec.decode_synth(torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev), base_seed=20260403, margin=60.0, ws=ws)
m = ws.masks[:ws.steps_run].cpu().numpy().astype(np.uint64).reshape(ws.steps_run, N, BEAM)
pc = np.zeros(m.shape, np.int64)
for b in range(37):
    pc += ((m >> np.uint64(b)) & np.uint64(1)).astype(np.int64)
C = pc.sum(-1)                                   # children per (step, utterance); carried hypotheses come on top
tlen = tl[None, :]
steps = np.arange(ws.steps_run)[:, None]
for name, sel in (("all steps", np.ones_like(C, bool)), ("steps inside the target", steps < tlen), ("steps after it", steps >= tlen)):
    c = C[sel]
    print(f"{name:26s} n {c.size:8d} mean {c.mean():6.1f}  <=64: {np.mean(c <= 64):.3f}  65-80: {np.mean((c > 64) & (c <= 80)):.3f}  "
          f"81-128: {np.mean((c > 80) & (c <= 128)):.3f}  >128: {np.mean(c > 128):.3f}  p50/p90/p99 {np.percentile(c, [50, 90, 99])}")
