"""Diagnostic: kernel durations of the decode loop's two kernels when each runs back to back on its own,
for comparison with their durations inside the alternating loop (run under rocprofv3 --kernel-trace)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from silent_speech_amd.decoder import DeviceTables, NSDecoder
from silent_speech_amd import synthetic as S
from silent_speech_amd.tables import NgramCounts

variant, N = "I", 4096
dev = torch.device("cuda", 0)
words = S.make_lexicon(200, 0)
lm = NgramCounts(5, floor_vsz=False).fit(S.make_sentences(words, 20000, 1))
tg, tl = S.pack_targets(S.make_sentences(words, N, 1000, 3, 7))
tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
dec = NSDecoder(DeviceTables.from_words(words, lm), 16, 64, variant=variant, lm_weight=0.05)
ws = dec.workspace(N)
dec.init(ws)
for t in range(12):                                   # reach a typical mid-decode state
    dec.synth_logits(ws, 20260403, 0, tgd, tld, 60.0)
    dec.step(ws)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
e[0].record()
for _ in range(64):
    dec.synth_logits(ws, 20260403, 0, tgd, tld, 60.0)
e[1].record()
torch.cuda.synchronize()
print("synth only: %.2f us per launch" % (e[0].elapsed_time(e[1]) * 1e3 / 64))
# the same step again and again: the state advances, logits stay (results are meaningless, timing is not)
e[2].record()
for _ in range(20):
    dec.step(ws)
e[3].record()
torch.cuda.synchronize()
print("step only: %.2f us per launch" % (e[2].elapsed_time(e[3]) * 1e3 / 20))
