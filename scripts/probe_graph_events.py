"""Probe: can torch.cuda graphs capture ctypes-launched kernels, and do timing events inside a graph work?"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from silent_speech_amd.decoder import DeviceTables, NSDecoder
from silent_speech_amd import synthetic as S
from silent_speech_amd.tables import NgramCounts

dev = torch.device("cuda", 0)
words = S.make_lexicon(200, 0)
lm = NgramCounts(5).fit(S.make_sentences(words, 20000, 1))
targets = S.make_sentences(words, 4096, 1000, 3, 7)
tg, tl = S.pack_targets(targets)
tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
dec = NSDecoder(DeviceTables.from_words(words, lm), 16, 64, variant="U", lm_weight=0.05)
ws = dec.workspace(4096)
ids0, lens0, _ = dec.decode_synth(tgd, tld, base_seed=1, margin=60.0, ws=ws, check_every=0)
ids0, lens0 = ids0.clone(), lens0.clone()
torch.cuda.synchronize()

s = torch.cuda.Stream()
try:
    evs = [(torch.cuda.Event(enable_timing=True, external=True), torch.cuda.Event(enable_timing=True, external=True)) for _ in range(64)]
    ext = True
except TypeError:
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    ext = False
print("external events supported by torch:", ext)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    dec.init(ws)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        dec.init(ws)
        for t in range(64):
            dec.synth_logits(ws, 1, 0, tgd, tld, 60.0)
            try:
                evs[t][0].record()
            except Exception as e:
                print("record in capture failed:", e); raise
            dec.step(ws)
            evs[t][1].record()
        ws.steps_run = 64
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    print("graph replay of a full 64-step decode: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
ids1, lens1, _ = dec.finish(ws)
print("same result as eager:", bool((ids1 == ids0).all()), bool((lens1 == lens0).all()))
try:
    d = [a.elapsed_time(b) * 1e3 for a, b in evs]
    print("in-graph event pairs (us): first 10", [round(x, 1) for x in d[:10]], "mean %.2f" % (sum(d) / len(d)))
except Exception as e:
    print("elapsed_time on in-graph events failed:", repr(e))
