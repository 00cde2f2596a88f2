"""Diagnostic: one decode of N utterances against the same N utterances as K shards decoded on K streams at once
(each shard's launches enqueued by nsd_decode_synth on its own stream).  Wall time per whole decode, both ways."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from silent_speech_amd import synthetic as S
from silent_speech_amd.decoder import DeviceTables, NSDecoder
from silent_speech_amd.tables import NgramCounts

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
variant = sys.argv[2] if len(sys.argv) > 2 else "I"
B, T = 16, 64
dev = torch.device("cuda", 0)
words = S.make_lexicon(200, 0)
lm = NgramCounts(5, floor_vsz=(variant == "U")).fit(S.make_sentences(words, 20000, 1))
dec = NSDecoder(DeviceTables.from_words(words, lm), B, T, variant=variant, lm_weight=0.05)
tg, tl = S.pack_targets(S.targets_for_range(words, 0, N))
tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)

def run(K, reps=20):
    bounds = [(N * k // K, N * (k + 1) // K) for k in range(K)]
    wss = [dec.workspace(hi - lo) for lo, hi in bounds]
    streams = [torch.cuda.Stream() for _ in range(K)]
    outs = None
    def once():
        res = []
        for (lo, hi), ws, st in zip(bounds, wss, streams):
            with torch.cuda.stream(st):
                res.append(dec.decode_synth_native(tgd[lo:hi], tld[lo:hi], base_seed=20260403, margin=60.0, utt0=lo, ws=ws))
        return res
    for _ in range(3):
        outs = once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        outs = once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ids = torch.cat([o[0] for o in outs]).cpu().numpy(); lens = torch.cat([o[1] for o in outs]).cpu().numpy()
    return dt, ids, lens

base = None
for K in (1, 2, 3, 4, 8):
    dt, ids, lens = run(K)
    if base is None:
        base = (ids.copy(), lens.copy())
    same = bool((ids == base[0]).all() and (lens == base[1]).all())
    print(f"{K} stream(s): {dt * 1e3:7.3f} ms per decode of {N} utterances ({dt * 1e6 / T:6.2f} us per step)  equal to one stream: {same}", flush=True)
