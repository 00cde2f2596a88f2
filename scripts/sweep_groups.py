"""Tuning sweep: nsd_step launch time by lanes-per-utterance (G), batch size and decoder variant."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from silent_speech_amd.decoder import DeviceTables, NSDecoder
from silent_speech_amd import synthetic as S
from silent_speech_amd.tables import NgramCounts

dev = torch.device("cuda", 0)
words = S.make_lexicon(200, 0)
corpus = S.make_sentences(words, 20000, 1)
Ns = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["4096", "32768", "131072"])]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for variant in ("U", "I"):
    lm = NgramCounts(5, floor_vsz=(variant == "U")).fit(corpus)
    tables = DeviceTables.from_words(words, lm)
    for N in Ns:
        tg, tl = S.pack_targets(S.make_sentences(words, N, 1000, 3, 7))
        tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
        ref = None
        for G in (16, 32, 64):
            if G < B:
                continue
            dec = NSDecoder(tables, B, 64, variant=variant, lm_weight=0.05, word_bonus=0.35, lanes_per_utt=G)
            ws = dec.workspace(N)
            sm = np.zeros(64, np.float32)
            for rep in range(3):
                t0 = time.perf_counter()
                ids, lens, sc = dec.decode_synth_native(tgd, tld, base_seed=20260403, margin=60.0, ws=ws, step_ms=sm)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            key = (ids.cpu().numpy().tobytes(), sc.cpu().numpy().tobytes())
            same = ref is None or key == ref
            ref = ref or key
            print(f"{variant} N={N:7d} B={B} G={G:2d}: decode {dt*1e3:8.2f} ms  step launches mean {sm.mean()*1e3:8.1f} us  "
                  f"max {sm.max()*1e3:8.1f} us  first8 {[round(float(x)*1e3) for x in sm[:8]]}  same_result={same}", flush=True)
