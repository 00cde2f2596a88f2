"""Diagnostic: nsd_step launch duration (HIP events with the kernel's own timestamps) against the number of
utterances, on the benchmark workload.  usage: launch_vs_n.py [variant] [beam] [N ...]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from silent_speech_amd.decoder import DeviceTables, NSDecoder
from silent_speech_amd import synthetic as S
from silent_speech_amd.tables import NgramCounts
variant = sys.argv[1] if len(sys.argv) > 1 else "I"
BEAM = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sizes = [int(x) for x in sys.argv[3:]] or [1, 64, 256, 1024, 2048, 4096, 8192, 16384]
dev = torch.device("cuda", 0)
words = S.make_lexicon(200, 0)
lm = NgramCounts(5, floor_vsz=(variant == "U")).fit(S.make_sentences(words, 20000, 1))
tables = DeviceTables.from_words(words, lm)
for N in sizes:
    targets = S.targets_for_range(words, 0, N)
    tg, tl = S.pack_targets(targets)
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
    dec = NSDecoder(tables, BEAM, 64, variant=variant, lm_weight=0.05, word_bonus=0.35)
    ws = dec.workspace(N)
    sm = np.zeros(64, np.float32)
    rows = []
    for rep in range(6):
        dec.decode_synth_native(tgd, tld, base_seed=20260403, margin=60.0, ws=ws, step_ms=sm)
        if rep >= 2:
            rows.append(sm[:ws.steps_run].copy() * 1e3)
    d = np.stack(rows)
    print(f"N {N:6d}: nsd_step launch mean {d.mean():7.2f} us  (steps 10-63: {d[:, 10:].mean():7.2f} us, min step {d.mean(0).min():6.2f}, max step {d.mean(0).max():6.2f})", flush=True)
