"""Diagnostic: mean nsd_step launch duration of each of several consecutive event-carrying decodes of the large batch,
first in a fresh process, then after the default-size workload ran in the same process."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from silent_speech_amd.decoder import DeviceTables, NSDecoder
from silent_speech_amd import synthetic as S
from silent_speech_amd.tables import NgramCounts

dev = torch.device("cuda", 0)
words = S.make_lexicon(200, 0)
lm = NgramCounts(5, floor_vsz=False).fit(S.make_sentences(words, 20000, 1))
tables = DeviceTables.from_words(words, lm)


def run(N, reps, label):
    tg, tl = S.pack_targets(S.targets_for_range(words, 0, N))
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
    dec = NSDecoder(tables, 16, 64, variant="I", lm_weight=0.05)
    ws = dec.workspace(N)
    for _ in range(2):
        dec.decode_synth_native(tgd, tld, base_seed=20260403, margin=60.0, ws=ws)
    torch.cuda.synchronize()
    sm = np.zeros(64, np.float32)
    out = []
    for _ in range(reps):
        dec.decode_synth_native(tgd, tld, base_seed=20260403, margin=60.0, ws=ws, step_ms=sm)
        out.append(sm[:ws.steps_run].mean() * 1e3)
    print(label, N, " ".join(f"{x:.1f}" for x in out), flush=True)


order = sys.argv[1] if len(sys.argv) > 1 else "big-first"
if order == "big-first":
    run(131072, 6, "fresh process:")
run(4096, 6, "default size:")
run(131072, 6, "after the default size:")
