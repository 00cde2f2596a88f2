"""Diagnostic: how much of one nsd_step wave's time is a cold start?

A library built with -DNSD_STAMPS -DNSD_REPEAT=3 runs the whole step three times inside one launch on the same
input (distinct in / out states), stamping the shader clock at the phase boundaries of every run.  The first run
sees what a real launch sees (instruction cache, address translations and data caches as the previous kernels left
them); the later runs find everything in place.  The difference is what a launch pays for starting cold.

    python scripts/phase_repeat.py [variant] [beam] [N ...]     (build the library beforehand: --build)
"""
import ctypes as C
import subprocess
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from silent_speech_amd import build as B

SO = ROOT / "silent-speech_amd" / "libnsdecode_repeat.so"
REPS = 3
if len(sys.argv) > 1 and sys.argv[1] == "--build":
    cmd = [B.hipcc(), *B.FLAGS, "-DNSD_STAMPS", f"-DNSD_REPEAT={REPS}", f"-I{ROOT/'include'}", f"-I{B.CSRC}",
           *[str(B.CSRC / s) for s in B.SOURCES], "-o", str(SO)]
    subprocess.run(cmd, check=True)
    print("built", SO)
    sys.exit(0)

import numpy as np, torch
from silent_speech_amd import lib
from silent_speech_amd.decoder import DeviceTables, NSDecoder
from silent_speech_amd import synthetic as S
from silent_speech_amd.tables import NgramCounts

variant = sys.argv[1] if len(sys.argv) > 1 else "I"
BEAM = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sizes = [int(x) for x in sys.argv[3:]] or [64, 4096]
dev = torch.device("cuda", 0)
L = lib.load()
R = C.CDLL(str(SO))
R.nsd_step.argtypes = L.nsd_step.argtypes
R.nsd_debug_set_stamps.argtypes = [C.c_void_p]
words = S.make_lexicon(200, 0)
lm = NgramCounts(5, floor_vsz=(variant == "U")).fit(S.make_sentences(words, 20000, 1))
tables = DeviceTables.from_words(words, lm)
names = ["state+records", "passes", "bounds+score", "select", "write"]
for N in sizes:
    tg, tl = S.pack_targets(S.make_sentences(words, N, 1000, 3, 7))
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
    dec = NSDecoder(tables, BEAM, 64, variant=variant, lm_weight=0.05, word_bonus=0.35)
    ws, ws2 = dec.workspace(N), dec.workspace(N)
    stamps = torch.zeros(N * REPS, 16, dtype=torch.int64, device=dev)
    assert R.nsd_debug_set_stamps(stamps.data_ptr()) == 0
    dec.init(ws)
    for t in range(24):
        dec.synth_logits(ws, 20260403, 0, tgd, tld, 60.0)
        if t in (8, 20, 21):
            stamps.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = R.nsd_step(tables.handle, C.byref(dec.params), ws.logits.data_ptr(), C.byref(ws.ptrs), C.byref(ws2.ptrs),
                            ws2.backptr.data_ptr(), ws2.token.data_ptr(), None, N, BEAM, 40,
                            torch.cuda.current_stream().cuda_stream)
            e1.record()
            assert rc == 0, rc
            torch.cuda.synchronize()
            s = stamps.cpu().numpy().reshape(N, REPS, 16)
            act = s[:, 0, 5] > 0
            print(f"N {N} step {t}: launch of {REPS} runs {e0.elapsed_time(e1) * 1e3:.1f} us, {act.sum()} active waves, "
                  f"candidates {s[act, 0, 10].mean():.0f}")
            for r in range(REPS):
                d = np.diff(s[act, r, :6].astype(np.float64), axis=1)
                tot = (s[act, r, 5] - s[act, r, 0]).astype(np.float64)
                gap = (s[act, r, 0] - s[act, r - 1, 5]).astype(np.float64).mean() if r else 0.0
                print(f"   run {r}: wave mean {tot.mean():7.0f} p99 {np.percentile(tot, 99):7.0f} max {tot.max():7.0f} ticks | "
                      + "  ".join(f"{n} {d[:, i].mean():6.0f}" for i, n in enumerate(names))
                      + f" | logits wait {(s[act, r, 6] - s[act, r, 1]).mean():.0f}  gap before run {gap:.0f}")
        dec.step(ws)
