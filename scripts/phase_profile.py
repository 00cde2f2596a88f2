"""Diagnostic: where does one nsd_step wave spend its cycles?

Builds a separate library with -DNSD_STAMPS (lane 0 stores the shader clock at the phase boundaries into a
buffer of its own), runs the benchmark decode step by step and prints the mean cycles per phase for chosen
steps.  The stamps forbid overlaps the real kernel has: read the shares, not the totals.
"""
import subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from silent_speech_amd import build as B, lib

so = ROOT / "silent-speech_amd" / "libnsdecode_stamps.so"
if len(sys.argv) > 1 and sys.argv[1] == "--build":      # on the CPU box, before the GPU call
    cmd = [B.hipcc(), *B.FLAGS, "-DNSD_STAMPS", f"-I{ROOT/'include'}", f"-I{B.CSRC}", *[str(B.CSRC / s) for s in B.SOURCES], "-o", str(so)]
    subprocess.run(cmd, check=True)
    print("built", so)
    sys.exit(0)
if not so.exists():
    raise SystemExit(f"{so} is missing: run `python scripts/phase_profile.py --build` first")
lib.SO_PATH = so
from silent_speech_amd.decoder import DeviceTables, NSDecoder
from silent_speech_amd import synthetic as S
from silent_speech_amd.tables import NgramCounts
import ctypes as C

variant = sys.argv[1] if len(sys.argv) > 1 else "U"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
BEAM = int(sys.argv[3]) if len(sys.argv) > 3 else 16
OPEN = len(sys.argv) > 4 and sys.argv[4] == "open"
dev = torch.device("cuda", 0)
words = S.make_lexicon(100000, 1, S.BASE_CHARS[:36], 2, 12) if OPEN else S.make_lexicon(200, 0)
lm = NgramCounts(5, floor_vsz=(variant == "U")).fit(S.make_sentences(words, 20000, 1))
tg, tl = S.pack_targets(S.make_sentences(words, N, 1000, 3, 7))
tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
dec = NSDecoder(DeviceTables.from_words(words, lm), BEAM, 64, variant=variant, lm_weight=0.05, word_bonus=0.35)
ws = dec.workspace(N)
stamps = torch.zeros(N, 16, dtype=torch.int64, device=dev)   # one row per wave (a wave carries 64/G utterances)
L = lib.load()
L.nsd_debug_set_stamps.argtypes = [C.c_void_p]
assert L.nsd_debug_set_stamps(stamps.data_ptr()) == 0
dec.init(ws)
names = ["state+records", "passes(softmax,emit)", "score", "select", "write"]
for t in range(40):
    dec.synth_logits(ws, 20260403, 0, tgd, tld, 60.0)
    stamps.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); dec.step(ws); e1.record()
    torch.cuda.synchronize()
    if t in (0, 1, 2, 3, 5, 10, 20, 30, 39):
        s = stamps.cpu().numpy()
        act = s[:, 5] > 0
        d = np.diff(s[act, :6], axis=1).astype(np.float64)
        fine = {"pass0: until its logits are there": s[act, 6] - s[act, 1], "final select: threshold": s[act, 9] - s[act, 3],
                "final select: rank": s[act, 4] - s[act, 9]}
        tot = (s[act, 5] - s[act, 0]).astype(np.float64)
        Cc = s[act, 10].astype(np.int64)
        print(f"      per wave: candidates mean {Cc.mean():.0f} max {Cc.max()}  live {s[act, 11].mean():.1f}  tie re-ranks {s[act, 12].mean():.3f}  "
              f"threshold iterations {s[act, 13].mean():.2f}  prunes {s[act, 14].mean():.2f}  LDS-path searches {s[act, 15].mean():.2f}  "
              f"inside passes: scoring before prunes {s[act, 7].mean():.0f}  prunes {s[act, 8].mean():.0f} ticks (slots 7/8 hold these totals)")
        for lo_, hi_ in ((0, 64), (64, 96), (96, 128), (128, 160), (160, 1000)):
            m = (Cc > lo_) & (Cc <= hi_)
            if m.any():
                print(f"      C in ({lo_},{hi_}]: {m.sum():5d} waves, duration mean {tot[m].mean():7.0f} max {tot[m].max():7.0f} | "
                      + "  ".join(f"{n.split('(')[0]} {d[m, i].mean():6.0f}" for i, n in enumerate(names)))
        # (clock values of different waves are not comparable: every XCD counts on its own)
        slow = tot >= np.percentile(tot, 99)
        print(f"      slowest 1 % of the waves ({slow.sum()}): candidates mean {Cc[slow].mean():.0f}  duration mean {tot[slow].mean():.0f} | "
              + "  ".join(f"{n.split('(')[0]} {d[slow, i].mean():6.0f}" for i, n in enumerate(names))
              + " | pass0 logits wait %.0f" % (s[act, 6] - s[act, 1])[slow].mean())
        pct = np.percentile(tot, [50, 90, 99, 100])
        print(f"   wave duration p50/p90/p99/max {pct.round()}  phase max: "
              + "  ".join(f"{n.split('(')[0]} {d[:,i].max():.0f}" for i, n in enumerate(names)))
        print(f"step {t:2d}: active waves {act.sum():5d}  kernel {e0.elapsed_time(e1)*1e3:6.1f} us  "
              f"wave mean {tot.mean():8.0f} ticks | " + "  ".join(f"{n} {d[:,i].mean():7.0f}" for i, n in enumerate(names)) +
              " || " + "  ".join(f"{k} {v.mean():6.0f}" for k, v in fine.items()))
