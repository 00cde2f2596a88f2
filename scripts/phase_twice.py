"""Diagnostic: wave durations of the same nsd_step launched twice in a row on identical input (stamps build).
If the second launch is tighter, the first paid for cold data / translations; if not, the cost is per launch.
    python scripts/phase_twice.py --build | [variant] [beam] [N]"""
import ctypes as C, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from silent_speech_amd import build as B
SO = ROOT / "silent-speech_amd" / "libnsdecode_stamps.so"
if len(sys.argv) > 1 and sys.argv[1] == "--build":
    subprocess.run([B.hipcc(), *B.FLAGS, "-DNSD_STAMPS", f"-I{ROOT/'include'}", f"-I{B.CSRC}",
                    *[str(B.CSRC / s) for s in B.SOURCES], "-o", str(SO)], check=True)
    sys.exit(0)
import numpy as np, torch
from silent_speech_amd import lib
from silent_speech_amd.decoder import DeviceTables, NSDecoder
from silent_speech_amd import synthetic as S
from silent_speech_amd.tables import NgramCounts
variant = sys.argv[1] if len(sys.argv) > 1 else "I"
BEAM = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
dev = torch.device("cuda", 0)
L = lib.load()
R = C.CDLL(str(SO))
R.nsd_step.argtypes = L.nsd_step.argtypes
R.nsd_debug_set_stamps.argtypes = [C.c_void_p]
words = S.make_lexicon(200, 0)
lm = NgramCounts(5, floor_vsz=(variant == "U")).fit(S.make_sentences(words, 20000, 1))
tables = DeviceTables.from_words(words, lm)
tg, tl = S.pack_targets(S.make_sentences(words, N, 1000, 3, 7))
tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
dec = NSDecoder(tables, BEAM, 64, variant=variant, lm_weight=0.05, word_bonus=0.35)
ws, ws2 = dec.workspace(N), dec.workspace(N)
stA = torch.zeros(N, 16, dtype=torch.int64, device=dev)
stB = torch.zeros(N, 16, dtype=torch.int64, device=dev)
names = ["state+records", "passes", "bounds+score", "select", "write"]
stream = torch.cuda.current_stream().cuda_stream
def launch(st):
    assert R.nsd_debug_set_stamps(st.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = R.nsd_step(tables.handle, C.byref(dec.params), ws.logits.data_ptr(), C.byref(ws.ptrs), C.byref(ws2.ptrs),
                    ws2.backptr.data_ptr(), ws2.token.data_ptr(), None, N, BEAM, 40, stream)
    e1.record()
    assert rc == 0
    return e0, e1
def show(tag, st, ev):
    s = st.cpu().numpy()
    act = s[:, 5] > 0
    d = np.diff(s[act, :6].astype(np.float64), axis=1)
    tot = (s[act, 5] - s[act, 0]).astype(np.float64)
    p = np.percentile(tot, [50, 90, 99, 100]).round()
    print(f"  {tag}: {ev[0].elapsed_time(ev[1]) * 1e3:6.1f} us  wave mean {tot.mean():7.0f}  p50/p90/p99/max {p} | "
          + "  ".join(f"{n} {d[:, i].mean():6.0f}" for i, n in enumerate(names)))
dec.init(ws)
for t in range(24):
    dec.synth_logits(ws, 20260403, 0, tgd, tld, 60.0)
    if t in (8, 20, 21):
        stA.zero_(); stB.zero_()
        torch.cuda.synchronize()
        a = launch(stA)
        b = launch(stB)
        torch.cuda.synchronize()
        print(f"N {N} step {t}")
        show("first launch ", stA, a)
        show("second launch", stB, b)
    dec.step(ws)
