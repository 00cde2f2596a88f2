"""Diagnostic: instructions per phase of one nsd_step launch.

Builds the library four more times with -DNSD_STOP_AT=k (the kernel returns at phase boundary k), decodes the
benchmark batch for some steps with the normal library and then launches every truncated build, and the normal one,
REPS times each on that same state and logits (the truncated kernels write nothing back).  Run under
`rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES ...`: scripts/phase_insts.sh reads the counter
file by dispatch order and prints the differences.

  build:  python scripts/phase_insts.py build      (here, before the GPU call)
  run:    python scripts/phase_insts.py run [variant] [utts] [beam] [step] [open]
"""
import ctypes as C, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from silent_speech_amd import build as B

OUT = ROOT / "build" / "diag"
STOPS = (1, 2, 3, 9, 12, 4)
REPS = 3

if sys.argv[1] == "build":
    OUT.mkdir(parents=True, exist_ok=True)
    for k in STOPS:
        so = OUT / f"libnsd_stop{k}.so"
        subprocess.run([B.hipcc(), *B.FLAGS, f"-DNSD_STOP_AT={k}", f"-I{ROOT/'include'}", f"-I{B.CSRC}",
                        *[str(B.CSRC / s) for s in B.SOURCES], "-o", str(so)], check=True)
        print("built", so)
    sys.exit(0)

import torch
from silent_speech_amd import lib, synthetic as S
from silent_speech_amd.decoder import DeviceTables, NSDecoder
from silent_speech_amd.tables import NgramCounts

variant = sys.argv[2] if len(sys.argv) > 2 else "I"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
BEAM = int(sys.argv[4]) if len(sys.argv) > 4 else 16
STEP = int(sys.argv[5]) if len(sys.argv) > 5 else 20
OPEN = len(sys.argv) > 6 and sys.argv[6] == "open"
dev = torch.device("cuda", 0)
words = S.make_lexicon(100000, 1, S.BASE_CHARS[:36], 2, 12) if OPEN else S.make_lexicon(200, 0)
lm = NgramCounts(5, floor_vsz=(variant == "U")).fit(S.make_sentences(words, 20000, 1))
tg, tl = S.pack_targets(S.targets_for_range(words, 0, N))
tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
dec = NSDecoder(DeviceTables.from_words(words, lm), BEAM, 64, variant=variant, lm_weight=0.05, word_bonus=0.35)
ws = dec.workspace(N)
dec.init(ws)
for t in range(STEP):
    dec.synth_logits(ws, 20260403, 0, tgd, tld, 60.0)
    dec.step(ws)
dec.synth_logits(ws, 20260403, 0, tgd, tld, 60.0)
torch.cuda.synchronize()
L = lib.load()
from silent_speech_amd.decoder import VOCAB
NB = ws.N * ws.B
step_args = (dec.tables.handle, C.byref(dec.params), ws.logits.data_ptr(), C.byref(ws.ptrs), C.byref(ws.ptrs),
             ws.backptr.data_ptr() + STEP * NB, ws.token.data_ptr() + STEP * NB, None, ws.N, ws.B, VOCAB,
             torch.cuda.current_stream().cuda_stream)
for k in STOPS:
    D = C.CDLL(str(OUT / f"libnsd_stop{k}.so"))
    D.nsd_step.argtypes = L.nsd_step.argtypes
    D.nsd_step.restype = C.c_int
    for _ in range(REPS):
        rc = D.nsd_step(*step_args)
        assert rc == 0, rc
    torch.cuda.synchronize()
for _ in range(REPS):
    assert L.nsd_step(*step_args) == 0
torch.cuda.synchronize()
print("done: last", REPS * (len(STOPS) + 1), "nsd_step dispatches are stop", STOPS, "then the whole kernel,", REPS, "each")
