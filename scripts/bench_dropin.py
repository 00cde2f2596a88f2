"""The drop-in entry point with host-side logits: constrained_beam_decode(step_fn, ...) one utterance at a time, step_fn
returning a CPU tensor as a model on another device would hand it over, so every step pays the host -> device copy of
its rows and the device -> host read of the step's outcome.  Beside it the Python restatement of the reference on the
same calls.  usage: bench_dropin.py [n_utts] [beam] [variant]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from oracle import ns_oracle as O, synth
from silent_speech_amd import TextTransform
from silent_speech_amd.decoder import constrained_beam_decode, ids_to_text

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
beam = int(sys.argv[2]) if len(sys.argv) > 2 else 16
variant = sys.argv[3] if len(sys.argv) > 3 else "I"
torch.set_num_threads(1)
words = synth.make_lexicon(200, 0)
corpus = synth.make_sentences(words, 20000, 1)
targets = synth.make_sentences(words, n, 1000, 3, 7)
trie = O.build_trie(words)
lm = O.NgramLM(5, floor_vsz=(variant == "U")).fit(corpus)
calls = [0]


def make_step(u):
    useed, tgt = synth.utt_seed(20260403, u), synth.text_to_tokens(targets[u])

    def step_fn(ids):
        calls[0] += 1
        return torch.from_numpy(synth.logits_row(useed, ids, tgt, 60.0)).unsqueeze(0)
    return step_fn


kw = dict(lm=lm, lm_weight=0.05, oov_penalty=1.0, variant=variant)
constrained_beam_decode(make_step(0), TextTransform, beam, 64, trie, **kw)          # tables built, kernels loaded
torch.cuda.synchronize()
calls[0] = 0
t0 = time.perf_counter()
got = [constrained_beam_decode(make_step(u), TextTransform, beam, 64, trie, **kw) for u in range(n)]
torch.cuda.synchronize()
t_gpu, c_gpu = time.perf_counter() - t0, calls[0]
# the synthetic step_fn alone, on the same call sequence length
t0 = time.perf_counter()
f = make_step(0)
for _ in range(c_gpu):
    f([37, 1, 2, 3])
t_fn = time.perf_counter() - t0
calls[0] = 0
t0 = time.perf_counter()
want = [O.constrained_beam_decode(make_step(u), O.TextTransformLike, beam, 64, trie, **kw) for u in range(n)]
t_cpu = time.perf_counter() - t0
assert got == want, "drop-in and restatement disagree"
chars = sum(len(ids_to_text(TextTransform, g)) for g in got)
steps = sum(len(g) - 1 for g in got)
print(f"variant {variant} beam {beam}: {n} utterances, {chars} chars, {c_gpu} step_fn calls ({t_fn:.2f} s inside the synthetic step_fn)")
print(f"  drop-in (HIP, host logits): {t_gpu:.2f} s = {chars / t_gpu:.0f} chars/s, {1e3 * (t_gpu - t_fn) / max(steps, 1):.3f} ms per step outside step_fn")
print(f"  Python restatement, 1 core: {t_cpu:.2f} s = {chars / t_cpu:.0f} chars/s, {1e3 * (t_cpu - t_fn) / max(steps, 1):.3f} ms per step outside step_fn")
