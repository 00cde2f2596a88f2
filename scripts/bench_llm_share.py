"""Scorer share of a decode step next to the LLM forward.

One GPU, N utterances x B hypotheses: every step runs the KV-cached AR head (Llama-3.2-3B-shaped stand-in with
random bf16 weights, torch / hipBLASLt: silent_speech_amd/ar_head.py) and then one nsd_step launch on its
logits.  Both are bracketed by events on the launch stream; the JSON line gives the mean per-step times and the
scorer's share.  Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
"""
import argparse, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from silent_speech_amd import synthetic as S
from silent_speech_amd.ar_head import LLAMA_3_2_3B, HipLlamaStep, LlamaDims, TorchLlamaStep
from silent_speech_amd.decoder import DeviceTables, NSDecoder
from silent_speech_amd.tables import NgramCounts

ap = argparse.ArgumentParser()
ap.add_argument("--utts", type=int, default=64)
ap.add_argument("--beam", type=int, default=16)
ap.add_argument("--context", type=int, default=96, help="prompt + EMG adapter positions per utterance")
ap.add_argument("--tmax", type=int, default=64)
ap.add_argument("--small", action="store_true", help="a 4-layer, 512-wide model (smoke run)")
ap.add_argument("--variant", default="I")
ap.add_argument("--head", default="hip", choices=["hip", "torch"],
                help="hip: everything between the GEMMs in the library's kernels, cache never re-ordered; "
                     "torch: torch ops + nsd_kv_reorder")
args = ap.parse_args()

dev = torch.device("cuda", 0)
dims = LlamaDims(512, 4, 8, 2, 64, 1408) if args.small else LLAMA_3_2_3B
words = S.make_lexicon(200, 0)
lm = NgramCounts(5, floor_vsz=(args.variant == "U")).fit(S.make_sentences(words, 20000, 1))
dec = NSDecoder(DeviceTables.from_words(words, lm), args.beam, args.tmax, variant=args.variant, lm_weight=0.05)
head = (HipLlamaStep if args.head == "hip" else TorchLlamaStep)(dims, dtype=torch.bfloat16, device=dev, seed=0)
context = torch.randn(args.utts, args.context, dims.hidden, device=dev, dtype=torch.bfloat16) * 0.5


def decode(timed):
    ws = dec.workspace(args.utts)
    dec.init(ws)
    head.encode_context(context, args.beam, args.tmax)
    fn = head.step_fn_batched()
    ev = []
    for t in range(args.tmax):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        logits = fn(None, None, ws)
        e[1].record()
        dec.step(ws, logits.contiguous())
        e[2].record()
        ev.append(e)
    ids, lens, _ = dec.finish(ws)
    torch.cuda.synchronize()
    if not timed:
        return None
    llm = [a.elapsed_time(b) for a, b, _ in ev]
    sc = [b.elapsed_time(c) for _, b, c in ev]
    return llm, sc, int(lens.sum().item())


decode(False)
llm, sc, n_ids = decode(True)
tl, ts = sum(llm) / len(llm), sum(sc) / len(sc)
print(json.dumps({
    "head": args.head,
    "workload": f"{args.utts} utterances x beam {args.beam}, context {args.context}, T_max {args.tmax}, variant {args.variant}, "
                f"AR head {'small test model' if args.small else 'Llama-3.2-3B-shaped, random bf16 weights'}",
    "llm_step_ms_mean": tl, "llm_step_ms_first_last": [llm[0], llm[-1]],
    "scorer_step_ms_mean": ts, "scorer_share_of_step": ts / (tl + ts),
    "steps": len(llm), "ids_decoded": n_ids}))
