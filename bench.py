#!/usr/bin/env python3
"""NS-decode throughput benchmark: NS-decoded characters per second.

One "step" = one full NS decode of the resident batch: up to T_max scorer
launches (nsd_step) fed by the synthetic integer-hash AR head (nsd_synth_logits,
the stand-in for the frozen LLM's character head), early exit when every
utterance has finished, back-trace, and - with more than one rank - the final
gather of token ids.  Inputs (tables, planted targets) are resident in HBM
before the timed region starts.

Default workload (N = 1): the benchmark's second configuration - 4096 synthetic utterances,
beam 16, closed 200-word template lexicon trie + character 5-gram LM, decoded
by the copy of the NS decoder that inference_emg_llm.py itself calls (variant
I, inference_emg_llm.py:222-300: no EOS option, so every utterance runs all
T_max = 64 steps with 16 live hypotheses), beta 0.05, gamma 1.0, planted-target
margin 60.  The scripts/ns_utils.py copy (variant U, EOS offered at every
step) is measured in the same run and reported under "variant_U".  With
--gpus N every rank decodes its own 4096-utterance shard (weak scaling) and
rank 0 gathers the ids.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak (spec)
VARIANT_SRC = {"U": "scripts/ns_utils.py:194", "I": "inference_emg_llm.py:222", "E": "eval_ar.py:214"}


def build_workload(n_utts: int, utt0: int, lex_words: int, corpus_n: int, order: int, open_lexicon: bool):
    """Seeded lexicon / corpus / planted targets."""
    from silent_speech_amd import synthetic as synth
    if open_lexicon:
        words = synth.make_lexicon(lex_words, 1, synth.BASE_CHARS[:36], 2, 12)
    else:
        words = synth.make_lexicon(lex_words, 0, synth.BASE_CHARS[:26], 2, 8)
    corpus = synth.make_sentences(words, corpus_n, 1)
    # targets are drawn per global utterance block so that ranks decode different utterances
    targets = synth.make_sentences(words, n_utts, 1000 + utt0 // max(n_utts, 1), 3, 7)
    tg, tl = synth.pack_targets(targets)
    return words, corpus, targets, tg, tl


def popcount64(a: np.ndarray) -> np.ndarray:
    a = a.astype(np.uint64)
    a = a - ((a >> np.uint64(1)) & np.uint64(0x5555555555555555))
    a = (a & np.uint64(0x3333333333333333)) + ((a >> np.uint64(2)) & np.uint64(0x3333333333333333))
    a = (a + (a >> np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
    return ((a * np.uint64(0x0101010101010101)) >> np.uint64(56)).astype(np.int64)


def algorithmic_bytes(dec, tgd, tld, args):
    """Untimed instrumented decode: per executed step, live hypotheses and lexicon-allowed children.

    Per live hypothesis the scorer must move: its 40 fp32 logits (160 B), its state in and out, its trie
    node record, back-pointer and token (245 B in all), and two table slots (interior + terminal n-gram)
    for each allowed child and for the hypothesis itself: 245 + 32 * (c + 1) bytes.
    """
    ws = dec.workspace(tgd.shape[0], with_masks=True)
    ws.token.fill_(0xFF)
    dec.decode_synth(tgd, tld, base_seed=args.seed, margin=args.margin, utt0=args.utt0, ws=ws, check_every=1)
    torch.cuda.synchronize()
    T, N, B = ws.steps_run, ws.N, ws.B
    tok = ws.token[:T].cpu().numpy()
    masks = ws.masks[:T].cpu().numpy().view(np.uint64)
    live = np.zeros(T, np.int64)
    live[0] = N
    for t in range(1, T):
        live[t] = int(((tok[t - 1] != 0xFF) & (tok[t - 1] != 38)).sum())
    cands = np.array([int(popcount64(masks[t] & np.uint64((1 << 37) - 1)).sum()) for t in range(T)], np.int64)
    per_step = live * 245 + 32 * (cands + live)
    return T, live, cands, per_step


def cpu_baseline(words, corpus, targets, args, gpu_texts):
    """The reference algorithm on one host core: the Python restatement (oracle/ns_oracle.py: dict trie,
    whole-string n-gram score per candidate, one step_fn call and one torch log_softmax per live
    hypothesis, stable sort) on a bounded sample of the same utterances."""
    from oracle import ns_oracle as O, synth
    torch.set_num_threads(1)
    trie = O.build_trie(words)
    lm = O.NgramLM(args.order, floor_vsz=(args.variant == "U")).fit(corpus)
    n = min(args.cpu_utts, len(targets))
    chars = 0
    mism = 0
    t0 = time.perf_counter()
    for u in range(n):
        useed = synth.utt_seed(args.seed, args.utt0 + u)
        tgt = synth.text_to_tokens(targets[u])
        step_fn = lambda ids: torch.from_numpy(synth.logits_row(useed, ids, tgt, args.margin)).unsqueeze(0)
        ids = O.constrained_beam_decode(step_fn, O.TextTransformLike, args.beam, args.tmax, trie, lm,
                                        lm_weight=args.beta, oov_penalty=args.gamma, variant=args.variant,
                                        word_bonus=args.kappa)
        text = O.ids_to_text(O.TextTransformLike, ids)
        chars += len(text)
        mism += int(text != gpu_texts[u])
        if time.perf_counter() - t0 > args.cpu_seconds:
            n = u + 1
            break
    dt = time.perf_counter() - t0
    return {"value": chars / dt, "unit": "chars/s", "cores": 1, "kind": "port",
            "sample": f"first {n} utterances of the same workload ({chars} chars, {dt:.1f} s), Python "
                      f"restatement of the reference decoder, torch.set_num_threads(1)",
            "transcript_mismatches_vs_gpu": mism}


def _cpu_shard(job):
    """Worker of the all-cores CPU figure: decodes its own utterances until the time is up."""
    (words, corpus, targets, u_first, seed, utt0, margin, beam, tmax, order, variant, beta, gamma, kappa, seconds) = job
    from oracle import ns_oracle as O, synth
    torch.set_num_threads(1)
    trie = O.build_trie(words)
    lm = O.NgramLM(order, floor_vsz=(variant == "U")).fit(corpus)
    chars = done = 0
    t0 = time.perf_counter()
    for k, target in enumerate(targets):
        useed = synth.utt_seed(seed, utt0 + u_first + k)
        tgt = synth.text_to_tokens(target)
        step_fn = lambda ids: torch.from_numpy(synth.logits_row(useed, ids, tgt, margin)).unsqueeze(0)
        ids = O.constrained_beam_decode(step_fn, O.TextTransformLike, beam, tmax, trie, lm, lm_weight=beta,
                                        oov_penalty=gamma, variant=variant, word_bonus=kappa)
        chars += len(O.ids_to_text(O.TextTransformLike, ids))
        done += 1
        if time.perf_counter() - t0 > seconds:
            break
    return chars, done, time.perf_counter() - t0


def cpu_all_cores(words, corpus, targets, args):
    """The same Python restatement with the utterances sharded over worker processes, one per core of this
    job's CPU share: the figure a user of the reference gets by running one copy per core."""
    import multiprocessing as mp
    cores = max(1, min(args.cpu_procs, os.cpu_count() or 1))
    per = max(1, min(len(targets) // cores, args.cpu_utts))
    jobs = [(words, corpus, targets[w * per:(w + 1) * per], w * per, args.seed, args.utt0, args.margin, args.beam,
             args.tmax, args.order, args.variant, args.beta, args.gamma, args.kappa, args.cpu_seconds_all)
            for w in range(cores)]
    from concurrent.futures import ProcessPoolExecutor
    # a worker that dies breaks the pool with an exception instead of hanging the bench
    with ProcessPoolExecutor(cores, mp_context=mp.get_context("spawn")) as pool:
        out = list(pool.map(_cpu_shard, jobs, timeout=args.cpu_seconds_all * 4 + 120))
    chars = sum(o[0] for o in out)
    wall = max(o[2] for o in out)
    return {"value": chars / wall, "unit": "chars/s", "cores": cores,
            "sample": f"{sum(o[1] for o in out)} utterances over {cores} processes ({chars} chars, {wall:.1f} s "
                      f"after start-up)"}


def secondary_variant(args, variant, words, corpus, tgd, tld, targets):
    """Same batch through another copy of the decoder: chars/s and the nsd_step roofline, briefly."""
    import copy
    from silent_speech_amd.decoder import DeviceTables, NSDecoder, ids_to_text
    from silent_speech_amd import TextTransform
    from silent_speech_amd.tables import NgramCounts
    a2 = copy.copy(args)
    a2.variant = variant
    lm = NgramCounts(args.order, floor_vsz=(variant == "U")).fit(corpus)
    dec = NSDecoder(DeviceTables.from_words(words, lm), args.beam, args.tmax, variant=variant, lm_weight=args.beta,
                    oov_penalty=args.gamma, word_bonus=args.kappa)
    ws = dec.workspace(args.utts)
    sm = np.zeros(max(args.tmax, 1), np.float32)
    durs = []
    reps = max(3, args.steps // 4)
    for _ in range(2):
        dec.decode_synth_native(tgd, tld, base_seed=args.seed, margin=args.margin, utt0=args.utt0, ws=ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        timed = (k % args.event_every == 0)
        dec.decode_synth_native(tgd, tld, base_seed=args.seed, margin=args.margin, utt0=args.utt0, ws=ws,
                                step_ms=sm if timed else None)
        if timed:
            durs.append(sm[:ws.steps_run].copy())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ids_np, lens_np = ws.out_ids[:args.utts].cpu().numpy(), ws.out_len[:args.utts].cpu().numpy()
    texts = [ids_to_text(TextTransform, ids_np[u, :lens_np[u]].tolist()) for u in range(args.utts)]
    chars = sum(len(t) for t in texts)
    T, live, cands, per_step = algorithmic_bytes(dec, tgd, tld, a2)
    steps_run = ws.steps_run
    bytes_per_launch = float(per_step[:steps_run].sum()) / max(steps_run, 1)
    d = np.concatenate(durs).astype(np.float64) * 1e-3
    ach = bytes_per_launch / float(d.mean()) / 1e9
    return {"source": VARIANT_SRC[variant], "value": chars * reps / dt, "unit": "chars/s", "decodes": reps,
            "ms_per_decode": 1e3 * dt / reps, "chars_per_decode": chars,
            "targets_recovered_exactly": sum(int(a == b) for a, b in zip(texts, targets)),
            "roofline": {"achieved": ach, "frac": ach / HBM_PEAK_GBPS, "unit": "GB/s",
                         "avg_launch_us": float(d.mean()) * 1e6, "algorithmic_bytes_per_launch": bytes_per_launch}}


def default_workload(a) -> bool:
    return (a.utts == 4096 and a.beam == 16 and a.tmax == 64 and a.variant == "I" and a.order == 5
            and a.lex_words == 200 and not a.open_lexicon and a.margin == 60.0 and a.beta == 0.05)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=4096, help="utterances per GPU")
    ap.add_argument("--beam", type=int, default=16)
    ap.add_argument("--tmax", type=int, default=64)
    ap.add_argument("--variant", default="I", choices=["U", "I", "E"])
    ap.add_argument("--also", default="U", help="second variant measured briefly in the same run ('' = none)")
    ap.add_argument("--order", type=int, default=5)
    ap.add_argument("--beta", type=float, default=0.05)
    ap.add_argument("--gamma", type=float, default=1.0)
    ap.add_argument("--kappa", type=float, default=0.35)
    ap.add_argument("--margin", type=float, default=60.0)
    ap.add_argument("--lex-words", type=int, default=200)
    ap.add_argument("--open-lexicon", action="store_true", help="100k-word a-z0-9 lexicon (configs[2])")
    ap.add_argument("--corpus", type=int, default=20000)
    ap.add_argument("--seed", type=int, default=20260403)
    ap.add_argument("--check-every", type=int, default=0,
                    help="poll the done counter every this many steps (0 = enqueue all T_max steps)")
    ap.add_argument("--event-every", type=int, default=4, help="time the nsd_step launches of every n-th decode")
    ap.add_argument("--cpu-utts", type=int, default=512)
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    ap.add_argument("--cpu-seconds-all", type=float, default=8.0,
                    help="time limit of the all-cores CPU figure (0 = skip)")
    ap.add_argument("--cpu-procs", type=int, default=16, help="worker processes of the all-cores CPU figure")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) on a multi-GPU node; gloo to rehearse "
                                                             "several ranks on one GPU")
    ap.add_argument("--no-events", action="store_true", help="skip per-launch HIP events (pure throughput run)")
    args = ap.parse_args()

    if os.environ.get("NSD_BENCH_WATCHDOG"):        # diagnostic: periodic stack dumps of a run that seems stuck
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["NSD_BENCH_WATCHDOG"]), repeat=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    ndev = torch.cuda.device_count()
    if local >= ndev and args.dist_backend == "nccl":
        raise SystemExit(f"LOCAL_RANK {local} but only {ndev} GPU(s) visible")
    local_dev = local % max(ndev, 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    import torch.distributed as dist
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    from silent_speech_amd.dist import gather_results, last_gathered, unpack_gathered
    from silent_speech_amd import lib as nsd_lib
    from silent_speech_amd.tables import NgramCounts

    if args.open_lexicon and args.lex_words == 200:
        args.lex_words = 100000
    args.utt0 = rank * args.utts
    words, corpus, targets, tg, tl = build_workload(args.utts, args.utt0, args.lex_words, args.corpus,
                                                    args.order, args.open_lexicon)
    lm = NgramCounts(args.order, floor_vsz=(args.variant == "U")).fit(corpus)
    tables = DeviceTables.from_words(words, lm)
    dec = NSDecoder(tables, args.beam, args.tmax, variant=args.variant, lm_weight=args.beta,
                    oov_penalty=args.gamma, word_bonus=args.kappa)
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
    ws = dec.workspace(args.utts)
    n_total = args.utts * world

    use_events = not args.no_events
    step_ms = np.zeros(max(args.tmax, 1), np.float32)
    durs = []                                   # per-launch nsd_step durations (ms) from HIP events

    def one_decode(timed: bool, k: int = 0):
        # the whole step loop is enqueued by native code; on every `event_every`-th timed decode each
        # nsd_step launch is bracketed by HIP events on the launch stream
        sm = step_ms if (timed and use_events and k % args.event_every == 0) else None
        ids, lens, _ = dec.decode_synth_native(tgd, tld, base_seed=args.seed, margin=args.margin, utt0=args.utt0,
                                               check_every=args.check_every, ws=ws, step_ms=sm)
        if sm is not None:
            durs.append(sm[:ws.steps_run].copy())
        # the exchange runs on every decode; rank 0 turns the receive buffer into (ids, lens) once, after the loop
        gather_results(ids, lens, n_total, dst=0, unpack=False)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_decode(False)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_decode(True, k)
    fence()
    dt = time.perf_counter() - t0
    if world > 1 and rank == 0:
        # the gathered block of rank 0 must be its own result
        per, width = args.utts, ws.out_ids.shape[1]
        g_ids, g_len = unpack_gathered(last_gathered(per, width, world, dev), n_total, width)
        if not (torch.equal(g_ids[:args.utts], ws.out_ids[:args.utts])
                and torch.equal(g_len[:args.utts], ws.out_len[:args.utts].to(g_len.dtype))):
            raise RuntimeError("gathered results differ from rank 0's own decode")
    rdev = dev if args.dist_backend == "nccl" else torch.device("cpu")     # gloo reduces host tensors
    tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # characters decoded by this rank (identical every repetition: the decode is deterministic)
    dec.raise_on_topk_error(ws)
    ids_np, lens_np = ws.out_ids[:args.utts].cpu().numpy(), ws.out_len[:args.utts].cpu().numpy()
    from silent_speech_amd.decoder import ids_to_text
    from silent_speech_amd import TextTransform
    texts = [ids_to_text(TextTransform, ids_np[u, :lens_np[u]].tolist()) for u in range(args.utts)]
    chars = torch.tensor([sum(len(s) for s in texts)], dtype=torch.float64, device=rdev)
    exact = torch.tensor([sum(int(a == b) for a, b in zip(texts, targets))], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(chars)
        dist.all_reduce(exact)
    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    result = {
        "metric": "NS-decoded chars/sec (beam 16, trie+5-gram)" if (args.beam == 16 and args.order == 5)
                  else f"NS-decoded chars/sec (beam {args.beam}, trie+{args.order}-gram)",
        "value": float(chars.item()) * args.steps / dt,
        "unit": "chars/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": ("configs[2]: open 100k-word lexicon" if args.open_lexicon else
                                "configs[1]: closed 200-word template lexicon") +
                               f", {args.utts} utterances per GPU, beam {args.beam}, char {args.order}-gram, "
                               f"variant {args.variant} ({VARIANT_SRC[args.variant]}), beta {args.beta}, gamma {args.gamma}, "
                               f"margin {args.margin}, T_max {args.tmax}",
                   "utterances_total": n_total, "parallelism": f"utterance-sharded x{world}, replicated tables",
                   "chars_per_decode": float(chars.item()),
                   "targets_recovered_exactly": int(exact.item()),
                   "scorer_launches_per_decode": ws.steps_run},
    }

    # roofline of the dominant kernel (nsd_step): algorithmic bytes / measured launch duration
    T, live, cands, per_step = algorithmic_bytes(dec, tgd, tld, args)
    if len(per_step) < ws.steps_run:            # launches after every utterance finished move no bytes
        pad = ws.steps_run - len(per_step)
        per_step = np.concatenate([per_step, np.zeros(pad, np.int64)])
        live = np.concatenate([live, np.zeros(pad, np.int64)])
        cands = np.concatenate([cands, np.zeros(pad, np.int64)])
    bytes_per_launch = float(per_step[:ws.steps_run].sum()) / max(ws.steps_run, 1)
    if use_events and durs:
        d = np.concatenate(durs).astype(np.float64) * 1e-3                 # seconds
        avg = float(d.mean())
        achieved = bytes_per_launch / avg / 1e9
        result["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                              "kernel": "nsd_step_kernel", "launches_timed": int(d.size),
                              "avg_launch_us": avg * 1e6, "algorithmic_bytes_per_launch": bytes_per_launch,
                              "mean_live_hyps_per_launch": float(live.sum() / max(ws.steps_run, 1)),
                              "mean_children_per_live_hyp": float(cands.sum() / max(live.sum(), 1)),
                              "busiest_launch": {"us": float(d.reshape(len(durs), -1).mean(0).max() * 1e6),
                                                 "GBps": float((per_step[:ws.steps_run] /
                                                                np.maximum(d.reshape(len(durs), -1).mean(0)[:len(per_step[:ws.steps_run])], 1e-9)).max() / 1e9)}}
    if "roofline" in result and default_workload(args):
        # HBM bytes per nsd_step launch from separate rocprofv3 --pmc passes of this same command
        # (profiles/r01_pmc_traffic.json, written by scripts/collect_traffic.py): 2 x FETCH_SIZE + WRITE_SIZE
        try:
            tr = json.loads((ROOT / "profiles" / "r01_pmc_traffic.json").read_text())
            result["roofline"]["traffic"] = tr["traffic_bytes_per_launch"]
            result["roofline"]["traffic_source"] = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
        except Exception:
            pass
    if args.also and args.also != args.variant and world == 1:
        result["variant_" + args.also] = secondary_variant(args, args.also, words, corpus, tgd, tld, targets)
    if not args.no_cpu and world == 1:
        result["cpu_baseline"] = cpu_baseline(words, corpus, targets, args, texts)
        result["cpu_baseline"]["host_cpus"] = os.cpu_count()
        if args.cpu_seconds_all > 0:
            result["cpu_baseline"]["all_cores"] = cpu_all_cores(words, corpus, targets, args)
        result["speedup_vs_cpu_1core"] = result["value"] / result["cpu_baseline"]["value"]
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
