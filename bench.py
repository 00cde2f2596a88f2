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
step) is measured in the same run and reported under "variant_U".

Several GPUs: utterances are independent, so rank r decodes its own contiguous
block with a replica of the tables and rank 0 gathers the ids (one collective
per decode).  --gpus N without WORLD_SIZE in the environment starts the N
ranks itself (fresh child processes; the parent touches no GPU); under
torchrun the ranks are taken from the environment.  Default: every rank
decodes --utts utterances (weak scaling).  --utts-total M splits M utterances
of the global numbering over the ranks (strong scaling, the 1M-utterance
configuration): rank r decodes shard_range(M, r, N).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--utts-total M]

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# numpy and torch are imported by worker(): the launching parent of a multi-GPU run needs neither
np = None
torch = None

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak (spec)
VARIANT_SRC = {"U": "scripts/ns_utils.py:194", "I": "inference_emg_llm.py:222", "E": "eval_ar.py:214"}


def build_workload(n_utts: int, utt0: int, lex_words: int, corpus_n: int, order: int, open_lexicon: bool):
    """Seeded lexicon / corpus / planted targets of the utterances [utt0, utt0 + n_utts) of the global numbering."""
    from silent_speech_amd import synthetic as synth
    if open_lexicon:
        words = synth.make_lexicon(lex_words, 1, synth.BASE_CHARS[:36], 2, 12)
    else:
        words = synth.make_lexicon(lex_words, 0, synth.BASE_CHARS[:26], 2, 8)
    corpus = synth.make_sentences(words, corpus_n, 1)
    targets = synth.targets_for_range(words, utt0, utt0 + n_utts)
    tg, tl = synth.pack_targets(targets)
    return words, corpus, targets, tg, tl


def decoded_text_stats(ids_np, lens_np, tg, tl):
    """Per utterance, from the decoded ids (BOS first): number of characters of ids_to_text(ids)
    (scripts/ns_utils.py:284-296: skip BOS, stop at EOS, keep base characters, strip), whether that text equals
    the planted target and whether it starts with it.  Vectorised (a 1M-utterance shard is one call)."""
    n, W = ids_np.shape
    col = np.arange(W)[None, :]
    valid = (col >= 1) & (col < lens_np[:, None])
    eos = valid & (ids_np == 38)
    first_eos = np.where(eos.any(1), eos.argmax(1), W)
    keep = valid & (col < first_eos[:, None]) & (ids_np < 37)
    nonsp = keep & (ids_np != 36)
    any_ns = nonsp.any(1)
    first = np.where(any_ns, nonsp.argmax(1), W)
    last = np.where(any_ns, W - 1 - nonsp[:, ::-1].argmax(1), -1)
    text = keep & (col >= first[:, None]) & (col <= last[:, None])
    nchar = text.sum(1)
    # position of every text character inside its text, compared with the target at that position
    pos = np.cumsum(text, 1) - 1
    S = tg.shape[1]
    tpos = np.clip(pos, 0, S - 1)
    tchar = np.take_along_axis(tg.astype(np.int16), tpos, 1)
    in_target = pos < tl[:, None]
    mism = text & in_target & (tchar != ids_np)
    starts = (~mism.any(1)) & (nchar >= tl)
    exact = starts & (nchar == tl)
    return nchar.astype(np.int64), exact, starts


def popcount64_t(a):
    """Set bits of every int64 element (device tensor)."""
    m1, m2, m4 = 0x5555555555555555, 0x3333333333333333, 0x0F0F0F0F0F0F0F0F
    a = a - ((a >> 1) & m1)
    a = (a & m2) + ((a >> 2) & m2)
    a = (a + (a >> 4)) & m4
    return ((a * 0x0101010101010101) >> 56) & 0xFF


def algorithmic_bytes(dec, tgd, tld, args):
    """Untimed instrumented decode: per executed step, live hypotheses and lexicon-allowed children.

    Per live hypothesis the scorer must move: its 40 fp32 logits (160 B), its state in and out, its trie
    node record, back-pointer and token (245 B in all), and two table slots (interior + terminal n-gram)
    for each allowed child and for the hypothesis itself: 245 + 32 * (c + 1) bytes.
    The step loop is driven from here, one launch at a time; the lexicon masks of each step go to one reused
    buffer and are counted on the device (a 1M-utterance shard needs no [T, N*B] mask array).
    """
    ws = dec.workspace(tgd.shape[0], with_masks=False)
    masks = torch.zeros(max(ws.N * ws.B, 1), dtype=torch.int64, device=ws.score.device)
    dec.init(ws)
    live, cands = [], []
    low37 = (1 << 37) - 1
    for t in range(dec.T):
        masks.zero_()
        dec.synth_logits(ws, args.seed, args.utt0, tgd, tld, args.margin)
        um = ws.utt[:ws.N].to(torch.int64)
        meta = ws.meta[:ws.N * ws.B].view(ws.N, ws.B)
        nb = (um & 0xFF).view(-1, 1)
        is_live = ((torch.arange(ws.B, device=meta.device).view(1, -1) < nb) & ((meta & 1) == 0)
                   & ((um & 0x100) == 0).view(-1, 1))
        n_live = int(is_live.sum().item())
        dec.step(ws, mask_out=masks)
        live.append(n_live)
        cands.append(int(popcount64_t(masks & low37).sum().item()))
        if dec.all_done(ws):
            break
    T = len(live)
    live, cands = np.array(live, np.int64), np.array(cands, np.int64)
    per_step = live * 245 + 32 * (cands + live)
    return T, live, cands, per_step


def cpu_baseline(words, corpus, targets, args, gpu_texts):
    """The reference algorithm on one host core: the Python restatement (oracle/ns_oracle.py: dict trie,
    whole-string n-gram score per candidate, one step_fn call and one torch log_softmax per live
    hypothesis, stable sort) on a bounded sample of the same utterances."""
    import torch
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
    import torch
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


def timed_launches(dec, ws, tgd, tld, args, reps):
    """nsd_step launch durations (seconds, one row per decode) from HIP events on the launch stream, taken on
    decodes of their own, outside any throughput window."""
    sm = np.zeros(max(args.tmax, 1), np.float32)
    rows = []
    # one untimed decode first: the host-side bookkeeping before this call leaves the GPU idle long enough for its
    # clocks to drop, and the first decode after such a pause runs 5-8 % slower per launch than the back-to-back
    # decodes of the throughput loop, which are what the launch time is meant to describe
    dec.decode_synth_native(tgd, tld, base_seed=args.seed, margin=args.margin, utt0=args.utt0,
                            check_every=args.check_every, ws=ws)
    for _ in range(reps):
        dec.decode_synth_native(tgd, tld, base_seed=args.seed, margin=args.margin, utt0=args.utt0,
                                check_every=args.check_every, ws=ws, step_ms=sm)
        rows.append(sm[:ws.steps_run].astype(np.float64) * 1e-3)
    n = min(len(r) for r in rows)
    return np.stack([r[:n] for r in rows])


def roofline_of(per_step, live, cands, d, steps_run):
    """HBM roofline of nsd_step: algorithmic bytes per launch over the mean event duration of a launch."""
    if len(per_step) < steps_run:               # launches after every utterance finished move no bytes
        pad = steps_run - len(per_step)
        per_step = np.concatenate([per_step, np.zeros(pad, np.int64)])
        live = np.concatenate([live, np.zeros(pad, np.int64)])
        cands = np.concatenate([cands, np.zeros(pad, np.int64)])
    per_step = per_step[:steps_run]
    bytes_per_launch = float(per_step.sum()) / max(steps_run, 1)
    avg = float(d.mean())
    achieved = bytes_per_launch / avg / 1e9
    per_launch = d.mean(0)
    k = min(len(per_launch), len(per_step))
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel": "nsd_step_kernel",
            "launches_timed": int(d.size), "avg_launch_us": avg * 1e6,
            "avg_launch_us_per_timed_decode": [float(x) * 1e6 for x in d.mean(1)],
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "mean_live_hyps_per_launch": float(live[:steps_run].sum() / max(steps_run, 1)),
            "mean_children_per_live_hyp": float(cands[:steps_run].sum() / max(live[:steps_run].sum(), 1)),
            "busiest_launch": {"us": float(per_launch.max() * 1e6),
                               "GBps": float((per_step[:k] / np.maximum(per_launch[:k], 1e-9)).max() / 1e9)}}


def secondary_variant(args, variant, words, corpus, tgd, tld, tg, tl):
    """Same batch through another copy of the decoder: chars/s and the nsd_step roofline, briefly."""
    import copy
    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    from silent_speech_amd.tables import NgramCounts
    a2 = copy.copy(args)
    a2.variant = variant
    lm = NgramCounts(args.order, floor_vsz=(variant == "U")).fit(corpus)
    dec = NSDecoder(DeviceTables.from_words(words, lm), args.beam, args.tmax, variant=variant, lm_weight=args.beta,
                    oov_penalty=args.gamma, word_bonus=args.kappa)
    ws = dec.workspace(args.utts)
    reps = max(3, args.steps // 4)
    for _ in range(2):
        dec.decode_synth_native(tgd, tld, base_seed=args.seed, margin=args.margin, utt0=args.utt0, ws=ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dec.decode_synth_native(tgd, tld, base_seed=args.seed, margin=args.margin, utt0=args.utt0, ws=ws)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nchar, exact, starts = decoded_text_stats(ws.out_ids[:args.utts].cpu().numpy(),
                                              ws.out_len[:args.utts].cpu().numpy(), tg, tl)
    chars = int(nchar.sum())
    steps_run = ws.steps_run
    d = timed_launches(dec, ws, tgd, tld, a2, 2)
    T, live, cands, per_step = algorithmic_bytes(dec, tgd, tld, a2)
    r = roofline_of(per_step, live, cands, d, steps_run)
    return {"source": VARIANT_SRC[variant], "value": chars * reps / dt, "unit": "chars/s", "decodes": reps,
            "ms_per_decode": 1e3 * dt / reps, "chars_per_decode": chars,
            "targets_recovered_exactly": int(exact.sum()),
            "target_chars_per_s": float(tl[starts].sum()) * reps / dt,
            "roofline": {k: r[k] for k in ("achieved", "frac", "unit", "avg_launch_us",
                                           "algorithmic_bytes_per_launch")}}


def cpu_ctc_rerank(words, corpus, targets, ctc_np, args, gpu_final, gpu_device_choice):
    """inference_emg_llm.py:432-464 for one utterance at a time on one host core: the Python restatement of the
    beam search, the frame argmax + collapse, postprocess of both transcripts and the LM choice; on a bounded
    sample of the same utterances.  Also recomputes the choice without postprocess, which is what nsd_rerank_lm
    takes on the device."""
    import torch
    from oracle import ns_oracle as O, synth
    torch.set_num_threads(1)
    TT = O.TextTransformLike
    trie = O.build_trie(words)
    lm = O.NgramLM(args.order, floor_vsz=(args.variant == "U")).fit(corpus)
    n = min(args.cpu_utts, len(targets))
    chars = mism = mism_dev = took_ctc = 0
    t0 = time.perf_counter()
    for u in range(n):
        useed = synth.utt_seed(args.seed, args.utt0 + u)
        tgt = synth.text_to_tokens(targets[u])
        step_fn = lambda ids: torch.from_numpy(synth.logits_row(useed, ids, tgt, args.margin)).unsqueeze(0)
        ids = O.constrained_beam_decode(step_fn, TT, args.beam, args.tmax, trie, lm, lm_weight=args.beta,
                                        oov_penalty=args.gamma, variant=args.variant, word_bonus=args.kappa)
        raw_ar = O.ids_to_text(TT, ids)
        raw_ctc = O.ctc_hypothesis(ctc_np[u], TT)
        hyp_ar = O.postprocess(raw_ar, trie, words)
        hyp_ctc = O.postprocess(raw_ctc, trie, words)
        hyp = O.rerank_lm(hyp_ar, hyp_ctc, lm)
        chars += len(hyp)
        took_ctc += int(hyp is hyp_ctc and hyp_ctc != hyp_ar)
        mism += int(hyp != gpu_final[u])
        mism_dev += int(O.rerank_lm(raw_ar, raw_ctc, lm) != gpu_device_choice[u])
        if time.perf_counter() - t0 > args.cpu_seconds:
            n = u + 1
            break
    dt = time.perf_counter() - t0
    return {"value": chars / dt, "unit": "chars/s", "cores": 1, "kind": "port",
            "sample": f"first {n} utterances of the same workload ({chars} final characters, {dt:.1f} s): Python "
                      f"restatement of inference_emg_llm.py:432-464 (beam search, CTC argmax + collapse, postprocess, "
                      f"LM choice), torch.set_num_threads(1)",
            "ctc_transcript_chosen": took_ctc,
            "final_transcript_mismatches_vs_gpu": mism,
            "device_choice_mismatches_vs_restatement_without_postprocess": mism_dev}


def ctc_rerank_workload(args, dev):
    """The CTC proposal + rerank configuration on one GPU.  One step = AR decode of the batch (T_max scorer launches fed
    by the synthetic AR head, back-trace), nsd_ctc_greedy over the resident (N, frames, 40) CTC logits and
    nsd_rerank_lm, which scores both transcripts of every utterance with the n-gram tables and picks one
    (inference_emg_llm.py:441-464 without the host-side postprocess).  The driver's complete tail, postprocess
    included, is run once through transcribe_batch (host string work, timed separately) and compared with the CPU
    restatement on the sampled utterances."""
    from silent_speech_amd import TextTransform, synthetic as synth
    from silent_speech_amd.decoder import (DeviceTables, NSDecoder, ctc_greedy_batch, ids_to_text, rerank_lm_batch,
                                           transcribe_batch)
    from silent_speech_amd.tables import NgramCounts
    args.utt0 = 0
    words, corpus, targets, tg, tl = build_workload(args.utts, 0, args.lex_words, args.corpus, args.order, False)
    lm = NgramCounts(args.order, floor_vsz=(args.variant == "U")).fit(corpus)
    tables = DeviceTables.from_words(words, lm)
    dec = NSDecoder(tables, args.beam, args.tmax, variant=args.variant, lm_weight=args.beta, oov_penalty=args.gamma,
                    word_bonus=args.kappa, lanes_per_utt=args.lanes)
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
    ctc_np = synth.ctc_frame_logits(tg, tl, 0, args.ctc_frames, margin=args.ctc_margin)
    ctc = torch.from_numpy(ctc_np).to(dev)
    ws = dec.workspace(args.utts)
    N, Tp = args.utts, args.ctc_frames

    def one_step():
        ids, lens, _ = dec.decode_synth_native(tgd, tld, base_seed=args.seed, margin=args.margin, utt0=0, ws=ws)
        ci, cl = ctc_greedy_batch(ctc)
        choice, scores = rerank_lm_batch(tables, ids, lens, ci, cl)
        return ids, lens, ci, cl, choice

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ids, lens, ci, cl, choice = one_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dec.raise_on_topk_error(ws)

    ids_np, lens_np = ids.cpu().numpy(), lens.cpu().numpy()
    ci_np, cl_np, ch_np = ci.cpu().numpy(), cl.cpu().numpy(), choice.cpu().numpy()
    TT = TextTransform
    picked = []
    for u in range(N):
        if ch_np[u]:
            picked.append(ids_to_text(TT, [TT.BOS_IDX] + ci_np[u, :cl_np[u]].tolist() + [TT.EOS_IDX]))
        else:
            picked.append(ids_to_text(TT, ids_np[u, :lens_np[u]].tolist()))
    chars = sum(len(s) for s in picked)
    exact = sum(int(s == t) for s, t in zip(picked, targets))

    # kernel durations: HIP events on the launch stream (torch's current stream is the one the library launches on)
    def timed(fn, reps=20):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in ev])) * 1e-3
    t_ctc = timed(lambda: ctc_greedy_batch(ctc))
    t_rr = timed(lambda: rerank_lm_batch(tables, ids, lens, ci, cl))
    ctc_bytes = float(N) * Tp * 160 + float(cl_np.sum()) + 4.0 * N
    d = timed_launches(dec, ws, tgd, tld, args, args.timed_decodes)
    T, live, cands, per_step = algorithmic_bytes(dec, tgd, tld, args)
    roof = roofline_of(per_step, live, cands, d, ws.steps_run)
    roof["traffic_note"] = "see profiles/ for the rocprofv3 passes of this command"

    # the driver's complete tail once, postprocess (host, difflib) included
    lex = set(words)
    t1 = time.perf_counter()
    final = transcribe_batch(TT, ids, lens, tables, lex.__contains__, words, ctc_logits=ctc)
    t_tail = time.perf_counter() - t1

    result = {
        "metric": f"NS-decoded chars/sec (beam {args.beam}, trie+{args.order}-gram), CTC proposal + LM rerank",
        "value": chars * args.steps / dt, "unit": "chars/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"configs[3]: CTC-head proposals + NS rerank, {N} utterances, beam {args.beam}, closed "
                               f"200-word template lexicon, char {args.order}-gram, variant {args.variant} "
                               f"({VARIANT_SRC[args.variant]}) AR decode of T_max {args.tmax} + frame argmax / collapse "
                               f"over ({N}, {Tp}, 40) CTC logits (planted target held 1-3 frames per character, "
                               f"blanks, margin {args.ctc_margin} against noise in [-2, 2)) + LM choice between the "
                               f"two transcripts (inference_emg_llm.py:441-464)",
                   "parallelism": "one GPU",
                   "chars_per_step": chars, "ctc_transcript_chosen": int(ch_np.sum()),
                   "chosen_transcript_equals_target": exact,
                   "value_counts": "characters of the transcript nsd_rerank_lm picks (ids_to_text of the AR ids or "
                                   "of the collapsed CTC ids, no postprocess); the copy of the decoder without an "
                                   "EOS option decodes all T_max steps, so an AR transcript runs past its target"},
        "roofline": roof,
        "ctc_greedy": {"kernel": "nsd_ctc_greedy_kernel", "bound": "hbm", "avg_launch_us": t_ctc * 1e6,
                       "algorithmic_bytes_per_launch": ctc_bytes, "achieved": ctc_bytes / t_ctc / 1e9,
                       "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ctc_bytes / t_ctc / 1e9 / HBM_PEAK_GBPS,
                       "note": f"{ctc_bytes / 1e6:.0f} MB of logits per launch: below the 256 MB Infinity Cache, so "
                               f"repeated launches over the same array may be served from it (events recorded around "
                               f"the call include about 2 us of launch overhead)"},
        "rerank_lm": {"kernel": "nsd_rerank_lm_kernel", "avg_launch_us": t_rr * 1e6},
        "share_of_step": {"ar_decode_ms": 1e3 * dt / args.steps - 1e3 * (t_ctc + t_rr), "ctc_greedy_ms": 1e3 * t_ctc,
                          "rerank_lm_ms": 1e3 * t_rr},
        "final_transcripts": {"seconds": t_tail, "chars": sum(len(s) for s in final),
                              "note": "transcribe_batch once: CTC argmax on the GPU, ids_to_text, postprocess (difflib "
                                      "snap of out-of-lexicon words) and LM choice on the host, one core"},
    }
    if not args.no_cpu:
        result["cpu_baseline"] = cpu_ctc_rerank(words, corpus, targets, ctc_np, args, final, picked)
        result["cpu_baseline"]["host_cpus"] = os.cpu_count()
        result["speedup_vs_cpu_1core"] = result["value"] / result["cpu_baseline"]["value"]
    return result


def shards_in_flight(args, dec, tgd, tld, ids_ref, lens_ref, chars, target_chars, dev, shards=2):
    """The same utterances decoded as `shards` halves on streams of their own (each shard's launches enqueued by
    nsd_decode_synth on its stream): a launch lasts as long as its slowest wave, and the other shard's launches
    fill that tail.  Reported beside `value`, which stays the one-stream decode whose launch the roofline describes."""
    import torch
    n = tgd.shape[0]
    bounds = [(n * k // shards, n * (k + 1) // shards) for k in range(shards)]
    wss = [dec.workspace(hi - lo) for lo, hi in bounds]
    streams = [torch.cuda.Stream(device=dev) for _ in bounds]

    def once():
        for (lo, hi), w, st in zip(bounds, wss, streams):
            with torch.cuda.stream(st):
                dec.decode_synth_native(tgd[lo:hi], tld[lo:hi], base_seed=args.seed, margin=args.margin,
                                        utt0=args.utt0 + lo, check_every=args.check_every, ws=w)

    torch.cuda.synchronize()
    for _ in range(max(args.warmup, 1)):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        once()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ids = torch.cat([w.out_ids[:hi - lo] for (lo, hi), w in zip(bounds, wss)])
    lens = torch.cat([w.out_len[:hi - lo] for (lo, hi), w in zip(bounds, wss)])
    same = bool(torch.equal(ids, ids_ref) and torch.equal(lens, lens_ref))
    return {"shards": shards, "value": chars * args.steps / dt, "unit": "chars/s",
            "target_chars_per_s": target_chars * args.steps / dt, "ms_per_step": 1e3 * dt / args.steps,
            "equal_to_the_one_stream_decode": same,
            "note": "same utterances, same kernels, two half-batches on two streams of one GPU"}


def large_batch(args, words, corpus, n_utts, dev):
    """The same decode at a batch that fills the GPU many times over (one GPU's share of the 1M-utterance
    configuration): chars/s and the nsd_step roofline, with the fields of the main line."""
    import copy
    from silent_speech_amd import synthetic as synth
    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    from silent_speech_amd.tables import NgramCounts
    a2 = copy.copy(args)
    a2.utts = n_utts
    tg, tl = synth.pack_targets(synth.targets_for_range(words, a2.utt0, a2.utt0 + n_utts))
    lm = NgramCounts(args.order, floor_vsz=(args.variant == "U")).fit(corpus)
    dec = NSDecoder(DeviceTables.from_words(words, lm), args.beam, args.tmax, variant=args.variant,
                    lm_weight=args.beta, oov_penalty=args.gamma, word_bonus=args.kappa, lanes_per_utt=args.lanes)
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
    ws = dec.workspace(n_utts)
    reps = 3
    dec.decode_synth_native(tgd, tld, base_seed=args.seed, margin=args.margin, utt0=a2.utt0, ws=ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dec.decode_synth_native(tgd, tld, base_seed=args.seed, margin=args.margin, utt0=a2.utt0, ws=ws)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dec.raise_on_topk_error(ws)
    nchar, exact, starts = decoded_text_stats(ws.out_ids[:n_utts].cpu().numpy(), ws.out_len[:n_utts].cpu().numpy(),
                                              tg, tl)
    steps_run = ws.steps_run
    d = timed_launches(dec, ws, tgd, tld, a2, 2)
    T, live, cands, per_step = algorithmic_bytes(dec, tgd, tld, a2)
    r = roofline_of(per_step, live, cands, d, steps_run)
    r["traffic"], src = profiled_traffic(a2)
    if src:
        r["traffic_note"] = f"from the separate rocprofv3 --pmc passes kept in {src} (same build, not this run)"
    return {"workload": f"{n_utts} utterances on one GPU, otherwise as the main line (variant {args.variant}, beam "
                        f"{args.beam}, char {args.order}-gram, T_max {args.tmax})",
            "value": float(nchar.sum()) * reps / dt, "unit": "chars/s", "decodes": reps,
            "ms_per_decode": 1e3 * dt / reps, "chars_per_decode": int(nchar.sum()),
            "targets_recovered_as_prefix": int(starts.sum()),
            "target_chars_per_s": float(tl[starts].sum()) * reps / dt,
            "roofline": r}


def profiled_traffic(args):
    """HBM-side bytes per nsd_step launch for this workload, if separate rocprofv3 --pmc passes of this very command
    are kept under profiles/ (FETCH_SIZE and WRITE_SIZE in runs of their own, FETCH_SIZE doubled: the gfx950 correction
    for wide coalesced reads; scripts/pmc_traffic.sh, scripts/collect_traffic.py).  The counters cannot be read from
    inside this process, so the figure is the one those passes measured on the same build, not a live one."""
    if not (args.variant == "I" and args.beam == 16 and args.order == 5 and args.tmax == 64 and args.margin == 60.0
            and args.beta == 0.05 and args.gamma == 1.0 and args.lanes == 0 and args.lds_trie == 0) and \
            not (args.open_lexicon and args.beam == 32):
        return None, None
    if args.open_lexicon:
        name = "r03_pmc_traffic_open_lexicon_b32.json" if (args.beam == 32 and args.utts == 4096 and
                                                          args.lex_words == 100000 and args.variant == "I") else None
    else:
        name = {4096: "r03_pmc_traffic.json", 131072: "r03_pmc_traffic_131072.json"}.get(args.utts) \
            if args.lex_words == 200 else None
    f = ROOT / "profiles" / name if name else None
    if f is None or not f.exists():
        return None, None
    return float(json.loads(f.read_text())["traffic_bytes_per_launch"]), f"profiles/{name}"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=4096, help="utterances per GPU (weak scaling)")
    ap.add_argument("--utts-total", type=int, default=0,
                    help="split this many utterances over the ranks instead (strong scaling; the 1M-utterance "
                         "configuration is --utts-total 1000000)")
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
    ap.add_argument("--timed-decodes", type=int, default=4,
                    help="decodes after the throughput loop whose nsd_step launches carry HIP events")
    ap.add_argument("--cpu-utts", type=int, default=512)
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    ap.add_argument("--cpu-seconds-all", type=float, default=8.0,
                    help="time limit of the all-cores CPU figure (0 = skip)")
    ap.add_argument("--cpu-procs", type=int, default=16, help="worker processes of the all-cores CPU figure")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) on a multi-GPU node; gloo to rehearse "
                                                             "several ranks on one GPU")
    ap.add_argument("--no-events", action="store_true", help="skip per-launch HIP events (pure throughput run)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes of a wave per utterance in nsd_step (0 = library's "
                                                         "choice; 16, 32 or 64)")
    ap.add_argument("--lds-trie", type=int, default=0, help="node records of the upper trie levels every workgroup "
                                                            "stages in LDS (0 = none, the default)")
    ap.add_argument("--shards-in-flight", type=int, default=0,
                    help="also time the same decode as this many shards on streams of their own (one GPU; e.g. 2); "
                         "`value` stays the one-stream decode.  Off by default: the shards' launches run the same "
                         "kernel instantiation, so a kernel trace of the command would average them into its row")
    ap.add_argument("--large-utts", type=int, default=131072,
                    help="utterances of the large_batch block measured after the main line on one GPU (0 = skip)")
    ap.add_argument("--ctc-rerank", action="store_true",
                    help="the CTC proposal + rerank configuration: AR decode, nsd_ctc_greedy over (N, frames, 40) CTC "
                         "logits, nsd_rerank_lm (inference_emg_llm.py:441-464), one GPU")
    ap.add_argument("--ctc-frames", type=int, default=256)
    ap.add_argument("--ctc-margin", type=float, default=3.98)
    ap.add_argument("--launch-timeout", type=float, default=3000.0,
                    help="seconds after which the self-started ranks of --gpus N are ended")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: start the ranks ourselves, before anything here touches a GPU
        from silent_speech_amd.launch import spawn_ranks
        rc = spawn_ranks([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], args.gpus,
                         timeout=args.launch_timeout)
        raise SystemExit(rc)
    worker(args)


def worker(args):
    global np, torch
    import numpy as _np
    import torch as _torch
    np, torch = _np, _torch
    # only the JSON line may reach stdout: libraries that print there (gloo's connection notes) go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("NSD_BENCH_WATCHDOG"):        # diagnostic: periodic stack dumps of a run that seems stuck
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["NSD_BENCH_WATCHDOG"]), repeat=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
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
    # under a launcher (WORLD_SIZE set) the process group and the final gather run whatever the rank count, so that
    # a one-rank torchrun line exercises the same RCCL calls as an eight-rank one
    grouped = world > 1 or "WORLD_SIZE" in os.environ
    if grouped:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)
        if dist.get_world_size() != world:
            raise SystemExit(f"process group reports {dist.get_world_size()} ranks, expected {world}")

    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    from silent_speech_amd.dist import gather_results, last_gathered, shard_range, unpack_gathered
    from silent_speech_amd.tables import NgramCounts

    if args.ctc_rerank:
        if grouped:
            raise SystemExit("--ctc-rerank is a one-GPU workload")
        result = ctc_rerank_workload(args, dev)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
        return
    if args.open_lexicon and args.lex_words == 200:
        args.lex_words = 100000
    strong = args.utts_total > 0
    if strong:
        lo, hi = shard_range(args.utts_total, rank, world)
        args.utts, args.utt0, n_total = hi - lo, lo, args.utts_total
    else:
        args.utt0, n_total = rank * args.utts, args.utts * world
    words, corpus, targets, tg, tl = build_workload(args.utts, args.utt0, args.lex_words, args.corpus,
                                                    args.order, args.open_lexicon)
    lm = NgramCounts(args.order, floor_vsz=(args.variant == "U")).fit(corpus)
    tables = DeviceTables.from_words(words, lm)
    dec = NSDecoder(tables, args.beam, args.tmax, variant=args.variant, lm_weight=args.beta,
                    oov_penalty=args.gamma, word_bonus=args.kappa, lanes_per_utt=args.lanes,
                    lds_trie_nodes=args.lds_trie)
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
    ws = dec.workspace(args.utts)

    def one_decode():
        # the whole step loop is enqueued by native code
        ids, lens, _ = dec.decode_synth_native(tgd, tld, base_seed=args.seed, margin=args.margin, utt0=args.utt0,
                                               check_every=args.check_every, ws=ws)
        # the exchange runs on every decode; rank 0 turns the receive buffer into (ids, lens) once, after the loop
        gather_results(ids, lens, n_total, dst=0, unpack=False, force=grouped)

    def fence():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_decode()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_decode()
    fence()
    dt = time.perf_counter() - t0
    steps_run = ws.steps_run
    if grouped and rank == 0:
        # the gathered block of rank 0 must be its own result
        per, width = (n_total + world - 1) // world, ws.out_ids.shape[1]
        g_ids, g_len = unpack_gathered(last_gathered(per, width, world, dev), n_total, width)
        if g_ids.shape[0] != n_total:
            raise RuntimeError(f"gathered {g_ids.shape[0]} utterances, expected {n_total}")
        if not (torch.equal(g_ids[:args.utts], ws.out_ids[:args.utts])
                and torch.equal(g_len[:args.utts], ws.out_len[:args.utts].to(g_len.dtype))):
            raise RuntimeError("gathered results differ from rank 0's own decode")
    rdev = dev if args.dist_backend == "nccl" else torch.device("cpu")     # gloo reduces host tensors
    tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
    if grouped:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # characters decoded by this rank (identical every repetition: the decode is deterministic)
    dec.raise_on_topk_error(ws)
    ids_np, lens_np = ws.out_ids[:args.utts].cpu().numpy(), ws.out_len[:args.utts].cpu().numpy()
    nchar, exact, starts = decoded_text_stats(ids_np, lens_np, tg, tl)
    sums = torch.tensor([float(nchar.sum()), float(exact.sum()), float(starts.sum()), float(tl[starts].sum()),
                         float(args.utts)], dtype=torch.float64, device=rdev)
    per_rank = torch.zeros(world, dtype=torch.float64, device=rdev)
    per_rank[rank] = float(args.utts)
    if grouped:
        dist.all_reduce(sums)
        dist.all_reduce(per_rank)
    if rank != 0:
        if grouped:
            dist.barrier()
            dist.destroy_process_group()
        return
    chars, n_exact, n_starts, target_chars, n_decoded = (float(x) for x in sums.tolist())
    if int(n_decoded) != n_total:
        raise RuntimeError(f"ranks decoded {int(n_decoded)} utterances in all, expected {n_total}")

    shape = (f"{n_total} utterances split over {world} GPU(s)" if strong else f"{args.utts} utterances per GPU")
    result = {
        "metric": "NS-decoded chars/sec (beam 16, trie+5-gram)" if (args.beam == 16 and args.order == 5)
                  else f"NS-decoded chars/sec (beam {args.beam}, trie+{args.order}-gram)",
        "value": chars * args.steps / dt,
        "unit": "chars/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": ("configs[2]: open 100k-word lexicon (random-access stress: random words, most "
                                "n-grams unseen)" if args.open_lexicon else
                                ("configs[4]: " if strong and n_total >= 1000000 else "configs[1]: ") +
                                "closed 200-word template lexicon") +
                               f", {shape}, beam {args.beam}, char {args.order}-gram, "
                               f"variant {args.variant} ({VARIANT_SRC[args.variant]}), beta {args.beta}, gamma {args.gamma}, "
                               f"margin {args.margin}, T_max {args.tmax}" +
                               ("; this copy of the decoder has no EOS option and decodes all T_max steps whatever the "
                                "target's length, so `value` counts the characters after the target's end too "
                                f"({100.0 * (1.0 - target_chars / max(chars, 1.0)):.0f} % of them here): "
                                "target_chars_per_s is the transcript throughput" if args.variant == "I" else ""),
                   "utterances_total": n_total, "utterances_per_rank": [int(x) for x in per_rank.tolist()],
                   "ranks_reported_by_process_group": dist.get_world_size() if grouped else 1,
                   "process_group": (dist.get_backend() if grouped else None),
                   "parallelism": f"utterance-sharded x{world}, replicated tables",
                   "chars_per_decode": chars,
                   "targets_recovered_exactly": int(n_exact),
                   "targets_recovered_as_prefix": int(n_starts),
                   "target_chars_per_decode": target_chars,
                   "scorer_launches_per_decode": steps_run},
        # characters of planted targets that came back (the decoded text starts with the target; the copy of the
        # decoder without an EOS option keeps decoding flat logits after the target's end)
        "target_chars_per_s": target_chars * args.steps / dt,
    }

    # roofline of the dominant kernel (nsd_step): algorithmic bytes / launch duration from HIP events, measured
    # on decodes of their own after the throughput window
    if not args.no_events and args.timed_decodes > 0:
        d = timed_launches(dec, ws, tgd, tld, args, args.timed_decodes)
        T, live, cands, per_step = algorithmic_bytes(dec, tgd, tld, args)
        result["roofline"] = roofline_of(per_step, live, cands, d, steps_run)
        traffic, src = profiled_traffic(args)
        result["roofline"]["traffic"] = traffic
        result["roofline"]["traffic_note"] = (
            f"HBM-side bytes per launch from the separate rocprofv3 --pmc passes kept in {src} (FETCH_SIZE and "
            f"WRITE_SIZE in runs of their own on this command, FETCH_SIZE doubled as the gfx950 correction): measured "
            f"on this build, not in this run" if src else
            "HBM-side bytes need separate rocprofv3 --pmc passes and are not measured in this run; see profiles/ for "
            "the passes taken on the benchmark commands")
    if args.shards_in_flight > 1 and world == 1 and args.utts >= 2 * args.shards_in_flight:
        result["two_shards_in_flight" if args.shards_in_flight == 2 else "shards_in_flight"] = shards_in_flight(
            args, dec, tgd, tld, ws.out_ids[:args.utts], ws.out_len[:args.utts], chars, target_chars, dev,
            args.shards_in_flight)
    if args.also and args.also != args.variant and world == 1:
        result["variant_" + args.also] = secondary_variant(args, args.also, words, corpus, tgd, tld, tg, tl)
    if args.large_utts > args.utts and world == 1 and not args.open_lexicon:
        del ws
        result["large_batch"] = large_batch(args, words, corpus, args.large_utts, dev)
    if not args.no_cpu and world == 1:
        from silent_speech_amd.decoder import ids_to_text
        from silent_speech_amd import TextTransform
        n_cmp = min(args.cpu_utts, args.utts)
        texts = [ids_to_text(TextTransform, ids_np[u, :lens_np[u]].tolist()) for u in range(n_cmp)]
        result["cpu_baseline"] = cpu_baseline(words, corpus, targets, args, texts)
        result["cpu_baseline"]["host_cpus"] = os.cpu_count()
        if args.cpu_seconds_all > 0:
            result["cpu_baseline"]["all_cores"] = cpu_all_cores(words, corpus, targets, args)
        result["speedup_vs_cpu_1core"] = result["value"] / result["cpu_baseline"]["value"]
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(result) + "\n").encode())
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
