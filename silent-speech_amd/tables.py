"""Table compiler: lexicon trie and char n-gram counts -> flat device tables.

Host-side replacement for the reference's dict-of-dicts ``Trie``
(scripts/ns_utils.py:31-65) and ``CharNgramLM`` dictionaries
(scripts/ns_utils.py:86-150): the trie becomes an array of 16-byte node
records numbered breadth-first, the n-gram counts become two open-addressed
hash tables of fp64 log-probabilities computed here with ``math.log`` exactly
as ``CharNgramLM.logp`` computes them, so the device only adds.
"""
from __future__ import annotations

import math
from typing import Iterable, Optional, Sequence

import numpy as np

BASE_CHARS = "abcdefghijklmnopqrstuvwxyz0123456789 "
SPACE_ID = 36
BOS_ID, EOS_ID, PAD_ID, VOCAB = 37, 38, 39, 40
MAX_ORDER = 10
KEY_EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)

NODE_DTYPE = np.dtype([("mask", "<u8"), ("child_base", "<u4"), ("reserved", "<u4")])
GRAM_DTYPE = np.dtype([("key", "<u8"), ("logp", "<f8")])
CTX_DTYPE = np.dtype([("key", "<u8"), ("term", "<f8"), ("ctx_id", "<u8"), ("backoff", "<f8")])
CTXREC_DTYPE = np.dtype([("backoff", "<f8"), ("term", "<f8"), ("symmask", "<u8"), ("child_base", "<u4"),
                         ("reserved", "<u4")])
GRAMREC_DTYPE = np.dtype([("logp", "<f8"), ("term_next", "<f8"), ("next_ctx", "<u4"), ("reserved", "<u4")])
CTX_NONE = 0xFFFFFFFF

_CHILD_CHARS = BASE_CHARS[:36]
_TOK = {c: i for i, c in enumerate(BASE_CHARS)}
SYM = {c: i + 1 for i, c in enumerate(BASE_CHARS)}
SYM["^"] = 38
SYM["$"] = 39


class HostTrie:
    """Minimal dict trie with the reference's duck type (.next, .end)."""

    __slots__ = ("next", "end")

    def __init__(self):
        self.next = {}
        self.end = False

    @classmethod
    def from_words(cls, words: Iterable[str]) -> "HostTrie":
        """Words are stripped and lower-cased, empties dropped
        (build_trie_from_words, scripts/ns_utils.py:59-65)."""
        root = cls()
        for w in words:
            w = (w or "").strip().lower()
            if not w:
                continue
            node = root
            for ch in w:
                nxt = node.next.get(ch)
                if nxt is None:
                    nxt = node.next[ch] = cls()
                node = nxt
            node.end = True
        return root


def load_lexicon_words(path: str, extra: Optional[str] = None) -> list[str]:
    """Lexicon + optional phrasebook lines, stripped, lower-cased, unique, sorted
    (load_lexicon, scripts/ns_utils.py:68-80)."""
    from pathlib import Path
    words: list[str] = []
    for p in (path, extra):
        if p and Path(p).exists():
            words += [l.strip().lower() for l in Path(p).read_text(encoding="utf-8").splitlines() if l.strip()]
    return sorted(set(words))


def flatten_trie(trie) -> np.ndarray:
    """Breadth-first array of node records from any object tree exposing
    ``.next`` (dict char -> node) and ``.end``.

    Only edges labelled with one of the 36 non-space base characters become
    nodes: an edge with any other label can never be taken by the decoder
    (scripts/ns_utils.py:246-247), and after a space the decoder restarts
    from the root (scripts/ns_utils.py:235), so a literal space edge only
    sets the "space may follow" bit.
    """
    if getattr(trie, "end", False) or " " in trie.next:
        # the empty prefix as a word (or a word starting with a space) would let a space follow a space;
        # the reference's loaders can never build such a trie (empty and unstripped words are dropped)
        raise ValueError("the trie root must not be a word or have a space edge")
    order = [trie]
    masks: list[int] = []
    bases: list[int] = []
    i = 0
    while i < len(order):
        nd = order[i]
        kids = sorted((_TOK[c], ch) for c, ch in nd.next.items() if c in _CHILD_CHARS and len(c) == 1)
        mask = 0
        for tok, _ in kids:
            mask |= 1 << tok
        if nd.end or " " in nd.next:
            mask |= 1 << SPACE_ID
        if nd.end:
            mask |= 1 << 37
        masks.append(mask)
        bases.append(len(order))
        order.extend(ch for _, ch in kids)
        i += 1
    out = np.zeros(len(order), dtype=NODE_DTYPE)
    out["mask"] = np.array(masks, dtype=np.uint64)
    out["child_base"] = np.array(bases, dtype=np.uint32)
    return out


def mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser (the slot hash, nsd_hash64)."""
    z = z.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def pack_key(s: str) -> Optional[int]:
    k = 0
    for ch in s:
        c = SYM.get(ch)
        if c is None:
            return None
        k = (k << 6) | c
    return k


def _capacity(n: int, spare: int = 2) -> int:
    """Power-of-two table size with at least `spare` slots per key."""
    cap = 16
    while cap < spare * n:
        cap *= 2
    return cap


def _place(keys: np.ndarray, cap: int) -> np.ndarray:
    """Linear-probing slot of every (distinct) key."""
    n = keys.shape[0]
    slot = np.full(n, -1, dtype=np.int64)
    used = np.zeros(cap, dtype=bool)
    pos = (mix64(keys) & np.uint64(cap - 1)).astype(np.int64)
    pending = np.arange(n)
    while pending.size:
        p = pos[pending]
        free = ~used[p]
        cand, pc = pending[free], p[free]
        _, first = np.unique(pc, return_index=True)
        win = cand[first]
        slot[win] = pos[win]
        used[pos[win]] = True
        lost = np.ones(pending.size, dtype=bool)
        lost[np.flatnonzero(free)[first]] = False
        pending = pending[lost]
        pos[pending] = (pos[pending] + 1) & (cap - 1)
    return slot


def _place_two_choice(keys: np.ndarray, cap: int, max_kicks: int = 2000):
    """Slot of every (distinct) key under two-choice (cuckoo) placement: a key goes to slot h & (cap - 1) or to
    slot (h >> 32) & (cap - 1), h = mix64(key).  Returns None when some key cannot be placed at this capacity."""
    n = keys.shape[0]
    h = mix64(keys)
    h1 = (h & np.uint64(cap - 1)).astype(np.int64)
    h2 = ((h >> np.uint64(32)) & np.uint64(cap - 1)).astype(np.int64)
    owner = np.full(cap, -1, dtype=np.int64)
    slot = np.full(n, -1, dtype=np.int64)
    pending = np.arange(n)
    for cand in (h1, h2):                         # first come, first served: most keys settle here
        if not pending.size:
            break
        p = cand[pending]
        free = owner[p] < 0
        idx, pos = pending[free], p[free]
        _, first = np.unique(pos, return_index=True)
        win = idx[first]
        owner[cand[win]] = win
        slot[win] = cand[win]
        lost = np.ones(pending.size, dtype=bool)
        lost[np.flatnonzero(free)[first]] = False
        pending = pending[lost]
    rng = np.random.RandomState(12345)
    for k in pending.tolist():                    # the rest: evict along a random walk
        cur = int(k)
        pos = int(h1[cur])
        for _ in range(max_kicks):
            other = int(owner[pos])
            owner[pos] = cur
            slot[cur] = pos
            if other < 0:
                cur = -1
                break
            cur = other
            a, b = int(h1[cur]), int(h2[cur])
            pos = b if pos == a else a
            if a == b or rng.rand() < 0.02:        # both choices coincide, or break a cycle: restart elsewhere
                pos = a if rng.rand() < 0.5 else b
        if cur >= 0:
            return None
    return slot


class NgramCounts:
    """Character n-gram counts with CharNgramLM's attribute names.

    ``fit`` restates CharNgramLM.fit (scripts/ns_utils.py:99-109): every
    n-gram of '^' + text.strip().lower() + '$' and its (n-1)-gram context are
    counted, V collects the characters seen.  ``floor_vsz`` picks the
    smoothing vocabulary of scripts/ns_utils.py:148 (True) or of the in-file
    copies (False)."""

    def __init__(self, order: int = 3, floor_vsz: bool = True):
        self.order = int(order)
        self.floor_vsz = bool(floor_vsz)
        self.counts: dict = {}
        self.context: dict = {}
        self.V: set = set()

    def fit(self, texts: Iterable[str]) -> "NgramCounts":
        n = self.order
        counts, context = self.counts, self.context
        for t in texts:
            s = "^" + (t or "").strip().lower() + "$"
            if len(s) >= n:
                self.V.update(s)
            for i in range(len(s) - n + 1):
                g = s[i:i + n]
                counts[g] = counts.get(g, 0) + 1
                c = g[:-1]
                context[c] = context.get(c, 0) + 1
        return self


class NgramTables:
    """Flat hash tables of one CharNgramLM-like object."""

    def __init__(self, order: int, grams: np.ndarray, ctxs: np.ndarray, unseen: float, vsz: int,
                 ctx_recs: np.ndarray, gram_recs: np.ndarray, init_ctx: int):
        self.order, self.grams, self.ctxs, self.unseen, self.vsz = order, grams, ctxs, unseen, vsz
        self.ctx_recs, self.gram_recs, self.init_ctx = ctx_recs, gram_recs, init_ctx


def compile_ngram(lm, floor_vsz: Optional[bool] = None) -> NgramTables:
    """``lm`` exposes ``order``, ``counts`` (n-gram -> count), ``context``
    ((n-1)-gram -> count) and ``V`` (set of characters), as CharNgramLM does.

    ``floor_vsz`` selects the smoothing vocabulary: max(1, |V|)
    (scripts/ns_utils.py:148) or |V| (inference_emg_llm.py:158,
    eval_ar.py:162).  When None it is read from ``lm.floor_vsz`` or, for an
    empty V (the only case where the two differ), probed from ``lm.logp``.
    """
    order = int(lm.order)
    if not 1 <= order <= MAX_ORDER:
        raise ValueError(f"n-gram order {order} outside [1, {MAX_ORDER}]")
    nV = len(lm.V)
    if floor_vsz is None:
        floor_vsz = getattr(lm, "floor_vsz", None)
    if floor_vsz is None:
        floor_vsz = True if nV else bool(float(lm.logp("a" * max(order, 1))) < 0.0)
    vsz = max(1, nV) if floor_vsz else nV
    log = math.log
    context = lm.context
    counts = lm.counts

    gk, gv = [], []
    for g, c in counts.items():
        k = pack_key(g) if len(g) == order else None
        if k is None:
            continue
        gk.append(k)
        gv.append(log((c + 1) / (context.get(g[:-1], 0) + vsz + 1)))
    ck, cb, ct = [], [], []
    for ctx, C in context.items():
        k = pack_key(ctx) if len(ctx) == order - 1 else None
        if k is None:
            continue
        ck.append(k)
        cb.append(log((0 + 1) / (C + vsz + 1)))
        ct.append(log((counts.get(ctx + "$", 0) + 1) / (C + vsz + 1)))

    gkeys = np.array(gk, dtype=np.uint64)
    gvals = np.array(gv, dtype=np.float64)
    ckeys = np.array(ck, dtype=np.uint64)
    cback = np.array(cb, dtype=np.float64)
    cterm = np.array(ct, dtype=np.float64)
    unseen = log((0 + 1) / (0 + vsz + 1))

    # context automaton: contexts in ascending key order, each one's seen n-grams in ascending symbol order
    corder = np.argsort(ckeys, kind="stable")
    ckeys, cback, cterm = ckeys[corder], cback[corder], cterm[corder]
    gorder = np.argsort(gkeys, kind="stable")            # (ctx << 6 | sym) ascending = grouped by ctx, then sym
    gkeys, gvals = gkeys[gorder], gvals[gorder]
    ctx_recs = np.zeros(len(ckeys), dtype=CTXREC_DTYPE)
    ctx_recs["backoff"], ctx_recs["term"] = cback, cterm
    gram_recs = np.zeros(len(gkeys), dtype=GRAMREC_DTYPE)
    gram_recs["logp"] = gvals
    if len(gkeys):
        sym_mask = np.uint64((1 << (6 * (order - 1))) - 1) if order >= 2 else np.uint64(0)
        gctx = gkeys >> np.uint64(6)
        gsym = (gkeys & np.uint64(63)).astype(np.uint64)
        owner = np.searchsorted(ckeys, gctx)
        ok = (owner < len(ckeys))
        ok[ok] = ckeys[owner[ok]] == gctx[ok]
        if not ok.all():
            raise ValueError("n-gram whose context is not in lm.context")
        first = np.searchsorted(gctx, ckeys, side="left")
        ctx_recs["child_base"] = first.astype(np.uint32)
        np.bitwise_or.at(ctx_recs["symmask"], owner, np.uint64(1) << gsym)
        nkeys = gkeys & sym_mask
        nxt = np.searchsorted(ckeys, nkeys)
        hit = nxt < len(ckeys)
        hit[hit] = ckeys[nxt[hit]] == nkeys[hit]
        gram_recs["next_ctx"] = np.where(hit, nxt, CTX_NONE).astype(np.uint32)
        gram_recs["term_next"] = np.where(hit, cterm[np.minimum(nxt, len(ckeys) - 1)], unseen)
    init_key = {1: 0, 2: SYM["^"]}.get(order)
    init_ctx = CTX_NONE
    if init_key is not None and len(ckeys):
        j = int(np.searchsorted(ckeys, np.uint64(init_key)))
        if j < len(ckeys) and int(ckeys[j]) == init_key:
            init_ctx = j

    grams = np.zeros(_capacity(len(gkeys)), dtype=GRAM_DTYPE)
    grams["key"] = KEY_EMPTY
    if len(gkeys):
        sl = _place(gkeys, grams.shape[0])
        grams["key"][sl] = gkeys
        grams["logp"][sl] = gvals
    # the decoder asks this table about every unseen transition, mostly for keys it does not hold; two-choice
    # placement answers with two independent reads either way (no probe chains)
    cap = _capacity(len(ckeys), 3)
    sl = _place_two_choice(ckeys, cap) if len(ckeys) else None
    while len(ckeys) and sl is None:
        cap *= 2
        sl = _place_two_choice(ckeys, cap)
    ctxs = np.zeros(cap, dtype=CTX_DTYPE)
    ctxs["key"] = KEY_EMPTY
    if len(ckeys):
        ctxs["key"][sl] = ckeys
        ctxs["backoff"][sl] = cback
        ctxs["term"][sl] = cterm
        ctxs["ctx_id"][sl] = np.arange(len(ckeys), dtype=np.uint64)
    return NgramTables(order, grams, ctxs, unseen, vsz, ctx_recs, gram_recs, init_ctx)


# ----------------------------------------------------------------------
# host-side lookups over the flat tables (used by the CPU-only tests of the
# table compiler; the decoder itself only reads them on the device)
# ----------------------------------------------------------------------
def host_find_ctx(table: np.ndarray, key: int) -> int:
    """Slot of a context key in the two-choice table, or -1."""
    cap = table.shape[0]
    h = int(mix64(np.array([key], dtype=np.uint64))[0])
    for pos in (h & (cap - 1), (h >> 32) & (cap - 1)):
        if int(table["key"][pos]) == key:
            return pos
    return -1


def host_find(table: np.ndarray, key: int) -> int:
    """Slot of an n-gram key in the linear-probing table, or -1."""
    cap = table.shape[0]
    h = int(mix64(np.array([key], dtype=np.uint64))[0]) & (cap - 1)
    keys = table["key"]
    while True:
        k = int(keys[h])
        if k == key:
            return h
        if k == int(KEY_EMPTY):
            return -1
        h = (h + 1) & (cap - 1)


def host_logp(t: NgramTables, text: str) -> float:
    """Whole-string score from the flat tables, same summation order as
    CharNgramLM.logp (scripts/ns_utils.py:138-150)."""
    if not text:
        return -1e9
    s = "^" + text.strip().lower() + "$"
    n = t.order
    total = 0.0
    for i in range(len(s) - n + 1):
        g = s[i:i + n]
        gk, ck = pack_key(g), pack_key(g[:-1])
        v = None
        if gk is not None:
            j = host_find(t.grams, gk)
            if j >= 0:
                v = float(t.grams["logp"][j])
        if v is None:
            j = host_find_ctx(t.ctxs, ck) if ck is not None else -1
            v = float(t.ctxs["backoff"][j]) if j >= 0 else t.unseen
        total += v
    return total


def host_trie_walk(nodes: np.ndarray, word: str) -> int:
    """Node index of ``word`` (base characters only) or -1."""
    cur = 0
    for ch in word:
        tok = _TOK.get(ch, 99)
        mask = int(nodes["mask"][cur])
        if tok >= 36 or not (mask >> tok) & 1:
            return -1
        cur = int(nodes["child_base"][cur]) + bin(mask & ((1 << tok) - 1)).count("1")
    return cur


def host_automaton_logp(t: NgramTables, text: str) -> float:
    """Whole-string score by walking the context automaton (ctx_recs / gram_recs) the way nsd_step does,
    falling back to the context hash on unseen transitions; same summation order as CharNgramLM.logp."""
    if not text:
        return -1e9
    s = "^" + text.strip().lower() + "$"
    n = t.order
    syms = [SYM[c] for c in s]
    total = 0.0
    key, cid = 0, CTX_NONE
    sym_mask = (1 << (6 * (n - 1))) - 1 if n >= 2 else 0

    def find(k):
        j = host_find_ctx(t.ctxs, k)
        return int(t.ctxs["ctx_id"][j]) if j >= 0 else CTX_NONE

    if n == 1:
        cid = find(0)
    for i, sym in enumerate(syms):
        have = i                                   # symbols before this one
        nxt_id = CTX_NONE
        resolved = False
        if have >= n - 1 and i + 1 >= n:           # the window ending at symbol i exists
            if cid != CTX_NONE:
                rec = t.ctx_recs[cid]
                if (int(rec["symmask"]) >> sym) & 1:
                    g = t.gram_recs[int(rec["child_base"]) + bin(int(rec["symmask"]) & ((1 << sym) - 1)).count("1")]
                    total += float(g["logp"])
                    nxt_id, resolved = int(g["next_ctx"]), True
                else:
                    total += float(rec["backoff"])
            else:
                total += t.unseen
        key = ((key << 6) | sym) & sym_mask if n >= 2 else 0
        if have + 1 >= n - 1:
            cid = nxt_id if resolved else find(key)
        else:
            cid = CTX_NONE
    return total


# ----------------------------------------------------------------------
# on-disk format of the replicated tables: one .npz (arrays only, nothing is unpickled on load)
# ----------------------------------------------------------------------
TABLE_FORMAT_VERSION = 2


def save_tables(path: str, nodes: np.ndarray, ngram: Optional[NgramTables]) -> None:
    """Write the flat trie and (optionally) the n-gram tables to ``path`` (.npz)."""
    arrays = {"version": np.array([TABLE_FORMAT_VERSION], np.int64), "nodes": nodes.view(np.uint8).reshape(-1)}
    if ngram is not None:
        arrays.update(order=np.array([ngram.order], np.int64), vsz=np.array([ngram.vsz], np.int64),
                      unseen=np.array([ngram.unseen], np.float64), init_ctx=np.array([ngram.init_ctx], np.int64),
                      grams=ngram.grams.view(np.uint8).reshape(-1), ctxs=ngram.ctxs.view(np.uint8).reshape(-1),
                      ctx_recs=ngram.ctx_recs.view(np.uint8).reshape(-1),
                      gram_recs=ngram.gram_recs.view(np.uint8).reshape(-1))
    np.savez(path, **arrays)


def load_tables(path: str):
    """Inverse of save_tables: (nodes, NgramTables or None)."""
    with np.load(path, allow_pickle=False) as z:
        if int(z["version"][0]) != TABLE_FORMAT_VERSION:
            raise ValueError(f"unsupported table file version {int(z['version'][0])}")
        nodes = z["nodes"].view(NODE_DTYPE).copy()
        if "order" not in z.files:
            return nodes, None
        ngram = NgramTables(int(z["order"][0]), z["grams"].view(GRAM_DTYPE).copy(), z["ctxs"].view(CTX_DTYPE).copy(),
                            float(z["unseen"][0]), int(z["vsz"][0]), z["ctx_recs"].view(CTXREC_DTYPE).copy(),
                            z["gram_recs"].view(GRAMREC_DTYPE).copy(), int(z["init_ctx"][0]))
    return nodes, ngram
