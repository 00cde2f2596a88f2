"""Batched, KV-cached AR head: what `EMGTextModel.step` (inference_emg_llm.py:83-88) computes, for all
N utterances x B hypotheses of a decode step at once.

The reference re-runs the frozen decoder-only LLM over the whole sequence [context ; char_embed(ids)] for one
hypothesis at a time and keeps the last position (inference_emg_llm.py:427-429).  Here the context (soft prompt,
prompt embeddings, EMG adapter output) is run once per utterance and its keys/values are shared by the
utterance's hypotheses; every step then feeds ONE new token per hypothesis, after re-ordering the cache of the
generated positions by the scorer's back-pointers.  `teacher_logits` scores whole candidate transcripts against the
same cached contexts in one causal pass (the `ar_logprob` of eval_ar.py:574-588).  The dense contractions stay torch / hipBLASLt; this module
is plumbing around them (no hand-written kernels), and its output feeds `NSDecoder.step`.

Architecture = transformers' LlamaModel (pre-norm, RMSNorm, rotary embedding in the rotate-half form, grouped
query attention, SwiGLU MLP, final norm) + `char_embed` + `lm_head`; `from_hf` takes the weights of such a
model so that the two can be compared position by position.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F


@dataclass
class LlamaDims:
    hidden: int
    layers: int
    heads: int
    kv_heads: int
    head_dim: int
    intermediate: int
    rope_theta: float = 500000.0
    rms_eps: float = 1e-5
    vocab: int = 40


# meta-llama/Llama-3.2-3B, the backbone the reference loads by default (inference_emg_llm.py:327)
LLAMA_3_2_3B = LlamaDims(hidden=3072, layers=28, heads=24, kv_heads=8, head_dim=128, intermediate=8192)


def _rms(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    xf = x.float()
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * xf.to(x.dtype)


def _rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x [..., hd]; cos/sin broadcastable [..., hd] (both halves hold the same angles)."""
    h = x.shape[-1] // 2
    rot = torch.cat((-x[..., h:], x[..., :h]), dim=-1)
    return x * cos + rot * sin


class TorchLlamaStep:
    """KV-cached batched equivalent of EMGTextModel.step for N utterances x B hypotheses."""

    def __init__(self, dims: LlamaDims, *, dtype=torch.bfloat16, device=None, seed: int = 0,
                 weights: Optional[dict] = None):
        self.d = dims
        self.dtype = dtype
        self.device = torch.device(device if device is not None else "cuda")
        if weights is None:
            weights = self._random_weights(seed)
        self.w = {k: v.to(self.device, dtype).contiguous() for k, v in weights.items()}
        inv = 1.0 / (dims.rope_theta ** (torch.arange(0, dims.head_dim, 2, dtype=torch.float32) / dims.head_dim))
        self.inv_freq = inv.to(self.device)
        self.scale = 1.0 / math.sqrt(dims.head_dim)
        self.N = self.B = self.S = self.T = 0

    # ------------------------------------------------------------------ weights
    def _random_weights(self, seed: int) -> dict:
        d = self.d
        # drawn on the target device: a 3B-parameter stand-in takes seconds there, minutes on the host
        g = torch.Generator(device=self.device).manual_seed(seed)
        r = lambda *s: (torch.randn(*s, generator=g, device=self.device) * 0.02).to(self.dtype)
        w = {"char_embed": r(d.vocab, d.hidden), "lm_head.w": r(d.vocab, d.hidden), "lm_head.b": torch.zeros(d.vocab),
             "norm": torch.ones(d.hidden)}
        for l in range(d.layers):
            w[f"{l}.ln1"] = torch.ones(d.hidden)
            w[f"{l}.ln2"] = torch.ones(d.hidden)
            w[f"{l}.qkv"] = r((d.heads + 2 * d.kv_heads) * d.head_dim, d.hidden)
            w[f"{l}.o"] = r(d.hidden, d.heads * d.head_dim)
            w[f"{l}.gate_up"] = r(2 * d.intermediate, d.hidden)
            w[f"{l}.down"] = r(d.hidden, d.intermediate)
        return w

    @classmethod
    def from_hf(cls, llama, char_embed: torch.Tensor, lm_head_w: torch.Tensor, lm_head_b: torch.Tensor, *,
                dtype=torch.float32, device=None) -> "TorchLlamaStep":
        """From a transformers LlamaModel (no rope scaling) and the reference wrapper's char_embed / lm_head."""
        c = llama.config
        hd = getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads
        theta = getattr(c, "rope_theta", None)
        if theta is None:
            theta = (getattr(c, "rope_parameters", None) or {}).get("rope_theta", 10000.0)
        dims = LlamaDims(c.hidden_size, c.num_hidden_layers, c.num_attention_heads, c.num_key_value_heads, hd,
                         c.intermediate_size, float(theta), float(c.rms_norm_eps), char_embed.shape[0])
        sd = llama.state_dict()
        w = {"char_embed": char_embed, "lm_head.w": lm_head_w, "lm_head.b": lm_head_b, "norm": sd["norm.weight"]}
        for l in range(dims.layers):
            p = f"layers.{l}."
            w[f"{l}.ln1"] = sd[p + "input_layernorm.weight"]
            w[f"{l}.ln2"] = sd[p + "post_attention_layernorm.weight"]
            w[f"{l}.qkv"] = torch.cat([sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"],
                                       sd[p + "self_attn.v_proj.weight"]], 0)
            w[f"{l}.o"] = sd[p + "self_attn.o_proj.weight"]
            w[f"{l}.gate_up"] = torch.cat([sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]], 0)
            w[f"{l}.down"] = sd[p + "mlp.down_proj.weight"]
        return cls(dims, dtype=dtype, device=device, weights={k: v.detach().clone() for k, v in w.items()})

    # ------------------------------------------------------------------ helpers
    def _cos_sin(self, pos: torch.Tensor):
        ang = pos.float()[..., None] * self.inv_freq                    # [..., hd/2]
        ang = torch.cat((ang, ang), dim=-1)
        return ang.cos().to(self.dtype), ang.sin().to(self.dtype)

    def _split_qkv(self, qkv: torch.Tensor):
        d = self.d
        q, k, v = qkv.split([d.heads * d.head_dim, d.kv_heads * d.head_dim, d.kv_heads * d.head_dim], dim=-1)
        return (q.unflatten(-1, (d.heads, d.head_dim)), k.unflatten(-1, (d.kv_heads, d.head_dim)),
                v.unflatten(-1, (d.kv_heads, d.head_dim)))

    def _mlp(self, l: int, x: torch.Tensor) -> torch.Tensor:
        gu = F.linear(x, self.w[f"{l}.gate_up"])
        g, u = gu.split(self.d.intermediate, dim=-1)
        return F.linear(F.silu(g) * u, self.w[f"{l}.down"])

    # ------------------------------------------------------------------ context (prefill)
    @torch.no_grad()
    def encode_context(self, context: torch.Tensor, beam: int, max_new_tokens: int) -> None:
        """context [N, S, hidden] (soft prompt + prompt embeddings + EMG adapter output, what
        EMGTextModel.encode_context returns): one causal pass; keys/values are kept per utterance."""
        d = self.d
        x = context.to(self.device, self.dtype)
        N, S, _ = x.shape
        self.N, self.B, self.S, self.T = N, int(beam), S, int(max_new_tokens)
        M = N * self.B
        cos, sin = self._cos_sin(torch.arange(S, device=self.device))    # [S, hd]
        self.kp, self.vp = [], []
        for l in range(d.layers):
            h = _rms(x, self.w[f"{l}.ln1"], d.rms_eps)
            q, k, v = self._split_qkv(F.linear(h, self.w[f"{l}.qkv"]))   # [N, S, heads, hd]
            q = _rope(q, cos[None, :, None, :], sin[None, :, None, :]).transpose(1, 2)
            k = _rope(k, cos[None, :, None, :], sin[None, :, None, :]).transpose(1, 2)   # [N, kvh, S, hd]
            v = v.transpose(1, 2)
            a = F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True)
            x = x + F.linear(a.transpose(1, 2).flatten(-2), self.w[f"{l}.o"])
            x = x + self._mlp(l, _rms(x, self.w[f"{l}.ln2"], d.rms_eps))
            self.kp.append(k.contiguous())
            self.vp.append(v.contiguous())
        # generated positions: [M, kvh, T, hd] per layer, two copies for the re-ordering
        shape = (d.layers, M, d.kv_heads, max(self.T, 1), d.head_dim)
        self.kg = [torch.zeros(shape, dtype=self.dtype, device=self.device) for _ in range(2)]
        self.vg = [torch.zeros(shape, dtype=self.dtype, device=self.device) for _ in range(2)]
        self.cur = 0
        self.t = 0

    # ------------------------------------------------------------------ one decode step
    @torch.no_grad()
    def step(self, tokens: torch.Tensor, parent: Optional[torch.Tensor] = None,
             backptr: Optional[torch.Tensor] = None) -> torch.Tensor:
        """tokens [N*B] int64: the last token of every hypothesis (BOS at the first step); parent [N*B] int64:
        global index of the hypothesis each one extends (the scorer's back-pointer + u*B), None at the first
        step; alternatively backptr [N*B] uint8, the scorer's back-pointers themselves, which on the GPU are
        applied to the cache by one nsd_kv_reorder launch per tensor.  Returns float32 logits [N*B, vocab]."""
        d, t, S, N, B = self.d, self.t, self.S, self.N, self.B
        M = N * B
        if t >= self.T:
            raise ValueError("all max_new_tokens steps already ran")
        if backptr is not None and t > 0 and self.device.type != "cuda":
            base = (torch.arange(N, device=self.device) * B).repeat_interleave(B)
            parent = base + backptr.long().clamp(max=B - 1)
            backptr = None
        reorder = parent is not None and t > 0 and backptr is None
        kg_old, vg_old = self.kg[self.cur], self.vg[self.cur]
        if reorder or (backptr is not None and t > 0):
            self.cur = 1 - self.cur
        kg, vg = self.kg[self.cur], self.vg[self.cur]
        if backptr is not None and t > 0:
            # generated keys/values follow their hypotheses: one batched copy of the first t positions
            from . import lib as _lib
            Lh = _lib.load()
            row_bytes = d.head_dim * kg.element_size()
            st = torch.cuda.current_stream().cuda_stream
            for src, dst in ((kg_old, kg), (vg_old, vg)):
                _lib.check(Lh.nsd_kv_reorder(src.data_ptr(), dst.data_ptr(), backptr.data_ptr(), d.layers, M, B,
                                             d.kv_heads, max(self.T, 1), t, row_bytes, st), "nsd_kv_reorder")
        x = F.embedding(tokens, self.w["char_embed"])                     # [M, hidden]
        cos, sin = self._cos_sin(torch.full((1,), S + t, device=self.device))
        g = d.heads // d.kv_heads
        for l in range(d.layers):
            h = _rms(x, self.w[f"{l}.ln1"], d.rms_eps)
            q, k, v = self._split_qkv(F.linear(h, self.w[f"{l}.qkv"]))   # [M, heads, hd]
            q = _rope(q, cos, sin)
            if reorder:
                # generated keys/values follow their hypotheses (layer by layer, into the other copy)
                kg[l, :, :, :t] = kg_old[l].index_select(0, parent)[:, :, :t]
                vg[l, :, :, :t] = vg_old[l].index_select(0, parent)[:, :, :t]
            kg[l, :, :, t] = _rope(k, cos, sin)
            vg[l, :, :, t] = v
            # context part: the B hypotheses of an utterance share keys/values -> extra query rows
            qn = q.view(N, B, d.kv_heads, g, d.head_dim).permute(0, 2, 1, 3, 4).reshape(N, d.kv_heads, B * g, d.head_dim)
            sp = torch.matmul(qn, self.kp[l].transpose(-1, -2)) * self.scale          # [N, kvh, B*g, S]
            # generated part: per hypothesis
            qm = q.view(M, d.kv_heads, g, d.head_dim)
            sg = torch.matmul(qm, kg[l, :, :, :t + 1].transpose(-1, -2)) * self.scale  # [M, kvh, g, t+1]
            sp = sp.view(N, d.kv_heads, B, g, S).permute(0, 2, 1, 3, 4).reshape(M, d.kv_heads, g, S)
            p = torch.softmax(torch.cat((sp, sg), dim=-1).float(), dim=-1).to(self.dtype)
            pp, pg = p[..., :S], p[..., S:]
            ppn = pp.view(N, B, d.kv_heads, g, S).permute(0, 2, 1, 3, 4).reshape(N, d.kv_heads, B * g, S)
            op = torch.matmul(ppn, self.vp[l]).view(N, d.kv_heads, B, g, d.head_dim).permute(0, 2, 1, 3, 4)
            og = torch.matmul(pg, vg[l, :, :, :t + 1])                                   # [M, kvh, g, hd]
            a = (op.reshape(M, d.kv_heads, g, d.head_dim) + og).reshape(M, d.heads * d.head_dim)
            x = x + F.linear(a, self.w[f"{l}.o"])
            x = x + self._mlp(l, _rms(x, self.w[f"{l}.ln2"], d.rms_eps))
        x = _rms(x, self.w["norm"], d.rms_eps)
        self.t = t + 1
        return F.linear(x, self.w["lm_head.w"], self.w["lm_head.b"]).float()

    # ------------------------------------------------------------------ teacher-forced scoring (rerank E)
    @torch.no_grad()
    def teacher_logits(self, seq: torch.Tensor, lens: torch.Tensor, utt: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Logits after every prefix of M token sequences, against the contexts cached by `encode_context`:
        row i of sequence m is what the reference's `step_fn(seq[m, :i+1])` returns, which its `ar_logprob`
        (eval_ar.py:574-588) obtains from one whole-sequence forward per prefix.  Here it is ONE causal pass over
        [context ; char_embed(seq[:, :-1])] per sequence, the context keys / values taken from the cache.

        seq [M, L] integer (BOS first, padded with any valid id), lens [M] (only documents what the caller may read:
        rows at or past lens[m]-1 come from padding); utt [M] = index of the utterance whose context sequence m
        continues (default m mod N: the AR candidates of all utterances, then the CTC candidates, as
        `decoder.transcribe_batch_joint` lays them out).  Returns float32 [M, L-1, vocab] — the `teacher_logits`
        callable that function asks for."""
        d, S, N = self.d, self.S, self.N
        if N == 0:
            raise ValueError("encode_context has not run")
        seq = seq.to(self.device).long()
        M, L = seq.shape
        if L < 2:
            return torch.zeros(M, 0, d.vocab, dtype=torch.float32, device=self.device)
        if utt is None:
            utt = torch.arange(M, device=self.device) % N
        utt = utt.to(self.device).long()
        L1 = L - 1
        x = F.embedding(seq[:, :L1], self.w["char_embed"])               # [M, L1, hidden]
        cos, sin = self._cos_sin(S + torch.arange(L1, device=self.device))
        cos, sin = cos[None, :, None, :], sin[None, :, None, :]
        # new position i sees the whole context and new positions <= i
        mask = torch.cat((torch.ones(L1, S, dtype=torch.bool, device=self.device),
                          torch.ones(L1, L1, dtype=torch.bool, device=self.device).tril()), dim=1)
        for l in range(d.layers):
            h = _rms(x, self.w[f"{l}.ln1"], d.rms_eps)
            q, k, v = self._split_qkv(F.linear(h, self.w[f"{l}.qkv"]))   # [M, L1, heads, hd]
            q = _rope(q, cos, sin).transpose(1, 2)
            k = torch.cat((self.kp[l].index_select(0, utt), _rope(k, cos, sin).transpose(1, 2)), dim=2)
            v = torch.cat((self.vp[l].index_select(0, utt), v.transpose(1, 2)), dim=2)
            a = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, enable_gqa=True)
            x = x + F.linear(a.transpose(1, 2).flatten(-2), self.w[f"{l}.o"])
            x = x + self._mlp(l, _rms(x, self.w[f"{l}.ln2"], d.rms_eps))
        x = _rms(x, self.w["norm"], d.rms_eps)
        return F.linear(x, self.w["lm_head.w"], self.w["lm_head.b"]).float()

    # ------------------------------------------------------------------ glue for NSDecoder.decode_batch
    def step_fn_batched(self):
        return self._step_fn_batched()

    def _step_fn_batched(self):
        """Callable for `NSDecoder.decode_batch(..., needs_ids=False)`: reads the scorer's back-pointers and
        tokens of the previous step from the workspace."""
        def fn(ids, lens, ws):
            t = ws.steps_run
            if t == 0:
                tok = torch.full((self.N * self.B,), 37, dtype=torch.int64, device=self.device)   # BOS
                return self.step(tok, None)
            tok = ws.token[t - 1, :self.N * self.B].long()
            tok = torch.where(tok >= self.d.vocab, torch.zeros_like(tok), tok)     # carried / absent slots: any id
            return self.step(tok, backptr=ws.backptr[t - 1, :self.N * self.B])
        return fn


class HipLlamaStep(TorchLlamaStep):
    """The same step with everything between the dense contractions in HIP kernels of libnsdecode.so
    (nsd_add_rmsnorm, nsd_rope_kv_append, nsd_attn_step, nsd_swiglu) and a generated-position cache that is never
    re-ordered: position s is a slab [M, kv_heads, hd] written once, and a hypothesis reads the slab rows of its
    ancestors, found through the scorer's back-pointer array (nsd_ancestors).  GPU only."""

    _DT = {torch.float32: 0, torch.bfloat16: 1}

    def __init__(self, *a, use_mfma: bool = True, **kw):
        super().__init__(*a, **kw)
        self.use_mfma = bool(use_mfma)
        if self.dtype not in self._DT:
            raise ValueError("HipLlamaStep stores fp32 or bf16")
        if self.device.type != "cuda":
            raise ValueError("HipLlamaStep runs on the GPU only")
        from . import lib as _lib
        self._lib, self._L = _lib, _lib.load()

    @torch.no_grad()
    def encode_context(self, context: torch.Tensor, beam: int, max_new_tokens: int) -> None:
        super().encode_context(context, beam, max_new_tokens)
        d, M, T = self.d, self.N * self.B, max(self.T, 1)
        del self.kg, self.vg
        kw = dict(dtype=self.dtype, device=self.device)
        self.kgen = torch.zeros(d.layers, T, M, d.kv_heads, d.head_dim, **kw)
        self.vgen = torch.zeros(d.layers, T, M, d.kv_heads, d.head_dim, **kw)
        self.anc = torch.zeros(T, M, dtype=torch.uint8, device=self.device)
        self._own_bp = torch.zeros(T, M, dtype=torch.uint8, device=self.device)
        self._buf = {}
        # context part on the matrix cores (bf16, head_dim 64 / 128, up to 256 context positions): keys padded to a
        # multiple of 32 rows, values transposed; per query row the softmax state the per-hypothesis kernel continues
        S32 = (self.S + 31) // 32 * 32
        self.mfma_ctx = (self.dtype == torch.bfloat16 and d.head_dim in (64, 128) and 0 < self.S and S32 <= 256
                         and self.use_mfma)
        if self.mfma_ctx:
            self.S32 = S32
            self.kpad, self.vt = [], []
            for l in range(d.layers):
                kp = torch.zeros(self.N, d.kv_heads, S32, d.head_dim, **kw)
                kp[:, :, :self.S] = self.kp[l]
                vt = torch.zeros(self.N, d.kv_heads, d.head_dim, S32, **kw)
                vt[:, :, :, :self.S] = self.vp[l].transpose(-1, -2)
                self.kpad.append(kp)
                self.vt.append(vt)
            self.part_o = torch.empty(M * d.heads, d.head_dim, dtype=torch.float32, device=self.device)
            self.part_m = torch.empty(M * d.heads, dtype=torch.float32, device=self.device)
            self.part_l = torch.empty(M * d.heads, dtype=torch.float32, device=self.device)

    def _tmp(self, name, *shape):
        b = self._buf.get(name)
        if b is None or b.shape != shape:
            b = self._buf[name] = torch.empty(*shape, dtype=self.dtype, device=self.device)
        return b

    @torch.no_grad()
    def step(self, tokens: torch.Tensor, parent: Optional[torch.Tensor] = None,
             backptr: Optional[torch.Tensor] = None, backptr_all: Optional[torch.Tensor] = None) -> torch.Tensor:
        """As TorchLlamaStep.step.  The parents of the previous scorer step come as `backptr` (u8 [M], in-beam
        ranks) or `parent` (global indices), and are kept in a [T, M] history of this object; or pass the scorer's
        own back-pointer array `backptr_all` (u8 [T, M], `Workspace.backptr`) and nothing is copied."""
        d, t, S, N, B = self.d, self.t, self.S, self.N, self.B
        M, L, lib, dt = N * B, self._L, self._lib, self._DT[self.dtype]
        if t >= self.T:
            raise ValueError("all max_new_tokens steps already ran")
        st = torch.cuda.current_stream().cuda_stream
        if t > 0:
            if backptr_all is None:
                if backptr is None:
                    if parent is None:
                        raise ValueError("the parents of the previous step are needed from the second step on")
                    backptr = (parent % B).to(torch.uint8)
                self._own_bp[t - 1] = backptr
                backptr_all = self._own_bp
            lib.check(L.nsd_ancestors(backptr_all.data_ptr(), t, N, B, self.anc.data_ptr(), st), "nsd_ancestors")
        ang = (S + t) * self.inv_freq
        ang = torch.cat((ang, ang))
        if self.dtype == torch.float32:
            cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
        else:                                           # the angles' cos / sin rounded to the storage type, as torch does
            cos, sin = ang.cos().to(self.dtype).float().contiguous(), ang.sin().to(self.dtype).float().contiguous()
        x = F.embedding(tokens, self.w["char_embed"]).contiguous()        # [M, hidden]
        h = self._tmp("h", M, d.hidden)
        q = self._tmp("q", M, d.heads, d.head_dim)
        att = self._tmp("att", M, d.heads * d.head_dim)
        act = self._tmp("act", M, d.intermediate)
        delta = None
        for l in range(d.layers):
            lib.check(L.nsd_add_rmsnorm(x.data_ptr(), delta.data_ptr() if delta is not None else None,
                                        self.w[f"{l}.ln1"].data_ptr(), d.rms_eps, M, d.hidden, dt, x.data_ptr(),
                                        h.data_ptr(), st), "nsd_add_rmsnorm")
            qkv = F.linear(h, self.w[f"{l}.qkv"])
            lib.check(L.nsd_rope_kv_append(qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), M, d.heads, d.kv_heads,
                                           d.head_dim, dt, q.data_ptr(), self.kgen[l, t].data_ptr(),
                                           self.vgen[l, t].data_ptr(), st), "nsd_rope_kv_append")
            po = pm = pl = None
            if self.mfma_ctx:
                po, pm, pl = self.part_o.data_ptr(), self.part_m.data_ptr(), self.part_l.data_ptr()
                lib.check(L.nsd_attn_ctx(q.data_ptr(), self.kpad[l].data_ptr(), self.vt[l].data_ptr(), N, B, d.heads,
                                         d.kv_heads, d.head_dim, S, self.S32, self.scale, po, pm, pl, st), "nsd_attn_ctx")
            lib.check(L.nsd_attn_step(q.data_ptr(), self.kp[l].data_ptr(), self.vp[l].data_ptr(),
                                      self.kgen[l].data_ptr(), self.vgen[l].data_ptr(), self.anc.data_ptr(),
                                      att.data_ptr(), N, B, d.heads, d.kv_heads, d.head_dim, S, t, self.scale, dt,
                                      po, pm, pl, st), "nsd_attn_step")
            o = F.linear(att, self.w[f"{l}.o"])
            lib.check(L.nsd_add_rmsnorm(x.data_ptr(), o.data_ptr(), self.w[f"{l}.ln2"].data_ptr(), d.rms_eps, M,
                                        d.hidden, dt, x.data_ptr(), h.data_ptr(), st), "nsd_add_rmsnorm")
            gu = F.linear(h, self.w[f"{l}.gate_up"])
            lib.check(L.nsd_swiglu(gu.data_ptr(), M, d.intermediate, dt, act.data_ptr(), st), "nsd_swiglu")
            delta = F.linear(act, self.w[f"{l}.down"])
        lib.check(L.nsd_add_rmsnorm(x.data_ptr(), delta.data_ptr() if delta is not None else None,
                                    self.w["norm"].data_ptr(), d.rms_eps, M, d.hidden, dt, x.data_ptr(), h.data_ptr(), st),
                  "nsd_add_rmsnorm")
        self.t = t + 1
        return F.linear(h, self.w["lm_head.w"], self.w["lm_head.b"]).float()

    def step_fn_batched(self):
        """Callable for `NSDecoder.decode_batch(..., needs_ids=False)`: the scorer's back-pointer array is read in
        place."""
        def fn(ids, lens, ws):
            t = ws.steps_run
            M = self.N * self.B
            if t == 0:
                return self.step(torch.full((M,), 37, dtype=torch.int64, device=self.device))
            tok = ws.token[t - 1, :M].long()
            tok = torch.where(tok >= self.d.vocab, torch.zeros_like(tok), tok)     # carried / absent slots: any id
            return self.step(tok, backptr_all=ws.backptr)
        return fn
