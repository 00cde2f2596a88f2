"""The KV-cached batched AR head against the reference's own formulation: transformers' LlamaModel run over the
whole sequence [context ; char_embed(ids)] per hypothesis, last position, lm_head (inference_emg_llm.py:83-88),
with a small random-weight configuration.  fp32 on both sides; tolerance 2e-4 absolute on logits of order 1."""
import numpy as np
import pytest
import torch


def _reference_setup(seed=0):
    from transformers import LlamaConfig, LlamaModel
    cfg = LlamaConfig(hidden_size=64, intermediate_size=160, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=16, vocab_size=8, max_position_embeddings=256,
                      rope_theta=500000.0, rms_norm_eps=1e-5, attention_bias=False, mlp_bias=False)
    torch.manual_seed(seed)
    llm = LlamaModel(cfg).eval()
    char_embed = torch.nn.Embedding(40, 64, padding_idx=39)
    lm_head = torch.nn.Linear(64, 40)
    return llm, char_embed, lm_head


def _reference_step(llm, char_embed, lm_head, context, ids):
    """EMGTextModel.step for one hypothesis: context [1, S, D], ids list."""
    with torch.no_grad():
        tok = char_embed(torch.tensor(ids, dtype=torch.long).unsqueeze(0))
        h = llm(inputs_embeds=torch.cat([context, tok], 1)).last_hidden_state[:, -1, :]
        return lm_head(h)[0]


def _run(device, use_backptr=False, hip=False):
    from silent_speech_amd.ar_head import HipLlamaStep, TorchLlamaStep
    if hip:
        TorchLlamaStep = HipLlamaStep                         # noqa: F811  (same interface)
    llm, char_embed, lm_head = _reference_setup()
    N, B, S, T = 3, 4, 9, 6
    torch.manual_seed(1)
    context = torch.randn(N, S, 64) * 0.5
    head = TorchLlamaStep.from_hf(llm, char_embed.weight.data, lm_head.weight.data, lm_head.bias.data,
                                  dtype=torch.float32, device=device)
    head.encode_context(context, beam=B, max_new_tokens=T)
    rs = np.random.RandomState(2)
    hist = [[37] for _ in range(N * B)]                     # every hypothesis starts from BOS
    parent = None
    worst = 0.0
    for t in range(T):
        tok = torch.tensor([h[-1] for h in hist], dtype=torch.long, device=device)
        if use_backptr and parent is not None:
            logits = head.step(tok, backptr=(parent % B).to(torch.uint8)).cpu()      # nsd_kv_reorder on the GPU
        else:
            logits = head.step(tok, parent).cpu()
        for m in range(N * B):
            want = _reference_step(llm, char_embed, lm_head, context[m // B:m // B + 1], hist[m])
            worst = max(worst, float((logits[m] - want).abs().max()))
        # a random beam re-ordering inside each utterance, as the scorer's back-pointers produce
        par = np.concatenate([u * B + rs.randint(0, B, B) for u in range(N)])
        new_tok = rs.randint(0, 37, N * B)
        hist = [hist[p] + [int(c)] for p, c in zip(par, new_tok)]
        parent = torch.tensor(par, dtype=torch.long, device=device)
    return worst


def test_cached_step_matches_full_sequence_forward_cpu():
    assert _run("cpu") < 2e-4


@pytest.mark.gpu
def test_cached_step_matches_full_sequence_forward_gpu():
    assert _run("cuda") < 2e-4
    assert _run("cuda", use_backptr=True) < 2e-4


def test_backptr_form_on_cpu():
    assert _run("cpu", use_backptr=True) < 2e-4


def _teacher(device):
    """teacher_logits against the reference's ar_logprob formulation (eval_ar.py:574-588): one whole-sequence
    forward per prefix, last position."""
    from silent_speech_amd.ar_head import TorchLlamaStep
    llm, char_embed, lm_head = _reference_setup()
    N, S = 3, 7
    torch.manual_seed(3)
    context = torch.randn(N, S, 64) * 0.5
    head = TorchLlamaStep.from_hf(llm, char_embed.weight.data, lm_head.weight.data, lm_head.bias.data,
                                  dtype=torch.float32, device=device)
    head.encode_context(context, beam=2, max_new_tokens=4)
    rs = np.random.RandomState(4)
    lens = [6, 2, 9, 4, 5, 3]                                  # 2N sequences: utterance = m mod N
    L = max(lens)
    seq = np.full((len(lens), L), 36, np.int64)
    for m, n in enumerate(lens):
        seq[m, :n] = [37] + list(rs.randint(0, 37, n - 2)) + [38]
    got = head.teacher_logits(torch.from_numpy(seq).to(device), torch.tensor(lens, device=device)).cpu()
    assert got.shape == (len(lens), L - 1, 40)
    worst = 0.0
    for m, n in enumerate(lens):
        for i in range(n - 1):
            want = _reference_step(llm, char_embed, lm_head, context[m % N:m % N + 1], [int(t) for t in seq[m, :i + 1]])
            worst = max(worst, float((got[m, i] - want).abs().max()))
    return worst


def test_teacher_forced_logits_match_one_forward_per_prefix_cpu():
    assert _teacher("cpu") < 2e-4


@pytest.mark.gpu
def test_teacher_forced_logits_match_one_forward_per_prefix_gpu():
    assert _teacher("cuda") < 2e-4


@pytest.mark.gpu
def test_kv_reorder_equals_index_select():
    """nsd_kv_reorder against torch.index_select, bit for bit, bf16 and fp32 rows, partial position ranges."""
    from silent_speech_amd import lib
    Lh = lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    for dtype, L, N, B, H, T, t, hd in ((torch.bfloat16, 3, 5, 16, 8, 24, 13, 128), (torch.float32, 2, 7, 4, 2, 9, 9, 16),
                                        (torch.bfloat16, 1, 1, 64, 1, 5, 1, 8)):
        M = N * B
        src = torch.randn(L, M, H, T, hd, generator=g, device=dev).to(dtype)
        dst = torch.full_like(src, 7.0)
        bp = torch.randint(0, B, (M,), generator=g, device=dev, dtype=torch.uint8)
        lib.check(Lh.nsd_kv_reorder(src.data_ptr(), dst.data_ptr(), bp.data_ptr(), L, M, B, H, T, t,
                                    hd * src.element_size(), torch.cuda.current_stream().cuda_stream), "nsd_kv_reorder")
        parent = (torch.arange(N, device=dev) * B).repeat_interleave(B) + bp.long()
        want = src.index_select(1, parent)[:, :, :, :t]
        assert torch.equal(dst[:, :, :, :t], want)
        assert bool((dst[:, :, :, t:] == 7.0).all())          # positions past t are left alone



@pytest.mark.gpu
def test_hip_step_matches_full_sequence_forward_gpu():
    """The step whose non-GEMM part runs in the library's kernels (ancestor table instead of a cache re-ordering,
    fused add + RMSNorm, rotary embedding + cache append, attention over [shared context ; ancestor rows], SwiGLU)
    against transformers' LlamaModel over the whole sequence per hypothesis, fp32, same tolerance."""
    assert _run("cuda", hip=True) < 2e-4
    assert _run("cuda", use_backptr=True, hip=True) < 2e-4


def _attn_reference(q, kctx, vctx, kgen, vgen, anc, N, B, S, t, scale):
    """Plain torch fp64: per hypothesis, keys = context rows of its utterance + its ancestors' slab rows."""
    M, heads, hd = q.shape
    kvh = kgen.shape[2]
    g = heads // kvh
    out = torch.zeros(M, heads, hd, dtype=torch.float64)
    for m in range(M):
        u = m // B
        slots = [u * B + int(anc[s, m]) for s in range(t)] + [m]
        for h in range(kvh):
            k = torch.cat([kctx[u, h].double(), torch.stack([kgen[s, slots[s], h] for s in range(t + 1)]).double()])
            v = torch.cat([vctx[u, h].double(), torch.stack([vgen[s, slots[s], h] for s in range(t + 1)]).double()])
            for j in range(g):
                p = torch.softmax((k @ q[m, h * g + j].double()) * scale, 0)
                out[m, h * g + j] = p @ v
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,N,B,heads,kvh,hd,S,t", [
    (torch.float32, 2, 4, 4, 2, 16, 9, 5), (torch.float32, 1, 1, 1, 1, 4, 0, 0), (torch.float32, 2, 17, 8, 1, 64, 70, 3),
    (torch.bfloat16, 2, 16, 24, 8, 128, 96, 63), (torch.bfloat16, 1, 33, 6, 2, 128, 130, 40),
    (torch.bfloat16, 3, 5, 4, 1, 256, 7, 64), (torch.float32, 2, 3, 3, 3, 32, 64, 0),
    (torch.bfloat16, 2, 7, 8, 2, 64, 33, 10), (torch.bfloat16, 1, 64, 3, 1, 128, 256, 2), (torch.bfloat16, 3, 1, 2, 2, 64, 1, 0),
])
def test_attention_step_kernel(dtype, N, B, heads, kvh, hd, S, t):
    from silent_speech_amd import lib
    L = lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(hd + t)
    M, T = N * B, t + 2
    r = lambda *s: torch.randn(*s, generator=g).to(dtype)
    q, kctx, vctx = r(M, heads, hd), r(N, kvh, max(S, 1), hd)[:, :, :S].contiguous(), r(N, kvh, max(S, 1), hd)[:, :, :S].contiguous()
    kgen, vgen = r(T, M, kvh, hd), r(T, M, kvh, hd)
    bp = torch.randint(0, B, (T, M), generator=g, dtype=torch.uint8)
    bp[:, ::7] = 200                                            # slots a beam does not use: clamped to rank 0
    anc = torch.zeros(T, M, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    lib.check(L.nsd_ancestors(bp.to(dev).data_ptr(), t, N, B, anc.data_ptr(), st), "nsd_ancestors")
    want_anc = torch.zeros(T, M, dtype=torch.uint8)
    for m in range(M):
        a = m % B
        for s in range(t - 1, -1, -1):
            b = int(bp[s, (m // B) * B + a])
            a = b if b < B else 0
            want_anc[s, m] = a
    assert torch.equal(anc.cpu()[:t], want_anc[:t])
    out = torch.zeros(M, heads, hd, dtype=dtype, device=dev)
    scale = 1.0 / hd ** 0.5
    dq, dk, dv, dkg, dvg = (x.to(dev) for x in (q, kctx if S else torch.zeros(1, dtype=dtype), vctx if S else torch.zeros(1, dtype=dtype), kgen, vgen))
    lib.check(L.nsd_attn_step(dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dkg.data_ptr(), dvg.data_ptr(), anc.data_ptr(),
                              out.data_ptr(), N, B, heads, kvh, hd, S, t, scale, 0 if dtype == torch.float32 else 1,
                              None, None, None, st), "nsd_attn_step")
    want = _attn_reference(q, kctx, vctx, kgen, vgen, want_anc, N, B, S, t, scale)
    err = float((out.cpu().double() - want).abs().max())
    assert err < (2e-5 if dtype == torch.float32 else 2e-2), err
    if dtype == torch.bfloat16 and hd in (64, 128) and 0 < S <= 256:
        # the context part on the matrix cores, continued by the per-hypothesis kernel
        S32 = (S + 31) // 32 * 32
        kpad = torch.zeros(N, kvh, S32, hd, dtype=dtype, device=dev)
        kpad[:, :, :S] = dk
        vt = torch.zeros(N, kvh, hd, S32, dtype=dtype, device=dev)
        vt[:, :, :, :S] = dv.transpose(-1, -2)
        po = torch.full((M * heads, hd), float("nan"), dtype=torch.float32, device=dev)
        pm = torch.full((M * heads,), float("nan"), dtype=torch.float32, device=dev)
        pl = torch.full((M * heads,), float("nan"), dtype=torch.float32, device=dev)
        lib.check(L.nsd_attn_ctx(dq.data_ptr(), kpad.data_ptr(), vt.data_ptr(), N, B, heads, kvh, hd, S, S32, scale,
                                 po.data_ptr(), pm.data_ptr(), pl.data_ptr(), st), "nsd_attn_ctx")
        out2 = torch.zeros_like(out)
        lib.check(L.nsd_attn_step(dq.data_ptr(), None, None, dkg.data_ptr(), dvg.data_ptr(), anc.data_ptr(),
                                  out2.data_ptr(), N, B, heads, kvh, hd, S, t, scale, 1, po.data_ptr(), pm.data_ptr(),
                                  pl.data_ptr(), st), "nsd_attn_step")
        assert not bool(torch.isnan(po).any()) and not bool(torch.isnan(pl).any())
        err2 = float((out2.cpu().double() - want).abs().max())
        assert err2 < 2e-2, err2


@pytest.mark.gpu
def test_hip_step_equals_torch_step_bf16():
    """A mid-size bf16 model (head_dim 128, three query heads per kv head, as Llama-3.2-3B): the fused step
    against the torch-only step through several re-ordered steps."""
    from silent_speech_amd.ar_head import HipLlamaStep, LlamaDims, TorchLlamaStep
    dims = LlamaDims(hidden=768, layers=3, heads=6, kv_heads=2, head_dim=128, intermediate=1024)
    dev = torch.device("cuda", 0)
    a = TorchLlamaStep(dims, dtype=torch.bfloat16, device=dev, seed=5)
    b = HipLlamaStep(dims, dtype=torch.bfloat16, device=dev, weights={k: v.clone() for k, v in a.w.items()})
    N, B, S, T = 3, 16, 40, 8
    ctx = torch.randn(N, S, dims.hidden, device=dev, dtype=torch.bfloat16) * 0.5
    a.encode_context(ctx, B, T)
    b.encode_context(ctx, B, T)
    rs = np.random.RandomState(0)
    tok = torch.full((N * B,), 37, dtype=torch.int64, device=dev)
    bp = None
    for t in range(T):
        la = a.step(tok, backptr=bp)
        lb = b.step(tok, backptr=bp)
        scale = float(la.abs().max())
        assert float((la - lb).abs().max()) < 0.04 * max(scale, 1.0), t
        bp = torch.from_numpy(rs.randint(0, B, N * B).astype(np.uint8)).to(dev)
        tok = torch.from_numpy(rs.randint(0, 37, N * B)).to(dev)


@pytest.mark.gpu
def test_joint_rerank_end_to_end_through_the_ar_head():
    """eval_ar.py:503-596 with a real (small, random-weight) decoder-only model behind step_fn: variant E beam decode
    through the KV-cached HIP step, CTC proposal, post-processing, then the joint choice with `ar_logprob` from ONE
    teacher-forced pass (`teacher_logits`).  Expected values: the restatement of that tail with step_fn = one
    whole-sequence transformers forward per prefix, as the reference computes it (fp32 on both sides: scores within
    5e-3, the choice identical wherever the two scores are further apart than 2e-2)."""
    from oracle import ns_oracle as O, synth
    from silent_speech_amd import TextTransform
    from silent_speech_amd.ar_head import HipLlamaStep
    from silent_speech_amd.decoder import DeviceTables, NSDecoder, transcribe_batch_joint
    dev = torch.device("cuda", 0)
    llm, char_embed, lm_head = _reference_setup(5)
    words = synth.make_lexicon(60, 3, max_len=6)
    lm = O.NgramLM(3, floor_vsz=False).fit(synth.make_sentences(words, 2000, 1))
    tables = DeviceTables.from_words(words, lm)
    trie = O.build_trie(words)
    TT = O.TextTransformLike
    N, B, S, T, beta = 5, 4, 6, 14, 0.3
    torch.manual_seed(6)
    context = torch.randn(N, S, 64) * 0.5
    with torch.no_grad():
        lm_head.weight.mul_(30.0)                                     # logits of order 1, so that the beam has preferences
    head = HipLlamaStep.from_hf(llm, char_embed.weight.data, lm_head.weight.data, lm_head.bias.data,
                                dtype=torch.float32, device=dev)
    head.encode_context(context, beam=B, max_new_tokens=T)
    dec = NSDecoder(tables, B, T, variant="E", lm_weight=beta, oov_penalty=1.0, word_bonus=0.35, min_len_for_eos=3)
    ar_ids, ar_len, _ = dec.decode_batch(head.step_fn_batched(), N, needs_ids=False, check_every=0)
    rng = np.random.RandomState(7)
    ctc_targets = synth.make_sentences(words, N, 9, 1, 2)
    Tp = 60
    x = rng.randn(N, Tp, 40).astype(np.float32)
    for u, t in enumerate(ctc_targets):
        f = 0
        for tok in synth.text_to_tokens(t)[: Tp // 3]:
            x[u, f:f + 2, tok] += 8.0; x[u, f + 2, 39] += 8.0; f += 3
        x[u, f:, 39] += 8.0
    final, scores = transcribe_batch_joint(TextTransform, ar_ids, ar_len, tables, trie, words,
                                           torch.from_numpy(x).to(dev), head.teacher_logits, beta)
    ar_np, al = ar_ids.cpu().numpy(), ar_len.cpu().numpy()
    checked = 0
    for u in range(N):
        hyp_ar = O.postprocess(O.ids_to_text(TT, ar_np[u, :al[u]].tolist()), trie, words)
        hyp_ctc = O.postprocess(O.ctc_hypothesis(x[u], TT), trie, words)
        step_fn = lambda ids, u=u: _reference_step(llm, char_embed, lm_head, context[u:u + 1], ids).unsqueeze(0)   # noqa: E731
        ref = [O.ar_logprob(s, step_fn, TT, O.lsm_torch) + beta * float(lm.logp(s)) for s in (hyp_ar, hyp_ctc)]
        for c in (0, 1):
            assert abs(scores[u, c] - ref[c]) <= 5e-3 * max(1.0, abs(ref[c]) / 10), (u, c, scores[u], ref)
        if abs(ref[0] - ref[1]) > 2e-2:
            assert final[u] == (hyp_ctc if ref[1] > ref[0] else hyp_ar), (u, ref)
            checked += 1
    assert checked > 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,hidden,rows,with_delta", [
    (torch.float32, 64, 7, True), (torch.float32, 3072, 33, False), (torch.bfloat16, 3072, 130, True),
    (torch.bfloat16, 8192, 9, True), (torch.bfloat16, 17408, 5, True), (torch.float32, 8200, 3, True)])
def test_add_rmsnorm_kernel(dtype, hidden, rows, with_delta):
    """nsd_add_rmsnorm (residual add + RMSNorm; one workgroup per row, or one wave per row beyond 16 waves) against
    torch: the sum is stored in the element type, the norm is taken of the stored sum in fp32."""
    import ctypes as C
    from silent_speech_amd import lib as _lib
    L = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(hidden + rows)
    x = (torch.randn(rows, hidden, generator=g, device=dev) * 0.7).to(dtype)
    d = (torch.randn(rows, hidden, generator=g, device=dev) * 0.3).to(dtype) if with_delta else None
    w = (1.0 + 0.1 * torch.randn(hidden, generator=g, device=dev)).to(dtype)
    xo, h = torch.empty_like(x), torch.empty_like(x)
    rc = L.nsd_add_rmsnorm(x.data_ptr(), d.data_ptr() if with_delta else None, w.data_ptr(), 1e-5, rows, hidden,
                           0 if dtype == torch.float32 else 1, xo.data_ptr() if with_delta else None, h.data_ptr(),
                           torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    s = (x.float() + d.float()).to(dtype) if with_delta else x
    sf = s.float()
    want = w.float() * (sf * torch.rsqrt(sf.pow(2).mean(-1, keepdim=True) + 1e-5)).to(dtype).float()
    if with_delta:
        assert torch.equal(xo, s)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert float((h.float() - want).abs().max()) <= tol * float(want.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,heads,kvh,hd,M", [
    (torch.float32, 4, 2, 16, 5), (torch.float32, 3, 1, 20, 4), (torch.bfloat16, 24, 8, 128, 37),
    (torch.bfloat16, 4, 4, 64, 9), (torch.bfloat16, 2, 1, 24, 3)])
def test_rope_kv_append_kernel(dtype, heads, kvh, hd, M):
    """nsd_rope_kv_append (16-byte pieces when hd / 2 allows, element pairs otherwise) against the rotate-half form in
    torch: q and k rotated, v copied, each into its own buffer."""
    from silent_speech_amd import lib as _lib
    from silent_speech_amd.ar_head import _rope
    L = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(hd + M)
    qkv = torch.randn(M, heads + 2 * kvh, hd, generator=g, device=dev).to(dtype)
    ang = torch.rand(hd // 2, generator=g, device=dev) * 6.0
    ang = torch.cat((ang, ang))
    cos, sin = ang.cos().to(dtype).float().contiguous(), ang.sin().to(dtype).float().contiguous()
    q = torch.empty(M, heads, hd, dtype=dtype, device=dev)
    k = torch.empty(M, kvh, hd, dtype=dtype, device=dev)
    v = torch.empty(M, kvh, hd, dtype=dtype, device=dev)
    rc = L.nsd_rope_kv_append(qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), M, heads, kvh, hd,
                              0 if dtype == torch.float32 else 1, q.data_ptr(), k.data_ptr(), v.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    want = _rope(qkv[:, :heads + kvh], cos.to(dtype), sin.to(dtype))
    tol = 1e-6 if dtype == torch.float32 else 1.6e-2
    assert float((q.float() - want[:, :heads].float()).abs().max()) <= tol
    assert float((k.float() - want[:, heads:].float()).abs().max()) <= tol
    assert torch.equal(v, qkv[:, heads + kvh:])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,rows,inter", [(torch.float32, 5, 160), (torch.bfloat16, 33, 8192), (torch.bfloat16, 3, 64)])
def test_swiglu_kernel(dtype, rows, inter):
    """nsd_swiglu against torch: silu(gate) rounded to the element type, times up."""
    from silent_speech_amd import lib as _lib
    L = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(inter + rows)
    gu = (torch.randn(rows, 2 * inter, generator=g, device=dev) * 3.0).to(dtype)
    out = torch.empty(rows, inter, dtype=dtype, device=dev)
    rc = L.nsd_swiglu(gu.data_ptr(), rows, inter, 0 if dtype == torch.float32 else 1, out.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    gate, up = gu[:, :inter].float(), gu[:, inter:].float()
    want = (torch.nn.functional.silu(gate).to(dtype).float() * up).to(dtype).float()
    tol = 3e-6 if dtype == torch.float32 else 1.6e-2
    err = (out.float() - want).abs() / want.abs().clamp(min=1.0)
    assert float(err.max()) <= tol
