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


def _run(device, use_backptr=False):
    from silent_speech_amd.ar_head import TorchLlamaStep
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

