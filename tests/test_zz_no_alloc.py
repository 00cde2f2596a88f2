"""nsd_decode_synth_noalloc (the entry point that allocates nothing: its workspaces live in a scratch array built
into the library) against nsd_decode_synth with caller-owned workspaces, through the C ABI: ids, lengths and
fp64 scores bit-equal, on two calls in a row, the second with a smaller batch (the array keeps what the first
call left), then a larger one again."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _workload(n_utts, seed):
    from silent_speech_amd import synthetic as S
    words = S.make_lexicon(200, 0)
    corpus = S.make_sentences(words, 3000, 1)
    tg, tl = S.pack_targets(S.make_sentences(words, n_utts, seed, 1, 5))
    return words, corpus, tg, tl


@pytest.mark.parametrize("variant", ["U", "I", "E"])
def test_noalloc_decode_equals_the_workspace_decode(variant):
    from silent_speech_amd import lib
    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    from silent_speech_amd.tables import NgramCounts
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = lib.load()
    words, corpus, _, _ = _workload(1, 0)
    tables = DeviceTables.from_words(words, NgramCounts(5, floor_vsz=(variant == "U")).fit(corpus))
    st = torch.cuda.current_stream().cuda_stream
    # (utterances, beam, steps): a large batch, a smaller one (stale state of the first call all around it), a
    # wider beam with fewer utterances (other strides inside the same array), and the largest batch allowed
    for k, (N, B, T) in enumerate([(1500, 16, 48), (200, 8, 30), (300, 32, 40), (4096, 16, 64)]):
        _, _, tg, tl = _workload(N, 50 + k)
        tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
        dec = NSDecoder(tables, B, T, variant=variant, lm_weight=0.05, oov_penalty=1.0, word_bonus=0.35,
                        min_len_for_eos=4)
        ids_a, lens_a, sc_a = dec.decode_synth_native(tgd, tld, base_seed=77, margin=25.0, utt0=10 * k)
        torch.cuda.synchronize()
        out_ids = torch.full((N, T + 1), 0xEE, dtype=torch.uint8, device=dev)
        out_len = torch.full((N,), -5, dtype=torch.int32, device=dev)
        out_sc = torch.full((N,), 123.0, dtype=torch.float64, device=dev)
        free_before = torch.cuda.mem_get_info()[0]
        rc = L.nsd_decode_synth_noalloc(tables.handle, C.byref(dec.params), 77, 10 * k, tgd.data_ptr(), tld.data_ptr(),
                                        tgd.shape[1], 25.0, T, N, B, out_ids.data_ptr(), out_len.data_ptr(),
                                        out_sc.data_ptr(), st)
        assert rc == 0, lib.load().nsd_strerror(rc)
        torch.cuda.synchronize()
        # no device memory taken by the call, the first one included: the scratch address was resolved when the
        # tables were created, and the decode just above has loaded the kernels
        assert torch.cuda.mem_get_info()[0] == free_before
        la, lb = lens_a.cpu().numpy(), out_len.cpu().numpy()
        assert (la == lb).all(), (variant, k)
        ia, ib = ids_a.cpu().numpy(), out_ids.cpu().numpy()
        for u in range(N):
            assert (ia[u, :la[u]] == ib[u, :la[u]]).all(), (variant, k, u)
        assert (sc_a.cpu().numpy().view(np.int64) == out_sc.cpu().numpy().view(np.int64)).all(), (variant, k)


def test_noalloc_decode_rejects_what_does_not_fit():
    from silent_speech_amd import lib
    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    L = lib.load()
    torch.cuda.set_device(0)
    tables = DeviceTables.from_words(["ab", "b"], None)
    dec = NSDecoder(tables, 16, 64)
    z = torch.zeros(8, dtype=torch.int32, device="cuda")
    args = (tables.handle, C.byref(dec.params), 1, 0, z.data_ptr(), z.data_ptr(), 1, 1.0)
    tail = (z.data_ptr(), z.data_ptr(), z.data_ptr(), None)
    assert L.nsd_decode_synth_noalloc(*args, 64, 4097, 16, *tail) == -8           # N * B beyond the scratch
    assert L.nsd_decode_synth_noalloc(*args, 128, 4096, 16, *tail) == -8          # N * B * T beyond the scratch
    assert L.nsd_decode_synth_noalloc(*args, 64, 0, 16, *tail) == 0               # nothing to do
    assert b"scratch" in L.nsd_strerror(-8)
