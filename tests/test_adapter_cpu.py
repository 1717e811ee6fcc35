"""HF adapter (SURVEY §8 row "HF adapter boundary", BASELINE config 5 family) on CPU: a random-init
Qwen3 sharded over 2 gloo ranks through substitute_hf_flash_attn / update_ring_flash_attn_params,
with the CPU oracle as operator backend, must reproduce the single-process eager model (logits and
all parameter gradients).  Checks the adapter wiring + llama3 all-gather / reduce-scatter schedule."""
import pytest
import torch

from conftest import free_port
import _adapter_worker as AW

CFG = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
           num_key_value_heads=2, head_dim=16, vocab_size=128, max_position_embeddings=256)


@pytest.mark.parametrize("W,stride", [(2, 1), (2, 2), (8, 1)])
def test_qwen3_ring_adapter_matches_eager(W, stride):
    cu = [0, 23, 70, 96]                       # deliberately not rank aligned (W = 8: 12 tokens per rank,
    #                                            8 unaligned slices through update_ring_flash_attn_params)
    ref_logits, ref_grads = AW.reference(CFG, cu, torch.float32, torch.device("cpu"))
    logits, grads = AW.run_world(W, CFG, cu, use_hip=False, heads_k_stride=stride, port=free_port())
    assert logits.shape == ref_logits.shape
    assert (logits - ref_logits).abs().max() < 2e-4 * max(1.0, ref_logits.abs().max().item())
    for n, g in ref_grads.items():
        assert (grads[n] - g).abs().max() <= 5e-4 * max(1.0, g.abs().max().item()), n


def test_use_ring_attn_switch_and_param_guard(single_rank_group):
    from ring_flash_attn.adapters import hf_adapter as A

    A.DATA_PARAMS.clear()
    q = torch.zeros(1, 4, 2, 8)
    with pytest.raises(RuntimeError, match="update_ring_flash_attn_params"):
        A._ring_attention(q, q, q, dropout=0.0, softmax_scale=None, causal=True)
    with pytest.raises(AssertionError):
        A._ring_attention(q, q, q, dropout=0.0, softmax_scale=None, causal=False)
    A.use_ring_attn(False)
    assert A.RING_ATTN_SWITCH is False
    A.use_ring_attn(True)


def test_sliding_window_is_honoured(single_rank_group):
    """a model configured with a sliding window shorter than the key length must not silently get
    full causal attention.  The window is forwarded like the reference does (hf_adapter.py:121-128:
    window_size=(w, w) when the key length exceeds it) and the kernels implement it (flash_attn semantics);
    a window that covers the whole key range is a no-op."""
    from ring_flash_attn import backend
    from ring_flash_attn import _testing
    from ring_flash_attn.adapters import hf_adapter as A
    from oracle import flash_attn_ref as O
    from oracle.oracle_backend import OracleBackend

    _testing.set_backend(OracleBackend())
    try:
        A.substitute_hf_flash_attn(None, 1)
        A.update_ring_flash_attn_params(torch.tensor([0, 24], dtype=torch.int32), None)
        g = torch.Generator().manual_seed(2)
        q, k, v = (torch.randn(1, 24, 2, 16, generator=g).to(torch.bfloat16) for _ in range(3))
        full = A._ring_attention(q, k, v, dropout=0.0, softmax_scale=None, causal=True, sliding_window=4096)
        win = A._ring_attention(q, k, v, dropout=0.0, softmax_scale=None, causal=True, sliding_window=4)
        ref_full, _ = O.full_attention_fp64(q, k, v, True)
        ref_win, _ = O.full_attention_fp64(q, k, v, True, window=(4, 4))
        assert (full.double() - ref_full).abs().max() < 2e-2
        assert (win.double() - ref_win).abs().max() < 2e-2
        assert (ref_win - ref_full).abs().max() > 0.1
    finally:
        _testing.set_backend(None)
        A.DATA_PARAMS.clear()


def test_sliding_window_through_the_adapter_over_several_ranks():
    """the adapter's sliding-window path had no multi-rank test.  A Qwen3 whose layers all use sliding
    attention (window 8) runs through substitute_hf_flash_attn on 1, 2 and 4 gloo ranks (llama3 schedule: the window
    is applied by ONE kernel call over the gathered keys, with flash_attn's (w, w) semantics as the reference adapter
    forwards it): logits and parameter gradients must not depend on the number of ranks, and must differ from the
    same model without a window."""
    cfg = dict(CFG, sliding_window=8, use_sliding_window=True, max_window_layers=0,
               layer_types=["sliding_attention"] * CFG["num_hidden_layers"])
    cu = [0, 23, 70, 96]
    one, g_one = AW.run_world(1, cfg, cu, use_hip=False, heads_k_stride=1, port=free_port())
    full, _ = AW.run_world(1, CFG, cu, use_hip=False, heads_k_stride=1, port=free_port())
    assert (one - full).abs().max() > 1e-3, "the window changed nothing"
    for W in (2, 4):
        logits, grads = AW.run_world(W, cfg, cu, use_hip=False, heads_k_stride=1, port=free_port())
        assert (logits - one).abs().max() < 2e-4 * max(1.0, one.abs().max().item()), W
        for n, g in g_one.items():
            assert (grads[n] - g).abs().max() <= 5e-4 * max(1.0, g.abs().max().item()), (W, n)
