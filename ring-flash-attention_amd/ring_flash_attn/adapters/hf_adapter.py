"""Hugging Face transformers adapter: every attention layer of a model calls
`llama3_flash_attn_varlen_func` on this rank's slice of the packed token stream.

Keeps the three entry points of /root/reference/ring_flash_attn/adapters/hf_adapter.py —
`substitute_hf_flash_attn(process_group, heads_k_stride)` (:361),
`update_ring_flash_attn_params(cu_seqlens, process_group)` (:42), `use_ring_attn(flag)` (:65) —
and the same per-micro-batch protocol (README.md:15-68 of the reference):

    substitute_hf_flash_attn(group, heads_k_stride=1)         # once
    model = AutoModelForCausalLM.from_config(cfg, attn_implementation="ring_attn")
    update_ring_flash_attn_params(cu_seqlens, group)           # per micro-batch (global cu_seqlens)
    model(input_ids=local_chunk, position_ids=local_positions)

Re-targeted to transformers >= 4.48 / 5.x, where attention implementations are looked up in
`ALL_ATTENTION_FUNCTIONS` / `AttentionInterface` (the reference imports private symbols that
no longer exist in transformers 5 — `_flash_supports_window_size`, `is_flash_attn_greater_or_equal`,
hf_adapter.py:9-19 — and `attn_implementation="flash_attention_2"` refuses to load without the
CUDA `flash_attn` wheel).  The adapter therefore registers its own implementation name,
"ring_attn", and — for parity with hf_adapter.py:392-393 — also takes over the
"flash_attention_2" slot and the module-level `_flash_attention_forward` hook.
"""
import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from ..llama3_flash_attn_varlen import (
    llama3_flash_attn_varlen_func,
    llama3_flash_attn_prepare_cu_seqlens,
)

# NB: the name must not contain "flash": transformers 5.x treats any such name as a flash-attention
# kernel request and tries to import the CUDA `flash_attn` package / a hub kernel for it.
ATTN_IMPLEMENTATION = "ring_attn"

DATA_PARAMS = {}
RING_ATTN_SWITCH = True
_STATE = {"group": None, "heads_k_stride": 1, "fallback": None}


def update_ring_flash_attn_params(cu_seqlens: torch.Tensor, process_group: dist.ProcessGroup):
    """Derive this rank's (cu_seqlens_q, cu_seqlens_k, max lens, local_k_slice) from the GLOBAL
    cu_seqlens of the packed micro-batch; cached for all layers (reference hf_adapter.py:42-62)."""
    world_size = dist.get_world_size(group=process_group)
    rank = dist.get_rank(group=process_group)
    (
        cu_seqlens_q,
        cu_seqlens_k,
        max_seqlen_q,
        max_seqlen_k,
        local_k_slice,
    ) = llama3_flash_attn_prepare_cu_seqlens(cu_seqlens, True, rank, world_size)
    DATA_PARAMS.update(
        {
            "cu_seqlens_q": cu_seqlens_q,
            "cu_seqlens_k": cu_seqlens_k,
            "max_seqlen_q": max_seqlen_q,
            "max_seqlen_k": max_seqlen_k,
            "local_k_slice": local_k_slice,
        }
    )


def use_ring_attn(flag):
    global RING_ATTN_SWITCH
    RING_ATTN_SWITCH = flag


def _ring_attention(query_states, key_states, value_states, *, dropout, softmax_scale, causal,
                    softcap=None, deterministic=None, sliding_window=None):
    """(1,S,H,D) local q/k/v -> (1,S,H,D).  Same guards as reference hf_adapter.py:137-147."""
    # reference hf_adapter.py:121-128: a configured sliding window that is shorter than the (local) key length
    # is forwarded as window_size=(w, w); llama3_flash_attn_varlen_func applies it in the kernels (flash_attn
    # window semantics; with `causal` the right bound is 0), on any world size: it gathers K/V, so one kernel call
    # sees every key a query may attend to.
    window_size = (-1, -1)
    if sliding_window is not None and key_states.shape[1] > sliding_window:
        window_size = (sliding_window, sliding_window)
    assert softcap is None, "llama3_flash_attn_varlen_func does not support softcap yet."
    assert causal, "only causal attention is supported yet."
    assert query_states.size(0) == 1, "varlen data should be processed in advance."
    if not DATA_PARAMS:
        raise RuntimeError("call update_ring_flash_attn_params(cu_seqlens, group) before the model forward")
    if deterministic is None:
        deterministic = os.environ.get("FLASH_ATTENTION_DETERMINISTIC", "0") == "1"
    attn_output = llama3_flash_attn_varlen_func(
        query_states.squeeze(dim=0),
        key_states.squeeze(dim=0),
        value_states.squeeze(dim=0),
        cu_seqlens_q=DATA_PARAMS["cu_seqlens_q"],
        cu_seqlens_k=DATA_PARAMS["cu_seqlens_k"],
        max_seqlen_q=DATA_PARAMS["max_seqlen_q"],
        max_seqlen_k=DATA_PARAMS["max_seqlen_k"],
        heads_k_stride=_STATE["heads_k_stride"],
        local_k_slice=DATA_PARAMS["local_k_slice"],
        dropout_p=dropout,
        softmax_scale=softmax_scale,
        causal=causal,
        window_size=window_size,
        deterministic=deterministic,
        group=_STATE["group"],
    )
    return attn_output.unsqueeze(dim=0)


def _target_dtype(query: torch.Tensor, module: torch.nn.Module):
    if query.dtype != torch.float32:
        return None
    if torch.is_autocast_enabled():
        return torch.get_autocast_gpu_dtype()
    if hasattr(module.config, "_pre_quantization_dtype"):
        return module.config._pre_quantization_dtype
    return next(layer for layer in module.modules() if isinstance(layer, torch.nn.Linear)).weight.dtype


def ring_flash_attention_forward(
    module: torch.nn.Module,
    query: torch.Tensor,
    key: torch.Tensor,
    value: torch.Tensor,
    attention_mask: Optional[torch.Tensor],
    dropout: float = 0.0,
    scaling: Optional[float] = None,
    sliding_window: Optional[int] = None,
    softcap: Optional[float] = None,
    **kwargs,
) -> Tuple[torch.Tensor, None]:
    """AttentionInterface entry (query (B,H,S,D), key/value (B,Hk,S,D)) — the counterpart of
    reference hf_adapter.py:293-358."""
    if not RING_ATTN_SWITCH and _STATE["fallback"] is not None:
        return _STATE["fallback"](module, query, key, value, attention_mask, dropout=dropout, scaling=scaling,
                                  sliding_window=sliding_window, softcap=softcap, **kwargs)
    original_dtype = query.dtype
    target_dtype = _target_dtype(query, module)
    # FA layout is (B,S,H,D)
    query = query.transpose(1, 2)
    key = key.transpose(1, 2)
    value = value.transpose(1, 2)
    if target_dtype is not None:
        query, key, value = query.to(target_dtype), key.to(target_dtype), value.to(target_dtype)
    is_causal = kwargs.get("is_causal", None)
    if is_causal is None:
        is_causal = getattr(module, "is_causal", True)
    attn_output = _ring_attention(query, key, value, dropout=dropout, softmax_scale=scaling, causal=is_causal,
                                  softcap=softcap, deterministic=kwargs.get("deterministic", None),
                                  sliding_window=sliding_window)
    return attn_output.to(original_dtype), None


def _patched_flash_attention_forward(old_fn):
    def _flash_attention_forward(query_states, key_states, value_states, attention_mask, query_length,
                                 is_causal, dropout=0.0, position_ids=None, softmax_scale=None,
                                 sliding_window=None, use_top_left_mask=False, softcap=None,
                                 deterministic=None, **kwargs):
        if not RING_ATTN_SWITCH:
            return old_fn(query_states, key_states, value_states, attention_mask, query_length, is_causal,
                          dropout=dropout, position_ids=position_ids, softmax_scale=softmax_scale,
                          sliding_window=sliding_window, use_top_left_mask=use_top_left_mask, softcap=softcap,
                          deterministic=deterministic, **kwargs)
        causal = is_causal if not use_top_left_mask else (is_causal and query_length != 1)
        return _ring_attention(query_states, key_states, value_states, dropout=dropout,
                               softmax_scale=softmax_scale, causal=causal, softcap=softcap,
                               deterministic=deterministic, sliding_window=sliding_window)

    return _flash_attention_forward


def substitute_hf_flash_attn(process_group: dist.ProcessGroup, heads_k_stride: int):
    """Route transformers' attention through ring attention on `process_group`."""
    import transformers

    _STATE["group"] = process_group
    _STATE["heads_k_stride"] = heads_k_stride
    try:
        from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS
    except Exception as e:  # pragma: no cover - very old transformers
        raise ValueError(
            f"The current transformer version {transformers.__version__} is not supported "
            "(needs the ALL_ATTENTION_FUNCTIONS attention registry, transformers >= 4.48)."
        ) from e

    if _STATE["fallback"] is None:
        for name in ("flash_attention_2", "sdpa"):
            try:
                fn = ALL_ATTENTION_FUNCTIONS[name]
            except Exception:
                continue
            if fn is not ring_flash_attention_forward:
                _STATE["fallback"] = fn
                break

    try:
        from transformers import AttentionInterface

        AttentionInterface.register(ATTN_IMPLEMENTATION, ring_flash_attention_forward)
    except Exception:
        ALL_ATTENTION_FUNCTIONS[ATTN_IMPLEMENTATION] = ring_flash_attention_forward
    # parity with the reference: the stock flash-attention slot now means ring attention too
    ALL_ATTENTION_FUNCTIONS["flash_attention_2"] = ring_flash_attention_forward

    try:
        import transformers.modeling_flash_attention_utils as fa_utils

        old = fa_utils._flash_attention_forward
        if not getattr(old, "_rfa_patched", False):
            new = _patched_flash_attention_forward(old)
            new._rfa_patched = True
            fa_utils._flash_attention_forward = new
    except Exception:
        pass
