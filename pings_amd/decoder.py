"""Decoder MLPs (model/decoder.py) — HIP fused Linear-ReLU-Linear behind `Decoder.mlp_batch`.

`mlp_batch(decoder, x)` evaluates the reference's `Decoder.mlp_batch(x)` (decoder.py:84-98; the
chunking by `infer_bs` there is a memory workaround with no numerical effect, each row is
independent).  `decoder` is the reference's `Decoder` object (or anything exposing `layers`,
`lout`, `use_leaky_relu`): its parameters are used in place, so optimiser steps and
`state_dict()` behave as before.

HIP tensors go through `pings_mlp_forward/backward` (csrc/mlp.hip); host tensors (CPU unit tests
of the host logic) use the module's own torch layers.
"""
from __future__ import annotations

import torch


def _supported(decoder) -> bool:
    return len(decoder.layers) == 1 and not getattr(decoder, "use_leaky_relu", False) \
        and decoder.layers[0].bias is not None and decoder.lout.bias is not None


def mlp_batch(decoder, features: torch.Tensor) -> torch.Tensor:
    if features.is_cuda and _supported(decoder):
        from . import mlp as _mlp

        l0, lo = decoder.layers[0], decoder.lout
        return _mlp.fused_mlp(features, l0.weight, l0.bias, lo.weight, lo.bias)
    return decoder.mlp_batch(features)
