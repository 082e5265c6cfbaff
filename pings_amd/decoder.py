"""Decoder MLPs (model/decoder.py) — HIP fused Linear-ReLU-Linear behind `Decoder.mlp_batch`.

`mlp_batch(decoder, x)` evaluates the reference's `Decoder.mlp_batch(x)` (decoder.py:84-98; the
chunking by `infer_bs` there is a memory workaround with no numerical effect, each row is
independent).  `decoder` is the reference's `Decoder` object (or anything exposing `layers`,
`lout`, `use_leaky_relu`): its parameters are used in place, so optimiser steps and
`state_dict()` behave as before.

HIP tensors go through `pings_mlp_forward/backward` (csrc/mlp.hip).  Host tensors raise: there is no
CPU path (the CPU restatement used by the tests is the decoder's own `mlp_batch`, called from oracle/).
Decoder shapes the kernel does not cover (more than one hidden level, leaky ReLU, no bias, input > 64, output > 32
or a hidden width outside {32, 64, 96, 128} — none of the shipped configs, pings.py:147-172) run the module's own
torch layers ON THE DEVICE (`decoder.mlp_batch`): still no CPU path, but not a hand-written kernel either.
"""
from __future__ import annotations

import torch


def _supported(decoder) -> bool:
    if not (len(decoder.layers) == 1 and not getattr(decoder, "use_leaky_relu", False)
            and decoder.layers[0].bias is not None and decoder.lout.bias is not None):
        return False
    from . import mlp as _mlp

    W1, W2 = decoder.layers[0].weight, decoder.lout.weight
    return _mlp.supported(int(W1.shape[1]), int(W1.shape[0]), int(W2.shape[0]))


def mlp_batch(decoder, features: torch.Tensor) -> torch.Tensor:
    if not features.is_cuda:
        from . import _lib

        raise _lib.PingsHipError("decoder.mlp_batch runs on the HIP device only (got a CPU tensor); "
                                 "there is no CPU fallback")
    if _supported(decoder):
        from . import mlp as _mlp

        l0, lo = decoder.layers[0], decoder.lout
        return _mlp.fused_mlp(features, l0.weight, l0.bias, lo.weight, lo.bias)
    orig = getattr(type(decoder), "_pings_mlp_batch_torch", None)   # set by install(): the class's own method
    return orig(decoder, features) if orig is not None else decoder.mlp_batch(features)


def mlp(decoder, features: torch.Tensor) -> torch.Tensor:
    """`Decoder.mlp(features)` (model/decoder.py:62-82) for [N, IN] or [N, K, IN] inputs through the fused MFMA kernels.
    Every head of the class goes through it — `sdf` (:100-104), `occupancy` (:115-117), `sem_label_prob` / `sem_label`
    (:119-126) and `regress_color` (:133-134; the mapper's colour-on SDF loop, utils/mapper.py:866-870, and the mesher's
    colour head) — so rebinding this one method moves them all off the torch GEMMs.  Decoder shapes the kernels do not
    cover (more than one hidden level, leaky ReLU, no bias, widths outside `mlp.supported`) run the module's own torch
    layers on the device."""
    if not features.is_cuda:
        from . import _lib

        raise _lib.PingsHipError("decoder.mlp runs on the HIP device only (got a CPU tensor); there is no CPU fallback")
    if _supported(decoder) and features.numel() > 0:
        from . import mlp as _mlp

        l0, lo = decoder.layers[0], decoder.lout
        y = _mlp.fused_mlp(features.reshape(-1, features.shape[-1]), l0.weight, l0.bias, lo.weight, lo.bias)
        return y.view(*features.shape[:-1], y.shape[-1])
    orig = getattr(type(decoder), "_pings_mlp_torch", None)     # set by install(): the class's own method
    if orig is None:
        cls_mlp = getattr(type(decoder), "mlp", None)           # not installed: the class's `mlp` is still its own
        if cls_mlp is not None and cls_mlp is not mlp:
            return cls_mlp(decoder, features)
        raise NotImplementedError("decoder.mlp: this decoder shape has no fused kernel and the class's own `mlp` was "
                                  "not kept (call pings_amd.decoder.install(Decoder))")
    return orig(decoder, features)


def sdf(decoder, features: torch.Tensor) -> torch.Tensor:
    """`Decoder.sdf(features)` (model/decoder.py:100-104): `mlp(features).squeeze(1) * sdf_scale` for [N, IN] or
    [N, K, IN] inputs — the decoder call of the mapper's training / inference loops (utils/mapper.py:537,574,858,1508,
    2275) — through the fused MFMA kernel pair.  First-order backward in HIP; a recorded backward (create_graph=True,
    only taken with `numerical_grad: False`) is the graph node `mlp._FusedMLPBackward`, whose own backward is
    `pings_mlp_double_backward` (csrc/mlp.hip) for this decoder shape."""
    if not features.is_cuda:
        from . import _lib

        raise _lib.PingsHipError("decoder.sdf runs on the HIP device only (got a CPU tensor); there is no CPU fallback")
    if _supported(decoder) and features.shape[0] > 0:
        from . import mlp as _mlp

        l0, lo = decoder.layers[0], decoder.lout
        y = _mlp.fused_mlp(features.reshape(-1, features.shape[-1]), l0.weight, l0.bias, lo.weight, lo.bias)
        return y.view(*features.shape[:-1], y.shape[-1]).squeeze(1) * decoder.sdf_scale
    orig = getattr(type(decoder), "_pings_mlp_torch", None)
    return (orig(decoder, features) if orig is not None else decoder.mlp(features)).squeeze(1) * decoder.sdf_scale


def install(decoder_cls) -> None:
    """`Decoder.mlp` — and with it `sdf`, `occupancy`, `sem_label_prob`, `sem_label`, `regress_color` — and
    `Decoder.mlp_batch` of the reference class through the fused kernels (the parameters stay the module's own)."""
    if hasattr(decoder_cls, "mlp") and not hasattr(decoder_cls, "_pings_mlp_torch"):
        decoder_cls._pings_mlp_torch = decoder_cls.mlp
        decoder_cls.mlp = mlp
    decoder_cls.sdf = sdf
    if hasattr(decoder_cls, "mlp_batch") and not hasattr(decoder_cls, "_pings_mlp_batch_torch"):
        decoder_cls._pings_mlp_batch_torch = decoder_cls.mlp_batch
        decoder_cls.mlp_batch = lambda self, features: mlp_batch(self, features)


def mlp_batch_group(decoders, features, fc=None):
    """`[d.mlp_batch(x) for d, x in zip(decoders, features)]` with ONE kernel launch each way when every decoder has
    the shape of the shipped spawn decoders (hidden 128, input <= 32; pings.py:156-160), else decoder by decoder."""
    if not features[0].is_cuda:
        from . import _lib

        raise _lib.PingsHipError("decoder.mlp_batch runs on the HIP device only (got a CPU tensor); "
                                 "there is no CPU fallback")
    if all(_supported(d) for d in decoders):
        from . import mlp as _mlp

        params = [(d.layers[0].weight, d.layers[0].bias, d.lout.weight, d.lout.bias) for d in decoders]
        return _mlp.fused_mlp_group(list(features), params, fc)
    if fc is not None:
        raise NotImplementedError("device-counted rows need the grouped MFMA decoders (hidden 128, input <= 32)")
    return [mlp_batch(d, x) for d, x in zip(decoders, features)]
