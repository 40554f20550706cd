"""Memo of kernel operands DERIVED from a module's parameters (eval-mode BatchNorm folded into weights, kernel layouts, padding).

The fused eval paths used to rebuild them on every forward -- some 25 tiny launches per Point-Transformer layer, 1.7 ms of an 11 ms
forward.  An entry is valid while every source tensor is the same object, at the same version counter (optimiser steps,
`load_state_dict`, BatchNorm's running statistics all write in place and bump it), with the same storage address, device and dtype:
`module.to(device)` / `.cuda()` / `.double()` / `param.data = ...` keep the Parameter OBJECT and its version but swap its storage,
which the address / device / dtype part of the key catches.  What no key can see is an in-place write through `.data`
(`p.data.mul_(0)`) or through a raw pointer: nothing in this package writes parameters that way; code that does must call
`invalidate(module)` (or set TGN_DERIVED_MEMO=0, which rebuilds the operands on every forward).  `copy.deepcopy` / pickling of a
module drop the memo (the copies would alias the original's operands).

A fresh entry is followed by a synchronisation of the building stream, so that a later forward on another stream (HotPath, the
side-stream sampling of the Point-Transformer pyramid) never reads operands whose producing kernels are still queued.  Nothing is
STORED while a HIP graph is being captured (the tensors would live in the graph's private pool), but entries made by the warm-up
forwards are used: a captured eval forward refers to the operands it was captured with, so re-capture after changing weights (as
for anything else a graph bakes in).  Inference-mode tensors carry no version counter and are not memoised."""
import os

import torch

ENABLED = os.environ.get("TGN_DERIVED_MEMO", "1") != "0"
_SLOT = "_tgn_derived"


class _Store(dict):
    """The per-module memo.  It never travels with the module: a deep copy or a pickle gets an empty one."""

    def __deepcopy__(self, memo):
        return _Store()

    def __reduce__(self):
        return (_Store, ())


def sources(*modules):
    """Parameters and buffers the operands of `modules` depend on (no recursion: pass the leaf modules)."""
    out = []
    for m in modules:
        out += [p for p in m.parameters(recurse=False)] + [b for b in m.buffers(recurse=False)]
    return out


def invalidate(module):
    """Drop every memoised operand of `module` and its children (after writing parameters behind autograd's back)."""
    for m in module.modules():
        m.__dict__.pop(_SLOT, None)


def _key(tensors):
    return tuple((t._version, t.data_ptr(), t.device, t.dtype) for t in tensors)


def cached(owner, slot, tensors, extra, build):
    """build() memoised on `owner` (an nn.Module) under `slot`, valid while `tensors` are the same objects with the same version,
    storage address, device and dtype, and `extra` (shape parameters of the derivation) is equal."""
    if not ENABLED:
        return build()
    tensors = tuple(t for t in tensors if t is not None)
    try:
        key = _key(tensors)
    except RuntimeError:                                        # inference tensors
        return build()
    store = owner.__dict__.get(_SLOT)
    if not isinstance(store, _Store):
        store = owner.__dict__[_SLOT] = _Store()
    hit = store.get(slot)
    if (hit is not None and hit[0] == extra and hit[2] == key and len(hit[1]) == len(tensors)
            and all(a is b for a, b in zip(hit[1], tensors))):
        return hit[3]                                           # (also inside a capture: entries made before it are ordinary memory)
    on_gpu = any(t.is_cuda for t in tensors)
    if on_gpu and torch.cuda.is_current_stream_capturing():
        return build()                                          # built inside the graph, not kept
    value = build()
    if on_gpu:
        torch.cuda.current_stream().synchronize()
    store[slot] = (extra, tensors, key, value)
    return value
