"""Memo of kernel operands DERIVED from a module's parameters (eval-mode BatchNorm folded into weights, kernel layouts, padding).

The fused eval paths used to rebuild them on every forward -- some 25 tiny launches per Point-Transformer layer, 1.7 ms of an 11 ms
forward.  An entry is rebuilt when one of its source tensors is replaced (identity) or written (torch's version counter: optimiser
steps, `load_state_dict`, BatchNorm's running statistics all write in place); `.to()` / `.float()` make new tensors.  Writes torch
cannot see (raw pointers) do not bump the counter: nothing in this package writes parameters that way.

A fresh entry is followed by a synchronisation of the building stream, so that a later forward on another stream (HotPath, the
side-stream sampling of the Point-Transformer pyramid) never reads operands whose producing kernels are still queued.  Nothing is
STORED while a HIP graph is being captured (the tensors would live in the graph's private pool), but entries made by the warm-up
forwards are used: a captured eval forward refers to the operands it was captured with, so re-capture after changing weights (as
for anything else a graph bakes in).  Inference-mode tensors carry no version counter and are not memoised."""
import torch


def sources(*modules):
    """Parameters and buffers the operands of `modules` depend on (no recursion: pass the leaf modules)."""
    out = []
    for m in modules:
        out += [p for p in m.parameters(recurse=False)] + [b for b in m.buffers(recurse=False)]
    return out


def cached(owner, slot, tensors, extra, build):
    """build() memoised on `owner` (an nn.Module) under `slot`, valid while `tensors` are the same objects at the same versions and
    `extra` (shape parameters of the derivation) is equal."""
    tensors = tuple(t for t in tensors if t is not None)
    try:
        versions = tuple(t._version for t in tensors)
    except RuntimeError:                                        # inference tensors
        return build()
    store = owner.__dict__.setdefault("_tgn_derived", {})
    hit = store.get(slot)
    if (hit is not None and hit[0] == extra and hit[2] == versions and len(hit[1]) == len(tensors)
            and all(a is b for a, b in zip(hit[1], tensors))):
        return hit[3]                                           # (also inside a capture: entries made before it are ordinary memory)
    on_gpu = any(t.is_cuda for t in tensors)
    if on_gpu and torch.cuda.is_current_stream_capturing():
        return build()                                          # built inside the graph, not kept
    value = build()
    if on_gpu:
        torch.cuda.current_stream().synchronize()
    store[slot] = (extra, tensors, versions, value)
    return value
