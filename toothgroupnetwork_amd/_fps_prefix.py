"""Book-keeping for "FPS of an FPS result is the identity" (include/tgn_pointops.h,
tgn_furthestsampling[_dense]_prefix).

Every FPS launch that takes part leaves a per-cloud certificate next to its sampled coordinates.  A later FPS call whose
input has the layout of a recorded result is offered that result as `prefix_ref` + `prefix_in`; the kernel compares the
input with it bit for bit, per cloud, before it believes the certificate -- provenance by content, so nothing here has to
be right for the result to be right, with two exceptions this module takes care of:

* the recorded coordinates must still be what the kernel wrote.  The book keeps a PRIVATE copy of them (3 floats per
  sample): the tensor the caller received may be written to in ways no version counter sees (`.data`, raw-pointer writes
  through the pointops_cuda shim, other kernels), and comparing a tensor with itself would always pass;
* a result is only offered on the stream that produced it: on another stream nothing orders the reading launch after
  the writing one, and the caching allocator may recycle a dropped entry's memory while it is still being read.

The shortcut is OPT-IN per call site (`use_prefix`): the operators (`furthestsampling`, `farthest_point_sample`, ...) run
every launch for real unless TGN_FPS_PREFIX=1; the call sites that chain sampling levels by construction -- the
set-abstraction modules in eval mode, the Point-Transformer sampling pyramid -- ask for it; TGN_FPS_PREFIX=0 turns it off
everywhere.
"""
import os

import torch

_env = os.environ.get("TGN_FPS_PREFIX")
FORCE = None if _env is None or _env == "" else (_env != "0")    # None: the call site decides


def use_prefix(site_opt_in, module_flag=None):
    """module_flag: the importing module's FPS_PREFIX attribute (tests flip it); None = not forced either way."""
    if module_flag is not None:
        return bool(module_flag)
    return bool(site_opt_in)


class PrefixBook:
    def __init__(self, cap=8):
        self.cap = cap
        self.entries = []   # (key, device, stream, private copy of the coordinates, certificate), most recent last
        self.stats = {"offered": 0}

    def clear(self):
        del self.entries[:]

    def offer(self, key, device):
        """(certificate, reference coordinates) of the most recent result with this layout, or (None, None)."""
        stream = torch.cuda.current_stream(device).cuda_stream
        for k, dev, st, ref, cert in reversed(self.entries):
            if k == key and dev == device and st == stream:
                self.stats["offered"] += 1
                return cert, ref
        return None, None

    def adopt(self, device, from_stream, to_stream):
        """`to_stream` has just been made to wait for `from_stream` (wait_stream / an event): results recorded on the latter may be
        offered on the former from now on.  The copies are marked as in use on `to_stream` for the caching allocator."""
        src, dst = from_stream.cuda_stream, to_stream.cuda_stream
        if src == dst:
            return
        have = {(k, id(ref)) for k, dev, st, ref, cert in self.entries if dev == device and st == dst}
        for k, dev, st, ref, cert in list(self.entries):
            if dev == device and st == src and (k, id(ref)) not in have:
                for t in (ref, cert):
                    if t is not None:
                        t.record_stream(to_stream)
                self.entries.append((k, dev, dst, ref, cert))
        del self.entries[:-2 * self.cap]

    def record(self, key, device, new_xyz, cert, shared):
        """Remember a result.  shared: the caller holds `new_xyz` too -- the book keeps its own copy."""
        if shared:
            new_xyz = new_xyz.detach().clone()
        self.entries.append((key, device, torch.cuda.current_stream(device).cuda_stream, new_xyz, cert))
        del self.entries[:-self.cap]
