"""Book-keeping for "FPS of an FPS result is the identity" (include/tgn_pointops.h,
tgn_furthestsampling[_dense]_prefix).

Every FPS launch leaves a per-cloud certificate next to its sampled coordinates.  A later FPS call whose input has
the layout of a recorded result is offered that result as `prefix_ref` + `prefix_in`; the kernel compares the input
with it bit for bit, per cloud, before it believes the certificate -- provenance by content, so nothing here has to
be right for the result to be right, with two exceptions this module takes care of:

* the recorded coordinates must still be what the kernel wrote.  The book keeps the tensor the caller also received
  together with its version counter and forgets the entry when the counter has moved (an in-place write); tensors
  without a version counter (torch.inference_mode) are kept as a private copy instead;
* a result is only offered on the stream that produced it: on another stream nothing orders the reading launch after
  the writing one, and the caching allocator may recycle a dropped entry's memory while it is still being read.
"""
import torch


class PrefixBook:
    def __init__(self, cap=8):
        self.cap = cap
        self.entries = []   # (key, device, stream, ref tensor, version or None, certificate), most recent last
        self.stats = {"offered": 0}

    def clear(self):
        del self.entries[:]

    def offer(self, key, device):
        """(certificate, reference coordinates) of the most recent live result with this layout, or (None, None)."""
        stream = torch.cuda.current_stream(device).cuda_stream
        for i in range(len(self.entries) - 1, -1, -1):
            k, dev, st, ref, ver, cert = self.entries[i]
            if ver is not None and ref._version != ver:
                del self.entries[i]     # written to since: no longer what the kernel produced
                continue
            if k == key and dev == device and st == stream:
                self.stats["offered"] += 1
                return cert, ref
        return None, None

    def record(self, key, device, new_xyz, cert, shared):
        """Remember a result.  shared: the caller holds `new_xyz` too (it may write to it)."""
        ver = None
        if shared:
            try:
                ver = new_xyz._version
            except RuntimeError:        # inference tensor: no version counter -> private copy
                new_xyz = new_xyz.clone()
        self.entries.append((key, device, torch.cuda.current_stream(device).cuda_stream, new_xyz, ver, cert))
        del self.entries[:-self.cap]
