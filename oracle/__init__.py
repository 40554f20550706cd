"""CPU oracle for the point-cloud sampling/grouping hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  It is the checker, never the thing measured or shipped: the product
path (``toothgroupnetwork_amd``) never imports it and fails loudly without its HIP library.

``oracle.cpu``      ctypes front-end to ``pointops_oracle.c`` (plain C restatement, numpy in/out)
``oracle.ref_gpu``  ctypes front-end to ``oracle/_ref/libpointops_ref.so`` -- the reference's own
                    ``*_cuda_kernel.cu`` files compiled for gfx950 where they lie (GPU box only)
"""
