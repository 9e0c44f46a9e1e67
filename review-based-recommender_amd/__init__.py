"""MI355X-native review-encoder hot path (DeepCoNN / NARRE / D-ATT twin towers).

Layout
  csrc/        hand-written HIP kernels (gfx950) + the C ABI declared in include/rbr_hip.h
  build.py     hipcc driver -> csrc/librbr_hip.so (in-tree)
  _lib.py      ctypes binding of the C ABI (fails loudly when the library is missing)
  functional.py  torch.autograd.Functions over the C ABI
  models/      nn.Modules mirroring the reference's models/{deepconn,narre,dual_att}
  distributed.py  one-process-per-GPU data parallelism (RCCL gradient all-reduce)
  train_step.py   the trainer's optimisation step (train_deepconn_pp.py:161-168)

There is no CPU fallback anywhere in this package: every op raises if librbr_hip.so
cannot be loaded or a tensor is not on a HIP device.
"""
__version__ = "0.1.0"
