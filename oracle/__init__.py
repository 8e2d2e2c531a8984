"""CPU oracle for the FRVSR/TecoGAN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A torch-CPU fp32 (float64 on request) restatement of the reference's algorithm
(`/root/reference/lib/{ops,frvsr,Teco}.py`, `main.py:186-216`), each function
citing the reference file:line it follows.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package; `tecogan_amd` never does (a test enforces it).

Pinning status
--------------
* The *wiring* of the path (layer order, indices, packing, losses, optimiser
  schedule) and the in-tree numerics (`upscale_four`, `bicubic_four`,
  space-to-depth, D-input packing, loss formulas) are PINNED: the reference's
  own `lib/ops.py`, `lib/frvsr.py`, `lib/Teco.py` are imported unmodified on top
  of a TF1 stand-in (`oracle/tf1_shim`) by `oracle/make_golden.py`, and the
  outputs are committed under `tests/golden/`.
* The L0 numerics that live in TensorFlow itself (conv SAME padding,
  conv_transpose alignment, legacy resize, dense_image_warp, batch_norm, Adam,
  EMA) are restated from TF1.x semantics (tagged [TF1])
  because TensorFlow is not installable here: for those ops
  **parity is unpinned** by any run of real TensorFlow; they are held by the
  known-answer tests of SURVEY.md section 8(c) in `tests/test_oracle_kat.py`.
"""
