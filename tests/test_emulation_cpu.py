"""Lane-level CPU emulations of the kernels that were written after round 2's GPU budget was spent (csrc/hr_tail.hip,
csrc/conv_wgrad_tr.hip, conv3x3_dma3_kernel of csrc/conv3x3_dma.hip): LDS images, fragment addresses, MFMA operand / accumulator lane layouts, the documented semantic of
ds_read_b64_tr_b16, tile coverage and store masks, checked against the oracle / autograd.  They are what stands in for a GPU
parity run of those kernels until round 3 (their GPU tests are gated behind TG_TEST_UNVALIDATED=1)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + [str(a) for a in args], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_fused_hr_tail_kernel_emulation_matches_oracle():
    out = _run("emu_hr_tail.py", 1, 10, 18)               # 2 x 2 tiles, all four image borders, both column blocks
    assert "every output pixel written exactly once" in out


def test_transpose_read_wgrad_kernel_emulation_matches_autograd():
    out = _run("emu_wgrad_tr.py", 2, 8, 2)                # two images, split-K over two workgroups, top and bottom padding
    assert "dW rel err" in out


def test_deep_prefetch_wide_layer_kernel_addressing_and_wait_protocol():
    out = _run("emu_dma3.py")                              # conv3x3_dma3_kernel: LDS images per buffer, in-order vmcnt protocol
    assert "consistent" in out
