"""A compile-time regression guard that needs no GPU: every register-resident decode kernel of the built library
keeps wave-uniform control flow.  The disassembly of each kernel is scanned for EXEC-masked loops
(tools/scan_kernels.py); round 2 found two kernels where the compiler had produced hundreds of them -- results
correct, TM1280 1.7x and the TM6144 pair variant 100x slower -- which no parity test can see."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "build", "csrc", "decode_ms_f32.o")) or
                    not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs the built objects and llvm-objdump")
def test_decode_kernels_have_uniform_control_flow():
    import scan_kernels
    table = scan_kernels.scan()
    assert len(table) >= 60
    bad = {k: v for k, v in table.items()
           if v[1] > 16 and "decode_ms_f64_kernel" not in k[1]}     # the f64 workspace fallback (variant 100) is a plain loop kernel
    assert not bad, f"kernels with EXEC-masked loops (mis-structured control flow): {bad}"
