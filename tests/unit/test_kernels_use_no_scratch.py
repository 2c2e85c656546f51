"""No kernel of the library may spill registers: every ``.hip`` source is compiled to
assembly with the build's own flags (hipcc cross-compiles for gfx950 without a GPU) and
the code-object metadata of every kernel must name a private (scratch) segment of 0 bytes.

A spill is silent otherwise - the 256 x 256 GEMM tile once kept 45 registers of its
epilogue in scratch memory and nothing but a profile showed it."""

import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

from nanofed_amd import native_build as nb

SOURCES = sorted((nb.CSRC / "hip").glob("*.hip"))


def _kernel_scratch(asm: str) -> dict[str, int]:
    """{kernel symbol: private_segment_fixed_size} from the amdhsa metadata of ``asm``."""
    out: dict[str, int] = {}
    size = None
    for line in asm.splitlines():
        m = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", line)
        if m:
            size = int(m.group(1))
            continue
        m = re.match(r"\s+\.symbol:\s+(\S+)\.kd", line)
        if m and size is not None:
            out[m.group(1)] = size
            size = None
    return out


def _assemble(src, out_dir):
    out = out_dir / (src.stem + ".s")
    flags = [f for f in nb.HIPCC_FLAGS if f != "-fPIC"]
    proc = subprocess.run([nb._hipcc(), *flags, "-S", "--cuda-device-only", "-o", str(out), str(src)],
                          capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-2000:]
    return src.name, _kernel_scratch(out.read_text())


@pytest.mark.skipif(shutil.which(nb._hipcc()) is None, reason="no hipcc")
def test_every_kernel_has_an_empty_private_segment(tmp_path):
    with ThreadPoolExecutor(max_workers=4) as pool:
        tables = dict(pool.map(lambda s: _assemble(s, tmp_path), SOURCES))
    assert sum(len(t) for t in tables.values()) > 100
    gemm = [k for k in tables["gemm_ops.hip"] if "gemm_kernel" in k]
    assert len(gemm) >= 12, gemm
    spilled = {f"{name}:{k}": v for name, t in tables.items() for k, v in t.items() if v}
    assert not spilled, spilled


def test_metadata_parser_reads_symbol_and_size():
    text = """
    .private_segment_fixed_size: 184
    .sgpr_count:     60
    .symbol:         _Z4gemmv.kd
    .private_segment_fixed_size: 0
    .symbol:         _Z5otherv.kd
"""
    assert _kernel_scratch(text) == {"_Z4gemmv": 184, "_Z5otherv": 0}
