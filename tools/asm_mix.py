"""Static instruction mix per kernel of a gfx950 assembly file (hipcc -S --cuda-device-only).

    python tools/asm_mix.py file.s [kernel-substring ...]

Counts, per kernel, the instructions by class (MFMA, other VALU, LDS, vector memory, scalar
memory, waits, barriers) and the register / scratch figures of its descriptor.  The counts are
static (loops are counted once), which is what one compares between two builds of a kernel.
"""
from __future__ import annotations

import re
import sys
from collections import Counter


def classify(op: str) -> str:
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "scratch" if op.startswith("scratch_") else "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def parse(path: str) -> dict[str, dict]:
    kernels: dict[str, dict] = {}
    cur = None
    meta_re = re.compile(r"\.(sgpr_count|vgpr_count|agpr_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\d+)")
    name_re = re.compile(r"^\s+\.name:\s+(\S+)")
    metas: list[dict] = []
    for line in open(path):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L") and not line.startswith("\t"):
            cur = m.group(1)
            kernels.setdefault(cur, {"mix": Counter(), "ops": Counter()})
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None if line.startswith(".Lfunc_end") else cur
        if cur and line.startswith("\t") and not line.startswith("\t."):
            op = line.split()[0]
            if op.startswith(";"):
                continue
            kernels[cur]["mix"][classify(op)] += 1
            kernels[cur]["ops"][op] += 1
        mm = name_re.match(line)
        if mm:
            metas.append({"name": mm.group(1)})
        mm = meta_re.search(line)
        if mm and metas:
            metas[-1][mm.group(1)] = int(mm.group(2))
    for meta in metas:
        k = kernels.get(meta.get("name", ""))
        if k is not None:
            k["meta"] = meta
    return {k: v for k, v in kernels.items() if v["mix"]}


def main() -> None:
    path, filters = sys.argv[1], sys.argv[2:]
    for name, k in parse(path).items():
        if filters and not any(f in name for f in filters):
            continue
        meta = k.get("meta", {})
        mix = k["mix"]
        print(f"{name}\n   " + " ".join(f"{c}={mix[c]}" for c in
              ("mfma", "valu", "lds", "vmem", "smem", "salu", "wait", "barrier", "scratch"))
              + "  | " + " ".join(f"{a}={meta[a]}" for a in
              ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size") if a in meta))
        if "--ops" in sys.argv:
            print("   " + ", ".join(f"{o}:{n}" for o, n in k["ops"].most_common(25)))


if __name__ == "__main__":
    main()
