#!/usr/bin/env python3
"""Kernel resource usage (VGPRs, SGPRs, spills, scratch, static LDS) of every gfx950 kernel inside a fat binary (.so / .o):
    python3 tools/kernel_resources.py gpu-fpga-recommendation-system_amd/libfleetrec.so [name-regex]
The clang offload bundle is unpacked by hand (magic, entry table), each gfx950 code object goes through `llvm-readelf --notes`, whose
amdhsa.kernels records carry the numbers.  Used by tests/test_abi.py (no scratch in the persistent kernels) and by hand."""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"


def code_objects(path):
    blob = open(path, "rb").read()
    out, pos = [], 0
    while True:
        i = blob.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + tl].decode(errors="replace")
            p += tl
            if "gfx950" in triple and size:
                out.append(blob[i + off:i + off + size])
        pos = i + 24
    return out


def kernel_records(path):
    """-> list of dicts: name (demangled), vgpr_count, agpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count,
    private_segment_fixed_size, group_segment_fixed_size."""
    recs = []
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
            name = f.name
        txt = subprocess.run([READELF, "--notes", name], capture_output=True, text=True).stdout
        os.unlink(name)
        for blk in txt.split("- .agpr_count:")[1:]:
            def g(key, blk=blk):
                m = re.search(r"\.%s:\s+(\S+)" % key, blk)
                return m.group(1) if m else None
            nm = g("name") or "?"
            try:
                nm = subprocess.run([CXXFILT, nm], capture_output=True, text=True).stdout.strip() or nm
            except OSError:
                pass
            rec = {"name": nm, "agpr_count": int(blk.split()[0])}
            for key in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
                v = g(key)
                rec[key] = int(v) if v is not None and v.isdigit() else 0
            recs.append(rec)
    return recs


def main(argv):
    pat = argv[2] if len(argv) > 2 else None
    for r in sorted(kernel_records(argv[1]), key=lambda r: r["name"]):
        if pat and not re.search(pat, r["name"]):
            continue
        print("%-96s vgpr %3d agpr %d sgpr %3d spill v%d s%d scratch %d lds %d" % (r["name"][:96], r["vgpr_count"], r["agpr_count"], r["sgpr_count"],
              r["vgpr_spill_count"], r["sgpr_spill_count"], r["private_segment_fixed_size"], r["group_segment_fixed_size"]))


if __name__ == "__main__":
    main(sys.argv)
