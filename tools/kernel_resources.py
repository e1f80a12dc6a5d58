#!/usr/bin/env python3
"""List kernel resource usage (vgpr / sgpr / spills / scratch / LDS) of every gfx950 kernel in a fat binary (.so / .o)."""
import re, subprocess, sys, tempfile, os
def code_objects(path):
    blob = open(path, 'rb').read()
    out = []
    # the offload bundle: magic "__CLANG_OFFLOAD_BUNDLE__", u64 n, then n x (u64 offset, u64 size, u64 triple_len, triple)
    pos = 0
    import struct
    while True:
        i = blob.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
        if i < 0: break
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p); p += 24
            triple = blob[p:p + tl].decode(); p += tl
            if "gfx950" in triple and size:
                out.append(blob[i + off:i + off + size])
        pos = i + 24
    return out
def main(path, pat=None):
    for k, co in enumerate(code_objects(path)):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co); name = f.name
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", name], capture_output=True, text=True).stdout
        os.unlink(name)
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda key: (re.search(r"\.%s:\s+(\S+)" % key, blk) or [None, "?"])[1]
            nm = g("name")
            try:
                nm = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", nm], capture_output=True, text=True).stdout.strip()
            except Exception: pass
            if pat and not re.search(pat, nm): continue
            print("%-90s vgpr %s agpr %s sgpr %s spill v%s s%s scratch %s lds %s" % (nm[:90], g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
