"""The C-ABI boundary: libfleetrec.so loads, exports every symbol include/fleetrec.h declares, and a context asked for on a
device >= 0 refuses to compute without a gfx950 device (no silent CPU fallback; device = -1 asks for the CPU back-end:
tests/test_cpu_backend.py)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# fleetrec.h = the boundary proper (SURVEY 8(b)'s three spans, the driver core, the sharded mode); serving extensions and measurement hooks apart
HEADERS = ("fleetrec.h", "fleetrec_serving.h", "fleetrec_diag.h")


def declared_symbols():
    syms = set()
    for name in HEADERS:
        hdr = open(os.path.join(ROOT, "include", name)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        syms |= set(re.findall(r"\b(fr_[a-z0-9_]+)\s*\(", hdr))
    return sorted(syms)


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "fleetrec.h"\n#include "fleetrec_serving.h"\n#include "fleetrec_diag.h"\nint main(void){ fr_model_desc d; (void)d; return FR_ABI_VERSION - 6; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           "-c", str(src), "-o", str(tmp_path / "t.o")])


def test_exports_every_declared_symbol(fr):
    syms = declared_symbols()
    assert len(syms) >= 35 and sorted(fr.ABI_SYMBOLS) == syms
    L = ctypes.CDLL(fr.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), "libfleetrec.so does not export %s" % s
    assert fr.lib().fr_abi_version() == fr.ABI_VERSION == 6
    # nothing but the fr_* API is exported
    out = subprocess.check_output(["nm", "-D", "--defined-only", fr.LIB_PATH]).decode()
    exported = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert all(e.startswith("fr_") for e in exported), exported


def test_product_library_reads_no_environment_variable(fr):
    """the experiment knobs (FR_GEMM_ABLATE -- wrong results by design --, FR_GATHER_*, FR_FUSED_*, ...) exist only
    in the -DFR_EXPERIMENTS build (`make -C csrc exp` -> libfleetrec_exp.so); the shipped library carries none of their names and does
    not import getenv at all (the GPU suite additionally checks that setting them changes no score bit)."""
    blob = open(fr.LIB_PATH, "rb").read()
    for name in (b"FR_GEMM_ABLATE", b"FR_GEMM_ORDER", b"FR_GEMM_PRIO", b"FR_GEMM_PIPE", b"FR_GATHER_", b"FR_FUSED\0", b"FR_FUSED_HK", b"FR_FUSED_HS_ABLATE", b"FR_LP_GEMM_SPLITK", b"fc_splitk_gemm", b"FR_FUSED_GROUP", b"FR_FUSED_ITEMS",
                 b"FR_FUSED_M2", b"FR_FUSED_WPE", b"FR_FUSED_R1D", b"FR_LP_GEMM", b"FR_SUBMIT_ZEROCOPY", b"FR_SMALL_BLOCK_SERIAL"):
        assert name not in blob, name     # (FR_FUSED_MAX_QUEUE, a constant's name inside an error string, is not a variable)
    und = subprocess.check_output(["nm", "-D", "--undefined-only", fr.LIB_PATH]).decode()
    assert "getenv" not in und, "libfleetrec.so imports getenv"


def test_struct_layout_matches_header(fr, tmp_path):
    """ctypes mirrors of fr_table_desc / fr_segment / fr_model_desc have the C sizes."""
    src = tmp_path / "s.c"
    src.write_text('#include <stdio.h>\n#include "fleetrec.h"\nint main(void){printf("%zu %zu %zu\\n", sizeof(fr_table_desc), sizeof(fr_segment), sizeof(fr_model_desc));return 0;}\n')
    exe = tmp_path / "s"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [ctypes.sizeof(fr.TableDesc), ctypes.sizeof(fr.Segment), ctypes.sizeof(fr.ModelDesc)]


def test_no_silent_cpu_fallback(fr):
    """Without a device a context on device >= 0 is refused loudly: the CPU back-end exists, but only when asked for by name (device = -1).
    (This test only asserts on GPU-less machines.)"""
    if fr.device_count() > 0:
        pytest.skip("a HIP device is visible")
    for dev in (0, 3):
        with pytest.raises(fr.FleetRecError) as e:
            fr.Context(fr.Model.builtin(fr.MODEL_A), device=dev)
        assert e.value.status == fr.FR_ERR_NO_DEVICE and "never falls back to the CPU" in str(e.value)
    with pytest.raises(fr.FleetRecError) as e:
        fr.Context(fr.Model.builtin(fr.MODEL_A), device=-2)
    assert e.value.status == fr.FR_ERR_NO_DEVICE


def test_product_does_not_reference_oracle():
    """No file of the product package mentions the oracle (it is test infrastructure only)."""
    pkg = os.path.join(ROOT, "gpu-fpga-recommendation-system_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".inc", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                for needle in ("liboracle", "import oracle", "from oracle", "oracle/"):
                    if needle in txt:
                        # comments that cite the oracle file for the shared hash spec are allowed
                        lines = [l for l in txt.splitlines() if needle in l and not l.strip().startswith(("//", "#", "*", "/*"))]
                        assert not lines, (f, lines)


def test_core_header_is_the_three_spans_only():
    """include/fleetrec.h carries the boundary SURVEY 8(b) cuts (context, worker, hot-loop body, the driver core, the
    sharded mode); the serving extensions and the measurement hooks live in their own headers and stay out of it."""
    core = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "fleetrec.h")).read(), flags=re.S)
    core_syms = set(re.findall(r"\b(fr_[a-z0-9_]+)\s*\(", core))
    for s in ("fr_worker_push_device_list", "fr_ctx_set_small_block", "fr_worker_host_pending", "fr_worker_flush", "fr_worker_host_poll", "fr_worker_push_host",
              "fr_worker_stage_acquire", "fr_worker_push_staged", "fr_worker_last_kernel", "fr_ctx_set_gather_variant", "fr_worker_timer_start", "fr_worker_fc_layer_only"):
        assert s not in core_syms, s
    for s in ("fr_ctx_create", "fr_ctx_fill_tables", "fr_ctx_set_weights", "fr_worker_create", "fr_worker_idx_ptr", "fr_worker_submit", "fr_worker_sync",
              "fr_worker_gather_only", "fr_worker_fc_only", "fr_worker_submit_sharded", "fr_driver_run_resident"):
        assert s in core_syms, s
    assert len(core_syms) <= 64, len(core_syms)


def test_persistent_kernels_use_no_scratch(fr):
    """the product instantiations of the persistent K-outer bf16 kernel (fr_fused_tile_hs_kernel: BASELINE configs[2]'s
    kernel and its three other record widths) compile without a single spilled vector register and without a private (scratch) segment
    -- round 3's build spilled 11-14 registers per instantiation and reloaded them, behind an s_waitcnt vmcnt(0), in front of FC1's weight
    loads.  Read from the code object's metadata notes inside libfleetrec.so (tools/kernel_resources.py: the offload bundle is unpacked by
    hand, llvm-readelf --notes prints the per-kernel records)."""
    import importlib.util
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not available")
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    recs = [r for r in kr.kernel_records(fr.LIB_PATH) if "fr_fused_tile_hs_kernel" in r["name"]]
    assert len(recs) >= 4, [r["name"] for r in recs]
    for r in recs:
        assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0, r
        assert r["vgpr_count"] <= 168, r      # 12 waves per workgroup = 3 per SIMD
    # ... and no kernel of the library needs more than a few dwords of scratch (a regression guard for the others)
    worst = max(kr.kernel_records(fr.LIB_PATH), key=lambda r: r["private_segment_fixed_size"])
    assert worst["private_segment_fixed_size"] <= 64, worst


def test_lds_dma_kernels_touch_m0_only_through_the_asm(fr, tmp_path):
    """VERDICT r04 item 6(b).  The GEMM kernels move their operands global -> LDS with `buffer_load_dwordx4 ... lds` from inline asm that
    writes M0 (the LDS base of the wave-instruction) WITHOUT a clobber entry -- hipcc rejects "m0" in a clobber list -- on the argument
    that those kernels contain no other M0 user whose set-up the compiler could move across the asm (fr_gemm.hip, `dma`).  This test is
    that argument, checked by a machine at every build: every gfx950 kernel of libfleetrec.so that contains an LDS-DMA load is
    disassembled, and (1) every instruction that mentions m0 must be the asm's own `s_mov_b32 m0, s<N>`; (2) every LDS-DMA load must be
    IMMEDIATELY preceded by such a write (the asm's two lines were not separated)."""
    import importlib.util
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    n_kernels = n_loads = 0
    for i, co in enumerate(kr.code_objects(fr.LIB_PATH)):
        f = tmp_path / ("co%d.o" % i)
        f.write_bytes(co)
        asm = subprocess.run([objdump, "-d", "--no-show-raw-insn", str(f)], capture_output=True, text=True).stdout
        cur, body = None, {}
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1)
                body[cur] = []
            elif cur is not None and line.strip() and not line.lstrip().startswith("//"):
                body[cur].append(re.sub(r"\s*//.*$", "", line).strip())
        for name, ins in body.items():
            dma = [k for k, t in enumerate(ins) if re.match(r"buffer_load_dword(x[234])? .*\blds\b", t)]
            if not dma:
                continue
            n_kernels += 1
            n_loads += len(dma)
            for k, t in enumerate(ins):
                if re.search(r"\bm0\b", t):
                    assert re.match(r"s_mov_b32 m0, s\d+$", t), "%s: unexpected M0 user `%s`" % (name, t)
            for k in dma:
                assert k > 0 and re.match(r"s_mov_b32 m0, s\d+$", ins[k - 1]), "%s: LDS-DMA load not directly behind its M0 write: `%s` / `%s`" % (name, ins[k - 1], ins[k])
    assert n_kernels >= 6 and n_loads >= 100, (n_kernels, n_loads)   # the fc_lp_gemm / fc_gemm_pipe instantiations of three precisions

def test_phased_waves_kernels_keep_their_mfmas_between_the_barriers(fr, tmp_path):
    """fc_pp_gemm_kernel / fc_pp_gemm_n128_kernel split a K sub-step into a fetching and a multiplying phase fenced by s_barrier, with the two
    waves of a SIMD in opposite phases.  MFMAs are register-only instructions: without the sched_barrier(0) on both sides of every barrier
    hipcc moved 29 of a phase's 32 MFMAs behind the next barrier (the product build of round 5's first version: bf16 FC1 117 us for 97, scores
    unchanged -- no parity test can see it).  This test is the guard: in every instantiation of the shipped library the MFMAs sit in whole
    phases -- between two consecutive barriers there are either none or exactly one phase's worth (32 of 16x16x32 in bf16, 8 of 32x32x64 in
    fp8), and never a memory instruction among them."""
    import importlib.util
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    seen = 0
    for i, co in enumerate(kr.code_objects(fr.LIB_PATH)):
        f = tmp_path / ("pp%d.o" % i)
        f.write_bytes(co)
        asm = subprocess.run([objdump, "-d", "--no-show-raw-insn", str(f)], capture_output=True, text=True).stdout
        cur, body = None, {}
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1)
                body[cur] = []
            elif cur is not None and line.strip() and not line.lstrip().startswith("//"):
                body[cur].append(re.sub(r"\s*//.*$", "", line).strip())
        for name, ins in body.items():
            if "fc_pp_gemm" not in name or name.endswith(".kd"):
                continue
            seen += 1
            segs, cur_seg = [], []
            for t in ins:
                if t.startswith("s_barrier"):
                    segs.append(cur_seg)
                    cur_seg = []
                else:
                    cur_seg.append(t)
            segs.append(cur_seg)
            counts = [sum(1 for t in sg if t.startswith("v_mfma")) for sg in segs]
            bf16 = any("v_mfma_f32_16x16x32_bf16" in t for t in ins)
            want = 32 if bf16 else 8
            assert set(counts) <= {0, want} and want in counts, "%s: MFMAs per barrier-to-barrier segment %s (a phase is %d)" % (name, counts, want)
            for sg, c in zip(segs, counts):
                if c:
                    assert not any(re.match(r"(buffer_load|ds_read|ds_write|global_load|buffer_store|global_store)", t) for t in sg), "%s: memory instructions in the multiplying phase" % name
    assert seen >= 7, seen   # <1, 3, {0, 2, 8}>, <2, 2, {0, 2, 8}>, the two 128 x 256 forms


def test_repo_root_holds_only_the_allow_listed_files():
    """nine `hipcc -save-temps` leftovers sat tracked at the repo root for a round (4.7 MB on every GPU box).  What is
    tracked at the root is an allow-list; anything else (a compiler temp, a scratch script) fails here before it is committed twice."""
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("not a git checkout (a gpurun snapshot travels without .git)")
    try:
        out = subprocess.check_output(["git", "-C", ROOT, "ls-files"], stderr=subprocess.DEVNULL).decode()
    except (OSError, subprocess.CalledProcessError):
        pytest.skip("git not usable here")
    roots = sorted({p.split("/")[0] for p in out.splitlines() if p})
    allowed = re.compile(r"^(\.gitignore|\.gpurunignore|ADVICE\.md|BASELINE\.(json|md)|(BENCH|GPUTEST|MULTICHIP|SCALE)_r\d+\.json|DESIGN\.md|INTEGRATION\.md|"
                         r"PAPERS\.md|README\.md|SNIPPETS\.md|SURVEY\.md|VERDICT\.md|__graft_entry__\.py|bench\.py|pytest\.ini|"
                         r"gpu-fpga-recommendation-system_amd|include|oracle|profiles|tests|tools)$")
    stray = [r for r in roots if not allowed.match(r)]
    assert not stray, "tracked at the repo root but not on the allow-list: %s" % stray
