#!/bin/bash
# Host-side sanitizer passes (CPU only: this pool has no GPU sanitizer).  `make -C csrc san` builds the library's host sources with g++
# -fsanitize=address,undefined and -fsanitize=thread around the product's device objects; the CPU suite then runs against each through
# FR_LIB with the matching runtime preloaded into python.  Usage: bash tools/run_sanitizers.sh   (about 10 minutes on 8 cores)
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
PKG=gpu-fpga-recommendation-system_amd
make -s -C $PKG/csrc -j8 all san || exit 1
rc=0
# ASan + UBSan: the whole CPU suite (CPU back-end, host logic, the gloo ranks).  One test is left out: it greps the PRODUCT library for
# the experiment knobs' names, and a -g build carries every identifier in its debug info.  Leak checking is off: the interpreter's own.
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 FR_LIB=$PWD/$PKG/libfleetrec_asan.so \
    timeout 3000 python -m pytest tests -q -m "not gpu" -p no:cacheprovider \
    --deselect tests/test_abi.py::test_product_library_reads_no_environment_variable 2>&1 | tee /tmp/fr_asan.log | tail -3 || rc=1
grep -q "ERROR: AddressSanitizer\|runtime error:" /tmp/fr_asan.log && { echo "ASan/UBSan reports: /tmp/fr_asan.log"; rc=1; }
# TSan: the CPU back-end's tests (thread pool, workers on several host threads, the driver loops, and -- round 6 -- the table-sharded step
# with G = 2, 3, 8 ranks on the in-process host exchange: rendezvous, host streams, status words, bounded wait, destroy in flight).  Left out: the two tests that start
# other programs or fork (TSan does not support new threads in the child of a multi-threaded fork; both run under ASan above).
rm -f /tmp/fr_tsan_report.*
LD_PRELOAD="$(gcc -print-file-name=libtsan.so)" TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=0 log_path=/tmp/fr_tsan_report" \
FR_LIB=$PWD/$PKG/libfleetrec_tsan.so \
    timeout 900 python -m pytest tests/test_cpu_backend.py -q -p no:cacheprovider \
    --deselect tests/test_cpu_backend.py::test_server_answers_the_sender_on_the_cpu_back_end \
    --deselect tests/test_cpu_backend.py::test_server_shards_model_c_over_cpu_shard_contexts \
    --deselect tests/test_cpu_backend.py::test_a_forked_child_gets_a_thread_pool_of_its_own 2>&1 | tail -3 || rc=1
# a report counts when one of its frames is in this library (numpy's OpenBLAS threads synchronise in ways TSan cannot see: its
# dgemm_beta / array_dealloc pairs are reported on every run and are not ours)
python3 - <<'PY' || rc=1
import glob, re, sys
text = "".join(open(f).read() for f in glob.glob("/tmp/fr_tsan_report.*"))
reports = [r for r in text.split("==================") if "WARNING: ThreadSanitizer" in r]
ours = [r for r in reports if "libfleetrec" in r]
print("ThreadSanitizer reports: %d, with a frame in libfleetrec: %d" % (len(reports), len(ours)))
for r in ours[:3]:
    print(r)
sys.exit(1 if ours else 0)
PY
exit $rc
