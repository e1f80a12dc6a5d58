#!/bin/bash
# the C-ABI RCCL leg's child with a one-rank communicator; the unique id is drawn by a process that stays alive (as the bench parent does)
mkdir -p gpurun_out
python - <<'PY'
import subprocess, sys, json, os
import __graft_entry__ as g
fr = g.load_package()
idhex = fr.Comm.unique_id().hex()
out = os.path.abspath("gpurun_out/r06_cabi_child.json")
if os.path.exists(out): os.unlink(out)
p = subprocess.run([sys.executable, "bench.py", "--cabi-child", "0/1/0/%s/%s" % (idhex, out)], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
print("child rc", p.returncode, p.stderr.decode()[-300:])
print(open(out).read())
PY
