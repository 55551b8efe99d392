# the exact-order plan's time has two states per process (profiles/r05_plan_beside.txt): N fresh processes, each prints where its
# arena landed (address of the first block carved from it) beside the plan time of three handles
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-8}); do
  PFD_TOOL_RESERVE_GIB=${2:-100} python - <<'PY' 2>&1 | grep -v "^W2026\|^E2026"
import os, sys, subprocess
sys.path.insert(0, os.getcwd())
from pyflwdir_amd import _hip
_hip.reserve(int(os.environ["PFD_TOOL_RESERVE_GIB"]) << 30)
b = _hip.DeviceBuffer(8 << 20)
addr = b.addr
b.free()
sys.argv = ["plan_time.py", "30000", "30000", "3"]
os.environ.pop("PFD_TOOL_RESERVE_GIB")
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    exec(open("tools/plan_time.py").read())
line = buf.getvalue()
plan = line[line.index("exact_plan"):line.index("ms;")]
print(f"arena block at {addr:#x} (mod 1 GiB: {addr % (1 << 30):#x}, mod 2 MiB: {addr % (2 << 20):#x})  {plan}")
PY
done
