"""The test-only RCCL stand-in (tests/rccl_loopback) covers exactly what the product library binds: every `U nccl*`
symbol of libpfd_hip.so is defined by librccl_loopback.so, so that LD_PRELOAD leaves no call going to the real library
by accident.  (CPU: symbol tables only; the ranks-on-one-GPU runs are tests/test_gpu_rccl_loopback.py.)"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_DIR = os.path.join(ROOT, "tests", "rccl_loopback")


def _symbols(path, kind):
    out = subprocess.run(["nm", "-D", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if f" {kind} " in ln and ln.split()[-1].startswith("nccl")}


def test_stand_in_defines_every_nccl_symbol_the_library_binds():
    subprocess.check_call(["make", "-C", SHIM_DIR], stdout=subprocess.DEVNULL)
    from pyflwdir_amd.build import build

    lib = build()
    wanted = _symbols(lib, "U")
    have = _symbols(os.path.join(SHIM_DIR, "librccl_loopback.so"), "T")
    assert wanted and wanted <= have, sorted(wanted - have)
    # and the product does not know the stand-in exists
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pyflwdir_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                with open(os.path.join(dirpath, f), errors="replace") as fh:
                    assert "rccl_loopback" not in fh.read().lower(), f
