"""Static check of the hand-issued scalar loads of k_piece_grad (csrc/minco_kernels.h TabRow).  The basis-table rows are
requested from inline asm (`s_load_dwordx16` into an "=&s" output) and awaited in a LATER asm statement (`s_waitcnt lgkmcnt(0)`):
the compiler does not know the registers are still in flight in between, the hardware does not interlock scalar registers
against a pending scalar load, so any instruction that reads or writes those SGPRs before the wait -- a copy, a spill through
v_writelane, a reuse as an address -- would silently corrupt a table row.  It holds with this compiler and these flags; this
test makes it a checked property of the build instead of an observation: the unit is compiled to assembly with the product's
flags and the destination registers of every 8- / 16-dword scalar load must be untouched until the next `s_waitcnt lgkmcnt(0)`
(scalar loads return out of order: only a full wait counts)."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sgprs(text):
    regs = set()
    for lo, hi in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        regs.update(range(int(lo), int(hi) + 1))
    for r in re.findall(r"\bs(\d+)\b", text):
        regs.add(int(r))
    return regs


def test_no_instruction_touches_the_destination_of_a_scalar_load_before_its_wait():
    from allocnet_amd import build as b
    unit, uflags = b.UNITS[0]
    assert unit == "piece_grad_unit.hip"
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "pg.s")
        cmd = [b.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include")] + \
            b.probe_flags(b.MFMA_VGPR_FORM) + uflags + ["--cuda-device-only", "-S", os.path.join(b.SRC_DIR, unit), "-o", out]
        res = subprocess.run(cmd, capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-2000:]
        asm = open(out).read().splitlines()
    kernels = loads = 0
    name, pending = None, []          # pending: (destination registers, the load's line)
    for ln, line in enumerate(asm):
        m = re.match(r"^(_ZN4anet\w+):", line)
        if m:
            name, pending = m.group(1), []
            kernels += "k_piece_grad" in name
            continue
        if name is None or "k_piece_grad" not in name:
            continue
        code = line.split(";")[0].strip()
        if not code or code.endswith(":") or code.startswith("."):
            continue
        op, _, rest = code.partition(" ")
        if op == "s_waitcnt" and "lgkmcnt(0)" in rest:
            pending = []
            continue
        if op == "s_endpgm":
            name = None
            continue
        touched = _sgprs(rest)
        for dest, at in pending:
            assert not (touched & dest), f"{name}: line {ln + 1} `{code}` touches s{sorted(touched & dest)} of the load in flight at line {at + 1} `{asm[at].strip()}`"
        # (8- and 16-dword loads: the table rows -- and the kernel arguments, which pass by construction.  Narrower loads are
        #  the compiler's own and are not followed here: a linear scan across its branches would see false conflicts.)
        if op in ("s_load_dwordx8", "s_load_dwordx16"):
            dest = _sgprs(rest.split(",")[0])
            pending.append((dest, ln))
            loads += op == "s_load_dwordx16"
    assert kernels >= 9 and loads >= 18        # every instantiation was seen, and the hand-issued 16-dword loads with them
