#!/usr/bin/env python3
"""Load / wait structure of one kernel instantiation, from the compiler's ISA (no GPU needed).

    python tools/isa_load_trace.py minco_kernels.h 'anet::k_minco_propagate_axis<3, 16, true, 2>(anet::PropArgs)'
    python tools/isa_load_trace.py lbfgs_kernels.h 'anet::k_lbfgs_update_wave<8, 1>(anet::LbfgsArgs)'

Prints the kernel as a sequence of  L<n> (n vector-memory loads back to back),  W<k> (s_waitcnt vmcnt(k)),
S (store) and the number of other instructions in between.  A latency-bound kernel (one wave per SIMD) pays a full
L2 round trip for every `L.. <few> W0` pair: this view is what showed ~100 of them in the small-batch propagate
kernel and a readfirstlane behind every history load of the L-BFGS update.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    header, inst = sys.argv[1], sys.argv[2]
    m = re.match(r"\s*(.*?)\((.*)\)\s*$", inst)
    name, params = m.group(1), m.group(2)
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.hip")
        with open(src, "w") as fh:
            fh.write('#include "%s"\n' % header)
            if "<" in name:  # (kernels that are not templates are emitted by the include alone)
                fh.write("template __global__ void %s(%s);\n" % (name, params))
        asm = os.path.join(d, "t.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "allocnet_amd", "csrc"), "--cuda-device-only", "-S", src, "-o", asm],
                       check=True, stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    base = name.split("<")[0].split("::")[-1]
    starts = [k for k, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % base, l)]
    for i in starts:
        j = i
        while not lines[j].strip().startswith("s_endpgm"):
            j += 1
        ins = [x.strip() for x in lines[i:j] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
        out, alu = [], 0
        for x in ins:
            op = x.split()[0]
            if "load" in op and not op.startswith("s_"):
                tok = "L"
            elif op == "s_waitcnt" and "vmcnt" in x:
                tok = "W" + re.search(r"vmcnt\((\d+)\)", x).group(1)
            elif op.startswith(("global_store", "flat_store", "scratch_store")):
                tok = "S"
            else:
                alu += 1
                continue
            if alu:
                out.append(str(alu))
                alu = 0
            out.append(tok)
        out.append(str(alu))
        s = re.sub(r"(L )+", lambda mm: "L%d " % (len(mm.group(0)) // 2), " ".join(out))
        print(lines[i].split(":")[0])
        print(len(ins), "instructions")
        print(s)


if __name__ == "__main__":
    main()
