#!/usr/bin/env python3
"""Static instruction mix of one kernel and of each of its loops, from the gfx950 assembly hipcc emits
(compile-only, works without a GPU):

    python tools/isa_mix.py phant_amd/csrc/mpt_verify_v3.hip hash_deep_kernel

Backs the "180 VALU per Keccak round, the instruction minimum" figure of DESIGN.md section 7 with something
checkable: the round loop's VALU count by opcode, its scalar overhead, where the loads and waits sit."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, kernel = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as d:
        s_path = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I",
                               os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "phant_amd", "csrc"), "-S", "--cuda-device-only", "-o", s_path, src],
                              stderr=subprocess.DEVNULL)
        lines = open(s_path).read().splitlines()
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(kernel), l)][0]
    end = [i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i]][0]
    body = lines[start:end]

    def ops(seg):
        return [x.strip().split()[0] for x in seg if x.startswith("\t") and not x.strip().startswith((".", ";"))]

    def summary(o):
        c = collections.Counter(o)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        return (f"{len(o)} instr: VALU {valu} (bitop3 {c.get('v_bitop3_b32', 0)}, alignbit {c.get('v_alignbit_b32', 0)}, "
                f"alignbyte {c.get('v_alignbyte_b32', 0)}, xor {c.get('v_xor_b32_e32', 0) + c.get('v_xor_b32_e64', 0)}, "
                f"cndmask {c.get('v_cndmask_b32_e32', 0) + c.get('v_cndmask_b32_e64', 0)}, "
                f"cmp {sum(v for k, v in c.items() if k.startswith('v_cmp'))}, mov {c.get('v_mov_b32_e32', 0)}), "
                f"SALU {sum(v for k, v in c.items() if k.startswith('s_') and k not in ('s_nop', 's_waitcnt'))}, "
                f"s_waitcnt {c.get('s_waitcnt', 0)}, s_nop {c.get('s_nop', 0)}, "
                f"global_load {sum(v for k, v in c.items() if k.startswith('global_load'))}, "
                f"global_store {sum(v for k, v in c.items() if k.startswith('global_store'))}")

    print(f"{lines[start].split(':')[0]}\nwhole kernel: {summary(ops(body))}")
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    for i, l in enumerate(body):
        m = re.search(r"\s(s_cbranch\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(2) in labels and labels[m.group(2)] < i:
            print(f"loop {m.group(2)} (lines {labels[m.group(2)]}..{i} of the kernel): {summary(ops(body[labels[m.group(2)]:i + 1]))}")


if __name__ == "__main__":
    main()
