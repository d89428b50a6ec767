"""Static check of the gfx950 ISA of csrc/sht_rec.hip for the hazard the compiler cannot see inside inline assembly:
a VGPR written by a VALU instruction must not be read as the DPP source (src0 of a `*_dpp` instruction) within the next
two wait states.  Counts, per `*_dpp` instruction, the instructions between it and the nearest preceding VALU write of any
register of its DPP source; `s_nop N` counts N + 1.  Prints the worst case per kernel; exit 1 if any distance is < 2.

    python scripts/dev/check_dpp_hazard.py [path/to/sht_rec-hip-amdgcn-amd-amdhsa-gfx950.s]
Without an argument the file is compiled (hipcc --save-temps) into a temporary directory."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(path):
    text = open(path).read()
    worst_all = 99
    for km in re.finditer(r"^(_ZN3pxm\w+):.*?s_endpgm", text, re.S | re.M):
        name, body = km.group(1), km.group(0)
        if "_dpp" not in body:
            continue
        lines = [ln.strip() for ln in body.split("\n")]
        insts = [ln for ln in lines if ln and not ln.startswith((";", ".", "_ZN")) and not ln.endswith(":")]
        last_write = {}  # vgpr -> index (in wait states) of its last VALU write
        pos = 0
        worst, n = 99, 0
        for ins in insts:
            op = ins.split()[0]
            if op == "s_nop":
                pos += int(ins.split()[1]) + 1
                continue
            args = ins[len(op):].split(",")
            if "_dpp" in op:
                src = regs(args[1].split()[0]) if len(args) > 1 else set()
                for r in src:
                    if r in last_write:
                        worst = min(worst, pos - last_write[r] - 1)
                n += 1
            if op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                for r in regs(args[0].split()[0]):
                    last_write[r] = pos
            pos += 1
        print(f"{name[:70]:70s} {n:5d} DPP instructions, nearest VALU write of a DPP source: {worst if worst < 99 else 'none'} instructions before")
        worst_all = min(worst_all, worst)
    return worst_all


if __name__ == "__main__":
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        d = tempfile.mkdtemp(prefix="pxm_dpp_")
        src = os.path.join(ROOT, "pxmcmc_amd", "csrc", "sht_rec.hip")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-c", src, "-o", os.path.join(d, "o.o"),
                        "--save-temps=obj"], check=True, cwd=os.path.dirname(src))
        path = os.path.join(d, "sht_rec-hip-amdgcn-amd-amdhsa-gfx950.s")
    w = check(path)
    print("worst distance:", w)
    sys.exit(0 if w >= 2 else 1)
