"""Scans a gfx950 assembly listing for the hazard found in round 4 (profiles/r04_mfma_operand_war.txt): an instruction that WRITES a VGPR of an MFMA's
A / B operand within WINDOW instructions after that MFMA.  usage: python mfma_war_scan.py file.s [kernel-name-substring] [window]"""
import re
import sys

path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
window = int(sys.argv[3]) if len(sys.argv) > 3 else 16
txt = open(path).read()
kernels = re.findall(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, flags=re.S | re.M)


def regs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def dst_regs(line):
    parts = line.split(None, 1)
    if len(parts) < 2:
        return set()
    op = parts[0]
    if op.startswith(("s_", "ds_write", "global_store", "buffer_store", "global_load_lds", "v_cmp", ";")):
        return set()
    return regs(parts[1].split(",")[0])


for name, body in kernels:
    if sub not in name:
        continue
    lines = [l.strip() for l in body.split("\n")]
    lines = [l for l in lines if l and not l.startswith((";", ".")) and not l.endswith(":")]
    hits = 0
    for n, l in enumerate(lines):
        if not l.startswith("v_mfma"):
            continue
        ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
        ab = regs(ops[1]) | regs(ops[2])
        for k in range(1, window + 1):
            if n + k >= len(lines):
                break
            w = dst_regs(lines[n + k]) & ab
            if w and not lines[n + k].startswith("v_mfma"):
                hits += 1
                if hits <= 12:
                    print(f"{name[:60]}: +{k}: {l}   <-   {lines[n + k]}")
                break
    print(name[:80], "MFMA operand overwrites within", window, "instructions:", hits)
