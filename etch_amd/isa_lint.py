"""ISA lint for the kernels that issue memory operations from inline asm with hand-counted waits.

    python -m etch_amd.isa_lint etch_amd/csrc/so3conv_x.hip [-D...]      (cross-compiles for gfx950, no GPU needed)

Why.  `asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : ...)` / `global_load_dwordx4 %0, ...` return their data ASYNCHRONOUSLY, but the compiler
believes an asm statement's output is valid when the statement ends.  Between the load and the `s_waitcnt` asm that the source pairs it with, the
register allocator is free to copy the output to another register (live-range splitting under register pressure), to read it, or to reuse it: the
copy then captures stale data -- for a 16-byte LDS read typically in lanes 48 .. 63, the quarter of the wave that the LDS returns last -- and whether it
does depends on the build (register pressure) and on the run (LDS / L2 timing).  This is the mechanism behind profiles/r04_x32_cin32_miscompile.txt
(profiles/r05_x32_cin32_root_cause.txt).  The lint walks the final ISA of every kernel, models the two wait counters the way the hardware
defines them, and reports every instruction that touches a register while a load into it may still be in flight.

Counter model (gfx950 = gfx9 counters):
  vmcnt   vector-memory loads, LDS-direct loads (global_load_lds_*) and stores.  In order WITHIN each of the three kinds, out of order between them
          (measured in round 4): an operation with y younger operations of its OWN kind outstanding is complete after `vmcnt(N)` iff N <= y.
  lgkmcnt LDS operations (in order) and scalar memory reads (out of order: complete only after lgkmcnt(0)).
Control flow: kernels are walked linearly; every backward branch re-walks its loop body once with the state of the loop's end (a pending load that
crosses the back edge is seen by the first instructions of the next iteration).  Forward branches are walked through (conservative for hazards
on the fall-through path; the skipped path is covered by the target's own walk).
"""
import os
import re
import subprocess
import sys
import tempfile

_REG = re.compile(r"\b([vas])(\d+)\b|\b([vas])\[(\d+):(\d+)\]")
_WAIT = re.compile(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)")


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), i))
    return out


def _kind(op):
    """-> (counter, kind, has_register_destination)"""
    if op.startswith(("global_load_lds", "buffer_load_lds")) or (op.startswith("buffer_load") and op.endswith("_lds")):
        return "vm", "lds_dma", False
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load", "global_atomic", "buffer_atomic", "flat_atomic")):
        return "vm", "load", True
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
        return "vm", "store", False
    if op.startswith(("s_load", "s_buffer_load", "s_scratch_load")):
        return "lgkm", "smem", True
    if op.startswith("ds_"):
        has_dst = not op.startswith(("ds_write", "ds_store", "ds_nop", "ds_gws"))
        return "lgkm", "lds", has_dst
    return None, None, False


class _State:
    def __init__(self):
        self.q = {"vm": [], "lgkm": []}          # outstanding operations, oldest first: dicts(kind, dst, line, asm)

    def copy(self):
        s = _State()
        s.q = {k: list(v) for k, v in self.q.items()}
        return s

    def wait(self, counter, n):
        q = self.q[counter]
        keep = []
        # an operation stays possibly-outstanding iff (younger operations of its own kind) + 1 <= n; scalar reads: out of order, iff 1 <= n
        younger = {}
        for e in reversed(q):
            y = younger.get(e["kind"], 0)
            alive = (1 <= n) if e["kind"] == "smem" else (y + 1 <= n)
            if alive:
                keep.append(e)
            younger[e["kind"]] = y + 1
        self.q[counter] = list(reversed(keep))

    def pending(self):
        regs = {}
        for q in self.q.values():
            for e in q:
                for r in e["dst"]:
                    regs[r] = e
        return regs


def lint_asm(text, only_asm_loads=False):
    """-> list of findings: dict(kernel, line_no, instr, reg, load_line, load_instr, load_in_asm)"""
    findings = []
    lines = text.split("\n")
    # kernels = from a global function label to its s_endpgm
    i = 0
    label_re = re.compile(r"^([A-Za-z_.$][\w.$]*):")
    while i < len(lines):
        m = label_re.match(lines[i])
        if not (m and m.group(1).startswith("_Z")):
            i += 1
            continue
        kernel = m.group(1)
        body = []
        j = i + 1
        while j < len(lines) and not lines[j].lstrip().startswith(".end_amdhsa_kernel") and not (label_re.match(lines[j]) and label_re.match(lines[j]).group(1).startswith("_Z")):
            body.append((j + 1, lines[j]))
            if lines[j].strip().startswith("s_endpgm") and not any(b[1].strip().startswith(("s_cbranch", "s_branch")) for b in body[-1:]):
                pass
            j += 1
        findings += _lint_kernel(kernel, body, only_asm_loads)
        i = j
    return findings


def _lint_kernel(kernel, body, only_asm_loads):
    # flatten: (line_no, instr text, in_asm), label -> index
    prog, labels, in_asm = [], {}, False
    for no, raw in body:
        s = raw.split(";")[0].strip() if ";;#" not in raw else raw.strip()
        if ";;#ASMSTART" in raw:
            in_asm = True
            continue
        if ";;#ASMEND" in raw:
            in_asm = False
            continue
        if not s or s.startswith((".", "//")):
            m = re.match(r"^(\.?[\w.$]+):", raw.strip())
            if m:
                labels[m.group(1)] = len(prog)
            continue
        m = re.match(r"^(\.?[\w.$]+):", s)
        if m:
            labels[m.group(1)] = len(prog)
            continue
        prog.append((no, s, in_asm))
    findings, seen = [], set()

    def walk(lo, hi, st):
        for k in range(lo, hi):
            no, s, ia = prog[k]
            op = s.split()[0]
            operands = s[len(op):]
            if op == "s_waitcnt":
                for cm in _WAIT.finditer(operands):
                    if cm.group(1) == "vmcnt":
                        st.wait("vm", int(cm.group(2)))
                    elif cm.group(1) == "lgkmcnt":
                        st.wait("lgkm", int(cm.group(2)))
                if not _WAIT.search(operands):            # raw immediate: treat as a full wait of both (never emitted by this code base)
                    st.wait("vm", 0)
                    st.wait("lgkm", 0)
                continue
            touched = _regs(operands)
            pend = st.pending()
            for r in touched & set(pend):
                e = pend[r]
                if only_asm_loads and not e["asm"]:
                    continue
                key = (no, r)
                if key not in seen:
                    seen.add(key)
                    findings.append(dict(kernel=kernel, line_no=no, instr=s, reg="%s%d" % r, load_line=e["line"], load_instr=e["instr"], load_in_asm=e["asm"]))
            counter, kind, has_dst = _kind(op)
            if counter:
                dst = set()
                if has_dst:
                    first = operands.split(",")[0]
                    dst = _regs(first)
                st.q[counter].append(dict(kind=kind, dst=dst, line=no, instr=s, asm=ia))
        return st

    st = walk(0, len(prog), _State())
    # back edges: re-walk each loop body once with the state at the branch
    for k, (no, s, ia) in enumerate(prog):
        op = s.split()[0]
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= k:
                st_end = walk(0, k + 1, _State())
                walk(labels[tgt], k + 1, st_end.copy())
    return findings


MFMA_WINDOW = 24      # instructions: more than the 19 wait states the longest MFMA needs before a VALU may read its result (CDNA3 ISA 4.5)


def lint_mfma_asm_reads(text):
    """Second hazard class (round 5, found the hard way in csrc/mhsa_layer.hip): a VALU instruction written as INLINE ASM that reads a register a
    v_mfma wrote a few instructions earlier.  The compiler's hazard recogniser inserts the required wait states (s_nop) only in front of instructions
    it emitted itself, and the hardware does not interlock: the asm instruction reads the accumulator's OLD contents.  Safe patterns: a compiler-emitted
    instruction reads the result first (it gets the wait states; MFMAs retire in order, so everything older is complete too), or an asm block with
    >= 16 wait states of s_nop sits between (the guard in ml_attention).  -> findings: dict(kernel, line_no, instr, reg, mfma_line, distance)"""
    findings = []
    kernel, in_asm, idx = None, False, 0
    fresh = {}            # register -> (instruction index of the v_mfma that wrote it, line number)
    srcc = {}             # register -> (instruction index, line) of the youngest v_mfma that reads it as its accumulator INPUT (SrcC != vDst):
                          # an inline-asm VALU WRITE to it while that MFMA is still reading is the write-after-read form of the same blind spot
    label_re = re.compile(r"^([A-Za-z_.$][\w.$]*):")
    for no, raw in enumerate(text.split("\n"), 1):
        m = label_re.match(raw)
        if m and m.group(1).startswith("_Z"):
            kernel, fresh, srcc, idx = m.group(1), {}, {}, 0
            continue
        if ";;#ASMSTART" in raw:
            in_asm, nops = True, 0
            continue
        if ";;#ASMEND" in raw:
            in_asm = False
            if nops >= 16:
                fresh, srcc = {}, {}
            continue
        s = raw.split(";")[0].strip()
        if kernel is None or not s or s.startswith((".", "//")) or label_re.match(s):
            continue
        idx += 1
        op = s.split()[0]
        operands = s[len(op):]
        if op == "s_nop":
            if in_asm:
                nops += int(operands.strip() or 0) + 1
            continue
        parts = operands.split(",")
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            dst = _regs(parts[0])
            for r in dst:
                fresh[r] = (idx, no)
                srcc.pop(r, None)
            if len(parts) >= 4:
                for r in _regs(parts[3]) - dst:
                    srcc[r] = (idx, no)
            continue
        if not fresh and not srcc:
            continue
        has_dst = op.startswith("v_") or _kind(op)[2]
        srcs = _regs(",".join(parts[1:])) if has_dst else _regs(operands)
        hit = [r for r in srcs if r in fresh]
        if hit and in_asm and op.startswith("v_"):
            for r in hit:
                if idx - fresh[r][0] < MFMA_WINDOW:
                    findings.append(dict(kernel=kernel, line_no=no, instr=s, reg="%s%d" % r, mfma_line=fresh[r][1], distance=idx - fresh[r][0]))
        elif hit:       # a compiler-emitted read: it carries the wait states; everything at least as old is complete
            newest = max(fresh[r][0] for r in hit)
            fresh = {r: v for r, v in fresh.items() if v[0] > newest}
        if has_dst:     # overwritten by something else: no longer an MFMA result (write-after-MFMA hazards of compiler-emitted writes are the compiler's)
            wr = _regs(parts[0])
            for r in wr:
                fresh.pop(r, None)
            over = [r for r in wr if r in srcc]
            if over and in_asm and op.startswith("v_"):
                for r in over:
                    if idx - srcc[r][0] < MFMA_WINDOW // 2:
                        findings.append(dict(kernel=kernel, line_no=no, instr=s, reg="%s%d" % r, mfma_line=srcc[r][1], distance=idx - srcc[r][0], kind="write-after-read"))
            elif over:  # a compiler-emitted write got the wait states: that MFMA and every older one are past their reads
                newest = max(srcc[r][0] for r in over)
                srcc = {r: v for r, v in srcc.items() if v[0] > newest}
    return findings


def compile_to_asm(src, defines=(), extra=()):
    here = os.path.dirname(os.path.abspath(__file__))
    from . import build as B

    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [B.hipcc()] + B.BASE + list(defines) + list(extra) + ["--cuda-device-only", "-S", src, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=here)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stderr)
        return open(out).read()


def main(argv):
    """python -m etch_amd.isa_lint FILE.hip|FILE.s [-Dflags] [--all]
    default: only loads issued from inline asm (the ones whose waits are hand-written); --all also lists compiler-issued loads, for which the
    linear walk through forward branches over-reports (the compiler's own waits sit on paths the walk does not follow)."""
    src = argv[0]
    text = open(src).read() if src.endswith(".s") else compile_to_asm(os.path.abspath(src), [a for a in argv[1:] if a.startswith("-D")])
    f = lint_asm(text, only_asm_loads="--all" not in argv)
    by_kernel = {}
    for x in f:
        by_kernel.setdefault(x["kernel"], []).append(x)
    for k, v in by_kernel.items():
        print(f"{k}: {len(v)} finding(s)")
        for x in v[:40]:
            print(f"   line {x['line_no']}: `{x['instr']}` touches {x['reg']} while `{x['load_instr']}` (line {x['load_line']}, {'inline asm' if x['load_in_asm'] else 'compiler'}) may be in flight")
    print(f"{len(f)} finding(s) in {len(by_kernel)} kernel(s)")
    g = lint_mfma_asm_reads(text)
    for x in g[:40]:
        print(f"   {x['kernel']} line {x['line_no']}: inline-asm `{x['instr']}` reads {x['reg']} {x['distance']} instructions after the v_mfma of line {x['mfma_line']} wrote it (no wait states)")
    print(f"{len(g)} inline-asm VALU read(s) of a fresh MFMA result")
    return 1 if f or g else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
