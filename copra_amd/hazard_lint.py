#!/usr/bin/env python3
"""Lint of the compiled gfx950 kernels for ONE pattern: a matrix instruction (v_mfma_*) whose result is read too few wait states later
along SOME path of the control-flow graph.

Why it exists (round 4): hipcc of ROCm 7.2 left one instruction between a v_mfma_f64_4x4x4 and the v_mfma that reads its result as B
operand on the TAKEN side of a wave-uniform branch (the wait states had been counted on the fall-through side only); the hardware then
multiplies with the register's previous contents.  The emulator cannot see this, the GPU tests only where scheduling happens to expose it.

    python tools/mfma_hazard_lint.py --library [libcopra_hip.so]   the SHIPPED binary: code objects extracted and disassembled (seconds)
    python tools/mfma_hazard_lint.py [file.s ...]                  compiler output (default: compiles every translation unit with -S)

Wait states needed (what the compiler itself leaves in straight-line code of these kernels; s_nop N counts N + 1, every other
instruction 1):  f64 4x4x4 -> matrix operand A/B or vector ALU read: 6, LDS / memory store data: 9;  f64 16x16x4 -> 11 resp. 18;
result used as the accumulator (C operand) of the NEXT matrix instruction: 0 (the chained-accumulator case the hardware interlocks
for identical shapes; anything else as for A/B)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
NEED = {"4x4x4": (6, 9), "16x16x4": (11, 18)}
REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs(op):
    out = set()
    for m in REG.finditer(op):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def parse(path):
    """-> {kernel: (instructions, labels)}; instruction = (mnemonic, [operands], line number)"""
    kernels, cur, name = {}, None, None
    for ln, line in enumerate(open(path, errors="replace"), 1):
        s = line.split(";")[0].rstrip()
        if not s.strip():
            continue
        m = re.match(r"^(_Z\w+|copra_\w+):\s*$", s)
        if m:
            name, cur = m.group(1), ([], {})
            kernels[name] = cur
            continue
        if cur is None:
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            cur[1][m.group(1)] = len(cur[0])
            continue
        if not s.startswith("\t") or s.strip().startswith("."):
            continue
        parts = s.strip().split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        cur[0].append((parts[0], ops, ln))
        if parts[0] == "s_endpgm" and not any(i >= len(cur[0]) for i in cur[1].values()):
            pass
    return kernels


def parse_disassembly(path):
    """llvm-objdump -d of a gfx950 code object -> the same structure as parse(): branch targets become labels 'L<address>'"""
    kernels, cur, name = {}, None, None
    raw = []
    for ln, line in enumerate(open(path, errors="replace"), 1):
        m = re.match(r"^[0-9a-f]+ <(\w+)>:\s*$", line)
        if m:
            name, cur = m.group(1), ([], {})
            kernels[name] = cur
            raw.append((name, []))
            continue
        if cur is None or not line.startswith("\t"):
            continue
        m = re.match(r"^\t(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if not m:
            continue
        op, rest, addr = m.group(1), m.group(2), int(m.group(3), 16)
        ops = [o.strip() for o in rest.split(",")] if rest else []
        raw[-1][1].append((op, ops, ln, addr))
    for name, lst in raw:
        ins, labels = kernels[name]
        index = {a: i for i, (_, _, _, a) in enumerate(lst)}
        for op, ops, ln, addr in lst:
            if (op.startswith("s_cbranch") or op == "s_branch") and ops:
                imm = int(ops[0], 0) & 0xFFFF
                imm -= 0x10000 if imm & 0x8000 else 0
                tgt = addr + 4 + 4 * imm
                lab = "L%x" % tgt
                if tgt in index:
                    labels[lab] = index[tgt]
                ops = [lab]
            ins.append((op, ops, ln))
    return kernels


def lint_library(so_path):
    """the SHIPPED binary: extract the gfx950 code objects of libcopra_hip.so, disassemble, lint (seconds)"""
    with tempfile.TemporaryDirectory(prefix="copra_lint_") as tmp:  # (several MB of code objects and listings: removed on the way out)
        local = os.path.join(tmp, os.path.basename(so_path))
        with open(so_path, "rb") as f, open(local, "wb") as g:
            g.write(f.read())
        objdump = OBJDUMP
        subprocess.run([objdump, "--offloading", os.path.basename(local)], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        found, nobj, nmfma = [], 0, 0
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            nobj += 1
            dis = os.path.join(tmp, f + ".dis")
            with open(dis, "w") as out:
                subprocess.run([objdump, "-d", f], cwd=tmp, stdout=out, stderr=subprocess.DEVNULL, check=True)
            with open(dis, errors="replace") as listing:
                nmfma += sum(1 for line in listing if "\tv_mfma" in line)
            found += [(f,) + h for h in lint(dis, disassembly=True)]
        return found, nobj, nmfma


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def lint_code_object(path):
    """one gfx950 code object (e.g. what copra_batch_specialise compiled at run time) -> findings"""
    with tempfile.TemporaryDirectory(prefix="copra_lint_") as tmp:
        dis = os.path.join(tmp, os.path.basename(path) + ".dis")
        obj = os.path.abspath(path)
        with open(obj, "rb") as f:
            bundled = f.read(24).startswith(b"__CLANG_OFFLOAD_BUNDLE__")
        if bundled:  # (what hipcc --genco writes: host stub + device code object)
            raw = os.path.join(tmp, os.path.basename(path) + ".co")
            subprocess.run([os.path.join(os.path.dirname(OBJDUMP), "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + obj, "--output=" + raw], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            obj = raw
        with open(dis, "w") as out:
            subprocess.run([OBJDUMP, "-d", obj], stdout=out, stderr=subprocess.DEVNULL, check=True)
        return lint(dis, disassembly=True)


def successors(ins, labels, i):
    op, ops, _ = ins[i]
    if op == "s_endpgm":
        return []
    if op == "s_branch":
        return [labels[ops[0]]] if ops and ops[0] in labels else []
    if op.startswith("s_cbranch"):
        nxt = [i + 1] if i + 1 < len(ins) else []
        return nxt + ([labels[ops[0]]] if ops and ops[0] in labels else [])
    return [i + 1] if i + 1 < len(ins) else []


def lint(path, disassembly=False):
    found = []
    for name, (ins, labels) in (parse_disassembly(path) if disassembly else parse(path)).items():
        for i, (op, ops, ln) in enumerate(ins):
            if not op.startswith("v_mfma"):
                continue
            shape = next((k for k in NEED if k in op), None)
            if shape is None or not ops:
                continue
            need_alu, need_mem = NEED[shape]
            dst = regs(ops[0])
            # walk every path until the result is overwritten or enough wait states have passed
            stack, seen = [(j, 0, False) for j in successors(ins, labels, i)], {}
            while stack:
                j, waited, crossed = stack.pop()
                if waited >= need_mem or seen.get(j, 1 << 30) <= waited:
                    continue
                seen[j] = waited
                o2, ops2, ln2 = ins[j]
                srcs = ops2[1:] if (o2.startswith("v_") or o2.startswith("ds_read") or o2.startswith("global_load") or o2.startswith("buffer_load")) and not o2.startswith("v_cmp") else ops2
                if o2.startswith("ds_write") or o2.startswith("global_store") or o2.startswith("buffer_store") or o2.startswith("v_cmp") or o2.startswith("v_readlane"):
                    srcs = ops2 if not o2.startswith("v_readlane") else ops2[1:]
                reads = set().union(*[regs(x) for x in srcs]) if srcs else set()
                if reads & dst:
                    is_mem = o2.startswith(("ds_", "global_", "buffer_", "flat_", "scratch_"))
                    need = need_mem if is_mem else need_alu
                    if o2.startswith("v_mfma") and len(ops2) >= 4 and regs(ops2[3]) & dst and not (regs(ops2[1]) | regs(ops2[2])) & dst:
                        need = 0  # accumulator chain
                    if waited < need and crossed:
                        found.append((name, ln, op, ln2, o2, waited, need))
                    continue  # (the first reader on this path decides)
                if ops2 and (o2.startswith("v_") or o2.startswith("ds_read") or "load" in o2) and regs(ops2[0]) >= dst:
                    continue  # overwritten
                step = int(ops2[0]) + 1 if o2 == "s_nop" and ops2 else 1
                succ = successors(ins, labels, j)
                for t in succ:
                    stack.append((t, waited + step, crossed or t != j + 1 or o2.startswith("s_cbranch")))
    return found


def main():
    files = sys.argv[1:]
    tmp = None
    if files and files[0] == "--library":
        so = files[1] if len(files) > 1 else os.path.join(CSRC, "libcopra_hip.so")
        hits, nobj, nmfma = lint_library(so)
        for f, name, ln, op, ln2, o2, waited, need in hits:
            print("%s: %s\n    line %d %s -> line %d %s after %d wait state(s) on a path across a branch (needs %d)" % (f, name[:90], ln, op, ln2, o2, waited, need))
        print("%s: %d code object(s), %d matrix instructions, %d finding(s)" % (os.path.basename(so), nobj, nmfma, len(hits)))
        return 1 if hits else 0
    if not files:
        tmp = tempfile.mkdtemp(prefix="copra_lint_")
        srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
        procs = []
        for f in srcs:
            out = os.path.join(tmp, f[:-4] + ".s")
            files.append(out)
            procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-pass-failed", '-DCOPRA_SRC_HASH="lint"',
                                           "--cuda-device-only", "-S", "-o", out, f], cwd=CSRC, stderr=subprocess.DEVNULL))
        for p in procs:
            p.wait()
        import atexit
        import shutil
        atexit.register(shutil.rmtree, tmp, True)
    total = 0
    for f in files:
        hits = lint(f)
        total += len(hits)
        for name, ln, op, ln2, o2, waited, need in hits:
            print("%s: %s\n    line %d %s -> line %d %s after %d wait state(s) on a path across a branch (needs %d)" % (os.path.basename(f), name[:90], ln, op, ln2, o2, waited, need))
    print("%d file(s), %d finding(s)" % (len(files), total))
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
