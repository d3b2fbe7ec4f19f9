#!/usr/bin/env python3
"""Build-time lint of the gfx950 code object inside libsdrxgpu.so for the wait states that the kernels' INLINE ASSEMBLY relies on.

The compiler spaces dependent instructions itself (GCNHazardRecognizer) -- but only those it can see.  kernels_mfma*.hpp hold two
kinds of inline assembly whose hazards it cannot see, and whose spacing so far rested on today's instruction order:

  R1  `global_store_dwordx4 ... nt` (k_stage12m's streaming scratch stores): on gfx940 and later a VMEM store of more than 8
      bytes reads its data registers late; a VALU instruction that overwrites them needs 2 wait states behind the store.  (Round 5
      lost one set of a 2 304-channel bank to this before the `s_nop 1` went into the asm statement.)
  R2  `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0] neg_lo|neg_hi` (the P -+ Q combinations behind the f32 MFMAs): a VALU read of a
      register an MFMA writes needs wait states between the two.  For the 8-pass v_mfma_f32_16x16x4_f32 the compiler spaces its OWN
      such reads by 10 (k_stage12m6's compiler-generated combinations: `s_setprio 0; s_nop 8` = 1 + 9 between the MFMA and the
      v_sub_f32 that reads it); the kernels' inline assembly was written to 12 (8 passes + 3, + 1 for gfx950: the XDL row of LLVM's
      GFX940_*_PassWriteVgprVALUMemExpReadWaitStates), and 12 is what the lint asks for.  The kernels rely on the next phase's
      operand reads standing in between, and on `s_nop 7; s_nop 3` where there are none.

Both are checked on the DISASSEMBLY, over the control-flow graph of every kernel (a wait state is counted only if it lies on every
path): the lint walks back from each inline v_pk_add_f32 to the nearest v_mfma that writes one of its sources, and forward from each
nt store to the nearest VALU write of its data registers, counting one wait state per instruction and N + 1 per `s_nop N`, as the
hazard recogniser does.  Run by __graft_entry__.build() and tests/test_hazard_lint.py; exit status 1 and one line per violation.

  tools/hazard_lint.py [path/to/libsdrxgpu.so] [--verbose]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MFMA_READ_WAIT = 12      # v_mfma_f32_16x16x4_f32 (8 passes) -> VALU read of its result, gfx950
STORE_OVERWRITE_WAIT = 2  # VMEM store > 8 bytes -> VALU overwrite of its data registers, gfx940+

_reg = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")
_line = re.compile(r"^\t(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_target = re.compile(r"<[^>]*\+0x([0-9a-fA-F]+)>\s*$|<([^>+]*)>\s*$")


def disassemble(lib):
    """-> text of `llvm-objdump -d` of the gfx950 code object bundled in the shared library"""
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in os.listdir(td) if "gfx950" in f]
        if not co:
            raise SystemExit("hazard_lint: no gfx950 code object in " + lib)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", os.path.join(td, co[0])],
                              check=True, capture_output=True, text=True).stdout


def regs(text):
    """set of (file, index) the operand text names"""
    out = set()
    for m in _reg.finditer(text):
        if m.group(2) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(1), i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


class Ins:
    __slots__ = ("addr", "op", "args", "dst", "src", "target", "wait")

    def __init__(self, addr, op, args, tail):
        self.addr, self.op, self.args = addr, op, args
        parts = [p.strip() for p in args.split(",")] if args else []
        # destination = first operand of a VALU / MFMA instruction (v_cmp & co. write SGPRs / VCC: no VGPR named there)
        self.dst = regs(parts[0]) if parts and op.startswith("v_") else set()
        self.src = regs(",".join(parts[1:])) if op.startswith("v_") else regs(args)
        if op.startswith(("v_fmac", "v_mac")):
            self.src |= self.dst                       # (the accumulator is read as well as written)
        self.target = None
        if op.startswith(("s_cbranch", "s_branch")):
            m = _target.search(tail)
            if m and m.group(1):
                self.target = int(m.group(1), 16)       # offset from the function's start
        self.wait = (int(args.split()[0], 0) + 1) if op == "s_nop" else 1


def parse(text):
    """-> {kernel name: [Ins]} with branch targets resolved to instruction indices (Ins.target)"""
    funcs, cur, base = {}, None, 0
    for ln in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", ln)
        if m:
            base = int(m.group(1), 16)
            cur = funcs.setdefault(m.group(2), [])
            cur_base = base
            continue
        m = _line.match(ln)
        if m and cur is not None:
            ins = Ins(int(m.group(3), 16), m.group(1), m.group(2), ln)
            if ins.target is not None:
                ins.target += cur_base
            cur.append(ins)
    for name, code in funcs.items():
        at = {i.addr: k for k, i in enumerate(code)}
        for i in code:
            if i.target is not None:
                i.target = at.get(i.target)            # None: outside the function (never for these kernels)
    return funcs


def cfg(code):
    succ = [[] for _ in code]
    pred = [[] for _ in code]
    for k, i in enumerate(code):
        if i.target is not None:
            succ[k].append(i.target)
        if not (i.op in ("s_branch", "s_endpgm", "s_setpc_b64")) and k + 1 < len(code):
            succ[k].append(k + 1)
        for s in succ[k]:
            pred[s].append(k)
    return succ, pred


def is_inline_pk_add(i):
    return i.op == "v_pk_add_f32" and "op_sel:[0,1]" in i.args and "op_sel_hi:[1,0]" in i.args and ("neg_lo:[0,1]" in i.args or "neg_hi:[1,0]" in i.args)


def is_nt_store(i):
    return i.op in ("global_store_dwordx4", "global_store_dwordx3") and re.search(r"\bnt\b", i.args) is not None


def lint_function(name, code, verbose=False):
    succ, pred = cfg(code)
    bad, n_pk, n_st, closest_mfma, closest_st = [], 0, 0, None, None
    for k, ins in enumerate(code):
        if is_inline_pk_add(ins):
            n_pk += 1
            want = ins.src                     # both sources are MFMA accumulators
            # walk back over every path; state = (index, wait states collected between it and the read, registers still open)
            seen, stack = {}, [(p, 0, frozenset(want)) for p in pred[k]]
            while stack:
                j, w, open_ = stack.pop()
                if w >= MFMA_READ_WAIT or not open_:
                    continue
                key = (j, open_)
                if seen.get(key, 1 << 30) <= w:
                    continue
                seen[key] = w
                c = code[j]
                hit = c.dst & open_
                if hit and c.op.startswith("v_mfma"):
                    closest_mfma = w if closest_mfma is None else min(closest_mfma, w)
                    bad.append(f"{name}: v_pk_add_f32 at {ins.addr:#x} reads {sorted(hit)[0][0]}{sorted(hit)[0][1]} only {w} wait state(s) "
                               f"behind {c.op} at {c.addr:#x} (needs {MFMA_READ_WAIT})")
                    continue
                if hit:
                    open_ = open_ - hit        # another instruction defines them on this path: that is the compiler's to space
                for p in pred[j]:
                    stack.append((p, w + c.wait, open_))
        elif is_nt_store(ins):
            n_st += 1
            parts = [p.strip() for p in ins.args.split(",")]
            data = regs(parts[1])
            seen, stack = {}, [(s, 0) for s in succ[k]]
            while stack:
                j, w = stack.pop()
                if w >= STORE_OVERWRITE_WAIT or seen.get(j, 1 << 30) <= w:
                    continue
                seen[j] = w
                c = code[j]
                if c.op.startswith("v_") and (c.dst & data):
                    closest_st = w if closest_st is None else min(closest_st, w)
                    bad.append(f"{name}: {c.op} at {c.addr:#x} overwrites a data register of the nt store at {ins.addr:#x} after {w} wait state(s) "
                               f"(needs {STORE_OVERWRITE_WAIT})")
                    continue
                for s in succ[j]:
                    stack.append((s, w + c.wait))
    if verbose and (n_pk or n_st):
        print(f"  {name[:100]}: {n_pk} inline v_pk_add_f32, {n_st} nt stores, {len(bad)} violation(s)")
    return bad, n_pk, n_st


def lint(lib, verbose=False):
    funcs = parse(disassemble(lib))
    bad, n_pk, n_st, n_k = [], 0, 0, 0
    for name, code in funcs.items():
        b, p, s = lint_function(name, code, verbose)
        bad += b
        n_pk += p
        n_st += s
        n_k += 1 if (p or s) else 0
    return bad, {"kernels_with_inline_asm": n_k, "inline_pk_add": n_pk, "nt_stores": n_st, "kernels": len(funcs)}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sdrx_amd", "lib", "libsdrxgpu.so")
    bad, stats = lint(lib, "--verbose" in sys.argv)
    for b in bad:
        print("HAZARD", b)
    print(f"hazard_lint: {stats['kernels']} kernels, {stats['kernels_with_inline_asm']} with inline assembly: {stats['inline_pk_add']} v_pk_add_f32 "
          f"behind MFMAs, {stats['nt_stores']} nt stores; {len(bad)} violation(s)")
    if stats["inline_pk_add"] == 0 or stats["nt_stores"] == 0:
        print("hazard_lint: found none of the instructions it is there to check -- the patterns no longer match the build")
        sys.exit(2)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
