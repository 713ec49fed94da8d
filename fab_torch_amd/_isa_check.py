"""Static check of the GENERATED code of every kernel in libfabhip.so: no instruction may read or write a vector / accumulation
register that is the destination of a vector-memory load still in flight.

Why: the flow and spline kernels issue their weight loads from inline asm (`global_load_dwordx4` into VGPR / AGPR rings,
csrc/flow_device.h, stream_r8.h) and wait for them with hand-counted `s_waitcnt vmcnt(N)`.  hipcc does not know those loads
exist, so it may copy, spill or re-allocate such a register between the load and the wait that covers it - correct source,
wrong machine code, silently wrong samples.  That is a property of the compiled binary, so it is checked on the ISA: the
checker models the vmcnt queue (every VMEM load / store / atomic enters it in issue order, `s_waitcnt vmcnt(N)` retires all
but the newest N: gfx9 returns them in order) along EVERY control-flow path (basic blocks, both successors of a conditional
branch, loop back edges to a fixed point; the states of joining paths are merged conservatively) and flags an instruction that touches a register whose load may still
be in flight, and a VMEM load whose scalar base was written by v_readlane / v_readfirstlane fewer than 5 wait states earlier.
Compiler-tracked loads obey the same rule by construction, so every load is treated alike and every kernel is checked.
`_build.py` runs this after compiling and refuses to produce a library that fails it; tests/test_stream_kernel_isa.py and
tools/check_r8_isa.py run it too.  No GPU needed."""
import os
import re
import subprocess
import tempfile

LLVM = os.environ.get("FABHIP_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAX_STATES = 400000          # (block, queue state) pairs per kernel before the walk is declared inconclusive

_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
_VMEM = re.compile(r"^(global|buffer|flat|scratch|tbuffer)_(load|store|atomic)")
_VMCNT = re.compile(r"vmcnt\((\d+)\)")


def tools_available():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objdump", "llvm-objcopy", "clang-offload-bundler"))


def regs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.add(m.group(1) + m.group(2))
        else:
            out.update(m.group(3) + str(i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def disassemble_object(obj):
    """{kernel name: [(address, instruction text, branch target address or None)]} of the gfx950 code object inside `obj`."""
    tmp = tempfile.mkdtemp(prefix="fabisa")
    fat, elf = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.elf")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={elf}"])
    text = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", elf], text=True)
    return parse_objdump(text)


def parse_objdump(text):
    kernels, cur, start = {}, None, 0
    for raw in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", raw)
        if m:
            start = int(m.group(1), 16)
            cur = kernels.setdefault(m.group(2), [])
            continue
        if cur is None or "//" not in raw:
            continue
        ins, com = raw.split("//", 1)
        ins = ins.strip()
        ma = re.match(r"\s*([0-9A-Fa-f]+):", com)
        if not ins or not ma:
            continue
        tgt = None
        if ins.startswith("s_cbranch") or ins.startswith("s_branch"):
            mt = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", com)
            if mt:
                tgt = start + int(mt.group(1), 16)
            elif re.search(r"<[^>+]+>\s*$", com):
                tgt = start
        cur.append((int(ma.group(1), 16), ins, tgt))
    return kernels


def from_lines(lines):
    """Straight-line test input: a list of instruction strings -> the checker's instruction tuples."""
    return [(4 * i, l, None) for i, l in enumerate(lines)]


def _decode(texts, i):
    """(kind, dest, used, n, hazard) of instruction i: kind 'm' VMEM (enters the queue with `dest`), 'w' s_waitcnt vmcnt(n),
    'u' other instruction using vector registers `used`, '-' irrelevant."""
    ins = texts[i]
    mv = _VMEM.match(ins)
    if mv:
        ops = ins.split(None, 1)[1] if " " in ins else ""
        is_load = mv.group(2) == "load" or (mv.group(2) == "atomic" and " sc0" in ins)       # returning atomics write a VGPR
        lds_dma = "_lds_" in ins.split()[0] or re.search(r"\blds\b", ins)
        first = ops.split(",")[0] if ops else ""
        dest = frozenset(regs(first)) if (is_load and not lds_dma) else frozenset()
        used = frozenset(regs(ops) - dest)
        hazard = False                   # scalar base written by a lane read too recently (hardware hazard, 5 wait states)
        msb = re.search(r"s\[(\d+):(\d+)\]", ops)
        if msb:
            base = {"s" + msb.group(1), "s" + msb.group(2)}
            ws = 0
            for p in reversed(texts[max(0, i - 8):i]):
                if p.startswith("s_nop"):
                    ws += int(p.split()[1]) + 1
                    continue
                mm = re.match(r"v_read(?:first)?lane_b32 (s\d+)", p)
                if mm and mm.group(1) in base:
                    hazard = ws < 5
                    break
                if p.startswith("s_") and " " in p:      # a scalar-ALU write in between: the load reads THAT value, no hazard
                    d0 = p.split(None, 1)[1].split(",")[0].strip()
                    md = re.match(r"s\[(\d+):(\d+)\]$", d0)
                    base -= ({"s%d" % k for k in range(int(md.group(1)), int(md.group(2)) + 1)} if md else {d0})
                    if not base:
                        break
                ws += 1
                if ws >= 5:
                    break
        return ("m", dest, used, 0, hazard)
    if ins.startswith("s_waitcnt"):
        m = _VMCNT.search(ins)
        return ("w", None, None, int(m.group(1)), False) if m else ("-", None, None, 0, False)
    if not ins.startswith("s_"):
        used = frozenset(regs(ins))
        if used:
            return ("u", None, used, 0, False)
    return ("-", None, None, 0, False)


def _trim(q):
    while q and not q[0]:
        q = q[1:]
    if len(q) > 64:                      # vmcnt is a 6-bit counter: everything 63 or more loads back retires together
        q = (frozenset().union(*q[:len(q) - 63]),) + q[len(q) - 63:]
    return q


def _apply(dec, ins, q, bad, idx):
    kind, dest, used, n, hazard = dec
    if kind == "m":
        if q and used and not used.isdisjoint(frozenset().union(*q)):
            bad.append((idx, "touches a register with a load in flight", ins))
        if hazard:
            bad.append((idx, "scalar-base hazard", ins))
        return _trim(q + (dest,))
    if kind == "w":
        if n == 0:
            return ()
        return _trim(q[len(q) - n:]) if len(q) > n else q
    if kind == "u" and q:
        if not used.isdisjoint(frozenset().union(*q)):
            bad.append((idx, "touches a register with a load in flight", ins))
    return q


def _merge(q1, q2):
    """Conservative join of two in-flight queues (oldest first): entries are aligned from the NEWEST end, because
    `s_waitcnt vmcnt(N)` keeps the N newest whatever the path; a register is in flight at depth k if it is on either path."""
    if q1 == q2:
        return q1
    n = max(len(q1), len(q2))
    a = (frozenset(),) * (n - len(q1)) + q1
    b = (frozenset(),) * (n - len(q2)) + q2
    return _trim(tuple(x | y for x, y in zip(a, b)))


def check_kernel(name, insns):
    """List of (instruction index, what, text) findings; `insns` = parse_objdump()'s tuples or from_lines().
    Forward data-flow over the basic blocks to a fixed point (block entry state = _merge of the predecessors' exit states:
    loops are followed until nothing changes), then one reporting pass."""
    if insns and isinstance(insns[0], str):
        insns = from_lines(insns)
    addr_to_idx = {a: i for i, (a, _, _) in enumerate(insns)}
    leaders = {0}
    for i, (a, ins, tgt) in enumerate(insns):
        if ins.startswith("s_cbranch") or ins.startswith("s_branch") or ins.startswith("s_endpgm"):
            if i + 1 < len(insns):
                leaders.add(i + 1)
            if tgt is not None and tgt in addr_to_idx:
                leaders.add(addr_to_idx[tgt])
    leaders = sorted(leaders)
    block_end = {b: (leaders[k + 1] if k + 1 < len(leaders) else len(insns)) for k, b in enumerate(leaders)}
    texts = [x[1] for x in insns]
    dec = [_decode(texts, i) for i in range(len(insns))]

    def run_block(b, q, found):
        end = block_end[b]
        succ = [end] if end < len(insns) else []
        for i in range(b, end):
            a, ins, tgt = insns[i]
            q = _apply(dec[i], ins, q, found, i)
            if ins.startswith("s_endpgm"):
                succ = []
            elif ins.startswith("s_branch"):
                succ = [addr_to_idx[tgt]] if tgt in addr_to_idx else []
            elif ins.startswith("s_cbranch"):
                succ = ([end] if end < len(insns) else []) + ([addr_to_idx[tgt]] if tgt in addr_to_idx else [])
        return q, succ

    state = {0: ()}
    work = [0]
    steps = 0
    while work:
        b = work.pop()
        steps += 1
        if steps > MAX_STATES:
            return [(b, "inconclusive: no fixed point", name)]
        q, succ = run_block(b, state[b], [])
        for s_ in succ:
            new = q if s_ not in state else _merge(state[s_], q)
            if s_ not in state or new != state[s_]:
                state[s_] = new
                work.append(s_)
    bad = []
    for b in sorted(state):
        run_block(b, state[b], bad)
    return sorted(set(bad))


def check_adapt_fold(name, insns):
    """The in-kernel step-size rule of the 4- / 8-chain transition kernels (csrc/ais_kernels.hip: hmc_store_row_stats /
    hmc_adapt_last) orders its cross-workgroup traffic by HARDWARE behaviour, not by the HIP memory model: the
    per-chain values are written with device-scope relaxed atomic stores, which gfx950 lowers to write-through stores (sc1:
    visible in memory when vmcnt retires them), the wave waits vmcnt(0) and draws a ticket (returning atomic add), and the wave
    with the last ticket reads the values past the non-coherent L2 (sc1 loads).  A compiler that lowers those accesses
    differently would silently break the rule, so the lowering itself is checked on every build: in such a kernel the ticket
    must directly follow an `s_waitcnt vmcnt(0)` with no vector-memory instruction in between, at least two `sc1` dword stores
    must precede it, and at least 32 `sc1` dword loads (16 rows x 2 statistics) must follow it."""
    if not re.search(r"k_hmc_step_r[48]|k_spline_logprob_r8", name):      # (spline_r8.h: s8_fold_adapt_last, the same mechanism)
        return []
    texts = [x[1] for x in insns]
    tickets = [i for i, t in enumerate(texts) if t.startswith("global_atomic_add") and " sc0" in t]
    if not tickets:
        return []                                   # (an instantiation without the folded rule)
    bad = []
    for i in tickets:
        j = i - 1
        while j >= 0 and not (texts[j].startswith("s_waitcnt") and "vmcnt(0)" in texts[j]) and not _VMEM.match(texts[j]) and i - j < 48:
            j -= 1                                  # (weaker waits hipcc adds at joins in between are no-ops on an empty queue)
        if j < 0 or not (texts[j].startswith("s_waitcnt") and "vmcnt(0)" in texts[j]):
            bad.append((i, "ticket not directly behind s_waitcnt vmcnt(0)", texts[i]))
    n_st = sum(1 for t in texts[:tickets[-1]] if t.startswith("global_store_dword ") and " sc1" in t)
    n_ld = sum(1 for t in texts[tickets[0]:] if t.startswith("global_load_dword ") and " sc1" in t)
    if n_st < 2:
        bad.append((tickets[0], f"only {n_st} write-through (sc1) statistic stores in front of the ticket", texts[tickets[0]]))
    if n_ld < 32:
        bad.append((tickets[0], f"only {n_ld} sc1 statistic loads behind the ticket", texts[tickets[0]]))
    return bad


def check_bf16_mfma_overlap(name, insns):
    """No bf16 MFMA may write a destination that overlaps its A or B operand.  With the accumulators in architectural registers
    (_build.py: -amdgpu-mfma-vgpr-form) hipcc hands an operand that dies in the MFMA to it as the destination, and gfx950 then
    returns a wrong first row of every 4-row group for v_mfma_f32_16x16x32_bf16 / v_mfma_f32_4x4x4_16b_bf16 (found round 6: two
    fast-mode tests; round 5 met it with a ring slot as the destination).  flow_device.h / flow_r4f.h keep the operands alive until
    the result exists; this check refuses a build in which the allocator got around that."""
    bad = []

    def rng(tok):
        m = re.match(r"([va])\[(\d+):(\d+)\]$", tok) or re.match(r"([va])(\d+)()$", tok)
        if not m:
            return None
        lo = int(m.group(2))
        return m.group(1), lo, int(m.group(3)) if m.group(3) else lo

    for i, (_, text, _) in enumerate(insns):
        if not text.startswith("v_mfma") or "bf16" not in text:
            continue
        ops = [o.strip() for o in text.split(None, 1)[1].split(",")]
        if len(ops) < 3:
            continue
        d = rng(ops[0])
        for which, tok in (("A", ops[1]), ("B", ops[2])):
            r = rng(tok)
            if d and r and d[0] == r[0] and not (r[2] < d[1] or r[1] > d[2]):
                bad.append((i, f"bf16 MFMA destination overlaps its {which} operand", text))
    return bad


def check_object(obj, patterns=None, verbose=False):
    """{kernel: findings} for the kernels of `obj` whose name contains one of `patterns` (all when None)."""
    out = {}
    for name, insns in sorted(disassemble_object(obj).items()):
        if patterns and not any(p in name for p in patterns):
            continue
        if not insns:
            continue
        bad = check_kernel(name, insns) + check_adapt_fold(name, insns) + check_bf16_mfma_overlap(name, insns)
        out[name] = bad
        if verbose:
            n_ld = sum(1 for _, l, _ in insns if l.startswith("global_load_dwordx4"))
            print(f"{name[:110]}: {len(insns)} instructions, {n_ld} dwordx4 loads, {len(bad)} findings")
            for f in bad[:12]:
                print("   ", f)
    return out


if __name__ == "__main__":                # python -m fab_torch_amd._isa_check <object> ...: exit code 1 on findings
    import sys
    rc = 0
    for obj in sys.argv[1:]:
        for k, v in check_object(obj).items():
            if v:
                rc = 1
                print(f"{os.path.basename(obj)}: {k}: {v[:6]}")
    sys.exit(rc)
