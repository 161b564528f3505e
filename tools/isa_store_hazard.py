"""Scan the gfx950 code of libi2r_hip.so for wide vector-memory stores whose data registers are overwritten too soon.

Background ("store-data hazard"): a buffer_store_dwordx3/x4 reads its data VGPRs a little after it issues.  LLVM's hazard
recognizer keeps one wait state between such a store and a VALU write of its data registers ONLY when the store has no SGPR soffset
(the rule inherited from older GCN parts); with an SGPR soffset it allows the very next instruction to overwrite them.  On MI355X
that is not safe when other waves share the SIMD: conv1x1_pair_k<8,4,2> stored 16 lanes of the NEXT fragment's value whenever a
second program ran beside it (tools/race_bisect.py).  The kernels therefore store through buf_st16() (csrc/i2r_common.h), which
keeps the data registers alive for two more wait states, and this scan proves that no wide store is left with fewer.

usage: isa_store_hazard.py [lib.so] [min wait states, default 2]   -> exit code 1 if a store has fewer
(CPU only: objcopy + llvm-objdump from /opt/rocm)"""
import os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "intra-and-inter-human-relation-network-for-mpee_amd", "libi2r_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

WIDE_STORE = re.compile(r"^(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\b")


def code_objects(lib, tmp):
    """every gfx950 code object of the library's .hip_fatbin section (one clang offload bundle per translation unit) -> file paths"""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    d = open(fat, "rb").read()
    out = []
    for m in re.finditer(MAGIC, d):
        base = m.start()
        n, = struct.unpack_from("<Q", d, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", d, p)
            triple = d[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                path = os.path.join(tmp, "co_%d.o" % len(out))
                open(path, "wb").write(d[base + off:base + off + size])
                out.append(path)
    return out


def regs(tok):
    """'v[18:21]' / 'v5' -> set of VGPR numbers; anything else -> empty"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def dst_regs(mnem, ops):
    """VGPRs the instruction writes right away (VALU / MFMA results; loads land far later and are counted by vmcnt)"""
    if not mnem.startswith("v_") or not ops:
        return set()
    if mnem.startswith("v_cmp") or mnem.startswith("v_cmpx"):
        return set()
    out = regs(ops[0])
    if mnem.startswith(("v_swap", "v_permlane")) and len(ops) > 1:
        out |= regs(ops[1])
    return out


def wait_states(mnem, ops):
    if mnem == "s_nop":
        return int(ops[0], 0) + 1
    return 1


def scan(path, need):
    txt = subprocess.check_output([OBJDUMP, "-d", "--no-show-raw-insn", path], text=True)
    kernel, insts, bad, n_stores = None, [], [], 0
    per_kernel = {}
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            kernel = m.group(1)
            per_kernel[kernel] = []
            continue
        if kernel is None or not line.startswith("\t"):
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        parts = body.split(None, 1)
        mnem = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        per_kernel[kernel].append((mnem, ops, body))
    for kernel, insts in per_kernel.items():
        for i, (mnem, ops, body) in enumerate(insts):
            if not WIDE_STORE.match(mnem):
                continue
            n_stores += 1
            # data operand: buffer_store vdata, vaddr, srsrc, soffset ...; global_store vaddr, vdata, saddr; flat_store vaddr, vdata
            data = regs(ops[0]) if mnem.startswith("buffer") else regs(ops[1])
            ws, j = 0, i + 1
            while j < len(insts) and ws < need:
                m2, o2, b2 = insts[j]
                if m2.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
                    break  # (a taken branch costs more than the wait states in question; the fall-through is scanned from its own position)
                if dst_regs(m2, o2) & data:
                    bad.append((kernel, body, b2, ws))
                    break
                ws += wait_states(m2, o2)
                j += 1
    return n_stores, bad


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else LIB
    need = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    with tempfile.TemporaryDirectory() as tmp:
        cos = code_objects(lib, tmp)
        total, bad = 0, []
        for co in cos:
            n, b = scan(co, need)
            total += n
            bad += b
    print("%d code objects, %d wide stores, %d with a data register overwritten within %d wait states" % (len(cos), total, len(bad), need))
    for kernel, st, wr, ws in bad:
        print("  %s\n      %s\n      %s   (after %d wait states)" % (kernel, st, wr, ws))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
