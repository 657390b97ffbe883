"""Opcode histogram of the iteration loops of a decode kernel, from any object or executable holding a gfx950
code object (build/csrc/*.o, a tools/kbench.hip binary).  A loop = a backward branch whose span holds exactly
two workgroup barriers (variable phase + check phase).
    python tools/loop_mix.py build/kb/tm5120_i8_old decode_ms_kernel [edges per thread]"""
import collections, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"


def loops_of(obj, kernel):
    return {k: [sp for _, _, sp in v] for k, v in spans_of(obj, kernel).items()}


def spans_of(obj, kernel):
    """{kernel: [(first, last, instructions)]}: positions in the kernel's instruction list, so that nested spans can be told apart
    (a span that holds another one is not an iteration loop but a path around it: e.g. the jump from a finished codeword's
    epilogue back into the loop for the next codeword of a one-wave kernel)"""
    tmp = tempfile.mkdtemp()
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={tmp}/fat", obj, "/dev/null"])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={tmp}/fat", f"--output={tmp}/co", "--unbundle"])
    dis = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", f"{tmp}/co"], text=True).split("\n")
    out, cur, body = {}, None, []
    for l in dis + ["0 <end>:"]:
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
        if m:
            if cur and kernel in cur and body:
                out[cur] = body
            cur, body = m.group(1), []
            continue
        if cur and "//" in l:
            text, tail = l.split("//", 1)
            addr = int(tail.split(":")[0].strip(), 16)
            tgt = re.search(r"<[^>]*\+0x([0-9a-f]+)>", tail)
            body.append((addr, text.strip(), int(tgt.group(1), 16) if tgt else None))
    res = {}
    for k, body in out.items():
        base, loops = body[0][0], []
        index = {b[0]: i for i, b in enumerate(body)}
        for i, (addr, text, tgt) in enumerate(body):
            if (text.startswith("s_cbranch") or text.startswith("s_branch")) and tgt is not None and base + tgt < addr:
                j = index.get(base + tgt)
                if j is None:
                    continue
                span = [t for _, t, _ in body[j:i + 1]]
                if sum(1 for t in span if t.startswith("s_barrier")) == 2:
                    loops.append((j, i, span))
        res[k] = loops
    return res


if __name__ == "__main__":
    obj, kernel = sys.argv[1], sys.argv[2]
    edges = float(sys.argv[3]) if len(sys.argv) > 3 else None
    for k, loops in loops_of(obj, kernel).items():
        print(k[:100], len(loops), "loops")
        for sp in loops:
            ops = collections.Counter(l.split()[0] for l in sp)
            valu = sum(c for o, c in ops.items() if o.startswith("v_"))
            print(f"  loop: {len(sp)} instructions, {valu} VALU" + (f" ({valu / edges:.2f} per edge)" if edges else "")
                  + f", {sum(c for o, c in ops.items() if o.startswith('ds_'))} ds, {sum(c for o, c in ops.items() if o.startswith('s_'))} salu")
            print("    " + "  ".join(f"{o}:{c}" for o, c in ops.most_common(40)))
