"""Instruction mix of the TM8192 f32 pair kernel's iteration loop, from the library's own code object, and the
cycle-weighted VALU issue cost it implies (tools/ubench/valu_rate.hip gives the per-class issue cost at four waves
per SIMD: 2 cycles for plain VOP1/VOP2 ALU operations, 4 for min/max/compare and the VOP3-only forms, ~3 for
v_bitop3_b32).  Writes profiles/<round>/valu_mix_tm8192_f32.json.
    python tools/valu_mix.py r03_final"""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
name = sys.argv[1] if len(sys.argv) > 1 else "r03_final"
tmp = tempfile.mkdtemp()
subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={tmp}/fat", os.path.join(ROOT, "build/csrc/decode_ms_f32_part_1.o"), "/dev/null"])
subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={tmp}/fat", f"--output={tmp}/co", "--unbundle"])
dis = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", f"{tmp}/co"], text=True).split("\n")
cur, body = None, []          # (address, text, branch target or None)
for l in dis:
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
    if m:
        cur = m.group(1)
        continue
    if cur and "decode_ms_pair_kernelILi8EfLi2E" in cur and "//" in l:      # the form-2 instantiation (the one a 25-iteration launch runs)
        text, tail = l.split("//", 1)
        addr = int(tail.split(":")[0].strip(), 16)
        tgt = re.search(r"<[^>]*\+0x([0-9a-f]+)>", tail)
        body.append((addr, text.strip(), int(tgt.group(1), 16) if tgt else None))
base = body[0][0]
# the iteration loops are the backward branches whose span holds exactly two workgroup barriers; one per quarter
# body and clamp mode.  Take a clamp-FREE one (no v_min_f32 cap operations, clamp-form self-correction = v_fmac + v_med3: the mode the
# benchmark's frames run in).
loops = []
for i, (addr, text, tgt) in enumerate(body):
    if text.startswith("s_cbranch") or text.startswith("s_branch"):
        if tgt is not None and base + tgt < addr:
            j = next(k for k, b in enumerate(body) if b[0] == base + tgt)
            span = [t for _, t, _ in body[j:i + 1]]
            if sum(1 for t in span if t.startswith("s_barrier")) == 2:
                loops.append(span)
loops.sort(key=lambda sp: (sum(1 for t in sp if t.startswith("v_min_f32")), -sum(1 for t in sp if t.startswith("v_fmac_f32")), len(sp)))
# among the clamp-free copies (as many v_min_f32 / v_mul_f32 as the first) take the one with the LARGEST VALU count: since
# iteration 0 is peeled the compiler rotates some copies of the loop, and a backward branch can then span a body that
# lacks the rotated part
key = lambda sp: (sum(1 for t in sp if t.startswith("v_min_f32")), sum(1 for t in sp if t.startswith("v_fmac_f32")))
same = [sp for sp in loops if key(sp) == key(loops[0])]
quarter = collections.Counter(sum(1 for t in sp if t.startswith("ds_")) for sp in same).most_common(1)[0][0]
loop = max((sp for sp in same if sum(1 for t in sp if t.startswith("ds_")) == quarter), key=lambda sp: sum(1 for t in sp if t.startswith("v_")))
FOUR = re.compile(r"^v_(min3|max3|med3|min_|max_|cmp|cmpx|cndmask_b32_e64|bfi|and_or|or3|add3|perm|alignbit|mad|pk_|lshl_add|lshl_or|xad)")      # (v_fma_f32 / v_fmac_f32 / v_mul_legacy_f32 issue at the full rate on gfx950: profiles/r03_final/valu_rate.txt)
cls = collections.Counter()
ops = collections.Counter()
for l in loop:
    if not l.startswith("v_"):
        continue
    op = l.split()[0]
    ops[op] += 1
    if op.startswith("v_bitop3"):
        cls["bitop3 (3 cycles)"] += 1
    elif FOUR.match(op):
        cls["4-cycle (min/max/med3/compare/VOP3-only)"] += 1
    else:
        cls["2-cycle (plain VOP1/VOP2)"] += 1
n = sum(cls.values())
avg = (2 * cls["2-cycle (plain VOP1/VOP2)"] + 3 * cls["bitop3 (3 cycles)"] + 4 * cls["4-cycle (min/max/med3/compare/VOP3-only)"]) / n
out = {"kernel": "decode_ms_pair_kernel<8, float, 2>, one quarter body, clamp-free mode, one iteration (variable + check phase)",
       "valu_instructions_per_wave_iteration": n, "classes": dict(cls), "avg_issue_cycles_per_instruction": avg,
       "other": {"ds": sum(1 for l in loop if l.startswith("ds_")), "salu": sum(1 for l in loop if l.startswith("s_"))},
       "top_opcodes": ops.most_common(14),
       "iteration_loops_found": len(loops),
       "note": "the loop = a backward branch spanning exactly two workgroup barriers; the clamp-free copy of one quarter body; issue costs per class from tools/ubench (valu_rate / valu_pairs, 4 waves per SIMD)"}
os.makedirs(os.path.join(ROOT, "profiles", name), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", name, "valu_mix_tm8192_f32.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
