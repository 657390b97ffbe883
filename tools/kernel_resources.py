"""VGPRs, spilled VGPRs, SGPRs, LDS bytes and scratch of every kernel in build/csrc/*.o (from the code objects' metadata notes).
    python tools/kernel_resources.py [substring ...]      # only kernels whose demangled name contains every substring"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def resources(pattern="build/csrc/*.o"):
    rows = []
    for obj in sorted(glob.glob(os.path.join(ROOT, pattern))):
        tmp = tempfile.mkdtemp()
        r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={tmp}/fat", obj, "/dev/null"], capture_output=True)
        if r.returncode != 0:
            continue
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={tmp}/fat", f"--output={tmp}/co", "--unbundle"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", f"{tmp}/co"], text=True)
        for blk in notes.split("  - .agpr_count:")[1:]:
            get = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
            name = get("name")
            dem = subprocess.check_output(["c++filt", name], text=True).strip()
            rows.append((os.path.basename(obj), dem, get("vgpr_count"), get("vgpr_spill_count"), get("sgpr_count"),
                         get("group_segment_fixed_size"), get("private_segment_fixed_size")))
    return rows


if __name__ == "__main__":
    want = sys.argv[1:]
    for obj, dem, v, sp, s, lds, scr in resources():
        short = re.sub(r"^void ldpc::", "", dem).split("(")[0]
        if all(w in short for w in want):
            print(f"{obj:26s} {short:64s} vgpr {v:>4s} spill {sp:>4s} sgpr {s:>4s} lds {lds:>7s} scratch {scr:>5s}")
