"""Scan the code objects of the built library for signs of a mis-structured kernel: the count of EXEC-masked
loops (s_cbranch_execnz) per kernel.  The decode kernels' control flow is wave-uniform by construction (scalar
branches on the quarter a wave works on, on the verdict, on the clamp mode); when the compiler fails to see that,
or turns a per-lane select chain into branches, it emits EXEC-masked regions and loops -- correct, and 1.7x
(TM1280, round 2) to 100x (pair kernel on TM6144) slower.  A handful are legitimate (the sub-wave codeword
handling of the TC codes, the guarded epilogue stores).
    python tools/scan_kernels.py            # table of every kernel in build/csrc/decode_ms_*.o"""
import collections, glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def scan(pattern="build/csrc/decode_ms_*.o"):
    """{(object, kernel): (instructions, exec loops)}"""
    out = {}
    for obj in sorted(glob.glob(os.path.join(ROOT, pattern))):
        tmp = tempfile.mkdtemp()
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={tmp}/fat", obj, "/dev/null"])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={tmp}/fat", f"--output={tmp}/co", "--unbundle"])
        dis = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", f"{tmp}/co"], text=True).split("\n")
        cur = None
        for l in dis:
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
            if m:
                cur = (os.path.basename(obj), m.group(1))
                out[cur] = [0, 0]
                continue
            if cur and "//" in l:
                out[cur][0] += 1
                if l.strip().startswith("s_cbranch_execnz"):
                    out[cur][1] += 1
    return out


if __name__ == "__main__":
    for (obj, k), (n, loops) in scan().items():
        print(f"{obj:24s} {k[:64]:64s} {n:6d} instructions  {loops:4d} exec loops{'   <== look at this one' if loops > 16 else ''}")
