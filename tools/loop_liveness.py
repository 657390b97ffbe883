"""VGPR liveness through the main loop of a kernel (the longest backward branch's body): live-in count and the number of simultaneously
live registers per 100-instruction window -- where a kernel's register pressure peaks, independent of what the allocator reports.
    python tools/loop_liveness.py <object.o> [kernel-name-substring]"""
import re, subprocess, sys, tempfile
L = '/opt/rocm/lib/llvm/bin'


def disasm(obj):
    T = tempfile.mkdtemp()
    subprocess.check_call([f'{L}/llvm-objcopy', '--dump-section', f'.hip_fatbin={T}/fat', obj, '/dev/null'])
    subprocess.check_call([f'{L}/clang-offload-bundler', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--input={T}/fat', f'--output={T}/co', '--unbundle'])
    return subprocess.check_output([f'{L}/llvm-objdump', '-d', f'{T}/co'], text=True)


def regs(tok):
    out = []
    for a, b in re.findall(r'v\[(\d+):(\d+)\]', tok):
        out += list(range(int(a), int(b) + 1))
    tok = re.sub(r'[av]\[\d+:\d+\]', '', tok)
    out += [int(x) for x in re.findall(r'\bv(\d+)\b', tok)]
    return out


NO_DST = ('ds_write', 'scratch_store', 'global_store', 'buffer_store', 's_', 'v_cmp', 'v_cmpx', 'ds_bpermute_b32x')


def analyse(text, want=''):
    for fn in re.split(r'\n(?=[0-9a-f]{16} <)', text):
        head = fn.split('\n', 1)[0]
        if want not in head or '>:' not in head:
            continue
        lines = fn.split('\n')
        start = int(head.split()[0], 16)
        addr = {}
        for i, l in enumerate(lines):
            m = re.search(r'//\s*([0-9A-F]{12}):', l)
            if m:
                addr[int(m.group(1), 16)] = i
        best = None
        for i, l in enumerate(lines):
            m = re.search(r'(s_cbranch\w*|s_branch)\s+\d+\s+//\s*([0-9A-F]{12}):.*\+0x([0-9a-f]+)>', l)
            if m:
                t = addr.get(start + int(m.group(3), 16))
                if t is not None and t < i and (best is None or i - t < best[1] - best[0]) and i - t > 500:
                    best = (t, i)
        if best is None:
            continue
        body = []
        for l in lines[best[0]:best[1]]:
            code = l.split('//')[0].strip()
            m = re.match(r'(\S+)\s+(.*)', code)
            if not m:
                continue
            op, parts = m.group(1), [a.strip() for a in m.group(2).split(',')]
            dst, src = ([], parts) if op.startswith(NO_DST) else (parts[:1], parts[1:])
            body.append((op, set(r for t in dst for r in regs(t)), set(r for t in src for r in regs(t))))
        live = set()
        for _ in range(2):                      # twice around the loop
            prof = []
            for op, d, s in reversed(body):
                live = (live - d) | s
                prof.append(len(live))
        prof.reverse()
        print(head.split('<')[1][:60], ': loop of', len(body), 'instructions, live-in', len(live), 'peak', max(prof))
        print('   live per 100:', [max(prof[i:i + 100]) for i in range(0, len(prof), 100)])


if __name__ == '__main__':
    analyse(disasm(sys.argv[1]), sys.argv[2] if len(sys.argv) > 2 else '')
