"""Static check of k_compat_split's machine code (hipcc -S, no GPU needed).

The kernel loads the next tile's E_0 straight into the accumulators with inline-assembly loads the compiler cannot see
(phl_meanfield.hip: PHL_E0_LOAD_HIDDEN) and waits for them itself.  That is only sound while the compiler does not touch
those registers between the load and the wait -- no copy, no spill, no move to AGPRs.  This script compiles the file to
assembly and verifies, for every instance of the kernel:
  * no scratch, no AGPRs, at most 256 VGPRs;
  * every instruction that names an accumulator register and is neither a matrix instruction nor a global load / store
    sits in the prologue (before the first barrier) or in a matrix slot (between a v_mfma and the slot's barrier: the softmax).
Exit code 0 = ok.  Used by tests/test_cabi_and_host.py."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check(asm_text):
    problems, seen = [], 0
    for m in re.finditer(r'^(_ZN\S*k_compat_split\S*):[^\n]*\n(.*?)\.end_amdhsa_kernel', asm_text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        seen += 1
        lines = body.split('\n')
        acc = set()
        for l in lines:
            h = re.search(r'global_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[', l)
            if h:
                acc.update(range(int(h.group(1)), int(h.group(2)) + 1))
        if len(acc) != 128:
            problems.append(f'{name}: {len(acc)} accumulator registers found through the hidden loads, expected 128')
            continue
        barriers, in_matrix = 0, False
        for i, l in enumerate(lines):
            t = l.strip()
            if not t or t[0] in ';.':
                continue
            op = t.split()[0]
            if op == 's_barrier':
                barriers, in_matrix = barriers + 1, False
                continue
            if op.startswith('v_mfma'):
                in_matrix = True
                continue
            if op.startswith('global_load') or op.startswith('global_store'):
                continue
            regs = set()
            for r in re.finditer(r'v\[(\d+):(\d+)\]', t):
                regs.update(range(int(r.group(1)), int(r.group(2)) + 1))
            for r in re.finditer(r'\bv(\d+)\b', t):
                regs.add(int(r.group(1)))
            if regs & acc and barriers > 0 and not in_matrix:
                problems.append(f'{name}: line {i}: `{t}` touches an accumulator outside a matrix slot')
        meta = asm_text[m.end():m.end() + 6000]
        for key, limit in (('ScratchSize', 0), ('NumAgprs', 0), ('NumVgprs', 256)):
            v = re.search(r'; %s: (\d+)' % key, meta)
            if not v or int(v.group(1)) > limit:
                problems.append(f'{name}: {key} = {v.group(1) if v else "?"} (limit {limit})')
    if seen != 4:
        problems.append(f'{seen} instances of k_compat_split found, expected 4')
    return problems


def main():
    src = os.path.join(ROOT, 'depth-estimation_amd', 'csrc', 'phl_meanfield.hip')
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'mf.s')
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
               '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.dirname(src), '-S', '--cuda-device-only', src, '-o', out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        problems = check(open(out).read())
    for p in problems:
        print(p)
    print('k_compat_split machine code:', 'FAILED' if problems else 'ok')
    return 1 if problems else 0


if __name__ == '__main__':
    sys.exit(main())
