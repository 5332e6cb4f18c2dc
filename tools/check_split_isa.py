"""Static check of k_compat_split's machine code (hipcc -S, no GPU needed).

The kernel loads the next tile's E_0 straight into the accumulators with inline-assembly loads the compiler cannot see
(phl_meanfield.hip: PHL_E0_LOAD_HIDDEN) and waits for them itself.  That is only sound while the compiler does not touch
those registers between the load and the wait -- no copy, no spill, no move to AGPRs.  This script compiles the file to
assembly and verifies, for every instance of the kernel:
  * no scratch, no AGPRs, at most 256 VGPRs;
  * in listing order, no instruction names an accumulator register between the hidden load into it and the kernel's own
    counted wait for those loads (marked CSP_E0_LANDED in the assembly); the loads are unconditional, so they sit in the
    loop's straight-line code -- if block placement ever moves one behind the last marker the script says so.
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
        def names(t):
            regs = set()
            for r in re.finditer(r'v\[(\d+):(\d+)\]', t):
                regs.update(range(int(r.group(1)), int(r.group(2)) + 1))
            for r in re.finditer(r'\bv(\d+)\b', t):
                regs.add(int(r.group(1)))
            return regs

        def walk(lo, hi, pending):           # listing order; returns the accumulators still in flight at `hi`
            for i in range(lo, hi):
                t = lines[i].strip()
                if 'CSP_E0_LANDED' in t:     # the kernel's own counted wait for those loads
                    pending = set()
                    continue
                if not t or t[0] in ';.':
                    continue
                h = re.search(r'global_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[', t)
                if h:
                    pending = pending | set(range(int(h.group(1)), int(h.group(2)) + 1))
                    continue
                if names(t) & pending:
                    problems.append(f'{name}: line {i}: `{t}` names an accumulator whose hidden load may still be in flight')
            return pending

        head = [i for i, l in enumerate(lines) if 'Loop Header: Depth=1' in l]
        mark = [i for i, l in enumerate(lines) if 'CSP_E0_LANDED' in l]
        if len(head) != 1 or len(mark) != 1 or mark[0] < head[0]:
            problems.append(f'{name}: expected one tile loop with one CSP_E0_LANDED marker inside (found {len(head)} / {len(mark)})')
            continue
        left = walk(0, len(lines), set())
        # the loads sit in the epilogue half, the wait at the top of the next iteration: around the back edge
        if walk(head[0], mark[0] + 1, left):
            problems.append(f'{name}: hidden loads not covered by the marker at the top of the loop')
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
