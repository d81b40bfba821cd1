#!/usr/bin/env python
"""ISA lint for the hazard of DESIGN.md section 4c: a VGPR that a DS instruction in flight still reads (address or data) is
overwritten by one of the next few instructions.  hipcc schedules such writes freely; on gfx950 the write raced the operand
read of the second of two back-to-back ds_bpermute when the writer was v_accvgpr_read_b32 (window-attention backward, round 3).

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -c kernel.hip -o kernel.s;  python tools/isa_lint.py kernel.s ...

Reports (kernel, line, DS instruction, overwriting instruction).  --all: any VALU writer, not only v_accvgpr_read_b32."""
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.fullmatch(r'v(\d+)', tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def lint_file(path, any_writer=False, window=3):
    """Sites of one assembly file: [(file, line, kernel, ds instruction, overwriting instruction, its line)]."""
    sites = []
    kern = None
    lines = open(path).read().split('\n')
    code = []
    for ln, line in enumerate(lines, 1):
        t = line.strip()
        if t.endswith(':') and not t.startswith('.') and not t.startswith(';'):
            kern = t[:-1]
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
            continue
        code.append((ln, kern, t.split(';')[0].strip()))
    for i, (ln, k, ins) in enumerate(code):
        op = ins.split()[0]
        if not op.startswith('ds_'):
            continue
        ops = [o.strip() for o in ins[len(op):].split(',')]
        ops = [o.split()[0] for o in ops if o]
        if op.startswith('ds_read') or op.startswith('ds_load'):
            src = set().union(*[regs(o) for o in ops[1:2]])
        elif op.startswith('ds_bpermute') or op.startswith('ds_permute') or op.startswith('ds_swizzle'):
            src = set().union(*[regs(o) for o in ops[1:]])
        else:                                   # stores: address + data
            src = set().union(*[regs(o) for o in ops])
        if not src:
            continue
        for ln2, k2, ins2 in code[i + 1:i + 1 + window]:
            op2 = ins2.split()[0]
            if op2.startswith('s_') and op2 not in ('s_nop',):
                if op2.startswith('s_waitcnt') or op2.startswith('s_barrier') or op2.startswith('s_cbranch'):
                    break
                continue
            if not op2.startswith('v_'):
                continue
            if not any_writer and op2 != 'v_accvgpr_read_b32':
                continue
            if op2.startswith('v_cmp') or op2.startswith('v_mfma') or op2.startswith('v_accvgpr_write'):
                continue
            dst = regs(ins2[len(op2):].split(',')[0].strip().split()[0])
            if dst & src:
                sites.append((path.split('/')[-1], ln, (k or '')[:60], ins, ins2, ln2))
                break
    return sites


def kernel_resources(path):
    """{kernel symbol: {'vgpr', 'agpr', 'scratch', 'occupancy', 'spill_vgpr', 'spill_sgpr'}} from the '; Kernel info:' trailer
    behind every kernel and the .amdgpu_metadata block at the end of a hipcc -S file."""
    out, kern, meta = {}, None, None
    pat = {'vgpr': r'; NumVgprs: (\d+)', 'agpr': r'; NumAgprs: (\d+)', 'scratch': r'; ScratchSize: (\d+)',
           'occupancy': r'; Occupancy: (\d+)'}
    for line in open(path):
        t = line.strip()
        m = re.match(r'\.amdhsa_kernel (\S+)', t)
        if m:
            kern = m.group(1)
            out.setdefault(kern, {})
            continue
        m = re.match(r'\.name:\s+(\S+)', t)
        if m:
            meta = m.group(1)
            continue
        m = re.match(r'\.(vgpr|sgpr)_spill_count:\s+(\d+)', t)
        if m and meta is not None:
            out.setdefault(meta, {})['spill_' + m.group(1)] = int(m.group(2))
            continue
        if kern is None or not t.startswith(';'):
            continue
        for key, rx in pat.items():
            m = re.match(rx, t)
            if m:
                out[kern][key] = int(m.group(1))
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    any_writer = '--all' in sys.argv
    total = 0
    for path in args:
        for f, ln, k, ins, ins2, ln2 in lint_file(path, any_writer):
            total += 1
            print(f'{f}:{ln} [{k}] {ins}   <-   {ins2} (line {ln2})')
        if '--resources' in sys.argv:
            for k, r in kernel_resources(path).items():
                if r.get('spill_vgpr', 0) or r.get('scratch', 0):
                    print(f'{path.split("/")[-1]}: {k[:90]} spills: {r}')
    print(f'{total} site(s)')


if __name__ == '__main__':
    main()
